! nested_hip.F90 -- the reference's `program nested` (nested_loops/nested.F90: namelist sizes -> arrays initialised
! with representative connectivity and topography -> data to the device once -> nIters timed iterations of the
! loop nest -> result checked to errTol) with libnlk_hip.so (include/nlk_hip.h) in the place of its OpenACC /
! OpenMP-offload / YAKL / Kokkos forms.
!
!   ./nested_hip [namelist-file [variant [dumpfile]]]
!     namelist  &nested_nml nIters, nEdges, nCells, nVertLevels, nAdv /   (the reference's nested.nml, :46-54 there;
!               "-" or absent: its shipped values 100, 25600, 2800, 100, 10)
!     variant   0 EXACT (bit-identical to the reference loop), 1 FAST
!     dumpfile  every array the program built + the result, stream access: the arrays are drawn with the compiler's
!               random_number like the reference's (:65-103), so a checker needs them to reproduce the result
! Checks it makes itself (the program has no CPU loop: the product has no CPU path): the host-array call against
! the device-resident call (two entry points, one kernel), and EXACT against FAST to the reference's errTol = 1e-10
! (nested_vars.F90:36) on the scale of the fluxes, printed in the reference's words when it fails.
module nlk_hip_mod
  use iso_c_binding
  implicit none
  integer, parameter :: RKIND = c_double
  interface
    integer(c_int) function nlk_high_order_flux(nEdges, nCells, nVertLevels, nvldim, nAdv, nAdvCellsForEdge, &
        advCellsForEdge, minLevelCell, maxLevelCell, tracerCur, normalThicknessFlux, advMaskHighOrder, &
        advCoefs, advCoefs3rd, coef3rdOrder, highOrderFlx) bind(C, name="nlk_high_order_flux")
      import :: c_int, c_double
      integer(c_int), value :: nEdges, nCells, nVertLevels, nvldim, nAdv
      integer(c_int), intent(in) :: nAdvCellsForEdge(*), advCellsForEdge(*), minLevelCell(*), maxLevelCell(*)
      real(c_double), intent(in) :: tracerCur(*), normalThicknessFlux(*), advMaskHighOrder(*), advCoefs(*), advCoefs3rd(*)
      real(c_double), value :: coef3rdOrder
      real(c_double) :: highOrderFlx(*)
    end function
    ! the data-resident form: device pointers, asynchronous on a stream
    integer(c_int) function nlk_high_order_flux_device(nEdges, nCells, nVertLevels, nvldim, nAdv, nAdvCellsForEdge, &
        advCellsForEdge, minLevelCell, maxLevelCell, tracerCur, normalThicknessFlux, advMaskHighOrder, &
        advCoefs, advCoefs3rd, coef3rdOrder, highOrderFlx, stream) bind(C, name="nlk_high_order_flux_device")
      import :: c_int, c_double, c_ptr
      integer(c_int), value :: nEdges, nCells, nVertLevels, nvldim, nAdv
      type(c_ptr), value :: nAdvCellsForEdge, advCellsForEdge, minLevelCell, maxLevelCell
      type(c_ptr), value :: tracerCur, normalThicknessFlux, advMaskHighOrder, advCoefs, advCoefs3rd
      real(c_double), value :: coef3rdOrder
      type(c_ptr), value :: highOrderFlx, stream
    end function
    integer(c_int) function nlk_set_variant(v) bind(C, name="nlk_set_variant")
      import :: c_int
      integer(c_int), value :: v
    end function
    ! the HIP runtime calls that stand where the reference has `!$acc enter data copyin` / `update host` (:170-175)
    integer(c_int) function hipMalloc(p, bytes) bind(C, name="hipMalloc")
      import :: c_int, c_ptr, c_size_t
      type(c_ptr) :: p
      integer(c_size_t), value :: bytes
    end function
    integer(c_int) function hipFree(p) bind(C, name="hipFree")
      import :: c_int, c_ptr
      type(c_ptr), value :: p
    end function
    integer(c_int) function hipMemcpy(dst, src, bytes, kind) bind(C, name="hipMemcpy")
      import :: c_int, c_ptr, c_size_t
      type(c_ptr), value :: dst, src
      integer(c_size_t), value :: bytes
      integer(c_int), value :: kind      ! 1 host -> device, 2 device -> host
    end function
    integer(c_int) function hipDeviceSynchronize() bind(C, name="hipDeviceSynchronize")
      import :: c_int
    end function
  end interface
contains
  subroutine must(rc, what)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: what
    if (rc /= 0) then
      write(*,*) what, ' failed with code ', rc
      error stop 1
    end if
  end subroutine must
end module nlk_hip_mod

program nested_hip
  use nlk_hip_mod
  implicit none
  integer :: nIters = 100, nEdges = 25600, nCells = 2800, nVertLevels = 100, nAdv = 10
  namelist /nested_nml/ nIters, nEdges, nCells, nVertLevels, nAdv
  real(RKIND), parameter :: errTol = 1.e-10_RKIND            ! nested_vars.F90:36
  real(RKIND) :: coef3rdOrder, randNum, relErr, worst, worstScaled, floorVal
  integer :: nvldim, variant, iCell, iEdge, i, k, n, ios, nbad, nrel
  integer(8) :: t1, t2, rt
  character(len=512) :: nmlfile, arg, dumpfile
  integer(c_int), allocatable, target :: nAdvCellsForEdge(:), advCellsForEdge(:,:), minLevelCell(:), maxLevelCell(:)
  real(RKIND), allocatable, target :: tracerCur(:,:), normalThicknessFlux(:,:), advMaskHighOrder(:,:), &
                                      advCoefs(:,:), advCoefs3rd(:,:), highOrderFlx(:,:), flxDev(:,:), flxOther(:,:)
  type(c_ptr) :: d(10)
  integer, allocatable :: seed(:)

  nmlfile = '-'; variant = 0; dumpfile = '-'
  if (command_argument_count() >= 1) call get_command_argument(1, nmlfile)
  if (command_argument_count() >= 2) then
    call get_command_argument(2, arg); read(arg,*) variant
  end if
  if (command_argument_count() >= 3) call get_command_argument(3, dumpfile)
  if (trim(nmlfile) /= '-') then
    open(unit=15, file=trim(nmlfile), status='old', iostat=ios)
    if (ios /= 0) error stop 'cannot open the namelist file'
    read(15, nml=nested_nml)
    close(15)
  end if
  nvldim = nVertLevels
  coef3rdOrder = real(2.14, RKIND)       ! nested_vars.F90:35 holds the literal in default (single) precision
  allocate(nAdvCellsForEdge(nEdges), advCellsForEdge(nAdv,nEdges), minLevelCell(nCells), maxLevelCell(nCells), &
           tracerCur(nvldim,nCells), normalThicknessFlux(nvldim,nEdges), advMaskHighOrder(nvldim,nEdges), &
           advCoefs(nAdv,nEdges), advCoefs3rd(nAdv,nEdges), highOrderFlx(nvldim,nEdges), flxDev(nvldim,nEdges), &
           flxOther(nvldim,nEdges))
  call random_seed(size=n)
  allocate(seed(n)); seed = 20241; call random_seed(put=seed)

  ! ---- the reference's initialisation laws (:58-108): random connectivity (its worst case), a topography with
  !      about half the cells at full depth, tracers and coefficients of order 15 - 21
  do iCell = 1, nCells
    minLevelCell(iCell) = 1
    call random_number(randNum)
    maxLevelCell(iCell) = min(max(3, nint(randNum*nVertLevels*2.0)), nVertLevels)
  end do
  tracerCur = 0.0_RKIND
  do iCell = 1, nCells
    do k = minLevelCell(iCell), maxLevelCell(iCell)
      call random_number(randNum)
      tracerCur(k,iCell) = 15.0_RKIND*randNum
    end do
  end do
  do iEdge = 1, nEdges
    nAdvCellsForEdge(iEdge) = nAdv
    do i = 1, nAdv
      call random_number(randNum); advCellsForEdge(i,iEdge) = int(nCells*randNum) + 1
      call random_number(randNum); advCoefs(i,iEdge) = 20.0_RKIND*randNum
      call random_number(randNum); advCoefs3rd(i,iEdge) = 21.0_RKIND*randNum
    end do
    do k = 1, nVertLevels
      call random_number(randNum)
      normalThicknessFlux(k,iEdge) = 15.0_RKIND*(0.5_RKIND - randNum)
      advMaskHighOrder(k,iEdge) = 1.0_RKIND
    end do
  end do
  highOrderFlx = 0.0_RKIND; flxDev = 0.0_RKIND; flxOther = 0.0_RKIND
  i = nlk_set_variant(int(variant, c_int))

  ! ---- the drop-in call on host arrays (transfers included)
  call system_clock(t1)
  call must(nlk_high_order_flux(nEdges, nCells, nVertLevels, nvldim, nAdv, nAdvCellsForEdge, advCellsForEdge, &
            minLevelCell, maxLevelCell, tracerCur, normalThicknessFlux, advMaskHighOrder, advCoefs, advCoefs3rd, &
            coef3rdOrder, highOrderFlx), 'nlk_high_order_flux')
  call system_clock(t2, rt)
  write(*,'(a,f12.6)') ' Host-array call (transfers included, first call), seconds: ', dble(t2-t1)/dble(rt)

  ! ---- data to the device once, nIters timed iterations, result back (:160-200)
  call system_clock(t1)
  call to_device(d(1), c_loc(nAdvCellsForEdge), 4_c_size_t*nEdges)
  call to_device(d(2), c_loc(advCellsForEdge), 4_c_size_t*nAdv*nEdges)
  call to_device(d(3), c_loc(minLevelCell), 4_c_size_t*nCells)
  call to_device(d(4), c_loc(maxLevelCell), 4_c_size_t*nCells)
  call to_device(d(5), c_loc(tracerCur), 8_c_size_t*nvldim*nCells)
  call to_device(d(6), c_loc(normalThicknessFlux), 8_c_size_t*nvldim*nEdges)
  call to_device(d(7), c_loc(advMaskHighOrder), 8_c_size_t*nvldim*nEdges)
  call to_device(d(8), c_loc(advCoefs), 8_c_size_t*nAdv*nEdges)
  call to_device(d(9), c_loc(advCoefs3rd), 8_c_size_t*nAdv*nEdges)
  call to_device(d(10), c_loc(flxDev), 8_c_size_t*nvldim*nEdges)
  call system_clock(t2, rt)
  write(*,'(a,f12.6)') ' Data transfer, seconds: ', dble(t2-t1)/dble(rt)
  call must(run_on_device(), 'nlk_high_order_flux_device')        ! (untimed first launch)
  call must(hipDeviceSynchronize(), 'hipDeviceSynchronize')
  call system_clock(t1)
  do n = 1, nIters
    call must(run_on_device(), 'nlk_high_order_flux_device')
  end do
  call must(hipDeviceSynchronize(), 'hipDeviceSynchronize')
  call system_clock(t2, rt)
  write(*,'(a,i0,a,f12.6,a,es12.4)') ' HIP kernel, ', nIters, ' iterations, seconds: ', dble(t2-t1)/dble(rt), &
       '   edge-level fluxes / s: ', dble(nIters)*dble(nEdges)*dble(nVertLevels)/(dble(t2-t1)/dble(rt))
  call must(hipMemcpy(c_loc(flxDev), d(10), 8_c_size_t*nvldim*nEdges, 2_c_int), 'hipMemcpy (result)')
  if (any(flxDev /= highOrderFlx)) then
    print *, 'Error computing highOrderFlx: the device-resident call differs from the host-array call'
    error stop 2
  end if
  write(*,*) 'Self-check (host-array call against device-resident call): identical'

  ! ---- the other variant, device-resident, against the first to the reference's tolerance (:215-225)
  i = nlk_set_variant(int(1 - variant, c_int))
  call must(run_on_device(), 'nlk_high_order_flux_device (other variant)')
  call must(hipMemcpy(c_loc(flxOther), d(10), 8_c_size_t*nvldim*nEdges, 2_c_int), 'hipMemcpy (result)')
  ! The program's metric (difference relative to the value, :215-220) is printed; the two variants round differently,
  ! so a flux whose terms -- of the order of the largest flux -- cancel to 1e-8 of their size shows 1e-10 there without
  ! being wrong: what fails the check is a difference beyond errTol relative to max(|value|, 1e-6 x the largest flux).
  nbad = 0; worst = 0.0_RKIND; worstScaled = 0.0_RKIND; nrel = 0
  floorVal = 1.0e-6_RKIND*maxval(abs(highOrderFlx))
  do iEdge = 1, nEdges
    do k = 1, nVertLevels
      relErr = abs(flxOther(k,iEdge) - highOrderFlx(k,iEdge))
      worstScaled = max(worstScaled, relErr/max(abs(highOrderFlx(k,iEdge)), floorVal))
      if (relErr/max(abs(highOrderFlx(k,iEdge)), floorVal) > errTol) then
        nbad = nbad + 1
        if (nbad <= 5) print *, 'Error computing highOrderFlx, EXACT against FAST: ', k, iEdge, &
                                 highOrderFlx(k,iEdge), flxOther(k,iEdge)
      end if
      if (highOrderFlx(k,iEdge) /= 0.0_RKIND) relErr = relErr/abs(highOrderFlx(k,iEdge))
      worst = max(worst, relErr)
      if (relErr > errTol) nrel = nrel + 1
    end do
  end do
  write(*,'(a,es12.4,a,i0,a)') ' Self-check (EXACT against FAST), largest relative difference: ', worst, &
       '   (', nrel, ' cancelling sums beyond 1e-10 of their own value)'
  write(*,'(a,es12.4,a,i0)') ' Self-check (EXACT against FAST), largest difference on the scale of the fluxes: ', &
       worstScaled, '   beyond errTol: ', nbad
  if (nbad > 0) error stop 3
  do i = 1, 10
    call must(hipFree(d(i)), 'hipFree')
  end do

  if (trim(dumpfile) /= '-') then
    open(unit=11, file=trim(dumpfile), access='stream', form='unformatted', status='replace')
    write(11) int(nEdges,4), int(nCells,4), int(nVertLevels,4), int(nvldim,4), int(nAdv,4), coef3rdOrder
    write(11) nAdvCellsForEdge, advCellsForEdge, minLevelCell, maxLevelCell
    write(11) tracerCur, normalThicknessFlux, advMaskHighOrder, advCoefs, advCoefs3rd, highOrderFlx
    close(11)
  end if

contains
  subroutine to_device(dp, hp, bytes)
    type(c_ptr), intent(out) :: dp
    type(c_ptr), intent(in) :: hp
    integer(c_size_t), intent(in) :: bytes
    call must(hipMalloc(dp, bytes), 'hipMalloc')
    call must(hipMemcpy(dp, hp, bytes, 1_c_int), 'hipMemcpy (to the device)')
  end subroutine to_device

  integer(c_int) function run_on_device() result(rc)
    rc = nlk_high_order_flux_device(nEdges, nCells, nVertLevels, nvldim, nAdv, d(1), d(2), d(3), d(4), d(5), d(6), &
                                    d(7), d(8), d(9), coef3rdOrder, d(10), c_null_ptr)
  end function run_on_device
end program nested_hip
