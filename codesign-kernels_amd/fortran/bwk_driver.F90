! bwk_driver.F90 -- the reference's `program biharmonic_wk_scalar_kernel`
! (atmosphere/biharmonic_wk_kernel.F90:545-580: allocate -> initialize_data -> routine -> L2 norm against a
! trusted result) with libbwk_hip.so (include/bwk_hip.h) in the place of its OpenACC variants.
!
!   ./bwk_driver [nelemd [variant [dumpfile [reffile]]]]
!     nelemd    elements (reference: 16, :12-14); nlev = 72, qsize = 40, np = 4 as the reference fixes them (:8-10)
!     variant   0 EXACT (bit-identical to the reference CPU routine), 1 FAST (FMA contraction)
!     dumpfile  qtens after the call, stream access ("-" = none)
!     reffile   a trusted qtens (stream access): the reference's own `L2 norm` line (:69-73) is printed against it
! The driver carries no CPU routine (the product has no CPU path).  Without a reffile it still validates itself:
! the other variant is run on the same data and the L2 norm between the two is printed ("Self-check").
module bwk_hip_mod
  use iso_c_binding
  implicit none
  integer, parameter :: rk = c_double, np = 4, nlev = 72, qsize = 40
  ! type element_t of the reference (:23-27), same components in the same order; bind(C) makes an element
  ! the 144 contiguous doubles the C-ABI takes
  type, bind(C) :: element_t
    real(rk) :: Dinv(np,np,2,2)
    real(rk) :: spheremp(np,np)
    real(rk) :: tensorVisc(np,np,2,2)
  end type element_t
  interface
    integer(c_int) function bwk_biharmonic_wk_scalar(nelemd, nlev_, qsize_, qtens, dvv, elem) &
        bind(C, name="bwk_biharmonic_wk_scalar")
      import :: c_int, c_int64_t, c_double, element_t
      integer(c_int64_t), value :: nelemd
      integer(c_int), value :: nlev_, qsize_
      real(c_double) :: qtens(*)
      real(c_double), intent(in) :: dvv(*)
      type(element_t), intent(in) :: elem(*)
    end function
    integer(c_int) function bwk_set_variant(v) bind(C, name="bwk_set_variant")
      import :: c_int
      integer(c_int), value :: v
    end function
    function bwk_last_error() bind(C, name="bwk_last_error") result(p)
      import :: c_ptr
      type(c_ptr) :: p
    end function
  end interface
contains
  ! the reference's portable generator (:77-91), the law its initialize_data draws every input from
  subroutine lcg_fill(n, a, state)
    integer(c_int64_t), intent(in) :: n
    real(rk), intent(out) :: a(n)
    integer, intent(inout) :: state
    integer(c_int64_t) :: i
    do i = 1, n
      state = mod(1301*state + 97, 131072)
      a(i) = state / 131072.0_rk
    end do
  end subroutine lcg_fill

  subroutine fail_with_library_text(what)
    character(*), intent(in) :: what
    character(kind=c_char), pointer :: s(:)
    integer :: n
    call c_f_pointer(bwk_last_error(), s, [512])
    n = 0
    do while (n < 512)
      if (s(n+1) == c_null_char) exit
      n = n + 1
    end do
    write(*,*) what, ': ', s(1:n)
    error stop 1
  end subroutine fail_with_library_text
end module bwk_hip_mod

program bwk_driver
  use bwk_hip_mod
  implicit none
  integer(c_int64_t) :: nelemd
  integer :: variant, state, ie, ios
  integer(8) :: t1, t2, rt
  character(len=512) :: arg, dumpfile, reffile
  real(rk) :: dvv(np,np), l2
  type(element_t), allocatable :: elem(:)
  real(rk), allocatable :: qtens(:,:,:,:,:), qtens_first(:,:,:,:,:), trusted(:,:,:,:,:)

  nelemd = 16; variant = 0; dumpfile = '-'; reffile = '-'
  if (command_argument_count() >= 1) then
    call get_command_argument(1, arg); read(arg,*) nelemd
  end if
  if (command_argument_count() >= 2) then
    call get_command_argument(2, arg); read(arg,*) variant
  end if
  if (command_argument_count() >= 3) call get_command_argument(3, dumpfile)
  if (command_argument_count() >= 4) call get_command_argument(4, reffile)
  allocate(elem(nelemd), qtens(np,np,nlev,qsize,nelemd), qtens_first(np,np,nlev,qsize,nelemd))

  call initialize_data()
  ie = bwk_set_variant(int(variant, c_int))
  call system_clock(t1)
  if (bwk_biharmonic_wk_scalar(nelemd, int(nlev,c_int), int(qsize,c_int), qtens, dvv, elem) /= 0) &
    call fail_with_library_text('bwk_biharmonic_wk_scalar')
  call system_clock(t2, rt)
  write(*,'(a,es14.6,a)') ' HIP  time: ', dble(t2-t1)/dble(rt), '  (host arrays: transfers included, first call)'
  qtens_first = qtens
  if (trim(dumpfile) /= '-') then
    open(unit=11, file=trim(dumpfile), access='stream', form='unformatted', status='replace')
    write(11) qtens
    close(11)
  end if
  if (trim(reffile) /= '-') then
    allocate(trusted(np,np,nlev,qsize,nelemd))
    open(unit=12, file=trim(reffile), access='stream', form='unformatted', status='old', iostat=ios)
    if (ios /= 0) error stop 'cannot open the reference file'
    read(12) trusted
    close(12)
    l2 = sqrt( sum( (qtens - trusted)**2 ) / sum(trusted**2) )     ! compute_l2norm, :69-73
    write(*,*) 'HIP  L2 norm: ', l2
  end if

  ! the other variant on the same data: a second, independently compiled kernel
  call initialize_data()
  ie = bwk_set_variant(int(1 - variant, c_int))
  call system_clock(t1)
  if (bwk_biharmonic_wk_scalar(nelemd, int(nlev,c_int), int(qsize,c_int), qtens, dvv, elem) /= 0) &
    call fail_with_library_text('bwk_biharmonic_wk_scalar (other variant)')
  call system_clock(t2, rt)
  write(*,'(a,es14.6,a)') ' HIP  time: ', dble(t2-t1)/dble(rt), '  (other variant, second call)'
  l2 = sqrt( sum( (qtens - qtens_first)**2 ) / sum(qtens_first**2) )
  write(*,*) 'Self-check (EXACT against FAST) L2 norm: ', l2
  if (.not. (l2 < 1.0e-13_rk)) error stop 'self-check failed'

contains
  ! the reference's initialize_data (:48-58): Dvv with a reset generator, then per element Dinv, spheremp,
  ! tensorVisc, then qtens -- every array in its storage order
  subroutine initialize_data()
    state = 11
    call lcg_fill(int(np*np, c_int64_t), dvv, state)
    do ie = 1, int(nelemd)
      call lcg_fill(int(np*np*4, c_int64_t), elem(ie)%Dinv, state)
      call lcg_fill(int(np*np, c_int64_t), elem(ie)%spheremp, state)
      call lcg_fill(int(np*np*4, c_int64_t), elem(ie)%tensorVisc, state)
    end do
    call lcg_fill(int(np*np,c_int64_t)*nlev*qsize*nelemd, qtens, state)
  end subroutine initialize_data
end program bwk_driver
