!> Driver of the MI355X build: the role of the reference's `program test_advect`
!! (mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:3-68, :645-683).
!!
!!   ./advect [ncrms nx nz [dist [variant [dumpfile [reffile [ntracers [ngpus [mode]]]]]]]]
!!   ./advect file.nml        sizes from a namelist (the reference's other mini-app is configured
!!                            this way, nested_loops/nested.nml:1-7):
!!                            &advect_nml ncrms=65536, nx=32, nz=28, dist=1, variant=1, ntracers=25,
!!                                        ngpus=8, dumpfile='-', reffile='-', mode='auto' /
!!
!! mode = host   : global arrays on the HOST (the reference's own situation: init on the host, `update
!!                 device`, kernels, `update host`), everything below;
!!        device : global arrays generated and kept on the ROOT GPU, scattered / gathered from there
!!                 (RCCL over xGMI for ngpus > 1), checksums on the device -- no host array of the
!!                 problem's size: BASELINE.json configs[4] (524288 x 25 tracers, 107.6 GB of f);
!!        auto   : device when one copy of f exceeds 4 GiB, else host (default).
!!
!! ntracers > 1: the tracer-batched call (same u,w,rho,rhow,adz for every tracer, tracer index
!! slowest); ngpus > 1: the ncrms axis sharded over the GPUs of the node, inputs scattered and
!! outputs gathered over RCCL (include/mpdata_hip.h section 3b).
!!
!! Sequence (reference :48-58): init() -> advect_scalar2D(f,u,w,rho,rhow,flux)
!! -> save() -> compare().  Sizes are run-time (the reference fixes them at compile time,
!! :7-9); inputs come from a portable seeded generator instead of the
!! compiler's random_number (:649-660), in the same fill order.  Timing lines
!! keep the reference's format (:640, :239).  The reference's compare() step
!! (:679-683) needs the CPU routine's result; this program does not contain a
!! CPU advection routine.  Instead compare() takes the trusted result from
!! `reffile` (stream binary, f then flux -- the format save() writes, and the
!! format the reference executable built by oracle/build_ref.py dumps) and
!! prints the reference's two "Relative L1 Error" lines (:681-682);
!! tests/test_fortran_driver.py feeds it the oracle's result.  `dumpfile` / `reffile`
!! may be `-` to skip.  Without any file the host mode still validates itself: its drop-in call and its
!! resident run go through two independent kernel families of the library, and the program prints the
!! same two lines for one against the other ("Self-check").
program test_advect
  use iso_c_binding
  use mpdata_grid
  use mpdata_hip_mod
  implicit none

  real(rp), allocatable :: f(:,:,:,:,:), u(:,:,:,:), w(:,:,:,:), rho(:,:), rhow(:,:), flux(:,:,:)
  real(rp), allocatable :: f_in(:,:,:,:,:), f_call(:,:,:,:,:), flux_call(:,:,:)
  integer(c_int64_t) :: n_arg
  integer :: nx_arg, nz_arg, dist, variant, rc, nt_arg, ng_arg, ranks
  character(len=512) :: arg, dumpfile, reffile
  character(len=16) :: mode
  real(c_double) :: d_kms, d_wall, d_sumf, d_sumflux
  integer(8) :: t1, t2, tr
  real(rp) :: kms
  real(c_double) :: sc_s, ga_s
  integer(c_int64_t) :: sc_b, ga_b

  n_arg = 64; nx_arg = 32; nz_arg = 28; dist = 1; variant = 0; dumpfile = ''; reffile = ''
  nt_arg = 1; ng_arg = 1; mode = 'auto'
  call get_command_argument(1, arg)
  if (command_argument_count() == 1 .and. index(arg, '.nml') > 0) then
    call read_namelist(trim(arg), n_arg, nx_arg, nz_arg, dist, variant, nt_arg, ng_arg, dumpfile, reffile, mode)
  end if
  if (command_argument_count() >= 3) then
    call get_command_argument(1, arg); read(arg, *) n_arg
    call get_command_argument(2, arg); read(arg, *) nx_arg
    call get_command_argument(3, arg); read(arg, *) nz_arg
  end if
  if (command_argument_count() >= 4) then
    call get_command_argument(4, arg); read(arg, *) dist
  end if
  if (command_argument_count() >= 5) then
    call get_command_argument(5, arg); read(arg, *) variant
  end if
  if (command_argument_count() >= 6) call get_command_argument(6, dumpfile)
  if (command_argument_count() >= 7) call get_command_argument(7, reffile)
  if (command_argument_count() >= 8) then
    call get_command_argument(8, arg); read(arg, *) nt_arg
  end if
  if (command_argument_count() >= 9) then
    call get_command_argument(9, arg); read(arg, *) ng_arg
  end if
  if (command_argument_count() >= 10) call get_command_argument(10, mode)
  if (trim(dumpfile) == '-') dumpfile = ''
  if (trim(reffile) == '-') reffile = ''

  call grid_set(n_arg, nx_arg, nz_arg, nt_arg, ng_arg)
  if (trim(mode) == 'auto') then
    mode = 'host'
    if (nslices*int(nx+6, 8)*int(nzm, 8)*int(ntracers, 8)*8_8 > 4294967296_8) mode = 'device'
  end if
  if (trim(mode) == 'device') then
    ! ---- the problem lives on the root GPU: no host array of its size (see the header)
    write(*,*) 'ncrms, nx, nz, ntracers, ngpus: ', nslices, nx, nz, ntracers, ngpus
    write(*,*) 'mode: device (global arrays generated and kept on the root GPU)'
    rc = mpdata_set_variant(int(variant, c_int))
    call advect_device_problem_run(100_8, dist, d_kms, d_wall, d_sumf, d_sumflux, ranks)
    write(*,*) 'RCCL ranks seen (ncclCommCount; 0 = no communicator): ', ranks
    write(*,*) 'HIP Timing: ', d_wall
    write(*,*) 'HIP kernel (hipEvent) seconds: ', d_kms*1.0d-3
    call print_transfer_stats()
    write(*,*) 'cell updates per call: ', nslices*int(nx,8)*int(nzm,8)*int(ntracers,8)
    write(*,*) 'checksum f   : ', d_sumf
    write(*,*) 'checksum flux: ', d_sumflux
  else
    call run_host_mode()
  end if

contains

  !> the reference's own situation: global arrays on the host (init on the host, `update device`, kernels,
  !! `update host`; :48-58, :105-110, :237-242)
  subroutine run_host_mode()
  allocate(f(nslices, -2:nx+3, 1, nzm, ntracers), u(nslices, -1:nx+3, 1, nzm), w(nslices, -1:nx+2, 1, nz))
  allocate(rho(nslices, nzm), rhow(nslices, nz), flux(nslices, nz, ntracers))
  write(*,*) 'ncrms, nx, nz, ntracers, ngpus: ', nslices, nx, nz, ntracers, ngpus
  rc = mpdata_set_variant(int(variant, c_int))

  ! ---- the drop-in call: host arrays, transfers inside (reference :53 pattern).  Twice, as the
  !      reference's validation sequence does (:52-58 and the two timing pairs of
  !      results/advect.pgiacc.17.7:2-13): the first call also pays the device / context start-up
  call init()
  allocate(f_in, source=f)
  call system_clock(t1)
  call advect_scalar2D(f, u, w, rho, rhow, flux)
  call system_clock(t2, tr)
  write(*,*) 'HIP Timing (first call: context start-up + transfers + kernel): ', dble(t2-t1)/dble(tr)
  call init()
  call system_clock(t1)
  call advect_scalar2D(f, u, w, rho, rhow, flux)
  call system_clock(t2, tr)
  write(*,*) 'HIP Timing (call, transfers included): ', dble(t2-t1)/dble(tr)
  call print_transfer_stats()
  call save()
  call compare()
  allocate(f_call, source=f)          ! (kept for the in-program self-check below)
  allocate(flux_call, source=flux)

  ! ---- device-resident: the reference's timed region (kernels only, :110-:238)
  call init()
  call advect_resident_begin(f, u, w, rho, rhow, flux)
  call advect_resident_run()            ! warm-up (the reference's first OpenACC call pays it too)
  call advect_resident_end(f, flux)
  call init()
  call advect_resident_begin(f, u, w, rho, rhow, flux)
  call system_clock(t1)
  call advect_resident_run(kms)
  call system_clock(t2, tr)
  write(*,*) 'HIP Timing: ', dble(t2-t1)/dble(tr)
  write(*,*) 'HIP kernel (hipEvent) seconds: ', kms*1.0e-3_rp
  call advect_resident_end(f, flux)
  call print_transfer_stats()
  write(*,*) 'cell updates per call: ', nslices*int(nx,8)*int(nzm,8)*int(ntracers,8)
  write(*,*) 'checksum f   : ', sum(f)
  write(*,*) 'checksum flux: ', sum(flux(:,1:nzm,:))
  ! ---- in-program self-check, the reference's `init -> variant -> compare` (:52-58) without a CPU routine:
  !      the drop-in call above and the resident run just finished go through two INDEPENDENT kernel families
  !      of the library (reference-layout x-march kernel / plan-layout wave-major kernel, different lane
  !      mappings, data paths and summation trees) on the same init() data.  EXACT: f must agree bit for bit
  !      (0.0), flux to rounding; FAST: both to rounding.
  write(*,*) 'Self-check (host-call kernel vs resident-plan kernel):'
  write(*,*) 'Relative L1 Error - f    : ' , sum(abs( f    - f_call    )) / sum(abs( f_call    ))
  write(*,*) 'Relative L1 Error - flux : ' , sum(abs( flux(:,1:nzm,:) - flux_call(:,1:nzm,:) )) / sum(abs( flux_call(:,1:nzm,:) ))
  end subroutine run_host_mode

  !> the namelist form of the command line; names as in BASELINE.json / the C-ABI.  Values not
  !! named in the file keep the defaults passed in.
  subroutine read_namelist(path, o_ncrms, o_nx, o_nz, o_dist, o_variant, o_ntracers, o_ngpus, o_dump, o_ref, o_mode)
    character(*), intent(in) :: path
    integer(c_int64_t), intent(inout) :: o_ncrms
    integer, intent(inout) :: o_nx, o_nz, o_dist, o_variant, o_ntracers, o_ngpus
    character(len=512), intent(inout) :: o_dump, o_ref
    character(len=16), intent(inout) :: o_mode
    integer(c_int64_t) :: ncrms
    integer :: nx, nz, dist, variant, ntracers, ngpus, iu
    character(len=512) :: dumpfile, reffile
    character(len=16) :: mode
    namelist /advect_nml/ ncrms, nx, nz, dist, variant, ntracers, ngpus, dumpfile, reffile, mode
    ncrms = o_ncrms; nx = o_nx; nz = o_nz; dist = o_dist; variant = o_variant
    ntracers = o_ntracers; ngpus = o_ngpus; dumpfile = o_dump; reffile = o_ref; mode = o_mode
    open(newunit=iu, file=path, status='old', action='read')
    read(iu, nml=advect_nml)
    close(iu)
    o_ncrms = ncrms; o_nx = nx; o_nz = nz; o_dist = dist; o_variant = variant
    o_ntracers = ntracers; o_ngpus = ngpus; o_dump = dumpfile; o_ref = reffile; o_mode = mode
  end subroutine read_namelist

  subroutine print_transfer_stats()
    if (ngpus <= 1) return
    call advect_transfer_stats(sc_s, ga_s, sc_b, ga_b)
    write(*,*) 'scatter seconds, GB/s per peer link: ', sc_s, dble(sc_b)/max(sc_s, 1d-12)*1d-9
    write(*,*) 'gather  seconds, GB/s per peer link: ', ga_s, dble(ga_b)/max(ga_s, 1d-12)*1d-9
  end subroutine print_transfer_stats

  !> splitmix64-style counter generator, bit-identical to
  !! mpdata_fill_synthetic_device / the test oracle's generator: element j
  !! (0-based, Fortran order) of array `sid` = top53(mix(seed + sid*K + (j+1)*G)) * 2^-53 + shift
  subroutine fill(a, n, sid, seed, dist)
    integer(8), intent(in) :: n
    real(rp), intent(out) :: a(n)
    integer, intent(in) :: sid, dist
    integer(8), intent(in) :: seed
    integer(8), parameter :: golden = int(z'9E3779B97F4A7C15', 8), ksid = int(z'D1B54A32D192ED03', 8)
    integer(8), parameter :: m1 = int(z'BF58476D1CE4E5B9', 8), m2 = int(z'94D049BB133111EB', 8)
    integer(8) :: j, z, base
    real(c_double) :: shift   ! the law is defined in fp64; an fp32 build rounds the fp64 value
    shift = 0._c_double
    if (dist == 1) then
      if (sid == 2 .or. sid == 3) shift = -0.5_c_double
      if (sid == 0 .or. sid == 4 .or. sid == 5) shift = 0.5_c_double
    else if (dist == 3) then
      if (sid == 2 .or. sid == 3) shift = -0.5_c_double
    end if
    base = seed + int(sid, 8) * ksid
    do j = 1, n
      z = base + j * golden
      z = ieor(z, shiftr(z, 30)) * m1
      z = ieor(z, shiftr(z, 27)) * m2
      z = ieor(z, shiftr(z, 31))
      a(j) = real(real(shiftr(z, 11), c_double) * 2._c_double**(-53) + shift, rp)
    end do
  end subroutine fill

  !> reference :645-665: same arrays, same order (adz,f,u,w,rho,rhow,flux)
  subroutine init()
    integer(8), parameter :: seed = 100_8
    call fill(adz,  size(adz, kind=8),  0, seed, dist)
    call fill(f,    size(f, kind=8),    1, seed, dist)
    call fill(u,    size(u, kind=8),    2, seed, dist)
    call fill(w,    size(w, kind=8),    3, seed, dist)
    call fill(rho,  size(rho, kind=8),  4, seed, dist)
    call fill(rhow, size(rhow, kind=8), 5, seed, dist)
    call fill(flux, size(flux, kind=8), 6, seed, dist)
  end subroutine init

  !> reference :668-676 keeps copies for compare(); here the result is written
  !! out (stream binary: f then flux) when a dump file was named
  subroutine save()
    integer :: iu
    if (len_trim(dumpfile) == 0) return
    open(newunit=iu, file=trim(dumpfile), access='stream', form='unformatted', status='replace')
    write(iu) f
    write(iu) flux
    close(iu)
  end subroutine save

  !> reference :679-683, with the trusted result read from `reffile` instead of kept
  !! from an in-program CPU call
  subroutine compare()
    real(rp), allocatable :: f_save(:,:,:,:,:), flux_save(:,:,:)
    integer :: iu
    if (len_trim(reffile) == 0) return
    allocate(f_save, mold=f)
    allocate(flux_save, mold=flux)
    open(newunit=iu, file=trim(reffile), access='stream', form='unformatted', status='old')
    read(iu) f_save
    read(iu) flux_save
    close(iu)
    write(*,*) 'Relative L1 Error - f    : ' , sum(abs( f    - f_save    )) / sum(abs( f_save    ))
    write(*,*) 'Relative L1 Error - flux : ' , sum(abs( flux - flux_save )) / sum(abs( flux_save ))
  end subroutine compare

end program test_advect
