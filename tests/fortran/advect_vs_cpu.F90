!> TEST-ONLY driver (tests/test_fortran_cpu_leg.py builds and runs it; it is not part of the product and is the only
!! Fortran program of this repo that links the oracle): the reference's own validation sequence in ONE program --
!!   call advect_scalar2D_cpu(...) ; save ; init ; call advect_scalar2D_<accelerated>(...) ; compare
!! (mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:48-58, :668-683) -- with the CPU leg = the oracle's C
!! restatement of the reference routine (oracle/mpdata_oracle.c, pinned bit for bit to the reference program) and the
!! accelerated leg = the product's drop-in `call advect_scalar2D(f,u,w,rho,rhow,flux)` through the ISO_C_BINDING shim.
!! The product's own driver (codesign-kernels_amd/fortran/test_advect.F90) carries no CPU routine: the product has no
!! CPU path.   usage: advect_vs_cpu ncrms nx nz dist variant
program advect_vs_cpu
  use iso_c_binding
  use mpdata_grid
  use mpdata_hip_mod
  implicit none
  interface
    integer(c_int) function oracle_advect(ncrms, nx, nz, f, u, w, rho, rhow, adz, flux, nthreads) &
        bind(C, name="mpdata_oracle_advect")
      import :: c_int, c_int64_t, c_double
      integer(c_int64_t), value :: ncrms
      integer(c_int), value :: nx, nz, nthreads
      real(c_double) :: f(*), flux(*)
      real(c_double), intent(in) :: u(*), w(*), rho(*), rhow(*), adz(*)
    end function
    subroutine oracle_fill(a, sid, rows, ncrms_global, sl0, nloc, seed, dist) bind(C, name="mpdata_oracle_fill")
      import :: c_int, c_int64_t, c_double
      real(c_double) :: a(*)
      integer(c_int), value :: sid, dist
      integer(c_int64_t), value :: rows, ncrms_global, sl0, nloc, seed
    end subroutine
  end interface
  real(rp), allocatable :: f(:,:,:,:,:), u(:,:,:,:), w(:,:,:,:), rho(:,:), rhow(:,:), flux(:,:,:)
  real(rp), allocatable :: f_save(:,:,:,:,:), flux_save(:,:,:)
  integer(c_int64_t) :: n_arg
  integer :: nx_arg, nz_arg, dist, variant, rc
  integer(8) :: t1, t2, tr
  character(len=64) :: arg

  call get_command_argument(1, arg); read(arg, *) n_arg
  call get_command_argument(2, arg); read(arg, *) nx_arg
  call get_command_argument(3, arg); read(arg, *) nz_arg
  call get_command_argument(4, arg); read(arg, *) dist
  call get_command_argument(5, arg); read(arg, *) variant
  call grid_set(n_arg, nx_arg, nz_arg)
  rc = mpdata_set_variant(int(variant, c_int))
  allocate(f(nslices, -2:nx+3, 1, nzm, 1), u(nslices, -1:nx+3, 1, nzm), w(nslices, -1:nx+2, 1, nz))
  allocate(rho(nslices, nzm), rhow(nslices, nz), flux(nslices, nz, 1))
  allocate(f_save, mold=f); allocate(flux_save, mold=flux)

  ! ---- CPU leg (reference :48-50)
  call init()
  call system_clock(t1)
  rc = oracle_advect(nslices, nx, nz, f, u, w, rho, rhow, adz, flux, 1_c_int)
  call system_clock(t2, tr)
  if (rc /= 0) error stop 'oracle failed'
  write(*,*) 'CPU Timing: ', real(t2 - t1) / real(tr)
  f_save = f; flux_save = flux                                   ! save(), reference :668-676
  ! ---- accelerated leg (reference :52-58): same inputs again, the drop-in call, compare
  call init()
  call advect_scalar2D(f, u, w, rho, rhow, flux)                 ! (first call: device start-up)
  call init()
  call system_clock(t1)
  call advect_scalar2D(f, u, w, rho, rhow, flux)
  call system_clock(t2, tr)
  write(*,*) 'HIP Timing: ', real(t2 - t1) / real(tr)
  write(*,*) 'Relative L1 Error - f    : ' , sum(abs( f    - f_save    )) / sum(abs( f_save    ))   ! reference :681-682
  write(*,*) 'Relative L1 Error - flux : ' , sum(abs( flux - flux_save )) / sum(abs( flux_save ))
  write(*,*) 'max abs difference - f   : ' , maxval(abs(f - f_save))
contains
  subroutine init()   ! the seeded law of the product driver's init(), through the oracle's generator (same law)
    integer(c_int64_t), parameter :: seed = 100_8
    call oracle_fill(adz,  0_c_int, int(nzm, 8),               nslices, 0_8, nslices, seed, int(dist, c_int))
    call oracle_fill(f,    1_c_int, int((nx + 6) * nzm, 8),    nslices, 0_8, nslices, seed, int(dist, c_int))
    call oracle_fill(u,    2_c_int, int((nx + 5) * nzm, 8),    nslices, 0_8, nslices, seed, int(dist, c_int))
    call oracle_fill(w,    3_c_int, int((nx + 4) * nz, 8),     nslices, 0_8, nslices, seed, int(dist, c_int))
    call oracle_fill(rho,  4_c_int, int(nzm, 8),               nslices, 0_8, nslices, seed, int(dist, c_int))
    call oracle_fill(rhow, 5_c_int, int(nz, 8),                nslices, 0_8, nslices, seed, int(dist, c_int))
    call oracle_fill(flux, 6_c_int, int(nz, 8),                nslices, 0_8, nslices, seed, int(dist, c_int))
  end subroutine init
end program advect_vs_cpu
