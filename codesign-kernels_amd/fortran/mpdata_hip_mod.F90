!> Fortran face of libmpdata_hip.so (include/mpdata_hip.h): the thin
!! ISO_C_BINDING layer and the drop-in `advect_scalar2D(f,u,w,rho,rhow,flux)`.
!!
!! The dummy-argument list and the array shapes are those of the reference's
!! advect_scalar2D_cpu / _openacc_1 / _openacc_2
!! (mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:477-484, :247, :72);
!! sizes and adz come from module mpdata_grid, where the reference takes them
!! by host association (:7-30).  The binding style (bind(C,name=..), scalars by
!! value, arrays as bare pointers, persistent device state behind the library)
!! follows the reference's own Fortran->C++ precedent, nested_loops/cke_mod.F90:4-50.
!! There is no CPU implementation behind this interface: a failing HIP call
!! stops the program.
module mpdata_hip_mod
  use iso_c_binding
  use mpdata_grid
  implicit none
  private
  public :: advect_scalar2D, advect_resident_begin, advect_resident_run, advect_resident_end
  public :: mpdata_set_variant, mpdata_check, advect_transfer_stats
  public :: advect_device_problem_run

  ! the C entry points that carry reals exist per precision (include/mpdata_hip.h sections 1-3
  ! and 6); `make single=1` (-DMPDATA_SINGLE) binds the fp32 ones, rp = c_float
#ifdef MPDATA_SINGLE
#define MPDATA_C_ADVECT "mpdata_advect_scalar2d_f32"
#define MPDATA_C_PLAN_CREATE "mpdata_plan_create_f32"
#define MPDATA_C_PLAN_UPLOAD "mpdata_plan_upload_f32"
#define MPDATA_C_PLAN_DOWNLOAD "mpdata_plan_download_f32"
#else
#define MPDATA_C_ADVECT "mpdata_advect_scalar2d"
#define MPDATA_C_PLAN_CREATE "mpdata_plan_create"
#define MPDATA_C_PLAN_UPLOAD "mpdata_plan_upload"
#define MPDATA_C_PLAN_DOWNLOAD "mpdata_plan_download"
#endif

  interface
    integer(c_int) function mpdata_advect_scalar2d_c(ncrms, nx, nz, ntracers, f, u, w, rho, rhow, adz, flux) &
        bind(C, name=MPDATA_C_ADVECT)
      import :: c_int, c_int64_t, rp
      integer(c_int64_t), value :: ncrms
      integer(c_int), value :: nx, nz, ntracers
      real(rp) :: f(*), flux(*)
      real(rp), intent(in) :: u(*), w(*), rho(*), rhow(*), adz(*)
    end function
    integer(c_int) function mpdata_plan_create_c(ncrms, nx, nz, ntracers, plan) bind(C, name=MPDATA_C_PLAN_CREATE)
      import :: c_int, c_int64_t, c_ptr
      integer(c_int64_t), value :: ncrms
      integer(c_int), value :: nx, nz, ntracers
      type(c_ptr) :: plan
    end function
    integer(c_int) function mpdata_plan_create_multi_c(ncrms, nx, nz, ntracers, ngpus, plan) &
        bind(C, name="mpdata_plan_create_multi")
      import :: c_int, c_int64_t, c_ptr
      integer(c_int64_t), value :: ncrms
      integer(c_int), value :: nx, nz, ntracers, ngpus
      type(c_ptr) :: plan
    end function
    integer(c_int) function mpdata_plan_transfer_stats_c(plan, scatter_s, gather_s, sbytes, gbytes, transport) &
        bind(C, name="mpdata_plan_transfer_stats")
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: plan
      real(c_double) :: scatter_s, gather_s
      integer(c_int64_t) :: sbytes, gbytes
      integer(c_int) :: transport
    end function
    integer(c_int) function mpdata_plan_upload_c(plan, f, u, w, rho, rhow, adz, flux) bind(C, name=MPDATA_C_PLAN_UPLOAD)
      import :: c_int, c_ptr, rp
      type(c_ptr), value :: plan
      real(rp), intent(in) :: f(*), u(*), w(*), rho(*), rhow(*), adz(*), flux(*)
    end function
    integer(c_int) function mpdata_plan_run_c(plan) bind(C, name="mpdata_plan_run")
      import :: c_int, c_ptr
      type(c_ptr), value :: plan
    end function
    integer(c_int) function mpdata_plan_sync_c(plan) bind(C, name="mpdata_plan_sync")
      import :: c_int, c_ptr
      type(c_ptr), value :: plan
    end function
    integer(c_int) function mpdata_plan_download_c(plan, f, flux) bind(C, name=MPDATA_C_PLAN_DOWNLOAD)
      import :: c_int, c_ptr, rp
      type(c_ptr), value :: plan
      real(rp) :: f(*), flux(*)
    end function
    integer(c_int) function mpdata_plan_last_kernel_ms_c(plan, ms) bind(C, name="mpdata_plan_last_kernel_ms")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: plan
      real(c_double) :: ms
    end function
    integer(c_int) function mpdata_plan_destroy_c(plan) bind(C, name="mpdata_plan_destroy")
      import :: c_int, c_ptr
      type(c_ptr), value :: plan
    end function
    integer(c_int) function mpdata_set_variant(variant) bind(C, name="mpdata_set_variant")
      import :: c_int
      integer(c_int), value :: variant
    end function
    type(c_ptr) function mpdata_last_error_c() bind(C, name="mpdata_last_error")
      import :: c_ptr
    end function
    ! ---- device-resident mode: the global arrays live on the root GPU (include/mpdata_hip.h 3, 3b, 4, 4b)
    integer(c_int) function mpdata_device_alloc_c(ptr, bytes) bind(C, name="mpdata_device_alloc")
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr) :: ptr
      integer(c_int64_t), value :: bytes
    end function
    integer(c_int) function mpdata_plan_device_alloc_c(plan, ptr, bytes) bind(C, name="mpdata_plan_device_alloc")
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: plan
      type(c_ptr) :: ptr
      integer(c_int64_t), value :: bytes
    end function
    integer(c_int) function mpdata_device_free_c(ptr) bind(C, name="mpdata_device_free")
      import :: c_int, c_ptr
      type(c_ptr), value :: ptr
    end function
    integer(c_int) function mpdata_device_sum_c(a, n, blk, stride, s) bind(C, name="mpdata_device_sum")
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: a
      integer(c_int64_t), value :: n, blk, stride
      real(c_double) :: s
    end function
    integer(c_int) function mpdata_fill_synthetic_device_c(a, sid, rows, ncrms_global, sl0, nloc, seed, dist, stream) &
        bind(C, name="mpdata_fill_synthetic_device")
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: a, stream
      integer(c_int), value :: sid, dist
      integer(c_int64_t), value :: rows, ncrms_global, sl0, nloc, seed
    end function
    integer(c_int) function mpdata_plan_import_device_c(plan, f, u, w, rho, rhow, adz, flux, first, n) &
        bind(C, name="mpdata_plan_import_device")
      import :: c_int, c_ptr
      type(c_ptr), value :: plan, f, u, w, rho, rhow, adz, flux
      integer(c_int), value :: first, n
    end function
    integer(c_int) function mpdata_plan_export_device_c(plan, f, flux, first, n) bind(C, name="mpdata_plan_export_device")
      import :: c_int, c_ptr
      type(c_ptr), value :: plan, f, flux
      integer(c_int), value :: first, n
    end function
    integer(c_int) function mpdata_plan_ranks_seen_c(plan) bind(C, name="mpdata_plan_ranks_seen")
      import :: c_int, c_ptr
      type(c_ptr), value :: plan
    end function
  end interface

  type(c_ptr), save :: resident_plan = c_null_ptr
  ! scatter / gather record of the last multi-GPU transfer (advect_transfer_stats)
  real(c_double), save :: last_scatter_s = 0, last_gather_s = 0
  integer(c_int64_t), save :: last_scatter_bytes = 0, last_gather_bytes = 0

contains

  !> `error stop` with the library's message when a C-ABI call failed
  !! (the reference has no error path at all; a failure there is a crash).
  subroutine mpdata_check(rc, what)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: what
    character(kind=c_char), pointer :: msg(:)
    integer :: n
    if (rc == 0) return
    call c_f_pointer(mpdata_last_error_c(), msg, [512])
    n = 1
    do while (n < 512 .and. msg(n) /= c_null_char)
      n = n + 1
    end do
    write(*,*) 'libmpdata_hip: ', what, ' failed, rc=', rc, ': ', msg(1:n-1)
    error stop 1
  end subroutine mpdata_check

  !> Drop-in replacement of `call advect_scalar2D_openacc_N(f,u,w,rho,rhow,flux)`
  !! (reference :53, :57): synchronous, host arrays, transfers included (the
  !! reference's `!$acc update device/host`, :107 and :241).
  subroutine advect_scalar2D(f, u, w, rho, rhow, flux)
    real(rp), intent(inout) :: f    (nslices, -2:nx+3, 1, nzm, ntracers)
    real(rp), intent(in   ) :: u    (nslices, -1:nx+3, 1, nzm)
    real(rp), intent(in   ) :: w    (nslices, -1:nx+2, 1, nz )
    real(rp), intent(in   ) :: rho  (nslices, nzm)
    real(rp), intent(in   ) :: rhow (nslices, nz )
    real(rp), intent(inout) :: flux (nslices, nz, ntracers)   ! level nz is left as it came (reference :541,:624)
    type(c_ptr) :: plan
    if (ngpus <= 1) then
      call mpdata_check(mpdata_advect_scalar2d_c(nslices, nx, nz, ntracers, f, u, w, rho, rhow, adz, flux), &
                        'mpdata_advect_scalar2d')
    else
      ! ncrms sharded over `ngpus` GPUs: scatter (RCCL over xGMI), one kernel per GPU, gather
      call create_plan(plan)
      call mpdata_check(mpdata_plan_upload_c(plan, f, u, w, rho, rhow, adz, flux), 'mpdata_plan_upload')
      call mpdata_check(mpdata_plan_run_c(plan), 'mpdata_plan_run')
      call mpdata_check(mpdata_plan_sync_c(plan), 'mpdata_plan_sync')
      call mpdata_check(mpdata_plan_download_c(plan, f, flux), 'mpdata_plan_download')
      call record_stats(plan)
      call mpdata_check(mpdata_plan_destroy_c(plan), 'mpdata_plan_destroy')
    end if
  end subroutine advect_scalar2D

  subroutine create_plan(plan)
    type(c_ptr), intent(out) :: plan
    if (ngpus <= 1) then
      call mpdata_check(mpdata_plan_create_c(nslices, nx, nz, ntracers, plan), 'mpdata_plan_create')
    else
#ifdef MPDATA_SINGLE
      write(*,*) 'multi-GPU plans are fp64'
      error stop 1
#else
      call mpdata_check(mpdata_plan_create_multi_c(nslices, nx, nz, ntracers, ngpus, plan), 'mpdata_plan_create_multi')
#endif
    end if
  end subroutine create_plan

  subroutine record_stats(plan)
    type(c_ptr), intent(in) :: plan
    integer(c_int) :: tr
    if (ngpus <= 1) return
    call mpdata_check(mpdata_plan_transfer_stats_c(plan, last_scatter_s, last_gather_s, last_scatter_bytes, &
                                                   last_gather_bytes, tr), 'mpdata_plan_transfer_stats')
  end subroutine record_stats

  !> seconds and bytes per peer link of the last multi-GPU scatter (upload) / gather (download)
  subroutine advect_transfer_stats(scatter_s, gather_s, scatter_bytes, gather_bytes)
    real(c_double), intent(out) :: scatter_s, gather_s
    integer(c_int64_t), intent(out) :: scatter_bytes, gather_bytes
    scatter_s = last_scatter_s; gather_s = last_gather_s
    scatter_bytes = last_scatter_bytes; gather_bytes = last_gather_bytes
  end subroutine advect_transfer_stats

  !> The whole sequence with the GLOBAL arrays resident on the root GPU instead of the host: the
  !! inputs are generated there (mpdata_fill_synthetic_device: the law init() uses, element for
  !! element), handed to the plan by mpdata_plan_import_device -- for ngpus > 1 that is the scatter
  !! over RCCL / xGMI, the replacement of the reference's `update device` (:107) --, advected (the
  !! reference's timed region, :110-:238), gathered back (`update host`, :241) and summed on the
  !! device.  No host array of the problem's size exists: BASELINE.json configs[4] (ncrms = 524288 x
  !! 25 tracers: 107.6 GB of f) runs from a host with a few GB.  fp64 only.
  subroutine advect_device_problem_run(seed, dist, kernel_ms, wall_s, sum_f, sum_flux, ranks)
    integer(c_int64_t), intent(in) :: seed
    integer, intent(in) :: dist
    real(c_double), intent(out) :: kernel_ms, wall_s, sum_f, sum_flux
    integer, intent(out) :: ranks
#ifdef MPDATA_SINGLE
    write(*,*) 'the device-resident mode is fp64'
    error stop 1
#else
    type(c_ptr) :: plan, d(0:6)
    integer(c_int64_t) :: rows(0:6), n, nf, nx1
    integer(8) :: t1, t2, tr
    integer :: i
    ! generator ids (the reference's fill order, :654-660): 0 adz, 1 f, 2 u, 3 w, 4 rho, 5 rhow, 6 flux
    rows = [ int(nzm, 8), int(nx+6, 8)*nzm*ntracers, int(nx+5, 8)*nzm, int(nx+4, 8)*nz, int(nzm, 8), int(nz, 8), &
             int(nz, 8)*ntracers ]
    ! the plan first: the global arrays must live on ITS root GPU (shard 0's device; with
    ! MPDATA_MULTI_DEVICES=3,4 that is device 3, not the current one)
    call create_plan(plan)
    do i = 0, 6
      call mpdata_check(mpdata_plan_device_alloc_c(plan, d(i), rows(i)*nslices*8_8), 'mpdata_plan_device_alloc')
      call mpdata_check(mpdata_fill_synthetic_device_c(d(i), int(i, c_int), rows(i), nslices, 0_8, nslices, seed, &
                                                       int(dist, c_int), c_null_ptr), 'mpdata_fill_synthetic_device')
    end do
    ranks = mpdata_plan_ranks_seen_c(plan)
    ! scatter (ngpus > 1) / layout entry; twice: the first run is the warm-up the reference's first
    ! OpenACC call pays as well, the second import restores the inputs for the timed run
    do i = 1, 2
      call mpdata_check(mpdata_plan_import_device_c(plan, d(1), d(2), d(3), d(4), d(5), d(0), d(6), 0_c_int, &
                                                    int(ntracers, c_int)), 'mpdata_plan_import_device')
      call mpdata_check(mpdata_plan_sync_c(plan), 'mpdata_plan_sync')
      call system_clock(t1)
      call mpdata_check(mpdata_plan_run_c(plan), 'mpdata_plan_run')
      call mpdata_check(mpdata_plan_sync_c(plan), 'mpdata_plan_sync')
      call system_clock(t2, tr)
    end do
    wall_s = dble(t2-t1)/dble(tr)
    call mpdata_check(mpdata_plan_last_kernel_ms_c(plan, kernel_ms), 'mpdata_plan_last_kernel_ms')
    call mpdata_check(mpdata_plan_export_device_c(plan, d(1), d(6), 0_c_int, int(ntracers, c_int)), 'mpdata_plan_export_device')
    call mpdata_check(mpdata_plan_sync_c(plan), 'mpdata_plan_sync')
    call record_stats(plan)
    nf = rows(1)*nslices
    call mpdata_check(mpdata_device_sum_c(d(1), nf, nf, nf, sum_f), 'mpdata_device_sum')
    n = rows(6)*nslices; nx1 = int(nz, 8)*nslices     ! flux(:,1:nzm,:) -- level nz is never written (:541, :624)
    call mpdata_check(mpdata_device_sum_c(d(6), n, int(nzm, 8)*nslices, nx1, sum_flux), 'mpdata_device_sum')
    call mpdata_check(mpdata_plan_destroy_c(plan), 'mpdata_plan_destroy')
    do i = 0, 6
      call mpdata_check(mpdata_device_free_c(d(i)), 'mpdata_device_free')
    end do
#endif
  end subroutine advect_device_problem_run

  !> Device-resident form = the reference's timed region (:105-110, :237-242):
  !! begin = `enter data` + `update device`; run = the kernels + `wait`;
  !! end = `update host`.
  subroutine advect_resident_begin(f, u, w, rho, rhow, flux)
    real(rp), intent(in) :: f(nslices, -2:nx+3, 1, nzm, ntracers), u(nslices, -1:nx+3, 1, nzm), w(nslices, -1:nx+2, 1, nz)
    real(rp), intent(in) :: rho(nslices, nzm), rhow(nslices, nz), flux(nslices, nz, ntracers)
    call create_plan(resident_plan)
    call mpdata_check(mpdata_plan_upload_c(resident_plan, f, u, w, rho, rhow, adz, flux), 'mpdata_plan_upload')
  end subroutine advect_resident_begin

  subroutine advect_resident_run(kernel_ms)
    real(rp), intent(out), optional :: kernel_ms
    real(c_double) :: ms
    call mpdata_check(mpdata_plan_run_c(resident_plan), 'mpdata_plan_run')
    call mpdata_check(mpdata_plan_sync_c(resident_plan), 'mpdata_plan_sync')
    if (present(kernel_ms)) then
      call mpdata_check(mpdata_plan_last_kernel_ms_c(resident_plan, ms), 'mpdata_plan_last_kernel_ms')
      kernel_ms = real(ms, rp)
    end if
  end subroutine advect_resident_run

  subroutine advect_resident_end(f, flux)
    real(rp), intent(out) :: f(nslices, -2:nx+3, 1, nzm, ntracers)
    real(rp), intent(inout) :: flux(nslices, nz, ntracers)
    call mpdata_check(mpdata_plan_download_c(resident_plan, f, flux), 'mpdata_plan_download')
    call record_stats(resident_plan)
    call mpdata_check(mpdata_plan_destroy_c(resident_plan), 'mpdata_plan_destroy')
    resident_plan = c_null_ptr
  end subroutine advect_resident_end

end module mpdata_hip_mod
