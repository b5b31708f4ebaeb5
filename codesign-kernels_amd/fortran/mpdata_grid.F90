!> Sizes and vertical grid of the CRM slices.
!!
!! In the reference these are compile-time parameters and a global array at
!! program scope, taken "From grid.F90" of E3SM-MMF
!! (mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:7-30) and reached by
!! the advection routine through host association.  Here they are run-time
!! values in a module the driver sets before the first call.
module mpdata_grid
  use iso_c_binding, only: c_double, c_float, c_int, c_int64_t
  implicit none
#ifdef MPDATA_SINGLE
  !> the reference's precision switch (:12).  Note: the `selected_real_kind(7)` it asks for
  !! is fp64 on conforming compilers (IEEE single has 6 decimal digits); c_float is fp32.
  integer, parameter :: rp = c_float
#else
  integer, parameter :: rp = c_double          !< reference :13, selected_real_kind(13)
#endif
  integer(c_int64_t) :: nslices = 48            !< CRM instances ("ncrms"), reference :7
  integer(c_int)     :: nz = 58, nx = 32        !< reference :8-9
  integer(c_int)     :: nzm = 57                !< nz-1, reference :14
  integer(c_int)     :: ntracers = 1            !< this build's extension (tracer index slowest)
  integer(c_int)     :: ngpus = 1               !< GPUs the ncrms axis is sharded over (this build's extension)
  real(rp), allocatable :: adz(:,:)             !< (nslices,nzm), reference :30
contains
  subroutine grid_set(ncrms_in, nx_in, nz_in, ntracers_in, ngpus_in)
    integer(c_int64_t), intent(in) :: ncrms_in
    integer, intent(in) :: nx_in, nz_in
    integer, intent(in), optional :: ntracers_in, ngpus_in
    nslices = ncrms_in
    nx = nx_in
    nz = nz_in
    nzm = nz_in - 1
    ntracers = 1
    if (present(ntracers_in)) ntracers = ntracers_in
    ngpus = 1
    if (present(ngpus_in)) ngpus = ngpus_in
    if (allocated(adz)) deallocate(adz)
    allocate(adz(nslices, nzm))
  end subroutine grid_set
end module mpdata_grid
