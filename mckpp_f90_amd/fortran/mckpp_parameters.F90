!> Dimension integers of the column model (hot-path subset).
!! Same module and variable names as the reference (src/mckpp_parameters.F90:4-59)
!! so code that USEs them is unchanged; values are set by the caller
!! (mckpp_set_dimensions below stands in for the reference's namelist read,
!! src/mckpp_initialize_namelist_mod.F90:27-84).
module mckpp_parameters
  use iso_c_binding, only: c_double
  implicit none
  integer :: nz = 0, nzm1 = 0, nzp1 = 0
  integer :: nx = 0, ny = 0, npts = 0
  integer :: nvel = 2, nsclr = 2, nvp1 = 3, nsp1 = 3
  integer :: nztmax = 0, nzp1tmax = 0, ngrid = 1
  integer :: nsflxs = 9, njdt = 1, maxmodeadv = 6
  integer :: itermax = 200
  real(c_double) :: hmixtolfrac = 0.1_c_double
contains
  subroutine mckpp_set_dimensions(nx_in, ny_in, nz_in, nztmax_in)
    integer, intent(in) :: nx_in, ny_in, nz_in
    integer, intent(in), optional :: nztmax_in
    nx = nx_in; ny = ny_in; nz = nz_in
    npts = nx*ny; nzm1 = nz - 1; nzp1 = nz + 1
    nztmax = nzp1
    if (present(nztmax_in)) nztmax = max(nztmax_in, nzp1)
    nzp1tmax = nztmax + 1
    nvp1 = nvel + 1; nsp1 = nsclr + 1
  end subroutine mckpp_set_dimensions
end module mckpp_parameters
