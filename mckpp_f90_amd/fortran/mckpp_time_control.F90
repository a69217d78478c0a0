!> Model time step counter read by the physics (reference: module variable
!! `ntime`, src/mckpp_time_control.F90:13, read at bldepth_mod.F90:113 and
!! fluxes_mod.F90:103,110).
module mckpp_time_control
  implicit none
  integer :: ntime = 0
contains
  subroutine mckpp_update_time(nt)   ! src/mckpp_ocean_model_3D.F90:41
    integer, intent(in) :: nt
    ntime = nt
  end subroutine mckpp_update_time
end module mckpp_time_control
