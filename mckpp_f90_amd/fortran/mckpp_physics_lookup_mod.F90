!> wm/ws lookup tables.  Same interface as the reference
!! (src/mckpp_physics_lookup_mod.F90:11); the table is filled by the library's
!! host helper so host and device see identical entries.
module mckpp_physics_lookup_mod
  use mckpp_data_fields, only: kpp_const_type
  use mckpp_hip_binding, only: mckpp_host_lookup
  implicit none
contains
  subroutine mckpp_physics_lookup(kpp_const_fields)
    type(kpp_const_type), intent(inout) :: kpp_const_fields
    call mckpp_host_lookup(kpp_const_fields%vonk, kpp_const_fields%wmt, kpp_const_fields%wst)
  end subroutine mckpp_physics_lookup
end module mckpp_physics_lookup_mod
