!> mckpp_initialize_ocean_model with the reference's interface
!! (src/mckpp_initialize_ocean.F90:18): tridiagonal geometry factors on the
!! host, then the per-column initial vmix / seeds on the device.
module mckpp_initialize_ocean
  use iso_c_binding, only: c_int
  use mckpp_parameters
  use mckpp_data_fields, only: kpp_const_fields
  use mckpp_hip_binding
  use mckpp_hip_session
  use mckpp_time_control, only: ntime
  implicit none
contains
  subroutine mckpp_initialize_ocean_model()
    call mckpp_host_tri(nz, nztmax, kpp_const_fields%dto, kpp_const_fields%zm, kpp_const_fields%hm, &
                        kpp_const_fields%tri)
    if (kpp_const_fields%L_RESTART) then
      call mckpp_hip_push_state(force=.true.)
      return
    end if
    call mckpp_hip_push_state(force=.true.)
    call mckpp_hip_check(mckpp_hip_multi_init_ocean(mckpp_hip_multi_handle, int(ntime, c_int)), 'mckpp_hip_init_ocean')
    call mckpp_hip_pull_state(MCKPP_F_ALL)
  end subroutine mckpp_initialize_ocean_model
end module mckpp_initialize_ocean
