!> mckpp_physics_driver with the reference's interface: no arguments, works on
!! the module globals kpp_3d_fields / kpp_const_fields and reads ntime
!! (src/mckpp_physics_driver_mod.F90:15-73).  The OpenMP column loop with its
!! per-column 3D<->1D copies is replaced by one kernel launch per GPU over its
!! share of the run_physics columns (mckpp_hip_session: mckpp_hip_ndevices); state stays in HBM between calls and only the field
!! groups in mckpp_hip_output_mask come back each step (default: all of them, the reference's contract; a host may
!! opt into the scalar group alone and call mckpp_hip_sync_host when it wants to read the rest;
!! mckpp_physics_finalize brings everything back before it lets the devices go).
module mckpp_physics_driver_mod
  use iso_c_binding, only: c_int
  use mckpp_data_fields, only: kpp_3d_fields, kpp_const_fields
  use mckpp_hip_binding
  use mckpp_hip_session
  use mckpp_time_control, only: ntime
  implicit none
contains
  subroutine mckpp_physics_driver()
    call mckpp_hip_push_state()
    ! what mckpp_boundary_update may have rewritten since the last step (src/mckpp_ocean_model_3D.F90:51-55)
    if (mckpp_hip_ancillaries_every_step) call mckpp_hip_push_ancillaries()
    ! forcing written by mckpp_fluxes into sflux(:,1:6,5,0) (src/mckpp_fluxes_mod.F90:62-69)
    call mckpp_hip_check(mckpp_hip_multi_set_forcing(mckpp_hip_multi_handle, kpp_3d_fields%sflux), 'mckpp_hip_set_forcing')
    ! every device's shard is launched before anything waits (the download below is the first wait)
    call mckpp_hip_check(mckpp_hip_multi_step(mckpp_hip_multi_handle, int(ntime, c_int), 1_c_int), 'mckpp_hip_step')
    ! mckpp_physics_overrides_bottomtemp after the column loop (src/mckpp_physics_driver_mod.F90:67-71)
    if (kpp_const_fields%L_VARY_BOTTOM_TEMP) then
      if (.not. allocated(kpp_3d_fields%bottom_temp)) then
        write (0, '(a)') 'MCKPP-HIP ERROR: L_VARY_BOTTOM_TEMP needs kpp_3d_fields%bottom_temp (mckpp_allocate_3d_optional)'
        error stop 1
      end if
      call mckpp_hip_check(mckpp_hip_multi_bottomtemp(mckpp_hip_multi_handle, kpp_3d_fields%bottom_temp), 'mckpp_hip_bottomtemp')
    end if
    call mckpp_hip_device_advanced()
    call mckpp_hip_pull_state(mckpp_hip_output_mask)
    ! the reference's located warnings (src/mckpp_physics_ocnstep_mod.F90:184-191, 229-236; solvers.F90:140-148)
    if (mckpp_hip_warnings) call mckpp_hip_report_warnings(int(ntime))
  end subroutine mckpp_physics_driver

  subroutine mckpp_physics_finalize()
    call mckpp_hip_detach()
  end subroutine mckpp_physics_finalize
end module mckpp_physics_driver_mod
