!> Per-column entry point with the reference's signature
!! (src/mckpp_physics_ocnstep_mod.F90:43): one column in, one column out.
!! It runs the same device kernel on a one-column batch through a private
!! context, followed by check_profile like the driver loop does
!! (src/mckpp_physics_driver_mod.F90:54-55).  Meant for drop-in use and tests;
!! throughput comes from mckpp_physics_driver.
module mckpp_physics_ocnstep_mod
  use iso_c_binding
  use mckpp_data_fields, only: kpp_1d_type, kpp_const_type
  use mckpp_hip_binding
  use mckpp_hip_session, only: mckpp_hip_check, mckpp_hip_warnings, mckpp_hip_abort_on_zero_pivot, mckpp_hip_column_messages
  use mckpp_hip_onecol
  use mckpp_time_control, only: ntime
  implicit none
  private
  public :: mckpp_physics_ocnstep

contains

  subroutine mckpp_physics_ocnstep(kpp_1d_fields, kpp_const_fields)
    type(kpp_1d_type), intent(inout) :: kpp_1d_fields
    type(kpp_const_type), intent(in), target :: kpp_const_fields
    type(mckpp_state_ptrs_c) :: s
    integer(c_int32_t), target :: st(1), np(1)
    integer(c_int64_t), target :: nflag
    logical :: zero_pivot
    call onecol_attach(kpp_const_fields)
    call onecol_load(kpp_1d_fields, s)
    call mckpp_hip_check(mckpp_hip_step(h1, int(ntime, c_int), 1_c_int), 'mckpp_hip_step (ocnstep)')
    call mckpp_hip_check(mckpp_hip_download(h1, s, int(MCKPP_F_ALL, c_int32_t)), 'mckpp_hip_download (ocnstep)')
    call onecol_store_state(kpp_1d_fields)
    call onecol_store_diag(kpp_1d_fields)
    kpp_1d_fields%comp_flag = .false.
    if (mckpp_hip_warnings) then   ! the reference's located warnings (:184-191, 229-236; solvers.F90:140-148)
      call mckpp_hip_check(mckpp_hip_status(h1, c_loc(st), c_loc(nflag), c_loc(np)), 'mckpp_hip_status (ocnstep)')
      if (st(1) /= 0) then
        zero_pivot = .false.
        call mckpp_hip_column_messages(st(1), np(1), int(ntime), kpp_1d_fields%dlat, kpp_1d_fields%dlon, &
                                       int(kpp_1d_fields%point), .true., kpp_1d_fields%hmix, kpp_1d_fields%kmix, zero_pivot)
        if (zero_pivot .and. mckpp_hip_abort_on_zero_pivot) error stop 1
      end if
    end if
  end subroutine mckpp_physics_ocnstep

end module mckpp_physics_ocnstep_mod
