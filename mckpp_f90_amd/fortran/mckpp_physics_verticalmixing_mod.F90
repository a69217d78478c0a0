!> mckpp_physics_verticalmixing with the reference's signature
!! (src/mckpp_physics_verticalmixing_mod.F90:14; also called by
!! src/mckpp_initialize_ocean.F90:60): one KPP vertical-mixing evaluation for one column -
!! equation of state, surface fluxes, kppmix (rimix, ddmix, bldepth, blmix, enhance) - on the
!! device kernel's vmix-only mode.  Returns the boundary-layer depth and index in hmixn / kmixn
!! and leaves rho, cp, buoy, Rig, dbloc, Shsq, difm, difs, dift, ghat, wU(0,:), wX(0,:), wXNT in
!! kpp_1d_fields; uref / vref come back as the reference leaves them (scratch, :115-125).
!! U, X and the saved time levels are not touched.
module mckpp_physics_verticalmixing_mod
  use iso_c_binding
  use mckpp_data_fields, only: kpp_1d_type, kpp_const_type
  use mckpp_hip_binding
  use mckpp_hip_session, only: mckpp_hip_check
  use mckpp_hip_onecol
  use mckpp_time_control, only: ntime
  implicit none
  private
  public :: mckpp_physics_verticalmixing

contains

  subroutine mckpp_physics_verticalmixing(kpp_1d_fields, kpp_const_fields, hmixn, kmixn)
    type(kpp_1d_type), intent(inout) :: kpp_1d_fields
    type(kpp_const_type), intent(in), target :: kpp_const_fields
    real(c_double), intent(out) :: hmixn
    integer, intent(out) :: kmixn
    type(mckpp_state_ptrs_c) :: s
    call onecol_attach(kpp_const_fields)
    call onecol_load(kpp_1d_fields, s)
    call mckpp_hip_check(mckpp_hip_vmix_only(h1, int(ntime, c_int)), 'mckpp_hip_vmix_only')
    call mckpp_hip_check(mckpp_hip_download(h1, s, int(ior(MCKPP_F_SCALARS, MCKPP_F_DIAG), c_int32_t)), &
                         'mckpp_hip_download (verticalmixing)')
    hmixn = one%hmix(1)
    kmixn = nint(one%kmix(1))
    kpp_1d_fields%uref = one%uref(1); kpp_1d_fields%vref = one%vref(1)
    call onecol_store_diag(kpp_1d_fields)
  end subroutine mckpp_physics_verticalmixing

end module mckpp_physics_verticalmixing_mod
