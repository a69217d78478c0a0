!> One-column device batch behind the per-column entry points that keep the reference's signatures
!! (mckpp_physics_ocnstep, mckpp_physics_verticalmixing): a private context and a 1 x nzp1 image of
!! kpp_3d_type that a kpp_1d_type is copied into and out of.
module mckpp_hip_onecol
  use iso_c_binding
  use mckpp_parameters
  use mckpp_data_fields, only: kpp_1d_type, kpp_3d_type, kpp_const_type
  use mckpp_hip_binding
  use mckpp_hip_session, only: mckpp_hip_check, mckpp_hip_const_view, mckpp_hip_state_view, mckpp_hip_device
  implicit none
  private
  public :: onecol_attach, onecol_load, onecol_store_state, onecol_store_diag, h1, one

  type(c_ptr), save :: h1 = c_null_ptr
  type(kpp_3d_type), target, save :: one

contains

  subroutine alloc_one()
    if (allocated(one%U)) return
    allocate (one%U(1,nzp1,nvel), one%X(1,nzp1,nsclr), one%U_init(1,nzp1,nvel))
    allocate (one%Us(1,nzp1,nvel,0:1), one%Xs(1,nzp1,nsclr,0:1), one%hmixd(1,0:1))
    allocate (one%f(1), one%ocdepth(1), one%Sref(1), one%SSref(1), one%Ssurf(1))
    allocate (one%hmix(1), one%kmix(1), one%Tref(1), one%uref(1), one%vref(1))
    allocate (one%reset_flag(1), one%dampu_flag(1), one%dampv_flag(1), one%freeze_flag(1))
    allocate (one%dlat(1), one%dlon(1), one%sflux(1,nsflxs,5,0:njdt))
    allocate (one%old(1), one%new(1), one%jerlov(1), one%l_ocean(1), one%l_initflag(1), one%run_physics(1))
    allocate (one%rho(1,0:nzp1tmax), one%cp(1,0:nzp1tmax), one%buoy(1,nzp1tmax))
    allocate (one%difm(1,0:nztmax), one%difs(1,0:nztmax), one%dift(1,0:nztmax), one%ghat(1,nztmax))
    allocate (one%wU(1,0:nztmax,nvp1), one%wX(1,0:nztmax,nsp1), one%wXNT(1,0:nztmax,nsclr))
    allocate (one%Rig(1,nzp1), one%dbloc(1,nz), one%Shsq(1,nzp1), one%swfrac(1,nzp1), one%swdk_opt(1,0:nz))
    one%rho = 0; one%cp = 0; one%buoy = 0; one%difm = 0; one%difs = 0; one%dift = 0; one%ghat = 0
    one%wU = 0; one%wX = 0; one%wXNT = 0; one%Rig = 0; one%dbloc = 0; one%Shsq = 0; one%swfrac = 0; one%swdk_opt = 0
    one%dlat = 0; one%dlon = 0
  end subroutine alloc_one

  !> context for kpp_const_fields (created once)
  subroutine onecol_attach(kpp_const_fields)
    type(kpp_const_type), intent(in), target :: kpp_const_fields
    type(mckpp_const_c) :: c
    call alloc_one()
    if (.not. c_associated(h1)) then
      call mckpp_hip_const_view(kpp_const_fields, c)
      call mckpp_hip_check(mckpp_hip_init(c, mckpp_hip_device, h1), 'mckpp_hip_init (one column)')
    end if
  end subroutine onecol_attach

  !> kpp_1d_fields -> device
  subroutine onecol_load(q, s)
    type(kpp_1d_type), intent(in) :: q
    type(mckpp_state_ptrs_c), intent(out) :: s
    one%U(1,:,:) = q%U; one%X(1,:,:) = q%X; one%U_init(1,:,:) = q%U_init
    one%Us(1,:,:,:) = q%Us; one%Xs(1,:,:,:) = q%Xs; one%hmixd(1,:) = q%hmixd
    one%f(1) = q%f; one%ocdepth(1) = q%ocdepth; one%Sref(1) = q%Sref; one%SSref(1) = q%SSref
    one%Ssurf(1) = q%Ssurf; one%hmix(1) = q%hmix; one%kmix(1) = q%kmix; one%Tref(1) = q%Tref
    one%uref(1) = q%uref; one%vref(1) = q%vref; one%freeze_flag(1) = q%freeze_flag
    one%reset_flag(1) = 0; one%dampu_flag(1) = 0; one%dampv_flag(1) = 0
    one%sflux(1,:,:,:) = q%sflux
    one%old(1) = q%old; one%new(1) = q%new; one%jerlov(1) = q%jerlov
    one%l_ocean(1) = q%l_ocean; one%l_initflag(1) = q%l_initflag; one%run_physics(1) = .true.
    call mckpp_hip_state_view(one, 1, s)   ! optional-physics components stay unallocated -> NULL
    call mckpp_hip_check(mckpp_hip_upload(h1, s), 'mckpp_hip_upload (one column)')
  end subroutine onecol_load

  !> prognostic and saved state, scalars (what mckpp_physics_ocnstep changes)
  subroutine onecol_store_state(q)
    type(kpp_1d_type), intent(inout) :: q
    q%U = one%U(1,:,:); q%X = one%X(1,:,:); q%Us = one%Us(1,:,:,:); q%Xs = one%Xs(1,:,:,:)
    q%hmixd = one%hmixd(1,:); q%hmix = one%hmix(1); q%kmix = one%kmix(1); q%Tref = one%Tref(1)
    q%uref = one%uref(1); q%vref = one%vref(1); q%Ssurf = one%Ssurf(1)
    q%reset_flag = one%reset_flag(1); q%dampu_flag = one%dampu_flag(1); q%dampv_flag = one%dampv_flag(1)
    q%old = one%old(1); q%new = one%new(1)
  end subroutine onecol_store_state

  !> diagnostics of the last vmix
  subroutine onecol_store_diag(q)
    type(kpp_1d_type), intent(inout) :: q
    q%rho = one%rho(1,:); q%cp = one%cp(1,:); q%buoy = one%buoy(1,:)
    q%difm = one%difm(1,:); q%difs = one%difs(1,:); q%dift = one%dift(1,:); q%ghat = one%ghat(1,:)
    q%wU = one%wU(1,:,:); q%wX = one%wX(1,:,:); q%wXNT = one%wXNT(1,:,:)
    q%Rig = one%Rig(1,:); q%dbloc = one%dbloc(1,:); q%Shsq = one%Shsq(1,:)
    q%swfrac = one%swfrac(1,:); q%swdk_opt = one%swdk_opt(1,:)
  end subroutine onecol_store_diag

end module mckpp_hip_onecol
