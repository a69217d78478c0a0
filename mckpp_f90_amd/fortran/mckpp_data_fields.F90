!> Column-model state: the reference's derived types, restricted to what the
!! column-physics path reads or writes.  Type, component and module-variable
!! names and array shapes are the reference's (src/mckpp_data_fields.F90:8-101
!! kpp_3d_type, :187-346 kpp_const_type, allocation extents :353-447, 492-501)
!! so that `kpp_3d_fields%U(ipt,k,1)` means the same thing here.  Fortran owns
!! the storage; the HIP library only ever sees c_loc pointers to it.
module mckpp_data_fields
  use iso_c_binding, only: c_double, c_int
  use mckpp_parameters
  implicit none
  private :: c_double, c_int

  type kpp_3d_type
    ! prognostic profiles and their two saved time levels
    real(c_double), allocatable :: U(:,:,:), X(:,:,:), Us(:,:,:,:), Xs(:,:,:,:), U_init(:,:,:)
    real(c_double), allocatable :: hmixd(:,:)
    ! per-column scalars
    real(c_double), allocatable :: f(:), ocdepth(:), Sref(:), SSref(:), Ssurf(:)
    real(c_double), allocatable :: hmix(:), kmix(:), Tref(:), uref(:), vref(:)
    real(c_double), allocatable :: reset_flag(:), dampu_flag(:), dampv_flag(:), freeze_flag(:)
    real(c_double), allocatable :: dlat(:), dlon(:)
    real(c_double), allocatable :: sflux(:,:,:,:)
    ! surface forcing fields consumed by mckpp_fluxes (reference :76-83)
    real(c_double), allocatable :: taux(:), tauy(:), swf(:), lwf(:), lhf(:), shf(:), rain(:), snow(:)
    integer(c_int), allocatable :: old(:), new(:), jerlov(:)
    logical(c_int), allocatable :: l_ocean(:), l_initflag(:), run_physics(:)
    ! optional relaxation / flux-correction inputs and outputs (allocate the ones a switch needs:
    ! mckpp_allocate_3d_optional); same names and shapes as the reference
    real(c_double), allocatable :: relax_sst(:), SST0(:), fcorr_twod(:), relax_sal(:), relax_ocnT(:), fcorr(:)
    real(c_double), allocatable :: bottom_temp(:)   ! reference :68, read when L_VARY_BOTTOM_TEMP
    real(c_double), allocatable :: fcorr_withz(:,:), sfcorr_withz(:,:), ocnT_clim(:,:), sal_clim(:,:)
    real(c_double), allocatable :: tinc_fcorr(:,:), sinc_fcorr(:,:), ocnTcorr(:,:), scorr(:,:), advection(:,:,:)
    integer(c_int), allocatable :: nmodeadv(:,:), modeadv(:,:,:)
    ! what the last vmix / ocnint pass of a step leaves behind
    real(c_double), allocatable :: rho(:,:), cp(:,:), buoy(:,:)
    real(c_double), allocatable :: difm(:,:), difs(:,:), dift(:,:), ghat(:,:)
    real(c_double), allocatable :: wU(:,:,:), wX(:,:,:), wXNT(:,:,:)
    real(c_double), allocatable :: Rig(:,:), dbloc(:,:), Shsq(:,:), swfrac(:,:), swdk_opt(:,:)
  end type kpp_3d_type

  !> One column, level index fastest (src/mckpp_data_fields.F90:104-184): the
  !! argument type of mckpp_physics_ocnstep.
  type kpp_1d_type
    real(c_double), allocatable :: U(:,:), X(:,:), Us(:,:,:), Xs(:,:,:), U_init(:,:), hmixd(:)
    real(c_double), allocatable :: sflux(:,:,:)
    real(c_double), allocatable :: rho(:), cp(:), buoy(:), difm(:), difs(:), dift(:), ghat(:)
    real(c_double), allocatable :: wU(:,:), wX(:,:), wXNT(:,:), Rig(:), dbloc(:), Shsq(:), swfrac(:), swdk_opt(:)
    real(c_double) :: f = 0, ocdepth = -10000, Sref = 0, SSref = 0, Ssurf = 0, hmix = 0, kmix = 0
    real(c_double) :: Tref = 0, uref = 0, vref = 0, reset_flag = 0, dampu_flag = 0, dampv_flag = 0, freeze_flag = 0
    real(c_double) :: dlat = 0, dlon = 0
    integer(c_int) :: old = 0, new = 1, jerlov = 3, point = 1
    logical(c_int) :: l_ocean = .true., l_initflag = .false., comp_flag = .false.
  end type kpp_1d_type

  type kpp_const_type
    real(c_double) :: dto = 3600, grav = 9.816_c_double, vonk = 0.4_c_double, sice = 4, TK0 = 273.15_c_double
    real(c_double) :: EL = 2.5e6_c_double, FL = 334000, FLSN = 334000, dmax = 200, iso_thresh = 0.002_c_double
    real(c_double), allocatable :: zm(:), hm(:), dm(:), wmt(:,:), wst(:,:), tri(:,:,:)
    integer(c_int) :: iso_bot = 2, dt_uvdamp = 360, ndtocn = 1
    logical :: LKPP = .true., LRI = .true., LDD = .false., L_SSref = .true.
    logical :: L_RELAX_SST = .false., L_RELAX_CALCONLY = .false., L_FCORR = .false., L_FCORR_WITHZ = .false.
    logical :: L_SFCORR = .false., L_SFCORR_WITHZ = .false., L_RELAX_SAL = .false., L_RELAX_OCNT = .false.
    logical :: L_NO_FREEZE = .false., L_NO_ISOTHERM = .false., L_DAMP_CURR = .false.
    logical :: L_VARY_BOTTOM_TEMP = .false., L_RESTART = .false., L_STRETCHGRID = .false.
    logical :: L_FLUXDATA = .false., L_REST = .false., L_ADVECT = .false.
    character(len=200) :: ocnT_file = 'none', sal_file = 'none'
  end type kpp_const_type

  type(kpp_3d_type), target, save :: kpp_3d_fields
  type(kpp_const_type), target, save :: kpp_const_fields

contains

  subroutine mckpp_allocate_3d_fields()
    associate (s => kpp_3d_fields)
      allocate (s%U(npts,nzp1,nvel), s%X(npts,nzp1,nsclr), s%U_init(npts,nzp1,nvel))
      allocate (s%Us(npts,nzp1,nvel,0:1), s%Xs(npts,nzp1,nsclr,0:1), s%hmixd(npts,0:1))
      allocate (s%f(npts), s%ocdepth(npts), s%Sref(npts), s%SSref(npts), s%Ssurf(npts))
      allocate (s%hmix(npts), s%kmix(npts), s%Tref(npts), s%uref(npts), s%vref(npts))
      allocate (s%reset_flag(npts), s%dampu_flag(npts), s%dampv_flag(npts), s%freeze_flag(npts))
      allocate (s%dlat(npts), s%dlon(npts), s%sflux(npts,nsflxs,5,0:njdt))
      allocate (s%taux(npts), s%tauy(npts), s%swf(npts), s%lwf(npts), s%lhf(npts), s%shf(npts), s%rain(npts), s%snow(npts))
      s%taux = 0; s%tauy = 0; s%swf = 0; s%lwf = 0; s%lhf = 0; s%shf = 0; s%rain = 0; s%snow = 0
      allocate (s%old(npts), s%new(npts), s%jerlov(npts))
      allocate (s%l_ocean(npts), s%l_initflag(npts), s%run_physics(npts))
      allocate (s%rho(npts,0:nzp1tmax), s%cp(npts,0:nzp1tmax), s%buoy(npts,nzp1tmax))
      allocate (s%difm(npts,0:nztmax), s%difs(npts,0:nztmax), s%dift(npts,0:nztmax), s%ghat(npts,nztmax))
      allocate (s%wU(npts,0:nztmax,nvp1), s%wX(npts,0:nztmax,nsp1), s%wXNT(npts,0:nztmax,nsclr))
      allocate (s%Rig(npts,nzp1), s%dbloc(npts,nz), s%Shsq(npts,nzp1), s%swfrac(npts,nzp1), s%swdk_opt(npts,0:nz))
      s%U = 0; s%X = 0; s%U_init = 0; s%Us = 0; s%Xs = 0; s%hmixd = 0
      s%f = 0; s%ocdepth = -10000; s%Sref = 0; s%SSref = 0; s%Ssurf = 0
      s%hmix = 0; s%kmix = 0; s%Tref = 0; s%uref = 0; s%vref = 0
      s%reset_flag = 0; s%dampu_flag = 0; s%dampv_flag = 0; s%freeze_flag = 0
      s%dlat = 0; s%dlon = 0; s%sflux = 0
      s%old = 0; s%new = 1; s%jerlov = 3
      s%l_ocean = .true.; s%l_initflag = .false.; s%run_physics = .true.
      s%rho = 0; s%cp = 0; s%buoy = 0; s%difm = 0; s%difs = 0; s%dift = 0; s%ghat = 0
      s%wU = 0; s%wX = 0; s%wXNT = 0; s%Rig = 0; s%dbloc = 0; s%Shsq = 0; s%swfrac = 0; s%swdk_opt = 0
    end associate
  end subroutine mckpp_allocate_3d_fields

  !> Components only the optional switches touch (reference extents, data_fields.F90:380-398)
  subroutine mckpp_allocate_3d_optional()
    associate (s => kpp_3d_fields)
      allocate (s%relax_sst(npts), s%SST0(npts), s%fcorr_twod(npts), s%relax_sal(npts), s%relax_ocnT(npts), s%fcorr(npts))
      allocate (s%bottom_temp(npts))
      s%bottom_temp = 0
      allocate (s%fcorr_withz(npts,nzp1), s%sfcorr_withz(npts,nzp1), s%ocnT_clim(npts,nzp1), s%sal_clim(npts,nzp1))
      allocate (s%tinc_fcorr(npts,nzp1), s%sinc_fcorr(npts,nzp1), s%ocnTcorr(npts,nzp1), s%scorr(npts,nzp1))
      allocate (s%nmodeadv(npts,2), s%modeadv(npts,maxmodeadv,2), s%advection(npts,maxmodeadv,2))
      s%relax_sst = 0; s%SST0 = 0; s%fcorr_twod = 0; s%relax_sal = 0; s%relax_ocnT = 0; s%fcorr = 0
      s%fcorr_withz = 0; s%sfcorr_withz = 0; s%ocnT_clim = 0; s%sal_clim = 0
      s%tinc_fcorr = 0; s%sinc_fcorr = 0; s%ocnTcorr = 0; s%scorr = 0
      s%nmodeadv = 0; s%modeadv = 0; s%advection = 0
    end associate
  end subroutine mckpp_allocate_3d_optional

  subroutine mckpp_allocate_1d_fields(q)
    type(kpp_1d_type), intent(inout) :: q
    if (allocated(q%U)) return
    allocate (q%U(nzp1,nvel), q%X(nzp1,nsclr), q%U_init(nzp1,nvel), q%Us(nzp1,nvel,0:1), q%Xs(nzp1,nsclr,0:1))
    allocate (q%hmixd(0:1), q%sflux(nsflxs,5,0:njdt))
    allocate (q%rho(0:nzp1tmax), q%cp(0:nzp1tmax), q%buoy(nzp1tmax))
    allocate (q%difm(0:nztmax), q%difs(0:nztmax), q%dift(0:nztmax), q%ghat(nztmax))
    allocate (q%wU(0:nztmax,nvp1), q%wX(0:nztmax,nsp1), q%wXNT(0:nztmax,nsclr))
    allocate (q%Rig(nzp1), q%dbloc(nz), q%Shsq(nzp1), q%swfrac(nzp1), q%swdk_opt(0:nz))
    q%U = 0; q%X = 0; q%U_init = 0; q%Us = 0; q%Xs = 0; q%hmixd = 0; q%sflux = 0
    q%rho = 0; q%cp = 0; q%buoy = 0; q%difm = 0; q%difs = 0; q%dift = 0; q%ghat = 0
    q%wU = 0; q%wX = 0; q%wXNT = 0; q%Rig = 0; q%dbloc = 0; q%Shsq = 0; q%swfrac = 0; q%swdk_opt = 0
  end subroutine mckpp_allocate_1d_fields

  subroutine mckpp_allocate_const_fields()
    associate (c => kpp_const_fields)
      allocate (c%zm(nzp1), c%hm(nzp1), c%dm(0:nz))
      allocate (c%wmt(0:891,0:49), c%wst(0:891,0:49), c%tri(0:nztmax,0:1,ngrid))
      c%zm = 0; c%hm = 0; c%dm = 0; c%wmt = 0; c%wst = 0; c%tri = 0
    end associate
  end subroutine mckpp_allocate_const_fields

end module mckpp_data_fields
