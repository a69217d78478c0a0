!> iso_c_binding view of include/mckpp_hip.h: the only place the Fortran host
!! touches the HIP library.  Struct layouts mirror mckpp_const_c and
!! mckpp_state_ptrs_c member for member.
module mckpp_hip_binding
  use iso_c_binding
  implicit none

  ! per-column status bits (include/mckpp_hip.h, MCKPP_ST_*)
  integer(c_int32_t), parameter :: MCKPP_ST_ZERO_PIVOT = 1, MCKPP_ST_LONG_ITER = 2, MCKPP_ST_RETRIED = 4, MCKPP_ST_FAILED = 8, &
                                   MCKPP_ST_DODGY_OLDNEW = 16
  integer(c_int), parameter :: MCKPP_F_PROFILES = 1, MCKPP_F_SAVED = 2, MCKPP_F_SCALARS = 4, MCKPP_F_DIAG = 8
  integer(c_int), parameter :: MCKPP_F_RESTART = 7, MCKPP_F_ALL = 15
  ! MCKPP_OUT_* (XIOS field ids, src/mckpp_xios_io.F90:74-210) and the window operations
  integer(c_int), parameter :: MCKPP_OUT_U = 0, MCKPP_OUT_V = 1, MCKPP_OUT_T = 2, MCKPP_OUT_S_ANOM = 3, MCKPP_OUT_HMIX = 4, &
    MCKPP_OUT_S = 5, MCKPP_OUT_B = 6, MCKPP_OUT_WU = 7, MCKPP_OUT_WV = 8, MCKPP_OUT_WT = 9, MCKPP_OUT_WS = 10, &
    MCKPP_OUT_WB = 11, MCKPP_OUT_WTNT = 12, MCKPP_OUT_DIFM = 13, MCKPP_OUT_DIFT = 14, MCKPP_OUT_DIFS = 15, &
    MCKPP_OUT_RHO = 16, MCKPP_OUT_CP = 17, MCKPP_OUT_SCORR = 18, MCKPP_OUT_RIG = 19, MCKPP_OUT_DBLOC = 20, &
    MCKPP_OUT_SHSQ = 21, MCKPP_OUT_TINC_FCORR = 22, MCKPP_OUT_FCORR_Z = 23, MCKPP_OUT_SINC_FCORR = 24, &
    MCKPP_OUT_FCORR = 25, MCKPP_OUT_TAUX_IN = 26, MCKPP_OUT_TAUY_IN = 27, MCKPP_OUT_SOLAR_IN = 28, &
    MCKPP_OUT_NSOLAR_IN = 29, MCKPP_OUT_PMINUSE_IN = 30, MCKPP_OUT_FREEZE_FLAG = 31, MCKPP_OUT_COMP_FLAG = 32, &
    MCKPP_OUT_DAMPU_FLAG = 33, MCKPP_OUT_DAMPV_FLAG = 34
  integer(c_int), parameter :: MCKPP_OP_MEAN = 0, MCKPP_OP_MIN = 1, MCKPP_OP_MAX = 2, MCKPP_OP_INSTANT = 3

  type, bind(C) :: mckpp_const_c
    integer(c_int32_t) :: nz, nztmax, nsflxs, njdt, itermax
    integer(c_int32_t) :: LKPP, LRI, LDD, L_SSref
    integer(c_int32_t) :: L_RELAX_SST, L_RELAX_CALCONLY, L_FCORR, L_FCORR_WITHZ
    integer(c_int32_t) :: L_SFCORR, L_SFCORR_WITHZ, L_RELAX_SAL, L_RELAX_OCNT
    integer(c_int32_t) :: L_NO_FREEZE, L_NO_ISOTHERM, L_DAMP_CURR
    integer(c_int32_t) :: clim_present
    integer(c_int32_t) :: iso_bot, dt_uvdamp
    integer(c_int32_t) :: maxmodeadv, L_ADVECT
    real(c_double) :: hmixtolfrac, dto, grav, vonk, sice, iso_thresh
    type(c_ptr) :: zm, hm, dm, tri, wmt, wst
  end type mckpp_const_c

  type, bind(C) :: mckpp_state_ptrs_c
    integer(c_int64_t) :: npts
    type(c_ptr) :: U, X, Us, Xs, U_init, hmixd
    type(c_ptr) :: f, ocdepth, Sref, SSref, Ssurf
    type(c_ptr) :: hmix, kmix, Tref, uref, vref
    type(c_ptr) :: reset_flag, dampu_flag, dampv_flag, freeze_flag
    type(c_ptr) :: sflux
    type(c_ptr) :: old, new_, jerlov
    type(c_ptr) :: l_ocean, l_initflag, run_physics
    type(c_ptr) :: rho, cp, buoy, difm, difs, dift, wU, wX, wXNT, ghat, Rig, Shsq, dbloc, swfrac, swdk_opt
    type(c_ptr) :: relax_sst, SST0, fcorr_twod, relax_sal, relax_ocnT, fcorr
    type(c_ptr) :: fcorr_withz, sfcorr_withz, ocnT_clim, sal_clim
    type(c_ptr) :: tinc_fcorr, sinc_fcorr, ocnTcorr, scorr
    type(c_ptr) :: nmodeadv, modeadv, advection
  end type mckpp_state_ptrs_c

  interface
    function mckpp_hip_last_error() bind(C, name="mckpp_hip_last_error") result(p)
      import :: c_ptr
      type(c_ptr) :: p
    end function
    function mckpp_hip_device_count() bind(C, name="mckpp_hip_device_count") result(n)
      import :: c_int
      integer(c_int) :: n
    end function
    function mckpp_hip_init(c, device, handle) bind(C, name="mckpp_hip_init") result(rc)
      import :: c_int, c_ptr, mckpp_const_c
      type(mckpp_const_c), intent(in) :: c
      integer(c_int), value :: device
      type(c_ptr), intent(out) :: handle
      integer(c_int) :: rc
    end function
    function mckpp_hip_finalize(handle) bind(C, name="mckpp_hip_finalize") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int) :: rc
    end function
    subroutine mckpp_host_lookup(vonk, wmt, wst) bind(C, name="mckpp_host_lookup")
      import :: c_double
      real(c_double), value :: vonk
      real(c_double), intent(out) :: wmt(*), wst(*)
    end subroutine
    subroutine mckpp_host_tri(nz, nztmax, dto, zm, hm, tri) bind(C, name="mckpp_host_tri")
      import :: c_double, c_int32_t
      integer(c_int32_t), value :: nz, nztmax
      real(c_double), value :: dto
      real(c_double), intent(in) :: zm(*), hm(*)
      real(c_double), intent(out) :: tri(*)
    end subroutine
    function mckpp_hip_upload(handle, s) bind(C, name="mckpp_hip_upload") result(rc)
      import :: c_int, c_ptr, mckpp_state_ptrs_c
      type(c_ptr), value :: handle
      type(mckpp_state_ptrs_c), intent(in) :: s
      integer(c_int) :: rc
    end function
    function mckpp_hip_update_ancillaries(handle, s) bind(C, name="mckpp_hip_update_ancillaries") result(rc)
      import :: c_int, c_ptr, mckpp_state_ptrs_c
      type(c_ptr), value :: handle
      type(mckpp_state_ptrs_c), intent(in) :: s
      integer(c_int) :: rc
    end function
    function mckpp_hip_set_forcing(handle, sflux) bind(C, name="mckpp_hip_set_forcing") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      real(c_double), intent(in) :: sflux(*)
      integer(c_int) :: rc
    end function
    function mckpp_hip_fluxes(handle, ntime, taux, tauy, swf, lwf, lhf, shf, rain, snow, l_rest, flsn, el) &
        bind(C, name="mckpp_hip_fluxes") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      integer(c_int), value :: ntime, l_rest
      real(c_double), intent(in) :: taux(*), tauy(*), swf(*), lwf(*), lhf(*), shf(*), rain(*), snow(*)
      real(c_double), value :: flsn, el
      integer(c_int) :: rc
    end function
    function mckpp_hip_set_flux_series(handle, rec0, nrec, fields) bind(C, name="mckpp_hip_set_flux_series") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      integer(c_int), value :: rec0, nrec
      real(c_double), intent(in) :: fields(*)
      integer(c_int) :: rc
    end function
    function mckpp_hip_run_forced(handle, nt_first, nsteps, ndtocn, l_rest, flsn, el) &
        bind(C, name="mckpp_hip_run_forced") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      integer(c_int), value :: nt_first, nsteps, ndtocn, l_rest
      real(c_double), value :: flsn, el
      integer(c_int) :: rc
    end function
    function mckpp_hip_bottomtemp(handle, bottom_temp) bind(C, name="mckpp_hip_bottomtemp") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      real(c_double), intent(in) :: bottom_temp(*)
      integer(c_int) :: rc
    end function
    function mckpp_hip_save_restart(handle, path) bind(C, name="mckpp_hip_save_restart") result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: handle
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int) :: rc
    end function
    function mckpp_hip_load_restart(handle, path) bind(C, name="mckpp_hip_load_restart") result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: handle
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int) :: rc
    end function
    function mckpp_hip_set_diagnostics(handle, on) bind(C, name="mckpp_hip_set_diagnostics") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int), value :: on
      integer(c_int) :: rc
    end function
    function mckpp_hip_set_solver_mode(handle, mode) bind(C, name="mckpp_hip_set_solver_mode") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int), value :: mode
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_set_solver_mode(handle, mode) bind(C, name="mckpp_hip_multi_set_solver_mode") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int), value :: mode
      integer(c_int) :: rc
    end function
    function mckpp_hip_init_ocean(handle, ntime) bind(C, name="mckpp_hip_init_ocean") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int), value :: ntime
      integer(c_int) :: rc
    end function
    function mckpp_hip_step(handle, ntime, nsteps) bind(C, name="mckpp_hip_step") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int), value :: ntime, nsteps
      integer(c_int) :: rc
    end function
    ! ---- several GPUs behind one handle (include/mckpp_hip.h, mckpp_hip_multi_*) ----
    function mckpp_hip_multi_init(c, ndev, devices, handle) bind(C, name="mckpp_hip_multi_init") result(rc)
      import :: c_int, c_int32_t, c_ptr, mckpp_const_c
      type(mckpp_const_c), intent(in) :: c
      integer(c_int32_t), value :: ndev
      integer(c_int32_t), intent(in) :: devices(*)
      type(c_ptr), intent(out) :: handle
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_finalize(handle) bind(C, name="mckpp_hip_multi_finalize") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_ctx(handle, shard) bind(C, name="mckpp_hip_multi_ctx") result(h)
      import :: c_int32_t, c_ptr
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: shard
      type(c_ptr) :: h
    end function
    function mckpp_hip_multi_upload(handle, s) bind(C, name="mckpp_hip_multi_upload") result(rc)
      import :: c_int, c_ptr, mckpp_state_ptrs_c
      type(c_ptr), value :: handle
      type(mckpp_state_ptrs_c), intent(in) :: s
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_update_ancillaries(handle, s) bind(C, name="mckpp_hip_multi_update_ancillaries") result(rc)
      import :: c_int, c_ptr, mckpp_state_ptrs_c
      type(c_ptr), value :: handle
      type(mckpp_state_ptrs_c), intent(in) :: s
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_set_forcing(handle, sflux) bind(C, name="mckpp_hip_multi_set_forcing") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      real(c_double), intent(in) :: sflux(*)
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_fluxes(handle, ntime, taux, tauy, swf, lwf, lhf, shf, rain, snow, l_rest, flsn, el) &
        bind(C, name="mckpp_hip_multi_fluxes") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      integer(c_int), value :: ntime, l_rest
      real(c_double), intent(in) :: taux(*), tauy(*), swf(*), lwf(*), lhf(*), shf(*), rain(*), snow(*)
      real(c_double), value :: flsn, el
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_bottomtemp(handle, bottom_temp) bind(C, name="mckpp_hip_multi_bottomtemp") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      real(c_double), intent(in) :: bottom_temp(*)
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_init_ocean(handle, ntime) bind(C, name="mckpp_hip_multi_init_ocean") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int), value :: ntime
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_step(handle, ntime, nsteps) bind(C, name="mckpp_hip_multi_step") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int), value :: ntime, nsteps
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_synchronize(handle) bind(C, name="mckpp_hip_multi_synchronize") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_download(handle, s, mask) bind(C, name="mckpp_hip_multi_download") result(rc)
      import :: c_int, c_int32_t, c_ptr, mckpp_state_ptrs_c
      type(c_ptr), value :: handle
      type(mckpp_state_ptrs_c), intent(in) :: s
      integer(c_int32_t), value :: mask
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_gather(handle, field, root, out) bind(C, name="mckpp_hip_multi_gather") result(rc)
      import :: c_int, c_int32_t, c_ptr, c_double
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: field, root
      real(c_double), intent(inout) :: out(*)
      integer(c_int) :: rc
    end function
    ! ---- the forced time loop, the output windows and the restart set for all shards ----
    function mckpp_hip_multi_set_flux_series(handle, rec0, nrec, fields) bind(C, name="mckpp_hip_multi_set_flux_series") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      integer(c_int), value :: rec0, nrec
      real(c_double), intent(in) :: fields(*)
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_run_forced(handle, nt_first, nsteps, ndtocn, l_rest, flsn, el) &
        bind(C, name="mckpp_hip_multi_run_forced") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      integer(c_int), value :: nt_first, nsteps, ndtocn, l_rest
      real(c_double), value :: flsn, el
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_window_select(handle, fields, nfields) bind(C, name="mckpp_hip_multi_window_select") result(rc)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: handle
      integer(c_int32_t), intent(in) :: fields(*)
      integer(c_int32_t), value :: nfields
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_window_reset(handle) bind(C, name="mckpp_hip_multi_window_reset") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_window_accumulate(handle) bind(C, name="mckpp_hip_multi_window_accumulate") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_window_fetch(handle, field, op, out) bind(C, name="mckpp_hip_multi_window_fetch") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      integer(c_int), value :: field, op
      real(c_double), intent(inout) :: out(*)
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_save_restart(handle, path) bind(C, name="mckpp_hip_multi_save_restart") result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: handle
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_load_restart(handle, path) bind(C, name="mckpp_hip_multi_load_restart") result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: handle
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int) :: rc
    end function
    function mckpp_hip_multi_release_host_arrays(handle) bind(C, name="mckpp_hip_multi_release_host_arrays") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int) :: rc
    end function
    ! ---- output fields with XIOS' temporal operations on the device (MCKPP_OUT_* of include/mckpp_hip.h) ----
    function mckpp_hip_window_select(handle, fields, nfields) bind(C, name="mckpp_hip_window_select") result(rc)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: handle
      integer(c_int32_t), intent(in) :: fields(*)
      integer(c_int32_t), value :: nfields
      integer(c_int) :: rc
    end function
    function mckpp_hip_window_reset(handle) bind(C, name="mckpp_hip_window_reset") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int) :: rc
    end function
    function mckpp_hip_window_accumulate(handle) bind(C, name="mckpp_hip_window_accumulate") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int) :: rc
    end function
    function mckpp_hip_window_fetch(handle, field, op, out) bind(C, name="mckpp_hip_window_fetch") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      integer(c_int), value :: field, op
      real(c_double), intent(inout) :: out(*)
      integer(c_int) :: rc
    end function
    function mckpp_hip_vmix_only(handle, ntime) bind(C, name="mckpp_hip_vmix_only") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int), value :: ntime
      integer(c_int) :: rc
    end function
    function mckpp_hip_synchronize(handle) bind(C, name="mckpp_hip_synchronize") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int) :: rc
    end function
    function mckpp_hip_download(handle, s, mask) bind(C, name="mckpp_hip_download") result(rc)
      import :: c_int, c_ptr, c_int32_t, mckpp_state_ptrs_c
      type(c_ptr), value :: handle
      type(mckpp_state_ptrs_c), intent(inout) :: s
      integer(c_int32_t), value :: mask
      integer(c_int) :: rc
    end function
    function mckpp_hip_status(handle, per_col, n_flagged, npasses) bind(C, name="mckpp_hip_status") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle, per_col, n_flagged, npasses
      integer(c_int) :: rc
    end function
    !> status words and pass counts of the last step for all npts points of all devices (land: 0)
    function mckpp_hip_multi_status(m, per_col, n_flagged, npasses) bind(C, name="mckpp_hip_multi_status") result(rc)
      import :: c_int, c_ptr, c_int32_t, c_int64_t
      type(c_ptr), value :: m
      integer(c_int32_t), intent(out) :: per_col(*), npasses(*)
      integer(c_int64_t), intent(out) :: n_flagged
      integer(c_int) :: rc
    end function
    function mckpp_hip_last_kernel_ms(handle, ms, nlaunch) bind(C, name="mckpp_hip_last_kernel_ms") result(rc)
      import :: c_int, c_ptr, c_double, c_int32_t
      type(c_ptr), value :: handle
      real(c_double), intent(out) :: ms
      integer(c_int32_t), intent(out) :: nlaunch
      integer(c_int) :: rc
    end function
  end interface

contains

  !> Text of the last error as a Fortran string.
  function mckpp_hip_error_text() result(txt)
    character(len=:), allocatable :: txt
    character(kind=c_char), pointer :: p(:)
    type(c_ptr) :: cp
    integer :: n
    cp = mckpp_hip_last_error()
    txt = ''
    if (.not. c_associated(cp)) return
    call c_f_pointer(cp, p, [512])
    n = 0
    do while (n < 512)
      if (p(n+1) == c_null_char) exit
      n = n + 1
    end do
    allocate (character(len=n) :: txt)
    txt = transfer(p(1:n), txt)
  end function mckpp_hip_error_text

end module mckpp_hip_binding
