!> Device session behind the reference's call surface: owns the library handle,
!! builds the C views of kpp_const_fields / kpp_3d_fields, and moves state
!! between the Fortran arrays and HBM.  Errors from the library stop the run
!! with the library's message (the reference's mckpp_abort convention,
!! src/mckpp_abort_mod.F90:14) - there is no host fallback path.
module mckpp_hip_session
  use iso_c_binding
  use mckpp_parameters
  use mckpp_data_fields
  use mckpp_hip_binding
  implicit none
  private
  public :: mckpp_hip_attach, mckpp_hip_push_state, mckpp_hip_pull_state, mckpp_hip_detach
  public :: mckpp_hip_handle, mckpp_hip_check, mckpp_hip_output_mask, mckpp_hip_device
  public :: mckpp_hip_push_ancillaries, mckpp_hip_ancillaries_every_step
  public :: mckpp_hip_multi_handle, mckpp_hip_ndevices, mckpp_hip_device_list, mckpp_hip_gather_field
  public :: mckpp_hip_const_view, mckpp_hip_state_view, l2i
  public :: mckpp_hip_all_set_flux_series, mckpp_hip_all_run_forced, mckpp_hip_all_window_select
  public :: mckpp_hip_all_window_reset, mckpp_hip_all_window_accumulate, mckpp_hip_all_window_fetch
  public :: mckpp_hip_all_save_restart, mckpp_hip_all_load_restart, mckpp_hip_sync_host, mckpp_hip_device_advanced
  public :: mckpp_hip_host_behind
  public :: mckpp_hip_warnings, mckpp_hip_abort_on_zero_pivot, mckpp_hip_report_warnings, mckpp_hip_column_messages

  !> All devices of the run behind one handle (include/mckpp_hip.h, mckpp_hip_multi_*): the columns of
  !! kpp_3d_fields are dealt round-robin over mckpp_hip_ndevices GPUs, HIP devices mckpp_hip_device,
  !! mckpp_hip_device+1, ... unless mckpp_hip_device_list names them.  One Fortran process drives them
  !! all - no MPI - exactly as the reference's single mckpp_physics_driver call covers all npts
  !! (src/mckpp_physics_driver_mod.F90:27-65).  mckpp_hip_handle is shard 0 (the whole run when
  !! mckpp_hip_ndevices = 1).  Whatever covers all npts - the forced run, the output windows, the restart set,
  !! status - must go through the mckpp_hip_all_* wrappers below (the multi handle), never through shard 0 alone.
  type(c_ptr), save :: mckpp_hip_multi_handle = c_null_ptr
  integer(c_int), save :: mckpp_hip_ndevices = 1
  integer(c_int32_t), allocatable, save :: mckpp_hip_device_list(:)
  type(c_ptr), save :: mckpp_hip_handle = c_null_ptr
  !> Field groups mckpp_physics_driver copies back into kpp_3d_fields after every call.  Default MCKPP_F_ALL: the
  !! reference leaves ALL of kpp_3d_fields current after every call (src/mckpp_types_transfer.F90:199-327), and its
  !! time loop reads profiles and diagnostics from it every step (mckpp_output_control / mckpp_restart_control,
  !! src/mckpp_ocean_model_3D.F90:59-66; src/mckpp_xios_io.F90:86-137,413-431), so a drop-in host gets exactly that.
  !! OPT-IN, for a host that knows when it reads what: MCKPP_F_SCALARS keeps the per-step download to the scalar
  !! group (hmix, kmix, Tref, uref, vref, Ssurf, flags: 192 bytes per column) and leaves the rest on the device
  !! until mckpp_hip_sync_host([groups]) is called at an output / restart step (or reduced output fields are fetched
  !! with mckpp_hip_all_window_fetch); mckpp_hip_host_behind() tells which groups of kpp_3d_fields are stale.
  !! mckpp_physics_finalize brings everything back before it detaches.
  integer(c_int), save :: mckpp_hip_output_mask = MCKPP_F_ALL
  integer(c_int), save :: mckpp_hip_device = 0
  logical, save :: resident = .false.
  !> field groups of kpp_3d_fields that are older than the device state (a step ran and did not bring them back)
  integer(c_int), save :: host_behind = 0
  !> The reference's time loop lets mckpp_boundary_update rewrite SST0, the climatologies and the flux
  !! corrections on the host between steps (src/mckpp_ocean_model_3D.F90:51-55).  With this flag on
  !! (default) mckpp_physics_driver re-sends those inputs before every step when an optional switch
  !! reads them, so the device can never run on stale ancillaries; set it .false. and call
  !! mckpp_hip_push_ancillaries at the ndtupd* cadences to save the per-step upload.
  logical, save :: mckpp_hip_ancillaries_every_step = .true.
  !> The reference's located messages.  Its column loop writes a warning to stderr, with the column's longitude,
  !! latitude and point number, when a column iterates beyond itermax+1 passes or exhausts the ten retries of the
  !! instability trap (src/mckpp_physics_ocnstep_mod.F90:184-191, 229-236), and stops the run on a zero pivot of the
  !! tridiagonal solve (src/mckpp_physics_solvers.F90:140-148).  The device reports all of these as per-column status
  !! bits (include/mckpp_hip.h, MCKPP_ST_*); after every mckpp_physics_driver call the status words (8 bytes per
  !! column) come back and mckpp_hip_report_warnings writes the reference's messages for the flagged columns.
  !! .false.: the host reads mckpp_hip_multi_status itself when it wants to.
  logical, save :: mckpp_hip_warnings = .true.
  !> .true. (the reference: CALL MCKPP_ABORT in tridmat): a zero pivot anywhere stops the run after the messages.
  !! .false.: the step's result stands - the pivot replaced by 1.E-12, the statement behind the reference's abort.
  logical, save :: mckpp_hip_abort_on_zero_pivot = .true.

contains

  pure integer(c_int32_t) function l2i(l)
    logical, intent(in) :: l
    l2i = merge(1_c_int32_t, 0_c_int32_t, l)
  end function l2i

  subroutine mckpp_hip_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    if (rc /= 0) then
      write (0, '(4a)') 'MCKPP-HIP ERROR in ', where, ': ', mckpp_hip_error_text()
      error stop 1
    end if
  end subroutine mckpp_hip_check

  subroutine mckpp_hip_const_view(k, c)
    type(kpp_const_type), intent(in), target :: k
    type(mckpp_const_c), intent(out) :: c
    c%nz = nz; c%nztmax = nztmax; c%nsflxs = nsflxs; c%njdt = njdt; c%itermax = itermax
    c%LKPP = l2i(k%LKPP); c%LRI = l2i(k%LRI); c%LDD = l2i(k%LDD); c%L_SSref = l2i(k%L_SSref)
    c%L_RELAX_SST = l2i(k%L_RELAX_SST); c%L_RELAX_CALCONLY = l2i(k%L_RELAX_CALCONLY)
    c%L_FCORR = l2i(k%L_FCORR); c%L_FCORR_WITHZ = l2i(k%L_FCORR_WITHZ)
    c%L_SFCORR = l2i(k%L_SFCORR); c%L_SFCORR_WITHZ = l2i(k%L_SFCORR_WITHZ)
    c%L_RELAX_SAL = l2i(k%L_RELAX_SAL); c%L_RELAX_OCNT = l2i(k%L_RELAX_OCNT)
    c%L_NO_FREEZE = l2i(k%L_NO_FREEZE); c%L_NO_ISOTHERM = l2i(k%L_NO_ISOTHERM); c%L_DAMP_CURR = l2i(k%L_DAMP_CURR)
    ! src/mckpp_physics_overrides.F90:57-58
    c%clim_present = l2i(trim(k%ocnT_file) /= 'none' .and. trim(k%sal_file) /= 'none')
    c%iso_bot = k%iso_bot; c%dt_uvdamp = k%dt_uvdamp
    c%maxmodeadv = maxmodeadv; c%L_ADVECT = l2i(k%L_ADVECT)
    c%hmixtolfrac = hmixtolfrac; c%dto = k%dto; c%grav = k%grav; c%vonk = k%vonk; c%sice = k%sice
    c%iso_thresh = k%iso_thresh
    c%zm = c_loc(k%zm); c%hm = c_loc(k%hm); c%dm = c_loc(k%dm)
    c%tri = c_loc(k%tri); c%wmt = c_loc(k%wmt); c%wst = c_loc(k%wst)
  end subroutine mckpp_hip_const_view

  subroutine mckpp_hip_state_view(f, n, s)
    type(kpp_3d_type), intent(in), target :: f
    integer, intent(in) :: n
    type(mckpp_state_ptrs_c), intent(out) :: s
    s%npts = n
    s%U = c_loc(f%U); s%X = c_loc(f%X); s%Us = c_loc(f%Us); s%Xs = c_loc(f%Xs)
    s%U_init = c_loc(f%U_init); s%hmixd = c_loc(f%hmixd)
    s%f = c_loc(f%f); s%ocdepth = c_loc(f%ocdepth); s%Sref = c_loc(f%Sref); s%SSref = c_loc(f%SSref)
    s%Ssurf = c_loc(f%Ssurf); s%hmix = c_loc(f%hmix); s%kmix = c_loc(f%kmix); s%Tref = c_loc(f%Tref)
    s%uref = c_loc(f%uref); s%vref = c_loc(f%vref)
    s%reset_flag = c_loc(f%reset_flag); s%dampu_flag = c_loc(f%dampu_flag)
    s%dampv_flag = c_loc(f%dampv_flag); s%freeze_flag = c_loc(f%freeze_flag)
    s%sflux = c_loc(f%sflux)
    s%old = c_loc(f%old); s%new_ = c_loc(f%new); s%jerlov = c_loc(f%jerlov)
    s%l_ocean = c_loc(f%l_ocean); s%l_initflag = c_loc(f%l_initflag); s%run_physics = c_loc(f%run_physics)
    s%rho = c_loc(f%rho); s%cp = c_loc(f%cp); s%buoy = c_loc(f%buoy)
    s%difm = c_loc(f%difm); s%difs = c_loc(f%difs); s%dift = c_loc(f%dift)
    s%wU = c_loc(f%wU); s%wX = c_loc(f%wX); s%wXNT = c_loc(f%wXNT); s%ghat = c_loc(f%ghat)
    s%Rig = c_loc(f%Rig); s%Shsq = c_loc(f%Shsq); s%dbloc = c_loc(f%dbloc)
    s%swfrac = c_loc(f%swfrac); s%swdk_opt = c_loc(f%swdk_opt)
    s%relax_sst = opt1(f%relax_sst); s%SST0 = opt1(f%SST0); s%fcorr_twod = opt1(f%fcorr_twod)
    s%relax_sal = opt1(f%relax_sal); s%relax_ocnT = opt1(f%relax_ocnT); s%fcorr = opt1(f%fcorr)
    s%fcorr_withz = opt2(f%fcorr_withz); s%sfcorr_withz = opt2(f%sfcorr_withz)
    s%ocnT_clim = opt2(f%ocnT_clim); s%sal_clim = opt2(f%sal_clim)
    s%tinc_fcorr = opt2(f%tinc_fcorr); s%sinc_fcorr = opt2(f%sinc_fcorr)
    s%ocnTcorr = opt2(f%ocnTcorr); s%scorr = opt2(f%scorr)
    s%nmodeadv = c_null_ptr; s%modeadv = c_null_ptr; s%advection = c_null_ptr
    if (allocated(f%nmodeadv)) s%nmodeadv = c_loc(f%nmodeadv)
    if (allocated(f%modeadv)) s%modeadv = c_loc(f%modeadv)
    if (allocated(f%advection)) s%advection = c_loc(f%advection)
  end subroutine mckpp_hip_state_view

  !> c_loc of an optional (possibly unallocated) component
  function opt1(a) result(p)
    real(c_double), allocatable, target, intent(in) :: a(:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) p = c_loc(a)
  end function opt1
  function opt2(a) result(p)
    real(c_double), allocatable, target, intent(in) :: a(:,:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) p = c_loc(a)
  end function opt2

  !> Create the device context from kpp_const_fields (idempotent).
  subroutine mckpp_hip_attach()
    type(mckpp_const_c) :: c
    integer(c_int32_t), allocatable :: dev(:)
    integer :: i
    if (c_associated(mckpp_hip_multi_handle)) return
    call mckpp_hip_const_view(kpp_const_fields, c)
    if (allocated(mckpp_hip_device_list)) then
      dev = mckpp_hip_device_list
      mckpp_hip_ndevices = size(dev)
    else
      allocate (dev(mckpp_hip_ndevices))
      dev = [(int(mckpp_hip_device + i - 1, c_int32_t), i = 1, mckpp_hip_ndevices)]
    end if
    call mckpp_hip_check(mckpp_hip_multi_init(c, int(mckpp_hip_ndevices, c_int32_t), dev, mckpp_hip_multi_handle), &
                         'mckpp_hip_multi_init')
    mckpp_hip_handle = mckpp_hip_multi_ctx(mckpp_hip_multi_handle, 0_c_int32_t)
  end subroutine mckpp_hip_attach

  !> Optional-physics inputs only (relaxation, corrections, climatologies, advection): host -> HBM.
  subroutine mckpp_hip_push_ancillaries()
    type(mckpp_state_ptrs_c) :: s
    if (.not. resident) return   ! the full upload that is still to come carries them
    call mckpp_hip_state_view(kpp_3d_fields, npts, s)
    call mckpp_hip_check(mckpp_hip_multi_update_ancillaries(mckpp_hip_multi_handle, s), 'mckpp_hip_update_ancillaries')
  end subroutine mckpp_hip_push_ancillaries

  !> kpp_3d_fields -> HBM (once; afterwards the state lives on the device).  A forced re-upload
  !! first brings back every field group the per-step download left on the device, so host copies
  !! that are older than the device state cannot overwrite it.
  subroutine mckpp_hip_push_state(force)
    logical, intent(in), optional :: force
    type(mckpp_state_ptrs_c) :: s
    logical :: doit
    integer(c_int) :: missing
    doit = .not. resident
    if (present(force)) doit = doit .or. force
    if (.not. doit) return
    call mckpp_hip_attach()
    if (resident) then
      missing = host_behind
      if (missing /= 0) call mckpp_hip_pull_state(missing)
    end if
    call mckpp_hip_state_view(kpp_3d_fields, npts, s)
    call mckpp_hip_check(mckpp_hip_multi_upload(mckpp_hip_multi_handle, s), 'mckpp_hip_upload')
    resident = .true.
    host_behind = 0
  end subroutine mckpp_hip_push_state

  !> HBM -> kpp_3d_fields for the selected field groups.
  subroutine mckpp_hip_pull_state(mask)
    integer(c_int), intent(in) :: mask
    type(mckpp_state_ptrs_c) :: s
    if (mask == 0 .or. .not. resident) return
    call mckpp_hip_state_view(kpp_3d_fields, npts, s)
    call mckpp_hip_check(mckpp_hip_multi_download(mckpp_hip_multi_handle, s, int(mask, c_int32_t)), 'mckpp_hip_download')
    host_behind = iand(host_behind, not(mask))
  end subroutine mckpp_hip_pull_state

  !> the device state has moved on (a step, a forced run, a restart load): every field group of kpp_3d_fields is
  !! now older than it until a download brings it back
  subroutine mckpp_hip_device_advanced()
    host_behind = MCKPP_F_ALL
  end subroutine mckpp_hip_device_advanced

  !> Field groups (MCKPP_F_* bits) of kpp_3d_fields that are older than the device state: 0 with the default
  !! output mask; with a reduced mask, what an output or restart writer must mckpp_hip_sync_host first.
  integer(c_int) function mckpp_hip_host_behind()
    mckpp_hip_host_behind = host_behind
  end function mckpp_hip_host_behind

  !> The reference's warnings for the columns the last step flagged (all devices), in point order.  `nt`: the time
  !! step the messages name (ntime).
  subroutine mckpp_hip_report_warnings(nt)
    integer, intent(in) :: nt
    integer(c_int32_t), allocatable :: st(:), np(:)
    integer(c_int64_t) :: nflag
    integer :: ipt
    logical :: zero_pivot, have_scalars
    if (.not. resident) return
    allocate (st(npts), np(npts))
    call mckpp_hip_check(mckpp_hip_multi_status(mckpp_hip_multi_handle, st, nflag, np), 'mckpp_hip_multi_status')
    if (nflag == 0) return
    have_scalars = iand(host_behind, MCKPP_F_SCALARS) == 0
    zero_pivot = .false.
    do ipt = 1, npts
      if (st(ipt) == 0) cycle
      call mckpp_hip_column_messages(st(ipt), np(ipt), nt, kpp_3d_fields%dlat(ipt), kpp_3d_fields%dlon(ipt), ipt, &
                                     have_scalars, kpp_3d_fields%hmix(ipt), kpp_3d_fields%kmix(ipt), zero_pivot)
    end do
    if (zero_pivot .and. mckpp_hip_abort_on_zero_pivot) error stop 1   ! MCKPP_ABORT (src/mckpp_abort_mod.F90:7-16: STOP)
  end subroutine mckpp_hip_report_warnings

  !> One column's messages from its status word.  The values the reference prints beside the location that exist
  !! only inside its iteration (hmixest, the last difference) are not kept per column: the new hmix, kmix and the number
  !! of passes are printed instead (hmix and kmix when the caller's copies are current).
  subroutine mckpp_hip_column_messages(st, np, nt, dlat, dlon, ipt, have_scalars, hmix, kmix, zero_pivot)
    integer(c_int32_t), intent(in) :: st, np
    integer, intent(in) :: nt, ipt
    real(c_double), intent(in) :: dlat, dlon, hmix, kmix
    logical, intent(in) :: have_scalars
    logical, intent(inout) :: zero_pivot
    character(len=200) :: message
    character(len=*), parameter :: ocnstep = 'MCKPP_PHYSICS_OCNSTEP', tridmat = 'MCKPP_PHYSICS_SOLVERS_TRIDMAT'
    if (iand(st, MCKPP_ST_DODGY_OLDNEW) /= 0) then   ! ocnstep_mod.F90:93-102
      write (message, *) 'Dodgy value of old or new at ipt = ', ipt
      call warn(ocnstep, message)
    end if
    if (iand(st, MCKPP_ST_LONG_ITER) /= 0) then   ! :184-191
      write (message, *) 'long iteration at timestep', nt, ' location = (', dlon, ',', dlat, ')'
      call warn(ocnstep, message)
      if (have_scalars) then
        write (message, *) 'hmixnew=', hmix, ', kmixn = ', int(kmix), ', passes = ', np, ', ipt = ', ipt
      else
        write (message, *) 'passes = ', np, ', ipt = ', ipt
      end if
      call warn(ocnstep, message)
    end if
    if (iand(st, MCKPP_ST_FAILED) /= 0) then   ! :229-236
      write (message, *) 'Failed to find a reasonable solution in the semi-implicit integration after ', 10, ' iterations.'
      call warn(ocnstep, message)
      write (message, *) 'At point lat = ', dlat, ' lon =', dlon, ' ipt = ', ipt, ':'
      call warn(ocnstep, message)
    end if
    if (iand(st, MCKPP_ST_ZERO_PIVOT) /= 0) then   ! solvers.F90:140-148 (mckpp_print_error)
      zero_pivot = .true.
      write (0, *) 'Error in '//tridmat//':'
      write (0, *) 'Algorithm for solving tridiag matrix failed.'
      write (message, *) 'bet = 0 at timestep', nt, ' lat = ', dlat, ' lon =', dlon, ' ipt = ', ipt
      write (0, *) trim(adjustl(message))
    end if
  contains
    subroutine warn(routine, msg)   ! mckpp_print_warning, src/mckpp_log_messages.F90:52-63
      character(len=*), intent(in) :: routine, msg
      write (0, *) 'Warning in '//routine//':'
      write (0, *) trim(adjustl(msg))
    end subroutine warn
  end subroutine mckpp_hip_column_messages

  !> Output gather without a full download: field 0 U, 1 V, 2 T, 3 S -> out(npts,nzp1), 4 hmix -> out(npts);
  !! the shards' rows travel over the GPU interconnect to device `root` (0-based shard index) and cross
  !! PCIe once (SURVEY 8(e): the diagnostics gather is the path's only exchange).
  subroutine mckpp_hip_gather_field(field, root, out)
    integer, intent(in) :: field, root
    real(c_double), intent(inout) :: out(*)
    call mckpp_hip_check(mckpp_hip_multi_gather(mckpp_hip_multi_handle, int(field, c_int32_t), int(root, c_int32_t), out), &
                         'mckpp_hip_multi_gather')
  end subroutine mckpp_hip_gather_field

  !> kpp_3d_fields brought up to date with the device for the field groups the per-step download leaves there
  !! (all of them by default): what a host does before it reads profiles or diagnostics, e.g. at an output step.
  subroutine mckpp_hip_sync_host(groups)
    integer(c_int), intent(in), optional :: groups
    integer(c_int) :: g
    g = MCKPP_F_ALL
    if (present(groups)) g = groups
    call mckpp_hip_pull_state(iand(g, host_behind))
  end subroutine mckpp_hip_sync_host

  !> The reference's forced time loop without per-step host traffic (src/mckpp_ocean_model_3D.F90:38-58) on all
  !! devices: `nrec` records of the eight forcing fields, fields(npts, 8, nrec) in the order taux, tauy, swf, lwf,
  !! lhf, shf, rain, snow (record 1 is flux update number rec0 of the run), kept on the devices; then steps
  !! nt_first .. nt_first+nsteps-1 with mckpp_fluxes every ndtocn steps, one launch sequence per device.
  subroutine mckpp_hip_all_set_flux_series(rec0, nrec, fields)
    integer, intent(in) :: rec0, nrec
    real(c_double), intent(in) :: fields(*)
    call mckpp_hip_push_state()
    call mckpp_hip_check(mckpp_hip_multi_set_flux_series(mckpp_hip_multi_handle, int(rec0, c_int), int(nrec, c_int), fields), &
                         'mckpp_hip_multi_set_flux_series')
  end subroutine mckpp_hip_all_set_flux_series

  subroutine mckpp_hip_all_run_forced(nt_first, nsteps, ndtocn)
    integer, intent(in) :: nt_first, nsteps, ndtocn
    call mckpp_hip_push_state()
    call mckpp_hip_check(mckpp_hip_multi_run_forced(mckpp_hip_multi_handle, int(nt_first, c_int), int(nsteps, c_int), &
                         int(ndtocn, c_int), l2i(kpp_const_fields%L_REST), kpp_const_fields%FLSN, kpp_const_fields%EL), &
                         'mckpp_hip_multi_run_forced')
    call mckpp_hip_device_advanced()
  end subroutine mckpp_hip_all_run_forced

  !> Output windows on the devices (what XIOS does with the fields mckpp_xios_output_control sends,
  !! src/mckpp_xios_io.F90:74-210, run/iodef.xml:88-157): select the MCKPP_OUT_* fields, accumulate once after
  !! each step, fetch op 0 mean / 1 min / 2 max / 3 instant into out(npts[,nzp1]).
  subroutine mckpp_hip_all_window_select(fields)
    integer(c_int32_t), intent(in) :: fields(:)
    call mckpp_hip_attach()
    call mckpp_hip_check(mckpp_hip_multi_window_select(mckpp_hip_multi_handle, fields, int(size(fields), c_int32_t)), &
                         'mckpp_hip_multi_window_select')
  end subroutine mckpp_hip_all_window_select
  subroutine mckpp_hip_all_window_reset()
    call mckpp_hip_check(mckpp_hip_multi_window_reset(mckpp_hip_multi_handle), 'mckpp_hip_multi_window_reset')
  end subroutine mckpp_hip_all_window_reset
  subroutine mckpp_hip_all_window_accumulate()
    call mckpp_hip_check(mckpp_hip_multi_window_accumulate(mckpp_hip_multi_handle), 'mckpp_hip_multi_window_accumulate')
  end subroutine mckpp_hip_all_window_accumulate
  subroutine mckpp_hip_all_window_fetch(field, op, out)
    integer, intent(in) :: field, op
    real(c_double), intent(inout) :: out(*)
    call mckpp_hip_check(mckpp_hip_multi_window_fetch(mckpp_hip_multi_handle, int(field, c_int), int(op, c_int), out), &
                         'mckpp_hip_multi_window_fetch')
  end subroutine mckpp_hip_all_window_fetch

  !> Restart set of all devices (src/mckpp_xios_io.F90:368-465): one file per shard, <path>.<shard>of<ndevices>
  subroutine mckpp_hip_all_save_restart(path)
    character(len=*), intent(in) :: path
    call mckpp_hip_check(mckpp_hip_multi_save_restart(mckpp_hip_multi_handle, trim(path)//c_null_char), 'mckpp_hip_multi_save_restart')
  end subroutine mckpp_hip_all_save_restart
  subroutine mckpp_hip_all_load_restart(path)
    character(len=*), intent(in) :: path
    call mckpp_hip_push_state()   ! the shards' column maps come from the upload
    call mckpp_hip_check(mckpp_hip_multi_load_restart(mckpp_hip_multi_handle, trim(path)//c_null_char), 'mckpp_hip_multi_load_restart')
    call mckpp_hip_device_advanced()
  end subroutine mckpp_hip_all_load_restart

  !> Everything the per-step download left on the devices comes back first, then the handle goes (the arrays of
  !! kpp_3d_fields the library pinned are released with it).
  subroutine mckpp_hip_detach()
    integer(c_int) :: rc
    if (resident) call mckpp_hip_sync_host()
    if (c_associated(mckpp_hip_multi_handle)) rc = mckpp_hip_multi_finalize(mckpp_hip_multi_handle)
    mckpp_hip_multi_handle = c_null_ptr
    mckpp_hip_handle = c_null_ptr
    resident = .false.
    host_behind = 0
  end subroutine mckpp_hip_detach

end module mckpp_hip_session
