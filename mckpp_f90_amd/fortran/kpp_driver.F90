!> Forced column-physics run on the Fortran call surface, shaped like the
!! reference's main program (src/mckpp_ocean_model_3D.F90:23-70) minus file I/O:
!!   dimensions -> allocate -> grid -> lookup -> initial profiles ->
!!   mckpp_initialize_ocean_model -> DO nt: ntime, forcing, mckpp_physics_driver
!! Inputs come from a flat binary file written by the test harness so that the
!! bits are identical to the ones the C-ABI tests use; results go to a second
!! flat file.  Usage: kpp_driver <case.bin> <out.bin>
program kpp_driver
  use iso_c_binding
  use mckpp_parameters
  use mckpp_data_fields
  use mckpp_time_control
  use mckpp_hip_binding, only: MCKPP_F_SCALARS
  use mckpp_physics_lookup_mod, only: mckpp_physics_lookup
  use mckpp_initialize_ocean, only: mckpp_initialize_ocean_model
  use mckpp_physics_driver_mod, only: mckpp_physics_driver, mckpp_physics_finalize
  use mckpp_physics_ocnstep_mod, only: mckpp_physics_ocnstep
  use mckpp_physics_verticalmixing_mod, only: mckpp_physics_verticalmixing
  use mckpp_fluxes_mod, only: mckpp_fluxes
  use mckpp_hip_session, only: mckpp_hip_ndevices, mckpp_hip_device_list, mckpp_hip_gather_field, mckpp_hip_sync_host, &
                               mckpp_hip_output_mask, mckpp_hip_host_behind, &
                               mckpp_hip_all_set_flux_series, mckpp_hip_all_run_forced, mckpp_hip_all_window_select, &
                               mckpp_hip_all_window_reset, mckpp_hip_all_window_accumulate, mckpp_hip_all_window_fetch
  implicit none
  character(len=512) :: fin, fout
  integer :: u, nt, nsteps, ncol, nlev, use_1d, ipt, flags
  integer(c_int) :: hdr(8)
  real(c_double), allocatable :: sf6(:,:), mask(:), series(:,:,:)
  type(kpp_1d_type) :: q
  real(c_double) :: t0, t1
  real(c_double), allocatable :: vm_h(:), vm_k(:), vm_difm(:,:), vm_difs(:,:), vm_dift(:,:), vm_ghat(:,:)
  real(c_double) :: hmixn
  integer :: kmixn

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open (newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read (u) hdr
  ncol = hdr(1); nlev = hdr(2); nsteps = hdr(3); use_1d = hdr(4)
  ! flags: 1 forcing through mckpp_fluxes (constant forcing, L_FLUXDATA=.F.) every step; 2 L_VARY_BOTTOM_TEMP;
  !        4 after the run, mckpp_physics_verticalmixing on every column of the final state (appended to the output)
  !        8 hmix and T also through the output gather (appended); 16 the time loop as ONE forced run from flux
  !        records resident on the devices (constant forcing, the records mckpp_fluxes would assemble each step);
  !        32 the same step by step with an output window: mean hmix and maximum T of the run (appended)
  !        64 opt into the reduced per-step download (scalar group only) + mckpp_hip_sync_host before the output;
  !           without it every call of mckpp_physics_driver leaves all of kpp_3d_fields current, as the reference does
  !        128 itermax = 4 (columns run beyond itermax+1 passes: the reference's located warnings on stderr) with
  !           dlon = 0.5 ipt, dlat = -60 + 0.25 ipt
  flags = hdr(6)
  if (iand(flags, 64) /= 0) mckpp_hip_output_mask = MCKPP_F_SCALARS
  ! hdr(7) > 0: that many device shards; hdr(8) = 1 puts them all on HIP device 0 (one-GPU rehearsal of the
  ! multi-device path), otherwise devices 0 .. hdr(7)-1
  if (hdr(7) > 0) then
    mckpp_hip_ndevices = hdr(7)
    if (hdr(8) == 1) then
      allocate (mckpp_hip_device_list(hdr(7)))
      mckpp_hip_device_list = 0
    end if
  end if
  call mckpp_set_dimensions(ncol, 1, nlev, hdr(5))
  if (iand(flags, 128) /= 0) itermax = 4
  call mckpp_allocate_const_fields()
  call mckpp_allocate_3d_fields()
  read (u) kpp_const_fields%dto
  read (u) kpp_const_fields%zm, kpp_const_fields%hm, kpp_const_fields%dm
  read (u) kpp_3d_fields%U, kpp_3d_fields%X
  read (u) kpp_3d_fields%f, kpp_3d_fields%Sref, kpp_3d_fields%SSref, kpp_3d_fields%Ssurf, kpp_3d_fields%ocdepth
  read (u) kpp_3d_fields%jerlov
  allocate (mask(ncol), sf6(ncol, 6))
  read (u) mask
  read (u) sf6
  close (u)
  kpp_3d_fields%run_physics = mask > 0.5_c_double
  kpp_3d_fields%l_ocean = kpp_3d_fields%run_physics
  kpp_3d_fields%U_init = kpp_3d_fields%U
  if (iand(flags, 128) /= 0) then
    kpp_3d_fields%dlon = [(0.5_c_double * ipt, ipt = 1, ncol)]
    kpp_3d_fields%dlat = [(-60 + 0.25_c_double * ipt, ipt = 1, ncol)]
  end if
  if (iand(flags, 2) /= 0) then
    call mckpp_allocate_3d_optional()
    kpp_const_fields%L_VARY_BOTTOM_TEMP = .true.
    kpp_3d_fields%bottom_temp = kpp_3d_fields%X(:, nzp1, 1) + 0.125_c_double
  end if
  kpp_3d_fields%sflux = 0
  kpp_3d_fields%sflux(:, :, 5, 0) = 1e-20_c_double      ! mckpp_initialize_fluxes, src/mckpp_fluxes_mod.F90:19-32

  call mckpp_physics_lookup(kpp_const_fields)
  ntime = 0
  call mckpp_initialize_ocean_model()

  kpp_3d_fields%sflux(:, 1:6, 5, 0) = sf6
  call cpu_time(t0)
  if (iand(flags, 48) /= 0) then   ! the reference's loop (src/mckpp_ocean_model_3D.F90:38-58) on the devices
    allocate (series(ncol, 8, 1))
    series(:, 1, 1) = 0.01_c_double; series(:, 2, 1) = 0; series(:, 3, 1) = 200; series(:, 4, 1) = 0
    series(:, 5, 1) = -150; series(:, 6, 1) = 0; series(:, 7, 1) = 6e-5_c_double; series(:, 8, 1) = 0
    call mckpp_hip_all_set_flux_series(0, 1, series)
    if (iand(flags, 32) /= 0) then
      call mckpp_hip_all_window_select([4_c_int32_t, 2_c_int32_t])   ! MCKPP_OUT_HMIX, MCKPP_OUT_T
      call mckpp_hip_all_window_reset()
      do nt = 1, nsteps
        call mckpp_hip_all_run_forced(nt, 1, nsteps + 1)
        call mckpp_hip_all_window_accumulate()
      end do
    else
      call mckpp_hip_all_run_forced(1, nsteps, nsteps + 1)   ! one flux update (step 1), as ndtocn > nsteps
    end if
    nsteps = 0
  end if
  do nt = 1, nsteps
    call mckpp_update_time(nt)
    if (iand(flags, 1) /= 0) call mckpp_fluxes()      ! ndtocn = 1 (src/mckpp_ocean_model_3D.F90:44-48)
    if (use_1d == 0) then
      call mckpp_physics_driver()
    else
      ! the reference's inner loop body, one column at a time (physics_driver_mod.F90:46-63)
      do ipt = 1, npts
        if (.not. kpp_3d_fields%run_physics(ipt)) cycle
        call gather_1d(ipt, q)
        call mckpp_physics_ocnstep(q, kpp_const_fields)
        call scatter_1d(ipt, q)
      end do
    end if
  end do
  call cpu_time(t1)
  write (*, '(a,i0,a,i0,a,i0,a,f8.3,a)') 'kpp_driver: ', ncol, ' columns x ', nlev, ' levels, ', nsteps, &
        ' steps, ', t1 - t0, ' s host time'
  write (*, '(a,3es14.6)') 'kpp_driver: hmix min/mean/max ', minval(kpp_3d_fields%hmix, kpp_3d_fields%run_physics), &
        sum(kpp_3d_fields%hmix)/max(1, count(kpp_3d_fields%run_physics)), maxval(kpp_3d_fields%hmix)

  if (iand(flags, 64 + 48) == 0 .and. mckpp_hip_host_behind() /= 0) then   ! the default mask: nothing may be stale
    write (0, '(a,i0)') 'kpp_driver: kpp_3d_fields is behind the device after mckpp_physics_driver: ', mckpp_hip_host_behind()
    error stop 2
  end if
  call mckpp_hip_sync_host()   ! what a reduced per-step download or the forced run left on the devices
  open (newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write (u) kpp_3d_fields%U, kpp_3d_fields%X, kpp_3d_fields%Us, kpp_3d_fields%Xs
  write (u) kpp_3d_fields%hmix, kpp_3d_fields%kmix, kpp_3d_fields%hmixd, kpp_3d_fields%Tref, kpp_3d_fields%Ssurf
  write (u) kpp_3d_fields%old, kpp_3d_fields%new
  write (u) kpp_3d_fields%difm, kpp_3d_fields%ghat, kpp_3d_fields%rho
  if (iand(flags, 8) /= 0) then   ! the output gather (hmix, T) over the device interconnect instead of the download
    allocate (vm_h(ncol), vm_difm(ncol, nzp1))
    vm_h = -1; vm_difm = -1
    call mckpp_hip_gather_field(4, max(0, mckpp_hip_ndevices - 1), vm_h)
    call mckpp_hip_gather_field(2, 0, vm_difm)
    write (u) vm_h, vm_difm
    deallocate (vm_h, vm_difm)
  end if
  if (iand(flags, 32) /= 0) then
    allocate (vm_h(ncol), vm_difm(ncol, nzp1))
    vm_h = -1; vm_difm = -1
    call mckpp_hip_all_window_fetch(4, 0, vm_h)
    call mckpp_hip_all_window_fetch(2, 2, vm_difm)
    write (u) vm_h, vm_difm
    deallocate (vm_h, vm_difm)
  end if
  if (iand(flags, 4) /= 0) then
    allocate (vm_h(ncol), vm_k(ncol), vm_difm(ncol,0:nztmax), vm_difs(ncol,0:nztmax), vm_dift(ncol,0:nztmax), vm_ghat(ncol,nztmax))
    vm_h = 0; vm_k = 0; vm_difm = 0; vm_difs = 0; vm_dift = 0; vm_ghat = 0
    do ipt = 1, npts
      if (.not. kpp_3d_fields%run_physics(ipt)) cycle
      call gather_1d(ipt, q)
      call mckpp_physics_verticalmixing(q, kpp_const_fields, hmixn, kmixn)
      vm_h(ipt) = hmixn; vm_k(ipt) = kmixn
      vm_difm(ipt,:) = q%difm; vm_difs(ipt,:) = q%difs; vm_dift(ipt,:) = q%dift; vm_ghat(ipt,:) = q%ghat
    end do
    write (u) vm_h, vm_k, vm_difm, vm_difs, vm_dift, vm_ghat
  end if
  close (u)
  call mckpp_physics_finalize()

contains

  subroutine gather_1d(i, c)
    integer, intent(in) :: i
    type(kpp_1d_type), intent(inout) :: c
    call mckpp_allocate_1d_fields(c)
    c%U = kpp_3d_fields%U(i,:,:); c%X = kpp_3d_fields%X(i,:,:); c%U_init = kpp_3d_fields%U_init(i,:,:)
    c%Us = kpp_3d_fields%Us(i,:,:,:); c%Xs = kpp_3d_fields%Xs(i,:,:,:); c%hmixd = kpp_3d_fields%hmixd(i,:)
    c%sflux = kpp_3d_fields%sflux(i,:,:,:)
    c%f = kpp_3d_fields%f(i); c%ocdepth = kpp_3d_fields%ocdepth(i); c%Sref = kpp_3d_fields%Sref(i)
    c%SSref = kpp_3d_fields%SSref(i); c%Ssurf = kpp_3d_fields%Ssurf(i)
    c%hmix = kpp_3d_fields%hmix(i); c%kmix = kpp_3d_fields%kmix(i)
    c%old = kpp_3d_fields%old(i); c%new = kpp_3d_fields%new(i); c%jerlov = kpp_3d_fields%jerlov(i)
    c%l_ocean = kpp_3d_fields%l_ocean(i); c%l_initflag = kpp_3d_fields%l_initflag(i); c%point = i
    c%dlat = kpp_3d_fields%dlat(i); c%dlon = kpp_3d_fields%dlon(i)   ! src/mckpp_types_transfer.F90 (what the warnings name)
  end subroutine gather_1d

  subroutine scatter_1d(i, c)
    integer, intent(in) :: i
    type(kpp_1d_type), intent(in) :: c
    kpp_3d_fields%U(i,:,:) = c%U; kpp_3d_fields%X(i,:,:) = c%X
    kpp_3d_fields%Us(i,:,:,:) = c%Us; kpp_3d_fields%Xs(i,:,:,:) = c%Xs; kpp_3d_fields%hmixd(i,:) = c%hmixd
    kpp_3d_fields%hmix(i) = c%hmix; kpp_3d_fields%kmix(i) = c%kmix; kpp_3d_fields%Tref(i) = c%Tref
    kpp_3d_fields%Ssurf(i) = c%Ssurf; kpp_3d_fields%old(i) = c%old; kpp_3d_fields%new(i) = c%new
    kpp_3d_fields%difm(i,:) = c%difm; kpp_3d_fields%ghat(i,:) = c%ghat; kpp_3d_fields%rho(i,:) = c%rho
  end subroutine scatter_1d

end program kpp_driver
