!> mckpp_fluxes / mckpp_initialize_fluxes with the reference's interface
!! (src/mckpp_fluxes_mod.F90:19-89), the assembly done on the device.  With
!! L_FLUXDATA=.F. the reference's constant forcing is used (:41-49); otherwise
!! the caller (the reference: mckpp_read_fluxes) has filled kpp_3d_fields%taux..snow.
module mckpp_fluxes_mod
  use iso_c_binding, only: c_int, c_double
  use mckpp_data_fields, only: kpp_3d_fields, kpp_const_fields
  use mckpp_hip_binding
  use mckpp_hip_session
  use mckpp_time_control, only: ntime
  implicit none
contains
  subroutine mckpp_initialize_fluxes()
    kpp_3d_fields%wU = 0; kpp_3d_fields%wX = 0; kpp_3d_fields%wXNT = 0
    kpp_3d_fields%sflux = 0
    kpp_3d_fields%sflux(:, :, 5, 0) = 1e-20_c_double
  end subroutine mckpp_initialize_fluxes

  subroutine mckpp_fluxes()
    associate (s => kpp_3d_fields)
      if (.not. kpp_const_fields%L_FLUXDATA) then
        s%taux = 0.01_c_double; s%tauy = 0; s%swf = 200; s%lwf = 0
        s%lhf = -150; s%shf = 0; s%rain = 6e-5_c_double; s%snow = 0
      end if
      call mckpp_hip_push_state()
      call mckpp_hip_check(mckpp_hip_multi_fluxes(mckpp_hip_multi_handle, int(ntime, c_int), s%taux, s%tauy, s%swf, s%lwf, &
                           s%lhf, s%shf, s%rain, s%snow, l2i(kpp_const_fields%L_REST), kpp_const_fields%FLSN, &
                           kpp_const_fields%EL), 'mckpp_hip_fluxes')
      call mckpp_hip_pull_state(MCKPP_F_SCALARS)   ! sflux(:,1:6,5,0) back for callers that read it
    end associate
  end subroutine mckpp_fluxes
end module mckpp_fluxes_mod
