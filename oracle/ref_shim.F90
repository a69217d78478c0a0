! TEST INFRASTRUCTURE - not part of the product path.
!
! bind(C) entry points into the two hot-path reference modules that compile
! from their own source file alone (no USE of mckpp_data_fields, hence no
! dependency on the absent netcdf-fortran library):
!   /root/reference/src/mckpp_physics_state_equations.F90      (abk80, cpsw)
!   /root/reference/src/mckpp_physics_verticalmixing_z121_mod.F90 (z121)
! Built only by oracle/Makefile target `ref` into oracle/_ref/ and used by
! tests/ to pin oracle/mckpp_oracle.c bit-for-bit on those functions.
! Everything else on the hot path USEs mckpp_data_fields -> mckpp_netcdf_read
! -> netcdf (not in this image) and is therefore treated as unbuildable here
! (see DESIGN.md, "Oracle pinning").
!
! Compiled with -fdefault-real-8, so REAL == c_double.

subroutine ref_abk80(s, t, p, alpha, beta, kappa, sig0, sig) bind(C, name="ref_abk80")
  use iso_c_binding, only: c_double
  use mckpp_physics_state_equations, only: mckpp_abk80
  implicit none
  real(c_double), value :: s, t, p
  real(c_double), intent(inout) :: alpha, beta, kappa, sig0, sig
  call mckpp_abk80(s, t, p, alpha, beta, kappa, sig0, sig)
end subroutine ref_abk80

function ref_cpsw(s, t, p) bind(C, name="ref_cpsw") result(cp)
  use iso_c_binding, only: c_double
  use mckpp_physics_state_equations, only: mckpp_cpsw
  implicit none
  real(c_double), value :: s, t, p
  real(c_double) :: cp
  cp = mckpp_cpsw(s, t, p)
end function ref_cpsw

! Batched forms so a million-point sweep does not pay ctypes call overhead.
subroutine ref_abk80_batch(n, s, t, p, alpha, beta, sig0, sig) bind(C, name="ref_abk80_batch")
  use iso_c_binding, only: c_double, c_int
  use mckpp_physics_state_equations, only: mckpp_abk80
  implicit none
  integer(c_int), value :: n
  real(c_double), intent(in) :: s(n), t(n), p(n)
  real(c_double), intent(out) :: alpha(n), beta(n), sig0(n), sig(n)
  real(c_double) :: a, b, kap, s0, sg
  integer :: i
  do i = 1, n
     ! the model calls abk80 with alpha, beta non-zero and kappa (exppr) = 0
     ! (reference src/mckpp_physics_verticalmixing_mod.F90:47-61)
     a = 1.0; b = 1.0; kap = 0.0; s0 = 0.0; sg = 0.0
     call mckpp_abk80(s(i), t(i), p(i), a, b, kap, s0, sg)
     alpha(i) = a; beta(i) = b; sig0(i) = s0; sig(i) = sg
  end do
end subroutine ref_abk80_batch

subroutine ref_cpsw_batch(n, s, t, p, cp) bind(C, name="ref_cpsw_batch")
  use iso_c_binding, only: c_double, c_int
  use mckpp_physics_state_equations, only: mckpp_cpsw
  implicit none
  integer(c_int), value :: n
  real(c_double), intent(in) :: s(n), t(n), p(n)
  real(c_double), intent(out) :: cp(n)
  integer :: i
  do i = 1, n
     cp(i) = mckpp_cpsw(s(i), t(i), p(i))
  end do
end subroutine ref_cpsw_batch

subroutine ref_z121(kmp1, vlo, vhi, v, w) bind(C, name="ref_z121")
  use iso_c_binding, only: c_double, c_int
  use mckpp_physics_verticalmixing_z121_mod, only: mckpp_physics_verticalmixing_z121
  implicit none
  integer(c_int), value :: kmp1
  real(c_double), value :: vlo, vhi
  real(c_double), intent(inout) :: v(0:kmp1), w(0:kmp1)
  real(c_double) :: lo, hi
  integer :: k
  k = kmp1; lo = vlo; hi = vhi
  call mckpp_physics_verticalmixing_z121(k, lo, hi, v, w)
end subroutine ref_z121
