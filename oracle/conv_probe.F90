! TEST INFRASTRUCTURE - not part of the product path; no reference source involved.
!
! Compiler-convention probe.  The CPU oracle (oracle/mckpp_oracle.c) restates the reference's Fortran
! expressions in C and therefore has to lower a few Fortran constructs the way the reference's compiler
! does under its own flags (-fdefault-real-8, fcm-make-*.cfg:5):
!     x**3, x**4                integer powers      (wscale_mod.F90:92, blmix_mod.F90:95, lookup_mod.F90:44)
!     x**(1./2.), x**(1./3.), x**(1./4.)            (lookup_mod.F90:53-61, blmix_mod.F90:62)
!     unkinded real literals    (1.257, 98.96, 4.e-7, 0.033, 1./3. ...) promoted to double
! This file evaluates exactly those constructs with amdflang and the same flags as oracle/Makefile's
! `ref` target; tests/test_oracle_cpu.py compares the bits with the oracle's C lowering
! ((x*x)*x, ((x*x)*x)*x, sqrt, pow(x, 1./3.), pow(x, 1./4.), double literals).
subroutine conv_probe_powers(n, x, p3, p4, ph, pt, pq) bind(C, name="conv_probe_powers")
  use iso_c_binding, only: c_int, c_double
  implicit none
  integer(c_int), value :: n
  real(c_double), intent(in) :: x(n)
  real(c_double), intent(out) :: p3(n), p4(n), ph(n), pt(n), pq(n)
  integer :: i
  real :: y
  do i = 1, n
     y = x(i)
     p3(i) = y**3
     p4(i) = y**4
     ph(i) = y**(1./2.)
     pt(i) = y**(1./3.)
     pq(i) = y**(1./4.)
  end do
end subroutine conv_probe_powers

subroutine conv_probe_literals(out) bind(C, name="conv_probe_literals")
  use iso_c_binding, only: c_double
  implicit none
  real(c_double), intent(out) :: out(12)
  real :: a
  out(1) = 1.257
  out(2) = 8.380
  out(3) = 98.96
  out(4) = -28.86
  out(5) = 4.e-7
  out(6) = 0.033
  out(7) = 1./3.
  out(8) = 0.04/49.
  out(9) = 1.E-12
  out(10) = 6.536332E-9
  a = 0.1
  out(11) = a
  out(12) = real(kind(a), c_double)   ! 8 under -fdefault-real-8
end subroutine conv_probe_literals

! Round 5: the intrinsics the oracle lowers to libm / C operators.
!   EXP           swfrac_mod.F90:40,77, fluxes_mod.F90:134-135, ddmix_mod.F90:43        -> exp
!   SQRT          verticalmixing_mod.F90:83-85, bldepth_mod.F90:134, ocnstep_mod.F90:221 -> sqrt
!   ABS           bldepth_mod.F90:103,134,146, blmix_mod.F90:75-83                       -> fabs
!   SIGN(a, b)    bldepth_mod.F90:123,196,201, ocnstep_mod.F90:335                       -> copysign(fabs(a), b)
!   MAX / MIN / AMAX1 / AMIN1 with two to four arguments
!                 swfrac_mod.F90:37-38,75-76, bldepth_mod.F90:137,161,175, blmix_mod.F90:92-100,113,137,
!                 rimix_mod.F90:66-72, ddmix_mod.F90:33, verticalmixing_mod.F90:114,120, ocnstep_mod.F90:323
!                                                                                        -> a > b ? a : b, a < b ? a : b, left to right
!   ifix / int / float   blmix_mod.F90:68, wscale_mod.F90:65-77                          -> (int) truncation, (double)
!   x**2          enhance_mod.F90:36-44, verticalmixing_mod.F90:83,134-136, ddmix_mod.F90:34 -> x*x
! Loops over arrays, like swfrac_mod.F90:36-41 (so a vectorised math library, if the flags selected one, would show).
subroutine conv_probe_unary(n, x, e, s, a, q) bind(C, name="conv_probe_unary")
  use iso_c_binding, only: c_int, c_double
  implicit none
  integer(c_int), value :: n
  real(c_double), intent(in) :: x(n)
  real(c_double), intent(out) :: e(n), s(n), a(n), q(n)
  integer :: i
  real :: y
  do i = 1, n
     y = x(i)
     e(i) = exp(y)
     s(i) = sqrt(abs(y))
     a(i) = abs(y)
     q(i) = y**2
  end do
end subroutine conv_probe_unary

! the model's own forms: MAX(z*fact/a, rmin) then EXP (swfrac_mod.F90:74-77), EXP(z/a) (fluxes_mod.F90:134)
subroutine conv_probe_swfrac(n, z, fact, a1, a2, rfac, sw, sk) bind(C, name="conv_probe_swfrac")
  use iso_c_binding, only: c_int, c_double
  implicit none
  integer(c_int), value :: n
  real(c_double), value :: fact, a1, a2, rfac
  real(c_double), intent(in) :: z(n)
  real(c_double), intent(out) :: sw(n), sk(n)
  real, parameter :: rmin = -80.
  real :: r1, r2
  integer :: i
  do i = 1, n
     r1 = MAX(z(i)*fact/a1, rmin)
     r2 = MAX(z(i)*fact/a2, rmin)
     sw(i) = rfac * exp(r1) + (1.-rfac) * exp(r2)
     sk(i) = rfac * EXP( z(i) / a1 ) + ( 1.0 - rfac ) * EXP( z(i) / a2 )
  end do
end subroutine conv_probe_swfrac

subroutine conv_probe_binary(n, a, b, c, d, sg, sh, se, mx, mn, ax, an, m3, m4, x3) bind(C, name="conv_probe_binary")
  use iso_c_binding, only: c_int, c_double
  implicit none
  integer(c_int), value :: n
  real(c_double), intent(in) :: a(n), b(n), c(n), d(n)
  real(c_double), intent(out) :: sg(n), sh(n), se(n), mx(n), mn(n), ax(n), an(n), m3(n), m4(n), x3(n)
  real, parameter :: epsln = 1.e-16
  integer :: i
  do i = 1, n
     sg(i) = SIGN(a(i), b(i))
     sh(i) = 0.5 + SIGN(0.5, b(i))
     se(i) = 0.5 + SIGN(0.5, b(i) + epsln)
     mx(i) = MAX(a(i), b(i))
     mn(i) = MIN(a(i), b(i))
     ax(i) = AMAX1(a(i), b(i))
     an(i) = AMIN1(a(i), b(i))
     m3(i) = MIN(a(i), b(i), c(i))
     m4(i) = MIN(a(i), b(i), c(i), d(i))
     x3(i) = MAX(a(i), b(i), c(i))
  end do
end subroutine conv_probe_binary

subroutine conv_probe_casts(n, x, ifx, itr, icl, fl) bind(C, name="conv_probe_casts")
  use iso_c_binding, only: c_int, c_double
  implicit none
  integer(c_int), value :: n
  real(c_double), intent(in) :: x(n)
  integer(c_int), intent(out) :: ifx(n), itr(n), icl(n)
  real(c_double), intent(out) :: fl(n)
  real, parameter :: epsln = 1.e-20
  integer :: i, iz
  do i = 1, n
     ifx(i) = ifix(x(i) + epsln)
     itr(i) = int(x(i))
     iz = int(x(i))
     iz = min(iz, 890)
     iz = max(iz, 0)
     icl(i) = iz
     fl(i) = x(i) - float(iz)
  end do
end subroutine conv_probe_casts
