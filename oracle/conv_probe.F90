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
