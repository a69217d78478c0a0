// mckpp_colmath.h - device arithmetic of the column kernel and the batch test kernels: equation of
// state, specific heat, turbulent velocity scales, Jerlov transmission.  Every
// function restates one reference routine operation for operation (citations
// at each function) and is compiled with -ffp-contract=off.
#pragma once
#include "mckpp_device.h"
#include "mckpp_math.h"

namespace mckpp_dev {

constexpr int NI = 890, NJ = 48, NT = NI + 2;

__device__ __forceinline__ double dmax2(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ double dmin2(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double dsign(double a, double b) { return __builtin_copysign(__builtin_fabs(a), b); }

// ---------------------------------------------------------------------------
// IEEE-754 correctly rounded fp64 division with the reciprocal refinement
// factored out, so two quotients over one denominator (or a quotient whose
// denominator is known long before its numerator) pay for it once and keep it
// off the dependent chain.  This is the same instruction sequence the compiler
// emits for `n / d` (v_div_scale, v_rcp, 4 fma, mul, fma, v_div_fmas,
// v_div_fixup) split in two; when v_div_scale would rescale either operand
// (never for the magnitudes on this path) it falls back to `n / d`, so the
// result is the correctly rounded quotient in every case.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double rcp_refine(double d)
{
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

__device__ __forceinline__ double div_by_refined(double n, double d, double r)
{
  bool f0, f1;
  const double ds = __builtin_amdgcn_div_scale(n, d, false, &f0);
  const double ns = __builtin_amdgcn_div_scale(n, d, true, &f1);
  if (__builtin_expect(!(ds == d && ns == n), 0)) return n / d;
  const double q = ns * r;
  const double e = __builtin_fma(-d, q, ns);
  const double res = __builtin_amdgcn_div_fmas(e, r, q, f1);
  return __builtin_amdgcn_div_fixup(res, d, n);
}

// The same quotient without the v_div_scale pair and the rescaling select: bit-identical to
// `n / d` whenever v_div_scale would leave both operands alone, i.e. d and 1/d normal, the
// exponents of n and d less than 768 apart, and n either zero or |n| >= 2^-969 (below that the
// residual fma would underflow).  v_div_fixup keeps the IEEE results for zero / infinite / NaN
// operands, signed zeros included.  `r` may be any approximation of 1/d good to 2^-52 relative
// (rcp_refine(d), or the compile-time correctly rounded 1.0/d for a literal d): the corrected
// quotient fma(n - d*q, r, q) then rounds to the correctly rounded n/d, which is what IEEE
// division returns - so every use below stays bit-comparable with the CPU oracle's `/`.
// Used where the operand ranges are known (grid spacings, densities, O(1) physical factors).
__device__ __forceinline__ double div_fast(double n, double d, double r)
{
  const double q = n * r;
  const double e = __builtin_fma(-d, q, n);
  const double res = __builtin_fma(e, r, q);
  return __builtin_amdgcn_div_fixup(res, d, n);
}

// |x| < 2^-961 and x != 0: the numerators div_fast must not see (frexp's exponent of 0 is 0)
__device__ __forceinline__ bool tiny_nonzero(double x) { return __builtin_amdgcn_frexp_exp(x) < -960; }

// div_fast for numerators that can legitimately be tiny non-zero numbers (velocities that have
// diffused down a deep column): those take the full IEEE sequence.
__device__ __forceinline__ double div_fast_guarded(double n, double d, double r)
{
  if (__builtin_expect(tiny_nonzero(n), 0)) return n / d;
  return div_fast(n, d, r);
}

// Specific heat, src/mckpp_physics_state_equations.F90:7-58
__device__ __forceinline__ double cpsw_dev(double S, double T1, double P0)
{
  double T = T1;
  if (T < -2.) T = -2.;
  double P = div_fast(P0, 10., 1. / 10.);
  double SR = __builtin_sqrt(__builtin_fabs(S));
  double A = (-1.38385E-3 * T + 0.1072763) * T - 7.643575;
  double B = (5.148E-5 * T - 4.07718E-3) * T + 0.1770383;
  double C = (((2.093236E-5 * T - 2.654387E-3) * T + 0.1412855) * T - 3.720283) * T + 4217.4;
  double CP0 = (B * SR + A) * S + C;
  A = (((1.7168E-8 * T + 2.0357E-6) * T - 3.13885E-4) * T + 1.45747E-2) * T - 0.49592;
  B = (((2.2956E-11 * T - 4.0027E-9) * T + 2.87533E-7) * T - 1.08645E-5) * T + 2.4931E-4;
  C = ((6.136E-13 * T - 6.5637E-11) * T + 2.6380E-9) * T - 5.422E-8;
  double CP1 = ((C * P + B) * P + A) * P;
  A = (((-2.9179E-10 * T + 2.5941E-8) * T + 9.802E-7) * T - 1.28315E-4) * T + 4.9247E-3;
  B = (3.122E-8 * T - 1.517E-6) * T - 1.2331E-4;
  A = (A + B * SR) * S;
  B = ((1.8448E-11 * T - 2.3905E-9) * T + 1.17054E-7) * T - 2.9558E-6;
  B = (B + 9.971E-8 * SR) * S;
  C = (3.513E-13 * T - 1.7682E-11) * T + 5.540E-10;
  C = (C - 1.4300E-12 * T * SR) * S;
  double CP2 = ((C * P + B) * P + A) * P;
  return CP0 + CP1 + CP2;
}

// mckpp_abk80 as the model calls it (alpha, beta requested, kappa not, P > 0):
// Sig80 :371-476, Bet80 :206-240, Alf80 :244-317 of
// src/mckpp_physics_state_equations.F90.
__device__ __forceinline__ void abk80_dev(double S, double T1, double P, double &Alpha,
                                          double &Beta, double &Sig0)
{
  double T = T1;
  if (T < -2.) T = -2.;
  // Sig80
  double P0 = div_fast(P, 10.0, 1. / 10.0);
  double SR = __builtin_sqrt(__builtin_fabs(S));
  double R1 = ((((6.536332E-9 * T - 1.120083E-6) * T + 1.001685E-4) * T - 9.095290E-3) * T + 6.793952E-2) * T - .157406;
  double R2 = (((5.3875E-9 * T - 8.2467E-7) * T + 7.6438E-5) * T - 4.0899E-3) * T + 8.24493E-1;
  double R3 = (-1.6546E-6 * T + 1.0227E-4) * T - 5.72466E-3;
  double R4 = 4.8314E-4;
  Sig0 = (R4 * S + R3 * SR + R2) * S + R1;
  double Rho0 = 1000.0 + Sig0;
  double B1 = (-5.3009E-4 * T + 1.6483E-2) * T + 7.944E-2;
  double A1 = ((-6.1670E-5 * T + 1.09987E-2) * T - 0.603459) * T + 54.6746;
  double KW = (((-5.155288E-5 * T + 1.360477E-2) * T - 2.327105) * T + 148.4206) * T + 19652.21;
  double K0 = (B1 * SR + A1) * S + KW;
  double E = (9.1697E-10 * T + 2.0816E-8) * T - 9.9348E-7;
  double BW = (5.2787E-8 * T - 6.12293E-6) * T + 8.50935E-5;
  double B = BW + E * S;
  double D = 1.91075E-4;
  double C = (-1.6078E-6 * T - 1.0981E-5) * T + 2.2838E-3;
  double AW = ((-5.77905E-7 * T + 1.16092E-4) * T + 1.43713E-3) * T + 3.239908;
  double A = (D * SR + C) * S + AW;
  double K = (B * P0 + A) * P0 + K0;
  double PK = div_fast(P0, K, rcp_refine(K));
  const double omPK = 1.0 - PK, r_omPK = rcp_refine(omPK);   // three quotients over (1 - PK)
  double Sig = div_fast(1000.0 * PK + Sig0, omPK, r_omPK);
  double Rho = 1000.0 + Sig;
  const double r_Rho = rcp_refine(Rho);                      // two over Rho
  // Bet80
  double SR5 = SR * 1.5;
  double DRho = R2 + SR5 * R3 + (S + S) * R4;
  double DK0 = A1 + SR5 * B1;
  double DA = C + SR5 * D;
  double DB = E;
  double DK = (DB * P0 + DA) * P0 + DK0;
  const double KmP2 = (K - P0) * (K - P0);
  double ABFac = div_fast(Rho0 * P0, KmP2, rcp_refine(KmP2));
  Beta = div_fast(DRho, omPK, r_omPK) - ABFac * DK;
  Beta = div_fast(Beta, Rho, r_Rho);
  // Alf80
  R1 = (((.3268166E-7 * T - .4480332e-5) * T + .3005055e-3) * T - .1819058E-1) * T + 6.793952E-2;
  R2 = ((.215500E-7 * T - .247401E-5) * T + .152876E-3) * T - 4.0899E-3;
  R3 = -.33092E-5 * T + 1.0227E-4;
  double Alph0 = (R3 * SR + R2) * S + R1;
  B1 = -.106018E-2 * T + 1.6483E-2;
  A1 = (-.18501E-3 * T + .219974E-1) * T - 0.603459;
  KW = ((-.2062115E-3 * T + .4081431E-1) * T - .4654210E+1) * T + 148.4206;
  K0 = (B1 * SR + A1) * S + KW;
  E = .183394E-8 * T + 2.0816E-8;
  BW = .105574E-6 * T - 6.12293E-6;
  double AlphB = BW + E * S;
  C = -.32156E-5 * T - 1.0981E-5;
  AW = (-.1733715E-5 * T + .232184E-3) * T + 1.43713E-3;
  double AlphaA = C * S + AW;
  double AlphK = (AlphB * P0 + AlphaA) * P0 + K0;
  Alpha = div_fast(Alph0, omPK, r_omPK) - ABFac * AlphK;
  Alpha = div_fast(-Alpha, Rho, r_Rho);
}

// Sig0 alone: the part of Sig80 (:371-398) that abk80_dev's Sig0 comes from, same operations.  Everything
// else in abk80_dev serves Alpha and Beta.
__device__ __forceinline__ double sig0_dev(double S, double T1)
{
  double T = T1;
  if (T < -2.) T = -2.;
  double SR = __builtin_sqrt(__builtin_fabs(S));
  double R1 = ((((6.536332E-9 * T - 1.120083E-6) * T + 1.001685E-4) * T - 9.095290E-3) * T + 6.793952E-2) * T - .157406;
  double R2 = (((5.3875E-9 * T - 8.2467E-7) * T + 7.6438E-5) * T - 4.0899E-3) * T + 8.24493E-1;
  double R3 = (-1.6546E-6 * T + 1.0227E-4) * T - 5.72466E-3;
  double R4 = 4.8314E-4;
  return (R4 * S + R3 * SR + R2) * S + R1;
}

// Turbulent velocity scales, src/mckpp_physics_verticalmixing_wscale_mod.F90:12-97.
// ustar is constant over a vmix pass, so its table row / fraction are hoisted.
struct wscale_u {
  int ju;
  double ufrac, ustar, ucube;
};

__device__ __forceinline__ wscale_u wscale_prepare(double ustar)
{
  const double umin = 0.0, umax = 0.04;
  const double deltau = (umax - umin) / (NJ + 1);
  wscale_u w;
  double udiff = ustar - umin;
  const double uq = div_fast(udiff, deltau, 1. / deltau);
  int ju = (int)uq;
  ju = ju < NJ ? ju : NJ;
  ju = ju > 0 ? ju : 0;
  w.ju = ju;
  w.ufrac = uq - (double)ju;
  w.ustar = ustar;
  w.ucube = (ustar * ustar) * ustar;
  return w;
}

template <class KP>
__device__ __forceinline__ void wscale_dev(const KP &p, const wscale_u &w, double sigma,
                                           double hbl, double bfsfc, double &wm, double &ws)
{
  const double zmin = -4.e-7, zmax = 0.0, c1 = 5.0;
  const double deltaz = (zmax - zmin) / (NI + 1);
  double zehat = p.vonk * sigma * hbl * bfsfc;
  if (zehat <= zmax) {
    double zdiff = zehat - zmin;
    double q = div_fast(zdiff, deltaz, 1. / deltaz);
    int iz = (int)q;
    iz = iz < NI ? iz : NI;
    iz = iz > 0 ? iz : 0;
    double zfrac = q - (double)iz;
    double fzfrac = 1. - zfrac;
    const auto r0 = p.wtab + 2 * ((size_t)w.ju * NT + iz), r1 = r0 + 2 * NT;   // {wmt, wst} pairs
    const double t00x = r0[0], t00y = r0[1], t10x = r0[2], t10y = r0[3];
    const double t01x = r1[0], t01y = r1[1], t11x = r1[2], t11y = r1[3];
    double wam = (fzfrac)*t01x + zfrac * t11x;
    double wbm = (fzfrac)*t00x + zfrac * t10x;
    wm = (1. - w.ufrac) * wbm + w.ufrac * wam;
    double was = (fzfrac)*t01y + zfrac * t11y;
    double wbs = (fzfrac)*t00y + zfrac * t10y;
    ws = (1. - w.ufrac) * wbs + w.ufrac * was;
  } else {
    const double den = w.ucube + c1 * zehat;
    wm = div_fast(p.vonk * w.ustar * w.ucube, den, rcp_refine(den));
    ws = wm;
  }
}

// The same look-up in two stages, for a caller with several of them in flight: the eight table entries are
// fetched whatever the sign of zehat (the index is clamped into the table either way), so the loads of two
// look-ups can be issued together and their latencies overlap; wscale_finish then does what wscale_dev does.
struct wscale_t {
  double zehat, zfrac;
  double t00x, t00y, t10x, t10y, t01x, t01y, t11x, t11y;
};

template <class KP>
__device__ __forceinline__ wscale_t wscale_fetch(const KP &p, const wscale_u &w, double sigma, double hbl, double bfsfc)
{
  const double zmin = -4.e-7, zmax = 0.0;
  const double deltaz = (zmax - zmin) / (NI + 1);
  wscale_t t;
  t.zehat = p.vonk * sigma * hbl * bfsfc;
  const double zdiff = t.zehat - zmin;
  const double q = div_fast(zdiff, deltaz, 1. / deltaz);
  int iz = (int)q;
  iz = iz < NI ? iz : NI;
  iz = iz > 0 ? iz : 0;
  t.zfrac = q - (double)iz;
  const auto r0 = p.wtab + 2 * ((size_t)w.ju * NT + iz), r1 = r0 + 2 * NT;
  t.t00x = r0[0]; t.t00y = r0[1]; t.t10x = r0[2]; t.t10y = r0[3];
  t.t01x = r1[0]; t.t01y = r1[1]; t.t11x = r1[2]; t.t11y = r1[3];
  return t;
}

template <class KP>
__device__ __forceinline__ void wscale_finish(const KP &p, const wscale_u &w, const wscale_t &t, double &wm, double &ws)
{
  const double zmax = 0.0, c1 = 5.0;
  if (t.zehat <= zmax) {
    const double zfrac = t.zfrac, fzfrac = 1. - zfrac;
    double wam = (fzfrac)*t.t01x + zfrac * t.t11x;
    double wbm = (fzfrac)*t.t00x + zfrac * t.t10x;
    wm = (1. - w.ufrac) * wbm + w.ufrac * wam;
    double was = (fzfrac)*t.t01y + zfrac * t.t11y;
    double wbs = (fzfrac)*t.t00y + zfrac * t.t10y;
    ws = (1. - w.ufrac) * wbs + w.ufrac * was;
  } else {
    const double den = w.ucube + c1 * t.zehat;
    wm = div_fast(p.vonk * w.ustar * w.ucube, den, rcp_refine(den));
    ws = wm;
  }
}

// Jerlov tables, src/mckpp_physics_swfrac_mod.F90:59-61 (constant memory: indexed by a run-time water type)
static __constant__ double jer_rfac_c[6] = {0, 0.58, 0.62, 0.67, 0.77, 0.78};
static __constant__ double jer_a1_c[6] = {0, 0.35, 0.6, 1.0, 1.5, 1.4};
static __constant__ double jer_a2_c[6] = {0, 23.0, 20.0, 17.0, 14.0, 7.9};
static __constant__ double jer_ra1_c[6] = {0, 1. / 0.35, 1. / 0.6, 1. / 1.0, 1. / 1.5, 1. / 1.4};   // correctly rounded 1/a1
static __constant__ double jer_ra2_c[6] = {0, 1. / 23.0, 1. / 20.0, 1. / 17.0, 1. / 14.0, 1. / 7.9};

__device__ __forceinline__ double swfrac_dev(double fact, double z, int jw)
{
  const double *rfac = jer_rfac_c, *a1 = jer_a1_c, *a2 = jer_a2_c;
  const double rmin = -80.;
  double r1 = dmax2(div_fast(z * fact, a1[jw], jer_ra1_c[jw]), rmin);
  double r2 = dmax2(div_fast(z * fact, a2[jw], jer_ra2_c[jw]), rmin);
  return rfac[jw] * mckpp_exp(r1) + (1. - rfac[jw]) * mckpp_exp(r2);
}


}  // namespace mckpp_dev
