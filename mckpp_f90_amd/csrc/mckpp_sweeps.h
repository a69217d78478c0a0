// mckpp_sweeps.h - the serial recurrences of a vmix + ocnint pass, shared by the cooperative
// column kernels (mckpp_kernels_wg.hip, mckpp_kernels_mw.hip).
//
// After a workgroup barrier ONE wave runs these for all W column slots of its workgroup with a
// lane per (slot[, system]): the recurrences stay serial and bit-faithful, but their instruction
// stream is paid once per workgroup instead of once per column.  The rows live in the slots'
// LDS blocks: slot s starts at slots + s*SS, row r at + r*NA, element k at + k (reference index).
#ifndef MCKPP_SWEEPS_H
#define MCKPP_SWEEPS_H
#include "mckpp_colmath.h"

#include <type_traits>

namespace mckpp_dev {

// per-slot LDS rows common to both kernels (each NA doubles)
enum { R_DM = 0, R_DT, R_DS, R_GH, R_YU, R_YT, R_YS, R_GM, R_GT, R_GS, R_BETM, R_YV, R_RB, R_COUNT };
// vmix scratch aliases (dead before the Thomas rows are built)
enum { R_U = R_YU, R_V = R_YT, R_B = R_YS, R_R = R_GM, R_DB = R_GT, R_DMO = R_GS, R_T = R_BETM };

// Rib(ku) = MAX(Rib(ku), Rib(ka)+epsln), bldepth_mod.F90:137; W lanes
// (the _n forms take the number of slots and the row to scan at run time: packed-lane kernel)
// Layout of a slot block: element (row a, level i) at a*RS + i*KS doubles.  The row-major kernels pass
// RS = NA, KS = 1; the packed-lane kernel keeps the rows interleaved per level (RS = 1, KS = row count).
// CS is the level stride of the shared grid-constant rows c_t0 / c_t1.
__device__ __forceinline__ void serial_scan_rib_n(int W, int R_SCAN, double *slots, int SS, int RS, int KS, int nz,
                                                  const int *sact, int sact_stride, int lane)
{
  const double epsln16 = 1.e-16;
  if (lane < W && sact[lane * sact_stride]) {
    double *r = slots + lane * SS + R_SCAN * RS;
    double rb = 0.0;
    int k = 2;
    for (; k + 3 <= nz; k += 4) {
      const double a0 = r[k * KS], a1 = r[(k + 1) * KS], a2 = r[(k + 2) * KS], a3 = r[(k + 3) * KS];
      rb = dmax2(a0, rb + epsln16); const double b0 = rb;
      rb = dmax2(a1, rb + epsln16); const double b1 = rb;
      rb = dmax2(a2, rb + epsln16); const double b2 = rb;
      rb = dmax2(a3, rb + epsln16);
      r[k * KS] = b0; r[(k + 1) * KS] = b1; r[(k + 2) * KS] = b2; r[(k + 3) * KS] = rb;
    }
    for (; k <= nz; ++k) {
      rb = dmax2(r[k * KS], rb + epsln16);
      r[k * KS] = rb;
    }
  }
}

template <int W>
__device__ __forceinline__ void serial_scan_rib(double *slots, int SS, int NA, int nz, const int *sact, int lane)
{
  serial_scan_rib_n(W, R_R, slots, SS, NA, 1, nz, sact, 1, lane);
}

// tridcof + tridmat (solvers.F90:14-44, 112-161), skewed by one level: iteration i forms
// gam(i) = cl(i-1)/bet(i-1) and y(i-1) = num(i-1)/bet(i-1) over the same denominator.  3W lanes
__device__ __forceinline__ void serial_thomas_uts_n(int W, double *slots, int SS, int RS, int KS, int CS, int nz,
                                                    const double *c_t0, const double *c_t1, const int *sact,
                                                    int sact_stride, int *sbad, int sbad_stride, int lane)
{
  if (lane < 3 * W) {
    const int sl = lane / 3, sys = lane - 3 * sl;
    if (sact[sl * sact_stride]) {
      double *base = slots + sl * SS;
      const double *d = base + (R_DM + sys) * RS;
      double *y = base + (R_YU + sys) * RS, *gm = base + (R_GM + sys) * RS;
      double *betm = base + R_BETM * RS, *rbm = base + R_RB * RS;
      int bad = 0;
      // The coefficients of tridcof share their products: with p(i) = tri(i,1) diff(i) and q(i) = tri(i,0) diff(i-1)
      //   cl(i) = -p(i), cu(i) = -q(i), cc(i) = (1 + p(i)) + q(i)      (solvers.F90:28-40, same roundings: a
      //   negation is exact), so a level forms p and q once; cu*x is -(q*x) exactly, hence cc - cu*gam = cc + q*gam.
      double dm1 = d[(1) * KS];
      double pm1 = c_t1[(1) * CS] * dm1;   // p(1)
      double bet = 1. + pm1;               // cc(1)
      double ynum = y[(1) * KS];           // y(1) = rhs(1)/bet, formed in the next level's step
      // One level of the skewed sweep.  The serial wave shares its SIMD with busy waves and issues one
      // fp64 instruction every ~6 cycles whether or not it depends on the previous one, so the sweep's time
      // is its instruction count: the common case is one straight basic block (pivot chain bet -> 1/bet ->
      // gam -> bet' interleaved with the solution chain, both on div_fast), and the two conditions that need
      // other arithmetic - a zero pivot, or a tiny non-zero solution numerator that div_fast must not see -
      // are detected at the end of the level before and sent through the slow copy of the step (IEEE
      // sequences), practically never.
      unsigned long long rare = __builtin_amdgcn_ballot_w64(tiny_nonzero(ynum));   // wave mask, lives in SGPRs
      auto level = [&](int i, double di, double t0, double t1, double rhs, auto slow) {
        if (slow.value && bet == 0.) { bad = 1; bet = 1.E-12; }   // solvers.F90:140-151 would stop here
        const double clm1 = -pm1;
        const double q = t0 * dm1;          // -cu(i)
        const double p = t1 * di;           // -cl(i)
        const double cc = (1. + p) + q;
        const double rb = rcp_refine(bet);
        const double g = slow.value ? div_by_refined(clm1, bet, rb) : div_fast(clm1, bet, rb);
        const double yprev = slow.value ? div_by_refined(ynum, bet, rb) : div_fast(ynum, bet, rb);
        if (sys == 0) { betm[(i - 1) * KS] = bet; rbm[(i - 1) * KS] = rb; }
        y[(i - 1) * KS] = yprev;
        gm[(i) * KS] = g;
        bet = cc + q * g;
        ynum = rhs + q * yprev;
        rare = __builtin_amdgcn_ballot_w64(tiny_nonzero(ynum)) | __builtin_amdgcn_ballot_w64(bet == 0.);
        dm1 = di; pm1 = p;
      };
      auto step = [&](int i, double di, double t0, double t1, double rhs) {
        if (__builtin_expect(rare != 0ull, 0)) level(i, di, t0, t1, rhs, std::true_type{});
        else level(i, di, t0, t1, rhs, std::false_type{});
      };
      {   // two levels per trip; each half's operands are fetched while the other half runs
        int i = 2;
        double a_d = d[(2) * KS], a_t0 = c_t0[(2) * CS], a_t1 = c_t1[(2) * CS], a_r = y[(2) * KS];
        for (; i + 1 <= nz; i += 2) {
          const double b_d = d[(i + 1) * KS], b_t0 = c_t0[(i + 1) * CS], b_t1 = c_t1[(i + 1) * CS], b_r = y[(i + 1) * KS];
          step(i, a_d, a_t0, a_t1, a_r);
          if (i + 2 <= nz) { a_d = d[(i + 2) * KS]; a_t0 = c_t0[(i + 2) * CS]; a_t1 = c_t1[(i + 2) * CS]; a_r = y[(i + 2) * KS]; }
          step(i + 1, b_d, b_t0, b_t1, b_r);
        }
        if (i <= nz) step(i, a_d, a_t0, a_t1, a_r);
      }
      if (bet == 0.) { bad = 1; bet = 1.E-12; }
      const double rbl = rcp_refine(bet);
      double yy = div_by_refined(ynum, bet, rbl);
      y[(nz) * KS] = yy;
      if (sys == 0) { betm[(nz) * KS] = bet; rbm[(nz) * KS] = rbl; }
      // back substitution, operands fetched four levels ahead
      int i = nz - 1;
      for (; i >= 4; i -= 4) {
        const double y0 = y[(i) * KS], y1 = y[(i - 1) * KS], y2 = y[(i - 2) * KS], y3 = y[(i - 3) * KS];
        const double g0 = gm[(i + 1) * KS], g1 = gm[(i) * KS], g2 = gm[(i - 1) * KS], g3 = gm[(i - 2) * KS];
        yy = y0 - g0 * yy; const double r0 = yy;
        yy = y1 - g1 * yy; const double r1 = yy;
        yy = y2 - g2 * yy; const double r2 = yy;
        yy = y3 - g3 * yy;
        y[(i) * KS] = r0; y[(i - 1) * KS] = r1; y[(i - 2) * KS] = r2; y[(i - 3) * KS] = yy;
      }
      for (; i >= 1; --i) {
        yy = y[(i) * KS] - gm[(i + 1) * KS] * yy;
        y[(i) * KS] = yy;
      }
      if (bad) sbad[sl * sbad_stride] = 1;
    }
  }
}
template <int W>
__device__ __forceinline__ void serial_thomas_uts(double *slots, int SS, int NA, int nz, const double *c_t0,
                                                  const double *c_t1, const int *sact, int *sbad, int lane)
{
  serial_thomas_uts_n(W, slots, SS, NA, 1, 1, nz, c_t0, c_t1, sact, 1, sbad, 1, lane);
}

// V on the stored momentum factorisation (bet, refined 1/bet, gam); W lanes
__device__ __forceinline__ void serial_thomas_v_n(int W, double *slots, int SS, int RS, int KS, int CS, int nz,
                                                  const double *c_t0, const int *sact, int sact_stride, int lane)
{
  if (lane < W && sact[lane * sact_stride]) {
    double *base = slots + lane * SS;
    const double *d = base + R_DM * RS, *gm = base + R_GM * RS, *betm = base + R_BETM * RS,
                 *rbm = base + R_RB * RS;
    double *y = base + R_YV * RS;
    double yy = div_by_refined(y[(1) * KS], betm[(1) * KS], rbm[(1) * KS]);
    y[(1) * KS] = yy;
    double dm1 = d[(1) * KS];
    // Here the quotient is the dependent chain itself, so it takes div_fast unconditionally; a tiny
    // non-zero numerator is noticed at the end of its level and the quotient is redone (IEEE
    // sequence) at the top of the next one, before anything has used it.  Two levels per trip.
    double nprev = 0.0, bprev = 1.0;
    unsigned long long rare = 0ull;   // wave mask of lanes whose last numerator was tiny
    auto vstep = [&](int i, double rhs, double t0, double b, double r, double di) {
      if (__builtin_expect(rare != 0ull, 0)) {
        if (tiny_nonzero(nprev)) { yy = nprev / bprev; y[(i - 1) * KS] = yy; }
      }
      const double cu = -t0 * dm1;
      const double n = rhs - cu * yy;
      yy = div_fast(n, b, r);
      rare = __builtin_amdgcn_ballot_w64(tiny_nonzero(n));
      y[(i) * KS] = yy;
      nprev = n; bprev = b;
      dm1 = di;
    };
    {
      int i = 2;
      double a_rhs = y[(2) * KS], a_t0 = c_t0[(2) * CS], a_b = betm[(2) * KS], a_r = rbm[(2) * KS], a_d = d[(2) * KS];
      for (; i + 1 <= nz; i += 2) {
        const double b_rhs = y[(i + 1) * KS], b_t0 = c_t0[(i + 1) * CS], b_b = betm[(i + 1) * KS], b_r = rbm[(i + 1) * KS], b_d = d[(i + 1) * KS];
        vstep(i, a_rhs, a_t0, a_b, a_r, a_d);
        if (i + 2 <= nz) { a_rhs = y[(i + 2) * KS]; a_t0 = c_t0[(i + 2) * CS]; a_b = betm[(i + 2) * KS]; a_r = rbm[(i + 2) * KS]; a_d = d[(i + 2) * KS]; }
        vstep(i + 1, b_rhs, b_t0, b_b, b_r, b_d);
      }
      if (i <= nz) vstep(i, a_rhs, a_t0, a_b, a_r, a_d);
    }
    if (__builtin_expect(rare != 0ull, 0)) {
      if (tiny_nonzero(nprev)) { yy = nprev / bprev; y[(nz) * KS] = yy; }
    }
    int i = nz - 1;
    for (; i >= 4; i -= 4) {
      const double y0 = y[(i) * KS], y1 = y[(i - 1) * KS], y2 = y[(i - 2) * KS], y3 = y[(i - 3) * KS];
      const double g0 = gm[(i + 1) * KS], g1 = gm[(i) * KS], g2 = gm[(i - 1) * KS], g3 = gm[(i - 2) * KS];
      yy = y0 - g0 * yy; const double r0 = yy;
      yy = y1 - g1 * yy; const double r1 = yy;
      yy = y2 - g2 * yy; const double r2 = yy;
      yy = y3 - g3 * yy;
      y[(i) * KS] = r0; y[(i - 1) * KS] = r1; y[(i - 2) * KS] = r2; y[(i - 3) * KS] = yy;
    }
    for (; i >= 1; --i) {
      yy = y[(i) * KS] - gm[(i + 1) * KS] * yy;
      y[(i) * KS] = yy;
    }
  }
}

template <int W>
__device__ __forceinline__ void serial_thomas_v(double *slots, int SS, int NA, int nz, const double *c_t0,
                                                const int *sact, int lane)
{
  serial_thomas_v_n(W, slots, SS, NA, 1, 1, nz, c_t0, sact, 1, lane);
}

}  // namespace mckpp_dev
#endif
