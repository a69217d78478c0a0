// mckpp_kernels_wg.hip - cooperative column kernel.
//
// A workgroup of W wavefronts keeps W water columns in flight, one per wave
// ("slot").  The level-parallel physics of a pass runs as in the one-wave
// kernel (lane = grid level), but the serial recurrences of all W columns are
// executed together by one wave with a lane per (column, system):
//   * bulk-Ri running maximum (bldepth_mod.F90:137)      : W lanes
//   * Thomas factorise + sweep for U, T, S (solvers.F90) : 3W lanes
//   * Thomas sweep for V on the stored U factorisation   : W lanes
// so their instruction stream is paid once per W columns instead of once per
// column.  Tridiagonal coefficients are formed inside the sweep from the
// diffusivity rows (tridcof, solvers.F90:14-44), which also removes nine LDS
// rows per column.  Slots run in pass lock-step (six s_barriers per pass) but
// are otherwise independent: each slot carries its own ocnstep iteration state
// and, when its column has converged, stores it and pulls the next column
// index from a global queue (persistent grid), so data-dependent pass counts
// (6..200+) never leave a slot idle.
//
// Register diet: only the twelve profile values per level (iterate, old,
// relaxation memory) plus talpha/sbeta live across phases; everything else
// crosses phases through the slot's LDS rows or its LDS scalar record, and
// per-column inputs are re-read with scalar loads where they are used.
//
// Arithmetic is identical, operation for operation, to k_column_pk
// (mckpp_kernels_pk.hip); the tests require bit-identical results from both.
#include "mckpp_sweeps.h"

#include <cstdio>
#include <type_traits>
#include <cstdlib>

namespace {

using namespace mckpp_dev;

// per-slot LDS rows: mckpp_sweeps.h; rho, cp rows exist only in the optional-physics build
enum { R_RHO = R_COUNT, R_CP, R_COUNT_EXT };

enum { S_EMPTY = 0, S_ACTIVE = 1, S_DONE = 2 };

// per-slot scalar record in LDS (wave-uniform values that cross phases)
enum { C_B0 = 0, C_B0SOL, C_USTAR, C_WU01, C_WU02, C_WX01, C_WX02, C_WXNT0, C_UREFNZ, C_VREFNZ,
       C_HBL, C_F, C_HMIXE, C_HMIXN, C_RHO0CP0, C_RRC /* refined 1/(rho0 cp0) */, C_COUNT = 16 };

#define WG_CONST_DOUBLES(NA_) (6 * (NA_))   // zm, hm, tri0, tri1, rdz (+misc), dtohk rows
template <int LPL>
__host__ __device__ constexpr int wg_na() { return 64 * LPL + 3; }   // == 3 (mod 32): bank-spread rows
template <int LPL, bool EXT = false>
__host__ __device__ constexpr int wg_slot_stride()
{
  int s = (EXT ? R_COUNT_EXT : R_COUNT) * wg_na<LPL>();
  while (s % 32 != 9) ++s;   // slot s, system m at offset 9s+3m (mod 32 doubles): all distinct
  return s;
}

#define WAVE_LDS_SYNC()                                        \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");     \
    __builtin_amdgcn_wave_barrier();                           \
  } while (0)

template <int LPL, int W, int MINW, bool EXT>
__global__ __launch_bounds__(64 * W, MINW) void k_column_wg(const mckpp_kparams *__restrict__ pp, const int ntime)
{
  // the parameter block is read through the scalar cache where it is used: passing its ~60
  // pointers by value pins >100 SGPRs for the whole kernel and turns into v_readlane traffic
  const mckpp_kparams &p = *pp;
  extern __shared__ double lds[];
  constexpr int NA = wg_na<LPL>();
  constexpr int SS = wg_slot_stride<LPL, EXT>();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: LDS row bases stay scalar
  const int nz = p.nz, nzp1 = p.nzp1;
  double *c_zm = lds, *c_hm = lds + NA, *c_t0 = lds + 2 * NA, *c_t1 = lds + 3 * NA;
  // grid-constant quotients and refined reciprocals (div_fast, mckpp_colmath.h), built once per workgroup:
  //   c_rdz[k] ~ 1/(zm(k)-zm(k+1)),  c_dtohk[k] = dto/hm(k),  c_misc = {~1/hm(1), ~1/vonk} in the two unused
  //   tail entries of c_rdz.  (LDS is what limits residency: 5 workgroups of 31.9 KB per CU, allocated in
  //   1280-byte granules, so there is no room for a third table.)
  double *c_rdz = lds + 4 * NA, *c_dtohk = lds + 5 * NA, *c_misc = c_rdz + (NA - 2);
  double *slots = lds + WG_CONST_DOUBLES(NA);
  double *my = slots + wave * SS;
  double *screc = slots + W * SS;   // [W][C_COUNT]
  double *sc = screc + wave * C_COUNT;
  int *sact = reinterpret_cast<int *>(screc + W * C_COUNT);
  int *sbad = sact + W;
  auto row = [&](int a) -> double * { return my + a * NA; };
  double *aDm = row(R_DM), *aDs = row(R_DS), *aDt = row(R_DT), *aGh = row(R_GH);
  double *aU = row(R_U), *aV = row(R_V), *aB = row(R_B), *aR = row(R_R), *aDb = row(R_DB),
         *aDmo = row(R_DMO), *aT = row(R_T);

  for (int i = threadIdx.x; i < NA; i += 64 * W) {
    c_zm[i] = p.zm[i];
    c_hm[i] = p.hm[i];
    c_t0[i] = p.tri0[i];
    c_t1[i] = p.tri1[i];
    if (i < NA - 2) c_rdz[i] = rcp_refine(p.zm[i] - p.zm[i + 1]);
    c_dtohk[i] = p.dto / p.hm[i];
  }
  if (threadIdx.x == 0) { c_misc[0] = rcp_refine(p.hm[1]); c_misc[1] = rcp_refine(p.vonk); }
  if (threadIdx.x < W) { sact[threadIdx.x] = 0; sbad[threadIdx.x] = 0; }
  __syncthreads();

  // ---- per-slot register state -------------------------------------------
  double U[LPL], V[LPL], T[LPL], S[LPL];
  double talpha[LPL], sbeta[LPL];
  double xt[LPL];   // EXT: tinc_fcorr of the latest pass (overrides.F90:87-88 adds to it)
  int kk[LPL];
  bool act[LPL], actz[LPL];
  FORJ {
    int k = lane + 64 * j + 1;
    kk[j] = k;
    act[j] = k <= nzp1;
    actz[j] = k <= nz;
    U[j] = V[j] = T[j] = S[j] = 0.0;
    talpha[j] = sbeta[j] = 0.0;
    xt[j] = 0.0;
  }
  const int lane_nz = (nz - 1) & 63, j_nz = (nz - 1) >> 6;
  const int lane_v1 = nzp1 & 63, j_v1 = nzp1 >> 6;
  const int lane_v2 = (nzp1 + 1) & 63, j_v2 = (nzp1 + 1) >> 6;

  // wave-uniform column state (integers: scalar registers)
  int state = S_EMPTY, col = 0;
  int old = 0, newi = 1, jer = 3, l_initflag = 0, status = 0, npass = 0, npass_try = 0, iconv = 0;
  int comp_flag = 0, kmixn = 0, kbl_pass = 0, nreset = 0;
  const double lambda = 0.5;
  const double epsln16 = 1.e-16, Ricr = 0.30, eps01 = 0.1, cekman = 0.7, cmonob = 1.0, epsln20 = 1.e-20;

  auto bcast_level = [&](const double (&x)[LPL], int src_lane, int src_j) {
    double v = 0;
    FORJ { double a = bcast(x[j], src_lane); if (j == src_j) v = a; }
    return v;
  };
  auto put = [&](int slot, double v) { if (lane == 0) sc[slot] = v; };
  // start-of-step profiles Uo/Xo (ocnstep_mod.F90:82-83) are the column's own U,V,T,S rows, which
  // nothing overwrites before finalize: re-read them (L2) where needed instead of pinning registers
  auto load_old = [&](const double *src, double (&x)[LPL]) {
    FORJ x[j] = act[j] ? src[(size_t)col * p.ld + lane + 64 * j] : 0.0;
  };
  auto old_bottom = [&](const double *src) -> double { return src[(size_t)col * p.ld + (nzp1 - 1)]; };
  auto rowoff = [&]() -> size_t { return (size_t)col * p.ld; };
  auto csrow = [&]() -> double * { return p.cs + (size_t)col * MCKPP_CS; };

  // (re)start the semi-implicit iteration from the saved time levels, ocnstep_mod.F90:91-112
  auto extrapolate = [&]() {
    FORJ {
      size_t o = rowoff() + lane + 64 * j;
      double uo = act[j] ? p.Us[old][o] : 0.0, un = act[j] ? p.Us[newi][o] : 0.0;
      double vo = act[j] ? p.Vs[old][o] : 0.0, vn = act[j] ? p.Vs[newi][o] : 0.0;
      double to = act[j] ? p.Ts[old][o] : 0.0, tn = act[j] ? p.Ts[newi][o] : 0.0;
      double so = act[j] ? p.Ss[old][o] : 0.0, sn = act[j] ? p.Ss[newi][o] : 0.0;
      U[j] = 2. * un - uo;
      V[j] = 2. * vn - vo;
      T[j] = 2. * tn - to;
      S[j] = 2. * sn - so;
      // the relaxation memory equals the new iterate (Ux = U, ocnstep_mod.F90:105,110): seed the
      // solution rows with it so the first under-relaxation returns it unchanged
      if (act[j]) { int k = kk[j]; row(R_YU)[k] = U[j]; row(R_YV)[k] = V[j]; row(R_YT)[k] = T[j]; row(R_YS)[k] = S[j]; }
    }
    WAVE_LDS_SYNC();
    npass_try = 0;
    iconv = 0;
  };

  // pull the next column from the queue and load it
  auto refill = [&]() {
    int c = 0;
    if (lane == 0) c = atomicAdd(p.qhead, 1);
    c = __builtin_amdgcn_readfirstlane(c);
    if (c >= p.ncol) { state = S_DONE; return; }
    col = c;
    state = S_ACTIVE;
    const int *ci = p.ci + (size_t)col * MCKPP_CI;
    old = ci[CI_OLD]; newi = ci[CI_NEW]; jer = ci[CI_JERLOV]; l_initflag = ci[CI_INITFLAG];
    status = 0; npass = 0; comp_flag = 1; nreset = 0;
    if (old < 0 || old > 1) { old = newi; status |= 16; }
    if (newi < 0 || newi > 1) { newi = old; status |= 16; }
    put(C_F, csrow()[CS_F]);
    put(C_WXNT0, 0.0);
    FORJ {
      size_t o = rowoff() + lane + 64 * j;
      U[j] = act[j] ? p.U[o] : 0.0; V[j] = act[j] ? p.V[o] : 0.0;
      T[j] = act[j] ? p.T[o] : 0.0; S[j] = act[j] ? p.S[o] : 0.0;
    }
    if (p.mode == MCKPP_MODE_STEP) extrapolate();
    if (p.mode == MCKPP_MODE_INIT) l_initflag = 1;   // initialize_ocean.F90:59
  };

  // is the coming pass possibly the last one of this column-step?  (diagnostics of the last
  // vmix are what the reference leaves behind, so only such passes need to write them)
  auto maybe_final = [&]() -> bool {
    if (p.mode != MCKPP_MODE_STEP) return true;
    return npass_try >= 3 && (iconv >= 2 || npass_try + 1 >= p.itermax);
  };

  // ---- phase A: EOS, surface fluxes, reference values, rimix, bulk-Ri pieces
  auto phaseA = [&]() {
    const double *cs = csrow();
    const double Sref = cs[CS_SREF];
    if (p.mode == MCKPP_MODE_STEP) {
      // under-relaxation, ocnstep_mod.F90:123-132 / :142-151.  The registers hold the previous
      // iterate (the reference's Ux/Xx), the slot's solution rows hold what ocnint returned.
      const double *yU = row(R_YU), *yV = row(R_YV), *yT = row(R_YT), *yS = row(R_YS);
      FORJ if (act[j]) {
        int k = kk[j];
        U[j] = lambda * U[j] + (1 - lambda) * yU[k];
        V[j] = lambda * V[j] + (1 - lambda) * yV[k];
        T[j] = lambda * T[j] + (1 - lambda) * yT[k];
        S[j] = lambda * S[j] + (1 - lambda) * yS[k];
      }
      WAVE_LDS_SYNC();
    }
    const double zm1 = first_lane(c_zm[1]), zm_kmp1 = first_lane(c_zm[nzp1]);
    double zmk[LPL], rho[LPL], cp[LPL], buoy[LPL];
    const double T1 = first_lane(T[0]);
    FORJ {
      int k = kk[j];
      zmk[j] = c_zm[k];
      double Sin = S[j] + Sref, Tin = T[j], Pin = -zmk[j];
      if (k == nzp1 + 1) { Sin = 0.0; Tin = T1; Pin = -zm1; }
      if (k == nzp1 + 2) { Sin = p.sice; Tin = T1; Pin = -zm1; }
      double al, be, s0;
      abk80_dev(Sin, Tin, Pin, al, be, s0);
      rho[j] = 1000. + s0;
      cp[j] = cpsw_dev(Sin, Tin, Pin);
      talpha[j] = al;
      sbeta[j] = be;
      buoy[j] = div_fast(-p.grav * s0, 1000., 1. / 1000.);
    }
    const double rhoh2o = first_lane(bcast_level(rho, lane_v1, j_v1)), rhob = first_lane(bcast_level(rho, lane_v2, j_v2));
    const double rho0 = first_lane(rho[0]), cp0 = first_lane(cp[0]);
    const double talpha0 = first_lane(talpha[0]), sbeta0 = first_lane(sbeta[0]);
    const double sflux1 = cs[CS_SFLUX1], sflux2 = cs[CS_SFLUX2], sflux3 = cs[CS_SFLUX3],
                 sflux4 = cs[CS_SFLUX4], sflux5 = cs[CS_SFLUX5], sflux6 = cs[CS_SFLUX6];
    const double Ssurf = cs[CS_SSURF];
    const double r_rho0 = rcp_refine(rho0), rho0cp0 = rho0 * cp0, r_rc = rcp_refine(rho0cp0);
    const double wU0_1 = first_lane(div_fast(-sflux1, rho0, r_rho0));   // verticalmixing_mod.F90:81-100
    const double wU0_2 = first_lane(div_fast(-sflux2, rho0, r_rho0));
    const double tau = __builtin_sqrt(sflux1 * sflux1 + sflux2 * sflux2) + 1.e-16;
    const double ustar = first_lane(__builtin_sqrt(div_fast(tau, rho0, r_rho0)));
    const double wX0_1 = first_lane(div_fast(div_fast(-sflux4, rho0, r_rho0), cp0, rcp_refine(cp0)));
    const double wX0_2 = first_lane(div_fast(Ssurf * sflux6, rhoh2o, rcp_refine(rhoh2o)) +
                                    div_fast((Ssurf - p.sice) * sflux5, rhob, rcp_refine(rhob)));
    const double B0 = first_lane(-p.grav * (talpha0 * wX0_1 - sbeta0 * wX0_2));
    const double B0sol = first_lane(div_fast(p.grav * talpha0 * sflux3, rho0cp0, r_rc));
    const wscale_u wu = wscale_prepare_uniform(ustar);
    const double r_vonk = first_lane(c_misc[1]);
    if (lane == 0) {
      sc[C_B0] = B0; sc[C_B0SOL] = B0sol; sc[C_USTAR] = ustar; sc[C_WU01] = wU0_1; sc[C_WU02] = wU0_2;
      sc[C_WX01] = wX0_1; sc[C_WX02] = wX0_2; sc[C_RHO0CP0] = rho0cp0; sc[C_RRC] = r_rc;
      if (ntime >= 1)   // wXNT(0,1), fluxes_mod.F90:110-116
        sc[C_WXNT0] = div_fast(-sflux3 * p.swdk_tab[jer * p.ldc], rho0cp0, r_rc);
    }

    FORJ if (act[j]) { aU[kk[j]] = U[j]; aV[kk[j]] = V[j]; aB[kk[j]] = buoy[j]; }
    double alphaDT[LPL], betaDS[LPL];
    FORJ { alphaDT[j] = 0.0; betaDS[j] = 0.0; }
    if constexpr (EXT) {
      double *aRho = row(R_RHO), *aCp = row(R_CP);
      FORJ if (act[j]) { aRho[kk[j]] = rho[j]; aCp[kk[j]] = cp[j]; }
      if (p.LDD) {   // alphaDT, betaDS across interfaces, verticalmixing_mod.F90:103-108
        double *sA = row(R_YV), *sB = row(R_RB), *sT = row(R_T), *sS = row(R_GH);
        FORJ if (act[j]) { int k = kk[j]; sA[k] = talpha[j]; sB[k] = sbeta[j]; sT[k] = T[j]; sS[k] = S[j]; }
        WAVE_LDS_SYNC();
        FORJ {
          int k = kk[j];
          alphaDT[j] = 0.5 * (talpha[j] + sA[k + 1]) * (T[j] - sT[k + 1]);
          betaDS[j] = 0.5 * (sbeta[j] + sB[k + 1]) * (S[j] - sS[k + 1]);
        }
      }
    }
    WAVE_LDS_SYNC();
    double Ritop[LPL], dVsq[LPL], dbloc[LPL], shsq[LPL], Rig[LPL], zdiff[LPL];
    {   // surface-layer reference values, verticalmixing_mod.F90:111-137
      const double U1 = first_lane(aU[1]), V1 = first_lane(aV[1]), Bu1 = first_lane(aB[1]);
      double zref[LPL], ur[LPL], vr[LPL], br[LPL];
      bool live[LPL];
      double rzref[LPL];
      FORJ {
        zref[j] = eps01 * zmk[j];
        rzref[j] = rcp_refine(zref[j]);
        double wz = dmax2(zm1, zref[j]);
        ur[j] = div_fast_guarded(U1 * wz, zref[j], rzref[j]);
        vr[j] = div_fast_guarded(V1 * wz, zref[j], rzref[j]);
        br[j] = div_fast(Bu1 * wz, zref[j], rzref[j]);
        live[j] = actz[j];
      }
      double zk = zm1, Uk = U1, Vk = V1, Bk = Bu1;
      for (int kl = 1; kl <= nz; ++kl) {
        const double zk1 = c_zm[kl + 1], Uk1 = aU[kl + 1], Vk1 = aV[kl + 1], Bk1 = aB[kl + 1];   // LDS broadcasts
        bool any = false;
        FORJ {
          live[j] = live[j] && !(zref[j] >= zk);
          any = any || live[j];
        }
        if (!__any(any)) break;
        const double dzk = zk - zk1, rdzk = first_lane(c_rdz[kl]);
        FORJ if (live[j]) {
          double wz = dmin2(zk - zk1, zk - zref[j]);
          double del = div_fast(0.5 * wz, dzk, rdzk);
          ur[j] = ur[j] - div_fast_guarded(wz * (Uk + del * (Uk1 - Uk)), zref[j], rzref[j]);
          vr[j] = vr[j] - div_fast_guarded(wz * (Vk + del * (Vk1 - Vk)), zref[j], rzref[j]);
          br[j] = br[j] - div_fast(wz * (Bk + del * (Bk1 - Bk)), zref[j], rzref[j]);
        }
        zk = zk1; Uk = Uk1; Vk = Vk1; Bk = Bk1;
      }
      FORJ {
        int k = kk[j];
        double bk1 = aB[k + 1], uk1 = aU[k + 1], vk1 = aV[k + 1];
        Ritop[j] = (zref[j] - zmk[j]) * (br[j] - buoy[j]);
        dbloc[j] = buoy[j] - bk1;
        dVsq[j] = (ur[j] - U[j]) * (ur[j] - U[j]) + (vr[j] - V[j]) * (vr[j] - V[j]);
        shsq[j] = (U[j] - uk1) * (U[j] - uk1) + (V[j] - vk1) * (V[j] - vk1);
      }
      if (p.mode != MCKPP_MODE_STEP) {
        put(C_UREFNZ, bcast_level(ur, lane_nz, j_nz));
        put(C_VREFNZ, bcast_level(vr, lane_nz, j_nz));
      }
    }
    // rimix + z121 (rimix_mod.F90:13-106, z121_mod.F90:7-45)
    FORJ {
      int k = kk[j];
      zdiff[j] = zmk[j] - c_zm[k + 1];
      const double shs = shsq[j] + 1.e-16;
      Rig[j] = div_fast(dbloc[j] * zdiff[j], shs, rcp_refine(shs));
      if (actz[j]) { aR[k] = Rig[j]; aDb[k] = dbloc[j]; }
      if (k == 1) aR[0] = 0.0;
      if (k == nzp1) aR[k] = 0.0;
    }
    if (p.diag && maybe_final()) {   // what the last vmix leaves behind (types_transfer.F90:199-327)
      FORJ {
        int k = kk[j];
        size_t o = rowoff() + k;
        if (act[j]) { p.rho[o] = rho[j]; p.cp[o] = cp[j]; p.buoy[o] = buoy[j]; p.talpha[o] = talpha[j]; p.sbeta[o] = sbeta[j]; }
        if (actz[j]) { p.Rig[o] = Rig[j]; p.dbloc[o] = dbloc[j]; p.Shsq[o] = shsq[j]; }
        if (k == 1) { p.rho[o - 1] = rho[j]; p.cp[o - 1] = cp[j]; p.talpha[o - 1] = talpha[j]; p.sbeta[o - 1] = sbeta[j]; }
      }
    }
    WAVE_LDS_SYNC();
    double dm_i[LPL], ds_i[LPL];
    FORJ {
      int k = kk[j];
      const double Riinfty = 0.8;
      double vm1 = aR[k - 1], vp1 = aR[k + 1];
      double wm1 = (k - 1 >= 1 && !((vm1 < 0.0) || (vm1 > Riinfty))) ? 1.0 : 0.0;
      double wp1 = (k + 1 <= nz && !((vp1 < 0.0) || (vp1 > Riinfty))) ? 1.0 : 0.0;
      double sm = wm1 * vm1 + 2. * Rig[j] + wp1 * vp1;
      double wait = wm1 + 2.0 + wp1;
      // wait is 2, 3 or 4: exact reciprocals but for 1/3 (correctly rounded literal)
      sm = div_fast(sm, wait, wait == 3.0 ? 1. / 3. : (wait == 2.0 ? 0.5 : 0.25));
      double Rigg = dmax2(sm, 0.0);
      double ratio = dmin2(div_fast(Rigg, Riinfty, 1. / Riinfty), 1.0);
      double fri = (1.0 - ratio * ratio);
      fri = fri * fri * fri;
      dm_i[j] = (0.0001 + fri * 0.005);
      ds_i[j] = (0.00001 + fri * 0.005);
    }
    double dt_i[LPL];
    FORJ dt_i[j] = ds_i[j];   // dift = difs, rimix_mod.F90:95-97
    if constexpr (EXT) {
      if (p.LDD) {   // ddmix_mod.F90:12-52
        const double Rrho0 = 1.9, dsfmax = 1.0e-4;
        FORJ {
          const double aDT = alphaDT[j], bDS = betaDS[j];
          if ((aDT > bDS) && (bDS > 0.)) {
            double Rrho = dmin2(aDT / bDS, Rrho0);
            double rr = ((Rrho - 1) / (Rrho0 - 1));
            double diffdd = 1.0 - rr * rr;
            diffdd = dsfmax * diffdd * diffdd * diffdd;
            dt_i[j] = dt_i[j] + diffdd * 0.8 / Rrho;
            ds_i[j] = ds_i[j] + diffdd;
          } else if ((aDT < 0.0) && (bDS < 0.0) && (aDT < bDS)) {
            double Rrho = aDT / bDS;
            double diffdd = 1.5e-6 * 9.0 * 0.101 * mckpp_exp(4.6 * mckpp_exp(-0.54 * (1 / Rrho - 1)));
            double prandtl = 0.15 * Rrho;
            if (Rrho > 0.5) prandtl = (1.85 - 0.85 / Rrho) * Rrho;
            dt_i[j] = dt_i[j] + diffdd;
            ds_i[j] = ds_i[j] + prandtl * diffdd;
          }
        }
      }
    }
    WAVE_LDS_SYNC();
    FORJ {   // interior diffusivities
      int k = kk[j];
      if (actz[j]) { aDm[k] = dm_i[j]; aDs[k] = ds_i[j]; aDt[k] = dt_i[j]; }
      if (k == nz) { aDm[k + 1] = dm_i[j]; aDs[k + 1] = ds_i[j]; aDt[k + 1] = dt_i[j]; }   // kppmix_mod.F90:82-84
      if (k == 1) { aDm[0] = 0.0; aDs[0] = 0.0; aDt[0] = 0.0; }
    }
    // bldepth, level-parallel part (bldepth_mod.F90:105-147)
    FORJ {
      int k = kk[j];
      double swf = p.swfrac_tab[jer * p.ldc + k];
      double bf = B0 + B0sol * (1. - swf);
      double st = 0.5 + dsign(0.5, bf + epsln16);
      double sg = st * 1. + (1. - st) * eps01;
      double wm, ws;
      wscale_dev(p, wu, sg, -zmk[j], bf, wm, ws);
      double dbm1 = aDb[k - 1];
      double bvsq = 0.5 * (div_fast(dbm1, c_zm[k - 1] - zmk[j], c_rdz[k - 1]) + div_fast(dbloc[j], zdiff[j], c_rdz[k]));
      double Vtsq = -zmk[j] * ws * __builtin_sqrt(__builtin_fabs(bvsq)) * p.Vtc;
      const double rawden = dVsq[j] + Vtsq + epsln16, bfa = __builtin_fabs(bf) + epsln16;
      double raw = div_fast(Ritop[j], rawden, rcp_refine(rawden));
      double dmo = div_fast(div_fast(cmonob * ustar * ustar * ustar, p.vonk, r_vonk), bfa, rcp_refine(bfa));
      dmo = st * dmo - (1. - st) * zm_kmp1;
      if (k >= 2 && actz[j]) { aR[k] = raw; aDmo[k] = dmo; }
      if (k == 1) { aR[1] = 0.0; aDmo[1] = -zm_kmp1; }
    }
  };

  // ---- optional terms of the T and S right-hand sides (ocnint_mod.F90:97-215) -----
  // Level k of this lane: relaxation / flux corrections / prescribed advection (rhsmod,
  // solvers.F90:176-335, salinity only).  Also leaves tinc_fcorr in xt and writes the
  // correction diagnostics when this may be the last pass.
  auto ext_rhs = [&](int k, int j, int kmixe, double To_k, double So_k, double &rhsT, double &rhsS) {
    const double dto = p.dto;
    const double *xs = p.xs + (size_t)col * MCKPP_XS;
    const double *aRho = row(R_RHO), *aCp = row(R_CP);
    const double rhok = aRho[k], cpk = aCp[k];
    const size_t oin = rowoff() + (k - 1);
    if (k == 1) {
      if (p.L_RELAX_SST && !p.L_FCORR_WITHZ && !p.L_FCORR) {   // :97-114
        const double relax_sst = xs[XS_RELAX_SST], SST0 = xs[XS_SST0];
        double fc = 0.0;
        if (relax_sst > 1.e-10) {
          if (!p.L_RELAX_CALCONLY) rhsT = rhsT + dto * relax_sst * (SST0 - To_k) * p.dm[kmixe] / c_hm[1];
          fc = relax_sst * (SST0 - To_k) * p.dm[kmixe] * rhok * cpk;
        }
        csrow()[CS_FCORR] = fc;
      }
      if (p.L_FCORR && !p.L_RELAX_SST && !p.L_FCORR_WITHZ)     // :121-125
        rhsT = rhsT + dto * xs[XS_FCORR_TWOD] / (rhok * cpk * c_hm[1]);
    }
    double tinc = 0.;                                           // :133-160
    if (p.L_FCORR_WITHZ && !p.L_FCORR) tinc = dto * p.fcorr_withz[oin] / (rhok * cpk);
    if (p.L_RELAX_OCNT) tinc = tinc + dto * xs[XS_RELAX_OCNT] * (p.ocnT_clim[oin] - To_k);
    rhsT = rhsT + tinc;
    xt[j] = tinc;
    const double ocnTcorr = tinc * rhok * cpk / dto;
    // prescribed advection of salinity, rhsmod with jsclr = 2 (:179-184)
    const int *ai = p.adv_i + (size_t)col * (p.maxmodeadv + 1);
    const double *ad = p.adv_d + (size_t)col * (p.maxmodeadv + 1);
    const int nmode = ai[0];
    const int nzi = nz, km = kmixe;
    for (int im = 0; im < nmode; ++im) {
      const int mode = ai[1 + im];
      if (mode <= 0) continue;
      const double fact = dto * ad[im] * 0.033;
      if (mode == 1) {
        if (k == 1) rhsS = rhsS + fact / c_hm[1];
      } else if (mode == 2) {
        const double delta = p.hsum[km - 1];
        if (k <= km - 1) rhsS = rhsS + fact / delta;
      } else if (mode == 3) {
        const double delta = p.hsum[nzi];
        if (k <= nzi) rhsS = rhsS + fact / delta;
      } else if (mode == 4) {
        const int nzend = nzi - 1;
        int n1 = 0;
        do { n1 = n1 + 1; } while (c_zm[n1] >= -100. && n1 < nzp1);
        double delta = 0.0;
        for (int n = n1; n <= nzend; ++n) delta = delta + c_hm[n];
        if (k >= n1 && k <= nzend) rhsS = rhsS + fact / delta;
      } else if (mode == 5) {
        if (k == nzi) rhsS = rhsS + fact / c_hm[nzi];
      } else if (mode == 6 || mode == 7) {
        int n1, n2 = 0;
        double depth, dmax, delta = 0.0;
        if (mode == 6) { n1 = 1; depth = c_hm[1]; dmax = p.dm[km] - 0.5 * (c_hm[km] + c_hm[km - 1]); }
        else { n1 = km - 1; depth = p.dm[km] - 0.5 * c_hm[km]; dmax = 100.; }
        for (int n = n1; n <= nzi; ++n) {
          n2 = n;
          delta = delta + c_hm[n];
          depth = depth + c_hm[n + 1];
          if (depth >= dmax) break;
        }
        if (k >= n1 && k <= n2) rhsS = rhsS + fact / delta;
      }
    }
    double sinc = 0.;                                           // :187-213
    if (p.L_SFCORR_WITHZ && !p.L_SFCORR) sinc = dto * p.sfcorr_withz[oin];
    if (p.L_RELAX_SAL) sinc = sinc + dto * xs[XS_RELAX_SAL] * (p.sal_clim[oin] - So_k);
    rhsS = rhsS + sinc;
    if (maybe_final()) {
      const size_t o = rowoff() + k;
      p.tinc_fcorr[o] = tinc; p.ocnTcorr[o] = ocnTcorr; p.sinc_fcorr[o] = sinc; p.scorr[o] = sinc / dto;
    }
  };

  // ---- phase C: hbl/kbl, blmix, enhance, combine, right-hand sides ---------
  auto phaseC = [&](bool do_ocnint) {
    const double *cs = csrow();
    const double B0 = first_lane(sc[C_B0]), B0sol = first_lane(sc[C_B0SOL]), ustar = first_lane(sc[C_USTAR]),
                 f = first_lane(sc[C_F]);
    const double ocdepth = cs[CS_OCDEPTH];
    const double zm_kmp1 = first_lane(c_zm[nzp1]);
    const wscale_u wu = wscale_prepare_uniform(ustar);
    const double fa = __builtin_fabs(f) + epsln16;
    const double hek = first_lane(div_fast(cekman * ustar, fa, rcp_refine(fa)));
    double zmk[LPL];
    FORJ zmk[j] = c_zm[kk[j]];
    int kbl = nz;
    double hbl = first_lane(-c_zm[nz]);
    {
      bool found = false;
      FORJ {
        int k = kk[j];
        double swf = p.swfrac_tab[jer * p.ldc + k];
        double bf = B0 + B0sol * (1. - swf);
        double stab = 0.5 + dsign(0.5, bf + epsln16);
        double Rka = aR[k - 1], Rku = aR[k], dmoa = aDmo[k - 1], dmou = aDmo[k];
        double zkm1 = c_zm[k - 1];
        double hri = -zkm1 + (zkm1 - zmk[j]) * (Ricr - Rka) / (Rku - Rka);
        double hmonob;
        if (dmou <= (-zmk[j])) {
          hmonob = (dmou - dmoa) / (zkm1 - zmk[j]);
          hmonob = (dmou + hmonob * zmk[j]) / (1. - hmonob);
        } else {
          hmonob = -zm_kmp1;
        }
        double hekman = stab * hek - (1. - stab) * zm_kmp1;
        double hmin = dmin2(dmin2(dmin2(hri, hmonob), hekman), -ocdepth);
        bool hit = (k >= 2) && actz[j] && (hmin < -zmk[j]);
        if (hit && !l_initflag && (hmin < -zkm1)) {
          double hmin2 = dmin2(dmin2(hri, hmonob), -ocdepth);
          if (hmin2 < -zmk[j]) hmin = hmin2;
        }
        unsigned long long m = __ballot(hit);
        if (!found && m != 0ull) {
          int src = __ffsll((long long)m) - 1;
          found = true;
          kbl = src + 64 * j + 1;
          hbl = first_lane(bcast(hmin, src));
        }
      }
    }
    double bfsfc = swfrac_dev_wave(-1.0, hbl, jer, lane);
    bfsfc = B0 + B0sol * (1. - bfsfc);
    const double stable = first_lane(0.5 + dsign(0.5, bfsfc));
    bfsfc = first_lane(bfsfc + stable * epsln16);
    const double caseA = first_lane(0.5 + dsign(0.5, -c_zm[kbl] - 0.5 * c_hm[kbl] - hbl));
    double gat1[3], dat1[3], dkm1[3];
    const double r_hbl = first_lane(rcp_refine(hbl));   // every quotient over hbl below shares it
    {
      double wm, ws;
      double sigma = stable * 1.0 + (1. - stable) * eps01;
      wscale_dev(p, wu, sigma, hbl, bfsfc, wm, ws);
      int ifx = (int)(caseA + epsln20);
      int kn = ifx * (kbl - 1) + (1 - ifx) * kbl;
      double hmkn = c_hm[kn], hmkn1 = c_hm[kn + 1];
      const double r_hmkn = rcp_refine(hmkn), r_hmkn1 = rcp_refine(hmkn1);
      double delhat = 0.5 * hmkn - c_zm[kn] - hbl;
      double R = 1.0 - div_fast(delhat, hmkn, r_hmkn);
      const double *dd[3] = {aDm, aDs, aDt};
      double dp[3], dh[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double dvdzup = div_fast(dd[m][kn - 1] - dd[m][kn], hmkn, r_hmkn);
        double dvdzdn = div_fast(dd[m][kn] - dd[m][kn + 1], hmkn1, r_hmkn1);
        dp[m] = 0.5 * ((1. - R) * (dvdzup + __builtin_fabs(dvdzup)) + R * (dvdzdn + __builtin_fabs(dvdzdn)));
        dh[m] = dd[m][kn] + dp[m] * delhat;
      }
      double u4 = ((ustar * ustar) * ustar) * ustar;
      const double u4e = u4 + epsln20, wme = wm + epsln20, wse = ws + epsln20;
      const double r_wme = rcp_refine(wme), r_wse = rcp_refine(wse);
      double f1 = div_fast(stable * 5.0 * bfsfc, u4e, rcp_refine(u4e));
      gat1[0] = div_fast(div_fast(dh[0], hbl, r_hbl), wme, r_wme);
      dat1[0] = div_fast(-dp[0], wme, r_wme) + f1 * dh[0];
      dat1[0] = dmin2(dat1[0], 0.);
      gat1[1] = div_fast(div_fast(dh[1], hbl, r_hbl), wse, r_wse);
      dat1[1] = div_fast(-dp[1], wse, r_wse) + f1 * dh[1];
      dat1[1] = dmin2(dat1[1], 0.);
      gat1[2] = div_fast(div_fast(dh[2], hbl, r_hbl), wse, r_wse);
      dat1[2] = div_fast(-dp[2], wse, r_wse) + f1 * dh[2];
      dat1[2] = dmin2(dat1[2], 0.);
#pragma unroll
      for (int m = 0; m < 3; ++m) { gat1[m] = first_lane(gat1[m]); dat1[m] = first_lane(dat1[m]); }
    }
    {
      double wm, ws;
      double sig = div_fast(-c_zm[kbl - 1], hbl, r_hbl);
      double sigma = stable * sig + (1. - stable) * dmin2(sig, eps01);
      wscale_dev(p, wu, sigma, hbl, bfsfc, wm, ws);
      double a1 = sig - 2.;
      double a2 = 3. - 2. * sig;
      double a3 = sig - 1.;
      double Gm = a1 + a2 * gat1[0] + a3 * dat1[0];
      double Gs = a1 + a2 * gat1[1] + a3 * dat1[1];
      double Gt = a1 + a2 * gat1[2] + a3 * dat1[2];
      dkm1[0] = first_lane(hbl * wm * sig * (1. + sig * Gm));
      dkm1[1] = first_lane(hbl * ws * sig * (1. + sig * Gs));
      dkm1[2] = first_lane(hbl * ws * sig * (1. + sig * Gt));
    }
    double difm[LPL], difs[LPL], dift[LPL], ghat[LPL];
    FORJ {
      int k = kk[j];
      const double hk = c_hm[k];
      const double dm_i = aDm[k], ds_i = aDs[k], dt_i = aDt[k];   // interior values of phase A
      double wm, ws;
      double sig = div_fast(-zmk[j] + 0.5 * hk, hbl, r_hbl);
      double sigma = stable * sig + (1. - stable) * dmin2(sig, eps01);
      wscale_dev(p, wu, sigma, hbl, bfsfc, wm, ws);
      double a1 = sig - 2.;
      double a2 = 3. - 2. * sig;
      double a3 = sig - 1.;
      double Gm = a1 + a2 * gat1[0] + a3 * dat1[0];
      double Gs = a1 + a2 * gat1[1] + a3 * dat1[1];
      double Gt = a1 + a2 * gat1[2] + a3 * dat1[2];
      double b0 = hbl * wm * sig * (1. + sig * Gm);
      double b1 = hbl * ws * sig * (1. + sig * Gs);
      double b2 = hbl * ws * sig * (1. + sig * Gt);
      const double ghd = ws * hbl + epsln20;
      double gh = div_fast((1. - stable) * p.cg, ghd, rcp_refine(ghd));
      if (k == kbl - 1 && k <= nz - 1) {   // enhance_mod.F90:10-51
        double delta = div_fast(hbl + zmk[j], zmk[j] - c_zm[k + 1], c_rdz[k]);
        double omd = 1. - delta;
        double dkmp5 = caseA * dm_i + (1. - caseA) * b0;
        double dstar = (omd * omd) * dkm1[0] + (delta * delta) * dkmp5;
        b0 = omd * dm_i + delta * dstar;
        dkmp5 = caseA * ds_i + (1. - caseA) * b1;
        dstar = (omd * omd) * dkm1[1] + (delta * delta) * dkmp5;
        b1 = omd * ds_i + delta * dstar;
        dkmp5 = caseA * dt_i + (1. - caseA) * b2;
        dstar = (omd * omd) * dkm1[2] + (delta * delta) * dkmp5;
        b2 = omd * dt_i + delta * dstar;
        gh = (1. - caseA) * gh;
      }
      if (k < kbl) {   // kppmix_mod.F90:103-111
        difm[j] = b0; difs[j] = b1; dift[j] = b2; ghat[j] = gh;
      } else {
        difm[j] = dm_i; difs[j] = ds_i; dift[j] = dt_i; ghat[j] = 0.;
      }
      if (k >= nz) {   // verticalmixing_mod.F90:151-159
        difm[j] = 0.0001; difs[j] = 0.00001; dift[j] = 0.00001; ghat[j] = 0.0;
      }
    }
    put(C_HBL, hbl);
    kbl_pass = kbl;
    // final diffusivities into the rows the Thomas lanes (and finalize) read
    WAVE_LDS_SYNC();
    FORJ if (act[j]) {
      int k = kk[j];
      aDm[k] = difm[j]; aDs[k] = difs[j]; aDt[k] = dift[j]; aGh[k] = ghat[j];
    }
    if (!do_ocnint) return;
    WAVE_LDS_SYNC();
    double Uo[LPL], Vo[LPL], To[LPL], So[LPL];
    load_old(p.U, Uo); load_old(p.V, Vo); load_old(p.T, To); load_old(p.S, So);
    const double Uo_np = old_bottom(p.U), To_np = old_bottom(p.T), So_np = old_bottom(p.S);
    const double dto = p.dto, tri1_nz = first_lane(c_t1[nz]), hm1 = first_lane(c_hm[1]);
    const double wU0_1 = first_lane(sc[C_WU01]), wX0_1 = first_lane(sc[C_WX01]), wX0_2 = first_lane(sc[C_WX02]),
                 wXNT0 = first_lane(sc[C_WXNT0]);
    const double rho0cp0 = first_lane(sc[C_RHO0CP0]), r_rc = first_lane(sc[C_RRC]), sflux3 = cs[CS_SFLUX3];
    const double r_hm1 = first_lane(c_misc[0]);
    double *yU = row(R_YU), *yT = row(R_YT), *yS = row(R_YS);
    FORJ {
      int k = kk[j];
      if (!actz[j]) continue;
      const double dt_m1 = aDt[k - 1], ds_m1 = aDs[k - 1];
      const double gh_m1 = (k >= 2) ? aGh[k - 1] : 0.0;
      double wxnt = 0.0, wxnt_m1 = 0.0;   // wXNT(k,1), wXNT(k-1,1), fluxes_mod.F90:110-116
      if (ntime >= 1) {
        wxnt = div_fast(-sflux3 * p.swdk_tab[jer * p.ldc + k], rho0cp0, r_rc);
        wxnt_m1 = div_fast(-sflux3 * p.swdk_tab[jer * p.ldc + k - 1], rho0cp0, r_rc);
      }
      double rhsU;   // ocnint_mod.F90:51-58
      if (k == 1) rhsU = Uo[j] + dto * (f * .5 * (Vo[j] + V[j]) - div_fast(wU0_1, hm1, r_hm1));
      else rhsU = Uo[j] + dto * f * .5 * (Vo[j] + V[j]);
      if (k == nz) rhsU = rhsU + tri1_nz * difm[j] * Uo_np;
      double rhsT;   // tridrhs, solvers.F90:53-107 (npd = 1)
      const double dtohk = c_dtohk[k];   // dto/hm(k), the quotient itself tabulated
      if (k == 1)
        rhsT = To[j] + dtohk * (wX0_1 * dift[j] * ghat[j] - wX0_1 * 1.0 + wxnt - wXNT0);
      else
        rhsT = To[j] + dtohk * (wX0_1 * (dift[j] * ghat[j] - dt_m1 * gh_m1) + wxnt - wxnt_m1);
      if (k == nz && nz > 1) rhsT = rhsT + To_np * tri1_nz * dift[j];
      double rhsS;
      if (k == 1)
        rhsS = So[j] + dtohk * (wX0_2 * difs[j] * ghat[j] - wX0_2 * 1.0 + 0.0 - 0.0);
      else
        rhsS = So[j] + dtohk * (wX0_2 * (difs[j] * ghat[j] - ds_m1 * gh_m1) + 0.0 - 0.0);
      if (k == nz && nz > 1) rhsS = rhsS + So_np * tri1_nz * difs[j];
      if constexpr (EXT) ext_rhs(k, j, kbl, To[j], So[j], rhsT, rhsS);
      yU[k] = rhsU; yT[k] = rhsT; yS[k] = rhsS;
    }
    FORJ if (kk[j] == nzp1) { yU[nzp1] = Uo[j]; yT[nzp1] = To[j]; yS[nzp1] = So[j]; }   // solvers.F90:159
    if constexpr (EXT) {   // tinc_fcorr / ocnTcorr / sinc_fcorr / scorr of level nzp1 (ocnint_mod.F90:153-160, 207-213)
      FORJ if (kk[j] == nzp1) { double t = 0.0, s2 = 0.0; ext_rhs(nzp1, j, kbl, To[j], So[j], t, s2); }
    }
  };

  // ---- phase E: collect U,T,S; V right-hand side (ocnint_mod.F90:62-69) ----
  auto phaseE = [&]() {
    const double *yU = row(R_YU);
    double *yV = row(R_YV);
    double Uo[LPL], Vo[LPL];
    load_old(p.U, Uo); load_old(p.V, Vo);
    const double Vo_np = old_bottom(p.V);
    const double dto = p.dto, tri1_nz = first_lane(c_t1[nz]), hm1 = first_lane(c_hm[1]), f = first_lane(sc[C_F]),
                 wU0_2 = first_lane(sc[C_WU02]), r_hm1 = first_lane(c_misc[0]);
    FORJ {
      int k = kk[j];
      if (actz[j]) {
        const double un = yU[k];
        double rhsV;
        if (k == 1) rhsV = Vo[j] - dto * (f * .5 * (Uo[j] + un) + div_fast(wU0_2, hm1, r_hm1));
        else rhsV = Vo[j] - dto * f * .5 * (Uo[j] + un);
        if (k == nz) rhsV = rhsV + tri1_nz * aDm[k] * Vo_np;
        yV[k] = rhsV;
      } else if (act[j]) {   // yn(nzi+1) = yo(nzi+1), solvers.F90:159
        yV[k] = Vo[j];
      }
    }
  };

  // diagnostic fluxes (ocnstep_mod.F90:242-256 / initialize_ocean.F90:66-81) and stores
  auto finalize = [&]() {
    double *cs = csrow();
    int *ci = p.ci + (size_t)col * MCKPP_CI;
    const size_t ro = rowoff();
    if (p.diag) {
      const double wX0_1 = sc[C_WX01], wX0_2 = sc[C_WX02];
      double *tU = row(R_YU), *tV = row(R_YT), *tT = row(R_YS), *tS = row(R_GM);
      if (p.mode == MCKPP_MODE_STEP || p.mode == MCKPP_MODE_INIT) {
        WAVE_LDS_SYNC();
        FORJ if (act[j]) { int k = kk[j]; tU[k] = U[j]; tV[k] = V[j]; tT[k] = T[j]; tS[k] = S[j]; }
        WAVE_LDS_SYNC();
      }
      const double rho0cp0 = sc[C_RHO0CP0], sflux3 = cs[CS_SFLUX3];
      FORJ {
        int k = kk[j];
        size_t o = ro + k;
        const double dfm = aDm[k], dfs = aDs[k], dft = aDt[k], gh = aGh[k];
        if (act[j]) { p.difm[o] = dfm; p.difs[o] = dfs; p.dift[o] = dft; }
        if (actz[j]) {
          p.ghat[o] = gh;
          p.wXNT1[o] = (ntime >= 1) ? -sflux3 * p.swdk_tab[jer * p.ldc + k] / rho0cp0 : 0.0;
          if (p.mode == MCKPP_MODE_STEP || p.mode == MCKPP_MODE_INIT) {
            double deltaz = 0.5 * (c_hm[k] + c_hm[k + 1]);
            double uk1 = tU[k + 1], vk1 = tV[k + 1], tk1 = tT[k + 1], sk1 = tS[k + 1];
            double wX1 = -dfs * ((T[j] - tk1) / deltaz - gh * wX0_1);
            double wX2 = -dfs * ((S[j] - sk1) / deltaz - gh * wX0_2);
            if (p.LDD) wX1 = -dft * ((T[j] - tk1) / deltaz - gh * wX0_1);
            p.wX1[o] = wX1; p.wX2[o] = wX2;
            p.wX3[o] = p.grav * (talpha[j] * wX1 - sbeta[j] * wX2);
            p.wU1[o] = -dfm * (U[j] - uk1) / deltaz;
            p.wU2[o] = -dfm * (V[j] - vk1) / deltaz;
          }
        }
        if (k == 1) {   // index-0 entries
          p.difm[ro] = 0.0; p.difs[ro] = 0.0; p.dift[ro] = 0.0;
          p.wU1[ro] = sc[C_WU01]; p.wU2[ro] = sc[C_WU02];
          p.wX1[ro] = wX0_1; p.wX2[ro] = wX0_2; p.wX3[ro] = -sc[C_B0];
          p.wXNT1[ro] = sc[C_WXNT0];
        }
      }
    }
    if (p.mode == MCKPP_MODE_STEP) {
      const double uref = first_lane(U[0]), vref = first_lane(V[0]), Tref = first_lane(T[0]);
      double Ssurf;
      if (p.L_SSref) Ssurf = cs[CS_SSREF];
      else Ssurf = first_lane(S[0]) + cs[CS_SREF];
      double dampu = 0.0, dampv = 0.0, reset_out = 0.0, freeze = cs[CS_FREEZE];
      if constexpr (EXT) {
        if (p.L_DAMP_CURR) {   // ocnstep_mod.F90:317-340
          const double rr = (double)p.dt_uvdamp * (86400. / p.dto);
          const double inc = 1.0 / (double)nzp1;
          int nu = 0, nv = 0;
          FORJ {
            double a = 0.99 * __builtin_fabs(U[j]), b = (U[j] * U[j]) / rr;
            nu += __popcll(__ballot(act[j] && (b < a)));
            U[j] = U[j] - dsign(dmin2(a, b), U[j]);
            a = 0.99 * __builtin_fabs(V[j]); b = (V[j] * V[j]) / rr;
            nv += __popcll(__ballot(act[j] && (b < a)));
            V[j] = V[j] - dsign(dmin2(a, b), V[j]);
          }
          for (int i = 0; i < nu; ++i) dampu = dampu + inc;
          for (int i = 0; i < nv; ++i) dampv = dampv + inc;
        }
      }
      old = newi;
      newi = 1 - old;
      FORJ if (act[j]) {
        size_t o = ro + lane + 64 * j;
        p.Us[newi][o] = U[j]; p.Vs[newi][o] = V[j]; p.Ts[newi][o] = T[j]; p.Ss[newi][o] = S[j];
      }
      // check_profile, overrides.F90:42-125
      reset_out = (double)nreset;
      if (comp_flag) {
        if (EXT && p.clim_present) {   // :57-71
          FORJ if (act[j]) {
            size_t o = ro + lane + 64 * j;
            T[j] = p.ocnT_clim[o];
            S[j] = p.sal_clim[o];
          }
        }
        FORJ if (act[j]) {   // :66 / :76
          size_t o = ro + lane + 64 * j;
          U[j] = p.U_init[o];
          V[j] = p.V_init[o];
        }
        reset_out = 999.;
      }
      const int l_ocean = ci[CI_LOCEAN];
      if constexpr (EXT) {
        if (l_ocean && p.L_NO_FREEZE) {   // :85-94
          const double inc = 1.0 / (double)nzp1;
          int nf = 0;
          FORJ {
            bool cold = act[j] && (T[j] < -1.8);
            if (cold) { xt[j] = xt[j] + (-1.8 - T[j]); T[j] = -1.8; }
            nf += __popcll(__ballot(cold));
          }
          for (int i = 0; i < nf; ++i) freeze = freeze + inc;
          FORJ if (act[j]) p.tinc_fcorr[ro + kk[j]] = xt[j];
        }
      }
      if (EXT && l_ocean && p.L_NO_ISOTHERM) {   // :102-120
        double *tT = row(R_YU), *tD = row(R_YT), *tZ = row(R_YS);
        WAVE_LDS_SYNC();
        FORJ if (act[j]) tT[kk[j]] = T[j];
        WAVE_LDS_SYNC();
        FORJ {
          int k = kk[j];
          if (k >= 2 && act[j]) {
            double dz = c_zm[k] - c_zm[k - 1];
            tD[k] = __builtin_fabs((T[j] - tT[k - 1])) * dz;
            tZ[k] = dz;
          }
        }
        WAVE_LDS_SYNC();
        double dtdz_total = 0., dz_total = 0.;
        for (int k = 2; k <= p.iso_bot; ++k) {
          dtdz_total = dtdz_total + tD[k];
          dz_total = dz_total + tZ[k];
        }
        dtdz_total = dtdz_total / dz_total;
        if (__builtin_fabs(dtdz_total) < p.iso_thresh) {
          FORJ if (act[j]) {
            size_t o = ro + lane + 64 * j;
            T[j] = p.ocnT_clim[o];
            S[j] = p.sal_clim[o];
          }
          reset_out = (-1.) * reset_out;
        }
      } else {
        reset_out = 0.0;   // :121-123
      }
      FORJ if (act[j]) {
        size_t o = ro + lane + 64 * j;
        p.U[o] = U[j]; p.V[o] = V[j]; p.T[o] = T[j]; p.S[o] = S[j];
      }
      if (lane == 0) {
        const double hmixn = sc[C_HMIXN];
        cs[CS_HMIX] = hmixn;
        cs[CS_KMIX] = (double)kmixn;
        cs[CS_UREF] = uref; cs[CS_VREF] = vref; cs[CS_TREF] = Tref;
        cs[CS_SSURF] = Ssurf;
        cs[newi ? CS_HMIXD1 : CS_HMIXD0] = hmixn;
        cs[CS_RESET] = reset_out;
        cs[CS_DAMPU] = dampu; cs[CS_DAMPV] = dampv;
        cs[CS_FREEZE] = freeze;
        ci[CI_OLD] = old; ci[CI_NEW] = newi;
        ci[CI_STATUS] = status; ci[CI_NPASS] = npass;
      }
    } else if (p.mode == MCKPP_MODE_INIT) {
      const double Tref = first_lane(T[0]);
      FORJ if (act[j]) {
        size_t o = ro + lane + 64 * j;
        p.Us[0][o] = U[j]; p.Us[1][o] = U[j]; p.Vs[0][o] = V[j]; p.Vs[1][o] = V[j];
        p.Ts[0][o] = T[j]; p.Ts[1][o] = T[j]; p.Ss[0][o] = S[j]; p.Ss[1][o] = S[j];
      }
      if (lane == 0) {
        const double hbl = sc[C_HBL];
        cs[CS_HMIX] = hbl;
        cs[CS_KMIX] = (double)kbl_pass;
        cs[CS_TREF] = Tref;
        cs[CS_UREF] = sc[C_UREFNZ]; cs[CS_VREF] = sc[C_VREFNZ];
        cs[CS_HMIXD0] = hbl; cs[CS_HMIXD1] = hbl;
        ci[CI_OLD] = 0; ci[CI_NEW] = 1; ci[CI_INITFLAG] = 0;
        ci[CI_STATUS] = status; ci[CI_NPASS] = npass;
      }
    } else {
      if (p.mode == MCKPP_MODE_PASS) {
        FORJ if (act[j]) {
          size_t o = ro + lane + 64 * j;
          p.U[o] = U[j]; p.V[o] = V[j]; p.T[o] = T[j]; p.S[o] = S[j];
        }
      }
      if (lane == 0) {
        cs[CS_HMIX] = sc[C_HBL];
        cs[CS_KMIX] = (double)kbl_pass;
        cs[CS_UREF] = sc[C_UREFNZ]; cs[CS_VREF] = sc[C_VREFNZ];
        ci[CI_STATUS] = status; ci[CI_NPASS] = npass;
      }
    }
    state = S_EMPTY;
  };

  // ---- phase G: collect V; ocnstep control (ocnstep_mod.F90:122-236) ----
  auto phaseG = [&]() {
    auto load_solution = [&]() {   // U,V,T,S <- what the last ocnint returned
      const double *yU = row(R_YU), *yV = row(R_YV), *yT = row(R_YT), *yS = row(R_YS);
      FORJ if (act[j]) { int k = kk[j]; U[j] = yU[k]; V[j] = yV[k]; T[j] = yT[k]; S[j] = yS[k]; }
      WAVE_LDS_SYNC();
    };
    if (p.mode != MCKPP_MODE_INIT && sbad[wave]) status |= 1;
    ++npass;
    if (p.mode != MCKPP_MODE_STEP) {
      if (p.mode == MCKPP_MODE_PASS) load_solution();
      finalize();
      return;
    }
    ++npass_try;
    const double hbl = sc[C_HBL];
    if (npass_try <= 3) {   // compulsory passes
      put(C_HMIXE, hbl);
      return;
    }
    const double hmixn = hbl, hmixe = sc[C_HMIXE];
    kmixn = kbl_pass;
    put(C_HMIXN, hmixn);
    double tol = p.hmixtolfrac * c_hm[kmixn];
    if (kmixn == nzp1) tol = p.hmixtolfrac * c_hm[nz];
    if (__builtin_fabs(hmixn - hmixe) > tol) iconv = 0;
    else iconv = iconv + 1;
    if (iconv < 3) {
      if (npass_try < p.itermax) { put(C_HMIXE, hmixn); return; }
      else if (hmixn > hmixe) { put(C_HMIXE, hmixn); return; }
    }
    if (npass_try > (p.itermax + 1)) status |= 2;
    load_solution();
    // instability trap
    comp_flag = 0;
    double f = sc[C_F];
    WAVE_LDS_SYNC();
    FORJ if (act[j]) aT[kk[j]] = T[j];
    WAVE_LDS_SYNC();
    int nviol = 0;
    FORJ {
      int k = kk[j];
      double tk1 = aT[k + 1];
      bool v = actz[j] && (__builtin_fabs(U[j]) >= 10 || __builtin_fabs(V[j]) >= 10 ||
                           __builtin_fabs(T[j] - tk1) >= 10);
      nviol += __popcll(__ballot(v));
    }
    if (nviol > 0) {
      comp_flag = 1;
      for (int i = 0; i < nviol; ++i) f = f * 1.01;
    }
    if (!comp_flag) {
      double *t0 = row(R_YU), *t1 = row(R_YT), *t2 = row(R_YS), *t3 = row(R_GM);
      double Uo[LPL], Vo[LPL], To[LPL], So[LPL];
      load_old(p.U, Uo); load_old(p.V, Vo); load_old(p.T, To); load_old(p.S, So);
      WAVE_LDS_SYNC();
      FORJ if (act[j]) {
        int k = kk[j];
        const double hk = c_hm[k];
        t0[k] = (U[j] - Uo[j]) * (U[j] - Uo[j]) * hk / p.dm_nz;
        t1[k] = (V[j] - Vo[j]) * (V[j] - Vo[j]) * hk / p.dm_nz;
        t2[k] = (T[j] - To[j]) * (T[j] - To[j]) * hk / p.dm_nz;
        t3[k] = (S[j] - So[j]) * (S[j] - So[j]) * hk / p.dm_nz;
      }
      WAVE_LDS_SYNC();
      bool over = false;
      if (lane < 4) {
        const double *t = row(R_YU + lane);
        double sum = 0.;
        for (int k = 1; k <= nzp1; ++k) sum = sum + t[k];
        sum = __builtin_sqrt(sum);
        over = sum >= 1.0;
      }
      int nover = __popcll(__ballot(over));
      if (nover > 0) {
        comp_flag = 1;
        for (int i = 0; i < nover; ++i) f = f * 1.01;
      }
    }
    if (comp_flag) { status |= 4; put(C_F, f); }
    nreset = nreset + 1;
    if (nreset > 10) status |= 8;
    if (comp_flag && nreset <= 10) { extrapolate(); return; }   // retry, ocnstep_mod.F90:89
    finalize();
  };

  // ---- serial phases, one lane per (slot[, system]) ------------------------
  // The recurrences are latency chains, so operands are fetched one step ahead
  // (loads never wait behind the chain) and every division uses a reciprocal
  // refined off the chain (div_by_refined: same correctly rounded quotient).
  // ---- serial phases (mckpp_sweeps.h), run by one wave for all slots of the workgroup ----
  auto scan_rib = [&]() { if (wave == 0) serial_scan_rib<W>(slots, SS, NA, nz, sact, lane); };
  auto thomas_uts = [&]() { if (wave == 0) serial_thomas_uts<W>(slots, SS, NA, nz, c_t0, c_t1, sact, sbad, lane); };
  auto thomas_v = [&]() { if (wave == 0) serial_thomas_v<W>(slots, SS, NA, nz, c_t0, sact, lane); };

  // ---- persistent pass loop -------------------------------------------------
  // p.dbg != nullptr: wave 1 of every workgroup accumulates shader cycles per segment
  // (0 refill, 1 A, 2 wait, 3 scan+wait, 4 C, 5 wait, 6 UTS+wait, 7 E, 8 wait, 9 V+wait, 10 G, 11 passes)
  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = p.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
#define STAMP(i)                                              \
  do {                                                        \
    if (p.dbg) {                                              \
      unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
      tacc[i] += t_ - tlast;                                  \
      tlast = t_;                                             \
    }                                                         \
  } while (0)
  const bool do_ocnint = p.mode == MCKPP_MODE_STEP || p.mode == MCKPP_MODE_PASS;
  for (;;) {
    if (state == S_EMPTY) refill();
    const bool active = state == S_ACTIVE;
    if (lane == 0) { sact[wave] = active ? 1 : 0; sbad[wave] = 0; }
    if (!__syncthreads_or(active ? 1 : 0)) break;
    STAMP(0);
    if (active) phaseA();
    STAMP(1);
    __syncthreads();
    STAMP(2);
    scan_rib();
    __syncthreads();
    STAMP(3);
    if (active) phaseC(do_ocnint);
    STAMP(4);
    __syncthreads();
    STAMP(5);
    if (do_ocnint) thomas_uts();
    __syncthreads();
    STAMP(6);
    if (active && do_ocnint) phaseE();
    STAMP(7);
    __syncthreads();
    STAMP(8);
    if (do_ocnint) thomas_v();
    __syncthreads();
    STAMP(9);
    if (active) phaseG();
    STAMP(10);
    tacc[11] += 1;
  }
  if (p.dbg && wave == 1 && lane == 0)
    for (int i = 0; i < 12; ++i) atomicAdd(p.dbg + i, tacc[i]);
#undef STAMP
}

template <int LPL, int W, bool EXT>
size_t wg_lds_bytes()
{
  return (size_t)(WG_CONST_DOUBLES(wg_na<LPL>()) + W * wg_slot_stride<LPL, EXT>() + W * C_COUNT) * sizeof(double) +
         2 * W * sizeof(int);
}

template <int LPL, int W, int MINW, bool EXT = false>
hipError_t launch_wg(const mckpp_kparams &p, const mckpp_kparams *dp, int nblocks, hipStream_t stream)
{
  const size_t lds = wg_lds_bytes<LPL, W, EXT>();
  // per device (a process may hold contexts on several GPUs): the dynamic-LDS attribute and the occupancy query
  static bool attr_set[64] = {};
  static int max_blocks_dev[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_column_wg<LPL, W, MINW, EXT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(k_column_wg<LPL, W, MINW, EXT>), 64 * W, lds) != hipSuccess) nb = 0;
    max_blocks_dev[dev] = nb;
    attr_set[dev] = true;
  }
  const int max_blocks = max_blocks_dev[dev];
  g_mckpp_last_launch = {nblocks, 64 * W, max_blocks, lds};
  hipLaunchKernelGGL((k_column_wg<LPL, W, MINW, EXT>), dim3((unsigned)nblocks), dim3(64 * W), lds, stream, dp,
                     p.ntime);
  return hipGetLastError();
}

}  // namespace

// Persistent grid: enough workgroups to fill every CU at the occupancy LDS and
// registers allow, never more than there are W-column groups.
// MCKPP_WG=<W>[x<blocks per CU>] overrides the geometry (experiments).
// Only the one-level-per-lane instantiation (LPL = 1, columns of up to 61 levels) is built: deeper
// columns run k_column_pk.
hipError_t mckpp_launch_column_kernel_wg(const mckpp_kparams &p, const mckpp_kparams *dp, int num_cu, hipStream_t stream)
{
  if (p.ncol <= 0) return hipSuccess;
  if (p.nzp1 + 2 > 64) return hipErrorInvalidValue;
  static int envW = -1, envB = 0;
  if (envW < 0) {
    envW = 0;
    if (const char *e = getenv("MCKPP_WG")) {
      int w = 0, b = 0;
      if (sscanf(e, "%dx%d", &w, &b) >= 1) { envW = w; envB = b; }
    }
  }
  int W = 4, per_cu = 5;   // 31.9 KB LDS and <=96 VGPRs per wave: 20 waves per CU
  if (envW == 4 || envW == 8) { W = envW; per_cu = (W == 8) ? 2 : 5; }
  if (envB > 0) per_cu = envB;
  int nblocks = num_cu * per_cu;
  const int groups = (p.ncol + W - 1) / W;
  if (nblocks > groups) nblocks = groups;
  if (nblocks < 1) nblocks = 1;
  if (p.ext) {   // optional-physics build: W = 4, register budget of four workgroups per CU
    nblocks = num_cu * 4;
    if (nblocks > (p.ncol + 3) / 4) nblocks = (p.ncol + 3) / 4;
    if (nblocks < 1) nblocks = 1;
    return launch_wg<1, 4, 4, true>(p, dp, nblocks, stream);
  }
  if (W == 8) return (per_cu >= 2) ? launch_wg<1, 8, 4>(p, dp, nblocks, stream) : launch_wg<1, 8, 2>(p, dp, nblocks, stream);
  if (per_cu >= 5) return launch_wg<1, 4, 5>(p, dp, nblocks, stream);   // 96-VGPR build
  if (per_cu >= 4) return launch_wg<1, 4, 4>(p, dp, nblocks, stream);   // 128-VGPR build
  if (per_cu == 3) return launch_wg<1, 4, 3>(p, dp, nblocks, stream);   // 168-VGPR build
  return launch_wg<1, 4, 2>(p, dp, nblocks, stream);
}
