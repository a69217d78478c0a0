// mckpp_kernels_mw.hip - cooperative column kernel for deep columns (63..190 levels).
//
// Same algorithm, arithmetic and slot/queue organisation as k_column_wg
// (mckpp_kernels_wg.hip), but a water column is spread over WPS wavefronts with
// ONE level per lane instead of giving each lane several levels: the per-level
// instruction stream is paid once (not once per 64-level row), registers stay
// at the one-level budget, and a column with 70 levels keeps 2x the waves in
// flight.  The price is that the two waves of a column exchange neighbour
// levels and wave-uniform picks through the slot's LDS rows/records, so every
// intra-pass hand-off is a workgroup barrier (all slots of a workgroup run the
// sub-phases in lock-step; 14 s_barriers per pass) instead of a wave-local fence.
//
// Results are bit-identical to k_column_wg / k_column and to the CPU oracle.
// EXT=true carries the optional physics (two more LDS rows for rho/cp, four more
// barriers in the finish round for the cross-wave counts of check_profile).
#include "mckpp_sweeps.h"

#include <cstdio>
#include <type_traits>
#include <cstdlib>

namespace {

using namespace mckpp_dev;

enum { S_EMPTY = 0, S_ACTIVE = 1, S_DONE = 2 };
// per-slot double record
enum { C_B0 = 0, C_B0SOL, C_USTAR, C_WU01, C_WU02, C_WX01, C_WX02, C_WXNT0, C_UREFNZ, C_VREFNZ, C_RHO0CP0, C_RRC,
       X_RHO0, X_CP0, X_TALPHA0, X_SBETA0, X_RHOH2O, X_RHOB,
       X_CAND_HBL,            // + sub (WPS entries)
       C_COUNT = X_CAND_HBL + 4 };
// per-slot int record
enum { I_COL = 0, I_CAND_KBL /* + sub */, I_NVIOL = I_CAND_KBL + 4 /* + sub */, I_OVER = I_NVIOL + 4,
       I_NU = I_OVER + 4 /* + sub */, I_NV = I_NU + 4, I_NF = I_NV + 4, I_COUNT = I_NF + 4 };
enum { R_RHO = R_COUNT, R_CP, R_COUNT_EXT };   // rho, cp rows of the optional-physics build

template <int WPS>
__host__ __device__ constexpr int mw_na() { return 64 * WPS + 3; }
template <int WPS, bool EXT = false>
__host__ __device__ constexpr int mw_slot_stride()
{
  int s = (EXT ? R_COUNT_EXT : R_COUNT) * mw_na<WPS>();
  while (s % 32 != 9) ++s;
  return s;
}

template <int WPS, int W, int MINW, bool EXT>
__global__ __launch_bounds__(64 * WPS * W, MINW) void k_column_mw(const mckpp_kparams *__restrict__ pp, const int ntime)
{
  const mckpp_kparams &p = *pp;
  extern __shared__ double lds[];
  constexpr int NA = mw_na<WPS>();
  constexpr int SS = mw_slot_stride<WPS, EXT>();
  constexpr int NW = WPS * W;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slot = wv / WPS, sub = wv - slot * WPS;
  const bool lead = (sub == 0);            // the wave that owns level 1
  const int nz = p.nz, nzp1 = p.nzp1;
  double *c_zm = lds, *c_hm = lds + NA, *c_t0 = lds + 2 * NA, *c_t1 = lds + 3 * NA;
  // grid-constant refined reciprocals / quotients for div_fast (see k_column_wg)
  double *c_rdz = lds + 4 * NA, *c_dtohk = lds + 5 * NA, *c_misc = c_rdz + (NA - 2);
  double *slots = lds + 6 * NA;
  double *my = slots + slot * SS;
  double *screc = slots + W * SS;
  double *sc = screc + slot * C_COUNT;
  int *sirec = reinterpret_cast<int *>(screc + W * C_COUNT);
  int *si = sirec + slot * I_COUNT;
  int *sact = sirec + W * I_COUNT;
  int *sbad = sact + W;
  auto row = [&](int a) -> double * { return my + a * NA; };
  double *aDm = row(R_DM), *aDs = row(R_DS), *aDt = row(R_DT), *aGh = row(R_GH);
  double *aU = row(R_U), *aV = row(R_V), *aB = row(R_B), *aR = row(R_R), *aDb = row(R_DB),
         *aDmo = row(R_DMO), *aT = row(R_T);

  for (int i = threadIdx.x; i < NA; i += 64 * NW) {
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

  // ---- per-lane state: one level ------------------------------------------
  // k is laundered once per pass (top of the persistent loop) so that the compiler does not hoist
  // every k-indexed grid-constant LDS read out of the loop into a long-lived register pair
  const int k0 = lane + 64 * sub + 1;
  int k = k0;
  const bool act = k0 <= nzp1, actz = k0 <= nz;
  double U = 0, V = 0, T = 0, S = 0, talpha = 0, sbeta = 0;
  // values that cross sub-phases of one pass
  double buoy = 0, Ritop = 0, dVsq = 0, dbloc = 0, shsq = 0, Rig = 0, zdiff = 0, zmk = 0;
  double dm_i = 0, ds_i = 0, dt_i = 0, difm = 0, difs = 0, dift = 0, ghat = 0;
  double alphaDT = 0, betaDS = 0, xt = 0;   // EXT: LDD inputs; tinc_fcorr of the latest pass (overrides.F90:87-88)

  int state = S_EMPTY, col = 0;
  int old = 0, newi = 1, jer = 3, l_initflag = 0, status = 0, npass = 0, npass_try = 0, iconv = 0;
  int comp_flag = 0, kmixn = 0, kbl_pass = 0, nreset = 0;
  // wave-uniform doubles every wave of a slot carries identically (no LDS hand-off, so no race)
  double f_col = 0, hmixe = 0, hmixn = 0, hbl_pass = 0;
  const double lambda = 0.5;
  const double epsln16 = 1.e-16, Ricr = 0.30, eps01 = 0.1, cekman = 0.7, cmonob = 1.0, epsln20 = 1.e-20;

  auto put = [&](int s_, double v) { if (lead && lane == 0) sc[s_] = v; };
  auto rowoff = [&]() -> size_t { return (size_t)col * p.ld; };
  auto csrow = [&]() -> double * { return p.cs + (size_t)col * MCKPP_CS; };
  auto ld_old = [&](const double *src) -> double { return act ? src[(size_t)col * p.ld + (k - 1)] : 0.0; };
  auto old_bottom = [&](const double *src) -> double { return src[(size_t)col * p.ld + (nzp1 - 1)]; };

  auto extrapolate = [&]() {   // ocnstep_mod.F90:91-112; also seeds the solution rows (see k_column_wg)
    size_t o = rowoff() + (k - 1);
    double uo = act ? p.Us[old][o] : 0.0, un = act ? p.Us[newi][o] : 0.0;
    double vo = act ? p.Vs[old][o] : 0.0, vn = act ? p.Vs[newi][o] : 0.0;
    double to = act ? p.Ts[old][o] : 0.0, tn = act ? p.Ts[newi][o] : 0.0;
    double so = act ? p.Ss[old][o] : 0.0, sn = act ? p.Ss[newi][o] : 0.0;
    U = 2. * un - uo;
    V = 2. * vn - vo;
    T = 2. * tn - to;
    S = 2. * sn - so;
    if (act) { row(R_YU)[k] = U; row(R_YV)[k] = V; row(R_YT)[k] = T; row(R_YS)[k] = S; }
    npass_try = 0;
    iconv = 0;
  };

  auto load_column = [&]() {   // after the slot's leader has published the queue ticket
    const int c = si[I_COL];
    if (c >= p.ncol) { state = S_DONE; return; }
    col = c;
    state = S_ACTIVE;
    const int *ci = p.ci + (size_t)col * MCKPP_CI;
    old = ci[CI_OLD]; newi = ci[CI_NEW]; jer = ci[CI_JERLOV]; l_initflag = ci[CI_INITFLAG];
    status = 0; npass = 0; comp_flag = 1; nreset = 0;
    if (old < 0 || old > 1) { old = newi; status |= 16; }
    if (newi < 0 || newi > 1) { newi = old; status |= 16; }
    f_col = first_lane(csrow()[CS_F]);
    put(C_WXNT0, 0.0);
    size_t o = rowoff() + (k - 1);
    U = act ? p.U[o] : 0.0; V = act ? p.V[o] : 0.0; T = act ? p.T[o] : 0.0; S = act ? p.S[o] : 0.0;
    if (p.mode == MCKPP_MODE_STEP) extrapolate();
    if (p.mode == MCKPP_MODE_INIT) l_initflag = 1;
  };

  auto maybe_final = [&]() -> bool {
    if (p.mode != MCKPP_MODE_STEP) return true;
    return npass_try >= 3 && (iconv >= 2 || npass_try + 1 >= p.itermax);
  };

  // ---- sub-phases of a pass (citations: see the same code in k_column_wg) --
  auto A1 = [&]() {   // under-relaxation; publish T for the level-1 broadcast
    if (p.mode == MCKPP_MODE_STEP && act) {
      U = lambda * U + (1 - lambda) * row(R_YU)[k];
      V = lambda * V + (1 - lambda) * row(R_YV)[k];
      T = lambda * T + (1 - lambda) * row(R_YT)[k];
      S = lambda * S + (1 - lambda) * row(R_YS)[k];
    }
    if (act) aT[k] = T;
  };
  auto A2 = [&]() {   // equation of state (+2 virtual slots); publish level-1 and virtual-slot values
    const double Sref = csrow()[CS_SREF];
    const double zm1 = c_zm[1];
    const double T1 = aT[1];
    zmk = c_zm[k];
    double Sin = S + Sref, Tin = T, Pin = -zmk;
    if (k == nzp1 + 1) { Sin = 0.0; Tin = T1; Pin = -zm1; }
    if (k == nzp1 + 2) { Sin = p.sice; Tin = T1; Pin = -zm1; }
    double s0;
    abk80_dev(Sin, Tin, Pin, talpha, sbeta, s0);
    const double rho = 1000. + s0;
    const double cp = cpsw_dev(Sin, Tin, Pin);
    buoy = div_fast(-p.grav * s0, 1000., 1. / 1000.);
    if (k == 1) { sc[X_RHO0] = rho; sc[X_CP0] = cp; sc[X_TALPHA0] = talpha; sc[X_SBETA0] = sbeta; }
    if (k == nzp1 + 1) sc[X_RHOH2O] = rho;
    if (k == nzp1 + 2) sc[X_RHOB] = rho;
    if (act) { aU[k] = U; aV[k] = V; aB[k] = buoy; }
    if (p.diag && maybe_final()) {   // what the last vmix leaves behind (types_transfer.F90:199-327)
      size_t o = rowoff() + k;
      if (act) { p.rho[o] = rho; p.cp[o] = cp; p.buoy[o] = buoy; p.talpha[o] = talpha; p.sbeta[o] = sbeta; }
      if (k == 1) { p.rho[o - 1] = rho; p.cp[o - 1] = cp; p.talpha[o - 1] = talpha; p.sbeta[o - 1] = sbeta; }
    }
    if constexpr (EXT) {
      if (act) { row(R_RHO)[k] = rho; row(R_CP)[k] = cp; }
      if (p.LDD && act) {   // neighbours for alphaDT, betaDS (T is already in aT)
        row(R_YV)[k] = talpha; row(R_RB)[k] = sbeta; row(R_GH)[k] = S;
      }
    }
  };
  double B0 = 0, B0sol = 0, ustar = 0;   // wave-uniform, recomputed identically by every wave of the slot
  auto A3 = [&]() {   // surface fluxes, reference-level loop, Ri pieces
    // the surface fluxes are slot-uniform: the lead wave computes and publishes them, the other
    // waves of the slot pick B0, B0sol and ustar up from the record in A5 (two barriers later)
    if (lead) {
      const double *cs = csrow();
      const double rho0 = first_lane(sc[X_RHO0]), cp0 = first_lane(sc[X_CP0]);
      const double talpha0 = first_lane(sc[X_TALPHA0]), sbeta0 = first_lane(sc[X_SBETA0]);
      const double rhoh2o = first_lane(sc[X_RHOH2O]), rhob = first_lane(sc[X_RHOB]);
      const double sflux1 = cs[CS_SFLUX1], sflux2 = cs[CS_SFLUX2], sflux3 = cs[CS_SFLUX3],
                   sflux4 = cs[CS_SFLUX4], sflux5 = cs[CS_SFLUX5], sflux6 = cs[CS_SFLUX6];
      const double Ssurf = cs[CS_SSURF];
      const double r_rho0 = rcp_refine(rho0), rho0cp0 = rho0 * cp0, r_rc = rcp_refine(rho0cp0);
      const double wU0_1 = first_lane(div_fast(-sflux1, rho0, r_rho0));
      const double wU0_2 = first_lane(div_fast(-sflux2, rho0, r_rho0));
      const double tau = __builtin_sqrt(sflux1 * sflux1 + sflux2 * sflux2) + 1.e-16;
      ustar = first_lane(__builtin_sqrt(div_fast(tau, rho0, r_rho0)));
      const double wX0_1 = first_lane(div_fast(div_fast(-sflux4, rho0, r_rho0), cp0, rcp_refine(cp0)));
      const double wX0_2 = first_lane(div_fast(Ssurf * sflux6, rhoh2o, rcp_refine(rhoh2o)) +
                                      div_fast((Ssurf - p.sice) * sflux5, rhob, rcp_refine(rhob)));
      B0 = first_lane(-p.grav * (talpha0 * wX0_1 - sbeta0 * wX0_2));
      B0sol = first_lane(div_fast(p.grav * talpha0 * sflux3, rho0cp0, r_rc));
      if (lane == 0) {
        sc[C_B0] = B0; sc[C_B0SOL] = B0sol; sc[C_USTAR] = ustar; sc[C_WU01] = wU0_1; sc[C_WU02] = wU0_2;
        sc[C_WX01] = wX0_1; sc[C_WX02] = wX0_2; sc[C_RHO0CP0] = rho0cp0; sc[C_RRC] = r_rc;
        if (ntime >= 1) sc[C_WXNT0] = div_fast(-sflux3 * p.swdk_tab[jer * p.ldc], rho0cp0, r_rc);
      }
    }
    const double zm1 = first_lane(c_zm[1]);
    const double U1 = first_lane(aU[1]), V1 = first_lane(aV[1]), Bu1 = first_lane(aB[1]);
    const double zref = eps01 * zmk, rzref = rcp_refine(zref);
    double wz = dmax2(zm1, zref);
    double ur = div_fast_guarded(U1 * wz, zref, rzref), vr = div_fast_guarded(V1 * wz, zref, rzref),
           br = div_fast(Bu1 * wz, zref, rzref);
    bool live = actz;
    double zk = zm1, Uk = U1, Vk = V1, Bk = Bu1;
    for (int kl = 1; kl <= nz; ++kl) {
      const double zk1 = first_lane(c_zm[kl + 1]), Uk1 = first_lane(aU[kl + 1]), Vk1 = first_lane(aV[kl + 1]),
                   Bk1 = first_lane(aB[kl + 1]);
      live = live && !(zref >= zk);
      if (!__any(live)) break;
      if (live) {
        const double dzk = zk - zk1, rdzk = first_lane(c_rdz[kl]);
        double wz2 = dmin2(zk - zk1, zk - zref);
        double del = div_fast(0.5 * wz2, dzk, rdzk);
        ur = ur - div_fast_guarded(wz2 * (Uk + del * (Uk1 - Uk)), zref, rzref);
        vr = vr - div_fast_guarded(wz2 * (Vk + del * (Vk1 - Vk)), zref, rzref);
        br = br - div_fast(wz2 * (Bk + del * (Bk1 - Bk)), zref, rzref);
      }
      zk = zk1; Uk = Uk1; Vk = Vk1; Bk = Bk1;
    }
    if constexpr (EXT) {
      alphaDT = 0.0; betaDS = 0.0;
      if (p.LDD) {   // verticalmixing_mod.F90:103-108
        alphaDT = 0.5 * (talpha + row(R_YV)[k + 1]) * (T - aT[k + 1]);
        betaDS = 0.5 * (sbeta + row(R_RB)[k + 1]) * (S - row(R_GH)[k + 1]);
      }
    }
    const double bk1 = aB[k + 1], uk1 = aU[k + 1], vk1 = aV[k + 1];
    Ritop = (zref - zmk) * (br - buoy);
    dbloc = buoy - bk1;
    dVsq = (ur - U) * (ur - U) + (vr - V) * (vr - V);
    shsq = (U - uk1) * (U - uk1) + (V - vk1) * (V - vk1);
    if (p.mode != MCKPP_MODE_STEP && k == nz) { sc[C_UREFNZ] = ur; sc[C_VREFNZ] = vr; }
    zdiff = zmk - c_zm[k + 1];
    const double shs = shsq + 1.e-16;
    Rig = div_fast(dbloc * zdiff, shs, rcp_refine(shs));
    if (actz) { aR[k] = Rig; aDb[k] = dbloc; }
    if (k == 1) aR[0] = 0.0;
    if (k == nzp1) aR[k] = 0.0;
    if (p.diag && maybe_final()) {
      size_t o = rowoff() + k;
      if (actz) { p.Rig[o] = Rig; p.dbloc[o] = dbloc; p.Shsq[o] = shsq; }
    }
  };
  auto A4 = [&]() {   // rimix + z121; interior diffusivity rows
    const double Riinfty = 0.8;
    double vm1 = aR[k - 1], vp1 = aR[k + 1];
    double wm1 = (k - 1 >= 1 && !((vm1 < 0.0) || (vm1 > Riinfty))) ? 1.0 : 0.0;
    double wp1 = (k + 1 <= nz && !((vp1 < 0.0) || (vp1 > Riinfty))) ? 1.0 : 0.0;
    double sm = wm1 * vm1 + 2. * Rig + wp1 * vp1;
    double wait = wm1 + 2.0 + wp1;
    sm = div_fast(sm, wait, wait == 3.0 ? 1. / 3. : (wait == 2.0 ? 0.5 : 0.25));
    double Rigg = dmax2(sm, 0.0);
    double ratio = dmin2(div_fast(Rigg, Riinfty, 1. / Riinfty), 1.0);
    double fri = (1.0 - ratio * ratio);
    fri = fri * fri * fri;
    dm_i = (0.0001 + fri * 0.005);
    ds_i = (0.00001 + fri * 0.005);
    if constexpr (EXT) {
      dt_i = ds_i;   // dift = difs, rimix_mod.F90:95-97 (the default build keeps one register for both)
      if (p.LDD) {   // ddmix_mod.F90:12-52
        const double Rrho0 = 1.9, dsfmax = 1.0e-4;
        const double aDT = alphaDT, bDS = betaDS;
        if ((aDT > bDS) && (bDS > 0.)) {
          double Rrho = dmin2(aDT / bDS, Rrho0);
          double rr = ((Rrho - 1) / (Rrho0 - 1));
          double diffdd = 1.0 - rr * rr;
          diffdd = dsfmax * diffdd * diffdd * diffdd;
          dt_i = dt_i + diffdd * 0.8 / Rrho;
          ds_i = ds_i + diffdd;
        } else if ((aDT < 0.0) && (bDS < 0.0) && (aDT < bDS)) {
          double Rrho = aDT / bDS;
          double diffdd = 1.5e-6 * 9.0 * 0.101 * mckpp_exp(4.6 * mckpp_exp(-0.54 * (1 / Rrho - 1)));
          double prandtl = 0.15 * Rrho;
          if (Rrho > 0.5) prandtl = (1.85 - 0.85 / Rrho) * Rrho;
          dt_i = dt_i + diffdd;
          ds_i = ds_i + prandtl * diffdd;
        }
      }
    }
    const double dt_l = EXT ? dt_i : ds_i;
    if (actz) { aDm[k] = dm_i; aDs[k] = ds_i; aDt[k] = dt_l; }
    if (k == nz) { aDm[k + 1] = dm_i; aDs[k + 1] = ds_i; aDt[k + 1] = dt_l; }
    if (k == 1) { aDm[0] = 0.0; aDs[0] = 0.0; aDt[0] = 0.0; }
  };
  auto A5 = [&]() {   // bldepth, level-parallel part
    if (!lead) { B0 = first_lane(sc[C_B0]); B0sol = first_lane(sc[C_B0SOL]); ustar = first_lane(sc[C_USTAR]); }
    const wscale_u wu = wscale_prepare_uniform(ustar);
    const double zm_kmp1 = first_lane(c_zm[nzp1]);
    double swf = p.swfrac_tab[jer * p.ldc + k];
    double bf = B0 + B0sol * (1. - swf);
    double st = 0.5 + dsign(0.5, bf + epsln16);
    double sg = st * 1. + (1. - st) * eps01;
    double wm, ws;
    wscale_dev(p, wu, sg, -zmk, bf, wm, ws);
    double dbm1 = aDb[k - 1];
    double bvsq = 0.5 * (div_fast(dbm1, c_zm[k - 1] - zmk, c_rdz[k - 1]) + div_fast(dbloc, zdiff, c_rdz[k]));
    double Vtsq = -zmk * ws * __builtin_sqrt(__builtin_fabs(bvsq)) * p.Vtc;
    const double rawden = dVsq + Vtsq + epsln16, bfa = __builtin_fabs(bf) + epsln16;
    double raw = div_fast(Ritop, rawden, rcp_refine(rawden));
    double dmo = div_fast(div_fast(cmonob * ustar * ustar * ustar, p.vonk, first_lane(c_misc[1])), bfa, rcp_refine(bfa));
    dmo = st * dmo - (1. - st) * zm_kmp1;
    if (k >= 2 && actz) { aR[k] = raw; aDmo[k] = dmo; }
    if (k == 1) { aR[1] = 0.0; aDmo[1] = -zm_kmp1; }
  };
  auto C1 = [&]() {   // first level with hmin < -zm(k), per wave; candidates to the slot record
    const double f = f_col;
    const double ocdepth = csrow()[CS_OCDEPTH];
    const double zm_kmp1 = first_lane(c_zm[nzp1]);
    const double fa = __builtin_fabs(f) + epsln16;
    const double hek = first_lane(div_fast(cekman * ustar, fa, rcp_refine(fa)));
    double swf = p.swfrac_tab[jer * p.ldc + k];
    double bf = B0 + B0sol * (1. - swf);
    double stab = 0.5 + dsign(0.5, bf + epsln16);
    double Rka = aR[k - 1], Rku = aR[k], dmoa = aDmo[k - 1], dmou = aDmo[k];
    double zkm1 = c_zm[k - 1];
    double hri = -zkm1 + (zkm1 - zmk) * (Ricr - Rka) / (Rku - Rka);
    double hmonob;
    if (dmou <= (-zmk)) {
      hmonob = (dmou - dmoa) / (zkm1 - zmk);
      hmonob = (dmou + hmonob * zmk) / (1. - hmonob);
    } else {
      hmonob = -zm_kmp1;
    }
    double hekman = stab * hek - (1. - stab) * zm_kmp1;
    double hmin = dmin2(dmin2(dmin2(hri, hmonob), hekman), -ocdepth);
    bool hit = (k >= 2) && actz && (hmin < -zmk);
    if (hit && !l_initflag && (hmin < -zkm1)) {
      double hmin2 = dmin2(dmin2(hri, hmonob), -ocdepth);
      if (hmin2 < -zmk) hmin = hmin2;
    }
    unsigned long long m = __ballot(hit);
    int cand = 0x7fffffff;
    double ch = 0.0;
    if (m != 0ull) {
      int src = __ffsll((long long)m) - 1;
      cand = src + 64 * sub + 1;
      ch = bcast(hmin, src);
    }
    if (lane == 0) { si[I_CAND_KBL + sub] = cand; sc[X_CAND_HBL + sub] = ch; }
  };
  auto C2 = [&]() {   // hbl/kbl, blmix, enhance, combine, bottom limits -> final diffusivities (registers)
    int kbl = nz;
    double hbl = first_lane(-c_zm[nz]);
#pragma unroll
    for (int s_ = 0; s_ < WPS; ++s_) {
      const int ck = si[I_CAND_KBL + s_];
      if (kbl == nz && ck != 0x7fffffff && ck <= nz) {   // the shallowest wave with a hit wins
        bool first = true;
#pragma unroll
        for (int t_ = 0; t_ < s_; ++t_) first = first && (si[I_CAND_KBL + t_] == 0x7fffffff);
        if (first) { kbl = ck; hbl = first_lane(sc[X_CAND_HBL + s_]); }
      }
    }
    hbl_pass = hbl;
    kbl_pass = kbl;
    if (!__any(k < kbl)) {
      // this wave lies entirely below the boundary layer: its levels keep the interior values
      // (kppmix_mod.F90:103-111) and need none of the boundary-layer scalars
      const double dt_l = EXT ? dt_i : ds_i;
      difm = dm_i; difs = ds_i; dift = dt_l; ghat = 0.;
      if (k >= nz) { difm = 0.0001; difs = 0.00001; dift = 0.00001; ghat = 0.0; }   // verticalmixing_mod.F90:151-159
      return;
    }
    const wscale_u wu = wscale_prepare_uniform(ustar);
    double bfsfc = swfrac_dev_wave(-1.0, hbl, jer, lane);
    bfsfc = B0 + B0sol * (1. - bfsfc);
    const double stable = first_lane(0.5 + dsign(0.5, bfsfc));
    bfsfc = first_lane(bfsfc + stable * epsln16);
    const double caseA = first_lane(0.5 + dsign(0.5, -c_zm[kbl] - 0.5 * c_hm[kbl] - hbl));
    double gat1[3], dat1[3], dkm1[3];
    const double r_hbl = first_lane(rcp_refine(hbl));
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
    {
      const double hk = c_hm[k];
      const double dt_l = EXT ? dt_i : ds_i;
      double wm, ws;
      double sig = div_fast(-zmk + 0.5 * hk, hbl, r_hbl);
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
      if (k == kbl - 1 && k <= nz - 1) {
        double delta = div_fast(hbl + zmk, zmk - c_zm[k + 1], c_rdz[k]);
        double omd = 1. - delta;
        double dkmp5 = caseA * dm_i + (1. - caseA) * b0;
        double dstar = (omd * omd) * dkm1[0] + (delta * delta) * dkmp5;
        b0 = omd * dm_i + delta * dstar;
        dkmp5 = caseA * ds_i + (1. - caseA) * b1;
        dstar = (omd * omd) * dkm1[1] + (delta * delta) * dkmp5;
        b1 = omd * ds_i + delta * dstar;
        dkmp5 = caseA * dt_l + (1. - caseA) * b2;
        dstar = (omd * omd) * dkm1[2] + (delta * delta) * dkmp5;
        b2 = omd * dt_l + delta * dstar;
        gh = (1. - caseA) * gh;
      }
      if (k < kbl) { difm = b0; difs = b1; dift = b2; ghat = gh; }
      else { difm = dm_i; difs = ds_i; dift = dt_l; ghat = 0.; }
      if (k >= nz) { difm = 0.0001; difs = 0.00001; dift = 0.00001; ghat = 0.0; }
    }
  };
  auto C3 = [&]() {   // final diffusivities into the rows the Thomas lanes (and finalize) read
    if (act) { aDm[k] = difm; aDs[k] = difs; aDt[k] = dift; aGh[k] = ghat; }
  };
  // ---- optional terms of the T and S right-hand sides (ocnint_mod.F90:97-215), level k of this lane:
  // relaxation / flux corrections / prescribed advection (rhsmod, solvers.F90:176-335, salinity only)
  auto ext_rhs = [&](int kmixe, double To_k, double So_k, double &rhsT, double &rhsS) {
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
    xt = tinc;
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
  auto C4 = [&]() {   // right-hand sides of U, T, S
    const double *cs = csrow();
    const double f = f_col;
    const double Uo = ld_old(p.U), Vo = ld_old(p.V), To = ld_old(p.T), So = ld_old(p.S);
    const double Uo_np = old_bottom(p.U), To_np = old_bottom(p.T), So_np = old_bottom(p.S);
    const double dto = p.dto, tri1_nz = first_lane(c_t1[nz]), hm1 = first_lane(c_hm[1]);
    const double wU0_1 = first_lane(sc[C_WU01]), wX0_1 = first_lane(sc[C_WX01]), wX0_2 = first_lane(sc[C_WX02]),
                 wXNT0 = first_lane(sc[C_WXNT0]);
    const double rho0cp0 = first_lane(sc[C_RHO0CP0]), r_rc = first_lane(sc[C_RRC]), sflux3 = cs[CS_SFLUX3];
    const double r_hm1 = first_lane(c_misc[0]);
    double *yU = row(R_YU), *yT = row(R_YT), *yS = row(R_YS);
    if (actz) {
      const double dt_m1 = aDt[k - 1], ds_m1 = aDs[k - 1];
      const double gh_m1 = (k >= 2) ? aGh[k - 1] : 0.0;
      double wxnt = 0.0, wxnt_m1 = 0.0;
      if (ntime >= 1) {
        wxnt = div_fast(-sflux3 * p.swdk_tab[jer * p.ldc + k], rho0cp0, r_rc);
        wxnt_m1 = div_fast(-sflux3 * p.swdk_tab[jer * p.ldc + k - 1], rho0cp0, r_rc);
      }
      double rhsU;
      if (k == 1) rhsU = Uo + dto * (f * .5 * (Vo + V) - div_fast(wU0_1, hm1, r_hm1));
      else rhsU = Uo + dto * f * .5 * (Vo + V);
      if (k == nz) rhsU = rhsU + tri1_nz * difm * Uo_np;
      double rhsT;
      const double dtohk = c_dtohk[k];
      if (k == 1) rhsT = To + dtohk * (wX0_1 * dift * ghat - wX0_1 * 1.0 + wxnt - wXNT0);
      else rhsT = To + dtohk * (wX0_1 * (dift * ghat - dt_m1 * gh_m1) + wxnt - wxnt_m1);
      if (k == nz && nz > 1) rhsT = rhsT + To_np * tri1_nz * dift;
      double rhsS;
      if (k == 1) rhsS = So + dtohk * (wX0_2 * difs * ghat - wX0_2 * 1.0 + 0.0 - 0.0);
      else rhsS = So + dtohk * (wX0_2 * (difs * ghat - ds_m1 * gh_m1) + 0.0 - 0.0);
      if (k == nz && nz > 1) rhsS = rhsS + So_np * tri1_nz * difs;
      if constexpr (EXT) ext_rhs(kbl_pass, To, So, rhsT, rhsS);
      yU[k] = rhsU; yT[k] = rhsT; yS[k] = rhsS;
    }
    if (k == nzp1) {
      yU[k] = Uo; yT[k] = To; yS[k] = So;
      if constexpr (EXT) { double t = 0.0, s2 = 0.0; ext_rhs(kbl_pass, To, So, t, s2); }   // corrections of level nzp1, ocnint_mod.F90:153-160, 207-213
    }
  };
  auto E = [&]() {   // V right-hand side with the new U
    const double Uo = ld_old(p.U), Vo = ld_old(p.V);
    const double Vo_np = old_bottom(p.V);
    const double dto = p.dto, tri1_nz = first_lane(c_t1[nz]), hm1 = first_lane(c_hm[1]), f = f_col,
                 wU0_2 = first_lane(sc[C_WU02]), r_hm1 = first_lane(c_misc[0]);
    const double *yU = row(R_YU);
    double *yV = row(R_YV);
    if (actz) {
      const double un = yU[k];
      double rhsV;
      if (k == 1) rhsV = Vo - dto * (f * .5 * (Uo + un) + div_fast(wU0_2, hm1, r_hm1));
      else rhsV = Vo - dto * f * .5 * (Uo + un);
      if (k == nz) rhsV = rhsV + tri1_nz * aDm[k] * Vo_np;
      yV[k] = rhsV;
    } else if (act) {
      yV[k] = Vo;
    }
  };

  // serial phases (identical to k_column_wg, rows are just longer)
  // ---- serial phases (mckpp_sweeps.h), run by one wave for all slots of the workgroup ----
  auto scan_rib = [&]() { if (wv == 0) serial_scan_rib<W>(slots, SS, NA, nz, sact, lane); };
  auto thomas_uts = [&]() { if (wv == 0) serial_thomas_uts<W>(slots, SS, NA, nz, c_t0, c_t1, sact, sbad, lane); };
  auto thomas_v = [&]() { if (wv == 0) serial_thomas_v<W>(slots, SS, NA, nz, c_t0, sact, lane); };

  // ---- ocnstep control after a pass (ocnstep_mod.F90:122-192); every wave of the slot
  // takes the same decision from the same replicated values
  enum { F_NONE = 0, F_TRAP = 1, F_FINAL = 2 };
  int fin = F_NONE;
  auto G = [&]() {
    fin = F_NONE;
    if (p.mode != MCKPP_MODE_INIT && sbad[slot]) status |= 1;
    ++npass;
    if (p.mode != MCKPP_MODE_STEP) { fin = F_FINAL; return; }
    ++npass_try;
    if (npass_try <= 3) { hmixe = hbl_pass; return; }   // compulsory passes
    hmixn = hbl_pass;
    kmixn = kbl_pass;
    double tol = p.hmixtolfrac * c_hm[kmixn];
    if (kmixn == nzp1) tol = p.hmixtolfrac * c_hm[nz];
    if (__builtin_fabs(hmixn - hmixe) > tol) iconv = 0;
    else iconv = iconv + 1;
    if (iconv < 3) {
      if (npass_try < p.itermax) { hmixe = hmixn; return; }
      else if (hmixn > hmixe) { hmixe = hmixn; return; }
    }
    if (npass_try > (p.itermax + 1)) status |= 2;
    fin = F_TRAP;
  };

  // ---- persistent loop -----------------------------------------------------
  const bool do_ocnint = p.mode != MCKPP_MODE_INIT;
  for (;;) {
    k = k0;
    asm volatile("" : "+v"(k));
    // refill round (only when some slot of the workgroup is empty)
    if (__syncthreads_or(state == S_EMPTY ? 1 : 0)) {
      if (state == S_EMPTY && lead && lane == 0) si[I_COL] = atomicAdd(p.qhead, 1);
      __syncthreads();
      if (state == S_EMPTY) load_column();
    }
    const bool active = state == S_ACTIVE;
    if (lead && lane == 0) { sact[slot] = active ? 1 : 0; sbad[slot] = 0; }
    if (!__syncthreads_or(active ? 1 : 0)) break;
    if (active) A1();
    __syncthreads();
    if (active) A2();
    __syncthreads();
    if (active) A3();
    __syncthreads();
    if (active) A4();
    __syncthreads();
    if (active) A5();
    __syncthreads();
    scan_rib();
    __syncthreads();
    if (active) C1();
    __syncthreads();
    if (active) C2();
    __syncthreads();
    if (active) C3();
    __syncthreads();
    if (active && do_ocnint) C4();
    __syncthreads();
    if (do_ocnint) thomas_uts();
    __syncthreads();
    if (active && do_ocnint) E();
    __syncthreads();
    if (do_ocnint) thomas_v();
    __syncthreads();
    fin = F_NONE;
    if (active) G();
    if (!__syncthreads_or(fin != F_NONE ? 1 : 0)) continue;

    // ---- finish round: instability trap (STEP), then retry or finalize -------
    // All waves of the workgroup walk the same barriers; only flagged slots work.
    if (fin != F_NONE && do_ocnint && act) {   // U,V,T,S <- what the last ocnint returned
      U = row(R_YU)[k]; V = row(R_YV)[k]; T = row(R_YT)[k]; S = row(R_YS)[k];
    }
    if (fin == F_TRAP && act) aT[k] = T;
    __syncthreads();
    if (fin == F_TRAP) {   // ocnstep_mod.F90:200-207
      const double tk1 = aT[k + 1];
      const bool v = actz && (__builtin_fabs(U) >= 10 || __builtin_fabs(V) >= 10 || __builtin_fabs(T - tk1) >= 10);
      const int nv = __popcll(__ballot(v));
      if (lane == 0) si[I_NVIOL + sub] = nv;
    }
    __syncthreads();
    if (fin == F_TRAP) {
      comp_flag = 0;
      int nviol = 0;
#pragma unroll
      for (int s_ = 0; s_ < WPS; ++s_) nviol += si[I_NVIOL + s_];
      if (nviol > 0) {
        comp_flag = 1;
        for (int i = 0; i < nviol; ++i) f_col = f_col * 1.01;
      }
      if (!comp_flag && act) {   // :208-219
        const double Uo = ld_old(p.U), Vo = ld_old(p.V), To = ld_old(p.T), So = ld_old(p.S);
        const double hk = c_hm[k];
        row(R_YU)[k] = (U - Uo) * (U - Uo) * hk / p.dm_nz;
        row(R_YT)[k] = (V - Vo) * (V - Vo) * hk / p.dm_nz;
        row(R_YS)[k] = (T - To) * (T - To) * hk / p.dm_nz;
        row(R_GM)[k] = (S - So) * (S - So) * hk / p.dm_nz;
      }
    }
    __syncthreads();
    if (fin == F_TRAP && !comp_flag && lead) {
      bool over = false;
      if (lane < 4) {
        const double *t = row(R_YU + lane);
        double sum = 0.;
        for (int q = 1; q <= nzp1; ++q) sum = sum + t[q];
        sum = __builtin_sqrt(sum);
        over = sum >= 1.0;
      }
      const int nover = __popcll(__ballot(over));
      if (lane == 0) si[I_OVER] = nover;
    }
    __syncthreads();
    if (fin == F_TRAP) {
      if (!comp_flag) {
        const int nover = si[I_OVER];
        if (nover > 0) {
          comp_flag = 1;
          for (int i = 0; i < nover; ++i) f_col = f_col * 1.01;
        }
      }
      if (comp_flag) status |= 4;
      nreset = nreset + 1;
      if (nreset > 10) status |= 8;
      if (comp_flag && nreset <= 10) { extrapolate(); fin = F_NONE; }   // retry, ocnstep_mod.F90:89
      else fin = F_FINAL;
    }
    // the diagnostic fluxes need the k+1 neighbours of the final profiles
    if (fin == F_FINAL && p.diag && p.mode != MCKPP_MODE_PASS && act) {
      row(R_YU)[k] = U; row(R_YT)[k] = V; row(R_YS)[k] = T; row(R_GM)[k] = S;
    }
    __syncthreads();
    if (fin == F_FINAL) {
      double *cs = csrow();
      const size_t ro = rowoff();
      if (p.diag) {
        const double wX0_1 = first_lane(sc[C_WX01]), wX0_2 = first_lane(sc[C_WX02]);
        const double *tU = row(R_YU), *tV = row(R_YT), *tT = row(R_YS), *tS = row(R_GM);
        const double rho0cp0 = first_lane(sc[C_RHO0CP0]), sflux3 = cs[CS_SFLUX3];
        const size_t o = ro + k;
        const double dfm = aDm[k], dfs = aDs[k], dft = aDt[k], gh = aGh[k];
        if (act) { p.difm[o] = dfm; p.difs[o] = dfs; p.dift[o] = dft; }
        if (actz) {
          p.ghat[o] = gh;
          p.wXNT1[o] = (ntime >= 1) ? -sflux3 * p.swdk_tab[jer * p.ldc + k] / rho0cp0 : 0.0;
          if (p.mode != MCKPP_MODE_PASS) {
            double deltaz = 0.5 * (c_hm[k] + c_hm[k + 1]);
            double uk1 = tU[k + 1], vk1 = tV[k + 1], tk1 = tT[k + 1], sk1 = tS[k + 1];
            double wX1 = -dfs * ((T - tk1) / deltaz - gh * wX0_1);
            double wX2 = -dfs * ((S - sk1) / deltaz - gh * wX0_2);
            if (p.LDD) wX1 = -dft * ((T - tk1) / deltaz - gh * wX0_1);
            p.wX1[o] = wX1; p.wX2[o] = wX2;
            p.wX3[o] = p.grav * (talpha * wX1 - sbeta * wX2);
            p.wU1[o] = -dfm * (U - uk1) / deltaz;
            p.wU2[o] = -dfm * (V - vk1) / deltaz;
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
    // ---- outputs of the column-step.  STEP: ocnstep_mod.F90:305-353 + check_profile
    // (overrides.F90:42-125); the optional parts of check_profile count over all levels of the
    // column, i.e. over the slot's waves, so they go through the slot record between barriers
    // that every wave of the workgroup walks (EXT build only).
    const bool fstep = fin == F_FINAL && p.mode == MCKPP_MODE_STEP;
    double reset_out = 0.0, dampu = 0.0, dampv = 0.0, freeze = 0.0;
    int l_ocean = 0;
    if (fstep && k == 1) {   // level-1 references, before any override touches the profiles
      double *cs = csrow();
      cs[CS_UREF] = U; cs[CS_VREF] = V; cs[CS_TREF] = T;
      cs[CS_SSURF] = p.L_SSref ? cs[CS_SSREF] : S + cs[CS_SREF];
    }
    if constexpr (EXT) {
      if (fstep && p.L_DAMP_CURR) {   // ocnstep_mod.F90:317-340
        const double rr = (double)p.dt_uvdamp * (86400. / p.dto);
        double a = 0.99 * __builtin_fabs(U), b = (U * U) / rr;
        const int nu = __popcll(__ballot(act && (b < a)));
        U = U - dsign(dmin2(a, b), U);
        a = 0.99 * __builtin_fabs(V); b = (V * V) / rr;
        const int nv = __popcll(__ballot(act && (b < a)));
        V = V - dsign(dmin2(a, b), V);
        if (lane == 0) { si[I_NU + sub] = nu; si[I_NV + sub] = nv; }
      }
      __syncthreads();
    }
    if (fstep) {
      const size_t o = rowoff() + (k - 1);
      if constexpr (EXT) {
        if (p.L_DAMP_CURR) {
          const double inc = 1.0 / (double)nzp1;
          int nu = 0, nv = 0;
#pragma unroll
          for (int s_ = 0; s_ < WPS; ++s_) { nu += si[I_NU + s_]; nv += si[I_NV + s_]; }
          for (int i = 0; i < nu; ++i) dampu = dampu + inc;
          for (int i = 0; i < nv; ++i) dampv = dampv + inc;
        }
      }
      old = newi;
      newi = 1 - old;
      if (act) { p.Us[newi][o] = U; p.Vs[newi][o] = V; p.Ts[newi][o] = T; p.Ss[newi][o] = S; }
      reset_out = (double)nreset;
      if (comp_flag) {   // overrides.F90:57-78
        if (EXT && p.clim_present && act) { T = p.ocnT_clim[o]; S = p.sal_clim[o]; }
        if (act) { U = p.U_init[o]; V = p.V_init[o]; }
        reset_out = 999.;
      }
      if constexpr (EXT) {
        l_ocean = p.ci[(size_t)col * MCKPP_CI + CI_LOCEAN];
        freeze = csrow()[CS_FREEZE];
        if (l_ocean && p.L_NO_FREEZE) {   // :85-94
          const bool cold = act && (T < -1.8);
          if (cold) { xt = xt + (-1.8 - T); T = -1.8; }
          const int nf = __popcll(__ballot(cold));
          if (lane == 0) si[I_NF + sub] = nf;
          if (act) p.tinc_fcorr[rowoff() + k] = xt;
        }
      }
    }
    if constexpr (EXT) {
      __syncthreads();
      const bool iso = fstep && l_ocean && p.L_NO_ISOTHERM;
      if (fstep && l_ocean && p.L_NO_FREEZE) {
        const double inc = 1.0 / (double)nzp1;
        int nf = 0;
#pragma unroll
        for (int s_ = 0; s_ < WPS; ++s_) nf += si[I_NF + s_];
        for (int i = 0; i < nf; ++i) freeze = freeze + inc;
      }
      if (iso && act) row(R_YU)[k] = T;   // :102-120
      __syncthreads();
      if (iso && k >= 2 && act) {
        const double dz = c_zm[k] - c_zm[k - 1];
        row(R_YT)[k] = __builtin_fabs((T - row(R_YU)[k - 1])) * dz;
        row(R_YS)[k] = dz;
      }
      __syncthreads();
      if (iso) {
        const double *tD = row(R_YT), *tZ = row(R_YS);
        double dtdz_total = 0., dz_total = 0.;
        for (int q = 2; q <= p.iso_bot; ++q) {
          dtdz_total = dtdz_total + tD[q];
          dz_total = dz_total + tZ[q];
        }
        dtdz_total = dtdz_total / dz_total;
        if (__builtin_fabs(dtdz_total) < p.iso_thresh) {
          if (act) { const size_t o = rowoff() + (k - 1); T = p.ocnT_clim[o]; S = p.sal_clim[o]; }
          reset_out = (-1.) * reset_out;
        }
      } else if (fstep) {
        reset_out = 0.0;   // :121-123
      }
    } else {
      reset_out = 0.0;     // :121-123 (no isotherm check in the default physics)
    }
    if (fin == F_FINAL) {
      double *cs = csrow();
      int *ci = p.ci + (size_t)col * MCKPP_CI;
      const size_t o = rowoff() + (k - 1);
      if (p.mode == MCKPP_MODE_STEP) {
        if (act) { p.U[o] = U; p.V[o] = V; p.T[o] = T; p.S[o] = S; }
        if (k == 1) {
          cs[CS_HMIX] = hmixn;
          cs[CS_KMIX] = (double)kmixn;
          cs[newi ? CS_HMIXD1 : CS_HMIXD0] = hmixn;
          cs[CS_RESET] = reset_out;
          cs[CS_DAMPU] = dampu; cs[CS_DAMPV] = dampv;
          if constexpr (EXT) cs[CS_FREEZE] = freeze;
          ci[CI_OLD] = old; ci[CI_NEW] = newi;
          ci[CI_STATUS] = status; ci[CI_NPASS] = npass;
        }
      } else if (p.mode == MCKPP_MODE_INIT) {
        if (act) {
          p.Us[0][o] = U; p.Us[1][o] = U; p.Vs[0][o] = V; p.Vs[1][o] = V;
          p.Ts[0][o] = T; p.Ts[1][o] = T; p.Ss[0][o] = S; p.Ss[1][o] = S;
        }
        if (k == 1) {
          cs[CS_HMIX] = hbl_pass;
          cs[CS_KMIX] = (double)kbl_pass;
          cs[CS_TREF] = T;
          cs[CS_UREF] = sc[C_UREFNZ]; cs[CS_VREF] = sc[C_VREFNZ];
          cs[CS_HMIXD0] = hbl_pass; cs[CS_HMIXD1] = hbl_pass;
          ci[CI_OLD] = 0; ci[CI_NEW] = 1; ci[CI_INITFLAG] = 0;
          ci[CI_STATUS] = status; ci[CI_NPASS] = npass;
        }
      } else {
        if (act) { p.U[o] = U; p.V[o] = V; p.T[o] = T; p.S[o] = S; }
        if (k == 1) {
          cs[CS_HMIX] = hbl_pass;
          cs[CS_KMIX] = (double)kbl_pass;
          cs[CS_UREF] = sc[C_UREFNZ]; cs[CS_VREF] = sc[C_VREFNZ];
          ci[CI_STATUS] = status; ci[CI_NPASS] = npass;
        }
      }
      state = S_EMPTY;
    }
    // the barrier at the top of the loop separates these reads of the slot's rows/record from
    // the next column's load
  }
}

template <int WPS, int W, bool EXT>
size_t mw_lds_bytes()
{
  return (size_t)(6 * mw_na<WPS>() + W * mw_slot_stride<WPS, EXT>() + W * C_COUNT) * sizeof(double) +
         (size_t)(W * I_COUNT + 2 * W) * sizeof(int);
}

template <int WPS, int W, int MINW, bool EXT = false>
hipError_t launch_mw(const mckpp_kparams &p, const mckpp_kparams *dp, int nblocks, hipStream_t stream)
{
  const size_t lds = mw_lds_bytes<WPS, W, EXT>();
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_column_mw<WPS, W, MINW, EXT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  static int max_blocks = -1;
  if (max_blocks < 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(k_column_mw<WPS, W, MINW, EXT>), 64 * WPS * W, lds) != hipSuccess) nb = 0;
    max_blocks = nb;
  }
  g_mckpp_last_launch = {nblocks, 64 * WPS * W, max_blocks, lds};
  hipLaunchKernelGGL((k_column_mw<WPS, W, MINW, EXT>), dim3((unsigned)nblocks), dim3(64 * WPS * W), lds, stream, dp,
                     p.ntime);
  return hipGetLastError();
}

}  // namespace

// Deep columns (nzp1+2 > 64).  MCKPP_MW=<W>x<blocks per CU>[x<min waves per SIMD>] overrides the
// geometry of the default-physics build (experiments).
hipError_t mckpp_launch_column_kernel_mw(const mckpp_kparams &p, const mckpp_kparams *dp, int num_cu,
                                         hipStream_t stream)
{
  if (p.ncol <= 0) return hipSuccess;
  const int wps = (p.nzp1 + 2 + 63) / 64;
  if (p.ext) {   // optional-physics build: two more LDS rows per slot, 4 workgroups per CU
    const int W = (wps == 3) ? 1 : (wps == 2) ? 2 : 4;
    int nblocks = num_cu * 4;
    const int groups = (p.ncol + W - 1) / W;
    if (nblocks > groups) nblocks = groups;
    if (nblocks < 1) nblocks = 1;
    switch (wps) {
      case 1: return launch_mw<1, 4, 4, true>(p, dp, nblocks, stream);
      case 2: return launch_mw<2, 2, 4, true>(p, dp, nblocks, stream);
      case 3: return launch_mw<3, 1, 4, true>(p, dp, nblocks, stream);
      default: return hipErrorInvalidValue;
    }
  }
  static int envW = -1, envB = 0, envM = 0;
  if (envW < 0) {
    envW = 0;
    if (const char *e = getenv("MCKPP_MW")) {
      int w = 0, b = 0, m = 0;
      if (sscanf(e, "%dx%dx%d", &w, &b, &m) >= 1) { envW = w; envB = b; envM = m; }
    }
  }
  // defaults (measured, 1e5 columns): 128 VGPRs -> 4 waves per SIMD, 16 waves per CU
  //   2 waves per column: 2 columns x 4 workgroups per CU;  3 waves per column: 1 column x 5
  int W = (wps == 1) ? 4 : (wps == 2) ? 2 : 1;
  int per_cu = (wps == 1) ? 4 : (wps == 2) ? 4 : 5;
  int minw = 4;
  if (envW > 0) W = envW;
  if (envB > 0) per_cu = envB;
  if (envM > 0) minw = envM;
  int nblocks = num_cu * per_cu;
  const int groups = (p.ncol + W - 1) / W;
  if (nblocks > groups) nblocks = groups;
  if (nblocks < 1) nblocks = 1;
#define MW_CASE(WPS_, W_, M_) \
  if (wps == WPS_ && W == W_ && minw == M_) return launch_mw<WPS_, W_, M_>(p, dp, nblocks, stream);
#ifdef MCKPP_MW_PROBE
  MW_CASE(2, 2, 3)
#else
  MW_CASE(1, 4, 4)
  MW_CASE(2, 2, 4) MW_CASE(2, 4, 4) MW_CASE(2, 2, 3)
  MW_CASE(3, 1, 4) MW_CASE(3, 2, 4)
#endif
#undef MW_CASE
  return hipErrorInvalidValue;
}
