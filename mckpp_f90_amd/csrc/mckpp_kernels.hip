// mckpp_kernels.hip - gfx950 kernels for MC-KPP's per-column physics step.
//
// One 64-lane wavefront owns one water column: lane l holds grid levels
// k = l+1 (+64 for columns deeper than 62 levels), so the level-parallel work
// (equation of state, Richardson mixing, boundary-layer shape functions, the
// tridiagonal coefficients and right-hand sides) runs 64 wide out of
// registers, neighbour levels and the short serial recurrences (bulk-Ri
// running maximum, Thomas sweeps) go through the wave's private LDS rows, and
// the whole semi-implicit iteration of ocnstep stays on chip: HBM is touched
// once to load the column and once to store it.
//
// Arithmetic is written operation-for-operation in the reference's
// expression order and built with -ffp-contract=off, so results are bitwise
// reproducible against the CPU oracle (tests/), which restates
//   src/mckpp_physics_ocnstep_mod.F90, _verticalmixing_*.F90,
//   _state_equations.F90, _ocnint_mod.F90, _solvers.F90, _swfrac_mod.F90,
//   mckpp_fluxes_mod.F90:93-137, _overrides.F90:42-125.
// Function-level citations are given at each device function.
#include "mckpp_colmath.h"

namespace {

using namespace mckpp_dev;

// ---------------------------------------------------------------------------
// The column kernel.  LPL = levels per lane; block = one wavefront.
// ---------------------------------------------------------------------------
enum {
  A_ZM = 0, A_HM, A_DM, A_DS, A_DT, A_GH,   // persistent across a pass
  A_TH0,                                    // Thomas region: 3 systems x {cu,cc,cl,rhs,gam}
  A_RHV = A_TH0 + 15, A_BETM, A_COUNT
};
// vmix scratch rows alias the Thomas region (dead before ocnint builds it)
enum { A_U = A_TH0, A_V, A_B, A_R, A_DB, A_DMO, A_T };

template <int LPL>
struct lvl_regs {
  double U[LPL], V[LPL], T[LPL], S[LPL];      // current iterate
  double Uo[LPL], Vo[LPL], To[LPL], So[LPL];  // start-of-step profiles
  double Ux[LPL], Vx[LPL], Tx[LPL], Sx[LPL];  // under-relaxation memory
  double zm[LPL], hm[LPL], tri0[LPL], tri1[LPL], swf[LPL], swdk[LPL], swdkm[LPL];
  // outputs of the latest vmix pass
  double rho[LPL], cp[LPL], talpha[LPL], sbeta[LPL], buoy[LPL];
  double difm[LPL], difs[LPL], dift[LPL], ghat[LPL], Rig[LPL], dbloc[LPL], shsq[LPL], wxnt[LPL];
};

struct pass_scalars {
  double wU0_1, wU0_2, wX0_1, wX0_2, wX0_3, wXNT0, rhoh2o, uref_nz, vref_nz;
};

template <int LPL>
__global__ __launch_bounds__(64) void k_column(const mckpp_kparams p)
{
  extern __shared__ double lds[];
  constexpr int NA = 64 * LPL + 8;
  const int lane = threadIdx.x;
  const int col = blockIdx.x;
  const int nz = p.nz, nzp1 = p.nzp1;
  auto row = [&](int a) -> double * { return lds + a * NA; };
  double *c_zm = row(A_ZM), *c_hm = row(A_HM);
  double *aDm = row(A_DM), *aDs = row(A_DS), *aDt = row(A_DT), *aGh = row(A_GH);
  double *aU = row(A_U), *aV = row(A_V), *aB = row(A_B), *aR = row(A_R), *aDb = row(A_DB),
         *aDmo = row(A_DMO), *aT = row(A_T);

  for (int i = lane; i < NA; i += 64) {
    c_zm[i] = p.zm[i];
    c_hm[i] = p.hm[i];
  }
  __syncthreads();

  lvl_regs<LPL> r;
  int kk[LPL];
  bool act[LPL], actz[LPL];
  const size_t rowoff = (size_t)col * p.ld;
  double *cs = p.cs + (size_t)col * MCKPP_CS;
  int *ci = p.ci + (size_t)col * MCKPP_CI;

  int old = ci[CI_OLD], newi = ci[CI_NEW];
  const int jer = ci[CI_JERLOV];
  int l_initflag = ci[CI_INITFLAG];
  int status = 0;
  if (old < 0 || old > 1) { old = newi; status |= 16; }
  if (newi < 0 || newi > 1) { newi = old; status |= 16; }
  double f = cs[CS_F];
  double Ssurf = cs[CS_SSURF];
  const double Sref = cs[CS_SREF], SSref = cs[CS_SSREF], ocdepth = cs[CS_OCDEPTH];
  const double sflux1 = cs[CS_SFLUX1], sflux2 = cs[CS_SFLUX2], sflux3 = cs[CS_SFLUX3],
               sflux4 = cs[CS_SFLUX4], sflux5 = cs[CS_SFLUX5], sflux6 = cs[CS_SFLUX6];

  FORJ {
    int jj = lane + 64 * j;
    int k = jj + 1;
    kk[j] = k;
    act[j] = k <= nzp1;
    actz[j] = k <= nz;
    r.zm[j] = c_zm[k];
    r.hm[j] = c_hm[k];
    r.tri0[j] = p.tri0[k];
    r.tri1[j] = p.tri1[k];
    r.swf[j] = p.swfrac_tab[jer * p.ldc + k];
    r.swdk[j] = p.swdk_tab[jer * p.ldc + k];
    r.swdkm[j] = p.swdk_tab[jer * p.ldc + k - 1];
    r.U[j] = act[j] ? p.U[rowoff + jj] : 0.0;
    r.V[j] = act[j] ? p.V[rowoff + jj] : 0.0;
    r.T[j] = act[j] ? p.T[rowoff + jj] : 0.0;
    r.S[j] = act[j] ? p.S[rowoff + jj] : 0.0;
    r.Uo[j] = r.U[j]; r.Vo[j] = r.V[j]; r.To[j] = r.T[j]; r.So[j] = r.S[j];
    r.Ux[j] = r.U[j]; r.Vx[j] = r.V[j]; r.Tx[j] = r.T[j]; r.Sx[j] = r.S[j];
    r.wxnt[j] = 0.0;
  }
  const double zm1 = c_zm[1], hm1 = c_hm[1], zm_kmp1 = c_zm[nzp1];
  const int lane_nz = (nz - 1) & 63, j_nz = (nz - 1) >> 6;       // owner of level nz
  const int lane_np = (nzp1 - 1) & 63, j_np = (nzp1 - 1) >> 6;   // owner of level nzp1
  const int lane_v1 = nzp1 & 63, j_v1 = nzp1 >> 6;               // virtual EOS slot nzp1+1
  const int lane_v2 = (nzp1 + 1) & 63, j_v2 = (nzp1 + 1) >> 6;   // virtual EOS slot nzp1+2

  pass_scalars ps;
  ps.wXNT0 = 0.0;

  // ---- one vmix (+ ocnint) pass -------------------------------------------
  // verticalmixing_mod.F90:14-161, kppmix_mod.F90:25-126 and callees,
  // ocnint_mod.F90:19-221, solvers.F90:14-161.
  auto pass = [&](bool do_ocnint, double &hbl_out, int &kbl_out) {
    // -- equation of state on every level (+2 virtual slots for the fresh
    //    water / brine surface densities, verticalmixing_mod.F90:52-55)
    const double T1 = first_lane(r.T[0]);
    double sig0v[LPL];
    FORJ {
      int k = kk[j];
      double Sin = r.S[j] + Sref, Tin = r.T[j], Pin = -r.zm[j];
      if (k == nzp1 + 1) { Sin = 0.0; Tin = T1; Pin = -zm1; }
      if (k == nzp1 + 2) { Sin = p.sice; Tin = T1; Pin = -zm1; }
      double al, be, s0;
      abk80_dev(Sin, Tin, Pin, al, be, s0);
      sig0v[j] = s0;
      r.rho[j] = 1000. + s0;
      r.cp[j] = cpsw_dev(Sin, Tin, Pin);
      r.talpha[j] = al;
      r.sbeta[j] = be;
      r.buoy[j] = -p.grav * s0 / 1000.;
    }
    double rhoh2o = 0, rhob = 0;
    FORJ {
      double a = bcast(r.rho[j], lane_v1), b = bcast(r.rho[j], lane_v2);
      if (j == j_v1) rhoh2o = a;
      if (j == j_v2) rhob = b;
    }
    ps.rhoh2o = rhoh2o;
    const double rho0 = first_lane(r.rho[0]), cp0 = first_lane(r.cp[0]);
    const double talpha0 = first_lane(r.talpha[0]), sbeta0 = first_lane(r.sbeta[0]);

    // -- ntflux (fluxes_mod.F90:110-116) and surface kinematic fluxes (:81-100)
    double wxnt_m1[LPL];   // wXNT(k-1,1)
    FORJ wxnt_m1[j] = 0.0;
    if (p.ntime >= 1) {
      FORJ {
        r.wxnt[j] = -sflux3 * r.swdk[j] / (rho0 * cp0);
        wxnt_m1[j] = -sflux3 * r.swdkm[j] / (rho0 * cp0);
      }
      ps.wXNT0 = first_lane(wxnt_m1[0]);
    }
    ps.wU0_1 = -sflux1 / rho0;
    ps.wU0_2 = -sflux2 / rho0;
    const double tau = __builtin_sqrt(sflux1 * sflux1 + sflux2 * sflux2) + 1.e-16;
    const double ustar = __builtin_sqrt(tau / rho0);
    ps.wX0_1 = -sflux4 / rho0 / cp0;
    ps.wX0_2 = Ssurf * sflux6 / rhoh2o + (Ssurf - p.sice) * sflux5 / rhob;
    const double B0 = -p.grav * (talpha0 * ps.wX0_1 - sbeta0 * ps.wX0_2);
    ps.wX0_3 = -B0;
    const double B0sol = p.grav * talpha0 * sflux3 / (rho0 * cp0);
    const wscale_u wu = wscale_prepare(ustar);

    // -- surface-layer reference values, bulk-Ri numerator, shear (:111-137)
    __syncthreads();
    FORJ if (act[j]) { aU[kk[j]] = r.U[j]; aV[kk[j]] = r.V[j]; aB[kk[j]] = r.buoy[j]; }
    __syncthreads();
    double Ritop[LPL], dVsq[LPL];
    {
      const double epsilon = 0.1;
      const double U1 = aU[1], V1 = aV[1], Bu1 = aB[1];
      double zref[LPL], ur[LPL], vr[LPL], br[LPL];
      bool live[LPL];
      FORJ {
        zref[j] = epsilon * r.zm[j];
        double wz = dmax2(zm1, zref[j]);
        ur[j] = U1 * wz / zref[j];
        vr[j] = V1 * wz / zref[j];
        br[j] = Bu1 * wz / zref[j];
        live[j] = actz[j];
      }
      double zk = zm1, Uk = U1, Vk = V1, Bk = Bu1;
      for (int kl = 1; kl <= nz; ++kl) {
        const double zk1 = c_zm[kl + 1], Uk1 = aU[kl + 1], Vk1 = aV[kl + 1], Bk1 = aB[kl + 1];
        bool any = false;
        FORJ {
          live[j] = live[j] && !(zref[j] >= zk);
          any = any || live[j];
        }
        if (!__any(any)) break;
        FORJ if (live[j]) {
          double wz = dmin2(zk - zk1, zk - zref[j]);
          double del = 0.5 * wz / (zk - zk1);
          ur[j] = ur[j] - wz * (Uk + del * (Uk1 - Uk)) / zref[j];
          vr[j] = vr[j] - wz * (Vk + del * (Vk1 - Vk)) / zref[j];
          br[j] = br[j] - wz * (Bk + del * (Bk1 - Bk)) / zref[j];
        }
        zk = zk1; Uk = Uk1; Vk = Vk1; Bk = Bk1;
      }
      FORJ {
        int k = kk[j];
        double bk1 = aB[k + 1], uk1 = aU[k + 1], vk1 = aV[k + 1];
        Ritop[j] = (zref[j] - r.zm[j]) * (br[j] - r.buoy[j]);
        r.dbloc[j] = r.buoy[j] - bk1;
        dVsq[j] = (ur[j] - r.U[j]) * (ur[j] - r.U[j]) + (vr[j] - r.V[j]) * (vr[j] - r.V[j]);
        r.shsq[j] = (r.U[j] - uk1) * (r.U[j] - uk1) + (r.V[j] - vk1) * (r.V[j] - vk1);
      }
      double un = 0, vn = 0;
      FORJ {
        double a = bcast(ur[j], lane_nz), b = bcast(vr[j], lane_nz);
        if (j == j_nz) { un = a; vn = b; }
      }
      ps.uref_nz = un;
      ps.vref_nz = vn;
    }

    // -- rimix (rimix_mod.F90:13-106) with the 1-2-1 smoother (z121_mod.F90:7-45)
    double zdiff[LPL];   // zm(k) - zm(k+1)
    FORJ {
      int k = kk[j];
      zdiff[j] = r.zm[j] - c_zm[k + 1];
      r.Rig[j] = r.dbloc[j] * zdiff[j] / (r.shsq[j] + 1.e-16);
      if (actz[j]) { aR[k] = r.Rig[j]; aDb[k] = r.dbloc[j]; }
      if (k == 1) aR[0] = 0.0;
      if (k == nzp1) aR[k] = 0.0;
    }
    __syncthreads();
    double dm_i[LPL], ds_i[LPL], dt_i[LPL];   // interior diffusivities at interface k
    FORJ {
      int k = kk[j];
      const double Riinfty = 0.8;
      double vm1 = aR[k - 1], vp1 = aR[k + 1];
      double wm1 = (k - 1 >= 1 && !((vm1 < 0.0) || (vm1 > Riinfty))) ? 1.0 : 0.0;
      double wp1 = (k + 1 <= nz && !((vp1 < 0.0) || (vp1 > Riinfty))) ? 1.0 : 0.0;
      double sm = wm1 * vm1 + 2. * r.Rig[j] + wp1 * vp1;
      double wait = wm1 + 2.0 + wp1;
      sm = sm / wait;
      double Rigg = dmax2(sm, 0.0);
      double ratio = dmin2(Rigg / Riinfty, 1.0);
      double fri = (1.0 - ratio * ratio);
      fri = fri * fri * fri;
      dm_i[j] = (0.0001 + fri * 0.005);
      ds_i[j] = (0.00001 + fri * 0.005);
      dt_i[j] = ds_i[j];
    }
    __syncthreads();
    FORJ {
      int k = kk[j];
      if (actz[j]) { aDm[k] = dm_i[j]; aDs[k] = ds_i[j]; aDt[k] = dt_i[j]; }
      if (k == nz) { aDm[k + 1] = dm_i[j]; aDs[k + 1] = ds_i[j]; aDt[k + 1] = dt_i[j]; }  // kppmix_mod.F90:82-84
      if (k == 1) { aDm[0] = 0.0; aDs[0] = 0.0; aDt[0] = 0.0; }
    }

    // -- bldepth (bldepth_mod.F90:32-203): level-parallel part
    const double epsln16 = 1.e-16, Ricr = 0.30, eps01 = 0.1, cekman = 0.7, cmonob = 1.0;
    double bfs_k[LPL], stab_k[LPL];
    FORJ {
      int k = kk[j];
      double bf = B0 + B0sol * (1. - r.swf[j]);
      double st = 0.5 + dsign(0.5, bf + epsln16);
      double sg = st * 1. + (1. - st) * eps01;
      double wm, ws;
      wscale_dev(p, wu, sg, -r.zm[j], bf, wm, ws);
      double dbm1 = aDb[k - 1];   // row entry 0 is never used (k >= 2 only)
      double bvsq = 0.5 * (dbm1 / (c_zm[k - 1] - r.zm[j]) + r.dbloc[j] / zdiff[j]);
      double Vtsq = -r.zm[j] * ws * __builtin_sqrt(__builtin_fabs(bvsq)) * p.Vtc;
      double raw = Ritop[j] / (dVsq[j] + Vtsq + epsln16);
      double dmo = cmonob * ustar * ustar * ustar / p.vonk / (__builtin_fabs(bf) + epsln16);
      dmo = st * dmo - (1. - st) * zm_kmp1;
      bfs_k[j] = bf;
      stab_k[j] = st;
      if (k >= 2 && actz[j]) { aR[k] = raw; aDmo[k] = dmo; }
      if (k == 1) { aR[1] = 0.0; aDmo[1] = -zm_kmp1; }
    }
    __syncthreads();
    if (lane == 0) {   // Rib(ku) = MAX(Rib(ku), Rib(ka)+epsln), bldepth_mod.F90:137
      double rb = 0.0;
      for (int k = 2; k <= nz; ++k) {
        rb = dmax2(aR[k], rb + epsln16);
        aR[k] = rb;
      }
    }
    __syncthreads();
    const double hek = cekman * ustar / (__builtin_fabs(f) + epsln16);
    int kbl = nz;
    double hbl = -c_zm[nz];
    {
      bool found = false;
      FORJ {
        int k = kk[j];
        double Rka = aR[k - 1], Rku = aR[k], dmoa = aDmo[k - 1], dmou = aDmo[k];
        double zkm1 = c_zm[k - 1];
        double hri = -zkm1 + (zkm1 - r.zm[j]) * (Ricr - Rka) / (Rku - Rka);
        double hmonob;
        if (dmou <= (-r.zm[j])) {
          hmonob = (dmou - dmoa) / (zkm1 - r.zm[j]);
          hmonob = (dmou + hmonob * r.zm[j]) / (1. - hmonob);
        } else {
          hmonob = -zm_kmp1;
        }
        double hekman = stab_k[j] * hek - (1. - stab_k[j]) * zm_kmp1;
        double hmin = dmin2(dmin2(dmin2(hri, hmonob), hekman), -ocdepth);
        bool hit = (k >= 2) && actz[j] && (hmin < -r.zm[j]);
        if (hit && !l_initflag && (hmin < -zkm1)) {
          double hmin2 = dmin2(dmin2(hri, hmonob), -ocdepth);
          if (hmin2 < -r.zm[j]) hmin = hmin2;
        }
        unsigned long long m = __ballot(hit);
        if (!found && m != 0ull) {
          int src = __ffsll((long long)m) - 1;
          found = true;
          kbl = src + 64 * j + 1;
          hbl = bcast(hmin, src);
        }
      }
    }
    double bfsfc = swfrac_dev(-1.0, hbl, jer);
    bfsfc = B0 + B0sol * (1. - bfsfc);
    const double stable = 0.5 + dsign(0.5, bfsfc);
    bfsfc = bfsfc + stable * epsln16;
    const double caseA = 0.5 + dsign(0.5, -c_zm[kbl] - 0.5 * c_hm[kbl] - hbl);

    // -- blmix (blmix_mod.F90:13-151)
    const double epsln20 = 1.e-20;
    double gat1[3], dat1[3], dkm1[3];
    {
      double wm, ws;
      double sigma = stable * 1.0 + (1. - stable) * eps01;
      wscale_dev(p, wu, sigma, hbl, bfsfc, wm, ws);
      int ifx = (int)(caseA + epsln20);
      int kn = ifx * (kbl - 1) + (1 - ifx) * kbl;
      double hmkn = c_hm[kn], hmkn1 = c_hm[kn + 1];
      double delhat = 0.5 * hmkn - c_zm[kn] - hbl;
      double R = 1.0 - delhat / hmkn;
      const double *dd[3] = {aDm, aDs, aDt};
      double dp[3], dh[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double dvdzup = (dd[m][kn - 1] - dd[m][kn]) / hmkn;
        double dvdzdn = (dd[m][kn] - dd[m][kn + 1]) / hmkn1;
        dp[m] = 0.5 * ((1. - R) * (dvdzup + __builtin_fabs(dvdzup)) + R * (dvdzdn + __builtin_fabs(dvdzdn)));
        dh[m] = dd[m][kn] + dp[m] * delhat;
      }
      double u4 = ((ustar * ustar) * ustar) * ustar;
      double f1 = stable * 5.0 * bfsfc / (u4 + epsln20);
      gat1[0] = dh[0] / hbl / (wm + epsln20);
      dat1[0] = -dp[0] / (wm + epsln20) + f1 * dh[0];
      dat1[0] = dmin2(dat1[0], 0.);
      gat1[1] = dh[1] / hbl / (ws + epsln20);
      dat1[1] = -dp[1] / (ws + epsln20) + f1 * dh[1];
      dat1[1] = dmin2(dat1[1], 0.);
      gat1[2] = dh[2] / hbl / (ws + epsln20);
      dat1[2] = -dp[2] / (ws + epsln20) + f1 * dh[2];
      dat1[2] = dmin2(dat1[2], 0.);
    }
    double blmc[3][LPL];
    FORJ {
      double wm, ws;
      double sig = (-r.zm[j] + 0.5 * r.hm[j]) / hbl;
      double sigma = stable * sig + (1. - stable) * dmin2(sig, eps01);
      wscale_dev(p, wu, sigma, hbl, bfsfc, wm, ws);
      double a1 = sig - 2.;
      double a2 = 3. - 2. * sig;
      double a3 = sig - 1.;
      double Gm = a1 + a2 * gat1[0] + a3 * dat1[0];
      double Gs = a1 + a2 * gat1[1] + a3 * dat1[1];
      double Gt = a1 + a2 * gat1[2] + a3 * dat1[2];
      blmc[0][j] = hbl * wm * sig * (1. + sig * Gm);
      blmc[1][j] = hbl * ws * sig * (1. + sig * Gs);
      blmc[2][j] = hbl * ws * sig * (1. + sig * Gt);
      r.ghat[j] = (1. - stable) * p.cg / (ws * hbl + epsln20);
    }
    {
      double wm, ws;
      double sig = -c_zm[kbl - 1] / hbl;
      double sigma = stable * sig + (1. - stable) * dmin2(sig, eps01);
      wscale_dev(p, wu, sigma, hbl, bfsfc, wm, ws);
      double a1 = sig - 2.;
      double a2 = 3. - 2. * sig;
      double a3 = sig - 1.;
      double Gm = a1 + a2 * gat1[0] + a3 * dat1[0];
      double Gs = a1 + a2 * gat1[1] + a3 * dat1[1];
      double Gt = a1 + a2 * gat1[2] + a3 * dat1[2];
      dkm1[0] = hbl * wm * sig * (1. + sig * Gm);
      dkm1[1] = hbl * ws * sig * (1. + sig * Gs);
      dkm1[2] = hbl * ws * sig * (1. + sig * Gt);
    }
    // -- enhance (enhance_mod.F90:10-51), combine (kppmix_mod.F90:103-111),
    //    bottom limits (verticalmixing_mod.F90:151-159)
    FORJ {
      int k = kk[j];
      if (k == kbl - 1 && k <= nz - 1) {
        double delta = (hbl + r.zm[j]) / zdiff[j];
        double omd = 1. - delta;
        double dkmp5 = caseA * dm_i[j] + (1. - caseA) * blmc[0][j];
        double dstar = (omd * omd) * dkm1[0] + (delta * delta) * dkmp5;
        blmc[0][j] = omd * dm_i[j] + delta * dstar;
        dkmp5 = caseA * ds_i[j] + (1. - caseA) * blmc[1][j];
        dstar = (omd * omd) * dkm1[1] + (delta * delta) * dkmp5;
        blmc[1][j] = omd * ds_i[j] + delta * dstar;
        dkmp5 = caseA * dt_i[j] + (1. - caseA) * blmc[2][j];
        dstar = (omd * omd) * dkm1[2] + (delta * delta) * dkmp5;
        blmc[2][j] = omd * dt_i[j] + delta * dstar;
        r.ghat[j] = (1. - caseA) * r.ghat[j];
      }
      if (k < kbl) {
        r.difm[j] = blmc[0][j]; r.difs[j] = blmc[1][j]; r.dift[j] = blmc[2][j];
      } else {
        r.difm[j] = dm_i[j]; r.difs[j] = ds_i[j]; r.dift[j] = dt_i[j];
        r.ghat[j] = 0.;
      }
      if (k >= nz) {
        r.difm[j] = 0.0001; r.difs[j] = 0.00001; r.dift[j] = 0.00001;
        r.ghat[j] = 0.0;
      }
    }
    hbl_out = hbl;
    kbl_out = kbl;
    if (!do_ocnint) return;

    // -- ocnint: coefficients and right-hand sides, level-parallel
    __syncthreads();
    FORJ if (act[j]) {
      int k = kk[j];
      aDm[k] = r.difm[j]; aDs[k] = r.difs[j]; aDt[k] = r.dift[j]; aGh[k] = r.ghat[j];
    }
    __syncthreads();
    const double Uo_np = [&] { double v = 0; FORJ { double a = bcast(r.Uo[j], lane_np); if (j == j_np) v = a; } return v; }();
    const double Vo_np = [&] { double v = 0; FORJ { double a = bcast(r.Vo[j], lane_np); if (j == j_np) v = a; } return v; }();
    const double To_np = [&] { double v = 0; FORJ { double a = bcast(r.To[j], lane_np); if (j == j_np) v = a; } return v; }();
    const double So_np = [&] { double v = 0; FORJ { double a = bcast(r.So[j], lane_np); if (j == j_np) v = a; } return v; }();
    const double dto = p.dto;
    FORJ {
      int k = kk[j];
      if (!actz[j]) continue;
      const double dm_m1 = aDm[k - 1], dt_m1 = aDt[k - 1], ds_m1 = aDs[k - 1];
      const double gh_m1 = (k >= 2) ? aGh[k - 1] : 0.0;
      // tridcof, solvers.F90:14-44, for difm / dift / difs
      const double dk[3] = {r.difm[j], r.dift[j], r.difs[j]};
      const double dk1[3] = {dm_m1, dt_m1, ds_m1};
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double cu, cc, cl;
        if (k == 1) {
          cu = 0.;
          cc = 1. + r.tri1[j] * dk[m];
          cl = -r.tri1[j] * dk[m];
        } else {
          cu = -r.tri0[j] * dk1[m];
          cc = 1. + r.tri1[j] * dk[m] + r.tri0[j] * dk1[m];
          cl = -r.tri1[j] * dk[m];
        }
        if (k == nz) cl = 0.;
        double *th = row(A_TH0 + 5 * m);
        th[0 * NA + k] = cu;
        th[1 * NA + k] = cc;
        th[2 * NA + k] = cl;
      }
      // U right-hand side, ocnint_mod.F90:51-58
      double rhsU;
      if (k == 1) rhsU = r.Uo[j] + dto * (f * .5 * (r.Vo[j] + r.V[j]) - ps.wU0_1 / hm1);
      else rhsU = r.Uo[j] + dto * f * .5 * (r.Vo[j] + r.V[j]);
      if (k == nz) rhsU = rhsU + r.tri1[j] * r.difm[j] * Uo_np;
      row(A_TH0 + 0)[3 * NA + k] = rhsU;
      // T right-hand side: tridrhs (solvers.F90:53-107, npd=1) with ghat and solar terms
      double rhsT;
      const double hk = r.hm[j];
      if (k == 1)
        rhsT = r.To[j] + dto / hk * (ps.wX0_1 * r.dift[j] * r.ghat[j] - ps.wX0_1 * 1.0 + r.wxnt[j] - ps.wXNT0);
      else
        rhsT = r.To[j] + dto / hk * (ps.wX0_1 * (r.dift[j] * r.ghat[j] - dt_m1 * gh_m1) + r.wxnt[j] - wxnt_m1[j]);
      if (k == nz && nz > 1) rhsT = rhsT + To_np * r.tri1[j] * r.dift[j];
      row(A_TH0 + 5)[3 * NA + k] = rhsT;
      // S right-hand side (wXNT(:,2) is identically zero, fluxes_mod.F90:26)
      double rhsS;
      if (k == 1)
        rhsS = r.So[j] + dto / hk * (ps.wX0_2 * r.difs[j] * r.ghat[j] - ps.wX0_2 * 1.0 + 0.0 - 0.0);
      else
        rhsS = r.So[j] + dto / hk * (ps.wX0_2 * (r.difs[j] * r.ghat[j] - ds_m1 * gh_m1) + 0.0 - 0.0);
      if (k == nz && nz > 1) rhsS = rhsS + So_np * r.tri1[j] * r.difs[j];
      row(A_TH0 + 10)[3 * NA + k] = rhsS;
    }
    __syncthreads();
    // -- Thomas sweeps (solvers.F90:112-161): lanes 0,1,2 solve U, T, S
    int bad = 0;
    if (lane < 3) {
      double *th = row(A_TH0 + 5 * lane);
      double *cu = th, *cc = th + NA, *cl = th + 2 * NA, *rh = th + 3 * NA, *gm = th + 4 * NA;
      double *betm = row(A_BETM);
      double bet = cc[1];
      double y = rh[1] / bet;
      rh[1] = y;
      if (lane == 0) betm[1] = bet;
      for (int i = 2; i <= nz; ++i) {
        double g = cl[i - 1] / bet;
        bet = cc[i] - cu[i] * g;
        if (bet == 0.) { bad = 1; bet = 1.E-12; }
        y = (rh[i] - cu[i] * y) / bet;
        gm[i] = g;
        rh[i] = y;
        if (lane == 0) betm[i] = bet;
      }
      for (int i = nz - 1; i >= 1; --i) {
        y = rh[i] - gm[i + 1] * y;
        rh[i] = y;
      }
    }
    __syncthreads();
    {
      const double *yU = row(A_TH0 + 0) + 3 * NA, *yT = row(A_TH0 + 5) + 3 * NA, *yS = row(A_TH0 + 10) + 3 * NA;
      double *rhv = row(A_RHV);
      FORJ {
        int k = kk[j];
        if (actz[j]) {
          r.U[j] = yU[k]; r.T[j] = yT[k]; r.S[j] = yS[k];
          // V right-hand side with the just-updated U, ocnint_mod.F90:62-69
          double rhsV;
          if (k == 1) rhsV = r.Vo[j] - dto * (f * .5 * (r.Uo[j] + r.U[j]) + ps.wU0_2 / hm1);
          else rhsV = r.Vo[j] - dto * f * .5 * (r.Uo[j] + r.U[j]);
          if (k == nz) rhsV = rhsV + r.tri1[j] * r.difm[j] * Vo_np;
          rhv[k] = rhsV;
        } else if (act[j]) {   // yn(nzi+1) = yo(nzi+1), solvers.F90:159
          r.U[j] = r.Uo[j]; r.T[j] = r.To[j]; r.S[j] = r.So[j];
        }
      }
    }
    __syncthreads();
    if (lane == 0) {
      const double *cu = row(A_TH0), *gm = row(A_TH0) + 4 * NA, *betm = row(A_BETM);
      double *rh = row(A_RHV);
      double y = rh[1] / betm[1];
      rh[1] = y;
      for (int i = 2; i <= nz; ++i) {
        y = (rh[i] - cu[i] * y) / betm[i];
        rh[i] = y;
      }
      for (int i = nz - 1; i >= 1; --i) {
        y = rh[i] - gm[i + 1] * y;
        rh[i] = y;
      }
    }
    __syncthreads();
    {
      const double *yV = row(A_RHV);
      FORJ {
        if (actz[j]) r.V[j] = yV[kk[j]];
        else if (act[j]) r.V[j] = r.Vo[j];
      }
    }
    if (__any(bad)) status |= 1;
  };

  double hmix_out = 0;
  int kmix_out = 0, npass = 0;
  int comp_flag = 0;
  double reset_flag = 0.0;

  if (p.mode == MCKPP_MODE_STEP) {
    const double lambda = 0.5;
    double UsO[LPL], VsO[LPL], TsO[LPL], SsO[LPL], UsN[LPL], VsN[LPL], TsN[LPL], SsN[LPL];
    FORJ {
      int jj = lane + 64 * j;
      UsO[j] = act[j] ? p.Us[old][rowoff + jj] : 0.0; UsN[j] = act[j] ? p.Us[newi][rowoff + jj] : 0.0;
      VsO[j] = act[j] ? p.Vs[old][rowoff + jj] : 0.0; VsN[j] = act[j] ? p.Vs[newi][rowoff + jj] : 0.0;
      TsO[j] = act[j] ? p.Ts[old][rowoff + jj] : 0.0; TsN[j] = act[j] ? p.Ts[newi][rowoff + jj] : 0.0;
      SsO[j] = act[j] ? p.Ss[old][rowoff + jj] : 0.0; SsN[j] = act[j] ? p.Ss[newi][rowoff + jj] : 0.0;
    }
    comp_flag = 1;
    double hmixe = 0, hmixn = 0;
    int kmixn = 0;
    while (comp_flag && reset_flag <= 10.0) {                 // ocnstep_mod.F90:89
      FORJ {                                                  // :91-112
        r.U[j] = 2. * UsN[j] - UsO[j]; r.Ux[j] = r.U[j];
        r.V[j] = 2. * VsN[j] - VsO[j]; r.Vx[j] = r.V[j];
        r.T[j] = 2. * TsN[j] - TsO[j]; r.Tx[j] = r.T[j];
        r.S[j] = 2. * SsN[j] - SsO[j]; r.Sx[j] = r.S[j];
      }
      int npass_try = 0, iconv = 0;
      for (;;) {
        FORJ {                                                // :123-132 / :142-151
          r.U[j] = lambda * r.Ux[j] + (1 - lambda) * r.U[j]; r.Ux[j] = r.U[j];
          r.V[j] = lambda * r.Vx[j] + (1 - lambda) * r.V[j]; r.Vx[j] = r.V[j];
          r.T[j] = lambda * r.Tx[j] + (1 - lambda) * r.T[j]; r.Tx[j] = r.T[j];
          r.S[j] = lambda * r.Sx[j] + (1 - lambda) * r.S[j]; r.Sx[j] = r.S[j];
        }
        double h;
        int kb;
        pass(true, h, kb);
        ++npass;
        ++npass_try;
        if (npass_try <= 3) {                                 // compulsory passes, :122-135
          hmixe = h;
          continue;
        }
        hmixn = h; kmixn = kb;                                // :152-154 (iter == npass_try)
        double tol = p.hmixtolfrac * c_hm[kmixn];             // :157
        if (kmixn == nzp1) tol = p.hmixtolfrac * c_hm[nz];
        if (__builtin_fabs(hmixn - hmixe) > tol) iconv = 0;   // :159-169
        else iconv = iconv + 1;
        if (iconv < 3) {                                      // :170-183
          if (npass_try < p.itermax) { hmixe = hmixn; continue; }
          else if (hmixn > hmixe) { hmixe = hmixn; continue; }
        }
        if (npass_try > (p.itermax + 1)) status |= 2;         // :184-191
        break;
      }
      // instability trap, :200-227
      comp_flag = 0;
      __syncthreads();
      FORJ if (act[j]) aT[kk[j]] = r.T[j];
      __syncthreads();
      int nviol = 0;
      FORJ {
        int k = kk[j];
        double tk1 = aT[k + 1];
        bool v = actz[j] && (__builtin_fabs(r.U[j]) >= 10 || __builtin_fabs(r.V[j]) >= 10 ||
                             __builtin_fabs(r.T[j] - tk1) >= 10);
        nviol += __popcll(__ballot(v));
      }
      if (nviol > 0) {
        comp_flag = 1;
        for (int i = 0; i < nviol; ++i) f = f * 1.01;
      }
      if (!comp_flag) {
        double *t0 = row(A_TH0), *t1 = row(A_TH0 + 1), *t2 = row(A_TH0 + 2), *t3 = row(A_TH0 + 3);
        __syncthreads();
        FORJ if (act[j]) {
          int k = kk[j];
          t0[k] = (r.U[j] - r.Uo[j]) * (r.U[j] - r.Uo[j]) * r.hm[j] / p.dm_nz;
          t1[k] = (r.V[j] - r.Vo[j]) * (r.V[j] - r.Vo[j]) * r.hm[j] / p.dm_nz;
          t2[k] = (r.T[j] - r.To[j]) * (r.T[j] - r.To[j]) * r.hm[j] / p.dm_nz;
          t3[k] = (r.S[j] - r.So[j]) * (r.S[j] - r.So[j]) * r.hm[j] / p.dm_nz;
        }
        __syncthreads();
        bool over = false;
        if (lane < 4) {
          const double *t = row(A_TH0 + lane);
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
      if (comp_flag) status |= 4;
      reset_flag = reset_flag + 1;
      if (reset_flag > 10.0) status |= 8;
    }
    hmix_out = hmixn;
    kmix_out = kmixn;
  } else if (p.mode == MCKPP_MODE_INIT) {
    const int save = l_initflag;
    l_initflag = 1;                                           // initialize_ocean.F90:59
    pass(false, hmix_out, kmix_out);
    l_initflag = 0;
    (void)save;
    npass = 1;
  } else {
    pass(true, hmix_out, kmix_out);
    npass = 1;
  }

  // ---- diagnostic fluxes (ocnstep_mod.F90:242-256 / initialize_ocean.F90:66-81)
  double wX1[LPL], wX2[LPL], wX3[LPL], wU1[LPL], wU2[LPL];
  if (p.diag && p.mode != MCKPP_MODE_PASS) {
    __syncthreads();
    FORJ if (act[j]) { int k = kk[j]; aU[k] = r.U[j]; aV[k] = r.V[j]; aB[k] = r.T[j]; aR[k] = r.S[j]; }
    __syncthreads();
    FORJ {
      int k = kk[j];
      double deltaz = 0.5 * (r.hm[j] + c_hm[k + 1]);
      double uk1 = aU[k + 1], vk1 = aV[k + 1], tk1 = aB[k + 1], sk1 = aR[k + 1];
      wX1[j] = -r.difs[j] * ((r.T[j] - tk1) / deltaz - r.ghat[j] * ps.wX0_1);
      wX2[j] = -r.difs[j] * ((r.S[j] - sk1) / deltaz - r.ghat[j] * ps.wX0_2);
      if (p.LDD) wX1[j] = -r.dift[j] * ((r.T[j] - tk1) / deltaz - r.ghat[j] * ps.wX0_1);
      wX3[j] = p.grav * (r.talpha[j] * wX1[j] - r.sbeta[j] * wX2[j]);
      wU1[j] = -r.difm[j] * (r.U[j] - uk1) / deltaz;
      wU2[j] = -r.difm[j] * (r.V[j] - vk1) / deltaz;
    }
  }

  // ---- end of step: outputs (ocnstep_mod.F90:305-353), check_profile
  //      (overrides.F90:42-125, default switches), stores
  if (p.mode == MCKPP_MODE_STEP) {
    const double uref = first_lane(r.U[0]), vref = first_lane(r.V[0]), Tref = first_lane(r.T[0]);
    if (p.L_SSref) Ssurf = SSref;
    else Ssurf = first_lane(r.S[0]) + Sref;
    old = newi;
    newi = 1 - old;
    FORJ if (act[j]) {
      size_t o = rowoff + lane + 64 * j;
      p.Us[newi][o] = r.U[j]; p.Vs[newi][o] = r.V[j]; p.Ts[newi][o] = r.T[j]; p.Ss[newi][o] = r.S[j];
    }
    if (comp_flag) {   // overrides.F90:57-78 (no climatology: reset currents only)
      FORJ if (act[j]) {
        size_t o = rowoff + lane + 64 * j;
        r.U[j] = p.U_init[o];
        r.V[j] = p.V_init[o];
      }
    }
    FORJ if (act[j]) {
      size_t o = rowoff + lane + 64 * j;
      p.U[o] = r.U[j]; p.V[o] = r.V[j]; p.T[o] = r.T[j]; p.S[o] = r.S[j];
    }
    if (lane == 0) {
      cs[CS_HMIX] = hmix_out;
      cs[CS_KMIX] = (double)kmix_out;
      cs[CS_UREF] = uref; cs[CS_VREF] = vref; cs[CS_TREF] = Tref;
      cs[CS_SSURF] = Ssurf;
      cs[newi ? CS_HMIXD1 : CS_HMIXD0] = hmix_out;
      cs[CS_RESET] = 0.0;   // overrides.F90:121-123 with L_NO_ISOTHERM = .F.
      cs[CS_DAMPU] = 0.0; cs[CS_DAMPV] = 0.0;
      ci[CI_OLD] = old; ci[CI_NEW] = newi;
      ci[CI_STATUS] = status; ci[CI_NPASS] = npass;
    }
  } else if (p.mode == MCKPP_MODE_INIT) {
    const double Tref = first_lane(r.T[0]);
    FORJ if (act[j]) {
      size_t o = rowoff + lane + 64 * j;
      p.Us[0][o] = r.U[j]; p.Us[1][o] = r.U[j]; p.Vs[0][o] = r.V[j]; p.Vs[1][o] = r.V[j];
      p.Ts[0][o] = r.T[j]; p.Ts[1][o] = r.T[j]; p.Ss[0][o] = r.S[j]; p.Ss[1][o] = r.S[j];
    }
    if (lane == 0) {
      cs[CS_HMIX] = hmix_out;
      cs[CS_KMIX] = (double)kmix_out;
      cs[CS_TREF] = Tref;
      cs[CS_UREF] = ps.uref_nz; cs[CS_VREF] = ps.vref_nz;
      cs[CS_HMIXD0] = hmix_out; cs[CS_HMIXD1] = hmix_out;
      ci[CI_OLD] = 0; ci[CI_NEW] = 1; ci[CI_INITFLAG] = 0;
      ci[CI_STATUS] = status; ci[CI_NPASS] = npass;
    }
  } else {
    FORJ if (act[j]) {
      size_t o = rowoff + lane + 64 * j;
      p.U[o] = r.U[j]; p.V[o] = r.V[j]; p.T[o] = r.T[j]; p.S[o] = r.S[j];
    }
    if (lane == 0) {
      cs[CS_HMIX] = hmix_out;
      cs[CS_KMIX] = (double)kmix_out;
      cs[CS_UREF] = ps.uref_nz; cs[CS_VREF] = ps.vref_nz;
      ci[CI_STATUS] = status; ci[CI_NPASS] = npass;
    }
  }

  if (p.diag) {
    FORJ {
      int k = kk[j];
      size_t o = rowoff + k;
      if (act[j]) {
        p.rho[o] = r.rho[j]; p.cp[o] = r.cp[j]; p.buoy[o] = r.buoy[j];
        p.talpha[o] = r.talpha[j]; p.sbeta[o] = r.sbeta[j];
        p.difm[o] = r.difm[j]; p.difs[o] = r.difs[j]; p.dift[o] = r.dift[j];
      }
      if (actz[j]) {
        p.ghat[o] = r.ghat[j]; p.Rig[o] = r.Rig[j]; p.dbloc[o] = r.dbloc[j]; p.Shsq[o] = r.shsq[j];
        p.wXNT1[o] = r.wxnt[j];
        if (p.mode != MCKPP_MODE_PASS) {
          p.wX1[o] = wX1[j]; p.wX2[o] = wX2[j]; p.wX3[o] = wX3[j]; p.wU1[o] = wU1[j]; p.wU2[o] = wU2[j];
        }
      }
      if (k == 1) {   // index-0 entries
        p.rho[rowoff] = r.rho[j]; p.cp[rowoff] = r.cp[j];
        p.talpha[rowoff] = r.talpha[j]; p.sbeta[rowoff] = r.sbeta[j];
        p.difm[rowoff] = 0.0; p.difs[rowoff] = 0.0; p.dift[rowoff] = 0.0;
        p.wU1[rowoff] = ps.wU0_1; p.wU2[rowoff] = ps.wU0_2;
        p.wX1[rowoff] = ps.wX0_1; p.wX2[rowoff] = ps.wX0_2; p.wX3[rowoff] = ps.wX0_3;
        p.wXNT1[rowoff] = ps.wXNT0;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// small batch kernels used by the parity tests
// ---------------------------------------------------------------------------
__global__ void k_eos_batch(int64_t n, const double *s, const double *t, const double *p,
                            double *alpha, double *beta, double *sig0, double *cp)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a, b, s0;
  abk80_dev(s[i], t[i], p[i], a, b, s0);
  alpha[i] = a; beta[i] = b; sig0[i] = s0;
  cp[i] = cpsw_dev(s[i], t[i], p[i]);
}

// the exact-division helpers of mckpp_colmath.h, one quotient per thread:
// q[0] = div_fast, q[1] = div_fast_guarded, q[2] = div_by_refined, q[3] = the compiler's n / d
__global__ void k_div_batch(int64_t n, const double *num, const double *den, double *q)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = num[i], d = den[i], r = rcp_refine(d);
  q[i] = div_fast(a, d, r);
  q[n + i] = div_fast_guarded(a, d, r);
  q[2 * n + i] = div_by_refined(a, d, r);
  q[3 * n + i] = a / d;
}

__global__ void k_exp_batch(int64_t n, const double *x, double *y)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = mckpp_exp(x[i]);
}

// ---------------------------------------------------------------------------
// layout kernels: Fortran column-fastest A(npts, nlev) <-> device rows.
// 64x64 tile through LDS so both sides move whole 512-B segments.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_rows(const double *__restrict__ src3d, int64_t npts,
                                                     int nlev, int lev_off, const int *__restrict__ ipt,
                                                     int64_t ncol, double *__restrict__ dst, int ld,
                                                     int dst_off)
{
  __shared__ double tile[64][65];
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  const int l0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int l = ty; l < 64; l += 4) {          // read: columns fastest
    int64_t c = c0 + tx;
    int lev = l0 + l;
    double v = 0.0;
    if (c < ncol && lev < nlev) v = src3d[(int64_t)(lev + lev_off) * npts + ipt[c]];
    tile[l][tx] = v;
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {       // write: levels fastest
    int64_t c = c0 + cc;
    int lev = l0 + tx;
    if (c < ncol && lev < nlev) dst[c * ld + dst_off + lev] = tile[tx][cc];
  }
}

__global__ __launch_bounds__(256) void k_scatter_rows(const double *__restrict__ src, int ld, int src_off,
                                                      const int *__restrict__ ipt, int64_t ncol,
                                                      double *__restrict__ dst3d, int64_t npts, int nlev,
                                                      int lev_off)
{
  __shared__ double tile[64][65];
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  const int l0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int cc = ty; cc < 64; cc += 4) {
    int64_t c = c0 + cc;
    int lev = l0 + tx;
    double v = 0.0;
    if (c < ncol && lev < nlev) v = src[c * ld + src_off + lev];
    tile[tx][cc] = v;
  }
  __syncthreads();
  for (int l = ty; l < 64; l += 4) {
    int64_t c = c0 + tx;
    int lev = l0 + l;
    if (c < ncol && lev < nlev) dst3d[(int64_t)(lev + lev_off) * npts + ipt[c]] = tile[l][tx];
  }
}


// ---------------------------------------------------------------------------
// Surface-flux assembly + non-turbulent flux: mckpp_fluxes,
// src/mckpp_fluxes_mod.F90:35-89, and its ntflux call (:93-118).  One thread
// per column; inputs are the eight forcing fields compacted to resident columns.
// ---------------------------------------------------------------------------
__global__ void k_fluxes(mckpp_kparams p, int ntime, const double *__restrict__ f8, int l_rest, double flsn,
                         double el)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.ncol) return;
  double *cs = p.cs + (size_t)c * MCKPP_CS;
  const int *ci = p.ci + (size_t)c * MCKPP_CI;
  if (!ci[CI_LOCEAN]) return;                                   // fluxes_mod.F90:55
  const size_t n = (size_t)p.ncol;
  double taux = f8[c], tauy = f8[n + c], swf = f8[2 * n + c], lwf = f8[3 * n + c], lhf = f8[4 * n + c],
         shf = f8[5 * n + c], rain = f8[6 * n + c], snow = f8[7 * n + c];
  if ((taux == 0.0) && (tauy == 0.0)) taux = 1.e-10;            // :57-58
  double s1, s2, s3, s4, s5, s6;
  if (!l_rest) {                                                // :60-69
    s1 = taux; s2 = tauy; s3 = swf;
    s4 = lwf + lhf + shf - snow * flsn;
    s5 = 1e-10;
    s6 = rain + snow + (lhf / el);
  } else {                                                      // :70-77
    s1 = 1.e-10; s2 = 0.00; s3 = 300.00; s4 = -300.00; s5 = 0.00; s6 = 0.00;
  }
  cs[CS_SFLUX1] = s1; cs[CS_SFLUX2] = s2; cs[CS_SFLUX3] = s3;
  cs[CS_SFLUX4] = s4; cs[CS_SFLUX5] = s5; cs[CS_SFLUX6] = s6;
  if (p.diag && ntime >= 1) {                                   // ntflux, :110-116
    const size_t ro = (size_t)c * p.ld;
    const double rho0 = p.rho[ro], cp0 = p.cp[ro];
    const int jer = ci[CI_JERLOV];
    for (int k = 0; k <= p.nz; ++k) p.wXNT1[ro + k] = -s3 * p.swdk_tab[jer * p.ldc + k] / (rho0 * cp0);
  }
}

// ---------------------------------------------------------------------------
// mckpp_physics_overrides_bottomtemp, src/mckpp_physics_overrides.F90:12-24:
// prescribed temperature of the bottom grid point; one thread per column.
// ---------------------------------------------------------------------------
__global__ void k_bottomtemp(mckpp_kparams p, const double *__restrict__ bt)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.ncol) return;
  const size_t ro = (size_t)c * p.ld;
  const double b = bt[c];
  const double tinc = b - p.T[ro + p.nzp1 - 1];                                   // :16
  p.tinc_fcorr[ro + p.nzp1] = tinc;
  p.ocnTcorr[ro + p.nzp1] = tinc * p.rho[ro + p.nzp1] * p.cp[ro + p.nzp1] / p.dto;   // :17-19
  p.T[ro + p.nzp1 - 1] = b;                                                       // :20
}

// ---------------------------------------------------------------------------
// Output-window reductions: running sum / min / max of the
// profile rows and of hmix, replacing XIOS's temporal operations
// (run/iodef.xml:91-116) so only reduced fields leave the device.  Pure
// streaming: 16 B per lane per access, one pass over the rows per step.
// acc layout: [field][3][ncol*ld] with 3 = {sum, min, max}; hacc: [3][ncol].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_window_accumulate(const double2 *__restrict__ u, const double2 *__restrict__ v,
                                                         const double2 *__restrict__ t, const double2 *__restrict__ s,
                                                         double2 *__restrict__ acc, size_t n2, const double *__restrict__ cs,
                                                         double *__restrict__ hacc, int ncol, int first)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const double2 *src[4] = {u, v, t, s};
  if (i < n2) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const double2 x = src[f][i];
      double2 *a = acc + (size_t)f * 3 * n2;
      double2 sm = a[i], mn = a[n2 + i], mx = a[2 * n2 + i];
      if (first) { sm = make_double2(0.0, 0.0); mn = x; mx = x; }
      sm.x = sm.x + x.x; sm.y = sm.y + x.y;
      mn.x = x.x < mn.x ? x.x : mn.x; mn.y = x.y < mn.y ? x.y : mn.y;
      mx.x = x.x > mx.x ? x.x : mx.x; mx.y = x.y > mx.y ? x.y : mx.y;
      a[i] = sm; a[n2 + i] = mn; a[2 * n2 + i] = mx;
    }
  }
  if (i < (size_t)ncol) {
    const double h = cs[i * MCKPP_CS + CS_HMIX];
    double sm = hacc[i], mn = hacc[ncol + i], mx = hacc[2 * (size_t)ncol + i];
    if (first) { sm = 0.0; mn = h; mx = h; }
    hacc[i] = sm + h;
    hacc[ncol + i] = h < mn ? h : mn;
    hacc[2 * (size_t)ncol + i] = h > mx ? h : mx;
  }
}

__global__ void k_window_mean(const double *__restrict__ sum, double *__restrict__ out, size_t n, double count)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = sum[i] / count;
}

}  // namespace

size_t mckpp_column_kernel_lds_bytes(int nzp1)
{
  int lpl = (nzp1 + 2 + 63) / 64;
  return (size_t)A_COUNT * (64 * lpl + 8) * sizeof(double);
}

mckpp_launch_info g_mckpp_last_launch = {0, 0, 0, 0};

hipError_t mckpp_launch_column_kernel(const mckpp_kparams &p, hipStream_t stream)
{
  if (p.ncol <= 0) return hipSuccess;
  const int lpl = (p.nzp1 + 2 + 63) / 64;
  const size_t lds = mckpp_column_kernel_lds_bytes(p.nzp1);
  dim3 grid((unsigned)p.ncol), block(64);
  switch (lpl) {
    case 1: hipLaunchKernelGGL(k_column<1>, grid, block, lds, stream, p); break;
    case 2: hipLaunchKernelGGL(k_column<2>, grid, block, lds, stream, p); break;
    case 3: hipLaunchKernelGGL(k_column<3>, grid, block, lds, stream, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t mckpp_launch_eos_batch(int64_t n, const double *s, const double *t, const double *p,
                                  double *alpha, double *beta, double *sig0, double *cp,
                                  hipStream_t stream)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_eos_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, s, t, p,
                     alpha, beta, sig0, cp);
  return hipGetLastError();
}

hipError_t mckpp_launch_div_batch(int64_t n, const double *num, const double *den, double *q, hipStream_t stream)
{
  hipLaunchKernelGGL(k_div_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, num, den, q);
  return hipGetLastError();
}

hipError_t mckpp_launch_exp_batch(int64_t n, const double *x, double *y, hipStream_t stream)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_exp_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, x, y);
  return hipGetLastError();
}

hipError_t mckpp_launch_gather_rows(const double *src3d, int64_t npts, int nlev, int lev_off,
                                    const int *ipt, int64_t ncol, double *dst, int ld, int dst_off,
                                    hipStream_t stream)
{
  if (ncol <= 0 || nlev <= 0) return hipSuccess;
  dim3 grid((unsigned)((ncol + 63) / 64), (unsigned)((nlev + 63) / 64));
  hipLaunchKernelGGL(k_gather_rows, grid, dim3(256), 0, stream, src3d, npts, nlev, lev_off, ipt, ncol,
                     dst, ld, dst_off);
  return hipGetLastError();
}

hipError_t mckpp_launch_scatter_rows(const double *src, int ld, int src_off, const int *ipt,
                                     int64_t ncol, double *dst3d, int64_t npts, int nlev, int lev_off,
                                     hipStream_t stream)
{
  if (ncol <= 0 || nlev <= 0) return hipSuccess;
  dim3 grid((unsigned)((ncol + 63) / 64), (unsigned)((nlev + 63) / 64));
  hipLaunchKernelGGL(k_scatter_rows, grid, dim3(256), 0, stream, src, ld, src_off, ipt, ncol, dst3d,
                     npts, nlev, lev_off);
  return hipGetLastError();
}

hipError_t mckpp_launch_fluxes(const mckpp_kparams &p, int ntime, const double *f8, int l_rest, double flsn,
                               double el, hipStream_t stream)
{
  if (p.ncol <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_fluxes, dim3((unsigned)((p.ncol + 255) / 256)), dim3(256), 0, stream, p, ntime, f8, l_rest,
                     flsn, el);
  return hipGetLastError();
}

hipError_t mckpp_launch_bottomtemp(const mckpp_kparams &p, const double *bt, hipStream_t stream)
{
  if (p.ncol <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_bottomtemp, dim3((unsigned)((p.ncol + 255) / 256)), dim3(256), 0, stream, p, bt);
  return hipGetLastError();
}

hipError_t mckpp_launch_window_accumulate(const double *u, const double *v, const double *t, const double *s,
                                          double *acc, size_t nelem, const double *cs, double *hacc, int ncol,
                                          int first, hipStream_t stream)
{
  const size_t n2 = nelem / 2;   // rows are 64*LPL doubles: always even
  const size_t nthreads = n2 > (size_t)ncol ? n2 : (size_t)ncol;
  hipLaunchKernelGGL(k_window_accumulate, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<const double2 *>(u), reinterpret_cast<const double2 *>(v),
                     reinterpret_cast<const double2 *>(t), reinterpret_cast<const double2 *>(s),
                     reinterpret_cast<double2 *>(acc), n2, cs, hacc, ncol, first);
  return hipGetLastError();
}

hipError_t mckpp_launch_window_mean(const double *sum, double *out, size_t n, double count, hipStream_t stream)
{
  hipLaunchKernelGGL(k_window_mean, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sum, out, n, count);
  return hipGetLastError();
}
