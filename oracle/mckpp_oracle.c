/*
 * mckpp_oracle.c - CPU restatement of the MC-KPP per-column physics step.
 *
 * TEST INFRASTRUCTURE ONLY (see mckpp_oracle.h).  Compile with
 *   gcc -O2 -ffp-contract=off -fno-fast-math [-fopenmp]
 * so every +,-,*,/ and sqrt is a single correctly rounded IEEE-754 binary64
 * operation in source order.  Each function cites the reference file:line
 * (relative to /root/reference/src/) whose algorithm it restates.
 */
#include "mckpp_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Fortran MAX/MIN/SIGN on non-NaN operands */
static inline double fmax2(double a, double b) { return a > b ? a : b; }
static inline double fmin2(double a, double b) { return a < b ? a : b; }
static inline double fsign(double a, double b) { return copysign(fabs(a), b); }

/* ------------------------------------------------------------------------
 * Portable exp: Cody-Waite reduction + the classic rational kernel, written
 * with +,-,*,/ and integer ops only so the HIP kernels (which carry an
 * identical copy) reproduce it bit for bit.  Error < 1 ulp.  Used when
 * orc_const.exp_mode == 1; exp_mode == 0 calls libm's exp like the
 * reference's Fortran EXP intrinsic does.
 * ---------------------------------------------------------------------- */
double orc_exp_portable(double x)
{
  const double ln2hi = 6.93147180369123816490e-01;
  const double ln2lo = 1.90821492927058770002e-10;
  const double invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01;
  const double P2 = -2.77777777770155933842e-03;
  const double P3 = 6.61375632143793436117e-05;
  const double P4 = -1.65339022054652515390e-06;
  const double P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 709.0) return HUGE_VAL;
  if (x < -745.0) return 0.0;
  double t = invln2 * x;
  int k = (int)(t + (x < 0.0 ? -0.5 : 0.5));
  double fk = (double)k;
  double hi = x - fk * ln2hi;
  double lo = fk * ln2lo;
  double r = hi - lo;
  double tt = r * r;
  double c = r - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  /* scale by 2^k in two steps so subnormal results round once */
  union { double d; uint64_t u; } s;
  if (k >= -1021) {
    s.u = (uint64_t)(1023 + k) << 52;
    return y * s.d;
  }
  s.u = (uint64_t)(1023 + k + 1000) << 52;
  y = y * s.d;
  s.u = (uint64_t)(1023 - 1000) << 52;
  return y * s.d;
}

static inline double orc_exp(const orc_const *c, double x)
{
  return c->exp_mode ? orc_exp_portable(x) : exp(x);
}

/* ------------------------------------------------------------------------
 * Specific heat of sea water.  mckpp_physics_state_equations.F90:7-58
 * ---------------------------------------------------------------------- */
double orc_cpsw(double S, double T1, double P0)
{
  double T = T1;
  if (T < -2.) T = -2.;                                   /* :28-29 */
  double P = P0 / 10.;                                    /* :32 */
  double SR = sqrt(fabs(S));                              /* :35 */
  double A = (-1.38385E-3 * T + 0.1072763) * T - 7.643575;                     /* :37 */
  double B = (5.148E-5 * T - 4.07718E-3) * T + 0.1770383;                      /* :38 */
  double C = (((2.093236E-5 * T - 2.654387E-3) * T + 0.1412855) * T - 3.720283) * T + 4217.4; /* :39 */
  double CP0 = (B * SR + A) * S + C;                                           /* :40 */
  A = (((1.7168E-8 * T + 2.0357E-6) * T - 3.13885E-4) * T + 1.45747E-2) * T - 0.49592;       /* :42 */
  B = (((2.2956E-11 * T - 4.0027E-9) * T + 2.87533E-7) * T - 1.08645E-5) * T + 2.4931E-4;    /* :43 */
  C = ((6.136E-13 * T - 6.5637E-11) * T + 2.6380E-9) * T - 5.422E-8;                         /* :44 */
  double CP1 = ((C * P + B) * P + A) * P;                                      /* :45 */
  A = (((-2.9179E-10 * T + 2.5941E-8) * T + 9.802E-7) * T - 1.28315E-4) * T + 4.9247E-3;     /* :47 */
  B = (3.122E-8 * T - 1.517E-6) * T - 1.2331E-4;                               /* :48 */
  A = (A + B * SR) * S;                                                        /* :49 */
  B = ((1.8448E-11 * T - 2.3905E-9) * T + 1.17054E-7) * T - 2.9558E-6;         /* :50 */
  B = (B + 9.971E-8 * SR) * S;                                                 /* :51 */
  C = (3.513E-13 * T - 1.7682E-11) * T + 5.540E-10;                            /* :52 */
  C = (C - 1.4300E-12 * T * SR) * S;                                           /* :53 */
  double CP2 = ((C * P + B) * P + A) * P;                                      /* :54 */
  return CP0 + CP1 + CP2;                                                      /* :56 */
}

/* ------------------------------------------------------------------------
 * UNESCO-1980 equation of state package.
 * mckpp_physics_state_equations.F90:133-190 (abk80), :371-476 (Sig80 with
 * the BlkMod entry), :206-240 (Bet80), :244-317 (Alf80), :336-367 (Kap80).
 * The variables the reference threads through argument lists live in one
 * struct here.
 * ---------------------------------------------------------------------- */
typedef struct {
  double R1, R2, R3, R4, A, B, C, D, E, A1, B1, K, SR, P0, PK, Rho, Rho0, ABFac;
  int ABFlg, KapFlg;
} eos80;

/* BlkMod entry point, :434-476.  Returns 1 if the caller's Sig80 should go
 * on to compute Sig/Rho (i.e. not an early return). */
static int eos_blkmod(eos80 *e, double S, double T, double P)
{
  if (e->KapFlg) {                                        /* :441-444 */
    e->P0 = P / 10.0;
    e->SR = sqrt(fabs(S));
  }
  e->B1 = (-5.3009E-4 * T + 1.6483E-2) * T + 7.944E-2;                          /* :446 */
  e->A1 = ((-6.1670E-5 * T + 1.09987E-2) * T - 0.603459) * T + 54.6746;         /* :447 */
  double KW = (((-5.155288E-5 * T + 1.360477E-2) * T - 2.327105) * T + 148.4206) * T + 19652.21; /* :448 */
  double K0 = (e->B1 * e->SR + e->A1) * S + KW;                                 /* :449 */
  if (P == 0.0) {                                         /* :452-455 */
    e->K = K0;
    return 0;
  }
  e->E = (9.1697E-10 * T + 2.0816E-8) * T - 9.9348E-7;                          /* :458 */
  double BW = (5.2787E-8 * T - 6.12293E-6) * T + 8.50935E-5;                    /* :459 */
  e->B = BW + e->E * S;                                                         /* :460 */
  e->D = 1.91075E-4;                                                            /* :462 */
  e->C = (-1.6078E-6 * T - 1.0981E-5) * T + 2.2838E-3;                          /* :463 */
  double AW = ((-5.77905E-7 * T + 1.16092E-4) * T + 1.43713E-3) * T + 3.239908; /* :464 */
  e->A = (e->D * e->SR + e->C) * S + AW;                                        /* :465 */
  e->K = (e->B * e->P0 + e->A) * e->P0 + K0;                                    /* :468 */
  e->PK = e->P0 / e->K;                                                         /* :471 */
  if (e->KapFlg) return 0;                                                      /* :472 */
  return 1;
}

static void eos_sig80(eos80 *e, double S, double T, double P, double *Sig0, double *Sig)
{
  e->P0 = P / 10.0;                                       /* :407 */
  e->SR = sqrt(fabs(S));                                  /* :408 */
  e->KapFlg = 0;                                          /* :409 */
  e->R1 = ((((6.536332E-9 * T - 1.120083E-6) * T + 1.001685E-4) * T - 9.095290E-3) * T
           + 6.793952E-2) * T - .157406;                  /* :412-413 */
  e->R2 = (((5.3875E-9 * T - 8.2467E-7) * T + 7.6438E-5) * T - 4.0899E-3) * T + 8.24493E-1; /* :416 */
  e->R3 = (-1.6546E-6 * T + 1.0227E-4) * T - 5.72466E-3;  /* :417 */
  e->R4 = 4.8314E-4;                                      /* :418 */
  *Sig0 = (e->R4 * S + e->R3 * e->SR + e->R2) * S + e->R1; /* :419 */
  e->Rho0 = 1000.0 + *Sig0;                               /* :420 */
  if (P == 0.0) {                                         /* :423-427 */
    *Sig = *Sig0;
    e->Rho = e->Rho0;
    return;
  }
  if (!eos_blkmod(e, S, T, P)) return;
  *Sig = (1000.0 * e->PK + *Sig0) / (1.0 - e->PK);        /* :473 */
  e->Rho = 1000.0 + *Sig;                                 /* :474 */
}

static void eos_bet80(eos80 *e, double S, double P, double *Beta)
{
  double SR5 = e->SR * 1.5;                               /* :219 */
  double DRho = e->R2 + SR5 * e->R3 + (S + S) * e->R4;    /* :220 */
  if (P == 0) {                                           /* :221-224 */
    *Beta = DRho / e->Rho;
    return;
  }
  double DK0 = e->A1 + SR5 * e->B1;                       /* :227 */
  double DA = e->C + SR5 * e->D;                          /* :228 */
  double DB = e->E;                                       /* :229 */
  double DK = (DB * e->P0 + DA) * e->P0 + DK0;            /* :232 */
  e->ABFac = e->Rho0 * e->P0 / ((e->K - e->P0) * (e->K - e->P0)); /* :235 */
  e->ABFlg = 0;                                           /* :236 */
  *Beta = DRho / (1. - e->PK) - e->ABFac * DK;            /* :237 */
  *Beta = *Beta / e->Rho;                                 /* :238 */
}

static void eos_alf80(eos80 *e, double S, double T, double P, double *Alpha)
{
  e->R1 = (((.3268166E-7 * T - .4480332e-5) * T + .3005055e-3) * T - .1819058E-1) * T + 6.793952E-2; /* :282 */
  e->R2 = ((.215500E-7 * T - .247401E-5) * T + .152876E-3) * T - 4.0899E-3;     /* :285 */
  e->R3 = -.33092E-5 * T + 1.0227E-4;                                          /* :286 */
  double Alph0 = (e->R3 * e->SR + e->R2) * S + e->R1;                           /* :287 */
  if (P == 0.0) {                                         /* :290-293 */
    *Alpha = -Alph0 / e->Rho;
    return;
  }
  e->B1 = -.106018E-2 * T + 1.6483E-2;                                          /* :296 */
  e->A1 = (-.18501E-3 * T + .219974E-1) * T - 0.603459;                         /* :297 */
  double KW = ((-.2062115E-3 * T + .4081431E-1) * T - .4654210E+1) * T + 148.4206; /* :298 */
  double K0 = (e->B1 * e->SR + e->A1) * S + KW;                                 /* :299 */
  e->E = .183394E-8 * T + 2.0816E-8;                                            /* :302 */
  double BW = .105574E-6 * T - 6.12293E-6;                                      /* :303 */
  double AlphB = BW + e->E * S;                                                 /* :304 */
  e->C = -.32156E-5 * T - 1.0981E-5;                                            /* :305 */
  double AW = (-.1733715E-5 * T + .232184E-3) * T + 1.43713E-3;                 /* :306 */
  double AlphaA = e->C * S + AW;                                                /* :307 */
  double AlphK = (AlphB * e->P0 + AlphaA) * e->P0 + K0;                         /* :310 */
  if (e->ABFlg)                                                                 /* :311-313 */
    e->ABFac = e->Rho0 * e->P0 / ((e->K - e->P0) * (e->K - e->P0));
  *Alpha = Alph0 / (1. - e->PK) - e->ABFac * AlphK;                             /* :314 */
  *Alpha = -*Alpha / e->Rho;                                                    /* :315 */
}

static void eos_kap80(eos80 *e, double S, double T, double P, double *Kappa)
{
  if (P == 0) {                                           /* :348-355 */
    e->KapFlg = 1;
    eos_blkmod(e, S, T, P);
    *Kappa = 1.0 / e->K;
    return;
  }
  if (e->KapFlg) eos_blkmod(e, S, T, P);                  /* :358-361 */
  double DelK = e->A + (e->P0 + e->P0) * e->B;            /* :364 */
  *Kappa = (1. - e->PK * DelK) / (e->K - e->P0);          /* :365 */
}

void orc_abk80(double S, double T1, double P, double *alpha, double *beta,
               double *kappa, double *sig0, double *sig)
{
  eos80 e;
  memset(&e, 0, sizeof e);
  double T = T1;
  if (T < -2.) T = -2.;                                   /* :143-144 */
  e.KapFlg = 1;                                           /* :149 */
  e.ABFlg = 1;                                            /* :150 */
  if (*beta != 0) {                                       /* :155-161 */
    eos_sig80(&e, S, T, P, sig0, sig);
    eos_bet80(&e, S, P, beta);
  }
  if (*alpha != 0) {                                      /* :164-172 */
    if (e.KapFlg) eos_sig80(&e, S, T, P, sig0, sig);
    eos_alf80(&e, S, T, P, alpha);
  }
  if (e.KapFlg) {                                         /* :175-178 */
    *sig = 0.;
    *sig0 = 0.;
  }
  if (*kappa != 0) eos_kap80(&e, S, T, P, kappa);         /* :183-186 */
}

void orc_abk80_batch(int n, const double *s, const double *t, const double *p,
                     double *alpha, double *beta, double *sig0, double *sig)
{
  for (int i = 0; i < n; i++) {
    double a = 1.0, b = 1.0, kap = 0.0, s0 = 0.0, sg = 0.0;
    orc_abk80(s[i], t[i], p[i], &a, &b, &kap, &s0, &sg);
    alpha[i] = a; beta[i] = b; sig0[i] = s0; sig[i] = sg;
  }
}

void orc_cpsw_batch(int n, const double *s, const double *t, const double *p, double *cp)
{
  for (int i = 0; i < n; i++) cp[i] = orc_cpsw(s[i], t[i], p[i]);
}

/* ------------------------------------------------------------------------
 * 1-2-1 smoother.  mckpp_physics_verticalmixing_z121_mod.F90:7-45
 * V and w are indexed 0..kmp1.
 * ---------------------------------------------------------------------- */
void orc_z121(int kmp1, double vlo, double vhi, double *V, double *w)
{
  int km = kmp1 - 1;                                      /* :20 */
  w[0] = 0.0; w[kmp1] = 0.0; V[0] = 0.0; V[kmp1] = 0.0;    /* :22-25 */
  for (int k = 1; k <= km; k++) {                         /* :27-35 */
    if ((V[k] < vlo) || (V[k] > vhi)) w[k] = 0.0;
    else w[k] = 1.0;
  }
  for (int k = 1; k <= km; k++) {                         /* :37-43 */
    double tmp = V[k];
    V[k] = w[k - 1] * V[0] + 2. * V[k] + w[k + 1] * V[k + 1];
    double wait = w[k - 1] + 2.0 + w[k + 1];
    V[k] = V[k] / wait;
    V[0] = tmp;
  }
}

/* ------------------------------------------------------------------------
 * wm/ws lookup tables.  mckpp_physics_lookup_mod.F90:11-66.
 * Layout wmt(0:891,0:49) column-major: element (i,j) at [j*892 + i].
 * Integer powers follow amdflang's lowering measured in this container:
 * x**3 = (x*x)*x, x**4 = ((x*x)*x)*x.
 * ---------------------------------------------------------------------- */
#define TBL(i, j) ((j) * (ORC_NI + 2) + (i))
/* The C lowering this file uses for the Fortran power operators and literals on the path
 * (checked against amdflang's by tests/test_oracle_cpu.py::test_compiler_conventions, oracle/conv_probe.F90):
 *   x**3 -> (x*x)*x, x**4 -> ((x*x)*x)*x, x**(1./2.) -> sqrt(x), x**(1./3.) -> pow(x, 1./3.), x**(1./4.) -> pow(x, 1./4.) */
void orc_conv_probe(int n, const double *x, double *p3, double *p4, double *ph, double *pt, double *pq)
{
  for (int i = 0; i < n; ++i) {
    const double y = x[i];
    p3[i] = (y * y) * y;
    p4[i] = ((y * y) * y) * y;
    ph[i] = sqrt(y);
    pt[i] = pow(y, 1. / 3.);
    pq[i] = pow(y, 1. / 4.);
  }
}
void orc_conv_literals(double *out)
{
  out[0] = 1.257; out[1] = 8.380; out[2] = 98.96; out[3] = -28.86; out[4] = 4.e-7; out[5] = 0.033;
  out[6] = 1. / 3.; out[7] = 0.04 / 49.; out[8] = 1.E-12; out[9] = 6.536332E-9; out[10] = 0.1;
}

void orc_lookup(double vonk, double *wmt, double *wst) { orc_lookup_mode(vonk, wmt, wst, 0); }

void orc_lookup_mode(double vonk, double *wmt, double *wst, int half_pow_mode)
{
  const int ni = ORC_NI, nj = ORC_NJ;
  const double epsln = 1.e-20, c1 = 5.0, zmin = -4.e-7, zmax = 0.0, umin = 0.0, umax = 0.04;
  const double am = 1.257, cm = 8.380, c2 = 16.0, zetam = -0.2;
  const double as = -28.86, cs = 98.96, c3 = 16.0, zetas = -1.0;
  double deltaz = (zmax - zmin) / (ni + 1);               /* :42 */
  double deltau = (umax - umin) / (nj + 1);               /* :43 */
  for (int i = 0; i <= ni + 1; i++) {                     /* :45 */
    double zehat = deltaz * (i) + zmin;                   /* :46 */
    for (int j = 0; j <= nj + 1; j++) {
      double usta = deltau * (j) + umin;                  /* :48 */
      double u3 = (usta * usta) * usta;
      double zeta = zehat / (u3 + epsln);                 /* :49 */
      if (zehat >= 0.) {                                  /* :51-53 */
        wmt[TBL(i, j)] = vonk * usta / (1. + c1 * zeta);
        wst[TBL(i, j)] = wmt[TBL(i, j)];
      } else {
        if (zeta > zetam)                                 /* :55-59 */
          wmt[TBL(i, j)] = vonk * usta * pow(1. - c2 * zeta, 1. / 4.);
        else
          wmt[TBL(i, j)] = vonk * pow(am * u3 - cm * zehat, 1. / 3.);
        if (zeta > zetas)                                 /* :60-64 */
          wst[TBL(i, j)] = vonk * usta * (half_pow_mode ? pow(1. - c3 * zeta, 1. / 2.)
                                                        : sqrt(1. - c3 * zeta));   /* **(1./2.): a square root under amdflang, conv_probe.F90 */
        else
          wst[TBL(i, j)] = vonk * pow(as * u3 - cs * zehat, 1. / 3.);
      }
    }
  }
}

/* ------------------------------------------------------------------------
 * Turbulent velocity scales.  mckpp_physics_verticalmixing_wscale_mod.F90:12-97
 * ---------------------------------------------------------------------- */
void orc_wscale(const orc_const *c, double sigma, double hbl, double ustar,
                double bfsfc, double *wm, double *ws)
{
  const int ni = ORC_NI, nj = ORC_NJ;
  const double zmin = -4.e-7, zmax = 0.0, umin = 0.0, umax = 0.04, c1 = 5.0;
  double deltaz = (zmax - zmin) / (ni + 1);               /* :57 */
  double deltau = (umax - umin) / (nj + 1);               /* :58 */
  double zehat = c->vonk * sigma * hbl * bfsfc;           /* :61 */
  if (zehat <= zmax) {                                    /* :63 */
    double zdiff = zehat - zmin;
    int iz = (int)(zdiff / deltaz);                       /* :65 */
    if (iz > ni) iz = ni;
    if (iz < 0) iz = 0;
    int izp1 = iz + 1;
    double udiff = ustar - umin;
    int ju = (int)(udiff / deltau);                       /* :71 */
    if (ju > nj) ju = nj;
    if (ju < 0) ju = 0;
    int jup1 = ju + 1;
    double zfrac = zdiff / deltaz - (double)iz;           /* :76 */
    double ufrac = udiff / deltau - (double)ju;           /* :77 */
    double fzfrac = 1. - zfrac;
    double wam = (fzfrac) * c->wmt[TBL(iz, jup1)] + zfrac * c->wmt[TBL(izp1, jup1)]; /* :80-81 */
    double wbm = (fzfrac) * c->wmt[TBL(iz, ju)] + zfrac * c->wmt[TBL(izp1, ju)];     /* :82-83 */
    *wm = (1. - ufrac) * wbm + ufrac * wam;                                          /* :84 */
    double was = (fzfrac) * c->wst[TBL(iz, jup1)] + zfrac * c->wst[TBL(izp1, jup1)]; /* :86-87 */
    double wbs = (fzfrac) * c->wst[TBL(iz, ju)] + zfrac * c->wst[TBL(izp1, ju)];     /* :88-89 */
    *ws = (1. - ufrac) * wbs + ufrac * was;                                          /* :90 */
  } else {
    double ucube = (ustar * ustar) * ustar;               /* :92 */
    *wm = c->vonk * ustar * ucube / (ucube + c1 * zehat); /* :93 */
    *ws = *wm;
  }
}

/* Jerlov water-type tables: swfrac_mod.F90:31-33, fluxes_mod.F90:130-132 */
static const double jer_rfac[6] = {0, 0.58, 0.62, 0.67, 0.77, 0.78};
static const double jer_a1[6] = {0, 0.35, 0.6, 1.0, 1.5, 1.4};
static const double jer_a2[6] = {0, 23.0, 20.0, 17.0, 14.0, 7.9};

/* mckpp_physics_swfrac, swfrac_mod.F90:49-79 (and the body of swfrac_opt :36-41) */
double orc_swfrac(const orc_const *c, double fact, double z, int jwtype)
{
  const double rmin = -80.;
  double r1 = fmax2(z * fact / jer_a1[jwtype], rmin);     /* :74 */
  double r2 = fmax2(z * fact / jer_a2[jwtype], rmin);     /* :75 */
  return jer_rfac[jwtype] * orc_exp(c, r1) + (1. - jer_rfac[jwtype]) * orc_exp(c, r2); /* :76 */
}

/* mckpp_fluxes_swdk, fluxes_mod.F90:121-137 */
double orc_swdk(const orc_const *c, double z, int j)
{
  return jer_rfac[j] * orc_exp(c, z / jer_a1[j]) + (1.0 - jer_rfac[j]) * orc_exp(c, z / jer_a2[j]);
}

/* Round 5: the intrinsics this file lowers to libm calls and C operators, through the very helpers the physics
 * below uses (exp as orc_exp does with exp_mode 0, sqrt, fabs, fsign, fmax2 / fmin2 left to right, int casts),
 * against conv_probe.F90's conv_probe_unary / _swfrac / _binary / _casts. */
void orc_conv_unary(int n, const double *x, double *e, double *s, double *a, double *q)
{
  orc_const c;
  memset(&c, 0, sizeof c);   /* exp_mode 0: libm exp, "the reference's EXP" */
  for (int i = 0; i < n; ++i) {
    const double y = x[i];
    e[i] = orc_exp(&c, y);
    s[i] = sqrt(fabs(y));
    a[i] = fabs(y);
    q[i] = y * y;
  }
}
void orc_conv_swfrac(int n, const double *z, double fact, int jwtype, double *sw, double *sk)
{
  orc_const c;
  memset(&c, 0, sizeof c);
  for (int i = 0; i < n; ++i) {
    sw[i] = orc_swfrac(&c, fact, z[i], jwtype);
    sk[i] = orc_swdk(&c, z[i], jwtype);
  }
}
void orc_conv_jerlov(int jwtype, double *out) { out[0] = jer_a1[jwtype]; out[1] = jer_a2[jwtype]; out[2] = jer_rfac[jwtype]; }
void orc_conv_binary(int n, const double *a, const double *b, const double *c, const double *d, double *sg, double *sh,
                     double *se, double *mx, double *mn, double *ax, double *an, double *m3, double *m4, double *x3)
{
  const double epsln = 1.e-16;
  for (int i = 0; i < n; ++i) {
    sg[i] = fsign(a[i], b[i]);
    sh[i] = 0.5 + fsign(0.5, b[i]);
    se[i] = 0.5 + fsign(0.5, b[i] + epsln);
    mx[i] = fmax2(a[i], b[i]);
    mn[i] = fmin2(a[i], b[i]);
    ax[i] = fmax2(a[i], b[i]);
    an[i] = fmin2(a[i], b[i]);
    m3[i] = fmin2(fmin2(a[i], b[i]), c[i]);
    m4[i] = fmin2(fmin2(fmin2(a[i], b[i]), c[i]), d[i]);
    x3[i] = fmax2(fmax2(a[i], b[i]), c[i]);
  }
}
void orc_conv_casts(int n, const double *x, int *ifx, int *itr, int *icl, double *fl)
{
  const double epsln = 1.e-20;
  for (int i = 0; i < n; ++i) {
    ifx[i] = (int)(x[i] + epsln);
    itr[i] = (int)x[i];
    int iz = (int)x[i];
    iz = iz < 890 ? iz : 890;
    iz = iz > 0 ? iz : 0;
    icl[i] = iz;
    fl[i] = x[i] - (double)iz;
  }
}


/* ------------------------------------------------------------------------
 * Tridiagonal pieces.  mckpp_physics_solvers.F90
 * ---------------------------------------------------------------------- */
/* tridcof :14-44.  diff indexed 0..nzi; cu,cc,cl indexed 1..nzi. */
void orc_tridcof(const orc_const *c, const double *diff, int nzi, double *cu,
                 double *cc, double *cl)
{
  cu[1] = 0.;                                             /* :31 */
  cc[1] = 1. + c->tri1[1] * diff[1];                      /* :32 */
  cl[1] = -c->tri1[1] * diff[1];                          /* :33 */
  for (int i = 2; i <= nzi; i++) {                        /* :36-40 */
    cu[i] = -c->tri0[i] * diff[i - 1];
    cc[i] = 1. + c->tri1[i] * diff[i] + c->tri0[i] * diff[i - 1];
    cl[i] = -c->tri1[i] * diff[i];
  }
  cl[nzi] = 0.;                                           /* :43 */
}

/* tridrhs :53-107.  h, yo 1..nzi+1; ntflux, diff 0..nzi; ghat, rhs 1..nzi */
static void tridrhs(const orc_const *c, int npd, const double *h, const double *yo,
                    const double *ntflux, const double *diff, const double *ghat,
                    double sturflux, double ghatflux, double dto, int nzi, double *rhs)
{
  double divflx = 1.0 / (double)npd;                      /* :78 */
  rhs[1] = yo[1] + dto / h[1] * (ghatflux * diff[1] * ghat[1] - sturflux * divflx
                                 + ntflux[1] - ntflux[0]);          /* :81-82 */
  if (npd >= 2) {                                         /* :85-93 */
    for (int i = 2; i <= npd; i++)
      rhs[i] = yo[i] + dto / h[i] * (ghatflux * diff[i] * ghat[i]
                                     - ghatflux * diff[i - 1] * ghat[i - 1]
                                     - sturflux * divflx + ntflux[i] - ntflux[i - 1]);
  }
  for (int i = npd + 1; i <= nzi - 1; i++)                /* :96-99 */
    rhs[i] = yo[i] + dto / h[i] * (ghatflux * (diff[i] * ghat[i] - diff[i - 1] * ghat[i - 1])
                                   + ntflux[i] - ntflux[i - 1]);
  if (nzi > 1) {                                          /* :102-106 */
    int i = nzi;
    rhs[i] = yo[i] + dto / h[i] * (ghatflux * (diff[i] * ghat[i] - diff[i - 1] * ghat[i - 1])
                                   + ntflux[i] - ntflux[i - 1])
             + yo[i + 1] * c->tri1[i] * diff[i];
  }
}

/* tridmat :112-161.  Returns 1 if a zero pivot was met (the reference
 * prints and calls mckpp_abort -> STOP; the statement after the abort sets
 * bet=1e-12, which is what is done here so the caller can flag the column).
 * gam is caller scratch indexed 2..nzi. */
int orc_tridmat(const double *cu, const double *cc, const double *cl,
                const double *rhs, const double *yo, int nzi, double *yn, double *gam)
{
  int bad = 0;
  double bet = cc[1];                                     /* :135 */
  yn[1] = rhs[1] / bet;                                   /* :136 */
  for (int i = 2; i <= nzi; i++) {                        /* :137-154 */
    gam[i] = cl[i - 1] / bet;
    bet = cc[i] - cu[i] * gam[i];
    if (bet == 0.) {
      bad = 1;
      bet = 1.E-12;                                       /* :150 */
    }
    yn[i] = (rhs[i] - cu[i] * yn[i - 1]) / bet;           /* :153 */
  }
  for (int i = nzi - 1; i >= 1; i--)                      /* :156-158 */
    yn[i] = yn[i] - gam[i + 1] * yn[i + 1];
  yn[nzi + 1] = yo[nzi + 1];                              /* :159 */
  return bad;
}

/* The library's opt-in solver mode 1 (NOT the reference's operation order; orc_const.solver_mode): the same
 * tridiagonal system, eliminated from both ends at once - downward from level 1 as tridmat does
 * (solvers.F90:135-154) for the levels 1..m, m = nzi/2, upward from level nzi by the mirrored recurrence for the
 * levels nzi..m+1 - then the two remaining unknowns y(m), y(m+1) from their 2x2 system, then the two back
 * substitutions away from the middle.  Each half is a chain of nzi/2 dependent steps instead of nzi.  Every
 * operation is one IEEE operation in the order written here; the HIP kernels' solver mode 1 performs exactly these.
 *   upper half:  gam(i) = cl(i-1)/bet, bet = cc(i) - cu(i) gam(i), z(i) = (rhs(i) - cu(i) z(i-1))/bet
 *   lower half:  g(i+1) = cu(i+1)/bet, bet = cc(i) - cl(i) g(i+1), z(i) = (rhs(i) - cl(i) z(i+1))/bet
 *   middle:      y(m) = (z(m) - gam(m+1) z(m+1)) / (1 - gam(m+1) g(m+1)),  y(m+1) = z(m+1) - g(m+1) y(m)
 *   back:        y(i) = z(i) - gam(i+1) y(i+1), i = m-1..1;   y(i) = z(i) - g(i) y(i-1), i = m+2..nzi
 * A zero pivot is replaced as solvers.F90:140-151 does, in either half.  gam[2..m] holds the upper half's
 * multipliers, gam[m+2..nzi] the lower half's. */
int orc_tridmat_2e(const double *cu, const double *cc, const double *cl,
                   const double *rhs, const double *yo, int nzi, double *yn, double *gam)
{
  if (nzi < 2) return orc_tridmat(cu, cc, cl, rhs, yo, nzi, yn, gam);
  const int m = nzi / 2;
  int bad = 0;
  double bet = cc[1];
  yn[1] = rhs[1] / bet;
  for (int i = 2; i <= m; i++) {
    gam[i] = cl[i - 1] / bet;
    bet = cc[i] - cu[i] * gam[i];
    if (bet == 0.) { bad = 1; bet = 1.E-12; }
    yn[i] = (rhs[i] - cu[i] * yn[i - 1]) / bet;
  }
  const double gt = cl[m] / bet;                          /* gam(m+1) */
  double betb = cc[nzi];
  if (betb == 0.) { bad = 1; betb = 1.E-12; }             /* the lower half's first pivot is one of tridmat's checked ones */
  yn[nzi] = rhs[nzi] / betb;
  for (int i = nzi - 1; i >= m + 1; i--) {
    const double g = cu[i + 1] / betb;
    gam[i + 1] = g;
    betb = cc[i] - cl[i] * g;
    if (betb == 0.) { bad = 1; betb = 1.E-12; }
    yn[i] = (rhs[i] - cl[i] * yn[i + 1]) / betb;
  }
  const double gb = cu[m + 1] / betb;                     /* g(m+1) */
  double den = 1. - gt * gb;
  if (den == 0.) { bad = 1; den = 1.E-12; }               /* the pivot of the 2x2 system in the middle: treated like tridmat's */
  const double ym = (yn[m] - gt * yn[m + 1]) / den;
  const double ym1 = yn[m + 1] - gb * ym;
  yn[m] = ym;
  yn[m + 1] = ym1;
  for (int i = m - 1; i >= 1; i--) yn[i] = yn[i] - gam[i + 1] * yn[i + 1];
  for (int i = m + 2; i <= nzi; i++) yn[i] = yn[i] - gam[i] * yn[i - 1];
  yn[nzi + 1] = yo[nzi + 1];
  return bad;
}

static int tridmat_mode(const orc_const *c, const double *cu, const double *cc, const double *cl,
                        const double *rhs, const double *yo, int nzi, double *yn, double *gam)
{
  return c->solver_mode == 1 ? orc_tridmat_2e(cu, cc, cl, rhs, yo, nzi, yn, gam)
                             : orc_tridmat(cu, cc, cl, rhs, yo, nzi, yn, gam);
}

/* ------------------------------------------------------------------------
 * Per-column working state (the reference's kpp_1d_type, hot-path subset).
 * ---------------------------------------------------------------------- */
typedef struct {
  int nz, nzp1;
  double *U[3], *X[3];          /* [1..2][1..nzp1] */
  double *Us[3][2], *Xs[3][2];  /* [comp][time level][1..nzp1] */
  double *U_init[3];
  double *rho, *cp, *talpha, *sbeta;   /* 0..nzp1 */
  double *buoy;                        /* 1..nzp1 */
  double *difm, *difs, *dift;          /* 0..nzp1 */
  double *ghat;                        /* 1..nzp1 */
  double *wU[3], *wX[4], *wXNT[3];     /* 0..nzp1 */
  double *Rig, *dbloc, *Shsq;          /* 1..nzp1 */
  double *swfrac;                      /* 1..nzp1 */
  double *swdk_opt;                    /* 0..nz */
  double *tinc_fcorr, *sinc_fcorr, *ocnTcorr, *scorr;
  double *fcorr_withz, *sfcorr_withz, *ocnT_clim, *sal_clim;
  double rhoh2o, ocdepth, f, relax_sst, fcorr, SST0, fcorr_twod, relax_sal, relax_ocnT;
  double hmix, kmix, Tref, uref, vref, Ssurf, Sref, SSref;
  double reset_flag, dampu_flag, dampv_flag, freeze_flag;
  double sflux[7];
  double hmixd[2];
  int old, newi, jerlov, l_initflag, l_ocean, comp_flag;
  int nmodeadv[3];
  int modeadv[ORC_MAXMODEADV + 1][3];
  double advection[ORC_MAXMODEADV + 1][3];
  int status, npasses;
  /* scratch */
  double *dVsq, *Ritop, *alphaDT, *betaDS;
  double *blmc[4];
  double *cu, *cc, *cl, *rhs, *diff, *gcap, *ntflx[3], *gam;
  double *Uo[3], *Xo[3], *Ux[3], *Xx[3];
  double *mem;
} orc_col;

static orc_col *col_new(int nz)
{
  orc_col *q = (orc_col *)calloc(1, sizeof(orc_col));
  q->nz = nz;
  q->nzp1 = nz + 1;
  int n = nz + 4;                 /* indices 0..nzp1+1 usable */
  int narr = 80;
  q->mem = (double *)calloc((size_t)narr * n, sizeof(double));
  double *p = q->mem;
#define TAKE(x) do { (x) = p; p += n; } while (0)
  for (int l = 1; l <= 2; l++) {
    TAKE(q->U[l]); TAKE(q->X[l]); TAKE(q->U_init[l]);
    for (int t = 0; t < 2; t++) { TAKE(q->Us[l][t]); TAKE(q->Xs[l][t]); }
    TAKE(q->Uo[l]); TAKE(q->Xo[l]); TAKE(q->Ux[l]); TAKE(q->Xx[l]);
    TAKE(q->wXNT[l]); TAKE(q->ntflx[l]);
  }
  TAKE(q->rho); TAKE(q->cp); TAKE(q->talpha); TAKE(q->sbeta); TAKE(q->buoy);
  TAKE(q->difm); TAKE(q->difs); TAKE(q->dift); TAKE(q->ghat);
  for (int l = 1; l <= 2; l++) TAKE(q->wU[l]);
  for (int l = 1; l <= 3; l++) TAKE(q->wX[l]);
  TAKE(q->Rig); TAKE(q->dbloc); TAKE(q->Shsq); TAKE(q->swfrac); TAKE(q->swdk_opt);
  TAKE(q->tinc_fcorr); TAKE(q->sinc_fcorr); TAKE(q->ocnTcorr); TAKE(q->scorr);
  TAKE(q->fcorr_withz); TAKE(q->sfcorr_withz); TAKE(q->ocnT_clim); TAKE(q->sal_clim);
  TAKE(q->dVsq); TAKE(q->Ritop); TAKE(q->alphaDT); TAKE(q->betaDS);
  for (int l = 1; l <= 3; l++) TAKE(q->blmc[l]);
  TAKE(q->cu); TAKE(q->cc); TAKE(q->cl); TAKE(q->rhs); TAKE(q->diff); TAKE(q->gcap); TAKE(q->gam);
#undef TAKE
  if ((p - q->mem) > (long)narr * n) abort();
  return q;
}

static void col_free(orc_col *q)
{
  free(q->mem);
  free(q);
}

/* ------------------------------------------------------------------------
 * swfrac_opt.  mckpp_physics_swfrac_mod.F90:14-43
 * ---------------------------------------------------------------------- */
static void swfrac_opt(const orc_const *c, orc_col *q, double fact)
{
  for (int l = 1; l <= q->nzp1; l++)
    q->swfrac[l] = orc_swfrac(c, fact, c->zm[l], q->jerlov);
}

/* mckpp_fluxes_ntflux.  mckpp_fluxes_mod.F90:93-118 */
static void ntflux(const orc_const *c, orc_col *q, int ntime)
{
  if (ntime <= 1)                                         /* :103-108 */
    for (int k = 0; k <= q->nz; k++) q->swdk_opt[k] = orc_swdk(c, -c->dm[k], q->jerlov);
  if (ntime >= 1)                                         /* :110-116 */
    for (int k = 0; k <= q->nz; k++)
      q->wXNT[1][k] = -q->sflux[3] * q->swdk_opt[k] / (q->rho[0] * q->cp[0]);
}

/* ------------------------------------------------------------------------
 * rimix.  mckpp_physics_verticalmixing_rimix_mod.F90:13-106
 * ---------------------------------------------------------------------- */
static void rimix(const orc_const *c, orc_col *q, int km, int kmp1)
{
  const double epsln = 1.e-16, Riinfty = 0.8, Ricon = -0.2, difm0 = 0.005, difs0 = 0.005;
  const double difmiw = 0.0001, difsiw = 0.00001, difmcon = 0.0000, difscon = 0.0000;
  const double c1 = 1.0, c0 = 0.0;
  const int mRi = 1;
  for (int ki = 1; ki <= km; ki++) {                      /* :47-52 */
    q->Rig[ki] = q->dbloc[ki] * (c->zm[ki] - c->zm[ki + 1]) / (q->Shsq[ki] + epsln);
    q->dift[ki] = q->Rig[ki];
    q->difm[ki] = q->dift[ki];
  }
  for (int j = 1; j <= mRi; j++)                          /* :56-58 */
    orc_z121(kmp1, c0, Riinfty, q->difm, q->difs);
  for (int ki = 1; ki <= km; ki++) {                      /* :62-97 */
    double Rigg = fmax2(q->dift[ki], Ricon);
    double ratio = fmin2((Ricon - Rigg) / Ricon, c1);
    double fcon = (c1 - ratio * ratio);
    fcon = fcon * fcon * fcon;
    Rigg = fmax2(q->difm[ki], c0);
    ratio = fmin2(Rigg / Riinfty, c1);
    double fri = (c1 - ratio * ratio);
    fri = fri * fri * fri;
    q->difm[ki] = (difmiw + fcon * difmcon + fri * difm0);
    q->difs[ki] = (difsiw + fcon * difscon + fri * difs0);
    q->dift[ki] = q->difs[ki];
  }
  q->difm[0] = c0;                                        /* :102-104 */
  q->dift[0] = c0;
  q->difs[0] = c0;
}

/* ddmix.  mckpp_physics_verticalmixing_ddmix_mod.F90:12-52 */
static void ddmix(const orc_const *c, orc_col *q, int km)
{
  const double Rrho0 = 1.9, dsfmax = 1.0e-4;
  for (int ki = 1; ki <= km; ki++) {
    double aDT = q->alphaDT[ki], bDS = q->betaDS[ki];
    if ((aDT > bDS) && (bDS > 0.)) {                      /* :31-36 */
      double Rrho = fmin2(aDT / bDS, Rrho0);
      double r = ((Rrho - 1) / (Rrho0 - 1));
      double diffdd = 1.0 - r * r;
      diffdd = dsfmax * diffdd * diffdd * diffdd;
      q->dift[ki] = q->dift[ki] + diffdd * 0.8 / Rrho;
      q->difs[ki] = q->difs[ki] + diffdd;
    } else if ((aDT < 0.0) && (bDS < 0.0) && (aDT < bDS)) { /* :39-48 */
      double Rrho = aDT / bDS;
      double diffdd = 1.5e-6 * 9.0 * 0.101 * orc_exp(c, 4.6 * orc_exp(c, -0.54 * (1 / Rrho - 1)));
      double prandtl = 0.15 * Rrho;
      if (Rrho > 0.5) prandtl = (1.85 - 0.85 / Rrho) * Rrho;
      q->dift[ki] = q->dift[ki] + diffdd;
      q->difs[ki] = q->difs[ki] + prandtl * diffdd;
    }
  }
}

/* ------------------------------------------------------------------------
 * bldepth.  mckpp_physics_verticalmixing_bldepth_mod.F90:32-203
 * ---------------------------------------------------------------------- */
static void bldepth(const orc_const *c, orc_col *q, int ntime, int km, int kmp1,
                    const double *dVsq, const double *Ritop, double ustar, double Bo,
                    double Bosol, double *hbl_o, double *bfsfc_o, double *stable_o,
                    double *caseA_o, int *kbl_o)
{
  const double epsln = 1.e-16, Ricr = 0.30, epsilon = 0.1, cekman = 0.7, cmonob = 1.0;
  const double cs = 98.96, cv = 1.6, hbf = 1.0;
  const double *zm = c->zm;
  double Rib[3], dmo[3];
  double bfsfc = 0, stable = 0, sigma = 0, caseA = 0, wm, ws;
  double Vtc = cv * sqrt(0.2 / cs / epsilon) / (c->vonk * c->vonk) / Ricr;  /* :91 */
  int ka = 1, ku = 2;
  Rib[ka] = 0.0;                                          /* :99 */
  dmo[ka] = -zm[kmp1];
  int kbl = km;
  double hbl = -zm[km];
  double hek = cekman * ustar / (fabs(q->f) + epsln);     /* :103 */
  for (int kl = 2; kl <= km; kl++) {                      /* :105 */
    if (ntime <= 1 && kl == 2) swfrac_opt(c, q, hbf);     /* :113-115 */
    if (kbl >= km) {                                      /* :117-125 */
      caseA = -zm[kl];
      bfsfc = Bo + Bosol * (1. - q->swfrac[kl]);
      stable = 0.5 + fsign(0.5, bfsfc + epsln);
      sigma = stable * 1. + (1. - stable) * epsilon;
    }
    orc_wscale(c, sigma, caseA, ustar, bfsfc, &wm, &ws);  /* :128 */
    if (kbl >= km) {                                      /* :130 */
      double bvsq = 0.5 * (q->dbloc[kl - 1] / (zm[kl - 1] - zm[kl]) +
                           q->dbloc[kl] / (zm[kl] - zm[kl + 1]));           /* :132-133 */
      double Vtsq = -zm[kl] * ws * sqrt(fabs(bvsq)) * Vtc;                  /* :134 */
      Rib[ku] = Ritop[kl] / (dVsq[kl] + Vtsq + epsln);                      /* :136 */
      Rib[ku] = fmax2(Rib[ku], Rib[ka] + epsln);                            /* :137 */
      double hri = -zm[kl - 1] + (zm[kl - 1] - zm[kl]) * (Ricr - Rib[ka]) / (Rib[ku] - Rib[ka]); /* :139-140 */
      double fmonob = stable * 1.0;                                         /* :144 */
      dmo[ku] = cmonob * ustar * ustar * ustar / c->vonk / (fabs(bfsfc) + epsln); /* :145-146 */
      dmo[ku] = fmonob * dmo[ku] - (1. - fmonob) * zm[kmp1];                /* :147 */
      double hmonob;
      if (dmo[ku] <= (-zm[kl])) {                                           /* :148-153 */
        hmonob = (dmo[ku] - dmo[ka]) / (zm[kl - 1] - zm[kl]);
        hmonob = (dmo[ku] + hmonob * zm[kl]) / (1. - hmonob);
      } else {
        hmonob = -zm[kmp1];
      }
      double fekman = stable * 1.0;                                         /* :157 */
      double hekman = fekman * hek - (1. - fekman) * zm[kmp1];              /* :158 */
      double hmin = fmin2(fmin2(fmin2(hri, hmonob), hekman), -q->ocdepth);  /* :161 */
      if (hmin < -zm[kl]) {                                                 /* :162 */
        if (!q->l_initflag) {                                               /* :173-180 */
          if (hmin < -zm[kl - 1]) {
            double hmin2 = fmin2(fmin2(hri, hmonob), -q->ocdepth);
            if (hmin2 < -zm[kl]) hmin = hmin2;
          }
        }
        hbl = hmin;                                                         /* :182-183 */
        kbl = kl;
      }
    }
    int ksave = ka;                                       /* :188-190 */
    ka = ku;
    ku = ksave;
  }
  bfsfc = orc_swfrac(c, -1.0, hbl, q->jerlov);            /* :193 */
  bfsfc = Bo + Bosol * (1. - bfsfc);                      /* :195 */
  stable = 0.5 + fsign(0.5, bfsfc);                       /* :196 */
  bfsfc = bfsfc + stable * epsln;                         /* :197 */
  caseA = 0.5 + fsign(0.5, -zm[kbl] - 0.5 * c->hm[kbl] - hbl); /* :201 */
  *hbl_o = hbl; *bfsfc_o = bfsfc; *stable_o = stable; *caseA_o = caseA; *kbl_o = kbl;
}

/* ------------------------------------------------------------------------
 * blmix.  mckpp_physics_verticalmixing_blmix_mod.F90:13-151
 * ---------------------------------------------------------------------- */
static void blmix(const orc_const *c, orc_col *q, int km, double ustar, double bfsfc,
                  double hbl, double stable, double caseA, int kbl, double *dkm1 /*1..3*/)
{
  const double epsln = 1.e-20, epsilon = 0.1, c1 = 5.0, cs = 98.96, cstar = 5.0;
  const double *zm = c->zm, *hm = c->hm;
  double gat1[4], dat1[4], wm, ws;
  double cg = cstar * c->vonk * pow(cs * c->vonk * epsilon, 1. / 3.);       /* :62 */
  double sigma = stable * 1.0 + (1. - stable) * epsilon;                    /* :65 */
  orc_wscale(c, sigma, hbl, ustar, bfsfc, &wm, &ws);                        /* :67 */
  int ifx = (int)(caseA + epsln);
  int kn = ifx * (kbl - 1) + (1 - ifx) * kbl;                               /* :68 */
  double delhat = 0.5 * hm[kn] - zm[kn] - hbl;                              /* :71 */
  double R = 1.0 - delhat / hm[kn];                                         /* :72 */
  double dvdzup = (q->difm[kn - 1] - q->difm[kn]) / hm[kn];                 /* :73 */
  double dvdzdn = (q->difm[kn] - q->difm[kn + 1]) / hm[kn + 1];             /* :74 */
  double viscp = 0.5 * ((1. - R) * (dvdzup + fabs(dvdzup)) + R * (dvdzdn + fabs(dvdzdn))); /* :75 */
  dvdzup = (q->difs[kn - 1] - q->difs[kn]) / hm[kn];                        /* :77 */
  dvdzdn = (q->difs[kn] - q->difs[kn + 1]) / hm[kn + 1];
  double difsp = 0.5 * ((1. - R) * (dvdzup + fabs(dvdzup)) + R * (dvdzdn + fabs(dvdzdn)));
  dvdzup = (q->dift[kn - 1] - q->dift[kn]) / hm[kn];                        /* :81 */
  dvdzdn = (q->dift[kn] - q->dift[kn + 1]) / hm[kn + 1];
  double diftp = 0.5 * ((1. - R) * (dvdzup + fabs(dvdzup)) + R * (dvdzdn + fabs(dvdzdn)));
  double visch = q->difm[kn] + viscp * delhat;                              /* :85-87 */
  double difsh = q->difs[kn] + difsp * delhat;
  double difth = q->dift[kn] + diftp * delhat;
  double u4 = ((ustar * ustar) * ustar) * ustar;          /* ustar**4 as amdflang lowers it */
  double f1 = stable * c1 * bfsfc / (u4 + epsln);                           /* :89 */
  gat1[1] = visch / hbl / (wm + epsln);                                     /* :90 */
  dat1[1] = -viscp / (wm + epsln) + f1 * visch;
  dat1[1] = fmin2(dat1[1], 0.);
  gat1[2] = difsh / hbl / (ws + epsln);                                     /* :94 */
  dat1[2] = -difsp / (ws + epsln) + f1 * difsh;
  dat1[2] = fmin2(dat1[2], 0.);
  gat1[3] = difth / hbl / (ws + epsln);                                     /* :98 */
  dat1[3] = -diftp / (ws + epsln) + f1 * difth;
  dat1[3] = fmin2(dat1[3], 0.);
  for (int ki = 1; ki <= km; ki++) {                                        /* :110-133 */
    double sig = (-zm[ki] + 0.5 * hm[ki]) / hbl;
    sigma = stable * sig + (1. - stable) * fmin2(sig, epsilon);
    orc_wscale(c, sigma, hbl, ustar, bfsfc, &wm, &ws);
    sig = (-zm[ki] + 0.5 * hm[ki]) / hbl;
    double a1 = sig - 2.;
    double a2 = 3. - 2. * sig;
    double a3 = sig - 1.;
    double Gm = a1 + a2 * gat1[1] + a3 * dat1[1];
    double Gs = a1 + a2 * gat1[2] + a3 * dat1[2];
    double Gt = a1 + a2 * gat1[3] + a3 * dat1[3];
    q->blmc[1][ki] = hbl * wm * sig * (1. + sig * Gm);
    q->blmc[2][ki] = hbl * ws * sig * (1. + sig * Gs);
    q->blmc[3][ki] = hbl * ws * sig * (1. + sig * Gt);
    q->ghat[ki] = (1. - stable) * cg / (ws * hbl + epsln);
  }
  double sig = -zm[kbl - 1] / hbl;                                          /* :136 */
  sigma = stable * sig + (1. - stable) * fmin2(sig, epsilon);
  orc_wscale(c, sigma, hbl, ustar, bfsfc, &wm, &ws);
  sig = -zm[kbl - 1] / hbl;
  double a1 = sig - 2.;
  double a2 = 3. - 2. * sig;
  double a3 = sig - 1.;
  double Gm = a1 + a2 * gat1[1] + a3 * dat1[1];
  double Gs = a1 + a2 * gat1[2] + a3 * dat1[2];
  double Gt = a1 + a2 * gat1[3] + a3 * dat1[3];
  dkm1[1] = hbl * wm * sig * (1. + sig * Gm);                               /* :147-149 */
  dkm1[2] = hbl * ws * sig * (1. + sig * Gs);
  dkm1[3] = hbl * ws * sig * (1. + sig * Gt);
}

/* enhance.  mckpp_physics_verticalmixing_enhance_mod.F90:10-51 */
static void enhance(const orc_const *c, orc_col *q, int km, const double *dkm1,
                    double hbl, int kbl, double caseA)
{
  const double *zm = c->zm;
  for (int ki = 1; ki <= km - 1; ki++) {
    if (ki == (kbl - 1)) {
      double delta = (hbl + zm[ki]) / (zm[ki] - zm[ki + 1]);                /* :34 */
      double dkmp5 = caseA * q->difm[ki] + (1. - caseA) * q->blmc[1][ki];
      double dstar = ((1. - delta) * (1. - delta)) * dkm1[1] + (delta * delta) * dkmp5;
      q->blmc[1][ki] = (1. - delta) * q->difm[ki] + delta * dstar;
      dkmp5 = caseA * q->difs[ki] + (1. - caseA) * q->blmc[2][ki];
      dstar = ((1. - delta) * (1. - delta)) * dkm1[2] + (delta * delta) * dkmp5;
      q->blmc[2][ki] = (1. - delta) * q->difs[ki] + delta * dstar;
      dkmp5 = caseA * q->dift[ki] + (1. - caseA) * q->blmc[3][ki];
      dstar = ((1. - delta) * (1. - delta)) * dkm1[3] + (delta * delta) * dkmp5;
      q->blmc[3][ki] = (1. - delta) * q->dift[ki] + delta * dstar;
      q->ghat[ki] = (1. - caseA) * q->ghat[ki];                             /* :47 */
    }
  }
}

/* kppmix.  mckpp_physics_verticalmixing_kppmix_mod.F90:25-126 */
static void kppmix(const orc_const *c, orc_col *q, int ntime, int km, int kmp1,
                   const double *dVsq, double ustar, double Bo, double Bosol,
                   const double *Ritop, double *hbl, int *kbl)
{
  double bfsfc, caseA, stable, dkm1[4];
  for (int ki = 0; ki <= km; ki++) {                      /* :65-69 */
    q->difm[ki] = 0.0;
    q->difs[ki] = 0.0;
    q->dift[ki] = 0.0;
  }
  if (c->LRI) rimix(c, q, km, kmp1);                      /* :72-74 */
  if (c->LDD) ddmix(c, q, km);                            /* :77-79 */
  q->difm[kmp1] = q->difm[km];                            /* :82-84 */
  q->difs[kmp1] = q->difs[km];
  q->dift[kmp1] = q->dift[km];
  if (c->LKPP) {                                          /* :87 */
    bldepth(c, q, ntime, km, kmp1, dVsq, Ritop, ustar, Bo, Bosol, hbl, &bfsfc, &stable, &caseA, kbl);
    blmix(c, q, km, ustar, bfsfc, *hbl, stable, caseA, *kbl, dkm1);
    enhance(c, q, km, dkm1, *hbl, *kbl, caseA);
    for (int ki = 1; ki <= km; ki++) {                    /* :103-111 */
      if (ki < *kbl) {
        q->difm[ki] = q->blmc[1][ki];
        q->difs[ki] = q->blmc[2][ki];
        q->dift[ki] = q->blmc[3][ki];
      } else {
        q->ghat[ki] = 0.;
      }
    }
  }
}

/* ------------------------------------------------------------------------
 * vmix.  mckpp_physics_verticalmixing_mod.F90:14-161
 * ---------------------------------------------------------------------- */
static void vmix(const orc_const *c, orc_col *q, int ntime, double *hmixn, int *kmixn)
{
  const int nz = q->nz, nzp1 = q->nzp1;
  const double epsilon = 0.1;
  const double *zm = c->zm;
  double alpha = 1., beta = 1., exppr = 0.0, sigma0 = 0, sigma = 0;         /* :47-51 */
  orc_abk80(0.0, q->X[1][1], -zm[1], &alpha, &beta, &exppr, &sigma0, &sigma); /* :52 */
  q->rhoh2o = 1000. + sigma0;
  orc_abk80(c->sice, q->X[1][1], -zm[1], &alpha, &beta, &exppr, &sigma0, &sigma); /* :54 */
  double rhob = 1000. + sigma0;
  for (int k = 1; k <= nzp1; k++) {                       /* :59-68 */
    orc_abk80(q->X[2][k] + q->Sref, q->X[1][k], -zm[k], &alpha, &beta, &exppr, &sigma0, &sigma);
    q->rho[k] = 1000. + sigma0;
    q->cp[k] = orc_cpsw(q->X[2][k] + q->Sref, q->X[1][k], -zm[k]);
    q->talpha[k] = alpha;
    q->sbeta[k] = beta;
    q->buoy[k] = -c->grav * sigma0 / 1000.;
  }
  q->rho[0] = q->rho[1];                                  /* :70-73 */
  q->cp[0] = q->cp[1];
  q->talpha[0] = q->talpha[1];
  q->sbeta[0] = q->sbeta[1];
  ntflux(c, q, ntime);                                    /* :78 */
  q->wU[1][0] = -q->sflux[1] / q->rho[0];                 /* :81-82 */
  q->wU[2][0] = -q->sflux[2] / q->rho[0];
  double tau = sqrt(q->sflux[1] * q->sflux[1] + q->sflux[2] * q->sflux[2]) + 1.e-16; /* :83 */
  double ustar = sqrt(tau / q->rho[0]);                   /* :85 */
  q->wX[1][0] = -q->sflux[4] / q->rho[0] / q->cp[0];      /* :88 */
  q->wX[2][0] = q->Ssurf * q->sflux[6] / q->rhoh2o + (q->Ssurf - c->sice) * q->sflux[5] / rhob; /* :91-93 */
  double B0 = -c->grav * (q->talpha[0] * q->wX[1][0] - q->sbeta[0] * q->wX[2][0]); /* :96-97 */
  q->wX[3][0] = -B0;                                      /* :98 */
  double B0sol = c->grav * q->talpha[0] * q->sflux[3] / (q->rho[0] * q->cp[0]); /* :99-100 */
  for (int n = 1; n <= nz; n++) {                         /* :103-108 */
    q->alphaDT[n] = 0.5 * (q->talpha[n] + q->talpha[n + 1]) * (q->X[1][n] - q->X[1][n + 1]);
    q->betaDS[n] = 0.5 * (q->sbeta[n] + q->sbeta[n + 1]) * (q->X[2][n] - q->X[2][n + 1]);
  }
  const double *U1 = q->U[1], *U2 = q->U[2], *buoy = q->buoy;
  for (int n = 1; n <= nz; n++) {                         /* :111-137 */
    double zref = epsilon * zm[n];
    double wz = fmax2(zm[1], zref);
    q->uref = U1[1] * wz / zref;
    q->vref = U2[1] * wz / zref;
    double bref = buoy[1] * wz / zref;
    for (int kl = 1; kl <= nz; kl++) {
      if (zref >= zm[kl]) break;                          /* :119 */
      wz = fmin2(zm[kl] - zm[kl + 1], zm[kl] - zref);
      double del = 0.5 * wz / (zm[kl] - zm[kl + 1]);
      q->uref = q->uref - wz * (U1[kl] + del * (U1[kl + 1] - U1[kl])) / zref;
      q->vref = q->vref - wz * (U2[kl] + del * (U2[kl + 1] - U2[kl])) / zref;
      bref = bref - wz * (buoy[kl] + del * (buoy[kl + 1] - buoy[kl])) / zref;
    }
    q->Ritop[n] = (zref - zm[n]) * (bref - buoy[n]);      /* :130 */
    q->dbloc[n] = buoy[n] - buoy[n + 1];                  /* :133 */
    q->dVsq[n] = (q->uref - U1[n]) * (q->uref - U1[n]) + (q->vref - U2[n]) * (q->vref - U2[n]); /* :134 */
    q->Shsq[n] = (U1[n] - U1[n + 1]) * (U1[n] - U1[n + 1]) +
                 (U2[n] - U2[n + 1]) * (U2[n] - U2[n + 1]);                /* :135-136 */
  }
  kppmix(c, q, ntime, nz, nzp1, q->dVsq, ustar, B0, B0sol, q->Ritop, hmixn, kmixn); /* :139 */
  const double dlimit = 0.00001, vlimit = 0.0001;         /* :151-152 */
  for (int k = nz; k <= nzp1; k++) {                      /* :154-158 */
    q->difm[k] = vlimit;
    q->difs[k] = dlimit;
    q->dift[k] = dlimit;
  }
  q->ghat[nz] = 0.0;                                      /* :159 */
}

/* ------------------------------------------------------------------------
 * rhsmod (prescribed advection).  mckpp_physics_solvers.F90:176-335
 * ---------------------------------------------------------------------- */
static void rhsmod(const orc_const *c, orc_col *q, int jsclr, int mode, double A,
                   double dto, int km, double dm, int nzi, double *rhs)
{
  const double *hm = c->hm;
  double fact = 0, delta;
  if (mode <= 0) return;                                  /* :207 */
  double Am = A;                                          /* :221 */
#define FACT(n) do { if (jsclr == 1) fact = dto * Am / (q->rho[n] * q->cp[n]); \
                     if (jsclr == 2) fact = dto * Am * 0.033; } while (0)
  if (mode == 1) {                                        /* :223-227 */
    FACT(1);
    rhs[1] = rhs[1] + fact / hm[1];
  } else if (mode == 2) {                                 /* :229-239 */
    delta = 0.0;
    for (int n = 1; n <= km - 1; n++) delta = delta + hm[n];
    for (int n = 1; n <= km - 1; n++) { FACT(n); rhs[n] = rhs[n] + fact / delta; }
  } else if (mode == 3) {                                 /* :241-251 */
    delta = 0.0;
    for (int n = 1; n <= nzi; n++) delta = delta + hm[n];
    for (int n = 1; n <= nzi; n++) { FACT(n); rhs[n] = rhs[n] + fact / delta; }
  } else if (mode == 4) {                                 /* :253-267 */
    int nzend = nzi - 1, n1 = 0;
    do { n1 = n1 + 1; } while (c->zm[n1] >= -100.);
    delta = 0.0;
    for (int n = n1; n <= nzend; n++) delta = delta + hm[n];
    for (int n = n1; n <= nzend; n++) { FACT(n); rhs[n] = rhs[n] + fact / delta; }
  } else if (mode == 5) {                                 /* :269-273 */
    FACT(nzi);
    rhs[nzi] = rhs[nzi] + fact / hm[nzi];
  } else {
    int n1, n2 = 0;
    double depth, dmax;
    if (mode == 6) {                                      /* :291-304 */
      n1 = 1;
      depth = hm[1];
      dmax = dm - 0.5 * (hm[km] + hm[km - 1]);
      delta = 0.0;
      for (int n = n1; n <= nzi; n++) {
        n2 = n;
        delta = delta + hm[n];
        depth = depth + hm[n + 1];
        if (depth >= dmax) break;
      }
    } else if (mode == 7) {                               /* :306-318 */
      n1 = km - 1;
      depth = dm - 0.5 * hm[km];
      dmax = 100.;
      delta = 0.0;
      for (int n = n1; n <= nzi; n++) {
        n2 = n;
        delta = delta + hm[n];
        depth = depth + hm[n + 1];
        if (depth >= dmax) break;
      }
    } else {
      return;                                             /* :320-324 would abort */
    }
    for (int n = n1; n <= n2; n++) { FACT(n); rhs[n] = rhs[n] + fact / delta; } /* :327-331 */
  }
#undef FACT
}

/* ------------------------------------------------------------------------
 * ocnint.  mckpp_physics_ocnint_mod.F90:19-221
 * ---------------------------------------------------------------------- */
static void ocnint(const orc_const *c, orc_col *q, int kmixe, double *const *Uo, double *const *Xo)
{
  const int NZ = q->nz, NZP1 = q->nzp1;
  const double dto = c->dto;
  const double *hm = c->hm;
  double *cu = q->cu, *cc = q->cc, *cl = q->cl, *rhs = q->rhs, *diff = q->diff, *gcap = q->gcap;
  double ftemp = q->f;                                    /* :43 */
  int i, npd;
  for (int k = 0; k <= NZP1; k++) diff[k] = q->difm[k];   /* :45-47 */
  orc_tridcof(c, diff, NZ, cu, cc, cl);                   /* :48 */
  rhs[1] = Uo[1][1] + dto * (ftemp * .5 * (Uo[2][1] + q->U[2][1]) - q->wU[1][0] / hm[1]); /* :51-52 */
  for (i = 2; i <= NZ - 1; i++)                           /* :53-55 */
    rhs[i] = Uo[1][i] + dto * ftemp * .5 * (Uo[2][i] + q->U[2][i]);
  i = NZ;                                                 /* :56-58 */
  rhs[i] = Uo[1][i] + dto * ftemp * .5 * (Uo[2][i] + q->U[2][i]) + c->tri1[i] * q->difm[i] * Uo[1][i + 1];
  if (tridmat_mode(c, cu, cc, cl, rhs, Uo[1], NZ, q->U[1], q->gam)) q->status |= ORC_ST_ZERO_PIVOT; /* :59 */
  rhs[1] = Uo[2][1] - dto * (ftemp * .5 * (Uo[1][1] + q->U[1][1]) + q->wU[2][0] / hm[1]); /* :62-63 */
  for (i = 2; i <= NZ - 1; i++)                           /* :64-66 */
    rhs[i] = Uo[2][i] - dto * ftemp * .5 * (Uo[1][i] + q->U[1][i]);
  i = NZ;                                                 /* :67-69 */
  rhs[i] = Uo[2][i] - dto * ftemp * .5 * (Uo[1][i] + q->U[1][i]) + c->tri1[i] * q->difm[i] * Uo[2][i + 1];
  npd = 1;                                                /* :70 */
  if (tridmat_mode(c, cu, cc, cl, rhs, Uo[2], NZ, q->U[2], q->gam)) q->status |= ORC_ST_ZERO_PIVOT; /* :71 */

  double ghatflux = q->wX[1][0];                          /* :82-83 */
  double sturflux = q->wX[1][0];
  diff[0] = q->dift[0];                                   /* :84 */
  q->ntflx[1][0] = q->wXNT[1][0];
  for (int k = 1; k <= NZP1; k++) {                       /* :86-90 */
    diff[k] = q->dift[k];
    gcap[k] = q->ghat[k];
    q->ntflx[1][k] = q->wXNT[1][k];
  }
  orc_tridcof(c, diff, NZ, cu, cc, cl);                   /* :91 */
  tridrhs(c, npd, hm, Xo[1], q->ntflx[1], diff, gcap, sturflux, ghatflux, dto, NZ, rhs); /* :93-94 */
  if (c->L_RELAX_SST && !c->L_FCORR_WITHZ && !c->L_FCORR) {                 /* :97-114 */
    if (q->relax_sst > 1.e-10) {
      if (!c->L_RELAX_CALCONLY)
        rhs[1] = rhs[1] + dto * q->relax_sst * (q->SST0 - Xo[1][1]) * c->dm[kmixe] / hm[1];
      q->fcorr = q->relax_sst * (q->SST0 - Xo[1][1]) * c->dm[kmixe] * q->rho[1] * q->cp[1];
    } else {
      q->fcorr = 0.0;
    }
  }
  if (c->L_FCORR && !c->L_RELAX_SST && !c->L_FCORR_WITHZ)                   /* :121-125 */
    rhs[1] = rhs[1] + dto * q->fcorr_twod / (q->rho[1] * q->cp[1] * hm[1]);
  for (int k = 1; k <= NZP1; k++) q->tinc_fcorr[k] = 0.;                    /* :133 */
  if (c->L_FCORR_WITHZ && !c->L_FCORR)                                      /* :134-139 */
    for (int k = 1; k <= NZP1; k++)
      q->tinc_fcorr[k] = dto * q->fcorr_withz[k] / (q->rho[k] * q->cp[k]);
  if (c->L_RELAX_OCNT)                                                      /* :144-152 */
    for (int k = 1; k <= NZP1; k++)
      q->tinc_fcorr[k] = q->tinc_fcorr[k] + dto * q->relax_ocnT * (q->ocnT_clim[k] - Xo[1][k]);
  for (int k = 1; k <= NZP1; k++) {                                         /* :153-160 */
    rhs[k] = rhs[k] + q->tinc_fcorr[k];
    q->ocnTcorr[k] = q->tinc_fcorr[k] * q->rho[k] * q->cp[k] / dto;
  }
  if (tridmat_mode(c, cu, cc, cl, rhs, Xo[1], NZ, q->X[1], q->gam)) q->status |= ORC_ST_ZERO_PIVOT; /* :162 */

  for (int k = 0; k <= NZP1; k++) diff[k] = q->difs[k];   /* :165-167 */
  orc_tridcof(c, diff, NZ, cu, cc, cl);                   /* :168 */
  for (int n = 2; n <= 2; n++) {                          /* :169, NSCLR = 2 */
    for (int k = 0; k <= NZP1; k++) q->ntflx[n][k] = q->wXNT[n][k];         /* :170-172 */
    ghatflux = q->wX[n][0];
    sturflux = q->wX[n][0];
    tridrhs(c, npd, hm, Xo[n], q->ntflx[n], diff, gcap, sturflux, ghatflux, dto, NZ, rhs); /* :175-176 */
    for (int imode = 1; imode <= q->nmodeadv[2]; imode++) {                 /* :179-184 */
      int adv_mode = q->modeadv[imode][2];
      double adv_mag = q->advection[imode][2];
      rhsmod(c, q, 2, adv_mode, adv_mag, dto, kmixe, c->dm[kmixe], NZ, rhs);
    }
    if (n == 2) {                                                           /* :187-215 */
      for (int k = 1; k <= NZP1; k++) q->sinc_fcorr[k] = 0.;
      if (c->L_SFCORR_WITHZ && !c->L_SFCORR)
        for (int k = 1; k <= NZP1; k++) q->sinc_fcorr[k] = dto * q->sfcorr_withz[k];
      if (c->L_RELAX_SAL)
        for (int k = 1; k <= NZP1; k++)
          q->sinc_fcorr[k] = q->sinc_fcorr[k] + dto * q->relax_sal * (q->sal_clim[k] - Xo[n][k]);
      for (int k = 1; k <= NZP1; k++) {
        rhs[k] = rhs[k] + q->sinc_fcorr[k];
        q->scorr[k] = q->sinc_fcorr[k] / dto;
      }
    }
    if (tridmat_mode(c, cu, cc, cl, rhs, Xo[n], NZ, q->X[n], q->gam)) q->status |= ORC_ST_ZERO_PIVOT; /* :218 */
  }
}

/* ------------------------------------------------------------------------
 * ocnstep.  mckpp_physics_ocnstep_mod.F90:43-357
 * ---------------------------------------------------------------------- */
static void relax_profiles(orc_col *q, double lambda)
{
  for (int k = 1; k <= q->nzp1; k++) {                    /* :123-132 / :142-151 */
    for (int l = 1; l <= 2; l++) {
      q->U[l][k] = lambda * q->Ux[l][k] + (1 - lambda) * q->U[l][k];
      q->Ux[l][k] = q->U[l][k];
    }
    for (int l = 1; l <= 2; l++) {
      q->X[l][k] = lambda * q->Xx[l][k] + (1 - lambda) * q->X[l][k];
      q->Xx[l][k] = q->X[l][k];
    }
  }
}

static void ocnstep(const orc_const *c, orc_col *q, int ntime)
{
  const int NZ = q->nz, NZP1 = q->nzp1;
  const int comp_iter_max = 10;                           /* :71 */
  const double rmsd_threshold[5] = {0, 1, 1, 1, 1};       /* :77 */
  const double lambda = 0.5;                              /* :78 */
  double hmixe = 0, hmixn = 0, tol;
  int kmixe = 0, kmixn = 0, iter, iconv;
  double rmsd[5];
  for (int k = 1; k <= NZP1; k++) {                       /* :82-83 */
    q->Uo[1][k] = q->U[1][k]; q->Uo[2][k] = q->U[2][k];
    q->Xo[1][k] = q->X[1][k]; q->Xo[2][k] = q->X[2][k];
  }
  q->comp_flag = 1;                                       /* :84-87 */
  q->reset_flag = 0;
  q->dampu_flag = 0;
  q->dampv_flag = 0;
  q->npasses = 0;
  while (q->comp_flag && q->reset_flag <= comp_iter_max) { /* :89 */
    if (q->old < 0 || q->old > 1) { q->old = q->newi; q->status |= ORC_ST_DODGY_OLDNEW; } /* :93-97 */
    if (q->newi < 0 || q->newi > 1) { q->newi = q->old; q->status |= ORC_ST_DODGY_OLDNEW; } /* :98-102 */
    for (int k = 1; k <= NZP1; k++) {                     /* :91-112 */
      for (int l = 1; l <= 2; l++) {
        q->U[l][k] = 2. * q->Us[l][q->newi][k] - q->Us[l][q->old][k];
        q->Ux[l][k] = q->U[l][k];
      }
      for (int l = 1; l <= 2; l++) {
        q->X[l][k] = 2. * q->Xs[l][q->newi][k] - q->Xs[l][q->old][k];
        q->Xx[l][k] = q->X[l][k];
      }
    }
    iter = 0;                                             /* :116-117 */
    iconv = 0;
    for (iter = 0; iter <= 2; iter++) {                   /* :122-135 */
      relax_profiles(q, lambda);
      vmix(c, q, ntime, &hmixe, &kmixe);
      ocnint(c, q, kmixe, q->Uo, q->Xo);
      q->npasses++;
    }
    /* Fortran DO leaves iter = 3 here */
    if (c->LKPP) {                                        /* :140 */
      for (;;) {                                          /* label 45 */
        relax_profiles(q, lambda);                        /* :142-151 */
        vmix(c, q, ntime, &hmixn, &kmixn);                /* :152 */
        ocnint(c, q, kmixn, q->Uo, q->Xo);                /* :153 */
        q->npasses++;
        iter = iter + 1;                                  /* :154 */
        tol = c->hmixtolfrac * c->hm[kmixn];              /* :157 */
        if (kmixn == NZP1) tol = c->hmixtolfrac * c->hm[NZ]; /* :158 */
        if (fabs(hmixn - hmixe) > tol) iconv = 0;         /* :159-169 */
        else iconv = iconv + 1;
        if (iconv < 3) {                                  /* :170-183 */
          if (iter < c->itermax) {
            hmixe = hmixn;
            kmixe = kmixn;
            continue;
          } else {
            if (hmixn > hmixe) {
              hmixe = hmixn;
              kmixe = kmixn;
              continue;
            }
          }
        }
        if (iter > (c->itermax + 1)) q->status |= ORC_ST_LONG_ITER; /* :184-191 */
        break;
      }
    }
    q->comp_flag = 0;                                     /* :200 */
    for (int k = 1; k <= NZ; k++) {                       /* :201-207 */
      if (fabs(q->U[1][k]) >= 10 || fabs(q->U[2][k]) >= 10 ||
          fabs(q->X[1][k] - q->X[1][k + 1]) >= 10) {
        q->comp_flag = 1;
        q->f = q->f * 1.01;
      }
    }
    if (!q->comp_flag) {                                  /* :208-227 */
      rmsd[1] = rmsd[2] = rmsd[3] = rmsd[4] = 0.;
      for (int k = 1; k <= NZP1; k++) {
        rmsd[1] = rmsd[1] + (q->U[1][k] - q->Uo[1][k]) * (q->U[1][k] - q->Uo[1][k]) * c->hm[k] / c->dm[NZ];
        rmsd[2] = rmsd[2] + (q->U[2][k] - q->Uo[2][k]) * (q->U[2][k] - q->Uo[2][k]) * c->hm[k] / c->dm[NZ];
        rmsd[3] = rmsd[3] + (q->X[1][k] - q->Xo[1][k]) * (q->X[1][k] - q->Xo[1][k]) * c->hm[k] / c->dm[NZ];
        rmsd[4] = rmsd[4] + (q->X[2][k] - q->Xo[2][k]) * (q->X[2][k] - q->Xo[2][k]) * c->hm[k] / c->dm[NZ];
      }
      for (int k = 1; k <= 4; k++) {
        rmsd[k] = sqrt(rmsd[k]);
        if (rmsd[k] >= rmsd_threshold[k]) {
          q->comp_flag = 1;
          q->f = q->f * 1.01;
        }
      }
    }
    if (q->comp_flag) q->status |= ORC_ST_RETRIED;
    q->reset_flag = q->reset_flag + 1;                    /* :228 */
    if (q->reset_flag > comp_iter_max) q->status |= ORC_ST_FAILED; /* :229-236 */
  }
  for (int k = 1; k <= NZ; k++) {                         /* :242-256 */
    double deltaz = 0.5 * (c->hm[k] + c->hm[k + 1]);
    for (int n = 1; n <= 2; n++)
      q->wX[n][k] = -q->difs[k] * ((q->X[n][k] - q->X[n][k + 1]) / deltaz - q->ghat[k] * q->wX[n][0]);
    if (c->LDD)
      q->wX[1][k] = -q->dift[k] * ((q->X[1][k] - q->X[1][k + 1]) / deltaz - q->ghat[k] * q->wX[1][0]);
    q->wX[3][k] = c->grav * (q->talpha[k] * q->wX[1][k] - q->sbeta[k] * q->wX[2][k]);
    for (int n = 1; n <= 2; n++)
      q->wU[n][k] = -q->difm[k] * (q->U[n][k] - q->U[n][k + 1]) / deltaz;
  }
  /* :258-276 energetics: locals only, never stored - omitted */
  q->hmix = hmixn;                                        /* :305-314 */
  q->kmix = (double)kmixn;
  q->uref = q->U[1][1];
  q->vref = q->U[2][1];
  q->Tref = q->X[1][1];
  if (c->L_SSref) q->Ssurf = q->SSref;
  else q->Ssurf = q->X[2][1] + q->Sref;
  if (c->L_DAMP_CURR) {                                   /* :317-340 */
    double dampU[3] = {0, 0., 0.};
    for (int k = 1; k <= NZP1; k++) {
      for (int l = 1; l <= 2; l++) {
        double a = 0.99 * fabs(q->U[l][k]);
        double b = (q->U[l][k] * q->U[l][k]) / ((double)c->dt_uvdamp * (86400. / c->dto));
        double Ui = fmin2(a, b);
        if (b < a) dampU[l] = dampU[l] + 1.0 / (double)NZP1;
        q->U[l][k] = q->U[l][k] - fsign(Ui, q->U[l][k]);
      }
    }
    q->dampu_flag = dampU[1];
    q->dampv_flag = dampU[2];
  }
  q->old = q->newi;                                       /* :343-353 */
  q->newi = 1 - q->old;
  q->hmixd[q->newi] = q->hmix;
  for (int k = 1; k <= NZP1; k++) {
    for (int l = 1; l <= 2; l++) q->Us[l][q->newi][k] = q->U[l][k];
    for (int l = 1; l <= 2; l++) q->Xs[l][q->newi][k] = q->X[l][k];
  }
}

/* ------------------------------------------------------------------------
 * check_profile.  mckpp_physics_overrides.F90:42-125
 * ---------------------------------------------------------------------- */
static void check_profile(const orc_const *c, orc_col *q)
{
  const int NZP1 = q->nzp1;
  if (q->comp_flag && c->clim_present) {                  /* :57-71 */
    for (int k = 1; k <= NZP1; k++) {
      q->X[1][k] = q->ocnT_clim[k];
      q->X[2][k] = q->sal_clim[k];
      q->U[1][k] = q->U_init[1][k];
      q->U[2][k] = q->U_init[2][k];
    }
    q->reset_flag = 999;
  } else if (q->comp_flag) {                              /* :72-78 */
    for (int k = 1; k <= NZP1; k++) {
      q->U[1][k] = q->U_init[1][k];
      q->U[2][k] = q->U_init[2][k];
    }
    q->reset_flag = 999;
  }
  if (q->l_ocean && c->L_NO_FREEZE) {                     /* :85-94 */
    for (int z = 1; z <= NZP1; z++) {
      if (q->X[1][z] < -1.8) {
        q->tinc_fcorr[z] = q->tinc_fcorr[z] + (-1.8 - q->X[1][z]);
        q->X[1][z] = -1.8;
        q->freeze_flag = q->freeze_flag + 1.0 / (double)NZP1;
      }
    }
  }
  if (q->l_ocean && c->L_NO_ISOTHERM) {                   /* :102-120 */
    double dtdz_total = 0., dz_total = 0.;
    for (int j = 2; j <= c->iso_bot; j++) {
      double dz = c->zm[j] - c->zm[j - 1];
      dtdz_total = dtdz_total + fabs((q->X[1][j] - q->X[1][j - 1])) * dz;
      dz_total = dz_total + dz;
    }
    dtdz_total = dtdz_total / dz_total;
    if (fabs(dtdz_total) < c->iso_thresh) {
      for (int k = 1; k <= NZP1; k++) {
        q->X[1][k] = q->ocnT_clim[k];
        q->X[2][k] = q->sal_clim[k];
      }
      q->reset_flag = (-1.) * q->reset_flag;
    }
  } else {
    q->reset_flag = 0;                                    /* :121-123 */
  }
}

/* ------------------------------------------------------------------------
 * grid / geometry helpers (host-side setup the tests share)
 * ---------------------------------------------------------------------- */
/* uniform grid: mckpp_initialize_geography_mod.F90:57-74 (l_stretchgrid=.F.) */
void orc_make_grid_uniform(int nz, double dmax, double *zm, double *hm, double *dm)
{
  double hsum = 0.0;
  for (int i = 1; i <= nz; i++) {
    hm[i] = dmax / (double)nz;
    zm[i] = 0.0 - (hsum + 0.5 * hm[i]);
    hsum = hsum + hm[i];
    dm[i] = hsum;
  }
  dm[0] = 0.0;
  hm[nz + 1] = 1.e-10;
  zm[nz + 1] = -dmax;
}

/* tri factors: mckpp_initialize_ocean.F90:30-43.  c->tri0/tri1 indexed 0..nz */
void orc_make_tri(const orc_const *c)
{
  const int NZ = c->nz;
  c->tri1[0] = c->dto / c->hm[1];
  c->tri0[0] = 0.0;
  c->tri0[1] = 0.0;
  c->tri1[1] = c->dto / c->hm[1] / (c->zm[1] - c->zm[2]);
  for (int k = 2; k <= NZ; k++) {
    c->tri1[k] = c->dto / c->hm[k] / (c->zm[k] - c->zm[k + 1]);
    c->tri0[k] = c->dto / c->hm[k] / (c->zm[k - 1] - c->zm[k]);
  }
}

/* Coriolis parameter: mckpp_initialize_geography_mod.F90:78-88, twopi = 8*atan(1) */
double orc_coriolis(double dlat)
{
  double twopi = 8 * atan(1.);
  if (fabs(dlat) < 2.5)
    return 2. * (twopi / 86164.) * sin(2.5 * twopi / 360.) * fsign(1., dlat);
  return 2. * (twopi / 86164.) * sin(dlat * twopi / 360.);
}

/* ------------------------------------------------------------------------
 * batch <-> column transfer.  Mirrors the persistence contract of
 * mckpp_types_transfer.F90:15-193 (3d -> 1d) and :199-327 (1d -> 3d).
 * ---------------------------------------------------------------------- */
orc_batch *orc_batch_new(long ncol, int ld)
{
  orc_batch *b = (orc_batch *)calloc(1, sizeof(orc_batch));
  b->ncol = ncol;
  b->ld = ld;
  return b;
}

void orc_batch_free(orc_batch *b) { free(b); }

int orc_batch_set(orc_batch *b, const char *name, void *ptr)
{
#define F(x) if (!strcmp(name, #x)) { b->x = ptr; return 0; }
  F(U) F(V) F(T) F(S) F(Us0) F(Us1) F(Vs0) F(Vs1) F(Ts0) F(Ts1) F(Ss0) F(Ss1)
  F(U_init) F(V_init) F(f) F(Ssurf) F(Sref) F(SSref) F(ocdepth) F(sflux) F(hmixd)
  F(hmix) F(kmix) F(uref) F(vref) F(Tref) F(reset_flag) F(dampu_flag) F(dampv_flag)
  F(freeze_flag) F(fcorr) F(old) F(newi) F(jerlov) F(l_initflag) F(l_ocean) F(status) F(npasses)
  F(swfrac) F(swdk_opt) F(rho) F(cp) F(buoy) F(talpha) F(sbeta) F(difm) F(difs) F(dift)
  F(ghat) F(wU1) F(wU2) F(wX1) F(wX2) F(wX3) F(wXNT1) F(Rig) F(dbloc) F(Shsq)
  F(tinc_fcorr) F(sinc_fcorr) F(ocnTcorr) F(scorr) F(relax_sst) F(SST0) F(fcorr_twod)
  F(relax_sal) F(relax_ocnT) F(fcorr_withz) F(sfcorr_withz) F(ocnT_clim) F(sal_clim)
  F(nmodeadv) F(modeadv) F(advection)
#undef F
  return -1;
}

static void ld_arr(const orc_batch *b, const double *src, long col, double *dst, int lo, int hi)
{
  if (!src) { for (int k = lo; k <= hi; k++) dst[k] = 0.0; return; }
  const double *p = src + col * (long)b->ld;
  for (int k = lo; k <= hi; k++) dst[k] = p[k];
}

static void st_arr(const orc_batch *b, double *dstb, long col, const double *src, int lo, int hi)
{
  if (!dstb) return;
  double *p = dstb + col * (long)b->ld;
  for (int k = lo; k <= hi; k++) p[k] = src[k];
}

#define LDS(field, def) (b->field ? b->field[col] : (def))

static void gather(const orc_const *c, const orc_batch *b, long col, orc_col *q)
{
  const int nz = c->nz, nzp1 = nz + 1;
  ld_arr(b, b->U, col, q->U[1], 1, nzp1); ld_arr(b, b->V, col, q->U[2], 1, nzp1);
  ld_arr(b, b->T, col, q->X[1], 1, nzp1); ld_arr(b, b->S, col, q->X[2], 1, nzp1);
  ld_arr(b, b->Us0, col, q->Us[1][0], 1, nzp1); ld_arr(b, b->Us1, col, q->Us[1][1], 1, nzp1);
  ld_arr(b, b->Vs0, col, q->Us[2][0], 1, nzp1); ld_arr(b, b->Vs1, col, q->Us[2][1], 1, nzp1);
  ld_arr(b, b->Ts0, col, q->Xs[1][0], 1, nzp1); ld_arr(b, b->Ts1, col, q->Xs[1][1], 1, nzp1);
  ld_arr(b, b->Ss0, col, q->Xs[2][0], 1, nzp1); ld_arr(b, b->Ss1, col, q->Xs[2][1], 1, nzp1);
  ld_arr(b, b->U_init, col, q->U_init[1], 1, nzp1); ld_arr(b, b->V_init, col, q->U_init[2], 1, nzp1);
  ld_arr(b, b->rho, col, q->rho, 0, nzp1); ld_arr(b, b->cp, col, q->cp, 0, nzp1);
  ld_arr(b, b->buoy, col, q->buoy, 1, nzp1);
  ld_arr(b, b->difm, col, q->difm, 0, nzp1); ld_arr(b, b->difs, col, q->difs, 0, nzp1);
  ld_arr(b, b->dift, col, q->dift, 0, nzp1); ld_arr(b, b->ghat, col, q->ghat, 1, nzp1);
  ld_arr(b, b->wU1, col, q->wU[1], 0, nz); ld_arr(b, b->wU2, col, q->wU[2], 0, nz);
  ld_arr(b, b->wX1, col, q->wX[1], 0, nz); ld_arr(b, b->wX2, col, q->wX[2], 0, nz);
  ld_arr(b, b->wX3, col, q->wX[3], 0, nz);
  ld_arr(b, b->wXNT1, col, q->wXNT[1], 0, nz);
  for (int k = 0; k <= nzp1; k++) q->wXNT[2][k] = 0.0;    /* zeroed at init, never written (fluxes_mod.F90:26) */
  ld_arr(b, b->Rig, col, q->Rig, 1, nz); ld_arr(b, b->dbloc, col, q->dbloc, 1, nz);
  ld_arr(b, b->Shsq, col, q->Shsq, 1, nz);
  ld_arr(b, b->swfrac, col, q->swfrac, 1, nzp1); ld_arr(b, b->swdk_opt, col, q->swdk_opt, 0, nz);
  ld_arr(b, b->tinc_fcorr, col, q->tinc_fcorr, 1, nzp1); ld_arr(b, b->sinc_fcorr, col, q->sinc_fcorr, 1, nzp1);
  ld_arr(b, b->ocnTcorr, col, q->ocnTcorr, 1, nzp1); ld_arr(b, b->scorr, col, q->scorr, 1, nzp1);
  ld_arr(b, b->fcorr_withz, col, q->fcorr_withz, 1, nzp1); ld_arr(b, b->sfcorr_withz, col, q->sfcorr_withz, 1, nzp1);
  ld_arr(b, b->ocnT_clim, col, q->ocnT_clim, 1, nzp1); ld_arr(b, b->sal_clim, col, q->sal_clim, 1, nzp1);
  for (int k = 0; k <= nzp1; k++) { q->talpha[k] = 0.0; q->sbeta[k] = 0.0; } /* not transferred */
  for (int i = 1; i <= 6; i++) q->sflux[i] = b->sflux ? b->sflux[col * 6 + (i - 1)] : 0.0;
  q->hmixd[0] = b->hmixd ? b->hmixd[col * 2] : 0.0;
  q->hmixd[1] = b->hmixd ? b->hmixd[col * 2 + 1] : 0.0;
  q->f = LDS(f, 0.0); q->Ssurf = LDS(Ssurf, 0.0); q->Sref = LDS(Sref, 0.0); q->SSref = LDS(SSref, 0.0);
  q->ocdepth = LDS(ocdepth, -10000.0);
  q->hmix = LDS(hmix, 0.0); q->kmix = LDS(kmix, 0.0); q->uref = LDS(uref, 0.0);
  q->vref = LDS(vref, 0.0); q->Tref = LDS(Tref, 0.0);
  q->reset_flag = LDS(reset_flag, 0.0); q->dampu_flag = LDS(dampu_flag, 0.0);
  q->dampv_flag = LDS(dampv_flag, 0.0); q->freeze_flag = LDS(freeze_flag, 0.0);
  q->fcorr = LDS(fcorr, 0.0);
  q->relax_sst = LDS(relax_sst, 0.0); q->SST0 = LDS(SST0, 0.0); q->fcorr_twod = LDS(fcorr_twod, 0.0);
  q->relax_sal = LDS(relax_sal, 0.0); q->relax_ocnT = LDS(relax_ocnT, 0.0);
  q->old = b->old ? b->old[col] : 0;
  q->newi = b->newi ? b->newi[col] : 1;
  q->jerlov = b->jerlov ? b->jerlov[col] : 3;
  q->l_initflag = b->l_initflag ? b->l_initflag[col] : 0;
  q->l_ocean = b->l_ocean ? b->l_ocean[col] : 1;
  q->status = 0;
  q->npasses = 0;
  q->rhoh2o = 0.0;
  for (int i = 1; i <= 2; i++) {
    q->nmodeadv[i] = b->nmodeadv ? b->nmodeadv[col * 2 + (i - 1)] : 0;
    for (int j = 1; j <= ORC_MAXMODEADV; j++) {
      q->modeadv[j][i] = b->modeadv ? b->modeadv[(col * 2 + (i - 1)) * ORC_MAXMODEADV + (j - 1)] : 0;
      q->advection[j][i] = b->advection ? b->advection[(col * 2 + (i - 1)) * ORC_MAXMODEADV + (j - 1)] : 0.0;
    }
  }
}

#define STS(field, val) do { if (b->field) b->field[col] = (val); } while (0)

static void scatter(const orc_const *c, orc_batch *b, long col, const orc_col *q)
{
  const int nz = c->nz, nzp1 = nz + 1;
  st_arr(b, b->U, col, q->U[1], 1, nzp1); st_arr(b, b->V, col, q->U[2], 1, nzp1);
  st_arr(b, b->T, col, q->X[1], 1, nzp1); st_arr(b, b->S, col, q->X[2], 1, nzp1);
  st_arr(b, b->Us0, col, q->Us[1][0], 1, nzp1); st_arr(b, b->Us1, col, q->Us[1][1], 1, nzp1);
  st_arr(b, b->Vs0, col, q->Us[2][0], 1, nzp1); st_arr(b, b->Vs1, col, q->Us[2][1], 1, nzp1);
  st_arr(b, b->Ts0, col, q->Xs[1][0], 1, nzp1); st_arr(b, b->Ts1, col, q->Xs[1][1], 1, nzp1);
  st_arr(b, b->Ss0, col, q->Xs[2][0], 1, nzp1); st_arr(b, b->Ss1, col, q->Xs[2][1], 1, nzp1);
  st_arr(b, b->rho, col, q->rho, 0, nzp1); st_arr(b, b->cp, col, q->cp, 0, nzp1);
  st_arr(b, b->buoy, col, q->buoy, 1, nzp1);
  st_arr(b, b->talpha, col, q->talpha, 0, nzp1); st_arr(b, b->sbeta, col, q->sbeta, 0, nzp1);
  st_arr(b, b->difm, col, q->difm, 0, nzp1); st_arr(b, b->difs, col, q->difs, 0, nzp1);
  st_arr(b, b->dift, col, q->dift, 0, nzp1); st_arr(b, b->ghat, col, q->ghat, 1, nz);
  st_arr(b, b->wU1, col, q->wU[1], 0, nz); st_arr(b, b->wU2, col, q->wU[2], 0, nz);
  st_arr(b, b->wX1, col, q->wX[1], 0, nz); st_arr(b, b->wX2, col, q->wX[2], 0, nz);
  st_arr(b, b->wX3, col, q->wX[3], 0, nz); st_arr(b, b->wXNT1, col, q->wXNT[1], 0, nz);
  st_arr(b, b->Rig, col, q->Rig, 1, nz); st_arr(b, b->dbloc, col, q->dbloc, 1, nz);
  st_arr(b, b->Shsq, col, q->Shsq, 1, nz);
  st_arr(b, b->swfrac, col, q->swfrac, 1, nzp1); st_arr(b, b->swdk_opt, col, q->swdk_opt, 0, nz);
  st_arr(b, b->tinc_fcorr, col, q->tinc_fcorr, 1, nzp1); st_arr(b, b->sinc_fcorr, col, q->sinc_fcorr, 1, nzp1);
  st_arr(b, b->ocnTcorr, col, q->ocnTcorr, 1, nzp1); st_arr(b, b->scorr, col, q->scorr, 1, nzp1);
  if (b->hmixd) { b->hmixd[col * 2] = q->hmixd[0]; b->hmixd[col * 2 + 1] = q->hmixd[1]; }
  /* f is NOT written back (types_transfer.F90:199-327): the 1.01 perturbation is per step */
  STS(Ssurf, q->Ssurf); STS(hmix, q->hmix); STS(kmix, q->kmix); STS(uref, q->uref);
  STS(vref, q->vref); STS(Tref, q->Tref); STS(reset_flag, q->reset_flag);
  STS(dampu_flag, q->dampu_flag); STS(dampv_flag, q->dampv_flag); STS(freeze_flag, q->freeze_flag);
  STS(fcorr, q->fcorr);
  STS(old, q->old); STS(newi, q->newi); STS(l_initflag, q->l_initflag);
  STS(status, q->status); STS(npasses, q->npasses);
}

static int pick_threads(int nthreads)
{
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  return nthreads;
#else
  (void)nthreads;
  return 1;
#endif
}

/* mckpp_initialize_ocean_model, per-column part.  mckpp_initialize_ocean.F90:48-107 */
void orc_init_ocean(const orc_const *c, orc_batch *b, int ntime, int nthreads)
{
  const int NZ = c->nz, NZP1 = NZ + 1;
  int nt = pick_threads(nthreads);
  (void)nt;
#pragma omp parallel num_threads(nt)
  {
    orc_col *q = col_new(NZ);
#pragma omp for schedule(dynamic, 8)
    for (long col = 0; col < b->ncol; col++) {
      double hmix0;
      int kmix0;
      gather(c, b, col, q);
      q->l_initflag = 1;                                  /* :59 */
      vmix(c, q, ntime, &hmix0, &kmix0);
      q->l_initflag = 0;                                  /* :61 */
      q->hmix = hmix0;
      q->kmix = (double)kmix0;
      q->Tref = q->X[1][1];
      for (int k = 1; k <= NZ; k++) {                     /* :66-81 */
        double deltaz = 0.5 * (c->hm[k] + c->hm[k + 1]);
        for (int n = 1; n <= 2; n++)
          q->wX[n][k] = -q->difs[k] * ((q->X[n][k] - q->X[n][k + 1]) / deltaz - q->ghat[k] * q->wX[n][0]);
        if (c->LDD)
          q->wX[1][k] = -q->dift[k] * ((q->X[1][k] - q->X[1][k + 1]) / deltaz - q->ghat[k] * q->wX[1][0]);
        q->wX[3][k] = c->grav * (q->talpha[k] * q->wX[1][k] - q->sbeta[k] * q->wX[2][k]);
        for (int n = 1; n <= 2; n++)
          q->wU[n][k] = -q->difm[k] * (q->U[n][k] - q->U[n][k + 1]) / deltaz;
      }
      q->old = 0;                                         /* :86-100 */
      q->newi = 1;
      q->hmixd[0] = q->hmix;
      q->hmixd[1] = q->hmix;
      for (int k = 1; k <= NZP1; k++)
        for (int l = 1; l <= 2; l++) {
          q->Us[l][0][k] = q->U[l][k]; q->Us[l][1][k] = q->U[l][k];
          q->Xs[l][0][k] = q->X[l][k]; q->Xs[l][1][k] = q->X[l][k];
        }
      scatter(c, b, col, q);
    }
    col_free(q);
  }
}

/* mckpp_physics_driver.  mckpp_physics_driver_mod.F90:15-73 */
void orc_physics_driver(const orc_const *c, orc_batch *b, int ntime, int nthreads)
{
  int nt = pick_threads(nthreads);
  (void)nt;
#pragma omp parallel num_threads(nt)
  {
    orc_col *q = col_new(c->nz);
#pragma omp for schedule(dynamic, 8)
    for (long col = 0; col < b->ncol; col++) {
      gather(c, b, col, q);                               /* :50 */
      ocnstep(c, q, ntime);                               /* :54 */
      check_profile(c, q);                                /* :55 */
      scatter(c, b, col, q);                              /* :59 */
    }
    col_free(q);
  }
}

/* One vmix + ocnint pass per column, no relaxation/iteration ("kppmix +
 * tridiag only", BASELINE config 2).  Uo/Xo are taken equal to the current
 * U/X; hmix/kmix receive the diagnosed boundary-layer depth. */
void orc_vmix_batch(const orc_const *c, orc_batch *b, int ntime, int nthreads)
{
  int nt = pick_threads(nthreads);
  (void)nt;
#pragma omp parallel num_threads(nt)
  {
    orc_col *q = col_new(c->nz);
#pragma omp for schedule(dynamic, 8)
    for (long col = 0; col < b->ncol; col++) {
      double h;
      int km;
      gather(c, b, col, q);
      for (int k = 1; k <= q->nzp1; k++)
        for (int l = 1; l <= 2; l++) { q->Uo[l][k] = q->U[l][k]; q->Xo[l][k] = q->X[l][k]; }
      vmix(c, q, ntime, &h, &km);
      ocnint(c, q, km, q->Uo, q->Xo);
      q->hmix = h;
      q->kmix = (double)km;
      scatter(c, b, col, q);
    }
    col_free(q);
  }
}

/* mckpp_physics_verticalmixing alone (verticalmixing_mod.F90:14-161) on every column: hmix/kmix <- hmixn/kmixn */
void orc_vmix_only_batch(const orc_const *c, orc_batch *b, int ntime, int nthreads)
{
  int nt = pick_threads(nthreads);
  (void)nt;
#pragma omp parallel num_threads(nt)
  {
    orc_col *q = col_new(c->nz);
#pragma omp for schedule(dynamic, 8)
    for (long col = 0; col < b->ncol; col++) {
      double h;
      int km;
      gather(c, b, col, q);
      vmix(c, q, ntime, &h, &km);
      q->hmix = h;
      q->kmix = (double)km;
      scatter(c, b, col, q);
    }
    col_free(q);
  }
}

/* mckpp_fluxes.  mckpp_fluxes_mod.F90:35-89 (forcing arrays given, l_fluxdata semantics left to the caller) */
/* mckpp_physics_overrides_bottomtemp, overrides.F90:12-24 (called by the driver after the column
 * loop when L_VARY_BOTTOM_TEMP, physics_driver_mod.F90:67-71) */
void orc_bottomtemp(const orc_const *c, orc_batch *b, const double *bottom_temp)
{
  const int nzp1 = c->nz + 1;
  for (long col = 0; col < b->ncol; col++) {
    const long o = col * b->ld + nzp1;
    const double tinc = bottom_temp[col] - b->T[o];                      /* :16 */
    if (b->tinc_fcorr) b->tinc_fcorr[o] = tinc;
    if (b->ocnTcorr) b->ocnTcorr[o] = tinc * b->rho[o] * b->cp[o] / c->dto;   /* :17-19 */
    b->T[o] = bottom_temp[col];                                          /* :20 */
  }
}

void orc_fluxes(const orc_const *c, orc_batch *b, int ntime, const double *taux_in, const double *tauy,
                const double *swf, const double *lwf, const double *lhf, const double *shf, const double *rain,
                const double *snow, int l_rest, double flsn, double el)
{
  orc_col *q = col_new(c->nz);
  for (long col = 0; col < b->ncol; col++) {
    if (b->l_ocean && !b->l_ocean[col]) continue;            /* :55 */
    double taux = taux_in[col];
    if ((taux == 0.0) && (tauy[col] == 0.0)) taux = 1.e-10;  /* :57-58 */
    double *sf = b->sflux + col * 6;
    if (!l_rest) {                                           /* :60-69 */
      sf[0] = taux;
      sf[1] = tauy[col];
      sf[2] = swf[col];
      sf[3] = lwf[col] + lhf[col] + shf[col] - snow[col] * flsn;
      sf[4] = 1e-10;
      sf[5] = rain[col] + snow[col] + (lhf[col] / el);
    } else {                                                 /* :70-77 */
      sf[0] = 1.e-10; sf[1] = 0.00; sf[2] = 300.00; sf[3] = -300.00; sf[4] = 0.00; sf[5] = 0.00;
    }
    gather(c, b, col, q);                                    /* :80-82 */
    ntflux(c, q, ntime);
    scatter(c, b, col, q);
  }
  col_free(q);
}
