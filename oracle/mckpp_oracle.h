/*
 * mckpp_oracle.h - CPU restatement of the MC-KPP per-column physics step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (mckpp_f90_amd/,
 * include/) may include, link or call this.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * Plain C, serial per column, written from a reading of the reference's
 * Fortran (aosprey/mckpp-f90, /root/reference/src); each function cites the
 * file:line it follows.  Arithmetic follows the reference's expression order
 * so that, compiled without FMA contraction, it reproduces the bits of the
 * reference built here with amdflang -fdefault-real-8 (verified for the two
 * reference modules that can be built without stand-ins: the equation of
 * state and the z121 smoother - see oracle/Makefile target `ref`).
 *
 * PARITY PINNING STATUS: EOS (abk80, cpsw) and z121 are pinned bit-for-bit to
 * the compiled reference (oracle/_ref) and to the check values in the
 * reference's comments.  The remainder of the path (vmix/kppmix stack,
 * ocnint, solvers, ocnstep, overrides) cannot be built here without a
 * stand-in for the absent netcdf-fortran module, so for those functions the
 * oracle is "parity unpinned": a careful restatement with invariant tests.
 *
 * Array conventions: all level arrays are addressed with the reference's
 * Fortran indices (1..nzp1, or 0..nz / 0..nzp1 for interface arrays); the C
 * buffers are allocated one element longer so the index can be used as is.
 */
#ifndef MCKPP_ORACLE_H
#define MCKPP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NI 890   /* lookup table: zehat values (wscale_mod.F90:19) */
#define ORC_NJ 48    /* lookup table: ustar values (wscale_mod.F90:20) */
#define ORC_TABLE_LEN ((ORC_NI + 2) * (ORC_NJ + 2))
#define ORC_MAXMODEADV 6

/* status bits (the reference prints warnings / STOPs instead) */
#define ORC_ST_ZERO_PIVOT   1   /* solvers.F90:140-151 would have aborted   */
#define ORC_ST_LONG_ITER    2   /* ocnstep_mod.F90:184-191 warning           */
#define ORC_ST_RETRIED      4   /* ocnstep_mod.F90:200-227 trap fired        */
#define ORC_ST_FAILED       8   /* ocnstep_mod.F90:229-236: 10 retries spent */
#define ORC_ST_DODGY_OLDNEW 16  /* ocnstep_mod.F90:93-102                    */

typedef struct {
  int nz;               /* layers; nzp1 = nz + 1 grid points */
  int itermax;          /* initialize_namelist_mod.F90:31 (200) */
  double hmixtolfrac;   /* :32 (0.1) */
  double dto, grav, vonk, sice;
  int LKPP, LRI, LDD, L_SSref;
  int L_RELAX_SST, L_RELAX_CALCONLY, L_FCORR, L_FCORR_WITHZ;
  int L_SFCORR, L_SFCORR_WITHZ, L_RELAX_SAL, L_RELAX_OCNT;
  int L_NO_FREEZE, L_NO_ISOTHERM, L_DAMP_CURR;
  int clim_present;     /* ocnT_file /= 'none' .and. sal_file /= 'none' (overrides.F90:57-58) */
  int iso_bot;
  double iso_thresh;
  int dt_uvdamp;
  int exp_mode;         /* 0: libm exp (faithful); 1: portable exp shared bit-for-bit with the HIP kernels */
  int solver_mode;      /* 0: tridmat in the reference's order (solvers.F90:112-161); 1: the two-ended elimination of
                           orc_tridmat_2e (the library's opt-in solver mode, same operations as the HIP kernels) */
  /* grid, Fortran-indexed: zm[1..nzp1], hm[1..nzp1], dm[0..nz] */
  double *zm, *hm, *dm;
  /* tri(k,0,1) and tri(k,1,1), k = 0..nz (initialize_ocean.F90:34-43) */
  double *tri0, *tri1;
  /* lookup tables, Fortran layout wmt(0:891,0:49): [j*892 + i] */
  double *wmt, *wst;
} orc_const;

/* Batch of columns, level-fastest: a level array holds `ld` doubles per
 * column (ld >= nzp1 + 1) and is addressed arr[col*ld + fortran_index].
 * Pointers that are NULL are treated as absent (zeros on read, no write). */
typedef struct {
  long ncol;
  int ld;
  /* prognostic + saved profiles, index 1..nzp1 */
  double *U, *V, *T, *S;
  double *Us0, *Us1, *Vs0, *Vs1, *Ts0, *Ts1, *Ss0, *Ss1;
  double *U_init, *V_init;
  /* per-column scalars */
  double *f, *Ssurf, *Sref, *SSref, *ocdepth;
  double *sflux;        /* [ncol][6] = sflux(1:6,5,0) */
  double *hmixd;        /* [ncol][2] */
  double *hmix, *kmix, *uref, *vref, *Tref;
  double *reset_flag, *dampu_flag, *dampv_flag, *freeze_flag, *fcorr;
  int *old, *newi, *jerlov, *l_initflag, *l_ocean, *status, *npasses;
  /* persisted per-column optics (index 1..nzp1 / 0..nz) */
  double *swfrac, *swdk_opt;
  /* diagnostics written by the last vmix/ocnint of the step */
  double *rho, *cp;     /* 0..nzp1 */
  double *buoy;         /* 1..nzp1 */
  double *talpha, *sbeta; /* 0..nzp1 (kpp_1d only in the reference) */
  double *difm, *difs, *dift; /* 0..nzp1 */
  double *ghat;         /* 1..nz */
  double *wU1, *wU2;    /* 0..nz */
  double *wX1, *wX2, *wX3; /* 0..nz */
  double *wXNT1;        /* 0..nz */
  double *Rig, *dbloc, *Shsq; /* 1..nz */
  double *tinc_fcorr, *sinc_fcorr, *ocnTcorr, *scorr; /* 1..nzp1 */
  /* optional forcing-correction inputs (all switches default off) */
  double *relax_sst, *SST0, *fcorr_twod, *relax_sal, *relax_ocnT;
  double *fcorr_withz, *sfcorr_withz, *ocnT_clim, *sal_clim; /* 1..nzp1 */
  int *nmodeadv;        /* [ncol][2] */
  int *modeadv;         /* [ncol][2][ORC_MAXMODEADV] */
  double *advection;    /* [ncol][2][ORC_MAXMODEADV] */
} orc_batch;

/* ---- scalar / small functions (per-function fixtures) ---- */
double orc_exp_portable(double x);
double orc_cpsw(double S, double T1, double P0);
void orc_abk80(double S, double T1, double P, double *alpha, double *beta,
               double *kappa, double *sig0, double *sig);
void orc_abk80_batch(int n, const double *s, const double *t, const double *p,
                     double *alpha, double *beta, double *sig0, double *sig);
void orc_cpsw_batch(int n, const double *s, const double *t, const double *p, double *cp);
void orc_z121(int kmp1, double vlo, double vhi, double *V, double *w);
void orc_conv_probe(int n, const double *x, double *p3, double *p4, double *ph, double *pt, double *pq);
void orc_conv_literals(double *out);
void orc_conv_unary(int n, const double *x, double *e, double *s, double *a, double *q);
void orc_conv_swfrac(int n, const double *z, double fact, int jwtype, double *sw, double *sk);
void orc_conv_jerlov(int jwtype, double *out);
void orc_conv_binary(int n, const double *a, const double *b, const double *c, const double *d, double *sg, double *sh,
                     double *se, double *mx, double *mn, double *ax, double *an, double *m3, double *m4, double *x3);
void orc_conv_casts(int n, const double *x, int *ifx, int *itr, int *icl, double *fl);
void orc_lookup(double vonk, double *wmt, double *wst);
/* half_pow_mode 0: x**(1./2.) is a square root (amdflang, the compiler the oracle is pinned to);
 * 1: it is pow(x, 0.5) (how a compiler without that rewrite lowers lookup_mod.F90:60-62) */
void orc_lookup_mode(double vonk, double *wmt, double *wst, int half_pow_mode);
void orc_wscale(const orc_const *c, double sigma, double hbl, double ustar,
                double bfsfc, double *wm, double *ws);
double orc_swfrac(const orc_const *c, double fact, double z, int jwtype);
double orc_swdk(const orc_const *c, double z, int j);
void orc_tridcof(const orc_const *c, const double *diff, int nzi, double *cu,
                 double *cc, double *cl);
int orc_tridmat(const double *cu, const double *cc, const double *cl,
                const double *rhs, const double *yo, int nzi, double *yn,
                double *gam);
int orc_tridmat_2e(const double *cu, const double *cc, const double *cl,
                   const double *rhs, const double *yo, int nzi, double *yn,
                   double *gam);
void orc_make_grid_uniform(int nz, double dmax, double *zm, double *hm, double *dm);
void orc_make_tri(const orc_const *c);
double orc_coriolis(double dlat);

/* ---- batch helpers ---- */
orc_batch *orc_batch_new(long ncol, int ld);
void orc_batch_free(orc_batch *b);
int orc_batch_set(orc_batch *b, const char *name, void *ptr);

/* ---- the path ---- */
/* mckpp_initialize_ocean_model's per-column part (initialize_ocean.F90:48-107) */
void orc_init_ocean(const orc_const *c, orc_batch *b, int ntime, int nthreads);
/* mckpp_physics_driver (physics_driver_mod.F90:15-73): ocnstep + check_profile */
void orc_physics_driver(const orc_const *c, orc_batch *b, int ntime, int nthreads);
/* mckpp_fluxes (fluxes_mod.F90:35-89): sflux assembly + ntflux */
void orc_fluxes(const orc_const *c, orc_batch *b, int ntime, const double *taux, const double *tauy,
                const double *swf, const double *lwf, const double *lhf, const double *shf, const double *rain,
                const double *snow, int l_rest, double flsn, double el);
/* mckpp_physics_overrides_bottomtemp (overrides.F90:12-24) */
void orc_bottomtemp(const orc_const *c, orc_batch *b, const double *bottom_temp);
/* one vmix + ocnint pass on every column (config-2 style kernel-level check) */
void orc_vmix_only_batch(const orc_const *c, orc_batch *b, int ntime, int nthreads);
void orc_vmix_batch(const orc_const *c, orc_batch *b, int ntime, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
