// mckpp_device.h - shared between the HIP kernels and the host runtime.
//
// Device-resident layout (see DESIGN.md "Data layout in HBM"):
//   level arrays   double[ncol][ld]     ld = 64*LPL, level-fastest, one row
//                                        per water column (512 B at nz<=61)
//     profiles (U,V,T,S,Us*,Xs*,U_init)  element j  <-> reference level k=j+1
//     diagnostics (rho,cp,dif*,wU,wX...) element k  <-> reference index k
//   column record  double cs[ncol][MCKPP_CS]   (scalars in/out, in place)
//                  int    ci[ncol][MCKPP_CI]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MCKPP_CS 24
#define MCKPP_CI 8
#define MCKPP_XS 8
enum { XS_RELAX_SST = 0, XS_SST0, XS_FCORR_TWOD, XS_RELAX_SAL, XS_RELAX_OCNT };

// cs[] slots
enum {
  CS_F = 0, CS_SSURF, CS_SREF, CS_SSREF, CS_OCDEPTH,
  CS_SFLUX1, CS_SFLUX2, CS_SFLUX3, CS_SFLUX4, CS_SFLUX5, CS_SFLUX6,
  CS_HMIXD0, CS_HMIXD1, CS_HMIX, CS_KMIX, CS_UREF, CS_VREF, CS_TREF,
  CS_RESET, CS_DAMPU, CS_DAMPV, CS_FREEZE, CS_FCORR, CS_SPARE1
};
// ci[] slots
enum { CI_OLD = 0, CI_NEW, CI_JERLOV, CI_INITFLAG, CI_STATUS, CI_NPASS, CI_LOCEAN, CI_IPT };

// STEP: ocnstep + check_profile; INIT: initial vmix + seeds (initialize_ocean.F90); PASS: one vmix + ocnint
// (configs[1]); VMIX: mckpp_physics_verticalmixing alone - diagnostics and hmix/kmix only, state untouched
enum { MCKPP_MODE_STEP = 0, MCKPP_MODE_INIT = 1, MCKPP_MODE_PASS = 2, MCKPP_MODE_VMIX = 3 };

// The kernel parameter block, once with plain pointers (host code, by-value kernel arguments) and once with
// pointers typed as global memory: a pointer that a kernel loads from the block in memory is otherwise a generic
// pointer - flat instructions, 64-bit VGPR address arithmetic, LDS waits tied to memory traffic - whereas
// through mckpp_kparams_dev (same layout) every access is a global_load / global_store.
template <class T> using mckpp_ptr_plain = T *;
template <class T> using mckpp_ptr_global = __attribute__((address_space(1))) T *;

template <template <class> class P>
struct mckpp_kparams_t {
  int nz, nzp1, ncol, ld;
  int ntime, itermax, mode, diag;
  int L_SSref, LDD, clim_present;
  int l2pre;      // L2 works from per-layer terms formed once per column (deep reference-level sums; host-chosen)
  double hmixtolfrac, dto, grav, vonk, sice;
  double Vtc;     // bldepth_mod.F90:91, host-evaluated
  double cg;      // blmix_mod.F90:62, host-evaluated (libm pow)
  double dm_nz;   // dm(NZ)
  // constants, device pointers; Fortran-indexed, padded to ldc doubles
  P<const double> zm, hm, tri0, tri1;
  P<const double> swfrac_tab;  // [6][ldc]  swfrac(k), k=1..nzp1, per Jerlov type
  P<const double> swdk_tab;    // [6][ldc]  swdk_opt(k), k=0..nz
  int ldc;
  int LRI;        // rimix (kppmix_mod.F90:72-74); .FALSE.: the interior diffusivities stay zero, Rig is not formed
  int solver_mode;   // 0: tridiagonal sweeps in the reference's order (solvers.F90:112-161); 1: two-ended elimination (opt-in)
  int l3cap;      // > 0: L3 forms the bulk Richardson numbers at most down to this level before the scan asks for the
                  // rest (MCKPP_L3_CAP, tests: the second round of the scan on every pass)
  P<const double> wtab;        // [(NJ+2)][(NI+2)] pairs {wmt, wst}
  // state
  P<double> U, V, T, S;
  P<double> Us[2], Vs[2], Ts[2], Ss[2];
  P<const double> U_init, V_init;
  P<double> cs;
  P<int> ci;
  P<int> qhead;   // column queue head for the persistent cooperative kernel (zeroed per launch)
  int nsteps_launch;   // model steps this launch takes every column through (1: the step-per-launch path)
  int nqueues;         // nsteps_launch > 1: queues = XCDs of the device (qhead[0..nqueues-1]); column c is in queue c mod nqueues
  int xcc_queue[16];   // hardware XCC id -> queue index
  // forced run in one launch (mckpp_hip_run_forced): the flux records and how a step finds its own
  P<const double> series;   // [nrec][8][ncol]: taux, tauy, swf, lwf, lhf, shf, rain, snow; null: the forcing is what cs holds
  int series_rec0, ndtocn, l_rest;
  double flsn, el;
  P<int> qowner;  // [16] per queue: 0 free, else hardware XCC id + 1 of the XCD whose workgroups serve it (zeroed per launch)
  P<int> done;    // [ncol] steps of this launch a column has completed, then [ncol] steps of it that have been started
                  // (zeroed per launch; nsteps_launch > 1 only)
  P<unsigned long long> dbg;   // optional [32] phase-cycle accumulators (diagnostic builds of a run only)
  // optional physics: ext != 0 selects the kernel build that carries it
  int ext, L_RELAX_SST, L_RELAX_CALCONLY, L_FCORR, L_FCORR_WITHZ, L_SFCORR, L_SFCORR_WITHZ;
  int L_RELAX_SAL, L_RELAX_OCNT, L_NO_FREEZE, L_NO_ISOTHERM, L_DAMP_CURR, iso_bot, dt_uvdamp, maxmodeadv;
  double iso_thresh;
  P<const double> dm;      // dm(0:nz)
  P<const double> hsum;    // hsum(n) = hm(1)+...+hm(n), summed in that order (rhsmod's delta)
  P<const double> xs;      // [ncol][MCKPP_XS]: relax_sst, SST0, fcorr_twod, relax_sal, relax_ocnT
  P<const double> fcorr_withz, sfcorr_withz, ocnT_clim, sal_clim;   // profile rows
  P<double> tinc_fcorr, sinc_fcorr, ocnTcorr, scorr;                // diagnostic rows
  P<const int> adv_i;      // [ncol][1+maxmodeadv]: nmodeadv(2), modeadv(:,2)
  P<const double> adv_d;   // [ncol][maxmodeadv]:   advection(:,2)
  // diagnostics (all or none)
  P<double> rho, cp, buoy, talpha, sbeta, difm, difs, dift, ghat;
  P<double> wU1, wU2, wX1, wX2, wX3, wXNT1, Rig, dbloc, Shsq;
  // k_column_ps: the iterate's scratch rows, one block per (workgroup, slot) (mckpp_ps_scratch_doubles)
  P<double> scratch;
  size_t scratch_doubles;
  // Stragglers (k_column_ps, M0 / G_late): columns on their way to itermax are left alone in their workgroup.
  P<int> sync;        // [0] stragglers the device holds right now (zeroed per launch)
  int solo_after;     // a column past this many passes of a try is one (default 12; MCKPP_SOLO_AFTER)
  int view_kmax;      // most slots of a view (0: as many as the workgroup's waves other than the manager's hold items for; MCKPP_VIEW_KMAX)
  int solo_limit;     // workgroups leave their other slots empty for a straggler while the device holds at most this many
                      // (0: never - MCKPP_SOLO=0; default: workgroups / 32, at least 2; MCKPP_SOLO_LIMIT)
};
using mckpp_kparams = mckpp_kparams_t<mckpp_ptr_plain>;
using mckpp_kparams_dev = mckpp_kparams_t<mckpp_ptr_global>;
static_assert(sizeof(mckpp_kparams) == sizeof(mckpp_kparams_dev), "same layout");

// what the last cooperative-kernel launch looked like (for the residency check of the tests)
struct mckpp_launch_info { int nblocks, threads, max_blocks_per_cu; size_t lds_bytes; };

// launchers
// The column step: packed, stateless-lane cooperative kernel (mckpp_kernels_ps.hip), persistent grid; level
// phases loop over (slot, level) items.  `dp` is a device copy of `p` (every field but ntime is read from it;
// ntime is passed by value)
hipError_t mckpp_launch_column_kernel_ps(const mckpp_kparams &p, const mckpp_kparams *dp, int num_cu,
                                         hipStream_t stream, mckpp_launch_info *info);
size_t mckpp_ps_scratch_doubles(int nzp1, int variant, int num_cu);   // what p.scratch must hold (variant: 0 default physics, 1 optional, 2 optional with double diffusion)
hipError_t mckpp_launch_xcc_probe(unsigned *mask, hipStream_t stream);   // OR of 1 << XCC id over a 4096-workgroup grid
hipError_t mckpp_launch_eos_batch(int64_t n, const double *s, const double *t, const double *p,
                                  double *alpha, double *beta, double *sig0, double *cp,
                                  hipStream_t stream);
hipError_t mckpp_launch_div_batch(int64_t n, const double *num, const double *den, double *q, hipStream_t stream);
hipError_t mckpp_launch_exp_batch(int64_t n, const double *x, double *y, hipStream_t stream);
hipError_t mckpp_launch_bottomtemp(const mckpp_kparams &p, const double *bt, hipStream_t stream);
hipError_t mckpp_launch_fluxes(const mckpp_kparams &p, int ntime, const double *f8, int l_rest, double flsn,
                               double el, hipStream_t stream);
// record slots a download wants as (npts) slabs: doubles of cs[], ints of ci[]
struct mckpp_pack_list { int nd, ni; int dslot[MCKPP_CS]; int islot[MCKPP_CI]; };
hipError_t mckpp_launch_unpack_sflux(const double *slabs, const int *ipt, double *cs, int64_t ncol, int64_t npts, hipStream_t stream);
hipError_t mckpp_launch_pack_records(const double *cs, const int *ci, const int *ipt, int64_t ncol, int64_t npts,
                                     const mckpp_pack_list &l, double *dout, int *iout, hipStream_t stream);
hipError_t mckpp_launch_out_sample(const double *src, int src_ld, int src_off, const double *cs, int add_sref,
                                   int64_t ncol, int nlev, int ld_out, double *sum, double *mn, double *mx, int first,
                                   double *inst, hipStream_t stream);
hipError_t mckpp_launch_window_mean(const double *sum, double *out, size_t n, double count, hipStream_t stream);
// layout kernels: Fortran (npts-fastest) <-> device rows
hipError_t mckpp_launch_gather_rows(const double *src3d, int64_t npts, int nlev, int lev_off,
                                    const int *ipt, int64_t ncol, double *dst, int ld, int dst_off,
                                    hipStream_t stream);
hipError_t mckpp_launch_scatter_rows(const double *src, int ld, int src_off, const int *ipt,
                                     int64_t ncol, double *dst3d, int64_t npts, int nlev, int lev_off,
                                     hipStream_t stream);
