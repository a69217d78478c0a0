/*
 * mckpp_hip.h - C-ABI of the MI355X (gfx950) column-physics library.
 *
 * This is the drop-in boundary for MC-KPP's per-column physics step.  The
 * reference (aosprey/mckpp-f90) has no FFI seam of its own: the seam is the
 * Fortran call surface
 *     CALL mckpp_physics_driver()          src/mckpp_physics_driver_mod.F90:15
 *       -> mckpp_physics_ocnstep(kpp_1d_fields, kpp_const_fields)
 *                                          src/mckpp_physics_ocnstep_mod.F90:43
 *       -> mckpp_physics_overrides_check_profile   src/mckpp_physics_overrides.F90:42
 *     CALL mckpp_initialize_ocean_model()  src/mckpp_initialize_ocean.F90:18
 * operating on the module globals kpp_3d_fields / kpp_const_fields
 * (src/mckpp_data_fields.F90:348-349).  The entry points below are what an
 * iso_c_binding module in that code base binds (see INTEGRATION.md and
 * mckpp_f90_amd/fortran/mckpp_hip_binding.F90): plain pointers into the
 * Fortran-owned ALLOCATABLE components, plain sizes, int return codes.
 *
 * Conventions
 *  - every array pointer addresses a Fortran array in its native layout
 *    (column index `ipt` fastest), fp64 / int32 / 4-byte LOGICAL;
 *  - the library never frees or keeps host pointers beyond the call;
 *  - column state is device-resident between mckpp_hip_upload and
 *    mckpp_hip_download;
 *  - all functions return 0 on success, <0 on error (text from
 *    mckpp_hip_last_error()).  Numerical conditions never abort: they are
 *    reported per column through the status bitmask.
 *  - not re-entrant; call from one host thread per handle.
 */
#ifndef MCKPP_HIP_H
#define MCKPP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCKPP_NI 890 /* wmt/wst first extent  = NI+2 (src/mckpp_physics_lookup_mod.F90:20) */
#define MCKPP_NJ 48  /* wmt/wst second extent = NJ+2 (:21) */

/* per-column status bits (replace the reference's stderr warnings / STOP) */
#define MCKPP_ST_ZERO_PIVOT   1  /* src/mckpp_physics_solvers.F90:140-151 (reference STOPs) */
#define MCKPP_ST_LONG_ITER    2  /* src/mckpp_physics_ocnstep_mod.F90:184-191 */
#define MCKPP_ST_RETRIED      4  /* :200-227 instability trap fired at least once */
#define MCKPP_ST_FAILED       8  /* :229-236 ten retries exhausted */
#define MCKPP_ST_DODGY_OLDNEW 16 /* :93-102 */

/* kpp_const_type, hot-path subset (src/mckpp_data_fields.F90:187-346) plus
 * the mckpp_parameters integers the path reads (src/mckpp_parameters.F90). */
typedef struct mckpp_const_c {
  int32_t nz;        /* layers; nzp1 = nz+1 grid points               */
  int32_t nztmax;    /* >= nzp1 (sizes difm/difs/dift/wU/wX/wXNT/ghat) */
  int32_t nsflxs;    /* first flux extent of sflux (9)                */
  int32_t njdt;      /* sflux(npts,nsflxs,5,0:njdt)                   */
  int32_t itermax;   /* 200                                           */
  int32_t LKPP, LRI, LDD, L_SSref;
  int32_t L_RELAX_SST, L_RELAX_CALCONLY, L_FCORR, L_FCORR_WITHZ;
  int32_t L_SFCORR, L_SFCORR_WITHZ, L_RELAX_SAL, L_RELAX_OCNT;
  int32_t L_NO_FREEZE, L_NO_ISOTHERM, L_DAMP_CURR;
  int32_t clim_present; /* ocnT_file/='none' .and. sal_file/='none'   */
  int32_t iso_bot, dt_uvdamp;
  int32_t maxmodeadv; /* second extent of modeadv/advection (6)       */
  int32_t L_ADVECT;   /* prescribed advection on (src/mckpp_initialize_advection_mod.F90:65: nmodeadv=0 otherwise) */
  double hmixtolfrac, dto, grav, vonk, sice, iso_thresh;
  const double *zm;  /* zm(nzp1)                                      */
  const double *hm;  /* hm(nzp1)                                      */
  const double *dm;  /* dm(0:nz)                                      */
  const double *tri; /* tri(0:nztmax,0:1,ngrid): only (:,:,1) is read  */
  const double *wmt; /* wmt(0:891,0:49)                               */
  const double *wst; /* wst(0:891,0:49)                               */
} mckpp_const_c;

/* Pointers into kpp_3d_type components (src/mckpp_data_fields.F90:8-101,
 * extents :353-447).  A NULL pointer means "field not exchanged". */
typedef struct mckpp_state_ptrs_c {
  int64_t npts;
  /* prognostic and saved profiles */
  double *U;       /* U(npts,nzp1,nvel)            */
  double *X;       /* X(npts,nzp1,nsclr)           */
  double *Us;      /* Us(npts,nzp1,nvel,0:1)       */
  double *Xs;      /* Xs(npts,nzp1,nsclr,0:1)      */
  double *U_init;  /* U_init(npts,nzp1,nvel)       */
  double *hmixd;   /* hmixd(npts,0:1)              */
  /* per-column scalars (npts) */
  double *f, *ocdepth, *Sref, *SSref, *Ssurf;
  double *hmix, *kmix, *Tref, *uref, *vref;
  double *reset_flag, *dampu_flag, *dampv_flag, *freeze_flag;
  double *sflux;   /* sflux(npts,nsflxs,5,0:njdt); (:,1:6,5,0) is read */
  int32_t *old, *new_, *jerlov;
  int32_t *l_ocean, *l_initflag, *run_physics; /* 4-byte LOGICAL       */
  /* diagnostics of the last vmix/ocnint pass of the step */
  double *rho, *cp;           /* (npts,0:nzp1tmax)                     */
  double *buoy;               /* (npts,nzp1tmax)                       */
  double *difm, *difs, *dift; /* (npts,0:nztmax)                       */
  double *wU;                 /* (npts,0:nztmax,nvp1)                  */
  double *wX;                 /* (npts,0:nztmax,nsp1)                  */
  double *wXNT;               /* (npts,0:nztmax,nsclr)                 */
  double *ghat;               /* (npts,nztmax)                         */
  double *Rig, *Shsq;         /* (npts,nzp1)                           */
  double *dbloc;              /* (npts,nz)                             */
  double *swfrac;             /* (npts,nzp1)                           */
  double *swdk_opt;           /* (npts,0:nz)                           */
  /* optional forcing corrections / relaxation (src/mckpp_physics_ocnint_mod.F90:97-215,
   * src/mckpp_physics_overrides.F90:42-125); only read when the matching switch is on */
  double *relax_sst, *SST0, *fcorr_twod, *relax_sal, *relax_ocnT; /* (npts)        */
  double *fcorr;              /* (npts) out: diagnosed surface flux correction    */
  double *fcorr_withz, *sfcorr_withz, *ocnT_clim, *sal_clim;      /* (npts,nzp1)  */
  double *tinc_fcorr, *sinc_fcorr, *ocnTcorr, *scorr;             /* (npts,nzp1) out */
  int32_t *nmodeadv;          /* (npts,2)            - column 2 (salinity) is used */
  int32_t *modeadv;           /* (npts,maxmodeadv,2)                               */
  double *advection;          /* (npts,maxmodeadv,2)                               */
} mckpp_state_ptrs_c;

/* field_mask bits for mckpp_hip_download */
#define MCKPP_F_PROFILES 1u /* U, X                                         */
#define MCKPP_F_SAVED    2u /* Us, Xs, hmixd, old, new                      */
#define MCKPP_F_SCALARS  4u /* hmix,kmix,Tref,uref,vref,Ssurf,*_flag        */
#define MCKPP_F_DIAG     8u /* rho..swdk_opt (needs diagnostics enabled)    */
#define MCKPP_F_RESTART  (MCKPP_F_PROFILES | MCKPP_F_SAVED | MCKPP_F_SCALARS)
#define MCKPP_F_ALL      0xFu

typedef struct mckpp_hip_ctx *mckpp_hip_handle;

const char *mckpp_hip_last_error(void);

/* Twelve hex digits identifying the kernel sources this library was built from
 * (measurement bookkeeping: profiles/ records it next to counter values). */
const char *mckpp_hip_build_id(void);
/* The compiler (hipcc --version line) the kernels were built and gate-checked with: the register budget and the
 * hand-scheduled LDS reads of the column kernel are checked on the code THAT compiler generated (tools/check_build.py,
 * run by the library's Makefile); a build made with CHECK=0 carries "-unchecked" in its build id. */
const char *mckpp_hip_build_compiler(void);

/* Number of visible gfx950 devices (<0 on error). */
int mckpp_hip_device_count(void);

/* Create a context on HIP device `device`; copies grid, tri and the lookup
 * tables to the device.  Replaces nothing in the reference: it is the
 * device-side mirror of kpp_const_fields after mckpp_initialize_namelist /
 * mckpp_physics_lookup / the tri() set-up of initialize_ocean.F90:34-43. */
int mckpp_hip_init(const mckpp_const_c *c, int device, mckpp_hip_handle *out);
int mckpp_hip_finalize(mckpp_hip_handle h);

/* Host helpers with no device work (so a caller written in C can build
 * kpp_const_fields): mckpp_physics_lookup (src/mckpp_physics_lookup_mod.F90:11)
 * and the tri() factors (src/mckpp_initialize_ocean.F90:34-43). */
void mckpp_host_lookup(double vonk, double *wmt, double *wst);
void mckpp_host_tri(int32_t nz, int32_t nztmax, double dto, const double *zm,
                    const double *hm, double *tri);

/* 3D -> device: compacts run_physics columns, re-lays column-fastest Fortran
 * arrays into level-fastest padded rows.  Replaces the gather half of
 * mckpp_fields_3dto1d (src/mckpp_types_transfer.F90:15-193), once instead of
 * every step. */
int mckpp_hip_upload(mckpp_hip_handle h, const mckpp_state_ptrs_c *s);

/* New surface forcing: sflux(:,1:6,5,0) as written by mckpp_fluxes
 * (src/mckpp_fluxes_mod.F90:62-69).  `sflux` is the full Fortran array. */
int mckpp_hip_set_forcing(mckpp_hip_handle h, const double *sflux);

/* mckpp_fluxes (src/mckpp_fluxes_mod.F90:35-89) on the device: assembles
 * sflux(:,1:6,5,0) from the eight surface forcing fields (each npts, 3D
 * ordering: taux,tauy,swf,lwf,lhf,shf,rain,snow; kpp_3d_type components of
 * the same names, src/mckpp_data_fields.F90:76-83) for every l_ocean column
 * and refreshes wXNT(:,1) like its mckpp_fluxes_ntflux call (:93-118).
 * l_rest, flsn, el: kpp_const_fields%L_REST, %FLSN, %EL. */
int mckpp_hip_fluxes(mckpp_hip_handle h, int ntime, const double *taux, const double *tauy,
                     const double *swf, const double *lwf, const double *lhf, const double *shf,
                     const double *rain, const double *snow, int l_rest, double flsn, double el);

/* Forced runs without per-step host traffic.  mckpp_hip_set_flux_series keeps `nrec`
 * successive records of the eight surface forcing fields on the device (layout
 * fields[rec][8][npts], field order and meaning as mckpp_hip_fluxes: what
 * kpp_3d_fields%taux..snow hold after the reference's flux reader at each
 * update); record 0 of the array is flux update number `rec0` of the run.
 * mckpp_hip_run_forced then is the reference's time loop without output
 * (src/mckpp_ocean_model_3D.F90:38-58): for nt = nt_first .. nt_first+nsteps-1
 *   IF (MOD(nt-1, ndtocn) == 0) mckpp_fluxes with record (nt-1)/ndtocn
 *   mckpp_physics_driver
 * all on the context's stream; it fails before launching anything if a needed
 * record is not resident.  l_rest, flsn, el as for mckpp_hip_fluxes. */
int mckpp_hip_set_flux_series(mckpp_hip_handle h, int rec0, int nrec, const double *fields);
int mckpp_hip_run_forced(mckpp_hip_handle h, int nt_first, int nsteps, int ndtocn, int l_rest,
                         double flsn, double el);

/* mckpp_physics_overrides_bottomtemp (src/mckpp_physics_overrides.F90:12-24),
 * which mckpp_physics_driver calls after the column loop when
 * kpp_const_fields%L_VARY_BOTTOM_TEMP (src/mckpp_physics_driver_mod.F90:67-71):
 *   tinc_fcorr(ipt,NZP1) = bottom_temp(ipt) - X(ipt,NZP1,1)
 *   ocnTcorr(ipt,NZP1)   = tinc_fcorr(ipt,NZP1)*rho(ipt,NZP1)*cp(ipt,NZP1)/dto
 *   X(ipt,NZP1,1)        = bottom_temp(ipt)
 * on every device-resident column.  `bottom_temp` is kpp_3d_fields%bottom_temp
 * (npts, 3D ordering).  Needs the diagnostics on (rho, cp of the last vmix).
 * Land points are not resident and stay untouched (the reference also rewrites
 * their X(:,NZP1,1), which nothing reads). */
int mckpp_hip_bottomtemp(mckpp_hip_handle h, const double *bottom_temp);

/* Enable/disable writing of the MCKPP_F_DIAG fields by step/init (default on). */
int mckpp_hip_set_diagnostics(mckpp_hip_handle h, int on);

/* Tridiagonal solver mode of the implicit step (an extension: the reference has one solver,
 * tridmat, src/mckpp_physics_solvers.F90:112-161).
 *   0 (default)  tridmat's order of operations - results bit-identical to the CPU restatement of the reference;
 *   1            the same systems eliminated from both ends at once (levels 1..nz/2 downward, nz..nz/2+1 upward,
 *                a 2x2 system in the middle): half the dependent chain, results within rounding of mode 0
 *                (profiles/r04/parity_tolerance.json; the oracle's solver_mode=1 restates
 *                it operation for operation).
 * The environment variable MCKPP_SOLVER_MODE sets the default of new handles.  <0 on an unknown mode. */
int mckpp_hip_set_solver_mode(mckpp_hip_handle h, int mode);
int mckpp_hip_get_solver_mode(mckpp_hip_handle h);

/* mckpp_initialize_ocean_model's per-column part
 * (src/mckpp_initialize_ocean.F90:48-107): initial vmix with l_initflag,
 * hmix/kmix, initial diagnostic fluxes, old/new/Us/Xs/hmixd seeds. */
int mckpp_hip_init_ocean(mckpp_hip_handle h, int ntime);

/* mckpp_physics_driver (src/mckpp_physics_driver_mod.F90:15-73): nsteps calls,
 * step i run with ntime+i.  Asynchronous on the context's stream. */
int mckpp_hip_step(mckpp_hip_handle h, int ntime, int nsteps);

/* One vmix + ocnint pass per column with Uo=U, Xo=X ("kppmix + tridiag
 * only"): mckpp_physics_verticalmixing (src/mckpp_physics_verticalmixing_mod.F90:14)
 * followed by mckpp_physics_ocnint (src/mckpp_physics_ocnint_mod.F90:19). */
int mckpp_hip_vmix_pass(mckpp_hip_handle h, int ntime);

/* mckpp_physics_verticalmixing alone (src/mckpp_physics_verticalmixing_mod.F90:14-161; the reference
 * also calls it from src/mckpp_initialize_ocean.F90:60) on every resident column with the column's own
 * l_initflag: hmix / kmix receive its hmixn / kmixn, uref / vref its scratch values (:115-125), and the
 * MCKPP_F_DIAG fields of a vmix (rho, cp, buoy, Rig, dbloc, Shsq, difm, difs, dift, ghat, wU(0), wX(0),
 * wXNT) are rewritten; profiles, saved time levels and hmixd stay as they are. */
int mckpp_hip_vmix_only(mckpp_hip_handle h, int ntime);

int mckpp_hip_synchronize(mckpp_hip_handle h);

/* device -> 3D: scatter half of mckpp_fields_1dto3d
 * (src/mckpp_types_transfer.F90:199-327) for the fields selected.  The row
 * fields are re-laid on the device into the Fortran order and cross PCIe once
 * each, the transfer of one field running under the layout kernel of the next;
 * the per-column records come as one array.  The arrays `s` points to (and
 * those of mckpp_hip_upload, _update_ancillaries, _window_fetch) are pinned on
 * first use (hipHostRegister) so the transfers run at the bus rate; they stay
 * pinned until the context is finalised - call mckpp_hip_release_host_arrays
 * before freeing them earlier.  MCKPP_HIP_NO_HOST_REGISTER=1 in the environment
 * switches the pinning off. */
int mckpp_hip_download(mckpp_hip_handle h, mckpp_state_ptrs_c *s, uint32_t field_mask);
int mckpp_hip_release_host_arrays(mckpp_hip_handle h);

/* Restart set (reference: XIOS restart context, src/mckpp_xios_io.F90:368-387
 * write list, :436-465 read path, cadence src/mckpp_xios_control.F90:61-83):
 * U,V,T,S,CP,rho,hmix,kmix,Sref,SSref,Ssurf,Tref,old,new,Us,Vs,Ts,Ss,hmixd (and
 * the column map) straight between HBM and a flat binary file.  load needs a
 * context created with the same vertical grid; it replaces any resident state
 * (only after the whole file has been read and checked: a bad file leaves the
 * resident state as it was). */
int mckpp_hip_save_restart(mckpp_hip_handle h, const char *path);
int mckpp_hip_load_restart(mckpp_hip_handle h, const char *path);

/* What the reference's time loop rewrites on the host between steps when the
 * optional physics is on (mckpp_boundary_update, src/mckpp_ocean_model_3D.F90:51-55;
 * the ndtupd* cadences of src/mckpp_boundary_update.F90): relax_sst, SST0,
 * fcorr_twod, relax_sal, relax_ocnT, fcorr_withz, sfcorr_withz, ocnT_clim,
 * sal_clim, nmodeadv/modeadv/advection.  Only those members of `s` are read;
 * the prognostic state stays as it is on the device.  Also what an
 * optional-physics context needs after mckpp_hip_load_restart (the restart set
 * does not carry these inputs; stepping is refused until they are resident).
 * No-op on a default-physics context. */
int mckpp_hip_update_ancillaries(mckpp_hip_handle h, const mckpp_state_ptrs_c *s);

/* Output fields and their temporal operations on the device: what mckpp_xios_output_control sends
 * every step (src/mckpp_xios_io.F90:74-210) and what XIOS then does with it ("instant", "average",
 * "minimum", "maximum", run/iodef.xml:88-157), so that only reduced fields leave the GPU.
 * MCKPP_OUT_* names follow the XIOS field ids; 3-D fields come back as out(npts,nzp1) on the vertical
 * axis the reference sends them on (levels 1..nzp1 for u, v, T, S, B, rho, cp, Rig, dbloc (0 at nzp1),
 * Shsq and the correction increments; interfaces 0..nz for wu..wTnt and for difm/dift/difs, whose
 * shifted copy at :136-148 is dif*(0:nz)), 2-D fields as out(npts).  S is X(:,:,2)+Sref as at :108-111;
 * MCKPP_OUT_S_ANOM is the bare X(:,:,2).  cplwght (a coupling weight, not on this path) is not offered.
 *   window_select: the fields accumulate reduces (default u, v, T, S_ANOM, hmix); resets the window
 *   window_reset / window_accumulate: start of an output window / once after each step
 *   window_fetch: op 0 mean, 1 min, 2 max over the window (selected fields); op 3 instant - the field as
 *     it stands on the device now, any field, no accumulate needed.  Land points keep what `out` held. */
enum {
  MCKPP_OUT_U = 0, MCKPP_OUT_V, MCKPP_OUT_T, MCKPP_OUT_S_ANOM, MCKPP_OUT_HMIX,
  MCKPP_OUT_S, MCKPP_OUT_B, MCKPP_OUT_WU, MCKPP_OUT_WV, MCKPP_OUT_WT, MCKPP_OUT_WS, MCKPP_OUT_WB, MCKPP_OUT_WTNT,
  MCKPP_OUT_DIFM, MCKPP_OUT_DIFT, MCKPP_OUT_DIFS, MCKPP_OUT_RHO, MCKPP_OUT_CP, MCKPP_OUT_SCORR, MCKPP_OUT_RIG,
  MCKPP_OUT_DBLOC, MCKPP_OUT_SHSQ, MCKPP_OUT_TINC_FCORR, MCKPP_OUT_FCORR_Z, MCKPP_OUT_SINC_FCORR,
  MCKPP_OUT_FCORR, MCKPP_OUT_TAUX_IN, MCKPP_OUT_TAUY_IN, MCKPP_OUT_SOLAR_IN, MCKPP_OUT_NSOLAR_IN, MCKPP_OUT_PMINUSE_IN,
  MCKPP_OUT_FREEZE_FLAG, MCKPP_OUT_COMP_FLAG, MCKPP_OUT_DAMPU_FLAG, MCKPP_OUT_DAMPV_FLAG,
  MCKPP_OUT_COUNT
};
int mckpp_hip_window_select(mckpp_hip_handle h, const int32_t *fields, int32_t nfields);
int mckpp_hip_window_reset(mckpp_hip_handle h);
int mckpp_hip_window_accumulate(mckpp_hip_handle h);
int mckpp_hip_window_fetch(mckpp_hip_handle h, int field, int op, double *out);

/* Per-column status words (npts entries in 3D ordering; land = 0), number of
 * columns with a non-zero word, and (optional) vmix+ocnint passes per column
 * of the last step. Any output pointer may be NULL. */
int mckpp_hip_status(mckpp_hip_handle h, int32_t *per_col, int64_t *n_flagged,
                     int32_t *npasses);

/* Timing of the most recent step/init/vmix_pass call on the context's stream, from HIP events recorded around its
 * kernel launches: total ms and the number of model steps (or passes) they covered.  mckpp_hip_step(nt, n > 1) with
 * constant forcing is ONE launch that takes every column through all n steps (a column's step waits for that
 * column's previous step only - no barrier across the device between steps, so a column that runs to itermax delays
 * nothing but itself; same results bit for bit; MCKPP_MULTISTEP=0 restores a launch per step):
 * mckpp_hip_last_launch_count tells how many kernel launches the call made. */
int mckpp_hip_last_kernel_ms(mckpp_hip_handle h, double *ms, int32_t *nlaunch);
int32_t mckpp_hip_last_launch_count(mckpp_hip_handle h);

/* Name of the column kernel this context launches for its grid and switches
 * ("k_column_ps", "k_column_ps<EXT>"); static storage, never NULL. */
const char *mckpp_hip_kernel_name(mckpp_hip_handle h);

/* Residency of this context's most recent column-kernel launch: workgroups per CU the launcher asked
 * for (its geometry is chosen per launch from the column depth and count), how many the runtime says fit
 * (registers, LDS), threads and dynamic LDS bytes per workgroup.  The launcher's choice assumes
 * max_blocks_per_cu >= blocks_per_cu. */
int mckpp_hip_kernel_residency(mckpp_hip_handle h, int32_t *blocks_per_cu, int32_t *max_blocks_per_cu,
                               int32_t *threads_per_block, int64_t *lds_bytes_per_block);

/* Number of device-resident (run_physics) columns. */
int64_t mckpp_hip_ncolumns(mckpp_hip_handle h);

/* Kernel-level batch entry points on caller-provided host arrays (tests). */
int mckpp_hip_eos_batch(mckpp_hip_handle h, int64_t n, const double *s,
                        const double *t, const double *p, double *alpha,
                        double *beta, double *sig0, double *cp);
int mckpp_hip_exp_batch(mckpp_hip_handle h, int64_t n, const double *x, double *y);
/* The kernels' exact-division helpers on n operand pairs: q4[0..n) = div_fast,
 * q4[n..2n) = div_fast_guarded, q4[2n..3n) = div_by_refined, q4[3n..4n) = the
 * compiler's IEEE n/d (csrc/mckpp_colmath.h); for the tests. */
int mckpp_hip_div_batch(mckpp_hip_handle h, int64_t n, const double *num, const double *den, double *q4);

/* ---------------------------------------------------------------------------
 * Several GPUs of one node behind one handle, for a host that is a single
 * process (the reference's program is one OpenMP process,
 * src/mckpp_physics_driver_mod.F90:27-65: one call covers all npts).  The
 * run_physics columns are dealt round-robin to the devices; each shard is an
 * ordinary context (mckpp_hip_multi_ctx) and every per-context entry point
 * above may be applied to it with the full-size Fortran arrays - a shard reads
 * and writes only its own columns.  The multi_* forms below do exactly that
 * for every shard; multi_step returns once all shards are launched.  There is
 * no collective inside a step.  mckpp_hip_multi_gather is the output gather:
 * shards -> root device over the GPU interconnect, relayout there, one
 * device-to-host transfer.
 * --------------------------------------------------------------------------- */
typedef struct mckpp_hip_multi *mckpp_hip_multi_handle;

/* run_physics mask of shard `dev` of `ndev` (host helper, no device work): the
 * j-th run_physics point in ipt order belongs to shard j mod ndev.  Returns the
 * number of columns of the shard, <0 on bad arguments. */
int64_t mckpp_host_shard_mask(int64_t npts, const int32_t *run_physics, int32_t ndev, int32_t dev, int32_t *mask_out);

/* devices == NULL: HIP devices 0 .. ndev-1. */
int mckpp_hip_multi_init(const mckpp_const_c *c, int32_t ndev, const int32_t *devices, mckpp_hip_multi_handle *out);
int mckpp_hip_multi_finalize(mckpp_hip_multi_handle m);
int32_t mckpp_hip_multi_ndev(mckpp_hip_multi_handle m);
mckpp_hip_handle mckpp_hip_multi_ctx(mckpp_hip_multi_handle m, int32_t shard);
int mckpp_hip_multi_upload(mckpp_hip_multi_handle m, const mckpp_state_ptrs_c *s);
int mckpp_hip_multi_set_forcing(mckpp_hip_multi_handle m, const double *sflux);
int mckpp_hip_multi_set_diagnostics(mckpp_hip_multi_handle m, int on);
int mckpp_hip_multi_set_solver_mode(mckpp_hip_multi_handle m, int mode);
int mckpp_hip_multi_update_ancillaries(mckpp_hip_multi_handle m, const mckpp_state_ptrs_c *s);
int mckpp_hip_multi_bottomtemp(mckpp_hip_multi_handle m, const double *bottom_temp);
int mckpp_hip_multi_fluxes(mckpp_hip_multi_handle m, int ntime, const double *taux, const double *tauy,
                           const double *swf, const double *lwf, const double *lhf, const double *shf,
                           const double *rain, const double *snow, int l_rest, double flsn, double el);
int mckpp_hip_multi_init_ocean(mckpp_hip_multi_handle m, int ntime);
int mckpp_hip_multi_step(mckpp_hip_multi_handle m, int ntime, int nsteps);
int mckpp_hip_multi_synchronize(mckpp_hip_multi_handle m);
/* mckpp_hip_download over all shards (scatter half of src/mckpp_types_transfer.F90:199-327): every row
 * field goes through the gather (shards -> shard 0 -> host: one PCIe transfer per field whatever the number of
 * devices), the per-column records of each shard come to the host on their own. */
int mckpp_hip_multi_download(mckpp_hip_multi_handle m, mckpp_state_ptrs_c *s, uint32_t field_mask);
int mckpp_hip_multi_release_host_arrays(mckpp_hip_multi_handle m);
int mckpp_hip_multi_status(mckpp_hip_multi_handle m, int32_t *per_col, int64_t *n_flagged, int32_t *npasses);
int64_t mckpp_hip_multi_ncolumns(mckpp_hip_multi_handle m);
/* field: 0 U, 1 V, 2 T, 3 S -> out(npts,nzp1); 4 hmix -> out(npts); root: shard index that collects.  All
 * shards' peer copies are in flight at once (one stream per shard on the root device). */
int mckpp_hip_multi_gather(mckpp_hip_multi_handle m, int32_t field, int32_t root, double *out);
/* The reference's forced time loop (src/mckpp_ocean_model_3D.F90:38-58) over all shards: every shard keeps its
 * columns of the flux records (mckpp_hip_set_flux_series) and runs mckpp_hip_run_forced on its own stream;
 * returns once all shards are launched. */
int mckpp_hip_multi_set_flux_series(mckpp_hip_multi_handle m, int rec0, int nrec, const double *fields);
int mckpp_hip_multi_run_forced(mckpp_hip_multi_handle m, int nt_first, int nsteps, int ndtocn, int l_rest,
                               double flsn, double el);
/* Output windows (src/mckpp_xios_io.F90:74-210, run/iodef.xml:88-157) over all shards: each shard reduces
 * its own columns, window_fetch gathers the reduced rows like any other field. */
int mckpp_hip_multi_window_select(mckpp_hip_multi_handle m, const int32_t *fields, int32_t nfields);
int mckpp_hip_multi_window_reset(mckpp_hip_multi_handle m);
int mckpp_hip_multi_window_accumulate(mckpp_hip_multi_handle m);
int mckpp_hip_multi_window_fetch(mckpp_hip_multi_handle m, int field, int op, double *out);
/* Restart set (src/mckpp_xios_io.F90:368-465) of all shards: one file per shard, <path>.<shard>of<ndev>.
 * load needs the state uploaded first (it gives the shards their column maps) and refuses files written for
 * another number of shards or another land mask, before anything resident is replaced. */
int mckpp_hip_multi_save_restart(mckpp_hip_multi_handle m, const char *path);
int mckpp_hip_multi_load_restart(mckpp_hip_multi_handle m, const char *path);

#ifdef __cplusplus
}
#endif
#endif
