// mckpp_runtime.cpp - host side of the C-ABI declared in include/mckpp_hip.h.
//
// Owns the device-resident column state (level-fastest rows, one per
// run_physics column), the device copies of kpp_const_fields, one HIP stream
// and a pair of events per context.  No CPU fallback exists: every entry
// point that computes does so by launching the gfx950 kernels of
// mckpp_kernels.hip, and fails with an error code if HIP is unavailable.
#include "../../include/mckpp_hip.h"
#include "mckpp_device.h"
#include "mckpp_math.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return -1;
}

#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) return fail("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// The host loops over the columns (compaction of the forcing, scatter of the per-column records) on a few
// threads: at 1e5 columns they are milliseconds of a 3.4 ms step otherwise.  MCKPP_HIP_HOST_THREADS sets the count
// (default: the machine's, at most 8).
template <class F>
void for_columns(int64_t n, F &&body)
{
  static const int want = [] {
    const char *e = getenv("MCKPP_HIP_HOST_THREADS");
    int t = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return t < 1 ? 1 : t > 8 ? 8 : t;
  }();
  const int nt = n < 32768 ? 1 : want;
  if (nt == 1) { body((int64_t)0, n); return; }
  std::vector<std::thread> th;
  const int64_t chunk = (n + nt - 1) / nt;
  for (int t = 1; t < nt; ++t) {
    const int64_t a = t * chunk, b = a + chunk < n ? a + chunk : n;
    if (a < b) th.emplace_back([&body, a, b] { body(a, b); });
  }
  body((int64_t)0, chunk < n ? chunk : n);
  for (auto &x : th) x.join();
}

// profile rows (element j <-> level j+1)
enum { P_U = 0, P_V, P_T, P_S, P_US0, P_US1, P_VS0, P_VS1, P_TS0, P_TS1, P_SS0, P_SS1, P_UINIT, P_VINIT, P_COUNT };
// optional-physics input rows (allocated only when a switch needs them)
enum { E_FCORR_WITHZ = 0, E_SFCORR_WITHZ, E_OCNT_CLIM, E_SAL_CLIM, E_COUNT };
// optional-physics output rows (indexed by k like the diagnostics)
enum { O_TINC = 0, O_SINC, O_OCNTCORR, O_SCORR, O_COUNT };
// diagnostic rows (element k <-> reference index k)
enum { D_RHO = 0, D_CP, D_BUOY, D_TALPHA, D_SBETA, D_DIFM, D_DIFS, D_DIFT, D_GHAT, D_WU1, D_WU2,
       D_WX1, D_WX2, D_WX3, D_WXNT1, D_RIG, D_DBLOC, D_SHSQ, D_COUNT };

const double jer_rfac[6] = {0, 0.58, 0.62, 0.67, 0.77, 0.78};
const double jer_a1[6] = {0, 0.35, 0.6, 1.0, 1.5, 1.4};
const double jer_a2[6] = {0, 23.0, 20.0, 17.0, 14.0, 7.9};

}  // namespace

enum { QBLOCK_INTS = 64 };

struct mckpp_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  int nlaunch = 0;
  int nkernels = 0;   // kernel launches of the last call (a call of several steps may be one launch)
  mckpp_const_c c{};
  int nz = 0, nzp1 = 0, lpl = 1, ld = 64, ldc = 72;
  double Vtc = 0, cg = 0, dm_nz = 0;
  std::vector<double> h_swfrac_tab, h_swdk_tab;
  double *d_zm = nullptr, *d_hm = nullptr, *d_tri0 = nullptr, *d_tri1 = nullptr;
  double *d_swfrac_tab = nullptr, *d_swdk_tab = nullptr;
  double2 *d_wtab = nullptr;
  // state
  int64_t npts = 0, ncol = 0;
  std::vector<int> ipt;
  int *d_ipt = nullptr;
  double *d_prof[P_COUNT] = {};
  double *d_diag[D_COUNT] = {};
  bool ext = false;          // any optional-physics switch on: the context carries their input / output fields
  bool ext_kernel = false;   // ... and needs the kernel build with the N3 code (a T/S climatology for the reset of
                             // failed columns alone - the shipped namelist - does not: the default build reads it)
  double *d_ext_in[E_COUNT] = {};
  double *d_ext_out[O_COUNT] = {};
  double *d_xs = nullptr, *d_adv_d = nullptr, *d_dm = nullptr, *d_hsum = nullptr;
  int *d_adv_i = nullptr;
  // output-window reductions: selected MCKPP_OUT_* fields, three accumulators (sum, min, max) each
  std::vector<int> wsel{0, 1, 2, 3, 4};
  std::vector<double *> d_wacc;   // [wsel.size()], each 3 * (ncol*ld | ncol) doubles
  int window_count = 0;
  double *d_cs = nullptr;
  int *d_ci = nullptr;
  int *d_qhead = nullptr;  // QBLOCK_INTS ints, zeroed before every launch: [0..15] queue heads, [16..31] queue owners, [32] stragglers on the device
  int view_kmax = 0;   // mckpp_kparams_t::view_kmax (MCKPP_VIEW_KMAX)
  int solo_after = 12, solo_limit = 8;   // mckpp_kparams_t::solo_after / solo_limit (MCKPP_SOLO=0, MCKPP_SOLO_AFTER, MCKPP_SOLO_LIMIT)
  int *d_done = nullptr;   // [2][ncol] steps of a multi-step launch each column has completed, and has started (mckpp_kparams_t::done)
  bool multistep = true;   // mckpp_hip_step(nt, n > 1) as one launch (MCKPP_MULTISTEP=0: a launch per step)
  int nqueues = 0;         // XCDs of the device, found by a probe at init: the queues of such a launch
  int xcc_queue[16];       // hardware XCC id -> queue (-1: no workgroup of the probe ran there)
  int l3cap = 0;   // MCKPP_L3_CAP (tests): see mckpp_kparams_t::l3cap
  int solver_mode = 0;   // mckpp_hip_set_solver_mode / MCKPP_SOLVER_MODE
  unsigned long long *d_dbg = nullptr;
  mckpp_kparams *d_params = nullptr;   // device copy of the kernel parameter block
  // its source: two pinned host slots used in turn, so a call never waits for its own upload (a slot is reused
  // only when the copy that read it - two calls back - has completed)
  mckpp_kparams *h_params = nullptr;
  hipEvent_t ev_params[2] = {nullptr, nullptr};
  unsigned params_seq = 0;
  double *d_scratch = nullptr;         // k_column_ps: scratch rows of the iterate, per (workgroup, slot)
  size_t scratch_doubles = 0;
  int num_cu = 256;
  int l2pre = 0;   // the reference-level sums of the deepest level span many layers: form the layer terms once per column
  double *d_series = nullptr;   // [nrec][8][ncol] forcing records (mckpp_hip_set_flux_series)
  int series_rec0 = 0, series_nrec = 0;
  mckpp_launch_info last_launch{};   // geometry of this context's most recent cooperative launch
  double *d_stage = nullptr;
  size_t stage_elems = 0;
  // Row transfers (upload / download): two device staging buffers used in turn and a copy stream, so the PCIe
  // transfer of one field runs while the layout kernel of the next does; the caller's arrays are pinned
  // (hipHostRegister, once per array) so those transfers are asynchronous and run at the bus rate.
  hipStream_t copy_stream = nullptr;
  double *d_xfer[2] = {nullptr, nullptr};
  size_t xfer_elems[2] = {0, 0};
  hipEvent_t ev_lay[2] = {nullptr, nullptr}, ev_copy[2] = {nullptr, nullptr};   // layout kernel done / transfer done, per buffer
  unsigned xfer_seq = 0;
  std::vector<std::pair<const void *, size_t>> pinned;   // caller arrays this context registered
  std::vector<std::pair<const void *, size_t>> unpinnable;   // ... and those it could not (not tried again)
  // pinned host images of the column records and of the forcing staging
  double *h_cs = nullptr, *h_f = nullptr;
  int *h_ci = nullptr;
  size_t h_f_elems = 0;
  hipEvent_t ev_f = nullptr;
  // record slots as (npts) slabs in 3-D order, packed on the device (a context that holds every grid point)
  double *d_pack = nullptr, *h_pack = nullptr;   // packed record slabs of a download: device block, pinned host block
  int diag = 1;
  // optional-physics contexts: the relaxation / correction / advection inputs come with upload (or
  // update_ancillaries); load_restart does not carry them, so stepping is refused until they are there
  bool ext_inputs_resident = false;
};

extern "C" {

const char *mckpp_hip_last_error(void) { return g_err.c_str(); }

#ifndef MCKPP_BUILD_ID
#define MCKPP_BUILD_ID "unknown"
#endif
const char *mckpp_hip_build_id(void) { return MCKPP_BUILD_ID; }
#ifndef MCKPP_BUILD_COMPILER
#define MCKPP_BUILD_COMPILER "unknown"
#endif
const char *mckpp_hip_build_compiler(void) { return MCKPP_BUILD_COMPILER; }

int mckpp_hip_device_count(void)
{
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail("hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

// mckpp_physics_lookup, src/mckpp_physics_lookup_mod.F90:11-66 (host, libm pow).
void mckpp_host_lookup(double vonk, double *wmt, double *wst)
{
  const int ni = MCKPP_NI, nj = MCKPP_NJ, n1 = ni + 2;
  const double epsln = 1.e-20, c1 = 5.0, zmin = -4.e-7, zmax = 0.0, umin = 0.0, umax = 0.04;
  const double am = 1.257, cm = 8.380, c2 = 16.0, zetam = -0.2, as = -28.86, cs = 98.96, c3 = 16.0,
               zetas = -1.0;
  const double deltaz = (zmax - zmin) / (ni + 1);
  const double deltau = (umax - umin) / (nj + 1);
  for (int j = 0; j <= nj + 1; ++j) {
    const double usta = deltau * (j) + umin;
    const double u3 = (usta * usta) * usta;
    for (int i = 0; i <= ni + 1; ++i) {
      const double zehat = deltaz * (i) + zmin;
      const double zeta = zehat / (u3 + epsln);
      double wm, ws;
      if (zehat >= 0.) {
        wm = vonk * usta / (1. + c1 * zeta);
        ws = wm;
      } else {
        wm = (zeta > zetam) ? vonk * usta * std::pow(1. - c2 * zeta, 1. / 4.)
                            : vonk * std::pow(am * u3 - cm * zehat, 1. / 3.);
        // **(1./2.) is a square root in the reference's build (oracle/conv_probe.F90), not libm's pow
        ws = (zeta > zetas) ? vonk * usta * std::sqrt(1. - c3 * zeta)
                            : vonk * std::pow(as * u3 - cs * zehat, 1. / 3.);
      }
      wmt[(size_t)j * n1 + i] = wm;
      wst[(size_t)j * n1 + i] = ws;
    }
  }
}

// tri(0:nztmax,0:1,1), src/mckpp_initialize_ocean.F90:30-43.  zm, hm are zm(1:nzp1), hm(1:nzp1).
void mckpp_host_tri(int32_t nz, int32_t nztmax, double dto, const double *zm, const double *hm, double *tri)
{
  const int n1 = nztmax + 1;
  auto Z = [&](int k) { return zm[k - 1]; };
  auto H = [&](int k) { return hm[k - 1]; };
  double *t0 = tri, *t1 = tri + n1;
  for (int k = 0; k < n1; ++k) { t0[k] = 0.0; t1[k] = 0.0; }
  t1[0] = dto / H(1);
  t1[1] = dto / H(1) / (Z(1) - Z(2));
  for (int k = 2; k <= nz; ++k) {
    t1[k] = dto / H(k) / (Z(k) - Z(k + 1));
    t0[k] = dto / H(k) / (Z(k - 1) - Z(k));
  }
}

int mckpp_hip_init(const mckpp_const_c *c, int device, mckpp_hip_handle *out)
{
  if (!c || !out) return fail("mckpp_hip_init: null argument");
  if (c->nz < 2) return fail("mckpp_hip_init: nz=%d (need >= 2)", c->nz);
  if (c->nztmax < c->nz + 1) return fail("mckpp_hip_init: nztmax=%d < nzp1=%d", c->nztmax, c->nz + 1);
  if (!c->zm || !c->hm || !c->dm || !c->tri || !c->wmt || !c->wst)
    return fail("mckpp_hip_init: zm/hm/dm/tri/wmt/wst must all be set");
  // LKPP=.FALSE. is not a defined configuration of the reference: kppmix then never assigns hbl / kbl
  // (src/mckpp_physics_verticalmixing_kppmix_mod.F90:87-118 is skipped), which ocnstep stores as hmix / kmix and uses as
  // the index of dm() (src/mckpp_physics_ocnstep_mod.F90:305-314, ocnint_mod.F90:97-114) - uninitialised memory.  Refused.
  if (!c->LKPP) return fail("mckpp_hip_init: LKPP=.FALSE. leaves hmix/kmix unassigned in the reference (kppmix_mod.F90:87-118): not a defined configuration, not emulated");
  if (c->maxmodeadv < 0 || c->maxmodeadv > 16) return fail("mckpp_hip_init: maxmodeadv=%d", c->maxmodeadv);
  int solver_mode_env = 0;   // default of mckpp_hip_set_solver_mode
  if (const char *e = getenv("MCKPP_SOLVER_MODE")) {
    solver_mode_env = atoi(e);
    if (solver_mode_env < 0 || solver_mode_env > 1) return fail("mckpp_hip_init: MCKPP_SOLVER_MODE=%s (0: the reference's order, 1: two-ended)", e);
  }
  if (c->L_NO_ISOTHERM && (c->iso_bot < 2 || c->iso_bot > c->nz + 1))
    return fail("mckpp_hip_init: iso_bot=%d outside 2..nzp1", c->iso_bot);
  const int nzp1 = c->nz + 1;
  const int lpl = (nzp1 + 2 + 63) / 64;
  if (lpl > 8) return fail("mckpp_hip_init: nz=%d too deep (max 509 levels: profile rows are padded to at most 512 doubles)", c->nz);
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail("mckpp_hip_init: device %d of %d", device, ndev);
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail("mckpp_hip_init: device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);

  mckpp_hip_ctx *h = new mckpp_hip_ctx();
  // any failure below releases what has been created so far (the HIPCHK returns included)
  struct guard { mckpp_hip_ctx *p; ~guard() { if (p) mckpp_hip_finalize(p); } } g{h};
  h->device = device;
  h->c = *c;
  h->nz = c->nz;
  h->nzp1 = nzp1;
  h->lpl = lpl;
  h->ld = 64 * lpl;
  h->ldc = 64 * lpl + 8;
  h->num_cu = prop.multiProcessorCount;
  h->ext = c->LDD || c->L_RELAX_SST || c->L_FCORR || c->L_FCORR_WITHZ || c->L_SFCORR || c->L_SFCORR_WITHZ ||
           c->L_RELAX_SAL || c->L_RELAX_OCNT || c->L_NO_FREEZE || c->L_NO_ISOTHERM || c->L_DAMP_CURR ||
           c->clim_present || c->L_ADVECT;
  h->ext_kernel = c->LDD || c->L_RELAX_SST || c->L_FCORR || c->L_FCORR_WITHZ || c->L_SFCORR || c->L_SFCORR_WITHZ ||
                  c->L_RELAX_SAL || c->L_RELAX_OCNT || c->L_NO_FREEZE || c->L_NO_ISOTHERM || c->L_DAMP_CURR || c->L_ADVECT;
  HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIPCHK(hipMalloc(&h->d_qhead, QBLOCK_INTS * sizeof(int)));   // queue heads [16], queue owners [16], straggler count and spare [32]
  h->solo_after = 12;
  h->solo_limit = std::max(2, h->num_cu / 32);
  if (const char *e = getenv("MCKPP_SOLO")) { if (atoi(e) == 0) h->solo_limit = 0; }
  if (const char *e = getenv("MCKPP_SOLO_LIMIT")) h->solo_limit = std::max(0, atoi(e));
  if (const char *e = getenv("MCKPP_SOLO_AFTER")) h->solo_after = std::max(0, atoi(e));
  if (const char *e = getenv("MCKPP_VIEW_KMAX")) h->view_kmax = std::max(0, atoi(e));
  {   // the device's XCDs (a column of a multi-step launch stays on one: mckpp_kernels_ps.hip, M0)
    HIPCHK(hipMemsetAsync(h->d_qhead, 0, sizeof(int), h->stream));
    HIPCHK(mckpp_launch_xcc_probe(reinterpret_cast<unsigned *>(h->d_qhead), h->stream));
    unsigned mask = 0;
    HIPCHK(hipMemcpyAsync(&mask, h->d_qhead, sizeof mask, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->nqueues = 0;
    for (int i = 0; i < 16; ++i) h->xcc_queue[i] = (mask >> i & 1u) ? h->nqueues++ : -1;
    if (h->nqueues == 0) h->multistep = false;
    // HIP promises nothing about where workgroups run: a launch may leave an XCD the probe saw without any (its queue is
    // then adopted, whole, by another XCD: k_column_ps M0), or run workgroups on one the probe did not see (they start
    // without a queue and adopt one, or end).  MCKPP_XCC_DROP=<bit mask of XCC ids> takes the queue away from those XCDs'
    // workgroups - the queues stay - so that both paths run on a device where they otherwise never would (tests).
    if (const char *e = getenv("MCKPP_XCC_DROP")) {
      const unsigned drop = (unsigned)strtoul(e, nullptr, 0);
      for (int i = 0; i < 16; ++i) if (drop >> i & 1u) h->xcc_queue[i] = -1;
    }
    if (getenv("MCKPP_PS_VERBOSE"))
      fprintf(stderr, "[mckpp] XCC ids seen by the probe: mask 0x%x, %d queues\n", mask, h->nqueues);
  }
  HIPCHK(hipMalloc(&h->d_params, sizeof(mckpp_kparams)));
  h->scratch_doubles = mckpp_ps_scratch_doubles(nzp1, h->ext_kernel ? (c->LDD ? 2 : 1) : 0, h->num_cu);
  HIPCHK(hipMalloc(&h->d_scratch, h->scratch_doubles * sizeof(double)));
  HIPCHK(hipMemset(h->d_scratch, 0, h->scratch_doubles * sizeof(double)));
  if (getenv("MCKPP_STAMP")) {
    HIPCHK(hipMalloc(&h->d_dbg, 32 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(h->d_dbg, 0, 32 * sizeof(unsigned long long)));
  }
  HIPCHK(hipEventCreate(&h->ev0));
  HIPCHK(hipEventCreate(&h->ev1));
  HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  for (int b = 0; b < 2; ++b) {
    HIPCHK(hipEventCreateWithFlags(&h->ev_lay[b], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_copy[b], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_params[b], hipEventDisableTiming));
  }
  HIPCHK(hipEventCreateWithFlags(&h->ev_f, hipEventDisableTiming));
  HIPCHK(hipHostMalloc(&h->h_params, 2 * sizeof(mckpp_kparams), hipHostMallocDefault));

  const int ldc = h->ldc, nz = h->nz, n1 = c->nztmax + 1;
  std::vector<double> zm(ldc, 0.0), hm(ldc, 0.0), t0(ldc, 0.0), t1(ldc, 0.0);
  for (int k = 1; k <= nzp1; ++k) { zm[k] = c->zm[k - 1]; hm[k] = c->hm[k - 1]; }
  for (int k = 0; k <= nz; ++k) { t0[k] = c->tri[k]; t1[k] = c->tri[n1 + k]; }
  h->dm_nz = c->dm[nz];
  {   // layers above a tenth of the deepest level's depth = trips of the reference-level loop (verticalmixing_mod.F90:118-131)
      // of that level: 6 on a uniform 60-level grid, 10 at 100 levels, 25 on the stretched 69-level grid.  From 16 on, the
      // extra phase that forms the whole-layer terms once per column pays (measured: +2 % on the stretched grid, -0.7 % at 60)
    int nref = 0;
    for (int k = 1; k <= nz; ++k) nref += c->zm[k - 1] > 0.1 * c->zm[nz - 1];
    h->l2pre = (nref >= 16 && !c->LDD) ? 1 : 0;
    if (const char *e = getenv("MCKPP_L2PRE")) h->l2pre = (atoi(e) != 0 && !c->LDD) ? 1 : 0;
    if (const char *e = getenv("MCKPP_L3_CAP")) h->l3cap = atoi(e) > 0 ? atoi(e) : 0;
    h->solver_mode = solver_mode_env;
    if (const char *e = getenv("MCKPP_MULTISTEP")) h->multistep = atoi(e) != 0 && h->nqueues > 0;
  }
  {
    std::vector<double> dm(ldc, 0.0), hs(ldc, 0.0);
    for (int k = 0; k <= nz; ++k) dm[k] = c->dm[k];
    double acc = 0.0;
    for (int n = 1; n <= nzp1; ++n) { acc = acc + hm[n]; hs[n] = acc; }
    HIPCHK(hipMalloc(&h->d_dm, ldc * sizeof(double)));
    HIPCHK(hipMemcpy(h->d_dm, dm.data(), ldc * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&h->d_hsum, ldc * sizeof(double)));
    HIPCHK(hipMemcpy(h->d_hsum, hs.data(), ldc * sizeof(double), hipMemcpyHostToDevice));
  }
  // Jerlov tables: swfrac_opt (swfrac_mod.F90:36-41, fact = hbf = 1) and swdk_opt (fluxes_mod.F90:104-107)
  h->h_swfrac_tab.assign((size_t)6 * ldc, 0.0);
  h->h_swdk_tab.assign((size_t)6 * ldc, 0.0);
  for (int jw = 1; jw <= 5; ++jw) {
    for (int l = 1; l <= nzp1; ++l) {
      const double rmin = -80.;
      double r1 = zm[l] * 1.0 / jer_a1[jw]; r1 = r1 > rmin ? r1 : rmin;
      double r2 = zm[l] * 1.0 / jer_a2[jw]; r2 = r2 > rmin ? r2 : rmin;
      h->h_swfrac_tab[(size_t)jw * ldc + l] = jer_rfac[jw] * mckpp_exp(r1) + (1. - jer_rfac[jw]) * mckpp_exp(r2);
    }
    for (int k = 0; k <= nz; ++k) {
      const double z = -c->dm[k];
      h->h_swdk_tab[(size_t)jw * ldc + k] =
          jer_rfac[jw] * mckpp_exp(z / jer_a1[jw]) + (1.0 - jer_rfac[jw]) * mckpp_exp(z / jer_a2[jw]);
    }
  }
  {  // bldepth_mod.F90:91 and blmix_mod.F90:62
    const double cv = 1.6, cs = 98.96, epsilon = 0.1, Ricr = 0.30, cstar = 5.0;
    h->Vtc = cv * std::sqrt(0.2 / cs / epsilon) / (c->vonk * c->vonk) / Ricr;
    h->cg = cstar * c->vonk * std::pow(cs * c->vonk * epsilon, 1. / 3.);
  }
  const size_t nt = (size_t)(MCKPP_NI + 2) * (MCKPP_NJ + 2);
  std::vector<double2> wtab(nt);
  for (size_t i = 0; i < nt; ++i) wtab[i] = make_double2(c->wmt[i], c->wst[i]);

  auto up = [&](double **dst, const double *src, size_t n) -> hipError_t {
    hipError_t e = hipMalloc(dst, n * sizeof(double));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src, n * sizeof(double), hipMemcpyHostToDevice);
  };
  HIPCHK(up(&h->d_zm, zm.data(), ldc));
  HIPCHK(up(&h->d_hm, hm.data(), ldc));
  HIPCHK(up(&h->d_tri0, t0.data(), ldc));
  HIPCHK(up(&h->d_tri1, t1.data(), ldc));
  HIPCHK(up(&h->d_swfrac_tab, h->h_swfrac_tab.data(), (size_t)6 * ldc));
  HIPCHK(up(&h->d_swdk_tab, h->h_swdk_tab.data(), (size_t)6 * ldc));
  HIPCHK(hipMalloc(&h->d_wtab, nt * sizeof(double2)));
  HIPCHK(hipMemcpy(h->d_wtab, wtab.data(), nt * sizeof(double2), hipMemcpyHostToDevice));
  // the host pointers are not kept
  h->c.zm = h->c.hm = h->c.dm = h->c.tri = h->c.wmt = h->c.wst = nullptr;
  g.p = nullptr;
  *out = h;
  return 0;
}

static void free_state(mckpp_hip_ctx *h)
{
  for (auto &p : h->d_prof) { if (p) hipFree(p); p = nullptr; }
  for (auto &p : h->d_diag) { if (p) hipFree(p); p = nullptr; }
  for (auto &p : h->d_ext_in) { if (p) hipFree(p); p = nullptr; }
  for (auto &p : h->d_ext_out) { if (p) hipFree(p); p = nullptr; }
  if (h->d_xs) hipFree(h->d_xs);
  if (h->d_adv_d) hipFree(h->d_adv_d);
  if (h->d_adv_i) hipFree(h->d_adv_i);
  h->d_xs = nullptr; h->d_adv_d = nullptr; h->d_adv_i = nullptr;
  if (h->d_series) hipFree(h->d_series);
  h->d_series = nullptr; h->series_nrec = 0;
  for (auto *p : h->d_wacc) if (p) hipFree(p);
  h->d_wacc.clear(); h->window_count = 0;
  if (h->d_cs) hipFree(h->d_cs);
  if (h->d_ci) hipFree(h->d_ci);
  if (h->d_ipt) hipFree(h->d_ipt);
  if (h->d_stage) hipFree(h->d_stage);
  h->d_cs = nullptr; h->d_ci = nullptr; h->d_ipt = nullptr; h->d_stage = nullptr;
  h->stage_elems = 0;
  for (int b = 0; b < 2; ++b) { if (h->d_xfer[b]) hipFree(h->d_xfer[b]); h->d_xfer[b] = nullptr; h->xfer_elems[b] = 0; }
  if (h->d_done) hipFree(h->d_done);
  h->d_done = nullptr;
  if (h->d_pack) hipFree(h->d_pack);
  if (h->h_pack) hipHostFree(h->h_pack);
  h->d_pack = nullptr; h->h_pack = nullptr;
  if (h->h_cs) hipHostFree(h->h_cs);
  if (h->h_ci) hipHostFree(h->h_ci);
  if (h->h_f) hipHostFree(h->h_f);
  h->h_cs = nullptr; h->h_ci = nullptr; h->h_f = nullptr; h->h_f_elems = 0;
  h->ncol = 0; h->npts = 0;
}

// The caller's arrays stay pinned until the context goes (or mckpp_hip_release_host_arrays is called)
static void unpin_all(mckpp_hip_ctx *h)
{
  for (auto &r : h->pinned)
    if (hipHostUnregister(const_cast<void *>(r.first)) != hipSuccess) (void)hipGetLastError();
  h->pinned.clear();
  h->unpinnable.clear();
}

int mckpp_hip_finalize(mckpp_hip_handle h)
{
  if (!h) return 0;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->copy_stream) hipStreamSynchronize(h->copy_stream);
  unpin_all(h);
  free_state(h);
  hipFree(h->d_zm); hipFree(h->d_hm); hipFree(h->d_tri0); hipFree(h->d_tri1);
  hipFree(h->d_swfrac_tab); hipFree(h->d_swdk_tab); hipFree(h->d_wtab); hipFree(h->d_qhead); hipFree(h->d_params); hipFree(h->d_scratch); hipFree(h->d_dm); hipFree(h->d_hsum);
  if (h->d_dbg) hipFree(h->d_dbg);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  for (int b = 0; b < 2; ++b) {
    if (h->ev_lay[b]) hipEventDestroy(h->ev_lay[b]);
    if (h->ev_copy[b]) hipEventDestroy(h->ev_copy[b]);
    if (h->ev_params[b]) hipEventDestroy(h->ev_params[b]);
  }
  if (h->ev_f) hipEventDestroy(h->ev_f);
  if (h->h_params) hipHostFree(h->h_params);
  if (h->copy_stream) hipStreamDestroy(h->copy_stream);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int64_t mckpp_hip_ncolumns(mckpp_hip_handle h) { return h ? h->ncol : -1; }

static int ensure_stage(mckpp_hip_ctx *h, size_t elems)
{
  if (elems <= h->stage_elems) return 0;
  // a fluxes / unpack / window kernel queued by an earlier call may still be reading the block (those calls do not
  // end with a synchronisation): wait for this context's stream, not - through hipFree - for the whole device
  if (h->d_stage) HIPCHK(hipStreamSynchronize(h->stream));
  if (h->d_stage) hipFree(h->d_stage);
  h->d_stage = nullptr;
  h->stage_elems = 0;
  HIPCHK(hipMalloc(&h->d_stage, elems * sizeof(double)));
  h->stage_elems = elems;
  return 0;
}

// MCKPP_HIP_NO_HOST_REGISTER=1: never pin the caller's arrays (transfers then go through the runtime's own
// pageable path: correct, several times slower)
static bool no_host_register()
{
  static const bool v = getenv("MCKPP_HIP_NO_HOST_REGISTER") != nullptr;
  return v;
}

// Pin a caller array once (portable: every device's transfers benefit).  Failure is not an error: an array that
// is already pinned (by another context, or by the caller) or cannot be is simply transferred as it is.
static void pin_host(mckpp_hip_ctx *h, const void *ptr, size_t bytes)
{
  if (!ptr || bytes < ((size_t)1 << 18) || no_host_register()) return;
  const char *b = static_cast<const char *>(ptr);
  for (auto &r : h->pinned)        // inside a range this context pinned
    if (b >= static_cast<const char *>(r.first) && b + bytes <= static_cast<const char *>(r.first) + r.second) return;
  for (auto &r : h->unpinnable)    // tried before: pinned by someone else, or cannot be
    if (r.first == ptr && r.second == bytes) return;
  static const bool verbose = getenv("MCKPP_HIP_VERBOSE") != nullptr;
  const hipError_t e = hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterPortable);
  if (e == hipSuccess) h->pinned.emplace_back(ptr, bytes);
  else { (void)hipGetLastError(); h->unpinnable.emplace_back(ptr, bytes); }
  if (verbose) fprintf(stderr, "[mckpp] hipHostRegister(%p, %zu B): %s\n", ptr, bytes, e == hipSuccess ? "pinned" : hipGetErrorString(e));
}

static int ensure_xfer(mckpp_hip_ctx *h, unsigned b, size_t elems)
{
  if (elems <= h->xfer_elems[b]) return 0;
  HIPCHK(hipStreamSynchronize(h->stream));        // nothing in flight may still use the buffer
  HIPCHK(hipStreamSynchronize(h->copy_stream));
  if (h->d_xfer[b]) hipFree(h->d_xfer[b]);
  h->d_xfer[b] = nullptr;
  h->xfer_elems[b] = 0;
  HIPCHK(hipMalloc(&h->d_xfer[b], elems * sizeof(double)));
  h->xfer_elems[b] = elems;
  return 0;
}

// host Fortran slab (npts x nlev, from `src`) -> device rows.  Queued: the copy on the copy stream, the layout
// kernel behind it on the context's stream; the caller waits for the stream before `src` may change.
static int up_rows(mckpp_hip_ctx *h, const double *src, int nlev, double *dst, int dst_off, const double *whole = nullptr,
                   size_t whole_elems = 0)
{
  const size_t n = (size_t)h->npts * nlev;
  const unsigned b = h->xfer_seq++ & 1u;
  if (ensure_xfer(h, b, n)) return -1;
  // the array the slab belongs to is pinned as a whole (slabs of one array share pages at their boundaries)
  if (whole) pin_host(h, whole, whole_elems * sizeof(double));
  else pin_host(h, src, n * sizeof(double));
  HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_lay[b], 0));   // the layout kernel that last read this buffer
  HIPCHK(hipMemcpyAsync(h->d_xfer[b], src, n * sizeof(double), hipMemcpyHostToDevice, h->copy_stream));
  HIPCHK(hipEventRecord(h->ev_copy[b], h->copy_stream));
  HIPCHK(hipStreamWaitEvent(h->stream, h->ev_copy[b], 0));
  HIPCHK(mckpp_launch_gather_rows(h->d_xfer[b], h->npts, nlev, 0, h->d_ipt, h->ncol, dst, h->ld, dst_off, h->stream));
  HIPCHK(hipEventRecord(h->ev_lay[b], h->stream));
  return 0;
}

// device rows -> host Fortran slab; entries of non-resident (land) columns keep their host values.  Queued: the
// layout kernel on the context's stream, the copy to the host behind it on the copy stream; xfer_finish() before
// the caller reads `dst`.
static int down_rows(mckpp_hip_ctx *h, const double *src, int src_ld, int src_off, int nlev, double *dst,
                     const double *whole = nullptr, size_t whole_elems = 0)
{
  const size_t n = (size_t)h->npts * nlev;
  const unsigned b = h->xfer_seq++ & 1u;
  if (ensure_xfer(h, b, n)) return -1;
  if (whole) pin_host(h, whole, whole_elems * sizeof(double));
  else pin_host(h, dst, n * sizeof(double));
  HIPCHK(hipStreamWaitEvent(h->stream, h->ev_copy[b], 0));   // the transfer that last read this buffer
  if (h->ncol < h->npts)
    HIPCHK(hipMemcpyAsync(h->d_xfer[b], dst, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(mckpp_launch_scatter_rows(src, src_ld, src_off, h->d_ipt, h->ncol, h->d_xfer[b], h->npts, nlev, 0, h->stream));
  HIPCHK(hipEventRecord(h->ev_lay[b], h->stream));
  HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_lay[b], 0));
  HIPCHK(hipMemcpyAsync(dst, h->d_xfer[b], n * sizeof(double), hipMemcpyDeviceToHost, h->copy_stream));
  HIPCHK(hipEventRecord(h->ev_copy[b], h->copy_stream));
  return 0;
}

static int xfer_finish(mckpp_hip_ctx *h)
{
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->copy_stream));
  return 0;
}


// Device buffers of the column state for `ncol` resident columns out of `npts` grid points (every
// row zeroed).  Used by upload and by load_restart, so a context with the optional physics gets
// its input/output rows and per-column records from either.
static int alloc_state(mckpp_hip_ctx *h, int64_t npts, int64_t ncol)
{
  free_state(h);
  h->npts = npts;
  h->ncol = ncol;
  h->ext_inputs_resident = false;
  if (ncol <= 0) return 0;
  const size_t rowbytes = (size_t)ncol * h->ld * sizeof(double);
  for (auto &p : h->d_prof) { HIPCHK(hipMalloc(&p, rowbytes)); HIPCHK(hipMemsetAsync(p, 0, rowbytes, h->stream)); }
  for (auto &p : h->d_diag) { HIPCHK(hipMalloc(&p, rowbytes)); HIPCHK(hipMemsetAsync(p, 0, rowbytes, h->stream)); }
  HIPCHK(hipMalloc(&h->d_cs, (size_t)ncol * MCKPP_CS * sizeof(double)));
  HIPCHK(hipMalloc(&h->d_ci, (size_t)ncol * MCKPP_CI * sizeof(int)));
  HIPCHK(hipMalloc(&h->d_ipt, (size_t)ncol * sizeof(int)));
  HIPCHK(hipHostMalloc(&h->h_cs, (size_t)ncol * MCKPP_CS * sizeof(double), hipHostMallocDefault));
  HIPCHK(hipHostMalloc(&h->h_ci, (size_t)ncol * MCKPP_CI * sizeof(int), hipHostMallocDefault));
  if (h->ext) {
    for (auto &p : h->d_ext_in) { HIPCHK(hipMalloc(&p, rowbytes)); HIPCHK(hipMemsetAsync(p, 0, rowbytes, h->stream)); }
    for (auto &p : h->d_ext_out) { HIPCHK(hipMalloc(&p, rowbytes)); HIPCHK(hipMemsetAsync(p, 0, rowbytes, h->stream)); }
    const size_t nadv = (size_t)ncol * (h->c.maxmodeadv + 1);
    HIPCHK(hipMalloc(&h->d_xs, (size_t)ncol * MCKPP_XS * sizeof(double)));
    HIPCHK(hipMemsetAsync(h->d_xs, 0, (size_t)ncol * MCKPP_XS * sizeof(double), h->stream));
    HIPCHK(hipMalloc(&h->d_adv_d, nadv * sizeof(double)));
    HIPCHK(hipMemsetAsync(h->d_adv_d, 0, nadv * sizeof(double), h->stream));
    HIPCHK(hipMalloc(&h->d_adv_i, nadv * sizeof(int)));
    HIPCHK(hipMemsetAsync(h->d_adv_i, 0, nadv * sizeof(int), h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// Inputs of the optional physics: what mckpp_boundary_update and the ancillary readers
// rewrite on the host between steps (src/mckpp_ocean_model_3D.F90:51-55) - relaxation times and targets,
// flux corrections, climatologies, prescribed advection.  Rows go through the staging buffer; the
// per-column scalars are compacted on the host.
static int upload_ancillaries(mckpp_hip_ctx *h, const mckpp_state_ptrs_c *s, const char *who = "mckpp_hip_upload")
{
  const mckpp_const_c &k = h->c;
  const int64_t npts = h->npts, ncol = h->ncol;
  const int nzp1 = h->nzp1;
  const std::vector<int> &ipt = h->ipt;
  if ((k.L_FCORR_WITHZ && !s->fcorr_withz) || (k.L_SFCORR_WITHZ && !s->sfcorr_withz) ||
      ((k.L_RELAX_OCNT || k.clim_present || k.L_NO_ISOTHERM) && !s->ocnT_clim) ||
      ((k.L_RELAX_SAL || k.clim_present || k.L_NO_ISOTHERM) && !s->sal_clim) ||
      (k.L_RELAX_SST && (!s->relax_sst || !s->SST0)) || (k.L_FCORR && !s->fcorr_twod) ||
      (k.L_RELAX_SAL && !s->relax_sal) || (k.L_RELAX_OCNT && !s->relax_ocnT))
    return fail("%s: a switch is on but the field it reads is a NULL pointer", who);
  if (s->fcorr_withz && up_rows(h, s->fcorr_withz, nzp1, h->d_ext_in[E_FCORR_WITHZ], 0)) return -1;
  if (s->sfcorr_withz && up_rows(h, s->sfcorr_withz, nzp1, h->d_ext_in[E_SFCORR_WITHZ], 0)) return -1;
  if (s->ocnT_clim && up_rows(h, s->ocnT_clim, nzp1, h->d_ext_in[E_OCNT_CLIM], 0)) return -1;
  if (s->sal_clim && up_rows(h, s->sal_clim, nzp1, h->d_ext_in[E_SAL_CLIM], 0)) return -1;
  const int mm = k.maxmodeadv;
  std::vector<double> xs((size_t)ncol * MCKPP_XS, 0.0), ad((size_t)ncol * (mm + 1), 0.0);
  std::vector<int> ai((size_t)ncol * (mm + 1), 0);
  for (int64_t c = 0; c < ncol; ++c) {
    const int64_t i = ipt[c];
    double *x = &xs[(size_t)c * MCKPP_XS];
    x[XS_RELAX_SST] = s->relax_sst ? s->relax_sst[i] : 0.0;
    x[XS_SST0] = s->SST0 ? s->SST0[i] : 0.0;
    x[XS_FCORR_TWOD] = s->fcorr_twod ? s->fcorr_twod[i] : 0.0;
    x[XS_RELAX_SAL] = s->relax_sal ? s->relax_sal[i] : 0.0;
    x[XS_RELAX_OCNT] = s->relax_ocnT ? s->relax_ocnT[i] : 0.0;
    int nm = (k.L_ADVECT && s->nmodeadv) ? s->nmodeadv[i + npts * 1] : 0;   // nmodeadv(ipt,2)
    if (nm < 0 || nm > mm) return fail("%s: nmodeadv(%lld,2)=%d outside 0..%d", who, (long long)i + 1, nm, mm);
    ai[(size_t)c * (mm + 1)] = nm;
    for (int j = 0; j < mm; ++j) {
      ai[(size_t)c * (mm + 1) + 1 + j] = s->modeadv ? s->modeadv[i + npts * (j + (int64_t)mm * 1)] : 0;
      ad[(size_t)c * (mm + 1) + j] = s->advection ? s->advection[i + npts * (j + (int64_t)mm * 1)] : 0.0;
    }
  }
  HIPCHK(hipMemcpy(h->d_xs, xs.data(), xs.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_adv_d, ad.data(), ad.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_adv_i, ai.data(), ai.size() * sizeof(int), hipMemcpyHostToDevice));
  h->ext_inputs_resident = true;
  return 0;
}

static void fill_params(mckpp_hip_ctx *h, mckpp_kparams &p, int ntime, int mode);

int mckpp_hip_upload(mckpp_hip_handle h, const mckpp_state_ptrs_c *s)
{
  if (!h || !s) return fail("mckpp_hip_upload: null argument");
  if (s->npts <= 0) return fail("mckpp_hip_upload: npts=%lld", (long long)s->npts);
  if (!s->U || !s->X) return fail("mckpp_hip_upload: U and X are required");
  HIPCHK(hipSetDevice(h->device));
  const int64_t npts = s->npts;
  const int nzp1 = h->nzp1;
  std::vector<int> ipt;
  ipt.reserve(npts);
  for (int64_t i = 0; i < npts; ++i)
    if (!s->run_physics || s->run_physics[i]) ipt.push_back((int)i);
  const int64_t ncol = (int64_t)ipt.size();
  if (ncol != h->ncol || npts != h->npts) {
    if (alloc_state(h, npts, ncol)) return -1;
  }
  if (ipt != h->ipt && h->d_series) {   // resident flux records were compacted with the previous land mask
    hipFree(h->d_series);
    h->d_series = nullptr;
    h->series_nrec = 0;
  }
  h->ipt = ipt;
  if (ncol == 0) return 0;
  HIPCHK(hipMemcpyAsync(h->d_ipt, ipt.data(), (size_t)ncol * sizeof(int), hipMemcpyHostToDevice, h->stream));
  const size_t slab = (size_t)npts * nzp1;
  if (up_rows(h, s->U, nzp1, h->d_prof[P_U], 0, s->U, 2 * slab)) return -1;
  if (up_rows(h, s->U + slab, nzp1, h->d_prof[P_V], 0, s->U, 2 * slab)) return -1;
  if (up_rows(h, s->X, nzp1, h->d_prof[P_T], 0, s->X, 2 * slab)) return -1;
  if (up_rows(h, s->X + slab, nzp1, h->d_prof[P_S], 0, s->X, 2 * slab)) return -1;
  if (s->Us) {   // Us(npts,nzp1,nvel,0:1)
    if (up_rows(h, s->Us + 0 * slab, nzp1, h->d_prof[P_US0], 0, s->Us, 4 * slab)) return -1;
    if (up_rows(h, s->Us + 1 * slab, nzp1, h->d_prof[P_VS0], 0, s->Us, 4 * slab)) return -1;
    if (up_rows(h, s->Us + 2 * slab, nzp1, h->d_prof[P_US1], 0, s->Us, 4 * slab)) return -1;
    if (up_rows(h, s->Us + 3 * slab, nzp1, h->d_prof[P_VS1], 0, s->Us, 4 * slab)) return -1;
  }
  if (s->Xs) {
    if (up_rows(h, s->Xs + 0 * slab, nzp1, h->d_prof[P_TS0], 0, s->Xs, 4 * slab)) return -1;
    if (up_rows(h, s->Xs + 1 * slab, nzp1, h->d_prof[P_SS0], 0, s->Xs, 4 * slab)) return -1;
    if (up_rows(h, s->Xs + 2 * slab, nzp1, h->d_prof[P_TS1], 0, s->Xs, 4 * slab)) return -1;
    if (up_rows(h, s->Xs + 3 * slab, nzp1, h->d_prof[P_SS1], 0, s->Xs, 4 * slab)) return -1;
  }
  if (s->U_init) {
    if (up_rows(h, s->U_init, nzp1, h->d_prof[P_UINIT], 0, s->U_init, 2 * slab)) return -1;
    if (up_rows(h, s->U_init + slab, nzp1, h->d_prof[P_VINIT], 0, s->U_init, 2 * slab)) return -1;
  }
  if (h->ext && upload_ancillaries(h, s)) return -1;
  std::vector<double> cs((size_t)ncol * MCKPP_CS, 0.0);
  std::vector<int> ci((size_t)ncol * MCKPP_CI, 0);
  const int64_t fl_i = npts;                                        // stride of the flux index
  const int64_t fl_5 = npts * (int64_t)h->c.nsflxs * (5 - 1);       // offset of (:,:,5,0)
  for (int64_t c = 0; c < ncol; ++c) {
    const int64_t i = ipt[c];
    double *r = &cs[(size_t)c * MCKPP_CS];
    int *q = &ci[(size_t)c * MCKPP_CI];
    auto g = [&](const double *a, double def) { return a ? a[i] : def; };
    r[CS_F] = g(s->f, 0.0);
    r[CS_SSURF] = g(s->Ssurf, 0.0);
    r[CS_SREF] = g(s->Sref, 0.0);
    r[CS_SSREF] = g(s->SSref, 0.0);
    r[CS_OCDEPTH] = g(s->ocdepth, -10000.0);
    for (int m = 0; m < 6; ++m) r[CS_SFLUX1 + m] = s->sflux ? s->sflux[i + fl_i * m + fl_5] : 0.0;
    r[CS_HMIXD0] = s->hmixd ? s->hmixd[i] : 0.0;
    r[CS_HMIXD1] = s->hmixd ? s->hmixd[i + npts] : 0.0;
    r[CS_HMIX] = g(s->hmix, 0.0);
    r[CS_KMIX] = g(s->kmix, 0.0);
    r[CS_UREF] = g(s->uref, 0.0);
    r[CS_VREF] = g(s->vref, 0.0);
    r[CS_TREF] = g(s->Tref, 0.0);
    r[CS_RESET] = g(s->reset_flag, 0.0);
    r[CS_DAMPU] = g(s->dampu_flag, 0.0);
    r[CS_DAMPV] = g(s->dampv_flag, 0.0);
    r[CS_FREEZE] = g(s->freeze_flag, 0.0);
    r[CS_FCORR] = g(s->fcorr, 0.0);
    q[CI_OLD] = s->old ? s->old[i] : 0;
    q[CI_NEW] = s->new_ ? s->new_[i] : 1;
    q[CI_JERLOV] = s->jerlov ? s->jerlov[i] : 3;
    if (q[CI_JERLOV] < 1 || q[CI_JERLOV] > 5) return fail("mckpp_hip_upload: jerlov(%lld)=%d outside 1..5", (long long)i + 1, q[CI_JERLOV]);
    q[CI_INITFLAG] = s->l_initflag ? (s->l_initflag[i] != 0) : 0;
    q[CI_LOCEAN] = s->l_ocean ? (s->l_ocean[i] != 0) : 1;
    q[CI_IPT] = (int)i;
  }
  HIPCHK(hipMemcpyAsync(h->d_cs, cs.data(), cs.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_ci, ci.data(), ci.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

static int ensure_host_f(mckpp_hip_ctx *h, size_t elems)
{
  if (elems <= h->h_f_elems) return 0;
  HIPCHK(hipEventSynchronize(h->ev_f));
  if (h->h_f) hipHostFree(h->h_f);
  h->h_f = nullptr;
  h->h_f_elems = 0;
  HIPCHK(hipHostMalloc(&h->h_f, elems * sizeof(double), hipHostMallocDefault));
  h->h_f_elems = elems;
  return 0;
}

int mckpp_hip_set_forcing(mckpp_hip_handle h, const double *sflux)
{
  if (!h || !sflux) return fail("mckpp_hip_set_forcing: null argument");
  if (h->ncol == 0) return 0;
  HIPCHK(hipSetDevice(h->device));
  // sflux(:,1:6,5,0) is six contiguous (npts) slabs: up as they are (from the caller's array, pinned on first
  // use), compacted into the records' flux slots on the device.  The call returns when the slabs have been read -
  // the caller may rewrite its array - not when the records are updated (that is stream-ordered before the step).
  const size_t n6 = (size_t)h->npts * 6;
  const double *slabs = sflux + h->npts * (int64_t)h->c.nsflxs * 4;
  if (ensure_stage(h, n6)) return -1;
  pin_host(h, sflux, (size_t)h->npts * h->c.nsflxs * 5 * (size_t)(h->c.njdt + 1) * sizeof(double));
  HIPCHK(hipMemcpyAsync(h->d_stage, slabs, n6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipEventRecord(h->ev_f, h->stream));
  HIPCHK(mckpp_launch_unpack_sflux(h->d_stage, h->d_ipt, h->d_cs, h->ncol, h->npts, h->stream));
  HIPCHK(hipEventSynchronize(h->ev_f));
  return 0;
}

// mckpp_fluxes (src/mckpp_fluxes_mod.F90:35-89) on the device: eight forcing fields (npts each, 3D
// ordering) -> sflux(:,1:6,5,0) of every resident l_ocean column, plus the ntflux refresh of wXNT(:,1).
int mckpp_hip_fluxes(mckpp_hip_handle h, int ntime, const double *taux, const double *tauy, const double *swf,
                     const double *lwf, const double *lhf, const double *shf, const double *rain,
                     const double *snow, int l_rest, double flsn, double el)
{
  if (!h || !taux || !tauy || !swf || !lwf || !lhf || !shf || !rain || !snow)
    return fail("mckpp_hip_fluxes: null argument");
  if (h->ncol == 0) return 0;
  HIPCHK(hipSetDevice(h->device));
  const double *src[8] = {taux, tauy, swf, lwf, lhf, shf, rain, snow};
  const size_t n8 = (size_t)8 * h->ncol;
  if (ensure_host_f(h, n8) || ensure_stage(h, n8)) return -1;
  HIPCHK(hipEventSynchronize(h->ev_f));
  for (int m = 0; m < 8; ++m)
    for (int64_t c = 0; c < h->ncol; ++c) h->h_f[(size_t)m * h->ncol + c] = src[m][h->ipt[c]];
  HIPCHK(hipMemcpyAsync(h->d_stage, h->h_f, n8 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipEventRecord(h->ev_f, h->stream));
  mckpp_kparams p;
  fill_params(h, p, ntime, MCKPP_MODE_STEP);
  HIPCHK(mckpp_launch_fluxes(p, ntime, h->d_stage, l_rest, flsn, el, h->stream));
  return 0;
}

int mckpp_hip_bottomtemp(mckpp_hip_handle h, const double *bottom_temp)
{
  if (!h || !bottom_temp) return fail("mckpp_hip_bottomtemp: null argument");
  if (h->ncol == 0) return 0;
  if (!h->diag) return fail("mckpp_hip_bottomtemp: needs the diagnostics on (rho, cp of the last vmix)");
  HIPCHK(hipSetDevice(h->device));
  if (!h->d_ext_out[O_TINC]) {   // default-physics contexts carry no correction rows until someone needs them
    const size_t rowbytes = (size_t)h->ncol * h->ld * sizeof(double);
    for (auto &p : h->d_ext_out) { HIPCHK(hipMalloc(&p, rowbytes)); HIPCHK(hipMemsetAsync(p, 0, rowbytes, h->stream)); }
  }
  std::vector<double> bt((size_t)h->ncol);
  for (int64_t c = 0; c < h->ncol; ++c) bt[(size_t)c] = bottom_temp[h->ipt[c]];
  if (ensure_stage(h, bt.size())) return -1;
  HIPCHK(hipMemcpyAsync(h->d_stage, bt.data(), bt.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  mckpp_kparams p;
  fill_params(h, p, 0, MCKPP_MODE_STEP);
  HIPCHK(mckpp_launch_bottomtemp(p, h->d_stage, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// the caller's arrays pinned on behalf of this context go back to pageable memory (before the caller frees them)
int mckpp_hip_release_host_arrays(mckpp_hip_handle h)
{
  if (!h) return fail("null handle");
  HIPCHK(hipSetDevice(h->device));
  if (xfer_finish(h)) return -1;
  unpin_all(h);
  return 0;
}

int mckpp_hip_set_diagnostics(mckpp_hip_handle h, int on)
{
  if (!h) return fail("null handle");
  h->diag = on ? 1 : 0;
  return 0;
}

int mckpp_hip_set_solver_mode(mckpp_hip_handle h, int mode)
{
  if (!h) return fail("null handle");
  if (mode < 0 || mode > 1) return fail("mckpp_hip_set_solver_mode: unknown solver mode (0: the reference's order, 1: two-ended)");
  h->solver_mode = mode;
  return 0;
}

int mckpp_hip_get_solver_mode(mckpp_hip_handle h) { return h ? h->solver_mode : fail("null handle"); }

static void fill_params(mckpp_hip_ctx *h, mckpp_kparams &p, int ntime, int mode)
{
  memset(&p, 0, sizeof p);
  p.nz = h->nz; p.nzp1 = h->nzp1; p.ncol = (int)h->ncol; p.ld = h->ld;
  p.ntime = ntime; p.itermax = h->c.itermax; p.mode = mode; p.diag = h->diag;
  p.L_SSref = h->c.L_SSref; p.LDD = h->c.LDD; p.clim_present = h->c.clim_present;
  p.l2pre = h->l2pre; p.LRI = h->c.LRI ? 1 : 0; p.l3cap = h->l3cap; p.solver_mode = h->solver_mode;
  p.hmixtolfrac = h->c.hmixtolfrac; p.dto = h->c.dto; p.grav = h->c.grav; p.vonk = h->c.vonk; p.sice = h->c.sice;
  p.Vtc = h->Vtc; p.cg = h->cg; p.dm_nz = h->dm_nz;
  p.zm = h->d_zm; p.hm = h->d_hm; p.tri0 = h->d_tri0; p.tri1 = h->d_tri1;
  p.swfrac_tab = h->d_swfrac_tab; p.swdk_tab = h->d_swdk_tab; p.ldc = h->ldc; p.wtab = reinterpret_cast<const double *>(h->d_wtab);
  p.U = h->d_prof[P_U]; p.V = h->d_prof[P_V]; p.T = h->d_prof[P_T]; p.S = h->d_prof[P_S];
  p.Us[0] = h->d_prof[P_US0]; p.Us[1] = h->d_prof[P_US1]; p.Vs[0] = h->d_prof[P_VS0]; p.Vs[1] = h->d_prof[P_VS1];
  p.Ts[0] = h->d_prof[P_TS0]; p.Ts[1] = h->d_prof[P_TS1]; p.Ss[0] = h->d_prof[P_SS0]; p.Ss[1] = h->d_prof[P_SS1];
  p.U_init = h->d_prof[P_UINIT]; p.V_init = h->d_prof[P_VINIT];
  p.cs = h->d_cs; p.ci = h->d_ci; p.qhead = h->d_qhead; p.dbg = h->d_dbg;
  p.nsteps_launch = 1; p.done = h->d_done; p.nqueues = h->nqueues; p.qowner = h->d_qhead + 16;
  p.sync = h->d_qhead + 32; p.solo_after = h->solo_after; p.solo_limit = h->solo_limit; p.view_kmax = h->view_kmax;
  for (int i = 0; i < 16; ++i) p.xcc_queue[i] = h->xcc_queue[i];
  p.ext = h->ext_kernel ? 1 : 0;
  p.L_RELAX_SST = h->c.L_RELAX_SST; p.L_RELAX_CALCONLY = h->c.L_RELAX_CALCONLY; p.L_FCORR = h->c.L_FCORR;
  p.L_FCORR_WITHZ = h->c.L_FCORR_WITHZ; p.L_SFCORR = h->c.L_SFCORR; p.L_SFCORR_WITHZ = h->c.L_SFCORR_WITHZ;
  p.L_RELAX_SAL = h->c.L_RELAX_SAL; p.L_RELAX_OCNT = h->c.L_RELAX_OCNT; p.L_NO_FREEZE = h->c.L_NO_FREEZE;
  p.L_NO_ISOTHERM = h->c.L_NO_ISOTHERM; p.L_DAMP_CURR = h->c.L_DAMP_CURR; p.iso_bot = h->c.iso_bot;
  p.dt_uvdamp = h->c.dt_uvdamp; p.maxmodeadv = h->c.maxmodeadv; p.iso_thresh = h->c.iso_thresh;
  p.dm = h->d_dm; p.hsum = h->d_hsum; p.xs = h->d_xs; p.adv_i = h->d_adv_i; p.adv_d = h->d_adv_d;
  p.fcorr_withz = h->d_ext_in[E_FCORR_WITHZ]; p.sfcorr_withz = h->d_ext_in[E_SFCORR_WITHZ];
  p.ocnT_clim = h->d_ext_in[E_OCNT_CLIM]; p.sal_clim = h->d_ext_in[E_SAL_CLIM];
  p.tinc_fcorr = h->d_ext_out[O_TINC]; p.sinc_fcorr = h->d_ext_out[O_SINC];
  p.ocnTcorr = h->d_ext_out[O_OCNTCORR]; p.scorr = h->d_ext_out[O_SCORR];
  p.rho = h->d_diag[D_RHO]; p.cp = h->d_diag[D_CP]; p.buoy = h->d_diag[D_BUOY];
  p.talpha = h->d_diag[D_TALPHA]; p.sbeta = h->d_diag[D_SBETA];
  p.difm = h->d_diag[D_DIFM]; p.difs = h->d_diag[D_DIFS]; p.dift = h->d_diag[D_DIFT]; p.ghat = h->d_diag[D_GHAT];
  p.wU1 = h->d_diag[D_WU1]; p.wU2 = h->d_diag[D_WU2];
  p.wX1 = h->d_diag[D_WX1]; p.wX2 = h->d_diag[D_WX2]; p.wX3 = h->d_diag[D_WX3]; p.wXNT1 = h->d_diag[D_WXNT1];
  p.Rig = h->d_diag[D_RIG]; p.dbloc = h->d_diag[D_DBLOC]; p.Shsq = h->d_diag[D_SHSQ];
  p.scratch = h->d_scratch; p.scratch_doubles = h->scratch_doubles;
}

struct forced_run { int ndtocn, l_rest; double flsn, el; };

static int run(mckpp_hip_ctx *h, int ntime, int nsteps, int mode, const forced_run *forced = nullptr)
{
  if (!h) return fail("null handle");
  if (h->ncol == 0) { h->nlaunch = 0; h->timed = false; return 0; }
  if (h->ext && !h->ext_inputs_resident)
    return fail("optional-physics context: the relaxation / correction / advection inputs are not resident "
                "(after mckpp_hip_load_restart call mckpp_hip_update_ancillaries before stepping)");
  HIPCHK(hipSetDevice(h->device));
  {   // parameter block (identical for every launch of this call but ntime), from a pinned slot: no host wait
    const unsigned slot = h->params_seq++ & 1u;
    HIPCHK(hipEventSynchronize(h->ev_params[slot]));
    fill_params(h, h->h_params[slot], ntime, mode);
    HIPCHK(hipMemcpyAsync(h->d_params, &h->h_params[slot], sizeof(mckpp_kparams), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipEventRecord(h->ev_params[slot], h->stream));
  }
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  // Several steps of constant forcing (mckpp_hip_step with nsteps > 1): ONE launch takes every column through all of
  // them - ncol x nsteps tickets, a column's step waiting only for that column's previous step (k_column_ps, M0) -
  // instead of a launch per step, each of which would wait for its slowest column.  Same results bit for bit (the
  // columns are independent; every step still stores its outputs and diagnostics).  The forced run too: a step
  // that is a flux update assembles its column's forcing from the resident record itself.  Not for a step at ntime = 0.
  if (mode == MCKPP_MODE_STEP && nsteps > 1 && ntime >= 1 && h->multistep) {
    if (!h->d_done) HIPCHK(hipMalloc(&h->d_done, 2 * (size_t)h->ncol * sizeof(int)));   // done[ncol], then the steps started (k_column_ps, M0)
    const int per_launch = (int)std::max<int64_t>(1, ((int64_t)1 << 30) / h->ncol);   // tickets are 32-bit (per queue: fewer still)
    for (int i = 0; i < nsteps; i += per_launch) {
      const int n = nsteps - i < per_launch ? nsteps - i : per_launch;
      const unsigned slot = h->params_seq++ & 1u;   // this launch's parameter block (its step count differs from the call's first)
      HIPCHK(hipEventSynchronize(h->ev_params[slot]));
      fill_params(h, h->h_params[slot], ntime + i, mode);
      h->h_params[slot].nsteps_launch = n;
      if (forced) {   // the forced run: every step finds its flux record itself (k_column_ps, M0)
        mckpp_kparams &q = h->h_params[slot];
        q.series = h->d_series; q.series_rec0 = h->series_rec0; q.ndtocn = forced->ndtocn; q.l_rest = forced->l_rest;
        q.flsn = forced->flsn; q.el = forced->el;
      }
      HIPCHK(hipMemcpyAsync(h->d_params, &h->h_params[slot], sizeof(mckpp_kparams), hipMemcpyHostToDevice, h->stream));
      HIPCHK(hipEventRecord(h->ev_params[slot], h->stream));
      HIPCHK(hipMemsetAsync(h->d_qhead, 0, QBLOCK_INTS * sizeof(int), h->stream));
      HIPCHK(hipMemsetAsync(h->d_done, 0, 2 * (size_t)h->ncol * sizeof(int), h->stream));
      HIPCHK(mckpp_launch_column_kernel_ps(h->h_params[slot], h->d_params, h->num_cu, h->stream, &h->last_launch));
    }
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->nlaunch = nsteps;   // (mckpp_hip_last_kernel_ms: time per STEP, whatever the number of launches)
    h->nkernels = (nsteps + per_launch - 1) / per_launch;
    h->timed = true;
    return 0;
  }
  for (int i = 0; i < nsteps; ++i) {
    mckpp_kparams p;
    fill_params(h, p, ntime + i, mode);
    if (forced && (ntime + i - 1) % forced->ndtocn == 0) {   // ocean_model_3D.F90:44-48
      const int rec = (ntime + i - 1) / forced->ndtocn - h->series_rec0;
      HIPCHK(mckpp_launch_fluxes(p, ntime + i, h->d_series + (size_t)rec * 8 * (size_t)h->ncol, forced->l_rest,
                                 forced->flsn, forced->el, h->stream));
    }
    HIPCHK(hipMemsetAsync(h->d_qhead, 0, QBLOCK_INTS * sizeof(int), h->stream));
    HIPCHK(mckpp_launch_column_kernel_ps(p, h->d_params, h->num_cu, h->stream, &h->last_launch));
  }
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  h->nlaunch = nsteps;
  h->nkernels = nsteps;
  h->timed = true;
  return 0;
}

int mckpp_hip_init_ocean(mckpp_hip_handle h, int ntime) { return run(h, ntime, 1, MCKPP_MODE_INIT); }
int mckpp_hip_step(mckpp_hip_handle h, int ntime, int nsteps)
{
  if (nsteps < 0) return fail("mckpp_hip_step: nsteps=%d", nsteps);
  return run(h, ntime, nsteps, MCKPP_MODE_STEP);
}
int mckpp_hip_vmix_pass(mckpp_hip_handle h, int ntime) { return run(h, ntime, 1, MCKPP_MODE_PASS); }
int mckpp_hip_vmix_only(mckpp_hip_handle h, int ntime) { return run(h, ntime, 1, MCKPP_MODE_VMIX); }

int mckpp_hip_set_flux_series(mckpp_hip_handle h, int rec0, int nrec, const double *fields)
{
  if (!h || !fields) return fail("mckpp_hip_set_flux_series: null argument");
  if (nrec < 1 || rec0 < 0) return fail("mckpp_hip_set_flux_series: rec0=%d nrec=%d", rec0, nrec);
  if (h->npts <= 0) return fail("mckpp_hip_set_flux_series: upload the state first (the records are compacted to the resident columns)");
  HIPCHK(hipSetDevice(h->device));
  if (h->d_series) { HIPCHK(hipFree(h->d_series)); h->d_series = nullptr; h->series_nrec = 0; }
  h->series_rec0 = rec0;
  h->series_nrec = nrec;
  if (h->ncol == 0) return 0;
  const size_t n = (size_t)nrec * 8 * (size_t)h->ncol;
  std::vector<double> f(n);
  for (int r = 0; r < nrec; ++r)
    for (int m = 0; m < 8; ++m) {
      const double *src = fields + ((size_t)r * 8 + m) * (size_t)h->npts;
      double *dst = f.data() + ((size_t)r * 8 + m) * (size_t)h->ncol;
      for (int64_t c = 0; c < h->ncol; ++c) dst[c] = src[h->ipt[c]];
    }
  HIPCHK(hipMalloc(&h->d_series, n * sizeof(double)));
  HIPCHK(hipMemcpy(h->d_series, f.data(), n * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

int mckpp_hip_run_forced(mckpp_hip_handle h, int nt_first, int nsteps, int ndtocn, int l_rest, double flsn, double el)
{
  if (!h) return fail("null handle");
  if (nt_first < 1 || nsteps < 0 || ndtocn < 1)
    return fail("mckpp_hip_run_forced: nt_first=%d nsteps=%d ndtocn=%d", nt_first, nsteps, ndtocn);
  if (nsteps == 0) return 0;
  // every flux update of the span must be resident before anything is launched
  const int first_upd = (nt_first - 1 + ndtocn - 1) / ndtocn, last_upd = (nt_first + nsteps - 2) / ndtocn;
  if (first_upd <= last_upd &&
      (h->series_nrec == 0 || first_upd < h->series_rec0 || last_upd >= h->series_rec0 + h->series_nrec))
    return fail("mckpp_hip_run_forced: steps %d..%d need flux records %d..%d, resident are %d..%d", nt_first,
                nt_first + nsteps - 1, first_upd, last_upd, h->series_rec0, h->series_rec0 + h->series_nrec - 1);
  const forced_run fr{ndtocn, l_rest, flsn, el};
  return run(h, nt_first, nsteps, MCKPP_MODE_STEP, &fr);
}

int mckpp_hip_synchronize(mckpp_hip_handle h)
{
  if (!h) return fail("null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (getenv("MCKPP_LIST_DEBUG") && h->d_qhead) {   // after the last launch: the queues' heads, and (stamp builds) the stragglers' passes
    int qb[QBLOCK_INTS];
    HIPCHK(hipMemcpy(qb, h->d_qhead, sizeof qb, hipMemcpyDeviceToHost));
    fprintf(stderr, "[mckpp queues] stragglers left %d; (stamp builds: straggler passes in a view %d, beside others %d; workgroup passes with one %d | %d, their active slots %d | %d); heads",
            qb[32], qb[33], qb[34], qb[35], qb[36], qb[37], qb[38]);
    for (int q = 0; q < h->nqueues; ++q) fprintf(stderr, " %d", qb[q]);
    fprintf(stderr, "\n");
  }
  if (h->d_dbg) {   // MCKPP_STAMP=1: print and reset the per-segment cycle sums of the stamping wave of every workgroup
    unsigned long long t[32];
    HIPCHK(hipMemcpy(t, h->d_dbg, sizeof t, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(h->d_dbg, 0, sizeof t));
    if (t[31]) {
      const char *nm[23] = {"L1", "w", "M1L2", "w", "L3", "w", "M2", "w", "L4", "w", "M3", "w", "L5", "w", "L6", "w",
                            "M4back", "w", "L7", "w", "M5back", "w", "finish"};
      fprintf(stderr, "[mckpp stamps ps] wave-passes %llu; cycles per wave-pass:", t[31]);
      double tot = 0;
      for (int i = 0; i < 23; ++i) {
        if (i == 16 || i == 20) {   // the forward parts of the two sweeps have their own accumulators
          fprintf(stderr, " %s=%.0f", i == 16 ? "M4fwd" : "M5fwd", (double)t[i == 16 ? 24 : 25] / (double)t[31]);
          tot += (double)t[i == 16 ? 24 : 25];
        }
        fprintf(stderr, " %s=%.0f", nm[i], (double)t[i] / (double)t[31]);
        tot += (double)t[i];
      }
      {   // the finish round in parts (its own accumulators; "finish" above is what follows the last of them)
        const char *fn[5] = {"trap-terms", "trap-decision", "outputs", "time-level+check_profile", "refill"};
        double fin = (double)t[22];
        fprintf(stderr, " [finish:");
        for (int i = 0; i < 5; ++i) { fprintf(stderr, " %s=%.0f", fn[i], (double)t[26 + i] / (double)t[31]); fin += (double)t[26 + i]; tot += (double)t[26 + i]; }
        fprintf(stderr, " all=%.0f]", fin / (double)t[31]);
      }
      fprintf(stderr, " total=%.0f\n", tot / (double)t[31]);
    }
  }
  return 0;
}

const char *mckpp_hip_kernel_name(mckpp_hip_handle h)
{
  if (!h) return "none";
  return h->ext_kernel ? "k_column_ps<EXT>" : "k_column_ps";
}

int mckpp_hip_kernel_residency(mckpp_hip_handle h, int32_t *blocks_per_cu, int32_t *max_blocks_per_cu,
                               int32_t *threads_per_block, int64_t *lds_bytes_per_block)
{
  if (!h) return fail("null handle");
  if (h->last_launch.threads == 0)
    return fail("mckpp_hip_kernel_residency: no cooperative-kernel launch yet");
  if (blocks_per_cu) *blocks_per_cu = (h->last_launch.nblocks + h->num_cu - 1) / h->num_cu;
  if (max_blocks_per_cu) *max_blocks_per_cu = h->last_launch.max_blocks_per_cu;
  if (threads_per_block) *threads_per_block = h->last_launch.threads;
  if (lds_bytes_per_block) *lds_bytes_per_block = (int64_t)h->last_launch.lds_bytes;
  return 0;
}

int mckpp_hip_last_kernel_ms(mckpp_hip_handle h, double *ms, int32_t *nlaunch)
{
  if (!h) return fail("null handle");
  if (!h->timed) { if (ms) *ms = 0.0; if (nlaunch) *nlaunch = 0; return 0; }
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipEventSynchronize(h->ev1));
  float t = 0.f;
  HIPCHK(hipEventElapsedTime(&t, h->ev0, h->ev1));
  if (ms) *ms = (double)t;
  if (nlaunch) *nlaunch = h->nlaunch;
  return 0;
}

int32_t mckpp_hip_last_launch_count(mckpp_hip_handle h) { return h && h->timed ? h->nkernels : 0; }

// What a download moves, as a list: every row field of `mask` whose host pointer is set - the device rows it
// comes from (of THIS context), the offset of its first element in them, its number of levels, and where it goes
// in the caller's arrays.  The list depends on the mask and the pointers only, so the contexts of a multi-device
// handle produce lists that correspond entry by entry.
namespace {
struct row_xfer { const double *dev; int src_off, nlev; double *host; const double *whole; size_t whole_elems; };
}

static void download_plan(mckpp_hip_ctx *h, const mckpp_state_ptrs_c *s, uint32_t mask, std::vector<row_xfer> &plan)
{
  const int nzp1 = h->nzp1, nz = h->nz;
  const size_t slab = (size_t)h->npts * nzp1;
  const double *whole = nullptr;   // the array the next slabs belong to, and its size: what gets pinned
  size_t whole_elems = 0;
  auto add = [&](const double *dev, int off, int nlev, double *host) { plan.push_back({dev, off, nlev, host, whole, whole_elems}); };
  if ((mask & MCKPP_F_PROFILES)) {
    if (s->U) { whole = s->U; whole_elems = 2 * slab; add(h->d_prof[P_U], 0, nzp1, s->U); add(h->d_prof[P_V], 0, nzp1, s->U + slab); }
    if (s->X) { whole = s->X; whole_elems = 2 * slab; add(h->d_prof[P_T], 0, nzp1, s->X); add(h->d_prof[P_S], 0, nzp1, s->X + slab); }
  }
  if ((mask & MCKPP_F_SAVED)) {
    if (s->Us) {
      whole = s->Us; whole_elems = 4 * slab;
      add(h->d_prof[P_US0], 0, nzp1, s->Us + 0 * slab); add(h->d_prof[P_VS0], 0, nzp1, s->Us + 1 * slab);
      add(h->d_prof[P_US1], 0, nzp1, s->Us + 2 * slab); add(h->d_prof[P_VS1], 0, nzp1, s->Us + 3 * slab);
    }
    if (s->Xs) {
      whole = s->Xs; whole_elems = 4 * slab;
      add(h->d_prof[P_TS0], 0, nzp1, s->Xs + 0 * slab); add(h->d_prof[P_SS0], 0, nzp1, s->Xs + 1 * slab);
      add(h->d_prof[P_TS1], 0, nzp1, s->Xs + 2 * slab); add(h->d_prof[P_SS1], 0, nzp1, s->Xs + 3 * slab);
    }
  }
  whole = nullptr; whole_elems = 0;
  if ((mask & MCKPP_F_DIAG)) {
    const int n1 = h->c.nztmax + 1;        // extent of (0:nztmax)
    const size_t s1 = (size_t)h->npts * n1;
    if (s->rho) add(h->d_diag[D_RHO], 0, nzp1 + 1, s->rho);
    if (s->cp) add(h->d_diag[D_CP], 0, nzp1 + 1, s->cp);
    if (s->buoy) add(h->d_diag[D_BUOY], 1, nzp1, s->buoy);
    if (s->difm) add(h->d_diag[D_DIFM], 0, nzp1 + 1, s->difm);
    if (s->difs) add(h->d_diag[D_DIFS], 0, nzp1 + 1, s->difs);
    if (s->dift) add(h->d_diag[D_DIFT], 0, nzp1 + 1, s->dift);
    if (s->ghat) add(h->d_diag[D_GHAT], 1, nz, s->ghat);
    if (s->wU) { whole = s->wU; whole_elems = 2 * s1; add(h->d_diag[D_WU1], 0, nz + 1, s->wU); add(h->d_diag[D_WU2], 0, nz + 1, s->wU + s1); }
    if (s->wX) {
      whole = s->wX; whole_elems = 3 * s1;
      add(h->d_diag[D_WX1], 0, nz + 1, s->wX); add(h->d_diag[D_WX2], 0, nz + 1, s->wX + s1);
      add(h->d_diag[D_WX3], 0, nz + 1, s->wX + 2 * s1);
    }
    whole = nullptr; whole_elems = 0;
    if (s->wXNT) add(h->d_diag[D_WXNT1], 0, nz + 1, s->wXNT);
    if (s->Rig) add(h->d_diag[D_RIG], 1, nz, s->Rig);
    if (s->Shsq) add(h->d_diag[D_SHSQ], 1, nz, s->Shsq);
    if (s->dbloc) add(h->d_diag[D_DBLOC], 1, nz, s->dbloc);
    if (h->d_ext_out[O_TINC]) {
      if (s->tinc_fcorr) add(h->d_ext_out[O_TINC], 1, nzp1, s->tinc_fcorr);
      if (s->sinc_fcorr) add(h->d_ext_out[O_SINC], 1, nzp1, s->sinc_fcorr);
      if (s->ocnTcorr) add(h->d_ext_out[O_OCNTCORR], 1, nzp1, s->ocnTcorr);
      if (s->scorr) add(h->d_ext_out[O_SCORR], 1, nzp1, s->scorr);
    }
  }
}

// The per-column records of one context -> the caller's (npts) arrays: one transfer of each record array into
// pinned memory, then a host loop over the context's columns.
static int download_records(mckpp_hip_ctx *h, mckpp_state_ptrs_c *s, uint32_t mask)
{
  const int nzp1 = h->nzp1, nz = h->nz;
  const int64_t npts = h->npts, ncol = h->ncol;
  const bool want_tabs = (mask & MCKPP_F_DIAG) && (s->swfrac || s->swdk_opt);
  if (!(mask & (MCKPP_F_SAVED | MCKPP_F_SCALARS)) && !want_tabs) return 0;
  if (ncol == npts && !want_tabs) {
    // every grid point is a resident column: the wanted record slots are packed into (npts) slabs on the device and
    // land in the caller's arrays as they are - no host loop, half the bytes
    struct dst_d { double *p; };
    struct dst_i { int32_t *p; };
    mckpp_pack_list l{};
    dst_d dd[MCKPP_CS];
    dst_i di[MCKPP_CI];
    auto addd = [&](double *p, int slot) { if (p) { dd[l.nd].p = p; l.dslot[l.nd++] = slot; } };
    auto addi = [&](int32_t *p, int slot) { if (p) { di[l.ni].p = p; l.islot[l.ni++] = slot; } };
    if (mask & MCKPP_F_SAVED) {
      if (s->hmixd) { addd(s->hmixd, CS_HMIXD0); addd(s->hmixd + npts, CS_HMIXD1); }
      addi(s->old, CI_OLD); addi(s->new_, CI_NEW);
    }
    if (mask & MCKPP_F_SCALARS) {
      addd(s->hmix, CS_HMIX); addd(s->kmix, CS_KMIX); addd(s->Tref, CS_TREF); addd(s->uref, CS_UREF); addd(s->vref, CS_VREF);
      addd(s->Ssurf, CS_SSURF); addd(s->reset_flag, CS_RESET); addd(s->dampu_flag, CS_DAMPU); addd(s->dampv_flag, CS_DAMPV);
      addd(s->freeze_flag, CS_FREEZE);
      if (h->ext) addd(s->fcorr, CS_FCORR);
      addi(s->l_initflag, CI_INITFLAG);
      if (s->sflux)
        for (int m = 0; m < 6; ++m) addd(s->sflux + npts * m + npts * (int64_t)h->c.nsflxs * 4, CS_SFLUX1 + m);
    }
    // One device block (the double slabs, then the int slabs right behind them) and ONE transfer of it into a pinned
    // block of the library's own; the caller's arrays - a dozen of them, anywhere - are filled from there by the host
    // threads.  (A transfer per array costs a round trip each: 1.8 ms of a 4.7 ms drop-in step on a link that has
    // been idle, r03.)
    const size_t pack_bytes = (size_t)npts * (MCKPP_CS * sizeof(double) + MCKPP_CI * sizeof(int));
    if (!h->d_pack) HIPCHK(hipMalloc(&h->d_pack, pack_bytes));
    if (!h->h_pack) HIPCHK(hipHostMalloc(&h->h_pack, pack_bytes, hipHostMallocDefault));
    int *d_ipack = reinterpret_cast<int *>(h->d_pack + (size_t)npts * l.nd);
    HIPCHK(mckpp_launch_pack_records(h->d_cs, h->d_ci, h->d_ipt, ncol, npts, l, h->d_pack, d_ipack, h->stream));
    const size_t used = (size_t)npts * (l.nd * sizeof(double) + l.ni * sizeof(int));
    if (used) HIPCHK(hipMemcpyAsync(h->h_pack, h->d_pack, used, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const double *hd = h->h_pack;
    const int *hi = reinterpret_cast<const int *>(h->h_pack + (size_t)npts * l.nd);
    for_columns(npts, [&](int64_t c0, int64_t c1) {
      for (int j = 0; j < l.nd; ++j) memcpy(dd[j].p + c0, hd + (size_t)npts * j + c0, (size_t)(c1 - c0) * sizeof(double));
      for (int j = 0; j < l.ni; ++j) memcpy(di[j].p + c0, hi + (size_t)npts * j + c0, (size_t)(c1 - c0) * sizeof(int));
    });
    return 0;
  }
  if (mask & (MCKPP_F_SAVED | MCKPP_F_SCALARS))
    HIPCHK(hipMemcpyAsync(h->h_cs, h->d_cs, (size_t)ncol * MCKPP_CS * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(h->h_ci, h->d_ci, (size_t)ncol * MCKPP_CI * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  const double *cs = h->h_cs;
  const int *ci = h->h_ci;
  const bool saved = mask & MCKPP_F_SAVED, scal = mask & MCKPP_F_SCALARS;
  const bool fc = scal && s->fcorr && h->ext;
  double *sfl = (scal && s->sflux) ? s->sflux + npts * (int64_t)h->c.nsflxs * 4 : nullptr;   // sflux(:,1:6,5,0)
  const int *ipt = h->ipt.data();
  for_columns(ncol, [&](int64_t c0, int64_t c1) {
    for (int64_t c = c0; c < c1; ++c) {
      const int64_t i = ipt[c];
      const double *r = &cs[(size_t)c * MCKPP_CS];
      const int *q = &ci[(size_t)c * MCKPP_CI];
      if (saved) {
        if (s->hmixd) { s->hmixd[i] = r[CS_HMIXD0]; s->hmixd[i + npts] = r[CS_HMIXD1]; }
        if (s->old) s->old[i] = q[CI_OLD];
        if (s->new_) s->new_[i] = q[CI_NEW];
      }
      if (scal) {
        if (s->hmix) s->hmix[i] = r[CS_HMIX];
        if (s->kmix) s->kmix[i] = r[CS_KMIX];
        if (s->Tref) s->Tref[i] = r[CS_TREF];
        if (s->uref) s->uref[i] = r[CS_UREF];
        if (s->vref) s->vref[i] = r[CS_VREF];
        if (s->Ssurf) s->Ssurf[i] = r[CS_SSURF];
        if (s->reset_flag) s->reset_flag[i] = r[CS_RESET];
        if (s->dampu_flag) s->dampu_flag[i] = r[CS_DAMPU];
        if (s->dampv_flag) s->dampv_flag[i] = r[CS_DAMPV];
        if (s->freeze_flag) s->freeze_flag[i] = r[CS_FREEZE];
        if (s->l_initflag) s->l_initflag[i] = q[CI_INITFLAG];
        if (fc) s->fcorr[i] = r[CS_FCORR];
        if (sfl) for (int m = 0; m < 6; ++m) sfl[i + npts * m] = r[CS_SFLUX1 + m];
      }
    }
  });
  if (want_tabs) {   // level by level within a block of columns: every destination slab is written front to back
    const double *tf = h->h_swfrac_tab.data(), *tk = h->h_swdk_tab.data();
    const size_t ldc = (size_t)h->ldc;
    for_columns(ncol, [&](int64_t c0, int64_t c1) {
      if (s->swfrac)
        for (int l = 1; l <= nzp1; ++l) {
          double *dst = s->swfrac + npts * (int64_t)(l - 1);
          for (int64_t c = c0; c < c1; ++c) dst[ipt[c]] = tf[(size_t)ci[(size_t)c * MCKPP_CI + CI_JERLOV] * ldc + l];
        }
      if (s->swdk_opt)
        for (int k = 0; k <= nz; ++k) {
          double *dst = s->swdk_opt + npts * (int64_t)k;
          for (int64_t c = c0; c < c1; ++c) dst[ipt[c]] = tk[(size_t)ci[(size_t)c * MCKPP_CI + CI_JERLOV] * ldc + k];
        }
    });
  }
  return 0;
}

int mckpp_hip_download(mckpp_hip_handle h, mckpp_state_ptrs_c *s, uint32_t mask)
{
  if (!h || !s) return fail("mckpp_hip_download: null argument");
  if (s->npts != h->npts) return fail("mckpp_hip_download: npts=%lld but %lld were uploaded", (long long)s->npts, (long long)h->npts);
  if (h->ncol == 0) return 0;
  HIPCHK(hipSetDevice(h->device));
  std::vector<row_xfer> plan;
  download_plan(h, s, mask, plan);
  for (const row_xfer &x : plan)
    if (down_rows(h, x.dev, h->ld, x.src_off, x.nlev, x.host, x.whole, x.whole_elems)) return -1;
  if (download_records(h, s, mask)) return -1;   // (waits for the context's stream: the last step has finished)
  return xfer_finish(h);
}

// ---------------------------------------------------------------------------
// Restart set.  The reference writes U,V,T,S,CP,rho,hmix,kmix,Sref,SSref,Ssurf,
// Tref,old,new,Us,Vs,Ts,Ss,hmixd through XIOS (src/mckpp_xios_io.F90:368-387, 413-431) and reads
// them back at :436-465; here the same set (the device-resident state) goes to a flat binary file.
// ---------------------------------------------------------------------------
namespace {
struct restart_header {
  char magic[8];
  int32_t version, nz, ld, cs, ci, nprof;
  int64_t npts, ncol;
};
const char kRestartMagic[8] = {'M', 'C', 'K', 'P', 'P', 'R', 'S', '1'};
}  // namespace

int mckpp_hip_save_restart(mckpp_hip_handle h, const char *path)
{
  if (!h || !path) return fail("mckpp_hip_save_restart: null argument");
  if (h->ncol <= 0) return fail("mckpp_hip_save_restart: no resident columns");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  FILE *f = fopen(path, "wb");
  if (!f) return fail("mckpp_hip_save_restart: cannot open %s", path);
  restart_header hd{};
  memcpy(hd.magic, kRestartMagic, 8);
  hd.version = 1; hd.nz = h->nz; hd.ld = h->ld; hd.cs = MCKPP_CS; hd.ci = MCKPP_CI; hd.nprof = P_COUNT + 2;
  hd.npts = h->npts; hd.ncol = h->ncol;
  bool ok = fwrite(&hd, sizeof hd, 1, f) == 1;
  ok = ok && fwrite(h->ipt.data(), sizeof(int), (size_t)h->ncol, f) == (size_t)h->ncol;
  const size_t rowelems = (size_t)h->ncol * h->ld;
  std::vector<double> buf(rowelems);
  auto dump = [&](const double *d, size_t n) -> int {
    if (n > buf.size()) buf.resize(n);
    if (hipMemcpy(buf.data(), d, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return fwrite(buf.data(), sizeof(double), n, f) == n ? 0 : -1;
  };
  for (int i = 0; i < P_COUNT && ok; ++i) ok = dump(h->d_prof[i], rowelems) == 0;
  ok = ok && dump(h->d_diag[D_CP], rowelems) == 0 && dump(h->d_diag[D_RHO], rowelems) == 0;
  ok = ok && dump(h->d_cs, (size_t)h->ncol * MCKPP_CS) == 0;
  std::vector<int> ci((size_t)h->ncol * MCKPP_CI);
  ok = ok && hipMemcpy(ci.data(), h->d_ci, ci.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
  ok = ok && fwrite(ci.data(), sizeof(int), ci.size(), f) == ci.size();
  ok = (fclose(f) == 0) && ok;
  if (!ok) return fail("mckpp_hip_save_restart: write to %s failed", path);
  return 0;
}

int mckpp_hip_load_restart(mckpp_hip_handle h, const char *path)
{
  if (!h || !path) return fail("mckpp_hip_load_restart: null argument");
  HIPCHK(hipSetDevice(h->device));
  // The whole file is read and checked on the host first; the resident state is replaced only
  // once everything is known to be there and consistent.
  struct closer { FILE *f; ~closer() { if (f) fclose(f); } } fc{fopen(path, "rb")};
  FILE *f = fc.f;
  if (!f) return fail("mckpp_hip_load_restart: cannot open %s", path);
  restart_header hd{};
  if (fread(&hd, sizeof hd, 1, f) != 1 || memcmp(hd.magic, kRestartMagic, 8) != 0 || hd.version != 1)
    return fail("mckpp_hip_load_restart: %s is not a restart file of this library", path);
  if (hd.nz != h->nz || hd.ld != h->ld || hd.cs != MCKPP_CS || hd.ci != MCKPP_CI || hd.nprof != P_COUNT + 2 ||
      hd.ncol <= 0 || hd.ncol > hd.npts)
    return fail("mckpp_hip_load_restart: %s was written for nz=%d (context has nz=%d) or another layout", path,
                hd.nz, h->nz);
  const size_t ncol = (size_t)hd.ncol, rowelems = ncol * h->ld;
  std::vector<int> ipt(ncol), ci(ncol * MCKPP_CI);
  std::vector<double> rows((size_t)(P_COUNT + 2) * rowelems), cs(ncol * MCKPP_CS);
  bool ok = fread(ipt.data(), sizeof(int), ipt.size(), f) == ipt.size();
  ok = ok && fread(rows.data(), sizeof(double), rows.size(), f) == rows.size();
  ok = ok && fread(cs.data(), sizeof(double), cs.size(), f) == cs.size();
  ok = ok && fread(ci.data(), sizeof(int), ci.size(), f) == ci.size();
  if (!ok) return fail("mckpp_hip_load_restart: %s is truncated or unreadable", path);
  for (size_t c = 0; c < ncol; ++c) {   // the column map scatters into (npts) arrays at download
    if (ipt[c] < 0 || ipt[c] >= hd.npts || (c > 0 && ipt[c] <= ipt[c - 1]))
      return fail("mckpp_hip_load_restart: %s has a corrupt column map (entry %zu = %d, npts = %lld)", path, c,
                  ipt[c], (long long)hd.npts);
    const int jw = ci[c * MCKPP_CI + CI_JERLOV];
    if (jw < 1 || jw > 5) return fail("mckpp_hip_load_restart: %s: jerlov=%d in column %zu", path, jw, c);
  }
  const bool same_shape = hd.ncol == h->ncol && hd.npts == h->npts;
  if (!same_shape) {
    if (alloc_state(h, hd.npts, hd.ncol)) return -1;
  } else if (ipt != h->ipt) {
    if (h->d_series) {   // resident flux records were compacted with the previous land mask
      hipFree(h->d_series);
      h->d_series = nullptr;
      h->series_nrec = 0;
    }
    h->ext_inputs_resident = false;   // and so were the optional-physics inputs
  }
  h->ipt = ipt;
  HIPCHK(hipMemcpy(h->d_ipt, ipt.data(), ipt.size() * sizeof(int), hipMemcpyHostToDevice));
  for (int i = 0; i < P_COUNT; ++i)
    HIPCHK(hipMemcpy(h->d_prof[i], rows.data() + (size_t)i * rowelems, rowelems * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_diag[D_CP], rows.data() + (size_t)P_COUNT * rowelems, rowelems * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_diag[D_RHO], rows.data() + (size_t)(P_COUNT + 1) * rowelems, rowelems * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_cs, cs.data(), cs.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_ci, ci.data(), ci.size() * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}

// Re-upload of what the host rewrites between steps for the optional physics
// (mckpp_boundary_update, src/mckpp_ocean_model_3D.F90:51-55): SST0 / relaxation times, flux
// corrections, climatologies, prescribed advection.  Prognostic state is not touched.
int mckpp_hip_update_ancillaries(mckpp_hip_handle h, const mckpp_state_ptrs_c *s)
{
  if (!h || !s) return fail("mckpp_hip_update_ancillaries: null argument");
  if (h->npts <= 0) return fail("mckpp_hip_update_ancillaries: no resident state (upload or load_restart first)");
  if (s->npts != h->npts) return fail("mckpp_hip_update_ancillaries: npts=%lld but %lld are resident", (long long)s->npts, (long long)h->npts);
  if (!h->ext || h->ncol == 0) return 0;   // the default physics reads none of them
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  return upload_ancillaries(h, s, "mckpp_hip_update_ancillaries");
}

// ---------------------------------------------------------------------------
// Output-window reductions
// ---------------------------------------------------------------------------
namespace {
// Where an output field lives on the device.  3-D fields give nzp1 values per column (the vertical axes of
// src/mckpp_xios_io.F90:96-176: levels 1..nzp1, or interfaces 0..nz for fluxes and diffusivities, whose
// shifted copies temp_2d(:,1)=0, temp_2d(:,2:NZP1)=dif*(:,1:NZ) are exactly dif*(0:nz)); 2-D fields one.
struct out_desc { const double *src; int ld, off, nlev, add_sref; };

int out_field(mckpp_hip_ctx *h, int f, out_desc &d)
{
  const int nzp1 = h->nzp1;
  auto prof = [&](int p) { d = {h->d_prof[p], h->ld, 0, nzp1, 0}; return 0; };             // element j <-> level j+1
  auto diag = [&](int q, int off) { d = {h->d_diag[q], h->ld, off, nzp1, 0}; return 0; };   // element k <-> index k
  auto corr = [&](int o) {
    if (!h->d_ext_out[o]) return fail("mckpp_hip_window: field %d needs the correction rows (optional physics or bottomtemp)", f);
    d = {h->d_ext_out[o], h->ld, 1, nzp1, 0};
    return 0;
  };
  auto scal = [&](int slot) { d = {h->d_cs, MCKPP_CS, slot, 1, 0}; return 0; };
  switch (f) {
    case MCKPP_OUT_U: return prof(P_U);
    case MCKPP_OUT_V: return prof(P_V);
    case MCKPP_OUT_T: return prof(P_T);
    case MCKPP_OUT_S_ANOM: return prof(P_S);
    case MCKPP_OUT_HMIX: return scal(CS_HMIX);
    case MCKPP_OUT_S: prof(P_S); d.add_sref = 1; return 0;
    case MCKPP_OUT_B: return diag(D_BUOY, 1);
    case MCKPP_OUT_WU: return diag(D_WU1, 0);
    case MCKPP_OUT_WV: return diag(D_WU2, 0);
    case MCKPP_OUT_WT: return diag(D_WX1, 0);
    case MCKPP_OUT_WS: return diag(D_WX2, 0);
    case MCKPP_OUT_WB: return diag(D_WX3, 0);
    case MCKPP_OUT_WTNT: return diag(D_WXNT1, 0);
    case MCKPP_OUT_DIFM: return diag(D_DIFM, 0);
    case MCKPP_OUT_DIFT: return diag(D_DIFT, 0);
    case MCKPP_OUT_DIFS: return diag(D_DIFS, 0);
    case MCKPP_OUT_RHO: return diag(D_RHO, 1);
    case MCKPP_OUT_CP: return diag(D_CP, 1);
    case MCKPP_OUT_SCORR: return corr(O_SCORR);
    case MCKPP_OUT_RIG: return diag(D_RIG, 1);
    case MCKPP_OUT_DBLOC: return diag(D_DBLOC, 1);
    case MCKPP_OUT_SHSQ: return diag(D_SHSQ, 1);
    case MCKPP_OUT_TINC_FCORR: return corr(O_TINC);
    case MCKPP_OUT_FCORR_Z: return corr(O_OCNTCORR);
    case MCKPP_OUT_SINC_FCORR: return corr(O_SINC);
    case MCKPP_OUT_FCORR: return scal(CS_FCORR);
    case MCKPP_OUT_TAUX_IN: return scal(CS_SFLUX1);
    case MCKPP_OUT_TAUY_IN: return scal(CS_SFLUX2);
    case MCKPP_OUT_SOLAR_IN: return scal(CS_SFLUX3);
    case MCKPP_OUT_NSOLAR_IN: return scal(CS_SFLUX4);
    case MCKPP_OUT_PMINUSE_IN: return scal(CS_SFLUX6);
    case MCKPP_OUT_FREEZE_FLAG: return scal(CS_FREEZE);
    case MCKPP_OUT_COMP_FLAG: return scal(CS_RESET);
    case MCKPP_OUT_DAMPU_FLAG: return scal(CS_DAMPU);
    case MCKPP_OUT_DAMPV_FLAG: return scal(CS_DAMPV);
    default: return fail("mckpp_hip_window: unknown output field %d", f);
  }
}
}  // namespace

int mckpp_hip_window_select(mckpp_hip_handle h, const int32_t *fields, int32_t nfields)
{
  if (!h || (nfields > 0 && !fields) || nfields < 0) return fail("mckpp_hip_window_select: bad argument");
  for (int i = 0; i < nfields; ++i)
    if (fields[i] < 0 || fields[i] >= MCKPP_OUT_COUNT) return fail("mckpp_hip_window_select: unknown output field %d", fields[i]);
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (auto *p : h->d_wacc) if (p) hipFree(p);
  h->d_wacc.clear();
  h->wsel.assign(fields, fields + nfields);
  h->window_count = 0;
  return 0;
}

int mckpp_hip_window_reset(mckpp_hip_handle h)
{
  if (!h) return fail("null handle");
  h->window_count = 0;
  return 0;
}

int mckpp_hip_window_accumulate(mckpp_hip_handle h)
{
  if (!h) return fail("null handle");
  if (h->ncol == 0) return 0;
  HIPCHK(hipSetDevice(h->device));
  if (h->d_wacc.size() != h->wsel.size()) h->d_wacc.assign(h->wsel.size(), nullptr);
  for (size_t i = 0; i < h->wsel.size(); ++i) {
    out_desc d;
    if (out_field(h, h->wsel[i], d)) return -1;
    if ((h->wsel[i] >= MCKPP_OUT_B && h->wsel[i] <= MCKPP_OUT_SINC_FCORR) && !h->diag)
      return fail("mckpp_hip_window_accumulate: field %d is a diagnostic, and diagnostics are switched off", h->wsel[i]);
    const int ld_out = d.nlev == 1 ? 1 : h->ld;
    const size_t n = (size_t)h->ncol * ld_out;
    if (!h->d_wacc[i]) HIPCHK(hipMalloc(&h->d_wacc[i], 3 * n * sizeof(double)));
    HIPCHK(mckpp_launch_out_sample(d.src, d.ld, d.off, h->d_cs, d.add_sref, h->ncol, d.nlev, ld_out, h->d_wacc[i],
                                   h->d_wacc[i] + n, h->d_wacc[i] + 2 * n, h->window_count == 0, nullptr, h->stream));
  }
  h->window_count += 1;
  return 0;
}

// The reduced (or sampled) field of this context's columns, compacted, left in device memory: `src` rows of
// `ld_out` doubles, `nlev` of them meaningful.  (The mean is formed into the context's staging buffer.)
static int window_prepare(mckpp_hip_ctx *h, int field, int op, const double **src_out, int *ld_out_, int *nlev_)
{
  if (op < 0 || op > 3) return fail("mckpp_hip_window_fetch: op %d (0 mean, 1 min, 2 max, 3 instant)", op);
  out_desc d;
  if (out_field(h, field, d)) return -1;
  const int ld_out = d.nlev == 1 ? 1 : h->ld;
  *ld_out_ = ld_out;
  *nlev_ = d.nlev;
  *src_out = nullptr;
  if (h->ncol == 0) return 0;
  HIPCHK(hipSetDevice(h->device));
  const size_t n = (size_t)h->ncol * ld_out;
  if (ensure_stage(h, n)) return -1;
  double *tmp = h->d_stage;
  const double *src = nullptr;
  if (op == 3) {   // the field as it stands (XIOS operation "instant" at the output step)
    HIPCHK(mckpp_launch_out_sample(d.src, d.ld, d.off, h->d_cs, d.add_sref, h->ncol, d.nlev, ld_out, nullptr, nullptr,
                                   nullptr, 0, tmp, h->stream));
    src = tmp;
  } else {
    size_t i = 0;
    while (i < h->wsel.size() && h->wsel[i] != field) ++i;
    if (i == h->wsel.size()) return fail("mckpp_hip_window_fetch: field %d is not among the selected window fields", field);
    if (h->window_count == 0 || i >= h->d_wacc.size() || !h->d_wacc[i]) return fail("mckpp_hip_window_fetch: empty window");
    src = h->d_wacc[i] + (size_t)op * n;
    if (op == 0) {
      HIPCHK(mckpp_launch_window_mean(src, tmp, n, (double)h->window_count, h->stream));
      src = tmp;
    }
  }
  *src_out = src;
  return 0;
}

int mckpp_hip_window_fetch(mckpp_hip_handle h, int field, int op, double *out)
{
  if (!h || !out) return fail("mckpp_hip_window_fetch: null argument");
  const double *src = nullptr;
  int ld_out = 0, nlev = 0;
  if (window_prepare(h, field, op, &src, &ld_out, &nlev)) return -1;
  if (h->ncol == 0) return 0;
  if (down_rows(h, src, ld_out, 0, nlev, out)) return -1;
  return xfer_finish(h);
}

int mckpp_hip_status(mckpp_hip_handle h, int32_t *per_col, int64_t *n_flagged, int32_t *npasses)
{
  if (!h) return fail("null handle");
  if (per_col) for (int64_t i = 0; i < h->npts; ++i) per_col[i] = 0;
  if (npasses) for (int64_t i = 0; i < h->npts; ++i) npasses[i] = 0;
  int64_t nf = 0;
  if (h->ncol > 0) {
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    // (the whole records, one contiguous copy; a 2-D copy of the two ints per record is no faster at 1e5 columns)
    std::vector<int> ci((size_t)h->ncol * MCKPP_CI);
    HIPCHK(hipMemcpy(ci.data(), h->d_ci, ci.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < h->ncol; ++c) {
      const int st = ci[(size_t)c * MCKPP_CI + CI_STATUS];
      if (st) ++nf;
      if (per_col) per_col[h->ipt[c]] = st;
      if (npasses) npasses[h->ipt[c]] = ci[(size_t)c * MCKPP_CI + CI_NPASS];
    }
  }
  if (n_flagged) *n_flagged = nf;
  return 0;
}

int mckpp_hip_eos_batch(mckpp_hip_handle h, int64_t n, const double *s, const double *t, const double *p,
                        double *alpha, double *beta, double *sig0, double *cp)
{
  if (!h) return fail("null handle");
  if (n <= 0) return 0;
  HIPCHK(hipSetDevice(h->device));
  double *d = nullptr;
  const size_t nb = (size_t)n * sizeof(double);
  HIPCHK(hipMalloc(&d, 7 * nb));
  HIPCHK(hipMemcpy(d, s, nb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + n, t, nb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + 2 * n, p, nb, hipMemcpyHostToDevice));
  hipError_t e = mckpp_launch_eos_batch(n, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, d + 5 * n, d + 6 * n, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(alpha, d + 3 * n, nb, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(beta, d + 4 * n, nb, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(sig0, d + 5 * n, nb, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(cp, d + 6 * n, nb, hipMemcpyDeviceToHost);
  hipFree(d);
  HIPCHK(e);
  return 0;
}

int mckpp_hip_div_batch(mckpp_hip_handle h, int64_t n, const double *num, const double *den, double *q4)
{
  if (!h) return fail("null handle");
  if (n <= 0) return 0;
  HIPCHK(hipSetDevice(h->device));
  double *d = nullptr;
  const size_t nb = (size_t)n * sizeof(double);
  HIPCHK(hipMalloc(&d, 6 * nb));
  hipError_t e = hipMemcpy(d, num, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + n, den, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = mckpp_launch_div_batch(n, d, d + n, d + 2 * n, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(q4, d + 2 * n, 4 * nb, hipMemcpyDeviceToHost);
  hipFree(d);
  HIPCHK(e);
  return 0;
}

int mckpp_hip_exp_batch(mckpp_hip_handle h, int64_t n, const double *x, double *y)
{
  if (!h) return fail("null handle");
  if (n <= 0) return 0;
  HIPCHK(hipSetDevice(h->device));
  double *d = nullptr;
  const size_t nb = (size_t)n * sizeof(double);
  HIPCHK(hipMalloc(&d, 2 * nb));
  hipError_t e = hipMemcpy(d, x, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = mckpp_launch_exp_batch(n, d, d + n, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(y, d + n, nb, hipMemcpyDeviceToHost);
  hipFree(d);
  HIPCHK(e);
  return 0;
}


// ---------------------------------------------------------------------------
// Several GPUs behind one handle: the run_physics columns are dealt round-robin to the
// devices (pass counts vary along a latitude band, so a contiguous split would unbalance the shards),
// every device holds the constants, a step launches on every device's stream before anything waits,
// and nothing crosses devices inside a step.  The only exchange is the output gather to one root
// device over the GPU interconnect (xGMI peer copies), relayout there, one transfer to the host.
// ---------------------------------------------------------------------------
struct mckpp_hip_multi {
  std::vector<mckpp_hip_ctx *> ctx;
  int64_t npts = 0;
  std::vector<std::vector<int32_t>> mask;   // run_physics of each shard
  // root-side resources of the gather (owned by the root device of the last gather): the 3-D image, the staging
  // area the shards' rows arrive in, every shard's column map, one stream per shard and the events that order
  // a shard's copy behind its owner's stream and the final transfer behind all shards
  int root = -1;
  double *d_out = nullptr, *d_stage = nullptr;
  int *d_gipt = nullptr;
  size_t out_elems = 0, stage_elems = 0, gipt_elems = 0;
  bool gipt_valid = false;                  // d_gipt holds the maps of the current upload
  std::vector<hipStream_t> gstream;         // [ndev], on the root device
  std::vector<hipEvent_t> ev_owner;         // [ndev], each on its shard's device
  std::vector<hipEvent_t> ev_done;          // [ndev], on the root device
  hipEvent_t ev_init = nullptr, ev_gcopy = nullptr;   // on the root device: 3-D image ready for the shards / delivered to the host
};

// run_physics mask of shard `dev` of `ndev`: the j-th ocean point (in ipt order) goes to shard j mod ndev
int64_t mckpp_host_shard_mask(int64_t npts, const int32_t *run_physics, int32_t ndev, int32_t dev, int32_t *mask_out)
{
  if (npts < 0 || ndev < 1 || dev < 0 || dev >= ndev || !mask_out) return -1;
  int64_t j = 0, mine = 0;
  for (int64_t i = 0; i < npts; ++i) {
    const bool ocean = !run_physics || run_physics[i];
    mask_out[i] = (ocean && (j % ndev) == dev) ? 1 : 0;
    if (ocean) { mine += mask_out[i]; ++j; }
  }
  return mine;
}

int mckpp_hip_multi_finalize(mckpp_hip_multi_handle m);

// the root-side buffers, streams and events go with the root device they were created on
static void multi_release_root(mckpp_hip_multi *m)
{
  if (m->root < 0) return;
  hipSetDevice(m->ctx[m->root]->device);
  for (auto &st : m->gstream) if (st) { hipStreamSynchronize(st); hipStreamDestroy(st); }
  for (auto &e : m->ev_done) if (e) hipEventDestroy(e);
  if (m->ev_init) hipEventDestroy(m->ev_init);
  if (m->ev_gcopy) hipEventDestroy(m->ev_gcopy);
  m->gstream.clear(); m->ev_done.clear(); m->ev_init = nullptr; m->ev_gcopy = nullptr;
  if (m->d_out) hipFree(m->d_out);
  if (m->d_stage) hipFree(m->d_stage);
  if (m->d_gipt) hipFree(m->d_gipt);
  m->d_out = m->d_stage = nullptr; m->d_gipt = nullptr;
  m->out_elems = m->stage_elems = m->gipt_elems = 0;
  m->gipt_valid = false;
  m->root = -1;
}

int mckpp_hip_multi_init(const mckpp_const_c *c, int32_t ndev, const int32_t *devices, mckpp_hip_multi_handle *out)
{
  if (!c || !out || ndev < 1) return fail("mckpp_hip_multi_init: bad argument (ndev=%d)", ndev);
  mckpp_hip_multi *m = new mckpp_hip_multi();
  for (int d = 0; d < ndev; ++d) {
    mckpp_hip_handle h = nullptr;
    if (mckpp_hip_init(c, devices ? devices[d] : d, &h) != 0) {
      for (auto *x : m->ctx) mckpp_hip_finalize(x);
      delete m;
      return -1;   // message of mckpp_hip_init
    }
    m->ctx.push_back(h);
  }
  m->mask.resize(ndev);
  m->ev_owner.assign(ndev, nullptr);
  for (int d = 0; d < ndev; ++d) {
    if (hipSetDevice(m->ctx[d]->device) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_owner[d], hipEventDisableTiming) != hipSuccess) {
      mckpp_hip_multi_finalize(m);
      return fail("mckpp_hip_multi_init: cannot create the events of shard %d", d);
    }
  }
  *out = m;
  return 0;
}

int mckpp_hip_multi_finalize(mckpp_hip_multi_handle m)
{
  if (!m) return 0;
  multi_release_root(m);
  for (size_t d = 0; d < m->ev_owner.size(); ++d)
    if (m->ev_owner[d]) { hipSetDevice(m->ctx[d]->device); hipEventDestroy(m->ev_owner[d]); }
  for (auto *x : m->ctx) mckpp_hip_finalize(x);
  delete m;
  return 0;
}

int32_t mckpp_hip_multi_ndev(mckpp_hip_multi_handle m) { return m ? (int32_t)m->ctx.size() : -1; }
mckpp_hip_handle mckpp_hip_multi_ctx(mckpp_hip_multi_handle m, int32_t i)
{
  if (!m || i < 0 || i >= (int)m->ctx.size()) { fail("mckpp_hip_multi_ctx: shard %d", i); return nullptr; }
  return m->ctx[i];
}

int mckpp_hip_multi_upload(mckpp_hip_multi_handle m, const mckpp_state_ptrs_c *s)
{
  if (!m || !s) return fail("mckpp_hip_multi_upload: null argument");
  if (s->npts <= 0) return fail("mckpp_hip_multi_upload: npts=%lld", (long long)s->npts);
  const int ndev = (int)m->ctx.size();
  m->npts = s->npts;
  for (int d = 0; d < ndev; ++d) {
    m->mask[d].assign((size_t)s->npts, 0);
    mckpp_host_shard_mask(s->npts, s->run_physics, ndev, d, m->mask[d].data());
    mckpp_state_ptrs_c sd = *s;
    sd.run_physics = m->mask[d].data();
    if (mckpp_hip_upload(m->ctx[d], &sd) != 0) return -1;
  }
  m->gipt_valid = false;   // the root's copy of the column maps is of the previous upload
  return 0;
}

#define MULTI_EACH(call)                                           \
  do {                                                             \
    if (!m) return fail("null multi handle");                      \
    for (auto *x : m->ctx) { if ((call) != 0) return -1; }         \
    return 0;                                                      \
  } while (0)

int mckpp_hip_multi_set_forcing(mckpp_hip_multi_handle m, const double *sflux) { MULTI_EACH(mckpp_hip_set_forcing(x, sflux)); }
int mckpp_hip_multi_set_diagnostics(mckpp_hip_multi_handle m, int on) { MULTI_EACH(mckpp_hip_set_diagnostics(x, on)); }
int mckpp_hip_multi_set_solver_mode(mckpp_hip_multi_handle m, int mode) { MULTI_EACH(mckpp_hip_set_solver_mode(x, mode)); }
int mckpp_hip_multi_init_ocean(mckpp_hip_multi_handle m, int ntime) { MULTI_EACH(mckpp_hip_init_ocean(x, ntime)); }
// asynchronous on every device: all shards are launched before the caller can wait on any of them
int mckpp_hip_multi_step(mckpp_hip_multi_handle m, int ntime, int nsteps) { MULTI_EACH(mckpp_hip_step(x, ntime, nsteps)); }
int mckpp_hip_multi_synchronize(mckpp_hip_multi_handle m) { MULTI_EACH(mckpp_hip_synchronize(x)); }
int mckpp_hip_multi_update_ancillaries(mckpp_hip_multi_handle m, const mckpp_state_ptrs_c *s) { MULTI_EACH(mckpp_hip_update_ancillaries(x, s)); }
int mckpp_hip_multi_bottomtemp(mckpp_hip_multi_handle m, const double *bottom_temp) { MULTI_EACH(mckpp_hip_bottomtemp(x, bottom_temp)); }
int mckpp_hip_multi_fluxes(mckpp_hip_multi_handle m, int ntime, const double *taux, const double *tauy, const double *swf,
                           const double *lwf, const double *lhf, const double *shf, const double *rain, const double *snow,
                           int l_rest, double flsn, double el)
{
  MULTI_EACH(mckpp_hip_fluxes(x, ntime, taux, tauy, swf, lwf, lhf, shf, rain, snow, l_rest, flsn, el));
}
int64_t mckpp_hip_multi_ncolumns(mckpp_hip_multi_handle m)
{
  if (!m) return -1;
  int64_t n = 0;
  for (auto *x : m->ctx) n += x->ncol;
  return n;
}

int mckpp_hip_multi_status(mckpp_hip_multi_handle m, int32_t *per_col, int64_t *n_flagged, int32_t *npasses)
{
  if (!m) return fail("null multi handle");
  int64_t nf = 0;
  std::vector<int32_t> st, np_;
  if (per_col) { st.assign((size_t)m->npts, 0); for (int64_t i = 0; i < m->npts; ++i) per_col[i] = 0; }
  if (npasses) { np_.assign((size_t)m->npts, 0); for (int64_t i = 0; i < m->npts; ++i) npasses[i] = 0; }
  for (auto *x : m->ctx) {
    int64_t f = 0;
    if (mckpp_hip_status(x, per_col ? st.data() : nullptr, &f, npasses ? np_.data() : nullptr) != 0) return -1;
    nf += f;
    for (int64_t c = 0; c < x->ncol; ++c) {
      const int i = x->ipt[(size_t)c];
      if (per_col) per_col[i] = st[(size_t)i];
      if (npasses) npasses[i] = np_[(size_t)i];
    }
  }
  if (n_flagged) *n_flagged = nf;
  return 0;
}

// ---- The gather.  One row field of every shard -> the caller's (npts, nlev) array in the Fortran
// layout: the shards' rows travel device to device to shard `root` (peer copies over the GPU interconnect, each on
// its own stream of the root device behind an event on its owner's stream - so all of them are in flight at once,
// and a gather queued behind a step overlaps the other shards' tail), are re-laid there into the 3-D order by one
// kernel per shard (disjoint points), and cross PCIe once.  Land points keep what `out` held.  `src[d]` are the
// shards' device rows (ld doubles each), src_off the first element taken from a row.
static int multi_prepare_root(mckpp_hip_multi *m, int root, size_t nout, size_t nstage)
{
  const int ndev = (int)m->ctx.size();
  mckpp_hip_ctx *r = m->ctx[root];
  if (m->root != root) {
    multi_release_root(m);
    HIPCHK(hipSetDevice(r->device));
    m->root = root;
    m->gstream.assign(ndev, nullptr);
    m->ev_done.assign(ndev, nullptr);
    for (int d = 0; d < ndev; ++d) {
      HIPCHK(hipStreamCreateWithFlags(&m->gstream[d], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&m->ev_done[d], hipEventDisableTiming));
      if (d != root && m->ctx[d]->device != r->device) {
        hipError_t e = hipDeviceEnablePeerAccess(m->ctx[d]->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // staged copies still work
      }
    }
    HIPCHK(hipEventCreateWithFlags(&m->ev_init, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_gcopy, hipEventDisableTiming));
  }
  HIPCHK(hipSetDevice(r->device));
  size_t ngipt = 0;
  for (auto *x : m->ctx) ngipt += (size_t)x->ncol;
  const bool grow = nout > m->out_elems || nstage > m->stage_elems || ngipt > m->gipt_elems;
  if (grow) {   // nothing of an earlier gather may still be using the buffers
    for (auto &st : m->gstream) HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipStreamSynchronize(r->stream));
    HIPCHK(hipStreamSynchronize(r->copy_stream));
  }
  if (nout > m->out_elems) { if (m->d_out) hipFree(m->d_out); m->d_out = nullptr; HIPCHK(hipMalloc(&m->d_out, nout * sizeof(double))); m->out_elems = nout; }
  if (nstage > m->stage_elems) { if (m->d_stage) hipFree(m->d_stage); m->d_stage = nullptr; HIPCHK(hipMalloc(&m->d_stage, nstage * sizeof(double))); m->stage_elems = nstage; }
  if (ngipt > m->gipt_elems) { if (m->d_gipt) hipFree(m->d_gipt); m->d_gipt = nullptr; HIPCHK(hipMalloc(&m->d_gipt, ngipt * sizeof(int))); m->gipt_elems = ngipt; m->gipt_valid = false; }
  if (!m->gipt_valid) {   // the shards' column maps, once per upload
    size_t go = 0;
    for (auto *x : m->ctx) {
      if (x->ncol) HIPCHK(hipMemcpy(m->d_gipt + go, x->ipt.data(), (size_t)x->ncol * sizeof(int), hipMemcpyHostToDevice));
      go += (size_t)x->ncol;
    }
    m->gipt_valid = true;
  }
  return 0;
}

static int multi_gather_rows(mckpp_hip_multi *m, int root, const std::vector<const double *> &src, int ld, int src_off,
                             int nlev, double *out, const double *whole = nullptr, size_t whole_elems = 0)
{
  const int ndev = (int)m->ctx.size();
  mckpp_hip_ctx *r = m->ctx[root];
  const size_t nout = (size_t)m->npts * nlev;
  size_t nstage = 0, ncols = 0;
  for (int d = 0; d < ndev; ++d) {
    ncols += (size_t)m->ctx[d]->ncol;
    if (d != root) nstage += (size_t)m->ctx[d]->ncol * ld;   // the root's own rows are read where they are
  }
  if (multi_prepare_root(m, root, nout, nstage ? nstage : 1)) return -1;
  if (whole) pin_host(r, whole, whole_elems * sizeof(double));
  else pin_host(r, out, nout * sizeof(double));
  // the 3-D image: behind the transfer of the previous gather; land points keep the caller's values
  HIPCHK(hipStreamWaitEvent(r->stream, m->ev_gcopy, 0));
  if (ncols < (size_t)m->npts) HIPCHK(hipMemcpyAsync(m->d_out, out, nout * sizeof(double), hipMemcpyHostToDevice, r->stream));
  HIPCHK(hipEventRecord(m->ev_init, r->stream));
  size_t so = 0, go = 0;
  for (int d = 0; d < ndev; ++d) {
    mckpp_hip_ctx *x = m->ctx[d];
    if (x->ncol == 0) continue;
    const size_t n = (size_t)x->ncol * ld;
    hipStream_t gs = m->gstream[d];
    HIPCHK(hipSetDevice(x->device));
    HIPCHK(hipEventRecord(m->ev_owner[d], x->stream));   // whatever the owner's stream still has queued
    HIPCHK(hipSetDevice(r->device));
    HIPCHK(hipStreamWaitEvent(gs, m->ev_owner[d], 0));
    HIPCHK(hipStreamWaitEvent(gs, m->ev_init, 0));
    const double *rows = src[d];
    if (d != root) {
      if (x->device == r->device) HIPCHK(hipMemcpyAsync(m->d_stage + so, src[d], n * sizeof(double), hipMemcpyDeviceToDevice, gs));
      else HIPCHK(hipMemcpyPeerAsync(m->d_stage + so, r->device, src[d], x->device, n * sizeof(double), gs));
      rows = m->d_stage + so;
      so += n;
    }
    HIPCHK(mckpp_launch_scatter_rows(rows, ld, src_off, m->d_gipt + go, x->ncol, m->d_out, m->npts, nlev, 0, gs));
    HIPCHK(hipEventRecord(m->ev_done[d], gs));
    HIPCHK(hipStreamWaitEvent(r->copy_stream, m->ev_done[d], 0));
    go += (size_t)x->ncol;
  }
  HIPCHK(hipStreamWaitEvent(r->copy_stream, m->ev_init, 0));
  HIPCHK(hipMemcpyAsync(out, m->d_out, nout * sizeof(double), hipMemcpyDeviceToHost, r->copy_stream));
  HIPCHK(hipEventRecord(m->ev_gcopy, r->copy_stream));
  return 0;
}

// the gathers queued so far have delivered
static int multi_gather_finish(mckpp_hip_multi *m)
{
  if (m->root < 0) return 0;
  mckpp_hip_ctx *r = m->ctx[m->root];
  HIPCHK(hipSetDevice(r->device));
  HIPCHK(hipStreamSynchronize(r->copy_stream));
  return 0;
}

// `field` 0 U, 1 V, 2 T, 3 S -> out(npts,nzp1), 4 hmix -> out(npts)
int mckpp_hip_multi_gather(mckpp_hip_multi_handle m, int32_t field, int32_t root, double *out)
{
  if (!m || !out) return fail("mckpp_hip_multi_gather: null argument");
  const int ndev = (int)m->ctx.size();
  if (root < 0 || root >= ndev) return fail("mckpp_hip_multi_gather: root %d of %d", root, ndev);
  if (field < 0 || field > 4) return fail("mckpp_hip_multi_gather: field %d", field);
  if (m->npts <= 0) return fail("mckpp_hip_multi_gather: nothing uploaded");
  std::vector<const double *> src(ndev);
  for (int d = 0; d < ndev; ++d) {
    mckpp_hip_ctx *x = m->ctx[d];
    src[d] = field == 0 ? x->d_prof[P_U] : field == 1 ? x->d_prof[P_V] : field == 2 ? x->d_prof[P_T]
           : field == 3 ? x->d_prof[P_S] : x->d_cs;
  }
  mckpp_hip_ctx *r = m->ctx[root];
  if (multi_gather_rows(m, root, src, field < 4 ? r->ld : MCKPP_CS, field < 4 ? 0 : CS_HMIX, field < 4 ? r->nzp1 : 1, out))
    return -1;
  return multi_gather_finish(m);
}

// mckpp_hip_download for all shards: the per-column records of every shard come to the host on their own (each
// crosses PCIe once), every row field goes through the gather - one transfer per field, whatever the number
// of devices.
int mckpp_hip_multi_download(mckpp_hip_multi_handle m, mckpp_state_ptrs_c *s, uint32_t mask)
{
  if (!m || !s) return fail("mckpp_hip_multi_download: null argument");
  if (s->npts != m->npts) return fail("mckpp_hip_multi_download: npts=%lld but %lld were uploaded", (long long)s->npts, (long long)m->npts);
  const int ndev = (int)m->ctx.size();
  std::vector<std::vector<row_xfer>> plans(ndev);
  for (int d = 0; d < ndev; ++d) download_plan(m->ctx[d], s, mask, plans[d]);
  std::vector<const double *> src(ndev);
  for (size_t e = 0; e < plans[0].size(); ++e) {
    for (int d = 0; d < ndev; ++d) src[d] = plans[d][e].dev;
    const row_xfer &x = plans[0][e];
    if (multi_gather_rows(m, 0, src, m->ctx[0]->ld, x.src_off, x.nlev, x.host, x.whole, x.whole_elems)) return -1;
  }
  for (auto *x : m->ctx) {
    if (x->ncol == 0) continue;
    HIPCHK(hipSetDevice(x->device));
    if (download_records(x, s, mask)) return -1;
  }
  return multi_gather_finish(m);
}

// ---- the forced time loop, the output windows and the restart set for all shards
int mckpp_hip_multi_set_flux_series(mckpp_hip_multi_handle m, int rec0, int nrec, const double *fields)
{
  MULTI_EACH(mckpp_hip_set_flux_series(x, rec0, nrec, fields));
}
// asynchronous on every device, like mckpp_hip_multi_step
int mckpp_hip_multi_run_forced(mckpp_hip_multi_handle m, int nt_first, int nsteps, int ndtocn, int l_rest, double flsn, double el)
{
  MULTI_EACH(mckpp_hip_run_forced(x, nt_first, nsteps, ndtocn, l_rest, flsn, el));
}
int mckpp_hip_multi_window_select(mckpp_hip_multi_handle m, const int32_t *fields, int32_t nfields) { MULTI_EACH(mckpp_hip_window_select(x, fields, nfields)); }
int mckpp_hip_multi_window_reset(mckpp_hip_multi_handle m) { MULTI_EACH(mckpp_hip_window_reset(x)); }
int mckpp_hip_multi_window_accumulate(mckpp_hip_multi_handle m) { MULTI_EACH(mckpp_hip_window_accumulate(x)); }

// every shard reduces its own columns; the reduced rows are gathered like any other field
int mckpp_hip_multi_window_fetch(mckpp_hip_multi_handle m, int field, int op, double *out)
{
  if (!m || !out) return fail("mckpp_hip_multi_window_fetch: null argument");
  if (m->npts <= 0) return fail("mckpp_hip_multi_window_fetch: nothing uploaded");
  const int ndev = (int)m->ctx.size();
  std::vector<const double *> src(ndev, nullptr);
  int ld_out = 0, nlev = 0;
  for (int d = 0; d < ndev; ++d)
    if (window_prepare(m->ctx[d], field, op, &src[d], &ld_out, &nlev)) return -1;
  if (multi_gather_rows(m, 0, src, ld_out, 0, nlev, out)) return -1;
  return multi_gather_finish(m);
}

// one file per shard: <path>.<d>of<ndev>
static std::string shard_path(const char *path, int d, int ndev)
{
  return std::string(path) + "." + std::to_string(d) + "of" + std::to_string(ndev);
}
int mckpp_hip_multi_save_restart(mckpp_hip_multi_handle m, const char *path)
{
  if (!m || !path) return fail("mckpp_hip_multi_save_restart: null argument");
  const int ndev = (int)m->ctx.size();
  for (int d = 0; d < ndev; ++d)
    if (m->ctx[d]->ncol > 0 && mckpp_hip_save_restart(m->ctx[d], shard_path(path, d, ndev).c_str()) != 0) return -1;
  return 0;
}
// the files must have been written by a handle with the same number of shards over the same land mask: every
// shard's column map is checked against the one the current upload gave it before anything is replaced
int mckpp_hip_multi_load_restart(mckpp_hip_multi_handle m, const char *path)
{
  if (!m || !path) return fail("mckpp_hip_multi_load_restart: null argument");
  if (m->npts <= 0) return fail("mckpp_hip_multi_load_restart: upload the state first (the shards' column maps come from it)");
  const int ndev = (int)m->ctx.size();
  for (int d = 0; d < ndev; ++d) {   // pass 1: headers and column maps only
    mckpp_hip_ctx *x = m->ctx[d];
    if (x->ncol == 0) continue;
    const std::string sp = shard_path(path, d, ndev);
    FILE *f = fopen(sp.c_str(), "rb");
    if (!f) return fail("mckpp_hip_multi_load_restart: cannot open %s", sp.c_str());
    restart_header hd{};
    std::vector<int> ipt;
    bool ok = fread(&hd, sizeof hd, 1, f) == 1 && memcmp(hd.magic, kRestartMagic, 8) == 0 && hd.ncol == x->ncol && hd.npts == x->npts;
    if (ok) { ipt.resize((size_t)hd.ncol); ok = fread(ipt.data(), sizeof(int), ipt.size(), f) == ipt.size() && ipt == x->ipt; }
    fclose(f);
    if (!ok) return fail("mckpp_hip_multi_load_restart: %s does not belong to shard %d of %d of the uploaded columns", sp.c_str(), d, ndev);
  }
  for (int d = 0; d < ndev; ++d)
    if (m->ctx[d]->ncol > 0 && mckpp_hip_load_restart(m->ctx[d], shard_path(path, d, ndev).c_str()) != 0) return -1;
  return 0;
}

// the caller's arrays pinned on behalf of this handle go back to pageable memory (before the caller frees them)
int mckpp_hip_multi_release_host_arrays(mckpp_hip_multi_handle m)
{
  if (!m) return fail("null multi handle");
  for (auto *x : m->ctx) { HIPCHK(hipSetDevice(x->device)); if (xfer_finish(x)) return -1; unpin_all(x); }
  return 0;
}

}  // extern "C"
