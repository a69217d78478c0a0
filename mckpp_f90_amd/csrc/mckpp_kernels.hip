// mckpp_kernels.hip - the small gfx950 kernels around the column step: layout conversion between the
// Fortran arrays (column index fastest) and the device rows (level fastest), surface-flux assembly
// (mckpp_fluxes), the bottom-temperature override, output-window reductions, and the batch probes the
// parity tests use.  The column step itself lives in mckpp_kernels_ps.hip.
//
// Arithmetic is written operation-for-operation in the reference's expression order and built with
// -ffp-contract=off, so results are bitwise reproducible against the CPU oracle (tests/).
#include "mckpp_colmath.h"

namespace {

using namespace mckpp_dev;

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
// Output-window reductions: what XIOS does with the fields mckpp_xios_output_control
// sends (src/mckpp_xios_io.F90:74-210; operations instant / average / minimum / maximum of
// run/iodef.xml:88-157), on the device so that only reduced fields cross PCIe / xGMI.
// One output field = nlev values per column taken from a device row array (element src_off + l of
// the column's row, row length src_ld) or from the column record (src_ld = MCKPP_CS, nlev = 1),
// optionally plus the column's Sref (the reference sends S = X(:,:,2) + Sref).  One thread per
// (column, level): either one sample into `inst`, or the running sum / min / max.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_out_sample(const double *__restrict__ src, int src_ld, int src_off,
                                                   const double *__restrict__ cs, int add_sref, int64_t ncol, int nlev,
                                                   int ld_out, double *__restrict__ sum, double *__restrict__ mn,
                                                   double *__restrict__ mx, int first, double *__restrict__ inst)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)ncol * ld_out) return;
  const int64_t c = (int64_t)(i / ld_out);
  const int l = (int)(i - (size_t)c * ld_out);
  double v = 0.0;
  if (l < nlev) {
    v = src[(size_t)c * src_ld + src_off + l];
    if (add_sref) v = v + cs[(size_t)c * MCKPP_CS + CS_SREF];
  }
  if (inst) { inst[i] = v; return; }
  double s_ = sum[i], a = mn[i], b = mx[i];
  if (first) { s_ = 0.0; a = v; b = v; }
  sum[i] = s_ + v;
  mn[i] = v < a ? v : a;
  mx[i] = v > b ? v : b;
}

__global__ void k_window_mean(const double *__restrict__ sum, double *__restrict__ out, size_t n, double count)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = sum[i] / count;
}

}  // namespace


// Which XCDs does this device have?  Every workgroup of a large grid ORs the bit of the XCC it runs on (read from
// the hardware) into one word: the ids of the queues of a launch of several steps (k_column_ps, M0).
__global__ void k_xcc_probe(unsigned *mask)
{
  if (threadIdx.x == 0) atomicOr(mask, 1u << (__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20 /* HW_REG_XCC_ID, bits 3:0 */) & 15));
}

hipError_t mckpp_launch_xcc_probe(unsigned *mask, hipStream_t stream)
{
  hipLaunchKernelGGL(k_xcc_probe, dim3(4096), dim3(64), 0, stream, mask);
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

// The per-column records <-> the caller's (npts) arrays without a host loop, for a context whose columns are all of
// the grid's points or not: forcing slabs sflux(:,1:6,5,0) (six contiguous (npts) slabs) into the records' flux
// slots, and selected record slots out into (npts) slabs in 3-D order (land points of the slabs are not written).
__global__ __launch_bounds__(256) void k_unpack_sflux(const double *__restrict__ slabs, const int *__restrict__ ipt,
                                                      double *__restrict__ cs, int64_t ncol, int64_t npts)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncol) return;
  const int64_t i = ipt[c];
#pragma unroll
  for (int m = 0; m < 6; ++m) cs[c * MCKPP_CS + CS_SFLUX1 + m] = slabs[i + npts * m];
}

__global__ __launch_bounds__(256) void k_pack_records(const double *__restrict__ cs, const int *__restrict__ ci,
                                                      const int *__restrict__ ipt, int64_t ncol, int64_t npts,
                                                      mckpp_pack_list l, double *__restrict__ dout, int *__restrict__ iout)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncol) return;
  const int64_t i = ipt[c];
  for (int j = 0; j < l.nd; ++j) dout[i + npts * j] = cs[c * MCKPP_CS + l.dslot[j]];
  for (int j = 0; j < l.ni; ++j) iout[i + npts * j] = ci[c * MCKPP_CI + l.islot[j]];
}

hipError_t mckpp_launch_unpack_sflux(const double *slabs, const int *ipt, double *cs, int64_t ncol, int64_t npts, hipStream_t stream)
{
  if (ncol <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_unpack_sflux, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, stream, slabs, ipt, cs, ncol, npts);
  return hipGetLastError();
}

hipError_t mckpp_launch_pack_records(const double *cs, const int *ci, const int *ipt, int64_t ncol, int64_t npts,
                                     const mckpp_pack_list &l, double *dout, int *iout, hipStream_t stream)
{
  if (ncol <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_records, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, stream, cs, ci, ipt, ncol, npts, l, dout, iout);
  return hipGetLastError();
}

hipError_t mckpp_launch_out_sample(const double *src, int src_ld, int src_off, const double *cs, int add_sref,
                                   int64_t ncol, int nlev, int ld_out, double *sum, double *mn, double *mx, int first,
                                   double *inst, hipStream_t stream)
{
  const size_t n = (size_t)ncol * ld_out;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_out_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, src_ld, src_off, cs,
                     add_sref, ncol, nlev, ld_out, sum, mn, mx, first, inst);
  return hipGetLastError();
}

hipError_t mckpp_launch_window_mean(const double *sum, double *out, size_t n, double count, hipStream_t stream)
{
  hipLaunchKernelGGL(k_window_mean, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sum, out, n, count);
  return hipGetLastError();
}
