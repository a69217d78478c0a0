// What one LDS instruction costs the wave that issues it (single wave on a CU, nothing else running): N independent
// accesses back to back, one s_waitcnt at the end of a batch of eight, for each access type and number of active lanes;
// then the same stores with independent fp64 work between them (does the store's cost hide behind VALU work?).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/ldsissue.hip -o tools/ubench/ldsissue && tools/ubench/ldsissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
enum { RD64, RD2_64, RD128, WR64, WR2_64, WR128, WR64_FMA, WR2_64_FMA, WR128_FMA, FMA_ONLY, RD2_WR2_FMA, GLD2_FMA, GLD4_FMA, GST2_FMA, GST4_FMA, RD128_FMA, RD2_FMA, NT };
const char *names[NT] = {"ds_read_b64", "ds_read2_b64", "ds_read_b128", "ds_write_b64", "ds_write2_b64", "ds_write_b128",
                         "ds_write_b64 + 2 independent fma", "ds_write2_b64 + 2 independent fma", "ds_write_b128 + 2 independent fma",
                         "2 independent fma alone", "ds_read2_b64 + ds_write2_b64 + 4 fma",
                         "global_load_dwordx2 + 2 independent fma", "global_load_dwordx4 + 2 independent fma", "global_store_dwordx2 + 2 independent fma",
                         "global_store_dwordx4 + 2 independent fma", "ds_read_b128 + 2 independent fma", "ds_read2_b64 + 2 independent fma"};

template <int T>
__global__ void k(unsigned long long *cyc, double *out, int n, int nact, int lstride, double *g, int misalign)
{
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 9000; i += 64) lds[i] = 1e-3 * (1 + i % 13);
  __syncthreads();
  if (lane >= nact) return;
  unsigned a = (unsigned)(unsigned long long)(lds + lane * lstride);
  if (T == RD128 || T == WR128 || T == WR128_FMA || T == RD128_FMA) a = (a & ~15u) + (misalign ? 8u : 0u);
  double *gp = g + lane * lstride * 2;
  double x = 1.0 + lane, y = 2.0 + lane, z = 0.5, w = 0.25;
  d2 v0{x, y}, v1 = v0, v2 = v0, v3 = v0, v4 = v0, v5 = v0, v6 = v0, v7 = v0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; i += 8) {
#define RD(v, off) \
    if (T == RD64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v.x) : "v"(a), "n"(off * 8) : "memory"); \
    if (T == RD2_64) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a), "n"(off), "n"(off + 9) : "memory"); \
    if (T == RD128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(off * 16) : "memory");
#define WR(v, off) \
    if (T == WR64 || T == WR64_FMA) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(a), "v"(v.x), "n"(off * 8) : "memory"); \
    if (T == WR2_64 || T == WR2_64_FMA || T == RD2_WR2_FMA) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(a), "v"(v.x), "v"(v.y), "n"(off), "n"(off + 9) : "memory"); \
    if (T == WR128 || T == WR128_FMA) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(a), "v"(v), "n"(off * 16) : "memory");
#define FM() \
    if (T >= WR64_FMA) { x = __builtin_fma(x, z, w); y = __builtin_fma(y, z, w); asm volatile("" : "+v"(x), "+v"(y)); }
    if (T <= RD128) { RD(v0, 0) RD(v1, 1) RD(v2, 2) RD(v3, 3) RD(v4, 4) RD(v5, 5) RD(v6, 6) RD(v7, 7) }
    else if (T == RD2_WR2_FMA) {
      asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:9" : "=v"(v1) : "v"(a) : "memory"); FM() WR(v0, 20) FM()
      asm volatile("ds_read2_b64 %0, %1 offset0:1 offset1:10" : "=v"(v2) : "v"(a) : "memory"); FM() WR(v0, 21) FM()
      asm volatile("ds_read2_b64 %0, %1 offset0:2 offset1:11" : "=v"(v3) : "v"(a) : "memory"); FM() WR(v0, 22) FM()
      asm volatile("ds_read2_b64 %0, %1 offset0:3 offset1:12" : "=v"(v4) : "v"(a) : "memory"); FM() WR(v0, 23) FM()
    }
    else if (T == GLD2_FMA || T == GLD4_FMA || T == GST2_FMA || T == GST4_FMA || T == RD128_FMA || T == RD2_FMA) {
#define GX(v, off) \
      if (T == GLD2_FMA) asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(v.x) : "v"(gp), "n"(off * 16) : "memory"); \
      if (T == GLD4_FMA) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(gp), "n"(off * 16) : "memory"); \
      if (T == GST2_FMA) asm volatile("global_store_dwordx2 %0, %1, off offset:%2" : : "v"(gp), "v"(v.x), "n"(off * 16) : "memory"); \
      if (T == GST4_FMA) asm volatile("global_store_dwordx4 %0, %1, off offset:%2" : : "v"(gp), "v"(v), "n"(off * 16) : "memory"); \
      if (T == RD128_FMA) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(off * 16) : "memory"); \
      if (T == RD2_FMA) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a), "n"(off), "n"(off + 9) : "memory");
      GX(v0, 0) FM() GX(v1, 1) FM() GX(v2, 2) FM() GX(v3, 3) FM() GX(v4, 4) FM() GX(v5, 5) FM() GX(v6, 6) FM() GX(v7, 7) FM()
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : : "memory");
    }
    else { WR(v0, 0) FM() WR(v1, 1) FM() WR(v2, 2) FM() WR(v3, 3) FM() WR(v4, 4) FM() WR(v5, 5) FM() WR(v6, 6) FM() WR(v7, 7) FM() }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : : "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[0] = t1 - t0;
  out[lane] = x + y + v0.x + v1.x + v2.x + v3.x + v4.y + v5.x + v6.x + v7.y;
}
static double *g_scratch; static int g_misalign;
template <int T> void run(unsigned long long *c, double *d, int nact, int lstride)
{
  const int n = 8192;
  unsigned long long h = 0;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 80000);
  for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL((k<T>), dim3(1), dim3(64), 80000, 0, c, d, n, nact, lstride, g_scratch, g_misalign); hipDeviceSynchronize(); }
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-44s lanes=%2d lane stride %3d doubles%s: %6.1f cycles per %s\n", names[T], nact, lstride, g_misalign ? " (b128 at 8 mod 16)" : "", (double)h / (T == RD2_WR2_FMA ? n / 2 : n),
         T == RD2_WR2_FMA ? "(read2 + write2 + 4 fma)" : T == FMA_ONLY ? "pair of fma" : "access");
}
int main()
{
  unsigned long long *c; double *d;
  hipMalloc(&c, 8); hipMalloc(&d, 64 * 8); hipMalloc(&g_scratch, 1 << 20); hipMemset(g_scratch, 0, 1 << 20);
  for (int ls : {1, 141}) for (int nact : {64, 45, 15}) {
    if (ls == 141 && nact == 64) continue;   // 64 x 141 doubles do not fit
    run<RD64>(c, d, nact, ls); run<RD2_64>(c, d, nact, ls); run<RD128>(c, d, nact, ls);
    run<WR64>(c, d, nact, ls); run<WR2_64>(c, d, nact, ls); run<WR128>(c, d, nact, ls);
    run<FMA_ONLY>(c, d, nact, ls); run<WR64_FMA>(c, d, nact, ls); run<WR2_64_FMA>(c, d, nact, ls); run<WR128_FMA>(c, d, nact, ls);
    run<RD2_WR2_FMA>(c, d, nact, ls);
    run<RD2_FMA>(c, d, nact, ls); run<RD128_FMA>(c, d, nact, ls);
    g_misalign = 1; run<RD128>(c, d, nact, ls); run<RD128_FMA>(c, d, nact, ls); run<WR128>(c, d, nact, ls); g_misalign = 0;
    run<GLD2_FMA>(c, d, nact, ls); run<GLD4_FMA>(c, d, nact, ls); run<GST2_FMA>(c, d, nact, ls); run<GST4_FMA>(c, d, nact, ls);
  }
  return 0;
}
