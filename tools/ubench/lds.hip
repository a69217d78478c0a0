// What an LDS-fed recurrence costs a single wave: x = y_i - g_i * x over NL levels, four levels per trip, operands
// read with ds_read2_b64 one trip ahead (asm, explicit wait counts), results written back with ds_write2_b64.
// Variants: lane stride in LDS (doubles), level stride, with / without the stores, with / without the reads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int O0, int O1> __device__ __forceinline__ d2 rd(unsigned a)
{
  d2 v;
  asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a), "n"(O0), "n"(O1) : "memory");
  return v;
}
template <int O0, int O1> __device__ __forceinline__ void wr(unsigned a, double x, double y)
{
  asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(a), "v"(x), "v"(y), "n"(O0), "n"(O1) : "memory");
}
template <int N> __device__ __forceinline__ void wt(d2 &a, d2 &b, d2 &c, d2 &d) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N)); }

template <int LSTRIDE, bool STORES, bool READS>
__global__ void k(unsigned long long *cyc, double *out, int nl, int nact)
{
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 9000; i += 64) lds[i] = 1e-3 * (1 + i % 13);
  __syncthreads();
  if (lane >= nact) return;
  constexpr int KS = 9;
  const unsigned step = 4 * KS * 8;
  unsigned ay = (unsigned)(unsigned long long)(lds + (lane % 16) * LSTRIDE + 3), ag = ay + 32;
  double yy = 1.0 + lane;
  d2 a0 = rd<0, 9>(ay), a1 = rd<18, 27>(ay), a2 = rd<0, 9>(ag), a3 = rd<18, 27>(ag), b0, b1, b2, b3;
  auto body = [&](unsigned aw, const d2 &y32, const d2 &y10, const d2 &g32, const d2 &g10) {
    yy = y10.y - g10.y * yy; const double r0 = yy;
    yy = y10.x - g10.x * yy; const double r1 = yy;
    yy = y32.y - g32.y * yy; const double r2 = yy;
    yy = y32.x - g32.x * yy;
    if (STORES) { wr<18, 27>(aw, r1, r0); wr<0, 9>(aw, yy, r2); }
  };
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i + 8 <= nl; i += 8) {
    if (READS) { b0 = rd<0, 9>(ay + step); b1 = rd<18, 27>(ay + step); b2 = rd<0, 9>(ag + step); b3 = rd<18, 27>(ag + step); wt<4>(a0, a1, a2, a3); }
    else { b0 = a0; b1 = a1; b2 = a2; b3 = a3; asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)); }
    body(ay, a0, a1, a2, a3);
    if (READS) { a0 = rd<0, 9>(ay + 2 * step); a1 = rd<18, 27>(ay + 2 * step); a2 = rd<0, 9>(ag + 2 * step); a3 = rd<18, 27>(ag + 2 * step); wt<4>(b0, b1, b2, b3); }
    else { a0 = b0; a1 = b1; a2 = b2; a3 = b3; asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    body(ay + step, b0, b1, b2, b3);
    ay += 2 * step; ag += 2 * step;
    if (i % 64 == 56) { ay -= 16 * step; ag -= 16 * step; }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[0] = t1 - t0;
  out[lane] = yy + a0.x + a1.x + a2.x + a3.x;
}
template <int LSTRIDE, bool STORES, bool READS> void run(const char *name, unsigned long long *c, double *d, int nact)
{
  const int nl = 4096;
  unsigned long long h = 0;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<LSTRIDE, STORES, READS>), hipFuncAttributeMaxDynamicSharedMemorySize, 80000);
  for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL((k<LSTRIDE, STORES, READS>), dim3(1), dim3(64), 80000, 0, c, d, nl, nact); hipDeviceSynchronize(); }
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-64s lanes=%2d %6.1f cycles per level\n", name, nact, (double)h / nl);
}
int main()
{
  unsigned long long *c; double *d;
  hipMalloc(&c, 8); hipMalloc(&d, 64 * 8);
  for (int nact : {64, 45, 3}) {
    run<1, false, false>("recurrence only (no LDS)", c, d, nact);
    run<1, true, false>("+ stores, lane stride 1", c, d, nact);
    run<1, false, true>("+ reads one trip ahead, lane stride 1", c, d, nact);
    run<1, true, true>("+ reads + stores, lane stride 1", c, d, nact);
    run<567, true, true>("+ reads + stores, lane stride 567 doubles", c, d, nact);
    run<16, true, true>("+ reads + stores, lane stride 16 doubles (conflicts)", c, d, nact);
  }
  return 0;
}
