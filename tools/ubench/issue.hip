// fp64 issue rate of one CU: W waves of one workgroup, each running 4 independent v_fma_f64 chains; shader cycles per
// wave-instruction of the whole CU (4 SIMDs).  78.6 TFLOP/s at 2.4 GHz over 256 CUs is one wave-instruction per cycle per CU.
// Every wave stamps its own start and end: the SIMD's arbiter favours its oldest wave, which runs at single-wave
// speed while the others wait, so the time of wave 0 alone says nothing about the CU (first wave's time is printed too).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__global__ void k(double *out, unsigned long long *cyc, double a, double b)
{
  double x = a + threadIdx.x, y = b, z = a * 0.5, w = b * 0.25, u = a * 0.125, v = b * 0.0625;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int i = 0; i < N / 4; ++i) { x = __builtin_fma(x, y, z); w = __builtin_fma(w, y, z); u = __builtin_fma(u, y, z); v = __builtin_fma(v, y, z); }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
  out[threadIdx.x] = x + w + u + v;
}
#define NL (1 << 22)
__global__ void klong(double *out, unsigned long long *cyc, double a, double b)
{
  double x = a + threadIdx.x, y = b, z = a * 0.5, w = b * 0.25, u = a * 0.125, v = b * 0.0625;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 4
  for (int i = 0; i < NL / 4; ++i) { x = __builtin_fma(x, y, z); w = __builtin_fma(w, y, z); u = __builtin_fma(u, y, z); v = __builtin_fma(v, y, z); }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
  out[threadIdx.x] = x + w + u + v;
}
int main()
{
  double *d; unsigned long long *c; hipMalloc(&d, 1024 * 8); hipMalloc(&c, 32 * 8);
  for (int W : {1, 2, 4, 8, 12, 16}) {
    unsigned long long h[32], best = ~0ull, first = 0;
    for (int r = 0; r < 3; ++r) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64 * W), 0, 0, d, c, 1.0000001, 0.9999999); hipDeviceSynchronize();
      hipMemcpy(h, c, 2 * W * 8, hipMemcpyDeviceToHost);
      unsigned long long lo = ~0ull, hi = 0;
      for (int w = 0; w < W; ++w) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
      if (hi - lo < best) { best = hi - lo; first = h[1] - h[0]; }
    }
    printf("%2d waves on one CU: %6llu cycles from the first start to the last end (wave 0 alone: %llu) for %d fma per wave -> %.2f cycles per wave-instruction of the CU (%.2f per SIMD)\n",
           W, best, first, N, (double)best / (N * W), (double)best / (N * W) * (W < 4 ? W : 4));
  }
  // what a tick of s_memtime is: a long run of the 16-wave case against HIP event time (and s_memrealtime, 100 MHz);
  // wave 0's stamps only, so its ticks cover its own share of the run (see above), the event time all 16 waves
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long *c2; hipMalloc(&c2, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(klong, dim3(1), dim3(1024), 0, 0, d, c2, 1.0000001, 0.9999999);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, c2, 16, hipMemcpyDeviceToHost);
    printf("long run: %.3f ms by HIP events, %llu ticks of s_memtime (%.1f per us), %llu ticks of s_memrealtime (%.1f per us); %d fma per wave x 16 waves -> %.2f wave-fma per us per CU\n",
           ms, h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3), NL, 16.0 * NL / (ms * 1e3));
  }
  return 0;
}
