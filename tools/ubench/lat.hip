// Single-wave latency / issue micro-benchmarks for gfx950 (one workgroup of 64 lanes, few lanes active as in
// the manager phases of the column kernel).  Prints shader-clock cycles per operation.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
template <int KIND>
__global__ void k(double *out, unsigned long long *cyc, double a, double b, int nact)
{
  __shared__ double lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  if (lane >= nact) return;
  double x = a + lane, y = b, z = a * 0.5, w = b * 0.25;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (KIND == 0) {   // dependent fma chain
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, z);
  } else if (KIND == 1) {   // 4 independent fma chains
#pragma unroll 4
    for (int i = 0; i < N / 4; ++i) { x = __builtin_fma(x, y, z); w = __builtin_fma(w, y, z); a = __builtin_fma(a, y, z); b = __builtin_fma(b, y, x); }
    x += w + a + b;
  } else if (KIND == 2) {   // dependent mul then add
#pragma unroll 16
    for (int i = 0; i < N / 2; ++i) { x = x * y; x = x + z; }
  } else if (KIND == 3) {   // dependent rcp
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x) + 0.0 * z;
  } else if (KIND == 4) {   // dependent div_fixup
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_div_fixup(x, y, z);
  } else if (KIND == 5) {   // dependent LDS read chain (pointer chasing through values)
    int idx = lane;
#pragma unroll 8
    for (int i = 0; i < N; ++i) { double v = lds[idx & 4095]; idx = (int)v + idx + 13; }
    x = idx;
  } else if (KIND == 6) {   // v_cmp -> ballot -> scalar branch per iteration (never taken path) + fma
    for (int i = 0; i < N; ++i) {
      x = __builtin_fma(x, y, z);
      unsigned long long m = __builtin_amdgcn_ballot_w64(__builtin_amdgcn_frexp_exp(x) < -960);
      if (__builtin_expect(m != 0ull, 0)) x = x / y;
    }
  } else if (KIND == 7) {   // LDS write then dependent read of another address (round trip with write in flight)
#pragma unroll 8
    for (int i = 0; i < N; ++i) { lds[(lane * 67 + i) & 4095] = x; x = x + lds[(lane * 131 + 2 * i + 1) & 4095]; }
  } else if (KIND == 8) {   // the division step of the sweeps: rcp_refine + div_fast on a dependent value
    for (int i = 0; i < N / 8; ++i) {
      double r = __builtin_amdgcn_rcp(x);
      double e = __builtin_fma(-x, r, 1.0); r = __builtin_fma(r, e, r); e = __builtin_fma(-x, r, 1.0); r = __builtin_fma(r, e, r);
      double q = y * r; double e2 = __builtin_fma(-x, q, y); double res = __builtin_fma(e2, r, q);
      x = __builtin_amdgcn_div_fixup(res, x, y) + z;
    }
  }
  else if (KIND == 9) {   // back-substitution chain with every operand in its own VGPR pair: x = a_i - g_i * x
    double a0 = lds[lane], a1 = lds[lane + 64], a2 = lds[lane + 128], a3 = lds[lane + 192];
    double g0 = lds[lane + 256] * 1e-3, g1 = lds[lane + 320] * 1e-3, g2 = lds[lane + 384] * 1e-3, g3 = lds[lane + 448] * 1e-3;
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3));
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
    for (int i = 0; i < N / 8; ++i) {
      x = a0 - g0 * x; x = a1 - g1 * x; x = a2 - g2 * x; x = a3 - g3 * x;
    }
  } else if (KIND == 10) {   // the same chain, results also stored to LDS (one ds_write_b64 per level)
    double a0 = lds[lane], a1 = lds[lane + 64], a2 = lds[lane + 128], a3 = lds[lane + 192];
    double g0 = lds[lane + 256] * 1e-3, g1 = lds[lane + 320] * 1e-3, g2 = lds[lane + 384] * 1e-3, g3 = lds[lane + 448] * 1e-3;
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3));
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 8; ++i) {
      x = a0 - g0 * x; lds[1024 + lane] = x; x = a1 - g1 * x; lds[1100 + lane] = x; x = a2 - g2 * x; lds[1200 + lane] = x; x = a3 - g3 * x; lds[1300 + lane] = x;
    }
  } else if (KIND == 11) {   // the same chain, operands re-read from LDS every trip (read latency exposed once per 4 levels)
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 8; ++i) {
      const int o = (i & 7) * 64;
      double a0 = lds[lane + o], a1 = lds[lane + 64 + o], a2 = lds[lane + 128 + o], a3 = lds[lane + 192 + o];
      double g0 = lds[lane + 2048 + o], g1 = lds[lane + 2112 + o], g2 = lds[lane + 2176 + o], g3 = lds[lane + 2240 + o];
      x = a0 - g0 * x; x = a1 - g1 * x; x = a2 - g2 * x; x = a3 - g3 * x;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[0] = t1 - t0;
  out[lane] = x;
}
template <int KIND> void run(const char *name, double per, int nact, double *d, unsigned long long *c)
{
  unsigned long long h = 0;
  for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64), 0, 0, d, c, 1.0000001, 0.9999999, nact); hipDeviceSynchronize(); }
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-58s lanes=%2d  %7.1f cycles per op (s_memtime ticks)\n", name, nact, (double)h / per);
}
int main()
{
  double *d; unsigned long long *c;
  hipMalloc(&d, 64 * 8); hipMalloc(&c, 8);
  for (int nact : {64, 12}) {
    run<0>("dependent v_fma_f64 chain", N, nact, d, c);
    run<1>("4 independent v_fma_f64 chains (per fma)", N, nact, d, c);
    run<2>("dependent mul,add (per op)", N, nact, d, c);
    run<3>("dependent v_rcp_f64 (+add)", N, nact, d, c);
    run<4>("dependent v_div_fixup_f64", N, nact, d, c);
    run<5>("dependent LDS read chain (incl. cvt+add)", N, nact, d, c);
    run<6>("fma + frexp/cmp/ballot/scalar branch (per iteration)", N, nact, d, c);
    run<7>("LDS write + dependent LDS read + add (per iteration)", N, nact, d, c);
    run<8>("rcp_refine + div_fast + add, dependent (per iteration)", N / 8, nact, d, c);
    run<9>("x = a_i - g_i*x, all operands VGPRs (per level)", N / 2, nact, d, c);
    run<10>("  + one ds_write_b64 per level (per level)", N / 2, nact, d, c);
    run<11>("  operands from LDS each trip of 4, no prefetch (per level)", N / 2, nact, d, c);
  }
  // clock calibration: s_memtime ticks per microsecond
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, c, 1.0000001, 0.9999999, 64); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("kernel %.1f us wall for %llu ticks in the timed loop -> >= %.0f ticks/us\n", ms * 1e3, h, h / (ms * 1e3));
  return 0;
}
