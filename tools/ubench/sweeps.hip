// Micro-benchmark of the manager wave's serial sweeps of k_column_ps, alone in a workgroup with the kernel's LDS
// layout: shader cycles per level of each sweep.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -mllvm -disable-machine-licm \
//         -I mckpp_f90_amd/csrc tools/ubench/sweeps.hip -o tools/ubench/sweeps && tools/ubench/sweeps [W nz waves]
#include "../../mckpp_f90_amd/csrc/mckpp_kernels_ps.hip"

namespace {
enum { T_FUSED_FWD = 0, T_BACK, T_V_FWD, T_V_BACK, T_SCAN, T_CHAIN, T_2E_FWD, T_2E_BACK, T_2E_V, T_FUSED_FWD_OLD, T_2E_FWD_OLD, T_COUNT };
const char *t_name[T_COUNT] = {"U,T,S forward", "back substitution U,T,S", "V forward", "V back substitution", "bulk-Ri scan",
                               "register chain x = a_i - g_i x, nz levels (reference: no LDS)",
                               "two-ended U,T,S forward (waves 0 and 1)", "two-ended U,T,S middle + substitutions (waves 0 and 1)",
                               "two-ended V, all of it (one wave)",
                               "U,T,S forward, the compiler's two-level trip (ps_thomas_uts_fwd: rounds 3-4, double diffusion)",
                               "two-ended U,T,S forward, the compiler's two-level trip (ps_thomas2_uts_fwd: rounds 4-5, double diffusion)"};

template <int TEST>
__global__ __launch_bounds__(1024, 4) void k_sweep(int W, int nz, int busy, unsigned long long *cyc, double *out, int SS)
{
  extern __shared__ double lds[];
  constexpr int XV = 0, ROWS = Q_COUNT;
  const int nzp1 = nz + 1, L = nzp1 + 2, NL = ps_nl(L);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  double *cst = lds, *slots = lds + K_STRIDE * NL + 2, *screc = slots + W * SS;
  int *sirec = reinterpret_cast<int *>(screc + W * C_COUNT), *s_flags = sirec + W * I_COUNT;
  for (int i = tid; i < NL; i += blockDim.x) {
    cst[i * K_STRIDE + K_ZM] = -3.3 * i; cst[i * K_STRIDE + K_HM] = 3.3;
    cst[i * K_STRIDE + K_T0] = 300.0 + 0.1 * i; cst[i * K_STRIDE + K_T1] = 310.0 - 0.1 * i;
  }
  for (int i = tid; i < W * SS; i += blockDim.x) slots[i] = 1.e-3 * (1.0 + (i % 97) * 0.01);
  for (int i = tid; i < W * I_COUNT; i += blockDim.x) sirec[i] = (i % I_COUNT) == I_ACT ? 1 : 0;
  if (tid < 4) s_flags[tid] = 0;
  __syncthreads();
  for (int i = tid; i < W * NL; i += blockDim.x) {   // right-hand sides / solutions O(10), pivots O(1)
    const int sl = i / NL, k = i - sl * NL;
    double *my = slots + sl * SS + k * ROWS;
    my[Q_YU] = 0.1 + 0.001 * k; my[Q_YT] = 10.0 + 0.01 * k; my[Q_YS] = 0.2 - 0.001 * k; my[Q_YV] = 0.05 + 0.001 * k;
    if (TEST == T_FUSED_FWD || TEST == T_FUSED_FWD_OLD || TEST == T_2E_FWD || TEST == T_2E_FWD_OLD) { my[Q_DM] = 0.31 + 0.001 * k; my[Q_DT] = 0.29; my[Q_GM] = 0.3; my[Q_BET] = 0.28 + 0.001 * k; }   // p, q
    if (TEST == T_V_FWD || TEST == T_2E_V) { my[Q_BET] = 1.5 + 0.001 * k; my[Q_DT] = 1. / (1.5 + 0.001 * k); my[Q_GM] = 0.3; }   // pivots, reciprocals, q
    if (TEST == T_BACK || TEST == T_V_BACK || TEST == T_2E_BACK) { my[Q_DS] = -0.2; my[Q_BET] = -0.2; }   // gam
    if (TEST == T_2E_V) { my[Q_DS] = -0.2; my[Q_DM] = -0.2; }   // gam at the ends of its row, multipliers of the substitution
  }
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  double sink = 0.0;
  if (TEST == T_2E_FWD || TEST == T_2E_FWD_OLD || TEST == T_2E_BACK) {   // both waves; the time of wave 0 from its start to the barrier behind them
    t0 = __builtin_amdgcn_s_memtime();
    if (TEST == T_2E_FWD) {
      if (wv == 0) ps_thomas2_uts_fwd4<XV, 1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
      if (wv == 1) ps_thomas2_uts_fwd4<XV, -1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
    } else if (TEST == T_2E_FWD_OLD) {
      if (wv == 0) ps_thomas2_uts_fwd<XV, 1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
      if (wv == 1) ps_thomas2_uts_fwd<XV, -1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
    } else {
      if (wv == 0) ps_thomas2_uts_back<XV, 1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
      if (wv == 1) ps_thomas2_uts_back<XV, -1>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
    }
    __syncthreads();
    t1 = __builtin_amdgcn_s_memtime();
  } else if (wv == 0) {
    t0 = __builtin_amdgcn_s_memtime();
    if (TEST == T_FUSED_FWD)
      ps_thomas_uts_fwd4<XV>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
    if (TEST == T_FUSED_FWD_OLD)
      ps_thomas_uts_fwd<XV>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
    if (TEST == T_BACK) ps_thomas_uts_back<XV>(W, slots, SS, nz, sirec + I_ACT, I_COUNT, lane);
    if (TEST == T_V_FWD) ps_thomas_v_fwd(W, slots, SS, ROWS, nz, sirec + I_ACT, I_COUNT, lane);
    if (TEST == T_V_BACK) ps_thomas_v_back(W, slots, SS, ROWS, ps_sysrows<XV>::gam_m, nz, sirec + I_ACT, I_COUNT, lane);
    if (TEST == T_2E_V) ps_thomas2_v(W, slots, SS, ROWS, ps_sysrows<XV>::gam_m, nz, sirec + I_ACT, I_COUNT, sirec + I_BAD, I_COUNT, lane);
    if (TEST == T_SCAN) { int out[2]; ps_scan_rib(W, Q_YV, slots, SS, ROWS, 2, nz, sirec + I_ACT, I_COUNT, lane, 1.e300 /* never crossed: the whole column */, out); sink = out[0]; }
    if (TEST == T_CHAIN) {
      double a0 = slots[lane], a1 = slots[lane + 64], a2 = slots[lane + 128], a3 = slots[lane + 192];
      double g0 = slots[lane + 256], g1 = slots[lane + 320], g2 = slots[lane + 384], g3 = slots[lane + 448], x = 1.0 + lane;
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3));
      t0 = __builtin_amdgcn_s_memtime();
      for (int i = 0; i < nz / 4; ++i) { x = a0 - g0 * x; x = a1 - g1 * x; x = a2 - g2 * x; x = a3 - g3 * x; }
      sink = x;
    }
    t1 = __builtin_amdgcn_s_memtime();
  } else if (wv >= 1 && busy) {   // the other waves: fp64 work as a level phase of another workgroup would issue
    double x = 1.0 + lane, y = 0.999, z = 1e-3;
    for (int i = 0; i < busy; ++i) { x = __builtin_fma(x, y, z); y = __builtin_fma(y, 0.9999, z); }
    sink = x + y;
  }
  __syncthreads();
  if (tid == 0) { cyc[0] = t1 - t0; cyc[1] = 0; }
  if (sink == 12345.678) out[tid] = sink;
  if (tid < W) out[tid] = slots[tid * SS + 5 * ROWS + Q_YU];
}

template <int TEST> void run(int W, int nz, int waves, int busy, unsigned long long *dc, double *dout)
{
  const int L = nz + 3;
  const size_t lds = ps_lds_bytes(L, W, 0);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<TEST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  unsigned long long best = ~0ull, h[2] = {0, 0};
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k_sweep<TEST>, dim3(1), dim3(64 * waves), lds, 0, W, nz, busy, dc, dout, ps_ss(L, 0, W));
    hipDeviceSynchronize();
    hipMemcpy(h, dc, sizeof h, hipMemcpyDeviceToHost);
    if (h[0] < best) best = h[0];
  }
  printf("%-52s W=%2d nz=%3d waves=%2d busy=%d: %7llu cycles, %6.1f per level%s\n", t_name[TEST], W, nz, waves, busy, best,
         (double)best / nz, h[1] ? "  [STUCK]" : "");
}
}  // namespace

int main(int argc, char **argv)
{
  int W = argc > 1 ? atoi(argv[1]) : 15, nz = argc > 2 ? atoi(argv[2]) : 60, waves = argc > 3 ? atoi(argv[3]) : 8;
  unsigned long long *dc;
  double *dout;
  hipMalloc(&dc, 64);
  hipMalloc(&dout, 1024 * sizeof(double));
  for (int busy = 0; busy <= 20000; busy += 20000) {
    run<T_FUSED_FWD>(W, nz, waves, busy, dc, dout);
    run<T_FUSED_FWD_OLD>(W, nz, waves, busy, dc, dout);
    run<T_BACK>(W, nz, waves, busy, dc, dout);
    run<T_V_FWD>(W, nz, waves, busy, dc, dout);
    run<T_V_BACK>(W, nz, waves, busy, dc, dout);
    run<T_SCAN>(W, nz, waves, busy, dc, dout);
    run<T_CHAIN>(W, nz, waves, busy, dc, dout);
    if (busy == 0) {
      run<T_2E_FWD>(W, nz, waves, busy, dc, dout);
      run<T_2E_FWD_OLD>(W, nz, waves, busy, dc, dout);
      run<T_2E_BACK>(W, nz, waves, busy, dc, dout);
      run<T_2E_V>(W, nz, waves, busy, dc, dout);
    }
  }
  return 0;
}
