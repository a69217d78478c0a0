// hwid.hip - where do the waves of co-resident workgroups sit?  Launches the column kernel's shape (512 threads,
// 80 KB of LDS, two workgroups per CU, 512 workgroups) and records HW_REG_HW_ID + HW_REG_XCC_ID of every wave.
// Question: do the wave-0s (the manager waves) of the two workgroups of a CU share a SIMD?
//   hipcc -O2 --offload-arch=gfx950 hwid.hip -o hwid && ./hwid [threads] [lds_bytes] [nblocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void k_hwid(unsigned *out, int *arrived, int nblocks) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  unsigned hw = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4);    // HW_REG_HW_ID, all 32 bits
  unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20);   // HW_REG_XCC_ID[3:0]
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * nw + wave) * 2] = hw;
    out[(blockIdx.x * nw + wave) * 2 + 1] = xcc;
  }
  lds[threadIdx.x] = hw;
  // hold the workgroup until all have arrived (bounded), so that the co-residency is the column kernel's
  if (threadIdx.x == 0) {
    atomicAdd(arrived, 1);
    for (int i = 0; i < 2000000; ++i)
      if (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nblocks) break;
  }
  __syncthreads();
  if (lds[threadIdx.x] < 0) out[0] = 0;
}

int main(int argc, char **argv) {
  int threads = argc > 1 ? atoi(argv[1]) : 512;
  int ldsb = argc > 2 ? atoi(argv[2]) : 79896;
  int nblocks = argc > 3 ? atoi(argv[3]) : 512;
  int nw = threads / 64;
  unsigned *d_out; int *d_arr;
  hipMalloc(&d_out, sizeof(unsigned) * 2 * nblocks * nw);
  hipMalloc(&d_arr, sizeof(int));
  hipFuncSetAttribute((const void *)k_hwid, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(d_arr, 0, sizeof(int));
    hipLaunchKernelGGL(k_hwid, dim3(nblocks), dim3(threads), ldsb, 0, d_out, d_arr, nblocks);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<unsigned> h(2 * nblocks * nw);
    hipMemcpy(h.data(), d_out, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
    // key = xcc, se, sh, cu
    std::map<unsigned, std::vector<int>> cu_wgs;
    int pattern_rr = 0;
    for (int b = 0; b < nblocks; ++b) {
      unsigned hw0 = h[(b * nw) * 2], xcc = h[(b * nw) * 2 + 1] & 15;
      unsigned key = (xcc << 16) | (hw0 & 0xff00);   // CU_ID[11:8], SH_ID[12], SE_ID[15:13]
      cu_wgs[key].push_back(b);
      bool rr = true;
      for (int w = 1; w < nw; ++w) {
        unsigned s0 = (hw0 >> 4) & 3, sw = (h[(b * nw + w) * 2] >> 4) & 3;
        if (sw != ((s0 + w) & 3)) rr = false;
      }
      pattern_rr += rr;
    }
    int hist[8] = {0}, same = 0, two = 0;
    for (auto &kv : cu_wgs) {
      int n = (int)kv.second.size();
      hist[n < 7 ? n : 7]++;
      if (n == 2) {
        ++two;
        unsigned a = h[(kv.second[0] * nw) * 2], b = h[(kv.second[1] * nw) * 2];
        if (((a >> 4) & 3) == ((b >> 4) & 3)) ++same;
      }
    }
    printf("rep %d: %zu CUs; workgroups per CU histogram:", rep, cu_wgs.size());
    for (int i = 0; i < 8; ++i) if (hist[i]) printf(" %d:%d", i, hist[i]);
    printf("; waves round-robin over SIMDs from wave 0's: %d of %d workgroups; CUs with two workgroups whose wave-0s share a SIMD: %d of %d\n",
           pattern_rr, nblocks, same, two);
    if (rep == 0) {
      int shown = 0;
      for (auto &kv : cu_wgs) {
        if (shown++ >= 6) break;
        printf("  cu key %06x:", kv.first);
        for (int b : kv.second) {
          printf("  wg %d tg_id %u simd of waves", b, (h[(b * nw) * 2] >> 16) & 15);
          for (int w = 0; w < nw; ++w) printf(" %u", (h[(b * nw + w) * 2] >> 4) & 3);
          printf(" wave_id");
          for (int w = 0; w < nw; ++w) printf(" %u", h[(b * nw + w) * 2] & 15);
          printf(";");
        }
        printf("\n");
      }
    }
  }
  return 0;
}
