// Does hipMemcpy2D (device rows of 32 bytes -> host rows of 8 bytes, width 8) write beyond the last host row?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
int main()
{
  int bad = 0;
  for (int n : {1, 2, 3, 64, 70, 187, 561, 1500, 4096, 9000, 12500, 65000, 100000}) {
    int *d; hipMalloc(&d, (size_t)n * 8 * sizeof(int));
    std::vector<int> src((size_t)n * 8);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (int)i;
    hipMemcpy(d, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice);
    std::vector<unsigned char> host((size_t)n * 8 + 256, 0xAB);
    hipError_t e = hipMemcpy2D(host.data(), 8, d + 4, 32, 8, (size_t)n, hipMemcpyDeviceToHost);
    int wrong = 0, over = 0;
    for (int c = 0; c < n; ++c) { int v[2]; memcpy(v, host.data() + (size_t)c * 8, 8); if (v[0] != c * 8 + 4 || v[1] != c * 8 + 5) ++wrong; }
    for (size_t i = (size_t)n * 8; i < host.size(); ++i) if (host[i] != 0xAB) ++over;
    printf("n=%6d rc=%d wrong rows %d, bytes written past the end %d\n", n, (int)e, wrong, over);
    bad += wrong + over;
    hipFree(d);
  }
  printf(bad ? "FAULTY\n" : "clean\n");
  return bad != 0;
}
