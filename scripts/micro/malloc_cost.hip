#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  size_t fr, tot; hipMemGetInfo(&fr, &tot); printf("free %.1f GB total %.1f GB\n", fr / 1e9, tot / 1e9);
  for (int rep = 0; rep < 2; ++rep)
  for (size_t gb : {1, 4, 16, 64, 110}) {
    void* p = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc(&p, gb << 30);
    double t1 = now();
    if (e != hipSuccess) { printf("%zu GB: malloc failed\n", gb); continue; }
    hipMemset(p, 0, gb << 30); hipDeviceSynchronize();
    double t2 = now();
    hipMemset(p, 1, gb << 30); hipDeviceSynchronize();
    double t3 = now();
    hipFree(p);
    double t4 = now();
    printf("rep %d %3zu GB: malloc %.1f ms, first memset %.1f ms, second memset %.1f ms, free %.1f ms\n", rep, gb, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
  }
  return 0;
}
