// several allocations held at once: is it the size of one hipMalloc or the total that makes it slow?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  (void)hipFree(0);
  for (size_t gb : {8, 16, 24}) {
    std::vector<void*> ps;
    double total = 0;
    printf("%zu GB pieces:", gb);
    for (int i = 0; i < (gb == 8 ? 16 : (gb == 16 ? 10 : 6)); ++i) {
      void* p = nullptr;
      const double t0 = now();
      if (hipMalloc(&p, gb << 30) != hipSuccess) { printf(" fail"); break; }
      const double dt = now() - t0;
      total += dt;
      printf(" %.0f", dt);
      ps.push_back(p);
    }
    printf(" ms (sum %.0f ms for %zu GB)\n", total, ps.size() * gb);
    const double t0 = now();
    for (void* p : ps) (void)hipFree(p);
    printf("  freed in %.0f ms\n", now() - t0);
  }
  // and again after everything was freed once
  for (int rep = 0; rep < 2; ++rep) {
    void* p = nullptr;
    const double t0 = now();
    (void)hipMalloc(&p, (size_t)20 << 30);
    printf("20 GB after the frees: %.0f ms\n", now() - t0);
    (void)hipFree(p);
  }
  return 0;
}
