// vmm_fresh.hip -- what does memory the process has never had cost, by the way it is asked for?  (profiles/r6_vmm_fresh.md)
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/vmm_fresh.hip -o /tmp/vmm_fresh
//   /tmp/vmm_fresh malloc 64 4      fresh process: 16 x hipMalloc of 4 GB, each filled by a kernel
//   /tmp/vmm_fresh vmm 64 4         fresh process: one reserved range, 16 x (hipMemCreate + hipMemMap + hipMemSetAccess) of 4 GB, each filled
//   /tmp/vmm_fresh vmm 64 1         the same in chunks of 1 GB
//   /tmp/vmm_fresh busy 64 4        as `vmm`, while a kernel runs on another stream (does growing wait for the device?)
// profiles/r3_cold_start.md has the hipMalloc side: 30 - 40 ms per GB beyond the first ~16 GB of a process.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(uint32_t* p, size_t n, uint32_t salt) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i * 2654435761u + salt;
}
__global__ void spin(unsigned long long* out, long long cycles) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) { }
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}
int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "vmm";
  const size_t GB = (size_t)1 << 30;
  const size_t total = (size_t)(argc > 2 ? atoi(argv[2]) : 64) * GB, chunk = (size_t)(argc > 3 ? atoi(argv[3]) : 4) * GB;
  CK(hipSetDevice(0));
  double t = now_ms(); CK(hipFree(0)); printf("%s: runtime up in %.1f ms\n", mode, now_ms() - t);
  hipStream_t st2; CK(hipStreamCreate(&st2));
  unsigned long long* flag = nullptr;
  double sum_alloc = 0, sum_fill = 0;
  if (!strcmp(mode, "malloc")) {
    std::vector<void*> ps;
    for (size_t done = 0; done < total; done += chunk) {
      void* p = nullptr;
      t = now_ms(); CK(hipMalloc(&p, chunk)); const double ta = now_ms() - t;
      t = now_ms(); fill<<<4096, 256>>>((uint32_t*)p, chunk / 4, 3u); CK(hipDeviceSynchronize()); const double tf = now_ms() - t;
      printf("  %3zu -> %3zu GB: hipMalloc %8.1f ms, first fill %6.1f ms\n", done / GB, (done + chunk) / GB, ta, tf);
      sum_alloc += ta; sum_fill += tf; ps.push_back(p);
    }
    t = now_ms(); for (void* p : ps) fill<<<4096, 256>>>((uint32_t*)p, chunk / 4, 5u); CK(hipDeviceSynchronize());
    printf("  second fill of everything %.1f ms\n", now_ms() - t);
  } else {
    const bool busy = !strcmp(mode, "busy");
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    void* base = nullptr;
    t = now_ms(); CK(hipMemAddressReserve(&base, total, gran, nullptr, 0)); printf("  reserve %zu GB of addresses %.2f ms (granularity %zu)\n", total / GB, now_ms() - t, gran);
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (busy) { CK(hipMalloc(&flag, 8)); spin<<<256, 64, 0, st2>>>(flag, (long long)4e9); }  // ~2 s of a kernel on the other stream
    for (size_t done = 0; done < total; done += chunk) {
      hipMemGenericAllocationHandle_t hnd;
      t = now_ms(); CK(hipMemCreate(&hnd, chunk, &prop, 0)); const double t1 = now_ms();
      CK(hipMemMap((char*)base + done, chunk, 0, hnd, 0)); const double t2 = now_ms();
      CK(hipMemSetAccess((char*)base + done, chunk, &acc, 1)); const double t3 = now_ms();
      double tf = 0;
      if (!busy) { fill<<<4096, 256>>>((uint32_t*)((char*)base + done), chunk / 4, 3u); CK(hipDeviceSynchronize()); tf = now_ms() - t3; }
      printf("  %3zu -> %3zu GB: create %8.2f, map %6.2f, access %6.2f ms, first fill %6.1f ms\n", done / GB, (done + chunk) / GB, t1 - t, t2 - t1, t3 - t2, tf);
      sum_alloc += t3 - t; sum_fill += tf;
    }
    if (busy) { t = now_ms(); CK(hipDeviceSynchronize()); printf("  the spinning kernel ended %.1f ms after the last chunk was mapped\n", now_ms() - t); }
    t = now_ms(); fill<<<8192, 256>>>((uint32_t*)base, total / 4, 5u); CK(hipDeviceSynchronize());
    printf("  one fill over the whole range %.1f ms\n", now_ms() - t);
  }
  printf("%s %zu GB in chunks of %zu GB: allocation %.1f ms in all (%.2f ms per GB), first fills %.1f ms\n", mode, total / GB, chunk / GB, sum_alloc, sum_alloc / (double)(total / GB), sum_fill);
  return 0;
}
