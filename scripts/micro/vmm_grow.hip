// vmm_grow.hip -- can an arena grow in place on this driver, and what does it cost?  
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/vmm_grow.hip -o /tmp/vmm_grow && /tmp/vmm_grow
// (a) hipMalloc / hipFree / hipMalloc of a larger block: the second allocation lands on memory the process has just freed;
// (b) an address range reserved once, physical chunks mapped behind what is there; the old bytes must still be there.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void fill(uint32_t* p, size_t n, uint32_t salt) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i * 2654435761u + salt;
}
__global__ void check(const uint32_t* p, size_t n, uint32_t salt, unsigned long long* bad) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (p[i] != (uint32_t)i * 2654435761u + salt) atomicAdd(bad, 1ull);
}

int main() {
  CK(hipSetDevice(0));
  const size_t GB = (size_t)1 << 30;
  {  // (a)
    void *a = nullptr, *b = nullptr, *c = nullptr;
    double t = now_ms(); CK(hipMalloc(&a, 4 * GB)); printf("hipMalloc 4 GB (fresh)            %8.1f ms\n", now_ms() - t);
    fill<<<4096, 256>>>((uint32_t*)a, GB, 1u); CK(hipDeviceSynchronize());
    t = now_ms(); CK(hipFree(a)); printf("hipFree 4 GB                      %8.1f ms\n", now_ms() - t);
    t = now_ms(); CK(hipMalloc(&b, 8 * GB)); printf("hipMalloc 8 GB after the free     %8.1f ms\n", now_ms() - t);
    t = now_ms(); CK(hipMalloc(&c, 8 * GB)); printf("hipMalloc 8 GB more               %8.1f ms\n", now_ms() - t);
    CK(hipFree(b)); CK(hipFree(c));
    t = now_ms(); CK(hipMalloc(&b, 16 * GB)); printf("hipMalloc 16 GB after freeing 16  %8.1f ms\n", now_ms() - t);
    CK(hipFree(b));
  }
  {  // (b)
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu bytes\n", gran);
    const size_t reserve = 64 * GB, chunk = 4 * GB;
    void* base = nullptr;
    double t = now_ms(); CK(hipMemAddressReserve(&base, reserve, gran, nullptr, 0)); printf("reserve 64 GB of addresses        %8.1f ms\n", now_ms() - t);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    unsigned long long* bad = nullptr;
    CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    size_t mapped = 0;
    for (int step = 0; step < 4; ++step) {
      hipMemGenericAllocationHandle_t hnd;
      t = now_ms();
      CK(hipMemCreate(&hnd, chunk, &prop, 0));
      const double t1 = now_ms();
      CK(hipMemMap((char*)base + mapped, chunk, 0, hnd, 0));
      const double t2 = now_ms();
      CK(hipMemSetAccess((char*)base + mapped, chunk, &acc, 1));
      const double t3 = now_ms();
      hs.push_back(hnd);
      printf("grow %2zu -> %2zu GB: create %7.1f, map %6.1f, access %6.1f ms\n", mapped / GB, (mapped + chunk) / GB, t1 - t, t2 - t1, t3 - t2);
      fill<<<4096, 256>>>((uint32_t*)((char*)base + mapped), chunk / 4, 7u + (uint32_t)step);
      CK(hipDeviceSynchronize());
      mapped += chunk;
      for (int s2 = 0; s2 <= step; ++s2) check<<<4096, 256>>>((const uint32_t*)((char*)base + (size_t)s2 * chunk), chunk / 4, 7u + (uint32_t)s2, bad);
      CK(hipDeviceSynchronize());
    }
    unsigned long long nbad = 0;
    CK(hipMemcpy(&nbad, bad, 8, hipMemcpyDeviceToHost));
    printf("words that changed while the arena grew: %llu\n", nbad);
    // one kernel over the whole range (chunks are contiguous in the address space)
    t = now_ms(); fill<<<8192, 256>>>((uint32_t*)base, mapped / 4, 99u); CK(hipDeviceSynchronize()); printf("fill of the whole 16 GB range     %8.1f ms\n", now_ms() - t);
    // shrink and grow again: does a chunk that was released come back wiped, and at what price?
    t = now_ms();
    CK(hipMemUnmap((char*)base + mapped - chunk, chunk)); CK(hipMemRelease(hs.back())); hs.pop_back(); mapped -= chunk;
    printf("unmap + release of the last chunk %8.1f ms\n", now_ms() - t);
    {
      hipMemGenericAllocationHandle_t hnd;
      t = now_ms(); CK(hipMemCreate(&hnd, chunk, &prop, 0)); const double t1 = now_ms();
      CK(hipMemMap((char*)base + mapped, chunk, 0, hnd, 0)); CK(hipMemSetAccess((char*)base + mapped, chunk, &acc, 1));
      printf("create after the release %7.1f ms, map + access %6.1f ms\n", t1 - t, now_ms() - t1);
      hs.push_back(hnd); mapped += chunk;
    }
    CK(hipMemUnmap(base, mapped));
    for (auto hnd : hs) CK(hipMemRelease(hnd));
    CK(hipMemAddressFree(base, reserve));
    CK(hipFree(bad));
  }
  printf("done\n");
  return 0;
}
