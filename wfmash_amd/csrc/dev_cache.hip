// dev_cache.hip -- see dev_cache.h
#include "dev_cache.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/wfmash_hip.h"

namespace {
struct Live { size_t size; int dev; };
struct Cache {
  std::mutex mu;
  std::unordered_map<void*, Live> live;                          // every block handed out or cached
  std::map<std::pair<int, size_t>, std::vector<void*>> free_by;  // (device, class size) -> cached blocks
  size_t cached[64] = {};                                        // bytes waiting per device
};
Cache& cache() { static Cache* c = new Cache; return *c; }  // (never destroyed: handles may outlive static destructors)

size_t class_of(size_t bytes) {
  if (bytes < 4096) return 4096;
  int lg = 63 - __builtin_clzll((unsigned long long)bytes);
  const size_t step = (size_t)1 << (lg > 3 ? lg - 3 : 0);
  return (bytes + step - 1) / step * step;
}
// What may wait per device: a third of that device's memory (96 GB of an MI355X's 288), or WFM_DEV_CACHE_GB
size_t limit_bytes(int dev) {
  static const long long env = getenv("WFM_DEV_CACHE_GB") ? atoll(getenv("WFM_DEV_CACHE_GB")) : -1;
  if (env >= 0) return (size_t)env << 30;
  static size_t per_dev[64] = {};
  if (dev < 0 || dev >= 64) return (size_t)8 << 30;
  if (!per_dev[dev]) {
    size_t fr = 0, tot = 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); tot = (size_t)24 << 30; }
    if (cur != dev) (void)hipSetDevice(cur);
    per_dev[dev] = tot / 3;
  }
  return per_dev[dev];
}
// dev < 0: every device
size_t trim_locked(Cache& c, std::vector<void*>& out, int dev) {
  size_t bytes = 0;
  for (auto it = c.free_by.begin(); it != c.free_by.end();) {
    if (dev >= 0 && it->first.first != dev) { ++it; continue; }
    for (void* p : it->second) { out.push_back(p); bytes += it->first.second; c.live.erase(p); }
    const int d = it->first.first;
    if (d >= 0 && d < 64) c.cached[d] = 0;
    it = c.free_by.erase(it);
  }
  return bytes;
}
void put(void* p) {
  Cache& c = cache();
  std::vector<void*> evict;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.live.find(p);
    if (it == c.live.end()) { evict.push_back(p); }  // not ours (should not happen): plain hipFree
    else {
      const int dev = it->second.dev;
      c.free_by[{dev, it->second.size}].push_back(p);
      if (dev >= 0 && dev < 64) {
        c.cached[dev] += it->second.size;
        if (c.cached[dev] > limit_bytes(dev)) trim_locked(c, evict, dev);  // only the device that is over: the others are not stalled
      }
    }
  }
  for (void* q : evict) (void)hipFree(q);
}
}  // namespace

hipError_t wfm_dmalloc(void** p, size_t bytes) {
  Cache& c = cache();
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const size_t cls = class_of(bytes);
  {
    std::lock_guard<std::mutex> lk(c.mu);
    // the block's own class, or -- for blocks of 64 MB and more -- the smallest cached block of the device that holds the request and is less
    // than half as large again: a first hipMalloc of a gigabyte costs 30 - 70 ms on this driver whenever the process has used memory before
    // (gpurun_out/r5s.err: "device block of 2.85 GB took 171.6 ms"), a cached block of another call's size class costs nothing
    auto it = c.free_by.lower_bound({dev, cls});
    while (it != c.free_by.end() && it->first.first == dev && it->second.empty()) ++it;
    if (it != c.free_by.end() && it->first.first == dev && !it->second.empty() &&
        (it->first.second == cls || (cls >= ((size_t)64 << 20) && it->first.second <= cls + cls / 2))) {
      *p = it->second.back();
      it->second.pop_back();
      if (dev >= 0 && dev < 64) c.cached[dev] -= it->first.second;
      return hipSuccess;
    }
  }
  e = hipMalloc(p, cls);
  if (e != hipSuccess) {  // give everything cached back and try once more
    (void)hipGetLastError();
    (void)wfm_dcache_trim();
    e = hipMalloc(p, cls);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> lk(c.mu);
  c.live[*p] = Live{cls, dev};
  return hipSuccess;
}

void wfm_dfree_nosync(void* p) { if (p) put(p); }
void wfm_dfree(void* p) {
  if (!p) return;
  int dev = -1, cur = 0;
  {
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.live.find(p);
    if (it != c.live.end()) dev = it->second.dev;
  }
  (void)hipGetDevice(&cur);
  if (dev >= 0 && dev != cur) (void)hipSetDevice(dev);
  (void)hipDeviceSynchronize();
  if (dev >= 0 && dev != cur) (void)hipSetDevice(cur);
  put(p);
}

size_t wfm_dcache_trim(void) {
  Cache& c = cache();
  std::vector<void*> out;
  size_t bytes;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    bytes = trim_locked(c, out, -1);
  }
  for (void* q : out) (void)hipFree(q);
  return bytes;
}

extern "C" size_t wfm_trim_device_cache(void) { return wfm_dcache_trim(); }
