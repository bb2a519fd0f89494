// dev_cache.hip -- see dev_cache.h
//
// Round 6: the blocks of a megabyte and more come from ONE heap per device -- an address range reserved once (hipMemAddressReserve), physical
// chunks of a gigabyte mapped behind what is there as the heap grows (hipMemCreate / hipMemMap / hipMemSetAccess), a best-fit free list with
// coalescing inside it.  What that buys over the size-class cache it replaces (rounds 3 - 5):
//   * memory the process has never had costs ~30 ms per GB on this driver whichever call asks for it, beyond the first ~32 GB of a process
//     (scripts/micro/vmm_fresh.hip, profiles/r6_vmm_fresh.md: hipMemCreate 120.7 ms per 4 GB from the 33rd GB on, as constant as a scrub at 33 GB/s
//     would be; hipMalloc the same in one 2 s lump).  A class cache strands a block in its class -- the 2.85 GB work buffer of the map phase could
//     not become a 3.5 GB ring arena of the align phase, so a process paid for the SUM of its classes' peaks; a heap pays for the peak of what is
//     live at once, and hands the map phase's memory to the align phase.
//   * the first WFM_POOL_GB (default 24, inside what the driver hands out for nothing) are mapped when the heap is made -- at the first
//     wfm_create of a device, outside any call that is timed or waited for;
//   * growth maps chunks BEHIND the heap: nothing is freed, nothing copied, nothing that is live moves.
// Blocks under a megabyte (job lists, counters) keep the size-class cache on plain hipMalloc: they are many, short-lived and cost nothing.
// If the virtual-memory calls fail on a driver (hipErrorNotSupported), the heap falls back to hipMalloc'ed slabs of 4 GB with the same free list
// inside each slab.
#include "dev_cache.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/wfmash_hip.h"

namespace {
constexpr size_t MB = (size_t)1 << 20, GB = (size_t)1 << 30;
constexpr size_t SMALL_LIMIT = MB;       // below: the class cache on hipMalloc
constexpr size_t HEAP_ALIGN = 64 << 10;  // block sizes and addresses of the heap
constexpr size_t CHUNK = GB;             // physical granule the heap grows by
constexpr size_t SLAB = 4 * GB;          // (fallback without the virtual-memory calls)

struct Live { size_t size; int dev; bool heap; };

// ---- the heap of one device ----
struct Heap {
  bool ready = false, vmm = false;
  int dev = 0;
  char* base = nullptr;
  size_t reserved = 0, committed = 0, live_bytes = 0, peak_live = 0;
  std::vector<hipMemGenericAllocationHandle_t> chunks;
  std::vector<std::pair<char*, size_t>> slabs;  // fallback: (base, length) of every hipMalloc'ed slab
  std::map<char*, size_t> free_by_addr;      // free blocks, address order (for coalescing)
  std::multimap<size_t, char*> free_by_len;  // the same, by length (best fit)
  hipMemAllocationProp prop = {};

  void insert_free(char* p, size_t len) {
    // coalesce with the neighbours (never across the edge of two slabs of the fallback: a slab goes back to the driver whole)
    auto nx = free_by_addr.lower_bound(p);
    if (nx != free_by_addr.end() && p + len == nx->first && (vmm || !slab_edge(nx->first))) {
      len += nx->second;
      erase_len(nx->second, nx->first);
      nx = free_by_addr.erase(nx);
    }
    if (nx != free_by_addr.begin()) {
      auto pv = std::prev(nx);
      if (pv->first + pv->second == p && (vmm || !slab_edge(p))) {
        p = pv->first; len += pv->second;
        erase_len(pv->second, pv->first);
        free_by_addr.erase(pv);
      }
    }
    free_by_addr[p] = len;
    free_by_len.emplace(len, p);
  }
  bool slab_edge(char* p) const {
    for (const auto& s : slabs) if (s.first == p) return true;
    return false;
  }
  void erase_len(size_t len, char* p) {
    auto r = free_by_len.equal_range(len);
    for (auto it = r.first; it != r.second; ++it) if (it->second == p) { free_by_len.erase(it); return; }
  }
  char* take(size_t len) {
    auto it = free_by_len.lower_bound(len);
    if (it == free_by_len.end()) return nullptr;
    char* p = it->second;
    const size_t have = it->first;
    free_by_len.erase(it);
    free_by_addr.erase(p);
    if (have > len) { free_by_addr[p + len] = have - len; free_by_len.emplace(have - len, p + len); }
    return p;
  }
  // maps `bytes` (a multiple of CHUNK) behind the committed prefix
  hipError_t commit(size_t bytes) {
    if (!vmm) {
      void* s = nullptr;
      const size_t n = bytes > SLAB ? bytes : SLAB;  // one slab that holds the request (a block cannot span two)
      hipError_t e = hipMalloc(&s, n);
      if (e != hipSuccess) return e;
      slabs.emplace_back((char*)s, n);
      committed += n;
      insert_free((char*)s, n);
      return hipSuccess;
    }
    if (committed + bytes > reserved) return hipErrorOutOfMemory;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t first = committed;
    for (size_t done = 0; done < bytes; done += CHUNK) {
      hipMemGenericAllocationHandle_t hnd;
      hipError_t e = hipMemCreate(&hnd, CHUNK, &prop, 0);
      if (e == hipSuccess) {
        e = hipMemMap(base + committed, CHUNK, 0, hnd, 0);
        if (e == hipSuccess) e = hipMemSetAccess(base + committed, CHUNK, &acc, 1);
        if (e != hipSuccess) { (void)hipMemUnmap(base + committed, CHUNK); (void)hipMemRelease(hnd); }
      }
      if (e != hipSuccess) {
        if (committed > first) insert_free(base + first, committed - first);
        return e;
      }
      chunks.push_back(hnd);
      committed += CHUNK;
    }
    insert_free(base + first, committed - first);
    return hipSuccess;
  }
  // gives the free chunks at the heap's end back to the driver (whole chunks only; what is live never moves); returns the bytes released
  size_t trim() {
    size_t released = 0;
    if (!vmm) {
      for (size_t i = 0; i < slabs.size();) {
        auto it = free_by_addr.find(slabs[i].first);
        if (it != free_by_addr.end() && it->second == slabs[i].second) {  // the slab is one free block
          erase_len(it->second, it->first);
          free_by_addr.erase(it);
          (void)hipFree(slabs[i].first);
          committed -= slabs[i].second; released += slabs[i].second;
          slabs.erase(slabs.begin() + (long)i);
        } else ++i;
      }
      return released;
    }
    if (free_by_addr.empty()) return 0;
    auto last = std::prev(free_by_addr.end());
    if (last->first + last->second != base + committed) return 0;
    const size_t start = (size_t)(last->first - base);
    const size_t keep_to = (start + CHUNK - 1) / CHUNK * CHUNK;  // first chunk boundary inside the free tail
    if (keep_to >= committed) return 0;
    const size_t nrel = (committed - keep_to) / CHUNK;
    char* p = last->first;
    const size_t len = last->second;
    erase_len(len, p);
    free_by_addr.erase(last);
    for (size_t i = 0; i < nrel; ++i) {
      committed -= CHUNK;
      (void)hipMemUnmap(base + committed, CHUNK);
      (void)hipMemRelease(chunks.back());
      chunks.pop_back();
      released += CHUNK;
    }
    if (keep_to > start) insert_free(p, keep_to - start);
    return released;
  }
};

struct Cache {
  std::mutex mu;
  std::unordered_map<void*, Live> live;                          // every block handed out (heap or class cache) or waiting in the class cache
  std::map<std::pair<int, size_t>, std::vector<void*>> free_by;  // class cache of the small blocks: (device, class size) -> waiting blocks
  size_t cached[64] = {};
  Heap heap[64];
};
Cache& cache() { static Cache* c = new Cache; return *c; }  // (never destroyed: handles may outlive static destructors)

size_t class_of(size_t bytes) {
  if (bytes < 4096) return 4096;
  int lg = 63 - __builtin_clzll((unsigned long long)bytes);
  const size_t step = (size_t)1 << (lg > 3 ? lg - 3 : 0);
  return (bytes + step - 1) / step * step;
}
bool dbg() { static const bool d = getenv("WFM_DEBUG") != nullptr; return d; }

// the heap of the current device, made at its first use (the first wfm_create of the device: wfa_host.hip asks for a block there)
Heap* heap_of(Cache& c, int dev) {
  if (dev < 0 || dev >= 64) return nullptr;
  Heap& H = c.heap[dev];
  if (H.ready) return &H;
  H.ready = true;
  H.dev = dev;
  static const bool vmm_off = getenv("WFM_VMM") && atoi(getenv("WFM_VMM")) == 0;
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); tot = 64 * GB; fr = tot; }
  H.prop.type = hipMemAllocationTypePinned;
  H.prop.location.type = hipMemLocationTypeDevice;
  H.prop.location.id = dev;
  const auto t0 = std::chrono::steady_clock::now();
  if (!vmm_off) {
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &H.prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran && CHUNK % gran == 0) {
      void* b = nullptr;
      const size_t want = tot / CHUNK * CHUNK;  // addresses cost nothing: the whole device
      if (hipMemAddressReserve(&b, want, gran, nullptr, 0) == hipSuccess) { H.base = (char*)b; H.reserved = want; H.vmm = true; }
    }
    (void)hipGetLastError();
  }
  // what is mapped at once: inside the ~32 GB a process gets for nothing, and never more than half of what is free (a shared device)
  const long long pool_gb = getenv("WFM_POOL_GB") ? atoll(getenv("WFM_POOL_GB")) : 24;
  size_t pool = (size_t)(pool_gb > 0 ? pool_gb : 0) * GB;
  if (pool > fr / 2) pool = fr / 2 / CHUNK * CHUNK;
  if (pool) {
    const hipError_t e = H.commit(pool);
    if (e != hipSuccess) (void)hipGetLastError();
  }
  if (dbg())
    fprintf(stderr, "[wfm] device %d: heap of %s, %.0f GB of addresses, %.1f GB mapped at once in %.1f ms\n", dev, H.vmm ? "mapped chunks" : "hipMalloc slabs", (double)H.reserved / (double)GB,
            (double)H.committed / (double)GB, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  return &H;
}

// What may wait per device in the class cache of the small blocks
size_t small_limit_bytes() { return (size_t)2 << 30; }

size_t trim_small_locked(Cache& c, std::vector<void*>& out, int dev) {
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
    else if (it->second.heap) {
      Heap& H = c.heap[it->second.dev];
      H.live_bytes -= it->second.size;
      H.insert_free((char*)p, it->second.size);
      c.live.erase(it);
    } else {
      const int dev = it->second.dev;
      c.free_by[{dev, it->second.size}].push_back(p);
      if (dev >= 0 && dev < 64) {
        c.cached[dev] += it->second.size;
        if (c.cached[dev] > small_limit_bytes()) trim_small_locked(c, evict, dev);
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
  if (bytes >= SMALL_LIMIT) {
    const size_t len = (bytes + HEAP_ALIGN - 1) / HEAP_ALIGN * HEAP_ALIGN;
    std::lock_guard<std::mutex> lk(c.mu);
    Heap* H = heap_of(c, dev);
    if (H) {
      char* q = H->take(len);
      if (!q) {
        // grow: what is missing behind the free block at the heap's end (if there is one), in whole chunks
        size_t tail = 0;
        if (H->vmm && !H->free_by_addr.empty()) {
          auto last = std::prev(H->free_by_addr.end());
          if (last->first + last->second == H->base + H->committed) tail = last->second;
        }
        const size_t need = H->vmm ? ((len - (tail < len ? tail : 0)) + CHUNK - 1) / CHUNK * CHUNK : len;
        const auto t0 = std::chrono::steady_clock::now();
        e = H->commit(need);
        if (e != hipSuccess) {  // the small blocks' cache back to the driver, the heap's free tail as well, and once more
          (void)hipGetLastError();
          std::vector<void*> out;
          trim_small_locked(c, out, dev);
          for (void* s : out) (void)hipFree(s);
          e = H->commit(need);
          if (e != hipSuccess) { (void)hipGetLastError(); return hipErrorOutOfMemory; }
        }
        if (dbg() && need >= 2 * GB)
          fprintf(stderr, "[wfm] device %d: heap grew by %.0f GB to %.0f GB in %.1f ms (%.1f GB live)\n", dev, (double)need / (double)GB, (double)H->committed / (double)GB,
                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), (double)H->live_bytes / (double)GB);
        q = H->take(len);
        if (!q) return hipErrorOutOfMemory;
      }
      H->live_bytes += len;
      if (H->live_bytes > H->peak_live) H->peak_live = H->live_bytes;
      c.live[q] = Live{len, dev, true};
      *p = q;
      return hipSuccess;
    }
  }
  const size_t cls = class_of(bytes);
  {
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.free_by.find({dev, cls});
    if (it != c.free_by.end() && !it->second.empty()) {
      *p = it->second.back();
      it->second.pop_back();
      if (dev >= 0 && dev < 64) c.cached[dev] -= cls;
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
  c.live[*p] = Live{cls, dev, false};
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
    bytes = trim_small_locked(c, out, -1);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < 64; ++d) {
      if (!c.heap[d].ready || !c.heap[d].committed) continue;
      if (d != cur) (void)hipSetDevice(d);
      (void)hipDeviceSynchronize();
      bytes += c.heap[d].trim();
      if (d != cur) (void)hipSetDevice(cur);
    }
  }
  for (void* q : out) (void)hipFree(q);
  return bytes;
}

void wfm_dcache_warm(void) {
  Cache& c = cache();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(c.mu);
  (void)heap_of(c, dev);
}

size_t wfm_dcache_stats(int dev, size_t* committed, size_t* live, size_t* peak_live) {
  Cache& c = cache();
  std::lock_guard<std::mutex> lk(c.mu);
  if (dev < 0 || dev >= 64 || !c.heap[dev].ready) { if (committed) *committed = 0; if (live) *live = 0; if (peak_live) *peak_live = 0; return 0; }
  const Heap& H = c.heap[dev];
  if (committed) *committed = H.committed;
  if (live) *live = H.live_bytes;
  if (peak_live) *peak_live = H.peak_live;
  return H.committed;
}

extern "C" size_t wfm_trim_device_cache(void) { return wfm_dcache_trim(); }
