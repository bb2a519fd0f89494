// map_index.hip -- device-resident reference index of the map path.
//
// Replaces Sketch::build's index stage (src/map/include/winSketch.hpp:266-429): k-mer (minmer)
// frequency filter, the position lookup hash -> [IntervalPoint] (OPEN/CLOSE pairs, contiguous
// intervals of one hash fused) and the minmerIndex vector.  The reference builds per-thread
// hash maps and merges them; here the minmer intervals of all target sequences are ONE array in
// HBM that is radix-sorted by hash (rocPRIM, stable, so the (seqId, wpos) order inside a hash is
// kept), after which every step is a scan:
//   group heads -> frequency per unique hash -> threshold (host, from the histogram)
//   keep flags  -> chain heads (wpos != previous wpos_end) -> OPEN / CLOSE points
// Lookup is a binary search in the sorted unique-hash array instead of a hash map.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>  // rocprim's texture iterator calls host memset

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/wfmash_hip.h"
#include "wfa_handle.h"
#include "dev_cache.h"
#include "map_device.h"

struct wfm_index {
  int device = 0;
  int64_t n_windows = 0;       // minmer intervals given
  int64_t n_kept = 0;          // after the frequency filter (= minmerIndex size)
  int64_t n_unique = 0;        // unique hashes kept (= minmerPosLookupIndex size)
  int64_t n_points = 0;        // interval points
  uint64_t threshold = 0;
  int64_t filtered = 0;
  int adjusted = 0;
  uint64_t* d_uhash = nullptr;          // [n_unique] ascending
  int64_t* d_poff = nullptr;            // [n_unique + 1] offsets into points
  wfm_interval_point_t* d_points = nullptr;
  wfm_minmer_t* d_minmers = nullptr;    // [n_kept] in (seqId, wpos) input order
};

namespace {

#define HIPCHK(h, call)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      wfm_set_error((h), std::string(#call) + ": " + hipGetErrorString(e_));            \
      return WFM_E_HIP;                                                                 \
    }                                                                                   \
  } while (0)

struct Scratch {
  std::vector<void*> p;
  ~Scratch() { if (!p.empty()) (void)hipDeviceSynchronize(); for (void* q : p) if (q) wfm_dfree_nosync(q); }
  template <typename T> hipError_t alloc(T** out, size_t n) {
    hipError_t e = wfm_dmalloc((void**)out, std::max<size_t>(n, 1) * sizeof(T));
    if (e == hipSuccess) p.push_back(*out);
    return e;
  }
};

__global__ void iota_u32(uint32_t* v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}
__global__ void gather_hash(const wfm_minmer_t* m, uint64_t* k, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) k[i] = m[i].hash;
}
// head[j] = 1 iff sorted position j starts a new hash group
__global__ void mark_group_heads(const uint64_t* keys, uint32_t* head, int64_t n) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) head[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1u : 0u;
}
// group id (exclusive scan of heads, minus one) -> group start positions
__global__ void group_starts(const uint32_t* head, const uint32_t* gid_incl, int64_t* gstart, uint64_t* ghash, const uint64_t* keys, int64_t n) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n && head[j]) { gstart[gid_incl[j] - 1] = j; ghash[gid_incl[j] - 1] = keys[j]; }
}
__global__ void group_freq(const int64_t* gstart, uint32_t* freq, int64_t ng, int64_t n) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < ng) freq[g] = (uint32_t)((g + 1 < ng ? gstart[g + 1] : n) - gstart[g]);
}
// keep[j] for sorted position j; chain head = first kept entry of its hash, or wpos != previous wpos_end
__global__ void mark_keep_and_chain(const wfm_minmer_t* m, const uint32_t* order, const uint32_t* head, const uint32_t* gid_incl,
                                    const uint32_t* freq, uint64_t thr, uint32_t* keep, uint32_t* chead, uint32_t* gkeep, int64_t n) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t g = gid_incl[j] - 1;
  const uint32_t f = freq[g];
  const bool kp = !((uint64_t)f > thr && f > 10u);  // winSketch.hpp:381 (min_occ = 10)
  keep[j] = kp ? 1u : 0u;
  bool ch = false;
  if (kp) {
    if (head[j]) { ch = true; gkeep[g] = 1u; }
    else ch = m[order[j]].wpos != m[order[j - 1]].wpos_end;  // pos_list.back().pos != mi.wpos (winSketch.hpp:385-391)
  }
  chead[j] = ch ? 1u : 0u;
}
// one OPEN/CLOSE pair per chain: OPEN at the chain head's wpos, CLOSE at the last member's wpos_end
__global__ void emit_points(const wfm_minmer_t* m, const uint32_t* order, const uint32_t* keep, const uint32_t* chead,
                            const uint32_t* chain_incl, wfm_interval_point_t* pts, int64_t n) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n || !keep[j]) return;
  const wfm_minmer_t& mi = m[order[j]];
  const int64_t c = (int64_t)chain_incl[j] - 1;
  const bool last = (j + 1 >= n) || !keep[j + 1] || chead[j + 1];
  if (chead[j]) {
    wfm_interval_point_t o; o.pos = mi.wpos; o.hash = mi.hash; o.seqId = mi.seqId; o.side = 1; o.pad_[0] = o.pad_[1] = o.pad_[2] = 0;
    pts[2 * c] = o;
    // the CLOSE point is created together with the OPEN point and keeps the head's seqId even when a
    // later interval (possibly of the next sequence) extends it: only .pos is updated (winSketch.hpp:388-391)
    pts[2 * c + 1].hash = mi.hash; pts[2 * c + 1].seqId = mi.seqId; pts[2 * c + 1].side = -1;
    pts[2 * c + 1].pad_[0] = pts[2 * c + 1].pad_[1] = pts[2 * c + 1].pad_[2] = 0;
  }
  if (last) pts[2 * c + 1].pos = mi.wpos_end;
}
// per kept hash group: unique hash + offset of its first point
__global__ void emit_group_offsets(const int64_t* gstart, const uint64_t* ghash, const uint32_t* gkeep, const uint32_t* gkeep_incl,
                                   const uint32_t* chain_incl, const uint32_t* chead, uint64_t* uhash, int64_t* poff, int64_t ng) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng || !gkeep[g]) return;
  const int64_t u = (int64_t)gkeep_incl[g] - 1;
  const int64_t j = gstart[g];
  uhash[u] = ghash[g];
  poff[u] = 2 * ((int64_t)chain_incl[j] - (chead[j] ? 1 : 0));  // chains before this group
}
// minmerIndex: kept entries in input order
__global__ void mark_keep_input_order(const uint32_t* order, const uint32_t* keep, uint32_t* keep_in, int64_t n) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) keep_in[order[j]] = keep[j];
}
__global__ void compact_minmers(const wfm_minmer_t* m, const uint32_t* keep_in, const uint32_t* keep_in_incl, wfm_minmer_t* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keep_in[i]) out[keep_in_incl[i] - 1] = m[i];
}

template <typename T>
int inclusive_scan(wfm_handle_t* h, Scratch& sc, const T* in, T* out, int64_t n, hipStream_t st) {
  size_t tmp = 0;
  HIPCHK(h, rocprim::inclusive_scan(nullptr, tmp, in, out, (size_t)n, rocprim::plus<T>(), st));
  void* d_tmp = nullptr;
  HIPCHK(h, sc.alloc((char**)&d_tmp, tmp));
  HIPCHK(h, rocprim::inclusive_scan(d_tmp, tmp, in, out, (size_t)n, rocprim::plus<T>(), st));
  return WFM_OK;
}

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" {

int wfm_index_build(wfm_handle_t* h, const wfm_minmer_t* minmers, int64_t n, double max_kmer_freq, wfm_index_t** out) {
  if (!h || !out || (n && !minmers) || n < 0 || n >= (int64_t)1 << 31) return WFM_E_ARG;
  *out = nullptr;
  HIPCHK(h, hipSetDevice(wfm_device(h)));
  Scratch sc;
  wfm_minmer_t* d_m = nullptr;
  if (n > 0) {
    if (sc.alloc(&d_m, (size_t)n) != hipSuccess) { wfm_set_error(h, "out of device memory (index build)"); return WFM_E_NOMEM; }
    HIPCHK(h, hipMemcpyAsync(d_m, minmers, (size_t)n * sizeof(wfm_minmer_t), hipMemcpyHostToDevice, wfm_stream(h)));
  }
  return map_index_build_device(h, d_m, n, max_kmer_freq, out);
}

}  // extern "C"

// the index of n minmer intervals that are already on the device (input order = the reference's minmerIndex order)
int map_index_build_device(wfm_handle_t* h, const wfm_minmer_t* d_m, int64_t n, double max_kmer_freq, wfm_index_t** out) {
  if (!h || !out || n < 0 || n >= (int64_t)1 << 31 || (n && !d_m)) return WFM_E_ARG;
  *out = nullptr;
  HIPCHK(h, hipSetDevice(wfm_device(h)));
  hipStream_t st = wfm_stream(h);
  wfm_index* ix = new wfm_index();
  ix->device = wfm_device(h);
  ix->n_windows = n;
  if (n == 0) { *out = ix; return WFM_OK; }
  Scratch sc;
  uint64_t *d_k = nullptr, *d_k2 = nullptr, *d_ghash = nullptr;
  uint32_t *d_ord = nullptr, *d_ord2 = nullptr, *d_head = nullptr, *d_gid = nullptr, *d_freq = nullptr, *d_keep = nullptr,
           *d_chead = nullptr, *d_chain = nullptr, *d_gkeep = nullptr, *d_gkeep_incl = nullptr, *d_keep_in = nullptr, *d_keep_in_incl = nullptr;
  int64_t* d_gstart = nullptr;
#define ALLOC(p, cnt) do { if (sc.alloc(&(p), (size_t)(cnt)) != hipSuccess) { delete ix; wfm_set_error(h, "out of device memory (index build)"); return WFM_E_NOMEM; } } while (0)
  ALLOC(d_k, n); ALLOC(d_k2, n); ALLOC(d_ord, n); ALLOC(d_ord2, n); ALLOC(d_head, n); ALLOC(d_gid, n);
  ALLOC(d_keep, n); ALLOC(d_chead, n); ALLOC(d_chain, n); ALLOC(d_keep_in, n); ALLOC(d_keep_in_incl, n);
  const bool dbg = getenv("WFM_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  if (dbg) fprintf(stderr, "[wfm] index_build: %.1f MB of minmers", (double)n * 32 / 1e6);
  const double t_b = now();
  hipLaunchKernelGGL(gather_hash, grid_for(n), dim3(256), 0, st, d_m, d_k, n);
  hipLaunchKernelGGL(iota_u32, grid_for(n), dim3(256), 0, st, d_ord, n);
  {  // stable radix sort of (hash, input index)
    size_t tmp = 0;
    HIPCHK(h, rocprim::radix_sort_pairs(nullptr, tmp, d_k, d_k2, d_ord, d_ord2, (size_t)n, 0, 64, st));
    void* d_tmp = nullptr;
    ALLOC(*(char**)&d_tmp, tmp);
    HIPCHK(h, rocprim::radix_sort_pairs(d_tmp, tmp, d_k, d_k2, d_ord, d_ord2, (size_t)n, 0, 64, st));
  }
  hipLaunchKernelGGL(mark_group_heads, grid_for(n), dim3(256), 0, st, d_k2, d_head, n);
  int rc = inclusive_scan<uint32_t>(h, sc, d_head, d_gid, n, st);
  if (rc != WFM_OK) { delete ix; return rc; }
  uint32_t ng32 = 0;
  HIPCHK(h, hipMemcpyAsync(&ng32, d_gid + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  const int64_t ng = ng32;
  ALLOC(d_gstart, ng); ALLOC(d_ghash, ng); ALLOC(d_freq, ng); ALLOC(d_gkeep, ng); ALLOC(d_gkeep_incl, ng);
  hipLaunchKernelGGL(group_starts, grid_for(n), dim3(256), 0, st, d_head, d_gid, d_gstart, d_ghash, d_k2, n);
  hipLaunchKernelGGL(group_freq, grid_for(ng), dim3(256), 0, st, d_gstart, d_freq, ng, n);
  // ---- frequency threshold, exactly as winSketch.hpp:298-349 (host, from the histogram) ----
  std::vector<uint32_t> freq((size_t)ng);
  HIPCHK(h, hipMemcpyAsync(freq.data(), d_freq, (size_t)ng * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (dbg) fprintf(stderr, ", sort+groups+freq download (%lld groups) %.1f ms", (long long)ng, now() - t_b);
  const double t_c = now();
  const uint64_t min_occ = 10;
  uint64_t thr;
  if (max_kmer_freq <= 1.0) thr = std::max(min_occ, (uint64_t)((double)n * max_kmer_freq));
  else thr = std::max(min_occ, (uint64_t)max_kmer_freq);
  size_t would_pos = 0, would_unique = 0;
  for (uint32_t f : freq) if ((uint64_t)f > thr && f > min_occ) { ++would_unique; would_pos += f; }
  if (would_pos > (size_t)n / 2 || (double)would_unique > (double)freq.size() * 0.7) {
    std::vector<uint32_t> all(freq);
    std::sort(all.begin(), all.end());
    size_t keep_index = (size_t)((double)all.size() * 0.999);
    if (keep_index >= all.size()) keep_index = all.size() - 1;
    thr = std::max<uint64_t>(thr, all[keep_index]);
    ix->adjusted = 1;
  }
  ix->threshold = thr;
  if (dbg) fprintf(stderr, ", threshold on host %.1f ms", now() - t_c);
  const double t_d = now();
  HIPCHK(h, hipMemsetAsync(d_gkeep, 0, (size_t)ng * sizeof(uint32_t), st));
  hipLaunchKernelGGL(mark_keep_and_chain, grid_for(n), dim3(256), 0, st, d_m, d_ord2, d_head, d_gid, d_freq, thr, d_keep, d_chead, d_gkeep, n);
  rc = inclusive_scan<uint32_t>(h, sc, d_chead, d_chain, n, st);
  if (rc == WFM_OK) rc = inclusive_scan<uint32_t>(h, sc, d_gkeep, d_gkeep_incl, ng, st);
  hipLaunchKernelGGL(mark_keep_input_order, grid_for(n), dim3(256), 0, st, d_ord2, d_keep, d_keep_in, n);
  if (rc == WFM_OK) rc = inclusive_scan<uint32_t>(h, sc, d_keep_in, d_keep_in_incl, n, st);
  if (rc != WFM_OK) { delete ix; return rc; }
  uint32_t n_chain = 0, n_uniq = 0, n_kept = 0;
  HIPCHK(h, hipMemcpyAsync(&n_chain, d_chain + (n - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&n_uniq, d_gkeep_incl + (ng - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&n_kept, d_keep_in_incl + (n - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  ix->n_points = 2 * (int64_t)n_chain; ix->n_unique = n_uniq; ix->n_kept = n_kept; ix->filtered = n - (int64_t)n_kept;
  if (wfm_dmalloc((void**)&ix->d_uhash, std::max<size_t>(n_uniq, 1) * 8) != hipSuccess ||
      wfm_dmalloc((void**)&ix->d_poff, ((size_t)n_uniq + 1) * 8) != hipSuccess ||
      wfm_dmalloc((void**)&ix->d_points, std::max<size_t>((size_t)ix->n_points, 1) * sizeof(wfm_interval_point_t)) != hipSuccess ||
      wfm_dmalloc((void**)&ix->d_minmers, std::max<size_t>(n_kept, 1) * sizeof(wfm_minmer_t)) != hipSuccess) {
    wfm_index_free(h, ix); wfm_set_error(h, "out of device memory (index)"); return WFM_E_NOMEM;
  }
  hipLaunchKernelGGL(emit_points, grid_for(n), dim3(256), 0, st, d_m, d_ord2, d_keep, d_chead, d_chain, ix->d_points, n);
  hipLaunchKernelGGL(emit_group_offsets, grid_for(ng), dim3(256), 0, st, d_gstart, d_ghash, d_gkeep, d_gkeep_incl, d_chain, d_chead, ix->d_uhash, ix->d_poff, ng);
  HIPCHK(h, hipMemcpyAsync(ix->d_poff + n_uniq, &ix->n_points, 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(compact_minmers, grid_for(n), dim3(256), 0, st, d_m, d_keep_in, d_keep_in_incl, ix->d_minmers, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(st));
#undef ALLOC
  if (dbg) fprintf(stderr, ", emit %.1f ms\n", now() - t_d);
  *out = ix;
  return WFM_OK;
}

extern "C" {

int wfm_index_upload(wfm_handle_t* h, const uint64_t* uhash, const int64_t* poff, int64_t n_unique, const wfm_interval_point_t* points,
                     const wfm_minmer_t* minmers, int64_t n_kept, wfm_index_t** out) {
  if (!h || !out || n_unique < 0 || n_kept < 0 || (n_unique && (!uhash || !poff || !points)) || (n_kept && !minmers)) return WFM_E_ARG;
  *out = nullptr;
  for (int64_t u = 1; u < n_unique; ++u)
    if (uhash[u] <= uhash[u - 1]) { wfm_set_error(h, "wfm_index_upload: hashes must ascend strictly"); return WFM_E_ARG; }
  const int64_t n_points = n_unique ? poff[n_unique] : 0;
  if (n_unique && (poff[0] != 0 || n_points < 0)) { wfm_set_error(h, "wfm_index_upload: bad offsets"); return WFM_E_ARG; }
  HIPCHK(h, hipSetDevice(wfm_device(h)));
  hipStream_t st = wfm_stream(h);
  wfm_index* ix = new wfm_index();
  ix->device = wfm_device(h);
  ix->n_windows = n_kept; ix->n_kept = n_kept; ix->n_unique = n_unique; ix->n_points = n_points;
  const int64_t zero = 0;
  if (wfm_dmalloc((void**)&ix->d_uhash, std::max<size_t>((size_t)n_unique, 1) * 8) != hipSuccess ||
      wfm_dmalloc((void**)&ix->d_poff, ((size_t)n_unique + 1) * 8) != hipSuccess ||
      wfm_dmalloc((void**)&ix->d_points, std::max<size_t>((size_t)n_points, 1) * sizeof(wfm_interval_point_t)) != hipSuccess ||
      wfm_dmalloc((void**)&ix->d_minmers, std::max<size_t>((size_t)n_kept, 1) * sizeof(wfm_minmer_t)) != hipSuccess) {
    wfm_index_free(h, ix); wfm_set_error(h, "out of device memory (index)"); return WFM_E_NOMEM;
  }
  hipError_t e = hipSuccess;
  if (n_unique) e = hipMemcpyAsync(ix->d_uhash, uhash, (size_t)n_unique * 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = n_unique ? hipMemcpyAsync(ix->d_poff, poff, ((size_t)n_unique + 1) * 8, hipMemcpyHostToDevice, st)
                                    : hipMemcpyAsync(ix->d_poff, &zero, 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess && n_points) e = hipMemcpyAsync(ix->d_points, points, (size_t)n_points * sizeof(wfm_interval_point_t), hipMemcpyHostToDevice, st);
  if (e == hipSuccess && n_kept) e = hipMemcpyAsync(ix->d_minmers, minmers, (size_t)n_kept * sizeof(wfm_minmer_t), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { wfm_index_free(h, ix); wfm_set_error(h, std::string("wfm_index_upload: ") + hipGetErrorString(e)); return WFM_E_HIP; }
  *out = ix;
  return WFM_OK;
}

// The index on another GPU of the node: the four arrays go device to device (xGMI when the devices are peers,
// through the host otherwise -- hipMemcpyPeer picks the route); the copy belongs to dst.
int wfm_index_replicate(wfm_handle_t* src, const wfm_index_t* ix, wfm_handle_t* dst, wfm_index_t** out) {
  if (!src || !ix || !dst || !out) return WFM_E_ARG;
  *out = nullptr;
  const int sdev = ix->device, ddev = wfm_device(dst);
  HIPCHK(dst, hipSetDevice(ddev));
  if (sdev != ddev) {
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, ddev, sdev) == hipSuccess && can) {
      const hipError_t pe = hipDeviceEnablePeerAccess(sdev, 0);
      if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
      else if (pe == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
    }
  }
  wfm_index* cp = new wfm_index();
  cp->device = ddev;
  cp->n_windows = ix->n_windows; cp->n_kept = ix->n_kept; cp->n_unique = ix->n_unique; cp->n_points = ix->n_points;
  cp->threshold = ix->threshold; cp->filtered = ix->filtered; cp->adjusted = ix->adjusted;
  const size_t b_uhash = std::max<size_t>((size_t)ix->n_unique, 1) * 8, b_poff = ((size_t)ix->n_unique + 1) * 8,
               b_points = std::max<size_t>((size_t)ix->n_points, 1) * sizeof(wfm_interval_point_t),
               b_minmers = std::max<size_t>((size_t)ix->n_kept, 1) * sizeof(wfm_minmer_t);
  if (wfm_dmalloc((void**)&cp->d_uhash, b_uhash) != hipSuccess || wfm_dmalloc((void**)&cp->d_poff, b_poff) != hipSuccess ||
      wfm_dmalloc((void**)&cp->d_points, b_points) != hipSuccess || wfm_dmalloc((void**)&cp->d_minmers, b_minmers) != hipSuccess) {
    wfm_index_free(dst, cp); wfm_set_error(dst, "out of device memory (index copy)"); return WFM_E_NOMEM;
  }
  hipStream_t st = wfm_stream(dst);
  hipError_t e = hipSuccess;
  if (ix->n_unique) e = hipMemcpyPeerAsync(cp->d_uhash, ddev, ix->d_uhash, sdev, (size_t)ix->n_unique * 8, st);
  if (e == hipSuccess) e = hipMemcpyPeerAsync(cp->d_poff, ddev, ix->d_poff, sdev, b_poff, st);
  if (e == hipSuccess && ix->n_points) e = hipMemcpyPeerAsync(cp->d_points, ddev, ix->d_points, sdev, (size_t)ix->n_points * sizeof(wfm_interval_point_t), st);
  if (e == hipSuccess && ix->n_kept) e = hipMemcpyPeerAsync(cp->d_minmers, ddev, ix->d_minmers, sdev, (size_t)ix->n_kept * sizeof(wfm_minmer_t), st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { wfm_index_free(dst, cp); wfm_set_error(dst, std::string("wfm_index_replicate: ") + hipGetErrorString(e)); return WFM_E_HIP; }
  *out = cp;
  return WFM_OK;
}

void wfm_index_free(wfm_handle_t* h, wfm_index_t* ix) {
  if (!ix) return;
  if (h) (void)hipSetDevice(wfm_device(h));
  if (ix->d_uhash) (void)wfm_dfree(ix->d_uhash);
  if (ix->d_poff) (void)wfm_dfree(ix->d_poff);
  if (ix->d_points) (void)wfm_dfree(ix->d_points);
  if (ix->d_minmers) (void)wfm_dfree(ix->d_minmers);
  delete ix;
}

int wfm_index_info(const wfm_index_t* ix, wfm_index_info_t* out) {
  if (!ix || !out) return WFM_E_ARG;
  out->n_windows = ix->n_windows; out->n_kept = ix->n_kept; out->n_unique = ix->n_unique; out->n_points = ix->n_points;
  out->threshold = ix->threshold; out->filtered = ix->filtered; out->adjusted = ix->adjusted;
  return WFM_OK;
}

int wfm_index_download(wfm_handle_t* h, const wfm_index_t* ix, uint64_t* uhash, int64_t* poff, wfm_interval_point_t* points, wfm_minmer_t* minmers) {
  if (!h || !ix) return WFM_E_ARG;
  HIPCHK(h, hipSetDevice(ix->device));
  if (uhash && ix->n_unique) HIPCHK(h, hipMemcpy(uhash, ix->d_uhash, (size_t)ix->n_unique * 8, hipMemcpyDeviceToHost));
  if (poff && ix->d_poff) HIPCHK(h, hipMemcpy(poff, ix->d_poff, ((size_t)ix->n_unique + 1) * 8, hipMemcpyDeviceToHost));
  if (points && ix->n_points) HIPCHK(h, hipMemcpy(points, ix->d_points, (size_t)ix->n_points * sizeof(wfm_interval_point_t), hipMemcpyDeviceToHost));
  if (minmers && ix->n_kept) HIPCHK(h, hipMemcpy(minmers, ix->d_minmers, (size_t)ix->n_kept * sizeof(wfm_minmer_t), hipMemcpyDeviceToHost));
  return WFM_OK;
}

}  // extern "C"

// accessors for the mapping kernels (map_l1.hip)
const uint64_t* wfm_index_uhash(const wfm_index_t* ix) { return ix->d_uhash; }
const int64_t* wfm_index_poff(const wfm_index_t* ix) { return ix->d_poff; }
const wfm_interval_point_t* wfm_index_points(const wfm_index_t* ix) { return ix->d_points; }
const wfm_minmer_t* wfm_index_minmers(const wfm_index_t* ix) { return ix->d_minmers; }
int64_t wfm_index_n_unique(const wfm_index_t* ix) { return ix->n_unique; }
int64_t wfm_index_n_kept(const wfm_index_t* ix) { return ix->n_kept; }
