// map_prefilter.hip -- which k-mers of a hashed sequence the host winnowing has to see.
//
// addMinmers (commonFunc.hpp:440-708) keeps, per window of W = w-k+1 consecutive k-mers, the s smallest
// distinct canonical hashes.  A k-mer whose hash is above a threshold tau can only ever matter in a window
// that holds fewer than s distinct hashes <= tau.  So the stream the host replays is thinned on the device:
//
//   keep(i) = valid(i) and ( hash(i) <= tau  or  i lies in a window that may hold < s distinct hashes <= tau )
//
// "may hold": the distinct count of a window is bounded from below by its number of FRESH candidates -- k-mers
// with hash <= tau whose previous occurrence of the same hash is at least W positions back -- because two
// fresh candidates inside one window cannot share a hash.  Windows under the bound keep all their k-mers, so
// the host sees every such window exactly as the full stream would; in all other windows the sketch is full
// of hashes <= tau at every step, and a k-mer above tau is neither in it nor ever the smallest of the pool
// (minmers.cpp holds the argument next to the code that relies on it).  With tau set to let ~3 s of a window's
// W k-mers through, about one k-mer in eight survives in ordinary sequence (s = 39, W = 986); low-complexity
// and N-rich stretches stay whole.
//
// Per sequence: candidate flags -> scan/compact -> radix sort by (hash, position) -> fresh flags ->
// scan -> window counts -> scan (dilation by W) -> keep flags -> scan/compact (position, hash, strand).
// All of it is streaming integer work over n k-mers, bound by HBM bandwidth: ~60 B/k-mer in total.
#include <hip/hip_runtime.h>
#include "dev_cache.h"
#include <stdint.h>

#include <cstring>  // rocprim's texture iterator calls host memset
#include <mutex>
#include <string>

#include <rocprim/rocprim.hpp>

#include "map_device.h"

namespace {

#define HIPCHK(h, call)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      wfm_set_error((h), std::string(#call) + ": " + hipGetErrorString(e_));            \
      return e_ == hipErrorOutOfMemory ? WFM_E_NOMEM : WFM_E_HIP;                       \
    }                                                                                   \
  } while (0)

__global__ void pf_cand_kernel(const uint64_t* __restrict__ hash, const int8_t* __restrict__ strand, uint64_t tau, uint32_t* __restrict__ flag, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (strand[i] != 0 && hash[i] <= tau) ? 1u : 0u;
}

__global__ void pf_scatter_cand_kernel(const uint64_t* __restrict__ hash, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ incl,
                                       uint64_t* __restrict__ chash, uint32_t* __restrict__ cpos, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) { const uint32_t j = incl[i] - 1; chash[j] = hash[i]; cpos[j] = (uint32_t)i; }
}

// sorted by (hash, position): a candidate is fresh when the same hash did not occur in the W-1 positions before it
__global__ void pf_fresh_kernel(const uint64_t* __restrict__ key, const uint32_t* __restrict__ pos, int64_t m, uint32_t W, uint32_t* __restrict__ fresh) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  if (j == 0 || key[j] != key[j - 1] || pos[j] - pos[j - 1] >= W) fresh[pos[j]] = 1u;
}

// window a = k-mers [a, a+W): under the bound when it holds fewer than s fresh candidates
__global__ void pf_deficient_kernel(const uint32_t* __restrict__ F, int64_t n, int64_t W, uint32_t s, uint32_t* __restrict__ def) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  uint32_t d = 0;
  if (a + W <= n) d = (F[a + W - 1] - (a ? F[a - 1] : 0u)) < s ? 1u : 0u;
  def[a] = d;
}

__global__ void pf_keep_kernel(const uint64_t* __restrict__ hash, const int8_t* __restrict__ strand, uint64_t tau, const uint32_t* __restrict__ P,
                               int64_t n, int64_t W, uint32_t* __restrict__ keep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t kp = 0;
  if (strand[i] != 0) kp = (hash[i] <= tau || P[i] - (i >= W ? P[i - W] : 0u) > 0u) ? 1u : 0u;  // a window a in (i-W, i] is under the bound
  keep[i] = kp;
}

__global__ void pf_emit_kernel(const uint64_t* __restrict__ hash, const int8_t* __restrict__ strand, const uint32_t* __restrict__ keep,
                               const uint32_t* __restrict__ incl, uint32_t* __restrict__ pos, uint64_t* __restrict__ ohash, int8_t* __restrict__ ostrand, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keep[i]) { const uint32_t j = incl[i] - 1; pos[j] = (uint32_t)i; ohash[j] = hash[i]; ostrand[j] = strand[i]; }
}

// out[q] = number of kept positions < query[q]
__global__ void pf_lower_bound_kernel(const uint32_t* __restrict__ pos, int64_t m, const int64_t* __restrict__ query, int nq, int64_t* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const int64_t x = query[q];
  int64_t lo = 0, hi = m;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)pos[mid] < x) lo = mid + 1; else hi = mid;
  }
  out[q] = lo;
}

// a buffer of the caller's grow-only workspace, or (without one) an allocation that lives until the call returns
int need(wfm_handle_t* h, MapScratch& sc, MapThinWork::Buf* slot, size_t bytes, void** out) {
  if (!slot) {
    char* p = nullptr;
    HIPCHK(h, sc.alloc(&p, bytes));
    *out = p;
    return WFM_OK;
  }
  if (slot->bytes < bytes) {
    if (slot->p) (void)wfm_dfree(slot->p);
    slot->p = nullptr; slot->bytes = 0;
    const size_t want = bytes + bytes / 8;
    HIPCHK(h, wfm_dmalloc(&slot->p, want));
    slot->bytes = want;
  }
  *out = slot->p;
  return WFM_OK;
}

int scan_u32(wfm_handle_t* h, MapScratch& sc, MapThinWork* wk, const uint32_t* in, uint32_t* out, int64_t n, hipStream_t st) {
  size_t tmp = 0;
  HIPCHK(h, rocprim::inclusive_scan(nullptr, tmp, in, out, (size_t)n, rocprim::plus<uint32_t>(), st));
  void* d_tmp = nullptr;
  const int rc = need(h, sc, wk ? &wk->tmp : nullptr, tmp, &d_tmp);
  if (rc != WFM_OK) return rc;
  HIPCHK(h, rocprim::inclusive_scan(d_tmp, tmp, in, out, (size_t)n, rocprim::plus<uint32_t>(), st));
  return WFM_OK;
}

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

void map_thin_work_free(MapThinWork* wk) {
  (void)hipSetDevice(wk->device);
  for (MapThinWork::Buf* b : {&wk->a, &wk->b, &wk->ck, &wk->ck2, &wk->cp, &wk->cp2, &wk->tmp}) {
    if (b->p) (void)wfm_dfree(b->p);
    b->p = nullptr; b->bytes = 0;
  }
}

namespace {
struct DevPool {
  struct Blk { void* p; size_t bytes; int device; bool busy; };
  std::mutex mu;
  std::vector<Blk> blks;
};
DevPool& dev_pool() { static DevPool p; return p; }
}  // namespace

void* map_dev_pool_get(int device, size_t bytes) {
  if (bytes == 0) bytes = 16;
  DevPool& P = dev_pool();
  {
    std::lock_guard<std::mutex> lk(P.mu);
    DevPool::Blk* best = nullptr;
    for (auto& b : P.blks)  // the smallest free block that is large enough, and not more than four times too large
      if (!b.busy && b.device == device && b.bytes >= bytes && b.bytes <= 4 * bytes + 65536 && (!best || b.bytes < best->bytes)) best = &b;
    if (best) { best->busy = true; return best->p; }
  }
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  void* p = nullptr;
  const size_t want = bytes + bytes / 8 + 256;
  if (wfm_dmalloc(&p, want) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(P.mu);
  P.blks.push_back(DevPool::Blk{p, want, device, true});
  return p;
}
void map_dev_pool_put(int device, void* p) {
  if (!p) return;
  DevPool& P = dev_pool();
  {
    std::lock_guard<std::mutex> lk(P.mu);
    for (auto& b : P.blks)
      if (b.p == p) { b.busy = false; return; }
  }
  (void)hipSetDevice(device);
  (void)wfm_dfree(p);  // not one of the pool's
}
void map_dev_pool_trim() {
  DevPool& P = dev_pool();
  std::lock_guard<std::mutex> lk(P.mu);
  size_t o = 0;
  for (size_t i = 0; i < P.blks.size(); ++i) {
    if (P.blks[i].busy) { P.blks[o++] = P.blks[i]; continue; }
    (void)hipSetDevice(P.blks[i].device);
    (void)wfm_dfree(P.blks[i].p);
  }
  P.blks.resize(o);
}

void map_sparse_free(MapSparseSeq* s) {
  map_dev_pool_put(s->device, s->d_pos);
  map_dev_pool_put(s->device, s->d_hash);
  map_dev_pool_put(s->device, s->d_strand);
  s->d_pos = nullptr; s->d_hash = nullptr; s->d_strand = nullptr; s->m = 0;
}

int map_prefilter_device(wfm_handle_t* h, const MapHashedSeq* q, int64_t W, int s, uint64_t tau, MapSparseSeq* out, MapThinWork* wk) {
  out->d_pos = nullptr; out->d_hash = nullptr; out->d_strand = nullptr; out->m = 0; out->device = q->device;
  const int64_t n = q->nk;
  if (n <= 0) return WFM_OK;
  if (n >= ((int64_t)1 << 32) - 1 || W < 1 || s < 1) { wfm_set_error(h, "prefilter: sequence too long for 32-bit k-mer positions"); return WFM_E_UNSUPPORTED; }
  HIPCHK(h, hipSetDevice(q->device));
  hipStream_t st = wfm_stream(h);
  MapScratch sc;
  if (wk) wk->device = q->device;
  uint32_t *A = nullptr, *B = nullptr;
  int rc = need(h, sc, wk ? &wk->a : nullptr, (size_t)n * 4, (void**)&A);
  if (rc == WFM_OK) rc = need(h, sc, wk ? &wk->b : nullptr, (size_t)n * 4, (void**)&B);
  if (rc != WFM_OK) return rc;
  // candidates: valid and hash <= tau
  hipLaunchKernelGGL(pf_cand_kernel, grid_for(n), dim3(256), 0, st, q->d_hash, q->d_strand, tau, A, n);
  rc = scan_u32(h, sc, wk, A, B, n, st);
  if (rc != WFM_OK) return rc;
  uint32_t mc32 = 0;
  HIPCHK(h, hipMemcpyAsync(&mc32, B + (n - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  const int64_t mc = mc32;
  uint64_t *ck = nullptr, *ck2 = nullptr;
  uint32_t *cp = nullptr, *cp2 = nullptr;
  if (mc > 0) {
    rc = need(h, sc, wk ? &wk->ck : nullptr, (size_t)mc * 8, (void**)&ck);
    if (rc == WFM_OK) rc = need(h, sc, wk ? &wk->ck2 : nullptr, (size_t)mc * 8, (void**)&ck2);
    if (rc == WFM_OK) rc = need(h, sc, wk ? &wk->cp : nullptr, (size_t)mc * 4, (void**)&cp);
    if (rc == WFM_OK) rc = need(h, sc, wk ? &wk->cp2 : nullptr, (size_t)mc * 4, (void**)&cp2);
    if (rc != WFM_OK) return rc;
    hipLaunchKernelGGL(pf_scatter_cand_kernel, grid_for(n), dim3(256), 0, st, q->d_hash, A, B, ck, cp, n);
    size_t tmp = 0;
    HIPCHK(h, rocprim::radix_sort_pairs(nullptr, tmp, ck, ck2, cp, cp2, (size_t)mc, 0, 64, st));
    void* d_tmp = nullptr;
    rc = need(h, sc, wk ? &wk->tmp : nullptr, tmp, &d_tmp);
    if (rc != WFM_OK) return rc;
    HIPCHK(h, rocprim::radix_sort_pairs(d_tmp, tmp, ck, ck2, cp, cp2, (size_t)mc, 0, 64, st));  // stable: positions ascend within a hash
  }
  // fresh candidates -> per-window lower bound of the distinct count
  HIPCHK(h, hipMemsetAsync(A, 0, (size_t)n * 4, st));
  if (mc > 0) hipLaunchKernelGGL(pf_fresh_kernel, grid_for(mc), dim3(256), 0, st, ck2, cp2, mc, (uint32_t)std::min<int64_t>(W, 0xffffffffll), A);
  rc = scan_u32(h, sc, wk, A, B, n, st);
  if (rc != WFM_OK) return rc;
  hipLaunchKernelGGL(pf_deficient_kernel, grid_for(n), dim3(256), 0, st, B, n, W, (uint32_t)s, A);
  rc = scan_u32(h, sc, wk, A, B, n, st);
  if (rc != WFM_OK) return rc;
  hipLaunchKernelGGL(pf_keep_kernel, grid_for(n), dim3(256), 0, st, q->d_hash, q->d_strand, tau, B, n, W, A);
  rc = scan_u32(h, sc, wk, A, B, n, st);
  if (rc != WFM_OK) return rc;
  uint32_t m32 = 0;
  HIPCHK(h, hipMemcpyAsync(&m32, B + (n - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  out->m = m32;
  if (out->m > 0) {
    out->d_pos = (uint32_t*)map_dev_pool_get(out->device, (size_t)out->m * 4);
    out->d_hash = (uint64_t*)map_dev_pool_get(out->device, (size_t)out->m * 8);
    out->d_strand = (int8_t*)map_dev_pool_get(out->device, (size_t)out->m);
    if (!out->d_pos || !out->d_hash || !out->d_strand) { map_sparse_free(out); wfm_set_error(h, "out of device memory (kept k-mers)"); return WFM_E_NOMEM; }
    hipLaunchKernelGGL(pf_emit_kernel, grid_for(n), dim3(256), 0, st, q->d_hash, q->d_strand, A, B, out->d_pos, out->d_hash, out->d_strand, n);
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(st));  // the scratch arrays go out of scope
  return WFM_OK;
}

int map_sparse_lower_bound(wfm_handle_t* h, const MapSparseSeq* s, const int64_t* query, int nq, int64_t* out, hipStream_t stream) {
  if (nq <= 0) return WFM_OK;
  if (s->m == 0) { for (int i = 0; i < nq; ++i) out[i] = 0; return WFM_OK; }
  HIPCHK(h, hipSetDevice(s->device));
  hipStream_t st = stream ? stream : wfm_stream(h);
  struct Pooled {
    int device; void* p;
    ~Pooled() { map_dev_pool_put(device, p); }
  } blk{s->device, map_dev_pool_get(s->device, (size_t)nq * 16)};
  if (!blk.p) { wfm_set_error(h, "out of device memory (lower bounds)"); return WFM_E_NOMEM; }
  int64_t *d_q = (int64_t*)blk.p, *d_o = d_q + nq;
  HIPCHK(h, hipMemcpyAsync(d_q, query, (size_t)nq * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(pf_lower_bound_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, s->d_pos, s->m, d_q, nq, d_o);
  HIPCHK(h, hipMemcpyAsync(out, d_o, (size_t)nq * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  return WFM_OK;
}

// kept k-mers [c0, c1) into a ring slot: uint64 hash[mc] | uint32 pos[mc] | int8 strand[mc]
int map_stage_copy_sparse(MapStage* st, int slot, const MapSparseSeq* s, int64_t c0, int64_t c1) {
  const size_t mc = (size_t)std::max<int64_t>(0, c1 - c0);
  if (slot < 0 || slot >= st->nslots || mc * 13 > st->slot_bytes) return WFM_E_ARG;
  if (hipSetDevice(st->device) != hipSuccess) return WFM_E_HIP;
  char* dst = st->slot(slot);
  if (mc) {
    if (hipMemcpyAsync(dst, s->d_hash + c0, mc * 8, hipMemcpyDeviceToHost, st->stream) != hipSuccess) return WFM_E_HIP;
    if (hipMemcpyAsync(dst + mc * 8, s->d_pos + c0, mc * 4, hipMemcpyDeviceToHost, st->stream) != hipSuccess) return WFM_E_HIP;
    if (hipMemcpyAsync(dst + mc * 12, s->d_strand + c0, mc, hipMemcpyDeviceToHost, st->stream) != hipSuccess) return WFM_E_HIP;
  }
  return hipEventRecord(st->ev[(size_t)slot], st->stream) == hipSuccess ? WFM_OK : WFM_E_HIP;
}

int map_sparse_fetch_packed(const MapSparseSeq* s, int64_t c0, int64_t c1, char* dst) {
  if (hipSetDevice(s->device) != hipSuccess) return WFM_E_HIP;
  const size_t mc = (size_t)std::max<int64_t>(0, c1 - c0);
  if (mc) {
    if (hipMemcpy(dst, s->d_hash + c0, mc * 8, hipMemcpyDeviceToHost) != hipSuccess) return WFM_E_HIP;
    if (hipMemcpy(dst + mc * 8, s->d_pos + c0, mc * 4, hipMemcpyDeviceToHost) != hipSuccess) return WFM_E_HIP;
    if (hipMemcpy(dst + mc * 12, s->d_strand + c0, mc, hipMemcpyDeviceToHost) != hipSuccess) return WFM_E_HIP;
  }
  return WFM_OK;
}

extern "C" int64_t wfm_prefilter_kmers(wfm_handle_t* h, const char* seq, int64_t len, int k, int w, int s, double c_factor,
                                       uint32_t* pos, uint64_t* hash, int8_t* strand, int64_t cap) {
  if (!h || !seq || len < 0 || cap < 0 || (cap && (!pos || !hash || !strand))) return WFM_E_ARG;
  if (k < 1 || k > 32 || w < k || s < 1 || !(c_factor > 0)) { wfm_set_error(h, "need 1 <= k <= 32, w >= k, s >= 1, c_factor > 0"); return WFM_E_UNSUPPORTED; }
  if (len < k) return 0;
  const int64_t W = (int64_t)w - k + 1;
  const uint64_t tau = map_prefilter_tau(c_factor, s, W);
  MapHashedSeq q;
  int rc = map_hash_sequence_device(h, seq, len, k, &q);
  if (rc != WFM_OK) return rc;
  MapSparseSeq sp;
  rc = map_prefilter_device(h, &q, W, s, tau, &sp);
  map_hashed_free(&q);
  if (rc != WFM_OK) return rc;
  const int64_t m = sp.m, take = std::min(m, cap);
  hipError_t e = hipSuccess;
  if (take > 0) {
    e = hipMemcpy(pos, sp.d_pos, (size_t)take * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(hash, sp.d_hash, (size_t)take * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(strand, sp.d_strand, (size_t)take, hipMemcpyDeviceToHost);
  }
  map_sparse_free(&sp);
  if (e != hipSuccess) { wfm_set_error(h, std::string("hipMemcpy: ") + hipGetErrorString(e)); return WFM_E_HIP; }
  return m;
}

