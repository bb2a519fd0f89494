// map_finish.hip -- the closing steps of addMinmers on the device (commonFunc.hpp:660-706): records of more than w
// windows cut into pieces, strand signs, the order by (wpos, wpos_end), de-duplication.
//
// The order is the catch.  The reference sorts with std::sort and a comparison that looks at (wpos, wpos_end) only, so the
// order of records that tie is whatever libstdc++'s introsort leaves -- and it is part of the output (the host path runs the
// library's own routines for that reason, host/minmers.cpp sort_as_std).  Here the same arrangement is COMPUTED in
// data-parallel steps; libstdc++'s std::sort (bits/stl_algo.h, unchanged since GCC 4.9) is
//   introsort loop: while a range has more than 16 elements: depth budget spent -> heapsort it; else the median of
//     (first+1, middle, last-1) goes to `first` as the pivot, the unguarded Hoare partition runs over (first, last), the
//     right part is recursed into and the left part looped on;   then one insertion sort over everything.
// * The insertion sort is stable and sorts everything, so the final order is the STABLE sort of the arrangement the loop
//   leaves: ties stay as the partitions left them.
// * One partition, in parallel.  Let A = positions of (first, last) whose element is not less than the pivot, ascending,
//   and B = positions whose element is not greater, descending.  The two pointers of the Hoare loop stop exactly at A[0],
//   B[0], swap, go on to A[1], B[1], ... for as long as A[t] < B[t]: elements the pointers have not passed are untouched,
//   so the lists of the ORIGINAL arrangement stay valid.  With T = the number of such t, the swaps are the pairs
//   (A[t], B[t]), t < T, and the returned cut is min(A[T], B[T-1]) (a missing entry counts as infinity): the left
//   pointer's next stop is the next original element >= pivot or, if it comes first, the one swapped into B[T-1].
// * Ranges are independent, so all ranges of one recursion depth are partitioned by one launch, one workgroup each
//   (sortlike_level_kernel); children of more than 16 elements go to the next launch.  A range that spends its depth
//   budget (2 floor(log2 n) levels; not seen outside adversarial inputs) is heap-sorted by the library itself on the host.
// scripts/ has no part in this; the CPU suite holds a host restatement of these steps (map_sortlike_model) against
// std::sort, the GPU suite holds the kernels against it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

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

struct Seg { int64_t f, l; int32_t depth, pad_; };
constexpr int64_t SMALL_SEG = 1024;  // ranges up to this size are partitioned by one wave, larger ones by 16
constexpr uint32_t NONE = 0xFFFFFFFFu;

__device__ inline void swap_elems(uint64_t* key, uint32_t* idx, int64_t a, int64_t b) {
  const uint64_t k = key[a]; key[a] = key[b]; key[b] = k;
  const uint32_t i = idx[a]; idx[a] = idx[b]; idx[b] = i;
}

// One range per workgroup of NW waves: pivot, the lists A and B (positions relative to the array), the swaps, the children.
template <int NW>
__global__ void __launch_bounds__(NW * 64) sortlike_level_kernel(const Seg* cur, uint64_t* key, uint32_t* idx, uint32_t* A, uint32_t* B, Seg* next_big, Seg* next_small,
                                                                  Seg* heap, int* counts) {
  const Seg sg = cur[blockIdx.x];
  const int64_t f = sg.f, l = sg.l;
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  __shared__ uint64_t s_p;
  __shared__ int s_ge[NW], s_le[NW], s_T[NW];
  if (sg.depth == 0) {  // budget spent: the library's heapsort on the host
    if (tid == 0) heap[atomicAdd(&counts[2], 1)] = sg;
    return;
  }
  if (tid == 0) {  // __move_median_to_first(first, first + 1, mid, last - 1)
    const int64_t a = f + 1, b = f + (l - f) / 2, c = l - 1;
    const uint64_t ka = key[a], kb = key[b], kc = key[c];
    int64_t pick;
    if (ka < kb) { if (kb < kc) pick = b; else if (ka < kc) pick = c; else pick = a; }
    else if (ka < kc) pick = a;
    else if (kb < kc) pick = c;
    else pick = b;
    swap_elems(key, idx, f, pick);
    s_p = key[f];
  }
  __syncthreads();
  const uint64_t p = s_p;
  // every wave takes a contiguous share of (f, l), whole tiles of 64
  const int64_t lo = f + 1, m = l - lo;
  const int64_t share = ((m + NW - 1) / NW + 63) / 64 * 64;
  const int64_t w0 = lo + (int64_t)wv * share, w1 = w0 + share < l ? w0 + share : l;
  int cge = 0, cle = 0;
  for (int64_t base = w0; base < w1; base += 64) {
    const int64_t i = base + lane;
    uint64_t k = 0;
    const bool in = i < w1;
    if (in) k = key[i];
    cge += __popcll(__ballot(in && k >= p));
    cle += __popcll(__ballot(in && k <= p));
  }
  if (lane == 0) { s_ge[wv] = cge; s_le[wv] = cle; }
  __syncthreads();
  int oge = 0, ole = 0, nA = 0, nB = 0;
  for (int q = 0; q < NW; ++q) { if (q < wv) { oge += s_ge[q]; ole += s_le[q]; } nA += s_ge[q]; nB += s_le[q]; }
  for (int64_t base = w0; base < w1; base += 64) {
    const int64_t i = base + lane;
    uint64_t k = 0;
    const bool in = i < w1;
    if (in) k = key[i];
    const bool ge = in && k >= p, le = in && k <= p;
    const unsigned long long mg = __ballot(ge), ml = __ballot(le);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (ge) A[lo + oge + __popcll(mg & below)] = (uint32_t)(i - lo);
    if (le) B[lo + (nB - 1 - (ole + __popcll(ml & below)))] = (uint32_t)(i - lo);  // descending positions
    oge += __popcll(mg); ole += __popcll(ml);
  }
  __syncthreads();
  // T = number of t with A[t] < B[t] (the pairs that are swapped)
  const int nmin = nA < nB ? nA : nB;
  int cnt = 0;
  for (int t = tid; t < nmin; t += NW * 64) cnt += A[lo + t] < B[lo + t] ? 1 : 0;
  for (int o = 32; o; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) s_T[wv] = cnt;
  __syncthreads();
  int T = 0;
  for (int q = 0; q < NW; ++q) T += s_T[q];
  const int64_t INF = INT64_MAX;
  const int64_t ca = T < nA ? lo + (int64_t)A[lo + T] : INF, cb = T > 0 ? lo + (int64_t)B[lo + T - 1] : INF;
  const int64_t cut = ca < cb ? ca : cb;
  for (int t = tid; t < T; t += NW * 64) swap_elems(key, idx, lo + (int64_t)A[lo + t], lo + (int64_t)B[lo + t]);
  if (tid == 0) {
    const Seg kids[2] = {Seg{f, cut, sg.depth - 1, 0}, Seg{cut, l, sg.depth - 1, 0}};
    for (const Seg& c : kids) {
      const int64_t sz = c.l - c.f;
      if (sz <= 16) continue;
      if (sz > SMALL_SEG) next_big[atomicAdd(&counts[0], 1)] = c;
      else next_small[atomicAdd(&counts[1], 1)] = c;
    }
  }
}

// ---- cut / layout / de-duplication ----
__device__ inline bool dropped(const wfm_minmer_t& m) { return m.wpos < 0 || m.wpos_end < 0 || m.wpos == m.wpos_end; }
__device__ inline int pieces_of(const wfm_minmer_t& m, int w) { return (int)ceilf((float)(m.wpos_end - m.wpos) / (float)w); }

// how many records of at most w windows (0 / 1) and how many pieces each raw record turns into
__global__ void finish_count_kernel(const wfm_minmer_t* raw, int64_t n, int w, uint32_t* n_short, uint32_t* n_piece) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const wfm_minmer_t m = raw[i];
  uint32_t a = 0, b = 0;
  if (!dropped(m)) { if (m.wpos_end > m.wpos + w) b = (uint32_t)pieces_of(m, w); else a = 1; }
  n_short[i] = a; n_piece[i] = b;
}
// the array std::sort would see: the records of at most w windows in emission order, then the pieces of the longer ones
__global__ void finish_layout_kernel(const wfm_minmer_t* raw, int64_t n, int w, const uint32_t* off_short, const uint32_t* off_piece, uint32_t total_short,
                                     wfm_minmer_t* R, uint64_t* key, uint32_t* idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  wfm_minmer_t m = raw[i];
  if (dropped(m)) return;
  const int16_t st = m.strand < 0 ? (int16_t)-1 : (int16_t)1;  // every non-negative tally reads FWD (commonFunc.hpp:672)
  if (m.wpos_end > m.wpos + w) {
    const int np = pieces_of(m, w);
    uint32_t at = total_short + off_piece[i];
    for (int c = 0; c < np; ++c, ++at) {
      wfm_minmer_t q;
      q.hash = m.hash; q.wpos = m.wpos + (int64_t)c * w;
      const int64_t e = m.wpos + (int64_t)c * w + w;
      q.wpos_end = e < m.wpos_end ? e : m.wpos_end;
      q.seqId = m.seqId; q.strand = st; q.pad_ = 0;
      R[at] = q;
      key[at] = ((uint64_t)q.wpos << 32) | (uint64_t)(uint32_t)q.wpos_end;
      idx[at] = at;
    }
  } else {
    const uint32_t at = off_short[i];
    m.strand = st; m.pad_ = 0;
    R[at] = m;
    key[at] = ((uint64_t)m.wpos << 32) | (uint64_t)(uint32_t)m.wpos_end;
    idx[at] = at;
  }
}
// std::unique over (wpos, hash): a record goes when its predecessor in the sorted order has the same two
__global__ void finish_flag_kernel(const wfm_minmer_t* R, const uint32_t* order, int64_t n, uint32_t* keep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k = 1;
  if (i > 0) {
    const wfm_minmer_t a = R[order[i - 1]], b = R[order[i]];
    if (a.wpos == b.wpos && a.hash == b.hash) k = 0;
  }
  keep[i] = k;
}
__global__ void finish_emit_kernel(const wfm_minmer_t* R, const uint32_t* order, const uint32_t* keep, const uint32_t* off, int64_t n, wfm_minmer_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !keep[i]) return;
  out[off[i]] = R[order[i]];
}

int grow(MapFinishWork::Buf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return WFM_OK;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr; b.bytes = 0;
  const size_t want = bytes + bytes / 4 + 256;
  if (hipMalloc(&b.p, want) != hipSuccess) return WFM_E_NOMEM;
  b.bytes = want;
  return WFM_OK;
}

template <typename T>
int excl_scan(wfm_handle_t* h, MapFinishWork* wk, const T* in, T* out, size_t n, hipStream_t st) {
  size_t tmp = 0;
  HIPCHK(h, rocprim::exclusive_scan(nullptr, tmp, in, out, (T)0, n, rocprim::plus<T>(), st));
  if (grow(wk->tmp, tmp)) return WFM_E_NOMEM;
  HIPCHK(h, rocprim::exclusive_scan(wk->tmp.p, tmp, in, out, (T)0, n, rocprim::plus<T>(), st));
  return WFM_OK;
}

// the arrangement std::sort's introsort loop leaves, on (key, idx) pairs of the device
int sortlike_loop_device(wfm_handle_t* h, MapFinishWork* wk, uint64_t* key, uint32_t* idx, int64_t n, hipStream_t st, int* levels_out, int* heaps_out) {
  if (levels_out) *levels_out = 0;
  if (heaps_out) *heaps_out = 0;
  if (n <= 16) return WFM_OK;
  const size_t seg_cap = (size_t)n / 16 + 4;
  if (grow(wk->A, (size_t)n * 4) || grow(wk->B, (size_t)n * 4) || grow(wk->seg[0], seg_cap * sizeof(Seg)) || grow(wk->seg[1], seg_cap * sizeof(Seg)) ||
      grow(wk->seg[2], seg_cap * sizeof(Seg)) || grow(wk->seg[3], seg_cap * sizeof(Seg)) || grow(wk->heap, seg_cap * sizeof(Seg)) || grow(wk->counts, 64)) {
    wfm_set_error(h, "out of device memory (closing sort)");
    return WFM_E_NOMEM;
  }
  int lg = 0;
  while (((int64_t)1 << (lg + 1)) <= n) ++lg;
  const Seg root{0, n, 2 * lg, 0};
  Seg* big[2] = {(Seg*)wk->seg[0].p, (Seg*)wk->seg[1].p};
  Seg* small_[2] = {(Seg*)wk->seg[2].p, (Seg*)wk->seg[3].p};
  int* d_counts = (int*)wk->counts.p;
  int nbig = 0, nsmall = 0, nheap_total = 0;
  if (n > SMALL_SEG) { HIPCHK(h, hipMemcpyAsync(big[0], &root, sizeof(Seg), hipMemcpyHostToDevice, st)); nbig = 1; }
  else { HIPCHK(h, hipMemcpyAsync(small_[0], &root, sizeof(Seg), hipMemcpyHostToDevice, st)); nsmall = 1; }
  HIPCHK(h, hipMemsetAsync(d_counts, 0, 16, st));
  int cur = 0, levels = 0;
  while (nbig || nsmall) {
    // counts[0], counts[1]: the next level's lists; counts[2]: ranges for the heapsort (kept across levels)
    HIPCHK(h, hipMemsetAsync(d_counts, 0, 8, st));
    if (nbig)
      hipLaunchKernelGGL(sortlike_level_kernel<16>, dim3((unsigned)nbig), dim3(1024), 0, st, big[cur], key, idx, (uint32_t*)wk->A.p, (uint32_t*)wk->B.p, big[cur ^ 1],
                         small_[cur ^ 1], (Seg*)wk->heap.p, d_counts);
    if (nsmall)
      hipLaunchKernelGGL(sortlike_level_kernel<1>, dim3((unsigned)nsmall), dim3(64), 0, st, small_[cur], key, idx, (uint32_t*)wk->A.p, (uint32_t*)wk->B.p, big[cur ^ 1],
                         small_[cur ^ 1], (Seg*)wk->heap.p, d_counts);
    HIPCHK(h, hipGetLastError());
    int c[4] = {0, 0, 0, 0};
    HIPCHK(h, hipMemcpyAsync(c, d_counts, 12, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    nbig = c[0]; nsmall = c[1]; nheap_total = c[2];
    cur ^= 1;
    ++levels;
    if (levels > 4 * lg + 8) { wfm_set_error(h, "closing sort: the recursion does not end"); return WFM_E_HIP; }
  }
  if (nheap_total) {  // ranges that spent their depth budget: __partial_sort(first, last, last) = make_heap + sort_heap
    std::vector<Seg> hs((size_t)nheap_total);
    HIPCHK(h, hipMemcpy(hs.data(), wk->heap.p, (size_t)nheap_total * sizeof(Seg), hipMemcpyDeviceToHost));
    std::vector<std::pair<uint64_t, uint32_t>> v;
    std::vector<uint64_t> hk;
    std::vector<uint32_t> hi;
    for (const Seg& s : hs) {
      const size_t m = (size_t)(s.l - s.f);
      hk.resize(m); hi.resize(m); v.resize(m);
      HIPCHK(h, hipMemcpy(hk.data(), key + s.f, m * 8, hipMemcpyDeviceToHost));
      HIPCHK(h, hipMemcpy(hi.data(), idx + s.f, m * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < m; ++i) v[i] = {hk[i], hi[i]};
      auto lessk = [](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) { return a.first < b.first; };
      std::make_heap(v.begin(), v.end(), lessk);
      std::sort_heap(v.begin(), v.end(), lessk);
      for (size_t i = 0; i < m; ++i) { hk[i] = v[i].first; hi[i] = v[i].second; }
      HIPCHK(h, hipMemcpy(key + s.f, hk.data(), m * 8, hipMemcpyHostToDevice));
      HIPCHK(h, hipMemcpy(idx + s.f, hi.data(), m * 4, hipMemcpyHostToDevice));
    }
  }
  if (levels_out) *levels_out = levels;
  if (heaps_out) *heaps_out = nheap_total;
  return WFM_OK;
}

}  // namespace

void map_finish_work_free(MapFinishWork* wk) {
  if (!wk) return;
  for (MapFinishWork::Buf* b : {&wk->ns, &wk->np, &wk->os, &wk->op, &wk->R, &wk->key, &wk->idx, &wk->key2, &wk->idx2, &wk->A, &wk->B, &wk->seg[0], &wk->seg[1], &wk->seg[2],
                                &wk->seg[3], &wk->heap, &wk->counts, &wk->tmp, &wk->out}) {
    if (b->p) (void)hipFree(b->p);
    b->p = nullptr; b->bytes = 0;
  }
}

// Raw records of one sequence (emission order, on the device) -> addMinmers' records: *d_out (inside wk, valid until the next
// call) holds *n_out of them.
int map_finish_records_device(wfm_handle_t* h, const wfm_minmer_t* d_raw, int64_t n_raw, int w, MapFinishWork* wk, wfm_minmer_t** d_out, int64_t* n_out,
                              MapFinishInfo* info) {
  if (!h || !wk || !d_out || !n_out || (n_raw && !d_raw)) return WFM_E_ARG;
  *d_out = nullptr; *n_out = 0;
  MapFinishInfo inf{};
  if (n_raw == 0) { if (info) *info = inf; return WFM_OK; }
  if (n_raw >= ((int64_t)1 << 31)) { wfm_set_error(h, "too many records for the closing sort"); return WFM_E_UNSUPPORTED; }
  hipStream_t st = wfm_stream(h);
  const size_t nr = (size_t)n_raw;
  if (grow(wk->ns, nr * 4 + 4) || grow(wk->np, nr * 4 + 4) || grow(wk->os, nr * 4 + 4) || grow(wk->op, nr * 4 + 4)) { wfm_set_error(h, "out of device memory (closing sort)"); return WFM_E_NOMEM; }
  const unsigned gb = (unsigned)((nr + 255) / 256);
  hipLaunchKernelGGL(finish_count_kernel, dim3(gb), dim3(256), 0, st, d_raw, n_raw, w, (uint32_t*)wk->ns.p, (uint32_t*)wk->np.p);
  HIPCHK(h, hipGetLastError());
  int rc = excl_scan<uint32_t>(h, wk, (const uint32_t*)wk->ns.p, (uint32_t*)wk->os.p, nr, st);
  if (rc == WFM_OK) rc = excl_scan<uint32_t>(h, wk, (const uint32_t*)wk->np.p, (uint32_t*)wk->op.p, nr, st);
  if (rc != WFM_OK) return rc;
  uint32_t last[4] = {0, 0, 0, 0};
  HIPCHK(h, hipMemcpyAsync(&last[0], (uint32_t*)wk->os.p + nr - 1, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&last[1], (uint32_t*)wk->ns.p + nr - 1, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&last[2], (uint32_t*)wk->op.p + nr - 1, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&last[3], (uint32_t*)wk->np.p + nr - 1, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  const int64_t total_short = (int64_t)last[0] + last[1], total = total_short + (int64_t)last[2] + last[3];
  inf.laid_out = total;
  if (total == 0) { if (info) *info = inf; return WFM_OK; }
  if (total >= ((int64_t)1 << 31)) { wfm_set_error(h, "too many records for the closing sort"); return WFM_E_UNSUPPORTED; }
  const size_t nt = (size_t)total;
  if (grow(wk->R, nt * sizeof(wfm_minmer_t)) || grow(wk->key, nt * 8) || grow(wk->idx, nt * 4) || grow(wk->key2, nt * 8) || grow(wk->idx2, nt * 4) ||
      grow(wk->out, nt * sizeof(wfm_minmer_t))) {
    wfm_set_error(h, "out of device memory (closing sort)");
    return WFM_E_NOMEM;
  }
  wfm_minmer_t* R = (wfm_minmer_t*)wk->R.p;
  uint64_t* key = (uint64_t*)wk->key.p;
  uint32_t* idx = (uint32_t*)wk->idx.p;
  hipLaunchKernelGGL(finish_layout_kernel, dim3(gb), dim3(256), 0, st, d_raw, n_raw, w, (const uint32_t*)wk->os.p, (const uint32_t*)wk->op.p, (uint32_t)total_short, R, key, idx);
  HIPCHK(h, hipGetLastError());
  rc = sortlike_loop_device(h, wk, key, idx, total, st, &inf.levels, &inf.heap_ranges);
  if (rc != WFM_OK) return rc;
  {  // the insertion sort that closes std::sort: stable, over everything
    size_t tmp = 0;
    HIPCHK(h, rocprim::radix_sort_pairs(nullptr, tmp, key, (uint64_t*)wk->key2.p, idx, (uint32_t*)wk->idx2.p, nt, 0, 64, st));
    if (grow(wk->tmp, tmp)) { wfm_set_error(h, "out of device memory (closing sort)"); return WFM_E_NOMEM; }
    HIPCHK(h, rocprim::radix_sort_pairs(wk->tmp.p, tmp, key, (uint64_t*)wk->key2.p, idx, (uint32_t*)wk->idx2.p, nt, 0, 64, st));
  }
  const uint32_t* order = (const uint32_t*)wk->idx2.p;
  // keep flags and their offsets reuse the count buffers (nt may exceed nr: pieces)
  if (grow(wk->ns, nt * 4 + 4) || grow(wk->os, nt * 4 + 4)) { wfm_set_error(h, "out of device memory (closing sort)"); return WFM_E_NOMEM; }
  const unsigned gt = (unsigned)((nt + 255) / 256);
  hipLaunchKernelGGL(finish_flag_kernel, dim3(gt), dim3(256), 0, st, R, order, total, (uint32_t*)wk->ns.p);
  HIPCHK(h, hipGetLastError());
  rc = excl_scan<uint32_t>(h, wk, (const uint32_t*)wk->ns.p, (uint32_t*)wk->os.p, nt, st);
  if (rc != WFM_OK) return rc;
  hipLaunchKernelGGL(finish_emit_kernel, dim3(gt), dim3(256), 0, st, R, order, (const uint32_t*)wk->ns.p, (const uint32_t*)wk->os.p, total, (wfm_minmer_t*)wk->out.p);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(&last[0], (uint32_t*)wk->os.p + nt - 1, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&last[1], (uint32_t*)wk->ns.p + nt - 1, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  *d_out = (wfm_minmer_t*)wk->out.p;
  *n_out = (int64_t)last[0] + last[1];
  inf.records = *n_out;
  if (info) *info = inf;
  return WFM_OK;
}

// The same steps on the host, for the CPU suite: the arrangement the device computes (lists A and B, T, the cut, the
// closing stable sort), held against std::sort itself.
void map_sortlike_model(std::vector<std::pair<uint64_t, uint32_t>>& v) {
  const int64_t n = (int64_t)v.size();
  auto lessk = [](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) { return a.first < b.first; };
  if (n > 16) {
    std::vector<Seg> st;
    int lg = 0;
    while (((int64_t)1 << (lg + 1)) <= n) ++lg;
    st.push_back(Seg{0, n, 2 * lg, 0});
    std::vector<int64_t> A, B;
    while (!st.empty()) {
      const Seg s = st.back();
      st.pop_back();
      if (s.l - s.f <= 16) continue;
      if (s.depth == 0) { std::make_heap(v.begin() + s.f, v.begin() + s.l, lessk); std::sort_heap(v.begin() + s.f, v.begin() + s.l, lessk); continue; }
      const int64_t f = s.f, l = s.l, a = f + 1, b = f + (l - f) / 2, c = l - 1;
      int64_t pick;
      if (v[(size_t)a].first < v[(size_t)b].first) { if (v[(size_t)b].first < v[(size_t)c].first) pick = b; else if (v[(size_t)a].first < v[(size_t)c].first) pick = c; else pick = a; }
      else if (v[(size_t)a].first < v[(size_t)c].first) pick = a;
      else if (v[(size_t)b].first < v[(size_t)c].first) pick = c;
      else pick = b;
      std::swap(v[(size_t)f], v[(size_t)pick]);
      const uint64_t p = v[(size_t)f].first;
      A.clear(); B.clear();
      for (int64_t i = f + 1; i < l; ++i) if (v[(size_t)i].first >= p) A.push_back(i);
      for (int64_t i = l - 1; i > f; --i) if (v[(size_t)i].first <= p) B.push_back(i);
      size_t T = 0;
      while (T < A.size() && T < B.size() && A[T] < B[T]) ++T;
      for (size_t t = 0; t < T; ++t) std::swap(v[(size_t)A[t]], v[(size_t)B[t]]);
      const int64_t ca = T < A.size() ? A[T] : INT64_MAX, cb = T > 0 ? B[T - 1] : INT64_MAX;
      const int64_t cut = std::min(ca, cb);
      st.push_back(Seg{f, cut, s.depth - 1, 0});
      st.push_back(Seg{cut, l, s.depth - 1, 0});
    }
  }
  std::stable_sort(v.begin(), v.end(), lessk);
}

// Test hooks.  wfmh_test_sortlike_model: the host restatement on records (key = (wpos, wpos_end)), in place.
extern "C" void wfmh_test_sortlike_model(wfm_minmer_t* recs, int64_t n) {
  std::vector<std::pair<uint64_t, uint32_t>> v((size_t)n);
  for (int64_t i = 0; i < n; ++i) v[(size_t)i] = {((uint64_t)recs[i].wpos << 32) | (uint64_t)(uint32_t)recs[i].wpos_end, (uint32_t)i};
  map_sortlike_model(v);
  std::vector<wfm_minmer_t> out((size_t)n);
  for (int64_t i = 0; i < n; ++i) out[(size_t)i] = recs[v[(size_t)i].second];
  if (n) memcpy(recs, out.data(), (size_t)n * sizeof(wfm_minmer_t));
}
// wfm_finish_records: raw records (host) through the device's closing steps; returns the number of records written to out
extern "C" int64_t wfm_finish_records(wfm_handle_t* h, const wfm_minmer_t* raw, int64_t n, int w, wfm_minmer_t* out, int64_t cap, int32_t* levels, int32_t* heap_ranges) {
  if (!h || n < 0 || (n && !raw)) return WFM_E_ARG;
  if (hipSetDevice(wfm_device(h)) != hipSuccess) return WFM_E_HIP;
  wfm_minmer_t* d_raw = nullptr;
  if (n && hipMalloc((void**)&d_raw, (size_t)n * sizeof(wfm_minmer_t)) != hipSuccess) return WFM_E_NOMEM;
  if (n && hipMemcpy(d_raw, raw, (size_t)n * sizeof(wfm_minmer_t), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d_raw); return WFM_E_HIP; }
  MapFinishWork wk;
  wfm_minmer_t* d_out = nullptr;
  int64_t n_out = 0;
  MapFinishInfo inf;
  int rc = map_finish_records_device(h, d_raw, n, w, &wk, &d_out, &n_out, &inf);
  if (rc == WFM_OK && n_out && hipMemcpy(out, d_out, (size_t)std::min(n_out, cap) * sizeof(wfm_minmer_t), hipMemcpyDeviceToHost) != hipSuccess) rc = WFM_E_HIP;
  if (levels) *levels = inf.levels;
  if (heap_ranges) *heap_ranges = inf.heap_ranges;
  map_finish_work_free(&wk);
  if (d_raw) (void)hipFree(d_raw);
  return rc == WFM_OK ? n_out : rc;
}
