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
// * Ranges are independent, so all ranges of one recursion depth are partitioned side by side: ranges above 128 k elements by
//   many workgroups each (huge_* kernels: counts per tile, offsets, lists, cut, swaps), ranges up to 128 k by one workgroup
//   each (sortlike_level_kernel), and a range of at most 1 k elements by one wave that keeps it in LDS through all its
//   remaining depths (sortlike_small_kernel).  A range that spends its depth budget (2 floor(log2 n) levels; not seen
//   outside adversarial inputs) is heap-sorted by the library itself on the host.
// The CPU suite holds a host restatement of these steps (map_sortlike_model) against std::sort, the GPU suite holds the
// kernels against the host's finish_records (tests/test_minmers.py).
#include <hip/hip_runtime.h>
#include "dev_cache.h"
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
constexpr int64_t SMALL_SEG = 1024;       // ranges up to this size are finished by one wave in LDS, all their depths in one go
constexpr int64_t HUGE_SEG = 128 * 1024;  // ranges above this size are partitioned by many workgroups (tiles of HUGE_TILE elements)
constexpr int HUGE_TILE = 8192;
struct HugeTile { int32_t seg, tile; };
struct HugeInfo { uint64_t pivot; int32_t nA, nB, T, pad_; int64_t cut; };

__device__ inline void swap_elems(uint64_t* key, uint32_t* idx, int64_t a, int64_t b) {
  const uint64_t k = key[a]; key[a] = key[b]; key[b] = k;
  const uint32_t i = idx[a]; idx[a] = idx[b]; idx[b] = i;
}

// counts[0]: next level's big ranges, [1]: small ranges (all levels), [2]: ranges for the heapsort, [3]: next level's huge ranges
__device__ inline void push_child(const Seg& c, Seg* next_huge, Seg* next_big, Seg* small_, int* counts) {
  const int64_t sz = c.l - c.f;
  if (sz <= 16) return;
  if (sz > HUGE_SEG) next_huge[atomicAdd(&counts[3], 1)] = c;
  else if (sz > SMALL_SEG) next_big[atomicAdd(&counts[0], 1)] = c;
  else small_[atomicAdd(&counts[1], 1)] = c;
}
__device__ inline int64_t median_pick(const uint64_t* key, int64_t f, int64_t l) {  // __move_median_to_first(first, first + 1, mid, last - 1)
  const int64_t a = f + 1, b = f + (l - f) / 2, c = l - 1;
  const uint64_t ka = key[a], kb = key[b], kc = key[c];
  if (ka < kb) { if (kb < kc) return b; if (ka < kc) return c; return a; }
  if (ka < kc) return a;
  if (kb < kc) return c;
  return b;
}

// One range per workgroup of NW waves: pivot, the lists A and B (positions relative to the array), the swaps, the children.
template <int NW>
__global__ void __launch_bounds__(NW * 64) sortlike_level_kernel(const Seg* cur, uint64_t* key, uint32_t* idx, uint32_t* A, uint32_t* B, Seg* next_huge, Seg* next_big,
                                                                  Seg* next_small, Seg* heap, int* counts) {
  const Seg sg = cur[blockIdx.x];
  const int64_t f = sg.f, l = sg.l;
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  __shared__ uint64_t s_p;
  __shared__ int s_ge[NW], s_le[NW], s_T[NW];
  if (sg.depth == 0) {  // budget spent: the library's heapsort on the host
    if (tid == 0) heap[atomicAdd(&counts[2], 1)] = sg;
    return;
  }
  if (tid == 0) {
    swap_elems(key, idx, f, median_pick(key, f, l));
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
      push_child(c, next_huge, next_big, next_small, counts);
    }
  }
}

// ---- a range of more than HUGE_SEG elements: the same partition, spread over many workgroups ----
__global__ void huge_pivot_kernel(const Seg* segs, uint64_t* key, uint32_t* idx, HugeInfo* info, Seg* heap, int* counts) {
  const Seg sg = segs[blockIdx.x];
  if (threadIdx.x != 0) return;
  HugeInfo& I = info[blockIdx.x];
  I.nA = I.nB = I.T = 0; I.cut = -1;
  if (sg.depth == 0) { heap[atomicAdd(&counts[2], 1)] = sg; I.cut = -2; return; }  // budget spent: no partition
  swap_elems(key, idx, sg.f, median_pick(key, sg.f, sg.l));
  I.pivot = key[sg.f];
}
// elements >= pivot and <= pivot per tile
__global__ void __launch_bounds__(256) huge_count_kernel(const Seg* segs, const HugeTile* tiles, const uint64_t* key, const HugeInfo* info, int2* tile_cnt) {
  const HugeTile t = tiles[blockIdx.x];
  const HugeInfo I = info[t.seg];
  if (I.cut == -2) return;
  const Seg sg = segs[t.seg];
  const int64_t lo = sg.f + 1 + (int64_t)t.tile * HUGE_TILE, hi = lo + HUGE_TILE < sg.l ? lo + HUGE_TILE : sg.l;
  int cge = 0, cle = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { const uint64_t k = key[i]; cge += k >= I.pivot; cle += k <= I.pivot; }
  __shared__ int s_a[4], s_b[4];
  for (int o = 32; o; o >>= 1) { cge += __shfl_xor(cge, o, 64); cle += __shfl_xor(cle, o, 64); }
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = cge; s_b[threadIdx.x >> 6] = cle; }
  __syncthreads();
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = make_int2(s_a[0] + s_a[1] + s_a[2] + s_a[3], s_b[0] + s_b[1] + s_b[2] + s_b[3]);
}
// per range: its tiles' counts -> offsets (in place), totals
__global__ void __launch_bounds__(256) huge_offsets_kernel(const int* seg_tile0, int2* tile_cnt, HugeInfo* info) {
  const int s = blockIdx.x;
  if (info[s].cut == -2) return;
  const int t0 = seg_tile0[s], t1 = seg_tile0[s + 1];
  __shared__ int s_a[256], s_b[256];
  int ra = 0, rb = 0;  // running totals of the tiles before this round
  for (int base = t0; base < t1; base += 256) {
    const int t = base + (int)threadIdx.x;
    const int2 c = t < t1 ? tile_cnt[t] : make_int2(0, 0);
    s_a[threadIdx.x] = c.x; s_b[threadIdx.x] = c.y;
    __syncthreads();
    int pa = 0, pb = 0, ta = 0, tb = 0;
    for (int q = 0; q < 256; ++q) { if (q < (int)threadIdx.x) { pa += s_a[q]; pb += s_b[q]; } ta += s_a[q]; tb += s_b[q]; }
    if (t < t1) tile_cnt[t] = make_int2(ra + pa, rb + pb);
    ra += ta; rb += tb;
    __syncthreads();
  }
  if (threadIdx.x == 0) { info[s].nA = ra; info[s].nB = rb; }
}
// the lists
__global__ void __launch_bounds__(256) huge_lists_kernel(const Seg* segs, const HugeTile* tiles, const uint64_t* key, const HugeInfo* info, const int2* tile_off, uint32_t* A,
                                                         uint32_t* B) {
  const HugeTile t = tiles[blockIdx.x];
  const HugeInfo I = info[t.seg];
  if (I.cut == -2) return;
  const Seg sg = segs[t.seg];
  const int64_t lo0 = sg.f + 1;
  const int64_t lo = lo0 + (int64_t)t.tile * HUGE_TILE, hi = lo + HUGE_TILE < sg.l ? lo + HUGE_TILE : sg.l;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __shared__ int s_a[4], s_b[4];
  int oge = tile_off[blockIdx.x].x, ole = tile_off[blockIdx.x].y;
  for (int64_t base = lo; base < hi; base += 256) {
    const int64_t i = base + threadIdx.x;
    const bool in = i < hi;
    const uint64_t k = in ? key[i] : 0ull;
    const bool ge = in && k >= I.pivot, le = in && k <= I.pivot;
    const unsigned long long mg = __ballot(ge), ml = __ballot(le), below = (1ull << lane) - 1ull;
    if (lane == 0) { s_a[wv] = __popcll(mg); s_b[wv] = __popcll(ml); }
    __syncthreads();
    int wa = 0, wb = 0, ta = 0, tb = 0;
    for (int q = 0; q < 4; ++q) { if (q < wv) { wa += s_a[q]; wb += s_b[q]; } ta += s_a[q]; tb += s_b[q]; }
    if (ge) A[lo0 + oge + wa + __popcll(mg & below)] = (uint32_t)(i - lo0);
    if (le) B[lo0 + (I.nB - 1 - (ole + wb + __popcll(ml & below)))] = (uint32_t)(i - lo0);
    oge += ta; ole += tb;
    __syncthreads();
  }
}
// T and the cut (A[t] < B[t] holds for t < T and for no later t: a bisection), the children
__global__ void huge_cut_kernel(const Seg* segs, HugeInfo* info, const uint32_t* A, const uint32_t* B, Seg* next_huge, Seg* next_big, Seg* small_, int* counts) {
  const int s = blockIdx.x;
  if (threadIdx.x != 0 || info[s].cut == -2) return;
  const Seg sg = segs[s];
  HugeInfo& I = info[s];
  const int64_t lo = sg.f + 1;
  int a = 0, b = I.nA < I.nB ? I.nA : I.nB;  // T in [a, b]
  while (a < b) { const int m = a + (b - a) / 2; if (A[lo + m] < B[lo + m]) a = m + 1; else b = m; }
  const int T = a;
  const int64_t ca = T < I.nA ? lo + (int64_t)A[lo + T] : INT64_MAX, cb = T > 0 ? lo + (int64_t)B[lo + T - 1] : INT64_MAX;
  I.T = T; I.cut = ca < cb ? ca : cb;
  push_child(Seg{sg.f, I.cut, sg.depth - 1, 0}, next_huge, next_big, small_, counts);
  push_child(Seg{I.cut, sg.l, sg.depth - 1, 0}, next_huge, next_big, small_, counts);
}
__global__ void __launch_bounds__(256) huge_swap_kernel(const Seg* segs, const HugeTile* tiles, const HugeInfo* info, const uint32_t* A, const uint32_t* B, uint64_t* key,
                                                        uint32_t* idx) {
  const HugeTile t = tiles[blockIdx.x];
  const HugeInfo I = info[t.seg];
  if (I.cut == -2) return;
  const int64_t lo = segs[t.seg].f + 1;
  const int64_t t0 = (int64_t)t.tile * HUGE_TILE, t1 = t0 + HUGE_TILE < I.T ? t0 + HUGE_TILE : I.T;
  for (int64_t q = t0 + threadIdx.x; q < t1; q += 256) swap_elems(key, idx, lo + (int64_t)A[lo + q], lo + (int64_t)B[lo + q]);
}

// ---- a range of at most SMALL_SEG elements: one wave, everything in LDS, all depths ----
__global__ void __launch_bounds__(64) sortlike_small_kernel(const Seg* segs, uint64_t* key, uint32_t* idx, Seg* heap, int* counts) {
  const Seg sg = segs[blockIdx.x];
  const int m = (int)(sg.l - sg.f), lane = (int)threadIdx.x;
  __shared__ uint64_t k[SMALL_SEG];
  __shared__ uint32_t ix[SMALL_SEG];
  __shared__ uint16_t A[SMALL_SEG], B[SMALL_SEG];
  __shared__ int st_f[64], st_l[64], st_d[64];
  for (int i = lane; i < m; i += 64) { k[i] = key[sg.f + i]; ix[i] = idx[sg.f + i]; }
  __syncthreads();
  int sp = 1;
  if (lane == 0) { st_f[0] = 0; st_l[0] = m; st_d[0] = sg.depth; }
  __syncthreads();
  while (sp > 0) {
    --sp;
    int f = st_f[sp], l = st_l[sp], d = st_d[sp];
    __syncthreads();
    while (l - f > 16) {
      if (d == 0) {  // budget spent: the library's heapsort on the host
        if (lane == 0) heap[atomicAdd(&counts[2], 1)] = Seg{sg.f + f, sg.f + l, 0, 0};
        break;
      }
      --d;
      if (lane == 0) {
        const int a = f + 1, b = f + (l - f) / 2, c = l - 1;
        const uint64_t ka = k[a], kb = k[b], kc = k[c];
        int pick;
        if (ka < kb) { if (kb < kc) pick = b; else if (ka < kc) pick = c; else pick = a; }
        else if (ka < kc) pick = a;
        else if (kb < kc) pick = c;
        else pick = b;
        const uint64_t tk = k[f]; k[f] = k[pick]; k[pick] = tk;
        const uint32_t ti = ix[f]; ix[f] = ix[pick]; ix[pick] = ti;
      }
      __syncthreads();
      const uint64_t p = k[f];
      const int lo = f + 1;
      int nA = 0, nB = 0;
      for (int base = lo; base < l; base += 64) {
        const int i = base + lane;
        const bool in = i < l;
        const uint64_t kk = in ? k[i] : 0ull;
        const bool ge = in && kk >= p, le = in && kk <= p;
        const unsigned long long mg = __ballot(ge), ml = __ballot(le), below = (1ull << lane) - 1ull;
        if (ge) A[nA + __popcll(mg & below)] = (uint16_t)i;
        if (le) B[nB + __popcll(ml & below)] = (uint16_t)i;  // ascending here: B[t] of the text is B[nB - 1 - t]
        nA += __popcll(mg); nB += __popcll(ml);
      }
      __syncthreads();
      const int nmin = nA < nB ? nA : nB;
      int T = 0;
      for (int base = 0; base < nmin; base += 64) {
        const int t = base + lane;
        const unsigned long long mk = __ballot(t < nmin && A[t] < B[nB - 1 - t]);
        T += __popcll(mk);
        if (mk != ~0ull) break;  // the condition holds for a prefix only
      }
      const int ca = T < nA ? (int)A[T] : INT32_MAX, cb = T > 0 ? (int)B[nB - T] : INT32_MAX;
      const int cut = ca < cb ? ca : cb;
      for (int t = lane; t < T; t += 64) {
        const int i = A[t], j = B[nB - 1 - t];
        const uint64_t tk = k[i]; k[i] = k[j]; k[j] = tk;
        const uint32_t ti = ix[i]; ix[i] = ix[j]; ix[j] = ti;
      }
      __syncthreads();
      if (l - cut > 16) {
        if (lane == 0) { st_f[sp] = cut; st_l[sp] = l; st_d[sp] = d; }
        ++sp;
      }
      l = cut;
      __syncthreads();
    }
  }
  for (int i = lane; i < m; i += 64) { key[sg.f + i] = k[i]; idx[sg.f + i] = ix[i]; }
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

// (only the stream the block was used on is waited for: a device-wide wait would stall the other device thread's stream)
int grow(MapFinishWork::Buf& b, size_t bytes, hipStream_t st) {
  if (b.bytes >= bytes && b.p) return WFM_OK;
  if (b.p) { (void)hipStreamSynchronize(st); wfm_dfree_nosync(b.p); }
  b.p = nullptr; b.bytes = 0;
  const size_t want = bytes + bytes / 4 + 256;
  if (wfm_dmalloc(&b.p, want) != hipSuccess) return WFM_E_NOMEM;
  b.bytes = want;
  return WFM_OK;
}

template <typename T>
int excl_scan(wfm_handle_t* h, MapFinishWork* wk, const T* in, T* out, size_t n, hipStream_t st) {
  size_t tmp = 0;
  HIPCHK(h, rocprim::exclusive_scan(nullptr, tmp, in, out, (T)0, n, rocprim::plus<T>(), st));
  if (grow(wk->tmp, tmp, st)) return WFM_E_NOMEM;
  HIPCHK(h, rocprim::exclusive_scan(wk->tmp.p, tmp, in, out, (T)0, n, rocprim::plus<T>(), st));
  return WFM_OK;
}

// the arrangement std::sort's introsort loop leaves, on (key, idx) pairs of the device
int sortlike_loop_device(wfm_handle_t* h, MapFinishWork* wk, uint64_t* key, uint32_t* idx, int64_t n, hipStream_t st, int* levels_out, int* heaps_out) {
  if (levels_out) *levels_out = 0;
  if (heaps_out) *heaps_out = 0;
  if (n <= 16) return WFM_OK;
  const size_t cap_small = (size_t)n / 16 + 4, cap_big = (size_t)n / (size_t)SMALL_SEG + 4, cap_huge = (size_t)n / (size_t)HUGE_SEG + 4;
  const size_t cap_tiles = (size_t)n / HUGE_TILE + cap_huge + 4;
  if (grow(wk->A, (size_t)n * 4, st) || grow(wk->B, (size_t)n * 4, st) || grow(wk->seg[0], cap_big * sizeof(Seg), st) || grow(wk->seg[1], cap_big * sizeof(Seg), st) ||
      grow(wk->seg[2], cap_huge * sizeof(Seg), st) || grow(wk->seg[3], cap_huge * sizeof(Seg), st) || grow(wk->small_, cap_small * sizeof(Seg), st) ||
      grow(wk->heap, cap_small * sizeof(Seg), st) || grow(wk->counts, 64, st) || grow(wk->tiles, cap_tiles * sizeof(HugeTile), st) || grow(wk->tile_cnt, cap_tiles * sizeof(int2), st) ||
      grow(wk->tile0, (cap_huge + 1) * sizeof(int), st) || grow(wk->info, cap_huge * sizeof(HugeInfo), st)) {
    wfm_set_error(h, "out of device memory (closing sort)");
    return WFM_E_NOMEM;
  }
  int lg = 0;
  while (((int64_t)1 << (lg + 1)) <= n) ++lg;
  const Seg root{0, n, 2 * lg, 0};
  Seg* big[2] = {(Seg*)wk->seg[0].p, (Seg*)wk->seg[1].p};
  Seg* huge[2] = {(Seg*)wk->seg[2].p, (Seg*)wk->seg[3].p};
  Seg* small_ = (Seg*)wk->small_.p;
  Seg* heap = (Seg*)wk->heap.p;
  uint32_t *A = (uint32_t*)wk->A.p, *B = (uint32_t*)wk->B.p;
  int* d_counts = (int*)wk->counts.p;
  int nbig = 0, nhuge = 0, nsmall = 0, nheap_total = 0;
  HIPCHK(h, hipMemsetAsync(d_counts, 0, 16, st));
  if (n > HUGE_SEG) { HIPCHK(h, hipMemcpyAsync(huge[0], &root, sizeof(Seg), hipMemcpyHostToDevice, st)); nhuge = 1; }
  else if (n > SMALL_SEG) { HIPCHK(h, hipMemcpyAsync(big[0], &root, sizeof(Seg), hipMemcpyHostToDevice, st)); nbig = 1; }
  else { HIPCHK(h, hipMemcpyAsync(small_, &root, sizeof(Seg), hipMemcpyHostToDevice, st)); nsmall = 1; const int one = 1; HIPCHK(h, hipMemcpyAsync(d_counts + 1, &one, 4, hipMemcpyHostToDevice, st)); }
  int cur = 0, levels = 0;
  std::vector<Seg> hsegs;
  std::vector<HugeTile> tiles;
  std::vector<int> tile0;
  while (nbig || nhuge) {
    // counts[0] / [3]: the next level's big / huge ranges; [1] small ranges and [2] ranges for the heapsort keep counting
    HIPCHK(h, hipMemsetAsync(d_counts, 0, 4, st));
    HIPCHK(h, hipMemsetAsync(d_counts + 3, 0, 4, st));
    if (nhuge) {
      hsegs.resize((size_t)nhuge);
      HIPCHK(h, hipMemcpyAsync(hsegs.data(), huge[cur], (size_t)nhuge * sizeof(Seg), hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      tiles.clear(); tile0.assign(1, 0);
      for (int q = 0; q < nhuge; ++q) {
        const int64_t m = hsegs[(size_t)q].l - hsegs[(size_t)q].f - 1;
        const int nt = (int)((m + HUGE_TILE - 1) / HUGE_TILE);
        for (int t = 0; t < nt; ++t) tiles.push_back(HugeTile{q, t});
        tile0.push_back((int)tiles.size());
      }
      if (tiles.size() > cap_tiles) { wfm_set_error(h, "closing sort: tile list overflow"); return WFM_E_HIP; }
      HugeTile* d_tiles = (HugeTile*)wk->tiles.p;
      int2* d_tc = (int2*)wk->tile_cnt.p;
      int* d_t0 = (int*)wk->tile0.p;
      HugeInfo* d_info = (HugeInfo*)wk->info.p;
      HIPCHK(h, hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(HugeTile), hipMemcpyHostToDevice, st));
      HIPCHK(h, hipMemcpyAsync(d_t0, tile0.data(), tile0.size() * sizeof(int), hipMemcpyHostToDevice, st));
      const unsigned nt = (unsigned)tiles.size();
      hipLaunchKernelGGL(huge_pivot_kernel, dim3((unsigned)nhuge), dim3(64), 0, st, huge[cur], key, idx, d_info, heap, d_counts);
      hipLaunchKernelGGL(huge_count_kernel, dim3(nt), dim3(256), 0, st, huge[cur], d_tiles, key, d_info, d_tc);
      hipLaunchKernelGGL(huge_offsets_kernel, dim3((unsigned)nhuge), dim3(256), 0, st, d_t0, d_tc, d_info);
      hipLaunchKernelGGL(huge_lists_kernel, dim3(nt), dim3(256), 0, st, huge[cur], d_tiles, key, d_info, d_tc, A, B);
      hipLaunchKernelGGL(huge_cut_kernel, dim3((unsigned)nhuge), dim3(64), 0, st, huge[cur], d_info, A, B, huge[cur ^ 1], big[cur ^ 1], small_, d_counts);
      hipLaunchKernelGGL(huge_swap_kernel, dim3(nt), dim3(256), 0, st, huge[cur], d_tiles, d_info, A, B, key, idx);
    }
    if (nbig)
      hipLaunchKernelGGL(sortlike_level_kernel<16>, dim3((unsigned)nbig), dim3(1024), 0, st, big[cur], key, idx, A, B, huge[cur ^ 1], big[cur ^ 1], small_, heap, d_counts);
    HIPCHK(h, hipGetLastError());
    int c[4] = {0, 0, 0, 0};
    HIPCHK(h, hipMemcpyAsync(c, d_counts, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    nbig = c[0]; nsmall = c[1]; nheap_total = c[2]; nhuge = c[3];
    cur ^= 1;
    ++levels;
    if (levels > 4 * lg + 8) { wfm_set_error(h, "closing sort: the recursion does not end"); return WFM_E_HIP; }
  }
  if (nsmall) {
    hipLaunchKernelGGL(sortlike_small_kernel, dim3((unsigned)nsmall), dim3(64), 0, st, small_, key, idx, heap, d_counts);
    HIPCHK(h, hipGetLastError());
    int c[4] = {0, 0, 0, 0};
    HIPCHK(h, hipMemcpyAsync(c, d_counts, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    nheap_total = c[2];
  }
  if (nheap_total) {  // ranges that spent their depth budget: __partial_sort(first, last, last) = make_heap + sort_heap
    std::vector<Seg> hs((size_t)nheap_total);
    HIPCHK(h, hipMemcpyAsync(hs.data(), wk->heap.p, (size_t)nheap_total * sizeof(Seg), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    std::vector<std::pair<uint64_t, uint32_t>> v;
    std::vector<uint64_t> hk;
    std::vector<uint32_t> hi;
    for (const Seg& s : hs) {
      const size_t m = (size_t)(s.l - s.f);
      hk.resize(m); hi.resize(m); v.resize(m);
      HIPCHK(h, hipMemcpyAsync(hk.data(), key + s.f, m * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipMemcpyAsync(hi.data(), idx + s.f, m * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      for (size_t i = 0; i < m; ++i) v[i] = {hk[i], hi[i]};
      auto lessk = [](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) { return a.first < b.first; };
      std::make_heap(v.begin(), v.end(), lessk);
      std::sort_heap(v.begin(), v.end(), lessk);
      for (size_t i = 0; i < m; ++i) { hk[i] = v[i].first; hi[i] = v[i].second; }
      HIPCHK(h, hipMemcpyAsync(key + s.f, hk.data(), m * 8, hipMemcpyHostToDevice, st));
      HIPCHK(h, hipMemcpyAsync(idx + s.f, hi.data(), m * 4, hipMemcpyHostToDevice, st));
      HIPCHK(h, hipStreamSynchronize(st));
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
                                &wk->seg[3], &wk->small_, &wk->heap, &wk->counts, &wk->tiles, &wk->tile_cnt, &wk->tile0, &wk->info, &wk->tmp, &wk->out}) {
    if (b->p) (void)wfm_dfree(b->p);
    b->p = nullptr; b->bytes = 0;
  }
}

// Raw records of one sequence (emission order, on the device) -> addMinmers' records: *d_out (inside wk, valid until the next
// call) holds *n_out of them.
int map_finish_records_device(wfm_handle_t* h, const wfm_minmer_t* d_raw, int64_t n_raw, int w, MapFinishWork* wk, wfm_minmer_t** d_out, int64_t* n_out,
                              MapFinishInfo* info, hipStream_t stream) {
  if (!h || !wk || !d_out || !n_out || (n_raw && !d_raw)) return WFM_E_ARG;
  *d_out = nullptr; *n_out = 0;
  MapFinishInfo inf{};
  if (n_raw == 0) { if (info) *info = inf; return WFM_OK; }
  if (n_raw >= ((int64_t)1 << 31)) { wfm_set_error(h, "too many records for the closing sort"); return WFM_E_UNSUPPORTED; }
  hipStream_t st = stream ? stream : wfm_stream(h);
  const size_t nr = (size_t)n_raw;
  if (grow(wk->ns, nr * 4 + 4, st) || grow(wk->np, nr * 4 + 4, st) || grow(wk->os, nr * 4 + 4, st) || grow(wk->op, nr * 4 + 4, st)) { wfm_set_error(h, "out of device memory (closing sort)"); return WFM_E_NOMEM; }
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
  if (grow(wk->R, nt * sizeof(wfm_minmer_t), st) || grow(wk->key, nt * 8, st) || grow(wk->idx, nt * 4, st) || grow(wk->key2, nt * 8, st) || grow(wk->idx2, nt * 4, st) ||
      grow(wk->out, nt * sizeof(wfm_minmer_t), st)) {
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
    if (grow(wk->tmp, tmp, st)) { wfm_set_error(h, "out of device memory (closing sort)"); return WFM_E_NOMEM; }
    HIPCHK(h, rocprim::radix_sort_pairs(wk->tmp.p, tmp, key, (uint64_t*)wk->key2.p, idx, (uint32_t*)wk->idx2.p, nt, 0, 64, st));
  }
  const uint32_t* order = (const uint32_t*)wk->idx2.p;
  // keep flags and their offsets reuse the count buffers (nt may exceed nr: pieces)
  if (grow(wk->ns, nt * 4 + 4, st) || grow(wk->os, nt * 4 + 4, st)) { wfm_set_error(h, "out of device memory (closing sort)"); return WFM_E_NOMEM; }
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
  if (n && wfm_dmalloc((void**)&d_raw, (size_t)n * sizeof(wfm_minmer_t)) != hipSuccess) return WFM_E_NOMEM;
  if (n && hipMemcpy(d_raw, raw, (size_t)n * sizeof(wfm_minmer_t), hipMemcpyHostToDevice) != hipSuccess) { (void)wfm_dfree(d_raw); return WFM_E_HIP; }
  MapFinishWork wk;
  wfm_minmer_t* d_out = nullptr;
  int64_t n_out = 0;
  MapFinishInfo inf;
  int rc = map_finish_records_device(h, d_raw, n, w, &wk, &d_out, &n_out, &inf);
  if (rc == WFM_OK && n_out && hipMemcpy(out, d_out, (size_t)std::min(n_out, cap) * sizeof(wfm_minmer_t), hipMemcpyDeviceToHost) != hipSuccess) rc = WFM_E_HIP;
  if (levels) *levels = inf.levels;
  if (heap_ranges) *heap_ranges = inf.heap_ranges;
  map_finish_work_free(&wk);
  if (d_raw) (void)wfm_dfree(d_raw);
  return rc == WFM_OK ? n_out : rc;
}
