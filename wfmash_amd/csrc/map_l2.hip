// map_l2.hip -- L2 stage of the mashmap3 mapper on the GPU.
//
//   SlideMapper                 src/map/include/slidingMap.hpp:28-212
//   computeL2MappedRegions      src/map/include/mappingCore.hpp:307-442
//   doL2Mapping                 src/map/include/computeMap.hpp:989-1061
//
// Every L1 candidate is independent: the reference's best-first heap over a fragment's
// candidates only decides where the ANI cutoff stops, and that cutoff is monotone in the
// candidate's intersection size, so it is a per-candidate predicate here.  One lane slides one
// candidate: the rank-s pivot of SlideMapper moves lazily (at most one slot per operation), so
// the walk over the locus' minmer intervals has to be replayed in the reference's order,
// including the order in which libstdc++'s binary heap releases intervals with equal wpos_end.
// Per-candidate state (s+1 slots, the interval heap, the loci) lives in global arenas sized by a
// counting pass; the query sketch is shared read-only by all candidates of a fragment.
// The identity test (computeMap.hpp:1018-1024) depends only on (Q.sketchSize, shared) and comes
// in as host-built tables (skch::Stat needs the binomial quantile, which stays on the host).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/wfmash_hip.h"
#include "wfa_handle.h"
#include "map_device.h"

const wfm_minmer_t* wfm_index_minmers(const wfm_index_t* ix);
int64_t wfm_index_n_kept(const wfm_index_t* ix);

namespace {

#define HIPCHK(h, call)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      wfm_set_error((h), std::string(#call) + ": " + hipGetErrorString(e_));            \
      return WFM_E_HIP;                                                                 \
    }                                                                                   \
  } while (0)

using Scratch = MapScratch;

struct Slot { uint32_t nbi; int16_t vote; uint8_t active; int8_t qstrand; };  // slidingMapContainerValueType minus the hash
struct HeapEnt { int64_t wpos_end; int64_t idx; };
struct Locus { int64_t start, end, mean; int32_t shared; int32_t strand; };

struct DevParams {
  int w, sketch_size, stage1_topani;
};

// first index with (seqId, wpos) >= (seq, pos)
__device__ int64_t mi_lower_bound(const wfm_minmer_t* m, int64_t n, int32_t seq, int64_t pos) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const bool less = m[mid].seqId < seq || (m[mid].seqId == seq && m[mid].wpos < pos);
    if (less) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void l2_range_kernel(const wfm_l1_candidate_t* cand, int64_t ncand, const int32_t* qcount, const double* cutoff_j,
                                const wfm_minmer_t* mi, int64_t n_mi, DevParams P, int64_t* lo_out, uint32_t* heap_cap, uint32_t* loc_cap) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncand) return;
  const wfm_l1_candidate_t L = cand[c];
  const int qs = qcount[L.frag];
  bool ok = qs > 0;
  if (ok && P.stage1_topani) ok = !((double)L.intersectionSize / (double)qs < cutoff_j[qs]);
  int64_t lo = 0;
  uint32_t hc = 0, lc = 0;
  if (ok) {
    lo = mi_lower_bound(mi, n_mi, L.seqId, L.rangeStartPos - P.w - 1);
    const int64_t hi = mi_lower_bound(mi, n_mi, L.seqId, L.rangeEndPos + 1);
    hc = (uint32_t)(hi - lo);
    lc = (uint32_t)((L.rangeEndPos - L.rangeStartPos) / (P.w + 1) + 2);
  }
  lo_out[c] = ok ? lo : -1;
  heap_cap[c] = hc;
  loc_cap[c] = lc;
}

// --- libstdc++ binary-heap element order (bits/stl_heap.h), comp(l, r) = l.wpos_end > r.wpos_end
__device__ __forceinline__ void heap_sift_up(HeapEnt* a, int64_t hole, int64_t top, HeapEnt v) {
  int64_t parent = (hole - 1) / 2;
  while (hole > top && a[parent].wpos_end > v.wpos_end) {
    a[hole] = a[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a[hole] = v;
}
__device__ __forceinline__ void heap_pop(HeapEnt* a, int64_t& size) {
  if (size > 1) {
    const HeapEnt v = a[size - 1];
    const int64_t n = size - 1;
    int64_t hole = 0, child = 0;
    while (child < (n - 1) / 2) {
      child = 2 * (child + 1);
      if (a[child].wpos_end > a[child - 1].wpos_end) child--;
      a[hole] = a[child];
      hole = child;
    }
    if ((n & 1) == 0 && child == (n - 2) / 2) {
      child = 2 * (child + 1);
      a[hole] = a[child - 1];
      hole = child - 1;
    }
    heap_sift_up(a, hole, 0, v);
  }
  --size;
}

struct Slide {
  const wfm_minmer_t* q;  // query sketch, slot i (1..S) = q[i-1]
  Slot* st;               // S+1 slots, slot 0 = sentinel
  int S, pivot, piv_rank, shared, strand_votes, isect;
  __device__ __forceinline__ uint64_t hash(int i) const { return i == 0 ? 0ull : q[i - 1].hash; }
  __device__ void init() {
    st[0] = Slot{0, 0, 0, 0};
    for (int i = 1; i <= S; ++i) st[i] = Slot{1, 0, 0, (int8_t)q[i - 1].strand};
    pivot = S; piv_rank = S; shared = 0; strand_votes = 0; isect = 0;
  }
  __device__ __forceinline__ int loc(uint64_t h) const {
    int lo = 1, hi = S + 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (q[mid - 1].hash < h) lo = mid + 1; else hi = mid; }
    return lo;
  }
  __device__ void insert(uint64_t h, int strand) {
    const int i = loc(h);
    if (i == S + 1) return;
    const uint64_t hi_ = hash(i), hp = hash(pivot);
    Slot s = st[i];
    if (hi_ == h) {
      s.active = 1;
      s.vote = (int16_t)(s.vote + s.qstrand * strand);
      st[i] = s;
      isect++;
      if (hi_ <= hp) { shared++; strand_votes += s.vote; }
    } else {
      s.nbi++;
      st[i] = s;
      if (hi_ <= hp) piv_rank++;
      if (piv_rank > S) {
        const Slot p = st[pivot];
        shared -= p.active;
        strand_votes -= p.vote;
        piv_rank -= (int)p.nbi;
        pivot--;
      }
    }
  }
  __device__ void erase(uint64_t h) {
    const int i = loc(h);
    if (i == S + 1) return;
    const uint64_t hi_ = hash(i), hp = hash(pivot);
    Slot s = st[i];
    if (hi_ == h) {
      if (hi_ <= hp) { shared--; strand_votes -= s.vote; }
      s.active = 0; s.vote = 0;
      st[i] = s;
      isect--;
    } else {
      s.nbi--;
      st[i] = s;
      if (hi_ <= hp) piv_rank--;
      if (pivot + 1 != S + 1) {
        const Slot nx = st[pivot + 1];
        if (piv_rank + (int)nx.nbi <= S) {
          pivot++;
          shared += nx.active;
          strand_votes += nx.vote;
          piv_rank += (int)nx.nbi;
        }
      }
    }
  }
};

// one lane per candidate: computeL2MappedRegions (window length 0)
__global__ void l2_slide_kernel(const wfm_l1_candidate_t* cand, int64_t ncand, const wfm_minmer_t* qsketch, const int32_t* qcount, int s,
                                const wfm_minmer_t* mi, int64_t n_mi, const int64_t* lo_in, const uint64_t* heap_off, const uint64_t* loc_off,
                                Slot* slots, HeapEnt* heaps, Locus* loci, const uint8_t* keep, DevParams P, uint32_t* n_loci, uint32_t* n_keep) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncand) return;
  const int64_t lo = lo_in[c];
  if (lo < 0) { n_loci[c] = 0; n_keep[c] = 0; return; }
  const wfm_l1_candidate_t L = cand[c];
  Slide sm;
  sm.q = qsketch + (int64_t)L.frag * s;
  sm.S = qcount[L.frag];
  sm.st = slots + c * (int64_t)(P.sketch_size + 1);
  sm.init();
  HeapEnt* heap = heaps + heap_off[c];
  int64_t hsize = 0;
  Locus* out = loci + loc_off[c];
  uint32_t nout = 0;
  int best_sketch = 1;
  bool in_cand = false;
  Locus l2{0, 0, 0, 0, 0};
  auto finish = [&](int votes) {
    l2.mean = (l2.start + l2.end) / 2;
    l2.strand = votes >= 0 ? 1 : -1;
    if (nout == 0 || out[nout - 1].end + P.w < l2.start) out[nout++] = l2;
    else { out[nout - 1].end = l2.end; out[nout - 1].mean = (out[nout - 1].start + out[nout - 1].end) / 2; }
  };
  int64_t it = lo;
  while (it != n_mi) {
    const wfm_minmer_t m = mi[it];
    if (m.seqId != L.seqId || !(m.wpos < L.rangeStartPos)) break;
    if (m.wpos_end > L.rangeStartPos) {
      heap[hsize] = HeapEnt{m.wpos_end, it};
      heap_sift_up(heap, hsize, 0, heap[hsize]);
      ++hsize;
      sm.insert(m.hash, m.strand);
    }
    ++it;
  }
  while (it != n_mi) {
    const wfm_minmer_t m = mi[it];
    if (m.seqId != L.seqId || !(m.wpos <= L.rangeEndPos)) break;
    const int prev_votes = sm.strand_votes;
    while (hsize > 0 && heap[0].wpos_end <= m.wpos) {
      sm.erase(mi[heap[0].idx].hash);
      heap_pop(heap, hsize);
    }
    sm.insert(m.hash, m.strand);
    heap[hsize] = HeapEnt{m.wpos_end, it};
    heap_sift_up(heap, hsize, 0, heap[hsize]);
    ++hsize;
    if (sm.shared > best_sketch) {
      nout = 0;
      in_cand = true;
      best_sketch = sm.shared;
      l2.shared = sm.shared;
      l2.start = m.wpos;
      l2.end = m.wpos;
    } else if (sm.shared == best_sketch) {
      if (!in_cand) { l2.shared = sm.shared; l2.start = m.wpos; }
      in_cand = true;
      l2.end = m.wpos;
    } else {
      if (in_cand) { finish(prev_votes); l2 = Locus{0, 0, 0, 0, 0}; }
      in_cand = false;
    }
    ++it;
  }
  if (in_cand) finish(sm.strand_votes);
  uint32_t nk = 0;
  for (uint32_t i = 0; i < nout; ++i) nk += keep[sm.S * (P.sketch_size + 1) + out[i].shared];
  n_loci[c] = nout;
  n_keep[c] = nk;
}

__global__ void l2_emit_kernel(const wfm_l1_candidate_t* cand, int64_t ncand, const int32_t* qcount, const int32_t* q_len, const uint8_t* q_kc,
                               const uint64_t* loc_off, const Locus* loci, const uint32_t* n_loci, const uint64_t* out_off,
                               const uint8_t* keep, const uint16_t* ident, DevParams P, wfm_mapping_t* out, int32_t* out_frag) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncand) return;
  const wfm_l1_candidate_t L = cand[c];
  const int qs = qcount[L.frag];
  const Locus* in = loci + loc_off[c];
  uint64_t o = out_off[c];
  for (uint32_t i = 0; i < n_loci[c]; ++i) {
    const int t = qs * (P.sketch_size + 1) + in[i].shared;
    if (!keep[t]) continue;
    wfm_mapping_t r;
    r.refSeqId = (uint32_t)L.seqId;
    r.refStartPos = (uint32_t)in[i].mean;
    r.queryStartPos = 0;
    r.blockLength = (uint32_t)q_len[L.frag];
    r.n_merged = 1;
    r.conservedSketches = (uint32_t)in[i].shared;
    r.nucIdentity = ident[t];
    r.flags = in[i].strand < 0 ? 1 : 0;
    r.kmerComplexity = q_kc[L.frag];
    out[o] = r;
    out_frag[o] = L.frag;
    ++o;
  }
}

template <typename T>
int exclusive_scan_u64(wfm_handle_t* h, Scratch& sc, const T* in, uint64_t* out, int64_t n, hipStream_t st) {
  size_t tmp = 0;
  auto tin = rocprim::make_transform_iterator(in, [] __device__(T v) { return (uint64_t)v; });
  HIPCHK(h, rocprim::exclusive_scan(nullptr, tmp, tin, out, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), st));
  char* d_tmp = nullptr;
  HIPCHK(h, sc.alloc(&d_tmp, tmp));
  HIPCHK(h, rocprim::exclusive_scan(d_tmp, tmp, tin, out, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), st));
  return WFM_OK;
}

int total_of(wfm_handle_t* h, const uint64_t* d_off, const uint32_t* d_cnt, int64_t n, hipStream_t st, uint64_t* total) {
  uint64_t lo = 0; uint32_t lc = 0;
  HIPCHK(h, hipMemcpyAsync(&lo, d_off + (n - 1), 8, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&lc, d_cnt + (n - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  *total = lo + lc;
  return WFM_OK;
}

}  // namespace

int map_l2_device(wfm_handle_t* h, MapScratch& sc, const wfm_index_t* ix, const wfm_minmer_t* d_q, const int32_t* d_qcount,
                  const int32_t* d_qlen, const uint8_t* d_kc, int64_t nfrag, int s, const wfm_l1_candidate_t* d_cand,
                  int64_t ncand, const wfm_l2_params_t* prm, wfm_mapping_t** d_out_p, int32_t** d_frag_p, int64_t* n_out_p) {
  *d_out_p = nullptr; *d_frag_p = nullptr; *n_out_p = 0;
  (void)nfrag;
  if (ncand == 0) return WFM_OK;
  hipStream_t st = wfm_stream(h);
  const int S1 = prm->sketch_size + 1;
  DevParams P{prm->window_length, prm->sketch_size, prm->stage1_topANI_filter};
#define ALLOC(p, n) do { if (sc.alloc(&(p), (size_t)(n)) != hipSuccess) { wfm_set_error(h, "out of device memory (L2)"); return WFM_E_NOMEM; } } while (0)
  uint8_t* d_keep = nullptr; uint16_t* d_ident = nullptr; double* d_cut = nullptr; int64_t* d_lo = nullptr;
  uint32_t *d_hcap = nullptr, *d_lcap = nullptr, *d_nloci = nullptr, *d_nkeep = nullptr; uint64_t *d_hoff = nullptr, *d_loff = nullptr, *d_ooff = nullptr;
  ALLOC(d_keep, S1 * S1); ALLOC(d_ident, S1 * S1); ALLOC(d_cut, S1); ALLOC(d_lo, ncand); ALLOC(d_hcap, ncand); ALLOC(d_lcap, ncand);
  ALLOC(d_nloci, ncand); ALLOC(d_nkeep, ncand); ALLOC(d_hoff, ncand); ALLOC(d_loff, ncand); ALLOC(d_ooff, ncand);
  HIPCHK(h, hipMemcpyAsync(d_keep, prm->keep_table, (size_t)S1 * S1, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_ident, prm->ident_table, (size_t)S1 * S1 * 2, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_cut, prm->cutoff_j, (size_t)S1 * 8, hipMemcpyHostToDevice, st));
  const wfm_minmer_t* mi = wfm_index_minmers(ix);
  const int64_t n_mi = wfm_index_n_kept(ix);
  const dim3 g((unsigned)((ncand + 63) / 64)), b(64);
  hipLaunchKernelGGL(l2_range_kernel, g, b, 0, st, d_cand, ncand, d_qcount, d_cut, mi, n_mi, P, d_lo, d_hcap, d_lcap);
  int rc = exclusive_scan_u64<uint32_t>(h, sc, d_hcap, d_hoff, ncand, st);
  if (rc != WFM_OK) return rc;
  rc = exclusive_scan_u64<uint32_t>(h, sc, d_lcap, d_loff, ncand, st);
  if (rc != WFM_OK) return rc;
  uint64_t heap_total = 0, loc_total = 0;
  if ((rc = total_of(h, d_hoff, d_hcap, ncand, st, &heap_total)) != WFM_OK) return rc;
  if ((rc = total_of(h, d_loff, d_lcap, ncand, st, &loc_total)) != WFM_OK) return rc;
  Slot* d_slots = nullptr; HeapEnt* d_heaps = nullptr; Locus* d_loci = nullptr;
  ALLOC(d_slots, (size_t)ncand * S1); ALLOC(d_heaps, heap_total); ALLOC(d_loci, loc_total);
  hipLaunchKernelGGL(l2_slide_kernel, g, b, 0, st, d_cand, ncand, d_q, d_qcount, s, mi, n_mi, d_lo, d_hoff, d_loff, d_slots, d_heaps, d_loci,
                     d_keep, P, d_nloci, d_nkeep);
  HIPCHK(h, hipGetLastError());
  rc = exclusive_scan_u64<uint32_t>(h, sc, d_nkeep, d_ooff, ncand, st);
  if (rc != WFM_OK) return rc;
  uint64_t n_out = 0;
  if ((rc = total_of(h, d_ooff, d_nkeep, ncand, st, &n_out)) != WFM_OK) return rc;
  if (n_out > 0) {
    wfm_mapping_t* d_out = nullptr; int32_t* d_ofrag = nullptr;
    ALLOC(d_out, n_out); ALLOC(d_ofrag, n_out);
    hipLaunchKernelGGL(l2_emit_kernel, g, b, 0, st, d_cand, ncand, d_qcount, d_qlen, d_kc, d_loff, d_loci, d_nloci, d_ooff, d_keep, d_ident, P,
                       d_out, d_ofrag);
    HIPCHK(h, hipGetLastError());
    *d_out_p = d_out; *d_frag_p = d_ofrag;
  }
#undef ALLOC
  *n_out_p = (int64_t)n_out;
  return WFM_OK;
}

extern "C" int64_t wfm_map_l2(wfm_handle_t* h, const wfm_index_t* ix, const wfm_minmer_t* qsketch, const int32_t* qcount, const int32_t* q_len,
                              const uint8_t* q_kmer_complexity, int64_t nfrag, int s, const wfm_l1_candidate_t* cands, int64_t ncand,
                              const wfm_l2_params_t* prm, wfm_mapping_t* out, int32_t* out_frag, int64_t cap) {
  if (!h || !ix || !prm || nfrag < 0 || ncand < 0 || s < 1 || (nfrag && (!qsketch || !qcount || !q_len || !q_kmer_complexity)) || (ncand && !cands))
    return WFM_E_ARG;
  if (!prm->keep_table || !prm->ident_table || !prm->cutoff_j || prm->sketch_size < 1 || s > prm->sketch_size) return WFM_E_ARG;
  for (int64_t c = 0; c < ncand; ++c) {
    if (cands[c].frag < 0 || cands[c].frag >= nfrag) return WFM_E_ARG;
    if (q_len[cands[c].frag] != prm->window_length) { wfm_set_error(h, "wfm_map_l2: fragments must be window_length long"); return WFM_E_UNSUPPORTED; }
  }
  if (ncand == 0) return 0;
  HIPCHK(h, hipSetDevice(wfm_device(h)));
  hipStream_t st = wfm_stream(h);
  MapScratch sc;
  wfm_minmer_t* d_q = nullptr; int32_t *d_qcount = nullptr, *d_qlen = nullptr; uint8_t* d_kc = nullptr; wfm_l1_candidate_t* d_cand = nullptr;
  if (sc.alloc(&d_q, (size_t)nfrag * s) != hipSuccess || sc.alloc(&d_qcount, nfrag) != hipSuccess || sc.alloc(&d_qlen, nfrag) != hipSuccess ||
      sc.alloc(&d_kc, nfrag) != hipSuccess || sc.alloc(&d_cand, ncand) != hipSuccess) { wfm_set_error(h, "out of device memory (L2)"); return WFM_E_NOMEM; }
  HIPCHK(h, hipMemcpyAsync(d_q, qsketch, (size_t)nfrag * s * sizeof(wfm_minmer_t), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_qcount, qcount, (size_t)nfrag * 4, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_qlen, q_len, (size_t)nfrag * 4, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_kc, q_kmer_complexity, (size_t)nfrag, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_cand, cands, (size_t)ncand * sizeof(wfm_l1_candidate_t), hipMemcpyHostToDevice, st));
  wfm_mapping_t* d_out = nullptr; int32_t* d_ofrag = nullptr; int64_t n_out = 0;
  const int rc = map_l2_device(h, sc, ix, d_q, d_qcount, d_qlen, d_kc, nfrag, s, d_cand, ncand, prm, &d_out, &d_ofrag, &n_out);
  if (rc != WFM_OK) return rc;
  if (n_out > 0 && out && out_frag && cap > 0) {
    const size_t n_copy = (size_t)std::min<int64_t>(n_out, cap);
    HIPCHK(h, hipMemcpyAsync(out, d_out, n_copy * sizeof(wfm_mapping_t), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(out_frag, d_ofrag, n_copy * 4, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(h, hipStreamSynchronize(st));
  return n_out;
}
