// map_l1.hip -- L1 stage of the mashmap3 mapper on the GPU.
//
//   getSeedIntervalPoints      src/map/include/mappingCore.hpp:82-131
//   computeL1CandidateRegions  src/map/include/mappingCore.hpp:137-301
//   doL1Mapping (group loop)   src/map/include/computeMap.hpp:945-984
//
// The reference looks each query minmer up in a hash map and heap-merges the <= s point lists
// per fragment.  Here all fragments of a batch are processed together:
//   l1_lookup_kernel   binary search of every query hash in the index' sorted unique hashes
//   l1_gather_kernel   one workgroup per fragment copies the (group-filtered) interval points as
//                      packed 64-bit keys (seqId | pos | side) into the fragment's segment
//   rocPRIM segmented radix sort   = the k-way merge by (seqId, pos, side)
//   l1_sweep_kernel    one lane per fragment runs the two sweeps of computeL1CandidateRegions
//                      over its sorted segment (count pass, then emit pass)
// Fragments are exactly windowLength long in wfmash (computeMap.hpp:560-631), so the sweep
// window Q.len - windowLength is 0 and the hash_to_freq bookkeeping of the reference is inert;
// other lengths are rejected.
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

const uint64_t* wfm_index_uhash(const wfm_index_t* ix);
const int64_t* wfm_index_poff(const wfm_index_t* ix);
const wfm_interval_point_t* wfm_index_points(const wfm_index_t* ix);
int64_t wfm_index_n_unique(const wfm_index_t* ix);

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

constexpr int POS_BITS = 41;  // pos < 2^41, seqId < 2^22
__device__ __forceinline__ uint64_t pack_key(int32_t seq, int64_t pos, int side) {
  return ((uint64_t)(uint32_t)seq << (POS_BITS + 1)) | ((uint64_t)pos << 1) | (side > 0 ? 1ull : 0ull);  // CLOSE sorts before OPEN
}
__device__ __forceinline__ int32_t key_seq(uint64_t k) { return (int32_t)(k >> (POS_BITS + 1)); }
__device__ __forceinline__ int64_t key_pos(uint64_t k) { return (int64_t)((k >> 1) & ((1ull << POS_BITS) - 1)); }
__device__ __forceinline__ bool key_open(uint64_t k) { return (k & 1ull) != 0; }

struct DevParams {
  int w, sketch_size, min_hits_cached, cached_segment_length;
  int skip_self, skip_prefix, lower_triangular, stage1_topani, stage2_full_scan;
  int n_cutoffs, n_seq;
  double cutoff_div;  // max(1, sketchSize / ss_table_max)
};

// per (fragment, query minmer): index of the hash among the unique hashes, or -1
__global__ void l1_lookup_kernel(const wfm_minmer_t* q, const int32_t* qcount, int s, int64_t nfrag, const uint64_t* uhash, int64_t nu,
                                 const int64_t* poff, int32_t* slot_u, uint32_t* seg_cap) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nfrag * s) return;
  const int64_t f = t / s;
  const int i = (int)(t - f * s);
  int32_t u = -1;
  if (i < qcount[f]) {
    const uint64_t h = q[t].hash;
    int64_t lo = 0, hi = nu;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (uhash[mid] < h) lo = mid + 1; else hi = mid; }
    if (lo < nu && uhash[lo] == h) { u = (int32_t)lo; atomicAdd(&seg_cap[f], (uint32_t)(poff[lo + 1] - poff[lo])); }
  }
  slot_u[t] = u;
}

__global__ __launch_bounds__(256) void l1_gather_kernel(const int32_t* slot_u, const int32_t* qcount, const int32_t* q_seq, int s,
                                                        const int64_t* poff, const wfm_interval_point_t* pts, const int32_t* ref_group,
                                                        const uint64_t* seg_off, uint64_t* keys, uint32_t* seg_cnt, DevParams P) {
  const int64_t f = blockIdx.x;
  __shared__ unsigned s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int qs = q_seq[f];
  const int qg = ref_group[qs];
  uint64_t* out = keys + seg_off[f];
  for (int i = 0; i < qcount[f]; ++i) {
    const int32_t u = slot_u[f * s + i];
    if (u < 0) continue;
    const int64_t b = poff[u], e = poff[u + 1];
    for (int64_t p = b + threadIdx.x; p < e; p += blockDim.x) {
      const wfm_interval_point_t ip = pts[p];
      const int tg = ref_group[ip.seqId];
      const bool skip = (P.skip_self && qg == tg) || (P.skip_prefix && qg == tg) || (P.lower_triangular && qs <= ip.seqId);
      if (!skip) out[atomicAdd(&s_n, 1u)] = pack_key(ip.seqId, ip.pos, ip.side);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) seg_cnt[f] = s_n;
}

__global__ void l1_seg_end_kernel(const uint64_t* off, const uint32_t* cnt, uint64_t* end, int64_t n) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f < n) end[f] = off[f] + cnt[f];
}

struct Cand { int32_t seq; int64_t start, end; int32_t isect; };

// computeL1CandidateRegions for one group's sorted points (window length 0).  emit == nullptr: count only.
__device__ int l1_group(const uint64_t* k, int64_t n, int q_sketch, int minimum_hits, const DevParams& P, const int32_t* cutoffs,
                        wfm_l1_candidate_t* emit, int emitted, int32_t frag, Cand& back, bool& have_back) {
  if (n == 0) return emitted;
  int overlap = 0, best = 0;
  if (P.stage1_topani) {
    int64_t t = 0, l = 0;
    while (l != n) {
      while (t != n && ((key_seq(k[t]) == key_seq(k[l]) && key_pos(k[t]) <= key_pos(k[l])) || key_seq(k[t]) < key_seq(k[l]))) {
        if (!key_open(k[t])) overlap--;
        t++;
      }
      const int64_t cur = key_pos(k[l]);
      while (l != n && key_pos(k[l]) == cur) { if (key_open(k[l])) overlap++; l++; }
      best = max(best, overlap);
    }
    if (best < minimum_hits) return emitted;
    const int idx = (int)((double)min(best, q_sketch) / P.cutoff_div);
    minimum_hits = max(cutoffs[min(idx, P.n_cutoffs - 1)], minimum_hits);
  }
  bool in_cand = false;
  Cand c{0, 0, 0, 0};
  overlap = 0;
  int prev_overlap = 0;
  int32_t prev_seq = 0, cur_seq = key_seq(k[0]);
  int64_t prev_pos = 0, cur_pos = key_pos(k[0]);
  int64_t t = 0, l = 0;
  // flush one local optimum through the "join proximal local opts" step (mappingCore.hpp:287-300)
  auto flush = [&](const Cand& lc) {
    if (!have_back || lc.seq != back.seq || lc.start > back.end + P.w) {
      if (have_back) {
        if (emit) { wfm_l1_candidate_t o; o.seqId = back.seq; o.frag = frag; o.rangeStartPos = back.start; o.rangeEndPos = back.end; o.intersectionSize = back.isect; o.pad_ = 0; emit[emitted] = o; }
        ++emitted;
      }
      back = lc; have_back = true;
    } else {
      back.end = lc.end;
      back.isect = max(lc.isect, back.isect);
    }
  };
  while (l != n) {
    prev_overlap = overlap;
    while (t != n && ((key_seq(k[t]) == key_seq(k[l]) && key_pos(k[t]) <= key_pos(k[l])) || key_seq(k[t]) < key_seq(k[l]))) {
      if (!key_open(k[t])) overlap--;
      t++;
    }
    if (key_pos(k[l]) != cur_pos) { prev_seq = cur_seq; prev_pos = cur_pos; cur_seq = key_seq(k[l]); cur_pos = key_pos(k[l]); }
    while (l != n && key_pos(k[l]) == cur_pos) { if (key_open(k[l])) overlap++; l++; }
    if (prev_overlap >= minimum_hits) {
      if (c.seq != prev_seq && in_cand) { flush(c); c = Cand{0, 0, 0, 0}; in_cand = false; }
      if (!in_cand) { c.start = prev_pos; c.end = prev_pos; c.seq = prev_seq; c.isect = prev_overlap; in_cand = true; }
      else if (P.stage2_full_scan) { c.isect = max(c.isect, prev_overlap); c.end = prev_pos; }
      else if (c.isect < prev_overlap) { c.isect = prev_overlap; c.start = prev_pos; c.end = prev_pos; }
    } else {
      if (in_cand) { flush(c); c = Cand{0, 0, 0, 0}; }
      in_cand = false;
    }
  }
  if (in_cand) flush(c);
  return emitted;
}

// one lane per fragment: doL1Mapping's group loop + sweeps.  out == nullptr -> counts only.
__global__ void l1_sweep_kernel(const uint64_t* keys, const uint64_t* seg_off, const uint32_t* seg_cnt, const int32_t* qcount,
                                const int32_t* q_len, const uint8_t* q_active, const int32_t* ref_group, const int32_t* min_hits_by_q,
                                const int32_t* cutoffs, int64_t nfrag, DevParams P, uint32_t* out_count, const uint64_t* out_off,
                                wfm_l1_candidate_t* out) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nfrag) return;
  int emitted = 0;
  const int qs = qcount[f];
  if (qs > 0 && q_active[f]) {
    const uint64_t* k = keys + seg_off[f];
    const int64_t n = seg_cnt[f];
    const int min_hits = (q_len[f] == P.cached_segment_length) ? P.min_hits_cached : min_hits_by_q[min(qs, P.sketch_size)];
    wfm_l1_candidate_t* emit = out ? out + out_off[f] : nullptr;
    Cand back{0, 0, 0, 0};
    bool have_back = false;
    int64_t b = 0;
    while (b < n) {
      int64_t e = n;
      if (P.skip_prefix) {
        const int g = ref_group[key_seq(k[b])];
        e = b;
        while (e < n && ref_group[key_seq(k[e])] == g) ++e;
      }
      emitted = l1_group(k + b, e - b, qs, min_hits, P, cutoffs, emit, emitted, (int32_t)f, back, have_back);
      b = e;
    }
    if (have_back) {
      if (emit) { wfm_l1_candidate_t o; o.seqId = back.seq; o.frag = (int32_t)f; o.rangeStartPos = back.start; o.rangeEndPos = back.end; o.intersectionSize = back.isect; o.pad_ = 0; emit[emitted] = o; }
      ++emitted;
    }
  }
  if (!out) out_count[f] = (uint32_t)emitted;
}

// ---- the same sweeps, one wave per fragment ------------------------------------------------------------------------
// What is sequential in computeL1CandidateRegions is only the candidate bookkeeping; the overlap count after a position
// group is a difference of two prefix counts: the OPEN points up to the group's end minus the CLOSE points the trailing
// pointer has passed, which is every CLOSE up to the end of the group's first (seq, pos) run (keys are sorted by
// (seq, pos, side), CLOSE first; a position group is a run of equal pos whatever the seq, mappingCore.hpp:223-226, and the
// trailing pointer compares with the group's first key, :214-221).  A wave takes 64 keys at a time: prefix sums and
// running maxima by lane shuffles give every group's count at its last key; the lanes that end a group are the
// "elements" the reference's loop looks at one iteration later (prev_overlap, prev_pos, prev_seq), and the bookkeeping
// walks them in order from wave-uniform registers -- skipping a chunk outright when no element reaches minimum_hits and
// no candidate is open, which is nearly all of them.  The keys of the next chunk are in flight while one is worked on.
__device__ __forceinline__ int wv_incl_sum(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(v, d, 64); if (lane >= d) v += t; }
  return v;
}
__device__ __forceinline__ int wv_incl_max(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(v, d, 64); if (lane >= d) v = max(v, t); }
  return v;
}
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int64_t rl64(int64_t v, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffll), lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((uint64_t)v >> 32), lane);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

struct SweepCarry {
  int opens = 0, sends = 0;   // OPEN keys / ends of (seq, pos) runs before the chunk
  int gs_sends = 0;           // `sends` at the start of the position group the chunk begins in
  int32_t gs_seq = 0;         // seq of that group's first key
  int closes_passed = 0;      // CLOSE keys the trailing pointer has passed
  uint64_t last_key = 0;      // the key before the chunk
};

// One chunk: lane `lane` holds key k[base + lane] (`key`; `next0` = the key after the chunk's last).  Out, valid on lanes
// that end a position group: the overlap count after the group, its position, the seq of its first key.
__device__ __forceinline__ void l1_chunk(uint64_t key, uint64_t next0, int64_t base, int64_t n, int lane, SweepCarry& cy, int& ov, int64_t& pos,
                                         int32_t& seq_first, bool& gend) {
  const int64_t gi = base + lane;
  const bool valid = gi < n;
  uint64_t kp = __shfl_up(key, 1, 64), kn = __shfl_down(key, 1, 64);
  if (lane == 0) kp = cy.last_key;
  if (lane == 63) kn = next0;
  const int64_t p = key_pos(key);
  const int32_t sq = key_seq(key);
  const bool open = valid && key_open(key);
  const bool gstart = valid && (gi == 0 || key_pos(kp) != p);
  const bool last = gi == n - 1;
  gend = valid && (last || key_pos(kn) != p);
  const bool send = valid && (last || key_pos(kn) != p || key_seq(kn) != sq);
  const int O = cy.opens + wv_incl_sum(open ? 1 : 0, lane);
  const int Cc = (int)(gi + 1) - O;  // CLOSE keys among k[0 .. gi]
  const int S_excl = cy.sends + wv_incl_sum(send ? 1 : 0, lane) - (send ? 1 : 0);
  const int gsl = wv_incl_max(gstart ? lane : -1, lane);  // lane of the latest group start at or before this one
  const int s_at = __shfl(S_excl, max(gsl, 0), 64);
  const int32_t q_at = __shfl(sq, max(gsl, 0), 64);
  const int Sgs = gsl >= 0 ? s_at : cy.gs_sends;
  seq_first = gsl >= 0 ? q_at : cy.gs_seq;
  const bool first_send = send && S_excl == Sgs;  // the end of the group's first (seq, pos) run: where the trailing pointer stops
  const int passed = max(cy.closes_passed, wv_incl_max(first_send ? Cc : -1, lane));
  ov = O - passed;
  pos = p;
  cy.opens = rl(O, 63);
  cy.sends = rl(S_excl + (send ? 1 : 0), 63);
  cy.gs_sends = rl(Sgs, 63);
  cy.gs_seq = rl(seq_first, 63);
  cy.closes_passed = rl(passed, 63);
  const int64_t nv = min<int64_t>(64, n - base);
  cy.last_key = (uint64_t)rl64((int64_t)key, (int)nv - 1);
}

// computeL1CandidateRegions for one group's sorted points by one wave (all lanes return the same).
__device__ int l1_group_wave(const uint64_t* k, int64_t n, int q_sketch, int minimum_hits, const DevParams& P, const int32_t* cutoffs,
                             wfm_l1_candidate_t* emit, int emitted, int32_t frag, Cand& back, bool& have_back, int lane) {
  if (n == 0) return emitted;
  if (P.stage1_topani) {
    int best = 0;
    SweepCarry cy;
    uint64_t nxt = lane < n ? k[lane] : 0;
    for (int64_t base = 0; base < n; base += 64) {
      const uint64_t key = nxt;
      nxt = base + 64 + lane < n ? k[base + 64 + lane] : 0;
      int ov; int64_t pos; int32_t sf; bool gend;
      l1_chunk(key, (uint64_t)rl64((int64_t)nxt, 0), base, n, lane, cy, ov, pos, sf, gend);
      int m = gend ? ov : 0;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
      best = max(best, m);
    }
    if (best < minimum_hits) return emitted;
    const int idx = (int)((double)min(best, q_sketch) / P.cutoff_div);
    minimum_hits = max(cutoffs[min(idx, P.n_cutoffs - 1)], minimum_hits);
  }
  bool in_cand = false;
  Cand c{0, 0, 0, 0};
  auto flush = [&](const Cand& lc) {
    if (!have_back || lc.seq != back.seq || lc.start > back.end + P.w) {
      if (have_back) {
        if (emit && lane == 0) { wfm_l1_candidate_t o; o.seqId = back.seq; o.frag = frag; o.rangeStartPos = back.start; o.rangeEndPos = back.end; o.intersectionSize = back.isect; o.pad_ = 0; emit[emitted] = o; }
        ++emitted;
      }
      back = lc; have_back = true;
    } else {
      back.end = lc.end;
      back.isect = max(lc.isect, back.isect);
    }
  };
  SweepCarry cy;
  uint64_t nxt = lane < n ? k[lane] : 0;
  for (int64_t base = 0; base < n; base += 64) {
    const uint64_t key = nxt;
    nxt = base + 64 + lane < n ? k[base + 64 + lane] : 0;
    int ov; int64_t pos; int32_t sf; bool gend;
    l1_chunk(key, (uint64_t)rl64((int64_t)nxt, 0), base, n, lane, cy, ov, pos, sf, gend);
    // the elements of this chunk: every position group but the segment's last (the loop ends before it looks back at it)
    const bool elem = gend && base + lane != n - 1;
    unsigned long long E = __ballot(elem);
    const unsigned long long H = __ballot(elem && ov >= minimum_hits);
    if (H == 0) {
      if (in_cand && E) { flush(c); c = Cand{0, 0, 0, 0}; in_cand = false; }
      continue;
    }
    while (E) {
      if (!in_cand) {  // nothing open: straight to the next element that reaches minimum_hits
        const unsigned long long rest = H & E;
        if (!rest) break;
        E &= ~((1ull << __builtin_ctzll(rest)) - 1ull);
      }
      const int e = __builtin_ctzll(E);
      E &= E - 1ull;
      const int o = rl(ov, e);
      if (o >= minimum_hits) {
        const int64_t pp = rl64(pos, e);
        const int32_t sq = rl(sf, e);
        if (c.seq != sq && in_cand) { flush(c); c = Cand{0, 0, 0, 0}; in_cand = false; }
        if (!in_cand) { c.start = pp; c.end = pp; c.seq = sq; c.isect = o; in_cand = true; }
        else if (P.stage2_full_scan) { c.isect = max(c.isect, o); c.end = pp; }
        else if (c.isect < o) { c.isect = o; c.start = pp; c.end = pp; }
      } else {
        if (in_cand) { flush(c); c = Cand{0, 0, 0, 0}; }
        in_cand = false;
      }
    }
  }
  if (in_cand) flush(c);
  return emitted;
}

__global__ __launch_bounds__(64) void l1_sweep_wave_kernel(const uint64_t* keys, const uint64_t* seg_off, const uint32_t* seg_cnt, const int32_t* qcount,
                                                           const int32_t* q_len, const uint8_t* q_active, const int32_t* ref_group,
                                                           const int32_t* min_hits_by_q, const int32_t* cutoffs, int64_t nfrag, DevParams P,
                                                           uint32_t* out_count, const uint64_t* out_off, wfm_l1_candidate_t* out) {
  const int64_t f = blockIdx.x;
  const int lane = (int)threadIdx.x;
  if (f >= nfrag) return;
  int emitted = 0;
  const int qs = qcount[f];
  if (qs > 0 && q_active[f]) {
    const uint64_t* k = keys + seg_off[f];
    const int64_t n = seg_cnt[f];
    const int min_hits = (q_len[f] == P.cached_segment_length) ? P.min_hits_cached : min_hits_by_q[min(qs, P.sketch_size)];
    wfm_l1_candidate_t* emit = out ? out + out_off[f] : nullptr;
    Cand back{0, 0, 0, 0};
    bool have_back = false;
    int64_t b = 0;
    while (b < n) {
      int64_t e = n;
      if (P.skip_prefix) {  // the run of keys of one reference group (doL1Mapping's group loop, computeMap.hpp:945-984)
        const int g = ref_group[key_seq(k[b])];
        for (int64_t base = b; base < n; base += 64) {
          const int64_t gi = base + lane;
          const bool diff = gi < n && ref_group[key_seq(k[gi])] != g;
          const unsigned long long m = __ballot(diff);
          if (m) { e = base + __builtin_ctzll(m); break; }
        }
      }
      emitted = l1_group_wave(k + b, e - b, qs, min_hits, P, cutoffs, emit, emitted, (int32_t)f, back, have_back, lane);
      b = e;
    }
    if (have_back) {
      if (emit && lane == 0) { wfm_l1_candidate_t o; o.seqId = back.seq; o.frag = (int32_t)f; o.rangeStartPos = back.start; o.rangeEndPos = back.end; o.intersectionSize = back.isect; o.pad_ = 0; emit[emitted] = o; }
      ++emitted;
    }
  }
  if (!out && lane == 0) out_count[f] = (uint32_t)emitted;
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

}  // namespace

int map_l1_device(wfm_handle_t* h, MapScratch& sc, const wfm_index_t* ix, const wfm_minmer_t* d_q, const int32_t* d_qcount,
                  const int32_t* d_qseq, const int32_t* d_qlen, const uint8_t* d_act, int64_t nfrag, int s,
                  const wfm_l1_params_t* prm, wfm_l1_candidate_t** d_cands, int64_t* ncand) {
  *d_cands = nullptr; *ncand = 0;
  if (nfrag == 0) return WFM_OK;
  hipStream_t st = wfm_stream(h);
  DevParams P;
  P.w = prm->window_length; P.sketch_size = prm->sketch_size; P.min_hits_cached = prm->min_hits_cached;
  P.cached_segment_length = prm->cached_segment_length;
  P.skip_self = prm->skip_self; P.skip_prefix = prm->skip_prefix; P.lower_triangular = prm->lower_triangular;
  P.stage1_topani = prm->stage1_topANI_filter; P.stage2_full_scan = prm->stage2_full_scan;
  P.n_cutoffs = prm->n_cutoffs; P.n_seq = prm->n_seq;
  P.cutoff_div = std::max(1.0, (double)prm->sketch_size / 1000.0);  // fixed::ss_table_max
  int32_t *d_group = nullptr, *d_minhits = nullptr, *d_cut = nullptr, *d_slot = nullptr;
  uint32_t *d_cap = nullptr, *d_cnt = nullptr, *d_ocount = nullptr; uint64_t *d_off = nullptr, *d_ooff = nullptr;
#define ALLOC(p, n) do { if (sc.alloc(&(p), (size_t)(n)) != hipSuccess) { wfm_set_error(h, "out of device memory (L1)"); return WFM_E_NOMEM; } } while (0)
  ALLOC(d_group, prm->n_seq); ALLOC(d_minhits, prm->sketch_size + 1); ALLOC(d_cut, prm->n_cutoffs); ALLOC(d_slot, nfrag * s);
  ALLOC(d_cap, nfrag); ALLOC(d_cnt, nfrag); ALLOC(d_ocount, nfrag); ALLOC(d_off, nfrag + 1); ALLOC(d_ooff, nfrag + 1);
  HIPCHK(h, hipMemcpyAsync(d_group, prm->ref_group, (size_t)prm->n_seq * 4, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_minhits, prm->min_hits_by_qsketch, (size_t)(prm->sketch_size + 1) * 4, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_cut, prm->sketch_cutoffs, (size_t)prm->n_cutoffs * 4, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemsetAsync(d_cap, 0, (size_t)nfrag * 4, st));
  const int64_t nt = nfrag * s;
  hipLaunchKernelGGL(l1_lookup_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, d_q, d_qcount, s, nfrag,
                     wfm_index_uhash(ix), wfm_index_n_unique(ix), wfm_index_poff(ix), d_slot, d_cap);
  int rc = exclusive_scan_u64<uint32_t>(h, sc, d_cap, d_off, nfrag, st);
  if (rc != WFM_OK) return rc;
  uint64_t last_off = 0; uint32_t last_cap = 0;
  HIPCHK(h, hipMemcpyAsync(&last_off, d_off + (nfrag - 1), 8, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&last_cap, d_cap + (nfrag - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  const uint64_t total = last_off + last_cap;
  uint64_t *d_keys = nullptr, *d_keys2 = nullptr, *d_end = nullptr;
  ALLOC(d_keys, total); ALLOC(d_keys2, total); ALLOC(d_end, nfrag);
  hipLaunchKernelGGL(l1_gather_kernel, dim3((unsigned)nfrag), dim3(256), 0, st, d_slot, d_qcount, d_qseq, s, wfm_index_poff(ix),
                     wfm_index_points(ix), d_group, d_off, d_keys, d_cnt, P);
  HIPCHK(h, hipGetLastError());
  if (total > 0) {
    // segment f = [off[f], off[f] + cnt[f])
    hipLaunchKernelGGL(l1_seg_end_kernel, dim3((unsigned)((nfrag + 255) / 256)), dim3(256), 0, st, d_off, d_cnt, d_end, nfrag);
    const uint64_t* end_it = d_end;
    const uint64_t* beg_it = d_off;
    size_t tmp = 0;
    HIPCHK(h, rocprim::segmented_radix_sort_keys(nullptr, tmp, d_keys, d_keys2, (size_t)total, (unsigned)nfrag, beg_it, end_it, 0, 64, st));
    char* d_tmp = nullptr;
    ALLOC(d_tmp, tmp);
    HIPCHK(h, rocprim::segmented_radix_sort_keys(d_tmp, tmp, d_keys, d_keys2, (size_t)total, (unsigned)nfrag, beg_it, end_it, 0, 64, st));
  }
  // one wave per fragment (WFM_L1_WAVE=0: the one-lane-per-fragment form, kept for A/B runs and as a cross-check in the tests)
  const bool wave_form = !(getenv("WFM_L1_WAVE") && atoi(getenv("WFM_L1_WAVE")) == 0);
  const dim3 g(wave_form ? (unsigned)nfrag : (unsigned)((nfrag + 63) / 64)), b(64);
  auto sweep = wave_form ? l1_sweep_wave_kernel : l1_sweep_kernel;
  hipLaunchKernelGGL(sweep, g, b, 0, st, d_keys2, d_off, d_cnt, d_qcount, d_qlen, d_act, d_group, d_minhits, d_cut, nfrag, P,
                     d_ocount, (const uint64_t*)nullptr, (wfm_l1_candidate_t*)nullptr);
  rc = exclusive_scan_u64<uint32_t>(h, sc, d_ocount, d_ooff, nfrag, st);
  if (rc != WFM_OK) return rc;
  uint64_t lo2 = 0; uint32_t lc2 = 0;
  HIPCHK(h, hipMemcpyAsync(&lo2, d_ooff + (nfrag - 1), 8, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&lc2, d_ocount + (nfrag - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  const int64_t n_out = (int64_t)(lo2 + lc2);
  if (n_out > 0) {
    wfm_l1_candidate_t* d_out = nullptr;
    ALLOC(d_out, n_out);
    hipLaunchKernelGGL(sweep, g, b, 0, st, d_keys2, d_off, d_cnt, d_qcount, d_qlen, d_act, d_group, d_minhits, d_cut, nfrag, P,
                       d_ocount, d_ooff, d_out);
    HIPCHK(h, hipGetLastError());
    *d_cands = d_out;
  }
#undef ALLOC
  *ncand = n_out;
  return WFM_OK;
}

extern "C" int64_t wfm_map_l1(wfm_handle_t* h, const wfm_index_t* ix, const wfm_minmer_t* qsketch, const int32_t* qcount,
                              const int32_t* q_seq_id, const int32_t* q_len, const uint8_t* q_active, int64_t nfrag, int s,
                              const wfm_l1_params_t* prm, wfm_l1_candidate_t* out, int64_t cap) {
  if (!h || !ix || !prm || (nfrag && (!qsketch || !qcount || !q_seq_id || !q_len || !q_active)) || nfrag < 0 || s < 1) return WFM_E_ARG;
  if (!prm->ref_group || !prm->min_hits_by_qsketch || !prm->sketch_cutoffs || prm->n_cutoffs < 1) return WFM_E_ARG;
  for (int64_t f = 0; f < nfrag; ++f)
    if (q_active[f] && qcount[f] > 0 && q_len[f] != prm->window_length) { wfm_set_error(h, "wfm_map_l1: fragments must be window_length long"); return WFM_E_UNSUPPORTED; }
  if (nfrag == 0) return 0;
  HIPCHK(h, hipSetDevice(wfm_device(h)));
  hipStream_t st = wfm_stream(h);
  MapScratch sc;
  wfm_minmer_t* d_q = nullptr; int32_t *d_qcount = nullptr, *d_qseq = nullptr, *d_qlen = nullptr; uint8_t* d_act = nullptr;
  if (sc.alloc(&d_q, (size_t)nfrag * s) != hipSuccess || sc.alloc(&d_qcount, nfrag) != hipSuccess || sc.alloc(&d_qseq, nfrag) != hipSuccess ||
      sc.alloc(&d_qlen, nfrag) != hipSuccess || sc.alloc(&d_act, nfrag) != hipSuccess) { wfm_set_error(h, "out of device memory (L1)"); return WFM_E_NOMEM; }
  HIPCHK(h, hipMemcpyAsync(d_q, qsketch, (size_t)nfrag * s * sizeof(wfm_minmer_t), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_qcount, qcount, (size_t)nfrag * 4, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_qseq, q_seq_id, (size_t)nfrag * 4, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_qlen, q_len, (size_t)nfrag * 4, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_act, q_active, (size_t)nfrag, hipMemcpyHostToDevice, st));
  wfm_l1_candidate_t* d_out = nullptr; int64_t n_out = 0;
  const int rc = map_l1_device(h, sc, ix, d_q, d_qcount, d_qseq, d_qlen, d_act, nfrag, s, prm, &d_out, &n_out);
  if (rc != WFM_OK) return rc;
  if (n_out > 0 && out && cap > 0) {
    HIPCHK(h, hipMemcpyAsync(out, d_out, (size_t)std::min(n_out, cap) * sizeof(wfm_l1_candidate_t), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(h, hipStreamSynchronize(st));
  return n_out;
}
