// map_winnow.hip -- addMinmers' winnowing on the device (commonFunc.hpp:440-708).
//
// The thinned k-mer stream of a sequence (map_prefilter.hip) is cut into speculative chunks as host/minmers.cpp cuts
// it -- a chunk starts from an empty state two windows before its first k-mer -- and ONE WAVE winnows one chunk:
// the sketch (sorted, <= s hashes with their open interval, strand tally and occurrence list), the pool (window k-mers
// outside the sketch) and the list nodes live in the wave's LDS; searching the sketch, the pool's minimum and the
// shifts of an insertion are done by the 64 lanes side by side, the control flow (map_winnow_core.h) is the reference's
// and uniform across the wave.  The stream is read through two 64-entry windows in registers (coalesced refills).
// After the chunks: the live state every chunk started from is compared with the one its predecessor reached
// (winnow_check_kernel), intervals that were open across a boundary get their true start (winnow_resolve_*), and the
// records are gathered in emission order.  Whatever does not fit the device's fixed capacities, and every failed
// speculation, is reported to the caller, which then winnows that sequence on the host (host/minmers.cpp).
#include <hip/hip_runtime.h>
#include "dev_cache.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "map_device.h"
#include "map_winnow_core.h"

namespace {

#define HIPCHK(h, call)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      wfm_set_error((h), std::string(#call) + ": " + hipGetErrorString(e_));            \
      return e_ == hipErrorOutOfMemory ? WFM_E_NOMEM : WFM_E_HIP;                       \
    }                                                                                   \
  } while (0)

using wn::Chunk;
using wn::Params;
using wn::Rec;
using wn::SBIT;
using wn::UNK;

// ---------------------------------------------------------------------------------------------------------------------
// wave-wide primitives (one workgroup = one wave of 64 lanes)
// ---------------------------------------------------------------------------------------------------------------------
struct SkA { uint64_t h; uint32_t wpos; uint32_t tc; };   // hash, interval start, tally (16 bits) | occurrences << 16
struct SkB { uint32_t first; uint32_t ht; };             // first occurrence, list head | tail << 16 (nodes of the further ones)
constexpr uint32_t NIL16 = 0xFFFFu;

struct DevOps {
  // LDS
  SkA* ska; SkB* skb;
  uint64_t* pl_h; uint32_t* pl_r;
  uint32_t* nd_r; uint16_t* nd_n;
  // per lane: the two stream windows
  uint32_t cw_lo[2], cw_hi[2], cw_p[2]; int cw_s[2];
  // uniform
  int n, S, P, N;
  uint64_t skmax;  // hash of the sketch's last entry (n > 0)
  int ph, pe;  // the pool's entries: [ph, pe), ascending (hash, index); dead ones (k-mer left the window) in between
  uint32_t free_head;
  int64_t ch_base[2];
  int lane;
  const Params* prm;
  Rec* recs;
  uint32_t nrec, rec_cap, chunk_id, flags;

  __device__ __forceinline__ void flag(uint32_t f) { flags |= f; }
  __device__ __forceinline__ void put(uint32_t* p, int i, uint32_t v) { if (lane == 0) p[i] = v; }

  // ---- the stream, through two 64-entry windows held in registers, one entry per lane (0: around the oldest k-mer of the
  //      window, 1: around the next to arrive); an entry is read with v_readlane, a window is refilled by coalesced loads ----
  __device__ __forceinline__ void load_block(int which, uint32_t i) {
    const int64_t base = (int64_t)(i & ~63u);
    const int64_t j = base + lane;
    uint64_t hh = 0; uint32_t pp = 0; int ss = 0;
    if (j < prm->m) { hh = prm->hash[j]; pp = prm->pos[j]; ss = prm->strand[j]; }
    cw_lo[which] = (uint32_t)hh; cw_hi[which] = (uint32_t)(hh >> 32); cw_p[which] = pp; cw_s[which] = ss;
    ch_base[which] = base;
  }
  __device__ __forceinline__ void need(int which, uint32_t i) { if ((int64_t)(i & ~63u) != ch_base[which]) load_block(which, i); }
  __device__ __forceinline__ static int sl(uint32_t i) { return __builtin_amdgcn_readfirstlane((int)(i & 63u)); }
  __device__ __forceinline__ uint64_t fr_hash(uint32_t i) { need(0, i); const int l = sl(i); return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)cw_lo[0], l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)cw_hi[0], l) << 32); }
  __device__ __forceinline__ uint32_t fr_pos(uint32_t i) { need(0, i); return (uint32_t)__builtin_amdgcn_readlane((int)cw_p[0], sl(i)); }
  __device__ __forceinline__ int fr_strand(uint32_t i) { need(0, i); return __builtin_amdgcn_readlane(cw_s[0], sl(i)); }
  __device__ __forceinline__ uint64_t ar_hash(uint32_t i) { need(1, i); const int l = sl(i); return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)cw_lo[1], l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)cw_hi[1], l) << 32); }
  __device__ __forceinline__ uint32_t ar_pos(uint32_t i) { need(1, i); return (uint32_t)__builtin_amdgcn_readlane((int)cw_p[1], sl(i)); }
  __device__ __forceinline__ int ar_strand(uint32_t i) { need(1, i); return __builtin_amdgcn_readlane(cw_s[1], sl(i)); }

  // ---- sketch: entries 0 .. n-1, ascending hashes ----
  __device__ __forceinline__ int sk_n() const { return n; }
  __device__ __forceinline__ uint64_t sk_max() const { return skmax; }
  __device__ __forceinline__ uint64_t sk_hash(int r) const { return ska[r].h; }
  __device__ __forceinline__ uint32_t sk_wpos(int r) const { return ska[r].wpos; }
  __device__ __forceinline__ int sk_tally(int r) const { return (int)(int16_t)(ska[r].tc & 0xFFFFu); }
  __device__ __forceinline__ int sk_cnt(int r) const { return (int)(ska[r].tc >> 16); }
  __device__ __forceinline__ void sk_set_wpos(int r, uint32_t v) { if (lane == 0) ska[r].wpos = v; __syncthreads(); }
  __device__ __forceinline__ void sk_set_tally(int r, int v) {
    const uint32_t tc = ska[r].tc;
    __syncthreads();
    if (lane == 0) ska[r].tc = (tc & 0xFFFF0000u) | ((uint32_t)v & 0xFFFFu);
    __syncthreads();
  }
  __device__ __forceinline__ void sk_all_unknown() { for (int j = lane; j < n; j += 64) ska[j].wpos = UNK; __syncthreads(); }
  __device__ __forceinline__ int sk_find(uint64_t h) const {
    for (int base = 0; base < n; base += 64) {
      const int j = base + lane;
      const bool hit = j < n && ska[j].h == h;
      const unsigned long long b = __ballot(hit);
      if (b) return base + __ffsll((long long)b) - 1;
    }
    return -1;
  }
  __device__ __forceinline__ int sk_insert(uint64_t h, uint32_t wpos) {
    int p = 0;
    for (int base = 0; base < n; base += 64) {
      const int j = base + lane;
      p += __popcll(__ballot(j < n && ska[j].h < h));
    }
    if (n >= S) { flag(wn::F_STATE_FULL); return p < S ? p : S - 1; }  // cannot happen: the caller inserts below s <= S
    for (int hi = n; hi > p; hi -= 64) {  // entries [p, n) move up by one, the top block first
      const int lo = hi - 64 > p ? hi - 64 : p;
      const int j = lo + lane;
      const bool v = j < hi;
      SkA a{}; SkB b{};
      if (v) { a = ska[j]; b = skb[j]; }
      __syncthreads();
      if (v) { ska[j + 1] = a; skb[j + 1] = b; }
      __syncthreads();
    }
    if (lane == 0) { ska[p] = SkA{h, wpos, 0u}; skb[p] = SkB{0u, NIL16 | (NIL16 << 16)}; }
    ++n;
    __syncthreads();
    if (p == n - 1) skmax = h;
    return p;
  }
  __device__ __forceinline__ void sk_erase(int r) {
    // nodes of a list that is still there go back to the free list
    if ((ska[r].tc >> 16) > 1) {
      uint32_t node = skb[r].ht & 0xFFFFu;
      while (node != NIL16) {
        const uint32_t nx = nd_n[node];
        __syncthreads();
        if (lane == 0) nd_n[node] = (uint16_t)free_head;
        free_head = node;
        node = nx;
        __syncthreads();
      }
    }
    for (int lo = r + 1; lo < n; lo += 64) {  // entries (r, n) move down by one
      const int j = lo + lane;
      const bool v = j < n;
      SkA a{}; SkB b{};
      if (v) { a = ska[j]; b = skb[j]; }
      __syncthreads();
      if (v) { ska[j - 1] = a; skb[j - 1] = b; }
      __syncthreads();
    }
    --n;
    if (r == n && n > 0) skmax = ska[n - 1].h;
  }

  // ---- occurrence lists: the first occurrence sits in the sketch entry, further ones in list nodes ----
  __device__ __forceinline__ void occ_push(int r, uint32_t ref) {
    const uint32_t tc = ska[r].tc, cnt = tc >> 16;
    if (cnt == 0) {
      __syncthreads();
      if (lane == 0) { skb[r] = SkB{ref, NIL16 | (NIL16 << 16)}; ska[r].tc = (tc & 0xFFFFu) | (1u << 16); }
    } else {
      const uint32_t node = free_head;
      if (node == NIL16) { flag(wn::F_OCC_FULL); return; }
      const uint32_t nx = nd_n[node], ht = skb[r].ht, tl = ht >> 16;
      __syncthreads();
      if (lane == 0) {
        nd_r[node] = ref; nd_n[node] = (uint16_t)NIL16;
        if (tl == NIL16) skb[r].ht = node | (node << 16);
        else { nd_n[tl] = (uint16_t)node; skb[r].ht = (ht & 0xFFFFu) | (node << 16); }
        ska[r].tc = (tc & 0xFFFFu) | ((cnt + 1) << 16);
      }
      free_head = nx;
    }
    __syncthreads();
  }
  __device__ __forceinline__ uint32_t occ_pop(int r) {
    const uint32_t tc = ska[r].tc, cnt = tc >> 16;
    const SkB b = skb[r];
    uint32_t node = NIL16, nref = 0, nnext = NIL16;
    if (cnt > 1) { node = b.ht & 0xFFFFu; nref = nd_r[node]; nnext = nd_n[node]; }
    __syncthreads();
    if (lane == 0) {
      if (cnt > 1) {
        skb[r] = SkB{nref, nnext == NIL16 ? (NIL16 | (NIL16 << 16)) : (nnext | (b.ht & 0xFFFF0000u))};
        nd_n[node] = (uint16_t)free_head;
      }
      ska[r].tc = (tc & 0xFFFFu) | ((cnt - 1) << 16);
    }
    if (cnt > 1) free_head = node;
    __syncthreads();
    return b.first;
  }
  __device__ __forceinline__ void occ_copy(int r, uint32_t* st, int at) {
    const uint32_t cnt = ska[r].tc >> 16;
    if (cnt == 0) return;
    put(st, at, skb[r].first);
    uint32_t node = skb[r].ht & 0xFFFFu;
    for (uint32_t q = 1; q < cnt && node != NIL16; ++q) { put(st, at + (int)q, nd_r[node]); node = nd_n[node]; }
  }

  // ---- pool ----
  // first live entry at or after position q (-1: none)
  __device__ __forceinline__ int pool_scan(int q, uint32_t a_lo) const {
    for (int base = q; base < pe; base += 64) {
      const int j = base + lane;
      const unsigned long long b = __ballot(j < pe && wn::ref_idx(pl_r[j]) >= a_lo);
      if (b) return base + __ffsll((long long)b) - 1;
    }
    return -1;
  }
  __device__ __forceinline__ void pool_compact(uint32_t a_lo) {  // the live entries move to the front, order kept
    int out = 0;
    for (int base = ph; base < pe; base += 64) {
      const int j = base + lane;
      const bool v = j < pe;
      const uint32_t ref = v ? pl_r[j] : 0u;
      const uint64_t h = v ? pl_h[j] : 0ull;
      const bool live = v && wn::ref_idx(ref) >= a_lo;
      const unsigned long long mask = __ballot(live);
      const int pre = __popcll(mask & ((1ull << lane) - 1ull));
      __syncthreads();
      if (live) { pl_r[out + pre] = ref; pl_h[out + pre] = h; }
      out += __popcll(mask);
      __syncthreads();
    }
    ph = 0; pe = out;
  }
  __device__ __forceinline__ void pool_push(uint64_t h, uint32_t ref, uint32_t a_lo) {
    if (pe >= P) pool_compact(a_lo);
    if (pe >= P) { flag(wn::F_POOL_FULL); return; }
    const uint32_t idx = wn::ref_idx(ref);
    int below = 0;  // entries that stay in front of the new one
    for (int base = ph; base < pe; base += 64) {
      const int j = base + lane;
      bool lt = false;
      if (j < pe) { const uint64_t hj = pl_h[j]; lt = hj < h || (hj == h && wn::ref_idx(pl_r[j]) < idx); }
      below += __popcll(__ballot(lt));
    }
    const int p = ph + below;
    if (ph > 0 && below <= pe - p) {  // the front part is the shorter one: it moves down by one (the slot before it is free)
      for (int lo = ph; lo < p; lo += 64) {
        const int j = lo + lane;
        const bool v = j < p;
        uint64_t a = 0; uint32_t b = 0;
        if (v) { a = pl_h[j]; b = pl_r[j]; }
        __syncthreads();
        if (v) { pl_h[j - 1] = a; pl_r[j - 1] = b; }
        __syncthreads();
      }
      --ph;
      if (lane == 0) { pl_h[p - 1] = h; pl_r[p - 1] = ref; }
      __syncthreads();
      return;
    }
    for (int hi = pe; hi > p; hi -= 64) {  // entries [p, pe) move up by one, the top block first
      const int lo = hi - 64 > p ? hi - 64 : p;
      const int j = lo + lane;
      const bool v = j < hi;
      uint64_t a = 0; uint32_t b = 0;
      if (v) { a = pl_h[j]; b = pl_r[j]; }
      __syncthreads();
      if (v) { pl_h[j + 1] = a; pl_r[j + 1] = b; }
      __syncthreads();
    }
    if (lane == 0) { pl_h[p] = h; pl_r[p] = ref; }
    ++pe;
    __syncthreads();
  }
  // the smallest hash among the live entries; the dead ones in front of it are dropped
  __device__ __forceinline__ bool pool_min(uint32_t a_lo, uint64_t* mh) {
    const int q = pool_scan(ph, a_lo);
    if (q < 0) { ph = 0; pe = 0; return false; }
    ph = q;
    *mh = pl_h[q];
    if (pe - ph >= 128 && pe >= P / 2) pool_compact(a_lo);  // keeps the scans and shifts short
    return true;
  }
  // takes out the live entry of hash h with the smallest index: it is the first live one, or there is none
  __device__ __forceinline__ bool pool_pop_hash(uint64_t h, uint32_t a_lo, uint32_t* ref) {
    const int q = pool_scan(ph, a_lo);
    if (q < 0) { ph = 0; pe = 0; return false; }
    ph = q;
    if (pl_h[q] != h) return false;
    *ref = pl_r[q];
    ph = q + 1;
    return true;
  }
  __device__ __forceinline__ int pool_first(uint32_t a_lo) const { return pool_scan(ph, a_lo); }
  __device__ __forceinline__ int pool_next(int q, uint32_t a_lo) const { return pool_scan(q + 1, a_lo); }
  __device__ __forceinline__ uint64_t pool_hash(int q) const { return pl_h[q]; }
  __device__ __forceinline__ uint32_t pool_ref(int q) const { return pl_r[q]; }

  __device__ __forceinline__ void emit(uint64_t h, uint32_t wpos, uint32_t wend, int tally) {
    if (nrec < rec_cap) {
      if (lane == 0) { Rec r; r.hash = h; r.wpos = wpos; r.wend = wend; r.tally = tally; r.chunk = chunk_id; recs[nrec] = r; }
    } else {
      flag(wn::F_REC_FULL);
    }
    ++nrec;
  }
};

__host__ __device__ inline size_t winnow_lds_bytes(int S, int P, int N) {
  return (size_t)S * 16 + (size_t)S * 8 + (size_t)P * 8 + (size_t)P * 4 + (size_t)N * 4 + (size_t)N * 2 + 64;
}

template <bool REPLAY>
__global__ void __launch_bounds__(64) winnow_chunks_kernel(Params prm, const Chunk* chunks, const int* todo, int S, int P, int N, Rec* recs, uint32_t* rec_count,
                                                           uint32_t* st_begin, uint32_t* st_end, uint32_t* wp_end, uint32_t* flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int cj = todo ? todo[blockIdx.x] : (int)blockIdx.x;
  const Chunk ch = chunks[cj];
  DevOps o;
  unsigned char* p = lds;
  o.ska = (SkA*)p; p += (size_t)S * 16;
  o.skb = (SkB*)p; p += (size_t)S * 8;
  o.pl_h = (uint64_t*)p; p += (size_t)P * 8;
  o.pl_r = (uint32_t*)p; p += (size_t)P * 4;
  o.nd_r = (uint32_t*)p; p += (size_t)N * 4;
  o.nd_n = (uint16_t*)p; p += (size_t)N * 2;
  o.n = 0; o.ph = 0; o.pe = 0; o.S = S; o.P = P; o.N = N;
  o.lane = (int)threadIdx.x;
  for (int j = o.lane; j < N; j += 64) o.nd_n[j] = (uint16_t)(j + 1 < N ? j + 1 : (int)NIL16);
  o.free_head = N > 0 ? 0u : NIL16;
  o.ch_base[0] = -1; o.ch_base[1] = -1;
  o.prm = &prm;
  o.recs = recs + ch.rec_off;
  o.nrec = 0; o.rec_cap = (uint32_t)ch.rec_cap; o.chunk_id = (uint32_t)cj; o.flags = 0;
  __syncthreads();
  wn::Stream<DevOps> S_(o, prm, (uint32_t)ch.c0);
  const size_t cap = (size_t)prm.state_words;
  if (REPLAY) S_.run_replay(ch, st_end + (size_t)(cj - 1) * cap, wp_end + (size_t)(cj - 1) * prm.s, st_begin + (size_t)cj * cap, st_end + (size_t)cj * cap,
                            wp_end + (size_t)cj * prm.s, prm.state_words);
  else S_.run(ch, st_begin + (size_t)cj * cap, nullptr, st_end + (size_t)cj * cap, wp_end + (size_t)cj * prm.s, prm.state_words);
  if (o.lane == 0) {
    rec_count[cj] = o.nrec < o.rec_cap ? o.nrec : o.rec_cap;
    flags[cj] = o.flags;
  }
}

// does chunk j start from the state chunk j-1 reached?  (which: the boundaries to look at, or all of them)
__global__ void winnow_check_kernel(const uint32_t* st_begin, const uint32_t* st_end, int cap, int nchunks, const int* which, uint32_t* flags) {
  const int j = which ? which[blockIdx.x] : (int)blockIdx.x + 1;
  if (j < 1 || j >= nchunks) return;
  const uint32_t* a = st_begin + (size_t)j * cap;
  const uint32_t* b = st_end + (size_t)(j - 1) * cap;
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  const uint32_t na = a[0], nb = b[0];
  if (na != nb || na > (uint32_t)cap || na < 4) { if (threadIdx.x == 0) bad = 1; }
  else for (uint32_t i = threadIdx.x; i < na; i += blockDim.x) if (a[i] != b[i]) bad = 1;
  __syncthreads();
  if (threadIdx.x == 0) flags[j] = (flags[j] & ~(uint32_t)wn::F_MISMATCH) | (bad ? (uint32_t)wn::F_MISMATCH : 0u);
}

// sketch hashes of a snapshot: entry r starts after the entries before it (3 words + their occurrences)
__device__ inline int64_t snap_find(const uint32_t* st, uint64_t h) {
  const int n = (int)st[2];
  int at = 4;
  for (int r = 0; r < n; ++r) {
    const uint64_t hh = (uint64_t)st[at] | ((uint64_t)st[at + 1] << 32);
    if (hh == h) return r;
    at += 3 + (int)(st[at + 2] >> 16);
  }
  return -1;
}
__device__ inline uint64_t snap_hash(const uint32_t* st, int r) {
  int at = 4;
  for (int q = 0; q < r; ++q) at += 3 + (int)(st[at + 2] >> 16);
  return (uint64_t)st[at] | ((uint64_t)st[at + 1] << 32);
}

// open intervals of the end states that began before their chunk: their start is in the previous chunk's end state
// (which may itself still be waiting for ITS predecessor: the host repeats the launch while *pending is set)
__global__ void winnow_resolve_states_kernel(const uint32_t* st_end, uint32_t* wp_end, int cap, int s, int nchunks, uint32_t* pending, uint32_t* flags) {
  const int j = blockIdx.x + 1;
  if (j >= nchunks - 1) return;  // the last chunk has no end state (its open intervals were flushed as records)
  const uint32_t* me = st_end + (size_t)j * cap;
  const uint32_t* prev = st_end + (size_t)(j - 1) * cap;
  const int n = (int)me[2];
  for (int r = threadIdx.x; r < n; r += blockDim.x) {
    if (wp_end[(size_t)j * s + r] != UNK) continue;
    const int64_t q = snap_find(prev, snap_hash(me, r));
    if (q < 0) { atomicOr(&flags[j], (uint32_t)wn::F_MISMATCH); continue; }
    const uint32_t v = wp_end[(size_t)(j - 1) * s + q];
    if (v != UNK) wp_end[(size_t)j * s + r] = v; else atomicOr(pending, 1u);
  }
}

// records that closed an interval which was open when their chunk began
__global__ void winnow_resolve_records_kernel(Rec* recs, const Chunk* chunks, const uint32_t* rec_count, const uint32_t* st_end, const uint32_t* wp_end,
                                              int cap, int s, uint32_t* flags) {
  const int j = blockIdx.x;
  if (j == 0) return;
  const uint32_t n = rec_count[j];
  Rec* r = recs + chunks[j].rec_off;
  const uint32_t* prev = st_end + (size_t)(j - 1) * cap;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    if (r[i].wpos != UNK) continue;
    const int64_t q = snap_find(prev, r[i].hash);
    if (q < 0) { atomicOr(&flags[j], (uint32_t)wn::F_MISMATCH); continue; }
    const uint32_t v = wp_end[(size_t)(j - 1) * s + q];
    if (v == UNK) atomicOr(&flags[j], (uint32_t)wn::F_UNRESOLVED);
    r[i].wpos = v;
  }
}

// the chunks' records side by side, in emission order, as MinmerInfo records
__global__ void winnow_gather_kernel(const Rec* recs, const Chunk* chunks, const uint32_t* rec_count, const int64_t* out_off, int32_t seq_id, wfm_minmer_t* out) {
  const int j = blockIdx.x;
  const uint32_t n = rec_count[j];
  const Rec* r = recs + chunks[j].rec_off;
  wfm_minmer_t* o = out + out_off[j];
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    wfm_minmer_t m;
    m.hash = r[i].hash; m.wpos = (int64_t)r[i].wpos; m.wpos_end = (int64_t)r[i].wend; m.seqId = seq_id; m.strand = (int16_t)r[i].tally; m.pad_ = 0;
    o[i] = m;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// the same primitives on plain arrays (the CPU suite's model of the kernel)
// ---------------------------------------------------------------------------------------------------------------------
struct HostOps {
  struct Entry { uint64_t h; uint32_t wpos; int tally; std::vector<uint32_t> occ; };
  std::vector<Entry> sk;
  std::vector<std::pair<uint64_t, uint32_t>> pool;
  const Params* prm;
  std::vector<Rec>* out;
  uint32_t chunk_id = 0, flags = 0;
  int pool_cap = 0, occ_cap = 0;

  void flag(uint32_t f) { flags |= f; }
  void put(uint32_t* p, int i, uint32_t v) { p[i] = v; }
  uint64_t fr_hash(uint32_t i) const { return prm->hash[i]; }
  uint32_t fr_pos(uint32_t i) const { return prm->pos[i]; }
  int fr_strand(uint32_t i) const { return prm->strand[i]; }
  uint64_t ar_hash(uint32_t i) const { return prm->hash[i]; }
  uint32_t ar_pos(uint32_t i) const { return prm->pos[i]; }
  int ar_strand(uint32_t i) const { return prm->strand[i]; }
  int sk_n() const { return (int)sk.size(); }
  uint64_t sk_max() const { return sk.back().h; }
  uint64_t sk_hash(int r) const { return sk[(size_t)r].h; }
  uint32_t sk_wpos(int r) const { return sk[(size_t)r].wpos; }
  int sk_tally(int r) const { return sk[(size_t)r].tally; }
  int sk_cnt(int r) const { return (int)sk[(size_t)r].occ.size(); }
  void sk_set_wpos(int r, uint32_t v) { sk[(size_t)r].wpos = v; }
  void sk_set_tally(int r, int v) { sk[(size_t)r].tally = v; }
  void sk_all_unknown() { for (auto& e : sk) e.wpos = UNK; }
  int sk_find(uint64_t h) const { for (size_t r = 0; r < sk.size(); ++r) if (sk[r].h == h) return (int)r; return -1; }
  int sk_insert(uint64_t h, uint32_t wpos) {
    size_t p = 0;
    while (p < sk.size() && sk[p].h < h) ++p;
    sk.insert(sk.begin() + (long)p, Entry{h, wpos, 0, {}});
    return (int)p;
  }
  void sk_erase(int r) { sk.erase(sk.begin() + r); }
  size_t occ_total() const { size_t t = 0; for (const auto& e : sk) t += e.occ.size() > 1 ? e.occ.size() - 1 : 0; return t; }
  void occ_push(int r, uint32_t ref) {
    if (!sk[(size_t)r].occ.empty() && occ_total() >= (size_t)occ_cap) { flag(wn::F_OCC_FULL); return; }
    sk[(size_t)r].occ.push_back(ref);
  }
  uint32_t occ_pop(int r) { const uint32_t v = sk[(size_t)r].occ.front(); sk[(size_t)r].occ.erase(sk[(size_t)r].occ.begin()); return v; }
  void occ_copy(int r, uint32_t* st, int at) { for (size_t q = 0; q < sk[(size_t)r].occ.size(); ++q) st[at + (int)q] = sk[(size_t)r].occ[q]; }
  void pool_compact(uint32_t a_lo) {
    size_t o2 = 0;
    for (size_t j = 0; j < pool.size(); ++j) if (wn::ref_idx(pool[j].second) >= a_lo) pool[o2++] = pool[j];
    pool.resize(o2);
  }
  void pool_push(uint64_t h, uint32_t ref, uint32_t a_lo) {  // ascending (hash, index)
    if ((int)pool.size() >= pool_cap) pool_compact(a_lo);
    if ((int)pool.size() >= pool_cap) { flag(wn::F_POOL_FULL); return; }
    size_t p = 0;
    while (p < pool.size() && (pool[p].first < h || (pool[p].first == h && wn::ref_idx(pool[p].second) < wn::ref_idx(ref)))) ++p;
    pool.insert(pool.begin() + (long)p, std::make_pair(h, ref));
  }
  int pool_scan(int q, uint32_t a_lo) const {
    for (size_t j = (size_t)q; j < pool.size(); ++j) if (wn::ref_idx(pool[j].second) >= a_lo) return (int)j;
    return -1;
  }
  bool pool_min(uint32_t a_lo, uint64_t* mh) {
    const int q = pool_scan(0, a_lo);
    if (q < 0) { pool.clear(); return false; }
    pool.erase(pool.begin(), pool.begin() + q);
    *mh = pool[0].first;
    return true;
  }
  bool pool_pop_hash(uint64_t h, uint32_t a_lo, uint32_t* ref) {
    const int q = pool_scan(0, a_lo);
    if (q < 0) { pool.clear(); return false; }
    pool.erase(pool.begin(), pool.begin() + q);
    if (pool[0].first != h) return false;
    *ref = pool[0].second;
    pool.erase(pool.begin());
    return true;
  }
  int pool_first(uint32_t a_lo) const { return pool_scan(0, a_lo); }
  int pool_next(int q, uint32_t a_lo) const { return pool_scan(q + 1, a_lo); }
  uint64_t pool_hash(int q) const { return pool[(size_t)q].first; }
  uint32_t pool_ref(int q) const { return pool[(size_t)q].second; }
  void emit(uint64_t h, uint32_t wpos, uint32_t wend, int tally) { out->push_back(Rec{h, wpos, wend, tally, chunk_id}); }
};

int64_t snap_find_host(const uint32_t* st, uint64_t h) {
  const int n = (int)st[2];
  int at = 4;
  for (int r = 0; r < n; ++r) {
    const uint64_t hh = (uint64_t)st[at] | ((uint64_t)st[at + 1] << 32);
    if (hh == h) return r;
    at += 3 + (int)(st[at + 2] >> 16);
  }
  return -1;
}

// chunk boundaries as SeqJob::plan draws them, for chunks of about chunk_len k-mer starts
std::vector<int64_t> plan_bounds(int64_t nk, int64_t chunk_len) {
  std::vector<int64_t> b(1, 0);
  if (chunk_len > 0 && nk > chunk_len + chunk_len / 2)
    for (int64_t x = chunk_len; x + chunk_len / 2 < nk; x += chunk_len) b.push_back(x);
  b.push_back(nk);
  return b;
}

struct Caps { int S, P, N, state_words; };
// Capacities of a wave's LDS state.  The full ones hold any window (a window of w - k + 1 k-mers has no more than that
// many outside the sketch, and no more further occurrences); the small ones hold what a window of ordinary sequence
// needs -- about c x s kept k-mers -- so that many waves fit a CU.  A chunk that runs out of the small ones is run
// again with the full ones.
Caps caps_for(int k, int w, int s, bool small) {
  Caps c;
  const int W = w - k + 1;
  c.S = s;
  c.P = W + 8;
  c.N = std::min(W + 8, 65000);
  if (small) {
    c.P = std::min(c.P, (std::max(192, 8 * s) + 63) / 64 * 64);
    c.N = std::min(c.N, 64);
  }
  c.state_words = 4 + 3 * s + 4 * (W + 8);
  return c;
}

}  // namespace

// The model: the chunks one after the other on the host, with the kernel's control flow and capacities.  Returns the
// number of raw records (emission order, unknown starts resolved), or -1 when the device would hand the sequence back.
int64_t map_winnow_model(const uint32_t* pos, const uint64_t* hash, const int8_t* strand, int64_t m, int64_t len, int k, int w, int s, int32_t seq_id,
                         int64_t chunk_len, std::vector<wfm_minmer_t>* out, uint32_t* why, int force_replay, int* replays_out) {
  Params prm{};
  prm.k = k; prm.w = w; prm.s = s; prm.nk = len - k + 1; prm.m = m; prm.hash = hash; prm.pos = pos; prm.strand = strand;
  const Caps cp = caps_for(k, w, s, false);
  prm.pool_cap = cp.P; prm.occ_cap = cp.N; prm.state_words = cp.state_words;
  const std::vector<int64_t> bounds = plan_bounds(prm.nk, chunk_len);
  const size_t nc = bounds.size() - 1;
  std::vector<std::vector<Rec>> recs(nc);
  std::vector<std::vector<uint32_t>> st_begin(nc, std::vector<uint32_t>((size_t)cp.state_words, 0)), st_end(nc, std::vector<uint32_t>((size_t)cp.state_words, 0));
  std::vector<std::vector<uint32_t>> wp_end(nc, std::vector<uint32_t>((size_t)s, UNK)), wp_dummy(1, std::vector<uint32_t>((size_t)s, UNK));
  uint32_t flags = 0;
  std::vector<Chunk> chunks(nc);
  for (size_t j = 0; j < nc; ++j) {
    Chunk& ch = chunks[j];
    ch = Chunk{};
    ch.from = bounds[j]; ch.to = bounds[j + 1];
    ch.warm_from = j > 0 ? std::max<int64_t>(0, bounds[j] - 2 * (int64_t)w) : 0;
    ch.c0 = std::lower_bound(pos, pos + m, ch.warm_from, [](uint32_t p, int64_t x) { return (int64_t)p < x; }) - pos;
    ch.c1 = std::lower_bound(pos, pos + m, ch.from, [](uint32_t p, int64_t x) { return (int64_t)p < x; }) - pos;
    ch.first = j == 0; ch.last = j + 1 == nc;
    HostOps o;
    o.prm = &prm; o.out = &recs[j]; o.chunk_id = (uint32_t)j; o.pool_cap = cp.P; o.occ_cap = cp.N;
    wn::Stream<HostOps> S(o, prm, (uint32_t)ch.c0);
    S.run(ch, st_begin[j].data(), wp_dummy[0].data(), st_end[j].data(), wp_end[j].data(), cp.state_words);
    flags |= o.flags;
  }
  // failed speculations (force_replay: every second one counts as failed): the chunk once more, from its predecessor's state
  int replays = 0;
  for (size_t j = 1; j < nc; ++j) {
    const uint32_t na = st_begin[j][0];
    const bool differs = na != st_end[j - 1][0] || memcmp(st_begin[j].data(), st_end[j - 1].data(), (size_t)na * 4) != 0;
    if (!differs && !(force_replay && (j & 1))) continue;
    ++replays;
    recs[j].clear();
    std::fill(wp_end[j].begin(), wp_end[j].end(), UNK);
    HostOps o;
    o.prm = &prm; o.out = &recs[j]; o.chunk_id = (uint32_t)j; o.pool_cap = cp.P; o.occ_cap = cp.N;
    wn::Stream<HostOps> S(o, prm, (uint32_t)chunks[j].c0);
    S.run_replay(chunks[j], st_end[j - 1].data(), wp_end[j - 1].data(), st_begin[j].data(), st_end[j].data(), wp_end[j].data(), cp.state_words);
    flags |= o.flags;
  }
  if (replays_out) *replays_out = replays;
  if (!flags) {
    for (size_t j = 1; j + 1 < nc; ++j) {  // in order: the predecessor's starts are final
      const int n = (int)st_end[j][2];
      int at = 4;
      for (int r = 0; r < n; ++r) {
        const uint64_t h = (uint64_t)st_end[j][(size_t)at] | ((uint64_t)st_end[j][(size_t)at + 1] << 32);
        at += 3 + (int)(st_end[j][(size_t)at + 2] >> 16);
        if (wp_end[j][(size_t)r] != UNK) continue;
        const int64_t q = snap_find_host(st_end[j - 1].data(), h);
        if (q < 0) { flags |= wn::F_MISMATCH; continue; }
        wp_end[j][(size_t)r] = wp_end[j - 1][(size_t)q];
      }
    }
    for (size_t j = 1; j < nc; ++j)
      for (Rec& r : recs[j]) {
        if (r.wpos != UNK) continue;
        const int64_t q = snap_find_host(st_end[j - 1].data(), r.hash);
        if (q < 0) { flags |= wn::F_MISMATCH; continue; }
        r.wpos = wp_end[j - 1][(size_t)q];
        if (r.wpos == UNK) flags |= wn::F_UNRESOLVED;
      }
  }
  if (why) *why = flags;
  if (flags) return -1;
  out->clear();
  for (size_t j = 0; j < nc; ++j)
    for (const Rec& r : recs[j]) out->push_back(wfm_minmer_t{r.hash, (int64_t)r.wpos, (int64_t)r.wend, seq_id, (int16_t)r.tally, 0});
  return (int64_t)out->size();
}

void map_winnow_work_free(MapWinnowWork* wk) {
  if (!wk) return;
  for (MapWinnowWork::Buf* b : {&wk->chunks, &wk->recs, &wk->count, &wk->st_begin, &wk->st_end, &wk->wp_end, &wk->flags, &wk->off, &wk->out, &wk->todo}) {
    if (b->p) (void)wfm_dfree(b->p);
    b->p = nullptr; b->bytes = 0;
  }
}

namespace {
// (only the stream the block was used on is waited for: a device-wide wait would stall the other device thread's stream)
int grow(MapWinnowWork::Buf& b, size_t bytes, hipStream_t st) {
  if (b.bytes >= bytes && b.p) return WFM_OK;
  if (b.p) { (void)hipStreamSynchronize(st); wfm_dfree_nosync(b.p); }
  b.p = nullptr; b.bytes = 0;
  const size_t want = bytes + bytes / 4 + 256;
  if (wfm_dmalloc(&b.p, want) != hipSuccess) return WFM_E_NOMEM;
  b.bytes = want;
  return WFM_OK;
}
}  // namespace

// The kept k-mers of one sequence winnowed on the device.  WFM_OK: *d_out (in wk, valid until the next call) holds *n_out
// raw records in emission order -- interval starts resolved, still to be cut, ordered and de-duplicated
// (commonFunc.hpp:660-706).  1: not for the device (capacities, a failed speculation, ...): the caller winnows the
// sequence on the host; *why says which.
int map_winnow_sparse_device(wfm_handle_t* h, const MapSparseSeq* sp, int64_t len, int k, int w, int s, int32_t seq_id, int64_t chunk_len,
                             MapWinnowWork* wk, wfm_minmer_t** d_out, int64_t* n_out, MapWinnowInfo* info, hipStream_t stream) {
  if (!h || !sp || !wk || !d_out || !n_out) return WFM_E_ARG;
  *d_out = nullptr; *n_out = 0;
  MapWinnowInfo inf{};
  const Caps cp = caps_for(k, w, s, false), cs = caps_for(k, w, s, true);
  const size_t lds = winnow_lds_bytes(cp.S, cp.P, cp.N), lds_small = winnow_lds_bytes(cs.S, cs.P, cs.N);
  const int64_t nk = len - k + 1;
  if (lds_small > 64 * 1024 || sp->m >= ((int64_t)1 << 31) || nk >= ((int64_t)1 << 32) - 1 || s < 1) { inf.why = wn::F_STATE_FULL; if (info) *info = inf; return 1; }
  HIPCHK(h, hipSetDevice(sp->device));
  hipStream_t st = stream ? stream : wfm_stream(h);
  static const bool dbg = getenv("WFM_DEBUG") && atoi(getenv("WFM_DEBUG")) > 1;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  double t_alloc = 0, t_main = 0, t_check = 0, t_resolve = 0;
  const std::vector<int64_t> bounds = plan_bounds(nk, chunk_len);
  const int nc = (int)bounds.size() - 1;
  // kept k-mers before every boundary and warm-up start
  std::vector<int64_t> q((size_t)2 * nc + 1), r((size_t)2 * nc + 1);
  for (int c = 0; c <= nc; ++c) q[(size_t)c] = bounds[(size_t)c];
  for (int c = 0; c < nc; ++c) q[(size_t)nc + 1 + c] = c > 0 ? std::max<int64_t>(0, bounds[(size_t)c] - 2 * (int64_t)w) : 0;
  int rc = map_sparse_lower_bound(h, sp, q.data(), (int)q.size(), r.data(), st);
  if (rc != WFM_OK) return rc;
  std::vector<Chunk> chunks((size_t)nc);
  int64_t rec_total = 0;
  for (int c = 0; c < nc; ++c) {
    Chunk& ch = chunks[(size_t)c];
    ch.from = bounds[(size_t)c]; ch.to = bounds[(size_t)c + 1]; ch.warm_from = q[(size_t)nc + 1 + c];
    ch.c0 = r[(size_t)nc + 1 + c];
    ch.c1 = r[(size_t)c];
    const int64_t kept = r[(size_t)c + 1] - r[(size_t)c];
    // an iteration emits at most three records (leave, arrive, the swap).  The chunk's iterations are the arrivals of its
    // own kept k-mers and the departures of those and of the ones that were in the window when it began -- kept k-mers
    // of the two windows before it at most (a chunk inside a run of N has none of its own and still closes the
    // intervals of everything that leaves); the flush adds s
    const int64_t before = r[(size_t)c] - r[(size_t)nc + 1 + c];
    ch.rec_cap = (int32_t)std::min<int64_t>(3 * (2 * kept + before) + s + 64, INT32_MAX);
    ch.rec_off = rec_total;
    rec_total += ch.rec_cap;
    ch.first = c == 0; ch.last = c + 1 == nc; ch.pad_ = 0;
  }
  const size_t cap = (size_t)cp.state_words;
  if (grow(wk->chunks, (size_t)nc * sizeof(Chunk), st) || grow(wk->recs, (size_t)rec_total * sizeof(Rec), st) || grow(wk->count, (size_t)nc * 4, st) ||
      grow(wk->st_begin, (size_t)nc * cap * 4, st) || grow(wk->st_end, (size_t)nc * cap * 4, st) || grow(wk->wp_end, (size_t)nc * (size_t)s * 4, st) ||
      grow(wk->flags, (size_t)nc * 4 + 64, st) || grow(wk->off, (size_t)nc * 8, st)) {
    wfm_set_error(h, "out of device memory (winnowing)");
    return WFM_E_NOMEM;
  }
  t_alloc = now();
  Params prm{};
  prm.k = k; prm.w = w; prm.s = s; prm.nk = nk; prm.m = sp->m; prm.hash = sp->d_hash; prm.pos = sp->d_pos; prm.strand = sp->d_strand;
  prm.pool_cap = cp.P; prm.occ_cap = cp.N; prm.state_words = cp.state_words;
  Chunk* d_chunks = (Chunk*)wk->chunks.p;
  Rec* d_recs = (Rec*)wk->recs.p;
  uint32_t* d_count = (uint32_t*)wk->count.p;
  uint32_t* d_stb = (uint32_t*)wk->st_begin.p;
  uint32_t* d_ste = (uint32_t*)wk->st_end.p;
  uint32_t* d_wpe = (uint32_t*)wk->wp_end.p;
  uint32_t* d_flags = (uint32_t*)wk->flags.p;
  uint32_t* d_pending = d_flags + nc;
  HIPCHK(h, hipMemcpyAsync(d_chunks, chunks.data(), (size_t)nc * sizeof(Chunk), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemsetAsync(d_flags, 0, (size_t)nc * 4 + 64, st));
  HIPCHK(h, hipMemsetAsync(d_wpe, 0xFF, (size_t)nc * (size_t)s * 4, st));
  // the first words of the snapshots: a chunk that does not write one (first / last) must not compare equal by accident
  HIPCHK(h, hipMemsetAsync(d_stb, 0, (size_t)nc * cap * 4, st));
  HIPCHK(h, hipMemsetAsync(d_ste, 0, (size_t)nc * cap * 4, st));
  hipLaunchKernelGGL(winnow_chunks_kernel<false>, dim3((unsigned)nc), dim3(64), lds_small, st, prm, d_chunks, (const int*)nullptr, cs.S, cs.P, cs.N, d_recs, d_count, d_stb,
                     d_ste, d_wpe, d_flags);
  HIPCHK(h, hipGetLastError());
  std::vector<uint32_t> flags((size_t)nc), count((size_t)nc);
  if (cs.P < cp.P || cs.N < cp.N) {  // chunks that ran out of the small capacities: once more with the full ones
    HIPCHK(h, hipMemcpyAsync(flags.data(), d_flags, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    std::vector<int> todo;
    for (int c = 0; c < nc; ++c)
      if (flags[(size_t)c] & (wn::F_POOL_FULL | wn::F_OCC_FULL)) todo.push_back(c);
    inf.rerun_chunks = (int)todo.size();
    if (!todo.empty() && lds <= 64 * 1024) {
      if (grow(wk->todo, todo.size() * sizeof(int), st)) { wfm_set_error(h, "out of device memory (winnowing)"); return WFM_E_NOMEM; }
      HIPCHK(h, hipMemcpyAsync(wk->todo.p, todo.data(), todo.size() * sizeof(int), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(winnow_chunks_kernel<false>, dim3((unsigned)todo.size()), dim3(64), lds, st, prm, d_chunks, (const int*)wk->todo.p, cp.S, cp.P, cp.N, d_recs, d_count,
                         d_stb, d_ste, d_wpe, d_flags);
      HIPCHK(h, hipGetLastError());
    }
  }
  if (dbg) { (void)hipStreamSynchronize(st); }
  t_main = now();
  if (nc > 1) {
    hipLaunchKernelGGL(winnow_check_kernel, dim3((unsigned)(nc - 1)), dim3(64), 0, st, d_stb, d_ste, (int)cap, nc, (const int*)nullptr, d_flags);
    HIPCHK(h, hipMemcpyAsync(flags.data(), d_flags, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    // WFM_WINNOW_FORCE=1 (tests): every second speculation counts as failed
    static const int force = [] { const char* e = getenv("WFM_WINNOW_FORCE"); return e ? atoi(e) : 0; }();
    if (force == 1) {
      for (int c = 1; c < nc; c += 2) flags[(size_t)c] |= wn::F_MISMATCH;
      HIPCHK(h, hipMemcpyAsync(d_flags, flags.data(), (size_t)nc * 4, hipMemcpyHostToDevice, st));
    }
    // Failed speculations: the chunk runs again from the state its predecessor really reached (full capacities), then the
    // next boundary is looked at again.  Chunks whose predecessor is itself waiting for its replay wait for the next round.
    std::vector<int> todo, next;
    for (int round = 0; lds <= 64 * 1024; ++round) {
      todo.clear(); next.clear();
      for (int c = 1; c < nc; ++c)
        if ((flags[(size_t)c] & wn::F_MISMATCH) && !(flags[(size_t)c - 1] & wn::F_MISMATCH)) { todo.push_back(c); if (c + 1 < nc) next.push_back(c + 1); }
      if (todo.empty()) break;
      if (round > 2 * nc + 8) { inf.why |= wn::F_MISMATCH; break; }
      inf.replays += (int)todo.size();
      if (grow(wk->todo, (todo.size() + next.size() + 1) * sizeof(int), st)) { wfm_set_error(h, "out of device memory (winnowing)"); return WFM_E_NOMEM; }
      int* d_todo = (int*)wk->todo.p;
      HIPCHK(h, hipMemcpyAsync(d_todo, todo.data(), todo.size() * sizeof(int), hipMemcpyHostToDevice, st));
      if (!next.empty()) HIPCHK(h, hipMemcpyAsync(d_todo + todo.size(), next.data(), next.size() * sizeof(int), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(winnow_chunks_kernel<true>, dim3((unsigned)todo.size()), dim3(64), lds, st, prm, d_chunks, (const int*)d_todo, cp.S, cp.P, cp.N, d_recs, d_count, d_stb,
                         d_ste, d_wpe, d_flags);
      if (!next.empty())
        hipLaunchKernelGGL(winnow_check_kernel, dim3((unsigned)next.size()), dim3(64), 0, st, d_stb, d_ste, (int)cap, nc, (const int*)(d_todo + todo.size()), d_flags);
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipMemcpyAsync(flags.data(), d_flags, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
    }
    t_check = now();
    for (int it = 0; it < 64 && nc > 2; ++it) {
      HIPCHK(h, hipMemsetAsync(d_pending, 0, 4, st));
      hipLaunchKernelGGL(winnow_resolve_states_kernel, dim3((unsigned)(nc - 1)), dim3(64), 0, st, d_ste, d_wpe, (int)cap, s, nc, d_pending, d_flags);
      uint32_t pending = 0;
      HIPCHK(h, hipMemcpyAsync(&pending, d_pending, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      ++inf.resolve_rounds;
      if (!pending) break;
    }
    hipLaunchKernelGGL(winnow_resolve_records_kernel, dim3((unsigned)nc), dim3(256), 0, st, d_recs, d_chunks, d_count, d_ste, d_wpe, (int)cap, s, d_flags);
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(flags.data(), d_flags, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(count.data(), d_count, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  t_resolve = now();
  uint32_t why = 0;
  for (int c = 0; c < nc; ++c) { why |= flags[(size_t)c]; inf.bad_chunks += flags[(size_t)c] != 0; }
  inf.chunks = nc; inf.why = why;
  if (why) { if (info) *info = inf; return 1; }
  std::vector<int64_t> off((size_t)nc);
  int64_t total = 0;
  for (int c = 0; c < nc; ++c) { off[(size_t)c] = total; total += count[(size_t)c]; }
  if (grow(wk->out, (size_t)std::max<int64_t>(total, 1) * sizeof(wfm_minmer_t), st)) { wfm_set_error(h, "out of device memory (winnowing)"); return WFM_E_NOMEM; }
  HIPCHK(h, hipMemcpyAsync(wk->off.p, off.data(), (size_t)nc * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(winnow_gather_kernel, dim3((unsigned)nc), dim3(256), 0, st, d_recs, d_chunks, d_count, (const int64_t*)wk->off.p, seq_id, (wfm_minmer_t*)wk->out.p);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(st));
  if (dbg)
    fprintf(stderr, "[wfm] winnow: %d chunks, lower bounds + buffers %.2f ms, chunks kernel (+ reruns of %d) %.2f, check + %d replays %.2f, resolve %.2f, gather %.2f ms\n", nc,
            t_alloc - t0, inf.rerun_chunks, t_main - t_alloc, inf.replays, t_check - t_main, t_resolve - t_check, now() - t_resolve);
  *d_out = (wfm_minmer_t*)wk->out.p;
  *n_out = total;
  inf.records = total;
  if (info) *info = inf;
  return WFM_OK;
}
