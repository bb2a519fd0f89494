// wfa_kernels.hip -- gap-affine 2-piece wavefront alignment kernels for gfx950.
//
// Replaces the WFA2-lib calls made at wflign.cpp:136-148 (alignEnd2End,
// MemoryUltralow = BiWFA), :280-305 and :368-397 (alignEndsFree, MemoryMed).
// One alignment (sub-)problem per workgroup; wavefront offsets live in a
// per-job ring in HBM/L2 ([dir][component][32 rows][diagonal], diagonals
// contiguous -> coalesced), row ranges in LDS, wave-64 shuffles for the
// per-step reductions.  Integer DP: no MFMA.
//
//   wfa_bp_kernel    BiWFA breakpoint search (forward + reverse score-only
//                    wavefronts, 26-score scope kept in a 32-row ring)
//   wfa_base_kernel  unidirectional WFA with per-cell backtrace decisions
//                    (BiWFA leaves, ends-free patches, short sequences)
//   rle_compact_kernel  gathers the leaves' run-length CIGAR pieces
//
// Conventions: pattern -> v, text -> h, k = h - v, offset = h, NULL = -2^30.
// Every out-of-bounds cell (h>tlen or v>plen) is nulled in ALL components;
// this is result-equivalent to WFA2-lib (which nulls only M and trims ends):
// an out-of-bounds I/D offset can only feed out-of-bounds cells.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "wfa_device.h"

namespace wfm {

#ifdef WFM_PROFILE_SECTIONS
__device__ long long g_sec[8];
#define SEC_T(v) const long long v = clock64(); asm volatile("" ::: "memory")
#define SEC_ADD(i, a, b) sec[i] += (b) - (a)
#else
#define SEC_T(v)
#define SEC_ADD(i, a, b)
#endif

__device__ __forceinline__ uint64_t load8(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

// common prefix of a[n..) and b[n..) counted from 0, at most maxn: 32 bases per round while they last (the four
// loads of a round are independent: one memory round trip per 32 matching bases instead of per 8 -- at low
// divergence a run of matches is hundreds of bases long), then 8 at a time.  May return more than maxn when the
// last word read runs past it; callers cap.
__device__ __forceinline__ int lce_from(const uint8_t* a, const uint8_t* b, int n, int maxn) {
  while (n + 32 <= maxn) {
    const uint64_t x0 = load8(a + n) ^ load8(b + n), x1 = load8(a + n + 8) ^ load8(b + n + 8);
    const uint64_t x2 = load8(a + n + 16) ^ load8(b + n + 16), x3 = load8(a + n + 24) ^ load8(b + n + 24);
    if (x0 | x1 | x2 | x3) {
      if (x0) return n + (int)(__builtin_ctzll(x0) >> 3);
      if (x1) return n + 8 + (int)(__builtin_ctzll(x1) >> 3);
      if (x2) return n + 16 + (int)(__builtin_ctzll(x2) >> 3);
      return n + 24 + (int)(__builtin_ctzll(x3) >> 3);
    }
    n += 32;
  }
  while (n < maxn) {
    const uint64_t x = load8(a + n) ^ load8(b + n);
    if (x) return n + (int)(__builtin_ctzll(x) >> 3);
    n += 8;
  }
  return n;
}

// longest common extension of P[v..) and T[h..), bounded by the sub-problem ends
__device__ __forceinline__ int lce_bounded(const uint8_t* P, const uint8_t* T, int v, int h, int pl, int tl) {
  const int maxn = min(pl - v, tl - h);
  return min(lce_from(P + v, T + h, 0, maxn), maxn);
}

// ---- wave-cooperative long extensions (global-memory form; the tile kernel's LDS-window form follows further down) ----
// A run of matches is usually over within a few bases (a cell off the optimal path) -- or it is hundreds to thousands of
// bases long (the optimal path of a low-divergence record), and then one lane walks it 32 bases per round while the
// other 63 lanes of its wave, and behind the step's barrier the whole workgroup, wait.  So a lane only looks at the
// first 40 bases itself; runs that go on are finished by the whole wave, one pending lane after the other, 64 lanes x
// 8 bases = 512 bases per round trip.  Every lane of the wave must make the call (pend = false when it has nothing).
__device__ __forceinline__ int rdlane(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

__device__ __forceinline__ int wave_lce_tail_g(const uint8_t* P, const uint8_t* T, int v, int h, int n, int maxn, bool pend) {
  unsigned long long todo = __ballot(pend);
  const int lane = (int)(threadIdx.x & 63u);
  while (todo) {
    const int src = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(todo));
    todo &= todo - 1;
    const int v0 = rdlane(v, src), h0 = rdlane(h, src), mx = rdlane(maxn, src);
    int nn = rdlane(n, src);  // bases known to match so far (uniform)
    int res;
    for (;;) {
      const int off = nn + lane * 8;
      const bool past = off >= mx;
      uint64_t x = 0;
      if (!past) x = load8(P + v0 + off) ^ load8(T + h0 + off);
      const unsigned long long hit = __ballot(past || x != 0);
      if (hit) {
        const int f = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(hit));
        const unsigned xlo = (unsigned)rdlane((int)(uint32_t)x, f), xhi = (unsigned)rdlane((int)(uint32_t)(x >> 32), f);
        const uint64_t xf = ((uint64_t)xhi << 32) | xlo;
        const int at = nn + f * 8;
        res = at >= mx ? mx : min(mx, at + (xf ? (int)(__builtin_ctzll(xf) >> 3) : 0));
        break;
      }
      nn += 512;
    }
    if (lane == src) n = res;
  }
  return n;
}

// what a lane does on its own: 8 bases (x8 = their xor, already loaded), then one round of 32; more = the run goes on
__device__ __forceinline__ int lce_head40_g(const uint8_t* P, const uint8_t* T, int v, int h, uint64_t x8, int maxn, bool& more) {
  more = false;
  if (x8) return min((int)(__builtin_ctzll(x8) >> 3), maxn);
  if (maxn <= 8) return maxn;
  if (maxn < 40) return min(lce_from(P + v, T + h, 8, maxn), maxn);  // the end of the sequences is near
  const uint8_t* pp = P + (v + 8);
  const uint8_t* tt = T + (h + 8);
  const uint64_t x0 = load8(pp) ^ load8(tt), x1 = load8(pp + 8) ^ load8(tt + 8);
  const uint64_t x2 = load8(pp + 16) ^ load8(tt + 16), x3 = load8(pp + 24) ^ load8(tt + 24);
  if (x0) return 8 + (int)(__builtin_ctzll(x0) >> 3);
  if (x1) return 16 + (int)(__builtin_ctzll(x1) >> 3);
  if (x2) return 24 + (int)(__builtin_ctzll(x2) >> 3);
  if (x3) return 32 + (int)(__builtin_ctzll(x3) >> 3);
  more = maxn > 40;
  return 40;
}

// extension of one cell per lane from (v, h), at most maxn bases; every lane of the wave calls it (live = false: no cell)
__device__ __forceinline__ int wave_lce(const uint8_t* P, const uint8_t* T, int v, int h, int maxn, bool live) {
  bool more = false;
  int n = 0;
  if (live && maxn > 0) n = lce_head40_g(P, T, v, h, load8(P + v) ^ load8(T + h), maxn, more);
  if (__any(more)) n = wave_lce_tail_g(P, T, v, h, n, maxn, more);
  return n;
}

struct Src {
  const int32_t* p;  // p[k] addresses diagonal k
  int lo, hi;
};

__device__ __forceinline__ int ldk(const Src& r, int k) {
  return (k >= r.lo && k <= r.hi) ? r.p[k] : WF_NULL;
}

// Extension with at most two dependent memory round trips for runs up to 40 bases:
// 8 bytes first (ends almost every cell), then 32 bytes at once.
__device__ __forceinline__ int lce_bounded2(const uint8_t* P, const uint8_t* T, int v, int h, int pl, int tl) {
  const int maxn = min(pl - v, tl - h);
  if (maxn <= 0) return 0;
  const uint8_t* a = P + v;
  const uint8_t* b = T + h;
  uint64_t x = load8(a) ^ load8(b);
  if (x) return min((int)(__builtin_ctzll(x) >> 3), maxn);
  int n = 8;
  while (n < maxn) {
    const uint64_t x0 = load8(a + n) ^ load8(b + n), x1 = load8(a + n + 8) ^ load8(b + n + 8);
    const uint64_t x2 = load8(a + n + 16) ^ load8(b + n + 16), x3 = load8(a + n + 24) ^ load8(b + n + 24);
    if (x0) { n += (int)(__builtin_ctzll(x0) >> 3); break; }
    if (x1) { n += 8 + (int)(__builtin_ctzll(x1) >> 3); break; }
    if (x2) { n += 16 + (int)(__builtin_ctzll(x2) >> 3); break; }
    if (x3) { n += 24 + (int)(__builtin_ctzll(x3) >> 3); break; }
    n += 32;
  }
  return min(n, maxn);
}

__device__ __forceinline__ int valid_or_null(int off, int k, unsigned pl, unsigned tl) {
  const unsigned h = (unsigned)off, v = (unsigned)(off - k);
  return (h <= tl && v <= pl) ? off : WF_NULL;
}

// wave-64 max of non-negative values with DPP row shifts / broadcasts (result valid in lane 63)
__device__ __forceinline__ int wave_max_dpp63(int x) {
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));  // row_shr:1
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));  // row_shr:2
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));  // row_shr:4
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));  // row_shr:8
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));  // row_bcast:15
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));  // row_bcast:31
  return x;
}

__device__ __forceinline__ int wave_max(int x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x = max(x, __shfl_xor(x, d, 64));
  return x;
}

__device__ __forceinline__ int imin3(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(a, max(b, c)); }

// ---------------------------------------------------------------------------
// BiWFA breakpoint kernel
// ---------------------------------------------------------------------------
// Row ranges in closed form.  Score s reaches the diagonals [-s, s] (every change of diagonal costs at least e2 = 1), clipped to
// the problem -- and, when an upper bound `sub` of the problem's score is known (a BiWFA child is handed its score by its
// parent; a root may come with a hint), only the diagonals from which the end diagonal kinv = tl - pl is still within
// reach: |k - kinv| <= sub - s.  A cell outside cannot lie on an alignment of score <= sub, and no cell inside depends on one
// outside (a predecessor is one diagonal away at most and at least e2 cheaper), so the cells inside keep their exact
// values and every breakpoint of score <= sub is found where the reference finds it: the phase-1 trigger (the running
// maxima of the antidiagonals) can only fire LATER without the cells outside, never after a pair of cells of a real
// overlap exists, and phase 2 tests the rows that triggered.  For a record with 1 kb end gaps this removes half the cells.
struct Rng { int pl, tl, kb_lo, kb_hi; };  // kb_lo = kinv - sub, kb_hi = kinv + sub
__device__ __forceinline__ Rng make_rng(int pl, int tl, int sub) { Rng r; r.pl = pl; r.tl = tl; r.kb_lo = (tl - pl) - sub; r.kb_hi = (tl - pl) + sub; return r; }
__device__ __forceinline__ int rng_lo(const Rng& r, int s) { return max(max(-r.pl, -s), r.kb_lo + s); }
__device__ __forceinline__ int rng_hi(const Rng& r, int s) { return min(min(r.tl, s), r.kb_hi - s); }
// The diagonals a tile pass over the scores (s_from, s_to] has to hold: every bound at its loosest score of the block -- and
// the score bound as it stood 25 scores BEFORE the block: the edge a score bound sets moves inwards, so the rows the block
// starts from are wider than its own, and the snapshot it leaves behind must hold every row of the last 26 scores whole --
// a short last block (one that stops at the meeting point after a few steps) hands rows older than its own first row to
// phase 2, which reads each row over its full range.  (Until round 3 the bound was taken at s_from: the cells of the older
// rows beyond it never reached the output ring, and phase 2 read whatever the ring held there.  With the slack the bounds
// used to carry -- 56 for a child, 200 for a caller's guess -- those cells could not complete an overlap within the bound
// and stale values of the same job never made one up; an exact bound on a small batch, where rings are reused across
// jobs, did: a false breakpoint one point under the optimum.)
constexpr int RNG_BACK = 25;
__device__ __forceinline__ void rng_block(const Rng& r, int s_from, int s_to, int& L, int& R) {
  L = max(max(-r.pl, -s_to), r.kb_lo + s_from - RNG_BACK);
  R = min(min(r.tl, s_to), r.kb_hi - s_from + RNG_BACK);
}

__device__ __forceinline__ int bp_gap_open(const DevPen& pn, int cc) {
  return (cc == C_M) ? 0 : ((cc == C_I1 || cc == C_D1) ? pn.o1 : pn.o2);
}

__device__ __forceinline__ int sel_rng(int v, int k, int lo, int hi) { return (k >= lo && k <= hi) ? v : WF_NULL; }

// ---- the kernels for any penalties, once per ring depth (wfa_generic_inc.h) ----
namespace r32 {
#include "wfa_generic_inc.h"
}  // namespace r32
namespace r128 {
constexpr int RING = 128;  // shadows wfm::RING inside this namespace
constexpr int RMASK = RING - 1;
#include "wfa_generic_inc.h"
}  // namespace r128

// ---------------------------------------------------------------------------
// RLE compaction: one workgroup per problem; keeps slot order
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rle_compact_kernel(const uint32_t* __restrict__ rle, const int64_t* __restrict__ prob_off,
                                                          const int64_t* __restrict__ prob_cap, uint32_t* __restrict__ out,
                                                          unsigned long long* __restrict__ total, int64_t* __restrict__ out_start,
                                                          int32_t* __restrict__ out_count) {
  const int64_t base = prob_off[blockIdx.x];
  const int64_t cap = prob_cap[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  __shared__ int s_w[4];
  __shared__ long long s_start;
  __shared__ int s_cnt;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  int cnt = 0;
  for (int64_t i = tid; i < cap; i += blockDim.x) cnt += rle[base + i] != 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
  if (lane == 0 && cnt) atomicAdd(&s_cnt, cnt);
  __syncthreads();
  if (tid == 0) {
    s_start = (long long)atomicAdd(total, (unsigned long long)s_cnt);
    out_start[blockIdx.x] = s_start;
    out_count[blockIdx.x] = s_cnt;
  }
  __syncthreads();
  int64_t wbase = s_start;
  for (int64_t i0 = 0; i0 < cap; i0 += blockDim.x) {
    const int64_t i = i0 + tid;
    const uint32_t e = i < cap ? rle[base + i] : 0u;
    const unsigned long long mask = __ballot(e != 0);
    const int pre = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_w[wid] = __popcll(mask);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) { const int x = s_w[w2]; if (w2 < wid) woff += x; tot += x; }
    if (e) out[wbase + woff + pre] = e;
    wbase += tot;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Register-resident time tile (default penalty lags only): every thread owns C consecutive
// diagonals; the M history is a 25-deep register delay line, I/D keep e1 / e2 rows;
// left/right neighbours come from wave shuffles, wave edges go through a tiny LDS mailbox.
// Same contract as wfa_tile_kernel (snapshot in -> T steps -> snapshot out + per-step maxima).
// ---------------------------------------------------------------------------
// ---- sequence windows in LDS (register tile kernel) ----
// The extension is a chain of dependent loads (8 bases, compare, next 8 ...); from L2 that chain is the
// longest stall of a score step.  A tile only touches a few kilobases of either sequence during its T
// steps (offsets never decrease and fall off by ~5 bases per diagonal away from the furthest one), so a
// window of each sequence is staged in LDS once per tile; anything outside it still comes from global.
constexpr int SEQ_WIN = 8192;  // bytes per window
__device__ __forceinline__ uint64_t lds_load8(const uint32_t* win, unsigned off) {
  const uint32_t* w = win + (off >> 2);
  const unsigned sh = (off & 3u) * 8u;
  const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
  const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
  return ((uint64_t)hi << 32) | lo;
}
// 8 bases of P[v..) xor T[h..)
__device__ __forceinline__ uint64_t win_xor8(const uint8_t* P, const uint8_t* T, const uint32_t* winP, const uint32_t* winT, int v, int h,
                                             int wP0, int wT0) {
  const unsigned ov = (unsigned)(v - wP0), oh = (unsigned)(h - wT0);
  if (ov <= (unsigned)(SEQ_WIN - 8) && oh <= (unsigned)(SEQ_WIN - 8)) return lds_load8(winP, ov) ^ lds_load8(winT, oh);
  return load8(P + v) ^ load8(T + h);
}
// longest common extension from (v, h) on, at most maxn
__device__ __forceinline__ int win_lce(const uint8_t* P, const uint8_t* T, const uint32_t* winP, const uint32_t* winT, int v, int h, int maxn,
                                       int wP0, int wT0) {
  int n = 0;
  // 32 bases per round: the loads of a round do not depend on each other, so a long run of matches (low
  // divergence: hundreds of bases between two differences) costs one LDS / L2 round trip per 32 bases, not per 8
  while (n + 32 <= maxn) {
    const unsigned ov = (unsigned)(v + n - wP0), oh = (unsigned)(h + n - wT0);
    uint64_t x0, x1, x2, x3;
    if (ov <= (unsigned)(SEQ_WIN - 32) && oh <= (unsigned)(SEQ_WIN - 32)) {
      const uint32_t* a = winP + (ov >> 2);
      const uint32_t* b = winT + (oh >> 2);
      const unsigned sa = (ov & 3u) * 8u, sb = (oh & 3u) * 8u;
      uint32_t wa[9], wb[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) { wa[q] = a[q]; wb[q] = b[q]; }
      uint32_t d[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) d[q] = __builtin_amdgcn_alignbit(wa[q + 1], wa[q], sa) ^ __builtin_amdgcn_alignbit(wb[q + 1], wb[q], sb);
      x0 = ((uint64_t)d[1] << 32) | d[0]; x1 = ((uint64_t)d[3] << 32) | d[2];
      x2 = ((uint64_t)d[5] << 32) | d[4]; x3 = ((uint64_t)d[7] << 32) | d[6];
    } else {
      const uint8_t* pp = P + (v + n);
      const uint8_t* tt = T + (h + n);
      x0 = load8(pp) ^ load8(tt); x1 = load8(pp + 8) ^ load8(tt + 8);
      x2 = load8(pp + 16) ^ load8(tt + 16); x3 = load8(pp + 24) ^ load8(tt + 24);
    }
    if (x0 | x1 | x2 | x3) {
      if (x0) return n + (int)(__builtin_ctzll(x0) >> 3);
      if (x1) return n + 8 + (int)(__builtin_ctzll(x1) >> 3);
      if (x2) return n + 16 + (int)(__builtin_ctzll(x2) >> 3);
      return n + 24 + (int)(__builtin_ctzll(x3) >> 3);
    }
    n += 32;
  }
  while (n < maxn) {
    const uint64_t x = win_xor8(P, T, winP, winT, v + n, h + n, wP0, wT0);
    if (x) { n += (int)(__builtin_ctzll(x) >> 3); break; }
    n += 8;
  }
  return min(n, maxn);
}

// ---- wave-cooperative long extensions, reading through the LDS sequence windows (see wave_lce_tail_g) ----
template <bool USE_WIN>
__device__ __forceinline__ int wave_lce_tail(const uint8_t* P, const uint8_t* T, const uint32_t* winP, const uint32_t* winT, int v, int h,
                                             int n, int maxn, bool pend, int wP0, int wT0) {
  unsigned long long todo = __ballot(pend);
  const int lane = (int)(threadIdx.x & 63u);
  while (todo) {
    const int src = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(todo));
    todo &= todo - 1;
    const int v0 = rdlane(v, src), h0 = rdlane(h, src), mx = rdlane(maxn, src);
    int nn = rdlane(n, src);  // bases known to match so far (uniform)
    int res;
    for (;;) {
      const int off = nn + lane * 8;
      const bool past = off >= mx;
      uint64_t x = 0;
      if (!past) {
        if (USE_WIN) x = win_xor8(P, T, winP, winT, v0 + off, h0 + off, wP0, wT0);
        else x = load8(P + v0 + off) ^ load8(T + h0 + off);
      }
      const unsigned long long hit = __ballot(past || x != 0);
      if (hit) {
        const int f = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(hit));
        const unsigned xlo = (unsigned)rdlane((int)(uint32_t)x, f), xhi = (unsigned)rdlane((int)(uint32_t)(x >> 32), f);
        const uint64_t xf = ((uint64_t)xhi << 32) | xlo;
        const int at = nn + f * 8;
        res = at >= mx ? mx : min(mx, at + (xf ? (int)(__builtin_ctzll(xf) >> 3) : 0));
        break;
      }
      nn += 512;
    }
    if (lane == src) n = res;
  }
  return n;
}

// what a lane does on its own: 8 bases (x8 = their xor, already loaded), then one round of 32; more = the run goes on
template <bool USE_WIN>
__device__ __forceinline__ int lce_head40(const uint8_t* P, const uint8_t* T, const uint32_t* winP, const uint32_t* winT, int v, int h,
                                          uint64_t x8, int maxn, int wP0, int wT0, bool& more) {
  more = false;
  if (x8) return min((int)(__builtin_ctzll(x8) >> 3), maxn);
  if (maxn <= 8) return maxn;
  if (maxn < 40) {  // the end of the sequences is near: 8 bases at a time
    int n = 8;
    while (n < maxn) {
      const uint64_t x = USE_WIN ? win_xor8(P, T, winP, winT, v + n, h + n, wP0, wT0) : (load8(P + v + n) ^ load8(T + h + n));
      if (x) { n += (int)(__builtin_ctzll(x) >> 3); break; }
      n += 8;
    }
    return min(n, maxn);
  }
  uint64_t x0, x1, x2, x3;
  const unsigned ov = (unsigned)(v + 8 - wP0), oh = (unsigned)(h + 8 - wT0);
  if (USE_WIN && ov <= (unsigned)(SEQ_WIN - 32) && oh <= (unsigned)(SEQ_WIN - 32)) {
    const uint32_t* a = winP + (ov >> 2);
    const uint32_t* b = winT + (oh >> 2);
    const unsigned sa = (ov & 3u) * 8u, sb = (oh & 3u) * 8u;
    uint32_t wa[9], wb[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) { wa[q] = a[q]; wb[q] = b[q]; }
    uint32_t d[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) d[q] = __builtin_amdgcn_alignbit(wa[q + 1], wa[q], sa) ^ __builtin_amdgcn_alignbit(wb[q + 1], wb[q], sb);
    x0 = ((uint64_t)d[1] << 32) | d[0]; x1 = ((uint64_t)d[3] << 32) | d[2];
    x2 = ((uint64_t)d[5] << 32) | d[4]; x3 = ((uint64_t)d[7] << 32) | d[6];
  } else {
    const uint8_t* pp = P + (v + 8);
    const uint8_t* tt = T + (h + 8);
    x0 = load8(pp) ^ load8(tt); x1 = load8(pp + 8) ^ load8(tt + 8);
    x2 = load8(pp + 16) ^ load8(tt + 16); x3 = load8(pp + 24) ^ load8(tt + 24);
  }
  if (x0) return 8 + (int)(__builtin_ctzll(x0) >> 3);
  if (x1) return 16 + (int)(__builtin_ctzll(x1) >> 3);
  if (x2) return 24 + (int)(__builtin_ctzll(x2) >> 3);
  if (x3) return 32 + (int)(__builtin_ctzll(x3) >> 3);
  more = maxn > 40;
  return 40;
}

// P2 = true: the phase-2 form (see P2Job in wfa_device.h) -- the two directions start at their own scores (J.tf / J.tr), T
// is P2K, every row of the core goes to the job's P2 rows with its per-component maxima, and there is no output snapshot.
template <int C, int NTMAX, int LX, int LA, int LB, int E1, int E2, bool P2 = false, bool CUT = true>
__global__ __launch_bounds__(NTMAX) void wfa_tile_reg_kernel(const uint8_t* __restrict__ seq, int32_t* __restrict__ ring_arena,
                                                           const TileJob* __restrict__ jobs, const TileTask* __restrict__ tasks,
                                                           int32_t* __restrict__ mak_out, int T, int32_t* __restrict__ p2_arena = nullptr,
                                                           int32_t* __restrict__ p2max = nullptr) {
  constexpr int H = LB + 1;  // rows of M the output snapshot must hold: s_end-LB .. s_end
                             // (the overlap test of the step kernel looks LB rows behind the resume score)
  // The default lags x = 5, o1+e1 = 10, o2+e2 = 25 are all multiples of 5: a step only ever reads M rows of
  // its own residue class (s mod 5).  The M history is therefore kept as 5 delay lines of depth 6 per
  // diagonal and a step shifts ONE of them (6 moves) instead of a 26-deep line (25 moves).
  constexpr int NCL = 5, DEP = 6;
  static_assert(LX == 5 && LA == 10 && LB == 25 && E1 == 2 && E2 == 1, "lags");
  __shared__ int s_edge[2][16][2][4];  // [parity][wave][0: lane63 -> next wave, 1: lane0 -> previous wave][value]
  __shared__ __attribute__((aligned(16))) uint32_t s_winP[SEQ_WIN / 4 + 4], s_winT[SEQ_WIN / 4 + 4];
  __shared__ int s_wlo[2];
  extern __shared__ __attribute__((aligned(16))) int s_makr[];  // [T + 1]
  TileTask tk = tasks[blockIdx.x];
  const TileJob J = jobs[tk.job];
  if (!J.active) return;
  const int sbase = P2 ? (tk.dir == 0 ? J.tf : J.tr) : J.s0;  // score of the snapshot this direction starts from
  // CUT = false: no job of the launch carries a score bound that can bind (the host checks): the ranges are the triangle's,
  // and the step loop is spared the bound's bookkeeping (7 % of its instructions on C3)
  const Rng RG = make_rng(J.pl, J.tl, CUT ? J.sub : SUB_NONE);
  auto rlo = [&](int sc) { return CUT ? rng_lo(RG, sc) : max(-RG.pl, -sc); };
  auto rhi = [&](int sc) { return CUT ? rng_hi(RG, sc) : min(RG.tl, sc); };
  int halo = T;  // columns computed on either side of the core (the trapezoid loses one per step)
  {  // tasks carry (tile index, tile width): this block's diagonal range [-s1, s1], clipped to the problem, is cut
     // into tiles from its own left end, so every tile but the last is full
    const int s1 = sbase + T;
    int L, R;
    if (CUT) rng_block(RG, sbase, s1, L, R);
    else { L = max(-J.pl, -s1); R = min(J.tl, s1); }
    const int idx = tk.core_lo, core = tk.core_hi;
    tk.core_lo = L + idx * core;
    tk.core_hi = min(R, tk.core_lo + core - 1);
    if (tk.core_lo > R) return;
    if (tk.core_lo == L && tk.core_hi == R) halo = 0;  // one tile for the whole range: nothing beside it to take from
  }
  const int dir = tk.dir, tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = NT >> 6;
  if (tid < 2) s_wlo[tid] = INT32_MAX;
  const uint8_t* P = seq + (dir == 0 ? J.p_fwd : J.p_rev);
  const uint8_t* Tx = seq + (dir == 0 ? J.t_fwd : J.t_rev);
  const int pl = J.pl, tl = J.tl, s0 = sbase;
  const int kA = tk.core_lo - halo;
  const int k0 = kA + tid * C;  // first diagonal of this thread
  const int64_t width = J.width;
  const int32_t* rin = ring_arena + J.ring_in + J.koff + (int64_t)dir * 5 * RING * width;
  int32_t* rout = ring_arena + J.ring_out + J.koff + (int64_t)dir * 5 * RING * width;
  const int kmax = tk.core_hi + halo;  // last diagonal of the tile
  // a job whose meeting point is known runs its last block only up to it (per direction)
  const int Tn = (!P2 && J.mode == 1) ? (dir == 0 ? J.tf : J.tr) : T;

  // Mh[c][r][e] = M[sr - 5 e][k0+c], sr = the newest score <= current with (sr - s0) mod 5 == r
  int Mh[C][NCL][DEP];
  int I1h[C][E1], D1h[C][E1], I2h[C], D2h[C];
  // ---- snapshot load (rows <= s0), column-blocked ----
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int k = k0 + c;
    const bool kin = k <= kmax;
#pragma unroll
    for (int r = 0; r < NCL; ++r)
#pragma unroll
      for (int e = 0; e < DEP; ++e) Mh[c][r][e] = WF_NULL;
#pragma unroll
    for (int d = 0; d < H; ++d) {
      const int sc = s0 - d;
      const int v = (kin && sc >= 0 && k >= rlo(sc) && k <= rhi(sc)) ? rin[((int64_t)(C_M * RING + (sc & RMASK))) * width + k] : WF_NULL;
      Mh[c][(NCL - d % NCL) % NCL][d / NCL] = v;  // row s0-d: class (-d mod 5), the (d/5)-th newest of its class
    }
#pragma unroll
    for (int d = 0; d < E1; ++d) {
      const int sc = s0 - d;
      const bool ok = kin && sc >= 0 && k >= rlo(sc) && k <= rhi(sc);
      I1h[c][d] = ok ? rin[((int64_t)(C_I1 * RING + (sc & RMASK))) * width + k] : WF_NULL;
      D1h[c][d] = ok ? rin[((int64_t)(C_D1 * RING + (sc & RMASK))) * width + k] : WF_NULL;
    }
    {
      const bool ok = kin && s0 >= 0 && k >= rlo(s0) && k <= rhi(s0);
      I2h[c] = ok ? rin[((int64_t)(C_I2 * RING + (s0 & RMASK))) * width + k] : WF_NULL;
      D2h[c] = ok ? rin[((int64_t)(C_D2 * RING + (s0 & RMASK))) * width + k] : WF_NULL;
    }
  }
  for (int t = tid; t <= T; t += NT) s_makr[t] = 0;
  unsigned hmaxu[C];
  bool colok[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int k = k0 + c;
    colok[c] = (k >= -pl) && (k <= tl);
    hmaxu[c] = colok[c] ? (unsigned)min(tl, pl + k) : 0u;
  }
  // ---- sequence windows: every offset this tile will ever extend from is >= the smallest live offset of
  // its history (a cell's text offset h and pattern offset v = h - k never fall below its source's)
  {
    int hlo = INT32_MAX, vlo = INT32_MAX;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      int lo = INT32_MAX;
#pragma unroll
      for (int r = 0; r < NCL; ++r)
#pragma unroll
        for (int e = 0; e < DEP; ++e) lo = min(lo, Mh[c][r][e] >= 0 ? Mh[c][r][e] : INT32_MAX);
#pragma unroll
      for (int d = 0; d < E1; ++d) { lo = min(lo, I1h[c][d] >= 0 ? I1h[c][d] : INT32_MAX); lo = min(lo, D1h[c][d] >= 0 ? D1h[c][d] : INT32_MAX); }
      lo = min(lo, I2h[c] >= 0 ? I2h[c] : INT32_MAX);
      lo = min(lo, D2h[c] >= 0 ? D2h[c] : INT32_MAX);
      if (lo != INT32_MAX) { hlo = min(hlo, lo); vlo = min(vlo, lo - k); }
    }
    __syncthreads();  // s_wlo initialised
    if (hlo != INT32_MAX) { atomicMin(&s_wlo[0], hlo); atomicMin(&s_wlo[1], max(vlo, 0)); }
    __syncthreads();
  }
  const int wT0 = s_wlo[0] == INT32_MAX ? 0 : (s_wlo[0] & ~7), wP0 = s_wlo[1] == INT32_MAX ? 0 : (s_wlo[1] & ~7);
  for (int i = tid; i < SEQ_WIN / 8 + 1; i += NT) {
    // sequences are padded by 64 readable bytes past their end (SEQ_PAD); beyond that the window holds zeros,
    // which is harmless: extensions are cut at the sequence ends
    const int bt = wT0 + 8 * i, bp = wP0 + 8 * i;
    const uint64_t vt = bt + 8 <= tl + 64 ? load8(Tx + bt) : 0ull, vp = bp + 8 <= pl + 64 ? load8(P + bp) : 0ull;
    s_winT[2 * i] = (uint32_t)vt; s_winT[2 * i + 1] = (uint32_t)(vt >> 32);
    s_winP[2 * i] = (uint32_t)vp; s_winP[2 * i + 1] = (uint32_t)(vp >> 32);
  }
  __syncthreads();

  for (int tb = 0; tb < Tn; tb += NCL) {
#pragma unroll
  for (int jj = 1; jj <= NCL; ++jj) {
    const int t = tb + jj;
    if (t > Tn) break;
    const int cl = jj % NCL;  // residue class of this step's score: compile time after unrolling
    const int s = s0 + t;
    const int par = t & 1;
    // (Tried in round 3 and taken out again: letting a wave whose diagonals lie outside every row of the step skip to the
    // step's barrier.  On pangenome batches 40 % of the waves are such waves, and skipping them changed nothing -- the step's
    // time is the chain of its busiest wave, the idle ones cost issue slots nobody was waiting for -- while the test itself
    // made the step 10 % slower on C3.)
    // rows this step reads from its class: [0] = s-5, [1] = s-10, [4] = s-25
    // publish the wave-edge history values needed by the neighbouring waves in this step
    if (lane == 63) { int* e = s_edge[par][wv][0]; e[0] = Mh[C - 1][cl][1]; e[1] = Mh[C - 1][cl][4]; e[2] = I1h[C - 1][E1 - 1]; e[3] = I2h[C - 1]; }
    if (lane == 0)  { int* e = s_edge[par][wv][1]; e[0] = Mh[0][cl][1];     e[1] = Mh[0][cl][4];     e[2] = D1h[0][E1 - 1];     e[3] = D2h[0]; }
    __syncthreads();
    // left neighbour (k0 - 1) and right neighbour (k0 + C) values
    int lM10 = __shfl_up(Mh[C - 1][cl][1], 1, 64), lM25 = __shfl_up(Mh[C - 1][cl][4], 1, 64);
    int lI1 = __shfl_up(I1h[C - 1][E1 - 1], 1, 64), lI2 = __shfl_up(I2h[C - 1], 1, 64);
    int rM10 = __shfl_down(Mh[0][cl][1], 1, 64), rM25 = __shfl_down(Mh[0][cl][4], 1, 64);
    int rD1 = __shfl_down(D1h[0][E1 - 1], 1, 64), rD2 = __shfl_down(D2h[0], 1, 64);
    if (lane == 0) {
      if (wv > 0) { const int* e = s_edge[par][wv - 1][0]; lM10 = e[0]; lM25 = e[1]; lI1 = e[2]; lI2 = e[3]; }
      else { lM10 = lM25 = lI1 = lI2 = WF_NULL; }
    }
    if (lane == 63) {
      if (wv + 1 < nw) { const int* e = s_edge[par][wv + 1][1]; rM10 = e[0]; rM25 = e[1]; rD1 = e[2]; rD2 = e[3]; }
      else { rM10 = rM25 = rD1 = rD2 = WF_NULL; }
    }
    // The closed-form source ranges are nested, the oldest row (s - LB) is the narrowest:
    // a thread whose neighbourhood k0-1 .. k0+C lies inside it needs no range select at all.
    const int sb = s - LB;
    // (the triangle's bound is tightest at the oldest row, the score bound's at the newest: s - E2)
    const bool interior = CUT ? (sb >= 0 && (k0 - 1 >= max(rlo(sb), rlo(s - E2))) && (k0 + C <= min(rhi(sb), rhi(s - E2))))
                              : (sb >= 0 && (k0 - 1 >= rlo(sb)) && (k0 + C <= rhi(sb)));
    int nM[C], nI1[C], nI2[C], nD1[C], nD2[C];
    int mak = 0;
    const int cut_lo = RG.kb_lo + s, cut_hi = RG.kb_hi - s;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      int a10 = c == 0 ? lM10 : Mh[c - 1][cl][1], b10 = c == C - 1 ? rM10 : Mh[c + 1][cl][1];
      int a25 = c == 0 ? lM25 : Mh[c - 1][cl][4], b25 = c == C - 1 ? rM25 : Mh[c + 1][cl][4];
      int i1 = c == 0 ? lI1 : I1h[c - 1][E1 - 1], d1 = c == C - 1 ? rD1 : D1h[c + 1][E1 - 1];
      int i2 = c == 0 ? lI2 : I2h[c - 1], d2 = c == C - 1 ? rD2 : D2h[c + 1];
      int mx = Mh[c][cl][0];
      if (!interior) {
        const int sx = s - LX, sa = s - LA, se1 = s - E1, se2 = s - E2;
        const int lx = sx >= 0 ? rlo(sx) : 1, hx = sx >= 0 ? rhi(sx) : 0;
        const int la = sa >= 0 ? rlo(sa) : 1, ha = sa >= 0 ? rhi(sa) : 0;
        const int lb = sb >= 0 ? rlo(sb) : 1, hb = sb >= 0 ? rhi(sb) : 0;
        const int le1 = se1 >= 0 ? rlo(se1) : 1, he1 = se1 >= 0 ? rhi(se1) : 0;
        const int le2 = se2 >= 0 ? rlo(se2) : 1, he2 = se2 >= 0 ? rhi(se2) : 0;
        a10 = sel_rng(a10, k - 1, la, ha); i1 = sel_rng(i1, k - 1, le1, he1);
        a25 = sel_rng(a25, k - 1, lb, hb); i2 = sel_rng(i2, k - 1, le2, he2);
        b10 = sel_rng(b10, k + 1, la, ha); d1 = sel_rng(d1, k + 1, le1, he1);
        b25 = sel_rng(b25, k + 1, lb, hb); d2 = sel_rng(d2, k + 1, le2, he2);
        mx = sel_rng(mx, k, lx, hx);
      }
      // in-bounds <=> 0 <= offset <= min(tl, pl + k)   (h <= tl and h - k <= pl)
      const unsigned hm = hmaxu[c];
      int ins1 = max(a10, i1) + 1, ins2 = max(a25, i2) + 1, del1 = max(b10, d1), del2 = max(b25, d2), mis = mx + 1;
      ins1 = (unsigned)ins1 <= hm ? ins1 : WF_NULL;
      ins2 = (unsigned)ins2 <= hm ? ins2 : WF_NULL;
      del1 = (unsigned)del1 <= hm ? del1 : WF_NULL;
      del2 = (unsigned)del2 <= hm ? del2 : WF_NULL;
      mis = (unsigned)mis <= hm ? mis : WF_NULL;
      nI1[c] = ins1; nI2[c] = ins2; nD1[c] = del1; nD2[c] = del2;
      int m = max(imax3(ins1, ins2, mis), max(del1, del2));
      // columns outside [-pl, tl] hold no cell at all (hmaxu = 0 would let offset 0 through), nor do columns the score
      // bound has cut off (their values would never be read; their extensions would be done for nothing)
      nM[c] = (colok[c] && (!CUT || (k >= cut_lo && k <= cut_hi))) ? m : WF_NULL;
    }
    // extension: first 8 bases of all C cells in flight together
    uint64_t x[C];
    int maxn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c, m = nM[c];
      x[c] = 0; maxn[c] = 0;
      if (m >= 0) {
        maxn[c] = min(pl - (m - k), tl - m);
        x[c] = win_xor8(P, Tx, s_winP, s_winT, m - k, m, wP0, wT0);
      }
    }
    int ext[C];
    bool more[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      ext[c] = 0; more[c] = false;
      if (nM[c] >= 0) ext[c] = lce_head40<true>(P, Tx, s_winP, s_winT, nM[c] - (k0 + c), nM[c], x[c], maxn[c], wP0, wT0, more[c]);
    }
    // runs longer than 40 bases: the wave finishes them together (uniform control flow: every lane is here)
#pragma unroll
    for (int c = 0; c < C; ++c)
      if (__any(more[c])) ext[c] = wave_lce_tail<true>(P, Tx, s_winP, s_winT, nM[c] - (k0 + c), nM[c], ext[c], maxn[c], more[c], wP0, wT0);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      int m = nM[c];
      if (m >= 0) {
        m += min(ext[c], maxn[c]);
        nM[c] = m;
        if (k >= tk.core_lo && k <= tk.core_hi && k >= rlo(s) && k <= rhi(s)) mak = max(mak, 2 * m - k);
      }
    }
    if (P2) {
      // every row of the core is kept: five components into the job's P2 rows
      int32_t* prow = p2_arena + J.p2_off + J.koff2 + ((int64_t)(dir * 5) * P2K + (t - 1)) * J.w2;
      const int64_t cstride = (int64_t)P2K * J.w2;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int k = k0 + c;
        if (k >= tk.core_lo && k <= tk.core_hi && k >= rlo(s) && k <= rhi(s)) {
          prow[C_M * cstride + k] = nM[c]; prow[C_I1 * cstride + k] = nI1[c]; prow[C_I2 * cstride + k] = nI2[c];
          prow[C_D1 * cstride + k] = nD1[c]; prow[C_D2 * cstride + k] = nD2[c];
        }
      }
    }
    // stream the last H rows of I/D of the core to the output snapshot
    if (!P2 && t > Tn - H) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int k = k0 + c;
        if (k >= tk.core_lo && k <= tk.core_hi && k >= rlo(s) && k <= rhi(s)) {
          const int64_t ro = ((int64_t)(s & RMASK)) * width + k;
          rout[(int64_t)C_I1 * RING * width + ro] = nI1[c];
          rout[(int64_t)C_I2 * RING * width + ro] = nI2[c];
          rout[(int64_t)C_D1 * RING * width + ro] = nD1[c];
          rout[(int64_t)C_D2 * RING * width + ro] = nD2[c];
        }
      }
    }
    // advance the delay line of this step's class only
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int e = DEP - 1; e > 0; --e) Mh[c][cl][e] = Mh[c][cl][e - 1];
      Mh[c][cl][0] = nM[c];
#pragma unroll
      for (int d = E1 - 1; d > 0; --d) { I1h[c][d] = I1h[c][d - 1]; D1h[c][d] = D1h[c][d - 1]; }
      I1h[c][0] = nI1[c]; D1h[c][0] = nD1[c];
      I2h[c] = nI2[c]; D2h[c] = nD2[c];
    }
    mak = wave_max_dpp63(mak);
    if (lane == 63 && mak > 0) atomicMax(&s_makr[t], mak);
  }
  }
  // ---- output snapshot: the newest H rows of M for the core ----
  // row s_end - d lives in class (T - d) mod 5 at depth (d - (T - class) mod 5) / 5; T mod 5 is uniform, one
  // compile-time variant per value keeps the history in registers
  if (P2) return;
  const int s_end = s0 + Tn;
  if (Tn < H) {
    // a short last block: the I/D rows of scores <= s0 that the step kernel still looks at live in the input ring
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      if (k < tk.core_lo || k > tk.core_hi) continue;
      for (int d = Tn; d < H; ++d) {
        const int sc = s_end - d;
        if (sc < 0 || k < rlo(sc) || k > rhi(sc)) continue;
        const int64_t ro = ((int64_t)(sc & RMASK)) * width + k;
        rout[(int64_t)C_I1 * RING * width + ro] = rin[(int64_t)C_I1 * RING * width + ro];
        rout[(int64_t)C_I2 * RING * width + ro] = rin[(int64_t)C_I2 * RING * width + ro];
        rout[(int64_t)C_D1 * RING * width + ro] = rin[(int64_t)C_D1 * RING * width + ro];
        rout[(int64_t)C_D2 * RING * width + ro] = rin[(int64_t)C_D2 * RING * width + ro];
      }
    }
  }
  auto write_rows = [&](auto TR) {
    constexpr int tr = decltype(TR)::value;  // T mod 5
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      if (k < tk.core_lo || k > tk.core_hi) continue;
#pragma unroll
      for (int d = 0; d < H; ++d) {
        const int r = ((tr - d) % NCL + NCL) % NCL;        // class of row s_end - d
        const int back = ((tr - r) % NCL + NCL) % NCL;     // s_end - (newest score of class r)
        const int e = (d - back) / NCL;
        const int sc = s_end - d;
        if (sc >= 0 && k >= rlo(sc) && k <= rhi(sc)) rout[((int64_t)(C_M * RING + (sc & RMASK))) * width + k] = Mh[c][r][e];
      }
    }
  };
  switch (Tn % NCL) {
    case 0: write_rows(std::integral_constant<int, 0>{}); break;
    case 1: write_rows(std::integral_constant<int, 1>{}); break;
    case 2: write_rows(std::integral_constant<int, 2>{}); break;
    case 3: write_rows(std::integral_constant<int, 3>{}); break;
    default: write_rows(std::integral_constant<int, 4>{}); break;
  }
  __syncthreads();
  int32_t* mk = mak_out + ((int64_t)tk.job * 2 + dir) * T;
  for (int t = 1 + tid; t <= T; t += NT) if (s_makr[t] > 0) atomicMax(&mk[t - 1], s_makr[t]);
}

// Between two tile blocks: one lane per job replays the alternating forward / reverse checks of
// wavefront_bialign_find_breakpoint over the T per-score maxima of the block just computed.  A job whose
// wavefronts met inside the block (or that ran out of steps) goes inactive with its state still at the
// block's START -- wfa_bp_kernel redoes that block step by step; the others move on to the next block.
// one wave per job: the T per-score maxima are read once, prefix maxima replace the sequential replay
// (round 6) `coarse`: wfa_tile2_kernel kept ONE maximum per direction for a block that ends below TileJob::fine_s (in the block's last slot: the
// prefix maxima below make of it what the per-score maxima would have left at the block's end, and the running maxima are monotone, so "the
// directions met in this block" is decided exactly).  Such a block runs again with per-score maxima (mode 5) before the run up to the meeting point.
__global__ __launch_bounds__(64) void wfa_tile_advance_kernel(TileJob* __restrict__ jobs, int32_t* __restrict__ mak, int njobs, int T, DevPen pen,
                                                              int exact, int coarse, int launched) {
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= njobs) return;
  TileJob J = jobs[i];
  if (!J.active) return;
  if (coarse && (J.packed & 1)) {
    // `launched`: which instantiations of wfa_tile2_kernel ran this block (bit 0 without, bit 1 with per-score maxima).  The host leaves out the
    // one it expects no tile for; a job whose state asks for it all the same (mode 5 set by the block before, inside the host's chunk) has not
    // moved and waits for the next chunk
    // (the FINE instantiation launched alone took every tile: nobody waits then)
    const bool job_fine = J.mode == 5 || (J.mode == 0 && J.s0 + T >= J.fine_s);
    if (launched != 2 && !(launched & (job_fine ? 2 : 1))) return;
  }
  int32_t* mf = mak + ((int64_t)i * 2 + 0) * T;
  int32_t* mr = mak + ((int64_t)i * 2 + 1) * T;
  if (J.mode == 1) {
    // the block that stops at the meeting point has run: its output ring is the snapshot the step kernel starts phase 2 from
    for (int t = lane; t < T; t += 64) { mf[t] = 0; mr[t] = 0; }
    if (lane == 0) {
      const int64_t t = J.ring_in; J.ring_in = J.ring_out; J.ring_out = t;
      J.mode = 2; J.active = 0; J.nblocks += 1;
      jobs[i] = J;
    }
    return;
  }
  if (J.mode == 6) {
    // the block before s0 has run again, with the gap rows of its last 26 scores: ring_out holds the snapshot of s0 the short run up to the
    // meeting point needs; what was ring_in (the same snapshot without those rows) takes that run's output
    for (int t = lane; t < T; t += 64) { mf[t] = 0; mr[t] = 0; }
    if (lane == 0) {
      const int64_t t = J.ring_in; J.ring_in = J.ring_out; J.ring_out = t;
      J.mode = 1; J.reran += 1;
      jobs[i] = J;
    }
    return;
  }
  const int A = J.pl + J.tl - 1;
  int fm = J.fmax, rm = J.rmax;  // running maxima before the chunk of 64 scores at hand (uniform)
  int tf = 0, tr = 0, last_fwd = 0;
  bool term = false;
  for (int base = 0; base < T && !term; base += 64) {
    const int t = base + lane;
    const int vf = t < T ? mf[t] : 0, vr = t < T ? mr[t] : 0;
    int pf = vf, pr = vr;  // inclusive prefix maxima over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int uf = __shfl_up(pf, d, 64), ur = __shfl_up(pr, d, 64);
      if (lane >= d) { pf = max(pf, uf); pr = max(pr, ur); }
    }
    int er = __shfl_up(pr, 1, 64);  // exclusive prefix maximum of the reverse direction
    if (lane == 0) er = 0;
    const int fm_t = max(fm, pf), rm_t = max(rm, pr), rm_before = max(rm, er);
    // the reference alternates: forward step t, check, reverse step t, check
    const bool hit_f = t < T && fm_t + rm_before >= A, hit_r = t < T && fm_t + rm_t >= A;
    const unsigned long long bf = __ballot(hit_f), br = __ballot(hit_r);
    if (bf | br) {
      const int lf = bf ? __builtin_ctzll(bf) : 64, lr = br ? __builtin_ctzll(br) : 64;
      term = true;
      if (lf <= lr) { last_fwd = 1; tf = base + lf + 1; tr = base + lf; fm = __shfl(fm_t, lf, 64); rm = __shfl(rm_before, lf, 64); }
      else { last_fwd = 0; tf = tr = base + lr + 1; fm = __shfl(fm_t, lr, 64); rm = __shfl(rm_t, lr, 64); }
    } else {
      fm = __shfl(fm_t, 63, 64); rm = __shfl(rm_t, 63, 64);
    }
  }
  for (int t = lane; t < T; t += 64) { mf[t] = 0; mr[t] = 0; }
  if (lane != 0) return;
  const int64_t max_steps = (int64_t)(pen.o1 + pen.o2) * 4 + (int64_t)(J.pl + J.tl + 2) * max(pen.x, max(pen.e1, pen.e2)) * 2 + 256;
  J.nblocks += 1;
  const bool was_coarse = coarse && (launched & 1) && (J.packed & 1) && J.mode == 0 && J.s0 + T < J.fine_s;  // (wfa_tile2_kernel's own test)
  if (term && exact && was_coarse) {
    // the same block again, with a maximum per score this time: same input ring, same running maxima.  fine_s = -1 tells the host that this
    // job computed the block once more than nblocks says (its cell count), and keeps every later block of the job fine
    J.mode = 5; J.fine_s = -1; J.nblocks -= 1;
  } else if (term && exact) {
    // redo this block, but only up to the meeting point (forward tf steps, reverse tr): same input ring -- unless one of the two runs is
    // shorter than the rows phase 2 reads behind it and the snapshot holds no gap rows that deep (TileJob::ring_prev): the block before first
    J.mode = (J.ring_prev >= 0 && J.prev_ok && min(tf, tr) < 26) ? 6 : 1;
    J.tf = tf; J.tr = tr; J.last_fwd = last_fwd;
    J.fmax = fm; J.rmax = rm;
  } else if (term || 2 * (int64_t)(J.s0 + T) > max_steps) {
    J.active = 0;
  } else {
    J.fmax = fm; J.rmax = rm;
    J.s0 += T;
    J.mode = 0;
    if (J.ring_prev >= 0) {  // three rings: the input of the block just computed stays one block longer
      const int64_t t = J.ring_prev; J.ring_prev = J.ring_in; J.ring_in = J.ring_out; J.ring_out = t;
      J.prev_ok = 1;
    } else {
      const int64_t t = J.ring_in; J.ring_in = J.ring_out; J.ring_out = t;
    }
  }
  jobs[i] = J;
}

// ---------------------------------------------------------------------------
// Phase 2 from rows computed ahead (P2Job in wfa_device.h)
// ---------------------------------------------------------------------------
// row s of (direction d, component cc), addressable by diagonal: a snapshot row of the ring, or one of the P2 rows
__device__ __forceinline__ const int32_t* p2_row(const int32_t* ring, const int32_t* p2, const P2Job& J, int d, int cc, int s) {
  const int sd = d == 0 ? J.sf : J.sr;
  if (s <= sd) return ring + J.ring_in + J.koff + ((int64_t)((d * 5 + cc) * RING + (s & RMASK))) * J.width;
  return p2 + J.p2_off + J.koff2 + ((int64_t)((d * 5 + cc) * P2K + (s - sd - 1))) * J.w2;
}
__device__ __forceinline__ const int32_t* p2_maxrow(const int32_t* p2max, int job, const P2Job& J, int d, int s) {
  const int sd = d == 0 ? J.sf : J.sr;
  return p2max + (((int64_t)job * 2 + d) * P2ROWS + (s - (sd - 25))) * 5;
}
// Maxima of every row of the phase-2 window (the snapshot's rows sd-25 .. sd from the ring, sd+1 .. sd+P2K from the P2 rows):
// per component over the whole row (p2max) and per block of 64 diagonals (bmax).  One workgroup per (job, direction, row),
// one wave per block at a time.
__global__ __launch_bounds__(256) void wfa_p2_blockmax_kernel(const int32_t* __restrict__ ring, const int32_t* __restrict__ p2,
                                                             const P2Job* __restrict__ jobs, int32_t* __restrict__ bmax,
                                                             int32_t* __restrict__ p2max) {
  const int job = blockIdx.x / (2 * P2ROWS), r2 = blockIdx.x % (2 * P2ROWS), d = r2 / P2ROWS, r = r2 % P2ROWS;
  const P2Job J = jobs[job];
  const int s = (d == 0 ? J.sf : J.sr) - 25 + r;
  const Rng RG = make_rng(J.pl, J.tl, J.sub);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __shared__ int s_mx[5];
  if (threadIdx.x < 5) s_mx[threadIdx.x] = 0;
  __syncthreads();
  int32_t* bm = bmax + J.bm_off + ((int64_t)(d * P2ROWS + r) * 5) * J.nblk;
  int mx[5] = {0, 0, 0, 0, 0};
  const int lo = s >= 0 ? rng_lo(RG, s) : 1, hi = s >= 0 ? rng_hi(RG, s) : 0;
  const int32_t* row[5];
#pragma unroll
  for (int cc = 0; cc < 5; ++cc) row[cc] = s >= 0 ? p2_row(ring, p2, J, d, cc, s) : nullptr;
  for (int b = wv; b < J.nblk; b += nw) {
    const int k = (b << 6) - J.koff2 + lane;
    // The maxima are taken over ANTIDIAGONALS, 2 h - k = h + v, not over offsets: two cells on mirrored diagonals (k0 + k1 = tl - pl)
    // meet, o0 + o1 >= tl, iff their antidiagonals sum to >= tl + pl -- and a wavefront is flat in antidiagonals where its
    // offsets differ by half a block's width from one end of a block to the other.  (Offset maxima, until round 4: the
    // two directions of a job in phase 2 ARE within a block's width of touching everywhere, so hardly a block was pruned --
    // a 2.5 kb pair of unrelated sequences in an LPA level tested 13 M cell pairs, 3.4 ms for a launch whose average walk took 0.1.)
    int v[5] = {0, 0, 0, 0, 0};
    if (k >= lo && k <= hi) {
#pragma unroll
      for (int cc = 0; cc < 5; ++cc) { const int o = row[cc][k]; v[cc] = o >= 0 ? 2 * o - k : 0; }
    }
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) {
      const int m = wave_max_dpp63(v[cc]);
      if (lane == 63) { bm[(int64_t)cc * J.nblk + b] = m; mx[cc] = max(mx[cc], m); }
    }
  }
  if (lane == 63) {
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) if (mx[cc] > 0) atomicMax(&s_mx[cc], mx[cc]);
  }
  __syncthreads();
  if (threadIdx.x < 5) p2max[(((int64_t)job * 2 + d) * P2ROWS + r) * 5 + threadIdx.x] = s_mx[threadIdx.x];
}

// wavefront_bialign_find_breakpoint's second loop over rows that are all there already: one workgroup per job walks the
// tests in the reference's order -- the data-parallel half of a test (smallest diagonal per (row of the other direction,
// component) on which the offsets meet, pruned by the best breakpoint so far and by the row maxima) and one lane's replay of
// the nested conditions, exactly as wfa_bp_kernel does them -- but no row is computed between two tests.
// Running maxima of the block maxima over the rows of a direction: pb[row r] = max over rows 0 .. r.  What a test
// prunes with (whole blocks of the tested row, then single diagonals) must only be an upper bound of what the rows in its
// scope hold in a block; the rows before them hold less almost everywhere, so the bound loses little and costs one read.
__global__ __launch_bounds__(256) void wfa_p2_prefixmax_kernel(const P2Job* __restrict__ jobs, const int32_t* __restrict__ bmax,
                                                              int32_t* __restrict__ pbmax, int njobs) {
  const int job = blockIdx.y;
  const P2Job J = jobs[job];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;  // (direction, component, block)
  const int per_dir = 5 * J.nblk;
  if (e >= 2 * per_dir) return;
  const int d = e / per_dir, rest = e % per_dir;  // rest = cc * nblk + b
  const int32_t* src = bmax + J.bm_off + (int64_t)d * P2ROWS * per_dir + rest;
  int32_t* dst = pbmax + J.bm_off + (int64_t)d * P2ROWS * per_dir + rest;
  int run = 0;
  for (int r = 0; r < P2ROWS; ++r) {
    run = max(run, src[(int64_t)r * per_dir]);
    dst[(int64_t)r * per_dir] = run;
  }
}

__device__ unsigned long long g_p2cnt[8];  // WFM_P2_COUNT diagnostics: tests, tests with candidates, pairs listed, blocks tested cell by cell, pairs that met, most pairs in a round, most blocks one wave tested in a round

// The loop for one workgroup, P2G tests per round.  Every stage of a round is spread over the threads; the stages are
// separated by barriers (each costs a round trip or two to the job's rows in L2, which is what a test's time is made of --
// hence several tests per round):
//   pairs    per test: which (row i of the other direction, component) pairs can still improve the best breakpoint (score)
//            and can reach tl at all (row maxima); they are listed
//   scan     one wave per (test, component) and all its rows: the 64-diagonal blocks of the tested row in ascending order, each
//            against the block maxima of every row (one row per lane), then the cells of the rows that pass (four rows
//            in flight), every pair up to the first diagonal on which its offsets meet
//   pick     test after test in the reference's order: the pair its nested loop would end up with -- smallest score, first in
//            its order among equals -- and the loop's own end condition
// The tests of a round see the best breakpoint as it was when the round began: that only lets more pairs through the first
// two stages than a test on its own would look at; what a test takes is decided in `pick`, with the best of that moment.
constexpr int P2G = 8;
constexpr int P2WORK = 3072;  // blocks one round's work list holds (24 KB of LDS)
constexpr int P2CM = 128;     // widest rows (in blocks of 64 diagonals) whose column maxima are kept in LDS (20 KB); wider jobs prune by the rows' maxima only
                              // (with antidiagonal maxima those prune well on their own; C3's rows of 200 blocks: 13.9 -> 12.7 ms per step without)
__global__ __launch_bounds__(1024) void wfa_p2_overlap_kernel(const int32_t* __restrict__ ring, const int32_t* __restrict__ p2,
                                                             const P2Job* __restrict__ jobs, const int32_t* __restrict__ p2max,
                                                             const int32_t* __restrict__ bmax, const int32_t* __restrict__ pbmax,
                                                             BpResult* __restrict__ results, DevPen pen, int scope, int count, int work_cap, int cm_max) {
  const int job = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
  const P2Job J = jobs[job];
  const int nblk = J.nblk;
  const int32_t* bmj = bmax + J.bm_off;
  (void)pbmax;
  __shared__ int s_mink[P2G][P2ENT];
  __shared__ int s_k[P2G][4];   // per test: [2] = some pair can still matter
  __shared__ unsigned s_pmask[P2G][5];  // per (test, component): the rows of the other direction (bit i = row s1 - i) that can still matter
  __shared__ int s_state[8];    // sf, sr, last_fwd, best, status (0 running, 1 ended, 2 more tests than rows), tests done
  __shared__ int s_bp[8];
  __shared__ unsigned long long s_cells;
  __shared__ int s_rmax[2][P2ROWS][5];
  __shared__ int s_work[P2WORK][2];  // the blocks to look at this round: ((test, component) << 20 | block, the block's maximum in the tested row)
  __shared__ int s_nwork, s_ovf;
  __shared__ int s_ka[P2G * 5], s_kb[P2G * 5];  // per scan: the diagonals of the tested row that some row mirrors
  __shared__ int s_cm[P2G * 5][P2CM];           // per scan and block of the other direction: the largest value of the scan's rows there
  for (int i = tid; i < 2 * P2ROWS * 5; i += blockDim.x) ((int*)s_rmax)[i] = p2max[(int64_t)job * 2 * P2ROWS * 5 + i];
  if (tid < 8) s_bp[tid] = 0;
  // (with a bound of the job's score the walk starts as if a breakpoint of score bound + 1 were in hand: see wfa_bp_kernel)
  if (tid == 0) { s_state[0] = J.sf; s_state[1] = J.sr; s_state[2] = J.last_fwd; s_state[3] = min(J.sub < SUB_NONE ? J.sub + 1 : INT32_MAX, J.best0 > 0 ? J.best0 : INT32_MAX); s_state[4] = 0; s_state[5] = 0; s_cells = 0; }
  __syncthreads();
  const int pl = J.pl, tl = J.tl, kinv = tl - pl;
  const int aneed = tl + pl;  // the row / block maxima are antidiagonals (wfa_p2_blockmax_kernel): two cells can only meet when theirs sum to this
  const Rng RG = make_rng(pl, tl, J.sub);
  const int gopen = max(pen.o1, pen.o2);
  const long long t_begin = wall_clock64();
  long long t_cells = 0, t_list = 0, t_pick = 0;
  long long n_items = 0;
  int rounds = 0;
  for (;;) {
    ++rounds;
    // ---- state at the start of the round (uniform)
    const int sf0 = s_state[0], sr0 = s_state[1], lf0 = s_state[2], best = s_state[3], u0 = s_state[5];
    if (s_state[4] != 0) break;
    // test g of the round: the direction that stepped last is tested first, then the two alternate
    auto test_of = [&](int g, int& d0, int& s0, int& s1) {
      const int first = lf0 ? 0 : 1;
      const int a = first == 0 ? sf0 : sr0, b = first == 0 ? sr0 : sf0;
      if ((g & 1) == 0) { d0 = first; s0 = a + g / 2; s1 = b + g / 2; }
      else { d0 = first ^ 1; s0 = b + (g + 1) / 2; s1 = a + (g - 1) / 2; }
    };
    const int ng = min(P2G, P2TESTS - u0);  // tests with rows behind them
    // ---- pairs: the (test, row of the other direction, component) triples that can still matter: per (test, component)
    // a mask over the rows
    for (int i = tid; i < P2G * P2ENT; i += blockDim.x) ((int*)s_mink)[i] = INT32_MAX;
    if (tid < P2G * 5) ((unsigned*)s_pmask)[tid] = 0u;
    if (tid < P2G) s_k[tid][2] = 0;
    __syncthreads();
    for (int e = tid; e < ng * scope * 5; e += blockDim.x) {
      const int g = e / (scope * 5), pr = e % (scope * 5), i = pr / 5, cc = pr % 5;
      int d0, s0, s1;
      test_of(g, d0, s0, s1);
      const int d1 = d0 ^ 1, sd0 = d0 == 0 ? J.sf : J.sr, sd1 = d1 == 0 ? J.sf : J.sr, si = s1 - i;
      if (si >= 0 && s0 + si - pen.o2 < best && s0 + si - bp_gap_open(pen, cc) < best &&
          s_rmax[d0][s0 - (sd0 - 25)][cc] + s_rmax[d1][si - (sd1 - 25)][cc] >= aneed) {
        atomicOr(&s_pmask[g][cc], 1u << i);
        s_k[g][2] = 1;
      }
    }
    __syncthreads();
    const long long t_c0 = wall_clock64();
    // ---- scan.  Per (test, component) the rows of the other direction that can still matter are tested against the tested row
    // block by block (64 diagonals): a block that holds a value large enough for some row is looked at -- first its maximum
    // against the block maxima of every row, one row per lane, then, for the rows that pass, cell by cell, one diagonal per
    // lane, four rows in flight; per (test, row, component) the SMALLEST diagonal on which the offsets meet is kept (atomicMin:
    // the reference's loop only ever takes that one).
    // Round 4: the blocks of all (test, component) scans of the round go through ONE work list that all waves share.  Until
    // then a scan belonged to one wave, which walked its blocks in ascending order and dropped a row at its first hit -- and on
    // repeat-rich records (LPA's KIV-2 copies: every row of a direction that has crossed the text holds tl, no maximum prunes)
    // one scan of a hundred blocks kept one wave busy for milliseconds while fifteen waited at the barrier: a launch of 400 jobs
    // took 3.4 ms for an average walk of 0.1 ms.  A row that has already met on a smaller diagonal is skipped by later blocks.
    // (Pair by pair this was the walk's whole cost in round 2: each of up to 26 x 5 pairs per test paid its own two or three
    // dependent round trips to find that the rows do not share a diagonal yet.)
    {
      if (tid == 0) { s_nwork = 0; s_ovf = 0; }
      __syncthreads();
      const long long t_l0 = wall_clock64();
      // -- list the blocks: one wave per (test, component)
      for (int it = wv; it < ng * 5; it += nw) {
        const int g = it / 5, cc = it % 5;
        const unsigned todo = s_pmask[g][cc];
        if (lane == 0) { s_ka[it] = 1; s_kb[it] = 0; }
        if (!todo) continue;
        int d0, s0, s1;
        test_of(g, d0, s0, s1);
        const int d1 = d0 ^ 1, sd0 = d0 == 0 ? J.sf : J.sr, sd1 = d1 == 0 ? J.sf : J.sr;
        // the diagonals of the tested row that some row of the mask mirrors
        int ka = INT32_MAX, kb = INT32_MIN;
        for (unsigned q = todo; q; q &= q - 1) {
          const int si = s1 - (int)__builtin_ctz(q);
          ka = min(ka, kinv - rng_hi(RG, si)); kb = max(kb, kinv - rng_lo(RG, si));
        }
        ka = max(ka, rng_lo(RG, s0)); kb = min(kb, rng_hi(RG, s0));
        if (ka > kb) continue;
        if (lane == 0) { s_ka[it] = ka; s_kb[it] = kb; }
        const int32_t* bm0 = bmj + ((int64_t)(d0 * P2ROWS + (s0 - (sd0 - 25))) * 5 + cc) * nblk;
        int M1 = 0;  // the largest value any of the rows holds anywhere
        for (unsigned q = todo; q; q &= q - 1) M1 = max(M1, s_rmax[d1][s1 - (int)__builtin_ctz(q) - (sd1 - 25)][cc]);
        // ... and per block of the OTHER direction the largest value any of the rows holds there: a block of the tested row is
        // only listed when the one or two blocks it mirrors can reach tl with it.  (M1 alone lets every block through once a
        // direction has crossed the text anywhere -- small unrelated or repeat-rich problems listed 1100 blocks per round and
        // paid a dependent round trip for each to find that no row passes: 3.4 ms for one job of LPA's, 0.1 ms on average.)
        const bool colmax = nblk <= cm_max;
        if (colmax) {
          for (int b1 = lane; b1 < nblk; b1 += 64) {
            int cm = 0;
            for (unsigned q = todo; q; q &= q - 1) {
              const int si = s1 - (int)__builtin_ctz(q);
              cm = max(cm, bmj[((int64_t)(d1 * P2ROWS + (si - (sd1 - 25))) * 5 + cc) * nblk + b1]);
            }
            s_cm[it][b1] = cm;
          }
          __threadfence_block();  // (written and read by this wave only)
        }
        const int B_lo = (ka + J.koff2) >> 6, B_hi = (kb + J.koff2) >> 6;
        for (int bb = B_lo; bb <= B_hi; bb += 64) {
          const int bl = bb + lane;
          const int bv = bl <= B_hi ? bm0[bl] : 0;
          int other = M1;
          if (colmax && bl <= B_hi) {
            const int k_lo = (bl << 6) - J.koff2, k_hi = k_lo + 63;  // the block's diagonals; their mirrors lie in at most two blocks
            const int b1a = min(max((kinv - k_hi + J.koff2) >> 6, 0), nblk - 1), b1b = min(max((kinv - k_lo + J.koff2) >> 6, 0), nblk - 1);
            other = max(s_cm[it][b1a], s_cm[it][b1b]);
          }
          const bool pass = bl <= B_hi && bv + other >= aneed;
          const unsigned long long m = __ballot(pass);
          if (!m) continue;
          int base = 0;
          if (lane == 0) base = atomicAdd(&s_nwork, (int)__popcll(m));
          base = rdlane(base, 0);
          const int at = base + (int)__popcll(m & ((1ull << lane) - 1ull));
          if (pass) {
            if (at < work_cap) { s_work[at][0] = (it << 20) | bl; s_work[at][1] = bv; }
            else s_ovf = 1;
          }
        }
      }
      __syncthreads();
      t_list += wall_clock64() - t_l0;
      n_items += s_nwork;
      const int nwork = min(s_nwork, work_cap);
      const bool ovf = s_ovf != 0;  // (more blocks than the list holds: the scans that did not fit run the old way below)
      unsigned c_rounds = 0, c_hits = 0;
      // one block against the rows of its scan: which rows' mirrored blocks can meet it, then the cells
      auto do_block = [&](int g, int cc, int b, int v0, int ka, int kb, unsigned todo, int d0, int s0, int s1) -> unsigned {
        const int d1 = d0 ^ 1, sd1 = d1 == 0 ? J.sf : J.sr;
        const int32_t* R0 = p2_row(ring, p2, J, d0, cc, s0);
        const int kb_lo = max(ka, (b << 6) - J.koff2), kb_hi = min(kb, (b << 6) - J.koff2 + 63);
        // (the tested row's cells are asked for before anything depends on them: one round trip less per block)
        const int k0 = (b << 6) - J.koff2 + lane, k1 = kinv - k0;
        const int o0 = (k0 >= kb_lo && k0 <= kb_hi) ? R0[k0] : WF_NULL;
        // rows whose mirrored block(s) can meet this one: one row per lane (a row that has met on a smaller diagonal is done)
        bool pass = false;
        if (lane < scope && ((todo >> lane) & 1u) && s_mink[g][lane * 5 + cc] > kb_lo) {
          const int si = s1 - lane;
          const int32_t* bm1 = bmj + ((int64_t)(d1 * P2ROWS + (si - (sd1 - 25))) * 5 + cc) * nblk;
          // the part of this block the row mirrors, and the row's one or two blocks that hold it
          const int q_lo = max(kb_lo, kinv - rng_hi(RG, si)), q_hi = min(kb_hi, kinv - rng_lo(RG, si));
          if (q_lo <= q_hi) {
            const int b1a = (kinv - q_hi + J.koff2) >> 6, b1b = (kinv - q_lo + J.koff2) >> 6;  // b1a <= b1b <= b1a + 1, inside the row
            pass = v0 + max(bm1[b1a], bm1[b1b]) >= aneed;
          }
        }
        unsigned rows = (unsigned)__ballot(pass);
        if (!rows) return todo;
        // the cells: one diagonal per lane against up to four rows at a time (eight in flight measured slower: the walk of
        // a heavy job is bound by the instruction issue of its 16 waves -- ~30 instructions per row and block -- not by round trips)
        constexpr int RG8 = 4;
        while (rows) {
          int ri[RG8], o1[RG8];
#pragma unroll
          for (int j = 0; j < RG8; ++j) {
            ri[j] = rows ? (int)__builtin_ctz(rows) : -1;
            if (rows) rows &= rows - 1;
          }
#pragma unroll
          for (int j = 0; j < RG8; ++j) {
            o1[j] = WF_NULL;
            if (ri[j] >= 0) {
              const int si = s1 - ri[j];
              if (k1 >= rng_lo(RG, si) && k1 <= rng_hi(RG, si)) o1[j] = p2_row(ring, p2, J, d1, cc, si)[k1];
            }
          }
          ++c_rounds;
#pragma unroll
          for (int j = 0; j < RG8; ++j) {
            if (ri[j] < 0) continue;
            const unsigned long long hh = __ballot(o0 + o1[j] >= tl);  // (a NULL offset is -2^30: the sum stays far below)
            if (hh) {
              if (lane == 0) atomicMin(&s_mink[g][ri[j] * 5 + cc], (b << 6) - J.koff2 + (int)__builtin_ctzll(hh));
              todo &= ~(1u << ri[j]);
              ++c_hits;
            }
          }
        }
        return todo;
      };
      for (int w = wv; w < nwork; w += nw) {
        const int it = s_work[w][0] >> 20, b = s_work[w][0] & 0xfffff, v0 = s_work[w][1];
        const int g = it / 5, cc = it % 5;
        int d0, s0, s1;
        test_of(g, d0, s0, s1);
        (void)do_block(g, cc, b, v0, s_ka[it], s_kb[it], s_pmask[g][cc], d0, s0, s1);
      }
      if (ovf) {
        // the blocks that did not fit the list: scan by scan, a wave each, in ascending order (every block again: the listed
        // ones find their rows done)
        for (int it = wv; it < ng * 5; it += nw) {
          const int g = it / 5, cc = it % 5;
          unsigned todo = s_pmask[g][cc];
          const int ka = s_ka[it], kb = s_kb[it];
          if (!todo || ka > kb) continue;
          int d0, s0, s1;
          test_of(g, d0, s0, s1);
          const int d1 = d0 ^ 1, sd0 = d0 == 0 ? J.sf : J.sr, sd1 = d1 == 0 ? J.sf : J.sr;
          const int32_t* bm0 = bmj + ((int64_t)(d0 * P2ROWS + (s0 - (sd0 - 25))) * 5 + cc) * nblk;
          int M1 = 0;
          for (unsigned q = todo; q; q &= q - 1) M1 = max(M1, s_rmax[d1][s1 - (int)__builtin_ctz(q) - (sd1 - 25)][cc]);
          const int B_lo = (ka + J.koff2) >> 6, B_hi = (kb + J.koff2) >> 6;
          for (int bb = B_lo; bb <= B_hi && todo; bb += 64) {
            const int bl = bb + lane;
            const int bv = bl <= B_hi ? bm0[bl] : 0;
            unsigned long long m = __ballot(bl <= B_hi && bv + M1 >= aneed);
            while (m && todo) {
              const int f = (int)__builtin_ctzll(m);
              m &= m - 1;
              todo = do_block(g, cc, bb + f, rdlane(bv, f), ka, kb, todo, d0, s0, s1);
            }
          }
        }
      }
      if (count == 1 && lane == 0) {  // WFM_P2_COUNT: rounds of cell tests (up to four rows each), pairs that met
        atomicAdd(&g_p2cnt[3], (unsigned long long)c_rounds); atomicAdd(&g_p2cnt[4], (unsigned long long)c_hits);
        atomicMax(&g_p2cnt[6], (unsigned long long)c_rounds);
      }
    }
    __syncthreads();
    t_cells += wall_clock64() - t_c0;
    const long long t_p0 = wall_clock64();
    // ---- pick, test after test.  The reference walks i = 0 .. scope-1 and, inside, D2, I2, D1, I1, M; it takes a hit when its
    // score is STRICTLY below the best so far (and skips ahead once a gap-open class can no longer beat it): with o2 >= o1 >= 0
    // the walk ends on the hit of smallest score, the first in that order among equals -- a minimum over (score, position)
    if (wv == 0) {
      int sf = sf0, sr = sr0, last_fwd = lf0, b = best, status = 0, u = u0;
      unsigned long long cells = 0;
      for (int g = 0; g < P2G; ++g) {
        int d0;  // direction whose newest row is tested, then the OTHER one advances
        if (last_fwd) {
          const int min_sr = (sr > scope - 1) ? sr - (scope - 1) : 0;
          if (sf + min_sr - gopen >= b) { status = 1; break; }
          d0 = 0;
        } else {
          const int min_sf = (sf > scope - 1) ? sf - (scope - 1) : 0;
          if (min_sf + sr - gopen >= b) { status = 1; break; }
          d0 = 1;
        }
        if (u >= P2TESTS) { status = 2; break; }
        const int s0 = d0 == 0 ? sf : sr, s1 = d0 == 0 ? sr : sf;
        if (count && lane == 0) { atomicAdd(&g_p2cnt[0], 1ull); if (s_k[g][2]) atomicAdd(&g_p2cnt[1], 1ull); }
        if (s_k[g][2]) {
          if (pen.o2 >= pen.o1) {
            long long key = INT64_MAX;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const int pr = lane + 64 * q;
              if (pr < scope * 5 && s_mink[g][pr] != INT32_MAX) {
                const int i = pr / 5, cc = pr % 5, si = s1 - i;
                const int sc = s0 + si - bp_gap_open(pen, cc);
                const int oi = cc == C_D2 ? 0 : (cc == C_I2 ? 1 : (cc == C_D1 ? 2 : (cc == C_I1 ? 3 : 4)));
                if (sc < b) key = min(key, ((long long)sc << 16) | (long long)(i * 5 + oi));
              }
            }
#pragma unroll
            for (int dlt = 32; dlt >= 1; dlt >>= 1) {
              const long long o = __shfl_xor(key, dlt, 64);
              key = min(key, o);
            }
            if (key != INT64_MAX) {
              const int pos = (int)(key & 0xffff), i = pos / 5, oi = pos % 5, si = s1 - i;
              const int cc = oi == 0 ? C_D2 : (oi == 1 ? C_I2 : (oi == 2 ? C_D1 : (oi == 3 ? C_I1 : C_M)));
              const int k0 = s_mink[g][i * 5 + cc], k1 = kinv - k0;
              b = (int)(key >> 16);
              if (lane == 0) {
                s_bp[0] = b;
                if (d0 == 0) { s_bp[1] = s0; s_bp[2] = si; s_bp[3] = k0; s_bp[4] = p2_row(ring, p2, J, 0, cc, s0)[k0]; }
                else         { s_bp[1] = si; s_bp[2] = s0; s_bp[3] = k1; s_bp[4] = p2_row(ring, p2, J, 0, cc, si)[k1]; }
                s_bp[5] = cc; s_bp[7] = 1;
              }
            }
          } else {  // unusual penalties: the walk itself (every lane runs it; lane 0 records)
            const int order[5] = {C_D2, C_I2, C_D1, C_I1, C_M};
            for (int i = 0; i < scope; ++i) {
              const int si = s1 - i;
              if (si < 0) break;
              for (int oi = 0; oi < 5; ++oi) {
                const int cc = order[oi];
                const int gop = bp_gap_open(pen, cc);
                if ((oi == 0 || oi == 2 || oi == 4) && s0 + si - gop >= b) break;
                const int k0 = s_mink[g][i * 5 + cc];
                if (k0 == INT32_MAX) continue;
                if (s0 + si - gop >= b) continue;
                const int k1 = kinv - k0;
                b = s0 + si - gop;
                if (lane == 0) {
                  s_bp[0] = b;
                  if (d0 == 0) { s_bp[1] = s0; s_bp[2] = si; s_bp[3] = k0; s_bp[4] = p2_row(ring, p2, J, 0, cc, s0)[k0]; }
                  else         { s_bp[1] = si; s_bp[2] = s0; s_bp[3] = k1; s_bp[4] = p2_row(ring, p2, J, 0, cc, si)[k1]; }
                  s_bp[5] = cc; s_bp[7] = 1;
                }
              }
            }
          }
        }
        // the other direction advances by one row (computed ahead: only the bookkeeping is left)
        if (d0 == 0) { ++sr; cells += (unsigned long long)(rng_hi(RG, sr) - rng_lo(RG, sr) + 1); last_fwd = 0; }
        else         { ++sf; cells += (unsigned long long)(rng_hi(RG, sf) - rng_lo(RG, sf) + 1); last_fwd = 1; }
        ++u;
      }
      if (lane == 0) {
        s_state[0] = sf; s_state[1] = sr; s_state[2] = last_fwd; s_state[3] = b; s_state[4] = status; s_state[5] = u;
        s_cells += cells;
      }
    }
    __syncthreads();
    t_pick += wall_clock64() - t_p0;
  }
  if (tid == 0) {
    BpResult r;
    r.status = s_state[4] == 2 ? WFM_DEV_P2_MORE : 0;
    if (r.status == 0 && !s_bp[7]) r.status = J.best0 > 0 ? WFM_DEV_P2_NOTHING : (J.sub < SUB_NONE ? WFM_DEV_BAND : WFM_DEV_UNREACHABLE);  // the walk ended without a (better) breakpoint
    r.score = s_state[3]; r.score_fwd = s_bp[1]; r.score_rev = s_bp[2]; r.k_fwd = s_bp[3]; r.off_fwd = s_bp[4]; r.comp = s_bp[7] ? s_bp[5] : -1;
    r.steps = s_state[0] + s_state[1];
    r.cells = s_cells;
    r.steps_p1 = J.sf + J.sr;
    r.ticks_p1 = (uint32_t)t_cells; r.ticks_p2 = (uint32_t)(wall_clock64() - t_begin); r.pad_ = rounds;  // diagnostics (WFM_DEBUG)
    r.ticks_list = (uint32_t)t_list; r.ticks_pick = (uint32_t)t_pick; r.work_items = (uint32_t)n_items; r.pad2_ = 0;
    results[job] = r;
  }
}

// ---------------------------------------------------------------------------
// An upper bound of a root's score: the cost of ONE valid global alignment, found greedily
// ---------------------------------------------------------------------------
// The tile kernels only compute the cells from which the end diagonal is still within reach of a known upper bound `sub` of
// the score (Rng).  A BiWFA child is handed its score by its parent; a root only had the caller's guess, which has to allow
// for what the caller cannot know (the divergence estimate of a sketch, +- 0.1 % of 50 kb = 300 points) -- and every point
// of slack is a diagonal on either side of every row: the roots of a pangenome batch ran 1150 diagonals wide where their
// children run 260.  Any alignment's cost is an upper bound of the optimal score, and for the records this matters for (long,
// a few differences per kilobase, offset by their padding) a good one is found by walking: one wave per root extends along
// the diagonal it is on, 512 bases per round trip, and at a difference lets its lanes try the edits side by side -- lane 0 a
// substitution, lanes 1..31 a deletion of that many bases, lanes 32..62 an insertion -- and takes the cheapest edit after which
// 16 bases match (or a sequence ends).  The first diagonal is searched the same way over shifts up to 4096 (the padding of the
// target window).  Where no edit qualifies a few times in a row, or edits come thicker than one per 64 bases, the walk gives up
// (-1): the job keeps the caller's guess.  The bound is rigorous whatever the walk chooses -- it is charged at least the
// gap-affine cost of the ops it spells -- so a root that runs under it cannot fail; the host keeps the retry all the same.
__global__ __launch_bounds__(64) void wfa_bound_kernel(const uint8_t* __restrict__ seq, const BoundJob* __restrict__ jobs,
                                                       int32_t* __restrict__ out, DevPen pen, int njobs) {
  const int i = blockIdx.x, lane = (int)threadIdx.x;
  if (i >= njobs) return;
  const BoundJob J = jobs[i];
  const uint8_t* P = seq + J.p_off;
  const uint8_t* T = seq + J.t_off;
  const int pl = J.pl, tl = J.tl;
  auto gap = [&](int L) -> long long { return L <= 0 ? 0ll : min((long long)pen.o1 + (long long)L * pen.e1, (long long)pen.o2 + (long long)L * pen.e2); };
  auto give_up = [&]() { if (lane == 0) out[i] = -1; };
  if (min(pl, tl) < 256) { give_up(); return; }
  constexpr int PROBE = 24, RUN = 16, LOOK = 32;
  int v = 0, h = 0;
  long long score = 0;
  // ---- the first diagonal: the smallest shift (D: the pattern runs ahead, I: the text) under which PROBE bases agree
  {
    bool found = false;
    const int amax = min(4096, max(pl, tl) - 128);
    for (int q0 = 0; q0 <= 96 && !found; q0 += 48) {
      for (int base = 0; base < amax && !found; base += 32) {
        const bool del = lane < 32;
        const int sh = base + (lane & 31);
        bool ok = false;
        if (del) {
          if (sh + q0 + PROBE <= pl && q0 + PROBE <= tl)
            ok = load8(P + sh + q0) == load8(T + q0) && load8(P + sh + q0 + 8) == load8(T + q0 + 8) && load8(P + sh + q0 + 16) == load8(T + q0 + 16);
        } else if (sh > 0) {
          if (q0 + PROBE <= pl && sh + q0 + PROBE <= tl)
            ok = load8(P + q0) == load8(T + sh + q0) && load8(P + q0 + 8) == load8(T + sh + q0 + 8) && load8(P + q0 + 16) == load8(T + sh + q0 + 16);
        }
        const unsigned long long hit = __ballot(ok);
        if (hit) {
          const unsigned hd = (unsigned)(hit & 0xffffffffull), hi = (unsigned)(hit >> 32);
          const int a = hd ? base + (int)__builtin_ctz(hd) : INT32_MAX, b = hi ? base + (int)__builtin_ctz(hi) : INT32_MAX;
          if (a <= b) { v = a; score += gap(a); } else { h = b; score += gap(b); }
          found = true;
        }
      }
    }
    if (!found) { give_up(); return; }
  }
  const int max_events = 64 + (pl + tl) / 64;
  int events = 0, forced = 0;
  for (;;) {
    const int maxn = min(pl - v, tl - h);
    int n = wave_lce_tail_g(P, T, v, h, 0, maxn, lane == 0 && maxn > 0);
    n = rdlane(n, 0);
    v += n; h += n;
    if (v >= pl || h >= tl) break;
    if (++events > max_events) { give_up(); return; }
    // the edits side by side
    int dv = 0, dh = 0;
    long long cost = 0;
    bool cand = true;
    if (lane == 0) { dv = 1; dh = 1; cost = pen.x; }
    else if (lane < 32) { dv = lane; cost = gap(lane); }
    else if (lane < 63) { dh = lane - 31; cost = gap(lane - 31); }
    else cand = false;
    const int v2 = v + dv, h2 = h + dh;
    cand = cand && v2 <= pl && h2 <= tl;
    int m = 0, mx = 0;
    if (cand) {
      mx = min(min(pl - v2, tl - h2), LOOK);
      m = min(lce_from(P + v2, T + h2, 0, mx), mx);
    }
    const bool pass = cand && (m >= RUN || m == min(pl - v2, tl - h2));
    long long key = pass ? ((cost << 8) | (long long)lane) : INT64_MAX;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) key = min(key, __shfl_xor(key, d, 64));
    if (key != INT64_MAX) {
      const int w = (int)(key & 0xff);
      score += key >> 8;
      v += rdlane(dv, w); h += rdlane(dh, w);
      forced = 0;
    } else {
      if (++forced > 6) { give_up(); return; }
      score += pen.x; ++v; ++h;  // a substitution whatever follows: the next difference is looked at on its own
    }
  }
  score += gap(pl - v) + gap(tl - h);
  if (lane == 0) out[i] = score < (long long)(SUB_NONE - 1) ? (int32_t)score : -1;
}
void launch_bound(const uint8_t* seq, const BoundJob* jobs, int32_t* out, int njobs, DevPen pen, hipStream_t st) {
  hipLaunchKernelGGL(wfa_bound_kernel, dim3(njobs), dim3(64), 0, st, seq, jobs, out, pen, njobs);
}

// reversed copies of pattern and text behind the forward ones, each followed by `pad` zero bytes
__global__ __launch_bounds__(256) void seq_reverse_kernel(uint8_t* __restrict__ seq, const SeqRev* __restrict__ jobs, int pad) {
  const SeqRev J = jobs[blockIdx.x >> 1];
  const bool text = blockIdx.x & 1;
  const uint8_t* src = seq + (text ? J.t_fwd : J.p_fwd);
  uint8_t* dst = seq + (text ? J.t_rev : J.p_rev);
  const int n = text ? J.tlen : J.plen;
  for (int q = threadIdx.x; q < n + pad; q += blockDim.x) dst[q] = q < n ? src[n - 1 - q] : (uint8_t)0;
}
void launch_reverse(uint8_t* seq, const SeqRev* jobs, int njobs, int pad, hipStream_t st) {
  hipLaunchKernelGGL(seq_reverse_kernel, dim3(njobs * 2), dim3(256), 0, st, seq, jobs, pad);
}

#ifdef WFM_PROFILE_SECTIONS
void read_sections(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sec), sizeof(long long) * 8); }
#endif
// host-callable launchers
// ---------------------------------------------------------------------------
// ring_rows: 32 (RING) or 128 (RING_BIG) -- the depth the job rings were laid out with (wfa_host.hip: by the penalties' scope)
void launch_bp(const uint8_t* seq, int32_t* ring, const BpJob* jobs, BpResult* res, int njobs, int threads,
               DevPen pen, int scope, int ring_rows, hipStream_t st) {
  if (ring_rows == RING) hipLaunchKernelGGL(r32::wfa_bp_kernel, dim3(njobs), dim3(threads), 0, st, seq, ring, jobs, res, pen, scope);
  else hipLaunchKernelGGL(r128::wfa_bp_kernel, dim3(njobs), dim3(threads), 0, st, seq, ring, jobs, res, pen, scope);
}
void launch_tile_init(const uint8_t* seq, int32_t* ring, const TileJob* jobs, int32_t* mak0, int njobs, int ring_rows, hipStream_t st) {
  if (ring_rows == RING) hipLaunchKernelGGL(r32::wfa_tile_init_kernel, dim3(njobs * 2), dim3(64), 0, st, seq, ring, jobs, mak0, njobs);
  else hipLaunchKernelGGL(r128::wfa_tile_init_kernel, dim3(njobs * 2), dim3(64), 0, st, seq, ring, jobs, mak0, njobs);
}
void launch_tile(const uint8_t* seq, int32_t* ring, const TileJob* jobs, const TileTask* tasks, int32_t* mak, int ntasks,
                 int threads, int T, int Wt, size_t lds_bytes, DevPen pen, int scope, int ring_rows, hipStream_t st) {
  // the attribute is per device and several devices / host threads share the process: set it on every launch of this
  // (non-default, WFM_TILE_REG=0) form rather than cache it
  if (ring_rows == RING) {
    (void)hipFuncSetAttribute((const void*)r32::wfa_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(r32::wfa_tile_kernel, dim3(ntasks), dim3(threads), lds_bytes, st, seq, ring, jobs, tasks, mak, T, Wt, pen, scope);
  } else {
    (void)hipFuncSetAttribute((const void*)r128::wfa_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(r128::wfa_tile_kernel, dim3(ntasks), dim3(threads), lds_bytes, st, seq, ring, jobs, tasks, mak, T, Wt, pen, scope);
  }
}
void launch_tile_advance(TileJob* jobs, int32_t* mak, int njobs, int T, DevPen pen, int exact, int coarse, int launched, hipStream_t st) {
  hipLaunchKernelGGL(wfa_tile_advance_kernel, dim3(njobs), dim3(64), 0, st, jobs, mak, njobs, T, pen, exact, coarse, launched);
}
void launch_tile_reg(const uint8_t* seq, int32_t* ring, const TileJob* jobs, const TileTask* tasks, int32_t* mak, int ntasks,
                     int threads, int T, int C, bool cut, hipStream_t st) {
  const size_t lds = (size_t)(T + 1) * 4;
  if (C == 4) hipLaunchKernelGGL((wfa_tile_reg_kernel<4, 256, 5, 10, 25, 2, 1>), dim3(ntasks), dim3(threads), lds, st, seq, ring, jobs, tasks, mak, T);
  else if (cut) hipLaunchKernelGGL((wfa_tile_reg_kernel<2, 1024, 5, 10, 25, 2, 1, false, true>), dim3(ntasks), dim3(threads), lds, st, seq, ring, jobs, tasks, mak, T);
  else hipLaunchKernelGGL((wfa_tile_reg_kernel<2, 1024, 5, 10, 25, 2, 1, false, false>), dim3(ntasks), dim3(threads), lds, st, seq, ring, jobs, tasks, mak, T);
}
void launch_tile_p2(const uint8_t* seq, int32_t* ring, const TileJob* jobs, const TileTask* tasks, int ntasks, int threads,
                    int32_t* p2, bool cut, hipStream_t st) {
  const size_t lds = (size_t)(P2K + 1) * 4;
  if (cut) hipLaunchKernelGGL((wfa_tile_reg_kernel<2, 1024, 5, 10, 25, 2, 1, true, true>), dim3(ntasks), dim3(threads), lds, st, seq, ring, jobs, tasks,
                              (int32_t*)nullptr, P2K, p2, (int32_t*)nullptr);
  else hipLaunchKernelGGL((wfa_tile_reg_kernel<2, 1024, 5, 10, 25, 2, 1, true, false>), dim3(ntasks), dim3(threads), lds, st, seq, ring, jobs, tasks,
                          (int32_t*)nullptr, P2K, p2, (int32_t*)nullptr);
}
// One workgroup per (job, direction, component, ring row): row s = sd + P2K - r of the P2 rows into slot s mod RING of the ring;
// what lies outside the row's own range is NULL (the slot held an older, wider or narrower, row).
__global__ __launch_bounds__(256) void wfa_p2_to_ring_kernel(int32_t* __restrict__ ring, const int32_t* __restrict__ p2, const P2Job* __restrict__ jobs) {
  const int job = blockIdx.x / (2 * 5 * RING), rest = blockIdx.x % (2 * 5 * RING), d = rest / (5 * RING), cc = (rest / RING) % 5, r = rest % RING;
  const P2Job J = jobs[job];
  const int sd = d == 0 ? J.sf : J.sr, s = sd + P2K - r;
  const Rng RG = make_rng(J.pl, J.tl, J.sub);
  const int lo = rng_lo(RG, s), hi = rng_hi(RG, s);
  const int32_t* src = p2 + J.p2_off + J.koff2 + ((int64_t)((d * 5 + cc) * P2K + (s - sd - 1))) * J.w2;
  int32_t* dst = ring + J.ring_in + ((int64_t)((d * 5 + cc) * RING + (s & RMASK))) * J.width;
  for (int c = threadIdx.x; c < J.width; c += blockDim.x) {
    const int k = c - J.koff;
    dst[c] = (k >= lo && k <= hi) ? src[k] : WF_NULL;
  }
}
void launch_p2_to_ring(int32_t* ring, const int32_t* p2, const P2Job* jobs, int njobs, hipStream_t st) {
  static_assert(P2K >= RING, "the snapshot after a round of phase 2 is taken from the P2 rows alone");
  hipLaunchKernelGGL(wfa_p2_to_ring_kernel, dim3((unsigned)njobs * 2 * 5 * RING), dim3(256), 0, st, ring, p2, jobs);
}
void launch_p2_blockmax(const int32_t* ring, const int32_t* p2, const P2Job* jobs, int32_t* bmax, int32_t* p2max, int njobs, hipStream_t st) {
  hipLaunchKernelGGL(wfa_p2_blockmax_kernel, dim3(njobs * 2 * P2ROWS), dim3(256), 0, st, ring, p2, jobs, bmax, p2max);
}
void launch_p2_overlap(const int32_t* ring, const int32_t* p2, const P2Job* jobs, const int32_t* p2max, const int32_t* bmax, int32_t* pbmax,
                       BpResult* res, int njobs, int threads, int max_nblk, DevPen pen, int scope, hipStream_t st) {
  static const int count = getenv("WFM_P2_COUNT") ? atoi(getenv("WFM_P2_COUNT")) : 0;
  (void)max_nblk;  // (the walk prunes with each row's own block maxima; the running maxima over the rows are not needed any more)
  static const int work_cap = getenv("WFM_P2_WORKCAP") ? std::max(1, std::min(P2WORK, atoi(getenv("WFM_P2_WORKCAP")))) : P2WORK;  // (tests: a small list forces the overflow path)
  static const int cm_max = getenv("WFM_P2_COLMAX") ? std::max(0, std::min(P2CM, atoi(getenv("WFM_P2_COLMAX")))) : P2CM;  // rows up to this many blocks get column maxima (0: none)
  hipLaunchKernelGGL(wfa_p2_overlap_kernel, dim3(njobs), dim3(threads), 0, st, ring, p2, jobs, p2max, bmax, pbmax, res, pen, scope, count, work_cap, cm_max);
}
void p2_counters(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p2cnt), sizeof(unsigned long long) * 8); }
void launch_base(const uint8_t* seq, int32_t* a32, uint8_t* a8, uint32_t* rle, const BaseJob* jobs, BaseResult* res,
                 int njobs, DevPen pen, bool wide, int ring_rows, hipStream_t st) {
  if (ring_rows == RING) {
    if (wide) hipLaunchKernelGGL(r32::wfa_base_kernel<1024>, dim3(njobs), dim3(1024), 0, st, seq, a32, a8, rle, jobs, res, pen);
    else hipLaunchKernelGGL(r32::wfa_base_kernel<256>, dim3(njobs), dim3(256), 0, st, seq, a32, a8, rle, jobs, res, pen);
  } else {
    if (wide) hipLaunchKernelGGL(r128::wfa_base_kernel<1024>, dim3(njobs), dim3(1024), 0, st, seq, a32, a8, rle, jobs, res, pen);
    else hipLaunchKernelGGL(r128::wfa_base_kernel<256>, dim3(njobs), dim3(256), 0, st, seq, a32, a8, rle, jobs, res, pen);
  }
}
void launch_compact(const uint32_t* rle, const int64_t* off, const int64_t* cap, uint32_t* out, unsigned long long* total,
                    int64_t* out_start, int32_t* out_count, int nprob, hipStream_t st) {
  hipLaunchKernelGGL(rle_compact_kernel, dim3(nprob), dim3(256), 0, st, rle, off, cap, out, total, out_start, out_count);
}

}  // namespace wfm
