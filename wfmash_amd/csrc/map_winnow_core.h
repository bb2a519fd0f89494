// map_winnow_core.h -- the window bookkeeping of addMinmers (commonFunc.hpp:440-708) for one speculative chunk of a
// thinned k-mer stream, written once for two sets of primitives: the wave-wide ones of map_winnow.hip's kernel (sketch,
// pool and occurrence lists in LDS, searched by the 64 lanes side by side) and plain arrays on the host (the model the CPU
// suite holds against the reference's records).  The control flow below is the same in both and is uniform across a
// wave: every lane takes the same branches, only the primitives spread work over the lanes.
//
// What is replayed is host/minmers.cpp's Winnower::advance_sparse with its leave / arrive / maintain steps -- see there
// for why the thinned stream gives the full stream's records.  Two things are put differently, neither observable:
//   * the pool is the SET of window k-mers outside the sketch, kept in (hash, index) order; an entry whose k-mer has left
//     the window is never looked at again (the reference pops such entries lazily when they surface; they are never
//     chosen -- Winnower::maintain);
//   * maintain() is only entered when it can do something: after a sketch entry was erased, after a k-mer went to the
//     pool that beats the sketch's largest hash (or found the sketch short), and before it has ever run.  In between
//     its post-condition -- sketch full and no pool entry below its largest hash, or the pool empty -- cannot break:
//     k-mers only leave the pool's live set or enter it above the sketch;
//   * a k-mer is named by its index in the stream of kept k-mers: it is in the window iff index >= a_lo (the oldest
//     k-mer that has not left yet), and its position is only needed for that oldest one.
// The one place where an expired entry could act in the reference -- a refill that takes more than one hash at a window
// other than the first -- raises F_REFILL here whether or not an expired entry is involved; the sequence then goes to the
// host's winnower, which has the reference's lazy heap.
#ifndef WFM_MAP_WINNOW_CORE_H_
#define WFM_MAP_WINNOW_CORE_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wn {

constexpr uint32_t UNK = 0xFFFFFFFFu;   // start of an interval that was open when the chunk began
constexpr uint32_t SBIT = 0x80000000u;  // stream reference: index | SBIT when the k-mer's strand is -1
constexpr uint32_t NIL = 0xFFFFFFFFu;
enum { F_POOL_FULL = 1, F_OCC_FULL = 2, F_REC_FULL = 4, F_REFILL = 8, F_STATE_FULL = 16, F_MISMATCH = 32, F_UNRESOLVED = 64 };

struct Params {
  int k, w, s;
  int64_t nk;               // k-mer starts of the sequence (len - k + 1)
  int64_t m;                // kept k-mers
  const uint64_t* hash;     // the kept k-mers, ascending positions
  const uint32_t* pos;
  const int8_t* strand;
  int pool_cap, occ_cap, state_words;
};
struct Chunk {
  int64_t warm_from, from, to;  // k-mer starts: warm-up [warm_from, from), the chunk's own [from, to)
  int64_t c0;                   // first kept k-mer with pos >= warm_from
  int64_t c1;                   // first kept k-mer with pos >= from
  int64_t rec_off;              // where the chunk's records go
  int32_t rec_cap, first, last, pad_;
};
struct Rec { uint64_t hash; uint32_t wpos, wend; int32_t tally; uint32_t chunk; };

__host__ __device__ inline uint32_t ref_idx(uint32_t ref) { return ref & ~SBIT; }
__host__ __device__ inline int ref_strand(uint32_t ref) { return (ref & SBIT) ? -1 : 1; }

template <class Ops>
struct Stream {
  Ops& o;
  const int k, w, s;
  const int64_t nk, m;
  uint32_t a_lo, c;  // the window's kept k-mers: stream[a_lo, c); c = the next one to arrive
  bool emit_on;
  bool dirty;        // maintain() has something to do

  __host__ __device__ Stream(Ops& ops, const Params& p, uint32_t c0) : o(ops), k(p.k), w(p.w), s(p.s), nk(p.nk), m(p.m), a_lo(c0), c(c0), emit_on(true), dirty(true) {}

  __host__ __device__ __forceinline__ void emit(uint64_t h, uint32_t wpos, int64_t wend, int tally) {
    if (emit_on) o.emit(h, wpos, (uint32_t)wend, tally);
  }

  // ---- the k-mer that fell out of the window (Winnower::leave) ----
  __host__ __device__ __forceinline__ void leave(int64_t win) {
    if (a_lo >= c) return;
    if ((int64_t)o.fr_pos(a_lo) >= win) return;
    const uint64_t lh = o.fr_hash(a_lo);
    const int ls = o.fr_strand(a_lo);
    const int n = o.sk_n();
    const int r = (n > 0 && lh <= o.sk_max()) ? o.sk_find(lh) : -1;
    if (r >= 0) {
      if (o.sk_cnt(r) == 1) {
        emit(lh, o.sk_wpos(r), win, o.sk_tally(r));
        o.sk_erase(r);
        dirty = true;
      } else {
        const int t = o.sk_tally(r);
        if (t - ls == 0 || t == 0) {  // the tally reaches or leaves zero: the interval is split here
          emit(lh, o.sk_wpos(r), win, t);
          o.sk_set_wpos(r, (uint32_t)win);
        }
        o.sk_set_tally(r, (int)(int16_t)(t - ls));
        if (o.sk_cnt(r) > 0) (void)o.occ_pop(r);
      }
    }
    ++a_lo;
  }

  // ---- a kept k-mer enters the window (Winnower::arrive) ----
  __host__ __device__ __forceinline__ void arrive(int64_t win) {
    const uint64_t h = o.ar_hash(c);
    const int st = o.ar_strand(c);
    const uint32_t ref = c | (st < 0 ? SBIT : 0u);
    const int n = o.sk_n();
    const int r = (n > 0 && h <= o.sk_max()) ? o.sk_find(h) : -1;
    if (r >= 0) {
      o.occ_push(r, ref);
      const int t = o.sk_tally(r);
      if (t + st == 0 || t == 0) {
        emit(h, o.sk_wpos(r), win, t);
        o.sk_set_wpos(r, (uint32_t)win);
      }
      o.sk_set_tally(r, (int)(int16_t)(t + st));
    } else {
      o.pool_push(h, ref, a_lo);
      if (n < s || h < o.sk_max()) dirty = true;
    }
    ++c;
  }

  // does the k-mer behind `ref` start after `win`?  (in the window, and not the one that leaves next if that one starts AT win)
  __host__ __device__ __forceinline__ bool starts_after(uint32_t ref, int64_t win) {
    const uint32_t i = ref_idx(ref);
    if (i < a_lo) return false;
    if (i == a_lo && (int64_t)o.fr_pos(a_lo) == win) return false;
    return true;
  }

  // ---- keep the sketch at the s smallest hashes of the window (Winnower::maintain) ----
  __host__ __device__ __forceinline__ void maintain(int64_t win) {
    uint64_t mh = 0;
    bool have = o.pool_min(a_lo, &mh);
    const int n = o.sk_n();
    if (n > 0 && have && n == s && mh < o.sk_max()) {
      const int r = n - 1;
      const uint64_t out_h = o.sk_hash(r);
      emit(out_h, o.sk_wpos(r), win, o.sk_tally(r));
      while (o.sk_cnt(r) > 0) {
        const uint32_t ref = o.occ_pop(r);
        if (starts_after(ref, win)) o.pool_push(out_h, ref, a_lo);  // strictly after, as the reference (commonFunc.hpp:615)
      }
      o.sk_erase(r);  // the pool's minimum is still mh: what went in is larger
    }
    int iter = 0;
    while (have && o.sk_n() < s) {
      if (iter > 0 && win > 0) o.flag(F_REFILL);
      const int r = o.sk_insert(mh, (uint32_t)win);
      int t = 0;
      uint32_t ref;
      while (o.pool_pop_hash(mh, a_lo, &ref)) {  // ascending positions, as the heap hands them out
        o.occ_push(r, ref);
        t += ref_strand(ref);
      }
      o.sk_set_tally(r, (int)(int16_t)t);
      ++iter;
      have = o.pool_min(a_lo, &mh);
    }
  }

  // the stream for k-mer starts [from, to) (Winnower::advance_sparse): only the iterations in which a kept k-mer arrives
  // or leaves, and the one that completes the first window, do anything
  __host__ __device__ __forceinline__ void advance(int64_t from, int64_t to) {
    const int64_t W = (int64_t)w - k + 1, first_full = (int64_t)w - k;
    const int64_t never = INT64_MAX;
    int64_t i = from;
    for (;;) {
      int64_t ia = never;
      if ((int64_t)c < m) ia = (int64_t)o.ar_pos(c);
      int64_t il = never;
      if (a_lo < c) { il = (int64_t)o.fr_pos(a_lo) + W; if (il < i) il = i; }
      const int64_t i0 = (first_full >= i && first_full >= from) ? first_full : never;
      int64_t nx = ia < il ? ia : il;
      if (i0 < nx) nx = i0;
      i = nx;
      if (i >= to) break;
      const int64_t win = i + k - w;
      leave(win);
      if (ia == i) arrive(win);
      if (win >= 0 && dirty) { maintain(win); dirty = false; }
      ++i;
    }
  }

  // remaining open intervals close at len - k + 1 (commonFunc.hpp:647-658)
  __host__ __device__ __forceinline__ void flush_end() {
    const int n = o.sk_n();
    for (int r = 0; r < n && r < s; ++r) emit(o.sk_hash(r), o.sk_wpos(r), nk, o.sk_tally(r));
  }

  // Everything the stream's future depends on, in a comparable form (Winnower::live_state), as 32-bit words:
  //   [0] words used  [1] a_lo  [2] sketch entries  [3] pool entries
  //   per sketch entry (ascending hashes): hash lo, hash hi, tally (16 bits) | occurrences << 16, the occurrences' references
  //   per pool entry (ascending (hash, index)): hash lo, hash hi, reference
  // The starts of the open intervals are not part of it; they go to wpos[0 .. s).
  __host__ __device__ __forceinline__ void snapshot(uint32_t* st, uint32_t* wpos, int cap) {
    int at = 4;
    const int n = o.sk_n();
    bool full = false;
    for (int r = 0; r < n; ++r) {
      const uint64_t h = o.sk_hash(r);
      const int cnt = o.sk_cnt(r);
      if (at + 3 + cnt > cap) { full = true; break; }
      o.put(st, at, (uint32_t)h); o.put(st, at + 1, (uint32_t)(h >> 32));
      o.put(st, at + 2, ((uint32_t)o.sk_tally(r) & 0xFFFFu) | ((uint32_t)cnt << 16));
      at += 3;
      o.occ_copy(r, st, at);
      at += cnt;
      if (wpos) o.put(wpos, r, o.sk_wpos(r));
    }
    int np = 0;
    for (int q = full ? -1 : o.pool_first(a_lo); q >= 0; q = o.pool_next(q, a_lo)) {
      if (at + 3 > cap) { full = true; break; }
      const uint64_t ph = o.pool_hash(q);
      o.put(st, at, (uint32_t)ph); o.put(st, at + 1, (uint32_t)(ph >> 32)); o.put(st, at + 2, o.pool_ref(q));
      at += 3;
      ++np;
    }
    if (full) o.flag(F_STATE_FULL);
    o.put(st, 0, (uint32_t)at); o.put(st, 1, a_lo); o.put(st, 2, (uint32_t)n); o.put(st, 3, (uint32_t)np);
  }

  // take over the state another chunk reached (its snapshot and interval starts): sketch entries in ascending order, their
  // occurrences, the pool
  __host__ __device__ __forceinline__ void load(const uint32_t* st, const uint32_t* wpos) {
    a_lo = st[1];
    const int n = (int)st[2], np = (int)st[3];
    int at = 4;
    for (int r = 0; r < n; ++r) {
      const uint64_t h = (uint64_t)st[at] | ((uint64_t)st[at + 1] << 32);
      const uint32_t tc = st[at + 2];
      const int cnt = (int)(tc >> 16);
      const int rr = o.sk_insert(h, wpos[r]);
      for (int q = 0; q < cnt; ++q) o.occ_push(rr, st[at + 3 + q]);
      o.sk_set_tally(rr, (int)(int16_t)(tc & 0xFFFFu));
      at += 3 + cnt;
    }
    for (int q = 0; q < np; ++q) {
      const uint64_t h = (uint64_t)st[at] | ((uint64_t)st[at + 1] << 32);
      o.pool_push(h, st[at + 2], a_lo);
      at += 3;
    }
  }

  // the replay after a failed speculation: the chunk once more, from the state its predecessor really reached.  The state
  // it starts from is also what it is compared with from now on (st_begin).
  __host__ __device__ __forceinline__ void run_replay(const Chunk& ch, const uint32_t* prev_end, const uint32_t* prev_wpos, uint32_t* st_begin, uint32_t* st_end,
                                      uint32_t* wpos_end, int cap) {
    load(prev_end, prev_wpos);
    c = (uint32_t)ch.c1;
    const int nw = (int)prev_end[0];
    for (int i = 0; i < nw; ++i) o.put(st_begin, i, prev_end[i]);
    advance(ch.from, ch.to);
    if (ch.last) flush_end();
    else snapshot(st_end, wpos_end, cap);
  }

  // one chunk: warm-up from an empty state two windows before it, then the chunk itself
  __host__ __device__ __forceinline__ void run(const Chunk& ch, uint32_t* st_begin, uint32_t* wpos_begin, uint32_t* st_end, uint32_t* wpos_end, int cap) {
    if (!ch.first) {
      emit_on = false;
      advance(ch.warm_from, ch.from);
      emit_on = true;
      o.sk_all_unknown();
      snapshot(st_begin, wpos_begin, cap);
    }
    advance(ch.from, ch.to);
    if (ch.last) flush_end();
    else snapshot(st_end, wpos_end, cap);
  }
};

}  // namespace wn
#endif
