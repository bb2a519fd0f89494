// minmers.cpp -- winnowed minmer intervals of a target sequence.
//
// Restates CommonFunc::addMinmers (src/map/include/commonFunc.hpp:440-708).  The k-mer
// hashing (two MurmurHash3 per base, the expensive part) runs on the GPU
// (kmer_hash_kernel through wfm_hash_kmers); the sliding-window bookkeeping stays on the host:
// the reference's algorithm is a sequential stream whose lazy heap clean-up, strand-tally
// splits and tie handling shape the output, so it is replayed as it is.
//
// Parallelism.  One sequence per worker is the reference's own (winSketch.hpp:200-239).  On top
// of that a long sequence is cut into chunks that are winnowed SPECULATIVELY in parallel: chunk j
// starts from an empty state two windows before its first k-mer, and what the stream's future
// depends on -- the k-mers still in the window, the sketch with its occurrence lists and strand
// tallies, the live part of the pool, the N counter -- is a function of the last window only,
// EXCEPT after rare bookkeeping anomalies whose effect can outlive a window inside long tandem
// repeats.  So nothing is assumed: at every chunk boundary the live state reached by chunk j-1 is
// compared with the one chunk j started from; on a mismatch chunk j is replayed sequentially from
// chunk j-1's state.  (Expired pool entries are not compared: an expired entry is only ever
// popped, it can never be chosen -- see advance().)  Intervals that are open across a boundary
// get their true start from the previous chunk when the chunks are stitched.
//
// Data movement.  The hashes of a sequence (10 B per base with strands and the normalised bases) stay
// on the device, where the stream is first THINNED (map_prefilter.hip): a k-mer whose hash is above a
// threshold is dropped unless a window may need it, which leaves about one k-mer in eight; the workers
// replay only the kept ones (Winnower::advance_sparse holds the argument why that changes nothing).  A
// streaming thread copies them chunk by chunk into a small ring of pinned slots (asynchronous, link
// speed), a worker takes a slot's bytes into its own reused buffer and winnows from there.  The host never
// holds whole-sequence arrays.  A replay fetches its chunk again.  One stream per sequence
// (wfm_add_minmers, threads == 1) stays dense: the tests hold the two forms against each other.
//
// State (names follow the roles, not the reference's identifiers):
//   arrivals  every valid k-mer still inside (or lingering behind) the window, arrival order
//   sketch    ordered map hash -> open interval + occurrences: the <= s smallest hashes
//   pool      lazy min-heap (hash, pos) of window k-mers that are not in the sketch
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/wfmash_hip.h"
#include "../csrc/dev_cache.h"
#include "../csrc/map_device.h"
#include "../csrc/wfa_handle.h"

namespace {

inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return k;
}
// MurmurHash3_x64_128 low word (src/common/murmur3.h:226-302); host copy for the few k-mers
// the device kernel deliberately does not hash (see below).
uint64_t murmur_lo(const uint8_t* d, int len, uint32_t seed) {
  uint64_t h1 = seed, h2 = seed;
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  const int nb = len / 16;
  for (int i = 0; i < nb; ++i) {
    uint64_t k1, k2;
    memcpy(&k1, d + 16 * i, 8); memcpy(&k2, d + 16 * i + 8, 8);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t* t = d + nb * 16;
  const int r = len & 15;
  uint64_t k1 = 0, k2 = 0;
  for (int i = r - 1; i >= 8; --i) k2 ^= (uint64_t)t[i] << (8 * (i - 8));
  if (r > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
  for (int i = std::min(r, 8) - 1; i >= 0; --i) k1 ^= (uint64_t)t[i] << (8 * i);
  if (r > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
  h1 += h2; h2 += h1; h1 = fmix64(h1); h2 = fmix64(h2);
  return h1 + h2;
}

struct Occ {
  int64_t pos;
  int16_t strand;
  bool operator==(const Occ& o) const { return pos == o.pos && strand == o.strand; }
};
struct PoolItem { uint64_t hash; int64_t pos; int16_t strand; };
struct Open { wfm_minmer_t mi; std::deque<Occ> occ; };

inline bool pool_after(const PoolItem& a, const PoolItem& b) {  // min-heap on (hash, pos)
  return std::tie(a.hash, a.pos) > std::tie(b.hash, b.pos);
}

constexpr int64_t kUnknownStart = std::numeric_limits<int64_t>::min();  // start of an interval opened before the chunk

// What a stream reads of its sequence: k-mer starts [kmer0, ...) and bases [base0, ...).
// norm: upper-cased / N-masked bases; hash/strand: canonical hash and strand per k-mer start as
// wfm_hash_kmers returns them (strand 0 = contains N or palindromic).
struct Slice {
  const uint64_t* hash = nullptr;
  const int8_t* strand = nullptr;
  const char* norm = nullptr;
  int64_t kmer0 = 0, base0 = 0;
};

class Winnower {
 public:
  Winnower(const Slice& d, int64_t len, int k, int w, int s, int32_t seq_id)
      : d_(d), len_(len), k_(k), w_(w), s_(s), seq_id_(seq_id), rc_((size_t)k) {}

  std::vector<wfm_minmer_t> out;  // raw interval records in emission order

  // the stream for k-mer starts [from, to)
  void advance(int64_t from, int64_t to) {
    if (sp_.pos) { advance_sparse(from, to); return; }
    const int k = k_, w = w_;
    const int64_t k0 = d_.kmer0, b0 = d_.base0;
    const char* norm = d_.norm;
    for (int64_t i = from; i < to; ++i) {
      const char* seq = norm + (i - b0);  // the k-mer's bases: seq[0..k)
      const int64_t win = i + k - w;  // id of the window that ends with this k-mer
      tidy_pool(win);
      // canonical hash: from the device, except k-mers that contain an N the reference does not notice
      // (NOTE: no initial scan, commonFunc.hpp:473: an N inside the first k-1 bases of the sequence
      //  is only seen when it is the LAST base of a k-mer)
      uint64_t hf_min; int16_t strand; bool asym;
      if (d_.strand[i - k0] != 0) { hf_min = d_.hash[i - k0]; strand = d_.strand[i - k0]; asym = true; }
      else asym = unnoticed_n_kmer(seq, &hf_min, &strand);
      leave(win);
      if (seq[k - 1] == 'N') ambig_ = k;
      if (asym && ambig_ == 0) arrive(i, win, hf_min, strand);
      if (ambig_ > 0) --ambig_;
      if (win >= 0) maintain(win);
    }
  }

  // remaining open intervals close at len - k + 1 (commonFunc.hpp:647-658)
  void flush_end() {
    uint64_t rank = 1;
    for (auto it = sketch_.begin(); it != sketch_.end() && rank <= (uint64_t)s_; ++it, ++rank) {
      if (it->second.mi.wpos != -1) {
        it->second.mi.wpos_end = len_ - k_ + 1;
        out.push_back(it->second.mi);
      }
    }
  }

  // ---- speculative chunks ----
  // forget where the currently open intervals started (they started before this chunk)
  void mark_open_unknown() {
    for (auto& e : sketch_) e.second.mi.wpos = kUnknownStart;
  }
  // everything the stream's future depends on at k-mer start `next`, in a comparable form
  struct Live {
    int ambig;
    std::vector<std::tuple<uint64_t, int16_t, int64_t>> arrivals;
    std::vector<std::tuple<uint64_t, int16_t, std::vector<Occ>>> sketch;  // hash, tally, occurrences
    std::vector<std::tuple<uint64_t, int64_t, int16_t>> pool;             // live entries, sorted
    bool operator==(const Live& o) const { return ambig == o.ambig && arrivals == o.arrivals && sketch == o.sketch && pool == o.pool; }
  };
  Live live_state(int64_t next) const {
    Live L;
    L.ambig = ambig_;
    L.arrivals.assign(arrivals_.begin(), arrivals_.end());
    for (const auto& e : sketch_) L.sketch.emplace_back(e.first, e.second.mi.strand, std::vector<Occ>(e.second.occ.begin(), e.second.occ.end()));
    const int64_t win = next + k_ - w_;
    for (const auto& p : pool_)
      if (p.pos >= win) L.pool.emplace_back(p.hash, p.pos, p.strand);
    std::sort(L.pool.begin(), L.pool.end());
    return L;
  }
  // true start of the interval of `hash` that is open in this state (kUnknownStart if none)
  int64_t open_start(uint64_t hash) const {
    auto it = sketch_.find(hash);
    return it == sketch_.end() ? kUnknownStart : it->second.mi.wpos;
  }
  // give records and still-open intervals that began before this chunk their start from `prev`,
  // the state the stream was in when this chunk began
  void resolve_unknown_starts(const Winnower& prev) {
    for (auto& m : out)
      if (m.wpos == kUnknownStart) m.wpos = prev.open_start(m.hash);
    for (auto& e : sketch_)
      if (e.second.mi.wpos == kUnknownStart) e.second.mi.wpos = prev.open_start(e.first);
  }
  // continue from another stream's state (the replay after a failed speculation)
  void take_state(const Winnower& o) {
    arrivals_ = o.arrivals_; sketch_ = o.sketch_; pool_ = o.pool_; ambig_ = o.ambig_;
  }

 private:
  // a k-mer the device marks invalid (strand 0): palindromic, or it contains an N.  The reference hashes the
  // ones whose N it has not noticed; whether the k-mer then enters the stream is the ambiguity counter's call.
  bool unnoticed_n_kmer(const char* seq, uint64_t* hf_min, int16_t* strand) {
    const int k = k_;
    bool has_n = false;
    for (int j = 0; j < k; ++j) has_n |= seq[j] == 'N';
    if (!has_n) { *hf_min = 0; *strand = 0; return false; }  // hashFwd == hashBwd
    for (int j = 0; j < k; ++j) {
      const char c = seq[j];
      rc_[(size_t)(k - 1 - j)] = (uint8_t)(c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c);
    }
    const uint64_t hf = murmur_lo((const uint8_t*)seq, k, 42u), hb = murmur_lo(rc_.data(), k, 42u);
    *hf_min = std::min(hf, hb); *strand = hf < hb ? 1 : -1;
    return hf != hb;
  }
  void tidy_pool(int64_t win) {
    if (pool_.size() > (size_t)2 * (size_t)w_) {
      pool_.erase(std::remove_if(pool_.begin(), pool_.end(), [win](const PoolItem& p) { return p.pos < win; }), pool_.end());
      std::make_heap(pool_.begin(), pool_.end(), pool_after);
    }
  }
  // ---- the k-mer that fell out of the window (one per iteration) ----
  void leave(int64_t win) {
    if (!arrivals_.empty() && std::get<2>(arrivals_.front()) < win) {
      const uint64_t lh = std::get<0>(arrivals_.front());
      const int16_t ls = std::get<1>(arrivals_.front());
      if (!sketch_.empty() && lh <= std::prev(sketch_.end())->first) {
        auto it = sketch_.find(lh);
        if (it != sketch_.end()) {
          Open& o = it->second;
          if (o.occ.size() == 1) {
            o.mi.wpos_end = win;
            out.push_back(o.mi);
            sketch_.erase(it);
          } else {
            if (o.mi.strand - ls == 0 || o.mi.strand == 0) {  // tally reaches or leaves zero: split the interval
              o.mi.wpos_end = win;
              out.push_back(o.mi);
              o.mi.wpos = win;
              o.mi.wpos_end = -1;
            }
            o.mi.strand = (int16_t)(o.mi.strand - ls);
            if (!o.occ.empty()) o.occ.pop_front();
          }
        }
      }
      arrivals_.pop_front();
    }
  }
  // ---- a valid k-mer enters the window ----
  void arrive(int64_t i, int64_t win, uint64_t hf_min, int16_t strand) {
    arrivals_.emplace_back(hf_min, strand, i);
    auto it = sketch_.find(hf_min);
    if (it != sketch_.end()) {
      Open& o = it->second;
      o.occ.push_back(Occ{i, strand});
      if (o.mi.strand + strand == 0 || o.mi.strand == 0) {
        o.mi.wpos_end = win;
        out.push_back(o.mi);
        o.mi.wpos = win;
        o.mi.wpos_end = -1;
      }
      o.mi.strand = (int16_t)(o.mi.strand + strand);
    } else {
      pool_.push_back(PoolItem{hf_min, i, strand});
      std::push_heap(pool_.begin(), pool_.end(), pool_after);
    }
  }
  // ---- keep the sketch at the s smallest hashes of the window ----
  // Expired pool entries never get chosen: the loop below clears them off the top, after it the
  // top is live, one step frees at most one sketch slot (one k-mer leaves per step; the swap
  // against the largest needs a full sketch), and a sketch with two or more free slots means
  // the pool ran empty earlier -- so the refill takes the live top and stops, or empties the pool.
  void maintain(int64_t win) {
    const size_t s = (size_t)s_;
    while (!pool_.empty() && pool_.front().pos < win) { std::pop_heap(pool_.begin(), pool_.end(), pool_after); pool_.pop_back(); }
    if (!sketch_.empty() && !pool_.empty() && sketch_.size() == s && pool_.front().hash < std::prev(sketch_.end())->first) {
      auto last = std::prev(sketch_.end());
      last->second.mi.wpos_end = win;
      out.push_back(last->second.mi);
      for (const Occ& oc : last->second.occ) {
        if (oc.pos > win) {  // strictly greater, as the reference (commonFunc.hpp:615)
          pool_.push_back(PoolItem{last->first, oc.pos, oc.strand});
          std::push_heap(pool_.begin(), pool_.end(), pool_after);
        }
      }
      sketch_.erase(last);
    }
    while (!pool_.empty() && sketch_.size() < s) {
      if (pool_.front().pos < win) {  // drops ONE expired item, then takes whatever is on top (commonFunc.hpp:627-633)
        ++expired_in_refill_;         // never expected (see above); chunked runs fall back to one stream if it happens
        std::pop_heap(pool_.begin(), pool_.end(), pool_after);
        pool_.pop_back();
        if (pool_.empty()) break;  // the reference reads an empty heap here (undefined); stop instead
      }
      const PoolItem top = pool_.front();
      Open& o = sketch_[top.hash];
      o.mi = wfm_minmer_t{top.hash, win, -1, seq_id_, 0, 0};
      while (!pool_.empty() && pool_.front().hash == top.hash) {
        o.occ.push_back(Occ{pool_.front().pos, pool_.front().strand});
        o.mi.strand = (int16_t)(o.mi.strand + pool_.front().strand);
        std::pop_heap(pool_.begin(), pool_.end(), pool_after);
        pool_.pop_back();
      }
    }
  }

  // The thinned stream (map_prefilter.hip): only the kept k-mers are known, each one valid.  What the full
  // stream does in an iteration whose arriving and leaving k-mers were both dropped is nothing:
  //  * a dropped k-mer x has hash > tau and every window that holds it has >= s distinct hashes <= tau, all of
  //    them kept.  By induction over the steps the sketch of such a window is full and <= tau after maintain():
  //    x is never found in it (arrive/leave only touch x's own hash), never beats its largest entry (the swap),
  //    and a refill -- one free slot, after a kept k-mer left -- takes the pool's live minimum, which is kept;
  //  * with the sketch full and the pool's live minimum >= its largest entry (true after every maintain(),
  //    since one step adds at most one small k-mer to the pool and the swap/refill takes exactly that one),
  //    maintain() only pops expired entries off the top, which the next call would do just as well;
  //  * windows with fewer fresh candidates keep ALL their k-mers, so there both streams hold the same live set.
  // So only three kinds of iterations are run: a kept k-mer arrives, a kept k-mer leaves (W steps after it
  // arrived), and the one where the first window completes.  Expired pool entries differ between the two
  // streams (dropped k-mers are never pooled); they are unobservable unless one takes part in a refill, which
  // the counter catches as it does for the chunks -- the fall-back is then the FULL single stream.
  void advance_sparse(int64_t from, int64_t to) {
    const int64_t W = (int64_t)w_ - k_ + 1;  // a k-mer that arrived at p leaves at p + W
    const int64_t first_full = (int64_t)w_ - k_;  // win == 0
    size_t c = (size_t)(std::lower_bound(sp_.pos, sp_.pos + sp_.n, from, [](uint32_t p, int64_t x) { return (int64_t)p < x; }) - sp_.pos);
    size_t e = (size_t)(std::lower_bound(extra_.begin(), extra_.end(), from, [](const PoolItem& p, int64_t x) { return p.pos < x; }) - extra_.begin());
    const int64_t never = std::numeric_limits<int64_t>::max();
    int64_t i = from;
    for (;;) {
      const int64_t ia = c < sp_.n ? (int64_t)sp_.pos[c] : never;
      const int64_t ie = e < extra_.size() ? extra_[e].pos : never;
      const int64_t il = arrivals_.empty() ? never : std::max(i, std::get<2>(arrivals_.front()) + W);
      const int64_t i0 = (first_full >= i && first_full >= from) ? first_full : never;
      i = std::min(std::min(ia, ie), std::min(il, i0));
      if (i >= to) break;
      const int64_t win = i + k_ - w_;
      tidy_pool(win);
      leave(win);
      if (ie == i) { arrive(i, win, extra_[e].hash, extra_[e].strand); ++e; }
      else if (ia == i) { arrive(i, win, sp_.hash[c], sp_.strand[c]); ++c; }
      if (win >= 0) maintain(win);
      ++i;
    }
  }

 public:
  // the kept k-mers of the stream (ascending positions)
  struct Sparse { const uint32_t* pos = nullptr; const uint64_t* hash = nullptr; const int8_t* strand = nullptr; size_t n = 0; };
  // Thinned form.  The k-mers the device cannot judge -- an N among the first k-1 bases of the sequence is not
  // noticed by the reference (see advance()) -- are hashed here from `head` (the first 2k bases, or all).
  Winnower(const Sparse& sp, const char* head, int64_t head_len, int64_t len, int k, int w, int s, int32_t seq_id)
      : len_(len), k_(k), w_(w), s_(s), seq_id_(seq_id), rc_((size_t)k), sp_(sp) {
    for (int64_t i = 0; i < k - 1 && i + k <= head_len; ++i) {
      bool late_n = false;  // an N the reference notices: at base k-1 or later
      for (int64_t b = std::max<int64_t>(i, k - 1); b < i + k; ++b) late_n |= head[b] == 'N';
      uint64_t hf; int16_t st;
      if (!late_n && unnoticed_n_kmer(head + i, &hf, &st)) extra_.push_back(PoolItem{hf, i, st});
    }
  }

 private:
  Slice d_;
  int64_t len_;
  int k_, w_, s_;
  int32_t seq_id_;
  std::deque<std::tuple<uint64_t, int16_t, int64_t>> arrivals_;
  std::map<uint64_t, Open> sketch_;
  std::vector<PoolItem> pool_;
  std::vector<uint8_t> rc_;
  int ambig_ = 0;
  int64_t expired_in_refill_ = 0;
  Sparse sp_;
  std::vector<PoolItem> extra_;  // thinned form: the unnoticed-N k-mers at the start of the sequence

 public:
  int64_t expired_in_refill() const { return expired_in_refill_; }
};

// std::sort, run by several threads, with std::sort's result to the last tie.
// The reference orders a sequence's records by (wpos, wpos_end) alone (commonFunc.hpp:696), so where records tie
// their order is whatever libstdc++'s introsort leaves -- and that order is part of the output.  The same routine
// is therefore run here, only its independent halves side by side: introsort partitions, recurses into the right
// part and loops on the left; the two parts never touch each other's elements, so handing the right part to
// another thread changes nothing, and the closing insertion pass is the library's own.
#if defined(__GLIBCXX__)
template <typename It, typename Cmp>
void introsort_loop_spread(It first, It last, long depth_limit, Cmp comp, long spawn_above) {
  std::vector<std::thread> helpers;
  while (last - first > (long)std::_S_threshold) {
    if (depth_limit == 0) { std::__partial_sort(first, last, last, comp); break; }
    --depth_limit;
    It cut = std::__unguarded_partition_pivot(first, last, comp);
    if (last - cut > spawn_above) helpers.emplace_back([=] { introsort_loop_spread(cut, last, depth_limit, comp, spawn_above); });
    else std::__introsort_loop(cut, last, depth_limit, comp);
    last = cut;
  }
  for (auto& t : helpers) t.join();
}
template <typename It, typename Compare>
void sort_as_std(It first, It last, Compare comp, int threads) {
  if (first == last) return;
  const long n = last - first;
  if (threads <= 1 || n < ((long)1 << 18)) { std::sort(first, last, comp); return; }
  auto c = __gnu_cxx::__ops::__iter_comp_iter(comp);
  introsort_loop_spread(first, last, std::__lg(n) * 2, c, std::max<long>(n / (2 * (long)threads), (long)1 << 15));
  std::__final_insertion_sort(first, last, c);
}
#else
template <typename It, typename Compare>
void sort_as_std(It first, It last, Compare comp, int) { std::sort(first, last, comp); }
#endif

// The records of one sequence: one allocation of the final size, 2 MB aligned and marked for huge pages (a
// chromosome's records are hundreds of MB; with 4 kB pages the first touch of such an array is mostly page faults).
struct RecBuf {
  wfm_minmer_t* p = nullptr;
  size_t n = 0;
  RecBuf() = default;
  RecBuf(const RecBuf&) = delete;
  RecBuf& operator=(const RecBuf&) = delete;
  ~RecBuf() { release(); }
  void release() { free(p); p = nullptr; n = 0; }
  void allocate(size_t count) {
    release();
    if (!count) return;
    const size_t huge = (size_t)2 << 20, bytes = (count * sizeof(wfm_minmer_t) + huge - 1) / huge * huge;
    p = static_cast<wfm_minmer_t*>(aligned_alloc(huge, bytes));
    if (!p) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
    (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
    n = count;
  }
  void assign(const std::vector<wfm_minmer_t>& v) { allocate(v.size()); if (n) memcpy(p, v.data(), n * sizeof(wfm_minmer_t)); }
  size_t size() const { return n; }
  const wfm_minmer_t* data() const { return p; }
  const wfm_minmer_t& operator[](size_t i) const { return p[i]; }
};

inline bool by_window(const wfm_minmer_t& l, const wfm_minmer_t& r) { return std::tie(l.wpos, l.wpos_end) < std::tie(r.wpos, r.wpos_end); }
inline bool dropped_record(const wfm_minmer_t& m) { return m.wpos < 0 || m.wpos_end < 0 || m.wpos == m.wpos_end; }
inline int16_t strand_sign(int16_t tally) { return tally < 0 ? (int16_t)-1 : (int16_t)1; }  // every non-negative tally (0 included) reads FWD (commonFunc.hpp:672)
inline int pieces_of(const wfm_minmer_t& m, int w) { return (int)std::ceil(float(m.wpos_end - m.wpos) / float(w)); }

// strand sign, chunks of at most w windows, order, de-duplication (commonFunc.hpp:660-706)
void finish_records(std::vector<wfm_minmer_t>& out, int w, int threads = 1) {
  out.erase(std::remove_if(out.begin(), out.end(), dropped_record), out.end());
  std::vector<wfm_minmer_t> chunks;
  for (auto& m : out) {
    m.strand = strand_sign(m.strand);
    if (m.wpos_end > m.wpos + w) {
      const int n = pieces_of(m, w);
      for (int c = 0; c < n; ++c)
        chunks.push_back(wfm_minmer_t{m.hash, m.wpos + (int64_t)c * w, std::min(m.wpos + (int64_t)c * w + w, m.wpos_end), m.seqId, m.strand, 0});
    }
  }
  out.erase(std::remove_if(out.begin(), out.end(), [w](const wfm_minmer_t& m) { return m.wpos_end - m.wpos > w; }), out.end());
  out.insert(out.end(), chunks.begin(), chunks.end());
  sort_as_std(out.begin(), out.end(), by_window, threads);
  out.erase(std::unique(out.begin(), out.end(), [](const wfm_minmer_t& l, const wfm_minmer_t& r) { return l.wpos == r.wpos && l.hash == r.hash; }), out.end());
}

// finish_records for records that sit in per-chunk lists (emission order = list after list): the array std::sort
// sees -- the records of at most w windows in emission order, then the pieces of the longer ones in emission order
// -- is laid out at its final size and filled by `threads` threads, each list into its own place.
void finish_lists(const std::vector<const std::vector<wfm_minmer_t>*>& lists, int w, int threads, RecBuf& out, double* ms = nullptr) {
  const size_t nl = lists.size();
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  std::vector<size_t> n_short(nl + 1, 0), n_piece(nl + 1, 0);
  auto spread = [&](auto&& fn) {
    std::atomic<size_t> next{0};
    auto work = [&] { for (size_t j; (j = next.fetch_add(1)) < nl;) fn(j); };
    std::vector<std::thread> pool;
    for (int t = 1; t < std::min<int>(std::max(1, threads), (int)nl); ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
  };
  spread([&](size_t j) {
    size_t a = 0, b = 0;
    for (const wfm_minmer_t& m : *lists[j]) {
      if (dropped_record(m)) continue;
      if (m.wpos_end > m.wpos + w) b += (size_t)pieces_of(m, w); else ++a;
    }
    n_short[j + 1] = a; n_piece[j + 1] = b;
  });
  for (size_t j = 0; j < nl; ++j) { n_short[j + 1] += n_short[j]; n_piece[j + 1] += n_piece[j]; }
  const size_t total_short = n_short[nl], total = total_short + n_piece[nl];
  out.allocate(total);
  const double t1 = now();
  spread([&](size_t j) {
    wfm_minmer_t* a = out.p + n_short[j];
    wfm_minmer_t* b = out.p + total_short + n_piece[j];
    for (const wfm_minmer_t& m : *lists[j]) {
      if (dropped_record(m)) continue;
      const int16_t st = strand_sign(m.strand);
      if (m.wpos_end > m.wpos + w) {
        const int n = pieces_of(m, w);
        for (int c = 0; c < n; ++c)
          *b++ = wfm_minmer_t{m.hash, m.wpos + (int64_t)c * w, std::min(m.wpos + (int64_t)c * w + w, m.wpos_end), m.seqId, st, 0};
      } else {
        *a = m; a->strand = st; ++a;
      }
    }
  });
  const double t2 = now();
  sort_as_std(out.p, out.p + total, by_window, threads);
  const double t3 = now();
  out.n = (size_t)(std::unique(out.p, out.p + total, [](const wfm_minmer_t& l, const wfm_minmer_t& r) { return l.wpos == r.wpos && l.hash == r.hash; }) - out.p);
  if (ms) { ms[0] = t1 - t0; ms[1] = t2 - t1; ms[2] = t3 - t2; ms[3] = now() - t3; }
}

void normalise(char* p, int64_t n) {  // makeUpperCaseAndValidDNA (commonFunc.hpp:132-142)
  for (int64_t i = 0; i < n; ++i) {
    char c = p[i];
    if (c > 96 && c < 123) c -= 32;
    p[i] = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'N';
  }
}

// the whole sequence as one stream
void winnow(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id, const uint64_t* hash, const int8_t* strand,
            std::vector<wfm_minmer_t>& out) {
  Winnower W(Slice{hash, strand, seq, 0, 0}, len, k, w, s, seq_id);
  W.advance(0, len - k + 1);
  W.flush_end();
  out = std::move(W.out);
  finish_records(out, w);
}

// ---- one sequence cut into speculative chunks ----
struct SeqJob {
  int64_t idx = 0;           // position in the caller's list
  int32_t seq_id = 0;
  int64_t len = 0, nk = 0;
  int k = 0, w = 0, s = 0;
  // whole-sequence host arrays: only when the sequence is winnowed from ordinary memory (one stream
  // per sequence, the test hooks, the fall-back in stitch()); the streamed form never holds them
  std::unique_ptr<char[]> norm;
  std::unique_ptr<uint64_t[]> hash;
  std::unique_ptr<int8_t[]> strand;
  std::vector<int64_t> bounds;                          // chunk j covers k-mer starts [bounds[j], bounds[j+1])
  std::vector<std::unique_ptr<Winnower>> chunk;         // state at bounds[j+1], records of the chunk
  std::vector<Winnower::Live> started_from;             // live state chunk j began with (j >= 1)
  std::atomic<int> pending{0};
  int replays = 0;
  int fetch_rc = WFM_OK;
  std::atomic<bool> stitched{false};   // result is final
  int sort_threads = 1;                // for the closing sort of stitch()
  double ms_stitch = 0;
  double ms_parts[6] = {0, 0, 0, 0, 0, 0};  // compare, count, fill, sort, unique, release
  RecBuf result;
  MapHashedSeq dev;          // the hashed sequence on the device: the source of every slice
  bool on_device = false;
  // thinned form (map_prefilter.hip): the kept k-mers, on the device; chunk j reads [cidx_warm[j], cidx[j+1])
  bool thinned = false;
  MapSparseSeq sparse;
  std::string head;                      // the first normalised bases: the k-mers the device cannot judge (Winnower)
  std::vector<int64_t> cidx, cidx_warm;  // kept k-mers before bounds[j] / before warm_from(j)
  const char* raw = nullptr;             // the caller's sequence, handle and GPU lock: to hash it again if the
  wfm_handle_t* handle = nullptr;        //   dense stream is needed after all (stitch()'s fall-back)
  std::mutex* gpu_mu = nullptr;
  std::vector<wfm_minmer_t> dev_raw;     // winnowed on the device (map_winnow.hip): raw records in emission order, to be finished
  bool dev_winnowed = false;
  bool for_device = false;               // thinned, and nothing that keeps it from the device winnower
  wfm_minmer_t* d_result = nullptr;      // winnowed AND finished on the device (map_finish.hip): the sequence's records, on the device
  int64_t n_result = 0;
  std::vector<uint32_t> h_pos;           // test hook: the kept k-mers in host memory
  std::vector<uint64_t> h_hash;
  std::vector<int8_t> h_strand;

  // what a stream reads: a dense slice or kept k-mers
  struct View { Slice d; Winnower::Sparse sp; };
  std::unique_ptr<Winnower> make(const View& v) const {
    if (thinned && v.sp.pos) return std::make_unique<Winnower>(v.sp, head.data(), (int64_t)head.size(), len, k, w, s, seq_id);
    return std::make_unique<Winnower>(v.d, len, k, w, s, seq_id);
  }
  // the packed kept-k-mer layout of map_device.h in `buf`
  static Winnower::Sparse packed_sparse(const char* buf, size_t mc) {
    Winnower::Sparse sp;
    sp.hash = reinterpret_cast<const uint64_t*>(buf);
    sp.pos = reinterpret_cast<const uint32_t*>(buf + mc * 8);
    sp.strand = reinterpret_cast<const int8_t*>(buf + mc * 12);
    sp.n = mc;
    return sp;
  }

  Slice whole() const { return Slice{hash.get(), strand.get(), norm.get(), 0, 0}; }
  int64_t warm_from(size_t j) const { return j > 0 ? std::max<int64_t>(0, bounds[j] - 2 * (int64_t)w) : 0; }
  // what chunk j reads: k-mer starts [kf, kt) (two windows of warm-up first) and their bases [kf, bt)
  void chunk_range(size_t j, int64_t* kf, int64_t* kt, int64_t* bt) const {
    *kf = warm_from(j); *kt = bounds[j + 1]; *bt = bounds[j + 1] + k - 1;
  }

  void plan(int64_t chunk_len) {
    bounds.assign(1, 0);
    // a chunk must dwarf its two-window warm-up; no chunk is longer than 1.5 x chunk_len
    if (chunk_len >= 64 * (int64_t)w && nk > chunk_len + chunk_len / 2)
      for (int64_t b = chunk_len; b + chunk_len / 2 < nk; b += chunk_len) bounds.push_back(b);
    bounds.push_back(nk);
    const size_t n = bounds.size() - 1;
    chunk.resize(n);
    started_from.resize(n);
    pending.store((int)n);
  }
  void run_chunk(size_t j, const View& v) {
    auto W = make(v);
    if (j > 0) {
      W->advance(warm_from(j), bounds[j]);  // warm-up: records are not this chunk's
      W->out.clear();
      W->mark_open_unknown();
      started_from[j] = W->live_state(bounds[j]);
    }
    W->advance(bounds[j], bounds[j + 1]);
    chunk[j] = std::move(W);
  }
  // the packed slice layout of map_device.h in `buf`
  static Slice packed(const char* buf, int64_t kf, int64_t kt) {
    const size_t n = (size_t)(kt - kf);
    return Slice{reinterpret_cast<const uint64_t*>(buf), reinterpret_cast<const int8_t*>(buf + n * 8), buf + n * 9, kf, kf};
  }
  // k-mer starts [kf, kt) with their bases, from the host arrays or (streamed form) from the device
  Slice refetch(int64_t kf, int64_t kt, std::vector<char>& buf) {
    if (hash) return whole();
    buf.resize(map_stage_bytes(kt - kf, kt - kf + k - 1));
    if (!on_device && raw && handle && gpu_mu) {
      // thinned form: the dense arrays were only borrowed; hash the sequence again (the rare fall-back)
      std::lock_guard<std::mutex> g(*gpu_mu);
      const int hrc = map_hash_sequence_device(handle, raw, len, k, &dev);
      if (hrc == WFM_OK) on_device = true; else fetch_rc = hrc;
    }
    if (!on_device) { if (fetch_rc == WFM_OK) fetch_rc = WFM_E_HIP; memset(buf.data(), 0, buf.size()); return packed(buf.data(), kf, kt); }
    const int rc = map_hashed_fetch_packed(&dev, kf, kt, kf, kt + k - 1, buf.data());
    if (rc != WFM_OK) { fetch_rc = rc; memset(buf.data(), 0, buf.size()); }
    return packed(buf.data(), kf, kt);
  }
  // what a replay of chunk j reads
  View refetch_chunk(size_t j, std::vector<char>& buf) {
    View v;
    if (!thinned) { v.d = refetch(bounds[j], bounds[j + 1], buf); return v; }
    const int64_t c0 = cidx[j], c1 = cidx[j + 1];
    if (!h_pos.empty() || c1 == c0) {
      static const uint32_t none = 0;
      v.sp.pos = h_pos.empty() ? &none : h_pos.data() + c0; v.sp.hash = h_hash.data() + c0; v.sp.strand = h_strand.data() + c0; v.sp.n = (size_t)(c1 - c0);
      return v;
    }
    buf.resize((size_t)(c1 - c0) * 13);
    const int rc = map_sparse_fetch_packed(&sparse, c0, c1, buf.data());
    if (rc != WFM_OK) { fetch_rc = rc; memset(buf.data(), 0, buf.size()); }
    v.sp = packed_sparse(buf.data(), (size_t)(c1 - c0));
    return v;
  }
  // the closing cut / sort / de-duplication of records the device winnowed
  void finish_device_records() {
    std::vector<const std::vector<wfm_minmer_t>*> lists(1, &dev_raw);
    finish_lists(lists, w, sort_threads, result, ms_parts + 1);
    std::vector<wfm_minmer_t>().swap(dev_raw);
  }
  // sequential: check every speculation against the state the previous chunk really reached
  void stitch() {
    // WFM_WINNOW_FORCE (tests): 1 = treat every speculation as failed, 2 = take the one-stream fall-back
    static const int force = [] { const char* e = getenv("WFM_WINNOW_FORCE"); return e ? atoi(e) : 0; }();
    int64_t expired = force == 2 ? 1 : 0;
    for (const auto& c : chunk) expired += c->expired_in_refill();
    std::vector<char> buf;
    if (chunk.size() > 1 && expired > 0) {
      // an expired pool entry took part in a refill: the chunks' histories of expired entries differ
      // from the single stream's, so only the single stream is trusted
      chunk.clear();
      started_from.clear();
      replays = -1;
      const Slice d = refetch(0, nk, buf);
      Winnower W(d, len, k, w, s, seq_id);
      W.advance(0, nk);
      W.flush_end();
      finish_records(W.out, w, sort_threads);
      result.assign(W.out);
      norm.reset(); hash.reset(); strand.reset();
      return;
    }
    const auto tb = std::chrono::steady_clock::now();
    for (size_t j = 1; j < chunk.size(); ++j) {
      const Winnower& prev = *chunk[j - 1];
      if (force == 1 || !(started_from[j] == prev.live_state(bounds[j]))) {
        auto R = make(refetch_chunk(j, buf));
        R->take_state(prev);
        R->advance(bounds[j], bounds[j + 1]);
        chunk[j] = std::move(R);
        ++replays;
      } else {
        chunk[j]->resolve_unknown_starts(prev);
      }
    }
    chunk.back()->flush_end();
    const auto tc = std::chrono::steady_clock::now();
    std::vector<const std::vector<wfm_minmer_t>*> lists;
    for (const auto& c : chunk) lists.push_back(&c->out);
    finish_lists(lists, w, sort_threads, result, ms_parts + 1);
    const auto td = std::chrono::steady_clock::now();
    chunk.clear();
    started_from.clear();
    ms_parts[0] = std::chrono::duration<double, std::milli>(tc - tb).count();
    ms_parts[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td).count();
    norm.reset(); hash.reset(); strand.reset();
  }
};

// The hash threshold of the thinned stream: ~c x s of the W k-mers of a window stay (WFM_PREFILTER_C);
// 0 = no thinning (WFM_PREFILTER=0, or more than half of the k-mers would stay anyway).
// c: every k-mer let through is an event of the winnower, every window that holds fewer than s candidates stays WHOLE (W events, and on the device
// a chunk that runs again with the full capacities) -- so as few as keep such windows out of ordinary sequence: the count of candidates in a window
// is about Poisson(c s), and c s = m with m - 5 sqrt(m) = s leaves one window in 10^8 - 10^9 short whatever s is (s = 23: c = 2.7, s = 39: 2.2), never
// more than the 3 of rounds 3 - 5.  Measured on a full-size C4 rank (s = 23, 8 x 249 Mbp; gpurun_out/r6bj, r6bk): index build 248 - 255 ms at c = 3,
// 236 - 246 at 2.7, 234 - 237 at 2.5, 300 at 2.4 (81 short windows per chromosome: their chunks run twice), 381 at 1.8; the records are the same bytes.
double prefilter_c(int s) {
  const double root = (5.0 + std::sqrt(25.0 + 4.0 * (double)std::max(1, s))) / 2.0;
  return std::min(3.0, root * root / (double)std::max(1, s));
}
uint64_t prefilter_tau(int s, int64_t W) {
  const char* on = getenv("WFM_PREFILTER");
  if (on && atoi(on) == 0) return 0;
  const char* ce = getenv("WFM_PREFILTER_C");
  const double c = ce ? atof(ce) : prefilter_c(s);
  const double share = c * (double)s / (double)std::max<int64_t>(1, W);
  if (!(share > 0) || share > 0.5) return 0;
  return map_prefilter_tau(c, s, W);
}

int64_t chunk_length() {
  const char* e = getenv("WFM_WINNOW_CHUNK");  // k-mers per speculative chunk; 0 = one stream per sequence
  return e ? atoll(e) : (int64_t)1 << 18;
}

}  // namespace

namespace {
// the device's selection of k-mers (map_prefilter.hip) restated from its definition: candidates (hash <= tau), fresh
// candidates, windows with fewer than s of them, and every valid k-mer of such windows
void thin_on_host(const uint64_t* hash, const int8_t* strand, int64_t n, int64_t W, int s, uint64_t tau, std::vector<uint32_t>& pos,
                  std::vector<uint64_t>& hs, std::vector<int8_t>& st) {
  std::vector<std::pair<uint64_t, int64_t>> cand;
  for (int64_t i = 0; i < n; ++i)
    if (strand[i] != 0 && hash[i] <= tau) cand.emplace_back(hash[i], i);
  std::sort(cand.begin(), cand.end());
  std::vector<uint32_t> F((size_t)n + 1, 0), P((size_t)n + 1, 0);  // prefix sums, shifted by one
  {
    std::vector<uint8_t> fresh((size_t)n, 0);
    for (size_t j = 0; j < cand.size(); ++j)
      if (j == 0 || cand[j].first != cand[j - 1].first || cand[j].second - cand[j - 1].second >= W) fresh[(size_t)cand[j].second] = 1;
    for (int64_t i = 0; i < n; ++i) F[(size_t)i + 1] = F[(size_t)i] + fresh[(size_t)i];
  }
  for (int64_t a = 0; a < n; ++a) {
    const uint32_t under = (a + W <= n && F[(size_t)(a + W)] - F[(size_t)a] < (uint32_t)s) ? 1u : 0u;
    P[(size_t)a + 1] = P[(size_t)a] + under;
  }
  for (int64_t i = 0; i < n; ++i) {
    if (strand[i] == 0) continue;
    const bool dense = P[(size_t)i + 1] - P[(size_t)std::max<int64_t>(0, i - W + 1)] > 0;  // a window a in (i-W, i] is under the bound
    if (hash[i] <= tau || dense) { pos.push_back((uint32_t)i); hs.push_back(hash[i]); st.push_back(strand[i]); }
  }
}
// Is there a k-mer among the first k-1 whose N the reference does not notice (Winnower's `extra_`)?  Those are hashed
// on the host and woven into the stream there; a sequence that has one is not winnowed on the device.
bool has_unnoticed_n(const char* head, int64_t head_len, int k) {
  for (int64_t i = 0; i < k - 1 && i + k <= head_len; ++i) {
    bool late_n = false, any_n = false;
    for (int64_t b = std::max<int64_t>(i, k - 1); b < i + k; ++b) late_n |= head[b] == 'N';
    for (int64_t b = i; b < i + k; ++b) any_n |= head[b] == 'N';
    if (!late_n && any_n) return true;
  }
  return false;
}
}  // namespace

extern "C" int64_t wfm_add_minmers(wfm_handle_t* h, const char* seq, int64_t len, int k, int w, int s, int32_t seq_id,
                                   wfm_minmer_t* out, int64_t cap) {
  const char* seqs[1] = {seq};
  return wfm_add_minmers_multi(h, seqs, &len, &seq_id, 1, k, w, s, 1, out, cap, nullptr);
}

// Many sequences at once.  The calling thread hashes one sequence after the other on the GPU; the
// hashes stay there.  A second thread streams them out chunk by chunk through a small ring of pinned
// slots (asynchronous copies at link speed, no page faults); `threads` host workers take a slot's
// bytes into their own reused buffer, hand the slot back and winnow the chunk -- sequences side by
// side (the reference's ThreadPool over buildHelper, winSketch.hpp:200-239) and, within a long
// sequence, its speculative chunks.  The host never holds more than one chunk per worker.
// Output is the concatenation in input order.
namespace {
// where the records of the sequences go: put() is called once per sequence, in input order, from the calling thread
struct MinmerSink {
  virtual ~MinmerSink() = default;
  virtual int put(const wfm_minmer_t* recs, int64_t n) = 0;
  virtual int put_device(const wfm_minmer_t* d_recs, int64_t n) = 0;  // the same, records on the device
};

// later: (optional) receives the release of the device work buffers and of the block pool instead of it being done before
// the return -- the caller that goes on to allocate gigabytes (the index) runs it afterwards: memory the driver has just
// been handed back is scrubbed in the background, and an allocation that follows on its heels waits for that
int64_t add_minmers_core(wfm_handle_t* h, const char* const* seqs, const int64_t* lens, const int32_t* seq_ids, int64_t nseq,
                         int k, int w, int s, int threads, MinmerSink& sink, int64_t* counts, std::function<void()>* later = nullptr) {
  if (!h || nseq < 0 || (nseq && (!seqs || !lens || !seq_ids))) return WFM_E_ARG;
  if (k < 1 || k > 32 || w < k || s < 1) { wfm_set_error(h, "need 1 <= k <= 32, w >= k, s >= 1"); return WFM_E_UNSUPPORTED; }
  const int nthreads = std::max(1, threads);
  const int64_t chunk_len = nthreads > 1 ? chunk_length() : 0;
  const bool streamed = chunk_len >= 64 * (int64_t)w;  // otherwise: whole sequences through ordinary memory
  MapStage* stage = nullptr;
  const int64_t W = (int64_t)w - k + 1;  // k-mers per window
  const uint64_t tau = streamed ? prefilter_tau(s, W) : 0;
  const int64_t slot_kmers = chunk_len + chunk_len / 2 + 2 * (int64_t)w + 1;
  const size_t slot_bytes = std::max(map_stage_bytes(slot_kmers, slot_kmers + k), (size_t)slot_kmers * 13);
  const int nslots = 16;
  if (streamed) {
    const int src = map_stage_acquire(h, slot_bytes, nslots, &stage);
    if (src != WFM_OK) return src;
  }

  std::vector<std::unique_ptr<SeqJob>> jobs((size_t)nseq);
  struct Task { SeqJob* job; size_t chunk; int slot; };
  std::deque<Task> queue;       // for the workers
  std::deque<SeqJob*> hashed;   // for the streaming thread
  std::vector<int> free_slots;
  for (int i = 0; i < (streamed ? stage->nslots : 0); ++i) free_slots.push_back(i);
  std::vector<SeqJob*> stitched;  // device arrays to release (done by the calling thread)
  std::mutex mu;
  std::condition_variable cv_work, cv_room, cv_slot, cv_hashed;
  bool done = false, hashed_done = false;
  std::atomic<int> async_rc{WFM_OK};
  std::atomic<bool> async_msg_set{false};  // the failing path left its own text in the handle: do not overwrite it
  int64_t inflight_bases = 0;
  const int64_t max_inflight = 1ll << 31;  // ~2 Gbp hashed but not yet winnowed: 10 B/base on the device

  auto worker = [&]() {
    std::vector<char> local;
    for (;;) {
      Task task;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return done || !queue.empty(); });
        if (queue.empty()) return;
        task = queue.front();
        queue.pop_front();
      }
      SeqJob* J = task.job;
      if (J->dev_winnowed) {  // winnowed on the device: only the closing sort is left
        const auto ts = std::chrono::steady_clock::now();
        J->finish_device_records();
        J->ms_stitch = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count();
        J->stitched.store(true, std::memory_order_release);
        {
          std::lock_guard<std::mutex> lk(mu);
          inflight_bases -= J->len;
          stitched.push_back(J);
        }
        cv_room.notify_one();
        continue;
      }
      SeqJob::View v;
      v.d = J->whole();
      if (task.slot >= 0) {
        int64_t kf, kt, bt;
        J->chunk_range(task.chunk, &kf, &kt, &bt);
        const size_t mc = J->thinned ? (size_t)(J->cidx[task.chunk + 1] - J->cidx_warm[task.chunk]) : 0;
        const size_t nbytes = J->thinned ? mc * 13 : map_stage_bytes(kt - kf, bt - kf);
        if (local.size() < std::max<size_t>(nbytes, 16)) local.resize(std::max(nbytes, stage->slot_bytes));
        const int wrc = map_stage_wait(stage, task.slot);
        if (wrc == WFM_OK) memcpy(local.data(), stage->slot(task.slot), nbytes);
        else { memset(local.data(), 0, nbytes); async_rc.store(wrc); }
        {
          std::lock_guard<std::mutex> lk(mu);
          free_slots.push_back(task.slot);
        }
        cv_slot.notify_one();
        if (J->thinned) {
          static const uint32_t none = 0;
          v.sp = SeqJob::packed_sparse(local.data(), mc);
          if (mc == 0) v.sp.pos = &none;  // an empty stream is still a thinned one
        } else {
          v.d = SeqJob::packed(local.data(), kf, kt);
        }
      }
      J->run_chunk(task.chunk, v);
      if (J->pending.fetch_sub(1) == 1) {  // last chunk of this sequence: stitch here
        const auto ts = std::chrono::steady_clock::now();
        J->stitch();
        J->ms_stitch = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count();
        if (J->fetch_rc != WFM_OK) async_rc.store(J->fetch_rc);
        J->stitched.store(true, std::memory_order_release);
        {
          std::lock_guard<std::mutex> lk(mu);
          inflight_bases -= J->len;
          stitched.push_back(J);
        }
        cv_room.notify_one();
      }
    }
  };
  // streams the chunks of every hashed sequence into the ring, in order
  auto streamer = [&]() {
    for (;;) {
      SeqJob* J;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_hashed.wait(lk, [&] { return hashed_done || !hashed.empty(); });
        if (hashed.empty()) break;
        J = hashed.front();
        hashed.pop_front();
      }
      for (size_t c = 0; c + 1 < J->bounds.size(); ++c) {
        int slot;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv_slot.wait(lk, [&] { return !free_slots.empty(); });
          slot = free_slots.back();
          free_slots.pop_back();
        }
        int64_t kf, kt, bt;
        J->chunk_range(c, &kf, &kt, &bt);
        const int crc = J->thinned ? map_stage_copy_sparse(stage, slot, &J->sparse, J->cidx_warm[c], J->cidx[c + 1])
                                   : map_stage_copy(stage, slot, &J->dev, kf, kt, kf, bt);
        if (crc != WFM_OK) async_rc.store(crc);  // the worker still runs (on whatever the slot holds): keeps the bookkeeping simple
        {
          std::lock_guard<std::mutex> lk(mu);
          queue.push_back(Task{J, c, slot});
        }
        cv_work.notify_one();
      }
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      done = true;
    }
    cv_work.notify_all();
  };
  auto release_stitched = [&]() {
    std::vector<SeqJob*> list;
    {
      std::lock_guard<std::mutex> lk(mu);
      list.swap(stitched);
    }
    for (SeqJob* J : list) {
      if (J->on_device) { map_hashed_free(&J->dev); J->on_device = false; }
      map_sparse_free(&J->sparse);
    }
  };

  // finished sequences leave in input order while the later ones are still being worked on
  int64_t next_out = 0, total = 0;
  int sink_rc = WFM_OK;
  auto flush_ready = [&](int64_t limit) {
    for (; next_out < limit; ++next_out) {
      SeqJob* J = jobs[(size_t)next_out].get();
      if (J && !J->stitched.load(std::memory_order_acquire)) break;
      const int64_t n = J ? (J->d_result ? J->n_result : (int64_t)J->result.size()) : 0;
      if (counts) counts[next_out] = n;
      if (n && sink_rc == WFM_OK) sink_rc = J->d_result ? sink.put_device(J->d_result, n) : sink.put(J->result.data(), n);
      total += n;
      if (J) J->result.release();
      if (J && J->d_result) {
        // the sink's device-to-device copy runs on the null stream and need not be over when the call returns; the
        // device thread's stream does not wait for the null stream, so the block must not go back to the pool before it is
        if (hipStreamSynchronize(nullptr) != hipSuccess && sink_rc == WFM_OK) sink_rc = WFM_E_HIP;
        map_dev_pool_put(wfm_device(h), J->d_result);
        J->d_result = nullptr;
      }
    }
  };
  std::mutex gpu_mu;  // the handle's stream and error string: this thread, and a worker that hashes a sequence again
  MapHashWork hash_work;
  MapThinWork thin_work;

  // several device threads, each with a stream and work buffers of its own: a sequence costs the device path 2 - 3 ms of
  // launches and round trips whatever its length, which the threads overlap (chromosome-sized sequences are one launch
  // set each and fill the device alone; a yeast genome is 128 short ones)
  const int n_dev_threads = std::max(1, std::min(16, getenv("WFM_WINNOW_DEV_THREADS") ? atoi(getenv("WFM_WINNOW_DEV_THREADS")) : 2));
  std::vector<MapWinnowWork> winnow_works((size_t)n_dev_threads);
  std::vector<MapFinishWork> finish_works((size_t)n_dev_threads);
  const bool dev_finish = !(getenv("WFM_FINISH_DEVICE") && atoi(getenv("WFM_FINISH_DEVICE")) == 0);
  int dev_levels = 0;
  int64_t dev_heaps = 0;
  const bool dev_winnow = !(getenv("WFM_WINNOW_DEVICE") && atoi(getenv("WFM_WINNOW_DEVICE")) == 0);
  // Sequences under 4 M k-mers stay with the host's workers.  Round 3 tried to move the switch down with several device
  // threads (each its own stream and buffers): on the C1 substitute (128 sequences of 0.2 - 1.5 Mbp) the index took 0.25 s with
  // 4 device threads from 256 k k-mers on, 0.19 - 0.21 s with the closing sort left to the host, against 0.13 - 0.15 s on the
  // host path (also with 32 host threads) -- a sequence costs the device path 2 - 4 ms of launches and stream
  // synchronisations whatever its length, the threads contend for the queue, and the feeding thread (hash + thin: 0.9 ms per
  // sequence) slows down beside them.  Only launch sets that span many sequences would change that.
  const int64_t dev_min = getenv("WFM_WINNOW_DEV_MIN") ? atoll(getenv("WFM_WINNOW_DEV_MIN")) : (int64_t)1 << 22;
  const int64_t dev_chunk = getenv("WFM_WINNOW_DEV_CHUNK") ? atoll(getenv("WFM_WINNOW_DEV_CHUNK")) : 0;  // 0: by sequence length
  int64_t dev_seqs = 0, dev_handed_back = 0, dev_chunks = 0, dev_replays = 0;
  uint32_t dev_why = 0;
  double ms_winnow = 0;
  // Winnowing and the closing sort of thinned sequences on the device (map_winnow.hip, map_finish.hip), on a stream and a
  // thread of their own: the calling thread is hashing and thinning the next sequence meanwhile.  A sequence the device
  // hands back joins the host path (the streamer's list).
  std::deque<SeqJob*> dev_queue;
  std::condition_variable cv_dev;
  bool dev_done = false;
  auto seq_finished = [&](SeqJob* J) {
    J->stitched.store(true, std::memory_order_release);
    {
      std::lock_guard<std::mutex> lk(mu);
      inflight_bases -= J->len;
    }
    cv_room.notify_one();
  };
  std::mutex dev_stat_mu;  // the counters below
  auto device_thread = [&](int dti) {
    (void)hipSetDevice(wfm_device(h));
    MapWinnowWork& winnow_work = winnow_works[(size_t)dti];
    MapFinishWork& finish_work = finish_works[(size_t)dti];
    hipStream_t st2 = nullptr;
    if (hipStreamCreateWithFlags(&st2, hipStreamNonBlocking) != hipSuccess) st2 = nullptr;
    for (;;) {
      SeqJob* J;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_dev.wait(lk, [&] { return dev_done || !dev_queue.empty(); });
        if (dev_queue.empty()) break;
        J = dev_queue.front();
        dev_queue.pop_front();
      }
      const auto tw = std::chrono::steady_clock::now();
      wfm_minmer_t* d_recs = nullptr;
      int64_t n_recs = 0;
      MapWinnowInfo wi;
      // chunk length: one wave per chunk, and about as many chunks as the device keeps resident at once (a chunk's two
      // windows of warm-up are its overhead: no chunk under four windows)
      const int64_t auto_chunk = std::min<int64_t>((J->nk + 6143) / 6144, (int64_t)1 << 16);
      // (no stream of its own: the sequence goes to the host's winnower like any other the device hands back)
      int wrc = st2 ? map_winnow_sparse_device(h, &J->sparse, J->len, k, w, s, J->seq_id, std::max<int64_t>(dev_chunk > 0 ? dev_chunk : auto_chunk, 4 * (int64_t)w),
                                               &winnow_work, &d_recs, &n_recs, &wi, st2)
                    : 1;
      { std::lock_guard<std::mutex> lk(dev_stat_mu); dev_chunks += wi.chunks; dev_replays += wi.replays; }
      if (wrc == WFM_OK && dev_finish) {  // the closing cut / sort / de-duplication on the device as well
        wfm_minmer_t* d_fin = nullptr;
        int64_t n_fin = 0;
        MapFinishInfo fi;
        wrc = map_finish_records_device(h, d_recs, n_recs, w, &finish_work, &d_fin, &n_fin, &fi, st2);
        if (wrc == WFM_OK && n_fin) {
          J->d_result = (wfm_minmer_t*)map_dev_pool_get(wfm_device(h), (size_t)n_fin * sizeof(wfm_minmer_t));
          if (!J->d_result || hipMemcpyAsync(J->d_result, d_fin, (size_t)n_fin * sizeof(wfm_minmer_t), hipMemcpyDeviceToDevice, st2) != hipSuccess ||
              hipStreamSynchronize(st2) != hipSuccess) {
            wfm_set_error(h, "out of device memory (minmer records)");
            wrc = WFM_E_NOMEM;
          }
        }
        if (wrc == WFM_OK) {
          J->n_result = n_fin;
          J->dev_winnowed = true;
          std::lock_guard<std::mutex> lk(dev_stat_mu);
          ++dev_seqs;
          dev_levels = std::max(dev_levels, fi.levels);
          dev_heaps += fi.heap_ranges;
        }
      } else if (wrc == WFM_OK) {  // WFM_FINISH_DEVICE=0: the closing sort by a host worker
        J->dev_raw.resize((size_t)n_recs);
        if (n_recs && (hipMemcpyAsync(J->dev_raw.data(), d_recs, (size_t)n_recs * sizeof(wfm_minmer_t), hipMemcpyDeviceToHost, st2) != hipSuccess ||
                       hipStreamSynchronize(st2) != hipSuccess)) {
          wfm_set_error(h, "device-to-host copy of minmer records failed");
          wrc = WFM_E_HIP;
        } else {
          J->dev_winnowed = true;
          std::lock_guard<std::mutex> lk(dev_stat_mu);
          ++dev_seqs;
        }
      }
      { std::lock_guard<std::mutex> lk(dev_stat_mu); ms_winnow += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count(); }
      if (wrc < 0) {  // an error: the sequence ends here with no records, the call fails (the failing path has set the message)
        async_msg_set.store(true);
        async_rc.store(wrc);
        map_sparse_free(&J->sparse);
        seq_finished(J);
      } else if (wrc == 1) {  // not for the device after all: the host's chunks
        { std::lock_guard<std::mutex> lk(dev_stat_mu); ++dev_handed_back; dev_why |= wi.why; }
        {
          std::lock_guard<std::mutex> lk(mu);
          hashed.push_back(J);
        }
        cv_hashed.notify_one();
      } else if (J->d_result || J->dev_raw.empty()) {  // winnowed and finished: nothing left to do
        map_sparse_free(&J->sparse);
        seq_finished(J);
      } else {
        map_sparse_free(&J->sparse);
        {
          std::lock_guard<std::mutex> lk(mu);
          queue.push_back(Task{J, 0, -1});
        }
        cv_work.notify_one();
      }
    }
    if (st2) (void)hipStreamDestroy(st2);
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker);
  std::vector<std::thread> dev_threads;
  for (int t = 0; t < n_dev_threads; ++t) dev_threads.emplace_back(device_thread, t);
  std::thread stream_thread;
  if (streamed) stream_thread = std::thread(streamer);
  int rc = WFM_OK;
  const auto t_start = std::chrono::steady_clock::now();
  double ms_hash = 0, ms_thin = 0;
  int64_t kept_kmers = 0, thinned_kmers = 0;
  for (int64_t i = 0; i < nseq && rc == WFM_OK; ++i) {
    const int64_t len = lens[i];
    if (!seqs[i] || len < 0) { rc = WFM_E_ARG; break; }
    if (len < k) continue;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv_room.wait(lk, [&] { return inflight_bases == 0 || inflight_bases + len <= max_inflight; });
      inflight_bases += len;
    }
    release_stitched();
    flush_ready(i);
    auto J = std::make_unique<SeqJob>();
    J->idx = i; J->seq_id = seq_ids[i]; J->len = len; J->nk = len - k + 1; J->k = k; J->w = w; J->s = s;
    J->sort_threads = std::min(16, nthreads);
    const auto t0 = std::chrono::steady_clock::now();
    const bool thin = tau != 0 && J->nk >= W && J->nk < ((int64_t)1 << 32) - 1;
    std::unique_lock<std::mutex> gpu(gpu_mu);
    // GPU: normalise + 2 x MurmurHash3 per base; the thinned form reuses one set of device buffers
    rc = thin ? map_hash_sequence_into(h, &hash_work, seqs[i], len, k, &J->dev) : map_hash_sequence_device(h, seqs[i], len, k, &J->dev);
    if (rc != WFM_OK) break;
    J->on_device = true;
    ms_hash += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    J->plan(streamed ? chunk_len : 0);
    const auto t1 = std::chrono::steady_clock::now();
    if (thin) {
      // thin the stream on the device; only the kept k-mers outlive this iteration
      J->raw = seqs[i]; J->handle = h; J->gpu_mu = &gpu_mu;
      rc = map_prefilter_device(h, &J->dev, W, s, tau, &J->sparse, &thin_work);
      if (rc != WFM_OK) { map_hashed_free(&J->dev); break; }
      J->thinned = true;
      kept_kmers += J->sparse.m;
      thinned_kmers += J->nk;
      J->head.assign((size_t)std::min<int64_t>(len, 2 * (int64_t)k), 'N');
      rc = map_hashed_fetch(&J->dev, 0, 0, 0, (int64_t)J->head.size(), nullptr, nullptr, &J->head[0]);
      const size_t nc = J->bounds.size() - 1;
      std::vector<int64_t> q(2 * nc + 1), r(2 * nc + 1);
      for (size_t c = 0; c <= nc; ++c) q[c] = J->bounds[c];
      for (size_t c = 0; c < nc; ++c) q[nc + 1 + c] = J->warm_from(c);
      if (rc == WFM_OK) rc = map_sparse_lower_bound(h, &J->sparse, q.data(), (int)q.size(), r.data());
      if (rc != WFM_OK) { map_hashed_free(&J->dev); map_sparse_free(&J->sparse); break; }
      J->cidx.assign(r.begin(), r.begin() + (long)nc + 1);
      J->cidx_warm.assign(r.begin() + (long)nc + 1, r.end());
      map_hashed_free(&J->dev);  // borrowed: just forgets the pointers
      J->on_device = false;
      // the winnowing itself goes to the device (map_winnow.hip, its own thread and stream below) unless the sequence starts
      // with a k-mer whose N the reference does not notice
      // -- or is short: the device path costs a few milliseconds per sequence in launches and round trips whatever its
      // length, the host's workers take short sequences side by side (WFM_WINNOW_DEV_MIN: k-mers from which on the device is used)
      J->for_device = dev_winnow && J->nk >= dev_min && !has_unnoticed_n(J->head.data(), (int64_t)J->head.size(), k);
    }
    gpu.unlock();
    ms_thin += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    SeqJob* Jp = J.get();
    jobs[(size_t)i] = std::move(J);
    if (Jp->for_device) {
      {
        std::lock_guard<std::mutex> lk(mu);
        dev_queue.push_back(Jp);
      }
      cv_dev.notify_one();
    } else if (streamed) {
      {
        std::lock_guard<std::mutex> lk(mu);
        hashed.push_back(Jp);
      }
      cv_hashed.notify_one();
    } else {
      Jp->norm.reset(new char[(size_t)len]);
      Jp->hash.reset(new uint64_t[(size_t)Jp->nk]);
      Jp->strand.reset(new int8_t[(size_t)Jp->nk]);
      rc = map_hashed_fetch(&Jp->dev, 0, Jp->nk, 0, len, Jp->hash.get(), Jp->strand.get(), Jp->norm.get());
      map_hashed_free(&Jp->dev);
      Jp->on_device = false;
      if (rc != WFM_OK) { wfm_set_error(h, "device-to-host copy of k-mer hashes failed"); jobs[(size_t)i].reset(); break; }  // never queued: nobody will stitch it
      {
        std::lock_guard<std::mutex> lk(mu);
        queue.push_back(Task{Jp, 0, -1});
      }
      cv_work.notify_one();
    }
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    dev_done = true;
  }
  cv_dev.notify_all();
  for (auto& t : dev_threads) t.join();  // before the streamer is told that nothing more will come: the device may still hand sequences back
  {
    std::lock_guard<std::mutex> lk(mu);
    hashed_done = true;
    if (!streamed) done = true;
  }
  cv_hashed.notify_all();
  cv_work.notify_all();
  const auto t_fed = std::chrono::steady_clock::now();
  {  // hand the sequences on in order as their stitches finish, while the later ones are still being stitched
    std::unique_lock<std::mutex> lk(mu);
    while (next_out < nseq) {
      SeqJob* J = jobs[(size_t)next_out].get();
      if (J && !J->stitched.load(std::memory_order_acquire)) { cv_room.wait(lk); continue; }  // a worker signals after every stitch
      lk.unlock();
      flush_ready(next_out + 1);
      lk.lock();
    }
  }
  if (stream_thread.joinable()) stream_thread.join();
  for (auto& t : pool) t.join();
  release_stitched();
  for (auto& J : jobs)
    if (J) { if (J->on_device) { map_hashed_free(&J->dev); J->on_device = false; } map_sparse_free(&J->sparse); if (J->d_result) { map_dev_pool_put(wfm_device(h), J->d_result); J->d_result = nullptr; } }  // after an error
  auto release_work = [hash_work, thin_work, winnow_works, finish_works]() mutable {
    map_hash_work_free(&hash_work);
    map_thin_work_free(&thin_work);
    for (auto& wk : winnow_works) map_winnow_work_free(&wk);
    for (auto& wk : finish_works) map_finish_work_free(&wk);
    map_dev_pool_trim();
  };
  if (getenv("WFM_DEBUG") && (dev_seqs || dev_handed_back))
    fprintf(stderr, "[wfm] winnowing on the device: %lld sequences in %lld chunks (WFM_WINNOW_DEV_CHUNK %lld), %lld chunks replayed after a failed speculation, %.1f ms (closing sort %s: %d levels at most, %lld ranges heap-sorted); %lld handed back to the host (why 0x%x)\n",
            (long long)dev_seqs, (long long)dev_chunks, (long long)dev_chunk, (long long)dev_replays, ms_winnow, dev_finish ? "on the device" : "on the host", dev_levels, (long long)dev_heaps,
            (long long)dev_handed_back, dev_why);
  if (getenv("WFM_DEBUG")) {
    int64_t nchunks = 0, replays = 0;
    double stitch_max = 0;
    const SeqJob* slowest = nullptr;
    for (const auto& J : jobs)
      if (J) { nchunks += (int64_t)J->bounds.size() - 1; replays += J->replays; if (J->ms_stitch >= stitch_max) { stitch_max = J->ms_stitch; slowest = J.get(); } }
    if (slowest)
      fprintf(stderr, "[wfm] longest stitch: compare %.1f, count %.1f, fill %.1f, sort %.1f, unique %.1f, release %.1f ms\n", slowest->ms_parts[0],
              slowest->ms_parts[1], slowest->ms_parts[2], slowest->ms_parts[3], slowest->ms_parts[4], slowest->ms_parts[5]);
    fprintf(stderr, "[wfm] add_minmers_multi: %lld sequences in %lld chunks (%lld replayed), %d workers, %s, %.1f %% of the k-mers kept: hashing thread %.1f ms (GPU hashing %.1f, thinning %.1f), drain %.1f ms (longest stitch %.1f)\n",
            (long long)nseq, (long long)nchunks, (long long)replays, nthreads, streamed ? "streamed through the pinned ring" : "whole sequences",
            thinned_kmers ? 100.0 * (double)kept_kmers / (double)thinned_kmers : 100.0,
            std::chrono::duration<double, std::milli>(t_fed - t_start).count(), ms_hash, ms_thin,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fed).count(), stitch_max);
  }
  if (rc == WFM_OK && async_rc.load() != WFM_OK) {
    rc = async_rc.load();
    if (!async_msg_set.load()) wfm_set_error(h, "device-to-host streaming of k-mer hashes failed");
  }
  if (rc != WFM_OK) { release_work(); return rc; }
  flush_ready(nseq);  // (the records of the last sequences leave their pooled blocks here)
  if (later) *later = release_work; else release_work();
  if (sink_rc != WFM_OK) return sink_rc;
  return total;
}

// records into the caller's array, as far as it goes
struct HostSink : MinmerSink {
  wfm_minmer_t* out; int64_t cap, at = 0;
  HostSink(wfm_minmer_t* o, int64_t c) : out(o), cap(c) {}
  int put(const wfm_minmer_t* recs, int64_t n) override {
    if (at < cap) memcpy(out + at, recs, (size_t)std::min(n, cap - at) * sizeof(wfm_minmer_t));
    at += n;
    return WFM_OK;
  }
  int put_device(const wfm_minmer_t* d_recs, int64_t n) override {
    if (at < cap && hipMemcpy(out + at, d_recs, (size_t)std::min(n, cap - at) * sizeof(wfm_minmer_t), hipMemcpyDeviceToHost) != hipSuccess) return WFM_E_HIP;
    at += n;
    return WFM_OK;
  }
};

// records into one growing device array (the index is built from it without a detour through host memory)
struct DeviceSink : MinmerSink {
  wfm_handle_t* h;
  wfm_minmer_t* d = nullptr;
  int64_t cap = 0, n = 0;
  DeviceSink(wfm_handle_t* hh, int64_t expect) : h(hh), cap(std::max<int64_t>(expect, 4096)) {}
  // (from and back to the per-device block cache like every other block of the map path: until round 5 this one -- 3 GB for a pangenome rank --
  // went to the driver with hipMalloc / hipFree at every map call, and whatever the NEXT allocation of the process was then landed on memory the
  // driver was still wiping: the map wall of a full-size rank read 0.6 s or 1.1 - 2 s depending on what had run before it)
  ~DeviceSink() override { if (d) wfm_dfree(d); }
  int put(const wfm_minmer_t* recs, int64_t m) override { return append(recs, m, hipMemcpyHostToDevice); }
  int put_device(const wfm_minmer_t* d_recs, int64_t m) override { return append(d_recs, m, hipMemcpyDeviceToDevice); }
  int append(const wfm_minmer_t* recs, int64_t m, hipMemcpyKind kind) {
    if (hipSetDevice(wfm_device(h)) != hipSuccess) return WFM_E_HIP;
    if (!d || n + m > cap) {
      int64_t want = d ? std::max(cap + cap / 2, n + m) : std::max(cap, m);
      wfm_minmer_t* nd = nullptr;
      if (wfm_dmalloc((void**)&nd, (size_t)want * sizeof(wfm_minmer_t)) != hipSuccess) {  // (the cache has given everything back and tried again by then)
        (void)hipGetLastError();
        wfm_set_error(h, "out of device memory (minmer intervals)");
        return WFM_E_NOMEM;
      }
      if (d && n && hipMemcpy(nd, d, (size_t)n * sizeof(wfm_minmer_t), hipMemcpyDeviceToDevice) != hipSuccess) { wfm_dfree(nd); return WFM_E_HIP; }
      if (d) wfm_dfree(d);
      d = nd; cap = want;
    }
    if (hipMemcpy(d + n, recs, (size_t)m * sizeof(wfm_minmer_t), kind) != hipSuccess) { wfm_set_error(h, "upload of minmer intervals failed"); return WFM_E_HIP; }
    n += m;
    return WFM_OK;
  }
};
}  // namespace

extern "C" int64_t wfm_add_minmers_multi(wfm_handle_t* h, const char* const* seqs, const int64_t* lens, const int32_t* seq_ids, int64_t nseq,
                                         int k, int w, int s, int threads, wfm_minmer_t* out, int64_t cap, int64_t* counts) {
  if (cap && !out) return WFM_E_ARG;
  HostSink sink(out, cap);
  return add_minmers_core(h, seqs, lens, seq_ids, nseq, k, w, s, threads, sink, counts);
}

// Sketch::build in one call (winSketch.hpp:175-457): minmer intervals of all sequences, then the index stage, with
// the intervals going from the workers straight to the device.  *n_windows (optional) = number of intervals.
extern "C" int wfm_index_build_sequences(wfm_handle_t* h, const char* const* seqs, const int64_t* lens, const int32_t* seq_ids, int64_t nseq,
                                         int k, int w, int s, int threads, double max_kmer_freq, wfm_index_t** out, int64_t* n_windows) {
  if (!h || !out) return WFM_E_ARG;
  *out = nullptr;
  int64_t bases = 0;
  for (int64_t i = 0; i < nseq && lens; ++i) bases += std::max<int64_t>(0, lens[i]);
  DeviceSink sink(h, bases / std::max(1, w) * (int64_t)s * 5 / 2 + 4096);  // about 2 s / w intervals per base
  std::function<void()> release;
  const int64_t n = add_minmers_core(h, seqs, lens, seq_ids, nseq, k, w, s, threads, sink, nullptr, &release);
  int rc = n < 0 ? (int)n : WFM_OK;
  if (n_windows && n >= 0) *n_windows = n;
  if (n > 0) rc = map_index_build_device(h, sink.d, n, max_kmer_freq, out);  // n == 0: no index, *out stays NULL
  if (release) release();
  return rc;
}

// Test hook (CPU test-suite): the host winnowing stage on caller-supplied k-mer hashes.
// chunk_len > 0 runs the speculative chunked form (single thread) and reports replays in *replays.
extern "C" int64_t wfmh_test_winnow_chunked(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id, const uint64_t* hash,
                                            const int8_t* strand, int64_t chunk_len, wfm_minmer_t* out, int64_t cap, int* replays) {
  if (len < k) return 0;
  SeqJob J;
  J.seq_id = seq_id; J.len = len; J.nk = len - k + 1; J.k = k; J.w = w; J.s = s;
  J.sort_threads = 4;
  J.norm.reset(new char[(size_t)len]);
  memcpy(J.norm.get(), seq, (size_t)len);
  normalise(J.norm.get(), len);
  J.hash.reset(new uint64_t[(size_t)J.nk]);
  J.strand.reset(new int8_t[(size_t)J.nk]);
  memcpy(J.hash.get(), hash, (size_t)J.nk * 8);
  memcpy(J.strand.get(), strand, (size_t)J.nk);
  J.bounds.assign(1, 0);
  if (chunk_len > 0)
    for (int64_t b = chunk_len; b < J.nk; b += chunk_len) J.bounds.push_back(b);
  J.bounds.push_back(J.nk);
  J.chunk.resize(J.bounds.size() - 1);
  J.started_from.resize(J.bounds.size() - 1);
  for (size_t c = 0; c + 1 < J.bounds.size(); ++c) { SeqJob::View v; v.d = J.whole(); J.run_chunk(c, v); }
  J.stitch();
  if (replays) *replays = J.replays;
  const int64_t n = (int64_t)J.result.size();
  for (int64_t i = 0; i < n && i < cap; ++i) out[i] = J.result[(size_t)i];
  return n;
}

extern "C" int64_t wfmh_test_winnow(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id,
                                    const uint64_t* hash, const int8_t* strand, wfm_minmer_t* out, int64_t cap) {
  if (len < k) return 0;
  std::string norm(seq, (size_t)len);
  normalise(&norm[0], len);
  std::vector<wfm_minmer_t> res;
  winnow(norm.data(), len, k, w, s, seq_id, hash, strand, res);
  const int64_t n = (int64_t)res.size();
  for (int64_t i = 0; i < n && i < cap; ++i) out[i] = res[(size_t)i];
  return n;
}

// Test hook (CPU test-suite): the thinned stream.  The device's selection (map_prefilter.hip) is restated
// here from its definition -- candidates, fresh candidates, windows under the bound, their dilation -- and
// the chunked winnowing then runs on the kept k-mers only.  kept_pos (optional, cap_kept entries) receives the
// kept positions, *n_kept their number: the GPU test holds wfm_prefilter_kmers against them.
extern "C" int64_t wfmh_test_winnow_thinned(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id, const uint64_t* hash,
                                            const int8_t* strand, double c_factor, int64_t chunk_len, wfm_minmer_t* out, int64_t cap,
                                            uint32_t* kept_pos, int64_t cap_kept, int64_t* n_kept, int* replays) {
  if (len < k) return 0;
  SeqJob J;
  J.seq_id = seq_id; J.len = len; J.nk = len - k + 1; J.k = k; J.w = w; J.s = s;
  J.sort_threads = 4;
  const int64_t n = J.nk, W = (int64_t)w - k + 1;
  J.norm.reset(new char[(size_t)len]);
  memcpy(J.norm.get(), seq, (size_t)len);
  normalise(J.norm.get(), len);
  J.hash.reset(new uint64_t[(size_t)n]);
  J.strand.reset(new int8_t[(size_t)n]);
  memcpy(J.hash.get(), hash, (size_t)n * 8);
  memcpy(J.strand.get(), strand, (size_t)n);
  thin_on_host(hash, strand, n, W, s, map_prefilter_tau(c_factor, s, W), J.h_pos, J.h_hash, J.h_strand);
  if (n_kept) *n_kept = (int64_t)J.h_pos.size();
  for (size_t i = 0; kept_pos && i < J.h_pos.size() && (int64_t)i < cap_kept; ++i) kept_pos[i] = J.h_pos[i];
  J.thinned = true;
  J.head.assign(J.norm.get(), (size_t)std::min<int64_t>(len, 2 * (int64_t)k));
  J.bounds.assign(1, 0);
  if (chunk_len > 0)
    for (int64_t b = chunk_len; b < n; b += chunk_len) J.bounds.push_back(b);
  J.bounds.push_back(n);
  const size_t nc = J.bounds.size() - 1;
  J.chunk.resize(nc);
  J.started_from.resize(nc);
  auto before = [&](int64_t x) { return (int64_t)(std::lower_bound(J.h_pos.begin(), J.h_pos.end(), x, [](uint32_t p, int64_t v) { return (int64_t)p < v; }) - J.h_pos.begin()); };
  for (size_t c = 0; c <= nc; ++c) J.cidx.push_back(before(J.bounds[c]));
  for (size_t c = 0; c < nc; ++c) J.cidx_warm.push_back(before(J.warm_from(c)));
  static const uint32_t none = 0;
  for (size_t c = 0; c < nc; ++c) {
    SeqJob::View v;
    const int64_t c0 = J.cidx_warm[c], c1 = J.cidx[c + 1];
    v.sp.pos = J.h_pos.empty() ? &none : J.h_pos.data() + c0; v.sp.hash = J.h_hash.data() + c0; v.sp.strand = J.h_strand.data() + c0; v.sp.n = (size_t)(c1 - c0);
    J.run_chunk(c, v);
  }
  J.stitch();
  if (replays) *replays = J.replays;
  const int64_t m = (int64_t)J.result.size();
  for (int64_t i = 0; i < m && i < cap; ++i) out[i] = J.result[(size_t)i];
  return m;
}

// Test hook: the closing sort of a sequence's records, as stitch() runs it (threads > 1: introsort's halves side by
// side; 1: std::sort itself).  The two must agree to the last tie.
extern "C" void wfmh_test_sort_records(wfm_minmer_t* recs, int64_t n, int threads) {
  sort_as_std(recs, recs + n, [](const wfm_minmer_t& l, const wfm_minmer_t& r) { return std::tie(l.wpos, l.wpos_end) < std::tie(r.wpos, r.wpos_end); }, threads);
}


// Test hook (CPU test-suite): the device winnower's control flow and capacities run on the host (map_winnow.hip's
// model) over the thinned stream, chunk after chunk, then the closing cut / sort / de-duplication.  Returns the number
// of records, or -1 when the device would hand the sequence back to the host (*why: wn::F_* bits; bit 31: an N in the
// first k-mers that the reference does not notice).
extern "C" int64_t wfmh_test_winnow_model(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id, const uint64_t* hash,
                                          const int8_t* strand, double c_factor, int64_t chunk_len, wfm_minmer_t* out, int64_t cap, uint32_t* why) {
  // chunk_len < 0: chunks of -chunk_len k-mers, and every second speculation counts as failed (the replay path)
  const int force_replay = chunk_len < 0;
  if (chunk_len < 0) chunk_len = -chunk_len;
  if (why) *why = 0;
  if (len < k) return 0;
  const int64_t n = len - k + 1, W = (int64_t)w - k + 1;
  std::string norm(seq, (size_t)len);
  normalise(&norm[0], len);
  if (has_unnoticed_n(norm.data(), std::min<int64_t>(len, 2 * (int64_t)k), k)) { if (why) *why = 1u << 31; return -1; }
  std::vector<uint32_t> pos; std::vector<uint64_t> hs; std::vector<int8_t> st;
  thin_on_host(hash, strand, n, W, s, map_prefilter_tau(c_factor, s, W), pos, hs, st);
  std::vector<wfm_minmer_t> recs;
  static const uint32_t none_p = 0; static const uint64_t none_h = 0; static const int8_t none_s = 0;
  const int64_t got = map_winnow_model(pos.empty() ? &none_p : pos.data(), hs.empty() ? &none_h : hs.data(), st.empty() ? &none_s : st.data(),
                                       (int64_t)pos.size(), len, k, w, s, seq_id, chunk_len, &recs, why, force_replay, nullptr);
  if (got < 0) return -1;
  finish_records(recs, w, 1);
  const int64_t m = (int64_t)recs.size();
  for (int64_t i = 0; i < m && i < cap; ++i) out[i] = recs[(size_t)i];
  return m;
}

// Test hook: the closing steps of a sequence's raw records on the host (cut, strand sign, std::sort, de-duplication), as
// finish_records runs them; returns the number of records.
extern "C" int64_t wfmh_test_finish_records(const wfm_minmer_t* raw, int64_t n, int w, wfm_minmer_t* out, int64_t cap) {
  std::vector<wfm_minmer_t> v(raw, raw + n);
  finish_records(v, w, 1);
  const int64_t m = (int64_t)v.size();
  for (int64_t i = 0; i < m && i < cap; ++i) out[i] = v[(size_t)i];
  return m;
}
