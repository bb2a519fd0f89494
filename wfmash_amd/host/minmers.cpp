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
// State (names follow the roles, not the reference's identifiers):
//   arrivals  every valid k-mer still inside (or lingering behind) the window, arrival order
//   sketch    ordered map hash -> open interval + occurrences: the <= s smallest hashes
//   pool      lazy min-heap (hash, pos) of window k-mers that are not in the sketch
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/wfmash_hip.h"
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

// seq: upper-cased / N-masked bases; hash/strand: canonical hash and strand per k-mer start as
// wfm_hash_kmers returns them (strand 0 = contains N or palindromic).
class Winnower {
 public:
  Winnower(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id, const uint64_t* hash, const int8_t* strand)
      : seq_(seq), len_(len), k_(k), w_(w), s_(s), seq_id_(seq_id), dev_hash_(hash), dev_strand_(strand), rc_((size_t)k) {}

  std::vector<wfm_minmer_t> out;  // raw interval records in emission order

  // the stream for k-mer starts [from, to)
  void advance(int64_t from, int64_t to) {
    const int k = k_, w = w_, s = s_;
    const char* seq = seq_;
    for (int64_t i = from; i < to; ++i) {
      const int64_t win = i + k - w;  // id of the window that ends with this k-mer
      if (pool_.size() > (size_t)2 * (size_t)w) {
        pool_.erase(std::remove_if(pool_.begin(), pool_.end(), [win](const PoolItem& p) { return p.pos < win; }), pool_.end());
        std::make_heap(pool_.begin(), pool_.end(), pool_after);
      }
      // canonical hash: from the device, except k-mers that contain an N the reference does not notice
      // (NOTE: no initial scan, commonFunc.hpp:473: an N inside the first k-1 bases of the sequence
      //  is only seen when it is the LAST base of a k-mer)
      uint64_t hf_min; int16_t strand; bool asym;
      if (dev_strand_[i] != 0) { hf_min = dev_hash_[i]; strand = dev_strand_[i]; asym = true; }
      else {
        bool has_n = false;
        for (int j = 0; j < k; ++j) has_n |= seq[i + j] == 'N';
        if (!has_n) { asym = false; hf_min = 0; strand = 0; }  // hashFwd == hashBwd
        else {
          for (int j = 0; j < k; ++j) {
            const char c = seq[i + j];
            rc_[(size_t)(k - 1 - j)] = (uint8_t)(c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c);
          }
          const uint64_t hf = murmur_lo((const uint8_t*)seq + i, k, 42u), hb = murmur_lo(rc_.data(), k, 42u);
          asym = hf != hb; hf_min = std::min(hf, hb); strand = hf < hb ? 1 : -1;
        }
      }
      // ---- the k-mer that fell out of the window (one per iteration) ----
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
      if (seq[i + k - 1] == 'N') ambig_ = k;
      if (asym && ambig_ == 0) {
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
      if (ambig_ > 0) --ambig_;
      // ---- keep the sketch at the s smallest hashes of the window ----
      // Expired pool entries never get chosen: the loop below clears them off the top, after it the
      // top is live, one step frees at most one sketch slot (one k-mer leaves per step; the swap
      // against the largest needs a full sketch), and a sketch with two or more free slots means
      // the pool ran empty earlier -- so the refill takes the live top and stops, or empties the pool.
      if (win >= 0) {
        while (!pool_.empty() && pool_.front().pos < win) { std::pop_heap(pool_.begin(), pool_.end(), pool_after); pool_.pop_back(); }
        if (!sketch_.empty() && !pool_.empty() && sketch_.size() == (size_t)s && pool_.front().hash < std::prev(sketch_.end())->first) {
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
        while (!pool_.empty() && sketch_.size() < (size_t)s) {
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
  const char* seq_;
  int64_t len_;
  int k_, w_, s_;
  int32_t seq_id_;
  const uint64_t* dev_hash_;
  const int8_t* dev_strand_;
  std::deque<std::tuple<uint64_t, int16_t, int64_t>> arrivals_;
  std::map<uint64_t, Open> sketch_;
  std::vector<PoolItem> pool_;
  std::vector<uint8_t> rc_;
  int ambig_ = 0;
  int64_t expired_in_refill_ = 0;

 public:
  int64_t expired_in_refill() const { return expired_in_refill_; }
};

// strand sign, chunks of at most w windows, order, de-duplication (commonFunc.hpp:660-706)
void finish_records(std::vector<wfm_minmer_t>& out, int w) {
  out.erase(std::remove_if(out.begin(), out.end(), [](const wfm_minmer_t& m) { return m.wpos < 0 || m.wpos_end < 0 || m.wpos == m.wpos_end; }), out.end());
  std::vector<wfm_minmer_t> chunks;
  for (auto& m : out) {
    m.strand = m.strand < 0 ? (int16_t)-1 : (int16_t)1;  // every non-negative tally (0 included) reads FWD (commonFunc.hpp:672)
    if (m.wpos_end > m.wpos + w) {
      const int n = (int)std::ceil(float(m.wpos_end - m.wpos) / float(w));
      for (int c = 0; c < n; ++c)
        chunks.push_back(wfm_minmer_t{m.hash, m.wpos + (int64_t)c * w, std::min(m.wpos + (int64_t)c * w + w, m.wpos_end), m.seqId, m.strand, 0});
    }
  }
  out.erase(std::remove_if(out.begin(), out.end(), [w](const wfm_minmer_t& m) { return m.wpos_end - m.wpos > w; }), out.end());
  out.insert(out.end(), chunks.begin(), chunks.end());
  std::sort(out.begin(), out.end(), [](const wfm_minmer_t& l, const wfm_minmer_t& r) { return std::tie(l.wpos, l.wpos_end) < std::tie(r.wpos, r.wpos_end); });
  out.erase(std::unique(out.begin(), out.end(), [](const wfm_minmer_t& l, const wfm_minmer_t& r) { return l.wpos == r.wpos && l.hash == r.hash; }), out.end());
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
  Winnower W(seq, len, k, w, s, seq_id, hash, strand);
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
  std::unique_ptr<char[]> norm;
  std::unique_ptr<uint64_t[]> hash;
  std::unique_ptr<int8_t[]> strand;
  std::vector<int64_t> bounds;                          // chunk j covers k-mer starts [bounds[j], bounds[j+1])
  std::vector<std::unique_ptr<Winnower>> chunk;         // state at bounds[j+1], records of the chunk
  std::vector<Winnower::Live> started_from;             // live state chunk j began with (j >= 1)
  std::atomic<int> pending{0};
  int replays = 0;
  std::vector<wfm_minmer_t> result;
  // The hashes stay on the device until the host arrays are resident: a device-to-host copy into
  // freshly allocated memory spends its time in page faults, not on PCIe, so the workers first-touch
  // the arrays slice by slice in parallel and the feeding thread then copies at link speed.
  MapHashedSeq dev;
  bool on_device = false;
  std::atomic<int> touched{0};

  // chunk j's slice of the host arrays: k-mer starts [bounds[j], bounds[j+1]) and the bases from k-1
  // past its first k-mer start to k-1 past its last (chunk 0 from base 0): every byte has one writer
  void touch(size_t j) {
    const size_t n = bounds.size() - 1;
    const int64_t base_from = j == 0 ? 0 : bounds[j] + k - 1;
    const int64_t base_to = j + 1 == n ? len : bounds[j + 1] + k - 1;
    memset(hash.get() + bounds[j], 0, (size_t)(bounds[j + 1] - bounds[j]) * 8);
    memset(strand.get() + bounds[j], 0, (size_t)(bounds[j + 1] - bounds[j]));
    if (base_to > base_from) memset(norm.get() + base_from, 0, (size_t)(base_to - base_from));
    touched.fetch_add(1, std::memory_order_release);
  }

  void plan(int64_t chunk_len) {
    bounds.assign(1, 0);
    // a chunk must dwarf its two-window warm-up; short sequences stay one stream
    if (chunk_len >= 64 * (int64_t)w && nk > 2 * chunk_len)
      for (int64_t b = chunk_len; b + chunk_len / 2 < nk; b += chunk_len) bounds.push_back(b);
    bounds.push_back(nk);
    const size_t n = bounds.size() - 1;
    chunk.resize(n);
    started_from.resize(n);
    pending.store((int)n);
  }
  void run_chunk(size_t j) {
    const int64_t warm_from = j > 0 ? std::max<int64_t>(0, bounds[j] - 2 * (int64_t)w) : 0;
    auto W = std::make_unique<Winnower>(norm.get(), len, k, w, s, seq_id, hash.get(), strand.get());
    if (j > 0) {
      W->advance(warm_from, bounds[j]);  // warm-up: records are not this chunk's
      W->out.clear();
      W->mark_open_unknown();
      started_from[j] = W->live_state(bounds[j]);
    }
    W->advance(bounds[j], bounds[j + 1]);
    chunk[j] = std::move(W);
  }
  // sequential: check every speculation against the state the previous chunk really reached
  void stitch() {
    int64_t expired = 0;
    for (const auto& c : chunk) expired += c->expired_in_refill();
    if (chunk.size() > 1 && expired > 0) {
      // an expired pool entry took part in a refill: the chunks' histories of expired entries differ
      // from the single stream's, so only the single stream is trusted
      chunk.clear();
      started_from.clear();
      replays = -1;
      winnow(norm.get(), len, k, w, s, seq_id, hash.get(), strand.get(), result);
      norm.reset(); hash.reset(); strand.reset();
      return;
    }
    for (size_t j = 1; j < chunk.size(); ++j) {
      const Winnower& prev = *chunk[j - 1];
      if (!(started_from[j] == prev.live_state(bounds[j]))) {
        auto R = std::make_unique<Winnower>(norm.get(), len, k, w, s, seq_id, hash.get(), strand.get());
        R->take_state(prev);
        R->advance(bounds[j], bounds[j + 1]);
        chunk[j] = std::move(R);
        ++replays;
      } else {
        chunk[j]->resolve_unknown_starts(prev);
      }
    }
    chunk.back()->flush_end();
    size_t total = 0;
    for (const auto& c : chunk) total += c->out.size();
    result.reserve(total);
    for (const auto& c : chunk) result.insert(result.end(), c->out.begin(), c->out.end());
    chunk.clear();
    started_from.clear();
    finish_records(result, w);
    norm.reset(); hash.reset(); strand.reset();
  }
};

int64_t chunk_length() {
  const char* e = getenv("WFM_WINNOW_CHUNK");  // k-mers per speculative chunk; 0 = one stream per sequence
  return e ? atoll(e) : (int64_t)1 << 20;
}

}  // namespace

extern "C" int64_t wfm_add_minmers(wfm_handle_t* h, const char* seq, int64_t len, int k, int w, int s, int32_t seq_id,
                                   wfm_minmer_t* out, int64_t cap) {
  const char* seqs[1] = {seq};
  return wfm_add_minmers_multi(h, seqs, &len, &seq_id, 1, k, w, s, 1, out, cap, nullptr);
}

// Many sequences at once: the calling thread feeds the GPU (one hashing pass per sequence) while
// `threads` host workers winnow what has been hashed -- sequences side by side (the reference's
// ThreadPool over buildHelper, winSketch.hpp:200-239) and, within a long sequence, its
// speculative chunks.  Output is the concatenation in input order.
extern "C" int64_t wfm_add_minmers_multi(wfm_handle_t* h, const char* const* seqs, const int64_t* lens, const int32_t* seq_ids, int64_t nseq,
                                         int k, int w, int s, int threads, wfm_minmer_t* out, int64_t cap, int64_t* counts) {
  if (!h || nseq < 0 || (nseq && (!seqs || !lens || !seq_ids)) || (cap && !out)) return WFM_E_ARG;
  if (k < 1 || k > 32 || w < k || s < 1) { wfm_set_error(h, "need 1 <= k <= 32, w >= k, s >= 1"); return WFM_E_UNSUPPORTED; }
  std::vector<std::unique_ptr<SeqJob>> jobs((size_t)nseq);
  struct Task { SeqJob* job; size_t chunk; bool touch; };
  std::deque<Task> queue;
  std::mutex mu;
  std::condition_variable cv_work, cv_room;
  bool done = false;
  int64_t inflight_bases = 0;
  const int64_t max_inflight = 1ll << 31;  // ~2 Gbp of hashed-but-not-winnowed sequence (10 B/base host, 10 B/base device)
  const int nthreads = std::max(1, threads);
  const int64_t chunk_len = nthreads > 1 ? chunk_length() : 0;
  auto worker = [&]() {
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
      if (task.touch) { J->touch(task.chunk); continue; }
      J->run_chunk(task.chunk);
      if (J->pending.fetch_sub(1) == 1) {  // last chunk of this sequence: stitch here
        J->stitch();
        {
          std::lock_guard<std::mutex> lk(mu);
          inflight_bases -= J->len;
        }
        cv_room.notify_one();
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker);
  int rc = WFM_OK;
  const auto t_start = std::chrono::steady_clock::now();
  double ms_hash = 0, ms_copy = 0;
  // Waves: (1) hash a wave of sequences on the GPU, results stay there; (2) the workers first-touch
  // the host arrays, slice by slice; (3) this thread -- the only one that talks to the GPU -- copies
  // each sequence over as soon as its arrays are resident and releases its chunks for winnowing.
  std::vector<SeqJob*> wave;
  int64_t wave_bases = 0;
  const int64_t wave_limit = 1ll << 29;
  auto release_wave = [&]() {
    if (wave.empty()) return;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv_room.wait(lk, [&] { return inflight_bases == 0 || inflight_bases + wave_bases <= max_inflight; });
      inflight_bases += wave_bases;
      for (SeqJob* Jp : wave)
        for (size_t c = 0; c + 1 < Jp->bounds.size(); ++c) queue.push_back(Task{Jp, c, true});
    }
    cv_work.notify_all();
    for (SeqJob* Jp : wave) {
      const int nchunks = (int)Jp->bounds.size() - 1;
      while (Jp->touched.load(std::memory_order_acquire) < nchunks) std::this_thread::yield();
      const auto t0 = std::chrono::steady_clock::now();
      const int crc = map_hashed_fetch(&Jp->dev, 0, Jp->nk, 0, Jp->len, Jp->hash.get(), Jp->strand.get(), Jp->norm.get());
      map_hashed_free(&Jp->dev);
      Jp->on_device = false;
      ms_copy += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (crc != WFM_OK && rc == WFM_OK) { rc = crc; wfm_set_error(h, "device-to-host copy of k-mer hashes failed"); }
      {
        std::lock_guard<std::mutex> lk(mu);
        for (int c = 0; c < nchunks; ++c) queue.push_back(Task{Jp, (size_t)c, false});  // winnowed even after an error: keeps the bookkeeping simple
      }
      cv_work.notify_all();
    }
    wave.clear();
    wave_bases = 0;
  };
  for (int64_t i = 0; i < nseq && rc == WFM_OK; ++i) {
    const int64_t len = lens[i];
    if (!seqs[i] || len < 0) { rc = WFM_E_ARG; break; }
    if (len < k) continue;
    auto J = std::make_unique<SeqJob>();
    J->idx = i; J->seq_id = seq_ids[i]; J->len = len; J->nk = len - k + 1; J->k = k; J->w = w; J->s = s;
    J->norm.reset(new char[(size_t)len]);
    J->hash.reset(new uint64_t[(size_t)J->nk]);
    J->strand.reset(new int8_t[(size_t)J->nk]);
    const auto t0 = std::chrono::steady_clock::now();
    rc = map_hash_sequence_device(h, seqs[i], len, k, &J->dev);  // GPU: normalise + 2 x MurmurHash3 per base
    if (rc != WFM_OK) break;
    J->on_device = true;
    ms_hash += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    J->plan(chunk_len);
    wave.push_back(J.get());
    wave_bases += len;
    jobs[(size_t)i] = std::move(J);
    if (wave_bases >= wave_limit) release_wave();
  }
  release_wave();
  {
    std::lock_guard<std::mutex> lk(mu);
    done = true;
  }
  cv_work.notify_all();
  const auto t_fed = std::chrono::steady_clock::now();
  for (auto& t : pool) t.join();
  if (getenv("WFM_DEBUG")) {
    int64_t nchunks = 0, replays = 0;
    for (const auto& J : jobs)
      if (J) { nchunks += (int64_t)J->bounds.size() - 1; replays += J->replays; }
    fprintf(stderr, "[wfm] add_minmers_multi: %lld sequences in %lld chunks (%lld replayed), %d workers: feeding %.1f ms (GPU hashing %.1f, D2H %.1f), drain %.1f ms\n",
            (long long)nseq, (long long)nchunks, (long long)replays, nthreads, std::chrono::duration<double, std::milli>(t_fed - t_start).count(),
            ms_hash, ms_copy, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fed).count());
  }
  if (rc != WFM_OK) return rc;
  int64_t total = 0;
  for (int64_t i = 0; i < nseq; ++i) {
    const size_t n = jobs[(size_t)i] ? jobs[(size_t)i]->result.size() : 0;
    if (counts) counts[i] = (int64_t)n;
    if (n && total < cap) memcpy(out + total, jobs[(size_t)i]->result.data(), (size_t)std::min<int64_t>((int64_t)n, cap - total) * sizeof(wfm_minmer_t));
    total += (int64_t)n;
  }
  return total;
}

// Test hook (CPU test-suite): the host winnowing stage on caller-supplied k-mer hashes.
// chunk_len > 0 runs the speculative chunked form (single thread) and reports replays in *replays.
extern "C" int64_t wfmh_test_winnow_chunked(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id, const uint64_t* hash,
                                            const int8_t* strand, int64_t chunk_len, wfm_minmer_t* out, int64_t cap, int* replays) {
  if (len < k) return 0;
  SeqJob J;
  J.seq_id = seq_id; J.len = len; J.nk = len - k + 1; J.k = k; J.w = w; J.s = s;
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
  for (size_t c = 0; c + 1 < J.bounds.size(); ++c) J.run_chunk(c);
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
