// minmers.cpp -- winnowed minmer intervals of a target sequence.
//
// Restates CommonFunc::addMinmers (src/map/include/commonFunc.hpp:440-708).  The k-mer
// hashing (two MurmurHash3 per base, the expensive part) runs on the GPU
// (kmer_hash_kernel through wfm_hash_kmers); the sliding-window bookkeeping stays on the host:
// the reference's algorithm is a sequential stream whose lazy heap clean-up, strand-tally
// splits and tie handling shape the output, so a data-parallel
// reformulation would not be bit-exact.  One sequence per host thread is the reference's own
// parallelism (winSketch.hpp:200-239).
//
// State (names follow the roles, not the reference's identifiers):
//   arrivals  every valid k-mer still inside (or lingering behind) the window, arrival order
//   sketch    ordered map hash -> open interval + occurrences: the <= s smallest hashes
//   pool      lazy min-heap (hash, pos) of window k-mers that are not in the sketch
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/wfmash_hip.h"
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

struct Occ { int64_t pos; int16_t strand; };
struct PoolItem { uint64_t hash; int64_t pos; int16_t strand; };
struct Open { wfm_minmer_t mi; std::deque<Occ> occ; };

inline bool pool_after(const PoolItem& a, const PoolItem& b) {  // min-heap on (hash, pos)
  return std::tie(a.hash, a.pos) > std::tie(b.hash, b.pos);
}

}  // namespace

// seq: upper-cased / N-masked bases; hash/strand: canonical hash and strand per k-mer start as
// wfm_hash_kmers returns them (strand 0 = contains N or palindromic).
static void winnow(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id,
                   const uint64_t* dev_hash, const int8_t* dev_strand, std::vector<wfm_minmer_t>& out) {
  std::deque<std::tuple<uint64_t, int16_t, int64_t>> arrivals;
  std::map<uint64_t, Open> sketch;
  std::vector<PoolItem> pool;
  std::vector<uint8_t> rc((size_t)k);
  int ambig = 0;  // NOTE: no initial scan here (commonFunc.hpp:473): an N inside the first k-1 bases
                  // is only seen when it is the LAST base of a k-mer
  for (int64_t i = 0; i + k <= len; ++i) {
    const int64_t win = i + k - w;  // id of the window that ends with this k-mer
    if (pool.size() > (size_t)2 * (size_t)w) {
      pool.erase(std::remove_if(pool.begin(), pool.end(), [win](const PoolItem& p) { return p.pos < win; }), pool.end());
      std::make_heap(pool.begin(), pool.end(), pool_after);
    }
    // canonical hash: from the device, except k-mers that contain an N the reference does not notice
    uint64_t hf_min; int16_t strand; bool asym;
    if (dev_strand[i] != 0) { hf_min = dev_hash[i]; strand = dev_strand[i]; asym = true; }
    else {
      bool has_n = false;
      for (int j = 0; j < k; ++j) has_n |= seq[i + j] == 'N';
      if (!has_n) { asym = false; hf_min = 0; strand = 0; }  // hashFwd == hashBwd
      else {
        for (int j = 0; j < k; ++j) {
          const char c = seq[i + j];
          rc[(size_t)(k - 1 - j)] = (uint8_t)(c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c);
        }
        const uint64_t hf = murmur_lo((const uint8_t*)seq + i, k, 42u), hb = murmur_lo(rc.data(), k, 42u);
        asym = hf != hb; hf_min = std::min(hf, hb); strand = hf < hb ? 1 : -1;
      }
    }
    // ---- the k-mer that fell out of the window (one per iteration) ----
    if (!arrivals.empty() && std::get<2>(arrivals.front()) < win) {
      const uint64_t lh = std::get<0>(arrivals.front());
      const int16_t ls = std::get<1>(arrivals.front());
      if (!sketch.empty() && lh <= std::prev(sketch.end())->first) {
        auto it = sketch.find(lh);
        if (it != sketch.end()) {
          Open& o = it->second;
          if (o.occ.size() == 1) {
            o.mi.wpos_end = win;
            out.push_back(o.mi);
            sketch.erase(it);
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
      arrivals.pop_front();
    }
    if (seq[i + k - 1] == 'N') ambig = k;
    if (asym && ambig == 0) {
      arrivals.emplace_back(hf_min, strand, i);
      auto it = sketch.find(hf_min);
      if (it != sketch.end()) {
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
        pool.push_back(PoolItem{hf_min, i, strand});
        std::push_heap(pool.begin(), pool.end(), pool_after);
      }
    }
    if (ambig > 0) --ambig;
    // ---- keep the sketch at the s smallest hashes of the window ----
    if (win >= 0) {
      while (!pool.empty() && pool.front().pos < win) { std::pop_heap(pool.begin(), pool.end(), pool_after); pool.pop_back(); }
      if (!sketch.empty() && !pool.empty() && sketch.size() == (size_t)s && pool.front().hash < std::prev(sketch.end())->first) {
        auto last = std::prev(sketch.end());
        last->second.mi.wpos_end = win;
        out.push_back(last->second.mi);
        for (const Occ& oc : last->second.occ) {
          if (oc.pos > win) {  // strictly greater, as the reference (commonFunc.hpp:615)
            pool.push_back(PoolItem{last->first, oc.pos, oc.strand});
            std::push_heap(pool.begin(), pool.end(), pool_after);
          }
        }
        sketch.erase(last);
      }
      while (!pool.empty() && sketch.size() < (size_t)s) {
        if (pool.front().pos < win) {  // drops ONE expired item, then takes whatever is on top (commonFunc.hpp:627-633)
          std::pop_heap(pool.begin(), pool.end(), pool_after);
          pool.pop_back();
          if (pool.empty()) break;  // the reference reads an empty heap here (undefined); stop instead
        }
        const PoolItem top = pool.front();
        Open& o = sketch[top.hash];
        o.mi = wfm_minmer_t{top.hash, win, -1, seq_id, 0, 0};
        while (!pool.empty() && pool.front().hash == top.hash) {
          o.occ.push_back(Occ{pool.front().pos, pool.front().strand});
          o.mi.strand = (int16_t)(o.mi.strand + pool.front().strand);
          std::pop_heap(pool.begin(), pool.end(), pool_after);
          pool.pop_back();
        }
      }
    }
  }
  // remaining open intervals close at len - k + 1 (commonFunc.hpp:647-658)
  {
    uint64_t rank = 1;
    for (auto it = sketch.begin(); it != sketch.end() && rank <= (uint64_t)s; ++it, ++rank) {
      if (it->second.mi.wpos != -1) {
        it->second.mi.wpos_end = len - k + 1;
        out.push_back(it->second.mi);
      }
    }
  }
  out.erase(std::remove_if(out.begin(), out.end(), [](const wfm_minmer_t& m) { return m.wpos < 0 || m.wpos_end < 0 || m.wpos == m.wpos_end; }), out.end());
  // strand sign, then chunks of at most w windows (commonFunc.hpp:670-693)
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

extern "C" int64_t wfm_add_minmers(wfm_handle_t* h, const char* seq, int64_t len, int k, int w, int s, int32_t seq_id,
                                   wfm_minmer_t* out, int64_t cap) {
  if (!h || !seq || len < 0 || (cap && !out)) return WFM_E_ARG;
  if (k < 1 || k > 32 || w < k || s < 1) { wfm_set_error(h, "need 1 <= k <= 32, w >= k, s >= 1"); return WFM_E_UNSUPPORTED; }
  if (len < k) return 0;
  const int64_t nk = len - k + 1;
  std::vector<uint64_t> hash((size_t)nk);
  std::vector<int8_t> strand((size_t)nk);
  const int rc = wfm_hash_kmers(h, seq, len, k, hash.data(), strand.data());  // GPU: normalise + 2 x MurmurHash3 per base
  if (rc != WFM_OK) return rc;
  std::string norm(seq, (size_t)len);
  for (auto& c : norm) {  // makeUpperCaseAndValidDNA (commonFunc.hpp:132-142)
    if (c > 96 && c < 123) c -= 32;
    if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) c = 'N';
  }
  std::vector<wfm_minmer_t> res;
  winnow(norm.data(), len, k, w, s, seq_id, hash.data(), strand.data(), res);
  const int64_t n = (int64_t)res.size();
  for (int64_t i = 0; i < n && i < cap; ++i) out[i] = res[(size_t)i];
  return n;
}

// Many sequences at once: the calling thread feeds the GPU (one hashing pass per sequence) while
// `threads` host workers winnow the sequences already hashed -- one sequence per worker, the
// reference's own parallelism (winSketch.hpp:200-239, ThreadPool over buildHelper).  Output is the
// concatenation in input order, whatever order the workers finish in.
extern "C" int64_t wfm_add_minmers_multi(wfm_handle_t* h, const char* const* seqs, const int64_t* lens, const int32_t* seq_ids, int64_t nseq,
                                         int k, int w, int s, int threads, wfm_minmer_t* out, int64_t cap, int64_t* counts) {
  if (!h || nseq < 0 || (nseq && (!seqs || !lens || !seq_ids)) || (cap && !out)) return WFM_E_ARG;
  if (k < 1 || k > 32 || w < k || s < 1) { wfm_set_error(h, "need 1 <= k <= 32, w >= k, s >= 1"); return WFM_E_UNSUPPORTED; }
  struct Job {
    int64_t idx;
    std::string norm;
    std::vector<uint64_t> hash;
    std::vector<int8_t> strand;
  };
  std::vector<std::vector<wfm_minmer_t>> results((size_t)nseq);
  std::deque<std::unique_ptr<Job>> queue;
  std::mutex mu;
  std::condition_variable cv_work, cv_room;
  bool done = false;
  int64_t inflight_bases = 0;
  const int64_t max_inflight = 1ll << 31;  // ~2 Gbp of hashed-but-not-winnowed sequence (9 B/base of hashes)
  const int nthreads = std::max(1, threads);
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; ++t) {
    pool.emplace_back([&]() {
      for (;;) {
        std::unique_ptr<Job> job;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv_work.wait(lk, [&] { return done || !queue.empty(); });
          if (queue.empty()) return;
          job = std::move(queue.front());
          queue.pop_front();
        }
        const int64_t len = (int64_t)job->norm.size();
        for (auto& c : job->norm) {  // makeUpperCaseAndValidDNA (commonFunc.hpp:132-142)
          if (c > 96 && c < 123) c -= 32;
          if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) c = 'N';
        }
        winnow(job->norm.data(), len, k, w, s, seq_ids[job->idx], job->hash.data(), job->strand.data(), results[(size_t)job->idx]);
        {
          std::lock_guard<std::mutex> lk(mu);
          inflight_bases -= len;
        }
        cv_room.notify_one();
      }
    });
  }
  int rc = WFM_OK;
  for (int64_t i = 0; i < nseq && rc == WFM_OK; ++i) {
    const int64_t len = lens[i];
    if (!seqs[i] || len < 0) { rc = WFM_E_ARG; break; }
    if (len < k) continue;
    auto job = std::make_unique<Job>();
    job->idx = i;
    const int64_t nk = len - k + 1;
    job->hash.resize((size_t)nk);
    job->strand.resize((size_t)nk);
    rc = wfm_hash_kmers(h, seqs[i], len, k, job->hash.data(), job->strand.data());
    if (rc != WFM_OK) break;
    job->norm.assign(seqs[i], (size_t)len);  // upper-cased / N-masked by the worker
    {
      std::unique_lock<std::mutex> lk(mu);
      cv_room.wait(lk, [&] { return inflight_bases == 0 || inflight_bases + len <= max_inflight; });
      inflight_bases += len;
      queue.push_back(std::move(job));
    }
    cv_work.notify_one();
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    done = true;
  }
  cv_work.notify_all();
  for (auto& t : pool) t.join();
  if (rc != WFM_OK) return rc;
  int64_t total = 0;
  for (int64_t i = 0; i < nseq; ++i) {
    const auto& r = results[(size_t)i];
    if (counts) counts[i] = (int64_t)r.size();
    for (size_t j = 0; j < r.size(); ++j)
      if (total + (int64_t)j < cap) out[total + (int64_t)j] = r[j];
    total += (int64_t)r.size();
  }
  return total;
}

// Test hook (CPU test-suite): the host winnowing stage on caller-supplied k-mer hashes.
extern "C" int64_t wfmh_test_winnow(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id,
                                    const uint64_t* hash, const int8_t* strand, wfm_minmer_t* out, int64_t cap) {
  if (len < k) return 0;
  std::string norm(seq, (size_t)len);
  for (auto& c : norm) {
    if (c > 96 && c < 123) c -= 32;
    if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) c = 'N';
  }
  std::vector<wfm_minmer_t> res;
  winnow(norm.data(), len, k, w, s, seq_id, hash, strand, res);
  const int64_t n = (int64_t)res.size();
  for (int64_t i = 0; i < n && i < cap; ++i) out[i] = res[(size_t)i];
  return n;
}
