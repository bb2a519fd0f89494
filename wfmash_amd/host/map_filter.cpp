#include "map_filter.hpp"
#include "parallel.hpp"

#include <cstring>
#include <memory>
#include <type_traits>
#include <algorithm>
#include <chrono>
#include <thread>
#include <atomic>
#include <cmath>
#include <exception>
#include <limits>
#include <locale>
#include <cstdio>
#include <map>
#include <mutex>
#include <numeric>
#include <set>
#include <tuple>

namespace skch {

// ---------------------------------------------------------------------------------------------
// plane sweeps (filter.hpp)
// ---------------------------------------------------------------------------------------------
namespace {

// The two sweeps differ only in the axis they run along.
struct QueryAxis {
  static double score(const MappingResultsVector_t& v, int x) {
    if (v[x].blockLength <= 0 || v[x].blockNucIdentity() <= 0) return std::numeric_limits<double>::lowest();
    return v[x].blockNucIdentity() * std::log(static_cast<double>(v[x].blockLength));
  }
  // strict "ranks before": higher score, then larger query start, then larger refSeqId
  static bool before(const MappingResultsVector_t& v, int x, int y) {
    const double xs = score(v, x), ys = score(v, y);
    return std::tie(xs, v[x].queryStartPos, v[x].refSeqId) > std::tie(ys, v[y].queryStartPos, v[y].refSeqId);
  }
  static double overlap(const MappingResultsVector_t& v, int x, int y) {
    const offset_t b = std::max(v[x].queryStartPos, v[y].queryStartPos);
    const offset_t e = std::min(v[x].queryEndPos(), v[y].queryEndPos());
    const offset_t ov = std::max(0, static_cast<int>(e - b));
    const offset_t xl = v[x].queryEndPos() - v[x].queryStartPos, yl = v[y].queryEndPos() - v[y].queryStartPos;
    return static_cast<double>(ov) / std::min(xl, yl);
  }
};

struct RefAxis {
  static double score(const MappingResultsVector_t& v, int x) { return v[x].blockNucIdentity() * log(v[x].blockLength); }
  static bool before(const MappingResultsVector_t& v, int x, int y) {
    const double xs = score(v, x), ys = score(v, y);
    return std::tie(xs, v[x].refStartPos) > std::tie(ys, v[y].refStartPos);
  }
  static double overlap(const MappingResultsVector_t& v, int x, int y) {
    const offset_t b = std::max(v[x].refStartPos, v[y].refStartPos);
    const offset_t e = std::min(v[x].refEndPos(), v[y].refEndPos());
    const offset_t ov = std::max(0, static_cast<int>(e - b));
    const offset_t xl = v[x].refEndPos() - v[x].refStartPos, yl = v[y].refEndPos() - v[y].refStartPos;
    return static_cast<double>(ov) / std::min(xl, yl);
  }
};

// Sweep-line status: the mappings crossing the current position, ranked by Axis::before.  It is a
// std::set on purpose: two mappings that compare equivalent collapse into one entry, and erasing
// one erases whichever is stored -- behaviour the output depends on.
template <class Axis>
class SweepLine {
 public:
  explicit SweepLine(MappingResultsVector_t& v) : v_(v), live_(Rank{&v}) {}
  void enter(int i) { live_.insert(i); }
  void leave(int i) { live_.erase(i); }

  // Helper::markGood (filter.hpp:95-165, :381-447)
  void markGood(int secondaryToKeep, bool dropRand, double overlapThreshold) {
    auto it = live_.begin();
    const auto top = live_.begin();
    int kept = 0;
    for (; it != live_.end(); ++it) {
      const bool worse_or_seen = Axis::score(v_, *top) > Axis::score(v_, *it) || !v_[*it].discard();
      if (worse_or_seen && kept > secondaryToKeep) break;
      v_[*it].setDiscard(false);
      ++kept;
    }
    const auto first_unkept = it;
    if (overlapThreshold < 1.0) {
      for (; it != live_.end(); ++it) {
        if (it == live_.begin()) continue;
        for (auto k = live_.begin(); k != first_unkept; ++k) {
          if (Axis::overlap(v_, *it, *k) > overlapThreshold) {
            v_[*it].setOverlapped(true);
            v_[*it].setDiscard(true);
            break;
          }
        }
      }
    }
    if (kept > secondaryToKeep && dropRand) {
      // ties beyond the quota: keep the ones with the largest (score, hash, address)
      std::vector<std::tuple<double, size_t, MappingResult*>> tied;
      for (int i : live_)
        if (!v_[i].discard()) tied.emplace_back(Axis::score(v_, i), v_[i].hash(), &v_[i]);
      std::sort(tied.begin(), tied.end(), std::greater<>{});
      for (auto& t : tied) std::get<2>(t)->setDiscard(true);
      kept = 0;
      for (auto& t : tied) {
        if (kept > secondaryToKeep) break;
        std::get<2>(t)->setDiscard(false);
        ++kept;
      }
    }
  }

 private:
  struct Rank {
    const MappingResultsVector_t* v;
    bool operator()(int x, int y) const { return Axis::before(*v, x, y); }
  };
  MappingResultsVector_t& v_;
  std::set<int, Rank> live_;
};

}  // namespace

namespace Filter {
namespace query {

void filterMappings(MappingResultsVector_t& readMappings, int secondaryToKeep, bool dropRand, double overlapThreshold) {
  if (readMappings.size() <= 1) return;
  for (auto& e : readMappings) { e.setDiscard(true); e.setOverlapped(false); }
  SweepLine<QueryAxis> line(readMappings);
  // (position, BEGIN/END, mapping).  The reference's schedule also carries 2N value-initialised
  // records (filter.hpp:194); they sort first and erase mapping 0 from a still empty status.
  typedef std::tuple<offset_t, int, int> Event;
  std::vector<Event> events;
  events.reserve(2 * readMappings.size());
  for (int i = 0; i < (int)readMappings.size(); ++i) {
    events.emplace_back(readMappings[i].queryStartPos, (int)event::BEGIN, i);
    events.emplace_back(readMappings[i].queryEndPos(), (int)event::END, i);
  }
  std::sort(events.begin(), events.end());
  for (size_t a = 0; a < events.size();) {
    size_t b = a;
    while (b < events.size() && std::get<0>(events[b]) == std::get<0>(events[a])) ++b;
    for (size_t k = a; k < b; ++k) {
      if (std::get<1>(events[k]) == event::BEGIN) line.enter(std::get<2>(events[k]));
      else line.leave(std::get<2>(events[k]));
    }
    line.markGood(secondaryToKeep, dropRand, overlapThreshold);
    a = b;
  }
  readMappings.erase(std::remove_if(readMappings.begin(), readMappings.end(), [](const MappingResult& e) { return e.discard() || e.overlapped(); }),
                     readMappings.end());
}

}  // namespace query

namespace ref {

void filterMappings(MappingResultsVector_t& readMappings, const SequenceIdManager& idManager, uint16_t secondaryToKeep, bool dropRand,
                    double overlapThreshold) {
  if (readMappings.size() <= 1) return;
  for (auto& e : readMappings) e.setDiscard(true);
  SweepLine<RefAxis> line(readMappings);
  typedef std::tuple<seqno_t, offset_t, int, int> Event;  // (ref sequence, offset, BEGIN/END, mapping)
  std::vector<Event> events;
  events.reserve(2 * readMappings.size());
  for (int i = 0; i < (int)readMappings.size(); ++i) {
    const MappingResult& m = readMappings[i];
    events.emplace_back((seqno_t)m.refSeqId, (offset_t)m.refStartPos, (int)event::BEGIN, i);
    // the END event sits one base past the mapping, rolling over into the next sequence (filter.hpp:455-468)
    seqno_t s = (seqno_t)m.refSeqId;
    offset_t o = m.refEndPos();
    if (o == idManager.getSequenceLength(s) - 1) { s += 1; o = 0; } else { o += 1; }
    events.emplace_back(s, o, (int)event::END, i);
  }
  std::sort(events.begin(), events.end());
  for (size_t a = 0; a < events.size();) {
    size_t b = a;
    while (b < events.size() && std::get<0>(events[b]) == std::get<0>(events[a]) && std::get<1>(events[b]) == std::get<1>(events[a])) ++b;
    for (size_t k = a; k < b; ++k) {
      if (std::get<2>(events[k]) == event::BEGIN) line.enter(std::get<3>(events[k]));
      else line.leave(std::get<3>(events[k]));
    }
    line.markGood(secondaryToKeep, dropRand, overlapThreshold);
    a = b;
  }
  readMappings.erase(std::remove_if(readMappings.begin(), readMappings.end(), [](const MappingResult& e) { return e.discard(); }), readMappings.end());
}

}  // namespace ref
}  // namespace Filter

// ---------------------------------------------------------------------------------------------
// MappingFilterUtils (mappingFilter.hpp)
// ---------------------------------------------------------------------------------------------
void MappingFilterUtils::filterWeakMappings(MappingResultsVector_t& readMappings, int64_t min_count, const Parameters& param,
                                            const SequenceIdManager& idManager, offset_t queryLen) {
  auto weak = [&](const MappingResult& e) {
    const bool at_boundary = e.queryStartPos < param.windowLength || e.queryEndPos() > queryLen - param.windowLength ||
                             e.refStartPos < param.windowLength ||
                             e.refEndPos() > idManager.getSequenceLength(e.refSeqId) - param.windowLength;
    // mappings touching a sequence end only need half the length / half the segment count
    if (at_boundary) return e.blockLength < param.block_length / 2 || e.n_merged < min_count / 2;
    return e.blockLength < param.block_length || e.n_merged < min_count;
  };
  readMappings.erase(std::remove_if(readMappings.begin(), readMappings.end(), weak), readMappings.end());
}

void MappingFilterUtils::filterFalseHighIdentity(MappingResultsVector_t& readMappings, const Parameters& param) {
  auto mismatched = [&](const MappingResult& e) {
    const int64_t q_l = (int64_t)e.queryEndPos() - (int64_t)e.queryStartPos;
    const int64_t r_l = (int64_t)e.refEndPos() - (int64_t)e.refStartPos;
    const uint64_t delta = std::abs(r_l - q_l);
    const double len_id_bound = (1.0 - (double)delta / (((double)q_l + r_l) / 2));
    return len_id_bound < std::min(0.7, std::pow((double)param.percentageIdentity, 3.0));  // pow(float, int) is double pow under g++
  };
  readMappings.erase(std::remove_if(readMappings.begin(), readMappings.end(), mismatched), readMappings.end());
}

void MappingFilterUtils::sparsifyMappings(MappingResultsVector_t& readMappings, const Parameters& param) {
  if (param.sparsity_hash_threshold == std::numeric_limits<uint64_t>::max()) return;
  readMappings.erase(std::remove_if(readMappings.begin(), readMappings.end(),
                                    [&](const MappingResult& e) { return e.hash() > param.sparsity_hash_threshold; }),
                     readMappings.end());
}

void MappingFilterUtils::filterByGroup(MappingResultsVector_t& unfilteredMappings, MappingResultsVector_t& filteredMappings, int n_mappings,
                                       bool filter_ref, const SequenceIdManager& idManager, const Parameters& param) {
  filteredMappings.reserve(unfilteredMappings.size());
  std::sort(unfilteredMappings.begin(), unfilteredMappings.end(), [](const MappingResult& a, const MappingResult& b) {
    return std::tie(a.refSeqId, a.refStartPos) < std::tie(b.refSeqId, b.refStartPos);
  });
  if (param.filterMode == filter::MAP || param.filterMode == filter::ONETOONE) {
    // one sweep per run of consecutive target sequences of the same group (all of them without -Y)
    size_t lo = 0;
    while (lo < unfilteredMappings.size()) {
      size_t hi = unfilteredMappings.size();
      if (param.skip_prefix) {
        const int g = idManager.getRefGroup(unfilteredMappings[lo].refSeqId);
        hi = lo;
        while (hi < unfilteredMappings.size() && idManager.getRefGroup(unfilteredMappings[hi].refSeqId) == g) ++hi;
      }
      MappingResultsVector_t run(unfilteredMappings.begin() + lo, unfilteredMappings.begin() + hi);
      std::sort(run.begin(), run.end(), [](const MappingResult& a, const MappingResult& b) {
        return std::tie(a.queryStartPos, a.refSeqId, a.refStartPos) < std::tie(b.queryStartPos, b.refSeqId, b.refStartPos);
      });
      if (filter_ref) Filter::ref::filterMappings(run, idManager, n_mappings, param.dropRand, param.overlap_threshold);
      else Filter::query::filterMappings(run, n_mappings, param.dropRand, param.overlap_threshold);
      filteredMappings.insert(filteredMappings.end(), run.begin(), run.end());
      lo = hi;
    }
  }
  std::sort(filteredMappings.begin(), filteredMappings.end(), [](const MappingResult& a, const MappingResult& b) {
    const auto as = a.strand(), bs = b.strand();
    return std::tie(a.queryStartPos, a.refSeqId, a.refStartPos, as) < std::tie(b.queryStartPos, b.refSeqId, b.refStartPos, bs);
  });
}

namespace {

// union by rank; equal ranks attach the larger id under the smaller (common/dset64.hpp:87-119).
// The representative ends up as a sort key, so the tie rule is part of the output.
class ChainSets {
 public:
  explicit ChainSets(size_t n) : parent_(n), rank_(n, 0) { std::iota(parent_.begin(), parent_.end(), (uint64_t)0); }
  uint64_t find(uint64_t x) {
    while (parent_[x] != x) {
      parent_[x] = parent_[parent_[x]];
      x = parent_[x];
    }
    return x;
  }
  void unite(uint64_t a, uint64_t b) {
    a = find(a); b = find(b);
    if (a == b) return;
    if (rank_[a] > rank_[b] || (rank_[a] == rank_[b] && a < b)) std::swap(a, b);
    parent_[a] = b;  // a: lower rank, or equal rank and larger id
    if (rank_[a] == rank_[b]) ++rank_[b];
  }

 private:
  std::vector<uint64_t> parent_;
  std::vector<uint64_t> rank_;
};

// Host threads this thread may use inside one call (set_filter_threads): a batch with one chromosome-sized query has
// nothing else to give its cores to.  The passes below are split into ranges; what they compute does not depend on it.
thread_local int tl_filter_threads = 1;
// most threads one data-parallel pass of the filters is split over (WFM_FILTER_PAR_CAP for A/B runs: every pass spawns and joins its threads)
static int par_cap() { static const int v = getenv("WFM_FILTER_PAR_CAP") ? std::max(1, std::min(33, atoi(getenv("WFM_FILTER_PAR_CAP")))) : 32; return v; }

template <class F>
void par_ranges(size_t n, F fn) {  // fn(lo, hi) over a partition of [0, n)
  const size_t T = n >= ((size_t)1 << 17) ? (size_t)std::min<int>(tl_filter_threads, par_cap()) : 1;
  if (T <= 1) { fn((size_t)0, n); return; }
  wfmash_host::parallel_for(T, (int)T, [&](size_t t) { fn(n * t / T, n * (t + 1) / T); });  // (the process's pool: parallel.hpp)
}

// the same for a few heavy items (a span of a chain stands for up to thousands of mappings): split whenever there are threads
template <class F>
void par_ranges_any(size_t n, F fn) {
  const size_t T = std::min<size_t>(n / 64, (size_t)std::min<int>(tl_filter_threads, par_cap()));
  if (T <= 1) { fn((size_t)0, n); return; }
  wfmash_host::parallel_for(T, (int)T, [&](size_t t) { fn(n * t / T, n * (t + 1) / T); });
}

// the same partition with the part's number: fn(t, lo, hi); returns the number of parts
template <class F>
size_t par_parts(size_t n, F fn) {
  const size_t T = n >= ((size_t)1 << 17) ? (size_t)std::max(1, std::min<int>(tl_filter_threads, par_cap())) : 1;
  if (T <= 1) { fn((size_t)0, (size_t)0, n); return 1; }
  wfmash_host::parallel_for(T, (int)T, [&](size_t t) { fn(t, n * t / T, n * (t + 1) / T); });
  return T;
}

template <typename T>
std::vector<T> permuted(const std::vector<T>& in, const std::vector<uint32_t>& p) {
  std::vector<T> out(in.size());
  par_ranges(p.size(), [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) out[i] = in[p[i]]; });
  return out;
}

// is less(q[i-1], q[i]) true for every i?
template <class Less>
bool strictly_ascending(const std::vector<uint32_t>& q, Less less) {
  std::atomic<bool> ok{true};
  par_ranges(q.size(), [&](size_t lo, size_t hi) {
    for (size_t i = std::max<size_t>(lo, 1); i < hi && ok.load(std::memory_order_relaxed); ++i)
      if (!less(q[i - 1], q[i])) ok.store(false, std::memory_order_relaxed);
  });
  return ok.load();
}

// f3, its first step: the caller may have the chaining order from the device (wfm_map_fragments_ordered) and have built
// readMappings in that order already; presorted_orig[i] = the position mapping i had in the reference's input order (fragment order:
// the ids the chain representatives are made of).  Consumed by the next chain_mappings call of this thread.
static thread_local const uint32_t* tl_presorted = nullptr;
static thread_local size_t tl_presorted_n = 0;

// Steps 1-4 of mergeMappingsInRange[WithChains] (mappingFilter.hpp:402-498, :593-675): sorts
// readMappings into chains and returns each mapping's chain representative.
std::vector<offset_t> chain_mappings(MappingResultsVector_t& readMappings, int max_dist, const Parameters& param) {
  const uint32_t* presorted = tl_presorted_n == readMappings.size() ? tl_presorted : nullptr;
  tl_presorted = nullptr; tl_presorted_n = 0;
  const size_t n = readMappings.size();
  static const bool tdbg = getenv("WFM_FILTER_TIMES") != nullptr;
  auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tt[8] = {0}; int ti = 0; tt[ti++] = tnow();
  std::vector<offset_t> chainOf(n);
  // (round 6: the link arrays and the permutations' scratch are raw memory filled by all threads -- a std::vector's constructor writes its 14 - 48 MB
  // on one thread, 30 ms of a chromosome-sized query's 150)
  struct RawFree { void operator()(void* q) const { free(q); } };
  std::unique_ptr<double, RawFree> linkScore_mem((double*)malloc(std::max<size_t>(n, 1) * sizeof(double)));
  std::unique_ptr<int64_t, RawFree> linkFrom_mem((int64_t*)malloc(std::max<size_t>(n, 1) * sizeof(int64_t)));
  if (!linkScore_mem || !linkFrom_mem) throw std::bad_alloc();
  double* const linkScore = linkScore_mem.get();
  int64_t* const linkFrom = linkFrom_mem.get();
  par_ranges(n, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) { chainOf[i] = (offset_t)i; linkScore[i] = std::numeric_limits<double>::max(); linkFrom[i] = std::numeric_limits<int64_t>::min(); }
  });

  std::vector<uint32_t> p(n);
  std::iota(p.begin(), p.end(), 0u);
  bool device_order = false;
  bool strict_order = false;  // the first order was found without a sort: every key strictly above its predecessor (no ties anywhere)
  if (presorted) {
    // the device's order is THE sorted order exactly when the keys ascend strictly in it (no ties: any sort gives this permutation)
    std::atomic<bool> ok{true};
    par_ranges(n, [&](size_t lo, size_t hi) {
      for (size_t i = std::max<size_t>(lo, 1); i < hi && ok.load(std::memory_order_relaxed); ++i) {
        const MappingResult &a = readMappings[i - 1], &b = readMappings[i];
        const int32_t sa = (int32_t)a.strand(), sb = (int32_t)b.strand();
        if (!(std::tie(a.refSeqId, sa, a.queryStartPos, a.refStartPos) < std::tie(b.refSeqId, sb, b.queryStartPos, b.refStartPos))) ok.store(false, std::memory_order_relaxed);
      }
    });
    if (ok.load()) {
      device_order = true;
      strict_order = true;
      par_ranges(n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) { p[i] = presorted[i]; chainOf[i] = (offset_t)presorted[i]; } });
    } else {
      // back to the reference's input order, then everything as without the shortcut
      MappingResultsVector_t orig(n);
      par_ranges(n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) orig[presorted[i]] = readMappings[i]; });
      readMappings.swap(orig);
    }
  }
  if (!device_order) {
    // the comparator only reads four fields: from a compact key array (16 B per mapping) the index sort touches
    // a third of the memory; same comparisons, same permutation
    struct Key { uint32_t ref; int32_t strand; uint32_t q, r; };
    std::vector<Key> key(n);
    par_ranges(n, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) key[i] = {(uint32_t)readMappings[i].refSeqId, (int32_t)readMappings[i].strand(), (uint32_t)readMappings[i].queryStartPos, (uint32_t)readMappings[i].refStartPos};
    });
    auto less = [&](uint32_t i, uint32_t j) {
      const Key &a = key[i], &b = key[j];
      return std::tie(a.ref, a.strand, a.q, a.r) < std::tie(b.ref, b.strand, b.q, b.r);
    };
    // The mappings of one query arrive in fragment order, per fragment by target position: within a (target,
    // strand) group they are usually in order already.  So: group them stably (a counting pass) and check; when
    // every key is strictly above its predecessor that IS the sorted order, whatever the sort algorithm.  Any tie
    // or inversion falls back to std::sort on the original order, whose tie order the reference's output has.
    bool sorted_fast = false;
    {
      // groups in key order: (ref, strand) with strand -1 before +1; a flat table when the ids are small and the strands
      // are the two the mapper knows, an ordered map otherwise
      std::vector<uint32_t> q(n);
      uint32_t max_ref = 0;
      bool two_strands = true;
      {
        std::vector<uint32_t> part_max(33, 0);
        std::vector<char> part_two(33, 1);
        par_parts(n, [&](size_t t, size_t lo, size_t hi) {
          uint32_t m = 0; bool two = true;
          for (size_t i = lo; i < hi; ++i) { m = std::max(m, key[i].ref); two &= key[i].strand == 1 || key[i].strand == -1; }
          part_max[t] = m; part_two[t] = two;
        });
        for (size_t t = 0; t < part_max.size(); ++t) { max_ref = std::max(max_ref, part_max[t]); two_strands &= part_two[t] != 0; }
      }
      if (two_strands && (uint64_t)max_ref * 2 + 2 <= ((uint64_t)1 << 16)) {
        // a stable counting pass, every part of the array with a histogram of its own: group g's members of part t go behind
        // those of the parts before it -- the order a single pass over the array leaves
        const size_t G = (size_t)max_ref * 2 + 2;
        auto slot = [](const Key& k) { return (size_t)k.ref * 2 + (k.strand > 0 ? 1 : 0); };
        std::vector<std::vector<size_t>> hist(33);
        const size_t T = par_parts(n, [&](size_t t, size_t lo, size_t hi) {
          std::vector<size_t>& hgm = hist[t];
          hgm.assign(G, 0);
          for (size_t i = lo; i < hi; ++i) ++hgm[slot(key[i])];
        });
        size_t at = 0;
        for (size_t g = 0; g < G; ++g)
          for (size_t t = 0; t < T; ++t) { const size_t c = hist[t][g]; hist[t][g] = at; at += c; }
        par_parts(n, [&](size_t t, size_t lo, size_t hi) {
          std::vector<size_t>& pos = hist[t];
          for (size_t i = lo; i < hi; ++i) q[pos[slot(key[i])]++] = (uint32_t)i;
        });
      } else if (two_strands && (uint64_t)max_ref * 2 + 2 <= ((uint64_t)1 << 22)) {
        std::vector<size_t> start((size_t)max_ref * 2 + 3, 0);
        auto slot = [](const Key& k) { return (size_t)k.ref * 2 + (k.strand > 0 ? 1 : 0); };
        for (size_t i = 0; i < n; ++i) ++start[slot(key[i]) + 1];
        for (size_t g = 1; g < start.size(); ++g) start[g] += start[g - 1];
        for (size_t i = 0; i < n; ++i) q[start[slot(key[i])]++] = (uint32_t)i;
      } else {
        std::map<uint64_t, uint32_t> count;
        for (size_t i = 0; i < n; ++i) ++count[((uint64_t)key[i].ref << 32) | ((uint32_t)key[i].strand ^ 0x80000000u)];
        std::map<uint64_t, size_t> start;
        size_t at = 0;
        for (const auto& kv : count) { start[kv.first] = at; at += kv.second; }
        for (size_t i = 0; i < n; ++i) q[start[((uint64_t)key[i].ref << 32) | ((uint32_t)key[i].strand ^ 0x80000000u)]++] = (uint32_t)i;
      }
      sorted_fast = strictly_ascending(q, less);
      if (sorted_fast) p.swap(q);
    }
    if (!sorted_fast) std::sort(p.begin(), p.end(), less);
    strict_order = sorted_fast;
  }
  tt[ti++] = tnow();
  // (one scratch array serves both permutations of the call: the second one writes into pages the first has left behind; raw memory, copied back
  // by all threads -- see above)
  std::unique_ptr<MappingResult, RawFree> scratch_mem((MappingResult*)malloc(std::max<size_t>(n, 1) * sizeof(MappingResult)));
  if (!scratch_mem) throw std::bad_alloc();
  MappingResult* const scratch = scratch_mem.get();
  static_assert(std::is_trivially_copyable<MappingResult>::value, "permute_mappings copies raw bytes");
  auto permute_mappings = [&]() {
    par_ranges(n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) std::memcpy((void*)&scratch[i], (const void*)&readMappings[p[i]], sizeof(MappingResult)); });
    par_ranges(n, [&](size_t lo, size_t hi) { std::memcpy((void*)&readMappings[lo], (const void*)&scratch[lo], (hi - lo) * sizeof(MappingResult)); });
  };
  if (!device_order) {
    permute_mappings();
    par_ranges(n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) chainOf[i] = (offset_t)p[i]; });  // chainOf was 0 .. n-1: permuted, it is p
  }
  tt[ti++] = tnow();

  // Within one (target, strand) run every mapping links to its closest admissible successor.  The runs do not see each
  // other, so they are worked through side by side; the unions are replayed afterwards in the reference's order (a
  // mapping's link is final before the loop reaches it: only its predecessors set it), because the representative a
  // chain ends up with -- a sort key of the output -- depends on that order.
  std::vector<std::pair<size_t, size_t>> runs;
  for (size_t lo = 0; lo < n;) {
    size_t hi = lo + 1;
    while (hi < n && readMappings[hi].refSeqId == readMappings[lo].refSeqId && readMappings[hi].strand() == readMappings[lo].strand()) ++hi;
    runs.emplace_back(lo, hi);
    lo = hi;
  }
  {
    std::atomic<size_t> next{0};
    auto work = [&]() {
      for (size_t rn; (rn = next.fetch_add(1)) < runs.size();) {
        const size_t lo = runs[rn].first, hi = runs[rn].second;
        for (size_t i = lo; i < hi; ++i) {
          double best = std::numeric_limits<double>::max();
          size_t best_j = hi;
          const MappingResult& a = readMappings[i];
          for (size_t j = i + 1; j < hi; ++j) {
            const MappingResult& b = readMappings[j];
            if (b.queryStartPos > a.queryEndPos() + max_dist) break;
            int64_t q_dist = b.queryStartPos - a.queryEndPos();
            if (q_dist < 0) q_dist = 0;
            const int64_t r_dist = (a.strand() == strnd::FWD) ? (b.refStartPos - a.refEndPos()) : (a.refStartPos - b.refEndPos());
            if (q_dist <= max_dist && r_dist >= -param.windowLength / 5 && r_dist <= max_dist) {
              const double d2 = (double)q_dist * q_dist + (double)r_dist * r_dist;
              if (d2 < best && d2 < linkScore[j]) { best = d2; best_j = j; }
            }
          }
          if (best_j != hi) { linkScore[best_j] = best; linkFrom[best_j] = (int64_t)i; }  // (the predecessor's POSITION; its id is chainOf[i])
        }
      }
    };
    const size_t T = n >= ((size_t)1 << 17) ? std::min<size_t>((size_t)tl_filter_threads, runs.size()) : 1;
    wfmash_host::parallel_for(T, (int)T, [&](size_t) { work(); });
  }
  tt[ti++] = tnow();
  // The links form PATHS: a mapping has one predecessor at most (linkFrom) and is the best successor of one mapping at most (each i writes one
  // best_j), always forwards inside its run.  The reference's unions (disjoint sets, by rank, equal ranks: the larger id under the smaller --
  // ChainSets above) walk a path in its order: the first union of a path joins two singletons and leaves the SMALLER id on top with rank 1, every
  // later one hangs a singleton of rank 0 under it.  So a chain's representative is min(id of its first member, id of its second) -- no
  // sets, no finds; and with a first order without ties the members of a chain already stand in (query, target) order, so the second order
  // (chain, query start, target start) is: chains by representative, members as they stand.  Runs side by side (paths do not leave a run),
  // then a prefix sum over the representatives.  WFM_FILTER_CLOSED_FORM=0 keeps the sets and the sort (the tests hold the two against each other).
  const bool closed_form_on = !(getenv("WFM_FILTER_CLOSED_FORM") && atoi(getenv("WFM_FILTER_CLOSED_FORM")) == 0);  // (read per call: the tests run both)
  const bool closed_form = closed_form_on && strict_order && n > 0 && n < ((size_t)1 << 31);
  if (closed_form) {
    struct Raw32 { uint32_t* q; explicit Raw32(size_t k) : q((uint32_t*)malloc(std::max<size_t>(k, 1) * sizeof(uint32_t))) { if (!q) throw std::bad_alloc(); } ~Raw32() { free(q); } };
    Raw32 head_m(n), rank_m(n), rep_m(n), start_m(n + 1);
    uint32_t* const head = head_m.q; uint32_t* const rank = rank_m.q; uint32_t* const rep = rep_m.q; uint32_t* const start = start_m.q;
    par_ranges(n + 1, [&](size_t lo, size_t hi) { std::memset(start + lo, 0, (hi - lo) * sizeof(uint32_t)); });
    {
      // start[r]: the size of the chain whose representative is r (an id is the representative of its own chain or of none: the slots of
      // different runs' chains never meet); a chain's size is written again by every member, the last one's stands
      std::atomic<size_t> next{0};
      auto work = [&]() {
        for (size_t rn; (rn = next.fetch_add(1)) < runs.size();) {
          const size_t lo = runs[rn].first, hi = runs[rn].second;
          for (size_t i = lo; i < hi; ++i) {
            if (linkScore[i] != std::numeric_limits<double>::max()) {
              const size_t pr = (size_t)linkFrom[i], hd = head[pr];
              const uint32_t rk = rank[pr] + 1;
              head[i] = (uint32_t)hd; rank[i] = rk;
              if (rk == 1) { start[(size_t)chainOf[hd]] = 0; rep[hd] = (uint32_t)std::min(chainOf[hd], chainOf[i]); }
              start[rep[hd]] = rk + 1;
            } else { head[i] = (uint32_t)i; rank[i] = 0; rep[i] = (uint32_t)chainOf[i]; start[(size_t)chainOf[i]] = 1; }
          }
        }
      };
      const size_t T = n >= ((size_t)1 << 17) ? std::min<size_t>((size_t)tl_filter_threads, runs.size()) : 1;
      wfmash_host::parallel_for(T, (int)T, [&](size_t) { work(); });
    }
    tt[ti++] = tnow();
    {
      // exclusive prefix sum of the chains' sizes over the representatives, in parts
      std::vector<uint64_t> part_sum(34, 0);
      const size_t T = par_parts(n, [&](size_t t, size_t lo, size_t hi) { uint64_t a = 0; for (size_t i = lo; i < hi; ++i) a += start[i]; part_sum[t + 1] = a; });
      for (size_t t = 0; t < T; ++t) part_sum[t + 1] += part_sum[t];
      par_parts(n, [&](size_t t, size_t lo, size_t hi) { uint32_t at = (uint32_t)part_sum[t]; for (size_t i = lo; i < hi; ++i) { const uint32_t c = start[i]; start[i] = at; at += c; } });
    }
    par_ranges(n, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) { const uint32_t r = rep[head[i]]; p[start[r] + rank[i]] = (uint32_t)i; }
    });
    par_ranges(n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) chainOf[i] = (offset_t)rep[head[i]]; });
  } else {
  ChainSets sets(n);
  // (the reference unites inside the loop and once more over everything afterwards: the second round finds every pair in one
  // set already and changes no representative, so it is not repeated here)
  for (size_t i = 0; i < n; ++i)
    if (linkScore[i] != std::numeric_limits<double>::max()) sets.unite(chainOf[i], chainOf[(size_t)linkFrom[i]]);
  for (size_t i = 0; i < n; ++i) chainOf[i] = (offset_t)sets.find(chainOf[i]);
  tt[ti++] = tnow();

  std::iota(p.begin(), p.end(), 0u);
  {
    struct Key { offset_t chain; uint32_t q, r; };
    std::vector<Key> key(n);
    par_ranges(n, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) key[i] = {chainOf[i], (uint32_t)readMappings[i].queryStartPos, (uint32_t)readMappings[i].refStartPos};
    });
    auto less = [&](uint32_t i, uint32_t j) {
      const Key &a = key[i], &b = key[j];
      return std::tie(a.chain, a.q, a.r) < std::tie(b.chain, b.q, b.r);
    };
    // the same shortcut: chain ids are representatives < n, the members of a chain already stand in (query,
    // target) order; a stable counting pass by chain id and a strictness check, std::sort otherwise
    bool sorted_fast = n > 0;
    for (size_t i = 0; i < n && sorted_fast; ++i) sorted_fast = key[i].chain >= 0 && (size_t)key[i].chain < n;
    if (sorted_fast) {
      std::vector<uint32_t> start(n + 1, 0), q(n);
      for (size_t i = 0; i < n; ++i) ++start[(size_t)key[i].chain + 1];
      for (size_t c = 0; c < n; ++c) start[c + 1] += start[c];
      for (size_t i = 0; i < n; ++i) q[start[(size_t)key[i].chain]++] = (uint32_t)i;
      sorted_fast = strictly_ascending(q, less);
      if (sorted_fast) p.swap(q);
    }
    if (!sorted_fast) std::sort(p.begin(), p.end(), less);
  }
  }
  tt[ti++] = tnow();
  permute_mappings();
  chainOf = permuted(chainOf, p);
  tt[ti++] = tnow();
  if (tdbg && n >= 100000)
    fprintf(stderr, "[filter] chain_mappings n=%zu, %d threads: order %.1f, permute %.1f, links %.1f, unions %.1f, order %.1f, permute %.1f ms\n", n, tl_filter_threads,
            tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3], tt[5] - tt[4], tt[6] - tt[5]);
  return chainOf;
}

// one merged record for members [first, last] of a chain (mappingFilter.hpp:536-562)
MappingResult merge_span(const MappingResultsVector_t& m, size_t first, size_t last) {
  MappingResult merged = m[first];
  const uint32_t q_start = m[first].queryStartPos;
  const uint32_t q_end = m[last].queryEndPos();
  uint32_t r_start = m[first].refStartPos;
  uint32_t r_end = m[last].refEndPos();
  double total_id = 0, total_comp = 0;
  uint32_t total_conserved = 0;
  for (size_t k = first; k <= last; ++k) {
    total_id += m[k].getNucIdentity();
    total_comp += m[k].getKmerComplexity();
    total_conserved += m[k].conservedSketches;
    if (merged.strand() == strnd::REV) {
      r_start = std::min(r_start, m[k].refStartPos);
      r_end = std::max(r_end, (uint32_t)m[k].refEndPos());
    }
  }
  merged.queryStartPos = q_start;
  merged.refStartPos = (merged.strand() == strnd::FWD) ? r_start : m[last].refStartPos;
  merged.blockLength = std::max(q_end - q_start, r_end - r_start);
  merged.n_merged = (uint32_t)(last - first + 1);
  merged.setNucIdentity(total_id / merged.n_merged);
  merged.setKmerComplexity(total_comp / merged.n_merged);
  merged.conservedSketches = total_conserved;
  return merged;
}

// The walk over the chains of a chain-sorted vector, splitting each at max_mapping_length (mappingFilter.hpp:500-571).
// The same walk for a chromosome-sized query's vector, chains side by side: the parts of the vector each list the spans of the chains that BEGIN
// in them (a chain is walked by one part, whatever it reaches into), the lists joined in part order are for_each_merged_span's sequence.
// chain_no: the chain's number in the order of the vector (the reference's dense chain id: first appearance in a chain-sorted vector).
struct SpanRec { size_t i, j, first, last; uint32_t chain_no; uint32_t pos; };
std::vector<SpanRec> collect_spans(const MappingResultsVector_t& m, const std::vector<offset_t>& chainOf, const Parameters& param) {
  const size_t n = m.size();
  std::vector<std::vector<SpanRec>> part(33);
  std::vector<uint32_t> chains_in(34, 0);
  const size_t T = par_parts(n, [&](size_t t, size_t lo, size_t hi) {
    std::vector<SpanRec>& out = part[t];
    uint32_t chains = 0;
    size_t i = lo;
    while (i < hi && i > 0 && chainOf[i] == chainOf[i - 1]) ++i;  // (the chain that reaches in from the part before is that part's)
    while (i < hi) {
      size_t j = i;
      while (j + 1 < n && chainOf[j + 1] == chainOf[i]) ++j;
      uint32_t pos = 1;
      for (size_t first = i; first <= j;) {
        size_t last = first;
        while (last + 1 <= j) {
          const offset_t query_span = m[last + 1].queryEndPos() - m[first].queryStartPos;
          const offset_t ref_span = m[last + 1].refEndPos() - m[first].refStartPos;
          if (std::max(query_span, ref_span) >= param.max_mapping_length) break;  // int64 against uint64, as the reference
          ++last;
        }
        out.push_back({i, j, first, last, chains, pos++});
        first = last + 1;
      }
      ++chains;
      i = j + 1;
    }
    chains_in[t + 1] = chains;
  });
  for (size_t t = 0; t < T; ++t) chains_in[t + 1] += chains_in[t];
  size_t total = 0;
  for (size_t t = 0; t < T; ++t) total += part[t].size();
  std::vector<SpanRec> all;
  all.reserve(total);
  for (size_t t = 0; t < T; ++t)
    for (SpanRec r : part[t]) { r.chain_no += chains_in[t]; all.push_back(r); }
  return all;
}

}  // namespace

MappingsWithChains MappingFilterUtils::mergeMappingsInRangeWithChains(MappingResultsVector_t& readMappings, int max_dist, const Parameters& param) {
  MappingsWithChains result;
  if (!param.split || readMappings.size() < 2) {
    result.mappings = readMappings;
    result.chainInfo.resize(readMappings.size());
    for (size_t i = 0; i < readMappings.size(); ++i) result.chainInfo[i] = {static_cast<uint32_t>(i), 1, 1};
    return result;
  }
  const std::vector<offset_t> chainOf = chain_mappings(readMappings, max_dist, param);
  // (representative -> sequential chain id in the order of first appearance: the vector is sorted by chain, so that is the chain's number;
  // chainPos counts a chain's spans from 1 in a uint16, as the reference's does)
  const std::vector<SpanRec> spans = collect_spans(readMappings, chainOf, param);
  result.mappings.resize(spans.size());
  result.chainInfo.resize(spans.size());
  par_ranges_any(spans.size(), [&](size_t lo, size_t hi) {
    for (size_t k = lo; k < hi; ++k) {
      const SpanRec& sp = spans[k];
      result.mappings[k] = merge_span(readMappings, sp.first, sp.last);
      result.chainInfo[k] = {sp.chain_no, (uint16_t)sp.pos, (uint16_t)(sp.j - sp.i + 1)};
    }
  });
  return result;
}

MappingResultsVector_t MappingFilterUtils::mergeMappingsInRange(MappingResultsVector_t& readMappings, int max_dist, const Parameters& param) {
  if (!param.split || readMappings.size() < 2) return readMappings;
  const std::vector<offset_t> chainOf = chain_mappings(readMappings, max_dist, param);
  const std::vector<SpanRec> spans = collect_spans(readMappings, chainOf, param);
  MappingResultsVector_t out(spans.size());
  par_ranges_any(spans.size(), [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) out[k] = merge_span(readMappings, spans[k].first, spans[k].last); });
  return out;
}

namespace {

// exact nearest-anchor distance in the (query midpoint, target midpoint) plane.  The reference
// walks a KD-tree (mappingFilter.hpp:45-127); only the minimum distance is used, so anchors are
// kept sorted by x and scanned outwards until |dx| alone exceeds the best distance.
class AnchorIndex {
 public:
  explicit AnchorIndex(const MappingResultsVector_t& anchors) {
    pts_.reserve(anchors.size());
    for (const auto& m : anchors) pts_.push_back({m.queryStartPos + m.blockLength * 0.5f, m.refStartPos + m.blockLength * 0.5f});
    std::sort(pts_.begin(), pts_.end(), [](const P& a, const P& b) { return a.x < b.x; });
  }
  float nearest(float x, float y) const {
    if (pts_.empty()) return std::numeric_limits<float>::infinity();
    float best = std::numeric_limits<float>::infinity();
    const size_t mid = std::lower_bound(pts_.begin(), pts_.end(), x, [](const P& a, float v) { return a.x < v; }) - pts_.begin();
    for (size_t i = mid; i < pts_.size(); ++i) {
      if (std::abs(x - pts_[i].x) > best) break;
      best = std::min(best, dist(pts_[i], x, y));
    }
    for (size_t i = mid; i-- > 0;) {
      if (std::abs(x - pts_[i].x) > best) break;
      best = std::min(best, dist(pts_[i], x, y));
    }
    return best;
  }

 private:
  struct P { float x, y; };
  static float dist(const P& a, float x, float y) { return std::sqrt((a.x - x) * (a.x - x) + (a.y - y) * (a.y - y)); }
  std::vector<P> pts_;
};

}  // namespace

void MappingFilterUtils::filterByScaffolds(MappingResultsVector_t& readMappings, const Parameters& param, const SequenceIdManager& idManager) {
  if (param.scaffold_gap <= 0) return;
  // scaffolds: chains formed with the (much larger) scaffold gap, long enough, and surviving a
  // plane sweep of their own
  static const bool tdbg = getenv("WFM_FILTER_TIMES") != nullptr;
  auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double ts0 = tnow();
  MappingResultsVector_t work = readMappings;
  const MappingResultsVector_t originals = work;
  MappingResultsVector_t scaffolds = mergeMappingsInRange(work, (int)param.scaffold_gap, param);
  const double ts1 = tnow();
  scaffolds.erase(std::remove_if(scaffolds.begin(), scaffolds.end(),
                                 [&](const MappingResult& m) { return m.blockLength < param.scaffold_min_length; }),
                  scaffolds.end());
  if (!scaffolds.empty() && (param.filterMode == filter::MAP || param.filterMode == filter::ONETOONE)) {
    MappingResultsVector_t kept;
    Parameters sweep = param;
    sweep.overlap_threshold = param.scaffold_overlap_threshold;
    filterByGroup(scaffolds, kept, param.numMappingsForScaffold - 1, false, idManager, sweep);
    scaffolds = std::move(kept);
  }
  // anchors: the original mappings lying inside a scaffold.  The reference tests every mapping against every
  // scaffold (mappingFilter.hpp:955-975) and may list a mapping more than once; only the SET matters (the anchors
  // feed a nearest-distance query), so scaffolds are ordered by (target, strand, query start) and a mapping is
  // tested against those that start at or before it and are long enough to still cover it.
  MappingResultsVector_t anchors;
  {
    struct Box { uint32_t ref; int strand; int64_t q0, q1, r0, r1; };
    std::vector<Box> boxes;
    boxes.reserve(scaffolds.size());
    for (const auto& c : scaffolds) boxes.push_back({(uint32_t)c.refSeqId, (int)c.strand(), (int64_t)c.queryStartPos, (int64_t)c.queryEndPos(), (int64_t)c.refStartPos, (int64_t)c.refEndPos()});
    std::sort(boxes.begin(), boxes.end(), [](const Box& a, const Box& b) { return std::tie(a.ref, a.strand, a.q0) < std::tie(b.ref, b.strand, b.q0); });
    // longest query span per (target, strand) group: a scaffold starting more than that before a mapping cannot contain it
    std::vector<int64_t> group_span(boxes.size(), 0);
    for (size_t lo = 0; lo < boxes.size();) {
      size_t hi = lo;
      int64_t span = 0;
      while (hi < boxes.size() && boxes[hi].ref == boxes[lo].ref && boxes[hi].strand == boxes[lo].strand) { span = std::max(span, boxes[hi].q1 - boxes[hi].q0); ++hi; }
      for (size_t k = lo; k < hi; ++k) group_span[k] = span;
      lo = hi;
    }
    // (every mapping on its own: flags side by side, then the anchors in the originals' order)
    std::vector<char> is_anchor(originals.size(), 0);
    par_ranges_any(originals.size(), [&](size_t lo_o, size_t hi_o) {
      for (size_t oi = lo_o; oi < hi_o; ++oi) {
        const MappingResult& orig = originals[oi];
        const uint32_t ref = (uint32_t)orig.refSeqId;
        const int strand = (int)orig.strand();
        const int64_t q0 = orig.queryStartPos, q1 = orig.queryEndPos(), r0 = orig.refStartPos, r1 = orig.refEndPos();
        // last box of the group with box.q0 <= q0, then backwards while the box can still reach q1
        size_t k = (size_t)(std::upper_bound(boxes.begin(), boxes.end(), std::make_tuple(ref, strand, q0), [](const std::tuple<uint32_t, int, int64_t>& v, const Box& b) {
                              return v < std::make_tuple(b.ref, b.strand, b.q0); }) - boxes.begin());
        while (k-- > 0) {
          const Box& b = boxes[k];
          if (b.ref != ref || b.strand != strand || b.q0 + group_span[k] < q1) break;
          if (b.q1 >= q1 && b.r0 <= r0 && b.r1 >= r1) { is_anchor[oi] = 1; break; }
        }
      }
    });
    for (size_t oi = 0; oi < originals.size(); ++oi) if (is_anchor[oi]) anchors.push_back(originals[oi]);
  }
  const double ts2 = tnow();
  if (readMappings.empty()) return;
  if (anchors.empty()) { readMappings.clear(); return; }
  const AnchorIndex index(anchors);
  const float max_dist = static_cast<float>(param.scaffold_max_deviation);
  MappingResultsVector_t keepers;
  std::vector<char> near(readMappings.size(), 0);
  par_ranges_any(readMappings.size(), [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      const MappingResult& m = readMappings[i];
      near[i] = index.nearest(m.queryStartPos + m.blockLength * 0.5f, m.refStartPos + m.blockLength * 0.5f) <= max_dist;
    }
  });
  for (size_t i = 0; i < readMappings.size(); ++i) if (near[i]) keepers.push_back(readMappings[i]);
  if (tdbg && readMappings.size() >= 10000)
    fprintf(stderr, "[filter] filterByScaffolds n=%zu: scaffold chains %.1f, their sweep + anchors %.1f (%zu anchors), nearest %.1f ms\n", readMappings.size(), ts1 - ts0, ts2 - ts1, anchors.size(), tnow() - ts2);
  readMappings = std::move(keepers);
}

// ---------------------------------------------------------------------------------------------
// Map::filterSubsetMappings (computeMap.hpp:1076-1165)
// ---------------------------------------------------------------------------------------------
void set_filter_threads(int threads) { tl_filter_threads = std::max(1, threads); }
void set_presorted_order(const uint32_t* orig_index, size_t n) { tl_presorted = orig_index; tl_presorted_n = n; }

FilteredMappingsResult filterSubsetMappings(MappingResultsVector_t& mappings, const Parameters& param, const SequenceIdManager& idManager,
                                            offset_t queryLen) {
  FilteredMappingsResult result;
  if (mappings.empty()) return result;
  static const bool tdbg = getenv("WFM_FILTER_TIMES") != nullptr;
  auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const size_t n_in = mappings.size();
  double tt[8] = {0}; int ti = 0; tt[ti++] = tnow();
  MappingsWithChains chained = MappingFilterUtils::mergeMappingsInRangeWithChains(mappings, (int)param.chain_gap, param);
  tt[ti++] = tnow();
  MappingResultsVector_t& merged = chained.mappings;
  if (param.mergeMappings && param.split) {
    MappingFilterUtils::filterWeakMappings(merged, (int64_t)std::floor(param.block_length / param.windowLength), param, idManager, queryLen);
    tt[ti++] = tnow();
    if (param.filterMode == filter::MAP || param.filterMode == filter::ONETOONE) {
      MappingResultsVector_t kept;
      // -n inf: uint32 max - 1 lands in an int as -2 (SURVEY 8a parity hazards)
      MappingFilterUtils::filterByGroup(merged, kept, param.numMappingsForSegment - 1, false, idManager, param);
      merged = std::move(kept);
    }
    tt[ti++] = tnow();
    if (param.filterLengthMismatches) MappingFilterUtils::filterFalseHighIdentity(merged, param);
    MappingFilterUtils::sparsifyMappings(merged, param);
    MappingFilterUtils::filterByScaffolds(merged, param, idManager);
    tt[ti++] = tnow();
    if (tdbg && n_in >= 100000)
      fprintf(stderr, "[filter] filterSubsetMappings n=%zu: chains + merge %.1f, weak %.1f, sweep %.1f, scaffolds %.1f ms\n", n_in, tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3]);
  } else {
    if (param.filterMode == filter::MAP || param.filterMode == filter::ONETOONE) {
      MappingResultsVector_t kept;
      MappingFilterUtils::filterByGroup(mappings, kept, param.numMappingsForSegment - 1, false, idManager, param);
      mappings = std::move(kept);
    }
    MappingFilterUtils::filterByScaffolds(mappings, param, idManager);
  }
  result.nonMergedMappings = std::move(mappings);
  result.mergedMappings = std::move(merged);
  result.nonMergedChainInfo.resize(result.nonMergedMappings.size());
  for (size_t i = 0; i < result.nonMergedMappings.size(); ++i) result.nonMergedChainInfo[i] = {static_cast<uint32_t>(i), 1, 1};
  result.mergedChainInfo = std::move(chained.chainInfo);
  return result;
}

// ---------------------------------------------------------------------------------------------
// MappingOutput (mappingOutput.hpp)
// ---------------------------------------------------------------------------------------------
void MappingOutput::mappingBoundarySanityCheck(offset_t queryLen, MappingResultsVector_t& readMappings, const SequenceIdManager& idManager) {
  // (every mapping on its own: a chromosome-sized query's 1.7 M of them are split over the threads this thread may use)
  std::exception_ptr failed;  // (an id the manager does not know throws: it must reach the caller, not end a helper thread)
  std::mutex failed_mu;
  par_ranges(readMappings.size(), [&](size_t lo, size_t hi) {
   try {
    for (size_t i = lo; i < hi; ++i) {
      MappingResult& e = readMappings[i];
      const offset_t refLen = idManager.getSequenceLength(e.refSeqId);
      if (e.refStartPos >= refLen) e.refStartPos = refLen - 1;
      if (e.refEndPos() < e.refStartPos) e.blockLength = 0;
      if (e.refEndPos() >= refLen) e.blockLength = refLen - 1 - e.refStartPos;
      if (e.queryStartPos >= queryLen) e.queryStartPos = queryLen;
      if (e.queryEndPos() < e.queryStartPos) e.blockLength = 0;
      if (e.queryEndPos() >= queryLen) e.blockLength = queryLen - e.queryStartPos;
    }
   } catch (...) {
    std::lock_guard<std::mutex> lk(failed_mu);
    if (!failed) failed = std::current_exception();
   }
  });
  if (failed) std::rethrow_exception(failed);
}

namespace {
// what operator<< writes for these types on a stream in its default state (decimal integers; floating point as %g with six digits)
inline void put_int(std::string& s, long long v) { char b[24]; const int k = snprintf(b, sizeof b, "%lld", v); s.append(b, (size_t)k); }
inline void put_uint(std::string& s, unsigned long long v) { char b[24]; const int k = snprintf(b, sizeof b, "%llu", v); s.append(b, (size_t)k); }
inline void put_real(std::string& s, double v) { char b[48]; const int k = snprintf(b, sizeof b, "%g", v); s.append(b, (size_t)k); }
}  // namespace

void MappingOutput::reportReadMappings(MappingResultsVector_t& readMappings, const ChainInfoVector_t& chainInfo, const std::string& queryName,
                                       std::ostream& outstrm, const SequenceIdManager& idManager, const Parameters& param, offset_t queryLen) {
  std::vector<size_t> order(readMappings.size());
  std::iota(order.begin(), order.end(), (size_t)0);
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return readMappings[a].queryStartPos < readMappings[b].queryStartPos; });
  const std::string sep = param.legacy_output ? " " : "\t";
  // A stream in its default state (what the mapper hands in) gets the records as text made side by side: the parts of `order` each into a
  // string of their own, written one after the other -- a chromosome-sized query's 36 k records were 26 ms of operator<< on one thread.  Any
  // other stream state takes the operators themselves.
  const bool plain = outstrm.flags() == (std::ios_base::skipws | std::ios_base::dec) && outstrm.precision() == 6 && outstrm.width() == 0 && outstrm.fill() == ' ' &&
                     outstrm.getloc() == std::locale::classic();
  if (plain) {
    const int legacy = param.legacy_output ? 1 : 0;
    std::vector<std::string> part(33);
    auto format = [&](size_t t, size_t lo, size_t hi) {
      std::string& o = part[t];
      o.reserve((hi - lo) * (queryName.size() + 160));
      for (size_t k = lo; k < hi; ++k) {
        const size_t idx = order[k];
        const MappingResult& e = readMappings[idx];
        const ChainInfo& chain = chainInfo[idx];
        const float fakeMapQ = e.getNucIdentity() == 1 ? 255 : std::round(-10.0 * std::log10(1 - (e.getNucIdentity())));
        o += queryName; o += sep; put_int(o, (long long)queryLen); o += sep; put_uint(o, e.queryStartPos); o += sep; put_int(o, (long long)(e.queryEndPos() - legacy)); o += sep;
        o += (e.strand() == strnd::FWD ? "+" : "-"); o += sep; o += idManager.getSequenceName(e.refSeqId); o += sep;
        put_int(o, (long long)idManager.getSequenceLength(e.refSeqId)); o += sep; put_uint(o, e.refStartPos); o += sep; put_int(o, (long long)(e.refEndPos() - legacy));
        if (!param.legacy_output) {
          o += sep; put_uint(o, e.conservedSketches); o += sep; put_uint(o, e.blockLength); o += sep; put_real(o, (double)fakeMapQ); o += sep; o += "id:f:"; put_real(o, (double)e.getNucIdentity());
          o += sep; o += "kc:f:"; put_real(o, (double)e.getKmerComplexity());
          if (!param.mergeMappings) { o += sep; o += "jc:f:"; put_real(o, 0.0); }
          else { o += sep; o += "ch:Z:"; put_uint(o, chain.chainId); o += "."; put_uint(o, chain.chainPos); o += "."; put_uint(o, chain.chainLen); }
        } else {
          o += sep; put_real(o, e.nucIdentity * 100.0);
        }
        o += "\n";
      }
    };
    const size_t n = order.size();
    const size_t T = std::max<size_t>(1, std::min<size_t>(n / 512, (size_t)std::min<int>(tl_filter_threads, par_cap())));
    if (T <= 1) format(0, 0, n);
    else wfmash_host::parallel_for(T, (int)T, [&](size_t t) { format(t, n * t / T, n * (t + 1) / T); });
    for (size_t t = 0; t < T; ++t) outstrm.write(part[t].data(), (std::streamsize)part[t].size());
    return;
  }
  for (size_t idx : order) {
    const MappingResult& e = readMappings[idx];
    const ChainInfo& chain = chainInfo[idx];
    const float fakeMapQ = e.getNucIdentity() == 1 ? 255 : std::round(-10.0 * std::log10(1 - (e.getNucIdentity())));
    outstrm << queryName << sep << queryLen << sep << e.queryStartPos << sep << e.queryEndPos() - (param.legacy_output ? 1 : 0) << sep
            << (e.strand() == strnd::FWD ? "+" : "-") << sep << idManager.getSequenceName(e.refSeqId) << sep
            << idManager.getSequenceLength(e.refSeqId) << sep << e.refStartPos << sep << e.refEndPos() - (param.legacy_output ? 1 : 0);
    if (!param.legacy_output) {
      outstrm << sep << e.conservedSketches << sep << e.blockLength << sep << fakeMapQ << sep << "id:f:" << e.getNucIdentity() << sep
              << "kc:f:" << e.getKmerComplexity();
      if (!param.mergeMappings) outstrm << sep << "jc:f:" << 0.0;
      else outstrm << sep << "ch:Z:" << chain.chainId << "." << chain.chainPos << "." << chain.chainLen;
    } else {
      outstrm << sep << e.nucIdentity * 100.0;
    }
    outstrm << "\n";
  }
}

void MappingOutput::reportReadMappings(MappingResultsVector_t& readMappings, const std::string& queryName, std::ostream& outstrm,
                                       const SequenceIdManager& idManager, const Parameters& param, offset_t queryLen) {
  ChainInfoVector_t own(readMappings.size());
  for (size_t i = 0; i < readMappings.size(); ++i) own[i] = {static_cast<uint32_t>(i), 1, 1};
  reportReadMappings(readMappings, own, queryName, outstrm, idManager, param, queryLen);
}

}  // namespace skch
