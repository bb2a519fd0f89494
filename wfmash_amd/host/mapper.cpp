#include "mapper.hpp"
#include "parallel.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <functional>
#include <deque>
#include <condition_variable>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <unordered_map>

#include "../csrc/wfa_handle.h"
#include "fasta.hpp"
#include "index_file.hpp"
#include "map_filter.hpp"
#include "map_stats.hpp"

namespace skch {

namespace {

constexpr float kConfidenceInterval = 0.95f;  // fixed::confidence_interval (map_parameters.hpp:125)
// (what a batch's mappings come back in: sized for the most a batch can yield -- 16 per fragment and more --, filled by the copy from the device up to
// what it did yield.  A std::vector writes all of it first: 144 MB of zeros and page faults per chromosome-sized query, on the device thread, before the
// mapping it waits for begins.  Raw memory, untouched beyond what the copy writes)
template <class T>
struct RawVec {
  T* p = nullptr; size_t n = 0, cap = 0;
  RawVec() = default;
  RawVec(const RawVec&) = delete;
  RawVec& operator=(const RawVec&) = delete;
  ~RawVec() { free(p); }
  void resize(size_t k) {
    if (k > cap) { free(p); p = (T*)malloc(k * sizeof(T)); if (!p) { cap = n = 0; throw std::bad_alloc(); } cap = k; }  // (never grown with content to keep: sized, then filled)
    n = k;
  }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T* data() { return p; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};

constexpr int64_t kBatchBases = 256ll << 20;  // query bases per wfm_map_fragments call

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// all FASTA files of a run, loaded once; name -> sequence
class SequenceSource {
 public:
  void add(const std::string& path) {
    if (stores_.count(path)) return;
    stores_.emplace(path, wfmash_host::open_shared(path));
    order_.push_back(path);
  }
  // the first file holding `name`, restricted to `files`
  bool find(const std::vector<std::string>& files, const std::string& name, wfmash_host::SeqView* seq) const {
    for (const auto& f : files) {
      const auto& st = *stores_.at(f);
      const int i = st.find(name);
      if (i >= 0) { *seq = st.sequence(i); return true; }
    }
    return false;
  }
  // indexed files: read the sequences find() will be asked for, side by side
  void preload(const std::vector<std::string>& files, const std::vector<std::string>& names, int threads) const {
    std::unordered_map<std::string, bool> seen;
    for (const auto& f : files) {
      const auto& st = *stores_.at(f);
      std::vector<int> which;
      for (const auto& n : names) {
        const int i = st.find(n);
        if (i >= 0 && !seen[n]) { which.push_back(i); seen[n] = true; }
      }
      if (!which.empty()) st.preload(which, threads);
    }
  }

 private:
  std::unordered_map<std::string, std::shared_ptr<wfmash_host::FastaStore>> stores_;
  std::vector<std::string> order_;
};

struct DeviceTables {
  std::vector<int32_t> ref_group, min_hits, cutoffs;
  std::vector<uint8_t> keep;
  std::vector<uint16_t> ident;
  std::vector<double> cutoff_j;
  wfm_map_params_t prm;
};

}  // namespace

Map::Map(const Parameters& p, wfm_handle_t* h) : Map(p, std::vector<wfm_handle_t*>{h}) {}

Map::Map(const Parameters& p, const std::vector<wfm_handle_t*>& hs) : param_(p), h_(hs.empty() ? nullptr : hs.front()), hs_(hs) {
  if (!h_ || std::find(hs_.begin(), hs_.end(), nullptr) != hs_.end()) throw std::runtime_error("no GPU handle");
  if (param_.querySequences.empty()) param_.querySequences = param_.refSequences;  // all-vs-all
  if (param_.sketchSize <= 0) {
    const double md = 1 - param_.percentageIdentity;
    const double dens = 0.02 * (1 + (md / 0.1));
    param_.sketchSize = dens * (param_.windowLength - param_.kmerSize);
  }
  if (param_.sketchSize < 1 || param_.sketchSize > param_.windowLength) throw std::runtime_error("sketch size must be in 1..window size");
  idManager_ = std::make_unique<SequenceIdManager>(param_.querySequences, param_.refSequences, param_.query_prefix,
                                                   std::vector<std::string>{param_.target_prefix}, std::string(1, param_.prefix_delim),
                                                   param_.query_list, param_.target_list);
  cached_minimum_hits_ = std::max(param_.minimum_hits, Stat::estimateMinimumHitsRelaxed(param_.sketchSize, param_.kmerSize,
                                                                                         param_.percentageIdentity, kConfidenceInterval));
}

int Map::mapQuery(MapSummary* summary) {
  MapSummary sum;
  const double t_begin = now_ms();
  const Parameters& P = param_;
  const SequenceIdManager& ids = *idManager_;
  const int S = P.sketchSize, k = P.kmerSize;
  const int64_t w = P.windowLength;

  // names as Map's constructor selects them (computeMap.hpp:162-190)
  std::vector<std::string> queryNames, targetNames;
  for (const auto& n : ids.getQuerySequenceNames()) {
    bool ok = P.query_prefix.empty();
    for (const auto& pre : P.query_prefix) ok = ok || n.compare(0, pre.size(), pre) == 0;
    if (ok) queryNames.push_back(n);
  }
  for (const auto& n : ids.getTargetSequenceNames())
    if (P.target_prefix.empty() || n.compare(0, P.target_prefix.size(), P.target_prefix) == 0) targetNames.push_back(n);

  SequenceSource src;
  for (const auto& f : P.refSequences) src.add(f);
  for (const auto& f : P.querySequences) src.add(f);
  src.preload(P.refSequences, targetNames, P.threads);
  src.preload(P.querySequences, queryNames, P.threads);

  // thresholds and tables the kernels take
  DeviceTables T;
  T.ref_group = ids.refGroupTable();
  T.min_hits.assign((size_t)S + 1, 0);
  for (int q = 1; q <= S; ++q)
    T.min_hits[q] = std::max(P.minimum_hits, Stat::estimateMinimumHitsRelaxed(q, k, P.percentageIdentity, kConfidenceInterval));
  if (P.stage1_topANI_filter) {
    const std::vector<int> c = Stat::sketch_cutoffs(S, k, P.ANIDiff, P.ANIDiffConf);
    T.cutoffs.assign(c.begin(), c.end());
  } else {
    T.cutoffs.assign((size_t)std::min<double>(S, 1000.0) + 1, 1);
  }
  Stat::l2_identity_tables(S, k, P.percentageIdentity, P.keep_low_pct_id, kConfidenceInterval, T.keep, T.ident);
  T.cutoff_j.assign((size_t)S + 1, 0.0);
  for (int q = 1; q <= S; ++q) T.cutoff_j[q] = Stat::l2_cutoff_j(q, k, P.ANIDiff, P.hgNumerator);
  std::memset(&T.prm, 0, sizeof(T.prm));
  T.prm.kmer_size = k;
  T.prm.kmer_complexity_threshold = P.kmerComplexityThreshold;
  wfm_l1_params_t& l1 = T.prm.l1;
  l1.window_length = (int32_t)w; l1.sketch_size = S; l1.min_hits_cached = cached_minimum_hits_; l1.cached_segment_length = (int32_t)w;
  l1.skip_self = P.skip_self; l1.skip_prefix = P.skip_prefix; l1.lower_triangular = P.lower_triangular;
  l1.stage1_topANI_filter = P.stage1_topANI_filter; l1.stage2_full_scan = P.stage2_full_scan;
  l1.n_seq = (int32_t)T.ref_group.size(); l1.ref_group = T.ref_group.data(); l1.min_hits_by_qsketch = T.min_hits.data();
  l1.sketch_cutoffs = T.cutoffs.data(); l1.n_cutoffs = (int32_t)T.cutoffs.size();
  wfm_l2_params_t& l2 = T.prm.l2;
  l2.window_length = (int32_t)w; l2.sketch_size = S; l2.stage1_topANI_filter = P.stage1_topANI_filter;
  l2.keep_table = T.keep.data(); l2.ident_table = T.ident.data(); l2.cutoff_j = T.cutoff_j.data();

  // createTargetSubsets (computeMap.hpp:295-327)
  std::vector<std::vector<std::string>> subsets;
  {
    int64_t index_by_size = P.index_by_size;
    if (!P.indexFilename.empty() && !P.create_index_only) {
      // "Using batch size N from index file" (computeMap.hpp:349-376)
      int64_t bs = 0;
      try { peek_index_file(P.indexFilename, &bs, nullptr); } catch (const std::exception& e) { wfm_set_error(h_, e.what()); return WFM_E_ARG; }
      if (bs > 0) index_by_size = bs;
    }
    const int64_t batch = index_by_size > 0 ? index_by_size : 5000000;
    std::vector<std::string> cur;
    uint64_t cur_size = 0;
    for (size_t i = 0; i < targetNames.size(); ++i) {
      cur.push_back(targetNames[i]);
      cur_size += ids.getSequenceLength(ids.getSequenceId(targetNames[i]));
      if (cur_size >= (uint64_t)batch || i + 1 == targetNames.size()) { subsets.push_back(cur); cur.clear(); cur_size = 0; }
    }
  }
  sum.targets = targetNames.size();
  sum.queries = queryNames.size();
  sum.subsets = subsets.size();
  for (const auto& n : targetNames) sum.target_bp += ids.getSequenceLength(ids.getSequenceId(n));
  for (const auto& n : queryNames) sum.query_bp += ids.getSequenceLength(ids.getSequenceId(n));

  const bool to_stdout = P.outFileName == "/dev/stdout" || P.outFileName == "-";
  std::ofstream file;
  if (!to_stdout) {
    file.open(P.outFileName);
    if (!file.is_open()) { wfm_set_error(h_, "cannot open output file " + P.outFileName); return WFM_E_ARG; }
  }
  std::ostream& out = to_stdout ? static_cast<std::ostream&>(std::cout) : file;
  std::map<seqno_t, MappingResultsVector_t> combined;  // one-to-one mode: everything is held back

  std::ifstream index_in;   // -I: the sub-indexes are read in file order, one per subset
  if (!P.indexFilename.empty() && !P.create_index_only) {
    index_in.open(P.indexFilename, std::ios::binary);
    if (!index_in) { wfm_set_error(h_, "unable to open index file for reading: " + P.indexFilename); return WFM_E_ARG; }
  }
  size_t subset_idx = 0;
  for (const auto& subset : subsets) {
    // ---- index of this subset (Sketch::build, or Sketch::readIndex with -I)
    double t0 = now_ms();
    wfm_index_t* ix = nullptr;
    const size_t this_subset = subset_idx++;
    if (index_in.is_open()) {
      SubIndex sub;
      try { read_sub_index(index_in, sub, *idManager_); } catch (const std::exception& e) { wfm_set_error(h_, e.what()); return WFM_E_ARG; }
      if (sub.windowLength != w || sub.sketchSize != S || sub.kmerSize != k) {  // readParameters (winSketch.hpp:713-737)
        wfm_set_error(h_, "parameters of the indexed sketch differ from the current ones: index w=" + std::to_string(sub.windowLength) + " s=" +
                              std::to_string(sub.sketchSize) + " k=" + std::to_string(sub.kmerSize));
        return WFM_E_ARG;
      }
      if (sub.names != subset) std::cerr << "[wfmash::mashmap] Warning: the sequences of index subset " << this_subset + 1 << " differ from the expected targets\n";
      if (!sub.minmers.empty()) {
        const int rc = wfm_index_upload(h_, sub.uhash.data(), sub.poff.data(), (int64_t)sub.uhash.size(), sub.points.data(), sub.minmers.data(),
                                        (int64_t)sub.minmers.size(), &ix);
        if (rc != WFM_OK) return rc;
      }
      sum.index_windows += sub.minmers.size();
    } else {
      std::vector<const char*> sp;
      std::vector<int64_t> sl;
      std::vector<int32_t> si;
      for (const auto& name : subset) {
        wfmash_host::SeqView seq;
        if (!src.find(P.refSequences, name, &seq)) { wfm_set_error(h_, "target sequence not found in FASTA: " + name); return WFM_E_ARG; }
        if ((int64_t)seq.size() < w) continue;  // "skipping short sequence" (winSketch.hpp:216-229)
        sp.push_back(seq.data()); sl.push_back((int64_t)seq.size()); si.push_back(ids.getSequenceId(name));
      }
      // minmer intervals (GPU hashing + thinning, host winnowing) and the index stage; the intervals never
      // sit in one host array
      int64_t n_windows = 0;
      const int rc = wfm_index_build_sequences(h_, sp.data(), sl.data(), si.data(), (int64_t)sp.size(), k, (int)w, S, P.threads,
                                               P.max_kmer_freq, &ix, &n_windows);
      if (rc != WFM_OK) return rc;
      sum.index_windows += (uint64_t)n_windows;
    }
    if (P.create_index_only) {
      // -W: write the sub-index, appended after the previous ones, and go on to the next subset (computeMap.hpp:405-415)
      SubIndex sub;
      sub.batch_idx = this_subset; sub.total_batches = subsets.size(); sub.batch_size = P.index_by_size;
      sub.names = subset; sub.windowLength = w; sub.sketchSize = S; sub.kmerSize = k;
      if (ix) {
        wfm_index_info_t inf;
        wfm_index_info(ix, &inf);
        sub.uhash.resize((size_t)inf.n_unique); sub.poff.resize((size_t)inf.n_unique + 1);
        sub.points.resize((size_t)inf.n_points); sub.minmers.resize((size_t)inf.n_kept);
        const int rc = wfm_index_download(h_, ix, sub.uhash.data(), sub.poff.data(), sub.points.data(), sub.minmers.data());
        wfm_index_free(h_, ix);
        if (rc != WFM_OK) return rc;
      } else {
        sub.poff.assign(1, 0);
      }
      std::ofstream index_out(P.indexFilename, this_subset ? std::ios::binary | std::ios::app : std::ios::binary);
      if (!index_out) { wfm_set_error(h_, "unable to open index file for writing: " + P.indexFilename); return WFM_E_ARG; }
      try { write_sub_index(index_out, sub, ids); } catch (const std::exception& e) { wfm_set_error(h_, e.what()); return WFM_E_ARG; }
      sum.ms_index += now_ms() - t0;
      continue;
    }
    sum.ms_index += now_ms() - t0;

    // ---- the other GPUs of the node receive a copy of the finished index (it is read-only from here on,
    // computeMap.hpp:431-484): built once per node, not once per GPU
    std::vector<wfm_index_t*> ixs(hs_.size(), nullptr);
    ixs[0] = ix;
    if (ix) {
      t0 = now_ms();
      // every device pulls its copy at the same time (the source's xGMI links to its peers are separate)
      std::vector<int> rcs(hs_.size(), WFM_OK);
      {
        std::vector<std::thread> pulls;
        for (size_t g = 1; g < hs_.size(); ++g) pulls.emplace_back([&, g] { rcs[g] = wfm_index_replicate(h_, ix, hs_[g], &ixs[g]); });
        for (auto& t : pulls) t.join();
      }
      for (size_t g = 1; g < hs_.size(); ++g) {
        if (rcs[g] != WFM_OK) {
          wfm_set_error(h_, std::string("index replication failed: ") + wfm_last_error(hs_[g]));
          for (size_t f = 0; f < hs_.size(); ++f) wfm_index_free(hs_[f], ixs[f]);
          return rcs[g];
        }
      }
      sum.ms_replicate += now_ms() - t0;
    }
    auto free_indexes = [&] { for (size_t g = 0; g < hs_.size(); ++g) if (ixs[g]) wfm_index_free(hs_[g], ixs[g]); };

    // ---- queries, in batches of whole sequences; a GPU takes the next batch when it is free, finished batches
    // are written in the order they were formed
    struct BatchQuery { std::string name; seqno_t id; offset_t len; int64_t base; int64_t first_frag; int nfrag; };
    // (a batch of one sequence -- a chromosome -- is mapped where the FASTA store holds it; several are laid end to end in `buffer`)
    struct Batch {
      std::vector<BatchQuery> bq;
      std::string buffer;
      const char* bases = nullptr;
      int64_t n_bases = 0;
      std::vector<int64_t> frag_off;
      std::vector<int32_t> frag_seq;
    };
    struct QueryOut { MappingResultsVector_t keep; std::string text; };
    struct BatchOut { std::vector<seqno_t> ids; std::vector<QueryOut> q; };
    int64_t batch_bases = kBatchBases;
    if (hs_.size() > 1) batch_bases = std::max<int64_t>(1, std::min<int64_t>(kBatchBases, (int64_t)(sum.query_bp / (2 * hs_.size()))));
    std::mutex read_mu, write_mu;
    size_t qi = 0;
    uint64_t next_seq = 0, next_write = 0;
    std::map<uint64_t, BatchOut> pending;
    std::atomic<int> error_rc{WFM_OK};
    const int threads_each = std::max(1, P.threads / (int)hs_.size());
    auto read_batch = [&](Batch& b) -> int64_t {
      std::lock_guard<std::mutex> lk(read_mu);
      b = Batch();
      wfmash_host::SeqView only;  // the batch's one sequence so far, not copied yet
      // (a batch of ONE sequence is mapped where the FASTA store holds it; a second one makes the batch a copy of both.  For chromosome-sized
      // queries that copy -- 2 x 249 MB into fresh pages, on the device thread, before every batch of the all-vs-all job -- was 170 ms per batch
      // beside 150 ms of mapping: a sequence that would push the copy past kCopyBases begins a batch of its own)
      constexpr int64_t kCopyBases = 64ll << 20;
      while (qi < queryNames.size() && (b.n_bases < batch_bases || b.bq.empty())) {
        const std::string& name = queryNames[qi];
        wfmash_host::SeqView seq;
        if (!src.find(P.querySequences, name, &seq) || seq.empty()) { ++qi; continue; }  // "not found or empty, skipping" (computeMap.hpp:534-537)
        if (!b.bq.empty() && b.n_bases + (int64_t)seq.size() > kCopyBases) break;
        ++qi;
        BatchQuery q{name, ids.getSequenceId(name), (offset_t)seq.size(), b.n_bases, (int64_t)b.frag_off.size(), 0};
        const int whole = (int)(q.len / w);
        for (int i = 0; i < whole; ++i) b.frag_off.push_back(q.base + (int64_t)i * w);
        q.nfrag = whole;
        if (whole >= 1 && q.len % w != 0) { b.frag_off.push_back(q.base + q.len - w); q.nfrag++; }  // anchored at the end
        b.frag_seq.insert(b.frag_seq.end(), (size_t)q.nfrag, q.id);
        if (b.bq.empty()) {
          only = seq;
        } else {
          if (b.buffer.empty()) b.buffer.assign(only.data(), only.size());
          b.buffer.append(seq.data(), seq.size());
        }
        b.n_bases += (int64_t)seq.size();
        b.bq.push_back(std::move(q));
      }
      if (b.bq.size() == 1) b.bases = only.data(); else b.bases = b.buffer.data();
      return b.bq.empty() ? -1 : (int64_t)next_seq++;
    };
    auto write_batch = [&](uint64_t seq, BatchOut&& bo) {
      std::lock_guard<std::mutex> lk(write_mu);
      pending.emplace(seq, std::move(bo));
      for (auto it = pending.begin(); it != pending.end() && it->first == next_write; it = pending.erase(it), ++next_write) {
        BatchOut& o = it->second;
        for (size_t qn = 0; qn < o.q.size(); ++qn) {
          if (P.filterMode == filter::ONETOONE) {
            auto& dst = combined[o.ids[qn]];
            dst.insert(dst.end(), o.q[qn].keep.begin(), o.q[qn].keep.end());
          } else {
            out << o.q[qn].text;
            sum.written += o.q[qn].keep.size();
          }
        }
      }
      out.flush();
    };
    std::vector<MapSummary> part(hs_.size());
    // an exception on a worker thread (FASTA I/O, bad_alloc, a filter throw) must come back as a WFM_E_* code like
    // everything else: the first message is kept, every thread is joined (Aligner::compute does the same)
    std::mutex err_mu;
    auto fail = [&](int rc, const std::string& what) {
      std::lock_guard<std::mutex> lk(err_mu);
      int expected = WFM_OK;
      if (error_rc.compare_exchange_strong(expected, rc)) wfm_set_error(h_, what);
    };
    // (round 6, f3's other half: SURVEY 8f-3) a batch's post-processing -- boundary check, chaining, sweep, scaffolds, PAF text: host work of
    // 70 ms per chromosome-sized query -- runs on a thread of its own BESIDE the device's mapping of the next batch (the reference runs a
    // query's filters inside that query's task, computeMap.hpp:634-688, while other queries' tasks map): one batch may wait, so memory
    // stays at two batches' mappings per GPU.  WFM_FILTER_OVERLAP=0: one after the other, as before.
    struct Work {
      Batch b;
      RawVec<wfm_mapping_t> maps;
      RawVec<int32_t> mfrag;
      RawVec<uint32_t> perm;  // the batch's mappings in chaining order (wfm_map_fragments_ordered), or perm[0] = ~0u
      int64_t seq = -1;
      MappingResultsVector_t* spare = nullptr;  // the device thread's spare vector (one filter stage runs at a time per device thread)
    };
    static const bool filter_overlap = !(getenv("WFM_FILTER_OVERLAP") && atoi(getenv("WFM_FILTER_OVERLAP")) == 0);
    auto worker_body = [&](size_t g) {
      wfm_handle_t* hg = hs_[g];
      MapSummary& ps = part[g];
      std::function<void(Work&)> filter_stage;  // (defined below: the second half of what used to be one loop body)
      std::mutex qmu;
      std::condition_variable qcv;
      std::deque<std::unique_ptr<Work>> queue;
      bool no_more = false;
      // (two filter threads per device thread since the mapping of a chromosome-sized query became shorter than its post-processing -- 52 against 60 ms:
      // the batches' texts are written in the batches' order whichever thread finishes first; WFM_FILTER_WORKERS=1: one, as before)
      static const size_t n_filt = getenv("WFM_FILTER_WORKERS") ? (size_t)std::max(1, std::min(4, atoi(getenv("WFM_FILTER_WORKERS")))) : 2;
      std::vector<std::thread> filt;
      MappingResultsVector_t spare_results;  // (see filter_stage: a chromosome-sized query's vector serves the next one; the filter threads have one each)
      // (what a chromosome-sized query's vector will about hold, told by the device thread before its mapping call: a filter thread that has no
      // vector from a query before it makes one while the device maps -- resize() writes every element and faults every page in, 9 ms that were
      // the first thing the post-processing did)
      std::atomic<size_t> spare_hint{0};
      auto filter_loop = [&]() {
        MappingResultsVector_t spare_own;
        try {
          for (;;) {
            {
              const size_t hint = spare_hint.load();
              if (hint && spare_own.capacity() < hint) spare_own.resize(hint);
            }
            std::unique_ptr<Work> wk;
            {
              std::unique_lock<std::mutex> lk(qmu);
              qcv.wait(lk, [&] { return !queue.empty() || no_more; });
              if (queue.empty()) return;
              wk = std::move(queue.front());
              queue.pop_front();
            }
            qcv.notify_all();
            wk->spare = &spare_own;
            if (error_rc.load() == WFM_OK) filter_stage(*wk);
          }
        } catch (const std::bad_alloc&) {
          fail(WFM_E_NOMEM, "out of host memory while post-processing mappings");
        } catch (const std::exception& e) {
          fail(WFM_E_ARG, std::string("post-processing failed: ") + e.what());
        }
      };
      struct Joiner {  // the filter threads are joined on every way out of this function
        std::vector<std::thread>& t; std::mutex& mu; std::condition_variable& cv; bool& flag;
        ~Joiner() { { std::lock_guard<std::mutex> lk(mu); flag = true; } cv.notify_all(); for (auto& x : t) if (x.joinable()) x.join(); }
      } joiner{filt, qmu, qcv, no_more};
      for (;;) {
        std::unique_ptr<Work> wkp(new Work());
        Work& W = *wkp;
        Batch& b = W.b;
        if (error_rc.load() != WFM_OK) break;
        const int64_t seq = read_batch(b);
        if (seq < 0) break;
        W.seq = seq;
        W.spare = &spare_results;
        double tb = now_ms();
        auto& maps = W.maps;
        auto& mfrag = W.mfrag;
        auto& perm = W.perm;
        std::vector<int32_t> frag_first;   // per fragment: the first fragment of its query
        static const bool dev_order = !(getenv("WFM_FILTER_DEVICE_ORDER") && atoi(getenv("WFM_FILTER_DEVICE_ORDER")) == 0);
        if (ixs[g] && !b.frag_off.empty()) {
          if (dev_order && P.split) {
            frag_first.resize(b.frag_off.size());
            for (const auto& q : b.bq)
              for (int64_t f = q.first_frag; f < q.first_frag + q.nfrag; ++f) frag_first[(size_t)f] = (int32_t)q.first_frag;
          }
          // a fragment of a pangenome maps about once per target haplotype; a too small buffer costs a
          // second pass over the batch, so be generous
          int64_t cap = (int64_t)b.frag_off.size() * std::min<int64_t>(256, std::max<int64_t>(16, 2 * (int64_t)subset.size())) + (1 << 16);
          if (filter_overlap && b.bq.size() == 1 && b.frag_off.size() >= ((size_t)1 << 16)) {
            // (a pangenome's fragment maps about once per target sequence of another group: one per target sequence is room enough; a vector that
            // turns out too small grows as any vector does)
            spare_hint.store(b.frag_off.size() * (size_t)std::min<int64_t>(16, std::max<int64_t>(1, (int64_t)subset.size())));
            if (filt.empty()) filt.emplace_back(filter_loop);
          }
          for (;;) {
            maps.resize((size_t)cap); mfrag.resize((size_t)cap);
            if (!frag_first.empty()) perm.resize((size_t)cap);
            const int64_t n = frag_first.empty()
                                  ? wfm_map_fragments(hg, ixs[g], b.bases, b.n_bases, b.frag_off.data(), b.frag_seq.data(), (int64_t)b.frag_off.size(), &T.prm,
                                                      maps.data(), mfrag.data(), cap)
                                  : wfm_map_fragments_ordered(hg, ixs[g], b.bases, b.n_bases, b.frag_off.data(), b.frag_seq.data(), (int64_t)b.frag_off.size(),
                                                              &T.prm, maps.data(), mfrag.data(), cap, frag_first.data(), perm.data());
            if (n < 0) {
              fail((int)n, wfm_last_error(hg));
              return;
            }
            if (n <= cap) { maps.resize((size_t)n); mfrag.resize((size_t)n); if (!perm.empty()) perm.resize((size_t)n); break; }  // (shrinking: the content stays)
            cap = n;
          }
        }
        ps.fragments += b.frag_off.size();
        ps.l2_mappings += maps.size();
        ps.ms_map += now_ms() - tb;
        if (!filter_stage) {
          filter_stage = [&, g](Work& FW) {
        Batch& b = FW.b;
        auto& maps = FW.maps;
        auto& mfrag = FW.mfrag;
        auto& perm = FW.perm;
        const int64_t seq = FW.seq;
        (void)g;
        // ---- per query: boundary check, filters, output (processFragment :124-128; query task :634-688)
        double tb = now_ms();
        // queries are independent here (the reference runs one Taskflow task per query); results are
        // written in query order afterwards
        const std::vector<BatchQuery>& bq = b.bq;
        std::vector<size_t> first_map(bq.size() + 1, maps.size());
        {
          size_t m = 0;
          for (size_t qn = 0; qn < bq.size(); ++qn) {
            first_map[qn] = m;
            while (m < maps.size() && mfrag[m] < bq[qn].first_frag + bq[qn].nfrag) ++m;
          }
        }
        const bool have_perm = !perm.empty() && perm.size() == maps.size() && perm[0] != 0xffffffffu;
        BatchOut bo;
        bo.q.resize(bq.size());
        for (const auto& q : bq) bo.ids.push_back(q.id);
        std::vector<QueryOut>& qout = bo.q;
        std::atomic<size_t> next{0};
        const int nt_filter = (int)std::min<size_t>((size_t)threads_each, bq.size());
        auto work = [&]() {
         try {
          set_filter_threads(std::max(1, threads_each / std::max(1, nt_filter)));  // few queries: each may use the idle threads
          for (size_t qn; error_rc.load() == WFM_OK && (qn = next.fetch_add(1)) < bq.size();) {
            const BatchQuery& q = bq[qn];
            static const bool tdbg = getenv("WFM_FILTER_TIMES") != nullptr;
            const double tq0 = now_ms();
            MappingResultsVector_t results;
            const size_t m0 = first_map[qn], nq = first_map[qn + 1] - first_map[qn];
            // (a chromosome-sized query is a batch of its own: its 48 MB vector is the one the query before it left behind -- resize() of a fresh
            // vector writes every element on this thread and faults every page in, 9 of the 13 ms this step took)
            const bool reuse = bq.size() == 1 && nq >= ((size_t)1 << 17);
            if (reuse) results.swap(FW.spare ? *FW.spare : results);
            std::vector<uint32_t> orig;  // (device order) position of every mapping in fragment order, within the query
            if (have_perm && nq >= 2) {
              // the query's mappings in chaining order, straight from the device's permutation (its queries are consecutive there as here)
              results.resize(nq);
              orig.resize(nq);
              std::atomic<bool> inside_a{true};
              auto fill = [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                  const size_t m = perm[m0 + i];
                  if (m < m0 || m - m0 >= nq) { inside_a.store(false); continue; }  // (a permutation that mixes queries would be a bug: sort on the host then)
                  MappingResult r;
                  std::memcpy(&r, &maps[m], sizeof(r));
                  r.queryStartPos += (uint32_t)((mfrag[m] - q.first_frag) * w);
                  results[i] = r;
                  orig[i] = (uint32_t)(m - m0);
                }
              };
              {
                const size_t T = nq >= ((size_t)1 << 17) ? (size_t)std::max(1, std::min(32, threads_each / std::max(1, nt_filter))) : 1;
                wfmash_host::parallel_for(T, (int)T, [&](size_t t) { fill(nq * t / T, nq * (t + 1) / T); });  // (the process's pool: parallel.hpp)
              }
              const bool inside = inside_a.load();
              if (!inside) {
                for (size_t i = 0; i < nq; ++i) {
                  MappingResult r;
                  std::memcpy(&r, &maps[m0 + i], sizeof(r));
                  r.queryStartPos += (uint32_t)((mfrag[m0 + i] - q.first_frag) * w);
                  results[i] = r;
                }
                orig.clear();
              }
            } else {
              results.clear();
              results.reserve(nq);
              for (size_t m = first_map[qn]; m < first_map[qn + 1]; ++m) {
                MappingResult r;
                std::memcpy(&r, &maps[m], sizeof(r));
                r.queryStartPos += (uint32_t)((mfrag[m] - q.first_frag) * w);  // fragmentIndex * windowLength, also for the anchored one
                results.push_back(r);
              }
            }
            const double tq1 = now_ms();
            MappingOutput::mappingBoundarySanityCheck(q.len, results, ids);
            const double tq2 = now_ms();
            if (!orig.empty()) set_presorted_order(orig.data(), orig.size());
            FilteredMappingsResult fr = filterSubsetMappings(results, P, ids, q.len);
            const double tq3 = now_ms();
            const bool merged = P.mergeMappings && P.split;
            MappingResultsVector_t& keep = merged ? fr.mergedMappings : fr.nonMergedMappings;
            const ChainInfoVector_t& chains = merged ? fr.mergedChainInfo : fr.nonMergedChainInfo;
            if (P.filterMode != filter::ONETOONE) {
              std::ostringstream os;
              MappingOutput::reportReadMappings(keep, chains, q.name, os, ids, P, q.len);
              qout[qn].text = os.str();
            }
            qout[qn].keep = std::move(keep);
            if (reuse && merged && FW.spare) FW.spare->swap(fr.nonMergedMappings);  // (the filters' input, handed back: nobody reads it after this)
            if (tdbg && nq >= 100000)
              fprintf(stderr, "[filter] query of %zu mappings: vector %.1f, boundary check %.1f, filterSubsetMappings %.1f, text %.1f ms (stage began %.1f ms before)\n", nq, tq1 - tq0, tq2 - tq1,
                      tq3 - tq2, now_ms() - tq3, tq0 - tb);
          }
         } catch (const std::bad_alloc&) {
          fail(WFM_E_NOMEM, "out of host memory while post-processing mappings");
         } catch (const std::exception& e) {
          fail(WFM_E_ARG, std::string("post-processing failed: ") + e.what());
         }
        };
        {
          const int nt = (int)std::min<size_t>((size_t)threads_each, bq.size());
          wfmash_host::parallel_for((size_t)nt, nt, [&](size_t) { work(); });  // (work() shares the queries out by its own counter and sets its thread's filter threads itself)
        }
        if (error_rc.load() != WFM_OK) return;
        const double tw0 = now_ms();
        write_batch((uint64_t)seq, std::move(bo));
        { std::lock_guard<std::mutex> lk(qmu); ps.ms_filter += now_ms() - tb; }
        if (getenv("WFM_FILTER_TIMES") && maps.size() >= 100000) fprintf(stderr, "[filter] stage of %zu mappings: %.1f ms in all, writing %.1f\n", maps.size(), now_ms() - tb, now_ms() - tw0);
          };
        }
        if (!filter_overlap) { filter_stage(W); continue; }
        if (filt.size() < n_filt) filt.emplace_back(filter_loop);  // (one more per batch until there are n_filt: a call of one batch starts one)
        {
          std::unique_lock<std::mutex> lk(qmu);
          qcv.wait(lk, [&] { return queue.size() < 1 || error_rc.load() != WFM_OK; });
          queue.push_back(std::move(wkp));
        }
        qcv.notify_all();
      }
    };
    auto worker = [&](size_t g) {
      try {
        worker_body(g);
      } catch (const std::bad_alloc&) {
        fail(WFM_E_NOMEM, "out of host memory in the map driver");
      } catch (const std::exception& e) {
        fail(WFM_E_ARG, std::string("map driver: ") + e.what());
      } catch (...) {
        fail(WFM_E_ARG, "map driver: unknown exception");
      }
    };
    {
      std::vector<std::thread> pool;
      try {
        for (size_t g = 1; g < hs_.size(); ++g) pool.emplace_back(worker, g);
      } catch (const std::exception& e) {
        fail(WFM_E_NOMEM, std::string("map driver: could not start a device thread: ") + e.what());
      }
      worker(0);
      for (auto& t : pool) t.join();
    }
    if (error_rc.load() != WFM_OK) { free_indexes(); return error_rc.load(); }
    {
      // the GPUs work side by side: the phase times of a subset are those of its slowest device
      double mm = 0, mf = 0;
      for (const MapSummary& ps : part) {
        sum.fragments += ps.fragments; sum.l2_mappings += ps.l2_mappings;
        mm = std::max(mm, ps.ms_map); mf = std::max(mf, ps.ms_filter);
      }
      sum.ms_map += mm; sum.ms_filter += mf;
    }
    free_indexes();
  }

  if (P.filterMode == filter::ONETOONE) {
    // final reference-axis pass (computeMap.hpp:790-866).  The reference walks unordered maps here;
    // ids ascending is used instead, which fixes the order of the output records.
    const double t0 = now_ms();
    std::map<seqno_t, MappingResultsVector_t> byTarget, final_;
    for (auto& [qid, v] : combined)
      for (auto& r : v) byTarget[(seqno_t)r.refSeqId].push_back(r);
    for (auto& [tid, v] : byTarget) {
      MappingResultsVector_t kept;
      MappingFilterUtils::filterByGroup(v, kept, P.numMappingsForSegment - 1, true, ids, P);
      for (const auto& r : kept)
        for (auto& [qid, orig] : combined)
          for (const auto& o : orig)
            if (o.refSeqId == r.refSeqId && o.refStartPos == r.refStartPos && o.queryStartPos == r.queryStartPos) { final_[qid].push_back(r); break; }
    }
    for (auto& [qid, v] : final_) {
      MappingOutput::reportReadMappings(v, ids.getSequenceName(qid), out, ids, P, ids.getSequenceLength(qid));
      sum.written += v.size();
    }
    out.flush();
    sum.ms_filter += now_ms() - t0;
  }
  sum.ms_total = now_ms() - t_begin;
  if (summary) *summary = sum;
  return WFM_OK;
}

}  // namespace skch
