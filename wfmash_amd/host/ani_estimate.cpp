#include "ani_estimate.hpp"

#include <algorithm>
#include <atomic>
#include <thread>
#include <map>
#include <memory>
#include <string>
#include <stdexcept>
#include <unordered_map>
#include <utility>
#include <vector>

#include "fasta.hpp"
#include "map_stats.hpp"

namespace skch {
namespace Stat {

namespace {
constexpr int kEstimationK = 21;             // map_stats.hpp:328
constexpr int kEstimationSketchSize = 4096;  // map_stats.hpp:330
constexpr double kFallbackIdentity = 0.70;   // fixed::percentage_identity

// StreamingMinHash keeps the smallest values with their multiplicity, so pooling is a multiset
// union cut back to the sketch size.
void pool(std::vector<hash_t>& group, const std::vector<hash_t>& add) {
  group.insert(group.end(), add.begin(), add.end());
  std::sort(group.begin(), group.end());
  if (group.size() > (size_t)kEstimationSketchSize) group.resize(kEstimationSketchSize);
}
}  // namespace

double estimate_identity_for_groups(const Parameters& params, const SequenceIdManager& idManager, wfm_handle_t* h) {
  return estimate_identity_for_groups(params, idManager, std::vector<wfm_handle_t*>{h});
}

double estimate_identity_for_groups(const Parameters& params, const SequenceIdManager& idManager, const std::vector<wfm_handle_t*>& hs) {
  if (hs.empty()) throw std::runtime_error("no GPU handle");
  struct Role { std::string file; bool is_query = false, is_target = false; };
  std::map<std::string, Role> roles;  // the reference walks a hash map here; the result does not depend on the order
  std::unordered_map<std::string, std::shared_ptr<wfmash_host::FastaStore>> stores;
  auto open = [&](const std::string& file) -> wfmash_host::FastaStore& {
    auto it = stores.find(file);
    if (it == stores.end()) it = stores.emplace(file, wfmash_host::open_shared(file)).first;
    return *it->second;
  };
  auto known = [&](const std::string& name) {
    try { return idManager.getSequenceName(idManager.getSequenceId(name)) == name; } catch (const std::exception&) { return false; }
  };
  for (const auto& file : params.querySequences) {
    auto& fa = open(file);
    for (int i = 0; i < fa.nseq(); ++i)
      if (known(fa.name(i))) { Role& r = roles[fa.name(i)]; r.file = file; r.is_query = true; }
  }
  for (const auto& file : params.refSequences) {
    auto& fa = open(file);
    for (int i = 0; i < fa.nseq(); ++i)
      if (known(fa.name(i))) { Role& r = roles[fa.name(i)]; if (r.file.empty()) r.file = file; r.is_target = true; }
  }
  std::map<int, std::vector<hash_t>> query_groups, target_groups;
  for (const auto& [name, role] : roles) {
    const int g = idManager.getRefGroup(idManager.getSequenceId(name));
    if (role.is_query) query_groups[g];
    if (role.is_target) target_groups[g];
  }
  int query_seq_count = 0, target_seq_count = 0;
  {  // indexed files: read the sequences side by side instead of one after the other
    std::unordered_map<std::string, std::vector<int>> want;
    for (const auto& [name, role] : roles) {
      const int i = open(role.file).find(name);
      if (i >= 0) want[role.file].push_back(i);
    }
    for (const auto& [file, which] : want) open(file).preload(which, params.threads);
  }
  // one sketch per sequence: the sequences are spread over the GPUs at hand (a queue, one host thread per device), the
  // sketches are pooled per group afterwards in the order of the names (the pool of a group does not depend on the order)
  struct Item { const std::string* name; const Role* role; wfmash_host::SeqView seq; std::vector<hash_t> sketch; };
  std::vector<Item> items;
  for (const auto& [name, role] : roles) {
    const wfmash_host::FastaStore& fa = open(role.file);
    const int64_t len = fa.seq_len(name);
    if (len <= 0) continue;  // "not found or empty, skipping"
    items.push_back(Item{&name, &role, fa.sequence(fa.find(name)), {}});
  }
  {
    std::atomic<size_t> next{0};
    std::vector<std::string> errors(hs.size());
    auto work = [&](size_t g) {
      std::vector<hash_t> sketch((size_t)kEstimationSketchSize);
      for (size_t i; (i = next.fetch_add(1)) < items.size();) {
        const wfmash_host::SeqView seq = items[i].seq;
        const int64_t n = wfm_minhash_sketch(hs[g], seq.data(), (int64_t)seq.size(), kEstimationK, kEstimationSketchSize, sketch.data());
        if (n < 0) { errors[g] = std::string("wfm_minhash_sketch failed: ") + wfm_last_error(hs[g]); return; }
        items[i].sketch.assign(sketch.begin(), sketch.begin() + n);
      }
    };
    std::vector<std::thread> pool_threads;
    for (size_t g = 1; g < hs.size(); ++g) pool_threads.emplace_back(work, g);
    work(0);
    for (auto& t : pool_threads) t.join();
    for (const auto& e : errors) if (!e.empty()) throw std::runtime_error(e);
  }
  for (const Item& it : items) {
    const int g = idManager.getRefGroup(idManager.getSequenceId(*it.name));
    if (it.role->is_query) { pool(query_groups[g], it.sketch); ++query_seq_count; }
    if (it.role->is_target) { pool(target_groups[g], it.sketch); ++target_seq_count; }
  }
  if (query_seq_count == 0 || target_seq_count == 0) return kFallbackIdentity;

  std::vector<double> anis;
  for (const auto& [qg, qs] : query_groups) {
    for (const auto& [tg, ts] : target_groups) {
      // same-group pairs are skipped; (A,B) and (B,A) are both taken: the reference's self-mode test
      // compares the addresses of two different members and never holds (map_stats.hpp:712-715)
      if (qg == tg || qs.empty() || ts.empty()) continue;
      size_t shared = 0, i = 0, j = 0;
      while (i < qs.size() && j < ts.size()) {
        if (qs[i] == ts[j]) { ++shared; ++i; ++j; }
        else if (qs[i] < ts[j]) ++i;
        else ++j;
      }
      if (shared == 0) continue;
      const double jaccard = static_cast<double>(shared) / std::min(qs.size(), ts.size());
      const double mash_dist = j2md(jaccard, kEstimationK);
      anis.push_back(1.0 - mash_dist);
    }
  }
  if (anis.empty()) return kFallbackIdentity;
  std::sort(anis.begin(), anis.end());
  size_t idx = (params.ani_percentile * anis.size()) / 100;
  if (idx >= anis.size()) idx = anis.size() - 1;
  double adjusted = anis[idx] + (params.ani_adjustment / 100.0);
  if (adjusted < 0.0) adjusted = 0.0;
  if (adjusted > 1.0) adjusted = 1.0;
  return adjusted;
}

}  // namespace Stat
}  // namespace skch
