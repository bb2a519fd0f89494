#include "index_file.hpp"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <istream>
#include <numeric>
#include <ostream>
#include <stdexcept>

namespace skch {

namespace {

template <typename T> void put(std::ostream& out, const T& v) { out.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <typename T> void get(std::istream& in, T& v) {
  in.read(reinterpret_cast<char*>(&v), sizeof(T));
  if (!in) throw std::runtime_error("index file: unexpected end of file");
}

static_assert(sizeof(wfm_minmer_t) == 32 && sizeof(wfm_interval_point_t) == 24, "record layouts of the reference (base_types.hpp:28-59)");

}  // namespace

void write_sub_index(std::ostream& out, const SubIndex& ix, const SequenceIdManager& ids) {
  // writeSubIndexHeader (winSketch.hpp:640-660)
  put(out, kIndexMagic);
  put(out, (uint64_t)ix.batch_idx);
  put(out, (uint64_t)ix.total_batches);
  put(out, (int64_t)ix.batch_size);
  put(out, (uint64_t)ix.names.size());
  for (const auto& n : ix.names) {
    put(out, (uint64_t)n.size());
    out.write(n.data(), (std::streamsize)n.size());
  }
  ids.exportIdMapping(out);
  // writeParameters (:606-612)
  put(out, (offset_t)ix.windowLength);
  put(out, (int)ix.sketchSize);
  put(out, (int)ix.kmerSize);
  // writeSketchBinary (:569-574)
  put(out, (uint64_t)ix.minmers.size());
  out.write(reinterpret_cast<const char*>(ix.minmers.data()), (std::streamsize)(ix.minmers.size() * sizeof(wfm_minmer_t)));
  // writePosListBinary (:579-593): keys in the order of their first interval in minmerIndex
  const size_t nk = ix.uhash.size();
  std::vector<int64_t> first((size_t)nk, (int64_t)ix.minmers.size());
  for (size_t i = ix.minmers.size(); i-- > 0;) {
    const size_t u = (size_t)(std::lower_bound(ix.uhash.begin(), ix.uhash.end(), ix.minmers[i].hash) - ix.uhash.begin());
    if (u < nk && ix.uhash[u] == ix.minmers[i].hash) first[u] = (int64_t)i;
  }
  std::vector<uint32_t> order(nk);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return first[a] < first[b]; });
  put(out, (uint64_t)nk);
  for (uint32_t u : order) {
    put(out, (uint64_t)ix.uhash[u]);
    const uint64_t np = (uint64_t)(ix.poff[u + 1] - ix.poff[u]);
    put(out, np);
    out.write(reinterpret_cast<const char*>(ix.points.data() + ix.poff[u]), (std::streamsize)(np * sizeof(wfm_interval_point_t)));
  }
  if (!out) throw std::runtime_error("index file: write failed");
}

void read_sub_index(std::istream& in, SubIndex& ix, SequenceIdManager& ids) {
  // readSubIndexHeader (winSketch.hpp:869-935) with its sanity bounds
  uint64_t magic = 0;
  get(in, magic);
  if (magic != kIndexMagic) throw std::runtime_error("index file: wrong magic number");
  get(in, ix.batch_idx);
  get(in, ix.total_batches);
  if (ix.total_batches < 1 || ix.total_batches > 1000 || ix.batch_idx >= ix.total_batches) throw std::runtime_error("index file: invalid batch information");
  get(in, ix.batch_size);
  uint64_t n_names = 0;
  get(in, n_names);
  if (n_names > 1000000) throw std::runtime_error("index file: invalid number of sequences");
  ix.names.clear();
  for (uint64_t i = 0; i < n_names; ++i) {
    uint64_t len = 0;
    get(in, len);
    if (len > 10000) throw std::runtime_error("index file: invalid sequence name length");
    std::string name((size_t)len, '\0');
    in.read(&name[0], (std::streamsize)len);
    ix.names.push_back(std::move(name));
  }
  if (!ids.importIdMapping(in)) throw std::runtime_error("index file: malformed id section");
  // readParameters (:713-737)
  get(in, ix.windowLength);
  get(in, ix.sketchSize);
  get(in, ix.kmerSize);
  // readSketchBinary (:677-683)
  uint64_t n = 0;
  get(in, n);
  ix.minmers.resize((size_t)n);
  in.read(reinterpret_cast<char*>(ix.minmers.data()), (std::streamsize)(n * sizeof(wfm_minmer_t)));
  if (!in) throw std::runtime_error("index file: unexpected end of file");
  for (auto& m : ix.minmers) m.pad_ = 0;
  // readPosListBinary (:688-709), then into the sorted form the device index uses
  uint64_t nk = 0;
  get(in, nk);
  std::vector<std::pair<uint64_t, std::vector<wfm_interval_point_t>>> lists((size_t)nk);
  uint64_t total = 0;
  for (uint64_t i = 0; i < nk; ++i) {
    uint64_t np = 0;
    get(in, lists[(size_t)i].first);
    get(in, np);
    lists[(size_t)i].second.resize((size_t)np);
    in.read(reinterpret_cast<char*>(lists[(size_t)i].second.data()), (std::streamsize)(np * sizeof(wfm_interval_point_t)));
    if (!in) throw std::runtime_error("index file: unexpected end of file");
    total += np;
  }
  std::sort(lists.begin(), lists.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  ix.uhash.clear(); ix.poff.assign(1, 0); ix.points.clear();
  ix.points.reserve((size_t)total);
  for (auto& l : lists) {
    ix.uhash.push_back(l.first);
    for (auto p : l.second) {
      wfm_interval_point_t q;
      std::memset(&q, 0, sizeof(q));  // padding bytes
      q.pos = p.pos; q.hash = p.hash; q.seqId = p.seqId; q.side = p.side;
      ix.points.push_back(q);
    }
    ix.poff.push_back((int64_t)ix.points.size());
  }
}

void peek_index_file(const std::string& path, int64_t* batch_size, uint64_t* total_batches) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("unable to open index file for reading: " + path);
  uint64_t magic = 0, idx = 0, total = 0;
  int64_t bs = 0;
  get(in, magic);
  if (magic != kIndexMagic) throw std::runtime_error("invalid index file format (wrong magic number): " + path);
  get(in, idx); get(in, total); get(in, bs);
  if (batch_size) *batch_size = bs;
  if (total_batches) *total_batches = total;
}

}  // namespace skch
