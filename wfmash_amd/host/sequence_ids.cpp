#include "sequence_ids.hpp"

#include <algorithm>
#include <fstream>
#include <istream>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <unordered_set>

#include "fasta.hpp"

namespace skch {

namespace {
bool starts_with(const std::string& s, const std::string& prefix) { return s.compare(0, prefix.size(), prefix) == 0; }
}  // namespace

SequenceIdManager::SequenceIdManager(const std::vector<std::string>& queryFiles, const std::vector<std::string>& targetFiles,
                                     const std::vector<std::string>& queryPrefixes, const std::vector<std::string>& targetPrefixes,
                                     const std::string& prefixDelim, const std::string& queryList, const std::string& targetList)
    : prefixDelim_(prefixDelim) {
  allPrefixes_ = queryPrefixes;
  allPrefixes_.insert(allPrefixes_.end(), targetPrefixes.begin(), targetPrefixes.end());
  // targets first, so that their ids do not depend on the query set
  for (const auto& f : targetFiles) readIndex(f, targetPrefixes, targetList, false);
  for (const auto& f : queryFiles) readIndex(f, queryPrefixes, queryList, true);
  buildRefGroups();
}

void SequenceIdManager::readIndex(const std::string& fasta, const std::vector<std::string>& prefixes, const std::string& listFile, bool isQuery) {
  std::unordered_set<std::string> allowed;
  if (!listFile.empty()) {
    std::ifstream lf(listFile);
    for (std::string name; std::getline(lf, name);) allowed.insert(name);
  }
  std::vector<std::pair<std::string, offset_t>> entries;
  std::ifstream fai(fasta + ".fai");
  if (fai.is_open()) {
    for (std::string line; std::getline(fai, line);) {
      std::istringstream iss(line);
      std::string name;
      offset_t len = 0;
      iss >> name >> len;
      entries.emplace_back(name, len);
    }
  } else {
    wfmash_host::FastaStore fa(fasta);  // throws when unreadable
    for (int i = 0; i < fa.nseq(); ++i) entries.emplace_back(fa.name(i), (offset_t)fa.length(i));
  }
  for (const auto& e : entries) {
    const bool prefix_ok = prefixes.empty() || std::any_of(prefixes.begin(), prefixes.end(), [&](const std::string& p) { return starts_with(e.first, p); });
    if (!prefix_ok || (!allowed.empty() && !allowed.count(e.first))) continue;
    addSequence(e.first, e.second);
    (isQuery ? queryNames_ : targetNames_).push_back(e.first);
  }
}

seqno_t SequenceIdManager::addSequence(const std::string& name, offset_t length) {
  auto it = idOf_.find(name);
  if (it != idOf_.end()) {
    metadata_[it->second].len = length;
    return it->second;
  }
  const seqno_t id = (seqno_t)metadata_.size();
  idOf_.emplace(name, id);
  metadata_.push_back(ContigInfo{name, length, 0});
  return id;
}

void SequenceIdManager::buildRefGroups() {
  if (metadata_.empty()) throw std::runtime_error("SequenceIdManager: no sequences indexed");
  groupKey_.clear();
  std::vector<std::pair<std::string, size_t>> order;
  order.reserve(metadata_.size());
  for (size_t i = 0; i < metadata_.size(); ++i) order.emplace_back(metadata_[i].name, i);
  std::sort(order.begin(), order.end());
  std::unordered_map<std::string, int> groupOf;
  int nGroups = 0;
  for (const auto& [name, idx] : order) {
    std::string key;
    for (const auto& p : allPrefixes_)
      if (starts_with(name, p)) { key = p; break; }
    if (key.empty() && !prefixDelim_.empty()) {
      const size_t pos = name.rfind(prefixDelim_);
      if (pos != std::string::npos) key = name.substr(0, pos);
    }
    if (key.empty()) key = name;
    auto ins = groupOf.emplace(key, nGroups + 1);
    if (ins.second) { ++nGroups; groupKey_[nGroups] = key; }
    metadata_[idx].groupId = ins.first->second;
  }
}

seqno_t SequenceIdManager::getSequenceId(const std::string& sequenceName) const {
  auto it = idOf_.find(sequenceName);
  if (it != idOf_.end()) return it->second;
  // partial match fallback (sequenceIds.hpp:219-226); lowest id wins here, the reference takes
  // whichever its hash map yields first
  for (size_t i = 0; i < metadata_.size(); ++i)
    if (metadata_[i].name.find(sequenceName) == 0 || sequenceName.find(metadata_[i].name) == 0) return (seqno_t)i;
  throw std::runtime_error("Sequence name not found: '" + sequenceName + "'");
}

const ContigInfo& SequenceIdManager::getContigInfo(seqno_t id) const {
  if (id >= 0 && id < (seqno_t)metadata_.size()) return metadata_[id];
  throw std::runtime_error("Invalid sequence ID: " + std::to_string(id));
}

int SequenceIdManager::getRefGroup(seqno_t seqId) const { return getContigInfo(seqId).groupId; }

std::string SequenceIdManager::getGroupPrefix(int groupId) const {
  auto it = groupKey_.find(groupId);
  return it != groupKey_.end() ? it->second : "group" + std::to_string(groupId);
}

std::vector<int32_t> SequenceIdManager::refGroupTable() const {
  std::vector<int32_t> t(metadata_.size());
  for (size_t i = 0; i < metadata_.size(); ++i) t[i] = metadata_[i].groupId;
  return t;
}

void SequenceIdManager::exportIdMapping(std::ostream& out) const {
  const uint64_t mapSize = idOf_.size();
  out.write(reinterpret_cast<const char*>(&mapSize), sizeof(mapSize));
  for (const auto& [name, id] : idOf_) {
    const uint64_t nameLength = name.size();
    out.write(reinterpret_cast<const char*>(&nameLength), sizeof(nameLength));
    out.write(name.c_str(), (std::streamsize)nameLength);
    out.write(reinterpret_cast<const char*>(&id), sizeof(id));
  }
  const seqno_t nextId = (seqno_t)metadata_.size();
  out.write(reinterpret_cast<const char*>(&nextId), sizeof(nextId));
}

bool SequenceIdManager::importIdMapping(std::istream& in) {
  uint64_t mapSize = 0;
  in.read(reinterpret_cast<char*>(&mapSize), sizeof(mapSize));
  if (!in || mapSize > 1000000) return false;  // the reference's sanity bound
  std::vector<std::pair<std::string, seqno_t>> entries;
  seqno_t maxId = 0;
  for (uint64_t i = 0; i < mapSize; ++i) {
    uint64_t nameLength = 0;
    in.read(reinterpret_cast<char*>(&nameLength), sizeof(nameLength));
    if (!in || nameLength > 10000) return false;
    std::string name((size_t)nameLength, '\0');
    in.read(&name[0], (std::streamsize)nameLength);
    seqno_t id = 0;
    in.read(reinterpret_cast<char*>(&id), sizeof(id));
    if (!in || id < 0) return false;
    maxId = std::max(maxId, id);
    entries.emplace_back(std::move(name), id);
  }
  seqno_t indexNextId = 0;
  in.read(reinterpret_cast<char*>(&indexNextId), sizeof(indexNextId));
  if (!in) return false;
  // the usual case: the run reads the files the index was made from -- nothing moves
  bool same = entries.size() == idOf_.size();
  for (const auto& e : entries) {
    auto it = idOf_.find(e.first);
    same = same && it != idOf_.end() && it->second == e.second;
  }
  if (same) return true;
  const auto oldIds = idOf_;
  const auto oldMeta = metadata_;
  idOf_.clear();
  targetNames_.clear();
  metadata_.assign((size_t)std::max<seqno_t>(indexNextId, maxId + 1), ContigInfo{std::string(), 0, 0});
  for (const auto& e : entries) {
    idOf_[e.first] = e.second;
    targetNames_.push_back(e.first);
    metadata_[(size_t)e.second].name = e.first;
    auto it = oldIds.find(e.first);
    if (it != oldIds.end()) metadata_[(size_t)e.second].len = oldMeta[(size_t)it->second].len;
  }
  for (const auto& q : queryNames_) {
    if (idOf_.count(q)) continue;
    auto it = oldIds.find(q);
    addSequence(q, it != oldIds.end() ? oldMeta[(size_t)it->second].len : 0);
  }
  buildRefGroups();
  return true;
}

}  // namespace skch
