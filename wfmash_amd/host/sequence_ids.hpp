// sequence_ids.hpp -- skch::SequenceIdManager (src/map/include/sequenceIds.hpp:16-443), SURVEY 8a m12.
//
// Sequence ids are handed out targets-first in .fai order, then queries (sequenceIds.hpp:358-373);
// a name seen twice keeps its first id.  Groups ("PanSN prefixes") are numbered from 1 in the order
// of the lexicographically sorted sequence names (sequenceIds.hpp:286-338); the group key of a name
// is the first matching user prefix, else the part before the LAST delimiter, else the whole name.
// The .fai next to each FASTA is read when it exists; otherwise the FASTA itself is scanned
// (the reference lets htslib create the .fai, faigz.h FAI_CREATE).  exportIdMapping / importIdMapping
// (sequenceIds.hpp:101-212) are the id section of the on-disk index (host/index_file.cpp).
#pragma once

#include <iosfwd>
#include <string>
#include <unordered_map>
#include <vector>

#include "map_types.hpp"

namespace skch {

class SequenceIdManager {
 public:
  SequenceIdManager(const std::vector<std::string>& queryFiles, const std::vector<std::string>& targetFiles,
                    const std::vector<std::string>& queryPrefixes, const std::vector<std::string>& targetPrefixes,
                    const std::string& prefixDelim, const std::string& queryList = "", const std::string& targetList = "");

  seqno_t getSequenceId(const std::string& sequenceName) const;  // throws std::runtime_error when unknown
  const ContigInfo& getContigInfo(seqno_t id) const;
  const std::string& getSequenceName(seqno_t id) const { return getContigInfo(id).name; }
  const offset_t& getSequenceLength(seqno_t id) const { return getContigInfo(id).len; }
  size_t size() const { return metadata_.size(); }
  const std::vector<ContigInfo>& getMetadata() const { return metadata_; }
  const std::vector<std::string>& getQuerySequenceNames() const { return queryNames_; }
  const std::vector<std::string>& getTargetSequenceNames() const { return targetNames_; }
  int getRefGroup(seqno_t seqId) const;
  std::string getGroupPrefix(int groupId) const;
  // groupId of every sequence id, the table the L1 kernels take
  std::vector<int32_t> refGroupTable() const;
  // the id section of an index file: (name length, name, id) per sequence in the map's own iteration order, then
  // the next free id (sequenceIds.hpp:101-115)
  void exportIdMapping(std::ostream& out) const;
  // takes names and ids from an index file; lengths of known names are kept, query sequences the file does
  // not know get fresh ids (the reference reloads them, sequenceIds.hpp:58-99,117-212); false = malformed
  bool importIdMapping(std::istream& in);

 private:
  seqno_t addSequence(const std::string& name, offset_t length);
  void readIndex(const std::string& fasta, const std::vector<std::string>& prefixes, const std::string& listFile, bool isQuery);
  void buildRefGroups();

  std::unordered_map<std::string, seqno_t> idOf_;
  std::vector<ContigInfo> metadata_;
  std::vector<std::string> queryNames_, targetNames_;
  std::vector<std::string> allPrefixes_;
  std::string prefixDelim_;
  std::unordered_map<int, std::string> groupKey_;
};

}  // namespace skch
