// map_types.hpp -- call-surface types of the map path, host side.
// Mirrors src/map/include/base_types.hpp (MappingResult :154-253, ChainInfo :261-265, ContigInfo
// :93-98, enums :100-135) and the fields of skch::Parameters (map_parameters.hpp:32-118) that the
// mapping, filtering and output stages read.  MappingResult has the byte layout of
// wfm_mapping_t so device output is used in place.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <limits>
#include <string>
#include <vector>

#include "../../include/wfmash_hip.h"

namespace skch {

typedef uint64_t hash_t;
typedef int64_t offset_t;
typedef int32_t seqno_t;
typedef int16_t strand_t;

namespace strnd { enum : strand_t { FWD = 1, REV = -1 }; }
namespace event { enum : int { BEGIN = 1, END = 2 }; }
namespace filter { enum : int { MAP = 1, ONETOONE = 2, NONE = 3 }; }  // base_types.hpp:128-135

struct ContigInfo {
  std::string name;
  offset_t len = 0;
  int groupId = 0;
};

struct MappingResult {
  uint32_t refSeqId = 0;
  uint32_t refStartPos = 0;
  uint32_t queryStartPos = 0;
  uint32_t blockLength = 0;
  uint32_t n_merged = 1;
  uint32_t conservedSketches = 0;
  uint16_t nucIdentity = 0;   // x 1e4
  uint8_t flags = 0;          // bit 0 reverse strand, bit 1 discard, bit 2 overlapped
  uint8_t kmerComplexity = 0; // x 100

  strand_t strand() const { return (flags & 1) ? strnd::REV : strnd::FWD; }
  bool discard() const { return (flags & 2) != 0; }
  bool overlapped() const { return (flags & 4) != 0; }
  void setStrand(strand_t s) { if (s == strnd::REV) flags |= 1; else flags &= ~1; }
  void setDiscard(bool d) { if (d) flags |= 2; else flags &= ~2; }
  void setOverlapped(bool o) { if (o) flags |= 4; else flags &= ~4; }
  float getNucIdentity() const { return nucIdentity / 10000.0f; }
  float getKmerComplexity() const { return kmerComplexity / 100.0f; }
  void setNucIdentity(float identity) { nucIdentity = static_cast<uint16_t>(roundf(identity * 10000.0f)); }
  void setKmerComplexity(float complexity) { kmerComplexity = static_cast<uint8_t>(roundf(complexity * 100.0f)); }
  offset_t refEndPos() const { return (uint32_t)(refStartPos + blockLength); }      // uint32 sum, as the reference (wraps at 2^32)
  offset_t queryEndPos() const { return (uint32_t)(queryStartPos + blockLength); }
  float blockNucIdentity() const { return getNucIdentity(); }
  // boost-style hash_combine over seven fields (base_types.hpp:143-148, :237-247); std::hash of an
  // integer is the identity in libstdc++
  size_t hash() const {
    size_t s = 0;
    auto mix = [&s](size_t v) { s ^= v + 0x9e3779b9 + (s << 6) + (s >> 2); };
    mix(refSeqId); mix(refStartPos); mix(queryStartPos); mix(blockLength); mix(nucIdentity); mix(conservedSketches); mix(flags);
    return s;
  }
};
static_assert(sizeof(MappingResult) == 28 && sizeof(wfm_mapping_t) == 28, "MappingResult must match wfm_mapping_t");

typedef std::vector<MappingResult> MappingResultsVector_t;

struct ChainInfo {
  uint32_t chainId;
  uint16_t chainPos;  // 1-based
  uint16_t chainLen;
};
typedef std::vector<ChainInfo> ChainInfoVector_t;

struct MappingsWithChains {
  MappingResultsVector_t mappings;
  ChainInfoVector_t chainInfo;
};

// skch::Parameters: the fields read by the stages built here, defaults of parse_args.hpp
struct Parameters {
  int kmerSize = 15;                       // parse_args.hpp:501-510
  offset_t windowLength = 1000;            // :337-339
  offset_t block_length = 0;               // :412-414
  offset_t chain_gap = 2000;               // :423-426
  uint64_t max_mapping_length = 50000;     // :480-482
  float percentageIdentity = 0.70f;        // map_parameters.hpp:126
  bool stage2_full_scan = true;            // parse_args.hpp (always on)
  bool stage1_topANI_filter = true;
  float ANIDiff = 0.0f;                    // map_parameters.hpp:127-128
  float ANIDiffConf = 0.999f;
  int filterMode = filter::MAP;            // parse_args.hpp:228-233
  uint32_t numMappingsForSegment = std::numeric_limits<uint32_t>::max();   // -n inf (:837-856)
  uint32_t numMappingsForScaffold = 1;     // :882
  bool dropRand = false;                   // :312
  int threads = 1;
  bool split = true;                       // :311
  bool lower_triangular = false;
  bool skip_self = true;                   // :171
  bool skip_prefix = true;                 // :184-188
  char prefix_delim = '#';
  bool mergeMappings = true;               // :315
  bool keep_low_pct_id = true;             // :173
  bool filterLengthMismatches = true;      // :698
  float kmerComplexityThreshold = 0;       // :656
  int sketchSize = 0;                      // :639-644
  double hgNumerator = 1.0;
  uint64_t sparsity_hash_threshold = std::numeric_limits<uint64_t>::max();  // :240
  double overlap_threshold = 0.95;         // :495-497
  double scaffold_overlap_threshold = 0.5; // :46
  int64_t scaffold_max_deviation = 100000; // :443-449
  int64_t scaffold_gap = 100000;           // :432-438
  int64_t scaffold_min_length = 10000;     // :454-461
  bool legacy_output = false;
  int64_t index_by_size = std::numeric_limits<int64_t>::max();  // :766-768
  std::string indexFilename;               // -W / -I (:745-758)
  bool create_index_only = false;          // -W: write the index and stop
  int minimum_hits = 3;                    // :729-731
  double max_kmer_freq = 0.0002;           // :735-737
  bool auto_pct_identity = true;           // -p ani50-2 (:41-43, :392-395); an explicit -p switches it off
  int ani_percentile = 50;
  float ani_adjustment = -2.0f;
  std::vector<std::string> refSequences, querySequences;
  std::string target_list, target_prefix, query_list;
  std::vector<std::string> query_prefix;
  std::string outFileName = "/dev/stdout";
};

}  // namespace skch
