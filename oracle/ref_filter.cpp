// oracle/ref_filter.cpp -- TEST INFRASTRUCTURE ONLY.
// Thin C wrapper around the REFERENCE'S OWN post-processing headers, compiled where they lie:
//   g++ -I/root/reference/src -I/root/reference/src/common   (oracle/Makefile, target `ref`)
// sequenceIds.hpp, filter.hpp, mappingFilter.hpp and mappingOutput.hpp compile unmodified and need
// no stand-in header; mappingFilter.hpp uses std::map without including <map>, which is why the
// standard header comes first here.  Nothing of the reference is copied into this repository.
//
// Map::filterSubsetMappings itself lives in computeMap.hpp, which cannot be compiled here
// (htslib, GSL, taskflow); the wrapper calls the same MappingFilterUtils functions in the order of
// computeMap.hpp:1076-1165, so every filter body that runs is the reference's.
#include <fcntl.h>
#include <unistd.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <map>
#include <sstream>
#include <string>

#include "map/include/sequenceIds.hpp"
#include "map/include/mappingFilter.hpp"
#include "map/include/mappingOutput.hpp"

#include "../include/wfmash_host.h"

namespace {

skch::Parameters to_ref(const wfmh_map_params_t& c) {
  skch::Parameters p{};
  p.kmerSize = c.kmer_size;
  p.windowLength = c.window_length;
  p.block_length = c.block_length;
  p.chain_gap = c.chain_gap;
  p.max_mapping_length = c.max_mapping_length;
  p.percentageIdentity = c.percentage_identity;
  p.sketchSize = c.sketch_size;
  p.filterMode = c.filter_mode;
  p.numMappingsForSegment = c.num_mappings_for_segment;
  p.numMappingsForScaffold = c.num_mappings_for_scaffold;
  p.dropRand = c.drop_rand != 0;
  p.split = c.split != 0;
  p.mergeMappings = c.merge_mappings != 0;
  p.skip_self = c.skip_self != 0;
  p.skip_prefix = c.skip_prefix != 0;
  p.lower_triangular = c.lower_triangular != 0;
  p.prefix_delim = c.prefix_delim;
  p.filterLengthMismatches = c.filter_length_mismatches != 0;
  p.sparsity_hash_threshold = c.sparsity_hash_threshold;
  p.overlap_threshold = c.overlap_threshold;
  p.scaffold_overlap_threshold = c.scaffold_overlap_threshold;
  p.scaffold_max_deviation = c.scaffold_max_deviation;
  p.scaffold_gap = c.scaffold_gap;
  p.scaffold_min_length = c.scaffold_min_length;
  p.legacy_output = c.legacy_output != 0;
  p.threads = c.threads > 0 ? c.threads : 1;
  return p;
}

struct QuietStderr {
  int saved, nul;
  QuietStderr() { fflush(stderr); saved = dup(2); nul = open("/dev/null", O_WRONLY); dup2(nul, 2); }
  ~QuietStderr() { fflush(stderr); dup2(saved, 2); close(saved); close(nul); }
};

}  // namespace

extern "C" char* ref_filter(const char* stage, const wfm_mapping_t* maps, int64_t n, const char* fasta, const char* query_name,
                            const wfmh_map_params_t* prm) {
  static_assert(sizeof(skch::MappingResult) == sizeof(wfm_mapping_t), "layout");
  std::string text;
  {
    QuietStderr quiet;  // the progress meter and the id manager talk on stderr
    const skch::Parameters p = to_ref(*prm);
    const std::string delim = p.prefix_delim ? std::string(1, p.prefix_delim) : std::string();
    skch::SequenceIdManager ids({std::string(fasta)}, {std::string(fasta)}, {}, {std::string()}, delim);
    skch::MappingResultsVector_t v((size_t)n);
    if (n) std::memcpy((void*)v.data(), maps, (size_t)n * sizeof(wfm_mapping_t));
    const skch::seqno_t qid = ids.getSequenceId(query_name);
    const skch::offset_t qlen = ids.getSequenceLength(qid);
    progress_meter::ProgressMeter progress((uint64_t)n * 1000 + 1000, "ref", false);
    std::ostringstream os;
    const std::string st(stage);
    typedef skch::MappingFilterUtils F;
    if (st == "subset") {
      std::string seq_dummy;
      {
        // mappingBoundarySanityCheck only reads input->len
        skch::InputSeqProgContainer in(std::string((size_t)qlen, 'A'), query_name, qid, progress);
        skch::MappingOutput::mappingBoundarySanityCheck(&in, v, ids);
      }
      skch::MappingResultsVector_t raw = v;
      auto chained = F::mergeMappingsInRangeWithChains(v, p.chain_gap, p, progress, qid, qlen);
      auto& merged = chained.mappings;
      if (p.mergeMappings && p.split) {
        F::filterWeakMappings(merged, std::floor(p.block_length / p.windowLength), p, ids, qlen);
        if (p.filterMode == skch::filter::MAP || p.filterMode == skch::filter::ONETOONE) {
          skch::MappingResultsVector_t kept;
          F::filterByGroup(merged, kept, p.numMappingsForSegment - 1, false, ids, p, progress);
          merged = std::move(kept);
        }
        if (p.filterLengthMismatches) F::filterFalseHighIdentity(merged, p);
        F::sparsifyMappings(merged, p);
        F::filterByScaffolds(merged, raw, p, ids, progress, qid, qlen);
        skch::MappingOutput::reportReadMappings(merged, chained.chainInfo, query_name, os, ids, p, nullptr, qlen);
      } else {
        if (p.filterMode == skch::filter::MAP || p.filterMode == skch::filter::ONETOONE) {
          skch::MappingResultsVector_t kept;
          F::filterByGroup(v, kept, p.numMappingsForSegment - 1, false, ids, p, progress);
          v = std::move(kept);
        }
        F::filterByScaffolds(v, raw, p, ids, progress, qid, qlen);
        skch::MappingOutput::reportReadMappings(v, query_name, os, ids, p, nullptr, qlen);
      }
    } else if (st == "onetoone") {
      skch::MappingResultsVector_t kept;
      F::filterByGroup(v, kept, p.numMappingsForSegment - 1, true, ids, p, progress);
      skch::MappingOutput::reportReadMappings(kept, query_name, os, ids, p, nullptr, qlen);
    }
    progress.finish();
    text = os.str();
  }
  char* out = (char*)malloc(text.size() + 1);
  std::memcpy(out, text.c_str(), text.size() + 1);
  return out;
}

extern "C" void ref_filter_free(char* p) { free(p); }

// The reference's own id section of an index file (SequenceIdManager::exportIdMapping, sequenceIds.hpp:101-115)
// for the sequences of `fasta`: pins the byte layout and the map iteration order of host/sequence_ids.cpp.
extern "C" int ref_export_ids(const char* fasta, char prefix_delim, const char* out_path) {
  QuietStderr quiet;
  const std::string delim = prefix_delim ? std::string(1, prefix_delim) : std::string();
  skch::SequenceIdManager ids({std::string(fasta)}, {std::string(fasta)}, {}, {std::string()}, delim);
  std::ofstream out(out_path, std::ios::binary);
  if (!out) return -1;
  ids.exportIdMapping(out);
  return out ? 0 : -1;
}

