// map_filter.hpp -- post-processing of a query's L2 mappings, host side.
//
// Restates, with the reference's integer/float behaviour and libstdc++ container semantics:
//   skch::Filter::query::filterMappings / ref::filterMappings   src/map/include/filter.hpp:172-235, :471-535
//   MappingFilterUtils::{filterWeakMappings :154, filterFalseHighIdentity :184, sparsifyMappings :203,
//     filterByGroup :220, mergeMappingsInRangeWithChains :382, mergeMappingsInRange :580,
//     filterByScaffolds :831}                                     src/map/include/mappingFilter.hpp
//   Map::filterSubsetMappings                                    src/map/include/computeMap.hpp:1076-1165
//   MappingOutput::{mappingBoundarySanityCheck :31, reportReadMappings :74}   src/map/include/mappingOutput.hpp
// This stage is serial per query in the reference and stays on the host; the GPU produces its input.
#pragma once

#include <ostream>
#include <string>

#include "map_types.hpp"
#include "sequence_ids.hpp"

namespace skch {

namespace Filter {
namespace query {
// plane sweep over the query axis: keeps, at every query position, the best-scoring mappings
void filterMappings(MappingResultsVector_t& readMappings, int secondaryToKeep, bool dropRand, double overlapThreshold);
}  // namespace query
namespace ref {
// plane sweep over the concatenated reference axis (one-to-one mode)
void filterMappings(MappingResultsVector_t& readMappings, const SequenceIdManager& idManager, uint16_t secondaryToKeep, bool dropRand,
                    double overlapThreshold);
}  // namespace ref
}  // namespace Filter

class MappingFilterUtils {
 public:
  static void filterWeakMappings(MappingResultsVector_t& readMappings, int64_t min_count, const Parameters& param,
                                 const SequenceIdManager& idManager, offset_t queryLen);
  static void filterFalseHighIdentity(MappingResultsVector_t& readMappings, const Parameters& param);
  static void sparsifyMappings(MappingResultsVector_t& readMappings, const Parameters& param);
  static void filterByGroup(MappingResultsVector_t& unfilteredMappings, MappingResultsVector_t& filteredMappings, int n_mappings,
                            bool filter_ref, const SequenceIdManager& idManager, const Parameters& param);
  static MappingsWithChains mergeMappingsInRangeWithChains(MappingResultsVector_t& readMappings, int max_dist, const Parameters& param);
  static MappingResultsVector_t mergeMappingsInRange(MappingResultsVector_t& readMappings, int max_dist, const Parameters& param);
  static void filterByScaffolds(MappingResultsVector_t& readMappings, const Parameters& param, const SequenceIdManager& idManager);
};

struct FilteredMappingsResult {
  MappingResultsVector_t nonMergedMappings, mergedMappings;
  ChainInfoVector_t nonMergedChainInfo, mergedChainInfo;
};
// host threads the CALLING thread may use inside filterSubsetMappings (default 1): a batch with a single long query
void set_filter_threads(int threads);
// f3, first step: the caller built the next query's mappings in chaining order from the device's permutation
// (wfm_map_fragments_ordered); orig_index[i] = mapping i's position in the reference's input order (fragment order).  chain_mappings checks the
// order and skips its own sort; the pointer must stay valid until the next filterSubsetMappings / mergeMappingsInRange of this thread returns.
void set_presorted_order(const uint32_t* orig_index, size_t n);

// Map::filterSubsetMappings: everything between a query's raw L2 mappings and what is printed
FilteredMappingsResult filterSubsetMappings(MappingResultsVector_t& mappings, const Parameters& param, const SequenceIdManager& idManager,
                                            offset_t queryLen);

class MappingOutput {
 public:
  static void mappingBoundarySanityCheck(offset_t queryLen, MappingResultsVector_t& readMappings, const SequenceIdManager& idManager);
  static void reportReadMappings(MappingResultsVector_t& readMappings, const ChainInfoVector_t& chainInfo, const std::string& queryName,
                                 std::ostream& outstrm, const SequenceIdManager& idManager, const Parameters& param, offset_t queryLen);
  static void reportReadMappings(MappingResultsVector_t& readMappings, const std::string& queryName, std::ostream& outstrm,
                                 const SequenceIdManager& idManager, const Parameters& param, offset_t queryLen);
};

}  // namespace skch
