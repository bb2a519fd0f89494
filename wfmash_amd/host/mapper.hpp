// mapper.hpp -- the map phase driver: counterpart of skch::Map (src/map/include/computeMap.hpp:61-230,
// mapQuery :329-872) on top of the C ABI of include/wfmash_hip.h.
//
//   targets  -> subsets of <= index_by_size bases (createTargetSubsets :295-327)
//   subset   -> wfm_add_minmers per sequence (>= windowLength long, winSketch.hpp:216) -> wfm_index_build
//   queries  -> windowLength fragments + one anchored at the end (:560-631), batched over whole
//               query sequences -> wfm_map_fragments (sketch, L1, L2 on the GPU)
//   per query and subset: mappingBoundarySanityCheck, filterSubsetMappings, reportReadMappings
//               (host, map_filter.hpp); one-to-one mode adds the reference-axis pass at the end (:790-866)
// The reference's Taskflow/thread-pool plumbing is replaced by batches; fragment results are taken
// in fragment order where the reference takes them in task-completion order.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/wfmash_hip.h"
#include "map_types.hpp"
#include "sequence_ids.hpp"

namespace skch {

struct MapSummary {
  uint64_t targets = 0, queries = 0, subsets = 0;
  uint64_t target_bp = 0, query_bp = 0;
  uint64_t index_windows = 0;       // minmer intervals over all subsets
  uint64_t fragments = 0;           // query fragments mapped (summed over subsets)
  uint64_t l2_mappings = 0;         // MappingResults leaving the GPU
  uint64_t written = 0;             // mapping PAF lines written
  double ms_index = 0, ms_replicate = 0, ms_map = 0, ms_filter = 0, ms_total = 0;
};

class Map {
 public:
  // p.sketchSize == 0 derives it from the identity (parse_args.hpp:642-644)
  Map(const Parameters& p, wfm_handle_t* h);
  // one handle per GPU of the node: the index is built on the first and copied to the others (it is read-only while
  // mapping, computeMap.hpp:431-484), batches of query sequences go to whichever device is free
  Map(const Parameters& p, const std::vector<wfm_handle_t*>& hs);
  // maps every query against every target subset and writes param.outFileName; returns 0 or WFM_E_*
  int mapQuery(MapSummary* summary = nullptr);
  const SequenceIdManager& ids() const { return *idManager_; }
  const Parameters& parameters() const { return param_; }

 private:
  Parameters param_;
  wfm_handle_t* h_;                // hs_[0]: builds the index, carries the error message
  std::vector<wfm_handle_t*> hs_;
  std::unique_ptr<SequenceIdManager> idManager_;
  int cached_minimum_hits_ = 0;
};

}  // namespace skch
