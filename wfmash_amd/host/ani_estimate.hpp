// ani_estimate.hpp -- automatic identity threshold: skch::Stat::estimate_identity_for_groups
// (src/map/include/map_stats.hpp:325-822), the default `-p ani50-2` path of main.cpp:72-128.
// Every sequence gets a bottom-4096 MinHash at k = 21 on the GPU (wfm_minhash_sketch); sketches are
// pooled per PanSN group, all query-group x target-group pairs of different groups give a Mash ANI,
// and the requested percentile of those, plus the adjustment, becomes the identity threshold.
#pragma once

#include <vector>

#include "../../include/wfmash_hip.h"
#include "map_types.hpp"
#include "sequence_ids.hpp"

namespace skch {
namespace Stat {

// returns the adjusted ANI in [0, 1]; fixed::percentage_identity (0.70) when nothing can be compared
double estimate_identity_for_groups(const Parameters& params, const SequenceIdManager& idManager, wfm_handle_t* h);
// the same with the sequences spread over several GPUs of the node (one sketch per sequence, independent of each other)
double estimate_identity_for_groups(const Parameters& params, const SequenceIdManager& idManager, const std::vector<wfm_handle_t*>& hs);

}  // namespace Stat
}  // namespace skch
