// capi_map.hpp -- conversion between the C parameter block and skch::Parameters
#pragma once
#include "../../include/wfmash_host.h"
#include "map_types.hpp"

namespace wfmash_host {
skch::Parameters to_parameters(const wfmh_map_params_t& c);
}
