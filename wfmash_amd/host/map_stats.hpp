// map_stats.hpp -- mashmap statistics of the map path, host side.
// Restates skch::Stat::{j2md, md2j, md_lower_bound, estimateMinimumHits,
// estimateMinimumHitsRelaxed} (src/map/include/map_stats.hpp:56-189) and Map::setProbs
// (src/map/include/computeMap.hpp:234-293) with the reference's float/double mix.  The
// reference calls GSL for three distribution functions (gsl_cdf_binomial_Q map_stats.hpp:109,
// gsl_ran_hypergeometric_pdf computeMap.hpp:248, gsl_cdf_hypergeometric_P computeMap.hpp:260);
// GSL is not available, so they are computed here from log-gamma sums.  Their results only
// feed integer thresholds.
#pragma once

#include <cstdint>
#include <vector>

namespace skch {
namespace Stat {

float j2md(float j, int k);
float md2j(float d, int k);
double binomial_Q(unsigned k, double p, unsigned n);                                   // P(X > k), X ~ Bin(n, p)
double hypergeometric_pdf(unsigned k, unsigned n1, unsigned n2, unsigned t);          // GSL argument order
double hypergeometric_P(unsigned k, unsigned n1, unsigned n2, unsigned t);            // P(X <= k)
float md_lower_bound(float d, int s, int k, float ci);
int estimateMinimumHits(int s, int k, float perc_identity);
int estimateMinimumHitsRelaxed(int s, int k, float perc_identity, float confidence_interval);
// Map::setProbs: sketchCutoffs[cmax], cmax = 0 .. min(s, 1000)
std::vector<int> sketch_cutoffs(int sketchSize, int kmerSize, float ANIDiff, float ANIDiffConf);
// per-(Q.sketchSize, shared) tables the L2 kernels take (computeMap.hpp:1001-1004, :1018-1036)
double l2_cutoff_j(int qs, int k, float ANIDiff, double hgNumerator);
void l2_identity_tables(int S, int k, float percentageIdentity, bool keep_low_pct_id, float ci, std::vector<uint8_t>& keep,
                        std::vector<uint16_t>& ident);

}  // namespace Stat
}  // namespace skch
