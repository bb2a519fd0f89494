#include "map_stats.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace skch {
namespace Stat {

float j2md(float j, int k) {
  if (j == 0) return 1.0f;
  if (j == 1) return 0.0f;
  const float mash_dist = 1 - std::pow(2 * j / (1 + j), 1.0 / k);  // float ratio, double pow (map_stats.hpp:64)
  return mash_dist;
}

float md2j(float d, int k) {
  const float sim = 1 - d;
  // std::pow(float, int) promotes both to double under g++ (map_stats.hpp:77); spelled out because hipcc adds a float overload
  const double sk = std::pow((double)sim, (double)k);
  const float jaccard = sk / (2 - sk);
  return jaccard;
}

namespace {
inline double log_choose(unsigned n, unsigned k) { return std::lgamma(n + 1.0) - std::lgamma(k + 1.0) - std::lgamma(n - k + 1.0); }
}  // namespace

double binomial_Q(unsigned k, double p, unsigned n) {
  if (k >= n) return 0.0;
  if (p <= 0.0) return 0.0;
  if (p >= 1.0) return 1.0;
  const double lp = std::log(p), lq = std::log1p(-p);
  // sum the shorter tail for accuracy
  if (k + 1 > n / 2) {
    double acc = 0;
    for (unsigned i = k + 1; i <= n; ++i) acc += std::exp(log_choose(n, i) + i * lp + (n - i) * lq);
    return std::min(acc, 1.0);
  }
  double acc = 0;
  for (unsigned i = 0; i <= k; ++i) acc += std::exp(log_choose(n, i) + i * lp + (n - i) * lq);
  return std::max(0.0, 1.0 - acc);
}

// k successes in t draws without replacement from n1 tagged + n2 untagged elements
double hypergeometric_pdf(unsigned k, unsigned n1, unsigned n2, unsigned t) {
  if (t > n1 + n2) t = n1 + n2;
  if (k > n1 || k > t) return 0.0;
  if (t > n2 && k + n2 < t) return 0.0;
  return std::exp(log_choose(n1, k) + log_choose(n2, t - k) - log_choose(n1 + n2, t));
}

double hypergeometric_P(unsigned k, unsigned n1, unsigned n2, unsigned t) {
  double acc = 0;
  const unsigned hi = std::min(k, std::min(n1, t));
  for (unsigned i = 0; i <= hi; ++i) acc += hypergeometric_pdf(i, n1, n2, t);
  return std::min(acc, 1.0);
}

float md_lower_bound(float d, int s, int k, float ci) {
  const float q2 = (1.0 - ci) / 2;
  int x = std::max(int(std::ceil(s * md2j(d, k))), 1);
  while (x <= s) {
    const double cdf_complement = binomial_Q((unsigned)(x - 1), md2j(d, k), (unsigned)s);
    if (cdf_complement < q2) { x--; break; }
    x++;
  }
  const float jaccard = float(x) / s;
  return j2md(jaccard, k);
}

int estimateMinimumHits(int s, int k, float perc_identity) {
  const float mash_dist = 1.0 - perc_identity;
  const float jaccard = md2j(mash_dist, k);
  return (int)std::ceil(1.0 * s * jaccard);
}

int estimateMinimumHitsRelaxed(int s, int k, float perc_identity, float confidence_interval) {
  const int first = estimateMinimumHits(s, k, perc_identity);
  int relaxed = first;
  for (int i = first; i >= 0; i--) {
    const float jaccard = 1.0 * i / s;
    const float d = j2md(jaccard, k);
    const float d_lower = md_lower_bound(d, s, k, confidence_interval);
    const float id_upper = 1.0 - d_lower;
    if (id_upper >= perc_identity) relaxed = i; else break;
  }
  return relaxed;
}

std::vector<int> sketch_cutoffs(int sketchSize, int kmerSize, float ANIDiff, float ANIDiffConf) {
  const float deltaANI = ANIDiff;
  const float min_p = 1 - ANIDiffConf;
  const int ss = (int)std::min<double>(sketchSize, 1000.0);  // fixed::ss_table_max
  std::vector<int> cutoffs((size_t)ss + 1, 1);
  std::vector<std::vector<double>> probs((size_t)ss + 1, std::vector<double>((size_t)ss + 1));
  for (int ci = 0; ci <= ss; ci++)
    for (int y = 0; y <= ci; y++) probs[ci][y] = hypergeometric_pdf((unsigned)y, (unsigned)ss, (unsigned)(ss - ci), (unsigned)ci);
  auto distDiff = [&](int cmax, int ci) {
    double prAbove = 0;
    for (double ymax = 0; ymax <= cmax; ymax++) {
      const double pymax = probs[cmax][(size_t)ymax];
      const double yi_cutoff = deltaANI == 0 ? ymax : std::floor(md2j(j2md(ymax / ss, kmerSize) + deltaANI, kmerSize) * ss);
      double pi_acc = (yi_cutoff - 1) >= 0 ? hypergeometric_P((unsigned)(yi_cutoff - 1), (unsigned)ss, (unsigned)(ss - ci), (unsigned)ci) : 0;
      pi_acc = 1 - pi_acc;
      prAbove += pymax * pi_acc;
      if (prAbove > min_p) return true;
    }
    return prAbove > min_p;
  };
  std::vector<int> range((size_t)ss + 1);
  std::iota(range.begin(), range.end(), 0);
  for (int cmax = 1; cmax <= ss; cmax++) {
    // std::upper_bound(begin, begin + ss, false, [](bool, int ci) { return distDiff(cmax, ci); }):
    // first ci in [0, ss) for which distDiff(cmax, ci) holds, else ss
    const int ci = (int)std::distance(range.begin(), std::upper_bound(range.begin(), range.begin() + ss, false,
                                                                       [&](bool, int c) { return distDiff(cmax, c); }));
    cutoffs[(size_t)cmax] = ci == 0 ? 1 : ci;
  }
  return cutoffs;
}

// Jaccard cutoff of the best-first L2 loop for a query sketch of qs minmers (computeMap.hpp:1001-1004)
double l2_cutoff_j(int qs, int k, float ANIDiff, double hgNumerator) {
  const double jaccardSimilarity = hgNumerator / qs;
  const double mash_dist = j2md(jaccardSimilarity, k);
  const double cutoff_ani = std::max(0.0, (1 - mash_dist) - ANIDiff);
  return md2j(1 - cutoff_ani, k);
}

// identity test and scaled identity of an L2 locus with `shared` of qs sketch elements
// (computeMap.hpp:1018-1036); index qs * (S + 1) + shared
void l2_identity_tables(int S, int k, float percentageIdentity, bool keep_low_pct_id, float ci, std::vector<uint8_t>& keep,
                        std::vector<uint16_t>& ident) {
  keep.assign((size_t)(S + 1) * (S + 1), 0);
  ident.assign((size_t)(S + 1) * (S + 1), 0);
  for (int qs = 1; qs <= S; ++qs) {
    for (int shared = 0; shared <= qs; ++shared) {
      const float mash_dist = j2md(1.0 * shared / qs, k);
      const float nucIdentity = (1 - mash_dist);
      const float nucIdentityUpperBound = 1 - md_lower_bound(mash_dist, qs, k, ci);
      const size_t t = (size_t)qs * (S + 1) + shared;
      keep[t] = ((keep_low_pct_id && nucIdentityUpperBound >= percentageIdentity) || nucIdentity >= percentageIdentity) ? 1 : 0;
      ident[t] = static_cast<uint16_t>(roundf(nucIdentity * 10000.0f));  // MappingResult::setNucIdentity
    }
  }
}

}  // namespace Stat
}  // namespace skch
