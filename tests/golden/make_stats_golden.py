#!/usr/bin/env python
"""Generates tests/golden/stats_golden.json.gz: the INTEGER threshold tables of the map statistics as an
independent implementation gives them -- oracle/map_stats.py + oracle/map_l2.py on scipy.stats (binom.sf, hypergeom.pmf,
hypergeom.cdf), where the reference calls GSL (gsl_cdf_binomial_Q map_stats.hpp:109, gsl_ran_hypergeometric_pdf
computeMap.hpp:248, gsl_cdf_hypergeometric_P computeMap.hpp:260) and the product sums log-gammas
(wfmash_amd/host/map_stats.cpp).  GSL is not in the image, so no vector can come from the reference itself; what the
two implementations must agree on are the thresholds, not the doubles behind them.
    python tests/golden/make_stats_golden.py"""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import map_l2 as L2  # noqa: E402
from oracle import map_stats as S  # noqa: E402

out = {"min_hits": [], "sketch_cutoffs": [], "l2_tables": []}
for k in (15, 19, 21):
    for s in (5, 19, 39, 49, 78, 156, 400):
        for ident in (0.70, 0.75, 0.80, 0.85, 0.90, 0.95, 0.98, 0.995):
            out["min_hits"].append([s, k, ident, S.estimate_minimum_hits(s, k, ident), S.estimate_minimum_hits_relaxed(s, k, ident, 0.95)])
# per-query-sketch-size thresholds as Map::mapQuery builds them (every q in 1..S), for the configs' sketch sizes
for s, ident in ((39, 0.90), (78, 0.70), (25, 0.98)):
    out["min_hits"] += [[q, 15, ident, S.estimate_minimum_hits(q, 15, ident), S.estimate_minimum_hits_relaxed(q, 15, ident, 0.95)]
                        for q in range(1, s + 1)]
for s, k, ad, ac in ((39, 15, 0.0, 0.999), (78, 15, 0.0, 0.999), (49, 15, 0.0, 0.999), (25, 21, 0.0, 0.999), (39, 15, 0.02, 0.999),
                     (60, 15, 0.05, 0.99), (130, 15, 0.0, 0.999)):
    out["sketch_cutoffs"].append({"s": s, "k": k, "ani_diff": ad, "ani_diff_conf": ac, "cutoffs": S.sketch_cutoffs(s, k, ad, ac)})
for s, k, ident in ((39, 15, 0.90), (78, 15, 0.70), (49, 15, 0.85), (25, 21, 0.98)):
    keep, idt = L2.identity_tables(s, k, ident)
    out["l2_tables"].append({"s": s, "k": k, "identity": ident, "ci": 0.95, "keep": keep.flatten().tolist(), "ident": idt.flatten().tolist()})
with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stats_golden.json.gz"), "wt") as f:
    json.dump(out, f)
print({k: len(v) for k, v in out.items()})
