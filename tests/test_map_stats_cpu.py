"""CPU tests of the map statistics: host C++ (wfmash_amd/host/map_stats.cpp)
against an independent numpy/scipy restatement.  The outputs that matter are the integer
thresholds (minimum hits, L1 sketch cutoffs), cross-checked over the parameter ranges the
configs use (k 15/21, s 39/78 ... , -p 70..99)."""
import pytest

from wfmash_amd import capi
from oracle import map_stats as S


def test_minimum_hits_tables():
    for k in (15, 19, 21):
        for s in (5, 39, 78, 156, 400):
            for ident in (0.70, 0.80, 0.85, 0.90, 0.95, 0.98, 0.995):
                got = capi.host_cigar_fn("min_hits", f"{s},{k},{ident},0.95")
                exp = f"{S.estimate_minimum_hits(s, k, ident)},{S.estimate_minimum_hits_relaxed(s, k, ident, 0.95)}"
                assert got == exp, (s, k, ident)


def test_known_values():
    # Jaccard <-> mash distance are inverse maps (map_stats.hpp:56-79)
    assert abs(float(S.j2md(S.md2j(0.1, 15), 15)) - 0.1) < 1e-5
    # -p 90 defaults: k=15, s=39
    assert capi.host_cigar_fn("min_hits", "39,15,0.9,0.95") == f"{S.estimate_minimum_hits(39, 15, 0.9)},{S.estimate_minimum_hits_relaxed(39, 15, 0.9, 0.95)}"


@pytest.mark.parametrize("s,k,ad,ac", [(39, 15, 0.0, 0.999), (78, 15, 0.0, 0.999), (25, 21, 0.0, 0.999), (39, 15, 0.02, 0.999), (60, 15, 0.05, 0.99)])
def test_sketch_cutoffs_match_scipy(s, k, ad, ac):
    got = [int(x) for x in capi.host_cigar_fn("sketch_cutoffs", f"{s},{k},{ad},{ac}").split(",")]
    assert got == S.sketch_cutoffs(s, k, ad, ac)


def test_threshold_tables_match_the_committed_fixture():
    """tests/golden/stats_golden.json.gz: the integer tables an independent scipy implementation gives
    (tests/golden/make_stats_golden.py); the host C++ (own log-gamma distribution functions) must give the same."""
    import gzip
    import json
    import os
    g = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stats_golden.json.gz"), "rt"))
    assert len(g["min_hits"]) > 300
    for s, k, ident, mh, mhr in g["min_hits"]:
        assert capi.host_cigar_fn("min_hits", f"{s},{k},{ident},0.95") == f"{mh},{mhr}", (s, k, ident)
    for c in g["sketch_cutoffs"]:
        got = [int(x) for x in capi.host_cigar_fn("sketch_cutoffs", f"{c['s']},{c['k']},{c['ani_diff']},{c['ani_diff_conf']}").split(",")]
        assert got == c["cutoffs"], (c["s"], c["k"])
    for c in g["l2_tables"]:
        got = capi.host_cigar_fn("l2_tables", f"{c['s']},{c['k']},{c['identity']},{c['ci']}").split(",")
        assert [int(x.split(":")[0]) for x in got] == c["keep"], (c["s"], c["identity"])
        assert [int(x.split(":")[1]) for x in got] == c["ident"], (c["s"], c["identity"])
