"""CPU tests of the host-side map post-processing: chaining/merging, weak
mapping filter, plane sweeps, scaffold filter and the mapping-PAF writer of wfmash_amd/host against
(a) golden output of the reference's own code (tests/golden/filter_golden.json.gz, made by
tests/golden/make_filter_golden.py from oracle/_ref/libref_filter.so) and (b) that library live
when it is present."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import pyfilter
from tests import filter_cases as FC
from wfmash_amd import capi

GOLDEN = json.loads(gzip.open(os.path.join(os.path.dirname(__file__), "golden", "filter_golden.json.gz")).read())
# sha256 of the reference's output (oracle/_ref/libref_filter.so) for the input test_large_query_passes_split_over_threads builds
LARGE_QUERY_SHA = "3a5eafce1f19cea2bb326cd397e8a21c82290f8f13d15b322e36b55f70426a4c"


@pytest.fixture(scope="module")
def fai(tmp_path_factory):
    return FC.write_fai(str(tmp_path_factory.mktemp("fai")))


def test_default_params_match_parse_args():
    p = capi.map_default_params()
    assert (p.kmer_size, p.window_length, p.chain_gap, p.max_mapping_length) == (15, 1000, 2000, 50000)
    assert p.num_mappings_for_segment == 0xFFFFFFFF and p.num_mappings_for_scaffold == 1 and p.filter_mode == 1
    assert (p.scaffold_gap, p.scaffold_max_deviation, p.scaffold_min_length) == (100000, 100000, 10000)
    assert p.prefix_delim == b"#" and p.skip_prefix == 1 and p.skip_self == 1 and p.minimum_hits == 3
    assert abs(p.overlap_threshold - 0.95) < 1e-12 and abs(p.max_kmer_freq - 0.0002) < 1e-12


@pytest.mark.parametrize("case", FC.CASES, ids=[c[0] for c in FC.CASES])
def test_filter_subset_matches_reference_golden(fai, case):
    name, query, seed, over = case
    m = FC.make_mappings(name, query, seed, over)
    P = capi.map_default_params(**over)
    got = capi.host_filter("subset", m, fai, query, P)
    assert got == GOLDEN[name]["subset"]
    if "onetoone" in GOLDEN[name]:
        assert capi.host_filter("onetoone", m, fai, query, P) == GOLDEN[name]["onetoone"]


def test_golden_is_not_trivial():
    n_lines = {k: len(v["subset"].splitlines()) for k, v in GOLDEN.items()}
    assert sum(1 for v in n_lines.values() if v > 10) >= 18
    assert n_lines["defaults"] > n_lines["n1"] > 0 and n_lines["no_merge"] > n_lines["defaults"]
    first = GOLDEN["defaults"]["subset"].splitlines()[0].split("\t")
    assert len(first) == 15 and first[12].startswith("id:f:") and first[14].startswith("ch:Z:")


@pytest.mark.skipif(not pyfilter.have_ref(), reason="oracle/_ref/libref_filter.so not built (needs /root/reference)")
def test_filter_subset_matches_reference_live(fai):
    """fresh seeds (not in the golden file) through the reference's own code, side by side"""
    total = 0
    for seed in range(100, 112):
        for over in ({}, {"num_mappings_for_segment": 2, "overlap_threshold": 0.7}, {"chain_gap": 8000, "block_length": 3000}):
            m = FC.make_mappings("live", "A#2#c1", seed, over)
            P = capi.map_default_params(**over)
            exp = pyfilter.ref_filter("subset", m, fai, "A#2#c1", P)
            assert capi.host_filter("subset", m, fai, "A#2#c1", P) == exp
            assert capi.host_filter("onetoone", m, fai, "A#2#c1", P) == pyfilter.ref_filter("onetoone", m, fai, "A#2#c1", P)
            total += len(exp.splitlines())
    assert total > 1000


def test_filter_input_order_fast_and_general_paths(fai):
    """The chaining sorts take a shortcut when the mappings arrive the way the GPU stages deliver them (fragment
    order, per fragment by target position) and fall back to std::sort otherwise: both orders of the same mappings,
    with the scaffold filter on, against the reference's own code."""
    if not pyfilter.have_ref():
        pytest.skip("oracle/_ref/libref_filter.so is built from /root/reference")
    for seed in (201, 202, 203):
        m = FC.make_mappings("live", "A#1#c1", seed, {})
        ordered = np.sort(m, order=["queryStartPos", "refSeqId", "refStartPos"])
        shuffled = ordered.copy()
        np.random.default_rng(seed).shuffle(shuffled)
        P = capi.map_default_params()
        for arr in (ordered, ordered[::-1].copy(), shuffled):
            assert capi.host_filter("subset", arr, fai, "A#1#c1", P) == pyfilter.ref_filter("subset", arr, fai, "A#1#c1", P)


def test_chain_representatives_in_closed_form_equal_the_disjoint_sets(fai, monkeypatch):
    """chain_mappings names a chain by min(id of its first member, id of its second) and orders chains by a prefix sum where the first
    order had no ties; WFM_FILTER_CLOSED_FORM=0 keeps the reference's disjoint sets and the second sort: same text on every case,
    chain gaps from one window to whole sequences (many singletons ... few long chains)"""
    for seed in range(300, 312):
        for over in ({}, {"chain_gap": 500}, {"chain_gap": 200000, "block_length": 3000}, {"num_mappings_for_segment": 3}):
            m = FC.make_mappings("live", "A#2#c1", seed, over)
            m = np.sort(m, order=["queryStartPos", "refSeqId", "refStartPos"])  # the order the GPU stages deliver
            P = capi.map_default_params(**over)
            monkeypatch.setenv("WFM_FILTER_CLOSED_FORM", "1")
            a = capi.host_filter("subset", m, fai, "A#2#c1", P)
            monkeypatch.setenv("WFM_FILTER_CLOSED_FORM", "0")
            assert capi.host_filter("subset", m, fai, "A#2#c1", P) == a


def test_sequence_id_manager_groups(fai, tmp_path):
    """ids follow the .fai order, groups the sorted names up to the LAST delimiter (sequenceIds.hpp:286-338)."""
    # a mapping onto every target prints its name and length through the id manager
    m = np.array([(t, 10, 0, 1000, 1, 10, 9000, 0, 90) for t in range(len(FC.NAMES))], dtype=FC.MAPPING_DTYPE)
    P = capi.map_default_params(scaffold_gap=0, filter_mode=3, merge_mappings=0)
    out = capi.host_filter("subset", m, fai, "A#1#c1", P).splitlines()
    assert [(l.split("\t")[5], int(l.split("\t")[6])) for l in out] == FC.NAMES
    with pytest.raises(capi.WfmError):
        capi.host_filter("subset", m, fai, "absent_sequence_name", P)


def test_large_query_passes_split_over_threads(tmp_path, monkeypatch):
    """a chromosome-sized query: the chaining passes are split over host threads above 128 k mappings (the batch has no other
    query to give the cores to); whatever the split, the output is the single-threaded one -- and the reference's, when its
    code is here"""
    import hashlib
    fa = str(tmp_path / "pan.fa")
    open(fa, "w").close()
    names = [f"hap{i}#1#chr1" for i in range(1, 9)]
    with open(fa + ".fai", "w") as f:
        for n in names:
            f.write(f"{n}\t70000000\t0\t60\t61\n")
    rng = np.random.default_rng(1)
    nf = 60000
    recs = []
    for t in range(1, 8):
        keep = rng.random(nf) < 0.995
        q = np.arange(nf, dtype=np.int64)[keep] * 1000
        m = np.zeros(int(keep.sum()), dtype=FC.MAPPING_DTYPE)
        m["refSeqId"] = t
        m["refStartPos"] = np.maximum(q + rng.integers(-40, 41, len(q)) + t * 137, 0)
        m["queryStartPos"] = q
        m["blockLength"] = 1000
        m["n_merged"] = 1
        m["conservedSketches"] = rng.integers(15, 24, len(q))
        m["nucIdentity"] = rng.integers(9970, 10000, len(q))
        m["kmerComplexity"] = 100
        if t == 3:  # an inverted stretch: both strands of one target
            m["flags"][20000:26000] = 1
        recs.append(m)
    allm = np.concatenate(recs)
    allm = allm[np.lexsort((allm["refSeqId"], allm["queryStartPos"]))]  # the order the GPU stages deliver: per fragment, by target
    assert len(allm) > (1 << 17)
    P = capi.map_default_params()
    out = {}
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("WFM_FILTER_THREADS", threads)
        out[threads] = capi.host_filter("subset", allm, fa, names[0], P)
    assert out["1"] == out["3"] == out["8"] and out["1"].count("\n") > 5000
    # the chains' representatives and their order in closed form (paths: min of the first two ids) against the disjoint sets and the sort
    monkeypatch.setenv("WFM_FILTER_CLOSED_FORM", "0")
    assert capi.host_filter("subset", allm, fa, names[0], P) == out["8"]
    monkeypatch.delenv("WFM_FILTER_CLOSED_FORM")
    if pyfilter.have_ref():
        assert out["8"] == pyfilter.ref_filter("subset", allm, fa, names[0], P)
    else:
        assert hashlib.sha256(out["8"].encode()).hexdigest() == LARGE_QUERY_SHA
