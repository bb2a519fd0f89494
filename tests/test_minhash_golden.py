"""m10's sketch pinned on the reference's own code: tests/golden/minhash_golden.json.gz was written by the reference's
StreamingMinHash / GroupedStreamingMinHash::processSequence (streamingMinHash.hpp; tests/golden/make_minhash_golden.py).
CPU: the restatement oracle/map_ani.py against it (and against the live reference build when present).
GPU: wfm_minhash_sketch and the host-side pooling against it."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import map_ani as ANI
from oracle import pymap

G = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "minhash_golden.json.gz"), "rt"))
SEQS = {k: v.encode() for k, v in G["seqs"].items()}


def _u64(hexes):
    return np.array([int(x, 16) for x in hexes], dtype=np.uint64)


def test_restatement_matches_reference_sketches():
    assert len(G["sketches"]) >= 30
    for c in G["sketches"]:
        got = ANI.minhash_sketch(SEQS[c["seq"]], c["k"], c["sketch_size"])
        want = _u64(c["hashes"])
        assert len(got) == len(want) and (got == want).all(), (c["seq"], c["k"], c["sketch_size"])
    # the cases do what they are there for: duplicates are kept, an ambiguous base among the first k blanks k-mers 0..k-1
    rep = next(c for c in G["sketches"] if c["seq"] == "repeat" and c["sketch_size"] == 4096)
    assert len(set(rep["hashes"])) < len(rep["hashes"]) / 4
    head = next(c for c in G["sketches"] if c["seq"] == "ambiguous_head" and c["k"] == 21 and c["sketch_size"] == 4096)
    h, st = pymap.hash_kmers(SEQS["ambiguous_head"], 21)
    assert (np.sort(h[21:][st[21:] != 0])[:4096] == _u64(head["hashes"])).all()


def test_restatement_matches_reference_pools_and_heap():
    for c in G["pools"]:
        sk = np.zeros(0, dtype=np.uint64)
        for m in c["members"]:
            sk = ANI.pool(sk, ANI.minhash_sketch(SEQS[m], c["k"], c["sketch_size"]), c["sketch_size"])
        want = _u64(c["hashes"])
        assert len(sk) == len(want) and (sk == want).all(), c["members"]
    for c in G["streams"]:
        want = c["sketch"]
        got = sorted(c["values"])[:c["sketch_size"]]  # bottom-k with multiplicity
        assert got == want, c


@pytest.mark.skipif(not pymap.have_ref(), reason="oracle/_ref is not built (needs /root/reference)")
def test_golden_is_what_the_reference_build_gives_now():
    for c in G["sketches"][::5]:
        v = pymap.ref_group_minhash([SEQS[c["seq"]]], [0], c["k"], c["sketch_size"], 0)
        assert (v == _u64(c["hashes"])).all()


@pytest.mark.gpu
def test_gpu_minhash_sketch_matches_reference_sketches(gpu):
    for c in G["sketches"]:
        got = gpu.minhash_sketch(SEQS[c["seq"]], k=c["k"], sketch_size=c["sketch_size"])
        want = _u64(c["hashes"])
        assert len(got) == len(want) and (got == want).all(), (c["seq"], c["k"], c["sketch_size"])
