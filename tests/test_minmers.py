"""addMinmers: the host winnowing stage against golden vectors generated from the
reference's own commonFunc.hpp (tests/golden/make_map_golden.py) -- CPU -- and the full
wfm_add_minmers (GPU hashing + host winnowing) on the GPU box.  Bit-exact, order included."""
import gzip
import json
import os
import random

import numpy as np
import pytest

from oracle import pymap
from wfmash_amd import capi, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "map_golden.json.gz")


def _as_list(m):
    return [[str(int(x["hash"])), int(x["wpos"]), int(x["wpos_end"]), int(x["strand"])] for x in m]


def _cpu_winnow(seq, k, w, s, seq_id):
    h, st = pymap.hash_kmers(seq, k)
    return capi.host_winnow(seq, k, w, s, seq_id, h, st)


def test_winnow_matches_reference_golden():
    gold = json.load(gzip.open(GOLD, "rt"))
    assert len(gold["minmers"]) >= 15
    for e in gold["minmers"]:
        seq = gold["seqs"][e["seq"]].encode()
        got = _cpu_winnow(seq, e["k"], e["w"], e["s"], 3)
        assert _as_list(got) == e["minmers"], (e["seq"], e["k"], e["w"], e["s"])
        assert all(int(x["seqId"]) == 3 for x in got)


@pytest.mark.skipif(not pymap.have_ref(), reason="reference build (oracle/_ref) only exists in the authoring container")
def test_winnow_matches_reference_live():
    rng = random.Random(2)
    for i in range(40):
        n = rng.choice([300, 1200, 4000, 9000])
        s = bytearray(synth.random_dna(500 + i, n))
        for _ in range(rng.randrange(0, 4)):
            p = rng.randrange(0, n)
            L = rng.randrange(1, 60)
            s[p:p + L] = (b"N" * L)[:max(0, min(L, n - p))]
        if rng.random() < 0.3:  # repeats: the lazy-heap quirks only show up on repetitive sequence
            unit = bytes(s[:rng.choice([7, 31, 150])])
            s = bytearray((unit * (n // len(unit) + 1))[:n])
            for _ in range(n // 50):
                s[rng.randrange(0, n)] = rng.choice(b"ACGT")
        if rng.random() < 0.2:
            s[:5] = b"NNACN"  # N inside the first k-1 bases (no initial scan in addMinmers)
        s = bytes(s[:n])
        k = rng.choice([15, 19])
        w = rng.choice([64, 256, 1000])
        sk = rng.choice([3, 12, 39])
        if n < w:
            continue
        ref = pymap.ref_add_minmers(s, k, w, sk, 5)
        got = _cpu_winnow(s, k, w, sk, 5)
        assert _as_list(got) == _as_list(ref), (i, n, k, w, sk)


@pytest.mark.gpu
def test_add_minmers_gpu_matches_golden(gpu):
    gold = json.load(gzip.open(GOLD, "rt"))
    for e in gold["minmers"]:
        seq = gold["seqs"][e["seq"]].encode()
        got = gpu.add_minmers(seq, e["k"], e["w"], e["s"], 3)
        assert _as_list(got) == e["minmers"], (e["seq"], e["k"], e["w"], e["s"])


@pytest.mark.gpu
def test_add_minmers_gpu_equals_cpu_stage_on_long_sequence(gpu):
    seq = synth.random_dna(31, 300000)
    got = gpu.add_minmers(seq, 15, 1000, 39, 0)
    exp = _cpu_winnow(seq, 15, 1000, 39, 0)
    assert len(got) == len(exp) > 5000
    assert (got == exp).all()
