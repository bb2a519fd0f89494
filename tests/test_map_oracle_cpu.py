"""CPU tests: the map-path oracle (oracle/map_oracle.cpp) against golden vectors generated
from the reference's own commonFunc.hpp (tests/golden/make_map_golden.py) and, where the
reference build exists (authoring container), against it live on random inputs."""
import gzip
import json
import os
import random

import numpy as np
import pytest

from oracle import pymap
from wfmash_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "map_golden.json.gz")


@pytest.fixture(scope="module")
def gold():
    return json.load(gzip.open(GOLD, "rt"))


def test_kmer_hash_known_answers(gold):
    for e in gold["kmer_hashes"]:
        assert pymap.get_hash(e["kmer"].encode()) == int(e["hash"])


def test_sketch_sequence_matches_golden(gold):
    assert len(gold["sketches"]) >= 20
    for e in gold["sketches"]:
        seq = gold["seqs"][e["seq"]].encode()
        m = pymap.sketch_sequence(seq, e["k"], e["s"], 7)
        got = [[str(int(x["hash"])), int(x["wpos"]), int(x["wpos_end"]), int(x["strand"])] for x in m]
        assert got == e["minmers"], (e["seq"], e["k"], e["s"])
        assert all(int(x["seqId"]) == 7 for x in m)


def test_hash_kmers_contract():
    s = bytearray(synth.random_dna(3, 400))
    s[50] = ord("N")
    s[200:204] = b"acgt"
    h, st = pymap.hash_kmers(bytes(s), 15)
    assert len(h) == 386
    assert (st[36:51] == 0).all() and (h[36:51] == np.uint64(2**64 - 1)).all()  # every k-mer covering the N
    up = bytes(s).upper().replace(b"N", b"A")
    # lower case is upper-cased before hashing (makeUpperCaseAndValidDNA)
    h2, _ = pymap.hash_kmers(bytes(s).upper(), 15)
    assert (h == h2).all()
    # palindromic k-mers (hashFwd == hashBwd) are skipped: even k only
    hp, sp = pymap.hash_kmers(b"ACGTACGTACGTACGT", 16)
    assert sp[0] == 0


@pytest.mark.skipif(not pymap.have_ref(), reason="reference build (oracle/_ref) only exists in the authoring container")
def test_oracle_matches_reference_live():
    rng = random.Random(1)
    for i in range(60):
        n = rng.choice([15, 16, 40, 300, 1000, 2500])
        s = bytearray(synth.random_dna(100 + i, n))
        for _ in range(rng.randrange(0, 4)):
            p = rng.randrange(0, n)
            s[p:p + rng.randrange(1, 30)] = b"N" * min(rng.randrange(1, 30), n - p)
        if rng.random() < 0.3:
            s = bytearray(bytes(s).lower())
        s = bytes(s[:n])
        k = rng.choice([11, 15, 16, 21, 31])
        if n < k:
            continue
        sk = rng.choice([1, 5, 39, 78, 200])
        a = pymap.sketch_sequence(s, k, sk, 2, "ref")
        b = pymap.sketch_sequence(s, k, sk, 2, "oracle")
        assert len(a) == len(b) and (a == b).all(), (n, k, sk)
        kmer = s[:k].upper()
        assert pymap.get_hash(kmer, "ref") == pymap.get_hash(kmer)
