"""GPU parity tests of the map-path kernels (wfm_hash_kmers, wfm_sketch_fragments) through the
C ABI against the CPU oracle and the golden vectors generated from the reference's own code.
Bit-exact (u64 hashes, positions, strands)."""
import gzip
import json
import os
import random

import numpy as np
import pytest

from oracle import pymap
from wfmash_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "map_golden.json.gz")


def _noisy(seed, n):
    rng = random.Random(seed)
    s = bytearray(synth.random_dna(seed, n))
    for _ in range(rng.randrange(0, 5)):
        p = rng.randrange(0, n)
        L = rng.randrange(1, 40)
        s[p:p + L] = (b"N" * L)[:max(0, min(L, n - p))]
    if rng.random() < 0.4:
        p = rng.randrange(0, max(1, n - 50))
        s[p:p + 50] = bytes(s[p:p + 50]).lower()
    return bytes(s[:n])


def test_hash_kmers_matches_oracle(gpu):
    for seed, n, k in [(1, 15, 15), (2, 16, 15), (3, 1000, 15), (4, 5000, 21), (5, 70000, 15), (6, 333, 11), (7, 2000, 31), (8, 900, 16), (9, 100, 32),
                       (10, 3000, 9), (11, 3000, 10), (12, 3000, 12), (13, 3000, 13), (14, 3000, 14), (15, 3000, 8), (16, 40000, 16),
                       (17, 3000, 17), (18, 3000, 19), (19, 30000, 21), (20, 3000, 23), (21, 3000, 24), (22, 3000, 25)]:  # 9..24: the word-wise kernel
        s = _noisy(seed, n)
        h, st = gpu.hash_kmers(s, k)
        eh, est = pymap.hash_kmers(s, k)
        assert (st == est).all(), (n, k)
        assert (h == eh).all(), (n, k)


def test_hash_kmers_edge_cases(gpu):
    h, st = gpu.hash_kmers(b"ACGT", 15)  # shorter than k: nothing
    assert len(h) == 0
    h, st = gpu.hash_kmers(b"N" * 100, 15)
    assert (st == 0).all()
    # the reference's own known answers
    gold = json.load(gzip.open(GOLD, "rt"))
    for e in gold["kmer_hashes"]:
        km = e["kmer"].encode()
        if b"N" in km:
            continue
        h, st = gpu.hash_kmers(km, len(km))
        eh, est = pymap.hash_kmers(km, len(km))
        assert h[0] == eh[0] and st[0] == est[0]


def test_sketch_fragments_match_golden(gpu):
    gold = json.load(gzip.open(GOLD, "rt"))
    for e in gold["sketches"]:
        seq = gold["seqs"][e["seq"]].encode()
        if len(seq) - e["k"] + 1 > 8192:
            continue
        got = gpu.sketch_fragments(seq, [0], [len(seq)], e["k"], e["s"], 7)[0]
        g = [[str(int(x["hash"])), int(x["wpos"]), int(x["wpos_end"]), int(x["strand"])] for x in got]
        assert g == e["minmers"], (e["seq"], e["k"], e["s"])
        assert all(int(x["seqId"]) == 7 for x in got)


def test_sketch_fragments_query_fragmentation(gpu):
    """1 kb query fragments as mapQuery cuts them (computeMap.hpp:560-631): floor(len/w) pieces
    plus one piece covering the last w bases when len % w != 0."""
    w, k, s = 1000, 15, 39
    seq = _noisy(42, 12345)
    offs = [i * w for i in range(len(seq) // w)]
    lens = [w] * len(offs)
    if len(seq) % w:
        offs.append(len(seq) - w)
        lens.append(w)
    got = gpu.sketch_fragments(seq, offs, lens, k, s, 3)
    assert len(got) == len(offs)
    for o, L, g in zip(offs, lens, got):
        e = pymap.sketch_sequence(seq[o:o + L], k, s, 3)
        assert len(g) == len(e)
        for name in ("hash", "wpos", "wpos_end", "seqId", "strand"):
            assert (g[name] == e[name]).all(), (o, name)


def test_sketch_fragments_ragged_and_degenerate(gpu):
    seq = b"A" * 500 + synth.random_dna(5, 700) + b"N" * 300 + (b"ACGTTGCA" * 100)
    offs = [0, 400, 1100, 1500, 0, 1490]
    lens = [500, 900, 400, 800, 14, 15]
    for k, s in [(15, 39), (21, 5), (15, 300)]:
        got = gpu.sketch_fragments(seq, offs, lens, k, s, 0)
        for o, L, g in zip(offs, lens, got):
            e = pymap.sketch_sequence(seq[o:o + L], k, s, 0) if L >= k else np.zeros(0, dtype=pymap.MINMER)
            assert len(g) == len(e), (o, L, k, s)
            for name in ("hash", "wpos", "wpos_end", "strand"):
                assert (g[name] == e[name]).all(), (o, L, k, s, name)


def test_sketch_large_batch_random(gpu):
    rng = random.Random(8)
    seq = _noisy(99, 400000)
    offs = [rng.randrange(0, len(seq) - 3000) for _ in range(500)]
    lens = [rng.choice([1000, 1000, 1000, 500, 2000, 3000]) for _ in offs]
    got = gpu.sketch_fragments(seq, offs, lens, 15, 39, 11)
    for i in rng.sample(range(len(offs)), 80):
        e = pymap.sketch_sequence(seq[offs[i]:offs[i] + lens[i]], 15, 39, 11)
        assert len(got[i]) == len(e) and (got[i]["hash"] == e["hash"]).all() and (got[i]["wpos_end"] == e["wpos_end"]).all()


def test_sketch_table_form_equals_sort_form(gpu, monkeypatch):
    """The threshold-and-table form of sketch_fragments against the sorting form on fragments whose number of distinct k-mers
    is anything between a handful and all of them (duplications by 2 - 4 x, microsatellites, N runs): thresholds that
    turn out too small are raised, tables that overflow are bisected, and the records are the same bytes."""
    rng = random.Random(77)
    parts = []
    for i in range(60):
        unit = synth.random_dna(3000 + i, rng.choice([40, 150, 400, 900]))
        reps = rng.choice([1, 2, 2, 3, 4])
        parts.append(unit * reps + synth.random_dna(4000 + i, rng.choice([0, 200, 700])))
        if i % 9 == 0:
            parts.append(b"N" * 37 + b"AC" * 120 + b"T" * 90)
    seq = b"".join(parts)
    offs, lens = [], []
    for _ in range(400):
        L = rng.choice([300, 1000, 1000, 2500, 5000, 12000])
        o = rng.randrange(0, len(seq) - L)
        offs.append(o); lens.append(L)
    for k, s in [(15, 23), (15, 80), (21, 300), (28, 40), (9, 500)]:
        monkeypatch.setenv("WFM_SKETCH_TABLE", "1")
        a = gpu.sketch_fragments(seq, offs, lens, k, s, 5)
        monkeypatch.setenv("WFM_SKETCH_TABLE", "0")
        b = gpu.sketch_fragments(seq, offs, lens, k, s, 5)
        for i, (x, y) in enumerate(zip(a, b)):
            assert len(x) == len(y) and x.tobytes() == y.tobytes(), (k, s, i, offs[i], lens[i])
        for i in rng.sample(range(len(offs)), 12):
            e = pymap.sketch_sequence(seq[offs[i]:offs[i] + lens[i]], k, s, 5)
            assert len(a[i]) == len(e) and (a[i]["hash"] == e["hash"]).all() and (a[i]["wpos"] == e["wpos"]).all() and (a[i]["strand"] == e["strand"]).all()
