"""The threshold-and-table form of sketchSequence (sketch_fragments_table_kernel) on the CPU: its model
(oracle/map_sketch_table.py) against sketchSequence itself -- the oracle's restatement and, where oracle/_ref is built, the
reference's own header -- with tables so small that thresholds overflow and are bisected, start values that are far too small
or too large, duplicated and low-complexity fragments."""
import random

import numpy as np
import pytest

from oracle import map_sketch_table as ST
from oracle import pymap
from wfmash_amd import synth


def _fragment(rng, seed):
    unit = synth.random_dna(seed, rng.choice([7, 40, 150, 600]))
    body = unit * rng.choice([1, 2, 3, 5]) + synth.random_dna(seed + 1, rng.choice([0, 100, 900]))
    if rng.random() < 0.3:
        body = body[:len(body) // 2] + b"N" * rng.choice([1, 20]) + body[len(body) // 2:]
    if rng.random() < 0.2:
        body += b"AC" * 60 + b"T" * 50
    return body


def _want(seq, k, s, which):
    e = pymap.sketch_sequence(seq, k, s, 0, which=which)
    return [(int(x["hash"]), int(x["wpos"]), int(x["wpos_end"]), int(x["strand"])) for x in e]


@pytest.mark.parametrize("which", ["oracle", "ref"])
def test_table_form_gives_sketch_sequence_whatever_path_the_threshold_takes(which):
    if which == "ref" and not pymap.have_ref():
        pytest.skip("oracle/_ref is not built in this checkout")
    rng = random.Random(11)
    paths = set()
    for trial in range(400):
        seq = _fragment(rng, 500 + trial)
        k = rng.choice([9, 15, 21])
        if len(seq) < k + 5:
            continue
        s = rng.choice([3, 10, 25, 60])
        h, st = pymap.hash_kmers(seq, k)
        want = _want(seq, k, s, which)
        for cap in (s, s + 1, 2 * s + 5, 100000):
            for tau0 in (None, 1, ST.TMAX, 1 << rng.randrange(40, 63)):
                got, tried = ST.sketch_table(h, st, s, cap, tau0)
                assert got == want, (trial, k, s, cap, tau0)
                up = any(b > a for a, b in zip(tried, tried[1:]))
                down = any(b < a for a, b in zip(tried, tried[1:]))
                paths.add((up, down))
    assert paths == {(False, False), (True, False), (False, True), (True, True)}  # straight, raised, bisected, both
