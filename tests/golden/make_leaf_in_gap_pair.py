#!/usr/bin/env python
"""Generates tests/golden/leaf_in_gap_pair.json.gz: one mapping record of the 40 Mbp C4 variant (synth.pangenome(8, 40_000_000):
query hap7#1#chr1 9257000-9300000 on the '-' strand against hap5#1#chr1 9296138-9339138 padded by 1 kb, as the align driver
cuts it) with the CPU oracle's score and CIGAR digest.  Kept as a fixture because regenerating it costs two 40 Mbp haplotypes.
    python tests/golden/make_leaf_in_gap_pair.py"""
import gzip
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from oracle import wflign_host as W  # noqa: E402
from wfmash_amd import synth  # noqa: E402

base = synth.random_backbone(0xC4, 40_000_000)
hap5 = synth.haplotype(base, (0xC4 << 8) + 4)
hap7 = synth.haplotype(base, (0xC4 << 8) + 6)
assert len(hap5) == 40071912 and len(hap7) == 40110425
q = W.revcomp(W.upper_valid_dna(hap7[9257000:9300000].tobytes()))
ref = W.upper_valid_dna(hap5[9296138 - 1000:9339138 + 1000].tobytes())
rc, ops, sc, _ = O.align_biwfa(ref, q)
assert rc == 0
with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "leaf_in_gap_pair.json.gz"), "wt") as f:
    json.dump({"pattern": ref.decode(), "text": q.decode(), "score": int(sc), "ops_sha": hashlib.sha256(ops).hexdigest(), "n_ops": len(ops)}, f)
print("score", sc, "ops", len(ops))
