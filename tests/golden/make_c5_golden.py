#!/usr/bin/env python
"""Generates tests/golden/c5_golden.json.gz: the first two full-size C5 pairs (100 kb at 15 % divergence,
synth.pairs("C5")) aligned by the CPU oracle (oracle/wfa2p.c, BiWFA).  About 10^10 wavefront cells per pair -- minutes of
one core each -- which is why the result is a committed fixture instead of a computation inside the GPU test:
    python tests/golden/make_c5_golden.py
Stored: score, number of ops, sha256 of the op string, and the run-length CIGAR itself (inputs are regenerated from the
seeded generator)."""
import gzip
import hashlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from wfmash_amd import synth  # noqa: E402

pairs = synth.pairs("C5", n_pairs=2)
t0 = time.time()
ops, scores, st, failed = O.align_batch_biwfa([p for p, _ in pairs], [q for _, q in pairs], nthreads=2)
assert failed == 0
out = {"config": "C5", "generator": "wfmash_amd.synth.pairs('C5', n_pairs=2)", "pairs": []}
for (p, q), o, sc in zip(pairs, ops, scores):
    rle = "".join(f"{len(m.group(0))}{m.group(0)[0]}" for m in re.finditer(r"M+|X+|I+|D+", o.decode()))
    out["pairs"].append({"plen": len(p), "tlen": len(q), "score": int(sc), "n_ops": len(o), "sha256": hashlib.sha256(o).hexdigest(), "rle": rle})
with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_golden.json.gz"), "wt") as f:
    json.dump(out, f)
print("scores", [p["score"] for p in out["pairs"]], "seconds", round(time.time() - t0, 1))
