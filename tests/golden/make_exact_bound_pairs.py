"""Fixture of the round-3 regression `exact bounds on a small batch`: three padded yeast-like records (pure end gaps and
perfect matches: the greedy bound equals the optimal score) that came back with a false breakpoint one point under the
optimum when their batch reused ring memory of other jobs.  The pairs were dumped from a failing GPU
run (WFM_DUMP_FAIL, scripts/c1_debug3.py); this script adds the oracle's answer (oracle/wfa2p.c) and writes the fixture.

usage: python tests/golden/make_exact_bound_pairs.py DIR_WITH_fail_N.txt"""
import glob
import gzip
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle as O  # noqa: E402

out, seen = [], set()
for fn in sorted(glob.glob(os.path.join(sys.argv[1], "fail_*.txt"))):
    raw = open(fn, "rb").read().split(b"\n")
    p, t = raw[1], raw[2]
    if (p, t) in seen:
        continue
    seen.add((p, t))
    rc, ops, sc, _ = O.align_biwfa(p, t)
    assert rc == 0
    out.append({"pattern": p.decode(), "text": t.decode(), "hint": int(raw[0].split()[2]), "score": int(sc), "ops": ops.decode()})
with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exact_bound_pairs.json.gz"), "wt") as f:
    json.dump(out, f)
print(len(out), "pairs")
