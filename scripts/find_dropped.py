"""Which mapping records of a run produced no aligned record, and why: re-aligns them one by one through the C ABI and
prints status / score of the main alignment (a record is dropped silently when the aligner's status is not 0, wflign.cpp:150,
or when the writer's filters reject it, wflign_patch.cpp:2611-2724)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi
from oracle import wflign_host as W

fa, mapping, aligned = sys.argv[1:4]
def key(f): return (f[0], f[2], f[3], f[4], f[5])
rows = [l.rstrip("\n") for l in open(mapping) if l.strip()]
seen = set()
for l in open(aligned):
    f = l.split("\t")
    seen.add((f[0], f[5], f[4]))
print("mapping rows", len(rows), file=sys.stderr)
import subprocess
h = capi.Handle(0)
names = {}
for line in capi.host_fasta(fa).splitlines()[1:]:
    n, ln = line.split("\t"); names[n] = int(ln)
# align the rows one at a time through wfmh_align_paf and report those that give nothing
import tempfile
d = tempfile.mkdtemp()
missing = []
CH = 256
for i in range(0, len(rows), CH):
    m = os.path.join(d, "m.paf"); a = os.path.join(d, "a.paf")
    open(m, "w").write("\n".join(rows[i:i + CH]) + "\n")
    s = capi.align_paf(h, fa, m, a)
    if s.written != s.records:
        for r in rows[i:i + CH]:
            open(m, "w").write(r + "\n")
            s1 = capi.align_paf(h, fa, m, a)
            if s1.written != 1:
                missing.append(r)
print(json.dumps({"missing": len(missing), "rows": missing[:5]}))
for r in missing[:3]:
    row = W.parse_mashmap_row(r, 1000, 1000)
    ref = capi.host_fasta(fa, row["refId"], row["rStartPos"], row["rEndPos"] - 1).encode()
    q = capi.host_fasta(fa, row["qId"], row["qStartPos"], row["qEndPos"] - 1).encode()
    ref, q = W.upper_valid_dna(ref), W.upper_valid_dna(q)
    if row["rev"]:
        q = W.revcomp(q)
    res = h.align([(ref, q)])[0]
    print(json.dumps({"plen": len(ref), "tlen": len(q), "status": res.status, "score": res.score, "cells": int(res.cells)}))
    for env in ("WFM_P2", "WFM_TILE", "WFM_BAND"):
        pass
