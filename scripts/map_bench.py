"""Map-phase timing on a synthetic PanSN pangenome (all-vs-all): N haplotypes of one random
backbone, each with its own SNPs/indels.  Prints one JSON line with the stage times wfmh_map
reports.  Usage: python scripts/map_bench.py [--haps 8] [--mbp 4] [--div 0.01] [--pct 0 (auto)]"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--haps", type=int, default=8)
ap.add_argument("--mbp", type=float, default=4.0)
ap.add_argument("--div", type=float, default=0.01)
ap.add_argument("--pct", type=float, default=0.0, help="identity threshold as a fraction; 0 = estimate (ani50-2)")
ap.add_argument("--keep", default="")
a = ap.parse_args()

L = int(a.mbp * 1e6)
base = synth.random_dna(0xA11, L)
d = a.keep or tempfile.mkdtemp()
fa = os.path.join(d, "pan.fa")
t0 = time.time()
with open(fa, "w") as f:
    for h in range(a.haps):
        s = synth.mutate(base, a.div, 0xBEEF + h).decode()
        f.write(f">hap{h}#1#chr1\n")
        for i in range(0, len(s), 80):
            f.write(s[i:i + 80] + "\n")
t_gen = time.time() - t0
h = capi.Handle(0)
ap_threads = os.cpu_count() or 1
P = capi.map_default_params(threads=ap_threads) if a.pct == 0 else capi.map_default_params(percentage_identity=a.pct, auto_pct_identity=0, threads=ap_threads)
out = os.path.join(d, "map.paf")
t0 = time.time()
s = capi.map_paf(h, fa, out, params=P)
wall = time.time() - t0
print(json.dumps({"haps": a.haps, "mbp_each": a.mbp, "gen_s": round(t_gen, 1), "wall_s": round(wall, 2), "pct": round(float(s.percentage_identity), 4),
                  "sketch": s.sketch_size, "windows": s.index_windows, "fragments": s.fragments, "l2_mappings": s.l2_mappings,
                  "written": s.written, "ms_index": round(s.ms_index), "ms_map": round(s.ms_map), "ms_filter": round(s.ms_filter),
                  "ms_total": round(s.ms_total), "query_mbp_per_s": round(s.query_bp / 1e6 / (s.ms_total / 1e3), 2)}))
h.close()
