"""The map phase of one rank of C4 (scripts/c4_rank.py's workload) run several times in one process under different
environment switches, on one generated FASTA: an A/B of host-side choices whose effect is a few tens of milliseconds.

Usage: python scripts/map_repeat.py [--haps 8] [--mbp 248.956422] [--reps 3] NAME=VALUE[,NAME=VALUE...] ...
Each argument is one variant (an empty string '' = the defaults).  Prints one JSON line per run and the md5 of the PAF."""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, dist, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--haps", type=int, default=8)
    ap.add_argument("--mbp", type=float, default=248.956422)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--all", action="store_true", help="every haplotype as a query (the all-vs-all job's map phase) instead of rank 0's share")
    ap.add_argument("variants", nargs="*", default=[""])
    a = ap.parse_args()
    d = tempfile.mkdtemp()
    fa = os.path.join(d, "c4.fa")
    names, lengths = synth.write_fasta(fa, synth.pangenome(a.haps, int(a.mbp * 1e6)))
    mine = list(names) if a.all else [names[i] for i in dist.shard_queries(lengths, 8)[0]]
    qlist = os.path.join(d, "queries.txt")
    open(qlist, "w").write("\n".join(mine) + "\n")
    P = capi.map_default_params(threads=os.cpu_count() or 1, query_list=qlist)
    h = capi.Handle(0)
    out = os.path.join(d, "o.paf")
    capi.map_paf(h, fa, out, params=P)  # warm-up: arenas, code objects
    for rep in range(a.reps):
        for v in a.variants:
            sets = dict(kv.split("=", 1) for kv in v.split(",") if kv)
            old = {k: os.environ.get(k) for k in sets}
            os.environ.update(sets)
            t0 = time.time()
            s = capi.map_paf(h, fa, out, params=P)
            wall = time.time() - t0
            for k, val in old.items():
                if val is None:
                    del os.environ[k]
                else:
                    os.environ[k] = val
            print(json.dumps({"variant": v, "rep": rep, "wall_s": round(wall, 3), "ms_identity": round(s.ms_identity), "ms_index": round(s.ms_index),
                              "ms_map": round(s.ms_map), "ms_filter": round(s.ms_filter), "ms_total": round(s.ms_total), "ms_wall": round(s.ms_wall),
                              "md5": hashlib.md5(open(out, "rb").read()).hexdigest()[:12]}), flush=True)
    h.close()


if __name__ == "__main__":
    main()
