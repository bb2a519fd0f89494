"""The bench line's secondary legs (C2 = LPA.subset all-vs-all -p 90 -P 50k; scaled C4 rank = 8 x 8 Mbp, one query
haplotype) run once each with WFM_DEBUG=1, so that the stage timings the library prints on stderr explain where the
wall time of `align_s` / `map_s` goes.  Usage: python scripts/legs_run.py [c2] [c4] [--threads N] [--mbp 8]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WFM_DEBUG", "1")
from wfmash_amd import capi, synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    threads = int(args[args.index("--threads") + 1]) if "--threads" in args else (os.cpu_count() or 1)
    mbp = float(args[args.index("--mbp") + 1]) if "--mbp" in args else 8.0
    reps = int(args[args.index("--reps") + 1]) if "--reps" in args else 2
    h = capi.Handle(0)
    with tempfile.TemporaryDirectory() as td:
        if "c2" in args or not [a for a in args if a in ("c2", "c4")]:
            lpa = os.path.join(ROOT, "tests", "golden", "LPA.subset.fa.gz")
            m, a = os.path.join(td, "lpa.m.paf"), os.path.join(td, "lpa.a.paf")
            for rep in range(reps):
                print(f"==== C2 pass {rep}", file=sys.stderr, flush=True)
                t1 = time.perf_counter()
                ms = capi.map_paf(h, lpa, m, params=capi.map_default_params(percentage_identity=0.9, auto_pct_identity=0, max_mapping_length=50000, threads=threads))
                t_map = time.perf_counter() - t1
                t1 = time.perf_counter()
                al = capi.align_paf(h, lpa, m, a, params={"threads": threads})
                t_al = time.perf_counter() - t1
                print(json.dumps({"leg": "C2", "pass": rep, "threads": threads, "map_s": t_map, "align_s": t_al, "records": int(al.records), "aligned_bp": int(al.aligned_bp),
                                  "aligned_bp_per_s_align": al.aligned_bp / t_al, "ms_gpu": al.ms_gpu, "gpu_share": al.ms_gpu * 1e-3 / t_al, "cells": int(al.cells)}), flush=True)
        if "c4" in args or not [a for a in args if a in ("c2", "c4")]:
            fa = os.path.join(td, "c4.fa")
            names, lengths = synth.write_fasta(fa, synth.pangenome(8, int(mbp * 1e6), n_sv=6))
            ql = os.path.join(td, "q.txt")
            open(ql, "w").write(names[0] + "\n")
            m, a = os.path.join(td, "m.paf"), os.path.join(td, "a.paf")
            for rep in range(reps):
                print(f"==== C4 rank ({mbp} Mbp) pass {rep}", file=sys.stderr, flush=True)
                t1 = time.perf_counter()
                ms = capi.map_paf(h, fa, m, params=capi.map_default_params(threads=threads, query_list=ql))
                t_map = time.perf_counter() - t1
                t1 = time.perf_counter()
                al = capi.align_paf(h, fa, m, a, params={"threads": threads})
                t_al = time.perf_counter() - t1
                print(json.dumps({"leg": "C4_rank_scaled", "mbp": mbp, "pass": rep, "threads": threads, "map_s": t_map, "ms_index": ms.ms_index, "ms_map": ms.ms_map,
                                  "ms_filter": ms.ms_filter, "align_s": t_al, "records": int(al.records), "aligned_bp": int(al.aligned_bp),
                                  "aligned_bp_per_s": al.aligned_bp / t_al, "ms_gpu": al.ms_gpu, "gpu_share": al.ms_gpu * 1e-3 / t_al, "cells": int(al.cells),
                                  "algorithmic_frac_gpu": 48.0 * al.cells / (al.ms_gpu * 1e-3) / 8e12 if al.ms_gpu else None}), flush=True)
    h.close()


if __name__ == "__main__":
    main()
