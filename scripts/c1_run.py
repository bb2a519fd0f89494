"""C1 substitute end to end (data/scerevisiae8.fa.gz is a missing blob; synth.yeast_like stands in for it): 8 strains x 16
chromosomes of one 12 Mbp genome, all-vs-all, defaults (identity from the ANI estimate, PanSN prefix grouping), map + align on one
GPU.  Prints one JSON line per pass.   python scripts/c1_run.py [--reps 2] [--threads N] [--genome-bp 12000000]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, synth  # noqa: E402


def main():
    a = sys.argv[1:]
    reps = int(a[a.index("--reps") + 1]) if "--reps" in a else 2
    threads = int(a[a.index("--threads") + 1]) if "--threads" in a else (os.cpu_count() or 1)
    gbp = int(a[a.index("--genome-bp") + 1]) if "--genome-bp" in a else 12_000_000
    h = capi.Handle(0)
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "c1.fa")
        names, lengths = synth.write_fasta(fa, synth.yeast_like(8, 16, gbp))
        m, al = os.path.join(td, "m.paf"), os.path.join(td, "a.paf")
        for rep in range(reps):
            t0 = time.perf_counter()
            ms = capi.map_paf(h, fa, m, params=capi.map_default_params(threads=threads))
            t_map = time.perf_counter() - t0
            t0 = time.perf_counter()
            s = capi.align_paf(h, fa, m, al, params={"threads": threads})
            t_al = time.perf_counter() - t0
            print(json.dumps({"config": "C1 substitute", "pass": rep, "sequences": len(names), "total_bp": int(sum(lengths)), "threads": threads,
                              "map_s": round(t_map, 3), "ms_identity": round(ms.ms_identity), "ms_index": round(ms.ms_index), "ms_map": round(ms.ms_map),
                              "ms_filter": round(ms.ms_filter), "mapping_records": int(ms.written), "pct": round(ms.percentage_identity, 4), "sketch": int(ms.sketch_size), "fragments": int(ms.fragments),
                              "align_s": round(t_al, 3), "records": int(s.records), "aligned_bp": int(s.aligned_bp),
                              "aligned_bp_per_s_align": round(s.aligned_bp / t_al), "aligned_bp_per_s_end_to_end": round(s.aligned_bp / (t_al + t_map)),
                              "ms_gpu": round(s.ms_gpu), "cells": int(s.cells), "batches": int(s.batches)}), flush=True)
    h.close()


if __name__ == "__main__":
    main()
