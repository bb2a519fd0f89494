"""m3 for sequences under the device winnower's threshold: the index build of the C1 substitute (128 chromosomes of
0.2 - 1.5 Mbp) and of LPA.subset (14 sequences of 0.1 - 0.3 Mbp) with the winnowing on the host's threads (the default below 4 M k-mers:
WFM_WINNOW_DEV_MIN) and with every sequence on the device winnower (WFM_WINNOW_DEV_MIN=0: one set of launches per sequence), at the host
thread counts given.  Map phase only; prints ms_index (median of --reps).  Run once per setting (the switches are read once per process):
   python scripts/winnow_short.py --threads 32 ; WFM_WINNOW_DEV_MIN=0 python scripts/winnow_short.py --threads 32"""
import json
import os
import statistics
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a = sys.argv[1:]
reps = int(a[a.index("--reps") + 1]) if "--reps" in a else 5
threads = int(a[a.index("--threads") + 1]) if "--threads" in a else 32
h = capi.Handle(0)
with tempfile.TemporaryDirectory() as td:
    fa = os.path.join(td, "c1.fa")
    names, lengths = synth.write_fasta(fa, synth.yeast_like(8, 16, 12_000_000))
    lpa = os.path.join(ROOT, "tests", "golden", "LPA.subset.fa.gz")
    for tag, path, prm in (("C1 substitute", fa, capi.map_default_params(threads=threads)),
                           ("LPA.subset", lpa, capi.map_default_params(percentage_identity=0.9, auto_pct_identity=0, max_mapping_length=50000, threads=threads))):
        idx, tot = [], []
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            ms = capi.map_paf(h, path, os.path.join(td, "m.paf"), params=prm)
            tot.append(time.perf_counter() - t0)
            idx.append(ms.ms_index)
        print(json.dumps({"workload": tag, "threads": threads, "WFM_WINNOW_DEV_MIN": os.environ.get("WFM_WINNOW_DEV_MIN", "default (4194304 k-mers)"),
                          "ms_index_median": round(statistics.median(idx[1:]), 1), "ms_index_min": round(min(idx[1:]), 1), "ms_index_first": round(idx[0], 1),
                          "map_s_median": round(statistics.median(tot[1:]), 3), "records": int(ms.written)}), flush=True)
h.close()
