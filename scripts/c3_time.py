#!/usr/bin/env python
"""Wall time of one C3 batch (64 pairs of 50 kb at 5 %) through wfm_align, `--reps` times after `--warmup`, statuses and CIGARs NOT checked: the
timing harness for experiments whose results are wrong by design and for A/B runs of two builds of the library (WFM_LIB=<file in wfmash_amd/>).
Prints the median and the spread; WFM_OVERLAP=0 gives the one-stream figure the per-launch numbers come from."""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wfmash_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--config", default="C3")
ap.add_argument("--pairs", type=int, default=64)
a = ap.parse_args()
h = capi.Handle(0)
pairs = synth.pairs(a.config, n_pairs=a.pairs)
ts = []
bad = 0
for i in range(a.warmup + a.reps):
    t0 = time.perf_counter()
    res = h.align(pairs)
    dt = (time.perf_counter() - t0) * 1e3
    if i >= a.warmup:
        ts.append(dt)
    bad += sum(r.status != 0 for r in res)
print(f"{a.config} x {a.pairs}: median {statistics.median(ts):.2f} ms  min {min(ts):.2f}  max {max(ts):.2f}  (host buffers in: upload included)  failed problems {bad}  lib {os.environ.get('WFM_LIB', 'default')}  overlap {os.environ.get('WFM_OVERLAP', '1')}")
h.close()
