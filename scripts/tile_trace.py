#!/usr/bin/env python
"""Where a score step of wfa_tile2_kernel spends its cycles (scripts/tile_trace.sh builds the instrumented library).  One C3 batch on one stream;
the stamps of the last launch of more than 1024 tiles: per wave and step  [3] step begins, [0] barrier let go, [1] recurrences issued (mailbox read
waited for), [2] extension known.  Prints the mean cycles of the four segments and the spread between the waves of a workgroup."""
import ctypes as C
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WFM_OVERLAP", "0")
from wfmash_amd import capi, synth
capi.LIB_PATH = os.path.join(ROOT, "wfmash_amd", "libwfmash_hip_trace.so")
L = capi.load()
L.wfm_debug_tile_trace.restype = C.c_int
L.wfm_debug_tile_trace.argtypes = [C.c_void_p, C.c_size_t]
h = capi.Handle(0)
pairs = synth.pairs("C3", n_pairs=int(os.environ.get("PAIRS", "64")))
res = h.align(pairs)
assert all(r.status == 0 for r in res)
n = 48 * 16 * 128 * 4
buf = np.zeros(n, dtype=np.uint64)
rc = L.wfm_debug_tile_trace(buf.ctypes.data_as(C.c_void_p), n)
assert rc == 0, rc
a = buf.reshape(48, 16, 128, 4).astype(np.int64)
T = 100
seg_names = ["barrier wait (3 -> 0)", "mailbox read + recurrences (0 -> 1)", "probe / extension (1 -> 2)", "bookkeeping, mailbox write, shifts (2 -> next 3)"]
tot = np.zeros(4); cnt = 0
per_step = []
spread = []
for b in range(48):
    waves = [w for w in range(16) if a[b, w, 5, 0] != 0]
    if len(waves) < 2:
        continue
    for w in waves:
        for t in range(12, T - 2):
            s3, s0, s1, s2 = a[b, w, t, 3], a[b, w, t, 0], a[b, w, t, 1], a[b, w, t, 2]
            n3 = a[b, w, t + 1, 3]
            if min(s3, s0, s1, s2, n3) == 0:
                continue
            d = np.array([s0 - s3, s1 - s0, s2 - s1, n3 - s2], dtype=np.float64)
            if (d < 0).any() or d.sum() > 1e6:
                continue
            tot += d; cnt += 1
            per_step.append(d.sum())
    # spread of the arrival at the barrier between the waves of the workgroup
    for t in range(12, T - 2):
        arr = np.array([a[b, w, t, 3] for w in waves])
        if (arr == 0).any():
            continue
        spread.append(arr.max() - arr.min())
print(f"steps sampled: {cnt}; cycles per step (s_memtime ticks): mean {np.mean(per_step):.0f}  median {np.median(per_step):.0f}  p90 {np.percentile(per_step, 90):.0f}")
for nme, v in zip(seg_names, tot / max(cnt, 1)):
    print(f"  {nme:55s} {v:8.0f}  ({100 * v / (tot.sum() / max(cnt, 1)):.0f} %)")
print(f"arrival spread between the waves of a workgroup at a step's begin: mean {np.mean(spread):.0f}  p90 {np.percentile(spread, 90):.0f}")
h.close()
