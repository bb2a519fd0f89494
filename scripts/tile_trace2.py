#!/usr/bin/env python
"""The split of a tile's life into snapshot load, steps and snapshot store (scripts/tile_trace.sh with TRACE_LEVEL=2 builds the library):
one s_memtime stamp per ten-step body and four around the loop, per wave, of the first 48 workgroups of the launch that begins at score 3000, a list of more than 1024 tiles,
of one C3 batch on one stream."""
import ctypes as C
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WFM_OVERLAP", "0")
from wfmash_amd import capi, synth
capi.LIB_PATH = os.path.join(ROOT, "wfmash_amd", os.environ.get("TRACE_LIB", "libwfmash_hip_trace.so"))
L = capi.load()
L.wfm_debug_tile_trace.restype = C.c_int
L.wfm_debug_tile_trace.argtypes = [C.c_void_p, C.c_size_t]
h = capi.Handle(0)
pairs = synth.pairs("C3", n_pairs=int(os.environ.get("PAIRS", "64")))
res = h.align(pairs)
print("statuses ok:", all(r.status == 0 for r in res))
n = 48 * 16 * 128 * 4
buf = np.zeros(n, dtype=np.uint64)
assert L.wfm_debug_tile_trace(buf.ctypes.data_as(C.c_void_p), n) == 0
a = buf.reshape(48, 16, 128, 4)[:, :, :, 0].astype(np.int64)
load, loop, store, life, body, ld_rows, ld_wlo, ld_win = [], [], [], [], [], [], [], []
cl_lo, cl_hi = 1, np.iinfo(np.int64).max  # (the kernel stamps one launch only: the blocks that begin at score 3000)
for b in range(48):
    for w in range(16):
        r = a[b, w]
        if r[120] == 0 or r[123] == 0 or r[123] < r[120] or r[123] - r[120] > 5_000_000 or r[120] < cl_lo or r[120] > cl_hi:
            continue
        if r[124] > r[120] and r[125] >= r[124] and r[121] >= r[125]:
            ld_rows.append(r[124] - r[120]); ld_wlo.append(r[125] - r[124]); ld_win.append(r[121] - r[125])
        load.append(r[121] - r[120]); loop.append(r[122] - r[121]); store.append(r[123] - r[122]); life.append(r[123] - r[120])
        for i in range(1, 9):
            d = r[i + 1] - r[i]
            if 0 < d < 1_000_000:
                body.append(d / 10.0)
f = lambda v: f"mean {np.mean(v):9.0f}  median {np.median(v):9.0f}  p90 {np.percentile(v, 90):9.0f}"
print(f"waves sampled: {len(life)}")
print("snapshot load + windows (begin -> loop):", f(load))
if ld_rows:
    print("  of it: the rows read                  ", f(ld_rows))
    print("         smallest live offset (LDS)     ", f(ld_wlo))
    print("         windows, constants, barrier    ", f(ld_win))
print("step loop (100 steps):                  ", f(loop))
print("snapshot store:                         ", f(store))
print("whole life of a wave:                   ", f(life))
print("cycles per step (bodies 1 .. 8):        ", f(body))
hist, edges = np.histogram(np.array(loop) / 100.0, bins=[0, 200, 400, 600, 800, 1000, 1200, 1400, 1600, 2000, 3000, 100000])
print("waves by cycles per step of their loop:", " ".join(f"<{int(e)}:{h}" for h, e in zip(hist, edges[1:])))
# the waves of one workgroup share their barriers: per workgroup the loop time of its slowest wave and the spread
wg = []
for b in range(48):
    ls = [a[b, w, 122] - a[b, w, 121] for w in range(16) if cl_lo <= a[b, w, 120] <= cl_hi and 0 < a[b, w, 122] - a[b, w, 121] < 5_000_000]
    if ls:
        wg.append((len(ls), min(ls), max(ls)))
print("per workgroup (waves, fastest loop, slowest loop):", wg[:24])
h.close()
