"""Align-only timing of the regimes a pangenome run is made of (records resident in HBM, wfm_align_resident):

  c4rec   N x (50 kb query at 0.2 % against its 52 kb target window): what the align driver hands over for a pangenome
          mapping -- the target is padded by 1 kb on both sides (-E), so the end-to-end alignment carries two ~1 kb end
          gaps and reaches a score of ~2.7 k whatever the divergence (the bulk of a C4 rank)
  light   N x 50 kb pairs at 0.2 % divergence, no padding (scores of a few hundred)
  sv      N x 50 kb pairs at 0.2 % with one 5-30 kb deletion or insertion inside (records that span a structural variant)
  c3      64 x 50 kb at 5 %  (bench.py's workload)
  c5      8 x 100 kb at 15 % (the deep-wavefront config)

Prints one JSON line per regime: ms per pass, aligned bp/s, cells, 48 B x cells / GPU kernel time against 8 TB/s, and the
kernel time split.  WFM_DEBUG=1 adds the per-level lines of the driver on stderr.

    python scripts/align_regimes.py [--regimes light,sv,c3,c5] [--n-light 4096] [--n-sv 128] [--steps 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, synth  # noqa: E402


def light_pairs(n, length=50_000, rate=0.002, seed=0x11):
    out = []
    for i in range(n):
        t = synth.random_dna(seed * 100003 + i, length)
        out.append((t, synth.mutate(t, rate, seed * 7919 + i)))
    return out


def c4rec_pairs(n, length=50_000, rate=0.002, pad=1000, seed=0x33):
    out = []
    for i in range(n):
        t = synth.random_dna(seed * 100003 + i, length + 2 * pad)
        out.append((t, synth.mutate(t[pad:pad + length], rate, seed * 7919 + i)))
    return out


def sv_pairs(n, length=50_000, rate=0.002, seed=0x22):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        t = synth.random_dna(seed * 100003 + i, length)
        q = bytearray(synth.mutate(t, rate, seed * 7919 + i))
        l = int(rng.integers(5_000, 30_001))
        p = int(rng.integers(2_000, len(q) - l - 2_000))
        if i % 2 == 0:
            del q[p:p + l]                                   # deletion in the query
        else:
            q[p:p] = synth.random_dna(seed * 31 + i, l)      # insertion
        out.append((t, bytes(q)))
    return out


def run(h, name, pairs, steps):
    ss = h.upload(pairs)
    qb = sum(len(q) for _, q in pairs)
    h.align_resident(ss, collect=False)  # warm-up (arenas)
    acc = dict(cells=0, ms_k=0.0, ms_tile=0.0, ms_bp=0.0, ms_base=0.0, levels=0, tile_launches=0, bp_launches=0, base_launches=0,
               p2_launches=0, p2_jobs=0, p2_more=0, busy=0.0)
    t0 = time.perf_counter()
    for _ in range(steps):
        failed = h.align_resident(ss, collect=False)
        assert failed == 0, failed
        st = h.stats()
        acc["cells"] += st.cells; acc["ms_k"] += st.ms_kernels; acc["ms_tile"] += st.ms_tile
        acc["ms_bp"] += st.ms_breakpoint - st.ms_tile; acc["ms_base"] += st.ms_base; acc["levels"] = st.levels
        acc["tile_launches"] += st.tile_launches; acc["bp_launches"] += st.bp_launches; acc["base_launches"] += st.base_launches
        acc["p2_launches"] += st.p2_launches; acc["p2_jobs"] += st.p2_jobs; acc["p2_more"] += st.p2_more; acc["busy"] += st.ms_any_busy
    dt = (time.perf_counter() - t0) / steps
    res = h._collect(ss)
    scores = np.array([r.score for r in res])
    out = {"regime": name, "pairs": len(pairs), "ms_per_pass": round(dt * 1e3, 2), "aligned_bp_per_s": round(qb / dt),
           "cells_per_pass": acc["cells"] // steps, "score_mean": float(scores.mean()), "score_max": int(scores.max()),
           "gcells_per_s_wall": round(acc["cells"] / steps / dt / 1e9, 2),
           "alg_frac_of_8TBs_wall": round(48.0 * acc["cells"] / steps / dt / 8e12, 3),
           "kernel_ms": {k[3:]: round(acc[k] / steps, 2) for k in ("ms_tile", "ms_bp", "ms_base")},
           "gpu_busy_ms": round(acc["busy"] / steps, 2), "alg_frac_of_8TBs_busy": round(48.0 * acc["cells"] / max(acc["busy"], 1e-9) / 8e9, 3),
           "launches": {k: acc[k] // steps for k in ("tile_launches", "bp_launches", "base_launches", "p2_launches", "p2_jobs", "p2_more")},
           "levels": acc["levels"]}
    ss.free()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--regimes", default="c4rec,light,sv,c3,c5")
    ap.add_argument("--n-light", type=int, default=1024)
    ap.add_argument("--n-sv", type=int, default=128)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    h = capi.Handle(0)
    for r in a.regimes.split(","):
        if r == "light":
            pairs = light_pairs(a.n_light)
        elif r == "c4rec":
            pairs = c4rec_pairs(a.n_light)
        elif r == "sv":
            pairs = sv_pairs(a.n_sv)
        elif r == "mix":
            pairs = light_pairs(a.n_light) + sv_pairs(a.n_sv)
        elif r == "c3":
            pairs = synth.pairs("C3")
        elif r == "c5":
            pairs = synth.pairs("C5", n_pairs=8)
        else:
            raise SystemExit(f"unknown regime {r}")
        print(json.dumps(run(h, r, pairs, a.steps)), flush=True)
    h.close()


if __name__ == "__main__":
    main()
