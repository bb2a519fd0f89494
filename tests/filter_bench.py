"""The host post-processing (f3: chaining, plane sweep, scaffolds, PAF text) of ONE chromosome-sized query on synthetic L2
mappings shaped like a full-size C4 rank's (249 k fragments x 7 target haplotypes, colinear with jitter, a few inversions and
noise): wall time of wfmh_test_filter("subset") per thread count, the stage times of chain_mappings (WFM_FILTER_TIMES), and --
with --check -- the text held against the reference's own filter code (oracle/_ref/libref_filter.so).  No GPU needed.

Usage: python tests/filter_bench.py [--frags 249000] [--threads 1,8,32] [--check]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi  # noqa: E402


def make(frags, haps=8, w=1000, seed=5):
    rng = np.random.default_rng(seed)
    L = frags * w + 437
    parts = []
    for t in range(haps):
        if t == 4:  # the query's own haplotype
            continue
        keep = rng.random(frags) < 0.995
        i = np.nonzero(keep)[0]
        shift = np.cumsum(rng.integers(-3, 4, frags))[i] + rng.integers(-20000, 20000)
        ref = np.clip(i * w + shift + rng.integers(-40, 41, len(i)), 0, L - w - 1)
        m = np.zeros(len(i), dtype=capi.MAPPING_DTYPE)
        m["refSeqId"] = t
        m["refStartPos"] = ref
        m["queryStartPos"] = i * w
        m["blockLength"] = w
        m["n_merged"] = 1
        m["conservedSketches"] = rng.integers(15, 24, len(i))
        m["nucIdentity"] = rng.integers(9600, 9990, len(i))
        m["kmerComplexity"] = rng.integers(80, 100, len(i))
        # an inverted stretch and some off-diagonal noise
        a = rng.integers(0, max(1, len(i) - 200))
        m["flags"][a:a + 150] = 1
        m["refStartPos"][a:a + 150] = m["refStartPos"][a:a + 150][::-1]
        noise = rng.random(len(i)) < 0.002
        m["refStartPos"][noise] = rng.integers(0, L - w - 1, int(noise.sum()))
        parts.append((i, m))
    # fragment order, per fragment by target (the order wfm_map_fragments returns)
    idx = np.concatenate([p[0] for p in parts])
    allm = np.concatenate([p[1] for p in parts])
    order = np.lexsort((allm["refSeqId"], idx))
    return allm[order], L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frags", type=int, default=249000)
    ap.add_argument("--threads", default="1,8,32")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    maps, L = make(a.frags)
    td = tempfile.mkdtemp()
    fa = os.path.join(td, "pan.fa")
    names = [f"hap{t + 1}#1#chr1" for t in range(8)]
    with open(fa, "wb") as f:
        f.truncate(8 * (L + L // 60 + 64))
    with open(fa + ".fai", "w") as f:
        off = 16
        for n in names:
            f.write(f"{n}\t{L}\t{off}\t60\t61\n")
            off += L + L // 60 + 32
    P = capi.map_default_params(percentage_identity=0.9787, auto_pct_identity=0, sketch_size=23)
    q = names[4]
    print(f"{len(maps)} mappings, query {q} of {L} bp", flush=True)
    texts = {}
    for t in [int(x) for x in a.threads.split(",")]:
        os.environ["WFM_FILTER_THREADS"] = str(t)
        best = 1e9
        for _ in range(a.reps):
            t0 = time.perf_counter()
            texts[t] = capi.host_filter("subset", maps, fa, q, P)
            best = min(best, time.perf_counter() - t0)
        print(f"threads {t}: {best * 1e3:.1f} ms, {texts[t].count(chr(10))} records", flush=True)
    assert len(set(texts.values())) == 1, "the text depends on the thread count"
    if a.check:
        from oracle import pyfilter
        t0 = time.perf_counter()
        exp = pyfilter.ref_filter("subset", maps, fa, q, P)
        print(f"reference filter code: {(time.perf_counter() - t0) * 1e3:.0f} ms, identical: {exp == next(iter(texts.values()))}", flush=True)


if __name__ == "__main__":
    main()
