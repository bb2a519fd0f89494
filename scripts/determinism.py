"""Repeated alignment of one batch of padded records: every run must give the same bytes (and no failures)."""
import os, sys, hashlib, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wfmash_amd import capi, synth

def rec(seed, L, snp, indel, pad_t, pad_q=0):
    base = synth.random_backbone(seed, L + 2 * pad_t + 64)
    hap = synth.haplotype(base, seed + 1, snp=snp, indel=indel, n_sv=0)
    t = base.tobytes()
    return t[:L + 2 * pad_t], hap.tobytes()[pad_t - pad_q:pad_t - pad_q + L]

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rng = np.random.default_rng(5)
    items = []
    for i in range(n):
        L = int(rng.choice([1000, 1400, 2000, 5000, 8000, 12000]))
        items.append(rec(1000 + i, L, float(rng.choice([0.003, 0.006, 0.012])), float(rng.choice([0.0003, 0.001])), int(rng.choice([300, 600, 1000])), int(rng.choice([0, 0, 100]))))
    h = capi.Handle(0)
    ref = None
    bad = 0
    for r in range(reps):
        res = h.align(items)
        fails = [i for i, x in enumerate(res) if x.status != 0]
        dig = hashlib.sha256(b"|".join((x.ops or b"F") for x in res)).hexdigest()[:12]
        if ref is None:
            ref = res
        diff = [i for i, (a, b) in enumerate(zip(ref, res)) if a.ops != b.ops]
        bad += bool(fails or diff)
        print(json.dumps({"rep": r, "digest": dig, "failed": fails[:5], "differ_from_first": diff[:5],
                          "sizes": [(len(items[i][0]), len(items[i][1])) for i in (fails + diff)[:3]]}), flush=True)
    print("nondeterministic or failing runs:", bad)

main()
