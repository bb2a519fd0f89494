"""Randomised parity campaign of the map path: small synthetic pangenomes (3 - 8 haplotypes of 120 - 500 kb; SNPs 0.1 - 4 %, short indels,
structural variants, an interspersed repeat family and a microsatellite so that frequent k-mers, ties and low-complexity windows occur) through
wfmh_map_paf with the defaults (identity estimate, index, L1 / L2, filters), one query haplotype per round held against the stage oracles on
the reference's own addMinmers and the reference's own filter code (oracle/_ref: this only runs where /root/reference was there at build time
or the built libraries travelled).  A tool for a GPU box beside the suite (it lives under tests/ because it calls the oracles; pytest does not collect it).

Usage: python tests/fuzz_map.py [--rounds 12] [--seed 1]
Prints one line per round and a JSON summary; exit code 1 if a round's records differ."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, synth  # noqa: E402
from oracle import map_ani as ANI  # noqa: E402
from oracle import map_pipeline as MP  # noqa: E402
from oracle import pyfilter, pymap  # noqa: E402


def make(rng, seed):
    n_haps = int(rng.integers(3, 9))
    length = int(rng.choice([120_000, 200_000, 350_000, 500_000]))
    base = synth.random_backbone(seed, length)
    # an interspersed repeat (copies at 2 - 6 % divergence) and a microsatellite
    rep = synth.random_backbone(seed + 7, int(rng.choice([300, 1200, 4000])))
    for _ in range(int(rng.integers(0, 25))):
        p = int(rng.integers(0, length - len(rep)))
        base[p:p + len(rep)] = synth.haplotype(rep, int(rng.integers(1 << 30)), snp=float(rng.choice([0.02, 0.06])), indel=0.0, n_sv=0)[:len(rep)]
    if rng.random() < 0.5:
        p = int(rng.integers(0, length - 3000))
        unit = np.frombuffer(rng.choice([b"AC", b"CAG", b"A", b"AATG"]), dtype=np.uint8)
        base[p:p + 3000] = np.resize(unit, 3000)
    snp = float(rng.choice([0.001, 0.005, 0.02, 0.04]))
    indel = float(rng.choice([0.0, 0.0001, 0.002]))
    recs = []
    for hp in range(n_haps):
        s = synth.haplotype(base, seed * 16 + hp, snp=snp, indel=indel, n_sv=int(rng.integers(0, 4)), sv_min=2_000, sv_max=20_000)
        recs.append((f"hap{hp + 1}#1#chr1", s.tobytes()))
    return recs, dict(haps=n_haps, length=length, snp=snp, indel=indel)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    if not (pymap.have_ref() and pyfilter.have_ref()):
        print(json.dumps({"error": "oracle/_ref is not built in this checkout"}))
        sys.exit(2)
    h = capi.Handle(0)
    bad = 0
    total = 0
    with tempfile.TemporaryDirectory() as td:
        for rnd in range(a.rounds):
            rng = np.random.default_rng(a.seed * 1000 + rnd)
            recs, what = make(rng, a.seed * 1000 + rnd)
            fa = os.path.join(td, f"f{rnd}.fa")
            names, _ = synth.write_fasta(fa, recs)
            m = os.path.join(td, f"f{rnd}.paf")
            t0 = time.time()
            capi.map_paf(h, fa, m, params=capi.map_default_params(threads=os.cpu_count() or 1))
            t_gpu = time.time() - t0
            got = open(m).read().splitlines()
            q = int(rng.integers(0, len(recs)))
            t0 = time.time()
            pct = np.float32(ANI.estimate_identity([s for _, s in recs], MP.ref_groups(names), 50, -2.0))
            S = MP.sketch_size(pct, 1000, 15)
            maps, _, _ = MP.map_queries(recs, pct, queries={q})
            exp = pyfilter.ref_filter("subset", maps[q], fa, names[q],
                                      capi.map_default_params(percentage_identity=float(pct), auto_pct_identity=0, sketch_size=S)).splitlines()
            t_or = time.time() - t0
            mine = [l for l in got if l.split("\t", 1)[0] == names[q]]
            same = mine == exp
            bad += not same
            total += len(exp)
            print(f"round {rnd}: {what}, identity {float(pct):.4f}, sketch {S}, query {names[q]}: {len(exp)} records, identical {same} (gpu {t_gpu:.2f} s, oracle {t_or:.1f} s)", flush=True)
            if not same:
                with open(os.path.join(os.path.dirname(m), f"bad{rnd}.txt"), "w") as f:
                    f.write("\n".join(mine) + "\n----\n" + "\n".join(exp) + "\n")
            os.unlink(fa)
    h.close()
    print(json.dumps({"rounds": a.rounds, "rounds_differ": bad, "records": total, "seed": a.seed}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
