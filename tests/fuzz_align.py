"""Randomised parity campaign of the align path: batches of mixed problems (substitutions / indels at 0 - 40 %, block deletions, insertions,
tandem duplications and inversions, microsatellites and homopolymers, N runs and soft-masked stretches, unrelated pairs, empty and one-base
sequences, very unequal lengths; BiWFA end-to-end and the two ends-free patch forms) through libwfmash_hip.so, every result held against the
CPU oracle (oracle/wfa2p.c; its calls run on a thread pool, ctypes releases the GIL).  A tool for a GPU box beside the suite (it lives under tests/ because it calls the oracle; pytest does not collect it):
the suite's own cases came out of runs like this one.

Usage: python tests/fuzz_align.py [--rounds 20] [--items 240] [--seed 1] [--max-len 9000] [--threads N]
Prints one line per round and a JSON summary; exit code 1 if anything differs (the failing problems are written to --dump as FASTA-like text)."""
import argparse
import json
import os
import random
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, synth  # noqa: E402
from oracle import pyoracle  # noqa: E402

COMP = bytes.maketrans(b"ACGTacgtN", b"TGCAtgcaN")


def revcomp(s):
    return s.translate(COMP)[::-1]


def gen_pair(rng, k, max_len):
    """one (pattern, text) pair; k picks the family"""
    L = rng.choice([0, 1, 2, 7, 33, 64, 100, 129, 255, 256, 257, 600, 1000, 1500, 2500, 4000, 6000, max_len])
    L = min(L, max_len)
    p = synth.random_dna(rng.getrandbits(40), L)
    rate = rng.choice([0.0, 0.001, 0.01, 0.03, 0.05, 0.1, 0.2, 0.4])
    t = synth.mutate(p, rate, rng.getrandbits(40)) if L else b""
    fam = k % 12
    if fam == 1 and L > 50:  # block deletion
        a = rng.randrange(0, L // 2); b = min(len(t), a + rng.choice([30, 200, 1000, 3000]))
        t = t[:a] + t[b:]
    elif fam == 2 and L > 50:  # block insertion
        a = rng.randrange(0, len(t) + 1)
        t = t[:a] + synth.random_dna(rng.getrandbits(40), rng.choice([30, 200, 1000, 3000])) + t[a:]
    elif fam == 3 and L > 200:  # tandem duplication
        a = rng.randrange(0, L // 2); n = rng.choice([20, 150, 800])
        t = t[:a + n] + t[a:a + n] * rng.choice([1, 2, 5]) + t[a + n:]
    elif fam == 4 and L > 200:  # inversion
        a = rng.randrange(0, L // 2); n = rng.choice([40, 300, 1500])
        t = t[:a] + revcomp(t[a:a + n]) + t[a + n:]
    elif fam == 5:  # microsatellite / homopolymer context
        unit = rng.choice([b"A", b"AC", b"CAG", b"AATG", b"T"])
        rep = unit * (rng.choice([20, 200, 1000]) // len(unit) + 1)
        a = rng.randrange(0, len(p) + 1)
        p = p[:a] + rep + p[a:]
        t = synth.mutate(p, rate, rng.getrandbits(40))
        if rng.random() < 0.5:  # a different repeat count on the other side
            b = rng.randrange(0, len(t) + 1)
            t = t[:b] + unit * rng.randrange(1, 40) + t[b:]
    elif fam == 6 and L > 20:  # N runs / soft-masked stretches (the byte kernels)
        a = rng.randrange(0, L); n = rng.choice([1, 5, 60, 400])
        if rng.random() < 0.5:
            p = p[:a] + b"N" * n + p[a + n:]
        else:
            p = p[:a] + p[a:a + n].lower() + p[a + n:]
        if rng.random() < 0.5 and len(t) > 10:
            b = rng.randrange(0, len(t))
            t = t[:b] + b"N" * rng.choice([1, 3, 50]) + t[b + 3:]
    elif fam == 7:  # unrelated
        t = synth.random_dna(rng.getrandbits(40), rng.choice([0, 1, 50, 400, 2000]))
    elif fam == 8:  # very unequal lengths around a shared core
        core = synth.random_dna(rng.getrandbits(40), rng.choice([50, 500, 2000]))
        p = synth.random_dna(rng.getrandbits(40), rng.choice([0, 100, 2500])) + core
        t = synth.mutate(core, rate, rng.getrandbits(40)) + synth.random_dna(rng.getrandbits(40), rng.choice([0, 80, 1800]))
    elif fam == 9:  # one side empty or a single base
        if rng.random() < 0.5:
            p = rng.choice([b"", b"A"])
        else:
            t = rng.choice([b"", b"C"])
    elif fam == 10 and L > 100:  # two diverged halves around an identical middle (breakpoints at the ends of long runs)
        h = L // 3
        t = synth.mutate(p[:h], 0.3, rng.getrandbits(40)) + p[h:2 * h] + synth.mutate(p[2 * h:], 0.3, rng.getrandbits(40))
    return p, t


def run(rounds=20, items=240, seed=1, max_len=9000, threads=None, pens=False, dump="", quiet=False):
    """the campaign; returns {"problems", "differ", "failed", ...}"""
    threads = threads or min(64, os.cpu_count() or 1)
    pyoracle.lib()
    h = capi.Handle(0)
    pool = ThreadPoolExecutor(threads)
    total = bad = failed = 0
    t_gpu = t_cpu = 0.0
    dumped = []
    for rnd in range(rounds):
        rng = random.Random(seed * 100003 + rnd)
        pen = None
        if pens:
            pen = rng.choice([None, (4, 6, 2, 12, 1), (3, 4, 1, 10, 1), (5, 8, 2, 60, 1), (6, 10, 3, 124, 1), (9, 40, 2, 100, 1), (33, 20, 2, 24, 1), (2, 3, 1, 8, 1), (7, 11, 3, 30, 2),
                              (1, 2, 1, 6, 1), (5, 8, 2, 24, 1), (6, 8, 2, 49, 1)])
        batch, calls = [], []
        for k in range(items):
            p, t = gen_pair(rng, rng.randrange(0, 12), max_len)
            mode = rng.random()
            if mode < 0.7 or not p or not t or max(len(p), len(t)) > 4000:
                batch.append((p, t))
                calls.append(("bi", p, t))
            else:  # the patch forms of do_biwfa_alignment: free beginnings (head) or free ends (tail)
                args = (len(p), 0, len(t), 0) if mode < 0.85 else (0, len(p), 0, len(t))
                batch.append((p, t, capi.WFM_MODE_ENDSFREE) + args)
                calls.append(("ef", p, t) + args)
        t0 = time.time()
        res = h.align(batch, pen)
        t_gpu += time.time() - t0
        t0 = time.time()

        def ref(c):
            if c[0] == "bi":
                return pyoracle.align_biwfa(c[1], c[2], pen)
            return pyoracle.align_endsfree(c[1], c[3], c[4], c[2], c[5], c[6], pen=pen)
        exp = list(pool.map(ref, calls))
        t_cpu += time.time() - t0
        rb = rf = 0
        for c, r, (rc, ops, sc, _) in zip(calls, res, exp):
            total += 1
            if rc != 0:
                continue  # (the oracle itself gave up: not a parity statement)
            if r.status != 0:
                rf += 1
            elif r.ops != ops or (c[0] == "bi" and r.score != sc):
                rb += 1
            else:
                continue
            if len(dumped) < 40:
                dumped.append({"round": rnd, "pen": pen, "kind": c[0], "args": list(c[3:]), "p": c[1].decode(), "t": c[2].decode(), "status": int(r.status),
                               "gpu_score": int(r.score), "oracle_score": int(sc)})
        bad += rb; failed += rf
        if not quiet:
            print(f"round {rnd} (penalties {pen}): {len(batch)} problems, {rb} differ, {rf} failed (gpu {t_gpu:.1f} s, oracle {t_cpu:.1f} s so far)", flush=True)
    h.close()
    pool.shutdown()
    if dumped and dump:
        with open(dump, "w") as f:
            json.dump(dumped, f)
    return {"problems": total, "differ": bad, "failed": failed, "seed": seed, "rounds": rounds, "items": items, "max_len": max_len}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--items", type=int, default=240)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-len", type=int, default=9000)
    ap.add_argument("--threads", type=int, default=min(64, os.cpu_count() or 1))
    ap.add_argument("--dump", default="")
    ap.add_argument("--pens", action="store_true", help="a penalty set per round (mismatch, o1, e1, o2, e2) drawn from sets within the supported scope instead of the defaults")
    a = ap.parse_args()
    out = run(a.rounds, a.items, a.seed, a.max_len, a.threads, a.pens, a.dump)
    print(json.dumps(out))
    sys.exit(1 if out["differ"] or out["failed"] else 0)


if __name__ == "__main__":
    main()
