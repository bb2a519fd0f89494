#!/usr/bin/env python
"""Generates tests/golden/map_golden.json from the REFERENCE'S OWN code
(oracle/_ref/libref_map.so = /root/reference/src/map/include/commonFunc.hpp compiled in
place by oracle/Makefile).  Run in the authoring container only:
    python tests/golden/make_map_golden.py
The fixture holds inputs and the reference's outputs (data, no reference source)."""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pymap  # noqa: E402
from wfmash_amd import synth  # noqa: E402

assert pymap.have_ref(), "build oracle/_ref first (make -C oracle)"


def lpa_slice(n=3000, off=100000):
    """A slice of the reference's LPA test data when available (real sequence, incl. soft-masked lower case)."""
    p = "/root/reference/data/LPA.subset.fa.gz"
    if not os.path.exists(p):
        return None
    seq = []
    for line in gzip.open(p, "rt"):
        if line.startswith(">"):
            if seq:
                break
            continue
        seq.append(line.strip())
    s = "".join(seq)
    return s[off:off + n].encode()


cases = []
seqs = {"rand1k": synth.random_dna(11, 1000), "rand5k": synth.random_dna(12, 5000)}
s = bytearray(synth.random_dna(13, 1200))
s[300:320] = b"N" * 20
s[700] = ord("n")
s[900:960] = bytes(s[900:960]).lower()
seqs["with_N_lower"] = bytes(s)
seqs["tandem"] = (b"ACGTTGCA" * 200)[:1000]
seqs["homopolymer_mix"] = b"A" * 300 + synth.random_dna(14, 400) + b"T" * 300
seqs["short"] = synth.random_dna(15, 40)
l = lpa_slice()
if l:
    seqs["lpa_slice"] = l

out = {"kmer_hashes": [], "sketches": [], "minmers": []}
for kmer in [b"ACGTACGTACGTACG", b"AAAAAAAAAAAAAAA", b"ACGTACGTACGTACGTACGTA", b"GATTACAGATTACAGA", b"ACGTNACGTACGTAC"]:
    out["kmer_hashes"].append({"kmer": kmer.decode(), "hash": str(pymap.get_hash(kmer, "ref"))})
for name, sq in seqs.items():
    for k, sk in [(15, 39), (15, 78), (21, 25), (17, 5)]:
        if len(sq) < k:
            continue
        m = pymap.sketch_sequence(sq, k, sk, 7, "ref")
        out["sketches"].append({"seq": name, "k": k, "s": sk,
                                "minmers": [[str(int(x["hash"])), int(x["wpos"]), int(x["wpos_end"]), int(x["strand"])] for x in m]})
    for k, w, sk in [(15, 100, 5), (15, 256, 12), (19, 64, 3)]:
        if len(sq) < w:
            continue
        m = pymap.ref_add_minmers(sq, k, w, sk, 3)
        out["minmers"].append({"seq": name, "k": k, "w": w, "s": sk,
                               "minmers": [[str(int(x["hash"])), int(x["wpos"]), int(x["wpos_end"]), int(x["strand"])] for x in m]})
out["seqs"] = {k: v.decode() for k, v in seqs.items()}
with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "map_golden.json.gz"), "wt") as f:
    json.dump(out, f)
print("sketch cases", len(out["sketches"]), "minmer cases", len(out["minmers"]))
