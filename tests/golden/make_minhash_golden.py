#!/usr/bin/env python
"""Generates tests/golden/minhash_golden.json.gz from the REFERENCE'S OWN StreamingMinHash /
GroupedStreamingMinHash::processSequence (streamingMinHash.hpp, compiled in place into oracle/_ref/libref_map.so by
oracle/Makefile; the k-mer loop is the one Stat::estimate_identity_for_groups runs, map_stats.hpp:569-616).
Run in the authoring container only:   python tests/golden/make_minhash_golden.py
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

base = synth.random_dna(77, 30000)
seqs = {
    "plain": base[:12000],
    "lower_and_iupac": base[:4000].lower() + b"RYKM" + base[4000:8000] + b"N" * 300 + base[8000:11000],
    "ambiguous_head": base[:7] + b"N" + base[8:6000],   # arms the counter: k-mers 0..k-1 are blanked
    "ambiguous_second": base[:25] + b"n" + base[26:5000],
    "repeat": synth.random_dna(5, 300) * 40,             # duplicates fill the sketch
    "short": base[:400],
    "tiny": base[:20],
    "all_n": b"N" * 500,
    "palindromes": b"ACGT" * 100 + base[:2000],
}
out = {"seqs": {k: v.decode() for k, v in seqs.items()}, "sketches": [], "pools": [], "streams": []}
for name, sq in seqs.items():
    for k, ss in ((21, 4096), (21, 128), (15, 64), (16, 50)):
        v = pymap.ref_group_minhash([sq], [0], k, ss, 0)
        out["sketches"].append({"seq": name, "k": k, "sketch_size": ss, "hashes": [format(int(x), "x") for x in v]})
# pooled group sketches (merge of per-sequence sketches, duplicates across sequences kept)
for members, k, ss in ((["plain", "repeat", "short"], 21, 512), (["lower_and_iupac", "plain"], 21, 4096), (["tiny", "all_n"], 21, 64)):
    v = pymap.ref_group_minhash([seqs[m] for m in members], [3] * len(members), k, ss, 3)
    out["pools"].append({"members": members, "k": k, "sketch_size": ss, "hashes": [format(int(x), "x") for x in v]})
# the heap alone on small integer streams (ties at the maximum, fewer values than the sketch holds)
for vals, ss in (([5, 3, 9, 3, 3, 7, 1, 9, 9, 2], 4), ([4, 4, 4, 4], 3), ([8, 6], 5), ([], 3), ([2, 9, 2, 9, 2, 9, 1], 6)):
    v = pymap.ref_streaming_minhash(vals, ss)
    out["streams"].append({"values": vals, "sketch_size": ss, "sketch": [int(x) for x in v]})
with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "minhash_golden.json.gz"), "wt") as f:
    json.dump(out, f)
print("sketches", len(out["sketches"]), "pools", len(out["pools"]), "streams", len(out["streams"]))
