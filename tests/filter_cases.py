"""Deterministic synthetic inputs for the map post-processing tests: a .fai of PanSN-named
sequences and, per case, one query's raw L2 mappings (fragment hits along colinear chains with
gaps, an inverted chain, repeats on several haplotypes, noise, exact score ties) plus the
parameter overrides of that case."""
import os
import random

import numpy as np

NAMES = [("A#1#c1", 300000), ("A#2#c1", 280000), ("B#1#c1", 310000), ("B#1#c2", 50000), ("C#1#c1", 290000), ("solo", 120000)]

MAPPING_DTYPE = np.dtype([("refSeqId", "<u4"), ("refStartPos", "<u4"), ("queryStartPos", "<u4"), ("blockLength", "<u4"),
                          ("n_merged", "<u4"), ("conservedSketches", "<u4"), ("nucIdentity", "<u2"), ("flags", "u1"),
                          ("kmerComplexity", "u1")])

UINT32_MAX = 0xFFFFFFFF

# (name, query, seed, parameter overrides)
CASES = [
    ("defaults", "A#1#c1", 1, {}),
    ("defaults_other_query", "C#1#c1", 2, {}),
    ("short_query", "B#1#c2", 3, {}),
    ("ungrouped_query", "solo", 4, {}),
    ("n1", "A#1#c1", 5, {"num_mappings_for_segment": 1}),
    ("n3", "A#1#c1", 6, {"num_mappings_for_segment": 3}),
    ("n2_droprand", "A#1#c1", 7, {"num_mappings_for_segment": 2, "drop_rand": 1}),
    ("no_merge", "A#1#c1", 8, {"merge_mappings": 0}),
    ("no_split", "A#1#c1", 9, {"split": 0, "scaffold_gap": 0}),
    ("no_split_scaffold", "A#1#c1", 26, {"split": 0}),
    ("filter_none", "A#1#c1", 10, {"filter_mode": 3}),
    ("one_to_one_mode", "A#1#c1", 11, {"filter_mode": 2}),
    ("no_scaffold", "A#1#c1", 12, {"scaffold_gap": 0}),
    ("tight_scaffold", "A#1#c1", 13, {"scaffold_gap": 20000, "scaffold_min_length": 30000, "scaffold_max_deviation": 5000}),
    ("block_length", "A#1#c1", 14, {"block_length": 5000}),
    ("small_chain_gap", "A#1#c1", 15, {"chain_gap": 500}),
    ("big_chain_gap_short_max_len", "A#1#c1", 16, {"chain_gap": 20000, "max_mapping_length": 10000}),
    ("no_prefix_groups", "A#1#c1", 17, {"skip_prefix": 0, "prefix_delim": b"\0"}),
    ("overlap_half", "A#1#c1", 18, {"overlap_threshold": 0.5, "num_mappings_for_segment": 4}),
    ("overlap_all", "A#1#c1", 19, {"overlap_threshold": 1.0, "num_mappings_for_segment": 2}),
    ("sparsify", "A#1#c1", 20, {"sparsity_hash_threshold": 1 << 63}),
    ("legacy_output", "A#1#c1", 21, {"legacy_output": 1}),
    ("scaffold_r2", "A#1#c1", 22, {"num_mappings_for_scaffold": 2, "scaffold_overlap_threshold": 0.9}),
    ("w500", "A#1#c1", 23, {"window_length": 500}),
    ("single_mapping", "A#1#c1", 24, {"scaffold_gap": 0}),
    ("empty", "A#1#c1", 25, {}),
]


def write_fai(dirname):
    """An (empty) FASTA plus its .fai: the reference's SequenceIdManager only reads the .fai."""
    fa = os.path.join(dirname, "pan.fa")
    open(fa, "w").close()
    with open(fa + ".fai", "w") as f:
        for n, l in NAMES:
            f.write(f"{n}\t{l}\t0\t60\t61\n")
    return fa


def make_mappings(name, query, seed, over):
    rng = random.Random(seed)
    w = over.get("window_length", 1000)
    qlen = dict(NAMES)[query]
    qid = [n for n, _ in NAMES].index(query)
    nfrag = qlen // w
    rows = []
    if name == "empty":
        return np.zeros(0, dtype=MAPPING_DTYPE)
    if name == "single_mapping":
        return np.array([(2, 1234, 5000, w, 1, 12, 9312, 0, 88)], dtype=MAPPING_DTYPE)
    for t, (tname, tlen) in enumerate(NAMES):
        if t == qid:
            continue
        off = rng.randrange(0, 4000)
        rev = rng.random() < 0.3
        ident0 = rng.randrange(8600, 9900)
        i = 0
        while i < nfrag:
            if rng.random() < 0.04:      # a gap in the chain (SV / unaligned stretch)
                i += rng.randrange(2, 40)
                off += rng.randrange(-3000, 30000)
                continue
            if rng.random() < 0.12:      # missed fragment
                i += 1
                continue
            q = i * w
            r = off + (q if not rev else (tlen - q - w)) + rng.randrange(-40, 40)
            if 0 <= r < tlen - w:
                ident = min(10000, max(7000, ident0 + rng.randrange(-300, 300)))
                rows.append((t, r, q, w, 1, rng.randrange(6, 30), ident, 1 if rev else 0, rng.randrange(70, 100)))
            i += 1
        # the final fragment is anchored at the query end but reported at nfrag*w (computeMap.hpp:124-128, :600-631)
        if qlen % w and rng.random() < 0.8:
            r = off + ((qlen - w) if not rev else 0)
            if 0 <= r < tlen - w:
                rows.append((t, r, nfrag * w, w, 1, 15, ident0, 1 if rev else 0, 90))
        # off-diagonal noise and a short repeat family
        for _ in range(rng.randrange(5, 40)):
            rows.append((t, rng.randrange(0, max(1, tlen - w)), rng.randrange(0, nfrag) * w, w, 1, rng.randrange(3, 10),
                         rng.randrange(7000, 9200), rng.randrange(2), rng.randrange(50, 100)))
    # exact ties: the same fragments hitting two haplotypes with identical identity
    for i in range(0, min(nfrag, 40), 3):
        for t in (1, 2):
            if t != qid:
                rows.append((t, 200000 + i * w, i * w, w, 1, 20, 9500, 0, 95))
    rng.shuffle(rows)
    return np.array(rows, dtype=MAPPING_DTYPE)
