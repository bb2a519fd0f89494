"""TEST INFRASTRUCTURE ONLY -- restatement of the automatic identity estimate:
  Stat::estimate_identity_for_groups   src/map/include/map_stats.hpp:325-822
  StreamingMinHash                     src/map/include/streamingMinHash.hpp:35-135
PARITY UNPINNED: map_stats.hpp needs GSL and htslib, neither is in the image, so the reference
cannot produce a vector for this function here.  K-mer hashes come from liboracle_map.so
(getHash restatement, pinned by the reference goldens)."""
import numpy as np

from oracle import map_stats as MS
from oracle import pymap

K = 21
SKETCH = 4096


def minhash_sketch(seq: bytes, k=K, sketch_size=SKETCH):
    """bottom-sketch_size canonical hashes with multiplicity of the k-mers the reference's loop accepts
    (map_stats.hpp:569-616): no non-ACGT base inside, strands hash differently; an ambiguous base
    among the first k bases arms the skip counter with k and so blanks k-mers 0..k-1."""
    if len(seq) < k:
        return np.zeros(0, dtype=np.uint64)
    h, st = pymap.hash_kmers(seq, k)
    ok = st != 0
    head = seq[:k].upper()
    if any(c not in b"ACGT" for c in head):
        ok[:k] = False
    v = np.sort(h[ok])
    return v[:sketch_size]


def pool(group, add, sketch_size=SKETCH):
    return np.sort(np.concatenate([group, add]))[:sketch_size]


def estimate_identity(seqs, groups, percentile=50, adjustment=-2.0, k=K, sketch_size=SKETCH):
    """seqs: list of bytes (all-vs-all: every sequence is query and target); groups: group id per sequence."""
    g_sk = {}
    for sq, g in zip(seqs, groups):
        if len(sq) == 0:
            continue
        g_sk[g] = pool(g_sk.get(g, np.zeros(0, dtype=np.uint64)), minhash_sketch(sq, k, sketch_size), sketch_size)
    anis = []
    for qg in sorted(g_sk):
        for tg in sorted(g_sk):
            if qg == tg or len(g_sk[qg]) == 0 or len(g_sk[tg]) == 0:
                continue
            a, b = g_sk[qg], g_sk[tg]
            i = j = shared = 0
            while i < len(a) and j < len(b):
                if a[i] == b[j]:
                    shared += 1; i += 1; j += 1
                elif a[i] < b[j]:
                    i += 1
                else:
                    j += 1
            if shared == 0:
                continue
            jac = shared / min(len(a), len(b))
            anis.append(1.0 - float(MS.j2md(np.float32(jac), k)))
    if not anis:
        return 0.70
    anis.sort()
    idx = min((percentile * len(anis)) // 100, len(anis) - 1)
    adj = anis[idx] + float(np.float32(adjustment)) / 100.0
    return min(1.0, max(0.0, adj))
