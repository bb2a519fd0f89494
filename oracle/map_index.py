"""TEST INFRASTRUCTURE ONLY -- restatement of Sketch::build's index stage
(src/map/include/winSketch.hpp:266-429) for a single thread (-t 1: sequences in file order):
frequency filter, hash -> [IntervalPoint] with contiguous intervals fused, and minmerIndex."""
from collections import OrderedDict

OPEN, CLOSE = 1, -1


def build_index(minmers, max_kmer_freq=0.0002):
    """minmers: list of (hash, wpos, wpos_end, seqId, strand) in (seqId, wpos) order.
    Returns (pos_lookup: {hash: [(pos, hash, seqId, side)]}, minmer_index, info)."""
    freqs = {}
    for m in minmers:
        freqs[m[0]] = freqs.get(m[0], 0) + 1
    total = len(minmers)
    min_occ = 10
    if max_kmer_freq <= 1.0:
        thr = max(min_occ, int(total * max_kmer_freq))
    else:
        thr = max(min_occ, int(max_kmer_freq))
    would_pos = sum(f for f in freqs.values() if f > thr and f > min_occ)
    would_unique = sum(1 for f in freqs.values() if f > thr and f > min_occ)
    adjusted = False
    if would_pos > total // 2 or would_unique > len(freqs) * 0.7:
        allf = sorted(freqs.values())
        keep_index = int(len(allf) * 0.999)
        if keep_index >= len(allf):
            keep_index = len(allf) - 1
        thr = max(thr, allf[keep_index])
        adjusted = True
    lookup = OrderedDict()
    index = []
    filtered = 0
    for m in minmers:
        h, wpos, wend, sid, strand = m
        f = freqs[h]
        if f > thr and f > min_occ:
            filtered += 1
            continue
        pl = lookup.setdefault(h, [])
        if not pl or pl[-1][1] != h or pl[-1][0] != wpos:
            pl.append([wpos, h, sid, OPEN])
            pl.append([wend, h, sid, CLOSE])
        else:
            pl[-1][0] = wend
        index.append(m)
    info = dict(n_windows=total, n_kept=len(index), n_unique=len(lookup), threshold=thr, filtered=filtered, adjusted=adjusted)
    return lookup, index, info
