"""TEST INFRASTRUCTURE ONLY -- the whole map phase (skch::Map::mapQuery, computeMap.hpp:329-872)
strung together from the stage oracles, for small inputs:
  target minmers   the reference's own addMinmers (oracle/_ref/libref_map.so) -- caller-supplied otherwise
  index            oracle/map_index.py
  query sketches   liboracle_map.so (sketchSequence restatement, pinned by the reference goldens)
  L1 / L2          oracle/map_l1.py, oracle/map_l2.py (PARITY UNPINNED restatements)
  post-processing  the reference's own filter code (oracle/_ref/libref_filter.so) -- caller-supplied otherwise
Returns the mapping PAF text in query order."""
import numpy as np

from oracle import map_index as MI
from oracle import map_l1 as L1
from oracle import map_l2 as L2
from oracle import map_stats as MS
from oracle import pymap

MAPPING_DTYPE = np.dtype([("refSeqId", "<u4"), ("refStartPos", "<u4"), ("queryStartPos", "<u4"), ("blockLength", "<u4"),
                          ("n_merged", "<u4"), ("conservedSketches", "<u4"), ("nucIdentity", "<u2"), ("flags", "u1"),
                          ("kmerComplexity", "u1")])


def ref_groups(names, delim="#"):
    """SequenceIdManager::buildRefGroups (sequenceIds.hpp:286-338) without user prefixes."""
    group, n = {}, 0
    out = [0] * len(names)
    for name, idx in sorted((nm, i) for i, nm in enumerate(names)):
        pos = name.rfind(delim) if delim else -1
        key = name[:pos] if pos >= 0 else name
        if key not in group:
            n += 1
            group[key] = n
        out[idx] = group[key]
    return out


def sketch_size(pct_identity, w, k):
    md = 1 - float(np.float32(pct_identity))
    dens = 0.02 * (1 + (md / 0.1))
    return int(dens * (w - k))


def fragments(seq_len, w):
    n = seq_len // w
    offs = [i * w for i in range(n)]
    if n >= 1 and seq_len % w:
        offs.append(seq_len - w)
    return offs


def map_queries(seqs, pct_identity, k=15, w=1000, s=None, minimum_hits=3, max_kmer_freq=0.0002, add_minmers=None,
                skip_self=True, skip_prefix=True, lower_triangular=False, kc_threshold=0.0, queries=None):
    """seqs: list of (name, bytes) in file order (all-vs-all).  Returns {query index: MAPPING_DTYPE array} of raw L2
    mappings (after processFragment's query offset, before the boundary check), and the group table."""
    names = [n for n, _ in seqs]
    group = ref_groups(names) if skip_prefix else ref_groups(names, "")
    S = s or sketch_size(pct_identity, w, k)
    pi = float(np.float32(pct_identity))
    if add_minmers is None:
        assert pymap.have_ref(), "no reference build here: pass add_minmers"
        add_minmers = lambda sq, sid: pymap.ref_add_minmers(sq, k, w, S, sid)
    mm = []
    for sid, (_, sq) in enumerate(seqs):
        if len(sq) >= w:
            mm += [(int(x["hash"]), int(x["wpos"]), int(x["wpos_end"]), int(x["seqId"]), int(x["strand"])) for x in add_minmers(sq, sid)]
    lookup, index, _ = MI.build_index(mm, max_kmer_freq)
    mh = [0] + [max(minimum_hits, MS.estimate_minimum_hits_relaxed(q, k, pi, 0.95)) for q in range(1, S + 1)]
    p1 = dict(window_length=w, sketch_size=S, min_hits_cached=mh[S], cached_segment_length=w, min_hits_by_qsketch=mh,
              sketch_cutoffs=MS.sketch_cutoffs(S, k, 0.0, 0.999), skip_self=skip_self, skip_prefix=skip_prefix,
              lower_triangular=lower_triangular, stage1_topani=True, stage2_full_scan=True)
    keep, ident = L2.identity_tables(S, k, pi)
    p2 = dict(window_length=w, sketch_size=S, stage1_topani=True, keep_table=keep, ident_table=ident,
              cutoff_j=[0.0] + [L2.cutoff_j(q, k) for q in range(1, S + 1)], skip_prefix=skip_prefix)
    out = {}
    for qid, (_, sq) in enumerate(seqs):
        if queries is not None and qid not in queries:  # (a C4-sized test maps one query haplotype against the whole index)
            continue
        rows = []
        for fi, off in enumerate(fragments(len(sq), w)):
            sk = pymap.sketch_sequence(sq[off:off + w], k, S)
            if len(sk) == 0:
                continue
            kc, kc_u8 = L2.kmer_complexity(int(sk["hash"][-1]), len(sk), w, k)
            if kc < np.float32(kc_threshold):
                continue
            hashes = [int(x) for x in sk["hash"]]
            cands = L1.do_l1_mapping(hashes, w, qid, lookup, group, p1)
            if not cands:
                continue
            cands = [dict(seqId=c["seqId"], start=c["start"], end=c["end"], isect=c["isect"]) for c in cands]
            qm = [(int(x["hash"]), int(x["strand"])) for x in sk]
            for r in L2.do_l2_mapping(qm, w, kc_u8, cands, index, group, p2):
                r = list(r)
                r[2] = (r[2] + fi * w) & 0xFFFFFFFF
                rows.append(tuple(r))
        out[qid] = np.array(rows, dtype=MAPPING_DTYPE) if rows else np.zeros(0, dtype=MAPPING_DTYPE)
    return out, group, S
