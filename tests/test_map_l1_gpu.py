"""GPU parity test of the L1 stage (SURVEY 8a m6+m7): wfm_map_l1 against the Python restatement
of getSeedIntervalPoints + computeL1CandidateRegions + doL1Mapping's group loop, on a synthetic
PanSN-style pangenome (groups of haplotypes, one reverse-complemented, repeats, an unrelated
sequence) mapped all-vs-all."""
import numpy as np
import pytest

from oracle import map_index as MI
from oracle import map_l1 as L1
from oracle import map_stats as MS
from oracle import wflign_host as W
from wfmash_amd import synth

pytestmark = pytest.mark.gpu

K = 15


def _pangenome(seed, L=24000):
    base = synth.random_dna(seed, L)
    unit = synth.random_dna(seed + 1, 700)
    base = base[:L // 2] + unit + base[L // 2:L // 2 + 3000] + unit + base[L // 2 + 3000:]  # a 2-copy repeat
    seqs, group = [], []
    for g in range(3):
        for hap in range(2):
            s = synth.mutate(base, 0.01 + 0.02 * g, seed * 100 + g * 10 + hap)
            if g == 2 and hap == 1:
                s = W.revcomp(s)
            seqs.append(s)
            group.append(g)
    seqs.append(synth.random_dna(seed + 7, 9000))  # unrelated
    group.append(3)
    return seqs, group


def _fragments(seq_len, w):
    """computeMap.hpp:560-631: non-overlapping windowLength fragments + one anchored at the end."""
    n = seq_len // w
    offs = [i * w for i in range(n)]
    if seq_len % w and seq_len >= w:
        offs.append(seq_len - w)
    return offs


def _params(w, s, ident, **over):
    mh = [max(3, MS.estimate_minimum_hits_relaxed(q, K, ident, 0.95)) if q else 0 for q in range(s + 1)]
    p = dict(window_length=w, sketch_size=s, min_hits_cached=mh[s], cached_segment_length=w, min_hits_by_qsketch=mh,
             sketch_cutoffs=MS.sketch_cutoffs(s, K, 0.0, 0.999), skip_self=True, skip_prefix=True, lower_triangular=False,
             stage1_topani=True, stage2_full_scan=True)
    p.update(over)
    return p


def _run(gpu, seqs, group, w, s, params, max_freq=0.0002):
    mm = np.concatenate([gpu.add_minmers(sq, K, w, s, sid) for sid, sq in enumerate(seqs)])
    ix = gpu.index_build(mm, max_freq)
    lookup, _, _ = MI.build_index([(int(x["hash"]), int(x["wpos"]), int(x["wpos_end"]), int(x["seqId"]), int(x["strand"])) for x in mm], max_freq)
    sk, qseq, qlen = [], [], []
    for sid, sq in enumerate(seqs):
        offs = _fragments(len(sq), w)
        sk += gpu.sketch_fragments(sq, offs, [w] * len(offs), K, s, sid)
        qseq += [sid] * len(offs)
        qlen += [w] * len(offs)
    nfrag = len(sk)
    flat = np.zeros(nfrag * s, dtype=sk[0].dtype)
    for f, m in enumerate(sk):
        flat[f * s:f * s + len(m)] = m
    qcount = [len(m) for m in sk]
    active = [1] * nfrag
    active[3] = 0  # one fragment below the kmer-complexity threshold
    got = gpu.map_l1(ix, flat, qcount, qseq, qlen, active, s, params, group)
    exp = []
    for f in range(nfrag):
        if not active[f]:
            continue
        hashes = [int(x) for x in sk[f]["hash"]]
        for c in L1.do_l1_mapping(hashes, qlen[f], qseq[f], lookup, group, params):
            exp.append((c["seqId"], f, c["start"], c["end"], c["isect"]))
    got_t = [(int(c["seqId"]), int(c["frag"]), int(c["rangeStartPos"]), int(c["rangeEndPos"]), int(c["intersectionSize"])) for c in got]
    ix.free()
    assert got_t == exp, (len(got_t), len(exp), [a for a, b in zip(got_t, exp) if a != b][:3])
    return got_t, nfrag


def test_l1_all_vs_all_default_filters(gpu):
    seqs, group = _pangenome(11)
    got, nfrag = _run(gpu, seqs, group, 1000, 25, _params(1000, 25, 0.85))
    assert len(got) > nfrag  # every related fragment finds the other groups' haplotypes


def test_l1_filter_switches(gpu):
    seqs, group = _pangenome(13, L=15000)
    base, _ = _run(gpu, seqs, group, 500, 16, _params(500, 16, 0.9))
    no_top, _ = _run(gpu, seqs, group, 500, 16, _params(500, 16, 0.9, stage1_topani=False))
    assert len(no_top) >= len(base)
    _run(gpu, seqs, group, 500, 16, _params(500, 16, 0.9, stage2_full_scan=False))
    # no group filter at all: self hits appear, one single group run
    allv, _ = _run(gpu, seqs, group, 500, 16, _params(500, 16, 0.9, skip_self=False, skip_prefix=False))
    assert (0, 0, 0, 0, 16) in allv and (0, 0, 0, 0, 16) not in base  # the fragment's own origin
    lt, _ = _run(gpu, seqs, group, 500, 16, _params(500, 16, 0.9, skip_self=False, skip_prefix=False, lower_triangular=True))
    assert lt and all(c[0] != 6 for c in lt)  # the last sequence is never a target under -L


def test_l1_rejects_ragged_fragment_lengths(gpu):
    seqs, group = _pangenome(17, L=6000)
    mm = np.concatenate([gpu.add_minmers(sq, K, 500, 10, sid) for sid, sq in enumerate(seqs)])
    ix = gpu.index_build(mm)
    sk = gpu.sketch_fragments(seqs[0], [0], [700], K, 10, 0)
    flat = np.zeros(10, dtype=sk[0].dtype)
    flat[:len(sk[0])] = sk[0]
    from wfmash_amd.capi import WfmError
    with pytest.raises(WfmError):
        gpu.map_l1(ix, flat, [len(sk[0])], [0], [700], [1], 10, _params(500, 10, 0.9), group)
    ix.free()


def test_l1_repeat_rich_lists_span_many_chunks(gpu, monkeypatch):
    """Tandem repeats put hundreds to thousands of interval points into a fragment's list: position groups and candidates
    then straddle the 64-key chunks of the wave-per-fragment sweep.  Both forms of the sweep against the oracle."""
    rng_seed = 23
    unit = synth.random_dna(rng_seed, 180)
    flank = synth.random_dna(rng_seed + 1, 9000)
    base = flank[:3000] + unit * 30 + flank[3000:6000] + unit * 12 + flank[6000:]
    seqs, group = [], []
    for g in range(4):
        for hap in range(2):
            seqs.append(synth.mutate(base, 0.004 * (g + 1), 1000 + g * 10 + hap))
            group.append(g)
    counts = []
    for wave in ("1", "0"):
        monkeypatch.setenv("WFM_L1_WAVE", wave)
        for kw in (dict(), dict(stage2_full_scan=False), dict(stage1_topani=False, skip_prefix=False, skip_self=False)):
            got, nfrag = _run(gpu, seqs, group, 500, 20, _params(500, 20, 0.9, **kw), max_freq=0.05)
            counts.append(len(got))
    assert counts[:3] == counts[3:] and min(counts) > 0
