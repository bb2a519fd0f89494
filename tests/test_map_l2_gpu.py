"""GPU parity test of the L2 stage: wfm_map_l2 against the Python restatement of
doL2Mapping + computeL2MappedRegions + SlideMapper, fed with the L1 candidates of wfm_map_l1."""
import numpy as np
import pytest

from oracle import map_l2 as L2
from tests.test_map_l1_gpu import K, _fragments, _pangenome, _params

pytestmark = pytest.mark.gpu


def _run(gpu, seqs, group, w, s, ident, **over):
    p1 = _params(w, s, ident, **over)
    mm = np.concatenate([gpu.add_minmers(sq, K, w, s, sid) for sid, sq in enumerate(seqs)])
    ix = gpu.index_build(mm)
    _, _, _, kept = ix.download()
    index = [(int(x["hash"]), int(x["wpos"]), int(x["wpos_end"]), int(x["seqId"]), int(x["strand"])) for x in kept]
    sk, qseq = [], []
    for sid, sq in enumerate(seqs):
        offs = _fragments(len(sq), w)
        sk += gpu.sketch_fragments(sq, offs, [w] * len(offs), K, s, sid)
        qseq += [sid] * len(offs)
    nfrag = len(sk)
    flat = np.zeros(nfrag * s, dtype=sk[0].dtype)
    for f, m in enumerate(sk):
        flat[f * s:f * s + len(m)] = m
    qcount = [len(m) for m in sk]
    qlen = [w] * nfrag
    kc = [L2.kmer_complexity(int(m["hash"][-1]), len(m), w, K)[1] if len(m) else 0 for m in sk]
    cands = gpu.map_l1(ix, flat, qcount, qseq, qlen, [1] * nfrag, s, p1, group)
    keep, idt = L2.identity_tables(s, K, ident)
    p2 = dict(window_length=w, sketch_size=s, stage1_topani=p1["stage1_topani"], keep_table=keep, ident_table=idt,
              cutoff_j=[0.0] + [L2.cutoff_j(q, K) for q in range(1, s + 1)], skip_prefix=p1["skip_prefix"])
    got, gfrag = gpu.map_l2(ix, flat, qcount, qlen, kc, s, cands, p2)
    names = got.dtype.names
    got_t = [(int(f),) + tuple(int(r[n]) for n in names) for r, f in zip(got, gfrag)]
    exp = []
    by_frag = {}
    for c in cands:
        by_frag.setdefault(int(c["frag"]), []).append(dict(seqId=int(c["seqId"]), start=int(c["rangeStartPos"]), end=int(c["rangeEndPos"]),
                                                            isect=int(c["intersectionSize"])))
    for f in sorted(by_frag):
        qm = [(int(x["hash"]), int(x["strand"])) for x in sk[f]]
        for r in L2.do_l2_mapping(qm, w, kc[f], by_frag[f], index, group, p2):
            exp.append((f,) + r)
    ix.free()
    assert len(got_t) == len(exp), (len(got_t), len(exp))
    diff = [(a, b) for a, b in zip(got_t, exp) if a != b]
    assert not diff, diff[:3]
    return got_t, cands


def test_l2_all_vs_all(gpu):
    seqs, group = _pangenome(11)
    got, cands = _run(gpu, seqs, group, 1000, 25, 0.85)
    assert len(got) > 100
    assert any(r[8] & 1 for r in got) and any(not (r[8] & 1) for r in got)  # both strands (one haplotype is reverse-complemented)
    # mapped positions are right: forward-strand hits of fragment i of a haplotype land near i*w on the others (small indel drift)
    assert max(r[7] for r in got) <= 10000 and min(r[7] for r in got) >= 7000  # identities between 70 and 100 %


def test_l2_high_identity_threshold_and_no_prefix_filter(gpu):
    seqs, group = _pangenome(13, L=15000)
    strict, _ = _run(gpu, seqs, group, 500, 16, 0.95)
    loose, _ = _run(gpu, seqs, group, 500, 16, 0.80)
    assert 0 < len(strict) < len(loose)
    _run(gpu, seqs, group, 500, 16, 0.9, skip_self=False, skip_prefix=False)
    _run(gpu, seqs, group, 500, 16, 0.9, stage1_topani=False)


def test_fused_map_fragments_equals_staged_path(gpu):
    """wfm_map_fragments (device-resident sketch -> L1 -> L2) against the staged calls, which the
    tests above pin against the restatement."""
    seqs, group = _pangenome(19, L=20000)
    w, s, ident = 1000, 25, 0.85
    p1 = _params(w, s, ident)
    mm = np.concatenate([gpu.add_minmers(sq, K, w, s, sid) for sid, sq in enumerate(seqs)])
    ix = gpu.index_build(mm)
    keep, idt = L2.identity_tables(s, K, ident)
    p2 = dict(window_length=w, sketch_size=s, stage1_topani=True, keep_table=keep, ident_table=idt,
              cutoff_j=[0.0] + [L2.cutoff_j(q, K) for q in range(1, s + 1)])
    buf = b"".join(seqs)
    base = np.cumsum([0] + [len(x) for x in seqs])
    off, sid_of, sk = [], [], []
    for sid, sq in enumerate(seqs):
        o = _fragments(len(sq), w)
        off += [int(base[sid]) + x for x in o]
        sid_of += [sid] * len(o)
        sk += gpu.sketch_fragments(sq, o, [w] * len(o), K, s, sid)
    nfrag = len(off)
    flat = np.zeros(nfrag * s, dtype=sk[0].dtype)
    for f, m in enumerate(sk):
        flat[f * s:f * s + len(m)] = m
    qcount = [len(m) for m in sk]
    kc = [L2.kmer_complexity(int(m["hash"][-1]), len(m), w, K)[1] for m in sk]
    cands = gpu.map_l1(ix, flat, qcount, sid_of, [w] * nfrag, [1] * nfrag, s, p1, group)
    exp, efrag = gpu.map_l2(ix, flat, qcount, [w] * nfrag, kc, s, cands, p2)
    got, gfrag = gpu.map_fragments(ix, buf, off, sid_of, K, p1, p2, group)
    assert len(got) == len(exp) > 100
    assert (gfrag == efrag).all() and got.tobytes() == exp.tobytes()
    # a complexity threshold above every fragment's complexity switches the whole batch off (computeMap.hpp:951)
    none, _ = gpu.map_fragments(ix, buf, off, sid_of, K, p1, p2, group, kc_threshold=10.0)
    assert len(none) == 0
    ix.free()
