"""CPU tests: the WFA oracle (oracle/wfa2p.c) against an independent O(nm) DP.

The reference holds no golden vector at the WFA2-lib boundary (SURVEY.md 8c:
"parity unpinned"), so the oracle is pinned by (i) optimal score == 5-state DP,
(ii) pafcheck-style CIGAR validity, (iii) CIGAR-implied score == optimal score.
"""
import random

import pytest

from wfmash_amd import synth


def _rand_pairs(seed, n, lens, rates):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        L = rng.choice(lens)
        p = synth.random_dna(seed * 1000 + i, L)
        t = synth.mutate(p, rng.choice(rates), seed * 7919 + i) if L else b""
        if rng.random() < 0.15:
            t = synth.random_dna(seed * 31 + i, rng.randrange(0, 200))
        out.append((p, t))
    return out


def test_uni_and_biwfa_scores_match_dp(oracle):
    for p, t in _rand_pairs(11, 120, [0, 1, 2, 7, 40, 100, 101, 160, 420, 900], [0.0, 0.02, 0.1, 0.3]):
        dp = oracle.dp_score(p, t)
        rc, ops, sc, _ = oracle.align_uni(p, t)
        assert rc == 0 and sc == dp
        assert oracle.ops_check(ops, p, t) == 0
        assert oracle.ops_score(ops) == dp
        rc, ops2, sc2, _ = oracle.align_biwfa(p, t)
        assert rc == 0 and sc2 == dp
        assert oracle.ops_check(ops2, p, t) == 0


def test_biwfa_recursion_is_exercised(oracle):
    p = synth.random_dna(5, 6000)
    t = synth.mutate(p, 0.1, 55)
    rc, ops, sc, st = oracle.align_biwfa(p, t)
    assert rc == 0
    assert st.bialign_calls >= 3 and st.base_calls >= 4 and st.max_depth >= 2
    assert sc == oracle.dp_score(p, t)
    assert oracle.ops_check(ops, p, t) == 0


def test_endsfree_scores_match_dp(oracle):
    rng = random.Random(3)
    for i in range(80):
        p = synth.random_dna(300 + i, rng.choice([1, 20, 150, 400]))
        t = synth.mutate(p, rng.choice([0.0, 0.05, 0.2]), 900 + i)
        if rng.random() < 0.4:
            t = synth.random_dna(77 + i, rng.randrange(1, 40)) + t
        if rng.random() < 0.4:
            p = synth.random_dna(99 + i, rng.randrange(1, 40)) + p
        if not t:
            continue
        for args in ((len(p), 0, len(t), 0), (0, len(p), 0, len(t))):  # head / tail patch forms (wflign.cpp:300-305, 392-397)
            dp = oracle.dp_score_endsfree(p, t, *args)
            rc, ops, sc, _ = oracle.align_endsfree(p, args[0], args[1], t, args[2], args[3])
            assert rc == 0 and sc == dp
            assert oracle.ops_check(ops, p, t) == 0


def test_component_halves_concatenate(oracle):
    """A BiWFA breakpoint splits the problem into two halves whose component-
    constrained alignments concatenate to an optimal alignment."""
    p = synth.random_dna(8, 3000)
    t = synth.mutate(p, 0.08, 88)
    rc, bp, _ = oracle.find_breakpoint(p, t)
    assert rc == 0
    dp = oracle.dp_score(p, t)
    assert bp.score == dp
    h, v = bp.offset_forward, bp.offset_forward - bp.k_forward
    rc0, ops0, _, _ = oracle.align_comp(p[:v], t[:h], 0, bp.component)
    rc1, ops1, _, _ = oracle.align_comp(p[v:], t[h:], bp.component, 0)
    assert rc0 == 0 and rc1 == 0
    ops = ops0 + ops1
    assert oracle.ops_check(ops, p, t) == 0
    assert oracle.ops_score(ops) == dp


def test_custom_penalties(oracle):
    pen = (4, 6, 2, 12, 1)
    for p, t in _rand_pairs(21, 30, [50, 300, 700], [0.05, 0.2]):
        rc, ops, sc, _ = oracle.align_biwfa(p, t, pen)
        assert rc == 0 and sc == oracle.dp_score(p, t, pen)


def test_legacy_reads_fixture_is_plausible(oracle):
    """test/data/regression/reads.255bps.paf predates the current writer (its
    CIGARs start/end with indels); it cannot be a byte-exact golden, but the
    alignment score of our oracle must not be worse than the legacy CIGAR's."""
    import gzip
    import os
    here = os.path.dirname(__file__)
    fa = os.path.join(here, "golden", "reads.255bps.fa.gz")
    paf = os.path.join(here, "golden", "reads.255bps.paf")
    if not (os.path.exists(fa) and os.path.exists(paf)):
        pytest.skip("fixture not present")
    seqs = {}
    name = None
    for line in gzip.open(fa, "rt"):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = ""
        elif name:
            seqs[name] += line.upper()
    import re
    n = 0
    for line in open(paf):
        f = line.rstrip("\n").split("\t")
        q, qs, qe, strand, tname, ts, te = f[0], int(f[2]), int(f[3]), f[4], f[5], int(f[7]), int(f[8])
        cg = [x for x in f if x.startswith("cg:Z:")][0][5:]
        if strand != "+":
            continue
        query = seqs[q][qs:qe].encode()
        target = seqs[tname][ts:te].encode()
        legacy_ops = b"".join((b"M" if op in "=M" else op.encode()) * int(cnt) for cnt, op in re.findall(r"(\d+)([=XIDM])", cg))
        # legacy "M"/"=" may hide mismatches: re-derive X from the sequences
        v = h = 0
        fixed = bytearray()
        for op in legacy_ops:
            c = chr(op)
            if c == "M":
                fixed.append(ord("M") if target[v] == query[h] else ord("X"))
                v += 1
                h += 1
            elif c == "X":
                fixed.append(op)
                v += 1
                h += 1
            elif c == "I":
                fixed.append(op)
                h += 1
            else:
                fixed.append(op)
                v += 1
        assert v == len(target) and h == len(query)
        rc, ops, sc, _ = oracle.align_biwfa(target, query)
        assert rc == 0 and sc <= oracle.ops_score(bytes(fixed))
        n += 1
    assert n >= 1


def test_a_score_bound_does_not_change_the_breakpoint(oracle):
    """The product cuts its wavefronts to the diagonals from which the end is within reach of a score bound (children of a
    BiWFA split: the score their parent found + two gap openings; roots: the caller's guess) and starts the overlap loop
    as if a breakpoint just above the bound were in hand.  The claim -- same breakpoint, field for field, whenever the
    bound holds; nothing found when it does not -- is checked here on the CPU restatement of the reference's loop, with
    the cut and the stand-in added to it (oracle/wfa2p.c: wfo_find_breakpoint_bounded), over padded records, structural
    differences, and begin / end components as the recursion produces them."""
    rng = random.Random(77)
    fields = ("score", "score_forward", "score_reverse", "k_forward", "k_reverse", "offset_forward", "offset_reverse", "component")
    checked = failed_as_expected = fewer_cells = 0
    for i in range(260):
        L = rng.choice([300, 900, 2500])
        core = synth.random_dna(9000 + i, L)
        q = synth.mutate(core, rng.choice([0.002, 0.02, 0.08]), 31 * i + 1)
        shape = i % 4
        if shape == 0:      # a query against its padded window: two end gaps
            tgt = synth.random_dna(50000 + i, rng.randrange(20, 400)) + core + synth.random_dna(60000 + i, rng.randrange(20, 400))
        elif shape == 1:    # one-sided padding
            tgt = core + synth.random_dna(60000 + i, rng.randrange(50, 600))
        elif shape == 2:    # a structural difference in the middle
            cut = rng.randrange(50, 300)
            tgt, q = core, q[:len(q) // 2] + q[len(q) // 2 + cut:]
        else:               # balanced
            tgt = core
        constrained = i % 3 == 0
        cb, ce = (rng.randrange(0, 5), rng.randrange(0, 5)) if constrained else (0, 0)
        rc0, bp0, st0 = oracle.find_breakpoint(tgt, q, cb, ce)
        if rc0 != 0:
            continue
        S = bp0.score
        slack = 2 * 24 + 8 if constrained else 0  # a child that begins or ends inside a gap: see wfa_host.hip (children's bounds)
        for sub in (S + slack, S + slack + 1, S + slack + 9, S + 300, 3 * S + 1000):
            rc, bp, st = oracle.find_breakpoint_bounded(tgt, q, sub, cb, ce)
            assert rc == 0, (i, shape, cb, ce, S, sub, rc)
            assert all(getattr(bp, f) == getattr(bp0, f) for f in fields), (i, shape, cb, ce, S, sub)
            assert st.cells <= st0.cells
            fewer_cells += st.cells < st0.cells
            checked += 1
        for sub in (S - 1, S // 2):
            if sub < 1:
                continue
            rc, bp, st = oracle.find_breakpoint_bounded(tgt, q, sub, cb, ce)
            assert rc != 0, (i, shape, cb, ce, S, sub, bp.score)
            failed_as_expected += 1
    assert checked > 700 and failed_as_expected > 250 and fewer_cells > 200, (checked, failed_as_expected, fewer_cells)


def test_cutting_the_overlap_loop_into_rounds_does_not_change_the_breakpoint(oracle):
    """The product's phase 2 runs in rounds of 2 x 32 tests: a job whose loop has not ended takes another round from the state it
    is in, the breakpoint so far set aside and its score handed on as if it were a bound (BpJob::best0); a round that ends the loop
    without a better one reports WFM_DEV_P2_NOTHING and the one set aside stands.  The protocol on the CPU restatement of the
    reference's loop (oracle/wfa2p.c: wfo_find_breakpoint_rounds): the same breakpoint, field for field, for any round length, with
    and without a score bound -- on records whose loop is long (unrelated sequences, moved blocks: the antidiagonals touch long
    before two cells share a diagonal) as well as ordinary ones."""
    rng = random.Random(41)
    fields = ("score", "score_forward", "score_reverse", "k_forward", "k_reverse", "offset_forward", "offset_reverse", "component")
    multi = checked = 0
    for i in range(120):
        shape = i % 4
        if shape == 0:    # unrelated
            a, b = synth.random_dna(100 + i, rng.choice([300, 700])), synth.random_dna(300 + i, rng.choice([250, 800]))
        elif shape == 1:  # a block moved to the end
            u, v, w = synth.random_dna(500 + i, 300), synth.random_dna(700 + i, 200), synth.random_dna(900 + i, 300)
            a, b = u + v + w, u + w + synth.mutate(v, 0.03, i)
        elif shape == 2:  # ordinary divergence
            a = synth.random_dna(1100 + i, 900)
            b = synth.mutate(a, rng.choice([0.02, 0.1]), 7 * i)
        else:             # padded window
            b = synth.random_dna(1300 + i, 600)
            a = synth.random_dna(1500 + i, 150) + synth.mutate(b, 0.01, 3 * i) + synth.random_dna(1700 + i, 90)
        cb, ce = (rng.randrange(0, 5), rng.randrange(0, 5)) if i % 5 == 0 else (0, 0)
        rc0, bp0, _ = oracle.find_breakpoint(a, b, cb, ce)
        if rc0 != 0:
            continue
        for per_round in (1, 2, 5, 16, 64):
            for sub in (-1, bp0.score + 2 * 24 + 8 + rng.randrange(0, 50)):
                rc, bp, rounds = oracle.find_breakpoint_rounds(a, b, per_round, sub, cb, ce)
                assert rc == 0, (i, shape, per_round, sub, rc)
                assert all(getattr(bp, f) == getattr(bp0, f) for f in fields), (i, shape, per_round, sub, rounds)
                multi += rounds > 1
                checked += 1
    assert checked > 900 and multi > 600, (checked, multi)
