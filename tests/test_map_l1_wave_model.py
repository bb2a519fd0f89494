"""The claim behind the wave-per-fragment L1 sweep (wfmash_amd/csrc/map_l1.hip), checked without a GPU: the chunked,
prefix-sum form of computeL1CandidateRegions (oracle/map_l1_wave.py, a lane-by-lane model of the kernel) gives the
candidates of the line-by-line restatement (oracle/map_l1.py) on arbitrary sorted point lists -- dense positions, the same
position on neighbouring sequences, closes and opens at one position, any chunk width."""
import random

from hypothesis import given, settings, strategies as st

from oracle import map_l1 as L1
from oracle import map_l1_wave as LW

CUTOFFS = [1] * 4 + [2] * 4 + [3] * 8 + [5] * 64


def _points(rng, n_seq, n_iv, span, width_max):
    """interval points of random minmer intervals: [pos, hash, seqId, side], sorted by (seqId, pos, side)"""
    pts = []
    for _ in range(n_iv):
        s = rng.randrange(n_seq)
        a = rng.randrange(span)
        b = a + rng.randrange(1, width_max)
        h = rng.randrange(1 << 20)
        pts.append([a, h, s, L1.OPEN])
        pts.append([b, h, s, L1.CLOSE])
    pts.sort(key=lambda p: (p[2], p[0], p[3]))
    return pts


def _both(pts, qs, mh, w, stage1, full, width, carry_in=None):
    a = L1.l1_candidates(pts, w, qs, mh, w, 50, CUTOFFS, stage1, full, l1=[dict(c) for c in (carry_in or [])])
    b = LW.l1_candidates_wave(pts, qs, mh, w, 50, CUTOFFS, stage1, full, l1=[dict(c) for c in (carry_in or [])], width=width)
    return a, b


@settings(max_examples=300, deadline=None)
@given(seed=st.integers(0, 10**9), n_seq=st.integers(1, 4), n_iv=st.integers(1, 60), span=st.sampled_from([4, 12, 40, 300]),
       width_max=st.sampled_from([2, 5, 30]), mh=st.integers(1, 6), stage1=st.booleans(), full=st.booleans(),
       width=st.sampled_from([3, 4, 5, 8, 64]))
def test_wave_form_equals_the_sequential_sweep(seed, n_seq, n_iv, span, width_max, mh, stage1, full, width):
    rng = random.Random(seed)
    pts = _points(rng, n_seq, n_iv, span, width_max)
    a, b = _both(pts, 25, mh, 20, stage1, full, width)
    assert a == b


def test_candidates_join_across_group_calls_and_chunk_borders():
    """doL1Mapping calls the sweep once per reference group with the same output list: a candidate of the next call may join the
    last one of the previous (mappingCore.hpp:287-300)."""
    rng = random.Random(5)
    for trial in range(200):
        carry = []
        ca, cb = [], []
        for g in range(3):
            pts = _points(rng, 2, rng.randrange(1, 40), 60, 8)
            for p in pts:
                p[2] += 2 * g  # the groups' sequences follow one another
            ca = L1.l1_candidates(pts, 20, 25, 2, 20, 50, CUTOFFS, trial % 2 == 0, trial % 3 != 0, l1=ca)
            cb = LW.l1_candidates_wave(pts, 25, 2, 20, 50, CUTOFFS, trial % 2 == 0, trial % 3 != 0, l1=cb, width=rng.choice([3, 4, 7, 64]))
        assert ca == cb, trial


def test_same_position_on_neighbouring_sequences():
    """The corner the count's second term exists for: a position group that spans two sequences (equal pos, adjacent in the
    order) -- the trailing pointer stops at the end of the FIRST sequence's run."""
    pts = [[5, 1, 0, L1.OPEN], [5, 2, 0, L1.OPEN], [9, 1, 0, L1.CLOSE], [9, 3, 0, L1.OPEN], [9, 4, 1, L1.CLOSE], [9, 5, 1, L1.OPEN],
           [9, 6, 1, L1.OPEN], [12, 5, 1, L1.CLOSE], [14, 6, 1, L1.CLOSE]]
    for width in (2, 3, 4, 64):
        for mh in (1, 2, 3):
            a, b = _both(pts, 25, mh, 20, False, True, width)
            assert a == b, (width, mh)


@settings(max_examples=300, deadline=None)
@given(seed=st.integers(0, 10**9), n=st.integers(1, 80), n_seq=st.integers(1, 5), span=st.sampled_from([2, 3, 6, 20]), mh=st.integers(1, 5),
       stage1=st.booleans(), full=st.booleans(), width=st.sampled_from([2, 3, 5, 8, 64]))
def test_wave_form_on_arbitrary_sorted_keys(seed, n, n_seq, span, mh, stage1, full, width):
    """Not only lists that come from intervals: any sorted list of (seq, pos, side) keys -- closes before their opens, position
    groups across three sequences -- has the reference's loop's answer (the count may go negative there, too)."""
    rng = random.Random(seed)
    pts = [[rng.randrange(span), rng.randrange(1 << 20), rng.randrange(n_seq), rng.choice([L1.OPEN, L1.OPEN, L1.CLOSE])] for _ in range(n)]
    pts.sort(key=lambda p: (p[2], p[0], p[3]))
    a, b = _both(pts, 25, mh, 3, stage1, full, width)
    assert a == b
