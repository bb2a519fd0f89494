"""GPU parity tests of the automatic identity estimate: wfm_minhash_sketch against
the StreamingMinHash restatement, and the threshold + sketch size wfmh_map derives from it."""
import numpy as np
import pytest

from oracle import map_ani as ANI
from oracle import map_pipeline as MP
from wfmash_amd import capi, synth
from tests.test_map_paf_gpu import _pangenome, _write_fasta

pytestmark = pytest.mark.gpu


def test_minhash_sketch_matches_restatement(gpu):
    base = synth.random_dna(77, 60000)
    cases = {
        "plain": base,
        "lower_and_iupac": base[:20000].lower() + b"RYKM" + base[20000:40000] + b"N" * 500 + base[40000:],
        "ambiguous_head": base[:7] + b"N" + base[8:30000],          # arms the counter for k-mers 0..20
        "repeat": synth.random_dna(5, 300) * 150,                    # duplicates fill the sketch
        "short": base[:2000],                                        # fewer k-mers than the sketch holds
        "tiny": base[:20],                                           # shorter than k
        "all_n": b"N" * 3000,
    }
    for name, sq in cases.items():
        got = gpu.minhash_sketch(sq)
        exp = ANI.minhash_sketch(sq)
        assert len(got) == len(exp), name
        assert (got == exp).all(), name
    assert len(gpu.minhash_sketch(cases["repeat"])) == 4096 and len(set(gpu.minhash_sketch(cases["repeat"]).tolist())) < 400
    small = gpu.minhash_sketch(base, k=15, sketch_size=64)
    assert (small == ANI.minhash_sketch(base, 15, 64)).all()


def test_minhash_sketch_of_a_long_sequence_selects_before_it_sorts(gpu):
    """beyond 1 Mbp the sketch comes from the hashes under a threshold (selected, then sorted) instead of a sort of
    all hashes; the result is the same multiset -- checked against a plain sort of the hash kernel's output, also
    when a repeat makes the smallest hashes occur many times, and with a sketch so large that the selection falls
    back to the full sort"""
    import numpy as np
    rng = np.random.default_rng(12)
    seq = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 3_000_000)])
    rep = seq[:1_500_000] + seq[1000:1400] * 500 + seq[1_500_000:2_500_000]
    for sq, n in ((seq, 4096), (rep, 4096), (seq, 60000)):
        h, st = gpu.hash_kmers(sq, 21)
        assert (st != 0).all()  # k odd, no N: every k-mer is valid
        want = np.sort(h)[:n]
        got = gpu.minhash_sketch(sq, k=21, sketch_size=n)
        assert len(got) == n and (got == want).all()
    # an ambiguous base among the first k bases blanks k-mers 0 .. k-1 (map_stats.hpp:574-580): in the one-pass form (hashing and
    # selection in one kernel, round 5) they are skipped, in the two-pass form their hashes are overwritten
    amb = seq[:7] + b"N" + seq[8:]
    h, st = gpu.hash_kmers(amb, 21)
    h = h.copy()
    h[:21] = np.iinfo(np.uint64).max
    h[st == 0] = np.iinfo(np.uint64).max
    for n in (4096, 60000):
        got = gpu.minhash_sketch(amb, k=21, sketch_size=n)
        assert len(got) == n and (got == np.sort(h)[:n]).all()


def test_sequence_cache_serves_the_same_buffer_and_only_that(gpu):
    """wfm_map_sequence_cache: inside a scope the normalised device copy of a sequence of a megabase and more is kept and found again by pointer,
    length and first / last bytes -- the second sketch and the k-mer hashes of the same buffer come from it and equal the uncached ones; the same
    memory with other contents (its first bases rewritten) is a different sequence; a change the fingerprint cannot see is the caller's contract
    (the scope's host memory does not change) and is NOT noticed; after the scope everything is uploaded again."""
    import ctypes as C
    rng = np.random.default_rng(5)
    arr = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 2_000_000)].copy()
    L = capi.load()
    L.wfm_map_sequence_cache.restype = C.c_int
    L.wfm_map_sequence_cache.argtypes = [C.c_void_p, C.c_int]

    def sketch(a, n=2048):
        out = np.zeros(n, dtype=np.uint64)
        f = L.wfm_minhash_sketch
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        m = f(gpu._p, a.ctypes.data, len(a), 21, n, out.ctypes.data)
        assert m == n
        return out

    plain = sketch(arr)
    assert L.wfm_map_sequence_cache(gpu._p, 1) == 0
    try:
        first = sketch(arr)      # uploads, keeps
        second = sketch(arr)     # from the kept copy
        assert (first == plain).all() and (second == plain).all()
        h1, _ = gpu.hash_kmers(bytes(arr[:0]) + arr.tobytes(), 21)  # another buffer with the same contents: its own upload
        saved = arr[:40].copy()
        arr[:40] = np.frombuffer(b"T" * 40, dtype=np.uint8)
        changed = sketch(arr)    # same pointer and length, other first bytes: not the kept copy
        want = np.sort(gpu.hash_kmers(arr.tobytes(), 21)[0])[:2048]
        assert (changed == want).all() and not (changed == plain).all()
        arr[:40] = saved
        mid = arr[1_000_000:1_000_040].copy()
        arr[1_000_000:1_000_040] = np.frombuffer(b"G" * 40, dtype=np.uint8)
        stale = sketch(arr)      # the fingerprint does not see the middle: the kept copy answers (the caller's contract)
        assert (stale == plain).all()
        arr[1_000_000:1_000_040] = mid
    finally:
        assert L.wfm_map_sequence_cache(gpu._p, 0) == 0
    arr[1_000_000:1_000_040] = np.frombuffer(b"G" * 40, dtype=np.uint8)
    after = sketch(arr)          # no scope: uploaded again
    assert (after == np.sort(gpu.hash_kmers(arr.tobytes(), 21)[0])[:2048]).all()


def test_auto_identity_drives_threshold_and_sketch_size(gpu, tmp_path):
    seqs = _pangenome(51)
    fa = str(tmp_path / "pan.fa")
    _write_fasta(fa, seqs)
    names = [n for n, _ in seqs]
    groups = MP.ref_groups(names)
    for pctl, adj in ((50, -2.0), (25, 0.0)):
        exp = ANI.estimate_identity([s for _, s in seqs], groups, pctl, adj)
        P = capi.map_default_params(ani_percentile=pctl, ani_adjustment=adj)
        assert P.auto_pct_identity == 1
        summ = capi.map_paf(gpu, fa, str(tmp_path / f"m{pctl}.paf"), params=P)
        assert summ.percentage_identity == np.float32(exp)
        assert 0.90 < summ.percentage_identity < 1.0
        assert summ.sketch_size == MP.sketch_size(np.float32(exp), 1000, 15)
        assert summ.written > 5
