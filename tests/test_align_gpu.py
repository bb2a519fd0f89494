"""GPU parity tests of the align path: libwfmash_hip.so (through the C ABI)
against the CPU oracle on the same seeded inputs.  Bit-exact on op strings."""
import random

import pytest

from wfmash_amd import capi, synth

pytestmark = pytest.mark.gpu


def _pairs(seed, n, lens, rates):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        L = rng.choice(lens)
        p = synth.random_dna(seed * 1000 + i, L)
        t = synth.mutate(p, rng.choice(rates), seed * 7919 + i) if L else b""
        r = rng.random()
        if r < 0.08:
            t = synth.random_dna(seed * 31 + i, rng.randrange(0, 400))
        elif r < 0.12:
            t = b""
        elif r < 0.16:
            p = b""
        out.append((p, t))
    return out


def _check_batch(gpu, oracle, items, pen=None):
    res = gpu.align(items, pen)
    n_bad = 0
    for (p, t), r in zip(items, res):
        rc, ops, sc, _ = oracle.align_biwfa(p, t, pen)
        assert rc == 0
        assert r.status == 0, (len(p), len(t), r.status)
        assert r.score == sc
        if r.ops != ops:
            n_bad += 1
    assert n_bad == 0, f"{n_bad}/{len(items)} CIGARs differ from the oracle"


def test_biwfa_matches_oracle_small(gpu, oracle):
    items = _pairs(1, 300, [0, 1, 2, 3, 15, 64, 99, 100, 101, 130, 256, 400, 777], [0.0, 0.01, 0.05, 0.15, 0.35])
    _check_batch(gpu, oracle, items)


def test_biwfa_matches_oracle_medium(gpu, oracle):
    items = _pairs(2, 96, [1000, 2500, 5000, 9000], [0.01, 0.05, 0.1, 0.2])
    _check_batch(gpu, oracle, items)


def test_biwfa_skewed_shapes(gpu, oracle):
    """Very unequal lengths: deep D/I components, breakpoints inside long gaps."""
    rng = random.Random(9)
    items = []
    for i in range(60):
        a = synth.random_dna(4000 + i, rng.choice([120, 600, 3000]))
        b = synth.random_dna(5000 + i, rng.choice([120, 150, 2000]))
        core = synth.random_dna(6000 + i, rng.choice([200, 1500]))
        items.append((a + core + b, synth.mutate(core, 0.05, 70 + i)))
        items.append((synth.mutate(core, 0.1, 170 + i), b + core + a))
    _check_batch(gpu, oracle, items)


def test_biwfa_low_complexity(gpu, oracle):
    """Homopolymers / tandem repeats create many equal-score paths: exercises every tie-break."""
    rng = random.Random(4)
    items = []
    for i in range(80):
        unit = synth.random_dna(900 + i, rng.choice([1, 2, 3, 7]))
        n = rng.choice([150, 700, 2500])
        p = (unit * (n // len(unit) + 1))[:n]
        t = synth.mutate(p, rng.choice([0.02, 0.1]), 333 + i)
        if rng.random() < 0.5:
            t = (unit * 400)[:max(1, n + rng.randrange(-100, 100))]
        items.append((p, t))
    _check_batch(gpu, oracle, items)


def test_biwfa_with_N_runs(gpu, oracle):
    items = []
    for i in range(20):
        p = bytearray(synth.random_dna(40 + i, 3000))
        p[1000:1300] = b"N" * 300
        t = bytearray(synth.mutate(bytes(p), 0.05, 400 + i))
        items.append((bytes(p), bytes(t)))
    _check_batch(gpu, oracle, items)


def test_wave_shifts_the_packed_tile_kernel_leans_on(gpu):
    """wfa_tile2.hip takes a lane's neighbours with DPP wave_shr:1 / wave_shl:1: lane i <- lane i -+ 1, the edge lanes keep NULL."""
    import ctypes as C
    import numpy as np
    out = np.zeros(128, dtype=np.int32)
    L = capi.load()
    L.wfm_selftest_dpp.restype = C.c_int
    L.wfm_selftest_dpp.argtypes = [C.c_void_p, C.c_void_p]
    assert L.wfm_selftest_dpp(gpu._p, out.ctypes.data) == 0
    null = -(1 << 30)
    assert out[:64].tolist() == [null] + [1000 + i for i in range(63)]
    assert out[64:].tolist() == [1000 + i for i in range(1, 64)] + [null]


def test_arenas_at_least_double_when_they_grow(gpu):
    """DevBuf::ensure (the rings, the phase-2 rows, the base arenas): a regrowth is a fresh hipMalloc, so it at least doubles; a request that fits
    allocates nothing.  (Until round 4 the old capacity was reset before it was read: the doubling never happened.)"""
    import ctypes as C
    L = capi.load()
    L.wfm_selftest_arena_growth.restype = C.c_int
    L.wfm_selftest_arena_growth.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    out = (C.c_size_t * 3)()
    assert L.wfm_selftest_arena_growth(gpu._p, 1 << 20, out) == 0
    c0, c1, c2 = out[0], out[1], out[2]
    assert c0 >= 1 << 20 and c1 >= 2 * c0 and c2 == c1


def test_packed_and_byte_kernels_in_one_batch(gpu, oracle):
    """A batch in which some problems are pure ACGT (2-bit mirror, wfa_tile2_kernel) and others hold an N or soft-masked bases
    (byte kernel): both kinds of tiles run side by side in every block of a level.  N matches N and nothing else; a lower-case
    base matches only itself (the C ABI compares bytes, as WFA2-lib does)."""
    items = []
    for i in range(24):
        p = bytearray(synth.random_dna(700 + i, 6000))
        if i % 3 == 1:
            p[2000:2100] = b"N" * 100
        t = bytearray(synth.mutate(bytes(p), 0.04, 7000 + i))
        if i % 3 == 2:
            t[3000:3050] = bytes(t[3000:3050]).lower()
        items.append((bytes(p), bytes(t)))
    _check_batch(gpu, oracle, items)
    # low divergence, long runs of matches: the wave-cooperative tail of the packed extension, windows left behind
    items = []
    for i in range(12):
        p = synth.random_dna(800 + i, 60000)
        items.append((p, synth.mutate(p, 0.002, 8000 + i)))
    _check_batch(gpu, oracle, items)


def test_custom_penalties(gpu, oracle):
    items = _pairs(5, 40, [80, 300, 2000], [0.05, 0.2])
    _check_batch(gpu, oracle, items, pen=(4, 6, 2, 12, 1))
    _check_batch(gpu, oracle, items, pen=(3, 4, 1, 10, 1))


def test_unsupported_penalties_fail_loudly(gpu):
    with pytest.raises(capi.WfmError):
        gpu.align([(b"ACGT" * 50, b"ACGA" * 50)], pen=(5, 8, 2, 200, 1))  # scope 202 > the 128-row rings
    with pytest.raises(capi.WfmError):
        gpu.align([(b"ACGT" * 50, b"ACGA" * 50)], pen=(0, 8, 2, 24, 1))   # mismatch 0: no wavefront order


def test_penalties_beyond_the_default_scope(gpu, oracle):
    """Any -g the reference accepts (parse_args.hpp:272-288) up to o2 + e2 = 125: scopes above 32 run on 128-row rings
    (step kernel, LDS tiles where they fit, base kernel).  Against the oracle, which is generic in its penalties."""
    items = _pairs(15, 36, [60, 150, 700, 2600], [0.03, 0.12])
    rng = random.Random(8)
    for i in range(10):  # long gaps: the second piece is what a large o2 prices
        a = synth.random_dna(8000 + i, rng.choice([300, 1500]))
        b = synth.random_dna(8100 + i, rng.choice([200, 900]))
        items.append((a + b, b) if i % 2 else (b, a + b))
    for pen in ((5, 8, 2, 60, 1), (6, 10, 3, 124, 1), (9, 40, 2, 100, 1), (33, 20, 2, 24, 1)):
        _check_batch(gpu, oracle, items, pen=pen)
    # and through the ends-free form (patches)
    p = synth.random_dna(8300, 900)
    t = synth.random_dna(8301, 40) + synth.mutate(p, 0.1, 8302)
    r = gpu.align([(p, t, capi.WFM_MODE_ENDSFREE, len(p), 0, len(t), 0)], pen=(5, 8, 2, 60, 1))[0]
    rc, ops, sc, _ = oracle.align_endsfree(p, len(p), 0, t, len(t), 0, pen=(5, 8, 2, 60, 1))
    assert rc == 0 and r.status == 0 and r.ops == ops


def test_endsfree_patches_match_oracle(gpu, oracle):
    """Head / tail patch forms exactly as do_biwfa_alignment issues them (wflign.cpp:300-305, 392-397)."""
    rng = random.Random(6)
    items, exp = [], []
    for i in range(120):
        p = synth.random_dna(700 + i, rng.choice([5, 60, 130, 400, 1500]))
        t = synth.mutate(p, rng.choice([0.0, 0.05, 0.2, 0.5]), 800 + i)
        if rng.random() < 0.4:
            t = synth.random_dna(70 + i, rng.randrange(1, 60)) + t
        if rng.random() < 0.4:
            p = synth.random_dna(90 + i, rng.randrange(1, 60)) + p
        if not t:
            t = b"A"
        for args in ((len(p), 0, len(t), 0), (0, len(p), 0, len(t))):
            items.append((p, t, capi.WFM_MODE_ENDSFREE, args[0], args[1], args[2], args[3]))
            exp.append(oracle.align_endsfree(p, args[0], args[1], t, args[2], args[3]))
    res = gpu.align(items)
    bad = 0
    for it, r, (rc, ops, sc, _) in zip(items, res, exp):
        assert rc == 0 and r.status == 0
        bad += r.ops != ops
    assert bad == 0, f"{bad}/{len(items)} ends-free CIGARs differ"


def test_wide_patches_on_tiles_match_oracle(gpu, oracle):
    """Ends-free patches whose score passes the second budget (1020): their third attempt has rows beyond the register kernel's 2048 diagonals
    and runs as tiles of it (wfa_base2t_kernel: blocks of 125 scores, halos, snapshots) -- head form (band around the end corner), tail form
    (rows as wide as the problem: nine tiles), an unrelated stretch in front, one score that ends exactly on a block's last row is not forced
    but the scores spread over several blocks.  CIGARs identical to the oracle's; and identical to the ring kernel's (WFM_BASE_TILES=0)."""
    import os
    items, exp = [], []
    for i, (L, div, pre) in enumerate([(3400, 0.10, 0), (3000, 0.14, 0), (3600, 0.08, 300), (2600, 0.2, 0), (3900, 0.07, 0)]):
        p = synth.random_dna(5100 + i, L)
        t = synth.random_dna(5200 + i, pre) + synth.mutate(p, div, 5300 + i)
        for args in ((len(p), 0, len(t), 0), (0, len(p), 0, len(t))):
            items.append((p, t, capi.WFM_MODE_ENDSFREE, args[0], args[1], args[2], args[3]))
            exp.append(oracle.align_endsfree(p, args[0], args[1], t, args[2], args[3]))
    res = gpu.align(items)
    fl = gpu.problem_flags(len(items))
    assert sum(bool(f & capi.WFM_PF_BASE_TILES) for f in fl) >= 6, fl
    scores = set()
    for it, r, f, (rc, ops, sc, _) in zip(items, res, fl, exp):
        assert rc == 0 and r.status == 0
        assert r.ops == ops, (len(it[0]), len(it[1]), it[3:], sc)
        if f & capi.WFM_PF_BASE_TILES:
            scores.add(r.score // 125)
    assert len(scores) >= 3  # (walks that end in different blocks)
    os.environ["WFM_BASE_TILES"] = "0"
    try:
        res0 = gpu.align(items)
    finally:
        del os.environ["WFM_BASE_TILES"]
    assert all(a.ops == b.ops and a.score == b.score for a, b in zip(res, res0))


def test_every_leaf_on_tiles_matches_oracle(gpu, oracle):
    """WFM_BASE_TILES=2 sends every leaf and patch with rows beyond 128 diagonals through wfa_base2t_kernel -- the leaves of BiWFA's recursion with
    their begin / end components (a child that begins or ends inside a gap), retried leaves, head and tail patches of all widths, mostly one tile
    per job: the mixed batch of the ends-free test and a batch of bialign problems with long gaps, identical to the oracle."""
    import os
    rng = random.Random(16)
    items, exp = [], []
    for i in range(60):
        p = synth.random_dna(1700 + i, rng.choice([60, 130, 400, 1500]))
        t = synth.mutate(p, rng.choice([0.0, 0.05, 0.2, 0.5]), 1800 + i) or b"A"
        if rng.random() < 0.4:
            t = synth.random_dna(170 + i, rng.randrange(1, 60)) + t
        for args in ((len(p), 0, len(t), 0), (0, len(p), 0, len(t))):
            items.append((p, t, capi.WFM_MODE_ENDSFREE, args[0], args[1], args[2], args[3]))
            exp.append(oracle.align_endsfree(p, args[0], args[1], t, args[2], args[3])[1])
    pairs = _pairs(9, 40, [300, 1200, 5000], [0.02, 0.1, 0.3])
    for i in range(12):  # long gaps: children that begin / end inside a gap
        a = synth.random_dna(2600 + i, rng.choice([1500, 4000]))
        cut = rng.randrange(200, len(a) - 200)
        gap = synth.random_dna(2700 + i, rng.choice([30, 300, 900]))
        b = synth.mutate(a[:cut], 0.03, 2800 + i) + gap + synth.mutate(a[cut:], 0.03, 2900 + i)
        pairs.append((a, b) if i % 2 else (b, a))
    for p, t in pairs:
        items.append((p, t))
        exp.append(oracle.align_biwfa(p, t)[1])
    os.environ["WFM_BASE_TILES"] = "2"
    try:
        res = gpu.align(items)
        fl = gpu.problem_flags(len(items))
    finally:
        del os.environ["WFM_BASE_TILES"]
    assert sum(bool(f & capi.WFM_PF_BASE_TILES) for f in fl) >= len(items) // 3
    bad = [i for i, (r, ops) in enumerate(zip(res, exp)) if r.status != 0 or r.ops != ops]
    assert not bad, f"{len(bad)}/{len(items)} differ: {bad[:8]}"


def test_uni_mode_matches_oracle(gpu, oracle):
    items = [(p, t, capi.WFM_MODE_END2END_UNI) for p, t in _pairs(8, 40, [50, 300, 1200], [0.02, 0.1, 0.3])]
    res = gpu.align(items)
    for (p, t, _), r in zip(items, res):
        rc, ops, sc, _ = oracle.align_uni(p, t)
        assert rc == 0 and r.status == 0 and r.ops == ops and r.score == sc


def test_c3_sized_pairs_properties(gpu, oracle):
    """BASELINE.json configs[2] at full size (8 of the 64 pairs here): CIGAR valid,
    implied score == reported score, and bit-identical to the oracle."""
    pairs = synth.pairs("C3", n_pairs=8)
    res = gpu.align(pairs)
    ops_cpu, scores, _, failed = oracle.align_batch_biwfa([p for p, _ in pairs], [q for _, q in pairs])
    assert failed == 0
    for (p, t), r, oc, sc in zip(pairs, res, ops_cpu, scores):
        assert r.status == 0
        assert oracle.ops_check(r.ops, p, t) == 0
        assert oracle.ops_score(r.ops) == r.score == int(sc)
        assert r.ops == oc


def test_memory_budget_chunking(oracle, monkeypatch):
    """A tiny device-memory budget forces the level loop to run in chunks; results must not change."""
    monkeypatch.setenv("WFM_MEM_BUDGET_MB", "24")
    h = capi.Handle(0)
    try:
        items = _pairs(12, 24, [3000, 6000], [0.05, 0.1])
        _check_batch(h, oracle, items)
    finally:
        h.close()


@pytest.mark.parametrize("band_root", ["4096", "200"], ids=["bands_hold", "bands_overflow"])
def test_narrow_rings_and_their_retries(oracle, monkeypatch, capfd, band_root):
    """When a level does not fit the memory budget, jobs get rings for the diagonals they are expected to reach only
    (low-divergence long records); a job that runs out of its band is run again on a full ring.  Forced here with a
    small budget; with 200 root scores most roots overflow (in the tile phase or in the step kernel) and are retried."""
    monkeypatch.setenv("WFM_MEM_BUDGET_MB", "96")
    monkeypatch.setenv("WFM_BAND_ROOT", band_root)
    monkeypatch.setenv("WFM_DEBUG", "1")
    monkeypatch.setenv("WFM_OVERLAP", "0")
    h = capi.Handle(0)
    try:
        items = _pairs(31, 40, [9000, 14000], [0.002, 0.01, 0.04])
        _check_batch(h, oracle, items)
    finally:
        h.close()
    err = capfd.readouterr().err
    lines = [l for l in err.splitlines() if "narrow rings" in l]
    assert lines, err[-1500:]
    jobs, retried = (int(x) for x in __import__("re").search(r"narrow rings: (\d+) jobs, (\d+) ran out", lines[-1]).groups())
    assert jobs > 0
    if band_root == "200":
        assert retried > 0, lines[-1]


def test_narrow_rings_can_be_switched_off(oracle, monkeypatch, capfd):
    monkeypatch.setenv("WFM_MEM_BUDGET_MB", "96")
    monkeypatch.setenv("WFM_BAND", "0")
    monkeypatch.setenv("WFM_DEBUG", "1")
    h = capi.Handle(0)
    try:
        _check_batch(h, oracle, _pairs(32, 12, [9000], [0.01]))
    finally:
        h.close()
    assert "narrow rings" not in capfd.readouterr().err


def test_repeated_calls_reuse_handle(gpu, oracle):
    items = _pairs(13, 10, [500, 2000], [0.05])
    a = gpu.align(items)
    b = gpu.align(items)
    assert [r.ops for r in a] == [r.ops for r in b]


def test_tiled_kernels_agree_with_step_kernel(oracle, monkeypatch):
    """Phase 1 has three implementations (step kernel, LDS time tiles, register time tiles).
    On multi-tile wavefronts (C3-sized pairs) all three must give identical op strings, and
    those must be optimal (score == oracle).  Regression for the 26-row snapshot depth."""
    pairs = synth.pairs("C3", n_pairs=24)[8:24]
    got = {}
    for name, env in (("step", {"WFM_TILE": "0"}), ("lds", {"WFM_TILE_REG": "0"}), ("reg", {}),
                      ("reg_small", {"WFM_TILE_THREADS": "256", "WFM_TILE_T": "32"}),
                      ("reg_bytes", {"WFM_TILE_V2": "0"}), ("reg_bytes_small", {"WFM_TILE_V2": "0", "WFM_TILE_THREADS": "256", "WFM_TILE_T": "32"}),
                      ("reg_r4form", {"WFM_TILE_FAST": "0"})):
        for k in ("WFM_TILE", "WFM_TILE_REG", "WFM_TILE_THREADS", "WFM_TILE_T", "WFM_TILE_V2", "WFM_TILE_FAST"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = capi.Handle(0)
        try:
            got[name] = h.align(pairs)
        finally:
            h.close()
    for name in ("lds", "reg", "reg_small", "reg_bytes", "reg_bytes_small", "reg_r4form"):
        bad = [i for i in range(len(pairs)) if got[name][i].ops != got["step"][i].ops]
        assert not bad, (name, bad)
    ops_cpu, scores, _, failed = oracle.align_batch_biwfa([p for p, _ in pairs[:4]], [q for _, q in pairs[:4]])
    assert failed == 0
    for i in range(4):
        assert got["reg"][i].ops == ops_cpu[i]


def test_generic_penalties_use_lds_tiles(gpu, oracle):
    """Non-default penalties take the LDS time-tile kernel (wfa_tile_kernel); sequences long
    enough for several tiles per wavefront and several score blocks."""
    items = []
    for i, (n, rate) in enumerate([(12000, 0.08), (20000, 0.05), (7000, 0.15), (16000, 0.1)]):
        p = synth.random_dna(7100 + i, n)
        items.append((p, synth.mutate(p, rate, 7200 + i)))
    for pen in ((4, 6, 2, 12, 1), (3, 4, 1, 10, 1), (5, 8, 2, 24, 2)):
        _check_batch(gpu, oracle, items, pen=pen)


def test_c5_shape_properties(gpu, oracle):
    """BASELINE.json configs[4] (100 kb, 15 %): deep wavefronts.  Full-size pairs are checked
    through size-independent properties (valid CIGAR, implied score == reported score,
    step kernel == tiled kernel); a 12 kb cut of the same generator is checked bit-exactly."""
    pairs = synth.pairs("C5", n_pairs=2)
    res = gpu.align(pairs)
    for (p, t), r in zip(pairs, res):
        assert r.status == 0
        assert oracle.ops_check(r.ops, p, t) == 0
        assert oracle.ops_score(r.ops) == r.score
    small = synth.pairs("C5", n_pairs=3, length=12000)
    _check_batch(gpu, oracle, small)


def test_c5_full_size_pairs_match_the_oracle_fixture(gpu):
    """BASELINE.json configs[4] at FULL size against the oracle: two 100 kb / 15 % pairs (~10^10 wavefront cells each,
    scores ~98 k).  The oracle's answer is a committed fixture (tests/golden/c5_golden.json.gz, written by
    tests/golden/make_c5_golden.py from oracle/wfa2p.c -- minutes of CPU per pair): score, length and every op of the
    CIGAR must be identical."""
    import gzip
    import hashlib
    import json
    import os
    import re
    g = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c5_golden.json.gz"), "rt"))
    pairs = synth.pairs("C5", n_pairs=len(g["pairs"]))
    res = gpu.align(pairs)
    for (p, t), r, want in zip(pairs, res, g["pairs"]):
        assert (len(p), len(t)) == (want["plen"], want["tlen"])
        assert r.status == 0 and r.score == want["score"] and len(r.ops) == want["n_ops"]
        assert hashlib.sha256(r.ops).hexdigest() == want["sha256"]
        rle = "".join(f"{len(m.group(0))}{m.group(0)[0]}" for m in re.finditer(r"M+|X+|I+|D+", r.ops.decode()))
        assert rle == want["rle"]
    assert min(w["score"] for w in g["pairs"]) > 90000


def test_leaf_that_ends_inside_a_long_gap(gpu, oracle):
    """A record of the 40 Mbp C4 variant (hap7 9257000-9300000 '-' against hap5) that round 1's kernels dropped with status
    -300: one BiWFA leaf is 5 x 155 bases and must END inside a D2 gap, so its own forward score (208) exceeds both the
    score its parent credited it with (184: the gap's opening is counted on the other side of the breakpoint) and the
    unconstrained all-gap bound the retry loop stopped at (205).  The fixture holds the two sequences and the oracle's answer."""
    import gzip
    import hashlib
    import json
    import os
    g = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "leaf_in_gap_pair.json.gz"), "rt"))
    p, t = g["pattern"].encode(), g["text"].encode()
    r = gpu.align([(p, t)])[0]
    assert r.status == 0 and r.score == g["score"] == 3044
    assert len(r.ops) == g["n_ops"] and hashlib.sha256(r.ops).hexdigest() == g["ops_sha"]
    rc, ops, sc, _ = oracle.align_biwfa(p, t)
    assert rc == 0 and ops == r.ops


def _padded_records(seed, n, qlen, pad, rate):
    """records as the align driver hands them over: a query against its target window, the window padded on both sides
    (pattern = target, text = query: wflign.cpp:136-148)"""
    rng = random.Random(seed)
    out = []
    for i in range(n):
        tgt = synth.random_dna(seed * 100 + i, qlen + rng.randrange(0, 2 * pad + 1))
        a = rng.randrange(0, len(tgt) - qlen + 1)
        q = synth.mutate(tgt[a:a + qlen], rate, seed * 57 + i)
        if rng.random() < 0.1:  # a structural difference in the middle: the score leaves every sensible guess behind
            cut = rng.randrange(200, 900)
            q = q[:qlen // 2] + q[qlen // 2 + cut:]
        out.append((tgt, q))
    return out


def test_score_hints_cut_the_wavefronts_not_the_result(gpu, oracle):
    """wfm_problem_t::score_hint: an upper bound of the score lets the kernels skip every diagonal from which the end is
    out of reach (children of a BiWFA split always carry their exact bound).  A generous hint, a tight one, one that is
    too small (the record runs again without it) and none at all must give the oracle's CIGAR -- and fewer cells."""
    items = _padded_records(3, 48, 6000, 400, 0.004) + _padded_records(4, 16, 12000, 1000, 0.02)
    exp = [oracle.align_biwfa(p, t) for p, t in items]
    assert all(e[0] == 0 for e in exp)
    E = capi.WFM_MODE_END2END_BIWFA
    plain = gpu.align([(p, t, E, 0, 0, 0, 0, 0) for p, t in items])
    cells = {}
    for name, hint in (("generous", lambda sc: 2 * sc + 500), ("tight", lambda sc: sc + 60), ("exact", lambda sc: sc), ("small", lambda sc: max(1, sc // 2)),
                       ("tiny", lambda sc: 1)):
        res = gpu.align([(p, t, E, 0, 0, 0, 0, hint(e[2])) for (p, t), e in zip(items, exp)])
        bad = [i for i, (r, e) in enumerate(zip(res, exp)) if r.status != 0 or r.score != e[2] or r.ops != e[1]]
        assert not bad, (name, bad[:5])
        cells[name] = sum(r.cells for r in res)
    assert all(r.status == 0 and r.ops == e[1] for r, e in zip(plain, exp))
    c0 = sum(r.cells for r in plain)
    assert cells["exact"] <= cells["tight"] < 0.95 * c0 and cells["generous"] <= c0, (cells, c0)
    assert cells["small"] >= c0  # every record ran again without its hint


def _expand(runs):
    return b"".join(op * n for n, op in runs)


def test_run_length_output_spells_the_oracles_ops(gpu, oracle):
    """wfm_align_batch_rle (the form the align driver consumes): per problem the runs spell the oracle's op string, no two
    neighbours share an op, ops_len / n_runs / score agree with the expanded form; mixed batch (BiWFA over three parts of
    the batch, ends-free patches, empty sequences), so the parts' run buffers are stitched into one."""
    rng = random.Random(77)
    items = _pairs(7, 120, [0, 3, 99, 130, 777, 2500, 6000], [0.0, 0.01, 0.05, 0.2])
    for i in range(30):
        p = synth.random_dna(9100 + i, rng.choice([60, 400, 1500]))
        t = synth.mutate(p, rng.choice([0.0, 0.05, 0.3]), 9200 + i) or b"A"
        if i % 2:
            items.append((p, t, capi.WFM_MODE_ENDSFREE, len(p), 0, len(t), 0))
        else:
            items.append((p, t, capi.WFM_MODE_ENDSFREE, 0, len(p), 0, len(t)))
    rng.shuffle(items)
    rle = gpu.align_rle(items)
    full = gpu.align(items)
    for it, (r, ops_len), f in zip(items, rle, full):
        p, t = it[0], it[1]
        if len(it) > 2:
            rc, ops, sc, _ = oracle.align_endsfree(p, it[3], it[4], t, it[5], it[6])
        else:
            rc, ops, sc, _ = oracle.align_biwfa(p, t)
        assert rc == 0 and r.status == 0 and f.status == 0
        assert _expand(r.ops) == ops == f.ops
        assert all(a[1] != b[1] for a, b in zip(r.ops, r.ops[1:])) and all(n > 0 for n, _ in r.ops)
        assert ops_len == len(ops) and r.n_runs == len(r.ops) == f.n_runs and r.score == f.score
        if len(it) == 2:
            assert r.score == sc


def _padded_record(seed, L, snp, indel, pad_t, pad_q=0, sv=0):
    """a mapping record as the align driver hands it over: query against its padded target window"""
    import numpy as np
    base = synth.random_backbone(seed, L + 2 * pad_t + 64)
    hap = synth.haplotype(base, seed + 1, snp=snp, indel=indel, n_sv=sv, sv_min=300, sv_max=900)
    t = base.tobytes()
    q = hap.tobytes()[pad_t - pad_q:pad_t - pad_q + L]
    return t[:L + 2 * pad_t], q


def test_score_bounds_are_upper_bounds_and_tight_on_padded_records(gpu, oracle):
    """wfm_score_bounds: never below the optimal score (oracle: the O(nm)-checked WFA score), -1 allowed; on low-divergence
    padded records -- what a pangenome batch is made of -- within a few per cent of it; divergent pairs give -1 or a bound."""
    items, kinds = [], []
    for i in range(24):
        t, q = _padded_record(500 + i, 6000 + 400 * i, 1e-3 * (1 + i % 4), 1e-4 * (1 + i % 3), pad_t=[1000, 300, 1000, 0][i % 4] if i % 4 != 3 else 64,
                              pad_q=[0, 0, 200, 0][i % 4])
        items.append((t, q)); kinds.append("padded")
    for i in range(6):  # the query runs ahead instead (an insertion opens the alignment)
        t, q = _padded_record(700 + i, 5000, 1e-3, 1e-4, pad_t=800)
        items.append((q, t)); kinds.append("swapped")
    for i in range(6):
        p = synth.random_dna(900 + i, 4000)
        items.append((p, synth.mutate(p, [0.02, 0.05, 0.15][i % 3], 950 + i))); kinds.append("divergent")
    for i in range(4):
        t, q = _padded_record(800 + i, 9000, 1e-3, 1e-4, pad_t=500, sv=2)
        items.append((t, q)); kinds.append("sv")
    items.append((synth.random_dna(1, 100), synth.random_dna(2, 120))); kinds.append("short")
    ub = gpu.score_bounds(items)
    tight = 0
    for (p, t), u, kind in zip(items, ub, kinds):
        rc, ops, sc, _ = oracle.align_biwfa(p, t)
        assert rc == 0
        assert u == -1 or u >= sc, (kind, len(p), len(t), int(u), sc)
        if kind in ("padded", "swapped"):
            assert u != -1, (kind, len(p), len(t))
            assert u <= sc + max(40, sc // 20), (kind, int(u), sc)
            tight += u <= sc + 10
        if kind == "short":
            assert u == -1
    assert tight >= 20
    # and the alignments themselves do not change (the bound only cuts cells no alignment of that score can touch)
    res = gpu.align(items)
    for (p, t), r in zip(items, res):
        rc, ops, sc, _ = oracle.align_biwfa(p, t)
        assert r.status == 0 and r.ops == ops and r.score == sc


def test_exact_bounds_on_small_batches_with_reused_rings(oracle):
    """Round-3 regression.  Three padded records whose greedy bound IS their optimal score (end gaps and perfect matches),
    aligned in small batches on a handle whose rings other jobs have just used: under an exact bound a short last tile block
    handed phase 2 rows older than its own first row, and the cells of those rows beyond the block's columns had never reached
    the output ring -- phase 2 read what the previous tenant had left there and took a false breakpoint one point under the
    optimum (the record was then dropped).  Fixture: tests/golden/exact_bound_pairs.json.gz (make_exact_bound_pairs.py)."""
    import gzip
    import json
    import os
    fix = json.load(gzip.open(os.path.join(os.path.dirname(__file__), "golden", "exact_bound_pairs.json.gz"), "rt"))
    assert len(fix) >= 3
    h = capi.Handle(0)
    try:
        dirty = [(synth.random_dna(4000 + i, 9000), synth.mutate(synth.random_dna(4000 + i, 9000), 0.03, 4100 + i)) for i in range(40)]
        pairs = [(e["pattern"].encode(), e["text"].encode()) for e in fix]
        ub = h.score_bounds(pairs)
        assert [int(u) for u in ub] == [e["score"] for e in fix]  # the bound is exact on these
        for rep in range(4):
            h.align(dirty)  # large offsets all over the rings
            res = h.align(pairs * 20)
            for j, r in enumerate(res):
                e = fix[j % len(fix)]
                assert r.status == 0 and r.score == e["score"] and r.ops == e["ops"].encode(), (rep, j)
    finally:
        h.close()


def test_phase2_walks_longer_than_the_rows_computed_ahead(gpu, oracle, monkeypatch):
    """Unrelated or rearranged sequences: the antidiagonals of the two directions touch long before any two cells share a
    diagonal, so the overlap loop runs for hundreds of tests -- more than the 2 x 32 the rows computed ahead cover.  Such jobs
    go further rounds of rows computed ahead (the breakpoint so far carried along) instead of the step-by-step kernel; the
    result is the oracle's either way, whatever the number of rounds allowed."""
    items = []
    for i, (a, b) in enumerate([(3000, 3000), (3000, 1000), (5000, 4200), (2500, 2600), (6000, 6000)]):
        items.append((synth.random_dna(900 + i, a), synth.random_dna(950 + i, b)))
    u, v, w = synth.random_dna(31, 4000), synth.random_dna(32, 3000), synth.random_dna(33, 4000)
    items.append((u + v + w, u + w + synth.mutate(v, 0.02, 5)))          # a block moved to the end
    items.append((u + v + v + w, synth.mutate(u + v + w, 0.03, 6)))       # a duplication on one side
    items.append((u + synth.random_dna(34, 2000) + w, u + synth.random_dna(35, 2500) + w))  # unrelated middles
    exp = [oracle.align_biwfa(p, t, None) for p, t in items]
    seen = []
    for rounds in (None, "1", "2"):
        if rounds is None:
            monkeypatch.delenv("WFM_P2_ROUNDS", raising=False)
        else:
            monkeypatch.setenv("WFM_P2_ROUNDS", rounds)
        h = capi.Handle(0)
        try:
            res = h.align(items)
            st = h.stats()
        finally:
            h.close()
        for (rc, ops, sc, _), r in zip(exp, res):
            assert rc == 0 and r.status == 0 and r.score == sc and r.ops == ops
        seen.append((st.p2_again, st.p2_more))
    assert seen[0][0] > 0, seen          # further rounds were taken
    assert seen[1][0] == 0 and seen[1][1] > 0, seen  # one round only: the step kernel finishes those jobs


def test_phase2_work_list_overflow_path(oracle, tmp_path):
    """wfa_p2_overlap_kernel lists the blocks of a round's scans in LDS (3072 entries) and runs what does not fit scan by scan;
    with a list of 4 entries (WFM_P2_WORKCAP, read once per process: a process of its own) nearly every round overflows.  Repeat
    units (no maximum prunes a row of a direction that has crossed the text), unrelated pairs and ordinary ones."""
    import subprocess
    import sys
    import textwrap
    script = tmp_path / "p2cap.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {str(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))!r})
        from wfmash_amd import capi, synth
        from oracle import pyoracle as O
        items = []
        for i in range(6):
            unit = synth.random_dna(910 + i, 700)
            p = synth.random_dna(920 + i, 3000) + unit * 6 + synth.random_dna(930 + i, 3000)
            t = synth.mutate(synth.random_dna(920 + i, 3000) + unit * 4 + synth.random_dna(930 + i, 3000), 0.03, 9400 + i)
            items.append((p, t))
        for i in range(6):
            p = synth.random_dna(940 + i, 9000)
            items.append((p, synth.mutate(p, 0.06, 9500 + i)))
        items.append((synth.random_dna(950, 5000), synth.random_dna(951, 5200)))
        h = capi.Handle(0)
        res = h.align(items)
        bad = 0
        for (p, t), r in zip(items, res):
            rc, ops, sc, _ = O.align_biwfa(p, t)
            assert rc == 0 and r.status == 0 and r.score == sc, (len(p), len(t), r.status, r.score, sc)
            bad += r.ops != ops
        h.close()
        assert bad == 0, bad
        print("P2CAP_OK")
    """))
    env = dict(__import__("os").environ, WFM_P2_WORKCAP="4")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "P2CAP_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_cells_the_rooflines_count_against_the_oracles_own_count(oracle, monkeypatch):
    """The numerator of every roofline figure is a count of (score, diagonal) cells made by the host in closed form
    (wfa_host.hip: h_cells_sum over the rows' ranges [max(-pl, -s), min(tl, s)] cut by the score bounds).  Held here against
    the oracle's own count (wfo_stats_t.cells: the width of every wavefront WFA2-lib's recursion computes, after trimming):
    with the bounds switched off -- no hints, no walk, children without their parents' scores -- the device computes the
    reference's rows, and its UNIQUE cells (the block in which a job's wavefronts meet is computed twice and counted once)
    must equal the oracle's within the tolerances written here; with the bounds on it computes fewer.
    Two inputs: C3-sized pairs (the bench line's workload: rows 16 k diagonals wide; measured 1.022 of the oracle's count, so the
    roofline's numerator overstates the reference's work by that much) and short mixed problems (measured 1.078), where what the device computes beyond the reference's rows weighs
    more: phase 2 computes its rows ahead in rounds of 32 where the reference stops at the row that ends its loop, a leaf's
    rows grow by a diagonal per score on either side where the reference trims NULL ends."""
    small = _pairs(41, 10, [2500, 6000, 12000], [0.02, 0.05, 0.1])
    small = [(p, t) for p, t in small if len(p) > 1000 and len(t) > 1000]
    for i in range(4):
        small.append(_padded_record(1200 + i, 8000 + 1000 * i, 1e-3, 1e-4, pad_t=700))
    big = synth.pairs("C3", n_pairs=2)

    def run(items, env):
        for k in ("WFM_SCORE_HINT", "WFM_BOUND", "WFM_SUB_SLACK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = capi.Handle(0)
        try:
            res = h.align(items)
            st = h.stats()
            assert all(r.status == 0 for r in res)
            return int(st.cells), int(st.cells - st.cells_tile + st.cells_tile_unique), int(st.cells_tile_unique)
        finally:
            h.close()

    for name, items, tol in (("C3-sized", big, 1.03), ("short mixed", small, 1.10)):
        want = 0
        for p, t in items:
            rc, ops, sc, st = oracle.align_biwfa(p, t)
            assert rc == 0
            want += int(st.cells)
        total_u, unique_u, tile_u = run(items, {"WFM_SCORE_HINT": "0", "WFM_BOUND": "0", "WFM_SUB_SLACK": str(1 << 28)})
        total_b, unique_b, tile_b = run(items, {})
        print(f"cells, {name}: oracle {want}; device without bounds {unique_u} unique ({unique_u / want:.4f} of the oracle's; tile kernel {tile_u}) / "
              f"{total_u} computed; with bounds {unique_b} unique / {total_b} computed")
        assert want <= unique_u <= tol * want, (name, want, unique_u)
        # (round 6: a root whose score nobody knows finds the block in which its directions meet with one maximum per block, runs it again
        # with per-score maxima and then up to the meeting point -- three runs of that block instead of two, which weighs on problems of a
        # handful of blocks like these: 1.32 measured on the short mixed set, 1.25 before)
        assert unique_u <= total_u <= 1.35 * unique_u
        assert unique_b <= unique_u


def test_problem_flags_name_the_rare_paths(gpu, oracle):
    """wfm_get_problem_flags: the diagnostic channel the sampled parity checks of bench.py and tests/test_configs_gpu.py draw
    their strata from.  A problem with an N must carry WFM_PF_BYTE_KERNEL, a patch that overflows 256 WFM_PF_BASE_RETRY, a
    root whose hint is far too small WFM_PF_ROOT_AGAIN -- and each of them still gives the oracle's CIGAR."""
    p = synth.random_dna(77, 6000)
    t = synth.mutate(p, 0.03, 78)
    pn = p[:3000] + b"N" * 7 + p[3007:]
    items = [(p, t), (pn, t)]
    res = gpu.align(items)
    fl = gpu.problem_flags(2)
    assert len(fl) == 2 and not (fl[0] & capi.WFM_PF_BYTE_KERNEL) and (fl[1] & capi.WFM_PF_BYTE_KERNEL)
    for (a, b), r in zip(items, res):
        rc, ops, sc, _ = oracle.align_biwfa(a, b)
        assert r.status == 0 and r.ops == ops
    # an ends-free patch of two unrelated kilobases: its score passes the first budget of 256
    a, b = synth.random_dna(81, 1500), synth.random_dna(82, 1400)
    items = [(a, b, capi.WFM_MODE_ENDSFREE, len(a), 0, len(b), 0), (a[:40], a[:40], capi.WFM_MODE_ENDSFREE, 40, 0, 40, 0)]
    res = gpu.align(items)
    fl = gpu.problem_flags(2)
    assert (fl[0] & capi.WFM_PF_BASE_RETRY) and not (fl[1] & capi.WFM_PF_BASE_RETRY)
    rc, ops, sc, _ = oracle.align_endsfree(a, len(a), 0, b, len(b), 0)
    assert res[0].status == 0 and res[0].ops == ops
    # a root with a hint far below its score runs again without it
    t2, q2 = _padded_record(1300, 9000, 5e-3, 1e-4, pad_t=800)
    res = gpu.align([(t2, q2, capi.WFM_MODE_END2END_BIWFA, 0, 0, 0, 0, 300)])
    fl = gpu.problem_flags(1)
    rc, ops, sc, _ = oracle.align_biwfa(t2, q2)
    assert sc > 400 and (fl[0] & capi.WFM_PF_ROOT_AGAIN)
    assert res[0].status == 0 and res[0].ops == ops


def test_legacy_reads_fixture_on_the_device(gpu, oracle):
    """The only CIGARs the reference tree holds for this path (test/data/regression/reads.255bps.paf over data/reads.255bps.fa.gz: three
    records of a writer that predates the current one -- they begin and end with indels, so they pin nothing byte for byte) as a
    plausibility check of the HIP path: the device's alignment of every forward record's spans must equal the oracle's op for op, and its
    score must not be worse than the legacy CIGAR's (tests/test_oracle_wfa.py::test_legacy_reads_fixture_is_plausible is the CPU twin)."""
    import gzip
    import os
    import re
    here = os.path.dirname(__file__)
    fa = os.path.join(here, "golden", "reads.255bps.fa.gz")
    paf = os.path.join(here, "golden", "reads.255bps.paf")
    if not (os.path.exists(fa) and os.path.exists(paf)):
        pytest.skip("fixture not present")
    seqs, name = {}, None
    for line in gzip.open(fa, "rt"):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = ""
        elif name:
            seqs[name] += line.upper()
    items, legacy = [], []
    for line in open(paf):
        f = line.rstrip("\n").split("\t")
        if f[4] != "+":
            continue
        query = seqs[f[0]][int(f[2]):int(f[3])].encode()
        target = seqs[f[5]][int(f[7]):int(f[8])].encode()
        cg = [x for x in f if x.startswith("cg:Z:")][0][5:]
        v = h = 0
        fixed = bytearray()
        for cnt, op in re.findall(r"(\d+)([=XIDM])", cg):
            for _ in range(int(cnt)):
                if op in "=MX":
                    fixed.append(ord("M") if target[v] == query[h] else ord("X")); v += 1; h += 1
                elif op == "I":
                    fixed.append(ord("I")); h += 1
                else:
                    fixed.append(ord("D")); v += 1
        assert v == len(target) and h == len(query)
        items.append((target, query))
        legacy.append(bytes(fixed))
    assert items
    res = gpu.align(items)
    for (p, t), r, lg in zip(items, res, legacy):
        rc, ops, sc, _ = oracle.align_biwfa(p, t)
        assert rc == 0 and r.status == 0 and r.ops == ops and r.score == sc
        assert r.score <= oracle.ops_score(lg)
