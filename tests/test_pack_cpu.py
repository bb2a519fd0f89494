"""The 2-bit packed extension of the tile kernel (wfmash_amd/csrc/wfa_pack.h) without a GPU: the host model that shares the
kernel's primitives -- the word layout of the mirror, 16 / 32 bases from any base offset, the first difference of two packed
words, the stages (16 bases, 64 more, then 32 at a time) -- against a byte-wise comparison, at every alignment of the two
sequences inside their buffers and across word boundaries."""
import ctypes as C
import random

import numpy as np

from wfmash_amd import capi


def _lib():
    L = capi.load()
    L.wfmh_test_packed_lce.restype = C.c_int
    L.wfmh_test_packed_lce.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]
    L.wfmh_test_is_acgt.restype = C.c_int
    L.wfmh_test_is_acgt.argtypes = [C.c_char_p, C.c_int64]
    return L


def _lce(p, t, v, h, maxn):
    n = 0
    while n < maxn and p[v + n] == t[h + n]:
        n += 1
    return n


def test_packed_runs_equal_bytewise_runs():
    L = _lib()
    rng = random.Random(4)
    for trial in range(400):
        n = rng.choice([40, 100, 300, 1200])
        p = bytes(rng.choice(b"ACGT") for _ in range(n))
        t = bytearray(p)
        for _ in range(rng.choice([0, 1, 3, 10])):  # a few differences: runs of every length, also past 80 bases (the tail's rounds)
            t[rng.randrange(n)] = rng.choice(b"ACGT")
        t = bytes(t)
        sp, st = rng.randrange(0, 40), rng.randrange(0, 40)  # where the sequences begin in their buffers: every alignment mod 16
        bufp = bytes(rng.choice(b"ACGT") for _ in range(sp)) + p + b"\0" * 64
        buft = bytes(rng.choice(b"ACGT") for _ in range(st)) + t + b"\0" * 64
        for _ in range(12):
            v, h = rng.randrange(n), rng.randrange(n)
            if rng.random() < 0.6:
                h = v  # on the diagonal the sequences share: long runs
            maxn = min(n - v, n - h)
            if rng.random() < 0.3:
                maxn = rng.randrange(0, maxn + 1)
            got = L.wfmh_test_packed_lce(bufp, len(bufp), sp, buft, len(buft), st, v, h, maxn)
            assert got == _lce(p, t, v, h, maxn), (trial, n, sp, st, v, h, maxn, got)


def test_runs_that_end_at_every_position_of_a_word():
    L = _lib()
    base = b"ACGT" * 64
    for start in range(16):
        for cut in range(0, 130):
            t = bytearray(base)
            if cut < len(t):
                t[cut] = ord("A") if base[cut] != ord("A") else ord("C")
            bufp = b"G" * start + base + b"\0" * 64
            buft = b"T" * (15 - start) + bytes(t) + b"\0" * 64
            got = L.wfmh_test_packed_lce(bufp, len(bufp), start, buft, len(buft), 15 - start, 0, 0, len(base))
            assert got == min(cut, len(base)), (start, cut, got)


def test_which_problems_are_packed():
    L = _lib()
    assert L.wfmh_test_is_acgt(b"ACGTTGCA", 8) == 1
    for bad in (b"ACGN", b"acgt", b"ACG\0", b"ACGU", b"ACG-"):
        assert L.wfmh_test_is_acgt(bad, len(bad)) == 0
    # the four codes are distinct (equality of codes == equality of bases for A C G T)
    codes = {(c >> 1) & 3 for c in b"ACGT"}
    assert len(codes) == 4
