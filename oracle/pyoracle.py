"""ctypes loader for the CPU parity oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (wfmash_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Penalties(C.Structure):
    _fields_ = [("x", C.c_int32), ("o1", C.c_int32), ("e1", C.c_int32),
                ("o2", C.c_int32), ("e2", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("cells", C.c_uint64), ("extend_bases", C.c_uint64),
                ("bialign_calls", C.c_uint32), ("base_calls", C.c_uint32),
                ("max_depth", C.c_uint32)]


class Breakpoint(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "score", "score_forward", "score_reverse", "k_forward", "k_reverse",
        "offset_forward", "offset_reverse", "component")]


DEFAULT_PEN = (5, 8, 2, 24, 1)  # parse_args.hpp:290-294
COMP = {"M": 0, "I1": 1, "I2": 2, "D1": 3, "D2": 4}


def build(force=False):
    """Compile the oracle shared objects (gcc only)."""
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_wfa.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        cp, ci, pi = C.c_char_p, C.c_int, C.POINTER(C.c_int)
        PP, SP = C.POINTER(Penalties), C.POINTER(Stats)
        L.wfo_dp_score.restype = C.c_int64
        L.wfo_dp_score.argtypes = [cp, ci, cp, ci, PP]
        L.wfo_dp_score_endsfree.restype = C.c_int64
        L.wfo_dp_score_endsfree.argtypes = [cp, ci, cp, ci, PP, ci, ci, ci, ci]
        for name in ("wfo_align_end2end_biwfa", "wfo_align_end2end_uni"):
            f = getattr(L, name)
            f.restype = ci
            f.argtypes = [cp, ci, cp, ci, PP, C.c_char_p, pi, pi, SP]
        L.wfo_align_endsfree.restype = ci
        L.wfo_align_endsfree.argtypes = [cp, ci, ci, ci, cp, ci, ci, ci, PP, C.c_char_p, pi, pi, SP]
        L.wfo_align_end2end_comp.restype = ci
        L.wfo_align_end2end_comp.argtypes = [cp, ci, cp, ci, PP, ci, ci, C.c_char_p, pi, pi, SP]
        L.wfo_find_breakpoint.restype = ci
        L.wfo_find_breakpoint.argtypes = [cp, ci, cp, ci, PP, ci, ci, C.POINTER(Breakpoint), SP]
        L.wfo_find_breakpoint_rounds.restype = ci
        L.wfo_find_breakpoint_rounds.argtypes = [cp, ci, cp, ci, PP, ci, ci, ci, ci, C.POINTER(Breakpoint), C.POINTER(ci), SP]
        L.wfo_find_breakpoint_bounded.restype = ci
        L.wfo_find_breakpoint_bounded.argtypes = [cp, ci, cp, ci, PP, ci, ci, ci, C.POINTER(Breakpoint), SP]
        L.wfo_ops_score.restype = C.c_int64
        L.wfo_ops_score.argtypes = [cp, ci, PP]
        L.wfo_ops_check.restype = ci
        L.wfo_ops_check.argtypes = [cp, ci, cp, ci, cp, ci]
        L.wfo_align_batch_biwfa.restype = ci
        L.wfo_align_batch_biwfa.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            ci, PP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, ci, SP]
        _LIB = L
    return _LIB


def _pen(pen):
    return Penalties(*(pen or DEFAULT_PEN))


def dp_score(pattern: bytes, text: bytes, pen=None) -> int:
    p = _pen(pen)
    return lib().wfo_dp_score(pattern, len(pattern), text, len(text), C.byref(p))


def dp_score_endsfree(pattern: bytes, text: bytes, pbf, pef, tbf, tef, pen=None) -> int:
    p = _pen(pen)
    return lib().wfo_dp_score_endsfree(pattern, len(pattern), text, len(text), C.byref(p), pbf, pef, tbf, tef)


def _run(fn, pattern, text, pen, *extra_mid):
    p = _pen(pen)
    buf = C.create_string_buffer(len(pattern) + len(text) + 2)
    nops, score, st = C.c_int(0), C.c_int(0), Stats()
    rc = fn(pattern, len(pattern), text, len(text), C.byref(p), *extra_mid, buf, C.byref(nops), C.byref(score), C.byref(st))
    return rc, buf.raw[:nops.value], score.value, st


def align_biwfa(pattern: bytes, text: bytes, pen=None):
    """alignEnd2End, MemoryUltralow.  Returns (status, ops, score, stats)."""
    return _run(lib().wfo_align_end2end_biwfa, pattern, text, pen)


def align_uni(pattern: bytes, text: bytes, pen=None):
    return _run(lib().wfo_align_end2end_uni, pattern, text, pen)


def align_comp(pattern: bytes, text: bytes, comp_begin, comp_end, pen=None):
    return _run(lib().wfo_align_end2end_comp, pattern, text, pen, comp_begin, comp_end)


def align_endsfree(pattern: bytes, pbf, pef, text: bytes, tbf, tef, pen=None):
    p = _pen(pen)
    buf = C.create_string_buffer(len(pattern) + len(text) + 2)
    nops, score, st = C.c_int(0), C.c_int(0), Stats()
    rc = lib().wfo_align_endsfree(pattern, len(pattern), pbf, pef, text, len(text), tbf, tef,
                                  C.byref(p), buf, C.byref(nops), C.byref(score), C.byref(st))
    return rc, buf.raw[:nops.value], score.value, st


def find_breakpoint(pattern: bytes, text: bytes, comp_begin=0, comp_end=0, pen=None):
    p = _pen(pen)
    bp, st = Breakpoint(), Stats()
    rc = lib().wfo_find_breakpoint(pattern, len(pattern), text, len(text), C.byref(p), comp_begin, comp_end,
                                   C.byref(bp), C.byref(st))
    return rc, bp, st


def find_breakpoint_bounded(pattern: bytes, text: bytes, sub: int, comp_begin=0, comp_end=0, pen=None):
    """the product's form of the breakpoint search under a bound of the score (rows cut to what can stay under it)"""
    p = _pen(pen)
    bp, st = Breakpoint(), Stats()
    rc = lib().wfo_find_breakpoint_bounded(pattern, len(pattern), text, len(text), C.byref(p), comp_begin, comp_end, sub,
                                           C.byref(bp), C.byref(st))
    return rc, bp, st


def find_breakpoint_rounds(pattern: bytes, text: bytes, tests_per_round: int, sub: int = -1, comp_begin=0, comp_end=0, pen=None):
    """the breakpoint search with its overlap loop cut every tests_per_round tests the way the product's phase 2 is (the
    breakpoint so far set aside, the next round seeded with its score); sub < 0: no score bound.  -> rc, bp, rounds"""
    p = _pen(pen)
    bp, st, rounds = Breakpoint(), Stats(), C.c_int(0)
    rc = lib().wfo_find_breakpoint_rounds(pattern, len(pattern), text, len(text), C.byref(p), comp_begin, comp_end, sub, tests_per_round,
                                          C.byref(bp), C.byref(rounds), C.byref(st))
    return rc, bp, rounds.value


def ops_score(ops: bytes, pen=None) -> int:
    p = _pen(pen)
    return lib().wfo_ops_score(ops, len(ops), C.byref(p))


def ops_check(ops: bytes, pattern: bytes, text: bytes) -> int:
    return lib().wfo_ops_check(ops, len(ops), pattern, len(pattern), text, len(text))


def align_batch_biwfa(patterns, texts, pen=None, nthreads=0):
    """Batch BiWFA on the host cores.  Returns (ops list, scores, Stats)."""
    n = len(patterns)
    blob = b"".join(patterns) + b"".join(texts)
    seqs = np.frombuffer(blob, dtype=np.uint8)
    plen = np.array([len(s) for s in patterns], dtype=np.int32)
    tlen = np.array([len(s) for s in texts], dtype=np.int32)
    poff = np.zeros(n, dtype=np.int64)
    toff = np.zeros(n, dtype=np.int64)
    if n:
        poff[1:] = np.cumsum(plen[:-1])
        toff[0] = plen.sum()
        toff[1:] = toff[0] + np.cumsum(tlen[:-1])
    cap = plen.astype(np.int64) + tlen + 2
    ooff = np.zeros(n, dtype=np.int64)
    if n:
        ooff[1:] = np.cumsum(cap[:-1])
    arena = np.zeros(int(cap.sum()) + 8, dtype=np.uint8)
    nops = np.zeros(n, dtype=np.int32)
    scores = np.zeros(n, dtype=np.int32)
    st = Stats()
    p = _pen(pen)
    failed = lib().wfo_align_batch_biwfa(seqs.ctypes.data, poff.ctypes.data, plen.ctypes.data,
                                         toff.ctypes.data, tlen.ctypes.data, n, C.byref(p),
                                         arena.ctypes.data, ooff.ctypes.data, nops.ctypes.data,
                                         scores.ctypes.data, nthreads, C.byref(st))
    ops = [arena[ooff[i]:ooff[i] + nops[i]].tobytes() if nops[i] >= 0 else None for i in range(n)]
    return ops, scores, st, failed
