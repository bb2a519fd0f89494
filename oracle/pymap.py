"""ctypes bindings of the map-path oracle (liboracle_map.so) and, when it has been built in
this container, of the reference's own code (oracle/_ref/libref_map.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MINMER = np.dtype([("hash", "<u8"), ("wpos", "<i8"), ("wpos_end", "<i8"), ("seqId", "<i4"), ("strand", "<i2"), ("pad", "<i2")])
_libs = {}


def _load(which):
    if which in _libs:
        return _libs[which]
    path = os.path.join(_HERE, "liboracle_map.so") if which == "oracle" else os.path.join(_HERE, "_ref", "libref_map.so")
    if not os.path.exists(path):
        if which == "oracle":
            import subprocess
            subprocess.check_call(["make", "-s", "-C", _HERE])
        else:
            _libs[which] = None
            return None
    L = C.CDLL(path)
    pre = "mo_" if which == "oracle" else "ref_"
    getattr(L, pre + "get_hash").restype = C.c_uint64
    getattr(L, pre + "get_hash").argtypes = [C.c_char_p, C.c_int]
    getattr(L, pre + "sketch_sequence").restype = C.c_int
    getattr(L, pre + "sketch_sequence").argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_int]
    if which == "oracle":
        L.mo_hash_kmers.restype = None
        L.mo_hash_kmers.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    else:
        L.ref_add_minmers.restype = C.c_int64
        L.ref_add_minmers.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_int64]
        L.ref_upper_valid.restype = None
        L.ref_upper_valid.argtypes = [C.c_char_p, C.c_int64]
        L.ref_revcomp.restype = None
        L.ref_revcomp.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        if hasattr(L, "ref_group_minhash"):
            L.ref_group_minhash.restype = C.c_int64
            L.ref_group_minhash.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_int64]
            L.ref_streaming_minhash.restype = C.c_int64
            L.ref_streaming_minhash.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64]
    _libs[which] = L
    return L


def have_ref():
    return _load("ref") is not None


def get_hash(kmer: bytes, which="oracle") -> int:
    L = _load(which)
    return getattr(L, ("mo_" if which == "oracle" else "ref_") + "get_hash")(kmer, len(kmer))


def hash_kmers(seq: bytes, k: int):
    n = max(len(seq) - k + 1, 0)
    h = np.zeros(n, dtype=np.uint64)
    st = np.zeros(n, dtype=np.int8)
    _load("oracle").mo_hash_kmers(seq, len(seq), k, h.ctypes.data, st.ctypes.data)
    return h, st


def sketch_sequence(seq: bytes, k: int, s: int, seq_id: int = 0, which="oracle"):
    L = _load(which)
    out = np.zeros(s + 1, dtype=MINMER)
    buf = C.create_string_buffer(seq, len(seq))  # the reference upper-cases in place
    n = getattr(L, ("mo_" if which == "oracle" else "ref_") + "sketch_sequence")(buf, len(seq), k, s, seq_id, out.ctypes.data, s + 1)
    return out[:n]


def ref_add_minmers(seq: bytes, k: int, w: int, s: int, seq_id: int = 0, cap=None):
    L = _load("ref")
    cap = int(cap) if cap else 4 * len(seq) + 64
    out = np.zeros(cap, dtype=MINMER)
    buf = C.create_string_buffer(seq, len(seq))
    n = L.ref_add_minmers(buf, len(seq), k, w, s, seq_id, out.ctypes.data, cap)
    assert n <= cap, (n, cap)
    return out[:n]


def ref_park_stderr(on: bool):
    """stderr on /dev/null for a whole multi-threaded leg of ref_add_minmers calls (the reference's ProgressMeter prints from threads of its own)."""
    L = _load("ref")
    if hasattr(L, "ref_park_stderr"):
        L.ref_park_stderr(1 if on else 0)


def ref_group_minhash(seqs, groups, k: int, sketch_size: int, want_group: int):
    """The reference's GroupedStreamingMinHash over `seqs` (list of bytes), sketch of group `want_group` ascending."""
    L = _load("ref")
    n = len(seqs)
    ptrs = (C.c_char_p * n)(*seqs)
    lens = np.array([len(x) for x in seqs], dtype=np.int64)
    grp = np.ascontiguousarray(groups, dtype=np.int32)
    out = np.zeros(sketch_size, dtype=np.uint64)
    m = L.ref_group_minhash(ptrs, lens.ctypes.data, grp.ctypes.data, n, k, sketch_size, want_group, out.ctypes.data, sketch_size)
    return out[:m]


def ref_streaming_minhash(values, sketch_size: int):
    L = _load("ref")
    v = np.ascontiguousarray(values, dtype=np.uint64)
    out = np.zeros(sketch_size, dtype=np.uint64)
    m = L.ref_streaming_minhash(v.ctypes.data, len(v), sketch_size, out.ctypes.data, sketch_size)
    return out[:m]
