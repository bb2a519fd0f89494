"""ctypes bindings of libwfmash_hip.so (the C ABI declared in include/wfmash_hip.h).

This is plumbing for tests and bench.py; the product is the shared library.
There is no CPU fallback: if the library is missing or no gfx950 device is
visible, loading / `Handle()` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WFM_LIB_PATH") or os.path.join(_HERE, "libwfmash_hip.so")  # WFM_LIB_PATH: A/B runs of two builds

WFM_MODE_END2END_BIWFA = 0
WFM_MODE_ENDSFREE = 1
WFM_MODE_END2END_UNI = 2

DEFAULT_PEN = (5, 8, 2, 24, 1)  # parse_args.hpp:290-294
# wfm_get_problem_flags (include/wfmash_hip.h): which of the rarer paths a problem took
WFM_PF_ROOT_AGAIN, WFM_PF_JOB_AGAIN, WFM_PF_BASE_RETRY, WFM_PF_BASE_RETRY2, WFM_PF_BYTE_KERNEL, WFM_PF_P2_ROUNDS, WFM_PF_RING_KERNEL, WFM_PF_BASE_TILES = 1, 2, 4, 8, 16, 32, 64, 128

EXPORTS = [
    "wfm_create", "wfm_destroy", "wfm_last_error", "wfm_device_name",
    "wfm_align_arena_bytes", "wfm_align_batch", "wfm_upload_sequences",
    "wfm_free_sequences", "wfm_align_resident", "wfm_get_stats",
    "wfm_hash_kmers", "wfm_sketch_fragments", "wfm_add_minmers",
    "wfm_index_build", "wfm_index_free", "wfm_index_info", "wfm_index_download",
    "wfm_map_l1", "wfm_map_l2", "wfm_map_fragments", "wfm_minhash_sketch", "wfm_add_minmers_multi",
    "wfm_prefilter_kmers", "wfm_index_build_sequences", "wfm_index_upload",
    "wfm_index_replicate", "wfm_device_count", "wfm_finish_records",
    "wfm_align_batch_rle", "wfm_align_resident_rle", "wfm_free_runs", "wfm_score_bounds", "wfm_get_busy_intervals", "wfm_trim_device_cache", "wfm_map_fragments_ordered", "wfm_map_sequence_cache", "wfm_selftest_dpp", "wfm_selftest_arena_growth", "wfm_set_concurrent_calls", "wfm_get_problem_flags",
]


class Penalties(C.Structure):
    _fields_ = [("x", C.c_int32), ("o1", C.c_int32), ("e1", C.c_int32),
                ("o2", C.c_int32), ("e2", C.c_int32)]


class Problem(C.Structure):
    _fields_ = [("pattern", C.c_char_p), ("plen", C.c_int32),
                ("text", C.c_char_p), ("tlen", C.c_int32),
                ("mode", C.c_int32),
                ("pattern_begin_free", C.c_int32), ("pattern_end_free", C.c_int32),
                ("text_begin_free", C.c_int32), ("text_end_free", C.c_int32),
                ("score_hint", C.c_int32), ("pad_", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("score", C.c_int32),
                ("ops_off", C.c_uint64), ("ops_len", C.c_uint32),
                ("n_runs", C.c_uint32), ("cells", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("cells", C.c_uint64), ("bytes_algorithmic", C.c_uint64),
                ("ms_kernels", C.c_double), ("ms_breakpoint", C.c_double),
                ("ms_base", C.c_double), ("ms_total", C.c_double),
                ("levels", C.c_uint32), ("bp_jobs", C.c_uint32), ("base_jobs", C.c_uint32),
                ("bp_launches", C.c_uint32), ("base_launches", C.c_uint32),
                ("cells_bp", C.c_uint64), ("cells_base", C.c_uint64),
                ("cells_tile", C.c_uint64), ("ms_tile", C.c_double),
                ("tile_launches", C.c_uint32), ("tile_tasks", C.c_uint32),
                ("ms_tile_busy", C.c_double), ("streams", C.c_uint32), ("pad_", C.c_uint32),
                ("cells_tile_unique", C.c_uint64), ("ms_bp_busy", C.c_double), ("ms_base_busy", C.c_double),
                ("ms_any_busy", C.c_double), ("p2_launches", C.c_uint32), ("p2_jobs", C.c_uint32), ("p2_more", C.c_uint32),
                ("p2_again", C.c_uint32)]


class Minmer(C.Structure):
    _fields_ = [("hash", C.c_uint64), ("wpos", C.c_int64), ("wpos_end", C.c_int64),
                ("seqId", C.c_int32), ("strand", C.c_int16), ("pad_", C.c_int16)]


POINT_DTYPE = np.dtype([("pos", "<i8"), ("hash", "<u8"), ("seqId", "<i4"), ("side", "i1"), ("pad_", "V3")])


class IndexInfo(C.Structure):
    _fields_ = [("n_windows", C.c_int64), ("n_kept", C.c_int64), ("n_unique", C.c_int64), ("n_points", C.c_int64),
                ("threshold", C.c_uint64), ("filtered", C.c_int64), ("adjusted", C.c_int32), ("pad_", C.c_int32)]


MINMER_DTYPE = np.dtype([("hash", "<u8"), ("wpos", "<i8"), ("wpos_end", "<i8"),
                         ("seqId", "<i4"), ("strand", "<i2"), ("pad_", "<i2")])
L1_DTYPE = np.dtype([("seqId", "<i4"), ("frag", "<i4"), ("rangeStartPos", "<i8"), ("rangeEndPos", "<i8"),
                     ("intersectionSize", "<i4"), ("pad_", "<i4")])


class MapParams(C.Structure):
    pass  # fields set after L1Params / L2Params are defined


MAPPING_DTYPE = np.dtype([("refSeqId", "<u4"), ("refStartPos", "<u4"), ("queryStartPos", "<u4"), ("blockLength", "<u4"),
                          ("n_merged", "<u4"), ("conservedSketches", "<u4"), ("nucIdentity", "<u2"), ("flags", "u1"),
                          ("kmerComplexity", "u1")])


class L2Params(C.Structure):
    _fields_ = [("window_length", C.c_int32), ("sketch_size", C.c_int32), ("stage1_topANI_filter", C.c_int32), ("pad_", C.c_int32),
                ("keep_table", C.c_void_p), ("ident_table", C.c_void_p), ("cutoff_j", C.c_void_p)]


class L1Params(C.Structure):
    _fields_ = [("window_length", C.c_int32), ("sketch_size", C.c_int32), ("min_hits_cached", C.c_int32),
                ("cached_segment_length", C.c_int32), ("skip_self", C.c_int32), ("skip_prefix", C.c_int32),
                ("lower_triangular", C.c_int32), ("stage1_topANI_filter", C.c_int32), ("stage2_full_scan", C.c_int32),
                ("n_seq", C.c_int32), ("ref_group", C.c_void_p), ("min_hits_by_qsketch", C.c_void_p),
                ("sketch_cutoffs", C.c_void_p), ("n_cutoffs", C.c_int32), ("pad_", C.c_int32)]


MapParams._fields_ = [("kmer_size", C.c_int32), ("kmer_complexity_threshold", C.c_float), ("l1", L1Params), ("l2", L2Params)]

_LIB = None


def load():
    """Load libwfmash_hip.so; raises if it has not been built."""
    global _LIB, LIB_PATH
    if _LIB is not None:
        return _LIB
    # (WFM_LIB: another build of the same library, for A/B runs of two kernels inside one process launch -- scripts/c3_time.py)
    if os.environ.get("WFM_LIB"):
        LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), os.environ["WFM_LIB"]) if not os.path.isabs(os.environ["WFM_LIB"]) else os.environ["WFM_LIB"]
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.wfm_create.restype = C.c_int
    L.wfm_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.wfm_destroy.restype = None
    L.wfm_destroy.argtypes = [vp]
    L.wfm_last_error.restype = C.c_char_p
    L.wfm_last_error.argtypes = [vp]
    L.wfm_device_name.restype = C.c_int
    L.wfm_device_name.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.wfm_align_arena_bytes.restype = C.c_size_t
    L.wfm_align_arena_bytes.argtypes = [C.POINTER(Problem), C.c_size_t]
    L.wfm_align_batch.restype = C.c_int
    L.wfm_align_batch.argtypes = [vp, C.POINTER(Penalties), C.POINTER(Problem), C.c_size_t,
                                  C.POINTER(Result), vp, C.c_size_t]
    L.wfm_upload_sequences.restype = C.c_int
    L.wfm_upload_sequences.argtypes = [vp, C.POINTER(Problem), C.c_size_t, C.POINTER(vp)]
    L.wfm_free_sequences.restype = None
    L.wfm_free_sequences.argtypes = [vp, vp]
    L.wfm_align_resident.restype = C.c_int
    L.wfm_align_resident.argtypes = [vp, C.POINTER(Penalties), vp, C.POINTER(Result), vp, C.c_size_t]
    L.wfm_get_stats.restype = C.c_int
    L.wfm_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.wfm_hash_kmers.restype = C.c_int
    L.wfm_hash_kmers.argtypes = [vp, vp, C.c_int64, C.c_int, vp, vp]
    L.wfm_sketch_fragments.restype = C.c_int
    L.wfm_sketch_fragments.argtypes = [vp, vp, C.c_int64, vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_int32, vp, vp]
    for name in EXPORTS:  # every symbol include/wfmash_hip.h declares must be exported
        getattr(L, name)
    _LIB = L
    return L


class WfmError(RuntimeError):
    pass


class AlignResult:
    __slots__ = ("status", "score", "ops", "n_runs", "cells")

    def __init__(self, status, score, ops, n_runs, cells):
        self.status, self.score, self.ops, self.n_runs, self.cells = status, score, ops, n_runs, cells


def _make_problems(items):
    """items: iterable of (pattern, text) or (pattern, text, mode, pbf, pef, tbf, tef[, score_hint])."""
    items = list(items)
    arr = (Problem * max(len(items), 1))()
    keep = []
    for i, it in enumerate(items):
        p, t = bytes(it[0]), bytes(it[1])
        keep.append((p, t))
        arr[i].pattern, arr[i].plen = p, len(p)
        arr[i].text, arr[i].tlen = t, len(t)
        if len(it) > 2:
            arr[i].mode = it[2]
            if len(it) > 3:
                (arr[i].pattern_begin_free, arr[i].pattern_end_free,
                 arr[i].text_begin_free, arr[i].text_end_free) = it[3:7]
                if len(it) > 7:
                    arr[i].score_hint = it[7]
        else:
            arr[i].mode = WFM_MODE_END2END_BIWFA
    return arr, keep, len(items)


class SeqSet:
    def __init__(self, handle, items):
        self._h = handle
        self.problems, self._keep, self.n = _make_problems(items)
        sp = C.c_void_p()
        rc = handle._L.wfm_upload_sequences(handle._p, self.problems, self.n, C.byref(sp))
        if rc != 0:
            raise WfmError(f"wfm_upload_sequences failed ({rc}): {handle.last_error()}")
        self._p = sp
        self.arena_bytes = handle._L.wfm_align_arena_bytes(self.problems, self.n)
        self.arena = np.zeros(self.arena_bytes + 8, dtype=np.uint8)
        self.results = (Result * max(self.n, 1))()

    def free(self):
        if self._p:
            self._h._L.wfm_free_sequences(self._h._p, self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Index:
    def __init__(self, handle, ptr):
        self._h, self._p = handle, ptr

    def info(self):
        inf = IndexInfo()
        self._h._L.wfm_index_info(self._p, C.byref(inf))
        return inf

    def download(self):
        inf = self.info()
        uh = np.zeros(inf.n_unique, dtype=np.uint64)
        po = np.zeros(inf.n_unique + 1, dtype=np.int64)
        pts = np.zeros(inf.n_points, dtype=POINT_DTYPE)
        mm = np.zeros(inf.n_kept, dtype=MINMER_DTYPE)
        rc = self._h._L.wfm_index_download(self._h._p, self._p, uh.ctypes.data, po.ctypes.data, pts.ctypes.data, mm.ctypes.data)
        if rc != 0:
            raise WfmError(f"wfm_index_download failed ({rc})")
        return uh, po, pts, mm

    def free(self):
        if self._p:
            self._h._L.wfm_index_free(self._h._p, self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Handle:
    """One handle per GPU (wfm_create)."""

    def __init__(self, device=0):
        self._L = load()
        p = C.c_void_p()
        rc = self._L.wfm_create(device, C.byref(p))
        if rc != 0:
            raise WfmError(f"wfm_create(device={device}) failed with {rc}: no usable gfx950 device "
                           "(there is no CPU fallback)")
        self._p = p

    def close(self):
        if self._p:
            self._L.wfm_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return self._L.wfm_last_error(self._p).decode()

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._L.wfm_device_name(self._p, buf, 256)
        return buf.value.decode()

    def stats(self):
        st = Stats()
        self._L.wfm_get_stats(self._p, C.byref(st))
        return st

    def problem_flags(self, n):
        """wfm_get_problem_flags: WFM_PF_* bits of the problems of the handle's last align call."""
        f = self._L.wfm_get_problem_flags
        f.restype = C.c_size_t
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        out = np.zeros(max(n, 1), dtype=np.uint32)
        have = f(self._p, out.ctypes.data, n)
        return out[:min(n, have)]

    def upload(self, items):
        return SeqSet(self, items)

    def align_resident(self, seqset, pen=None, collect=True):
        pn = Penalties(*(pen or DEFAULT_PEN))
        rc = self._L.wfm_align_resident(self._p, C.byref(pn), seqset._p, seqset.results,
                                        seqset.arena.ctypes.data, seqset.arena_bytes)
        if rc < 0:
            raise WfmError(f"wfm_align_resident failed ({rc}): {self.last_error()}")
        if not collect:
            return rc
        return self._collect(seqset)

    @staticmethod
    def _collect(seqset):
        out = []
        a = seqset.arena
        for i in range(seqset.n):
            r = seqset.results[i]
            ops = a[r.ops_off:r.ops_off + r.ops_len].tobytes() if r.status == 0 else None
            out.append(AlignResult(r.status, r.score, ops, r.n_runs, r.cells))
        return out

    def align(self, items, pen=None):
        """items: list of (pattern, text[, mode, pbf, pef, tbf, tef]). Returns [AlignResult]."""
        probs, keep, n = _make_problems(items)
        pn = Penalties(*(pen or DEFAULT_PEN))
        nbytes = self._L.wfm_align_arena_bytes(probs, n)
        arena = np.zeros(nbytes + 8, dtype=np.uint8)
        res = (Result * max(n, 1))()
        rc = self._L.wfm_align_batch(self._p, C.byref(pn), probs, n, res, arena.ctypes.data, nbytes)
        if rc < 0:
            raise WfmError(f"wfm_align_batch failed ({rc}): {self.last_error()}")
        out = []
        for i in range(n):
            r = res[i]
            ops = arena[r.ops_off:r.ops_off + r.ops_len].tobytes() if r.status == 0 else None
            out.append(AlignResult(r.status, r.score, ops, r.n_runs, r.cells))
        return out

    def score_bounds(self, items, pen=None):
        """wfm_score_bounds: per (pattern, text) an upper bound of the end-to-end score, or -1."""
        probs, keep, n = _make_problems(items)
        pn = Penalties(*(pen or DEFAULT_PEN))
        out = np.zeros(max(n, 1), dtype=np.int32)
        f = self._L.wfm_score_bounds
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        rc = f(self._p, C.byref(pn), probs, n, out.ctypes.data)
        if rc < 0:
            raise WfmError(f"wfm_score_bounds failed ({rc}): {self.last_error()}")
        return out[:n]

    def align_rle(self, items, pen=None):
        """wfm_align_batch_rle: the same problems, run-length output.  AlignResult.ops is the list of (length, op) runs
        with op in b'MXID' (adjacent runs never share an op), .n_runs their number; `ops_len` travels as an extra
        attribute on the tuple: returns [(AlignResult, ops_len)]."""
        probs, keep, n = _make_problems(items)
        pn = Penalties(*(pen or DEFAULT_PEN))
        res = (Result * max(n, 1))()
        runs = C.POINTER(C.c_uint32)()
        total = C.c_size_t(0)
        f = self._L.wfm_align_batch_rle
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_size_t)]
        self._L.wfm_free_runs.restype = None
        self._L.wfm_free_runs.argtypes = [C.POINTER(C.c_uint32)]
        rc = f(self._p, C.byref(pn), probs, n, res, C.byref(runs), C.byref(total))
        if rc < 0:
            raise WfmError(f"wfm_align_batch_rle failed ({rc}): {self.last_error()}")
        try:
            arr = np.ctypeslib.as_array(runs, shape=(total.value,)).copy() if total.value else np.zeros(0, dtype=np.uint32)
        finally:
            self._L.wfm_free_runs(runs)
        out = []
        for i in range(n):
            r = res[i]
            ops = None
            if r.status == 0:
                mine = arr[r.ops_off:r.ops_off + r.n_runs]
                ops = [(int(x) >> 2, b"MXID"[int(x) & 3:(int(x) & 3) + 1]) for x in mine]
            out.append((AlignResult(r.status, r.score, ops, r.n_runs, r.cells), int(r.ops_len)))
        return out

    # ---- map path ----
    def hash_kmers(self, seq: bytes, k: int):
        n = max(len(seq) - k + 1, 0)
        hashes = np.zeros(n, dtype=np.uint64)
        strand = np.zeros(n, dtype=np.int8)
        buf = np.frombuffer(seq, dtype=np.uint8)
        rc = self._L.wfm_hash_kmers(self._p, buf.ctypes.data, len(seq), k, hashes.ctypes.data, strand.ctypes.data)
        if rc < 0:
            raise WfmError(f"wfm_hash_kmers failed ({rc}): {self.last_error()}")
        return hashes, strand

    def sketch_fragments(self, seq: bytes, frag_off, frag_len, k: int, s: int, seq_id: int = 0):
        frag_off = np.ascontiguousarray(frag_off, dtype=np.int64)
        frag_len = np.ascontiguousarray(frag_len, dtype=np.int32)
        n = len(frag_off)
        out = np.zeros(n * s, dtype=MINMER_DTYPE)
        cnt = np.zeros(n, dtype=np.int32)
        buf = np.frombuffer(seq, dtype=np.uint8)
        rc = self._L.wfm_sketch_fragments(self._p, buf.ctypes.data, len(seq), frag_off.ctypes.data,
                                          frag_len.ctypes.data, n, k, s, seq_id, out.ctypes.data, cnt.ctypes.data)
        if rc < 0:
            raise WfmError(f"wfm_sketch_fragments failed ({rc}): {self.last_error()}")
        return [out[i * s:i * s + cnt[i]] for i in range(n)]

    def index_build(self, minmers, max_kmer_freq=0.0002):
        """wfm_index_build: device-resident reference index from the concatenated minmer intervals."""
        m = np.ascontiguousarray(minmers, dtype=MINMER_DTYPE)
        L = self._L
        L.wfm_index_build.restype = C.c_int
        L.wfm_index_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.POINTER(C.c_void_p)]
        L.wfm_index_free.restype = None
        L.wfm_index_free.argtypes = [C.c_void_p, C.c_void_p]
        L.wfm_index_info.argtypes = [C.c_void_p, C.POINTER(IndexInfo)]
        L.wfm_index_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        ix = C.c_void_p()
        rc = L.wfm_index_build(self._p, m.ctypes.data, len(m), max_kmer_freq, C.byref(ix))
        if rc != 0:
            raise WfmError(f"wfm_index_build failed ({rc}): {self.last_error()}")
        return Index(self, ix)

    def index_build_sequences(self, seqs, k: int, w: int, s: int, seq_ids=None, threads: int = 1, max_kmer_freq=0.0002):
        """wfm_index_build_sequences: Sketch::build in one call; returns (Index or None, number of minmer intervals)."""
        n = len(seqs)
        ids = np.ascontiguousarray(seq_ids if seq_ids is not None else range(n), dtype=np.int32)
        bufs = [np.frombuffer(x, dtype=np.uint8) for x in seqs]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        lens = np.array([len(x) for x in seqs], dtype=np.int64)
        L = self._L
        L.wfm_index_free.restype = None
        L.wfm_index_free.argtypes = [C.c_void_p, C.c_void_p]
        L.wfm_index_info.argtypes = [C.c_void_p, C.POINTER(IndexInfo)]
        L.wfm_index_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        f = L.wfm_index_build_sequences
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                      C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        ix = C.c_void_p()
        nw = C.c_int64(0)
        rc = f(self._p, ptrs, lens.ctypes.data, ids.ctypes.data, n, k, w, s, threads, max_kmer_freq, C.byref(ix), C.byref(nw))
        if rc != 0:
            raise WfmError(f"wfm_index_build_sequences failed ({rc}): {self.last_error()}")
        return (Index(self, ix) if ix.value else None), nw.value

    def map_l1(self, index, qsketch, qcount, q_seq_id, q_len, q_active, s, params, ref_group):
        """wfm_map_l1: L1 candidate regions of a batch of query fragments.  params: the dict of oracle/map_l1.py."""
        nfrag = len(qcount)
        q = np.ascontiguousarray(qsketch, dtype=MINMER_DTYPE)
        qc = np.ascontiguousarray(qcount, dtype=np.int32)
        qs = np.ascontiguousarray(q_seq_id, dtype=np.int32)
        ql = np.ascontiguousarray(q_len, dtype=np.int32)
        qa = np.ascontiguousarray(q_active, dtype=np.uint8)
        rg = np.ascontiguousarray(ref_group, dtype=np.int32)
        mh = np.ascontiguousarray(params["min_hits_by_qsketch"], dtype=np.int32)
        sc = np.ascontiguousarray(params["sketch_cutoffs"], dtype=np.int32)
        assert len(q) == nfrag * s and len(mh) == params["sketch_size"] + 1
        P = L1Params(params["window_length"], params["sketch_size"], params["min_hits_cached"], params["cached_segment_length"],
                     int(params["skip_self"]), int(params["skip_prefix"]), int(params["lower_triangular"]),
                     int(params["stage1_topani"]), int(params["stage2_full_scan"]), len(rg), rg.ctypes.data, mh.ctypes.data,
                     sc.ctypes.data, len(sc), 0)
        f = self._L.wfm_map_l1
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                      C.POINTER(L1Params), C.c_void_p, C.c_int64]
        cap = 1 << 16
        while True:
            out = np.zeros(cap, dtype=L1_DTYPE)
            n = f(self._p, index._p, q.ctypes.data, qc.ctypes.data, qs.ctypes.data, ql.ctypes.data, qa.ctypes.data, nfrag, s,
                  C.byref(P), out.ctypes.data, cap)
            if n < 0:
                raise WfmError(f"wfm_map_l1 failed ({n}): {self.last_error()}")
            if n <= cap:
                return out[:n]
            cap = int(n)

    def map_fragments(self, index, seq: bytes, frag_off, frag_seq_id, k, p1, p2, ref_group, kc_threshold=0.0):
        """wfm_map_fragments: sketch -> L1 -> L2 for fragments of p1["window_length"] bases of one buffer."""
        off = np.ascontiguousarray(frag_off, dtype=np.int64)
        sid = np.ascontiguousarray(frag_seq_id, dtype=np.int32)
        rg = np.ascontiguousarray(ref_group, dtype=np.int32)
        mh = np.ascontiguousarray(p1["min_hits_by_qsketch"], dtype=np.int32)
        sc = np.ascontiguousarray(p1["sketch_cutoffs"], dtype=np.int32)
        keep = np.ascontiguousarray(p2["keep_table"], dtype=np.uint8)
        ident = np.ascontiguousarray(p2["ident_table"], dtype=np.uint16)
        cut = np.ascontiguousarray(p2["cutoff_j"], dtype=np.float64)
        P = MapParams()
        P.kmer_size = k
        P.kmer_complexity_threshold = kc_threshold
        P.l1 = L1Params(p1["window_length"], p1["sketch_size"], p1["min_hits_cached"], p1["cached_segment_length"],
                        int(p1["skip_self"]), int(p1["skip_prefix"]), int(p1["lower_triangular"]), int(p1["stage1_topani"]),
                        int(p1["stage2_full_scan"]), len(rg), rg.ctypes.data, mh.ctypes.data, sc.ctypes.data, len(sc), 0)
        P.l2 = L2Params(p2["window_length"], p2["sketch_size"], int(p2["stage1_topani"]), 0, keep.ctypes.data, ident.ctypes.data,
                        cut.ctypes.data)
        buf = np.frombuffer(seq, dtype=np.uint8)
        f = self._L.wfm_map_fragments
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(MapParams),
                      C.c_void_p, C.c_void_p, C.c_int64]
        cap = 1 << 16
        while True:
            out = np.zeros(cap, dtype=MAPPING_DTYPE)
            frag = np.zeros(cap, dtype=np.int32)
            n = f(self._p, index._p, buf.ctypes.data, len(seq), off.ctypes.data, sid.ctypes.data, len(off), C.byref(P),
                  out.ctypes.data, frag.ctypes.data, cap)
            if n < 0:
                raise WfmError(f"wfm_map_fragments failed ({n}): {self.last_error()}")
            if n <= cap:
                return out[:n], frag[:n]
            cap = int(n)

    def minhash_sketch(self, seq: bytes, k: int = 21, sketch_size: int = 4096):
        """wfm_minhash_sketch: bottom-sketch_size canonical k-mer hashes (with multiplicity) of one sequence."""
        out = np.zeros(sketch_size, dtype=np.uint64)
        buf = np.frombuffer(seq, dtype=np.uint8)
        f = self._L.wfm_minhash_sketch
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        n = f(self._p, buf.ctypes.data, len(seq), k, sketch_size, out.ctypes.data)
        if n < 0:
            raise WfmError(f"wfm_minhash_sketch failed ({n}): {self.last_error()}")
        return out[:n]

    def map_l2(self, index, qsketch, qcount, q_len, q_kc, s, cands, params):
        """wfm_map_l2: mappings (MAPPING_DTYPE) + fragment ids for a batch of L1 candidates.
        params: dict(window_length, sketch_size, stage1_topani, keep_table, ident_table, cutoff_j)."""
        nfrag = len(qcount)
        q = np.ascontiguousarray(qsketch, dtype=MINMER_DTYPE)
        qc = np.ascontiguousarray(qcount, dtype=np.int32)
        ql = np.ascontiguousarray(q_len, dtype=np.int32)
        kc = np.ascontiguousarray(q_kc, dtype=np.uint8)
        cd = np.ascontiguousarray(cands, dtype=L1_DTYPE)
        keep = np.ascontiguousarray(params["keep_table"], dtype=np.uint8)
        ident = np.ascontiguousarray(params["ident_table"], dtype=np.uint16)
        cut = np.ascontiguousarray(params["cutoff_j"], dtype=np.float64)
        S1 = params["sketch_size"] + 1
        assert keep.size == S1 * S1 and ident.size == S1 * S1 and cut.size == S1 and len(q) == nfrag * s
        P = L2Params(params["window_length"], params["sketch_size"], int(params["stage1_topani"]), 0, keep.ctypes.data,
                     ident.ctypes.data, cut.ctypes.data)
        f = self._L.wfm_map_l2
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                      C.c_int64, C.POINTER(L2Params), C.c_void_p, C.c_void_p, C.c_int64]
        cap = 1 << 16
        while True:
            out = np.zeros(cap, dtype=MAPPING_DTYPE)
            frag = np.zeros(cap, dtype=np.int32)
            n = f(self._p, index._p, q.ctypes.data, qc.ctypes.data, ql.ctypes.data, kc.ctypes.data, nfrag, s, cd.ctypes.data, len(cd),
                  C.byref(P), out.ctypes.data, frag.ctypes.data, cap)
            if n < 0:
                raise WfmError(f"wfm_map_l2 failed ({n}): {self.last_error()}")
            if n <= cap:
                return out[:n], frag[:n]
            cap = int(n)

    def add_minmers_multi(self, seqs, k: int, w: int, s: int, seq_ids=None, threads: int = 1, cap=None):
        """wfm_add_minmers_multi: minmer intervals of several sequences (GPU hashing, threaded host winnowing);
        returns one array per sequence."""
        n = len(seqs)
        ids = np.ascontiguousarray(seq_ids if seq_ids is not None else range(n), dtype=np.int32)
        bufs = [np.frombuffer(x, dtype=np.uint8) for x in seqs]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        lens = np.array([len(x) for x in seqs], dtype=np.int64)
        cap = int(cap) if cap else 4 * int(lens.sum()) + 64
        out = np.zeros(cap, dtype=MINMER_DTYPE)
        counts = np.zeros(n, dtype=np.int64)
        f = self._L.wfm_add_minmers_multi
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                      C.c_void_p]
        tot = f(self._p, ptrs, lens.ctypes.data, ids.ctypes.data, n, k, w, s, threads, out.ctypes.data, cap, counts.ctypes.data)
        if tot < 0:
            raise WfmError(f"wfm_add_minmers_multi failed ({tot}): {self.last_error()}")
        offs = np.concatenate([[0], np.cumsum(counts)])
        return [out[offs[i]:offs[i + 1]] for i in range(n)]

    def finish_records(self, raw, w: int):
        """wfm_finish_records: the closing steps of addMinmers on the device (map_finish.hip) on raw interval records;
        returns (records, recursion levels, ranges heap-sorted on the host)."""
        raw = np.ascontiguousarray(raw, dtype=MINMER_DTYPE)
        f = self._L.wfm_finish_records
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        cap = 4 * len(raw) + 64
        lv, hp = C.c_int32(0), C.c_int32(0)
        while True:
            out = np.zeros(cap, dtype=MINMER_DTYPE)
            n = f(self._p, raw.ctypes.data, len(raw), w, out.ctypes.data, cap, C.byref(lv), C.byref(hp))
            if n < 0:
                raise WfmError(f"wfm_finish_records failed ({n}): {self.last_error()}")
            if n <= cap:
                return out[:n], lv.value, hp.value
            cap = int(n)

    def prefilter_kmers(self, seq: bytes, k: int, w: int, s: int, c_factor: float = 4.0):
        """wfm_prefilter_kmers: the k-mers the host winnowing gets to see; returns (pos, hash, strand)."""
        cap = len(seq) + 1
        pos = np.zeros(cap, dtype=np.uint32)
        hsh = np.zeros(cap, dtype=np.uint64)
        st = np.zeros(cap, dtype=np.int8)
        buf = np.frombuffer(seq, dtype=np.uint8)
        f = self._L.wfm_prefilter_kmers
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        n = f(self._p, buf.ctypes.data, len(seq), k, w, s, c_factor, pos.ctypes.data, hsh.ctypes.data, st.ctypes.data, cap)
        if n < 0:
            raise WfmError(f"wfm_prefilter_kmers failed ({n}): {self.last_error()}")
        return pos[:n], hsh[:n], st[:n]

    def add_minmers(self, seq: bytes, k: int, w: int, s: int, seq_id: int = 0):
        """wfm_add_minmers: winnowed minmer intervals of one target sequence."""
        cap = 4 * len(seq) + 64
        out = np.zeros(cap, dtype=MINMER_DTYPE)
        buf = np.frombuffer(seq, dtype=np.uint8)
        self._L.wfm_add_minmers.restype = C.c_int64
        self._L.wfm_add_minmers.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_int64]
        n = self._L.wfm_add_minmers(self._p, buf.ctypes.data, len(seq), k, w, s, seq_id, out.ctypes.data, cap)
        if n < 0:
            raise WfmError(f"wfm_add_minmers failed ({n}): {self.last_error()}")
        return out[:n]


# ---------------------------------------------------------------------------
# host-side align driver (include/wfmash_host.h)
# ---------------------------------------------------------------------------
HOST_EXPORTS = ["wfmh_test_packed_lce", "wfmh_test_is_acgt", "wfmh_align_default_params", "wfmh_align_paf", "wfmh_test_cigar", "wfmh_free", "wfmh_test_winnow",
                "wfmh_map_default_params", "wfmh_test_filter", "wfmh_map", "wfmh_test_winnow_chunked", "wfmh_test_fasta", "wfmh_test_winnow_thinned", "wfmh_test_sort_records", "wfmh_test_index_file",
                "wfmh_map_multi", "wfmh_align_paf_multi", "wfmh_test_winnow_model", "wfmh_test_sortlike_model", "wfmh_test_finish_records",
                "wfmh_release_sequences", "wfmh_test_fasta_shared"]


class MapSummary(C.Structure):
    _fields_ = [("targets", C.c_uint64), ("queries", C.c_uint64), ("subsets", C.c_uint64), ("target_bp", C.c_uint64),
                ("query_bp", C.c_uint64), ("index_windows", C.c_uint64), ("fragments", C.c_uint64), ("l2_mappings", C.c_uint64),
                ("written", C.c_uint64), ("percentage_identity", C.c_float), ("sketch_size", C.c_int32), ("ms_index", C.c_double), ("ms_map", C.c_double), ("ms_filter", C.c_double),
                ("ms_total", C.c_double), ("ms_replicate", C.c_double), ("ms_identity", C.c_double), ("ms_wall", C.c_double)]


def _handle_array(handles):
    arr = (C.c_void_p * len(handles))(*[h._p for h in handles])
    return arr


def map_paf_multi(handles, target_fasta: str, out_paf: str, query_fasta: str = None, params=None) -> "MapSummary":
    """wfmh_map_multi: the map phase over several GPUs of the node (index built once, copied to the others)."""
    L = load()
    L.wfmh_map_multi.restype = C.c_int
    L.wfmh_map_multi.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.POINTER(MapSummary)]
    s = MapSummary()
    rc = L.wfmh_map_multi(_handle_array(handles), len(handles), target_fasta.encode(), query_fasta.encode() if query_fasta else None,
                          out_paf.encode(), C.byref(params) if params is not None else None, C.byref(s))
    if rc != 0:
        raise WfmError(f"wfmh_map_multi failed ({rc}): {handles[0].last_error()}")
    return s


def map_paf(handle, target_fasta: str, out_paf: str, query_fasta: str = None, params=None) -> "MapSummary":
    """wfmh_map: the map phase on files (FASTA -> approximate mapping PAF)."""
    L = load()
    L.wfmh_map.restype = C.c_int
    L.wfmh_map.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.POINTER(MapSummary)]
    s = MapSummary()
    rc = L.wfmh_map(handle._p, target_fasta.encode(), query_fasta.encode() if query_fasta else None, out_paf.encode(),
                    C.byref(params) if params is not None else None, C.byref(s))
    if rc != 0:
        raise WfmError(f"wfmh_map failed ({rc}): {handle.last_error()}")
    return s


def host_fasta_shared(path: str, name: str) -> str:
    """wfmh_test_fasta_shared: a whole sequence through the per-path shared store, which is then kept as a map call keeps its stores."""
    L = load()
    L.wfmh_test_fasta_shared.restype = C.c_void_p
    L.wfmh_test_fasta_shared.argtypes = [C.c_char_p, C.c_char_p]
    L.wfmh_free.restype = None
    L.wfmh_free.argtypes = [C.c_void_p]
    p = L.wfmh_test_fasta_shared(path.encode(), name.encode())
    if not p:
        raise WfmError("wfmh_test_fasta_shared failed")
    s = C.string_at(p).decode()
    L.wfmh_free(p)
    if s.startswith("ERROR: "):
        raise WfmError(s)
    return s


def release_sequences() -> None:
    """wfmh_release_sequences: lets go of the sequences the last map call left loaded for the align phase."""
    L = load()
    L.wfmh_release_sequences.restype = None
    L.wfmh_release_sequences.argtypes = []
    L.wfmh_release_sequences()


class MapHostParams(C.Structure):
    """wfmh_map_params_t (include/wfmash_host.h)"""
    _fields_ = [("kmer_size", C.c_int32), ("window_length", C.c_int64), ("block_length", C.c_int64), ("chain_gap", C.c_int64),
                ("max_mapping_length", C.c_uint64), ("percentage_identity", C.c_float), ("sketch_size", C.c_int32),
                ("filter_mode", C.c_int32), ("num_mappings_for_segment", C.c_uint32), ("num_mappings_for_scaffold", C.c_uint32),
                ("drop_rand", C.c_int32), ("split", C.c_int32), ("merge_mappings", C.c_int32), ("skip_self", C.c_int32),
                ("skip_prefix", C.c_int32), ("lower_triangular", C.c_int32), ("prefix_delim", C.c_char),
                ("filter_length_mismatches", C.c_int32), ("sparsity_hash_threshold", C.c_uint64), ("overlap_threshold", C.c_double),
                ("scaffold_overlap_threshold", C.c_double), ("scaffold_max_deviation", C.c_int64), ("scaffold_gap", C.c_int64),
                ("scaffold_min_length", C.c_int64), ("legacy_output", C.c_int32), ("minimum_hits", C.c_int32),
                ("max_kmer_freq", C.c_double), ("index_by_size", C.c_int64), ("kmer_complexity_threshold", C.c_float),
                ("stage1_topani_filter", C.c_int32), ("stage2_full_scan", C.c_int32), ("ani_diff", C.c_float),
                ("ani_diff_conf", C.c_float), ("hg_numerator", C.c_double), ("threads", C.c_int32),
                ("auto_pct_identity", C.c_int32), ("ani_percentile", C.c_int32), ("ani_adjustment", C.c_float),
                ("target_prefix", C.c_char_p), ("target_list", C.c_char_p), ("query_prefix", C.c_char_p), ("query_list", C.c_char_p),
                ("index_file", C.c_char_p), ("write_index", C.c_int32), ("pad_", C.c_int32)]


def map_default_params(**over) -> MapHostParams:
    L = load()
    p = MapHostParams()
    L.wfmh_map_default_params.restype = None
    L.wfmh_map_default_params.argtypes = [C.POINTER(MapHostParams)]
    L.wfmh_map_default_params(C.byref(p))
    for k, v in over.items():
        if isinstance(v, str):
            v = v.encode()
        setattr(p, k, v)
    return p


def host_fasta(path: str, name: str = None, start: int = 0, end_inclusive: int = -1, whole: bool = False) -> str:
    """wfmh_test_fasta: the FASTA reader (random access through .fai/.gzi, or in-memory); no GPU needed.
    name None -> 'indexed|in-memory' and the name/length table."""
    L = load()
    L.wfmh_test_fasta.restype = C.c_void_p
    L.wfmh_test_fasta.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int]
    L.wfmh_free.restype = None
    L.wfmh_free.argtypes = [C.c_void_p]
    p = L.wfmh_test_fasta(path.encode(), name.encode() if name is not None else None, start, end_inclusive, int(whole))
    if not p:
        raise WfmError("wfmh_test_fasta failed")
    s = C.string_at(p).decode()
    L.wfmh_free(p)
    if s.startswith("ERROR: "):
        raise WfmError(s)
    return s


def host_filter(stage: str, mappings, fasta: str, query_name: str, params: MapHostParams) -> str:
    """wfmh_test_filter: the host-side post-processing on caller-supplied MappingResults (no GPU needed)."""
    L = load()
    m = np.ascontiguousarray(mappings, dtype=MAPPING_DTYPE)
    L.wfmh_test_filter.restype = C.c_void_p
    L.wfmh_test_filter.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_char_p, C.POINTER(MapHostParams)]
    L.wfmh_free.restype = None
    L.wfmh_free.argtypes = [C.c_void_p]
    p = L.wfmh_test_filter(stage.encode(), m.ctypes.data, len(m), fasta.encode(), query_name.encode(), C.byref(params))
    if not p:
        raise WfmError("wfmh_test_filter failed")
    s = C.string_at(p).decode()
    L.wfmh_free(p)
    if s.startswith("ERROR: "):
        raise WfmError(s)
    return s


class AlignParams(C.Structure):
    _fields_ = [("mismatch", C.c_int32), ("gap_open1", C.c_int32), ("gap_ext1", C.c_int32),
                ("gap_open2", C.c_int32), ("gap_ext2", C.c_int32),
                ("min_identity", C.c_float), ("min_alignment_length", C.c_uint64),
                ("min_block_identity", C.c_float), ("target_padding", C.c_uint64),
                ("query_padding", C.c_uint64), ("wflign_max_len_minor", C.c_uint64),
                ("disable_chain_patching", C.c_int32), ("sam_format", C.c_int32),
                ("emit_md_tag", C.c_int32), ("no_seq_in_sam", C.c_int32), ("threads", C.c_int32), ("pad_", C.c_int32)]


class AlignSummary(C.Structure):
    _fields_ = [("records", C.c_uint64), ("aligned_bp", C.c_uint64), ("written", C.c_uint64),
                ("skipped", C.c_uint64), ("cells", C.c_uint64), ("ms_gpu", C.c_double), ("ms_total", C.c_double),
                ("ms_rows", C.c_double), ("ms_fetch", C.c_double), ("ms_wflign", C.c_double), ("ms_text", C.c_double), ("batches", C.c_uint64),
                ("cells_tile", C.c_uint64), ("tile_launches", C.c_uint64), ("ms_tile", C.c_double), ("ms_tags", C.c_double)]


def _host():
    L = load()
    if not getattr(L, "_host_bound", False):
        L.wfmh_align_default_params.restype = None
        L.wfmh_align_default_params.argtypes = [C.POINTER(AlignParams)]
        L.wfmh_align_paf.restype = C.c_int
        L.wfmh_align_paf.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                     C.POINTER(AlignParams), C.POINTER(AlignSummary)]
        L.wfmh_test_cigar.restype = C.c_void_p
        L.wfmh_test_cigar.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_longlong, C.c_longlong]
        L.wfmh_free.restype = None
        L.wfmh_free.argtypes = [C.c_void_p]
        L._host_bound = True
    return L


def host_cigar_fn(fn, a=b"", b=b"", query=b"", target=b"", i0=0, i1=0) -> str:
    """Pure host-side CIGAR helpers of wfmash_amd/host (no GPU needed)."""
    L = _host()
    enc = lambda x: x if isinstance(x, bytes) else x.encode()
    p = L.wfmh_test_cigar(enc(fn), enc(a), enc(b), enc(query), enc(target), i0, i1)
    s = C.string_at(p).decode()
    L.wfmh_free(p)
    return s


def host_plan_batch_bytes(file_bytes, rows, row_bytes, row_bases_sum, batch_records, batch_bases, nworkers, ngpu=1, min_batches=1, level=True) -> int:
    """Aligner::plan_batch_bytes: bytes of the mapping file one batch may hold (2**64 - 1: no limit)."""
    L = _host()
    L.wfmh_test_plan_batch_bytes.restype = C.c_ulonglong
    L.wfmh_test_plan_batch_bytes.argtypes = [C.c_ulonglong] * 9 + [C.c_int]
    return int(L.wfmh_test_plan_batch_bytes(file_bytes, rows, row_bytes, row_bases_sum, batch_records, batch_bases, nworkers, ngpu, min_batches, 1 if level else 0))


def align_paf(handle, target_fasta, mapping_paf, out_paf, query_fasta=None, params=None):
    """wfmh_align_paf: the align phase on files (mapping PAF in, aligned PAF out)."""
    L = _host()
    prm = AlignParams()
    L.wfmh_align_default_params(C.byref(prm))
    for k, v in (params or {}).items():
        setattr(prm, k, v)
    summ = AlignSummary()
    rc = L.wfmh_align_paf(handle._p, target_fasta.encode(), query_fasta.encode() if query_fasta else None,
                          mapping_paf.encode(), out_paf.encode(), C.byref(prm), C.byref(summ))
    if rc != 0:
        raise WfmError(f"wfmh_align_paf failed ({rc}): {handle.last_error()}")
    return summ


def read_record_tags(path):
    """The align driver's diagnostic channel (WFM_RECORD_TAGS=<file>, host/aligner.cpp): {row of the mapping file: (tags, score, ok)}
    with tags = WFM_PF_* of the main alignment | of the head patch << 8 | of the tail patch << 16."""
    out = {}
    with open(path) as f:
        for line in f:
            a = line.split()
            if len(a) >= 4:
                out[int(a[0])] = (int(a[1]), int(a[2]), int(a[3]))
    return out


def stratified_rows(tags, n_rows, per_stratum=64, top_scores=32, uniform=64):
    """Rows of a mapping file to hold against the oracle, drawn where the align path has broken before rather than uniformly:
    every record (up to per_stratum each) whose root or a child ran again, whose patches went to a second / third budget or to the
    ring kernel, which ran on the byte kernels, whose overlap walk took several rounds; the top_scores highest scores; then every
    k-th row.  Returns (sorted rows, {stratum: count})."""
    strata = {
        "root_again": lambda t: t & WFM_PF_ROOT_AGAIN,
        "job_again": lambda t: t & WFM_PF_JOB_AGAIN,
        "patch_second_budget": lambda t: (t >> 8 | t >> 16) & WFM_PF_BASE_RETRY,
        "patch_third_budget": lambda t: (t >> 8 | t >> 16) & WFM_PF_BASE_RETRY2,
        "ring_kernel": lambda t: (t | t >> 8 | t >> 16) & WFM_PF_RING_KERNEL,
        "base_tiles": lambda t: (t >> 8 | t >> 16) & WFM_PF_BASE_TILES,
        "byte_kernel": lambda t: (t | t >> 8 | t >> 16) & WFM_PF_BYTE_KERNEL,
        "p2_rounds": lambda t: t & WFM_PF_P2_ROUNDS,
        "leaf_retry": lambda t: t & (WFM_PF_BASE_RETRY | WFM_PF_BASE_RETRY2),
    }
    rows, counts = set(), {}
    for name, pred in strata.items():
        hit = sorted(r for r, (t, _, ok) in tags.items() if pred(t))
        counts[name] = len(hit)
        step = max(1, len(hit) // per_stratum)
        rows.update(hit[::step][:per_stratum])
    by_score = sorted(tags, key=lambda r: -tags[r][1])[:top_scores]
    rows.update(by_score)
    counts["top_scores"] = len(by_score)
    if uniform and n_rows:
        rows.update(range(0, n_rows, max(1, n_rows // uniform)))
    counts["sampled"] = len(rows)
    return sorted(r for r in rows if r < n_rows), counts


def align_paf_multi(handles, target_fasta, mapping_paf, out_paf, query_fasta=None, params=None):
    """wfmh_align_paf_multi: the align phase over several GPUs of the node."""
    L = _host()
    L.wfmh_align_paf_multi.restype = C.c_int
    L.wfmh_align_paf_multi.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                       C.POINTER(AlignParams), C.POINTER(AlignSummary)]
    prm = AlignParams()
    L.wfmh_align_default_params(C.byref(prm))
    for k, v in (params or {}).items():
        setattr(prm, k, v)
    summ = AlignSummary()
    rc = L.wfmh_align_paf_multi(_handle_array(handles), len(handles), target_fasta.encode(), query_fasta.encode() if query_fasta else None,
                                mapping_paf.encode(), out_paf.encode(), C.byref(prm), C.byref(summ))
    if rc != 0:
        raise WfmError(f"wfmh_align_paf_multi failed ({rc}): {handles[0].last_error()}")
    return summ


def host_winnow(seq: bytes, k: int, w: int, s: int, seq_id: int, hashes, strands):
    """Host winnowing stage on caller-supplied canonical k-mer hashes (no GPU needed)."""
    L = load()
    L.wfmh_test_winnow.restype = C.c_int64
    L.wfmh_test_winnow.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    cap = 4 * len(seq) + 64
    out = np.zeros(cap, dtype=MINMER_DTYPE)
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    strands = np.ascontiguousarray(strands, dtype=np.int8)
    n = L.wfmh_test_winnow(seq, len(seq), k, w, s, seq_id, hashes.ctypes.data, strands.ctypes.data, out.ctypes.data, cap)
    return out[:n]


def host_index_file(op: str, fasta: str, out_path: str, in_path: str = None, prefix_delim: str = "#"):
    """wfmh_test_index_file: 'ids' (id section of fasta's sequences) or 'rewrite' (read + write every sub-index)."""
    L = load()
    f = L.wfmh_test_index_file
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_char_p, C.c_char, C.c_char_p, C.c_char_p]
    rc = f(op.encode(), fasta.encode(), (prefix_delim or "\0").encode(), in_path.encode() if in_path else None, out_path.encode())
    if rc != 0:
        raise WfmError(f"wfmh_test_index_file({op}) failed")


def host_sort_records(recs, threads: int):
    """wfmh_test_sort_records: the closing (wpos, wpos_end) sort of a sequence's records; returns a sorted copy."""
    L = load()
    L.wfmh_test_sort_records.restype = None
    L.wfmh_test_sort_records.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    a = np.array(recs, dtype=MINMER_DTYPE, copy=True)
    L.wfmh_test_sort_records(a.ctypes.data, len(a), threads)
    return a


def host_winnow_thinned(seq: bytes, k: int, w: int, s: int, seq_id: int, hashes, strands, c_factor: float = 4.0, chunk_len: int = 0):
    """The chunked host winnowing on the thinned stream (selection restated on the host); returns
    (minmers, kept positions, replays)."""
    L = load()
    f = L.wfmh_test_winnow_thinned
    f.restype = C.c_int64
    f.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_int64, C.c_void_p,
                  C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    cap = 4 * len(seq) + 64
    out = np.zeros(cap, dtype=MINMER_DTYPE)
    kept = np.zeros(len(seq) + 1, dtype=np.uint32)
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    strands = np.ascontiguousarray(strands, dtype=np.int8)
    rep = C.c_int(0)
    nk = C.c_int64(0)
    n = f(seq, len(seq), k, w, s, seq_id, hashes.ctypes.data, strands.ctypes.data, c_factor, chunk_len, out.ctypes.data, cap,
          kept.ctypes.data, len(kept), C.byref(nk), C.byref(rep))
    return out[:n], kept[:nk.value], rep.value


def host_winnow_model(seq: bytes, k: int, w: int, s: int, seq_id: int, hashes, strands, c_factor: float = 3.0, chunk_len: int = 0):
    """The device winnower's control flow and capacities on the host (map_winnow.hip's model) over the thinned stream;
    returns (minmers or None when the device would hand the sequence back, why-bits)."""
    L = load()
    f = L.wfmh_test_winnow_model
    f.restype = C.c_int64
    f.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_int64, C.c_void_p,
                  C.c_int64, C.POINTER(C.c_uint32)]
    cap = 4 * len(seq) + 64
    out = np.zeros(cap, dtype=MINMER_DTYPE)
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    strands = np.ascontiguousarray(strands, dtype=np.int8)
    why = C.c_uint32(0)
    n = f(seq, len(seq), k, w, s, seq_id, hashes.ctypes.data, strands.ctypes.data, c_factor, chunk_len, out.ctypes.data, cap, C.byref(why))
    return (None if n < 0 else out[:n]), why.value


def host_sortlike_model(recs):
    """map_finish.hip's data-parallel restatement of std::sort's arrangement, run on the host; returns the sorted copy."""
    L = load()
    f = L.wfmh_test_sortlike_model
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int64]
    out = np.ascontiguousarray(recs, dtype=MINMER_DTYPE).copy()
    f(out.ctypes.data, len(out))
    return out


def host_finish_records(raw, w: int):
    """cut / strand sign / std::sort / de-duplication of raw interval records on the host (finish_records)."""
    L = load()
    f = L.wfmh_test_finish_records
    f.restype = C.c_int64
    f.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64]
    raw = np.ascontiguousarray(raw, dtype=MINMER_DTYPE)
    cap = 4 * len(raw) + 64
    while True:
        out = np.zeros(cap, dtype=MINMER_DTYPE)
        n = f(raw.ctypes.data, len(raw), w, out.ctypes.data, cap)
        if n <= cap:
            return out[:n]
        cap = n


def host_winnow_chunked(seq: bytes, k: int, w: int, s: int, seq_id: int, hashes, strands, chunk_len: int):
    """The speculative chunked form of the host winnowing (single thread); returns (minmers, replays).
    replays = chunks whose speculation failed and were replayed; -1 = fell back to one stream."""
    L = load()
    f = L.wfmh_test_winnow_chunked
    f.restype = C.c_int64
    f.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                  C.POINTER(C.c_int)]
    cap = 4 * len(seq) + 64
    out = np.zeros(cap, dtype=MINMER_DTYPE)
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    strands = np.ascontiguousarray(strands, dtype=np.int8)
    rep = C.c_int(0)
    n = f(seq, len(seq), k, w, s, seq_id, hashes.ctypes.data, strands.ctypes.data, chunk_len, out.ctypes.data, cap, C.byref(rep))
    return out[:n], rep.value
