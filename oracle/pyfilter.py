"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/_ref/libref_filter.so, the reference's own
post-processing code (filter.hpp, mappingFilter.hpp, mappingOutput.hpp, sequenceIds.hpp) compiled
in place from /root/reference by oracle/Makefile."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_filter.so")
_LIB = None


def have_ref() -> bool:
    return os.path.exists(_PATH)


def ref_filter(stage: str, mappings, fasta: str, query_name: str, params) -> str:
    """params: wfmash_amd.capi.MapHostParams (the wrapper reads the same C struct)."""
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(_PATH)
        _LIB.ref_filter.restype = C.c_void_p
        _LIB.ref_filter.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_char_p, C.c_void_p]
        _LIB.ref_filter_free.restype = None
        _LIB.ref_filter_free.argtypes = [C.c_void_p]
    m = np.ascontiguousarray(mappings)
    assert m.dtype.itemsize == 28
    p = _LIB.ref_filter(stage.encode(), m.ctypes.data, len(m), fasta.encode(), query_name.encode(), C.byref(params))
    s = C.string_at(p).decode()
    _LIB.ref_filter_free(p)
    return s


def ref_export_ids(fasta: str, out_path: str, prefix_delim: str = "#") -> None:
    """the reference's SequenceIdManager::exportIdMapping for the sequences of `fasta` (its .fai)"""
    lib = C.CDLL(_PATH)
    lib.ref_export_ids.restype = C.c_int
    lib.ref_export_ids.argtypes = [C.c_char_p, C.c_char, C.c_char_p]
    if lib.ref_export_ids(fasta.encode(), (prefix_delim or "\0").encode(), out_path.encode()) != 0:
        raise RuntimeError("ref_export_ids failed")

