"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/_ref/libref_wflign.so -- the reference's own wflign sources
(wflign.cpp, wflign_patch.cpp, wflign_alignment.cpp, wflign_swizzle.cpp) compiled where they lie under /root/reference
against the product's WFAligner.hpp (oracle/ref_wflign.cpp, oracle/Makefile target `ref`).  Needs a GPU at run time:
every alignment the reference code asks for goes through the product's C ABI."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_ref", "libref_wflign.so")
_LIB = None


def available() -> bool:
    return os.path.exists(PATH)


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(PATH)
        L.ref_do_biwfa_alignment.restype = C.c_void_p
        L.ref_do_biwfa_alignment.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_char_p, C.c_char_p,
                                             C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_float, C.c_uint64, C.c_float,
                                             C.c_int32, C.c_int32, C.c_int32]
        L.ref_wflign_free.restype = None
        L.ref_wflign_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def do_biwfa_alignment(query_name, query: bytes, query_total_length, query_offset, query_is_rev, target_name, target_buf: bytes,
                       target_skip, target_total_length, target_offset, target_length, penalties=(5, 8, 2, 24, 1), emit_md_tag=False,
                       paf_format_else_sam=True, no_seq_in_sam=False, disable_chain_patching=False, min_identity=0.0,
                       min_alignment_length=32, min_block_identity=0.1, wflign_max_len_minor=128000, mashmap_estimated_identity=0.95,
                       chain_id=-1, chain_length=1, chain_pos=1) -> str:
    """wflign::wavefront::do_biwfa_alignment (wflign.cpp:108-483) as Aligner::processAlignment calls it
    (computeAlignments.hpp:661-723); returns what it wrote to its output stream."""
    L = lib()
    x, o1, e1, o2, e2 = penalties
    p = L.ref_do_biwfa_alignment(query_name.encode(), query, query_total_length, query_offset, len(query), int(query_is_rev),
                                 target_name.encode(), target_buf, target_skip, target_total_length, target_offset, target_length,
                                 x, o1, e1, o2, e2, int(emit_md_tag), int(paf_format_else_sam), int(no_seq_in_sam), int(disable_chain_patching),
                                 min_identity, min_alignment_length, min_block_identity, wflign_max_len_minor, mashmap_estimated_identity,
                                 chain_id, chain_length, chain_pos)
    s = C.string_at(p).decode()
    L.ref_wflign_free(p)
    if s.startswith("ERROR: "):
        raise RuntimeError(s)
    return s
