"""Boundary test in its strongest form: the REFERENCE'S OWN wflign.cpp / wflign_patch.cpp /
wflign_alignment.cpp / wflign_swizzle.cpp, compiled where they lie against the product's WFAligner.hpp
(oracle/_ref/libref_wflign.so), run do_biwfa_alignment (wflign.cpp:108-483) on the GPU through seam 1 -- and the record
each call writes must be, byte for byte, the record the product's batch pipeline (wfmh_align_paf:
host/aligner.cpp + host/wflign_hip.cpp) writes for the same mapping row.  This pins rows a3 (erosion, patching,
merging), a4 (swizzle) and a5 (PAF / SAM / MD writers) on the reference's code instead of on a restatement; what stays
a restatement on the expected side is parseMashmapRow and the window arithmetic of createSeqRecord (computeAlignments.hpp
needs htslib)."""
import pytest

from wfmash_amd import capi
from oracle import wflign_host as W
from oracle import pyref_wflign as R
from test_align_paf_gpu import _make_case

pytestmark = pytest.mark.gpu


def _need_ref():
    if not R.available():
        pytest.skip("oracle/_ref/libref_wflign.so is not built (it is compiled from /root/reference by `make -C oracle ref`; "
                    "a checkout without that tree cannot run this comparison)")


def _reference_records(lines, seqs, target_padding=1000, query_padding=1000, max_len_minor=128000, **kw):
    out = []
    for line in lines:
        try:
            row = W.parse_mashmap_row(line, target_padding, query_padding)
        except ValueError:
            continue
        ref, qry = seqs[row["refId"]], seqs[row["qId"]]
        head_pad = min(row["rStartPos"], max_len_minor)
        tail_pad = min(len(ref) - row["rEndPos"], max_len_minor)
        window = W.upper_valid_dna(ref[row["rStartPos"] - head_pad:row["rEndPos"] + tail_pad])
        q = W.upper_valid_dna(qry[row["qStartPos"]:row["qEndPos"]])
        if row["rev"]:
            q = W.revcomp(q)
        text = R.do_biwfa_alignment(row["qId"], q, len(qry), row["qStartPos"], row["rev"], row["refId"], window, head_pad, len(ref),
                                    row["rStartPos"], row["rEndPos"] - row["rStartPos"], wflign_max_len_minor=max_len_minor,
                                    mashmap_estimated_identity=float(row["mm_id"]), chain_id=row["chain_id"],
                                    chain_length=row["chain_length"], chain_pos=row["chain_pos"], **kw)
        if text:
            out.append(text)
    return out


@pytest.mark.parametrize("seed", [7, 58])
def test_paf_records_equal_the_reference_codes(gpu, tmp_path, seed):
    _need_ref()
    fa, paf, seqs, lines = _make_case(tmp_path, seed)
    out = str(tmp_path / "out.paf")
    capi.align_paf(gpu, fa, paf, out)
    got = [l.rstrip("\n") for l in open(out)]
    # processMappingRecord re-tokenises the writer's line and joins with single tabs (computeAlignments.hpp:484-525)
    exp = ["\t".join(t.split()) for t in _reference_records(lines, seqs)]
    assert len(got) == len(exp) and len(got) >= 20
    diff = [i for i, (a, b) in enumerate(zip(got, exp)) if a != b]
    assert not diff, (diff[:3], got[diff[0]][:300], exp[diff[0]][:300])


def test_no_patching_no_padding_equal_the_reference_codes(gpu, tmp_path):
    _need_ref()
    fa, paf, seqs, lines = _make_case(tmp_path, 21)
    out = str(tmp_path / "out.paf")
    capi.align_paf(gpu, fa, paf, out, params={"disable_chain_patching": 1, "target_padding": 0, "query_padding": 0})
    got = [l.rstrip("\n") for l in open(out)]
    exp = ["\t".join(t.split()) for t in _reference_records(lines, seqs, 0, 0, disable_chain_patching=True)]
    assert got == exp and len(got) >= 20


def test_sam_records_with_md_equal_the_reference_codes(gpu, tmp_path):
    _need_ref()
    fa, paf, seqs, lines = _make_case(tmp_path, 33)
    out = str(tmp_path / "out.sam")
    capi.align_paf(gpu, fa, paf, out, params={"sam_format": 1, "emit_md_tag": 1})
    body = [l for l in open(out) if not l.startswith("@")]
    exp = _reference_records(lines, seqs, paf_format_else_sam=False, emit_md_tag=True)
    assert len(body) == len(exp) and len(body) >= 20
    for a, b in zip(body, exp):
        assert a == b, (a[:200], b[:200])


def test_other_penalties_equal_the_reference_codes(gpu, tmp_path):
    _need_ref()
    fa, paf, seqs, lines = _make_case(tmp_path, 44)
    out = str(tmp_path / "out.paf")
    capi.align_paf(gpu, fa, paf, out, params={"mismatch": 4, "gap_open1": 6, "gap_ext1": 2, "gap_open2": 26, "gap_ext2": 1})
    got = [l.rstrip("\n") for l in open(out)]
    exp = ["\t".join(t.split()) for t in _reference_records(lines, seqs, penalties=(4, 6, 2, 26, 1))]
    assert got == exp and len(got) >= 20
