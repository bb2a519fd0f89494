"""CPU tests of the host-side align logic (wfmash_amd/host/*.cpp through the C test
hook) against the Python restatement of the reference (oracle/wflign_host.py), plus
known-answer cases read off the reference source."""
import random

import pytest

from wfmash_amd import capi
from oracle import wflign_host as W


def _rand_cigar(rng, n_ops, ops="=XID", maxlen=40, short_bias=True):
    out, prev = [], None
    for _ in range(n_ops):
        o = rng.choice([c for c in ops if c != prev])
        L = rng.choice([1, 1, 2, 3, 3, 4, 11, 12, rng.randrange(1, maxlen), rng.randrange(100, 400)]) if short_bias else rng.randrange(1, maxlen)
        out.append(f"{L}{o}")
        prev = o
    return "".join(out)


def test_erode_short_matches_known_answers():
    f = lambda c, h: capi.host_cigar_fn("erode", c, i0=3, i1=int(h))
    # wflign.cpp:55-76: I-match-D with both indels longer than the match is absorbed
    assert f("5I2=7D100=", True) == "7I9D100="
    assert f("100=5D3=4I", False) == "100=8D7I"
    # same-type indels, or an indel not longer than the match: untouched
    assert f("5I2=7I100=", True) == "5I2=7I100="
    assert f("2I2=7D100=", True) == "2I2=7D100="
    # head form only inspects ops[1] and ops[2] (end_idx = min(size-1, 3), wflign.cpp:47-50)
    assert f("100=5I2=7D100=", True) == "100=7I9D100="
    assert f("100=1X5I2=7D100=", True) == "100=1X5I2=7D100="
    assert f("100=1X5I2=7D100=", False) == "100=1X7I9D100="
    assert f("5=", True) == "5="


def test_host_cigar_helpers_match_oracle_random():
    rng = random.Random(17)
    for _ in range(400):
        c = _rand_cigar(rng, rng.randrange(1, 9))
        for head in (True, False):
            assert capi.host_cigar_fn("erode", c, i0=3, i1=int(head)) == W.erode_short_matches_in_cigar(c, 3, head)
        c2 = _rand_cigar(rng, rng.randrange(0, 5))
        assert capi.host_cigar_fn("merge", c, c2) == W.merge_adjacent_ops(c, c2)
        q, t, p = W.head_erosion(c)
        assert capi.host_cigar_fn("head_erosion", c) == f"{q},{t},{p}"
        q, t, i = W.tail_erosion(W.parse(c))
        assert capi.host_cigar_fn("tail_erosion", c) == f"{q},{t},{i}"
    ops = bytes(rng.choice(b"MXID") for _ in range(500))
    assert capi.host_cigar_fn("compress", ops) == W.compress(ops)


def test_erosion_stop_rules():
    # stops at the first op after >=128 bp are eroded on both axes and an >=11 '=' run was seen (wflign.cpp:252-260)
    c = "50=1X90=1X5=2I300="
    q, t, p = W.head_erosion(c)
    assert (q, t) == (141, 141) and c[p:] == "1X5=2I300="
    assert capi.host_cigar_fn("head_erosion", c) == f"{q},{t},{p}"
    # no long '=' run: erosion runs to MAX_ERODE_LENGTH
    c = "".join("10=1X" for _ in range(800))
    q, t, p = W.head_erosion(c)
    assert q >= 4096 and q < 4096 + 11
    assert capi.host_cigar_fn("head_erosion", c) == f"{q},{t},{p}"


def test_swizzle_matches_oracle():
    rng = random.Random(5)
    hit = 0
    for it in range(600):
        unit = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([1, 2, 3])))
        n, d = rng.randrange(1, 12), rng.randrange(1, 8)
        rest = bytes(rng.choice(b"ACGT") for _ in range(40))
        # tandem repeat so that the '=' run can slide across the deletion
        query = (unit * 30)[:n] + rest
        target = (unit * 30)[:n + d] + rest if rng.random() < 0.7 else bytes(rng.choice(b"ACGT") for _ in range(n + d)) + rest
        c = f"{n}={d}D40=" if rng.random() < 0.5 else f"{n}={d}D20=1X19="
        a = capi.host_cigar_fn("swap_start", c, query=query, target=target + b"ACGTACGT")
        assert a == W.try_swap_start_pattern(c, query, target + b"ACGTACGT")
        hit += a != c
        # end pattern: ... dD n=
        query2 = rest + (unit * 30)[:n]
        target2 = rest + (unit * 30)[:n + d]
        c2 = f"40={d}D{n}=" if rng.random() < 0.6 else f"20=1X19={d}D{n}="
        b = capi.host_cigar_fn("swap_end", c2, query=query2, target=target2 + b"TTTT")
        assert b == W.try_swap_end_pattern(c2, query2, target2 + b"TTTT")
        hit += b != c2
    assert hit > 50  # the swaps really fire


def test_swap_end_requires_eq_del_only():
    # wflign_swizzle.cpp:82-102: any X or I in the CIGAR vetoes the end swap
    q = b"ACGTACGTAA" + b"AA"
    t = b"ACGTACGTAA" + b"AAAA"
    assert capi.host_cigar_fn("swap_end", "10=2D2=", query=q, target=t) == "12=2D"
    q2 = b"ACGTTCGTAA" + b"AA"
    assert capi.host_cigar_fn("swap_end", "4=1X5=2D2=", query=q2, target=t) == "4=1X5=2D2="


def test_paf_writer_matches_oracle():
    rng = random.Random(23)
    for _ in range(300):
        c = _rand_cigar(rng, rng.randrange(1, 10))
        ops = W.parse(c)
        qlen = sum(n for n, o in ops if o in "=XI")
        tlen = sum(n for n, o in ops if o in "=XD")
        qoff, toff = rng.randrange(0, 5000), rng.randrange(0, 5000)
        rev = rng.random() < 0.5
        mm = rng.choice([0.95, 0.8731, 1.0, 0.7])
        cid, clen, cpos = rng.choice([(-1, 1, 1), (3, 4, 2), (7, 1, 1)])
        meta = f"q#1|{qoff + qlen + 77}|{qoff}|{qlen}|{int(rev)}|t#2|{toff + tlen + 99}|{toff}|{mm}|{cid}|{clen}|{cpos}"
        got = capi.host_cigar_fn("paf", c, meta)
        exp = W.write_alignment_paf(c, "q#1", qoff + qlen + 77, qoff, qlen, rev, "t#2", toff + tlen + 99, toff, mm, cid, clen, cpos)
        assert got == (exp or "")


def test_paf_writer_known_answer():
    # 100= : gi = bi = 1 -> mapq 255; ch:Z: is written id.LENGTH.pos (wflign_patch.cpp:2708)
    got = capi.host_cigar_fn("paf", "2D100=3I", "q|500|10|103|0|t|900|20|0.95|5|3|2")
    assert got == "q\t500\t10\t110\t+\tt\t900\t22\t122\t100\t100\t255\tgi:f:1\tbi:f:1\tmd:f:0.95\tch:Z:5.3.2\tcg:Z:100=\t"
    # reverse strand coordinates (wflign_patch.cpp:2669-2676)
    got = capi.host_cigar_fn("paf", "90=1X9=", "q|500|10|100|1|t|900|20|0.9|-1|1|1")
    assert got.split("\t")[2:5] == ["10", "110", "-"]


def test_parse_mashmap_row_padding_rules():
    line = "q1\t5000\t1000\t2000\t-\tt1\t9000\t3000\t4000\t50\t1000\t20\tid:f:0.93\tkc:f:0.9\tch:Z:7.1.3"
    r = capi.host_cigar_fn("parse_row", line, i0=1000, i1=1000).split(",")
    # first piece of a 3-piece chain: query start padding is computed but NOT stored (computeAlignments.hpp:278-288)
    assert r == ["q1", "1000", "2000", "-", "t1", "2000", "5000", "0.93", "7", "3", "1"]
    line2 = line.replace("ch:Z:7.1.3", "ch:Z:7.3.3")
    r = capi.host_cigar_fn("parse_row", line2, i0=1000, i1=1000).split(",")
    assert r[1:3] == ["1000", "3000"]  # last piece: end padded and stored
    line3 = line.replace("ch:Z:7.1.3", "ch:Z:7.1.1")
    r = capi.host_cigar_fn("parse_row", line3, i0=1000, i1=1000).split(",")
    assert r[1:3] == ["0", "3000"]
    assert capi.host_cigar_fn("parse_row", "a\tb\tc") == "ERROR"
    for ln in (line, line2, line3):
        row = W.parse_mashmap_row(ln, 1000, 1000)
        got = capi.host_cigar_fn("parse_row", ln, i0=1000, i1=1000).split(",")
        assert [str(row["qStartPos"]), str(row["qEndPos"]), str(row["rStartPos"]), str(row["rEndPos"])] == [got[1], got[2], got[5], got[6]]


def test_md_string_matches_oracle_and_known_answers():
    t = b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT"
    # 5= 1X 4= 2D 3= : MD counts matches, names the mismatched / deleted target bases
    assert capi.host_cigar_fn("md", "5=1X4=2D3=", target=t, i0=0) == "MD:Z:5C4^GT3"
    assert capi.host_cigar_fn("md", "5=1X4=2D3=", target=t, i0=0) == W.md_string("5=1X4=2D3=", 0, t)
    rng = random.Random(3)
    for _ in range(200):
        c = _rand_cigar(rng, rng.randrange(1, 8), maxlen=6, short_bias=False)
        tlen = sum(n for n, o in W.parse(c) if o in "=XD")
        tgt = bytes(rng.choice(b"ACGT") for _ in range(tlen + 4))
        assert capi.host_cigar_fn("md", c, target=tgt, i0=0) == W.md_string(c, 0, tgt)


# ---- the batch pipeline's forms on runs (wflign_hip.cpp: *_ops) against the text forms and the oracle ----

def test_run_forms_equal_the_text_forms_random():
    rng = random.Random(29)
    for _ in range(600):
        c = _rand_cigar(rng, rng.randrange(1, 11))
        for head in (True, False):
            assert capi.host_cigar_fn("erode_ops", c, i0=3, i1=int(head)) == W.erode_short_matches_in_cigar(c, 3, head)
        c2 = _rand_cigar(rng, rng.randrange(0, 5))
        assert capi.host_cigar_fn("merge_ops", c, c2) == W.merge_adjacent_ops(c, c2)
        q, t, p = W.head_erosion(c)
        assert capi.host_cigar_fn("head_erosion_ops", c) == f"{q},{t},{p}"


def test_runs_to_cigar():
    # run = (length << 2) | op, op 0 M (written '='), 1 X, 2 I, 3 D (include/wfmash_hip.h)
    runs = [(100 << 2) | 0, (1 << 2) | 1, (7 << 2) | 2, (3 << 2) | 3, (12 << 2) | 0]
    assert capi.host_cigar_fn("runs", ",".join(map(str, runs))) == "100=1X7I3D12="
    assert capi.host_cigar_fn("runs", "") == ""


def test_swizzle_run_forms_match_oracle():
    rng = random.Random(31)
    hit = 0
    for it in range(800):
        unit = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([1, 2, 3])))
        n, d = rng.randrange(1, 12), rng.randrange(1, 8)
        rest = bytes(rng.choice(b"ACGT") for _ in range(40))
        query = (unit * 30)[:n] + rest
        target = (unit * 30)[:n + d] + rest if rng.random() < 0.7 else bytes(rng.choice(b"ACGT") for _ in range(n + d)) + rest
        # (a following '=' run makes the swapped CIGAR merge: n= dD 40= -> dD (n+40)=)
        c = rng.choice([f"{n}={d}D40=", f"{n}={d}D20=1X19=", f"{n}={d}D3I40=", f"{n}={d}D"])
        a = capi.host_cigar_fn("swap_start_ops", c, query=query, target=target + b"ACGTACGT")
        assert a == W.try_swap_start_pattern(c, query, target + b"ACGTACGT"), (c, query, target)
        hit += a != c
        query2 = rest + (unit * 30)[:n]
        target2 = rest + (unit * 30)[:n + d] if rng.random() < 0.8 else rest + bytes(rng.choice(b"ACGT") for _ in range(n + d))
        c2 = rng.choice([f"40={d}D{n}=", f"20=1X19={d}D{n}=", f"2D38={d}D{n}=", f"{d}D{n}="])
        q2 = query2 if not c2.startswith("2D") else query2[2:]
        b = capi.host_cigar_fn("swap_end_ops", c2, query=q2, target=target2 + b"TTTT")
        assert b == W.try_swap_end_pattern(c2, q2, target2 + b"TTTT"), (c2, q2, target2)
        hit += b != c2
    assert hit > 80


def test_paf_writer_run_form_matches_oracle():
    """write_alignment_paf_ops emits the record as the align driver finally writes it: the reference writer's fields
    re-joined with single tabs (computeAlignments.hpp:484-525) and a newline"""
    rng = random.Random(37)
    for _ in range(400):
        c = _rand_cigar(rng, rng.randrange(1, 12), maxlen=3000 if rng.random() < 0.2 else 40)
        ops = W.parse(c)
        qlen = sum(n for n, o in ops if o in "=XI")
        tlen = sum(n for n, o in ops if o in "=XD")
        qoff, toff = rng.randrange(0, 5_000_000_000), rng.randrange(0, 5000)
        rev = rng.random() < 0.5
        mm = rng.choice([0.95, 0.8731, 1.0, 0.7, 0.999999, 0.123456789])
        cid, clen, cpos = rng.choice([(-1, 1, 1), (3, 4, 2), (7, 1, 1), (12345, 0, 1)])
        meta = f"q#1|{qoff + qlen + 77}|{qoff}|{qlen}|{int(rev)}|t#2|{toff + tlen + 99}|{toff}|{mm}|{cid}|{clen}|{cpos}"
        got = capi.host_cigar_fn("paf_ops", c, meta)
        exp = W.write_alignment_paf(c, "q#1", qoff + qlen + 77, qoff, qlen, rev, "t#2", toff + tlen + 99, toff, mm, cid, clen, cpos)
        assert got == ("\t".join(exp.split()) + "\n" if exp else ""), (c, meta)
        assert got == ("\t".join(capi.host_cigar_fn("paf", c, meta).split()) + "\n" if exp else "")


def test_batch_plan_levels_a_few_batches_over_the_workers():
    """Aligner::plan_batch_bytes (host/aligner.cpp): a mapping file of one batch stays one batch; a file of a few batches is cut
    into a multiple of the workers' number, never into fewer batches than the caps on records and bases demand; many batches are
    left to the caps; several GPUs get at least eight batches each; a file that cannot be rewound is read as it comes."""
    import math
    import random
    NOLIMIT = 2 ** 64 - 1
    P = capi.host_plan_batch_bytes
    rng = random.Random(3)
    for _ in range(3000):
        nworkers = rng.choice([1, 2, 3, 4, 6])
        rows = rng.randrange(1, 257)
        avg_line = rng.randrange(60, 400)
        avg_bases = rng.randrange(500, 60000)
        n_rec = rng.choice([1, 5, 300, 1536, 1537, 4000, 6000, 47719, 300000])
        file_bytes = n_rec * avg_line
        cap_rec, cap_bases = 1536, 160_000_000
        b = P(file_bytes, rows, rows * avg_line, rows * avg_bases, cap_rec, cap_bases, nworkers)
        need = math.ceil(max(n_rec / cap_rec, n_rec * avg_bases / cap_bases))
        if nworkers == 1 or need < 2 or need >= 8 * nworkers:
            assert b == NOLIMIT
            continue
        n_batches = math.ceil(file_bytes / b)
        assert n_batches >= need and n_batches <= need + nworkers
        want = (need + nworkers - 1) // nworkers * nworkers
        assert b == file_bytes // want + 1
    assert P(0, 0, 0, 0, 1536, 160_000_000, 3) == NOLIMIT                      # a FIFO: nothing known
    assert P(10_000, 50, 5000, 50 * 3000, 1536, 160_000_000, 3, level=False) == NOLIMIT
    assert P(10_000, 50, 5000, 50 * 3000, 1536, 160_000_000, 3, min_batches=5) == 10_000 // 5 + 1   # WFM_ALIGN_MIN_BATCHES
    assert P(1_000_000, 0, 0, 0, 1536, 160_000_000, 6, ngpu=2) == 1_000_000 // 16 + 1               # two GPUs: sixteen batches at least


def test_record_tags_and_stratified_rows(tmp_path):
    """The parity samples of bench.py and tests/test_configs_gpu.py are drawn by capi.stratified_rows from the align driver's record tags
    (WFM_RECORD_TAGS; wfm_get_problem_flags): every stratum that has members is represented, the highest scores are in, the uniform part
    covers the file, nothing is out of range or duplicated."""
    from wfmash_amd import capi
    n = 5000
    lines = []
    for r in range(n):
        t = 0
        if r % 97 == 0:
            t |= capi.WFM_PF_ROOT_AGAIN
        if r % 251 == 3:
            t |= capi.WFM_PF_BASE_RETRY << 8          # head patch on its second budget
        if r % 1009 == 5:
            t |= (capi.WFM_PF_BASE_RETRY | capi.WFM_PF_BASE_RETRY2 | capi.WFM_PF_RING_KERNEL) << 16  # tail patch on its third, ring kernel
        if r == 4321:
            t |= capi.WFM_PF_BYTE_KERNEL
        if r % 13 == 1:
            t |= capi.WFM_PF_P2_ROUNDS
        lines.append(f"{r}\t{t}\t{1000 + (r * 7919) % 3000}\t1\n")
    p = tmp_path / "tags.tsv"
    p.write_text("".join(reversed(lines)))  # (batches are written in any order)
    tags = capi.read_record_tags(str(p))
    assert len(tags) == n and tags[4321][0] & capi.WFM_PF_BYTE_KERNEL
    rows, counts = capi.stratified_rows(tags, n, per_stratum=8, top_scores=8, uniform=16)
    assert rows == sorted(set(rows)) and 0 <= rows[0] and rows[-1] < n
    assert counts["root_again"] == len(range(0, n, 97)) and counts["patch_second_budget"] > 0 and counts["patch_third_budget"] == counts["ring_kernel"] > 0
    assert counts["byte_kernel"] == 1 and 4321 in rows
    picked = set(rows)
    assert any(tags[r][0] & capi.WFM_PF_ROOT_AGAIN for r in picked)
    assert any((tags[r][0] >> 16) & capi.WFM_PF_BASE_RETRY2 for r in picked)
    assert any(tags[r][0] & capi.WFM_PF_P2_ROUNDS for r in picked)
    top = sorted(tags, key=lambda r: -tags[r][1])[:8]
    assert set(top) <= picked
    assert len(picked) <= 8 * 8 + 8 + 17 and counts["sampled"] == len(rows)
    # without tags (a run that wrote none): uniform rows only
    rows2, counts2 = capi.stratified_rows({}, 100, uniform=10)
    assert rows2 == list(range(0, 100, 10)) and counts2["root_again"] == 0
