"""GPU end-to-end test of the align phase on files: wfmh_align_paf (C++ Aligner batcher +
wflign pipeline on the GPU) against the Python restatement of the reference's align
phase driven by the CPU oracle.  PAF lines must be byte-identical."""
import gzip
import os
import random

import pytest

from wfmash_amd import capi, synth
from oracle import wflign_host as W

pytestmark = pytest.mark.gpu


def _write_fasta(path, seqs, gz=False, width=60):
    op = gzip.open if gz else open
    with op(path, "wt") as f:
        for name, s in seqs.items():
            f.write(f">{name} some description\n")
            s = s.decode()
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")


def _make_case(tmp_path, seed, n_hap=4, L=40000, gz=False):
    rng = random.Random(seed)
    base = synth.random_dna(seed, L)
    seqs = {}
    for h in range(n_hap):
        s = synth.mutate(base, rng.choice([0.01, 0.03, 0.06]), seed * 100 + h)
        if h == 1:  # lower-case and IUPAC noise must be normalised (makeUpperCaseAndValidDNA)
            b = bytearray(s)
            b[5000:5200] = bytes(b[5000:5200]).lower()
            b[9000:9005] = b"RYKMN"
            s = bytes(b)
        seqs[f"hap{h}#1#chr1"] = s
    seqs["hap9#1#chr1"] = W.revcomp(synth.mutate(base, 0.04, seed * 100 + 9))
    fa = str(tmp_path / ("pan.fa.gz" if gz else "pan.fa"))
    _write_fasta(fa, seqs, gz)
    names = list(seqs)
    lines = []
    chain = 0
    for _ in range(24):
        qn, tn = rng.sample(names, 2)
        qlen_total, tlen_total = len(seqs[qn]), len(seqs[tn])
        seg = rng.choice([1500, 4000, 9000, 15000])
        qs = rng.randrange(0, min(qlen_total, tlen_total) - seg - 600)
        qe = qs + seg
        rev = (qn == "hap9#1#chr1") != (tn == "hap9#1#chr1")
        if rev:
            # segment [qs,qe) of one strand corresponds to [len-qe, len-qs) on the other
            ts, te = tlen_total - qe, tlen_total - qs
        else:
            ts, te = qs, qe
        ts = max(0, ts + rng.randrange(-60, 60))
        te = min(tlen_total, te + rng.randrange(-60, 60))
        chain += 1
        n_pieces = rng.choice([1, 1, 2, 3])
        # split the mapping into consecutive chain pieces (ch:Z:id.pos.len, mappingOutput.hpp:121)
        cuts_q = [qs + (qe - qs) * k // n_pieces for k in range(n_pieces + 1)]
        cuts_t = [ts + (te - ts) * k // n_pieces for k in range(n_pieces + 1)]
        for k in range(n_pieces):
            if rev:
                tq0, tq1 = cuts_t[n_pieces - k - 1], cuts_t[n_pieces - k]
            else:
                tq0, tq1 = cuts_t[k], cuts_t[k + 1]
            lines.append("\t".join(map(str, [qn, qlen_total, cuts_q[k], cuts_q[k + 1], "-" if rev else "+", tn, tlen_total,
                                             tq0, tq1, 100, seg, 30, "id:f:0.95", "kc:f:0.9", f"ch:Z:{chain}.{k + 1}.{n_pieces}"])))
    lines.append("garbage line with too few columns")
    paf = str(tmp_path / "map.paf")
    with open(paf, "w") as f:
        f.write("\n".join(lines) + "\n")
    return fa, paf, seqs, lines


@pytest.mark.parametrize("gz", [False, True])
def test_align_paf_matches_reference_restatement(gpu, tmp_path, gz):
    fa, paf, seqs, lines = _make_case(tmp_path, 7 + int(gz), gz=gz)
    out = str(tmp_path / "out.paf")
    summ = capi.align_paf(gpu, fa, paf, out)
    got = [l.rstrip("\n") for l in open(out)]
    exp = W.align_mapping_lines(lines, seqs, seqs)
    assert summ.records == len(lines) - 1 and summ.skipped == 1
    assert len(got) == len(exp) and len(got) >= 20
    diff = [i for i, (a, b) in enumerate(zip(got, exp)) if a != b]
    assert not diff, (diff[:3], got[diff[0]][:200], exp[diff[0]][:200])
    # every record is a valid alignment of the FASTA (pafcheck-style, CMakeLists.txt:452)
    for line in got:
        f = line.split("\t")
        q, qs, qe, strand, t, ts, te = f[0], int(f[2]), int(f[3]), f[4], f[5], int(f[7]), int(f[8])
        cg = [x for x in f if x.startswith("cg:Z:")][0][5:]
        qseq = W.upper_valid_dna(seqs[q][qs:qe])
        if strand == "-":
            qseq = W.revcomp(qseq)
        tseq = W.upper_valid_dna(seqs[t][ts:te])
        qi = ti = 0
        for n, op in W.parse(cg):
            if op == "=":
                assert qseq[qi:qi + n] == tseq[ti:ti + n]
                qi += n; ti += n
            elif op == "X":
                assert all(qseq[qi + j] != tseq[ti + j] for j in range(n))
                qi += n; ti += n
            elif op == "I":
                qi += n
            else:
                ti += n
        assert qi == len(qseq) and ti == len(tseq)


def test_align_paf_no_patching_and_custom_params(gpu, tmp_path):
    fa, paf, seqs, lines = _make_case(tmp_path, 21)
    out = str(tmp_path / "out2.paf")
    capi.align_paf(gpu, fa, paf, out, params={"disable_chain_patching": 1, "target_padding": 0, "query_padding": 0})
    got = [l.rstrip("\n") for l in open(out)]
    exp = []
    for line in lines:
        try:
            row = W.parse_mashmap_row(line, 0, 0)
        except ValueError:
            continue
        ref, qry = seqs[row["refId"]], seqs[row["qId"]]
        tail_pad = min(len(ref) - row["rEndPos"], 128000)
        tav = W.upper_valid_dna(ref[row["rStartPos"]:row["rEndPos"] + tail_pad])
        tgt = tav[:row["rEndPos"] - row["rStartPos"]]
        q = W.upper_valid_dna(qry[row["qStartPos"]:row["qEndPos"]])
        if row["rev"]:
            q = W.revcomp(q)
        cg = W.do_biwfa_alignment(q, tgt, tav, None, disable_chain_patching=True)
        rec = W.write_alignment_paf(cg, row["qId"], len(qry), row["qStartPos"], len(q), row["rev"], row["refId"], len(ref),
                                    row["rStartPos"], row["mm_id"], row["chain_id"], row["chain_length"], row["chain_pos"])
        if rec:
            exp.append("\t".join(rec.split()))
    assert got == exp


def test_align_sam_with_md_matches_reference_restatement(gpu, tmp_path):
    """-a -d: SAM records with the MD:Z tag (wflign_patch.cpp:2480-2609)."""
    fa, paf, seqs, lines = _make_case(tmp_path, 33)
    out = str(tmp_path / "out.sam")
    capi.align_paf(gpu, fa, paf, out, params={"sam_format": 1, "emit_md_tag": 1})
    got = [l.rstrip("\n") for l in open(out)]
    header = [l for l in got if l.startswith("@")]
    body = [l for l in got if not l.startswith("@")]
    assert len(header) == len(seqs) + 1 and header[0].startswith("@SQ\tSN:hap0#1#chr1\tLN:")
    exp = []
    for line in lines:
        try:
            row = W.parse_mashmap_row(line, 1000, 1000)
        except ValueError:
            continue
        ref, qry = seqs[row["refId"]], seqs[row["qId"]]
        tail_pad = min(len(ref) - row["rEndPos"], 128000)
        tav = W.upper_valid_dna(ref[row["rStartPos"]:row["rEndPos"] + tail_pad])
        tgt = tav[:row["rEndPos"] - row["rStartPos"]]
        q = W.upper_valid_dna(qry[row["qStartPos"]:row["qEndPos"]])
        if row["rev"]:
            q = W.revcomp(q)
        cg = W.do_biwfa_alignment(q, tgt, tav, None)
        rec = W.write_alignment_sam(cg, row["qId"], row["qStartPos"], row["rev"], row["refId"], row["rStartPos"], row["mm_id"],
                                    row["chain_id"], row["chain_length"], row["chain_pos"], q, tav, emit_md_tag=True)
        if rec:
            exp.append(rec)
    assert len(body) == len(exp) and len(body) >= 20
    for a, b in zip(body, exp):
        assert a == b, (a[:150], b[:150])


def test_align_paf_bytes_do_not_depend_on_batching_or_on_a_seekable_input(gpu, tmp_path):
    """The align driver sizes its batches from the mapping file's size and first rows, and cuts small files into several batches
    on request: the output is the same bytes however the file is cut (WFM_ALIGN_MIN_BATCHES, WFM_ALIGN_LEVEL: read once per
    process, hence the subprocesses), and a mapping file that cannot be rewound (a FIFO) is read as it comes."""
    import subprocess
    import sys
    import threading
    fa, paf, seqs, lines = _make_case(tmp_path, 21)
    ref = str(tmp_path / "ref.paf")
    capi.align_paf(gpu, fa, paf, ref)
    want = open(ref, "rb").read()
    assert want.count(b"\n") >= 20
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\nfrom wfmash_amd import capi\nh = capi.Handle(0)\n"
            "s = capi.align_paf(h, sys.argv[1], sys.argv[2], sys.argv[3])\nprint(int(s.batches))\nh.close()\n") % root
    for env_add, min_batches in (({"WFM_ALIGN_MIN_BATCHES": "5"}, 4), ({"WFM_ALIGN_MIN_BATCHES": "3", "WFM_ALIGN_LEVEL": "0"}, 2)):
        out = str(tmp_path / "cut.paf")
        r = subprocess.run([sys.executable, "-c", code, fa, paf, out], env=dict(os.environ, **env_add), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert int(r.stdout.strip().splitlines()[-1]) >= min_batches
        assert open(out, "rb").read() == want
    fifo = str(tmp_path / "map.fifo")
    os.mkfifo(fifo)
    t = threading.Thread(target=lambda: open(fifo, "wb").write(open(paf, "rb").read()))
    t.start()
    out = str(tmp_path / "fifo.paf")
    capi.align_paf(gpu, fa, fifo, out)
    t.join()
    assert open(out, "rb").read() == want
