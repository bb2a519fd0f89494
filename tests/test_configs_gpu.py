"""BASELINE.json configs C1 and C4 as tests (C2: test_lpa_gpu.py, C3 / C5: test_align_gpu.py).

C1  data/scerevisiae8.fa.gz all-vs-all, default parameters.  The data file is a missing blob in the reference tree, so the
    input is the SUBSTITUTE SURVEY 8d allows: synth.yeast_like -- 8 strains x several chromosomes of one random genome,
    names STRAIN#1#chrN, strain divergence 0.3-1.2 % -- scaled down to test size.  Defaults mean -p ani50-2 (identity
    estimated from the data), -Y '#', -n inf, -P 50k, map + align.
C4  8 synthetic chr1 haplotypes all-vs-all -Y '#', sharded over the GPUs of a node; scaled-down haplotypes, and the
    sharding run (i) inside one process over several device handles (what `wfmash-hip --gpus N` does) and (ii) as one
    process per rank under torch.distributed (scripts/pangenome_run.py; ranks share the one GPU of the test box over gloo).
    Every variant must give the single-GPU bytes."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import map_ani as ANI
from oracle import map_pipeline as MP
from oracle import pyfilter, pymap
from oracle import wflign_host as W
from wfmash_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "wfmash_amd", "wfmash-hip")


def _check_records(text, seqs, min_records):
    """pafcheck's rule (CMakeLists.txt:452): every cg:Z: spells its two intervals base by base."""
    lines = text.splitlines()
    assert len(lines) >= min_records
    covered = {}
    for line in lines:
        f = line.split("\t")
        qs, qe, strand, ts, te = int(f[2]), int(f[3]), f[4], int(f[7]), int(f[8])
        cg = [x for x in f[12:] if x.startswith("cg:Z:")][0][5:]
        qseq = W.upper_valid_dna(seqs[f[0]][qs:qe])
        if strand == "-":
            qseq = W.revcomp(qseq)
        tseq = W.upper_valid_dna(seqs[f[5]][ts:te])
        q = np.frombuffer(qseq, dtype=np.uint8)
        t = np.frombuffer(tseq, dtype=np.uint8)
        qi = ti = 0
        for n, op in W.parse(cg):
            if op == "=":
                assert (q[qi:qi + n] == t[ti:ti + n]).all()
                qi += n; ti += n
            elif op == "X":
                assert (q[qi:qi + n] != t[ti:ti + n]).all()
                qi += n; ti += n
            elif op == "I":
                qi += n
            else:
                assert op == "D"
                ti += n
        assert qi == len(q) and ti == len(t)
        covered[(f[0], f[5])] = covered.get((f[0], f[5]), 0) + (qe - qs)
    return covered


def _handles(n):
    return [capi.Handle(0) for _ in range(n)]


def _oracle_lines(mapping_paf, fasta_path, out_path):
    """oracle/wflign_host.py over every row of a mapping file, many rows side by side -- in a process of its own (oracle/align_lines_cli.py
    forks its pool there): this process holds the HIP runtime and is not forked."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "align_lines_cli.py"), fasta_path, mapping_paf, out_path],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return [l.rstrip("\n") for l in open(out_path)]


# ---------------------------------------------------------------- C1 ----

def test_c1_shape_small_against_the_oracles(gpu, tmp_path):
    """Defaults end to end on a yeast-like set small enough for the stage oracles: the identity threshold is the ANI
    oracle's estimate (ani50-2), the mapping PAF is byte-identical to the stage oracles + the reference's own filter code,
    the aligned PAF to the align oracle."""
    if not (pymap.have_ref() and pyfilter.have_ref()):
        pytest.skip("oracle/_ref is not built (compiled from /root/reference by `make -C oracle ref`)")
    recs = [(n, s.tobytes()) for n, s in synth.yeast_like(4, 3, 72_000)]
    fa = str(tmp_path / "y.fa")
    synth.write_fasta(fa, recs)
    seqs = dict(recs)
    names = [n for n, _ in recs]
    groups = MP.ref_groups(names)
    pct = np.float32(ANI.estimate_identity([s for _, s in recs], groups, 50, -2.0))
    P = capi.map_default_params()  # defaults: ani50-2
    assert P.auto_pct_identity == 1 and P.ani_percentile == 50 and abs(P.ani_adjustment + 2.0) < 1e-6
    m = str(tmp_path / "m.paf")
    summ = capi.map_paf(gpu, fa, m, params=P)
    assert summ.percentage_identity == pct and 0.93 < pct < 0.999
    S = MP.sketch_size(pct, 1000, 15)
    assert summ.sketch_size == S
    maps, _, _ = MP.map_queries(recs, pct)
    Pexp = capi.map_default_params(percentage_identity=float(pct), auto_pct_identity=0, sketch_size=S)
    exp = "".join(pyfilter.ref_filter("subset", maps[q], fa, names[q], Pexp) for q in range(len(recs)))
    got = open(m).read()
    assert got == exp and len(got.splitlines()) >= 12
    a = str(tmp_path / "a.paf")
    capi.align_paf(gpu, fa, m, a)
    want = W.align_mapping_lines(got.splitlines(), seqs, seqs)
    assert [l.rstrip("\n") for l in open(a)] == want
    _check_records(open(a).read(), seqs, 12)


def test_c1_shape_all_vs_all_defaults(gpu, tmp_path):
    """8 strains x 8 chromosomes (12.8 Mbp), wfmash-hip with no option but -t: every record valid, every strain pair
    covered, and the bytes do not depend on how many device handles share the work."""
    recs = [(n, s.tobytes()) for n, s in synth.yeast_like(8, 8, 1_600_000)]
    fa = str(tmp_path / "yeast_like.fa")
    synth.write_fasta(fa, recs)
    seqs = dict(recs)
    out = str(tmp_path / "cli.paf")
    r = subprocess.run([CLI, "-t", "16", "--out", out, fa], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    covered = _check_records(text, seqs, 8 * 7 * 8)
    # no strain maps to itself (-Y '#'), and each chromosome of each strain is aligned to its homologue in the 7 others
    strains = synth.YEAST_STRAINS
    for (q, t), bp in covered.items():
        assert q.split("#")[0] != t.split("#")[0]
    for a in strains:
        for b in strains:
            if a == b:
                continue
            for c in range(1, 9):
                key = (f"{a}#1#chr{c}", f"{b}#1#chr{c}")
                assert covered.get(key, 0) > 0.85 * len(seqs[key[0]]), key
    # the same through the C ABI with three handles: identical bytes
    hs = _handles(3)
    try:
        m3, a3 = str(tmp_path / "m3.paf"), str(tmp_path / "a3.paf")
        capi.map_paf_multi(hs, fa, m3, params=capi.map_default_params(threads=16))
        capi.align_paf_multi(hs, fa, m3, a3, params={"threads": 16})
        assert open(a3).read() == text
    finally:
        for h in hs:
            h.close()


# ---------------------------------------------------------------- C4 ----

def test_c4_shape_sharded_runs_give_the_single_gpu_bytes(gpu, tmp_path):
    recs = [(n, s.tobytes()) for n, s in synth.pangenome(8, 1_500_000, n_sv=3, sv_min=5_000, sv_max=30_000)]
    fa = str(tmp_path / "c4.fa")
    names, lengths = synth.write_fasta(fa, recs)
    assert names == [f"hap{i}#1#chr1" for i in range(1, 9)]
    seqs = dict(recs)
    P = capi.map_default_params(threads=16)  # -Y '#' and ani50-2 are the defaults
    m1, a1 = str(tmp_path / "m1.paf"), str(tmp_path / "a1.paf")
    s1 = capi.map_paf(gpu, fa, m1, params=P)
    capi.align_paf(gpu, fa, m1, a1, params={"threads": 16})
    text = open(a1).read()
    covered = _check_records(text, seqs, 8 * 7 * 20)
    assert s1.percentage_identity > 0.95
    for q in names:
        for t in names:
            if q != t:
                assert covered.get((q, t), 0) > 0.9 * len(seqs[q]), (q, t)
    # (i) one process, four device handles: index built once and copied, batches go to whichever handle is free
    hs = _handles(4)
    try:
        m4, a4 = str(tmp_path / "m4.paf"), str(tmp_path / "a4.paf")
        s4 = capi.map_paf_multi(hs, fa, m4, params=P)
        assert open(m4).read() == open(m1).read()
        assert s4.fragments == s1.fragments and s4.l2_mappings == s1.l2_mappings and s4.written == s1.written
        capi.align_paf_multi(hs, fa, m4, a4, params={"threads": 16})
        assert open(a4).read() == text
    finally:
        for h in hs:
            h.close()
    # (ii) one process per rank (torch.distributed, gloo on this one-GPU box): the index is built by rank 0 only
    out = str(tmp_path / "ranks.paf")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "scripts", "pangenome_run.py"), fa, "--out", out, "--threads", "8"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    assert '"index_once_per_node": true' in r.stderr
    assert open(out).read() == text


def test_c4_shape_against_the_oracles(gpu, tmp_path):
    """C4 at a size where the production switches are the ones under test: 8 haplotypes `hapN#1#chr1` of 5.2 Mbp (above
    WFM_WINNOW_DEV_MIN: hashing, thinning, winnowing and the closing sort run on the device), defaults (-Y '#', ani50-2),
    `wfmash-hip -m` then the align phase.  Held against the oracles, not against the product:
      * identity threshold = the ANI oracle's estimate;
      * mapping PAF of one query haplotype byte-identical to the reference's own addMinmers (oracle/_ref/libref_map.so) ->
        index / L1 / L2 stage oracles -> the reference's own filter + output code (oracle/_ref/libref_filter.so);
      * aligned PAF of a sample of 240 mapping records (all query haplotypes) byte-identical to oracle/wflign_host.py over
        oracle/wfa2p.c; the sample's records appear unchanged in the run over the whole mapping file."""
    if not (pymap.have_ref() and pyfilter.have_ref()):
        pytest.skip("oracle/_ref is not built (compiled from /root/reference by `make -C oracle ref`)")
    recs = [(n, s.tobytes()) for n, s in synth.pangenome(8, 5_200_000, n_sv=4, sv_min=5_000, sv_max=50_000)]
    fa = str(tmp_path / "c4.fa")
    names, lengths = synth.write_fasta(fa, recs)
    assert names == [f"hap{i}#1#chr1" for i in range(1, 9)] and min(lengths) > 5_000_000
    seqs = dict(recs)
    m = str(tmp_path / "m.paf")
    env = {k: v for k, v in os.environ.items() if not k.startswith("WFM_")}
    env["WFM_DEBUG"] = "1"
    r = subprocess.run([CLI, "-m", "-t", "16", "--out", m, fa], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    dev = [l for l in r.stderr.splitlines() if "winnowing on the device" in l]
    assert dev and int(dev[-1].split("winnowing on the device:")[1].split("sequences")[0]) == 8 and "closing sort on the device" in dev[-1], r.stderr[-3000:]
    got = open(m).read()
    # -- map: one query haplotype against the index of all eight
    q = 2
    groups = MP.ref_groups(names)
    pct = np.float32(ANI.estimate_identity([s for _, s in recs], groups, 50, -2.0))
    S = MP.sketch_size(pct, 1000, 15)
    maps, _, _ = MP.map_queries(recs, pct, queries={q})
    Pexp = capi.map_default_params(percentage_identity=float(pct), auto_pct_identity=0, sketch_size=S)
    exp = pyfilter.ref_filter("subset", maps[q], fa, names[q], Pexp)
    mine = "".join(l + "\n" for l in got.splitlines() if l.split("\t", 1)[0] == names[q])
    assert len(maps[q]) > 30_000 and len(exp.splitlines()) >= 7 * 90
    assert mine == exp
    # -- align: a sample over all queries against the align oracle
    lines = got.splitlines()
    assert len(lines) >= 8 * 7 * 90
    sample = lines[::max(1, len(lines) // 240)][:240]
    ms, as_, a = str(tmp_path / "ms.paf"), str(tmp_path / "as.paf"), str(tmp_path / "a.paf")
    open(ms, "w").write("".join(l + "\n" for l in sample))
    capi.align_paf(gpu, fa, ms, as_, params={"threads": 16})
    got_s = [l.rstrip("\n") for l in open(as_)]
    want = W.align_mapping_lines(sample, seqs, seqs)
    assert len(want) >= 230 and got_s == want
    summ = capi.align_paf(gpu, fa, m, a, params={"threads": 16})
    full = [l.rstrip("\n") for l in open(a)]
    assert summ.written == len(full) >= len(lines) - 8
    assert set(got_s) <= set(full)
    it = iter(full)
    assert all(any(x == y for y in it) for x in got_s)  # ... and in the same order
    _check_records("".join(l + "\n" for l in full[::7]), seqs, 600)


def test_c4_scaled_rank_every_record_against_the_align_oracle(gpu, tmp_path, monkeypatch):
    """The bench line's scaled C4 rank (8 haplotypes of 8 Mbp with structural variants, `-Y '#'`, one query haplotype: ~1.2 k mapping
    records of ~50 kb) with EVERY aligned record byte-identical to oracle/wflign_host.py over oracle/wfa2p.c -- not a sample: the records
    that have broken before are the rare ones (a root that runs again, a patch on its third budget, a record across a structural
    variant), and the run's own tags (WFM_RECORD_TAGS, wfm_get_problem_flags) must show that this input holds them."""
    recs = [(n, s.tobytes()) for n, s in synth.pangenome(8, 8_000_000, n_sv=6)]
    fa = str(tmp_path / "c4.fa")
    names, lengths = synth.write_fasta(fa, recs)
    seqs = dict(recs)
    ql = str(tmp_path / "q.txt")
    open(ql, "w").write(names[0] + "\n")
    m, a, tg = str(tmp_path / "m.paf"), str(tmp_path / "a.paf"), str(tmp_path / "tags.tsv")
    capi.map_paf(gpu, fa, m, params=capi.map_default_params(threads=32, query_list=ql))
    lines = [l for l in open(m).read().splitlines() if l]
    assert len(lines) >= 1000
    monkeypatch.setenv("WFM_RECORD_TAGS", tg)
    summ = capi.align_paf(gpu, fa, m, a, params={"threads": 32})
    monkeypatch.delenv("WFM_RECORD_TAGS")
    got = [l.rstrip("\n") for l in open(a)]
    tags = capi.read_record_tags(tg)
    assert sorted(tags) == list(range(len(lines)))
    want = _oracle_lines(m, fa, str(tmp_path / "oracle.paf"))
    assert len(want) >= len(lines) - 8 and summ.written == len(got)
    bad = [i for i, (x, y) in enumerate(zip(got, want)) if x != y]
    assert not bad and len(got) == len(want), (len(got), len(want), bad[:5])
    rows, counts = capi.stratified_rows(tags, len(lines))
    print("strata of the scaled C4 rank:", counts)
    # the input holds the rare paths: patches that overflow their first budget (records across a structural variant)
    assert counts["patch_second_budget"] >= 8
