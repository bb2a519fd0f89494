"""GPU end-to-end test of the map phase on files (wfmh_map: index + sketch + L1 + L2 on the GPU,
post-processing on the host) against the stage oracles strung together (oracle/map_pipeline.py)
and the reference's own filter/output code where it is built.  The mapping PAF must be
byte-identical."""
import os

import pytest

from oracle import map_pipeline as MP
from oracle import pyfilter, pymap
from oracle import wflign_host as W
from wfmash_amd import capi, synth

pytestmark = pytest.mark.gpu


def _write_fasta(path, seqs, width=70):
    with open(path, "w") as f, open(path + ".fai", "w") as fai:
        off = 0
        for name, s in seqs:
            hdr = f">{name}\n"
            f.write(hdr)
            off += len(hdr)
            fai.write(f"{name}\t{len(s)}\t{off}\t{width}\t{width + 1}\n")
            t = s.decode()
            for i in range(0, len(t), width):
                f.write(t[i:i + width] + "\n")
            off += len(t) + (len(t) + width - 1) // width


def _pangenome(seed, L=30000):
    base = synth.random_dna(seed, L)
    unit = synth.random_dna(seed + 1, 900)
    base = base[:L // 3] + unit + base[L // 3:L // 3 + 4000] + unit + base[L // 3 + 4000:]
    seqs = []
    for g, gname in enumerate(["HG01", "HG02", "HG03"]):
        for hap in (1, 2):
            s = synth.mutate(base, 0.01 + 0.015 * g, seed * 100 + g * 10 + hap)
            if g == 1 and hap == 2:
                s = W.revcomp(s)
            if g == 2 and hap == 1:  # a large deletion and lower-case soft masking
                s = s[:8000] + s[12500:20000].lower() + s[20000:]
            seqs.append((f"{gname}#{hap}#chr1", s))
    seqs.append(("tiny#1#x", synth.random_dna(seed + 9, 700)))       # shorter than a window: never indexed, never mapped
    seqs.append(("other#1#chr2", synth.random_dna(seed + 7, 6500)))  # unrelated
    return seqs


def _expected(seqs, fa, P, pct, **kw):
    """The expected mapping PAF never comes from the product: minmer intervals from the reference's own addMinmers,
    post-processing by the reference's own filter code (oracle/_ref).  Without that build the comparison cannot be
    made and the test says so."""
    if not (pymap.have_ref() and pyfilter.have_ref()):
        pytest.skip("oracle/_ref is not built (it is compiled from /root/reference by `make -C oracle ref`): no independent "
                    "expected output for the map phase in this checkout")
    kw.pop("add_minmers", None)
    maps, group, S = MP.map_queries(seqs, pct, add_minmers=None, **kw)
    return "".join(pyfilter.ref_filter("subset", maps[q], fa, seqs[q][0], P) for q in range(len(seqs))), maps, S


@pytest.mark.parametrize("over", [{}, {"num_mappings_for_segment": 1, "chain_gap": 5000}, {"merge_mappings": 0, "scaffold_gap": 0}],
                         ids=["defaults", "n1_c5k", "no_merge"])
def test_map_paf_matches_stage_oracles(gpu, tmp_path, over):
    seqs = _pangenome(41)
    fa = str(tmp_path / "pan.fa")
    _write_fasta(fa, seqs)
    pct = 0.85
    P = capi.map_default_params(percentage_identity=pct, auto_pct_identity=0, **over)
    out = str(tmp_path / "map.paf")
    summ = capi.map_paf(gpu, fa, out, params=P)
    got = open(out).read()
    S = MP.sketch_size(pct, 1000, 15)
    exp, maps, S2 = _expected(seqs, fa, P, pct, add_minmers=lambda sq, sid: gpu.add_minmers(sq, 15, 1000, S, sid))
    assert S == S2 == 49
    assert summ.queries == len(seqs) and summ.targets == len(seqs) and summ.subsets == 1
    assert summ.l2_mappings == sum(len(m) for m in maps.values()) > 300
    assert got == exp
    assert summ.written == len(got.splitlines()) > 10
    # self and same-haplotype-group targets are never reported (-Y '#' default), the unrelated sequence maps nowhere
    for line in got.splitlines():
        f = line.split("\t")
        assert f[0].rsplit("#", 1)[0] != f[5].rsplit("#", 1)[0]
        assert "other" not in f[0] and "other" not in f[5] and "tiny" not in line


def test_map_paf_target_subsets_and_query_file(gpu, tmp_path):
    """-b style target batching (one index per subset) and a separate query file."""
    seqs = _pangenome(43, L=20000)
    fa = str(tmp_path / "t.fa")
    _write_fasta(fa, seqs)
    qfa = str(tmp_path / "q.fa")
    queries = [("sample#1#ctg", synth.mutate(seqs[0][1][3000:17000], 0.03, 99))]
    _write_fasta(qfa, queries)
    P = capi.map_default_params(percentage_identity=0.85, auto_pct_identity=0, index_by_size=45000)
    out = str(tmp_path / "m.paf")
    summ = capi.map_paf(gpu, fa, out, query_fasta=qfa, params=P)
    assert summ.subsets >= 3 and summ.queries == 1 and summ.targets == len(seqs)
    lines = open(out).read().splitlines()
    assert len(lines) >= 6
    hit = {l.split("\t")[5] for l in lines}
    assert {f"HG0{g}#{h}#chr1" for g in (1, 2, 3) for h in (1, 2)} <= hit
    for l in lines:
        f = l.split("\t")
        # the query was cut from this haplotype at 3000 (the end-anchored last fragment is reported at nfrag*w, computeMap.hpp:124-128)
        if f[5] == "HG01#1#chr1" and int(f[2]) < 13000:
            assert f[4] == "+" and abs((int(f[7]) - int(f[2])) - 3000) < 400


def test_cli_map_then_align_equals_two_phase_run(gpu, tmp_path):
    """wfmash-hip target.fa == wfmash-hip -m | wfmash-hip -i (the reference's two-phase restart), and every
    record is a valid base-level alignment of the FASTA (pafcheck-style)."""
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wfmash_amd", "wfmash-hip")
    seqs = _pangenome(47, L=24000)
    fa = str(tmp_path / "pan.fa")
    _write_fasta(fa, seqs)
    one, m, two = str(tmp_path / "one.paf"), str(tmp_path / "m.paf"), str(tmp_path / "two.paf")
    subprocess.check_call([cli, "-p", "85", "-t", "4", "--out", one, fa], cwd=str(tmp_path))
    subprocess.check_call([cli, "-m", "-p", "85", "--out", m, fa], cwd=str(tmp_path))
    subprocess.check_call([cli, "-i", m, "--out", two, fa], cwd=str(tmp_path))
    a, b = open(one).read(), open(two).read()
    assert a == b and len(a.splitlines()) >= 10
    assert not [f for f in os.listdir(tmp_path) if f.startswith("wfmash-")]  # the hand-off file is removed
    by_name = dict(seqs)
    for line in a.splitlines():
        f = line.split("\t")
        qs, qe, strand, ts, te = int(f[2]), int(f[3]), f[4], int(f[7]), int(f[8])
        cg = [x for x in f if x.startswith("cg:Z:")][0][5:]
        qseq = W.upper_valid_dna(by_name[f[0]][qs:qe])
        if strand == "-":
            qseq = W.revcomp(qseq)
        tseq = W.upper_valid_dna(by_name[f[5]][ts:te])
        qi = ti = 0
        for n, op in W.parse(cg):
            if op == "=":
                assert qseq[qi:qi + n] == tseq[ti:ti + n]
                qi += n; ti += n
            elif op == "X":
                qi += n; ti += n
            elif op == "I":
                qi += n
            else:
                ti += n
        assert qi == len(qseq) and ti == len(tseq)
    # auto identity (the default -p ani50-2) runs end to end too
    subprocess.check_call([cli, "-m", "--out", str(tmp_path / "auto.paf"), fa], cwd=str(tmp_path))
    assert len(open(tmp_path / "auto.paf").read().splitlines()) >= 6


def test_query_sharding_reproduces_single_run(gpu, tmp_path):
    """the multi-GPU map path: every rank indexes all targets and maps its shard of the queries
    (-A query list); merging the per-rank texts gives the single-GPU output.  Ranks are run one after
    the other on the one GPU here."""
    from wfmash_amd import dist as D
    seqs = _pangenome(53, L=20000)
    fa = str(tmp_path / "pan.fa")
    _write_fasta(fa, seqs)
    names = [n for n, _ in seqs]
    lens = [len(s) for _, s in seqs]
    single = str(tmp_path / "single.paf")
    capi.map_paf(gpu, fa, single, params=capi.map_default_params(percentage_identity=0.85, auto_pct_identity=0))

    def map_fn(mine):
        lst = str(tmp_path / f"q{abs(hash(tuple(mine))) % 10**8}.txt")
        with open(lst, "w") as f:
            f.write("\n".join(mine) + "\n")
        out = lst + ".paf"
        capi.map_paf(gpu, fa, out, params=capi.map_default_params(percentage_identity=0.85, auto_pct_identity=0, query_list=lst))
        return open(out).read()

    shards = D.shard_queries(lens, 3)
    texts = [map_fn([names[i] for i in sh]) for sh in shards]
    assert D.merge_query_blocks(texts, names) == open(single).read()
    assert sum(1 for t in texts if t) >= 2


def test_filter_modes_end_to_end(gpu, tmp_path):
    """one-to-one (-o), no filter (-f) and lower-triangular (-L) runs against the default run"""
    seqs = _pangenome(59, L=20000)
    fa = str(tmp_path / "pan.fa")
    _write_fasta(fa, seqs)
    order = {n: i for i, (n, _) in enumerate(seqs)}

    def run(tag, **over):
        out = str(tmp_path / f"{tag}.paf")
        capi.map_paf(gpu, fa, out, params=capi.map_default_params(percentage_identity=0.85, auto_pct_identity=0, **over))
        return [tuple(l.split("\t")[:12]) for l in open(out).read().splitlines()]

    base = run("base")
    o2o = run("o2o", filter_mode=2)
    none = run("none", filter_mode=3)
    lt = run("lt", lower_triangular=1)
    # each mode only removes records of the weaker one.  (The one-to-one pass re-attributes the kept mappings to queries
    # by coordinates, computeMap.hpp:823-838, so haplotypes with coincident coordinates can receive a record twice.)
    # so only the mapping itself (coordinates, strand, target, counts), not the query it is printed under, is a base record
    assert o2o and len(base) <= len(none) and set(base) <= set(none)
    assert {r[2:] for r in o2o} <= {r[2:] for r in base}
    assert lt and all(order[r[0]] > order[r[5]] for r in lt)
    assert {(r[0], r[5]) for r in lt} <= {(r[0], r[5]) for r in none}


def test_cli_window_size_drives_the_align_defaults(gpu, tmp_path):
    """-w other than 1k: the reference derives the paddings (min(w, 5000)) and the length cap of the align phase
    (128 w) from it (parse_args.hpp:593-620).  wfmash-hip -w 500 must give what the align oracle gives with those values
    on the CLI's own mappings -- and not what the 1 kb defaults give -- and the checks on -w / -p are the reference's."""
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wfmash_amd", "wfmash-hip")
    seqs = _pangenome(61, L=24000)
    fa = str(tmp_path / "pan.fa")
    _write_fasta(fa, seqs)
    by_name = dict(seqs)
    m, a = str(tmp_path / "m.paf"), str(tmp_path / "a.paf")
    subprocess.check_call([cli, "-m", "-w", "500", "-p", "85", "-t", "4", "--out", m, fa], cwd=str(tmp_path))
    subprocess.check_call([cli, "-w", "500", "-p", "85", "-t", "4", "--out", a, fa], cwd=str(tmp_path))
    lines = open(m).read().splitlines()
    got = open(a).read().splitlines()
    assert len(lines) >= 10
    want = W.align_mapping_lines(lines, by_name, by_name, target_padding=500, query_padding=500, max_len_minor=64000)
    assert got == want
    assert got != W.align_mapping_lines(lines, by_name, by_name)  # the 1 kb defaults align other windows
    # an explicit -E / -U wins over the derived value
    e = str(tmp_path / "e.paf")
    subprocess.check_call([cli, "-w", "500", "-p", "85", "-E", "200", "-U", "0", "-t", "4", "--out", e, fa], cwd=str(tmp_path))
    assert open(e).read().splitlines() == W.align_mapping_lines(lines, by_name, by_name, target_padding=200, query_padding=0, max_len_minor=64000)
    for bad in (["-w", "50"], ["-w", "20k"], ["-p", "40"], ["-p", "ani0"], ["-p", "ani120"], ["-i", m, "-W", str(tmp_path / "x.idx")]):
        r = subprocess.run([cli] + bad + ["--out", str(tmp_path / "bad.paf"), fa], cwd=str(tmp_path), capture_output=True, text=True)
        assert r.returncode == 1 and "ERROR" in r.stderr, bad
    ok = subprocess.run([cli, "-m", "-w", "20k", "-P", "50k", "-p", "85", "--out", str(tmp_path / "big.paf"), fa], cwd=str(tmp_path), capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr  # large windows are fine for mapping only
