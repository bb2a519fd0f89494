"""The one known answer the reference PUBLISHES for this path, and the map half of C2 / C5 on real and divergent data.

/root/reference/doc/performance-tuning.md:295-298 prints the reference binary's own index counters for
data/LPA.subset.fa.gz (ctest `wfmash-time-LPA`, CMakeLists.txt:436-439: `-p 80 -n 5`):
    Processed 8 sequences (0 skipped, 2317910 total bp), 20552 unique hashes, 258475 windows
    Filtered 179727/438202 k-mers occurring > 87 times (target: 0.02%)
The reference's own addMinmers (oracle/_ref/libref_map.so) reproduces all five figures at k = 15, w = 1000, s = 98 (the
sketch size of that binary, v0.22.0-184; HEAD's formula gives 59 at -p 80) -- so they pin m3 + m4 (winSketch.hpp:298-349:
threshold = max(10, floor(windows * 0.0002)), no adjustment below 50 % of the positions) on real, repeat-rich data
(the KIV-2 repeats: 41 % of the windows go).

The mapping / alignment figures of the same run (:307, :314: 861 records, 13012371 query bp, 14294162 aligned bp) are NOT
reproduced by HEAD's own filter code on HEAD's own L2 mappings under any single parameter (the closest is
`-l 3000`, 690 records / 12.64 Mbp; that binary's chaining differs), so they cannot pin anything; instead the product's
mapping PAF at those parameters is held, byte for byte, against the stage oracles + the reference's filter code."""
import gzip
import os

import numpy as np
import pytest

from oracle import map_pipeline as MP
from oracle import pyfilter, pymap
from oracle import wflign_host as W
from wfmash_amd import capi, synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
LPA = os.path.join(HERE, "golden", "LPA.subset.fa.gz")


def _read_fasta(path):
    seqs = []
    for line in gzip.open(path, "rt"):
        if line.startswith(">"):
            seqs.append([line[1:].split()[0], []])
        else:
            seqs[-1][1].append(line.strip())
    return [(n, "".join(v).encode()) for n, v in seqs]


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


def _need_ref():
    if not (pymap.have_ref() and pyfilter.have_ref()):
        pytest.skip("oracle/_ref is not built (compiled from /root/reference by `make -C oracle ref`): no independent expected output here")


def test_lpa_index_counters_equal_the_published_run(gpu):
    """doc/performance-tuning.md:295-298, figure by figure."""
    seqs = _read_fasta(LPA)
    assert len(seqs) == 8 and sum(len(s) for _, s in seqs) == 2317910
    ix, n_intervals = gpu.index_build_sequences([s for _, s in seqs], 15, 1000, 98, threads=8)
    try:
        info = ix.info()
        assert n_intervals == info.n_windows == 438202      # "179727/438202 k-mers"
        assert info.threshold == 87                          # "occurring > 87 times": floor(438202 * 0.0002)
        assert info.adjusted == 0
        assert info.filtered == 179727
        assert info.n_unique == 20552                        # "20552 unique hashes"
        assert info.n_kept == 258475                         # "258475 windows"
    finally:
        ix.free()
    # the same figures from the reference's own addMinmers + the index oracle, where the reference build is at hand
    if pymap.have_ref():
        from oracle import map_index as MI
        mm = []
        for sid, (_, sq) in enumerate(seqs):
            mm += [(int(x["hash"]), int(x["wpos"]), int(x["wpos_end"]), int(x["seqId"]), int(x["strand"])) for x in pymap.ref_add_minmers(sq, 15, 1000, 98, sid)]
        assert len(mm) == 438202
        lookup, index, st = MI.build_index(mm, 0.0002)
        assert len(lookup) == 20552 and len(index) == 258475


@pytest.mark.parametrize("case", ["published_p80_n5_s98", "c2_p90_P50k"])
def test_lpa_mapping_paf_against_the_stage_oracles(gpu, tmp_path, case):
    """The mapping PAF of LPA all-vs-all, byte for byte: minmer intervals from the reference's own addMinmers, index / L1 / L2
    from the stage oracles, filters + chaining + scaffolds + PAF text from the reference's own code."""
    _need_ref()
    seqs = _read_fasta(LPA)
    fa = str(tmp_path / "lpa.fa")
    _write_fasta(fa, seqs)
    if case == "published_p80_n5_s98":
        pct, s_given, over = 0.80, 98, dict(num_mappings_for_segment=5, sketch_size=98)
    else:
        pct, s_given, over = 0.90, None, dict(max_mapping_length=50000)
    P = capi.map_default_params(percentage_identity=pct, auto_pct_identity=0, threads=8, **over)
    out = str(tmp_path / "map.paf")
    summ = capi.map_paf(gpu, fa, out, params=P)
    got = open(out).read()
    maps, group, S = MP.map_queries(seqs, pct, s=s_given)
    assert S == (98 if s_given else 39)
    assert summ.l2_mappings == sum(len(m) for m in maps.values())
    exp = "".join(pyfilter.ref_filter("subset", maps[q], fa, seqs[q][0], P) for q in range(len(seqs)))
    assert got == exp
    lines = got.splitlines()
    qbp = sum(int(l.split("\t")[3]) - int(l.split("\t")[2]) for l in lines)
    if case == "published_p80_n5_s98":
        # what HEAD's own code gives at the published parameters (the published binary wrote 861 records / 13012371 bp: see the docstring)
        assert (len(lines), qbp) == (3990, 17834455)
    else:
        assert (len(lines), qbp) == (660, 20358650)


def test_c5_map_half_p70_on_divergent_segments(gpu, tmp_path):
    """C5's map half: -p 70 (s = 78) on 100 kb segments at 15 % divergence (70 / 15 / 15), against the stage oracles + the
    reference's filter code; every query must find its source."""
    _need_ref()
    n, L = 6, 100000
    targets = [(f"t{i}#1#seg", synth.random_dna(0xC5 + i, L)) for i in range(n)]
    queries = [(f"q{i}#1#seg", synth.mutate(targets[i][1], 0.15, 0xC50000 + i, p_sub=0.70, p_ins=0.15)) for i in range(n)]
    seqs = targets + queries
    fa = str(tmp_path / "c5.fa")
    _write_fasta(fa, seqs)
    pct = 0.70
    P = capi.map_default_params(percentage_identity=pct, auto_pct_identity=0, threads=8)
    out = str(tmp_path / "map.paf")
    summ = capi.map_paf(gpu, fa, out, params=P)
    got = open(out).read()
    maps, group, S = MP.map_queries(seqs, pct)
    assert S == 78
    assert summ.l2_mappings == sum(len(m) for m in maps.values()) > 0
    exp = "".join(pyfilter.ref_filter("subset", maps[q], fa, seqs[q][0], P) for q in range(len(seqs)))
    assert got == exp
    cov = {}
    for l in got.splitlines():
        f = l.split("\t")
        if f[0].startswith("q") and f[5] == "t" + f[0][1:]:
            cov[f[0]] = cov.get(f[0], 0) + int(f[3]) - int(f[2])
    assert len(cov) == n and all(v >= 0.8 * L for v in cov.values()), cov
