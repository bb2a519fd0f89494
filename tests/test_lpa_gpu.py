"""BASELINE configs[1] as an acceptance + parity test on the reference's own test data
(data/LPA.subset.fa.gz, 8 haplotype contigs of the LPA locus, 2.3 Mbp; committed as a fixture):
all-vs-all map + align at -p 90 -P 50k.  Mirrors the reference's ctest bar (CMakeLists.txt:436-464:
pafcheck-style CIGAR validation, coverage) and adds what it lacks: the aligned PAF must be
byte-identical to the CPU restatement of the align phase driven by the CPU oracle on the same mappings."""
import gzip
import os
import subprocess

import pytest

from oracle import wflign_host as W

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FASTA = os.path.join(HERE, "golden", "LPA.subset.fa.gz")
CLI = os.path.join(os.path.dirname(HERE), "wfmash_amd", "wfmash-hip")


def _read_fasta(path):
    seqs, name = {}, None
    for line in gzip.open(path, "rt"):
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = []
        else:
            seqs[name].append(line.strip())
    return {k: "".join(v).encode() for k, v in seqs.items()}


def test_lpa_all_vs_all_map_and_align(tmp_path):
    seqs = _read_fasta(FASTA)
    assert len(seqs) == 8 and sum(len(s) for s in seqs.values()) == 2317910
    m, aln = str(tmp_path / "map.paf"), str(tmp_path / "aln.paf")
    subprocess.check_call([CLI, "-m", "-p", "90", "-P", "50k", "-t", "8", "--out", m, FASTA], cwd=str(tmp_path))
    subprocess.check_call([CLI, "-i", m, "-t", "8", "--out", aln, FASTA], cwd=str(tmp_path))
    map_lines = open(m).read().splitlines()
    got = open(aln).read().splitlines()
    assert len(map_lines) >= 100 and len(got) >= 0.9 * len(map_lines)
    # --- pafcheck: every =/X column of every CIGAR agrees with the FASTA
    aligned_bp = 0
    cover = {n: bytearray(len(s)) for n, s in seqs.items()}
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
        aligned_bp += qe - qs
        cover[q][qs:qe] = b"\x01" * (qe - qs)
        cover[t][ts:te] = b"\x01" * (te - ts)
    assert aligned_bp > 10_000_000
    # --- coverage (scripts/test.sh bar of the reference's yeast test, 0.89): query or target intervals cover each contig
    for n, c in cover.items():
        assert sum(c) / len(c) >= 0.89, (n, sum(c) / len(c))
    # --- parity: the align phase on these mappings, CPU restatement + CPU oracle, byte-identical PAF
    exp = W.align_mapping_lines(map_lines, seqs, seqs)
    assert len(got) == len(exp)
    diff = [i for i, (a, b) in enumerate(zip(got, exp)) if a != b]
    assert not diff, (len(diff), got[diff[0]][:160], exp[diff[0]][:160])
