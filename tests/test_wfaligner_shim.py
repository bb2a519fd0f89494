"""The wfa::WFAligner shim (seam 1 of INTEGRATION.md): a translation unit written like the
reference's call sites (wflign.cpp:136-165, 280-309) must compile against it and, on a GPU box,
produce the oracle's op strings."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <cstdio>
#include <cstring>
#include <string>
#include "wfmash_amd/host/WFAligner.hpp"
int main(int argc, char** argv) {
  std::string target = argv[1], query = argv[2];
  wfa::WFAlignerGapAffine2Pieces wf_aligner(0, 5, 8, 2, 24, 1, wfa::WFAligner::Alignment, wfa::WFAligner::MemoryUltralow);
  wf_aligner.setHeuristicNone();
  const int status = wf_aligner.alignEnd2End(target.data(), (int)target.size(), query.data(), (int)query.size());
  if (status != 0) return 2;
  char* ops; int n;
  wf_aligner.getAlignment(&ops, &n);
  printf("%.*s\n", n, ops);
  wfa::WFAlignerGapAffine2Pieces head(0, 5, 8, 2, 24, 1, wfa::WFAligner::Alignment, wfa::WFAligner::MemoryMed);
  std::string hq = query.substr(0, 60), ht = target.substr(0, 64);
  if (head.alignEndsFree(ht, (int)ht.size(), 0, hq, (int)hq.size(), 0) != 0) return 3;
  printf("%s\n", head.getAlignment().c_str());
  return 0;
}
'''


def _build(tmp_path):
    src = tmp_path / "shim_user.cpp"
    src.write_text(SRC)
    exe = tmp_path / "shim_user"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + ROOT, str(src), "-o", str(exe),
                           "-L" + os.path.join(ROOT, "wfmash_amd"), "-lwfmash_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "wfmash_amd")])
    return str(exe)


def test_shim_compiles_and_links(tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
def test_shim_matches_oracle(tmp_path, oracle):
    from wfmash_amd import synth
    exe = _build(tmp_path)
    t = synth.random_dna(77, 700)
    q = synth.mutate(t, 0.06, 78)
    out = subprocess.check_output([exe, t.decode(), q.decode()]).decode().split("\n")
    rc, ops, _, _ = oracle.align_biwfa(t, q)
    assert out[0].encode() == ops
    rc, hops, _, _ = oracle.align_endsfree(t[:64], 64, 0, q[:60], 60, 0)
    assert out[1].encode() == hops
