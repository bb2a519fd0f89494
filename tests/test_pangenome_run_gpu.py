"""The node-level driver (scripts/pangenome_run.py): queries sharded over ranks, map + align per rank, PAF gathered
to rank 0.  On a one-GPU box two ranks share the device and gather over gloo; the merged output must be the
single-process output, which must be what the C ABI calls give when made directly."""
import os
import subprocess
import sys

import pytest

from wfmash_amd import capi
from test_map_paf_gpu import _pangenome, _write_fasta

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "pangenome_run.py")


def _run(cmd, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stderr


@pytest.mark.parametrize("approx", [True, False], ids=["map_only", "map_and_align"])
def test_two_ranks_reproduce_single_process(gpu, tmp_path, approx):
    seqs = _pangenome(73)
    fa = str(tmp_path / "pan.fa")
    _write_fasta(fa, seqs)
    flags = ["--pct", "0.9", "--threads", "4"] + (["-m"] if approx else [])
    one, two, direct = (str(tmp_path / n) for n in ("one.paf", "two.paf", "direct.paf"))
    _run([sys.executable, SCRIPT, fa, "--out", one] + flags, tmp_path)
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
          "--master-port", "29517", SCRIPT, fa, "--out", two] + flags, tmp_path)
    P = capi.map_default_params(percentage_identity=0.9, auto_pct_identity=0, threads=4)
    mapping = str(tmp_path / "direct.map.paf")
    capi.map_paf(gpu, fa, mapping, params=P)
    if approx:
        direct = mapping
    else:
        capi.align_paf(gpu, fa, mapping, direct, params={"threads": 4})
    want = open(direct).read()
    assert want.count("\n") > 10
    assert open(one).read() == want
    assert open(two).read() == want
