"""The RCCL leg of the multi-GPU plumbing run once on hardware: backend "nccl" (= RCCL on ROCm), device tensors end to end,
world_size 1 on the one GPU of the test box (two ranks cannot share a device under RCCL).  The collectives are the ones a
`bench.py --gpus N` / `scripts/pangenome_run.py` run issues: all_gather of the sizes, gather of the padded payloads
(wfmash_amd/dist.py: gather_bytes), the chunked file gather (gather_files) and the sharded map driver on top of them."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys, tempfile
    sys.path.insert(0, {root!r})
    import torch
    import torch.distributed as dist
    from wfmash_amd import dist as D
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    # gather_bytes on device tensors
    text = ("q1\\t100\\t0\\t100\\t+\\tt1\\t100\\t0\\t100\\t90\\t100\\t60\\n" * 1000).encode()
    payload = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
    parts = D.gather_bytes(payload, dist, dst=0)
    assert parts is not None and len(parts) == 1 and parts[0].is_cuda
    assert bytes(parts[0].cpu().numpy().tobytes()) == text
    empty = D.gather_bytes(torch.zeros(0, dtype=torch.uint8, device=dev), dist, dst=0)
    assert empty[0].numel() == 0
    # a real all_reduce as well (the barrier + max-over-ranks timing of bench.py)
    t = torch.tensor([3.5], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    assert float(t.item()) == 3.5
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "rank0.paf")
        open(p, "wb").write(text)
        paths = D.gather_files(p, dist, td, dst=0, device=dev, chunk_bytes=4096)
        assert paths == [p]
        names = ["qa", "qb", "qc"]
        out = D.map_sharded(lambda mine: "".join(f"{{n}}\\tx\\n" for n in mine), names, [300, 100, 200], dist=dist, device=dev)
        assert out == "qa\\tx\\nqb\\tx\\nqc\\tx\\n"
        outp = os.path.join(td, "merged.paf")
        def map_fn(mine):
            q = os.path.join(td, "mine.paf")
            open(q, "w").write("".join(f"{{n}}\\ty\\n" for n in mine))
            return q
        D.map_sharded_files(map_fn, names, [300, 100, 200], outp, td, dist=dist, device=dev)
        assert open(outp).read() == "qa\\ty\\nqb\\ty\\nqc\\ty\\n"
    dist.destroy_process_group()
    print("NCCL_OK")
""")


def test_gathers_over_rccl_with_device_tensors(tmp_path):
    script = tmp_path / "nccl_leg.py"
    script.write_text(SCRIPT.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
