"""bench.py's launch contract: `python bench.py --gpus N` without a torch.distributed environment starts N ranks
itself (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1); under the driver's own launcher
(WORLD_SIZE set) it joins the existing job.  --rank-check stops after the ranks have found each other (gloo, no GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def _json_lines(text):
    out = []
    dec = json.JSONDecoder()
    for line in text.splitlines():
        line = line.strip()
        while line.startswith("{"):  # two ranks may share a line of the pipe
            try:
                obj, end = dec.raw_decode(line)
            except ValueError:
                break
            out.append(obj)
            line = line[end:].lstrip()
    return out


def test_gpus_flag_launches_that_many_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rank-check"], cwd=ROOT, env=_clean_env(), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [j for j in _json_lines(r.stdout) if j.get("rank_check")]
    assert sorted(j["rank"] for j in lines) == [0, 1]
    assert all(j["n_gpus"] == 2 and j["ranks_seen"] == [0, 1] for j in lines)


def test_gpus_flag_must_agree_with_the_launcher():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rank-check"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_print_one_line_for_the_job():
    """The N>1 path end to end on the one-GPU box: two ranks share the device (gloo instead of RCCL), each aligns its own
    shard, the CIGAR payload is gathered to rank 0, rank 0 prints the one JSON line with n_gpus = 2."""
    env = dict(_clean_env(), WFM_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--pairs", "4", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [j for j in _json_lines(r.stdout) if "metric" in j]
    assert len(lines) == 1
    j = lines[0]
    assert j["n_gpus"] == 2 and j["steps"] == 1 and j["scaling"] == "weak"
    assert j["value"] > 0 and j["roofline"]["frac"] > 0


def test_strong_scaling_shards_one_file_over_the_ranks():
    """--scaling strong: the same --total-pairs records at every N, dealt out by dist.shard_records -- every record on
    exactly one rank, the shards level."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rank-check", "--scaling", "strong", "--total-pairs", "13"], cwd=ROOT,
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = sorted((j for j in _json_lines(r.stdout) if j.get("rank_check")), key=lambda j: j["rank"])
    assert [j["scaling"] for j in lines] == ["strong", "strong"]
    assert sorted(lines[0]["records"] + lines[1]["records"]) == list(range(13))
    assert abs(len(lines[0]["records"]) - len(lines[1]["records"])) <= 1
    one = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--rank-check", "--scaling", "strong", "--total-pairs", "13"], cwd=ROOT,
                         env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert [j["records"] for j in _json_lines(one.stdout) if j.get("rank_check")] == [list(range(13))]


@pytest.mark.gpu
def test_strong_scaling_two_ranks_on_one_gpu():
    env = dict(_clean_env(), WFM_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--scaling", "strong", "--total-pairs", "6",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    (j,) = [j for j in _json_lines(r.stdout) if "metric" in j]
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["pairs_total"] == 6 and j["config"]["pairs_per_gpu"] == 3
    assert j["value"] > 0


@pytest.mark.gpu
def test_c4_all_vs_all_two_ranks_give_the_records_of_one(tmp_path):
    """bench.py --config C4 (north_star's C4 as one job: all eight haplotypes against all eight, the queries sharded over the ranks for the
    map phase, the mapping records of all queries exchanged and sharded by weight for the align phase, the PAF gathered to rank 0): two ranks
    sharing the one GPU of the box must produce, byte for byte, the records one rank produces."""
    outs = []
    for n in (1, 2):
        out = tmp_path / f"c4.n{n}.paf"
        env = dict(_clean_env(), WFM_BENCH_SHARE_GPU="1")
        r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--config", "C4", "--c4-mbp", "1.5", "--steps", "1", "--warmup", "0",
                            "--no-cpu-baseline", "--out-paf", str(out)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-3000:]
        (j,) = [j for j in _json_lines(r.stdout) if "metric" in j]
        assert j["n_gpus"] == n and j["scaling"] == "strong" and j["value"] > 0
        assert j["config"]["queries_total"] == 8 and j["config"]["records_total"] > 50
        assert j["roofline"]["bound"] == "valu"
        outs.append(out.read_bytes())
    assert len(outs[0]) > 10000 and outs[0] == outs[1]
