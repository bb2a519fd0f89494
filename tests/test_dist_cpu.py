"""world_size-2 gloo test of the multi-GPU plumbing (records sharded over ranks, PAF payload
gathered to rank 0).  The data path has no collective; this is the only exchange."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wfmash_amd.dist import gather_bytes, shard_records
    weights = [(i * 7919) % 13 + 1 for i in range(40)]
    shards = shard_records(weights, world)
    mine = shards[rank]
    payload = ("|".join(f"rec{i}:{'=' * (weights[i])}" for i in mine)).encode()
    got = gather_bytes(torch.frombuffer(bytearray(payload), dtype=torch.uint8), dist, dst=0)
    if rank == 0:
        q.put([bytes(t.numpy().tobytes()) for t in got])
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_bytes_two_ranks():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from wfmash_amd.dist import shard_records
    weights = [(i * 7919) % 13 + 1 for i in range(40)]
    shards = shard_records(weights, world)
    assert len(out) == world
    for r in range(world):
        exp = ("|".join(f"rec{i}:{'=' * (weights[i])}" for i in shards[r])).encode()
        assert out[r] == exp
    # every record exactly once
    seen = sorted(int(x.split(b":")[0][3:]) for part in out for x in part.split(b"|"))
    assert seen == list(range(40))


# ---- map path: queries sharded over ranks, text gathered and re-ordered on rank 0 ----

_QUERIES = [(f"s{i}#1#c", 1000 * ((i * 37) % 11 + 1)) for i in range(9)]


def _fake_map(names):
    """stands in for wfmh_map on a query list: a query-dependent number of records per query"""
    return "".join(f"{n}\t{dict(_QUERIES)[n]}\t{j}\trest\n" for n in names for j in range(len(n) % 3 + (dict(_QUERIES)[n] // 4000)))


def _map_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wfmash_amd.dist import map_sharded
    names = [n for n, _ in _QUERIES]
    out = map_sharded(_fake_map, names, [l for _, l in _QUERIES], dist)
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_map_sharded_two_ranks_reproduces_single_process_order():
    from wfmash_amd.dist import map_sharded, shard_queries
    names = [n for n, _ in _QUERIES]
    single = map_sharded(_fake_map, names, [l for _, l in _QUERIES], None)
    assert single == _fake_map(names)
    shards = shard_queries([l for _, l in _QUERIES], 2)
    assert sorted(shards[0] + shards[1]) == list(range(len(names))) and shards[0] and shards[1]
    loads = [sum(_QUERIES[i][1] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= max(l for _, l in _QUERIES)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_map_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert out == single


# ---- the file form: result files gathered in chunks, merged by streaming ----

def _fake_map_file(work):
    def fn(names):
        path = os.path.join(work, f"part.{os.getpid()}.paf")
        with open(path, "w") as f:
            f.write(_fake_map(names))
        return path
    return fn


def _file_worker(rank, world, port, work, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import wfmash_amd.dist as D
    D.CHUNK_BYTES = 37  # many chunks per file
    names = [n for n, _ in _QUERIES]
    rank_dir = os.path.join(work, f"r{rank}")
    os.makedirs(rank_dir, exist_ok=True)
    paths_before = D.gather_files.__defaults__
    D.map_sharded_files(_fake_map_file(rank_dir), names, [l for _, l in _QUERIES], out_path, rank_dir, dist,
                        device=None)
    assert paths_before == D.gather_files.__defaults__
    dist.barrier()
    dist.destroy_process_group()


def test_map_sharded_files_two_ranks(tmp_path):
    import wfmash_amd.dist as D
    names = [n for n, _ in _QUERIES]
    single = str(tmp_path / "single.paf")
    D.map_sharded_files(_fake_map_file(str(tmp_path)), names, [l for _, l in _QUERIES], single, str(tmp_path), None)
    assert open(single).read() == _fake_map(names)
    # a query's records may come in several pieces (one per target subset): pieces keep their order
    a, b = tmp_path / "a", tmp_path / "b"
    a.write_text("q1\tx\nq1\ty\nq2\tz\nq1\tw\n")
    b.write_text("q3\tu\n")
    with open(tmp_path / "m", "wb") as f:
        D.merge_query_block_files([str(a), str(b)], ["q3", "q1", "q2"], f)
    assert (tmp_path / "m").read_text() == "q3\tu\nq1\tx\nq1\ty\nq1\tw\nq2\tz\n"
    world, port = 2, _free_port()
    two = str(tmp_path / "two.paf")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_file_worker, args=(r, world, port, str(tmp_path), two)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert open(two).read() == open(single).read()


def _all_gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wfmash_amd import dist as D
        text = "" if rank == 1 else "q%d\tline of rank %d\n" % (rank, rank) * (3 + rank)
        q.put((rank, D.all_gather_text(text, dist)))
    finally:
        dist.destroy_process_group()


def test_all_gather_text_three_ranks_one_of_them_empty():
    """dist.all_gather_text: the exchange between the query-sharded map phase and the record-sharded align phase of the all-vs-all bench
    (bench.py --config C4) -- every rank ends up with every rank's mapping text in rank order, a rank without queries contributes nothing."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_all_gather_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = ["q0\tline of rank 0\n" * 3, "", "q2\tline of rank 2\n" * 5]
    assert got[0] == want and got[1] == want and got[2] == want
