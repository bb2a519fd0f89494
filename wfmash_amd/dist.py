"""Multi-GPU plumbing of the align path: shard mapping records across ranks and
gather the variable-length PAF payload to rank 0.

Every mapping record is independent (computeAlignments.hpp:398-435), so there
is no data-path collective; the only exchange is the final gather of PAF bytes.
Works with backend "nccl" (= RCCL over xGMI, device tensors) and "gloo" (CPU).
"""
import torch


def shard_records(weights, world_size):
    """Greedy longest-first assignment to the least-loaded rank.

    Same heuristic as scripts/split_approx_mappings_in_chunks.py:19-27,47
    (weight = len * (1 - identity)); callers pass its square for WFA cost.
    Returns a list of index lists, one per rank; deterministic.
    """
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    loads = [0.0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += weights[i]
    for s in shards:
        s.sort()
    return shards


def gather_bytes(payload: torch.Tensor, dist, dst=0):
    """Gather 1-D uint8 tensors of different lengths to rank `dst`.

    Returns a list of per-rank uint8 tensors on `dst`, None elsewhere.
    One all_gather of the sizes + one gather of max-size padded buffers.
    """
    world = dist.get_world_size()
    rank = dist.get_rank()
    dev = payload.device
    size = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=dev)
    buf[:payload.numel()] = payload
    if rank == dst:
        recv = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.gather(buf, gather_list=recv, dst=dst)
        return [recv[r][:sizes[r]] for r in range(world)]
    dist.gather(buf, gather_list=None, dst=dst)
    return None


def all_gather_bytes(payload: torch.Tensor, dist):
    """Every rank ends up with every rank's 1-D uint8 tensor (different lengths), in rank order: one all_gather of the sizes, one of
    the buffers padded to the longest.  The exchange between the two phases of a sharded all-vs-all run: the map phase shards by
    QUERY (computeMap.hpp:565-599), the align phase by RECORD WEIGHT over the records of all queries
    (scripts/split_approx_mappings_in_chunks.py:19-27,47), so every rank needs every rank's mapping records."""
    world = dist.get_world_size()
    dev = payload.device
    size = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=dev)
    buf[:payload.numel()] = payload
    recv = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(recv, buf)
    return [recv[r][:sizes[r]] for r in range(world)]


def all_gather_text(text: str, dist, device=None):
    """all_gather_bytes for text (dist None: [text])."""
    if dist is None:
        return [text]
    payload = torch.frombuffer(bytearray(text.encode()), dtype=torch.uint8) if text else torch.zeros(0, dtype=torch.uint8)
    if device is not None:
        payload = payload.to(device)
    return [bytes(p.cpu().numpy().tobytes()).decode() for p in all_gather_bytes(payload, dist)]


# ---- map path: the queries shard, the target index is replicated on every rank ----

def shard_queries(lengths, world_size):
    """Query sequences are independent units of the map phase (one Taskflow task per query,
    computeMap.hpp:527-688): greedy longest-first onto the least-loaded rank, weight = sequence
    length (mapping cost is linear in the number of fragments)."""
    return shard_records([float(x) for x in lengths], world_size)


def merge_query_blocks(rank_texts, query_order):
    """Reassembles the single-GPU record order from per-rank mapping PAF texts: a rank prints the
    records of one query as one consecutive block (reportReadMappings runs once per query), and the
    single-GPU run prints the blocks in query order.  query_order: all query names in file order."""
    blocks = {}
    for text in rank_texts:
        for line in text.splitlines(keepends=True):
            blocks.setdefault(line.split("\t", 1)[0], []).append(line)
    return "".join("".join(blocks.get(q, [])) for q in query_order)


def map_sharded(map_fn, query_names, query_lengths, dist=None, device=None):
    """Runs the map phase with the queries sharded over the ranks of `dist` (None = single process).

    map_fn(names) -> mapping PAF text of those queries against ALL targets (e.g. a closure over
    capi.map_paf with a -A style query list; every rank builds the same target index).  No
    data-path collective: the only exchange is the gather of the PAF text to rank 0, which returns
    the text in single-GPU record order; other ranks return None."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    shards = shard_queries(query_lengths, world)
    mine = [query_names[i] for i in shards[rank]]
    text = map_fn(mine) if mine else ""
    if dist is None:
        return merge_query_blocks([text], query_names)
    payload = torch.frombuffer(bytearray(text.encode()), dtype=torch.uint8) if text else torch.zeros(0, dtype=torch.uint8)
    if device is not None:
        payload = payload.to(device)
    parts = gather_bytes(payload, dist, dst=0)
    if parts is None:
        return None
    return merge_query_blocks([bytes(p.cpu().numpy().tobytes()).decode() for p in parts], query_names)


# ---- the same without holding a rank's output in memory: result files, sent in chunks ----

CHUNK_BYTES = 64 << 20


def gather_files(path, dist, work_dir, dst=0, device=None, chunk_bytes=CHUNK_BYTES):
    """Every rank has its records in the file `path`; rank `dst` ends up with one file per rank (its own is `path`
    itself) and returns their paths in rank order, other ranks return None.  One all_gather of the sizes, then each
    file travels as point-to-point chunks of at most chunk_bytes (RCCL send/recv over xGMI with device tensors,
    gloo on the host): memory stays at one chunk per rank whatever the size of the output."""
    import os
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    size = torch.tensor([os.path.getsize(path)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(x.item()) for x in sizes]
    if rank == dst:
        paths = []
        for r in range(world):
            if r == dst:
                paths.append(path)
                continue
            dest = os.path.join(work_dir, f"gathered.rank{r}")
            with open(dest, "wb") as f:
                left = sizes[r]
                while left > 0:
                    n = min(left, chunk_bytes)
                    buf = torch.empty(n, dtype=torch.uint8, device=dev)
                    dist.recv(buf, src=r)
                    f.write(buf.cpu().numpy().tobytes())
                    left -= n
            paths.append(dest)
        return paths
    with open(path, "rb") as f:
        left = sizes[rank]
        while left > 0:
            n = min(left, chunk_bytes)
            buf = torch.frombuffer(bytearray(f.read(n)), dtype=torch.uint8).to(dev)
            dist.send(buf, dst=dst)
            left -= n
    return None


def merge_query_block_files(paths, query_order, out):
    """merge_query_blocks on files: one pass over every file notes where the records of each query lie (a rank prints a
    query's records consecutively, once per target subset), a second pass copies them to `out` (binary file object) in
    query order, a query's pieces in (rank, position) order.  Only the table of pieces is held in memory."""
    pieces = {}
    for fi, p in enumerate(paths):
        with open(p, "rb") as f:
            pos = 0
            cur, start = None, 0
            for line in f:
                q = line.split(b"\t", 1)[0]
                if q != cur:
                    if cur is not None:
                        pieces.setdefault(cur, []).append((fi, start, pos - start))
                    cur, start = q, pos
                pos += len(line)
            if cur is not None:
                pieces.setdefault(cur, []).append((fi, start, pos - start))
    handles = [open(p, "rb") for p in paths]
    try:
        for q in query_order:
            for fi, off, n in pieces.get(q.encode() if isinstance(q, str) else q, []):
                handles[fi].seek(off)
                left = n
                while left > 0:
                    b = handles[fi].read(min(left, 16 << 20))
                    out.write(b)
                    left -= len(b)
    finally:
        for h in handles:
            h.close()


def map_sharded_files(map_fn, query_names, query_lengths, out_path, work_dir, dist=None, device=None):
    """map_sharded for outputs that should not be held in memory: map_fn(names) -> path of a file with the records of
    those queries (None when there are none); rank 0 writes out_path in single-GPU record order."""
    import os
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    shards = shard_queries(query_lengths, world)
    mine = [query_names[i] for i in shards[rank]]
    path = map_fn(mine) if mine else None
    if path is None:
        path = os.path.join(work_dir, f"empty.rank{rank}")
        open(path, "wb").close()
    paths = [path] if dist is None else gather_files(path, dist, work_dir, dst=0, device=device)
    if paths is not None:
        with open(out_path, "wb") as out:
            merge_query_block_files(paths, query_names, out)
