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
