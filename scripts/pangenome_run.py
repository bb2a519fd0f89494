"""Map + align a pangenome FASTA on all GPUs of a node: one process per GPU (torch.distributed), queries sharded over
the ranks (independent units, no data-path collective).

  * the target index is built ONCE per node: rank 0 estimates the identity threshold (-p ani50-2), builds the index of
    every target subset and writes it as a `-W` index file into the node-local work directory; after a barrier every
    rank reads it (`-I`: wfm_index_upload) instead of winnowing the targets again (the index is read-only while mapping,
    computeMap.hpp:431-484);
  * every rank maps and aligns the queries dist.shard_queries gives it;
  * the records travel to rank 0 as chunks of the per-rank result files (RCCL send/recv over xGMI when the backend is
    nccl) and rank 0 writes them in the order a single-GPU run prints them -- no rank holds a whole output in memory.

    python scripts/pangenome_run.py target.fa --out out.paf                     # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        scripts/pangenome_run.py target.fa --out out.paf                        # 8 GPUs

`wfmash-hip --gpus N` does the same inside one process (host threads, device-to-device index copies); this script is the
one-process-per-GPU form.  With fewer GPUs than ranks (tests on a one-GPU box) the ranks share the devices and the gather
runs over gloo.  Not supported across ranks: the one-to-one filter (it needs all queries' mappings in one place).  With
several target subsets (-b) the records of a query are printed together, where a single-GPU run prints them subset
after subset."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, dist as wdist  # noqa: E402


def sequence_table(fasta):
    """names and lengths in file order (the .fai when there is one)"""
    rows = capi.host_fasta(fasta).splitlines()[1:]
    names = [r.split("\t")[0] for r in rows]
    return names, [int(r.split("\t")[1]) for r in rows]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("target")
    ap.add_argument("--out", default="/dev/stdout")
    ap.add_argument("-m", "--approx-mapping", action="store_true", help="mapping only")
    ap.add_argument("--pct", type=float, default=0.0, help="identity threshold as a fraction; 0 = estimate (ani50-2)")
    ap.add_argument("--threads", type=int, default=0, help="host threads per rank; 0 = cores / ranks (rank 0 uses all cores while it builds the index)")
    ap.add_argument("-b", "--batch", type=int, default=0, help="target batch size (-b)")
    ap.add_argument("--index-per-rank", action="store_true", help="every rank builds its own index (the round-1 behaviour, for comparison)")
    ap.add_argument("--work", default="", help="node-local directory for the index file and the per-rank results [a fresh one under $TMPDIR]")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    td = None
    dev_index = 0
    gather_device = None
    if world > 1:
        import torch
        import torch.distributed as td
        ngpu = torch.cuda.device_count()
        if ngpu == 0:
            raise SystemExit("no GPU visible (there is no CPU fallback)")
        dev_index = local % ngpu
        if ngpu >= world:
            torch.cuda.set_device(dev_index)
            td.init_process_group("nccl")  # = RCCL
            gather_device = torch.device("cuda", dev_index)
        else:
            td.init_process_group("gloo")
    cores = os.cpu_count() or 1
    threads = a.threads or max(1, cores // world)
    names, lengths = sequence_table(a.target)
    # one work directory per node, made by rank 0 and announced to the others
    shared = [a.work or (tempfile.mkdtemp(prefix="wfmash_node_") if rank == 0 else None)]
    if td is not None:
        td.broadcast_object_list(shared, src=0)
    shared = shared[0]
    work = os.path.join(shared, f"rank{rank}")
    os.makedirs(work, exist_ok=True)
    h = capi.Handle(dev_index)
    stats = {"records": 0, "aligned_bp": 0, "ms_index": 0.0, "ms_map": 0.0, "ms_align": 0.0}
    ok = False
    t_all = time.time()
    try:
        base = {}
        if a.pct:
            base.update(percentage_identity=a.pct, auto_pct_identity=0)
        if a.batch:
            base.update(index_by_size=a.batch)
        index_file = None
        if world > 1 and not a.index_per_rank:
            # ---- the index once per node: rank 0 builds and writes it, everybody reads it
            index_file = os.path.join(shared, "targets.idx")
            decided = [None]
            if rank == 0:
                t0 = time.time()
                s = capi.map_paf(h, a.target, os.path.join(work, "unused.paf"),
                                 params=capi.map_default_params(threads=a.threads or cores, index_file=index_file, write_index=1, **base))
                stats["ms_index"] = (time.time() - t0) * 1e3
                decided = [(float(s.percentage_identity), int(s.sketch_size))]
            td.broadcast_object_list(decided, src=0)  # doubles as the barrier after the file is complete
            pct, sketch = decided[0]
            base.update(percentage_identity=pct, auto_pct_identity=0, sketch_size=sketch, index_file=index_file, write_index=0)

        def map_and_align(mine):
            qlist = os.path.join(work, "queries.txt")
            with open(qlist, "w") as f:
                f.write("\n".join(mine) + "\n")
            mapping = os.path.join(work, "map.paf")
            t0 = time.time()
            capi.map_paf(h, a.target, mapping, params=capi.map_default_params(threads=threads, query_list=qlist, **base))
            stats["ms_map"] = (time.time() - t0) * 1e3
            result = mapping
            if not a.approx_mapping:
                result = os.path.join(work, "aln.paf")
                t0 = time.time()
                s = capi.align_paf(h, a.target, mapping, result, params={"threads": threads})
                stats["ms_align"] = (time.time() - t0) * 1e3
                stats["records"], stats["aligned_bp"] = int(s.records), int(s.aligned_bp)
            return result

        wdist.map_sharded_files(map_and_align, names, lengths, a.out, work, dist=td, device=gather_device)
        if td is not None:
            import torch
            t = torch.tensor([stats["records"], stats["aligned_bp"]], dtype=torch.int64, device=gather_device or "cpu")
            td.all_reduce(t)
            stats["records"], stats["aligned_bp"] = int(t[0]), int(t[1])
        wall = time.time() - t_all
        if rank == 0:
            lines = 0
            if a.out != "/dev/stdout":
                with open(a.out, "rb") as f:
                    lines = sum(chunk.count(b"\n") for chunk in iter(lambda: f.read(1 << 24), b""))
            print(json.dumps({"ranks": world, "queries": len(names), "lines": lines, "records": stats["records"],
                              "aligned_bp": stats["aligned_bp"], "wall_s": round(wall, 2),
                              "aligned_bp_per_s": round(stats["aligned_bp"] / wall) if wall > 0 else 0,
                              "index_once_per_node": index_file is not None, "rank0_ms_index": round(stats["ms_index"]),
                              "rank0_ms_map": round(stats["ms_map"]), "rank0_ms_align": round(stats["ms_align"])}), file=sys.stderr)
        ok = True
    finally:
        h.close()
        if td is not None and ok:
            td.barrier()  # nobody removes the node's directory while another rank still reads from it
        shutil.rmtree(work, ignore_errors=True)
        if rank == 0 and not a.work:
            shutil.rmtree(shared, ignore_errors=True)
        if td is not None:
            td.destroy_process_group()


if __name__ == "__main__":
    main()
