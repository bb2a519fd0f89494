"""Map + align a pangenome FASTA on all GPUs of a node: one process per GPU, queries sharded over the ranks,
every rank builds the (replicated) target index, maps and aligns its own queries, and the PAF text is
gathered to rank 0 (independent units, no data-path collective; RCCL over xGMI when the
backend is nccl).  Rank 0 writes the records in the order a single-GPU run prints them.

    python scripts/pangenome_run.py target.fa --out out.paf                     # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        scripts/pangenome_run.py target.fa --out out.paf                        # 8 GPUs

With fewer GPUs than ranks (tests on a one-GPU box) the ranks share the devices and the gather runs over gloo.
Not supported across ranks: the one-to-one filter (it needs all queries' mappings in one place)."""
import argparse
import json
import os
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
    ap.add_argument("--threads", type=int, default=0, help="host threads per rank; 0 = cores / ranks")
    ap.add_argument("-b", "--batch", type=int, default=0, help="target batch size (-b)")
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
    threads = a.threads or max(1, (os.cpu_count() or 1) // world)
    names, lengths = sequence_table(a.target)
    work = tempfile.mkdtemp(prefix=f"wfmash_rank{rank}_")
    h = capi.Handle(dev_index)
    stats = {"records": 0, "aligned_bp": 0, "ms_map": 0.0, "ms_align": 0.0}

    def map_and_align(mine):
        qlist = os.path.join(work, "queries.txt")
        with open(qlist, "w") as f:
            f.write("\n".join(mine) + "\n")
        over = dict(threads=threads, query_list=qlist)
        if a.pct:
            over.update(percentage_identity=a.pct, auto_pct_identity=0)
        if a.batch:
            over.update(index_by_size=a.batch)
        mapping = os.path.join(work, "map.paf")
        t0 = time.time()
        capi.map_paf(h, a.target, mapping, params=capi.map_default_params(**over))
        stats["ms_map"] = (time.time() - t0) * 1e3
        result = mapping
        if not a.approx_mapping:
            result = os.path.join(work, "aln.paf")
            t0 = time.time()
            s = capi.align_paf(h, a.target, mapping, result, params={"threads": threads})
            stats["ms_align"] = (time.time() - t0) * 1e3
            stats["records"], stats["aligned_bp"] = int(s.records), int(s.aligned_bp)
        return open(result).read()

    t_all = time.time()
    text = wdist.map_sharded(map_and_align, names, lengths, dist=td, device=gather_device)
    if td is not None:
        import torch
        t = torch.tensor([stats["records"], stats["aligned_bp"]], dtype=torch.int64, device=gather_device or "cpu")
        td.all_reduce(t)
        stats["records"], stats["aligned_bp"] = int(t[0]), int(t[1])
    wall = time.time() - t_all
    h.close()
    if rank == 0:
        with open(a.out, "w") as f:
            f.write(text)
        print(json.dumps({"ranks": world, "queries": len(names), "lines": text.count("\n"), "records": stats["records"],
                          "aligned_bp": stats["aligned_bp"], "wall_s": round(wall, 2),
                          "aligned_bp_per_s": round(stats["aligned_bp"] / wall) if wall > 0 else 0,
                          "rank0_ms_map": round(stats["ms_map"]), "rank0_ms_align": round(stats["ms_align"])}), file=sys.stderr)
    if td is not None:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
