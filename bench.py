#!/usr/bin/env python
"""bench.py -- aligned bases/sec of the wfmash align hot path on MI355X.

A "step" is one pass of the hot path (BiWFA gap-affine-2p alignment, penalties
5,8,2,24,1; wflign.cpp:136-148) over one batch of synthetic mapping records:
BASELINE.json configs[2], "synthetic 5%-divergence 64x50kb segment pairs,
WFA-only (mappings pre-supplied), 1 GPU" (generator: SURVEY.md 8d / wfmash_amd/synth.py).
configs[1] (LPA.subset all-vs-all) needs the reference's data file, which does
not travel to the GPU box; it is covered as a parity case, not a bench line.

Sequences are resident in HBM before the timed region (wfm_upload_sequences);
the timed region is K calls of wfm_align_resident (all recursion levels, the
backtrace, the CIGAR gather and the D2H copy of the CIGARs).

N>1: one process per GPU (torch.distributed, backend nccl = RCCL); every rank
aligns its own shard of mapping records (weak scaling, no data-path
collective) and the PAF-side payload (run-length CIGARs) is gathered to rank 0
inside the timed region, as the reference's cluster sharding would
(scripts/split_approx_mappings_in_chunks.py).  `python bench.py --gpus N` started
without a torch.distributed environment launches the N ranks itself (it re-executes
under `python -m torch.distributed.run --nproc-per-node N`, rendezvous on 127.0.0.1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=["C3", "C5"])
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs in the cpu_baseline sample (0 = auto)")
    ap.add_argument("--rank-check", action="store_true",
                    help="launch / join the ranks, print one line per rank and stop (no GPU needed: gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start one process per GPU ourselves
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
        raise SystemExit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch
    from wfmash_amd import capi, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rank_check:
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            seen = [None] * world
            dist.all_gather_object(seen, rank)
            dist.barrier()
            dist.destroy_process_group()
        else:
            seen = [0]
        sys.stdout.flush()
        os.write(1, (json.dumps({"rank_check": True, "rank": rank, "n_gpus": world, "ranks_seen": seen}) + "\n").encode())  # one write per line
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    ndev = torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > ndev and os.environ.get("WFM_BENCH_SHARE_GPU") != "1":
            raise SystemExit(f"bench.py: --gpus {world} but only {ndev} device(s) visible (WFM_BENCH_SHARE_GPU=1 lets ranks share "
                             "devices over gloo, for testing the launch path on one GPU)")
        local_rank = local_rank % ndev
        torch.cuda.set_device(local_rank)
        if world > ndev:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # collectives run on device tensors over RCCL; the one-GPU launch test (gloo) keeps them on the host
    comm_dev = dev if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")

    h = capi.Handle(local_rank)
    # this rank's shard of the mapping records (distinct seeds per rank)
    all_pairs = synth.pairs(args.config, n_pairs=args.pairs * world)
    mine = all_pairs[rank * args.pairs:(rank + 1) * args.pairs]
    seqset = h.upload(mine)
    query_bases = sum(len(q) for _, q in mine)  # "total aligned bp" = sum of query spans (computeAlignments.hpp:481,528)

    def gather_payload(seqset):
        """PAF-side payload gather to rank 0 (variable-length byte buffers)."""
        if dist is None:
            return
        from wfmash_amd.dist import gather_bytes
        n_bytes = sum(int(seqset.results[i].ops_len) for i in range(seqset.n))
        payload = seqset.arena[:n_bytes]
        gather_bytes(torch.from_numpy(payload).to(comm_dev), dist, dst=0)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        h.align_resident(seqset, collect=False)
        gather_payload(seqset)
    sync()
    t0 = time.perf_counter()
    cells_bp = 0
    cells_tile = 0
    ms_tile = 0.0
    ms_tile_busy = 0.0
    streams = 1
    tile_launches = 0
    ms_bp = 0.0
    ms_base = 0.0
    cells_total = 0
    bp_launches = 0
    for _ in range(args.steps):
        failed = h.align_resident(seqset, collect=False)
        if failed:
            raise SystemExit(f"{failed} alignments failed")
        st = h.stats()
        cells_bp += st.cells_bp
        cells_tile += st.cells_tile
        ms_tile += st.ms_tile
        ms_tile_busy += st.ms_tile_busy
        streams = max(streams, st.streams)
        tile_launches += st.tile_launches
        cells_total += st.cells
        ms_bp += st.ms_breakpoint
        ms_base += st.ms_base
        bp_launches += st.bp_launches
        gather_payload(seqset)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    out = None
    if rank == 0:
        value = query_bases * world * args.steps / dt
        # roofline of the dominant kernel: algorithmic bytes =
        # 48 B per computed (score,diagonal) cell (7 loads + 5 stores of int32
        # offsets) + the sequences read once per launch
        seq_bytes = sum(len(p) + len(q) for p, q in mine) * 2  # forward + reversed copies
        # dominant kernel: the time-tiled phase-1 kernel when it ran (default), else the step kernel
        if ms_tile > 0.5 * ms_bp:
            # With the batch split over two streams, launches of the kernel overlap in time: the kernel's bandwidth
            # is the bytes all its launches handle / the time during which it was running at all (the union of the
            # launch intervals, HIP events against one time origin), not / the sum of the stretched durations.
            dom, dom_cells, dom_ms, dom_launches = "wfa_tile_reg_kernel", cells_tile, (ms_tile_busy if streams > 1 else ms_tile), tile_launches
        else:
            dom, dom_cells, dom_ms, dom_launches = "wfa_bp_kernel", cells_bp - cells_tile, ms_bp - ms_tile, bp_launches
        alg_bytes = 48.0 * dom_cells + seq_bytes * args.steps
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        peak = 8000.0
        traffic = None  # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r1g_traffic.json")))
            if tj.get("kernel") == dom and args.config == "C3" and args.pairs == 64:
                traffic = tj["traffic_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "aligned bases/sec (whole node) + CIGAR-identical rate vs CPU ref",
            "value": value, "unit": "aligned bases/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {args.pairs} synthetic "
                                   f"{'5%' if args.config == 'C3' else '15%'}-divergence "
                                   f"{'50' if args.config == 'C3' else '100'}kb segment pairs per GPU, WFA-only "
                                   "(BiWFA gap-affine-2p 5,8,2,24,1; mappings pre-supplied)",
                       "pairs_per_gpu": args.pairs, "parallelism": f"records sharded over {world} GPU(s)"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": peak,
                         "unit": "GB/s", "frac": achieved / peak, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes / max(dom_launches, 1),
                         "cells_per_launch": dom_cells / max(dom_launches, 1),
                         "avg_launch_ms": (ms_tile if dom == "wfa_tile_reg_kernel" else dom_ms) / max(dom_launches, 1),
                         "launches": dom_launches,
                         "streams": streams,
                         "kernel_busy_ms_per_step": dom_ms / args.steps,
                         "note": "achieved = 48 B x computed (score,diagonal) cells / time the kernel was running (SURVEY 8d). "
                                 "The batch runs as up to three parts on as many streams, so launches of this kernel overlap each other: "
                                 "avg_launch_ms (what rocprofv3 shows per launch) is stretched by the sharing, the running "
                                 "time is the union of the launch intervals from HIP events; WFM_OVERLAP=0 gives the "
                                 "exclusive figure (0.93, profiles/r1d_align.md); profiles/r1g_align.md is this configuration. The tiled kernel keeps wavefront history "
                                 "in registers, so real HBM traffic (traffic, per launch) is far below the algorithmic bytes"},
            "kernel_ms_per_step": {"wfa_tile_reg_kernel": ms_tile / args.steps, "wfa_bp_kernel": (ms_bp - ms_tile) / args.steps,
                                   "wfa_base_kernel": ms_base / args.steps},
            "cells_per_step": cells_total / args.steps,
            "device": h.device_name(),
        }
        # ---- CPU baseline (oracle = "port") on a bounded sample of the same workload ----
        if not args.no_cpu_baseline:
            from oracle import pyoracle as O
            cores = os.cpu_count() or 1
            n_s = args.cpu_sample or min(len(mine), max(8, cores))
            threads = min(cores, n_s)
            sample = mine[:n_s]
            t1 = time.perf_counter()
            ops, scores, cst, failed = O.align_batch_biwfa([p for p, _ in sample], [q for _, q in sample], nthreads=threads)
            cdt = time.perf_counter() - t1
            res = h._collect(seqset)
            ident = sum(1 for i in range(n_s) if res[i].ops == ops[i])
            score_ident = sum(1 for i in range(n_s) if res[i].score == int(scores[i]))
            out["cpu_baseline"] = {"value": sum(len(q) for _, q in sample) / cdt, "unit": "aligned bases/s",
                                   "cores": threads, "kind": "port",
                                   "sample": f"first {n_s} of the {len(mine)} {args.config} pairs, oracle/wfa2p.c BiWFA, "
                                             f"{threads} OpenMP threads, {cdt:.1f} s",
                                   "host_cpu": _cpu_model()}
            out["cigar_identical_rate"] = ident / n_s
            out["score_identical_rate"] = score_ident / n_s
        print(json.dumps(out), flush=True)
    seqset.free()
    h.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
