#!/usr/bin/env python
"""bench.py -- aligned bases/sec of the wfmash align hot path on MI355X.

A "step" is one pass of the hot path (BiWFA gap-affine-2p alignment, penalties
5,8,2,24,1; wflign.cpp:136-148) over one batch of synthetic mapping records:
BASELINE.json configs[2], "synthetic 5%-divergence 64x50kb segment pairs,
WFA-only (mappings pre-supplied), 1 GPU" (generator: SURVEY.md 8d / wfmash_amd/synth.py).
The other configs are reported in the same line as `secondary` legs, driver-timed end to end
through the C ABI and each with a sampled parity check against the oracles: C1 (substitute:
synth.yeast_like, the data blob is absent), C2 (LPA.subset all-vs-all, the reference's test data,
a committed fixture), one rank of C4 at two sizes, C5; `value` stays C3, the largest
configuration BASELINE.json quotes on one GPU.

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

# The bound the tile kernel is measured against: vector-instruction issue.  A cell cannot cost fewer than VALU_FLOOR vector
# instructions in the present scheme (recurrences 9, score-bound / column test 2, one 16-base probe of both sequences 17, extension add 1,
# antidiagonal 2, its share of the neighbour exchange 4, of the block's maximum 1); a wave64 instruction of the step's mix occupies its SIMD for
# VALU_MEAN_ISSUE_CYCLES on average (measured: profiles/r6_valu_issue.md -- 4.2 cycles for max / min / compare / select / alignbit / ffbl / DPP /
# lshl, 2.2 for add / sub / and / or / xor / lshr / mov -- priced over the step's common path by scripts/isa_hot_path.py: 3.34 - 3.54).
VALU_FLOOR = 36.0
VALU_MEAN_ISSUE_CYCLES = 3.4
VALU_PEAK_CELLS = 256 * 4 * 2.4e9 * 64.0 / (VALU_FLOOR * VALU_MEAN_ISSUE_CYCLES)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)  # (the first passes of a process still size arenas and ramp clocks: one warm-up pass left 4 - 6 % in the first process of a box)
    ap.add_argument("--config", default="C3", choices=["C3", "C5", "C4"],
                    help="C3 / C5: synthetic pairs, WFA-only.  C4 (strong scaling by nature): a synthetic pangenome of eight haplotypes (--c4-mbp each) "
                         "all-vs-all, map + align inside the timed region -- queries dealt out over the ranks for the map phase, the mapping records of all "
                         "queries for the align phase (dist.shard_queries / dist.shard_records on the reference's own weights)")
    ap.add_argument("--out-paf", default="", help="--config C4: rank 0 writes the gathered alignment records of the last step here, sorted")
    ap.add_argument("--c4-mbp", type=float, default=8.0, help="haplotype length of --config C4 in Mbp (248.956422 = north_star's size)")
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --pairs records per GPU (the driver's SCALE runs); strong: one file of --total-pairs records "
                         "sharded over the ranks by dist.shard_records, the same file at every N")
    ap.add_argument("--total-pairs", type=int, default=512, help="records of the strong-scaling file (64 x 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs in the cpu_baseline sample (0 = auto)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (C5, C1 substitute, C4 ranks, C2)")
    ap.add_argument("--legs", default="", help="comma-separated subset of the secondary legs to run (C5,C1_substitute,C4_rank_scaled,C4_rank_40mbp,C4_rank_full,C4_all_vs_all,C2,map_parity); default all")
    ap.add_argument("--no-full-c4", action="store_true", help="skip the chr1-sized C4 rank among the secondary legs (2 GB of synthetic haplotypes: ~20 s to make)")
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
        shard = _shard(args, rank, world, None)
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
        os.write(1, (json.dumps({"rank_check": True, "rank": rank, "n_gpus": world, "ranks_seen": seen, "scaling": args.scaling,
                                 "records": shard}) + "\n").encode())  # one write per line
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
    if args.config == "C4":
        _strong_c4(args, h, capi, synth, dist, rank, world, comm_dev, torch)
        h.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    # this rank's shard of the mapping records (weak: distinct seeds per rank; strong: its share of the one file)
    n_all = args.pairs * world if args.scaling == "weak" else args.total_pairs
    all_pairs = synth.pairs(args.config, n_pairs=n_all)
    mine = [all_pairs[i] for i in _shard(args, rank, world, all_pairs)]
    seqset = h.upload(mine)
    query_bases = sum(len(q) for _, q in mine)  # "total aligned bp" = sum of query spans (computeAlignments.hpp:481,528)
    job_bases = sum(len(q) for _, q in all_pairs)  # of all ranks

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
    acc = _StatAcc()
    for _ in range(args.steps):
        failed = h.align_resident(seqset, collect=False)
        if failed:
            raise SystemExit(f"{failed} alignments failed")
        acc.add(h.stats())
        gather_payload(seqset)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    out = None
    if rank == 0:
        value = job_bases * args.steps / dt
        seq_bytes = sum(len(p) + len(q) for p, q in mine) * 2  # forward + reversed copies
        # one more, untimed, pass with the parts of the batch one after the other on one stream (WFM_OVERLAP=0): every
        # launch then has the GPU to itself, which is what rocprofv3's per-launch durations of that mode show
        excl = _StatAcc()
        os.environ["WFM_OVERLAP"] = "0"
        try:
            for _ in range(2):
                h.align_resident(seqset, collect=False)
                excl.add(h.stats())
        finally:
            del os.environ["WFM_OVERLAP"]
        roof = _roofline(acc, excl, seq_bytes, args)
        out = {
            "metric": "aligned bases/sec (whole node) + CIGAR-identical rate vs CPU ref",
            "value": value, "unit": "aligned bases/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {args.pairs if args.scaling == 'weak' else args.total_pairs} synthetic "
                                   f"{'5%' if args.config == 'C3' else '15%'}-divergence "
                                   f"{'50' if args.config == 'C3' else '100'}kb segment pairs "
                                   f"{'per GPU' if args.scaling == 'weak' else 'in all, sharded over the GPUs'}, WFA-only "
                                   "(BiWFA gap-affine-2p 5,8,2,24,1; mappings pre-supplied)",
                       "pairs_per_gpu": len(mine), "pairs_total": n_all, "parallelism": f"records sharded over {world} GPU(s)"},
            "roofline": roof,
            # time during which at least one launch of the kernel was running (union of the launch intervals over all
            # streams, HIP events against one origin), and the sum of the individual launch durations
            "kernel_busy_ms_per_step": {"wfa_tile2_kernel": acc.ms_tile_busy / args.steps, "wfa_bp_kernel": acc.ms_bp_busy / args.steps,
                                        "wfa_base_kernel": acc.ms_base_busy / args.steps, "any": acc.ms_any_busy / args.steps},
            "kernel_ms_per_step": {"wfa_tile2_kernel": acc.ms_tile / args.steps, "wfa_bp_kernel": (acc.ms_bp - acc.ms_tile) / args.steps,
                                   "wfa_base_kernel": acc.ms_base / args.steps},
            "cells_per_step": acc.cells / args.steps,
            "whole_step": {"algorithmic_GBps": (48.0 * acc.cells_unique + seq_bytes * args.steps) / dt / 1e9,
                           "frac": (48.0 * acc.cells_unique + seq_bytes * args.steps) / dt / 1e9 / 8000.0,
                           "note": "48 B x unique cells of all kernels / wall time of the step"},
            "device": h.device_name(),
        }
        # ---- CPU baseline (oracle = "port") on a bounded sample of the same workload ----
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed on rank 0 at N = 1 only
            from oracle import pyoracle as O
            cores = os.cpu_count() or 1
            n_s = args.cpu_sample or min(len(mine), max(8, cores))
            threads = min(cores, n_s)
            sample = mine[:n_s]
            t1 = time.perf_counter()
            ops, scores, cst, failed = O.align_batch_biwfa([p for p, _ in sample], [q for _, q in sample], nthreads=threads)
            cdt = time.perf_counter() - t1
            h.align_resident(seqset, collect=False)
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
            out["cpu_baseline_map"] = _cpu_baseline_map(h)
        if not args.no_secondary and world == 1:
            out["secondary"] = _secondary(h, capi, synth, full_c4=not args.no_full_c4, only=set(x for x in args.legs.split(",") if x))
            out["legs"] = _legs_summary(out)  # the last thing in the line: a reader of its tail sees every leg
        print(json.dumps(out), flush=True)
    seqset.free()
    h.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _strong_c4(args, h, capi, synth, dist, rank, world, comm_dev, torch):
    """north_star's C4 as one job: a synthetic pangenome of eight haplotypes, ALL-VS-ALL (`-Y '#'`), both phases inside the timed region.
    Map: the eight query haplotypes are dealt out over the ranks by length (dist.shard_queries: queries are independent tasks,
    computeMap.hpp:565-599), every rank builds the index of all eight targets (replicated) and maps its queries; the mapping records
    of all ranks are exchanged (dist.all_gather_text: the one real exchange step of the path); align: the records of ALL queries are dealt
    out by dist.shard_records on the weight the reference's cluster sharding uses (length x (1 - identity),
    scripts/split_approx_mappings_in_chunks.py:19-27,47; squared for WFA cost), every rank aligns its share end to end
    (wfmh_align_paf: sequence fetch, device batches, CIGAR surgery, PAF text) and the PAF text is gathered to rank 0.
    value = aligned bp of all ranks / the slowest rank's time for map + exchange + align + gather."""
    import tempfile
    from wfmash_amd import dist as D
    threads = max(1, (os.cpu_count() or 1) // max(1, world))
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "c4.fa")
        recs = synth.pangenome_parallel(8, int(args.c4_mbp * 1e6), n_sv=6 if args.c4_mbp <= 8 else 20, workers=min(8, threads))
        names, lengths = synth.write_fasta(fa, recs)
        del recs
        my_queries = [names[i] for i in D.shard_queries(lengths, world)[rank]]
        ql = os.path.join(td, f"q.rank{rank}.txt")
        open(ql, "w").write("".join(n + "\n" for n in my_queries))
        m_mine = os.path.join(td, f"m.rank{rank}.paf")
        mine = os.path.join(td, f"m.align.rank{rank}.paf")
        out_paf = os.path.join(td, f"a.rank{rank}.paf")
        state = {"lines": [], "shard": [], "t_map": 0.0, "t_xchg": 0.0, "ms": None}

        def weight(line):
            f = line.split("\t")
            ident = next((float(x[5:]) for x in f[12:] if x.startswith("id:f:")), 0.9)
            return (max(1.0, (int(f[3]) - int(f[2])) * max(1e-3, 1.0 - ident))) ** 2

        def one_pass():
            t1 = time.perf_counter()
            text = ""
            if my_queries:
                state["ms"] = capi.map_paf(h, fa, m_mine, params=capi.map_default_params(threads=threads, query_list=ql))
                text = open(m_mine).read()
            t2 = time.perf_counter()
            texts = D.all_gather_text(text, dist, device=comm_dev if comm_dev.type == "cuda" else None)
            lines = D.merge_query_blocks(texts, names).splitlines(keepends=True)
            shard = D.shard_records([weight(l) for l in lines], world)[rank]
            open(mine, "w").write("".join(lines[i] for i in shard))
            t3 = time.perf_counter()
            al = capi.align_paf(h, fa, mine, out_paf, params={"threads": threads})
            paths = [out_paf]
            if dist is not None:
                paths = D.gather_files(out_paf, dist, td, dst=0, device=comm_dev if comm_dev.type == "cuda" else None)
            state.update(lines=lines, shard=shard, paths=paths)
            state["t_map"] += t2 - t1
            state["t_xchg"] += t3 - t2
            return al

        def sync():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            one_pass()
        sync()
        state["t_map"] = state["t_xchg"] = 0.0
        t0 = time.perf_counter()
        bp = 0
        recs_n = 0
        ms_gpu = 0.0
        cells = cells_tile = launches = 0
        ms_tile = 0.0
        for _ in range(args.steps):
            al = one_pass()
            bp += int(al.aligned_bp); recs_n += int(al.records); ms_gpu += al.ms_gpu
            cells += int(al.cells); cells_tile += int(al.cells_tile); launches += int(al.tile_launches); ms_tile += al.ms_tile
        sync()
        dt = time.perf_counter() - t0
        lines, shard = state["lines"], state["shard"]
        roof = cpu = parity = None
        if rank == 0 and args.out_paf:  # the job's whole output (the records of all ranks, sorted: a rank's batches finish in any order) for the tests
            recs_all = []
            for pth in state["paths"]:
                recs_all += open(pth).read().splitlines(keepends=True)
            open(args.out_paf, "w").write("".join(sorted(recs_all)))
        if rank == 0:
            # the dominant kernel of the path (the tile kernels), from the run's own HIP events: 48 B x the unique cells of its launches / the sum of
            # their durations (launches of the four workers overlap: the sum is an upper bound of the kernel's own time, the figure a lower bound)
            peak = 8000.0
            ach = 48.0 * cells_tile / (ms_tile * 1e-3) / 1e9 if ms_tile > 0 else 0.0
            peak_cells = VALU_PEAK_CELLS
            ach_cells = cells_tile / (ms_tile * 1e-3) if ms_tile > 0 else 0.0
            roof = {"bound": "valu", "kernel": "wfa_tile2_kernel (rank 0)", "achieved": ach_cells, "peak": peak_cells, "unit": "cells/s", "frac": ach_cells / peak_cells,
                    "floor_valu_insts_per_cell": VALU_FLOOR, "mean_issue_cycles_per_valu_inst": VALU_MEAN_ISSUE_CYCLES,
                    "traffic": None, "launches_per_step": launches / max(1, args.steps), "avg_launch_ms": ms_tile / max(1, launches),
                    "hbm_yardstick": {"bound": "hbm", "achieved": ach, "peak": peak, "unit": "GB/s", "frac": ach / peak,
                                      "algorithmic_bytes_per_launch": 48.0 * cells_tile / max(1, launches),
                                      "note": "48 B per cell (SURVEY 8d): saturated -- the history lives in registers, frac passes 1 on C3 / C5 (DESIGN.md section 5)"},
                    "whole_path": {"algorithmic_frac_gpu": 48.0 * cells / (ms_gpu * 1e-3) / 8e12 if ms_gpu else None, "gpu_share_of_align": ms_gpu * 1e-3 / dt,
                                   "note": "48 B x the cells of ALL align kernels / the time any of them was running on rank 0's device"},
                    "note": "launch durations are summed over workers whose launches overlap (an upper bound of the kernel's own time); the yardstick and the VALU floor are "
                            "the C3 line's (DESIGN.md section 5)"}
            if world == 1 and not args.no_cpu_baseline:
                # the CPU baseline on a bounded sample of the same records: oracle/wflign_host.py over oracle/wfa2p.c, one record per thread
                from oracle import wflign_host as W
                from concurrent.futures import ThreadPoolExecutor
                seqs = {}
                name = None
                for ln in open(fa):
                    if ln.startswith(">"):
                        name = ln[1:].split()[0]; seqs[name] = []
                    else:
                        seqs[name].append(ln.strip())
                seqs = {k: "".join(v).encode() for k, v in seqs.items()}
                mylines = [lines[i].rstrip("\n") for i in shard]
                n_s = args.cpu_sample or min(len(mylines), 256)
                sample = mylines[::max(1, len(mylines) // n_s)][:n_s]
                nt = max(1, min(len(sample), os.cpu_count() or 1, 64))
                t1 = time.perf_counter()
                with ThreadPoolExecutor(nt) as ex:
                    parts = list(ex.map(lambda i: W.align_mapping_lines(sample[i::nt], seqs, seqs), range(nt)))
                cdt = time.perf_counter() - t1
                want = [w for p in parts for w in p]
                got = set(l.rstrip("\n") for l in open(out_paf))
                same = sum(1 for w in want if w in got)
                qbp = sum(int(l.split("\t")[3]) - int(l.split("\t")[2]) for l in sample)
                cpu = {"value": qbp / cdt, "unit": "aligned bases/s", "cores": nt, "kind": "port", "host_cpu": _cpu_model(),
                       "sample": f"{len(sample)} of the {len(mylines)} mapping records (every k-th), oracle/wflign_host.py over oracle/wfa2p.c (BiWFA + patches + record), "
                                 f"{nt} threads, {cdt:.1f} s"}
                parity = {"sampled_records": len(sample), "oracle_records": len(want), "identical": same, "cigar_identical_rate": same / max(1, len(want))}
        tot = torch.tensor([float(bp), float(recs_n), dt], dtype=torch.float64, device=comm_dev)
        mx = tot.clone()
        if dist is not None:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        if rank == 0:
            dtm = float(mx[2].item())
            print(json.dumps({
                "metric": "aligned bases/sec (whole node) + CIGAR-identical rate vs CPU ref", "value": float(tot[0].item()) / dtm, "unit": "aligned bases/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dtm / args.steps * 1e3, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
                "config": {"workload": f"C4: 8 synthetic haplotypes x {args.c4_mbp} Mbp all-vs-all, -Y '#', defaults (ani50-2); per step: map (queries sharded over the "
                                       f"GPUs by length, index of all eight replicated) + exchange of the {len(lines)} mapping records + align (records sharded by "
                                       "(length x (1 - identity))^2) + gather of the PAF text to rank 0",
                           "queries_total": len(names), "queries_rank0": len(my_queries), "records_total": len(lines), "records_rank0": len(shard),
                           "parallelism": f"queries, then records, sharded over {world} GPU(s)", "host_threads_per_rank": threads},
                "roofline": roof, "cpu_baseline": cpu, "cigar_identical": parity,
                "map_s_rank0_per_step": state["t_map"] / args.steps, "exchange_s_rank0_per_step": state["t_xchg"] / args.steps,
                "map_stages_ms_rank0": ({"identity": state["ms"].ms_identity, "index": state["ms"].ms_index, "map": state["ms"].ms_map, "filter": state["ms"].ms_filter} if state["ms"] else None),
                "records_per_step_all_ranks": float(tot[1].item()) / args.steps, "ms_gpu_rank0_per_step": ms_gpu / args.steps}), flush=True)


def _shard(args, rank, world, pairs):
    """Indices of this rank's mapping records.  weak: rank r takes records [r * pairs, (r + 1) * pairs) of a file that grows
    with N; strong: the --total-pairs records of one file go to the ranks the way the reference's cluster sharding deals
    them out (dist.shard_records: longest first onto the least-loaded rank; weight = WFA cost ~ (length x divergence)^2,
    scripts/split_approx_mappings_in_chunks.py:19-27)."""
    if args.scaling == "weak":
        return list(range(rank * args.pairs, (rank + 1) * args.pairs))
    from wfmash_amd.dist import shard_records
    if pairs is None:  # --rank-check: the shard sizes only need the record count (synthetic records are equally long)
        weights = [1.0] * args.total_pairs
    else:
        weights = [float(len(q)) ** 2 for _, q in pairs]
    return shard_records(weights, world)[rank]


class _StatAcc:
    """wfm_stats_t summed over passes"""
    FIELDS = ("cells", "cells_bp", "cells_tile", "cells_tile_unique", "ms_tile", "ms_tile_busy", "ms_bp_busy", "ms_base_busy", "ms_any_busy",
              "tile_launches", "bp_launches", "base_launches", "ms_base")

    def __init__(self):
        for f in self.FIELDS:
            setattr(self, f, 0)
        self.ms_bp = 0.0
        self.streams = 1
        self.passes = 0

    def add(self, st):
        for f in self.FIELDS:
            setattr(self, f, getattr(self, f) + getattr(st, f))
        self.ms_bp += st.ms_breakpoint
        self.streams = max(self.streams, st.streams)
        self.passes += 1

    @property
    def cells_unique(self):  # all kernels, the tile kernel's re-run block counted once
        return self.cells - (self.cells_tile - self.cells_tile_unique)


def _roofline(acc, excl, seq_bytes, args):
    """Dominant kernel = the time-tiled phase-1 kernel (wfa_tile_reg_kernel) when it ran, else the step kernel.
    achieved = ALGORITHMIC bytes / the kernel's running time: 48 B per (score, diagonal) cell the result needs (7 loads +
    5 stores of int32 offsets; the block a job runs twice is counted once) + the sequences once per pass."""
    peak = 8000.0
    tiled = acc.ms_tile > 0.5 * acc.ms_bp
    if tiled:
        dom, cells, cells_all, busy, launches, ms_sum = "wfa_tile2_kernel", acc.cells_tile_unique, acc.cells_tile, acc.ms_tile_busy, acc.tile_launches, acc.ms_tile
        e_cells, e_ms, e_launches = excl.cells_tile_unique, excl.ms_tile, excl.tile_launches
    else:
        dom, cells, cells_all, busy, launches, ms_sum = "wfa_bp_kernel", acc.cells_bp - acc.cells_tile, acc.cells_bp - acc.cells_tile, acc.ms_bp_busy, acc.bp_launches, acc.ms_bp - acc.ms_tile
        e_cells, e_ms, e_launches = excl.cells_bp - excl.cells_tile, excl.ms_bp - excl.ms_tile, excl.bp_launches
    alg = 48.0 * cells + seq_bytes * acc.passes
    achieved = alg / (busy * 1e-3) / 1e9 if busy > 0 else 0.0
    e_alg = 48.0 * e_cells + seq_bytes * excl.passes
    e_achieved = e_alg / (e_ms * 1e-3) / 1e9 if e_ms > 0 else 0.0
    traffic = hbm_frac = None  # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/)
    src = None
    for name in ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if tj.get("kernel") == dom and args.config == "C3" and args.pairs == 64:
            traffic = tj["traffic_bytes_per_launch"]
            if tj.get("avg_launch_ms"):
                hbm_frac = traffic / (tj["avg_launch_ms"] * 1e-3) / 1e9 / peak
            src = "profiles/" + name
            break
    issue = None  # issue-side figures of the same kernel from the committed SQ passes (profiles/): the 48 B/cell yardstick is
    for name in ("r6_sq.json", "r5_sq.json", "r4_sq.json"):  # saturated (C5 passes 1.0), what the kernel is really short of is issue slots and latency
        try:
            issue = json.load(open(os.path.join(ROOT, "profiles", name)))
            issue["source"] = "profiles/" + name
            break
        except (OSError, ValueError):
            continue
    # The bound that can still be missed: vector-instruction issue.  A cell of the tile kernel cannot cost fewer than
    # VALU_FLOOR vector instructions (recurrences 9, score-bound select 2, one 16-base probe of both sequences 17, extension add 1,
    # antidiagonal 2, its share of the neighbour exchange 4, of the per-step wave maximum 1): peak = lane-instructions per second / floor.
    peak_cells = VALU_PEAK_CELLS
    e_cells_per_s = e_cells / (e_ms * 1e-3) if e_ms > 0 else 0.0
    steps_per_wave = 100.0
    valu = {"bound": "valu", "unit": "cells/s", "floor_valu_insts_per_cell": VALU_FLOOR, "peak": peak_cells, "achieved": e_cells_per_s,
            "frac": e_cells_per_s / peak_cells,
            "note": "achieved = unique (score, diagonal) cells of the tile kernel / its exclusive running time, measured live (the same events as roofline.frac); "
                    "peak = 1024 SIMDs x 2.4 GHz x 64 lanes / (the instruction floor of a cell x the mean issue cycles of the step's instruction mix, "
                    "profiles/r6_valu_issue.md: 4.2 cycles per wave64 instruction for max / compare / select / alignbit / ffbl / DPP, 2.2 for add / sub / logic; "
                    "scripts/isa_hot_path.py prices the step's common path at 3.4). Lanes of a tile's halo and lanes without a cell issue as well and count against it.",
            "mean_issue_cycles_per_valu_inst": VALU_MEAN_ISSUE_CYCLES}
    if issue and issue.get("counters", {}).get("SQ_WAVES"):
        c = issue["counters"]
        valu["committed_profile"] = {"source": issue.get("source"), "valu_insts_per_wave_step": c.get("SQ_INSTS_VALU", 0) / c["SQ_WAVES"] / steps_per_wave,
                                     "salu_insts_per_wave_step": c.get("SQ_INSTS_SALU", 0) / c["SQ_WAVES"] / steps_per_wave,
                                     "floor_per_wave_step": VALU_FLOOR * 2, "valu_busy": issue.get("valu_frac"), "wait_frac": issue.get("wait_frac"),
                                     "frac_valu_from_counters": (VALU_FLOOR * 2) / max(1.0, c.get("SQ_INSTS_VALU", 0) / c["SQ_WAVES"] / steps_per_wave) * (issue.get("valu_frac") or 0.0),
                                     "note": "per wave and score step of 2 x 64 cells, from the committed SQ pass (the builder's box, not this run): floor / measured x VALU busy"}
    hbm = {"bound": "hbm", "achieved": e_achieved, "peak": peak, "unit": "GB/s", "frac": e_achieved / peak,
           "achieved_overlapped": achieved, "frac_overlapped": achieved / peak,
           "note": "the contract's yardstick, 48 B per cell (SURVEY 8d) -- saturated: the wavefront history never leaves the registers, real traffic is a fifth of it "
                   "(traffic), frac passes 1; exclusive = untimed WFM_OVERLAP=0 passes (what rocprofv3 --kernel-trace reproduces), overlapped = the timed region"}
    return {"bound": "valu", "achieved": valu["achieved"], "peak": valu["peak"], "unit": "cells/s", "frac": valu["frac"], "traffic": traffic,
            "valu": valu, "hbm_yardstick": hbm,
            "kernel": dom + (" (wfa_tile_reg_kernel for problems with an N or soft-masked bases: none in this workload)" if tiled else ""),
            # the figure rocprofv3 reproduces: launches one after the other on one stream (untimed passes with WFM_OVERLAP=0)
            "committed_profile": {"traffic": traffic, "hbm_frac": hbm_frac, "traffic_source": src,
                                  "valu_frac": issue.get("valu_frac") if issue else None, "wait_frac": issue.get("wait_frac") if issue else None, "issue_source": issue.get("source") if issue else None,
                                  "note": "read from the committed PMC passes under profiles/ (the builder's box, scripts/profile_r5.sh), NOT measured in this run; "
                                          "everything outside this key is measured live with HIP events"},
            "frac_exclusive": e_achieved / peak, "achieved_exclusive": e_achieved,
            "avg_launch_ms_exclusive": e_ms / max(e_launches, 1), "launches_exclusive_per_step": e_launches / max(excl.passes, 1),
            # (a "launch" here is one BLOCK of 100 scores: the events stand around the one or two instantiations of the tile kernel a block launches -- with
            # and without per-score maxima.  rocprofv3 lists the instantiations apart: their total times summed / the passes of the profiled process is this figure)
            "tile_kernel_ms_exclusive_per_pass": e_ms / max(excl.passes, 1),
            "algorithmic_bytes_per_launch_exclusive": e_alg / max(e_launches, 1),
            "algorithmic_bytes_per_launch": alg / max(launches, 1), "cells_per_launch": cells / max(launches, 1),
            "cells_computed_per_launch": cells_all / max(launches, 1),
            "avg_launch_ms": ms_sum / max(launches, 1), "launches": launches, "streams": acc.streams,
            "kernel_busy_ms_per_step": busy / max(acc.passes, 1),
            "note": "frac = unique (score, diagonal) cells of the dominant kernel / its exclusive running time (untimed WFM_OVERLAP=0 passes: launches one after "
                    "the other on one stream, what rocprofv3 --kernel-trace reproduces) against the vector-issue peak of the cell's instruction floor; hbm_yardstick = "
                    "the contract's 48 B per cell against 8 TB/s, which a kernel whose history lives in registers passes. DESIGN.md section 5"}


def _cpu_baseline_map(h):
    """Map-phase CPU baseline: the REFERENCE'S OWN CommonFunc::addMinmers (oracle/_ref/libref_map.so, compiled from
    /root/reference in the build container) on the host cores, one sequence per thread as Sketch::build runs it
    (winSketch.hpp:175-260), against the product's index build of the same sequences on the GPU."""
    import threading
    try:
        from oracle import pymap
        if not pymap.have_ref():
            return {"value": None, "kind": "reference", "note": "oracle/_ref/libref_map.so is not built in this checkout"}
        import numpy as np
        from wfmash_amd import synth
        cores = os.cpu_count() or 1
        n_seq, L = min(cores, 64), 2_000_000
        base = synth.random_backbone(0xB45E, L)
        seqs = [synth.haplotype(base, 0xB45E00 + i, n_sv=2).tobytes() for i in range(n_seq)]
        k, w, s_sz = 15, 1000, 39
        pymap.ref_park_stderr(True)  # (the reference's ProgressMeter threads: parked once around the leg, not per call)
        t1 = time.perf_counter()
        ths = [threading.Thread(target=pymap.ref_add_minmers, args=(sq, k, w, s_sz, i)) for i, sq in enumerate(seqs)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        cdt = time.perf_counter() - t1
        pymap.ref_park_stderr(False)
        t1 = time.perf_counter()
        h.add_minmers_multi(seqs, k, w, s_sz, threads=cores)
        gdt = time.perf_counter() - t1
        bp = sum(len(x) for x in seqs)
        return {"value": bp / cdt, "unit": "indexed bases/s", "cores": n_seq, "kind": "reference",
                "sample": f"{n_seq} synthetic haplotypes x {L // 1000000} Mbp, k={k} w={w} s={s_sz}: CommonFunc::addMinmers "
                          f"(commonFunc.hpp:440-708) one sequence per thread, {cdt:.1f} s",
                "gpu_value": bp / gdt, "gpu_note": f"wfm_add_minmers_multi of the same sequences (hashing and thinning on the device; sequences of 2 Mbp are winnowed by {cores} host threads side by side, from 4 Mbp on by the device), {gdt:.2f} s"}
    except Exception as e:  # the baseline is a report, never a reason to lose the bench line
        return {"value": None, "kind": "reference", "note": f"failed: {e}"}


def _align_fields(al, t_al):
    """what a wfmh_align_paf run says about itself: throughput, device share, host stages (summed over the batches)"""
    return {"align_s": t_al, "records": int(al.records), "aligned_bp": int(al.aligned_bp), "aligned_bp_per_s": al.aligned_bp / t_al,
            "cells": int(al.cells), "ms_gpu": al.ms_gpu, "gpu_share_of_align": al.ms_gpu * 1e-3 / t_al,
            "algorithmic_frac_gpu": 48.0 * al.cells / (al.ms_gpu * 1e-3) / 8e12 if al.ms_gpu else None,
            "batches": int(al.batches), "host_ms_summed_over_batches": {"rows": al.ms_rows, "fetch": al.ms_fetch, "wflign_incl_device_calls": al.ms_wflign, "text": al.ms_text,
                                                                      # (the records' tags the parity sample is drawn from are written inside the timed call: this is what it cost)
                                                                      "record_tags": al.ms_tags}}


def _sampled_cigar_identity(fa_seqs, map_lines, aln_path, n_sample, tags_path=None, **oracle_kw):
    """CIGAR-identical rate of a sample of a run's records against the align oracle (oracle/wflign_host.py over oracle/wfa2p.c):
    the sampled mapping lines are aligned by the oracle; each record must be, byte for byte, a line of the run's output.
    The sample is STRATIFIED where the run wrote its records' tags (WFM_RECORD_TAGS, host/aligner.cpp -> wfm_get_problem_flags): records whose
    root or a child ran again, whose patches went to a second / third score budget or to the ring kernel, which ran on the byte kernels, whose
    overlap walk took several rounds, the highest scores -- the paths that have broken before -- and then every k-th row."""
    from oracle import wflign_host as W
    strata = None
    rows = None
    if tags_path and os.path.exists(tags_path):
        from wfmash_amd import capi as _capi
        tags = _capi.read_record_tags(tags_path)
        if tags:
            rows, strata = _capi.stratified_rows(tags, len(map_lines), per_stratum=max(4, n_sample // 8), top_scores=max(4, n_sample // 8), uniform=n_sample)
    if rows is None:
        rows = list(range(0, len(map_lines), max(1, len(map_lines) // n_sample)))[:n_sample]
    sample = [map_lines[r] for r in rows]
    from concurrent.futures import ThreadPoolExecutor
    nt = max(1, min(len(sample), (os.cpu_count() or 1), 32))  # (the oracle is C behind ctypes: the calls run side by side)
    with ThreadPoolExecutor(nt) as ex:
        parts = list(ex.map(lambda i: W.align_mapping_lines(sample[i::nt], fa_seqs, fa_seqs, **oracle_kw), range(nt)))
    want = [w for p in parts for w in p]
    got = set(l.rstrip("\n") for l in open(aln_path))
    same = sum(1 for w in want if w in got)
    return {"sampled_records": len(sample), "oracle_records": len(want), "identical": same, "cigar_identical_rate": same / max(1, len(want)),
            "strata": strata if strata is not None else "uniform (no record tags)"}


def _mapping_identity(h, capi, synth, td):
    """mapping-coordinate identity (PAF columns 1-9 + ch:Z:) of the map path on a small pangenome against the
    stage oracles + the reference's own filter code; None where oracle/_ref did not travel"""
    from oracle import map_ani as ANI
    from oracle import map_pipeline as MP
    from oracle import pyfilter, pymap
    import numpy as np
    if not (pymap.have_ref() and pyfilter.have_ref()):
        return {"mapping_identical_rate": None, "note": "oracle/_ref is not built in this checkout"}
    recs = [(n, s.tobytes()) for n, s in synth.pangenome(8, 600_000, n_sv=2, sv_min=3_000, sv_max=20_000)]
    fa = os.path.join(td, "mi.fa")
    names, _ = synth.write_fasta(fa, recs)
    m = os.path.join(td, "mi.paf")
    capi.map_paf(h, fa, m, params=capi.map_default_params(threads=os.cpu_count() or 1))
    got = [l for l in open(m).read().splitlines()]
    q = 3
    pct = np.float32(ANI.estimate_identity([s for _, s in recs], MP.ref_groups(names), 50, -2.0))
    S = MP.sketch_size(pct, 1000, 15)
    maps, _, _ = MP.map_queries(recs, pct, queries={q})
    exp = pyfilter.ref_filter("subset", maps[q], fa, names[q], capi.map_default_params(percentage_identity=float(pct), auto_pct_identity=0, sketch_size=S)).splitlines()

    def key(l):
        f = l.split("\t")
        return tuple(f[:9]) + tuple(x for x in f[12:] if x.startswith("ch:Z:"))
    mine = [key(l) for l in got if l.split("\t", 1)[0] == names[q]]
    want = [key(l) for l in exp]
    same = len(set(mine) & set(want))
    return {"mapping_identical_rate": same / max(1, len(want)), "records_oracle": len(want), "records_gpu": len(mine), "byte_identical": [l for l in got if l.split("\t", 1)[0] == names[q]] == exp,
            "workload": "8 synthetic haplotypes x 0.6 Mbp, defaults: one query haplotype against the stage oracles + the reference's filter code"}


def _secondary(h, capi, synth, full_c4=True, only=None):
    """Driver-timed figures of the other configs (the bench line's `value` stays C3): C5 align-only, C4 ranks at two sizes, the C1
    substitute and C2 (LPA.subset all-vs-all) end to end through the C ABI, each with a parity check against the oracles."""
    import tempfile
    sec = {}
    threads = os.cpu_count() or 1
    want = lambda tag: not only or tag in only
    try:
        if not want("C5"):
            raise KeyError("skipped")
        pairs = synth.pairs("C5", n_pairs=8)
        ss = h.upload(pairs)
        h.align_resident(ss, collect=False)
        acc = _StatAcc()
        t1 = time.perf_counter()
        for _ in range(2):
            failed = h.align_resident(ss, collect=False)
            acc.add(h.stats())
        d = (time.perf_counter() - t1) / 2
        res = h._collect(ss)
        sec["C5"] = {"workload": "8 synthetic 15%-divergence 100kb pairs, WFA-only", "ms_per_pass": d * 1e3,
                     "aligned_bp_per_s": sum(len(q) for _, q in pairs) / d, "failed": int(failed), "score_mean": sum(r.score for r in res) / len(res),
                     "cells_per_pass": acc.cells_unique / 2, "algorithmic_frac_wall": 48.0 * acc.cells_unique / 2 / d / 8e12,
                     "kernel_busy_ms": {"tile": acc.ms_tile_busy / 2, "bp": acc.ms_bp_busy / 2, "base": acc.ms_base_busy / 2}}
        ss.free()
    except Exception as e:
        sec["C5"] = {"error": str(e)}
    with tempfile.TemporaryDirectory() as td:
        # (C1 first: its divergent batches size the align handles' arenas, as a run of its own would; after the C4 legs it would pay for
        # growing them step by step -- 7.3 s instead of 5.7)
        try:  # C1: the reference's CPU-runnable case; data/scerevisiae8.fa.gz is a missing blob, synth.yeast_like stands in
            if not want("C1_substitute"):
                raise KeyError("skipped")
            fa = os.path.join(td, "c1.fa")
            recs = [(n, s) for n, s in synth.yeast_like(8, 16, 12_000_000)]
            names, lengths = synth.write_fasta(fa, recs)
            m, a = os.path.join(td, "c1.m.paf"), os.path.join(td, "c1.a.paf")
            t1 = time.perf_counter()
            ms = capi.map_paf(h, fa, m, params=capi.map_default_params(threads=threads))
            t_map = time.perf_counter() - t1
            t1 = time.perf_counter()
            tg = os.path.join(td, "c1.tags")
            os.environ["WFM_RECORD_TAGS"] = tg  # (one line per record, written per batch: the parity sample is drawn from it)
            al = capi.align_paf(h, fa, m, a, params={"threads": threads})
            t_al = time.perf_counter() - t1
            os.environ.pop("WFM_RECORD_TAGS", None)
            leg = {"workload": f"C1 substitute: 8 yeast-like strains x 16 chromosomes ({sum(lengths) / 1e6:.0f} Mbp), all-vs-all, defaults (ani50-2), map + align",
                   "map_s": t_map, "ms_identity": ms.ms_identity, "ms_index": ms.ms_index, "ms_map": ms.ms_map, "ms_filter": ms.ms_filter,
                   "mapping_records": int(ms.written), "identity_threshold": float(ms.percentage_identity)}
            leg.update(_align_fields(al, t_al))
            leg["aligned_bp_per_s_end_to_end"] = al.aligned_bp / (t_map + t_al)
            seqs = {n: s.tobytes() for n, s in recs}
            leg["parity"] = _sampled_cigar_identity(seqs, open(m).read().splitlines(), a, 32, tags_path=tg)
            sec["C1_substitute"] = leg
            del recs, seqs
        except Exception as e:
            sec["C1_substitute"] = {"error": str(e)}
        legs = [("C4_rank_scaled", 8, 48), ("C4_rank_40mbp", 40, 64)]
        if full_c4:
            legs.append(("C4_rank_full", 248.956422, 64))  # north_star's own size: one chr1-sized haplotype against all eight
        for tag, mbp, n_cig in legs:
            try:  # one rank of C4: 8 haplotypes, one of them (1/8 of the queries) against the index of all eight
                if not want(tag):
                    raise KeyError("skipped")
                a = os.path.join(td, "a.paf")
                fa = os.path.join(td, f"c4_{mbp}.fa")
                t_g = time.perf_counter()
                recs = synth.pangenome_parallel(8, int(mbp * 1_000_000), n_sv=6 if mbp == 8 else 20, workers=min(8, threads))
                names, lengths = synth.write_fasta(fa, recs)
                t_gen = time.perf_counter() - t_g
                ql = os.path.join(td, "q.txt")
                open(ql, "w").write(names[0] + "\n")
                m = os.path.join(td, "m.paf")
                t1 = time.perf_counter()
                ms = capi.map_paf(h, fa, m, params=capi.map_default_params(threads=threads, query_list=ql))
                t_map = time.perf_counter() - t1
                tg = os.path.join(td, f"{tag}.tags")
                if os.path.exists(tg):
                    os.unlink(tg)
                os.environ["WFM_RECORD_TAGS"] = tg
                t1 = time.perf_counter()
                al = capi.align_paf(h, fa, m, a, params={"threads": threads})
                t_al = time.perf_counter() - t1
                os.environ.pop("WFM_RECORD_TAGS", None)
                leg = {"workload": f"8 synthetic haplotypes x {mbp} Mbp, -Y '#', defaults (ani50-2): rank 0 of 8 (one haplotype against all)",
                       "generate_s": t_gen, "map_s": t_map, "ms_identity": ms.ms_identity, "ms_index": ms.ms_index, "ms_map": ms.ms_map, "ms_filter": ms.ms_filter, "mapping_records": int(ms.written)}
                leg.update(_align_fields(al, t_al))
                leg["aligned_bp_per_s_map_and_align"] = al.aligned_bp / (t_map + t_al)
                if True:  # the same align phase once more: arenas sized, handles of the workers created (the first pass is what a one-shot run pays)
                    t1 = time.perf_counter()
                    al2 = capi.align_paf(h, fa, m, a, params={"threads": threads})
                    t2 = time.perf_counter() - t1
                    leg["second_pass"] = {"align_s": t2, "aligned_bp_per_s": al2.aligned_bp / t2, "ms_gpu": al2.ms_gpu,
                                          "algorithmic_frac_gpu": 48.0 * al2.cells / (al2.ms_gpu * 1e-3) / 8e12 if al2.ms_gpu else None}
                    if mbp >= 100:  # and the map phase: what the device heap had to grow by for the align phase is there now
                        t1 = time.perf_counter()
                        ms2 = capi.map_paf(h, fa, m, params=capi.map_default_params(threads=threads, query_list=ql))
                        leg["second_pass"].update({"map_s": time.perf_counter() - t1, "ms_identity": ms2.ms_identity, "ms_index": ms2.ms_index, "ms_map": ms2.ms_map, "ms_filter": ms2.ms_filter})
                seqs = {n: s.tobytes() for n, s in recs}
                leg["parity"] = _sampled_cigar_identity(seqs, open(m).read().splitlines(), a, n_cig, tags_path=tg)
                sec[tag] = leg
                if tag == "C4_rank_full" and want("C4_all_vs_all"):
                    # north_star's C4 as ONE job on one GPU -- the N = 1 point of its 1 / 2 / 4 / 8 curve (`--config C4` is the same job as a bench line
                    # of its own, with the queries and then the records sharded over the ranks): all eight haplotypes against all eight, map + align
                    try:
                        m8, a8, tg8 = os.path.join(td, "m8.paf"), os.path.join(td, "a8.paf"), os.path.join(td, "c4_all.tags")
                        t1 = time.perf_counter()
                        ms8 = capi.map_paf(h, fa, m8, params=capi.map_default_params(threads=threads))
                        t_map8 = time.perf_counter() - t1
                        os.environ["WFM_RECORD_TAGS"] = tg8
                        t1 = time.perf_counter()
                        al8 = capi.align_paf(h, fa, m8, a8, params={"threads": threads})
                        t_al8 = time.perf_counter() - t1
                        os.environ.pop("WFM_RECORD_TAGS", None)
                        leg8 = {"workload": f"C4 at north_star's size on ONE GPU: 8 synthetic haplotypes x {mbp} Mbp ALL-VS-ALL, -Y '#', defaults (ani50-2), map + align",
                                "queries": len(names), "map_s": t_map8, "ms_identity": ms8.ms_identity, "ms_index": ms8.ms_index, "ms_map": ms8.ms_map, "ms_filter": ms8.ms_filter,
                                "mapping_records": int(ms8.written)}
                        leg8.update(_align_fields(al8, t_al8))
                        leg8["wall_s"] = t_map8 + t_al8
                        leg8["aligned_bp_per_s_map_and_align"] = al8.aligned_bp / (t_map8 + t_al8)
                        leg8["parity"] = _sampled_cigar_identity(seqs, open(m8).read().splitlines(), a8, n_cig, tags_path=tg8)
                        sec["C4_all_vs_all"] = leg8
                    except Exception as e:
                        sec["C4_all_vs_all"] = {"error": str(e)}
                del recs, seqs
                os.unlink(fa)
            except Exception as e:
                sec[tag] = {"error": str(e)}
        try:  # C2: the reference's LPA test data (a committed fixture), all-vs-all -p 90 -P 50k
            if not want("C2"):
                raise KeyError("skipped")
            import gzip
            lpa = os.path.join(ROOT, "tests", "golden", "LPA.subset.fa.gz")
            m, a = os.path.join(td, "lpa.m.paf"), os.path.join(td, "lpa.a.paf")
            t1 = time.perf_counter()
            ms = capi.map_paf(h, lpa, m, params=capi.map_default_params(percentage_identity=0.9, auto_pct_identity=0, max_mapping_length=50000, threads=threads))
            t_map = time.perf_counter() - t1
            t1 = time.perf_counter()
            tg = os.path.join(td, "lpa.tags")
            os.environ["WFM_RECORD_TAGS"] = tg
            al = capi.align_paf(h, lpa, m, a, params={"threads": threads})
            t_al = time.perf_counter() - t1
            os.environ.pop("WFM_RECORD_TAGS", None)
            leg = {"workload": "LPA.subset.fa.gz all-vs-all, -p 90 -P 50k, map + align", "map_s": t_map, "mapping_records": int(ms.written)}
            leg.update(_align_fields(al, t_al))
            leg["aligned_bp_per_s_align"] = al.aligned_bp / t_al
            leg["aligned_bp_per_s_end_to_end"] = al.aligned_bp / (t_map + t_al)
            seqs, name = {}, None
            for line in gzip.open(lpa, "rt"):
                if line.startswith(">"):
                    name = line[1:].split()[0]
                    seqs[name] = []
                else:
                    seqs[name].append(line.strip())
            seqs = {k: "".join(v).encode() for k, v in seqs.items()}
            leg["parity"] = _sampled_cigar_identity(seqs, open(m).read().splitlines(), a, 16, tags_path=tg)
            sec["C2"] = leg
        except Exception as e:
            sec["C2"] = {"error": str(e)}
        try:
            if not want("map_parity"):
                raise KeyError("skipped")
            sec["map_parity"] = _mapping_identity(h, capi, synth, td)
        except Exception as e:
            sec["map_parity"] = {"error": str(e)}
    return {k: v for k, v in sec.items() if v.get("error") != "'skipped'"}


def _legs_summary(out):
    """Every leg in a few numbers, at the end of the JSON line: [align seconds, M aligned bp/s of the align phase, device-busy ms,
    48 B x cells / device-busy time / 8 TB/s, sampled CIGAR-identical rate, map seconds]; second_pass: [align seconds, device-busy ms, that fraction(, map seconds)]."""
    def r(x, n=3):
        return None if x is None else round(float(x), n)
    legs = {"C3": {"ms_per_step": r(out["ms_per_step"], 2), "Mbp_per_s": r(out["value"] / 1e6, 2), "tile_valu_frac": r(out["roofline"]["frac"]),
                   "tile_hbm_yardstick_frac": r(out["roofline"].get("hbm_yardstick", {}).get("frac")), "whole_step_frac": r(out["whole_step"]["frac"]),
                   "cigar_identical": out.get("cigar_identical_rate")}}
    for tag, leg in out.get("secondary", {}).items():
        if "error" in leg:
            legs[tag] = {"error": leg["error"][:80]}
        elif tag == "C5":
            legs[tag] = {"ms_per_pass": r(leg["ms_per_pass"], 1), "Mbp_per_s": r(leg["aligned_bp_per_s"] / 1e6, 2), "frac_wall": r(leg["algorithmic_frac_wall"])}
        elif tag == "map_parity":
            legs[tag] = {"mapping_identical": leg.get("mapping_identical_rate"), "byte_identical": leg.get("byte_identical")}
        else:
            legs[tag] = {"align_s": r(leg["align_s"]), "Mbp_per_s": r(leg["aligned_bp"] / leg["align_s"] / 1e6, 1), "ms_gpu": r(leg["ms_gpu"], 1),
                         "frac_gpu": r(leg.get("algorithmic_frac_gpu")), "cigar_identical": leg.get("parity", {}).get("cigar_identical_rate"),
                         "records": leg.get("records"), "map_s": r(leg.get("map_s"))}
            if "second_pass" in leg:
                legs[tag]["second_pass"] = [r(leg["second_pass"]["align_s"]), r(leg["second_pass"]["ms_gpu"], 1), r(leg["second_pass"]["algorithmic_frac_gpu"])]
                if "map_s" in leg["second_pass"]:
                    legs[tag]["second_pass"].append(r(leg["second_pass"]["map_s"]))
    return legs


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
