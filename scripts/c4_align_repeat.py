"""The align phase of one rank of C4 (scripts/c4_rank.py's haplotypes and mappings) several times in one process, so that the
second and later passes run on warm arenas; with WFM_DEBUG=1 the library's stage times and the workers' batch timelines go to
stderr.  Usage: python scripts/c4_align_repeat.py [--mbp 248.956422] [--reps 3] [--threads N]
Prints one JSON line per pass: align_s, ms_gpu, gpu_share, frac_gpu."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, dist, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--haps", type=int, default=8)
    ap.add_argument("--mbp", type=float, default=248.956422)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    d = tempfile.mkdtemp()
    fa = os.path.join(d, "c4.fa")
    names, lengths = synth.write_fasta(fa, synth.pangenome(a.haps, int(a.mbp * 1e6)))
    mine = [names[i] for i in dist.shard_queries(lengths, 8)[0]]
    ql = os.path.join(d, "q.txt")
    open(ql, "w").write("\n".join(mine) + "\n")
    h = capi.Handle(0)
    import hashlib
    m = os.path.join(d, "m.paf")
    for rep in range(a.reps):
        print(f"==== pass {rep}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        ms = capi.map_paf(h, fa, m, params=capi.map_default_params(threads=a.threads, query_list=ql))  # (every pass: the map phase at this thread count too)
        map_s = time.perf_counter() - t0
        out = os.path.join(d, f"a{rep}.paf")
        t0 = time.perf_counter()
        al = capi.align_paf(h, fa, m, out, params={"threads": a.threads})
        dt = time.perf_counter() - t0
        print(json.dumps({"pass": rep, "mbp": a.mbp, "threads": a.threads, "map_s": round(map_s, 3), "ms_identity": ms.ms_identity, "ms_index": ms.ms_index, "ms_map": ms.ms_map, "ms_filter": ms.ms_filter,
                          "align_s": round(dt, 4), "records": int(al.records), "aligned_bp": int(al.aligned_bp),
                          "Mbp_per_s": round(al.aligned_bp / dt / 1e6, 1), "Mbp_per_s_map_and_align": round(al.aligned_bp / (dt + map_s) / 1e6, 1), "ms_gpu": round(al.ms_gpu, 1), "gpu_share": round(al.ms_gpu * 1e-3 / dt, 3),
                          "frac_gpu": round(48.0 * al.cells / (al.ms_gpu * 1e-3) / 8e12, 3) if al.ms_gpu else None,
                          "host_ms_summed": {"rows": round(al.ms_rows, 1), "fetch": round(al.ms_fetch, 1), "wflign": round(al.ms_wflign, 1), "text": round(al.ms_text, 1)}, "batches": int(al.batches),
                          "md5_mapping": hashlib.md5(open(m, "rb").read()).hexdigest(), "md5_aligned": hashlib.md5(open(out, "rb").read()).hexdigest()}), flush=True)
    if a.reps > 1:
        same = all(open(os.path.join(d, f"a{r}.paf"), "rb").read() == open(os.path.join(d, "a0.paf"), "rb").read() for r in range(1, a.reps))
        print(json.dumps({"passes_byte_identical": same}), flush=True)
    h.close()


if __name__ == "__main__":
    main()
