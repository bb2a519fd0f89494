"""One rank's share of config C4 (8 haplotypes of a chr1-sized backbone, `-Y '#'`, 8 GPUs) on ONE GPU:
every rank builds the index of all haplotypes and maps the queries dist.shard_queries gives it, so
rank r of 8 is a complete dry run of what each GPU of the node does (no exchange step
besides the gather of PAF text).

The haplotypes are synthetic (real chr1 is not available): one random backbone, per haplotype 0.1 % SNPs,
0.01 % short indels and 20 structural variants of 10-100 kb (deletion / duplication / inversion).  The
generator is vectorised numpy (seeded), not the splitmix64 one of C3: 2 Gbp have to be made in seconds.

Usage: python scripts/c4_rank.py [--haps 8] [--mbp 248.956422] [--rank 0] [--world 8] [--pct 0 (auto)]
Prints one JSON line: sizes, stage times, records written, peak host RSS."""
import argparse
import json
import os
import resource
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, dist, synth  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--haps", type=int, default=8)
    ap.add_argument("--mbp", type=float, default=248.956422)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--pct", type=float, default=0.0, help="identity threshold as a fraction; 0 = estimate (ani50-2)")
    ap.add_argument("--keep", default="")
    ap.add_argument("--align", action="store_true", help="also run the align phase on the rank's mappings")
    a = ap.parse_args()

    d = a.keep or tempfile.mkdtemp()
    fa = os.path.join(d, "c4.fa")
    t0 = time.time()
    names, lengths = synth.write_fasta(fa, synth.pangenome(a.haps, int(a.mbp * 1e6)))
    t_gen = time.time() - t0

    mine = [names[i] for i in dist.shard_queries(lengths, a.world)[a.rank]]
    qlist = os.path.join(d, f"queries.rank{a.rank}.txt")
    with open(qlist, "w") as f:
        f.write("\n".join(mine) + "\n")
    threads = os.cpu_count() or 1
    over = dict(threads=threads, query_list=qlist)
    if a.pct:
        over.update(percentage_identity=a.pct, auto_pct_identity=0)
    P = capi.map_default_params(**over)
    h = capi.Handle(0)
    out = os.path.join(d, f"rank{a.rank}.paf")
    t0 = time.time()
    s = capi.map_paf(h, fa, out, params=P)
    wall = time.time() - t0
    align = None
    if a.align:
        aln = os.path.join(d, f"rank{a.rank}.aln.paf")
        t0 = time.time()
        sa = capi.align_paf(h, fa, out, aln)
        wall_a = time.time() - t0
        # every record's CIGAR must span exactly its coordinates (pafcheck's rule)
        import re
        bad_cg = n_rec = 0
        with open(aln) as f:
            for line in f:
                c = line.rstrip("\n").split("\t")
                cg = next(x[5:] for x in c[12:] if x.startswith("cg:Z:"))
                ql = tl = 0
                for num, op in re.findall(r"(\d+)([=XIDM])", cg):
                    ql += int(num) if op in "=XIM" else 0
                    tl += int(num) if op in "=XDM" else 0
                n_rec += 1
                bad_cg += (ql != int(c[3]) - int(c[2])) or (tl != int(c[8]) - int(c[7]))
        align = {"records": int(sa.records), "written": int(sa.written), "aligned_bp": int(sa.aligned_bp), "cells": int(sa.cells),
                 "ms_gpu": round(sa.ms_gpu), "ms_total": round(sa.ms_total), "wall_s": round(wall_a, 2),
                 "aligned_bp_per_s": round(sa.aligned_bp / (sa.ms_total / 1e3)), "cigar_span_errors": int(bad_cg), "checked": n_rec}
    h.close()
    # sanity of the output: every record starts inside its sequences, query ranges are exact and no query maps to
    # its own haplotype.  A target END may pass the sequence end by a few bases: a chain's block length is the
    # larger of its two spans (base_types.hpp:215-229) and is printed as such (mappingOutput.hpp:74-138); the
    # align driver clamps it.  Counted separately.
    bad = []
    past_end = 0
    span = 0
    ln = dict(zip(names, lengths))
    with open(out) as f:
        for line in f:
            c = line.split("\t")
            qs, qe, ts, te = int(c[2]), int(c[3]), int(c[7]), int(c[8])
            span += qe - qs
            past_end += te > ln[c[5]]
            if not (0 <= qs < qe <= ln[c[0]] and 0 <= ts < te and ts < ln[c[5]] and te <= ln[c[5]] + 1000) or c[0].split("#")[0] == c[5].split("#")[0]:
                bad.append(line.rstrip("\n")[:300])
    print(json.dumps({"config": f"C4 rank {a.rank} of {a.world}", "haps": a.haps, "target_bp": int(s.target_bp), "query_bp": int(s.query_bp),
                      "queries": mine, "threads": threads, "gen_s": round(t_gen, 1), "wall_s": round(wall, 2),
                      "pct": round(float(s.percentage_identity), 4), "sketch": s.sketch_size, "windows": int(s.index_windows),
                      "fragments": int(s.fragments), "l2_mappings": int(s.l2_mappings), "written": int(s.written), "bad_records": len(bad), "bad_examples": bad[:3], "target_end_past_length": int(past_end),
                      "query_span_mapped_per_target": round(span / max(1, int(s.query_bp)) / max(1, a.haps - 1), 4),
                      "ms_identity": round(s.ms_identity), "ms_wall": round(s.ms_wall), "ms_index": round(s.ms_index), "ms_map": round(s.ms_map), "ms_filter": round(s.ms_filter), "ms_total": round(s.ms_total),
                      "query_mbp_per_s": round(s.query_bp / 1e6 / (s.ms_total / 1e3), 2),
                      "align": align,
                      "peak_rss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)}))


if __name__ == "__main__":
    main()
