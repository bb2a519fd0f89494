"""Node-level run of the C4 shape through the product's own multi-GPU path (wfmh_map_multi + wfmh_align_paf_multi: what
`wfmash-hip --gpus N` calls): N device handles, the target index built once and copied to the others, query batches and
record batches going to whichever device is free.

On a node with fewer than N GPUs the handles share the devices round-robin (--share): the numbers then show what does NOT
grow with N (index cost per node) and that the bytes do not depend on N; they are not a scaling curve.

    python scripts/c4_node.py [--haps 8] [--mbp 40] [--gpus 1,2,4,8] [--share] [--align]
Prints one JSON line per N."""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--haps", type=int, default=8)
    ap.add_argument("--mbp", type=float, default=40.0)
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--share", action="store_true", help="more handles than devices: share them round-robin")
    ap.add_argument("--align", action="store_true")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    ndev = capi.load().wfm_device_count()
    threads = a.threads or (os.cpu_count() or 1)
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "c4.fa")
        t0 = time.time()
        names, lengths = synth.write_fasta(fa, synth.pangenome(a.haps, int(a.mbp * 1e6)))
        gen_s = time.time() - t0
        ref = None
        for n in [int(x) for x in a.gpus.split(",")]:
            if n > ndev and not a.share:
                print(json.dumps({"gpus": n, "skipped": f"only {ndev} device(s); --share lets handles share them"}), flush=True)
                continue
            hs = [capi.Handle(i % ndev) for i in range(n)]
            try:
                m = os.path.join(d, f"map{n}.paf")
                t0 = time.time()
                s = capi.map_paf_multi(hs, fa, m, params=capi.map_default_params(threads=threads))
                map_s = time.time() - t0
                out = {"gpus": n, "devices": min(n, ndev), "haps": a.haps, "target_bp": int(s.target_bp), "threads": threads, "gen_s": round(gen_s, 1),
                       "map_wall_s": round(map_s, 3), "ms_index": round(s.ms_index), "ms_replicate": round(s.ms_replicate), "ms_map": round(s.ms_map),
                       "ms_filter": round(s.ms_filter), "records": int(s.written), "pct": round(float(s.percentage_identity), 4)}
                digest = hashlib.sha256(open(m, "rb").read()).hexdigest()
                if a.align:
                    al = os.path.join(d, f"aln{n}.paf")
                    t0 = time.time()
                    r = capi.align_paf_multi(hs, fa, m, al, params={"threads": threads})
                    out.update(align_wall_s=round(time.time() - t0, 3), aligned_bp=int(r.aligned_bp), aligned_bp_per_s=round(r.aligned_bp / (time.time() - t0)),
                               cells=int(r.cells))
                    digest = hashlib.sha256(open(al, "rb").read()).hexdigest()
                    os.unlink(al)
                if ref is None:
                    ref = digest
                out["same_bytes_as_first_run"] = digest == ref
                print(json.dumps(out), flush=True)
            finally:
                for h in hs:
                    h.close()


if __name__ == "__main__":
    main()
