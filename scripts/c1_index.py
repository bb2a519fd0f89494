"""Index build of the C1 substitute (8 yeast-like strains x 16 chromosomes, 97 Mbp) through wfm_index_build_sequences:
wall time per setting of WFM_WINNOW_DEV_MIN (k-mers from which on a sequence is winnowed on the device)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, synth
recs = [(n, s.tobytes()) for n, s in synth.yeast_like(8, 16, 12_000_000)]
seqs = [s for _, s in recs]
h = capi.Handle(0)
for rep in range(3):
    t = time.perf_counter()
    ix, nwin = h.index_build_sequences(seqs, 15, 1000, 29, threads=int(os.environ.get("THREADS", os.cpu_count() or 1)))
    dt = time.perf_counter() - t
    print(json.dumps({"rep": rep, "sequences": len(seqs), "bp": sum(map(len, seqs)), "windows": int(nwin), "index_s": round(dt, 4), "dev_min": os.environ.get("WFM_WINNOW_DEV_MIN", "default")}), flush=True)
    if ix is not None:
        ix.free()
