"""The bench's order of legs in one process (a 40 Mbp C4 rank, then the C1 substitute) with WFM_DEBUG=1: what does the C1 leg wait for
when it runs on handles whose arenas the C4 leg has sized?"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WFM_DEBUG", "1")
from wfmash_amd import capi, synth
h = capi.Handle(0)
thr = os.cpu_count()
with tempfile.TemporaryDirectory() as td:
    fa = os.path.join(td, "c4.fa"); names, _ = synth.write_fasta(fa, synth.pangenome(8, 40_000_000))
    open(os.path.join(td, "q.txt"), "w").write(names[0] + "\n")
    m, a = os.path.join(td, "m.paf"), os.path.join(td, "a.paf")
    capi.map_paf(h, fa, m, params=capi.map_default_params(threads=thr, query_list=os.path.join(td, "q.txt")))
    t = time.perf_counter(); capi.align_paf(h, fa, m, a, params={"threads": thr}); print("C4 align", time.perf_counter() - t, flush=True)
    print("==== C1", file=sys.stderr, flush=True)
    fa = os.path.join(td, "c1.fa"); synth.write_fasta(fa, synth.yeast_like(8, 16, 12_000_000))
    capi.map_paf(h, fa, m, params=capi.map_default_params(threads=thr))
    for rep in range(2):
        t = time.perf_counter(); s = capi.align_paf(h, fa, m, a, params={"threads": thr}); print("C1 align", rep, time.perf_counter() - t, "busy", s.ms_gpu, flush=True)
