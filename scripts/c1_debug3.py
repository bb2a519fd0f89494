import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, synth
d = tempfile.mkdtemp()
fa = os.path.join(d, "y.fa")
synth.write_fasta(fa, [(n, s.tobytes()) for n, s in synth.yeast_like(8, 8, 1_600_000)])
h1 = capi.Handle(0)
m = os.path.join(d, "m.paf")
capi.map_paf(h1, fa, m, params=capi.map_default_params(threads=16))
nh = int(sys.argv[1]) if len(sys.argv) > 1 else 1
hs = [capi.Handle(0) for _ in range(nh)]
for rep in range(3):
    a3 = os.path.join(d, "a3.paf")
    capi.align_paf_multi(hs, fa, m, a3, params={"threads": 16})
    print(json.dumps({"handles": nh, "rep": rep, "records": sum(1 for _ in open(a3))}), flush=True)
