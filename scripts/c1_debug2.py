import hashlib, json, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, synth
d = tempfile.mkdtemp()
fa = os.path.join(d, "y.fa")
synth.write_fasta(fa, [(n, s.tobytes()) for n, s in synth.yeast_like(8, 8, 1_600_000)])
h1 = capi.Handle(0)
m = os.path.join(d, "m.paf")
capi.map_paf(h1, fa, m, params=capi.map_default_params(threads=16))
a1 = os.path.join(d, "a1.paf")
capi.align_paf(h1, fa, m, a1, params={"threads": 16})
one = open(a1).read().splitlines()
hs = [capi.Handle(0) for _ in range(3)]
for rep in range(3):
    a3 = os.path.join(d, "a3.paf")
    capi.align_paf_multi(hs, fa, m, a3, params={"threads": 16})
    three = open(a3).read().splitlines()
    k1 = {tuple(l.split("\t")[:9]): l for l in one}
    k3 = {tuple(l.split("\t")[:9]): l for l in three}
    diff_cg = [k for k in k1 if k in k3 and k1[k] != k3[k]]
    print(json.dumps({"rep": rep, "one": len(one), "three": len(three), "only_one": len(set(k1) - set(k3)), "only_three": len(set(k3) - set(k1)), "same_key_other_text": len(diff_cg),
                      "order_same": [tuple(l.split("\t")[:9]) for l in one] == [tuple(l.split("\t")[:9]) for l in three],
                      "example": [list(k) for k in list(set(k3) - set(k1))[:2]] + [list(k) for k in list(set(k1) - set(k3))[:2]]}), flush=True)
