import sys, time, random
sys.path.insert(0, '.')
from wfmash_amd import capi, synth
from oracle import pyoracle as O
h = capi.Handle(0)
print(h.device_name())
rng = random.Random(5)
def rnd(n): return bytes(rng.choice(b"ACGT") for _ in range(n))
items = []
for n, rate in [(0,0),(1,0),(50,0.1),(100,0.05),(150,0.05),(300,0.1),(1000,0.05),(3000,0.05),(3000,0.15),(8000,0.1),(20000,0.05)]:
    p = rnd(n); t = synth.mutate(p, rate, 77+n) if n else b""
    items.append((p,t))
items.append((rnd(500), b""))
items.append((b"", rnd(40)))
items.append((rnd(2000), rnd(300)))
items.append((rnd(300)*1, rnd(2500)))
p = rnd(5000); items.append((p,p))
t0=time.time(); res = h.align(items); print("gpu time", time.time()-t0)
st = h.stats(); print("levels",st.levels,"bp_jobs",st.bp_jobs,"base_jobs",st.base_jobs,"ms_bp",st.ms_breakpoint,"ms_base",st.ms_base,"cells",st.cells)
bad=0
for (p,t),r in zip(items,res):
    rc,ops,sc,s2 = O.align_biwfa(p,t)
    ok = (r.status==0 and r.ops==ops)
    chk = O.ops_check(r.ops,p,t) if r.ops is not None else 'NA'
    print(len(p),len(t),"status",r.status,"score",r.score,sc,"identical",ok,"check",chk, "gpu_cells", r.cells, "cpu_cells", s2.cells)
    bad += (not ok)
print("BAD",bad)
