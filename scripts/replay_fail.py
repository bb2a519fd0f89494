import os, sys, glob, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi
from oracle import pyoracle as O
h = capi.Handle(0)
seen = set()
for fn in sorted(glob.glob("scripts/fails_tmp/fail_*.txt")):
    raw = open(fn, "rb").read().split(b"\n")
    pl, tl, hint = map(int, raw[0].split())
    p, t = raw[1], raw[2]
    if (p, t) in seen:
        continue
    seen.add((p, t))
    rc, ops, sc, _ = O.align_biwfa(p, t)
    ub = int(h.score_bounds([(p, t)])[0])
    outs = []
    for env_bound in ("1",):
        r = h.align([(p, t)])[0]
        outs.append((r.status, r.score, r.ops == ops))
        rr = h.align([(p, t)] * 7 + [(p[:900], t[:800])])
        outs.append([(x.status, x.score) for x in rr[:3]])
    print(json.dumps({"file": os.path.basename(fn), "pl": pl, "tl": tl, "hint": hint, "oracle_score": sc, "ub": ub, "runs": outs}), flush=True)
