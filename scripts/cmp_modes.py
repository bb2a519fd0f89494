"""Debug helper: run the C3 batch through the step kernel only, the LDS tile kernel and the
register tile kernel and compare the op strings (GPU vs GPU), then spot-check vs the oracle."""
import os, subprocess, sys, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from wfmash_amd import capi, synth
    h = capi.Handle(0)
    pairs = synth.pairs("C3", n_pairs=int(sys.argv[3]))
    res = h.align(pairs)
    pickle.dump([(r.status, r.score, r.ops) for r in res], open(sys.argv[2], "wb"))
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "64"
outs = {}
for name, env in [("step", {"WFM_TILE": "0"}), ("lds", {"WFM_TILE_REG": "0"}), ("reg", {})]:
    e = dict(os.environ); e.update(env)
    f = f"/tmp/cmp_{name}.pkl"
    subprocess.check_call([sys.executable, __file__, "child", f, n], env=e)
    outs[name] = pickle.load(open(f, "rb"))
from oracle import pyoracle as O
from wfmash_amd import synth
pairs = synth.pairs("C3", n_pairs=int(n))
for name in ("lds", "reg"):
    bad = [i for i in range(len(pairs)) if outs[name][i][2] != outs["step"][i][2]]
    print(name, "differs from step kernel on pairs:", bad)
    for i in bad[:4]:
        st, sc, ops = outs[name][i]
        print("  pair", i, "status", st, "score", sc, "step score", outs["step"][i][1],
              "check", O.ops_check(ops, *pairs[i]) if ops else None, "implied", O.ops_score(ops) if ops else None)
