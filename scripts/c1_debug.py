"""C1-shaped all-vs-all (yeast-like substitute) map + align with WFM_DEBUG=1: prints the library's complaints about failed
problems and compares the aligned output of several switch settings (each in a process of its own)."""
import hashlib, json, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi, synth

def main():
    d = tempfile.mkdtemp()
    fa = os.path.join(d, "y.fa")
    synth.write_fasta(fa, [(n, s.tobytes()) for n, s in synth.yeast_like(8, 8, 1_600_000)])
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wfmash_amd", "wfmash-hip")
    m = os.path.join(d, "m.paf")
    subprocess.check_call([cli, "-m", "-t", "16", "--out", m, fa], cwd=d, stderr=subprocess.DEVNULL)
    nmap = sum(1 for _ in open(m))
    ref = None
    for env in sys.argv[1:] or ["WFM_X=0"]:
        e = dict(os.environ, WFM_DEBUG="1")
        for kv in env.split():
            k, v = kv.split("=")
            e[k] = v
        out = os.path.join(d, "a.paf")
        r = subprocess.run([cli, "-i", m, "-t", "16", "--out", out, fa], cwd=d, env=e, capture_output=True, text=True)
        lines = open(out).read().splitlines()
        dig = hashlib.sha256("\n".join(lines).encode()).hexdigest()[:12]
        keys = {tuple(l.split("\t")[:9]) for l in lines}
        if ref is None:
            ref = keys
        complaints = [l for l in r.stderr.splitlines() if "problem " in l or "unreachable" in l.lower() or "overflowed" in l]
        print(json.dumps({"env": env, "mappings": nmap, "records": len(lines), "digest": dig, "missing_vs_first": len(ref - keys), "extra_vs_first": len(keys - ref),
                          "complaints": complaints[:6], "n_complaints": len(complaints)}), flush=True)

main()
