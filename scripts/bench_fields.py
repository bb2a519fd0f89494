"""The figures of a bench line a reader looks at first (stdin or file: the JSON line of bench.py): C3, the roofline, every secondary leg's times."""
import json
import sys

src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
for l in src:
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    r = d["roofline"]
    print("C3 ms/step %.2f  value %.3e %s" % (d["ms_per_step"], d["value"], d["unit"]))
    print("roofline: bound %s frac %.3f (achieved %.3e / peak %.3e %s); hbm yardstick frac %.3f; launch excl %.4f ms x %.0f" % (
        r["bound"], r["frac"], r["achieved"], r["peak"], r["unit"], r.get("hbm_yardstick", {}).get("frac", float("nan")), r["avg_launch_ms_exclusive"], r["launches_exclusive_per_step"]))
    print("cpu_baseline:", {k: d["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind")} if d.get("cpu_baseline") else None)
    for name, leg in (d.get("secondary") or {}).items():
        if isinstance(leg, dict):
            keep = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in leg.items() if k in ("ms_per_step", "map_s", "align_s", "ms_gpu", "ms_filter", "gpu_share_of_align", "algorithmic_frac_gpu", "aligned_bp_per_s", "records", "aligned_bp_per_s_map_and_align", "wall_s", "queries")}
            par = leg.get("parity") or {}
            print(f"  {name}: {keep} parity {par.get('identical')}/{par.get('sampled_records')}")
