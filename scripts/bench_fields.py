import sys, json
for l in sys.stdin:
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    r = d["roofline"]
    print(sys.argv[1] if len(sys.argv) > 1 else "", "ms/step %.1f" % d["ms_per_step"], "frac %.3f excl %.3f" % (r["frac"], r["frac_exclusive"]), "launch excl %.4f ms x %.0f" % (r["avg_launch_ms_exclusive"], r["launches_exclusive_per_step"]),
          "busy", {k: round(v, 1) for k, v in d.get("kernel_busy_ms_per_step", {}).items()})
