#!/bin/bash
# A/B runs of the C3 bench line (no secondary legs, no CPU baseline) under environment settings given as arguments, e.g.
#   scripts/ab_c3.sh "" "WFM_TILE_FAST=0"
# prints per setting: ms/step, the tile kernel's exclusive launch time and busy time per step
root=$(cd "$(dirname "$0")/.." && pwd)
for setting in "$@"; do
  echo "==== setting: [$setting]"
  env $setting python $root/bench.py --steps ${STEPS:-10} --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c '
import sys, json
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        r = d["roofline"]
        print("  C3 ms/step %.2f  tile excl launch %.4f ms x %.0f  busy/step %s  cigar_identical %s" % (d["ms_per_step"], r.get("avg_launch_ms_exclusive", 0), r.get("launches_exclusive_per_step", 0), json.dumps(d.get("kernel_busy_ms_per_step")), d.get("cigar_identical_rate")))
'
done
