#!/bin/bash
# A/B runs of the C2 / scaled C4 legs (scripts/legs_run.py) under environment settings given as arguments, e.g.
#   scripts/ab_legs.sh "" "WFM_P2_THREADS=256" "WFM_TILE_V2=0"
# prints one line per (setting, leg, pass): align_s, ms_gpu, algorithmic_frac_gpu
root=$(cd "$(dirname "$0")/.." && pwd)
for setting in "$@"; do
  echo "==== setting: [$setting]"
  env $setting WFM_DEBUG=0 python $root/scripts/legs_run.py c4 c2 --reps 3 2>/dev/null | python -c '
import sys, json
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        print("  %-16s pass %d  align_s %.4f  ms_gpu %7.2f  map_s %.3f  aligned Mbp/s %7.1f  frac_gpu %s" % (d["leg"], d["pass"], d["align_s"], d["ms_gpu"], d["map_s"], d.get("aligned_bp_per_s", d.get("aligned_bp_per_s_align", 0)) / 1e6, ("%.3f" % (48.0 * d["cells"] / (d["ms_gpu"] * 1e-3) / 8e12)) if d["ms_gpu"] else "-"))
'
done
