#!/bin/bash
# A/B runs of the C1 substitute (scripts/c1_run.py, map + align, fresh process each) under environment settings given as arguments, e.g.
#   scripts/ab_c1.sh "" "WFM_P2_THREADS=256"
# prints per (setting, pass): align_s, ms_gpu
root=$(cd "$(dirname "$0")/.." && pwd)
for setting in "$@"; do
  echo "==== setting: [$setting]"
  env $setting python $root/scripts/c1_run.py --reps ${REPS:-2} 2>/dev/null | python -c '
import sys, json
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        print("  C1 pass %d  align_s %.3f  ms_gpu %7.1f  map_s %.3f  aligned Mbp/s %7.1f" % (d["pass"], d["align_s"], d["ms_gpu"], d["map_s"], d["aligned_bp_per_s_align"] / 1e6))
'
done
