#!/bin/bash
# A/B runs of north_star's C4 all-vs-all at full size on one GPU (bench.py's C4_rank_full + C4_all_vs_all legs, one fresh process each) under environment
# settings given as arguments, e.g.   scripts/ab_allvsall.sh "" "WFM_ALIGN_WORKERS=6"
root=$(cd "$(dirname "$0")/.." && pwd)
for setting in "$@"; do
  echo "==== setting: [$setting]"
  env $setting python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs C4_rank_full,C4_all_vs_all 2>/dev/null | python -c '
import sys, json
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        for k in ("C4_rank_full", "C4_all_vs_all"):
            v = d["legs"].get(k)
            if v: print("  %-14s %s" % (k, json.dumps({q: (round(v[q], 3) if isinstance(v[q], float) else v[q]) for q in ("map_s", "ms_filter", "align_s", "ms_gpu", "gpu_share_of_align", "wall_s", "aligned_bp_per_s_map_and_align") if q in v})))
'
done
