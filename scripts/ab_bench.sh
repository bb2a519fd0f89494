#!/bin/bash
# A/B runs of the whole bench line under environment settings given as arguments; prints the `legs` digest of each
#   scripts/ab_bench.sh "" "WFM_STREAMS=3 WFM_RING_CHUNK_GB=16"
root=$(cd "$(dirname "$0")/.." && pwd)
for setting in "$@"; do
  echo "==== setting: [$setting]"
  env $setting python $root/bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c '
import sys, json
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["legs"].items():
            print("  %-16s %s" % (k, json.dumps(v)))
'
done
