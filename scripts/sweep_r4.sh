#!/bin/bash
# A/B sweep of align-side switches on one rank of the 40 Mbp C4 variant, warm: three passes in one process (scripts/legs_debug.py c4 --mbp 40), the last two shown
root=$(cd "$(dirname "$0")/.." && pwd)
for setting in "$@"; do
  env $setting WFM_DEBUG=0 python $root/scripts/legs_debug.py c4 --mbp 40 --reps 3 2>/dev/null | python -c '
import sys, json
rows = [json.loads(l) for l in sys.stdin if l.startswith("{")]
print("  [%s]" % sys.argv[1], "  ".join("align %.3f s, busy %.0f ms, %.0f Mbp/s" % (d["align_s"], d["ms_gpu"], d["aligned_bp"] / d["align_s"] / 1e6) for d in rows[1:]))
' "$setting"
done
