#!/bin/bash
# C1 substitute in fresh processes: first (cold) and second pass of map + align, for ring chunk caps of 8 / 4 / 2 GB
for gb in 8 4 2; do echo "WFM_RING_CHUNK_GB=$gb"; WFM_RING_CHUNK_GB=$gb python scripts/c1_run.py --reps 2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print({k:j[k] for k in ('pass','map_s','align_s','ms_gpu','aligned_bp_per_s_align')})"; done
