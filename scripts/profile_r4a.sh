#!/bin/bash
# Round-4 baseline counters BEFORE the kernel work (VERDICT r3 "next" #2): SQ passes + kernel traces of the scaled C4 rank
# and of C2 with the round-3 kernels.  Output: gpurun_out/profiles_out/r4_c4_sq.md, r4_c2_sq.md (copied into profiles/).
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
po=$root/gpurun_out/profiles_out
mkdir -p "$po"
export TMPDIR=/tmp
tag=${1:-r4}
for leg in c4 c2; do
  cd /tmp
  rm -rf /tmp/p4_$leg
  rocprofv3 --kernel-trace --stats -d /tmp/p4_$leg/t -o t -- python $root/scripts/legs_debug.py $leg --reps 1 > /tmp/p4_${leg}_t.log 2>/dev/null
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/p4_$leg/a -o a -- python $root/scripts/legs_debug.py $leg --reps 1 > /tmp/p4_${leg}_a.log 2>/dev/null
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH -d /tmp/p4_$leg/b -o b -- python $root/scripts/legs_debug.py $leg --reps 1 > /tmp/p4_${leg}_b.log 2>/dev/null
  cd "$root"
  python scripts/sq_report.py "$po/${tag}_${leg}_sq.md" "$tag: $leg leg of scripts/legs_debug.py (--reps 1), kernel trace + two SQ counter passes of their own (rocprofv3 --pmc, no trace domains)" \
    "$(find /tmp/p4_$leg/t -name '*results.db' | head -1)" "$(find /tmp/p4_$leg/a -name '*results.db' | head -1)" "$(find /tmp/p4_$leg/b -name '*results.db' | head -1)" /tmp/p4_${leg}_t.log
done
ls -la "$po"
