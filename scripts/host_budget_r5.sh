#!/bin/bash
# The host budget of one rank of an 8-GPU node on round 5's code: the FULL-size C4 rank (8 x 249 Mbp, one query
# haplotype: 35.9 k records, 1.75 Gbp), map + align, three passes each with all host threads, nproc / 8 and nproc / 16.
# Output: gpurun_out/profiles_out/r5_host_budget.jsonl (one line per pass), summarised in profiles/r5_host_budget.md
root=$(cd "$(dirname "$0")/.." && pwd)
po=$root/gpurun_out/profiles_out; mkdir -p "$po"
n=$(nproc)
: > "$po/r5_host_budget.jsonl"
for t in $n $((n / 8)) $((n / 16)); do
  python "$root/scripts/c4_align_repeat.py" --threads $t --reps 3 2>/dev/null | grep '^{' >> "$po/r5_host_budget.jsonl"
done
echo "nproc $n"; cut -c1-700 "$po/r5_host_budget.jsonl"
