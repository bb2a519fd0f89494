#!/bin/bash
# The host side of one GPU's share of a node: the 40 Mbp C4 variant (all 8 query haplotypes, map + align) through one
# device handle with all host threads, then with nproc/8 and nproc/16 -- what a rank of an 8-GPU node has.
# Output: gpurun_out/profiles_out/r3_host_budget.jsonl (one line per thread count), summarised in profiles/r3_host_budget.md
root=$(cd "$(dirname "$0")/.." && pwd)
po=$root/gpurun_out/profiles_out; mkdir -p "$po"
n=$(nproc)
: > "$po/r3_host_budget.jsonl"
for t in $n $((n / 8)) $((n / 16)); do
  python "$root/scripts/c4_node.py" --gpus 1 --align --threads $t 2>/dev/null | grep '^{' >> "$po/r3_host_budget.jsonl"
done
echo "nproc $n"; cut -c1-600 "$po/r3_host_budget.jsonl"
