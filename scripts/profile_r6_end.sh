set -u
root=$GRAFT_REPO_ROOT
po=$root/gpurun_out/profiles_out_end
mkdir -p "$po"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/p6
rocprofv3 --kernel-trace --stats -d /tmp/p6/map -o t -- python $root/scripts/c4_rank.py > /tmp/p6_map.log 2>/dev/null
cd "$root"
python scripts/prof_summary.py "$po/r6_map.md" "r6 (round's end): map phase of one rank of C4 at full size (8 x 249 Mbp, one query haplotype: 249 k fragments; scripts/c4_rank.py)" "$(find /tmp/p6/map -name '*results.db' | head -1)" --bench /tmp/p6_map.log > /dev/null
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p6/c4 -o t -- python $root/scripts/c4_rank.py --mbp 40 --align > /tmp/p6_c4.log 2>&1
cd "$root"
python scripts/prof_summary.py "$po/r6_c4.md" "r6 (round's end): one rank of the 40 Mbp C4 variant, map + align (scripts/c4_rank.py --mbp 40 --align)" "$(find /tmp/p6/c4 -name '*results.db' | head -1)" --bench /tmp/p6_c4.log > /dev/null
ls -la "$po"
