#!/bin/bash
# rocprofv3 passes of one command, summarised into profiles/<name>.md by scripts/prof_summary.py:
#   kernel trace (+ --stats), then PMC passes in runs of their own (never together with a trace domain): FETCH_SIZE,
#   WRITE_SIZE, and SQ wave-cycle accounting.
# usage: scripts/profile_cmd.sh NAME "TITLE" [--sq] [--traffic] -- command ...
set -u
name=$1; title=$2; shift 2
sq=0; traffic=0
while [ "$1" != "--" ]; do case "$1" in --sq) sq=1;; --traffic) traffic=1;; esac; shift; done
shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$name
mkdir -p "$out" "$root/profiles"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$out/t" -o t -- "$@" > "$out/trace.log" 2>&1
dbs=()
if [ $traffic = 1 ]; then
  rocprofv3 --pmc FETCH_SIZE -d "$out/f" -o f -- "$@" > "$out/fetch.log" 2>&1; dbs+=("$out/f/f_results.db")
  rocprofv3 --pmc WRITE_SIZE -d "$out/w" -o w -- "$@" > "$out/write.log" 2>&1; dbs+=("$out/w/w_results.db")
fi
if [ $sq = 1 ]; then
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d "$out/s" -o s -- "$@" > "$out/sq.log" 2>&1
  dbs+=("$out/s/s_results.db")
fi
cd "$root"
python scripts/prof_summary.py "profiles/$name.md" "$title" "$out/t/t_results.db" "${dbs[@]}" --bench "$out/trace.log" > /dev/null
mkdir -p "$root/gpurun_out/profiles_out"
cp "profiles/$name.md" "$root/gpurun_out/profiles_out/"   # profiles/ does not travel back from the GPU box, gpurun_out/ does
rm -rf "$out"/t "$out"/f "$out"/w "$out"/s                 # the databases stay on the box (tens of MB)
tail -n +1 "profiles/$name.md" | head -60
