#!/bin/bash
# Round-3 profiles (run on the GPU box; everything lands in gpurun_out/profiles_out/, to be copied into profiles/):
#   r3_align_excl.md   rocprofv3 --kernel-trace of `WFM_OVERLAP=0 python bench.py` (C3): the per-launch figures of roofline.frac
#   r3_traffic.json    FETCH_SIZE / WRITE_SIZE passes of the same command (HBM bytes per launch of the dominant kernel)
#   r3_sq.json         SQ pass of the same command (valu_frac, wait_frac)
#   r3_c4.md           kernel trace of one rank of the 40 Mbp C4 variant, map + align
#   r3_c2.md           kernel trace + stage timings of C2 (LPA.subset all-vs-all)
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
po=$root/gpurun_out/profiles_out
mkdir -p "$po"
export TMPDIR=/tmp
cd /tmp
B="python $root/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline"
WFM_OVERLAP=0 rocprofv3 --kernel-trace --stats -d /tmp/p3/t -o t -- $B > /tmp/p3_trace.log 2>&1
WFM_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE -d /tmp/p3/f -o f -- $B > /tmp/p3_f.log 2>&1
WFM_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE -d /tmp/p3/w -o w -- $B > /tmp/p3_w.log 2>&1
WFM_OVERLAP=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/p3/s -o s -- $B > /tmp/p3_s.log 2>&1
cd "$root"
T=$(find /tmp/p3/t -name "*results.db" | head -1); F=$(find /tmp/p3/f -name "*results.db" | head -1); W=$(find /tmp/p3/w -name "*results.db" | head -1); S=$(find /tmp/p3/s -name "*results.db" | head -1)
python scripts/prof_summary.py "$po/r3_align_excl.md" "r3: C3, WFM_OVERLAP=0 (one stream, launches one after the other): the per-launch figures bench.py reports as roofline.frac" "$T" "$F" "$W" "$S" --bench /tmp/p3_trace.log > /dev/null
python - "$T" "$F" "$W" "$po/r3_traffic.json" <<'PY'
import json, sqlite3, sys
t, f, w, out = sys.argv[1:5]
like = "%wfa_tile_reg_kernel<2, 1024, 5, 10, 25, 2, 1, false%"
n, tot, avg = sqlite3.connect(t).execute("select count(*), sum(end-start)/1e6, avg(end-start)/1e6 from kernels where name like ?", (like,)).fetchall()[0]
fs, nf = sqlite3.connect(f).execute("select sum(value), count(*) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like ?", (like,)).fetchall()[0]
ws, nw = sqlite3.connect(w).execute("select sum(value), count(*) from counters_collection where counter_name='WRITE_SIZE' and kernel_name like ?", (like,)).fetchall()[0]
fb, wb = fs * 1024 * 2, ws * 1024
d = {"kernel": "wfa_tile_reg_kernel", "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `WFM_OVERLAP=0 python bench.py --steps 3 --warmup 1` (scripts/profile_r3.sh)",
     "dispatches": nf, "dispatches_trace": n, "fetch_size_kb_sum": fs, "write_size_kb_sum": ws, "fetch_bytes_corrected": fb, "write_bytes": wb,
     "traffic_bytes_per_launch": (fb + wb) / nf, "avg_launch_ms": avg, "total_ms": tot, "hbm_GBps_per_launch": (fb + wb) / nf / (avg * 1e-3) / 1e9,
     "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); per launch = (fetch + write) / dispatches"}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d))
PY
python scripts/make_sq_json.py "$T" "$S" "wfa_tile_reg_kernel<2, 1024, 5, 10, 25, 2, 1, false" "$po/r3_sq.json" "rocprofv3 --pmc SQ_* pass + kernel trace of \`WFM_OVERLAP=0 python bench.py --steps 3 --warmup 1\` (scripts/profile_r3.sh)"
# C4 rank (40 Mbp variant), default mode
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p3/c4 -o t -- python $root/scripts/c4_rank.py --mbp 40 --align > /tmp/p3_c4.log 2>&1
cd "$root"
python scripts/prof_summary.py "$po/r3_c4.md" "r3: one rank of the 40 Mbp C4 variant, map + align (scripts/c4_rank.py --mbp 40 --align)" "$(find /tmp/p3/c4 -name '*results.db' | head -1)" --bench /tmp/p3_c4.log > /dev/null
# C2 with stage timings
cd /tmp
WFM_DEBUG=1 rocprofv3 --kernel-trace --stats -d /tmp/p3/c2 -o t -- python $root/scripts/legs_debug.py c2 --reps 2 > /tmp/p3_c2.log 2> /tmp/p3_c2.err
cd "$root"
python scripts/prof_summary.py "$po/r3_c2.md" "r3: C2 (LPA.subset all-vs-all, -p 90 -P 50k), two passes in one process (scripts/legs_debug.py c2)" "$(find /tmp/p3/c2 -name '*results.db' | head -1)" --bench /tmp/p3_c2.log > /dev/null
{ echo; echo "## stage timings of the second (warm) pass (WFM_DEBUG=1)"; echo '```'; awk '/==== C2 pass 1/{f=1} f' /tmp/p3_c2.err | grep "wfmash::align\]\|wflign\]\|align_batch\|wall:\|score bounds\|add_minmers_multi\|index_build" | cut -c1-400; echo '```'; } >> "$po/r3_c2.md"
# C1 substitute (8 yeast-like strains all-vs-all), one pass
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p3/c1 -o t -- python $root/scripts/c1_run.py --reps 1 > /tmp/p3_c1.log 2>/dev/null
cd "$root"
python scripts/prof_summary.py "$po/r3_c1.md" "r3: C1 substitute (8 yeast-like strains x 16 chromosomes, 96 Mbp, all-vs-all, defaults), map + align, one pass in a fresh process (scripts/c1_run.py --reps 1)" "$(find /tmp/p3/c1 -name '*results.db' | head -1)" --bench /tmp/p3_c1.log > /dev/null
# the map phase of a full-size C4 rank (item 7's three kernels: sketch_fragments, l1_sweep, l2_slide per 249 k fragments)
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p3/map -o t -- python $root/scripts/c4_rank.py > /tmp/p3_map.log 2>/dev/null
cd "$root"
python scripts/prof_summary.py "$po/r3_map.md" "r3: map phase of one rank of C4 at full size (8 x 249 Mbp, one query haplotype: 249 k fragments; scripts/c4_rank.py)" "$(find /tmp/p3/map -name '*results.db' | head -1)" --bench /tmp/p3_map.log > /dev/null
ls -la "$po"
