#!/bin/bash
# Round-6 profiles (run on the GPU box; everything lands in gpurun_out/profiles_out/, to be copied into profiles/):
#   r6_align_excl.md   rocprofv3 --kernel-trace of `WFM_OVERLAP=0 python bench.py` (C3): the per-launch figures of roofline.frac
#   r6_traffic.json    FETCH_SIZE / WRITE_SIZE passes of the same command (HBM bytes per launch of the dominant kernel)
#   r6_sq.json         SQ pass of the same command (valu_frac, wait_frac, instructions per wave)
#   r6_c4.md / r6_c2.md / r6_c1.md / r6_map.md   kernel traces of the other legs
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
po=$root/gpurun_out/profiles_out
mkdir -p "$po"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/p6
B="python $root/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline"
WFM_OVERLAP=0 rocprofv3 --kernel-trace --stats -d /tmp/p6/t -o t -- $B > /tmp/p6_trace.log 2>&1
WFM_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE -d /tmp/p6/f -o f -- $B > /tmp/p6_f.log 2>&1
WFM_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE -d /tmp/p6/w -o w -- $B > /tmp/p6_w.log 2>&1
WFM_OVERLAP=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/p6/s -o s -- $B > /tmp/p6_s.log 2>&1
WFM_OVERLAP=0 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM -d /tmp/p6/s2 -o s2 -- $B > /tmp/p6_s2.log 2>&1
cd "$root"
T=$(find /tmp/p6/t -name "*results.db" | head -1); F=$(find /tmp/p6/f -name "*results.db" | head -1); W=$(find /tmp/p6/w -name "*results.db" | head -1); S=$(find /tmp/p6/s -name "*results.db" | head -1); S2=$(find /tmp/p6/s2 -name "*results.db" | head -1)
python scripts/prof_summary.py "$po/r6_align_excl.md" "r6: C3, WFM_OVERLAP=0 (one stream, launches one after the other): the per-launch figures bench.py reports as roofline.frac" "$T" "$F" "$W" "$S" --bench /tmp/p6_trace.log > /dev/null
python - "$T" "$po/r6_align_excl.md" <<'PY'
# how the table above and bench.py's live figures meet: bench.py times BLOCKS (its events stand around the one or two instantiations of the tile kernel a
# block launches), rocprofv3 lists the instantiations apart
import sqlite3, sys
t, out = sys.argv[1:3]
rows = sqlite3.connect(t).execute("select name, count(*), sum(end-start)/1e6 from kernels where name like '%wfa_tile2_kernel<1024, false, true,%' group by name").fetchall()
passes = 6  # --warmup 1 --steps 3 + bench.py's two untimed WFM_OVERLAP=0 passes
tot = sum(r[2] for r in rows)
with open(out, "a") as f:
    f.write("\n## how this meets bench.py's `roofline`\n\n")
    f.write("The profiled process runs %d passes of the batch (warm-up 1, steps 3, and the two untimed passes bench.py itself runs with `WFM_OVERLAP=0`); all of them are one chain of launches here.\n" % passes)
    f.write("The tile kernel's two phase-1 instantiations (with and without per-score maxima) together: %d launches, %.1f ms = **%.2f ms per pass** -- bench.py's `roofline.tile_kernel_ms_exclusive_per_pass`, which it reports as\n" % (sum(r[1] for r in rows), tot, tot / passes))
    f.write("`launches_exclusive_per_step` BLOCKS x `avg_launch_ms_exclusive` (its events stand around the one or two instantiations a block of 100 scores launches).\n")
PY
python - "$T" "$F" "$W" "$po/r6_traffic.json" <<'PY'
import json, sqlite3, sys
t, f, w, out = sys.argv[1:5]
like = "%wfa_tile2_kernel<1024, false, true, false%"
n, tot, avg = sqlite3.connect(t).execute("select count(*), sum(end-start)/1e6, avg(end-start)/1e6 from kernels where name like ?", (like,)).fetchall()[0]
fs, nf = sqlite3.connect(f).execute("select sum(value), count(*) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like ?", (like,)).fetchall()[0]
ws, nw = sqlite3.connect(w).execute("select sum(value), count(*) from counters_collection where counter_name='WRITE_SIZE' and kernel_name like ?", (like,)).fetchall()[0]
fb, wb = fs * 1024 * 2, ws * 1024
d = {"kernel": "wfa_tile2_kernel", "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `WFM_OVERLAP=0 python bench.py --steps 3 --warmup 1` (scripts/profile_r6.sh)",
     "dispatches": nf, "dispatches_trace": n, "fetch_size_kb_sum": fs, "write_size_kb_sum": ws, "fetch_bytes_corrected": fb, "write_bytes": wb,
     "traffic_bytes_per_launch": (fb + wb) / nf, "avg_launch_ms": avg, "total_ms": tot, "hbm_GBps_per_launch": (fb + wb) / nf / (avg * 1e-3) / 1e9,
     "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); per launch = (fetch + write) / dispatches; measured on the builder's box, not in the bench run"}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d))
PY
python scripts/make_sq_json.py "$T" "$S,$S2" "wfa_tile2_kernel<1024, false, true, false" "$po/r6_sq.json" "rocprofv3 --pmc SQ_* pass + kernel trace of \`WFM_OVERLAP=0 python bench.py --steps 3 --warmup 1\` (scripts/profile_r6.sh); measured on the builder's box, not in the bench run"
# C4 rank (40 Mbp variant), default mode
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p6/c4 -o t -- python $root/scripts/c4_rank.py --mbp 40 --align > /tmp/p6_c4.log 2>&1
cd "$root"
python scripts/prof_summary.py "$po/r6_c4.md" "r6: one rank of the 40 Mbp C4 variant, map + align (scripts/c4_rank.py --mbp 40 --align)" "$(find /tmp/p6/c4 -name '*results.db' | head -1)" --bench /tmp/p6_c4.log > /dev/null
# C2 with stage timings
cd /tmp
WFM_DEBUG=1 rocprofv3 --kernel-trace --stats -d /tmp/p6/c2 -o t -- python $root/scripts/legs_run.py c2 --reps 2 > /tmp/p6_c2.log 2> /tmp/p6_c2.err
cd "$root"
python scripts/prof_summary.py "$po/r6_c2.md" "r6: C2 (LPA.subset all-vs-all, -p 90 -P 50k), two passes in one process (scripts/legs_run.py c2)" "$(find /tmp/p6/c2 -name '*results.db' | head -1)" --bench /tmp/p6_c2.log > /dev/null
# C1 substitute (8 yeast-like strains all-vs-all), one pass
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p6/c1 -o t -- python $root/scripts/c1_run.py --reps 1 > /tmp/p6_c1.log 2>/dev/null
cd "$root"
python scripts/prof_summary.py "$po/r6_c1.md" "r6: C1 substitute (8 yeast-like strains x 16 chromosomes, 96 Mbp, all-vs-all, defaults), map + align, one pass in a fresh process (scripts/c1_run.py --reps 1)" "$(find /tmp/p6/c1 -name '*results.db' | head -1)" --bench /tmp/p6_c1.log > /dev/null
# the map phase of a full-size C4 rank
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p6/map -o t -- python $root/scripts/c4_rank.py > /tmp/p6_map.log 2>/dev/null
cd "$root"
python scripts/prof_summary.py "$po/r6_map.md" "r6: map phase of one rank of C4 at full size (8 x 249 Mbp, one query haplotype: 249 k fragments; scripts/c4_rank.py)" "$(find /tmp/p6/map -name '*results.db' | head -1)" --bench /tmp/p6_map.log > /dev/null
# C5
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p6/c5 -o t -- python $root/bench.py --config C5 --pairs 8 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline > /tmp/p6_c5.log 2>&1
cd "$root"
python scripts/prof_summary.py "$po/r6_c5.md" "r6: C5 (8 pairs of 100 kb at 15 %), align only (bench.py --config C5 --pairs 8)" "$(find /tmp/p6/c5 -name '*results.db' | head -1)" --bench /tmp/p6_c5.log > /dev/null
ls -la "$po"
# m3 for sequences under the device winnower's threshold: the host's threads against one set of launches per sequence
python scripts/winnow_short.py --threads 32 > "$po/r6_winnow_host.jsonl" 2>/dev/null
WFM_WINNOW_DEV_MIN=0 python scripts/winnow_short.py --threads 32 > "$po/r6_winnow_dev.jsonl" 2>/dev/null
python scripts/winnow_short.py --threads 256 > "$po/r6_winnow_host256.jsonl" 2>/dev/null
WFM_WINNOW_DEV_MIN=0 python scripts/winnow_short.py --threads 256 > "$po/r6_winnow_dev256.jsonl" 2>/dev/null
cat "$po"/r6_winnow_*.jsonl
