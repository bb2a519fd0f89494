#!/bin/bash
# SQ counters of the map phase's kernels on a full-size C4 rank (scripts/c4_rank.py): what the device winnower's time is made of.
# Two passes (the SQ counters do not all fit one); prints the winnow kernel's rows.  Output: gpurun_out/winnow_pmc/
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/winnow_pmc
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/wp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/wp/a -o a -- python $root/scripts/c4_rank.py > /tmp/wp_a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS -d /tmp/wp/b -o b -- python $root/scripts/c4_rank.py > /tmp/wp_b.log 2>&1
cd "$root"
python - "$(find /tmp/wp/a -name '*results.db' | head -1)" "$(find /tmp/wp/b -name '*results.db' | head -1)" "$out/winnow_pmc.json" <<'PY'
import json, sqlite3, sys
a, b, out = sys.argv[1:4]
res = {}
for db in (a, b):
    for name, c, v, n in sqlite3.connect(db).execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like '%winnow_chunks%' or kernel_name like '%pf_cand%' or kernel_name like '%l2_slide%' group by kernel_name, counter_name"):
        k = name[:name.index("(")] if "(" in name else name
        res.setdefault(k, {})[c] = v
        res[k]["dispatches"] = n
json.dump(res, open(out, "w"), indent=1)
for k, d in res.items():
    w = max(1.0, d.get("SQ_WAVES", 1))
    print(k, "dispatches", d.get("dispatches"), "waves %.0f" % w, " per wave: VALU %.0f SALU %.0f LDS %.0f VMEM %.0f SMEM %.0f BRANCH %.0f" % (d.get("SQ_INSTS_VALU", 0) / w, d.get("SQ_INSTS_SALU", 0) / w, d.get("SQ_INSTS_LDS", 0) / w,
          (d.get("SQ_INSTS_VMEM_RD", 0) + d.get("SQ_INSTS_VMEM_WR", 0)) / w, d.get("SQ_INSTS_SMEM", 0) / w, d.get("SQ_INSTS_BRANCH", 0) / w),
          " wave cycles (quad) %.3e wait_any %.2f wait_inst_any %.2f active_inst_any %.2f  wait_inst_lds(raw) %.3e bank conflicts %.3e busy_cu_cycles %.3e" % (d.get("SQ_WAVE_CYCLES", 0), d.get("SQ_WAIT_ANY", 0) / max(1, d.get("SQ_WAVE_CYCLES", 1)),
          d.get("SQ_WAIT_INST_ANY", 0) / max(1, d.get("SQ_WAVE_CYCLES", 1)), d.get("SQ_ACTIVE_INST_ANY", 0) / max(1, d.get("SQ_WAVE_CYCLES", 1)), d.get("SQ_WAIT_INST_LDS", 0), d.get("SQ_LDS_BANK_CONFLICT", 0), d.get("SQ_BUSY_CU_CYCLES", 0)))
PY
