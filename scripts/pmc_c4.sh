#!/bin/bash
# PMC passes (runs of their own, no trace domains) over the scaled C4 rank; per kernel sums into OUT.md
out=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/$out"
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  env WFM_OVERLAP=0 WFM_DEBUG=0 "$@" rocprofv3 --pmc $set -d /tmp/pmc$i -o p -- python $root/scripts/legs_debug.py c4 --reps 1 > $root/$out/pmc$i.log 2>&1
done
cd $root
python - $out <<'PY'
import sqlite3, sys, glob
out = sys.argv[1]
rows = {}
for i in (1, 2, 3):
    for db in glob.glob(f"/tmp/pmc{i}/**/*results.db", recursive=True):
        con = sqlite3.connect(db)
        for name, ctr, val, n in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like '%wfa_%' group by kernel_name, counter_name"):
            k = name.split("(")[0].replace("wfm::", "").replace("void ", "")[-70:]
            rows.setdefault(k, {})[ctr] = val
            rows[k]["dispatches"] = n
with open(f"{out}/pmc.md", "w") as f:
    for k, d in sorted(rows.items()):
        f.write(f"## {k}\n")
        for c, v in sorted(d.items()):
            f.write(f"- {c}: {v:.6g}\n")
print(open(f"{out}/pmc.md").read()[:6000])
PY
