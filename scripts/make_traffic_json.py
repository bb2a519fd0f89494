#!/usr/bin/env python
"""profiles/<name>_traffic.json from the rocprofv3 passes scripts/profile_cmd.sh --traffic left under gpurun_out/prof_<name>/:
HBM bytes per launch of one kernel = (FETCH_SIZE x 2 + WRITE_SIZE) / dispatches -- FETCH_SIZE doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (it reports half of a wide coalesced read), both counters in KB,
each collected in a pass of its own -- next to the kernel's average launch duration from the kernel trace.
usage: make_traffic_json.py NAME KERNEL_SUBSTRING OUT.json "source note" """
import json
import sqlite3
import sys

name, kern, out, note = sys.argv[1:5]
root = f"gpurun_out/prof_{name}"


def q(db, sql, *a):
    return sqlite3.connect(db).execute(sql, a).fetchall()


like = f"%{kern}%"
n, tot_ms, avg_ms = q(f"{root}/t/t_results.db", "select count(*), sum(end-start)/1e6, avg(end-start)/1e6 from kernels where name like ?", like)[0]
fetch, nf = q(f"{root}/f/f_results.db", "select sum(value), count(*) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like ?", like)[0]
write, nw = q(f"{root}/w/w_results.db", "select sum(value), count(*) from counters_collection where counter_name='WRITE_SIZE' and kernel_name like ?", like)[0]
fetch_b, write_b = fetch * 1024 * 2, write * 1024
d = {"kernel": kern, "source": note, "dispatches": nf, "dispatches_trace": n, "fetch_size_kb_sum": fetch, "write_size_kb_sum": write,
     "fetch_bytes_corrected": fetch_b, "write_bytes": write_b, "traffic_bytes_per_launch": (fetch_b + write_b) / nf,
     "avg_launch_ms": avg_ms, "total_ms": tot_ms,
     "hbm_GBps_per_launch": (fetch_b + write_b) / nf / (avg_ms * 1e-3) / 1e9,
     "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); per launch = (fetch + write) / dispatches"}
json.dump(d, open(out, "w"), indent=1)
import os
os.makedirs("gpurun_out/profiles_out", exist_ok=True)
json.dump(d, open(os.path.join("gpurun_out/profiles_out", os.path.basename(out)), "w"), indent=1)
print(json.dumps(d))
