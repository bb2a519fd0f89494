#!/usr/bin/env python
"""Issue-side figures of one kernel from a rocprofv3 SQ pass + a kernel trace of the same command:
  valu_frac = SQ_INSTS_VALU x the mean issue cycles of the kernel's instruction mix (MEAN_ISSUE_CYCLES below: measured,
              profiles/r6_valu_issue.md priced over the tile kernel's step by scripts/isa_hot_path.py; 4.2 for the pure max / compare kinds, 2.2 for
              add / logic) / the SIMD cycles the kernel's CUs were busy (SQ_BUSY_CU_CYCLES x 4 SIMDs when the pass holds it -- a counter, in
              cycles summed over the CUs -- else 1024 SIMDs x the kernel's total running time x 2.4 GHz).  gfx950 has no counter of
              vector-unit occupancy (SQ_ACTIVE_INST_VALU counts one quad-cycle per instruction whatever its kind): valu_frac stays derived.
  wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of its lifetime a wave spends waiting for anything)
usage: make_sq_json.py TRACE.db SQ.db KERNEL_SUBSTRING OUT.json "source note" """
import json
import sqlite3
import sys

trace, sq, kern, out, note = sys.argv[1:6]
MEAN_ISSUE_CYCLES = float(sys.argv[6]) if len(sys.argv) > 6 else 3.4
like = f"%{kern}%"
n, tot_ms = sqlite3.connect(trace).execute("select count(*), sum(end-start)/1e6 from kernels where name like ?", (like,)).fetchall()[0]
c = {}
for db in sq.split(","):  # (several passes: a pass holds at most eight SQ counters)
    for k_, v_ in sqlite3.connect(db).execute("select counter_name, sum(value) from counters_collection where kernel_name like ? group by counter_name", (like,)).fetchall():
        c.setdefault(k_, v_)
simd_cycles = c["SQ_BUSY_CU_CYCLES"] * 4 if c.get("SQ_BUSY_CU_CYCLES") else 1024 * tot_ms * 1e-3 * 2.4e9
d = {"kernel": kern, "source": note, "dispatches": n, "kernel_total_ms": tot_ms, "counters": c,
     "valu_frac": c.get("SQ_INSTS_VALU", 0) * MEAN_ISSUE_CYCLES / simd_cycles if simd_cycles else None,
     "mean_issue_cycles_per_valu_inst": MEAN_ISSUE_CYCLES, "simd_cycles": simd_cycles, "simd_cycles_from": "SQ_BUSY_CU_CYCLES x 4" if c.get("SQ_BUSY_CU_CYCLES") else "kernel time x 1024 SIMDs x 2.4 GHz",
     "salu_per_valu": c.get("SQ_INSTS_SALU", 0) / max(1, c.get("SQ_INSTS_VALU", 1)),
     "wait_frac": c.get("SQ_WAIT_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1)),
     "issue_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1)),
     "insts_per_wave": (c.get("SQ_INSTS_VALU", 0) + c.get("SQ_INSTS_SALU", 0) + c.get("SQ_INSTS_LDS", 0)) / max(1, c.get("SQ_WAVES", 1)),
     "note": "valu_frac = SQ_INSTS_VALU x mean issue cycles / busy SIMD cycles (derived: no occupancy counter on gfx950, profiles/r6_valu_issue.md); wait_frac and issue_frac are ratios of per-wave cycle counters (quad-cycles)"}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d))
