#!/usr/bin/env python
"""Issue-side figures of one kernel from a rocprofv3 SQ pass + a kernel trace of the same command:
  valu_frac = SQ_INSTS_VALU x 4 cycles (a wave64 VALU instruction occupies its 16-lane SIMD for 4 cycles) / (1024 SIMDs x the
              kernel's total running time x 2.4 GHz)
  wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of its lifetime a wave spends waiting for anything)
usage: make_sq_json.py TRACE.db SQ.db KERNEL_SUBSTRING OUT.json "source note" """
import json
import sqlite3
import sys

trace, sq, kern, out, note = sys.argv[1:6]
like = f"%{kern}%"
n, tot_ms = sqlite3.connect(trace).execute("select count(*), sum(end-start)/1e6 from kernels where name like ?", (like,)).fetchall()[0]
c = dict(sqlite3.connect(sq).execute("select counter_name, sum(value) from counters_collection where kernel_name like ? group by counter_name", (like,)).fetchall())
simd_cycles = 1024 * tot_ms * 1e-3 * 2.4e9
d = {"kernel": kern, "source": note, "dispatches": n, "kernel_total_ms": tot_ms, "counters": c,
     "valu_frac": c.get("SQ_INSTS_VALU", 0) * 4 / simd_cycles if simd_cycles else None,
     "salu_per_valu": c.get("SQ_INSTS_SALU", 0) / max(1, c.get("SQ_INSTS_VALU", 1)),
     "wait_frac": c.get("SQ_WAIT_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1)),
     "issue_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1)),
     "insts_per_wave": (c.get("SQ_INSTS_VALU", 0) + c.get("SQ_INSTS_SALU", 0) + c.get("SQ_INSTS_LDS", 0)) / max(1, c.get("SQ_WAVES", 1)),
     "note": "valu_frac: 256 CUs x 4 SIMDs at 2.4 GHz, 4 cycles per wave64 VALU instruction; wait_frac and issue_frac are ratios of per-wave cycle counters"}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d))
