#!/usr/bin/env python
"""List the launches of the kernels whose name contains PATTERN in time order, from a rocprofv3 (rocpd sqlite) kernel trace:
start (ms after the first listed launch), duration, grid and workgroup size.  usage: prof_timeline.py trace.db PATTERN [max_rows]"""
import sqlite3
import sys


def main():
    db, pat = sys.argv[1], sys.argv[2]
    lim = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    want = [c for c in ("name", "start", "end", "grid_size", "workgroup_size", "grid_x", "workgroup_x", "grid_size_x", "workgroup_size_x", "lds_size", "vgpr_count",
                        "accum_vgpr_count", "sgpr_count", "scratch_size") if c in cols]
    print("# columns available:", ", ".join(cols))
    rows = con.execute(f"select {', '.join(want)} from kernels where name like ? order by start", (f"%{pat}%",)).fetchall()
    if not rows:
        print("no launches match")
        return
    t0 = rows[0][want.index("start")]
    print("| # | start ms | dur ms | " + " | ".join(w for w in want if w not in ("name", "start", "end")) + " | kernel |")
    for i, r in enumerate(rows[:lim]):
        d = dict(zip(want, r))
        rest = " | ".join(str(d[w]) for w in want if w not in ("name", "start", "end"))
        print(f"| {i} | {(d['start'] - t0) / 1e6:.3f} | {(d['end'] - d['start']) / 1e6:.4f} | {rest} | {d['name'].split('(')[0][-60:]} |")


if __name__ == "__main__":
    main()
