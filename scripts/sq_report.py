#!/usr/bin/env python
"""Per-kernel table out of one rocprofv3 kernel trace and two SQ counter passes of the same command:
launches, total / average / maximum duration, waves, instructions per wave (VALU / SALU / LDS / VMEM), share of the wave
cycles spent waiting, VALU busy share of the kernel's own running time.
usage: sq_report.py OUT.md "title" TRACE.db SQ_A.db SQ_B.db [bench log]"""
import sqlite3
import sys

out, title, trace, sqa, sqb = sys.argv[1:6]
log = sys.argv[6] if len(sys.argv) > 6 else None


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("wfm::", "").replace("void ", "")
    return (n[:n.index("(")] if "(" in n else n) or name


kern = {}
for name, n, tot, avg, mx in sqlite3.connect(trace).execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, max(end-start)/1e6 from kernels group by name"):
    kern[short(name)] = dict(n=n, tot=tot, avg=avg, mx=mx)
ctr = {}
for db in (sqa, sqb):
    for name, c, v in sqlite3.connect(db).execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"):
        ctr.setdefault(short(name), {})[c] = v
all_ms = sum(k["tot"] for k in kern.values())
with open(out, "w") as f:
    f.write(f"# {title}\n\n")
    f.write(f"Sum of kernel time {all_ms:.1f} ms.  wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES; issue = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; valu_busy = SQ_INSTS_VALU x 3.4 cycles (the measured mean of the tile kernel's mix, "
            "profiles/r6_valu_issue.md; 4.2 for a pure max / compare stream) / (1024 SIMDs x kernel time x 2.4 GHz).\n\n")
    f.write("| kernel | launches | total ms | share | avg ms | max ms | waves | VALU/wave | SALU/wave | LDS/wave | VMEM/wave | wait | issue | valu_busy |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for k, d in sorted(kern.items(), key=lambda kv: -kv[1]["tot"]):
        if d["tot"] < 0.002 * all_ms:
            continue
        c = ctr.get(k, {})
        w = max(1.0, c.get("SQ_WAVES", 0))
        wc = max(1.0, c.get("SQ_WAVE_CYCLES", 0))
        vb = c.get("SQ_INSTS_VALU", 0) * 3.4 / (1024 * d["tot"] * 1e-3 * 2.4e9) if d["tot"] else 0
        f.write(f"| `{k[-90:]}` | {d['n']} | {d['tot']:.2f} | {d['tot'] / all_ms:.3f} | {d['avg']:.4f} | {d['mx']:.3f} | {w:.0f} | {c.get('SQ_INSTS_VALU', 0) / w:.0f} | "
                f"{c.get('SQ_INSTS_SALU', 0) / w:.0f} | {c.get('SQ_INSTS_LDS', 0) / w:.0f} | {(c.get('SQ_INSTS_VMEM_RD', 0) + c.get('SQ_INSTS_VMEM_WR', 0)) / w:.0f} | "
                f"{c.get('SQ_WAIT_ANY', 0) / wc:.2f} | {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | {vb:.3f} |\n")
    if log:
        f.write("\n## the run's own line\n```\n")
        for line in open(log):
            if line.startswith("{"):
                f.write(line[:1500])
        f.write("```\n")
print(open(out).read())
