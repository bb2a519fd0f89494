#!/usr/bin/env python
"""Summarise rocprofv3 (rocpd sqlite) outputs into a markdown file under profiles/.

usage: prof_summary.py OUT.md TITLE trace.db [pmc1.db pmc2.db ...] [--bench bench.log]
FETCH_SIZE / WRITE_SIZE are reported in KB as rocprofv3 gives them; on gfx950 FETCH_SIZE
under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -- both the raw
and the corrected figure are printed.
"""
import sqlite3
import sys


def short(name):
    """kernel name without the argument list; anonymous-namespace kernels start with '(anonymous namespace)::'"""
    n = name.replace("(anonymous namespace)::", "").replace("wfm::", "")
    n = n.split("(")[0]
    if "rocprim" in n:  # rocPRIM trampolines: keep the config name only
        import re
        m = re.search(r"wrapped_(\w+?)_config", n)
        n = "rocprim " + (m.group(1) if m else n.split("::")[-1][:40])
    return n[:90]


def q(db, sql):
    return sqlite3.connect(db).execute(sql).fetchall()


def main():
    args = sys.argv[1:]
    bench = None
    if "--bench" in args:
        i = args.index("--bench")
        bench = args[i + 1]
        del args[i:i + 2]
    out, title, trace, pmcs = args[0], args[1], args[2], args[3:]
    L = [f"# {title}\n\n"]
    if bench:
        for line in open(bench):
            if line.startswith("{"):
                L.append("bench line:\n```\n" + line.strip() + "\n```\n\n")
    L.append("## rocprofv3 --kernel-trace --stats\n\n| kernel | calls | total ms | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|---|\n")
    rows = q(trace, "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6 from kernels group by name order by 3 desc")
    tot = sum(r[2] for r in rows) or 1.0
    for r in rows:
        L.append("| %s | %d | %.3f | %.4f | %.4f | %.4f | %.1f |\n" % (short(r[0]), r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
    if pmcs:
        L.append("\n## PMC passes (separate runs, sums over all dispatches of the kernel)\n\n| kernel | counter | sum | dispatches |\n|---|---|---|---|\n")
        for db in pmcs:
            for r in q(db, "select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name not like '%rocprim%' and kernel_name not like '__amd%' group by kernel_name, counter_name order by 1, 2"):
                name = short(r[0])
                L.append("| %s | %s | %.6g | %d |\n" % (name, r[1], r[2], r[3]))
                if r[1] == "FETCH_SIZE":
                    L.append("| %s | FETCH bytes (KB*1024; x2 gfx950 correction) | %.4g (%.4g) | |\n" % (name, r[2] * 1024, r[2] * 2048))
                if r[1] == "WRITE_SIZE":
                    L.append("| %s | WRITE bytes (KB*1024) | %.4g | |\n" % (name, r[2] * 1024))
    open(out, "w").write("".join(L))
    print("".join(L))


if __name__ == "__main__":
    main()
