#!/usr/bin/env python
"""How many kernels run side by side, from a rocprofv3 (rocpd sqlite) kernel trace: over the window of the LAST burst of launches (gaps under
--gap ms belong to a burst), the time with 0 / 1 / 2 / 3+ kernels in flight, and per kernel name the time during which it ran ALONE (what no other
chain covered).  usage: prof_concurrency.py trace.db [--gap 20] [--pattern wfa_]"""
import argparse
import collections
import sqlite3


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("wfm::", "").split("(")[0][:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--gap", type=float, default=20.0)
    ap.add_argument("--pattern", default="")
    ap.add_argument("--burst", type=int, default=-1, help="which burst (default: the last)")
    a = ap.parse_args()
    rows = sqlite3.connect(a.db).execute("select name, start, end from kernels order by start").fetchall()
    rows = [r for r in rows if a.pattern in r[0]]
    bursts, cur = [], [rows[0]]
    for r in rows[1:]:
        if (r[1] - max(x[2] for x in cur[-64:])) / 1e6 > a.gap:
            bursts.append(cur)
            cur = []
        cur.append(r)
    bursts.append(cur)
    print(f"{len(rows)} launches in {len(bursts)} bursts: " + ", ".join(f"{len(b)} ({(max(x[2] for x in b) - b[0][1]) / 1e6:.1f} ms)" for b in bursts))
    b = bursts[a.burst]
    ev = []
    for n, s, e in b:
        ev.append((s, 1, n))
        ev.append((e, -1, n))
    ev.sort()
    live = collections.Counter()
    depth_ms = collections.Counter()
    alone = collections.Counter()
    t_prev = ev[0][0]
    for t, d, n in ev:
        k = sum(live.values())
        dt = (t - t_prev) / 1e6
        depth_ms[min(k, 4)] += dt
        if k == 1:
            alone[short(next(x for x in live if live[x] > 0))] += dt
        live[n] += d
        t_prev = t
    # the longest stretches without any kernel: where, after what, before what
    gaps = []
    ends = sorted(b, key=lambda r: r[1])
    cur_end, last_name = ends[0][2], ends[0][0]
    for n, s0, e0 in ends[1:]:
        if s0 > cur_end:
            gaps.append(((s0 - cur_end) / 1e6, (cur_end - b[0][1]) / 1e6, short(last_name), short(n)))
        if e0 > cur_end:
            cur_end, last_name = e0, n
    tot = sum(depth_ms.values())
    print(f"burst of {len(b)} launches, {tot:.2f} ms from its first launch to its last end")
    for k in sorted(depth_ms):
        print(f"  {k}{'+' if k == 4 else ''} kernels in flight: {depth_ms[k]:8.2f} ms  ({100 * depth_ms[k] / tot:.0f} %)")
    print("  longest stretches without a kernel (ms, at ms, after, before):")
    for g in sorted(gaps, reverse=True)[:14]:
        print(f"    {g[0]:6.2f} at {g[1]:7.2f}  after {g[2][:44]:44s} before {g[3][:44]}")
    print(f"    ({len(gaps)} stretches, {sum(g[0] for g in gaps):.2f} ms in all; {sum(g[0] for g in gaps if g[0] < 0.05):.2f} ms of it in stretches under 50 us)")
    print("  alone on the device, by kernel:")
    for n, ms in alone.most_common(12):
        print(f"    {ms:8.2f} ms  {n}")


if __name__ == "__main__":
    main()
