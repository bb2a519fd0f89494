for w in 4 6 4 6; do WFM_ALIGN_WORKERS=$w python bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs C4_rank_40mbp,C4_rank_full,C4_all_vs_all 2>/dev/null | W=$w python -c '
import sys,json,os
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); s=d["secondary"]
        a=s["C4_all_vs_all"]; r=s["C4_rank_full"]; m=s["C4_rank_40mbp"]
        print("workers", os.environ["W"], "all: align %.2f gpu %.0f map %.2f | rank: align %.3f second %.3f gpu %.0f | 40mbp: %.3f second %.3f" % (a["align_s"], a["ms_gpu"], a["map_s"], r["align_s"], r["second_pass"]["align_s"], r["ms_gpu"], m["align_s"], m["second_pass"]["align_s"]))
'; done
