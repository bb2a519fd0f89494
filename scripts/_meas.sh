WFM_DEBUG=1 python scripts/c4_rank.py --mbp 40 --align 2>/tmp/e.log >/dev/null
grep "phase 2 from rows\|tiled\|bp jobs\|base\|align_batch" /tmp/e.log | head -150 | cut -c1-220
