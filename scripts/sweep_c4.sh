#!/bin/bash
# env-variable sweeps on the scaled C4 rank (exclusive mode: one stream, so kernel times are not stretched by sharing);
# prints per config the warm pass's align wall time, device-busy time and the library's own phase timings
# usage: scripts/sweep_c4.sh OUTDIR "VAR=a VAR2=b" "VAR=c" ...
out=$1; shift
mkdir -p "$out"
i=0
for cfg in "$@"; do
  i=$((i+1))
  env WFM_OVERLAP=${WFM_OVERLAP:-0} $cfg timeout 200 python scripts/legs_debug.py c4 --reps 2 > "$out/s$i.json" 2> "$out/s$i.err"
  python - "$out/s$i.json" "$out/s$i.err" "$cfg" <<'PY'
import json, sys, re
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
d = rows[-1] if rows else {}
err = open(sys.argv[2]).read()
seg = err[err.rfind("pass 1"):]
walls = re.findall(r"wall: total ([\d.]+) ms \| levels ([\d.]+) \(tile phase ([\d.]+) incl\. kernels ([\d.]+); base phase ([\d.]+) incl\. kernels ([\d.]+); bp kernels ([\d.]+)\)", seg)
main = walls[0] if walls else None
print(f"{sys.argv[3]:45s} align_s {d.get('align_s', 0):.3f} gpu_ms {d.get('ms_gpu', 0):.1f} | main call: total {main[0] if main else '?'} tile-kernels {main[3] if main else '?'} tile-wall {main[2] if main else '?'} p2+bp {main[6] if main else '?'} base {main[5] if main else '?'}")
PY
done
