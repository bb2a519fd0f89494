mkdir -p gpurun_out/r6k
for e in 0 1 2 8 16 3; do
  TRACE_LIB=libwfmash_hip_trace_E$e.so timeout 150 python scripts/tile_trace2.py > gpurun_out/r6k/exp_$e.txt 2>&1
  echo "== EXP $e"; grep -E "statuses|cycles per step|step loop|waves by" gpurun_out/r6k/exp_$e.txt
done
