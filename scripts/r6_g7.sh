mkdir -p gpurun_out/r6g
python scripts/tile_trace.py > gpurun_out/r6g/trace_default.txt 2>&1
WFM_TILE_LDS_PAD=40000 python scripts/tile_trace.py > gpurun_out/r6g/trace_pad40k.txt 2>&1
WFM_TILE_LDS_PAD=60000 python scripts/tile_trace.py > gpurun_out/r6g/trace_pad60k.txt 2>&1
for f in gpurun_out/r6g/*.txt; do echo "== $f"; tail -n 7 $f; done
