mkdir -p gpurun_out/r6e
python scripts/tile_trace.py > gpurun_out/r6e/trace_default.txt 2>&1
WFM_TILE_COARSE=0 python scripts/tile_trace.py > gpurun_out/r6e/trace_fine.txt 2>&1
WFM_TILE_THREADS=256 python scripts/tile_trace.py > gpurun_out/r6e/trace_256.txt 2>&1
PAIRS=8 python scripts/tile_trace.py > gpurun_out/r6e/trace_8pairs.txt 2>&1
tail -8 gpurun_out/r6e/*.txt
