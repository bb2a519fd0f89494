mkdir -p gpurun_out/r6v
python scripts/tile_trace2.py > gpurun_out/r6v/t2.txt 2>&1; tail -n 9 gpurun_out/r6v/t2.txt
WFM_TILE_LDS_PAD=60000 python scripts/tile_trace2.py > gpurun_out/r6v/t2_pad.txt 2>&1; tail -n 9 gpurun_out/r6v/t2_pad.txt
