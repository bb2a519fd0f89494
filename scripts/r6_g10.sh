mkdir -p gpurun_out/r6j
python scripts/tile_trace2.py > gpurun_out/r6j/t2_default.txt 2>&1
tail -n 9 gpurun_out/r6j/t2_default.txt
