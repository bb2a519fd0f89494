mkdir -p gpurun_out/r6h
python scripts/tile_trace2.py > gpurun_out/r6h/t2_default.txt 2>&1
WFM_TILE_COARSE=0 python scripts/tile_trace2.py > gpurun_out/r6h/t2_fine.txt 2>&1
WFM_TILE_LDS_PAD=60000 python scripts/tile_trace2.py > gpurun_out/r6h/t2_pad60k.txt 2>&1
WFM_TILE_THREADS=256 python scripts/tile_trace2.py > gpurun_out/r6h/t2_256.txt 2>&1
for f in gpurun_out/r6h/*.txt; do echo "== $f"; tail -n 6 $f; done
