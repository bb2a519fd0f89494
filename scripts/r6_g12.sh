mkdir -p gpurun_out/r6l
timeout 1500 python -m pytest tests/test_align_gpu.py -x -q -m gpu > gpurun_out/r6l/align_tests.log 2>&1
tail -4 gpurun_out/r6l/align_tests.log
STEPS=10 bash scripts/ab_c3.sh "" "WFM_TILE_COARSE=0" "" > gpurun_out/r6l/ab.log 2>&1
cat gpurun_out/r6l/ab.log
