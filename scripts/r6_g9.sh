mkdir -p gpurun_out/r6i
timeout 1500 python -m pytest tests/test_align_gpu.py -x -q -m gpu > gpurun_out/r6i/align_tests.log 2>&1
tail -4 gpurun_out/r6i/align_tests.log
STEPS=10 bash scripts/ab_c3.sh "" "WFM_TILE_COARSE=0" "" > gpurun_out/r6i/ab.log 2>&1
cat gpurun_out/r6i/ab.log
python bench.py --config C5 --pairs 8 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('C5 ms/step', d['ms_per_step'], 'cigar', d.get('cigar_identical_rate'))"
