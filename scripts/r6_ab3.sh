mkdir -p gpurun_out/r6t
timeout 1500 python -m pytest tests/test_align_gpu.py -x -q -m gpu > gpurun_out/r6t/align_tests.log 2>&1; tail -4 gpurun_out/r6t/align_tests.log
for rep in 1 2; do
for setting in "WFM_X=1" "WFM_TILE_RING3=0"; do
  echo "==== [$setting] rep $rep"
  env $setting python scripts/c3_time.py --reps 10 --warmup 3
  env $setting WFM_OVERLAP=0 python scripts/c3_time.py --reps 4 --warmup 2
  env $setting WFM_DEBUG=0 python scripts/legs_run.py c2 c4 --reps 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(' ', d['leg'], 'pass', d['pass'], 'align_s %.4f ms_gpu %.1f' % (d['align_s'], d['ms_gpu']))"
done
done 2>&1 | tee gpurun_out/r6t/ab3.log
