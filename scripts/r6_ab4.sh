mkdir -p gpurun_out/r6y
timeout 1500 python -m pytest tests/test_align_gpu.py -x -q -m gpu > gpurun_out/r6y/align_tests.log 2>&1; tail -3 gpurun_out/r6y/align_tests.log
for rep in 1 2; do
for setting in "WFM_X=1" "WFM_LIB=libwfmash_hip_base.so"; do
  echo "==== [$setting] rep $rep"
  env $setting python scripts/c3_time.py --reps 8 --warmup 3
  env $setting WFM_DEBUG=0 python scripts/legs_run.py c2 c4 --reps 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(' ', d['leg'], 'pass', d['pass'], 'align_s %.4f ms_gpu %.1f' % (d['align_s'], d['ms_gpu']))"
  env $setting python scripts/c1_run.py --reps 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  C1 pass', d['pass'], 'align_s %.3f ms_gpu %d' % (d['align_s'], d['ms_gpu']))"
done
done 2>&1 | tee gpurun_out/r6y/ab4.log
