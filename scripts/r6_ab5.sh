mkdir -p gpurun_out/r6z
timeout 1500 python -m pytest tests/test_align_gpu.py tests/test_align_paf_gpu.py tests/test_ref_wflign_gpu.py -x -q -m gpu > gpurun_out/r6z/align_tests.log 2>&1; tail -3 gpurun_out/r6z/align_tests.log
timeout 1500 python -m pytest tests/test_lpa_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/r6z/cfg_tests.log 2>&1; tail -3 gpurun_out/r6z/cfg_tests.log
for rep in 1 2; do
for setting in "WFM_X=1" "WFM_LIB=libwfmash_hip_base.so"; do
  echo "==== [$setting] rep $rep"
  env $setting WFM_DEBUG=0 python scripts/legs_run.py c2 --reps 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(' ', d['leg'], 'pass', d['pass'], 'align_s %.4f ms_gpu %.1f' % (d['align_s'], d['ms_gpu']))"
done
done 2>&1 | tee gpurun_out/r6z/ab5.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/pc2 -o t -- python $GRAFT_REPO_ROOT/scripts/legs_run.py c2 --reps 2 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python scripts/prof_summary.py gpurun_out/r6z/c2_after.md "r6: C2 after the wide ring kernel change" "$(find /tmp/pc2 -name '*results.db' | head -1)" > /dev/null; sed -n 1,16p gpurun_out/r6z/c2_after.md | cut -c1-150
