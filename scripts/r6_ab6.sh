mkdir -p gpurun_out/r6z
timeout 1500 python -m pytest tests/test_align_gpu.py tests/test_align_paf_gpu.py tests/test_ref_wflign_gpu.py tests/test_lpa_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/r6z/tests6.log 2>&1; tail -3 gpurun_out/r6z/tests6.log
timeout 1500 python -m pytest tests/test_map_paf_gpu.py tests/test_bench_launch.py -x -q -m gpu > gpurun_out/r6z/tests6b.log 2>&1; tail -3 gpurun_out/r6z/tests6b.log
for rep in 1 2; do
for setting in "WFM_X=1" "WFM_LIB=libwfmash_hip_base.so"; do
  echo "==== [$setting] rep $rep"
  env $setting python scripts/c3_time.py --reps 6 --warmup 2
  env $setting WFM_DEBUG=0 python scripts/legs_run.py c2 --reps 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(' ', d['leg'], 'pass', d['pass'], 'align_s %.4f ms_gpu %.1f' % (d['align_s'], d['ms_gpu']))"
done
done 2>&1 | tee gpurun_out/r6z/ab6.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pc2 && rocprofv3 --kernel-trace --stats -d /tmp/pc2 -o t -- python $GRAFT_REPO_ROOT/scripts/legs_run.py c2 --reps 2 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python scripts/prof_summary.py gpurun_out/r6z/c2_after2.md "r6: C2 after the walk's mismatch runs" "$(find /tmp/pc2 -name '*results.db' | head -1)" > /dev/null; sed -n 5,14p gpurun_out/r6z/c2_after2.md | cut -c1-150
for s in "WFM_X=1" "WFM_FILTER_OVERLAP=0"; do echo "== [$s] all-vs-all map, 40 Mbp x 8"; env $s python scripts/c4_node.py --mbp 40 --gpus 1 2>&1 | tail -2 | cut -c1-600; done
