mkdir -p gpurun_out/r6w
timeout 2400 python -m pytest tests/test_map_paf_gpu.py tests/test_lpa_published_gpu.py tests/test_lpa_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/r6w/map_tests2.log 2>&1; tail -3 gpurun_out/r6w/map_tests2.log
for rep in 1 2; do
for setting in "WFM_X=1" "WFM_FILTER_DEVICE_ORDER=0"; do
  echo "==== [$setting] rep $rep"
  env $setting WFM_FILTER_TIMES=1 python scripts/c4_rank.py 2>&1 | grep -E "chain_mappings|filterSubset|ms_filter" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  ms_identity %d ms_index %d ms_map %d ms_filter %d wall %.2f' % (d['ms_identity'], d['ms_index'], d['ms_map'], d['ms_filter'], d['wall_s']))
    else: print(' ', l.strip()[:160])"
done
done 2>&1 | tee gpurun_out/r6w/f3_ab2.log
