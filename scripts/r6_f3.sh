mkdir -p gpurun_out/r6w
timeout 2400 python -m pytest tests/test_map_paf_gpu.py tests/test_lpa_published_gpu.py tests/test_lpa_gpu.py tests/test_configs_gpu.py tests/test_map_l2_gpu.py tests/test_index_file_gpu.py -x -q -m gpu > gpurun_out/r6w/map_tests.log 2>&1; tail -4 gpurun_out/r6w/map_tests.log
for rep in 1 2; do
for setting in "WFM_X=1" "WFM_FILTER_DEVICE_ORDER=0"; do
  echo "==== [$setting] rep $rep"
  env $setting WFM_FILTER_TIMES=1 python scripts/c4_rank.py 2>&1 | grep -E "chain_mappings|filterSubset|ms_filter|map_s|\"map\"|post" | tail -6
done
done 2>&1 | tee gpurun_out/r6w/f3_ab.log
