mkdir -p gpurun_out/r6x gpurun_out/r6z
timeout 1500 python -m pytest tests/test_align_gpu.py -x -q -m gpu > gpurun_out/r6z/tests7.log 2>&1; tail -2 gpurun_out/r6z/tests7.log
for rep in 1 2 3; do
for setting in "WFM_X=1" "WFM_LIB=libwfmash_hip_base.so"; do
  echo "==== [$setting] rep $rep"
  env $setting python scripts/c3_time.py --reps 10 --warmup 3
done
done 2>&1 | tee gpurun_out/r6z/ab7.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6x/bench_driver_style.json 2> gpurun_out/r6x/bench_driver_style.err
python scripts/bench_fields.py gpurun_out/r6x/bench_driver_style.json | head -3
