bash scripts/profile_r6.sh > gpurun_out/profile_r6.log 2>&1
tail -12 gpurun_out/profile_r6.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6x/bench_driver_style.json 2> gpurun_out/r6x/bench_driver_style.err
python scripts/bench_fields.py gpurun_out/r6x/bench_driver_style.json | head -4
