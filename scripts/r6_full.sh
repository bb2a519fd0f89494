mkdir -p gpurun_out/r6p
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r6p/gpu_tests.log 2>&1
tail -4 gpurun_out/r6p/gpu_tests.log
timeout 1200 python bench.py > gpurun_out/r6p/bench.json 2> gpurun_out/r6p/bench.err
python scripts/bench_fields.py gpurun_out/r6p/bench.json 2>/dev/null | head -40
