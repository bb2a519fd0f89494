mkdir -p gpurun_out/r6s
timeout 1500 python -m pytest tests/test_bench_launch.py -x -q -m gpu > gpurun_out/r6s/launch_tests.log 2>&1; tail -5 gpurun_out/r6s/launch_tests.log
timeout 900 python bench.py --config C4 --c4-mbp 8 --steps 2 --warmup 1 > gpurun_out/r6s/c4_8mbp.json 2> gpurun_out/r6s/c4_8mbp.err; tail -c 1800 gpurun_out/r6s/c4_8mbp.json; tail -3 gpurun_out/r6s/c4_8mbp.err
timeout 1500 python bench.py > gpurun_out/r6s/bench.json 2> gpurun_out/r6s/bench.err; python scripts/bench_fields.py gpurun_out/r6s/bench.json; tail -3 gpurun_out/r6s/bench.err
