mkdir -p gpurun_out/${OUT:-r6x}
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/${OUT:-r6x}/gpu_tests.log 2>&1
tail -3 gpurun_out/${OUT:-r6x}/gpu_tests.log
for i in 1 2 3; do
timeout 1200 python bench.py > gpurun_out/${OUT:-r6x}/bench$i.json 2> gpurun_out/${OUT:-r6x}/bench$i.err
python scripts/bench_fields.py gpurun_out/${OUT:-r6x}/bench$i.json
done
