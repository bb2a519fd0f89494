mkdir -p gpurun_out/r6a
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue scripts/micro/valu_issue.hip 2>/dev/null && timeout 300 /tmp/valu_issue > gpurun_out/r6a/valu_issue.md 2> gpurun_out/r6a/valu_issue.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "^\s*(Name|.*SQ_(ACTIVE|INST_CYCLES|BUSY|VALU|INSTS_VALU|THREAD_CYCLES|WAIT_INST|INST_LEVEL))" | head -80 > $GRAFT_REPO_ROOT/gpurun_out/r6a/counters.txt 2>&1
rocprofv3 --list-avail 2>/dev/null > $GRAFT_REPO_ROOT/gpurun_out/r6a/list_avail.txt
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
tail -c 3000 gpurun_out/r6a/bench.json | head -c 1500
cat gpurun_out/r6a/valu_issue.md
