mkdir -p gpurun_out/r6c
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue scripts/micro/valu_issue.hip 2>/dev/null && timeout 300 /tmp/valu_issue > gpurun_out/r6c/valu_issue.md 2> gpurun_out/r6c/valu_issue.err
grep -i "pair\|cndmask" gpurun_out/r6c/valu_issue.md
