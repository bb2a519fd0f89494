run() { echo "== $1 handles=$2"; env $1 WFM_DEBUG=1 python scripts/c1_debug3.py $2 2> gpurun_out/c1d3.err | tr '\n' ' '; echo; grep -c "outside\|status" gpurun_out/c1d3.err; }
run "WFM_ALIGN_OWN_HANDLES=0 WFM_ALIGN_MIN_BATCHES=24 WFM_STREAMS=1" 1
run "WFM_X=1" 3
run "WFM_ALIGN_MIN_BATCHES=60" 1
