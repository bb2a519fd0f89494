mkdir -p gpurun_out/r6u
for setting in "WFM_X=1" "WFM_TILE_T=80" "WFM_TILE_T=120" "WFM_TILE_T=150" "WFM_TILE_T=200" "WFM_TILE_THREADS=1024" "WFM_TILE_THREADS=1024 WFM_TILE_T=150" "WFM_TILE_THREADS=256" "WFM_STREAMS=2" "WFM_STREAMS=4" "WFM_TILE_CHUNK=3" "WFM_TILE_CHUNK=4" "WFM_X=2"; do
  echo "==== [$setting]"
  env $setting python scripts/c3_time.py --reps 8 --warmup 3
done 2>&1 | tee gpurun_out/r6u/sweep.log
