#!/bin/bash
# Builds wfmash_amd/libwfmash_hip_trace.so: the library with wfa_tile2.hip compiled -DWFM_TILE_TRACE=${TRACE_LEVEL:-1} (s_memtime stamps of a step's phases; never the
# shipped library).  Run here (hipcc cross-compiles); scripts/tile_trace.py runs it on the GPU box.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
python $root/__graft_entry__.py > /dev/null
mkdir -p $root/build/trace
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 -mllvm -unroll-threshold=1000000 -DWFM_TILE_TRACE=${TRACE_LEVEL:-1} -x hip -c $root/wfmash_amd/csrc/wfa_tile2.hip -o $root/build/trace/wfa_tile2.hip.o
objs=$(ls $root/build/obj/*.o | grep -v "wfa_tile2.hip.o$")
hipcc --offload-arch=gfx950 -fPIC -shared -o $root/wfmash_amd/libwfmash_hip_trace.so $objs $root/build/trace/wfa_tile2.hip.o -lz
ls -la $root/wfmash_amd/libwfmash_hip_trace.so
