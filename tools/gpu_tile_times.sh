#!/bin/bash
# dec_unfilter_kernel's per-tile time stamps (build variant tile_timing) for one case of tools/decode_device_timing.py
TAG=${1:-t}; CASE=${2:-8K RGBA grad}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
FPNG_TIMING_NOCHECK=1 FPNG_AMD_TILE_TIMES=$O/${TAG}_tiles.txt FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_tile_timing.so timeout 300 python tools/decode_device_timing.py 2 "$CASE" > $O/${TAG}_tiles.log 2>&1
python tools/tile_times_summary.py $O/${TAG}_tiles.txt | tee $O/${TAG}_tiles_summary.txt
rm -f $O/${TAG}_tiles.txt
