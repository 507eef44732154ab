#!/bin/bash
# dec_sync_kernel<false>'s per-workgroup time stamps (build variant sync_timing) for one case of tools/decode_device_timing.py
TAG=${1:-t}; CASE=${2:-8K RGBA grad}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
FPNG_TIMING_NOCHECK=1 FPNG_AMD_SYNC_TIMES=$O/${TAG}_sync.txt FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_sync_timing.so timeout 300 python tools/decode_device_timing.py 2 "$CASE" > $O/${TAG}_sync.log 2>&1
python tools/sync_times_summary.py $O/${TAG}_sync.txt | tee $O/${TAG}_sync_summary.txt
rm -f $O/${TAG}_sync.txt
