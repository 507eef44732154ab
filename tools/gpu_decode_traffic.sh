#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the decoder's kernels for one case of tools/decode_device_timing.py (separate --pmc passes)
TAG=${1:-t}; CASE=${2:-8K RGBA grad}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/traffic_$TAG; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  FPNG_TIMING_NOCHECK=1 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o sq -- python $R/tools/decode_device_timing.py 2 "$CASE" > $OUT/run_$C.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT "(dec_[a-z_]+(?:<[a-z]+>)?)" | grep -A2 "^dec_"
