#!/bin/bash
# kernel trace of tools/decode_device_timing.py for one content, libraries given by suffix ("." = the tree's): per kernel, calls and times
#   usage (through gpurun): bash tools/gpu_decode_trace.sh <tag> "<lib suffixes>" "<case>"
TAG=${1:-t}; LIBS=${2:-"."}; CASE=${3:-8K RGBA grad}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
OUT=$O/${TAG}_trace.txt; : > $OUT
cd /tmp; export TMPDIR=/tmp
for L in $LIBS; do
  [ "$L" = "." ] && L=""
  echo "== libfpng_amd$L.so: $CASE" >> $OUT
  FPNG_TIMING_NOCHECK=1 FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd$L.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_tr$L -o dec -- python $R/tools/decode_device_timing.py 3 "$CASE" > /dev/null 2>&1
  python $R/tools/prof_summary.py $(dirname $(find $O/${TAG}_tr$L -name "*kernel_stats.csv" | head -1)) dec_ 12 | grep "dec_" >> $OUT
done
cat $OUT
