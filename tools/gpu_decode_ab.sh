#!/bin/bash
# A/B of builds of the decoder on one box: kernel stats of one workload's step for each library given
#   tools/gpu_decode_ab.sh "<case name of tools/decode_device_timing.py>" <lib suffix> ...      ("" = the product, _base, _emit_nt ...)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
CASE=$1; shift
cd /tmp; export TMPDIR=/tmp
for V in "$@"; do
  [ "$V" = "-" ] && V=""
  L=$R/fpng_amd/lib/libfpng_amd$V.so
  echo "=== $L: $CASE"
  FPNG_TIMING_NOCHECK=1 FPNG_AMD_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ab$V -o dec -- python $R/tools/decode_device_timing.py 6 "$CASE" 2>&1 | grep "flags="
  python $R/tools/prof_summary.py $O/ab$V dec_ 7 | grep "dec_"
done
