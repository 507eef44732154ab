#!/bin/bash
# A/B of diagnostic builds of the decoder on one box: kernel stats of the 8 x 8K step for each library given
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for V in "$@"; do
  L=$R/fpng_amd/lib/libfpng_amd$V.so
  echo "=== $L"
  FPNG_TIMING_NOCHECK=1 FPNG_AMD_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ab$V -o dec -- python $R/tools/decode_device_timing.py 6 "8K RGBA grad" 2>&1 | grep "flags="
  python $R/tools/prof_summary.py $O/ab$V dec_ 7 | grep "dec_"
done
