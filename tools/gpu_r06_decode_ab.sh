#!/bin/bash
# round 6: the decode-once decoder against round 5's (libfpng_amd_r05dec.so = the tree before the decoder changed), one box, alternating.
#   usage (through gpurun): bash tools/gpu_r06_decode_ab.sh <tag> [case substring ...]
TAG=${1:-ab}; shift; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
OUT=$O/${TAG}_decode_ab.txt; : > $OUT
CASES=("$@"); [ ${#CASES[@]} -eq 0 ] && CASES=("8K RGBA grad" "photo" "8K RGBA solid" "1080p RGB grad" "4K UI glyphs")
for rep in 1 2; do
  for C in "${CASES[@]}"; do
    for L in _r05dec ""; do
      echo "== rep $rep lib libfpng_amd$L.so" >> $OUT
      FPNG_TIMING_PHASES=1 FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd$L.so timeout 300 python tools/decode_device_timing.py 8 "$C" 2>&1 | grep "flags=" | cut -c1-220 >> $OUT
    done
  done
done
cat $OUT
# kernel trace of the first case, both libraries
cd /tmp; export TMPDIR=/tmp
for L in _r05dec ""; do
  echo "== kernel trace, libfpng_amd$L.so: ${CASES[0]}" >> $OUT
  FPNG_TIMING_NOCHECK=1 FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd$L.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace$L -o dec -- python $R/tools/decode_device_timing.py 6 "${CASES[0]}" > /dev/null 2>&1
  python $R/tools/prof_summary.py $(dirname $(find $O/${TAG}_trace$L -name "*kernel_stats.csv" | head -1)) dec_ 9 | grep "dec_" >> $OUT
done
tail -40 $OUT
