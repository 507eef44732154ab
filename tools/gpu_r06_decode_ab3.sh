#!/bin/bash
# round 6: the current decoder against earlier builds kept as fpng_amd/lib/libfpng_amd_<name>.so, one box, alternating.
#   usage (through gpurun): bash tools/gpu_r06_decode_ab3.sh <tag> "<lib suffixes, '' = current>" [case substring ...]
TAG=${1:-ab}; LIBS=${2:-"_r05dec _r06q ."}; shift; shift; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
OUT=$O/${TAG}_decode_ab.txt; : > $OUT
timeout 900 python -m pytest tests/test_gpu_decode.py -x -q > $O/${TAG}_decode_tests.txt 2>&1; echo "pytest rc=$?" >> $O/${TAG}_decode_tests.txt
tail -4 $O/${TAG}_decode_tests.txt >> $OUT
CASES=("$@"); [ ${#CASES[@]} -eq 0 ] && CASES=("8K RGBA grad" "photo" "8K RGBA solid" "1080p RGB grad" "4K UI glyphs" "4K UI dither" "8K RGBA stripes" "512x512")
for rep in 1 2; do
  for C in "${CASES[@]}"; do
    for L in $LIBS; do
      [ "$L" = "." ] && L=""
      echo "== rep $rep lib libfpng_amd$L.so" >> $OUT
      FPNG_TIMING_PHASES=1 FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd$L.so timeout 300 python tools/decode_device_timing.py 8 "$C" 2>&1 | grep "flags=" | cut -c1-220 >> $OUT
    done
  done
done
cd /tmp; export TMPDIR=/tmp
echo "== kernel trace, current: ${CASES[0]}" >> $OUT
FPNG_TIMING_NOCHECK=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o dec -- python $R/tools/decode_device_timing.py 6 "${CASES[0]}" > /dev/null 2>&1
python $R/tools/prof_summary.py $(dirname $(find $O/${TAG}_trace -name "*kernel_stats.csv" | head -1)) dec_ 9 | grep "dec_" >> $OUT
cat $OUT
