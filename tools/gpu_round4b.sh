#!/bin/bash
# round 4, second half: same-box A/B of the decoder against the tree in front of a change (libfpng_amd_base.so, built from the commit
# in front: git stash; python -c "from fpng_amd import build; build.build_variant('base', [])"; git stash pop) + the decoder's GPU tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -s 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 | tee $O/r4b_tests.txt
for rep in 1 2; do
  for lib in libfpng_amd_base.so libfpng_amd.so; do
    for c in "8K RGBA grad" "photo 11 MP" "8K RGBA blocks" "8K RGBA stripes" "4K RGBA grad" "1080p RGB grad" "4K UI"; do
      echo "== $lib"; FPNG_TIMING_PHASES=1 FPNG_AMD_LIB=$R/fpng_amd/lib/$lib timeout 300 python tools/decode_device_timing.py 6 "$c" 2>&1 | grep "flags="
    done
  done
done | tee $O/r4b_decode_ab.txt
