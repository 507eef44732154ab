#!/bin/bash
# Round 5: device-resident decode batches cut into groups whose kernels alternate between two streams (un-filter under the next group's sync / emit)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
F="grep -v ^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
timeout 400 python -m pytest tests/test_gpu_decode.py -m gpu -x -q 2>&1 | $F | tail -4 | tee $O/r05_decode_groups_tests.txt
for G in 1 2 3 4; do for CASE in "8K RGBA grad x 8" "photo 11 MP RGB x 8" "4K UI glyphs"; do
  echo -n "groups $G: "; FPNG_TIMING_NOCHECK=1 FPNG_AMD_DECODE_DEVICE_GROUPS=$G timeout 200 python tools/decode_device_timing.py 6 "$CASE" 2>&1 | grep "flags="; done; done | tee $O/r05_decode_groups.txt
