#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
T() { env "$@" timeout 120 python tools/direct_timing.py $WL 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -1; }
L=$R/fpng_amd/lib/libfpng_amd
for WL in "7680x4320x4 8 0" "7680x4320x4 8 1" "1920x1080x3 256 0"; do
  for rep in 1 2; do T FPNG_AMD_LIB=${L}_rows4_w8.so; T FPNG_AMD_DIRECT=0; T FPNG_AMD_LIB=${L}_rows4_w6.so; T FPNG_AMD_LIB=${L}_asm_w4.so; T FPNG_AMD_LIB=${L}_asm_w4_rows6.so; T FPNG_AMD_LIB=${L}_asm_w2_rows6.so; done
done 2>&1 | tee gpurun_out/r05_rows_w6e.txt
