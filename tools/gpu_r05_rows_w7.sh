#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
T() { env "$@" timeout 120 python tools/direct_timing.py $WL 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -1; }
L=$R/fpng_amd/lib/libfpng_amd_rows_w7.so
for WL in "1920x1080x3 256 0" "512x512x3 1024 0" "1920x1080x4 256 0" "1920x1080x3 256 1" "3840x2160x3 32 0"; do
  for rep in 1 2; do T FPNG_AMD_DIRECT=0; T FPNG_AMD_LIB=$L; done
done 2>&1 | tee gpurun_out/r05_rows_w7.txt
