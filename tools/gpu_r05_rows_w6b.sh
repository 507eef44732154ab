#!/bin/bash
# Round 5: the 4-channel row kernel at six waves per SIMD (FPNG_ROWS_WPE4=6, now the product) against eight (build variant rows4_w8), same box, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
T() { env "$@" timeout 120 python tools/direct_timing.py $WL 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -1; }
L=$R/fpng_amd/lib/libfpng_amd_rows4_w8.so
timeout 280 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_kat or fuzz or flip_point" 2>&1 | tail -2
for WL in "7680x4320x4 8 0" "7680x4320x4 8 1" "3840x2160x4 16 0" "3840x2160x4 16 1" "1920x1080x4 256 0" "7680x4320x4 1 0" "3840x2160x4 1 0"; do
  T FPNG_AMD_LIB=$L; T FPNG_AMD_DIRECT=0; T FPNG_AMD_LIB=$L; T FPNG_AMD_DIRECT=0
done 2>&1 | tee gpurun_out/r05_rows_w6b.txt
