#!/bin/bash
# Round 5, encode_direct_kernel second session: after the round trips were taken out of the chunk's path -- parity, then where the time goes
# (timing-only builds: no look-back / no placement stores / no look behind / none of the three).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
F="grep -v ^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
timeout 280 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_kat or direct_placement" 2>&1 | $F | tail -8 | tee $O/r05_direct2_tests.txt
T() { env "$@" timeout 120 python tools/direct_timing.py $WL 2>&1 | $F | tail -1; }
L=$R/fpng_amd/lib
{ WL="7680x4320x4 8 0"
  T FPNG_AMD_DIRECT=0
  T FPNG_AMD_DIRECT=1
  T FPNG_AMD_PIECE_PX=1024
  T FPNG_AMD_PIECE_PX=2048 FPNG_AMD_LIB=$L/libfpng_amd_direct_win1280.so
  T FPNG_AMD_LIB=$L/libfpng_amd_abl_nolook.so
  T FPNG_AMD_LIB=$L/libfpng_amd_abl_nostore.so
  T FPNG_AMD_LIB=$L/libfpng_amd_abl_nobehind.so
  T FPNG_AMD_LIB=$L/libfpng_amd_abl_all.so
  T FPNG_AMD_LIB=$L/libfpng_amd_abl_all.so FPNG_AMD_PIECE_PX=7680
  T FPNG_AMD_LANES=1
  WL="1920x1080x3 256 0"
  T FPNG_AMD_DIRECT=0
  T FPNG_AMD_DIRECT=1
  T FPNG_AMD_LIB=$L/libfpng_amd_abl_all.so
  WL="512x512x3 1024 0"
  T FPNG_AMD_DIRECT=0
  T FPNG_AMD_DIRECT=1
  T FPNG_AMD_LIB=$L/libfpng_amd_abl_all.so
  WL="7680x4320x4 8 1"
  T FPNG_AMD_DIRECT=0
  T FPNG_AMD_DIRECT=1
} 2>&1 | tee $O/r05_direct2_timing.txt
