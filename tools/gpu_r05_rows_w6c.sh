R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
T() { env "$@" timeout 120 python tools/direct_timing.py $WL 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -1; }
L=$R/fpng_amd/lib/libfpng_amd_rows4_w8.so
timeout 280 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_kat or fuzz" 2>&1 | tail -1
for WL in "3072x1728x4 24 0" "3072x1728x4 24 1" "1920x1080x4 256 0" "7680x4320x4 8 0"; do
  T FPNG_AMD_LIB=$L; T FPNG_AMD_DIRECT=0; T FPNG_AMD_LIB=$L; T FPNG_AMD_DIRECT=0
done 2>&1 | tee gpurun_out/r05_rows_w6c.txt
# below the threshold the product uses eight waves: force six there to see where the crossover lies (a build with the threshold at 0)
