R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
T() { env "$@" timeout 120 python tools/direct_timing.py $WL 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -1; }
for WL in "512x512x3 1024 0" "3840x2160x4 16 0" "1920x1080x3 256 0" "1920x1080x4 256 0" "256x256x3 2048 0"; do
  T FPNG_AMD_DIRECT=0
  for RL in 15 14 13; do T FPNG_AMD_ASSEMBLE_RL=$RL; done
  T FPNG_AMD_DIRECT=0
done 2>&1 | tee gpurun_out/r05_assemble_rl2.txt
