#!/bin/bash
# Round 5: tools/probes/write_shapes says a workgroup per 4 KiB of consecutive memory writes at 6.8 TB/s, per 64 KiB at 5.7 -- assemble_kernel owns 64 KiB per workgroup on big batches.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
T() { env "$@" timeout 120 python tools/direct_timing.py $WL 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -1; }
{ for WL in "7680x4320x4 8 0" "1920x1080x3 256 0" "7680x4320x4 8 1"; do
  T FPNG_AMD_DIRECT=0
  for RL in 15 14 13 12; do T FPNG_AMD_ASSEMBLE_RL=$RL; done
  T FPNG_AMD_DIRECT=0
done; } 2>&1 | tee $O/r05_assemble_rl.txt
