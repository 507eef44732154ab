# round 5: one-image submissions against lanes (FPNG_AMD_LANES) and hardware queues (GPU_MAX_HW_QUEUES, ROCclr's default: 4)
for Q in 4 8 16; do for L in 2 3 4; do for B in 1 2 8; do
  echo -n "GPU_MAX_HW_QUEUES=$Q "; GPU_MAX_HW_QUEUES=$Q FPNG_AMD_LANES=$L timeout 120 python tools/submit_cost.py 7680 4320 4 $B 2>&1 | grep " x "
done; done; done
