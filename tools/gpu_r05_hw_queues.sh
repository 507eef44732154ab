# round 5: lanes x hardware queues on the other bench lines (bench.py's own timing; one box, alternating)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d.get('parity_checked'))"; }
for rep in 1 2; do for Q in 4 8; do for L in 2 4; do
  export GPU_MAX_HW_QUEUES=$Q FPNG_AMD_LANES=$L
  timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | grep "^{" | line "Q=$Q L=$L 8k_1pass"
  timeout 200 python bench.py --no-cpu-baseline --flags 1 2>/dev/null | grep "^{" | line "Q=$Q L=$L 8k_2pass"
  timeout 200 python bench.py --no-cpu-baseline --workload 1080p --batch 256 2>/dev/null | grep "^{" | line "Q=$Q L=$L 1080p_rgb_x256"
  timeout 200 python bench.py --no-cpu-baseline --workload 512 --batch 1024 2>/dev/null | grep "^{" | line "Q=$Q L=$L 512_rgb_x1024"
done; done; done
unset FPNG_AMD_LANES
for Q in 4 8; do for G in 1 2; do
  GPU_MAX_HW_QUEUES=$Q FPNG_AMD_DECODE_DEVICE_GROUPS=$G timeout 200 python bench.py --mode decode 2>/dev/null | grep "^{" | line "Q=$Q groups=$G decode_8k"
done; done
