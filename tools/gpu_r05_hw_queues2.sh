# round 5: the library's own default (eight hardware queues asked for at load, four lanes) against the old one, and more lanes over sixteen queues
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lanes or pipelin or golden_batch" 2>&1 | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d.get('parity_checked'))"; }
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | grep "^{" | line "$name 8k_1pass"
  env "$@" timeout 200 python bench.py --no-cpu-baseline --flags 1 2>/dev/null | grep "^{" | line "$name 8k_2pass"
  env "$@" timeout 200 python bench.py --no-cpu-baseline --workload 1080p --batch 256 2>/dev/null | grep "^{" | line "$name 1080p_rgb_x256"
  env "$@" timeout 200 python bench.py --no-cpu-baseline --workload 512 --batch 1024 2>/dev/null | grep "^{" | line "$name 512_rgb_x1024"
}
for rep in 1 2; do
  run "default" A=1
  run "old(Q4,L2)" FPNG_AMD_KEEP_HW_QUEUES=1
  run "Q16,L6" GPU_MAX_HW_QUEUES=16 FPNG_AMD_LANES=6
  run "Q16,L8" GPU_MAX_HW_QUEUES=16 FPNG_AMD_LANES=8
  run "Q8,L3" GPU_MAX_HW_QUEUES=8 FPNG_AMD_LANES=3
done
for B in 1 2 4; do python tools/submit_cost.py 7680 4320 4 $B 2>&1 | grep " x "; FPNG_AMD_KEEP_HW_QUEUES=1 python tools/submit_cost.py 7680 4320 4 $B 2>&1 | grep " x "; GPU_MAX_HW_QUEUES=16 FPNG_AMD_LANES=8 python tools/submit_cost.py 7680 4320 4 $B 2>&1 | grep " x "; done
