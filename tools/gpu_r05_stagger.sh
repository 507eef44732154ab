# round 5: the 2-pass stagger rule under four lanes, other shapes (the product now: off with four lanes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d.get('parity_checked'))"; }
for rep in 1 2; do for S in default 1; do
  [ $S = default ] && unset FPNG_AMD_STAGGER || export FPNG_AMD_STAGGER=$S
  timeout 200 python bench.py --no-cpu-baseline --flags 1 2>/dev/null | grep "^{" | line "stagger=$S 8k_2pass"
  timeout 200 python bench.py --no-cpu-baseline --flags 1 --workload 1080p --batch 256 2>/dev/null | grep "^{" | line "stagger=$S 1080p_rgb_x256_2pass"
  timeout 200 python bench.py --no-cpu-baseline --flags 1 --workload 512 --batch 1024 2>/dev/null | grep "^{" | line "stagger=$S 512_x1024_2pass"
  timeout 200 python bench.py --no-cpu-baseline --flags 1 --workload 4k --batch 16 2>/dev/null | grep "^{" | line "stagger=$S 4k_x16_2pass"
  timeout 200 python bench.py --no-cpu-baseline --flags 1 --batch 1 2>/dev/null | grep "^{" | line "stagger=$S 8k_x1_2pass"
done; done
