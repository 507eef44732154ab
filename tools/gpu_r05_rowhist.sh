# round 5: the histogram pass with per-row counts (hist_rows_kernel, FPNG_AMD_ROWHIST=1) against hist_kernel: same files, kernel time, step
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d.get('parity_checked'), d['roofline']['phase_ms'])"; }
for rep in 1 2; do for V in 0 1; do
  FPNG_AMD_ROWHIST=$V timeout 200 python bench.py --no-cpu-baseline --flags 1 2>/dev/null | grep "^{" | line "rowhist=$V 8k_2pass"
  FPNG_AMD_ROWHIST=$V timeout 200 python bench.py --no-cpu-baseline --flags 1 --workload 4k --batch 16 2>/dev/null | grep "^{" | line "rowhist=$V 4k_x16_2pass"
done; done
FPNG_AMD_ROWHIST=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_kat or two_pass or skewed" 2>&1 | tail -2
