# round 5: the chain's older knobs again under four lanes over eight hardware queues (stagger rule, waves per SIMD of the wide 4-channel walk)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d.get('parity_checked'))"; }
for rep in 1 2; do
  for fl in 0 1; do
    timeout 200 python bench.py --no-cpu-baseline --flags $fl 2>/dev/null | grep "^{" | line "default flags=$fl"
    FPNG_AMD_STAGGER=0 timeout 200 python bench.py --no-cpu-baseline --flags $fl 2>/dev/null | grep "^{" | line "stagger=0 flags=$fl"
    FPNG_AMD_STAGGER=1 timeout 200 python bench.py --no-cpu-baseline --flags $fl 2>/dev/null | grep "^{" | line "stagger=1 flags=$fl"
    FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_rows4_w8.so timeout 200 python bench.py --no-cpu-baseline --flags $fl 2>/dev/null | grep "^{" | line "rows4_w8 flags=$fl"
    FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_rows4_w6.so timeout 200 python bench.py --no-cpu-baseline --flags $fl 2>/dev/null | grep "^{" | line "rows4_w6 flags=$fl"
    FPNG_AMD_LANES=5 timeout 200 python bench.py --no-cpu-baseline --flags $fl 2>/dev/null | grep "^{" | line "lanes=5 flags=$fl"
  done
done
