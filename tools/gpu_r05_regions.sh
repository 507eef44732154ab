# round 5: why is the first timed region of the short-row lines slow since the bench cycles through eight sets of output buffers?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['runs'])"; }
for S in 8; do for L in 4; do
  FPNG_AMD_LANES=$L timeout 200 python bench.py --no-cpu-baseline --workload 1080p --batch 256 --regions 6 --out-sets $S 2>/dev/null | grep "^{" | line "sets=$S lanes=$L 1080p"
  FPNG_AMD_LANES=$L timeout 200 python bench.py --no-cpu-baseline --workload 512 --batch 1024 --regions 6 --out-sets $S 2>/dev/null | grep "^{" | line "sets=$S lanes=$L 512"
done; done
timeout 200 python bench.py --no-cpu-baseline --workload 1080p --batch 256 --regions 6 --prewarm 200 2>/dev/null | grep "^{" | line "sets=8 prewarm=200 1080p"
timeout 200 python bench.py --no-cpu-baseline --regions 6 2>/dev/null | grep "^{" | line "sets=8 8k"
