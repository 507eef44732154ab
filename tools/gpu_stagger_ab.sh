R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for a in "--flags 1" "--flags 1 --workload 4k --batch 128 --steps 10"; do for i in 1 2 3; do for S in 1 0; do
  FPNG_AMD_STAGGER=$S python bench.py --no-cpu-baseline --steps 20 --warmup 5 $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('stagger=$S [$a]', d['ms_per_step'])"
done; done; done
