R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for a in "" "--workload 1080p --batch 256" "--flags 1" "--workload 512 --batch 1024"; do
for i in 1 2 3; do for L in 2 3 4; do
  FPNG_AMD_LANES=$L python bench.py --no-cpu-baseline --steps 30 --warmup 5 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes=$L [$a]', d['ms_per_step'])"
done; done; done
