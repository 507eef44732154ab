# Same-box comparison with the round-1 tree: mkdir r1tree && git archive 081c225 | tar -x -C r1tree && (cd r1tree && python -m fpng_amd.build)
# (r1tree/ is git-ignored but travels to the GPU box); run through gpurun: bash tools/ab_r1.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
p() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline'].get('phase_ms'))"; }
for i in 1 2 3; do
  (cd r1tree && python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | p r1)
  python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | p cur
done
(cd r1tree && python bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload 1080p --batch 256 2>/dev/null | p r1_1080p)
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload 1080p --batch 256 2>/dev/null | p cur_1080p
(cd r1tree && python bench.py --no-cpu-baseline --steps 20 --warmup 5 --flags 1 2>/dev/null | p r1_2pass)
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --flags 1 2>/dev/null | p cur_2pass
