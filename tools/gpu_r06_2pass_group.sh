#!/bin/bash
# round 6, verdict item 5: the 2-pass chain in groups of g images (hist -> table -> walk per group) against the whole-submission order,
# same box, alternating.  usage (through gpurun): bash tools/gpu_r06_2pass_group.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
OUT=$O/r06_2pass_group.txt; : > $OUT
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"  {d['value']/1e3:8.1f} GP/s  {d['ms_per_step']:.4f} ms/step  runs {d['runs']}  parity {d['parity_checked']} ({d['parity_images']})")
except Exception as e:
    print("  FAILED", e, open(sys.argv[1]).read()[-400:])
PY
}
for rep in 1 2; do
for W in "--workload 8k --batch 8" "--workload 4k --batch 16" "--workload 4k --batch 64"; do
  for L in 4 2; do
    for G in 0 1 2 4; do
      echo "rep $rep: $W flags=1 lanes=$L group=$G" >> $OUT
      FPNG_AMD_LANES=$L FPNG_AMD_2PASS_GROUP=$G timeout 300 python bench.py $W --flags 1 --steps 20 --warmup 4 --no-cpu-baseline --decode-steps 1 --regions 3 > $O/tmp_2pg.json 2> $O/tmp_2pg.err
      line $O/tmp_2pg.json >> $OUT
    done
  done
done
done
cat $OUT
