#!/bin/bash
# One GPU-box session: bench line + rocprofv3 kernel stats + PMC traffic passes for the default workload and the others the
# review asked for.  tools/gpu_profile_workloads.sh <round>   (then tools/summarize_workloads.sh <round> locally)
RND=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out
run() { tag=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>$O/bench_${RND}_$tag.err | grep "^{" > $O/bench_${RND}_$tag.json
  python -c "import json; d=json.load(open('$O/bench_${RND}_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['runs'], d['parity_checked'], d['roofline']['phase_ms'])"
  bash tools/gpu_profile_round.sh ${RND}_$tag "$@" > /dev/null 2>&1
}
run 8k
export SKIP_CAL=1
run 8k_2pass --flags 1
run 1080p --workload 1080p --batch 256
run 512 --workload 512 --batch 1024
run 8k_noise --kind noise
run 4k --workload 4k --batch 16
