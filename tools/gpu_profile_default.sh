#!/bin/bash
# bench line + rocprofv3 kernel stats + PMC traffic passes (+ calibration) of the DEFAULT bench command: tools/gpu_profile_default.sh <round>
RND=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>$O/bench_${RND}_8k.err | grep "^{" > $O/bench_${RND}_8k.json
python -c "import json; d=json.load(open('$O/bench_${RND}_8k.json')); print('8k', d['value'], d['ms_per_step'], d['runs'], d['parity_checked'], d['roofline']['phase_ms'])"
bash tools/gpu_profile_round.sh ${RND}_8k > /dev/null 2>&1
ls $O/prof_${RND}_8k
