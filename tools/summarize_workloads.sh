#!/bin/bash
# after tools/gpu_profile_workloads.sh <round>: profiles/<round>_<workload>_{kernel_stats.csv,pmc_traffic.txt,bench.json}
RND=$1
declare -A ARGS=( [8k]="" [8k_2pass]="--flags 1" [1080p]="--workload 1080p --batch 256" [512]="--workload 512 --batch 1024" [8k_noise]="--kind noise" [4k]="--workload 4k --batch 16" )
for tag in 8k 8k_2pass 1080p 512 8k_noise 4k; do
  alg=$(python -c "import json; print(json.load(open('gpurun_out/bench_${RND}_$tag.json'))['roofline']['algorithmic_bytes_per_launch'])")
  python tools/summarize_profile.py ${RND}_$tag $alg ${RND}_8k ${ARGS[$tag]} | tail -12
  cp gpurun_out/bench_${RND}_$tag.json profiles/${RND}_${tag}_bench.json
done
