#!/bin/bash
# round 6's closing session on one box: the -m gpu suite, smoke(), the default bench line, kernel-trace + PMC passes of the default
# command and of --mode decode (tools/gpu_profile_round.sh), the decoder against round 5's on the same box.
#   usage (through gpurun): bash tools/gpu_r06_final.sh <tag>
TAG=${1:-r06f}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/gpu_session.sh $TAG
bash tools/gpu_profile_round.sh ${TAG}_8k > /dev/null 2>&1
SKIP_CAL=1 bash tools/gpu_profile_round.sh ${TAG}_decode_8k --mode decode --decode-steps 3 > /dev/null 2>&1
cd $R
timeout 600 python bench.py --mode decode > $O/${TAG}_bench_decode_8k.json 2> $O/${TAG}_bench_decode_8k.err
bash tools/gpu_r06_decode_ab3.sh ${TAG} "_r05dec ." "8K RGBA grad" "4K RGBA grad" "1080p RGB grad" "512x512" "photo" "8K RGBA solid" "8K RGBA blocks" "8K RGBA stripes" "8K RGBA noise" "4K UI" > /dev/null 2>&1
ls $O | grep $TAG
