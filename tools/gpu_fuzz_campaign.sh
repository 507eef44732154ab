#!/bin/bash
# A long run of the harness' two fuzz modes (reference fpng_test.cpp:381-682) through the drop-in on the GPU, every output
# compared byte for byte with the UNMODIFIED reference encoder (oracle/_ref) and decoded back.  -> gpurun_out/fuzz_campaign.txt
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; cd $R
T=$R/fpng_amd/lib/fpng_amd_test; J=$R/oracle/_ref/libfpng_ref.so; P=$R/tests/golden/real/example_rgb_f1.png
O=$R/gpurun_out/fuzz_campaign.txt; : > $O
run() { echo "## fpng_amd_test $* --judge libfpng_ref.so" >> $O; s=$(date +%s); LD_PRELOAD= timeout 600 $T "$@" --judge $J -o /tmp/fz.png > /tmp/fz.log 2>&1; rc=$?; tail -12 /tmp/fz.log >> $O; echo "rc=$rc, $(( $(date +%s) - s )) s" >> $O; }
run -e -n ${N1:-8000} $P
run -e -s -n ${N2:-4000} $P
run -e -a -n ${N2:-4000} $P
run -e -a -s -n ${N3:-2000} $P
run -E -n ${N4:-1500} -m 2049 synth:noise:8x8x3
run -E -s -n ${N5:-700} -m 2049 synth:noise:8x8x4
cat $O
