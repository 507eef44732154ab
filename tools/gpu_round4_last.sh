#!/bin/bash
# The last GPU-box session of round 4 (a few minutes of box time left): the corpus workflow over ordinary PNGs incl. the
# screenshot-like content (tools/make_corpus.py, tools/run_corpus.sh: every output byte-identical to the unmodified reference's),
# then as much of the -m gpu suite as the remaining time allows (the product's sources are those of profiles/r04_final_gpu_tests.txt).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
date +%s > $O/last_t0
python tools/make_corpus.py /tmp/corpus > /dev/null 2>$O/last_corpus_make.err
bash tools/run_corpus.sh /tmp/corpus $O/r04_corpus.csv 2>&1 | tail -3
echo "corpus lines: $(grep -c , $O/r04_corpus.csv), seconds so far: $(( $(date +%s) - $(cat $O/last_t0) ))"
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee $O/r04_last_tests.txt
echo "seconds: $(( $(date +%s) - $(cat $O/last_t0) ))"
