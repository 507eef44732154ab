#!/bin/bash
# The last GPU-box session of round 4: the -m gpu suite + smoke() on the final tree (the corpus workflow over ordinary PNGs incl. the
# screenshot-like content ran in the session before: profiles/r04_corpus.csv, and again inside the suite,
# test_command_line_harness_on_ordinary_png_files).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee $O/r04_last_tests.txt
timeout 30 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/r04_last_tests.txt
