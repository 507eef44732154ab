#!/bin/bash
# The FIRST GPU-box session of the next round (~6 minutes of box time), prepared at the end of round 4 when its GPU minutes were spent:
#  1. the -m gpu suite + smoke() on the tree as round 4 left it (its last host-side fixes -- the flag mask, the stored stream with one
#     zero byte more, the reserved length symbols -- were only run through their own targeted tests on a box);
#  2. tools/gpu_next_checks.py: the stored-or-compressed flip point, skewed histograms, token-edited megapixel files through the kernels;
#  3. the build variants of the decoder's emit walk (lean: fewer vector instructions; stage: the 16-byte groups put together in LDS) against
#     the product on the same box: their own parity (tests/test_gpu_decode.py against each library), then kernel stats of one step each;
#  4. the default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 | tee $O/r05_first_tests.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/r05_first_tests.txt
timeout 300 python tools/gpu_next_checks.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee $O/r05_next_checks.txt; echo "next checks rc=${PIPESTATUS[0]}" | tee -a $O/r05_next_checks.txt
# (build the variants in the dev container BEFORE the call -- hipcc cross-compiles there, the libraries travel with the snapshot:
#    for V in lean stage stage_lean lead96 lead64; do python -m fpng_amd.build --variant $V; done      -- 30 s each, not box time)
for V in lean stage stage_lean lead96 lead64; do [ -f fpng_amd/lib/libfpng_amd_$V.so ] || python -m fpng_amd.build --variant $V > /dev/null 2>&1; done
# (the variants' own parity first: the decoder's tests against each library, then one step's kernel stats next to the product's)
for V in lean stage_lean; do FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_$V.so timeout 200 python -m pytest tests/test_gpu_decode.py -m gpu -x -q 2>&1 | tail -2 | sed "s/^/[$V] /" | tee -a $O/r05_variants_tests.txt; done
for CASE in "8K RGBA grad x 8" "photo 11 MP RGB x 8" "4K UI glyphs"; do bash tools/gpu_decode_ab.sh "$CASE" - _lean _stage _stage_lean _lead96 _lead64 2>&1 | tee -a $O/r05_variants_ab.txt; done
timeout 600 python bench.py > $O/bench_r05_first.json 2> $O/bench_r05_first.err; tail -c 400 $O/bench_r05_first.json; echo
