# round 5: do one-image chains keep their local streams in the Infinity Cache when the streams are stored with the default policy?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do for B in 1 2; do for L in 3 4; do
  echo -n "product L=$L: "; FPNG_AMD_LANES=$L python tools/submit_cost.py 7680 4320 4 $B 2>&1 | grep " x "
  echo -n "nont    L=$L: "; FPNG_AMD_LANES=$L FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_nont.so python tools/submit_cost.py 7680 4320 4 $B 2>&1 | grep " x "
done; done; done
