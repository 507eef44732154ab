# round 5: the RGB lines under four lanes: seven waves per SIMD in the 3-channel walk (room for the other lanes' assemble), assemble's range size
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d.get('parity_checked'))"; }
for rep in 1 2; do
  for W in "1080p 256" "512 1024"; do set -- $W
    timeout 200 python bench.py --no-cpu-baseline --workload $1 --batch $2 2>/dev/null | grep "^{" | line "product $1x$2"
    FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_rows_w7.so timeout 200 python bench.py --no-cpu-baseline --workload $1 --batch $2 2>/dev/null | grep "^{" | line "rows_w7 $1x$2"
    for RL in 13 14 15 16; do FPNG_AMD_ASSEMBLE_RL=$RL timeout 200 python bench.py --no-cpu-baseline --workload $1 --batch $2 2>/dev/null | grep "^{" | line "assemble_rl=$RL $1x$2"; done
  done
done
