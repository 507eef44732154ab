R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 300 python tools/oneshot_split.py 2>&1 | grep " x "
timeout 300 python tools/oneshot_split.py 1920 1080 3 64 2>&1 | grep " x "
bash tools/gpu_r05_regions.sh 2>&1 | grep "sets="
