R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== 1080p x 256, 8 sets"; timeout 200 python tools/region_diag.py 1920 1080 3 256 8 2>&1 | grep region
echo "== 1080p x 256, 4 sets"; timeout 200 python tools/region_diag.py 1920 1080 3 256 4 2>&1 | grep region
echo "== 512 x 1024, 8 sets"; timeout 200 python tools/region_diag.py 512 512 3 1024 8 2>&1 | grep region
echo "== 1080p x 256, 8 sets, HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 timeout 200 python tools/region_diag.py 1920 1080 3 256 8 2>&1 | grep region
