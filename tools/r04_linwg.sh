R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for wg in 64 128 256 64 256; do
  echo -n "wg=$wg: "
  GFS_GICP_LIN_WG=$wg timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 5 2>&1 | tail -1 | grep -o '"value": [0-9.]*' | head -1
done
for wg in 64 256; do
  echo -n "c4 wg=$wg: "
  GFS_GICP_LIN_WG=$wg timeout 600 python bench.py --batch 64 --lanes 2 --steps 40 --warmup 3 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 5 2>&1 | tail -1 | grep -o '"value": [0-9.]*' | head -1
done
