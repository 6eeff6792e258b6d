R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for l in 1 2 3 4; do
  echo -n "c4 lanes=$l: "
  timeout 600 python bench.py --batch 64 --lanes $l --steps 40 --warmup 3 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 5 2>&1 | tail -1 | grep -o '"value": [0-9.]*' | head -1
done
for l in 1 2 4; do
  echo -n "c3 lanes=$l: "
  timeout 600 python bench.py --workload c3 --batch 32 --lanes $l --steps 30 --warmup 3 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 5 2>&1 | tail -1 | grep -o '"value": [0-9.]*' | head -1
done
for l in 3 6; do
  echo -n "c2 lanes=$l: "
  timeout 600 python bench.py --lanes $l --steps 30 --warmup 3 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 5 2>&1 | tail -1 | grep -o '"value": [0-9.]*' | head -1
done
