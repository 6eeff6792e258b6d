R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 0 --lanes 1 --serial 2>&1 | grep -E "OCTT|rror" | tail -16
