R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 1 --lanes 1 --serial 2>&1 | grep -E "PYR|rror|value" | tail -40
timeout 900 python -m pytest tests/test_gpu_orb_match.py tests/test_golden.py tests/test_gpu_host_mirror.py -x -q -m gpu 2>&1 | grep -v PYR | tail -3
