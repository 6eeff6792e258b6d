R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_batched.py -x -q -m gpu > $OUT/r04c_tests.log 2>&1; tail -4 $OUT/r04c_tests.log
cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/bench.py --batch 64 --lanes 1 --serial --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-klt --verify 8 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("c4s", d["value"], d["ms_per_step"], d.get("verified_pairs"), d["roofline"]["kernels_ms_per_step"])'
timeout 300 python $R/bench.py --batch 64 --lanes 2 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-klt --verify 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("c4", d["value"], d["ms_per_step"])'
timeout 300 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-klt --verify 8 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("c2", d["value"], d["ms_per_step"], d.get("verified_pairs"), d["roofline"]["kernels_ms_per_step"])'
