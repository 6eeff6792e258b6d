R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_gicp.py tests/test_frame_helpers.py -x -q -m gpu -k "key_merge or frame_rgbd or preprocess" > $OUT/r04f_tests.log 2>&1; tail -4 $OUT/r04f_tests.log
cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/bench.py --workload c3 --batch 32 --lanes 1 --serial --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-klt --verify 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("c3s", d["value"], d["ms_per_step"], d["roofline"]["kernels_ms_per_step"])'
timeout 300 python $R/bench.py --workload c3 --batch 32 --lanes 2 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-klt --verify 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("c3", d["value"], d["ms_per_step"])'
cd $R; python tools/stream_overhead.py 2>&1 | head -3
