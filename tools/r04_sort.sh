R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_gicp.py -x -q -m gpu -k "voxel_sort or 720p or wide_extent or preprocess" > $OUT/r04b_tests.log 2>&1; tail -5 $OUT/r04b_tests.log
cd /tmp; export TMPDIR=/tmp
for l in 2; do
  timeout 300 python $R/bench.py --workload c3 --batch 32 --lanes $l --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-klt --verify 8 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("c3", d["value"], d["ms_per_step"], d.get("verify"), d["roofline"]["kernels_ms_per_step"])'
done
