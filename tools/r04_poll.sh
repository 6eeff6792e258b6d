R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_batched.py -x -q -m gpu -k "not random_pairs and not voxel_sort" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
b() { timeout 300 python $R/bench.py "$@" --no-cpu-baseline --no-extras --no-klt --verify 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'; }
for mode in side inline side inline; do
  export GFS_GICP_POLL=$mode
  echo "$mode c4 $(b --batch 64 --lanes 2 --steps 40 --warmup 5) c3 $(b --workload c3 --batch 32 --lanes 2 --steps 40 --warmup 5) c2 $(b --steps 20 --warmup 3)"
done
cd $R
for mode in side inline; do GFS_GICP_POLL=$mode python tools/stream_probe.py 2>&1 | head -1 | cut -c1-300; done
