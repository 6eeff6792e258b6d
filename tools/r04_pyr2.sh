R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1200 python -m pytest tests/test_gpu_orb_match.py tests/test_golden.py tests/test_gpu_host_mirror.py tests/test_frame_helpers.py tests/test_gpu_batched.py -q -m gpu 2>&1 | tail -3
python tools/stream_probe.py 2>&1 | head -3
timeout 600 python bench.py --workload c3 --batch 32 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 3 --lanes 2 2>&1 | tail -1 | cut -c1-200
