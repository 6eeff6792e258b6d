R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_orb_match.py tests/test_gpu_host_mirror.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
python tools/stream_probe.py 2>&1 | head -12
