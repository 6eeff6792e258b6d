R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_sbp.py tests/test_gpu_sbp_of.py tests/test_gpu_pose.py -x -q -m gpu > $OUT/r04d_tests.log 2>&1; tail -4 $OUT/r04d_tests.log
timeout 300 python tools/stream_probe.py 2>&1 | head -4
