R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_lba.py tests/test_lba_adaptor.py -x -q -m gpu > $OUT/r04e_tests.log 2>&1; tail -3 $OUT/r04e_tests.log
for k in 1 2; do
timeout 300 python tools/lba_probe.py 2>&1 | tail -1
GFS_LBA_POLL=sync timeout 300 python tools/lba_probe.py 2>&1 | tail -1
done
