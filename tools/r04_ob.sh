R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_orb_match.py tests/test_golden.py tests/test_gpu_host_mirror.py -x -q -m gpu 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ob -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 3 --lanes 1 --serial > $OUT/ob.log 2>&1
f=$(ls $OUT/ob/*/*kernel_stats.csv | head -1); python -c "
import csv,sys
for r in csv.reader(open('$f')):
    if any(k in r[0] for k in ('k_pyr_area','k_fast_cells','k_blur7','k_orient','k_octree')): print(r[0][:50], r[1], r[3])
"; rm -rf $OUT/ob
tail -c 300 $OUT/ob.log | grep -o '"value": [0-9.]*' | head -1
