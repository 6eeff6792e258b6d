R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_batched.py -x -q -m gpu 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lin -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 3 --lanes 1 --serial > $OUT/lin.log 2>&1
f=$(ls $OUT/lin/*/*kernel_stats.csv | head -1); python -c "
import csv,sys
for r in csv.reader(open('$f')):
    if any(k in r[0] for k in ('k_gicp_',)): print(r[0][:50], r[1], r[2], r[3])
"; rm -rf $OUT/lin
tail -c 300 $OUT/lin.log | grep -o '"value": [0-9.]*' | head -1
timeout 600 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-klt --verify 4 --prime 5 2>&1 | tail -1 | grep -o '"value": [0-9.]*' | head -1
