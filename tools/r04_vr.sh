R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_batched.py -x -q -m gpu 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vr -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 3 --lanes 1 --serial > $OUT/vr.log 2>&1
f=$(ls $OUT/vr/*/*kernel_stats.csv | head -1); python -c "
import csv,sys
for r in csv.reader(open('$f')):
    if any(k in r[0] for k in ('k_voxel_reduce','k_cell_build','k_grid_fill','k_voxel_keys','k_cell_sort')): print(r[0][:50], r[1], r[3])
"; rm -rf $OUT/vr
