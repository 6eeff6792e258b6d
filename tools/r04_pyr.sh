R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_orb_match.py tests/test_golden.py tests/test_gpu_host_mirror.py -x -q -m gpu 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
for kb in 150; do
  export GFS_ORB_PYR_LDS_KB=$kb
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pyr_$kb -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 3 --lanes 1 --serial > $OUT/pyr_$kb.log 2>&1
  f=$(ls $OUT/pyr_$kb/*/*kernel_stats.csv | head -1); echo "== $kb"; python -c "
import csv,sys
for r in csv.reader(open('$f')):
    if any(k in r[0] for k in ('k_pyr_area','k_fast_cells','k_blur7')): print(r[0][:50], r[1], r[3])
"; rm -rf $OUT/pyr_$kb
  tail -c 300 $OUT/pyr_$kb.log | grep -o '"value": [0-9.]*' | head -1
done
