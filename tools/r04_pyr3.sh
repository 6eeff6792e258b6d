R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in lds hbm; do
  if [ $v = hbm ]; then export GFS_ORB_PYR_XTAB_HBM=1; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pyr_$v -- python $R/bench.py --workload c3 --batch 32 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 3 --lanes 1 --serial > $OUT/pyr_$v.log 2>&1
  f=$(ls $OUT/pyr_$v/*/*kernel_stats.csv | head -1); echo "== $v"; python -c "
import csv,sys
for r in csv.reader(open('$f')):
    if any(k in r[0] for k in ('k_pyr_area','k_fast_cells','k_blur7')): print(r[0][:50], r[1], r[3])
"; rm -rf $OUT/pyr_$v
done
