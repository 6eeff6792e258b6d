R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for cfg in "c4s --batch 64 --lanes 1 --serial" "c3s --workload c3 --batch 32 --lanes 1 --serial"; do
  set -- $cfg; tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04e_$tag -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 3 "$@" > $OUT/r04e_$tag.log 2>&1
  f=$(ls $OUT/r04e_$tag/*/*kernel_stats.csv | head -1); cp $f $OUT/r04e_${tag}_kernel_stats.csv; rm -rf $OUT/r04e_$tag
  echo "== $tag"; python -c "
import csv
rows=[r for r in csv.reader(open('$OUT/r04e_${tag}_kernel_stats.csv'))][1:]
tot=0
for r in rows:
    per_step=float(r[2])/15/1e6   # 3 prime + 2 warm + 10 steps
    tot+=per_step
    if per_step>0.03: print('%-44s %6s %8.3f ms/step'%(r[0][:44], r[1], per_step))
print('sum', round(tot,3))
"
  tail -c 300 $OUT/r04e_$tag.log | grep -o '"value": [0-9.]*' | head -1
done
