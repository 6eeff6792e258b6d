R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cd /tmp
for cfg in "c4 --batch 64 --lanes 2" "c4s --batch 64 --lanes 1 --serial" "c3 --workload c3 --batch 32 --lanes 2" "c3s --workload c3 --batch 32 --lanes 1 --serial"; do
  set -- $cfg; tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04b_$tag -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 3 "$@" > $OUT/r04b_$tag.log 2>&1
  f=$(ls $OUT/r04b_$tag/*/*kernel_stats.csv | head -1); cp $f $OUT/r04b_${tag}_kernel_stats.csv; rm -rf $OUT/r04b_$tag
  tail -c 400 $OUT/r04b_$tag.log
done
