R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for l in 1 2 3 4 6 8; do
  echo "lanes $l: $(timeout 300 python $R/bench.py --batch 64 --lanes $l --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-klt --verify 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])')" >> $OUT/r04_lanes.log
done
for l in 2 4; do
  echo "c3 lanes $l: $(timeout 300 python $R/bench.py --workload c3 --batch 32 --lanes $l --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-klt --verify 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])')" >> $OUT/r04_lanes.log
done
cat $OUT/r04_lanes.log
