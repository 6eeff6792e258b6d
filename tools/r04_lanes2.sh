R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { echo "$*: $(timeout 400 python $R/bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-klt --verify 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])')" >> $OUT/r04_lanes2.log; }
run --lanes 4
run --lanes 5 --batch 640
run --lanes 6 --batch 768
run --lanes 8 --batch 1024
run --lanes 4 --batch 768
run --lanes 3 --batch 384
run --lanes 4
cat $OUT/r04_lanes2.log
