#!/bin/bash
# Collects the measured evidence of a round on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh r02a'
# Outputs land in gpurun_out/<tag>_*; copy the summaries into profiles/ (see profiles/README.md).
# PMC counters are collected in their own passes (no trace domains), one counter set per pass.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp

# run <seconds> <log> <command...>: one retry when the pass was killed or failed (a killed WRITE_SIZE pass went unnoticed in round 4)
run() {
  local t=$1 log=$2; shift 2
  for attempt in 1 2; do
    timeout $t "$@" > $log 2>&1 && return 0
    echo "collect.sh: attempt $attempt of [$*] ended with rc $? (see $log)" >&2
  done
  echo "collect.sh: GIVING UP on [$*]" >&2
  return 1
}
NS="--no-supervisor"  # the profiler must see the process that launches the kernels

# 1. the bench line (default arguments), CPU baseline included; then the driver's exact command
timeout 1600 python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || echo "collect.sh: bench.py rc $?" >&2
timeout 1600 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_cmd.json 2> $OUT/${TAG}_bench_driver_cmd.err || echo "collect.sh: driver command rc $?" >&2

# 2. kernel trace + stats of the same workload (overlapped lanes)
run 600 $OUT/${TAG}_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- \
  python $R/bench.py $NS --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 0
# ... and one module at a time (the per-kernel durations the roofline object is computed from)
run 600 $OUT/${TAG}_stats_serial.log rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_serial -- \
  python $R/bench.py $NS --steps 5 --warmup 1 --serial --lanes 1 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 0

# 3. HBM-side traffic: FETCH_SIZE and WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md, PMC slots)
for c in FETCH_SIZE WRITE_SIZE; do
  run 600 $OUT/${TAG}_pmc_$c.log rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -- \
    python $R/bench.py $NS --steps 2 --warmup 1 --serial --lanes 1 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 0
done
# 4. issue / wait mix of the hot kernels (at most four counters a pass: a seven-counter pass hung for its whole timeout in round 4)
n=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  n=$((n+1)); d=pmc_sq$n; [ $n = 1 ] && d=pmc_sq
  run 300 $OUT/${TAG}_$d.log rocprofv3 --pmc $set --output-format csv -d $OUT/${TAG}_$d -- \
    python $R/bench.py $NS --steps 2 --warmup 1 --serial --lanes 1 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 0
done

python $R/profiles/summarize.py $OUT $TAG
tail -c 600 $OUT/${TAG}_bench.json
