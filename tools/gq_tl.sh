#!/bin/bash
# kernel trace of a small-batch bench run, per hardware queue (tools/probes/chain_timeline.py):  bash tools/gq_tl.sh <tag> [bench args]
export GFS_BENCH_NO_SUPERVISOR=1
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_tl -- \
  python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 2 "$@" > $OUT/${TAG}_tl.log 2>&1
grep '^{' $OUT/${TAG}_tl.log | python3 -c 'import sys,json
for l in sys.stdin: d=json.loads(l); print("bench under the profiler:", d["value"], d["ms_per_step"])'
python3 $R/tools/probes/chain_timeline.py $OUT/${TAG}_tl 0.6
find $OUT/${TAG}_tl -name '*kernel_trace.csv' -size +20M -delete
