#!/bin/bash
# The one parameterised gpurun helper (replaces round 4's tools/r04_*.sh one-offs).  Run on the GPU box through gpurun:
#   gpurun --timeout 900 -- 'bash tools/gq.sh quick'                 headline only (no side legs), JSON summary
#   gpurun --timeout 900 -- 'bash tools/gq.sh serial [ENV=V ...]'    per-kernel ms of one serial step (HIP events, 1 lane)
#   gpurun --timeout 900 -- 'bash tools/gq.sh stats <tag> [bench args]'   rocprofv3 --kernel-trace --stats of a serial step -> gpurun_out/<tag>_kernel_stats.csv
#   gpurun --timeout 900 -- 'bash tools/gq.sh test <pytest args>'    pytest -m gpu subset
#   gpurun --timeout 1800 -- 'bash tools/gq.sh driver'               the driver's exact command on this tree
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
MODE=${1:-quick}; shift || true
SIDE="--no-cpu-baseline --no-extras --no-klt"
summ() { python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d.get("roofline") or {}
        print(json.dumps({"value":d["value"],"ms_per_step":d["ms_per_step"],"verified":d.get("verified_pairs"),"frac":r.get("frac"),"kernels":r.get("kernels_ms_per_step")}))'; }
case $MODE in
  quick)  cd $R && timeout 800 python bench.py $SIDE --verify 8 --steps 20 --warmup 3 "$@" 2>$OUT/gq_quick.err | summ | tee $OUT/gq_quick.json ;;
  serial) cd $R && for kv in "$@"; do export "$kv"; done; timeout 800 python bench.py $SIDE --verify 8 --serial --lanes 1 --steps 5 --warmup 2 2>$OUT/gq_serial.err | summ | tee $OUT/gq_serial.json ;;
  stats)  TAG=${1:-gq}; shift || true; export GFS_BENCH_NO_SUPERVISOR=1; cd /tmp && export TMPDIR=/tmp
          timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python $R/bench.py $SIDE --verify 0 --serial --lanes 1 --steps 5 --warmup 1 --prime 0 "$@" > $OUT/${TAG}_stats.log 2>&1
          f=$(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/${TAG}_kernel_stats.csv && head -30 $OUT/${TAG}_kernel_stats.csv | cut -c1-150 ;;
  test)   cd $R && timeout 1500 python -m pytest -x -q -m gpu "$@" 2>&1 | tail -15 ;;
  driver) cd $R && timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err; echo "rc $?"; python3 -c 'import json;d=json.loads(open("'$OUT'/driver_cmd.json").read().strip().splitlines()[-1]);print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ("value","frac","kernel","achieved","traffic","error","cores","kind","median_ms")}) for k,v in d.items() if k!="config"})' ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
