#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
t0=$(date +%s)
timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err; echo "rc $? in $(( $(date +%s) - t0 )) s"
python3 -c "
import json
d=json.loads(open('gpurun_out/driver_cmd.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d.get('skipped_legs'), d.get('side_legs_incomplete'))
print({k:r.get(k) for k in ('traffic','traffic_raw_counters','traffic_source','traffic_live','traffic_committed','algorithmic_bytes_per_launch','launches_per_step')})"
