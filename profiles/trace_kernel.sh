#!/bin/bash
# Per-launch durations of one kernel over one serial step (1 lane, the whole batch per launch):
#   gpurun -- 'bash profiles/trace_kernel.sh k_gicp_linearize tag [ENV=VALUE ...]'
export GFS_BENCH_NO_SUPERVISOR=1  # the profiler must see the process that launches the kernels
set -u
K=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf $OUT/${TAG}_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- \
  python $R/bench.py --steps 1 --warmup 1 --serial --lanes 1 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 0 > $OUT/${TAG}_trace.log 2>&1
python - "$OUT/${TAG}_trace" "$K" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:40]))
rows.sort()
d = [round(x[1], 1) for x in rows]
print(sys.argv[2], "launches", len(d), "total_us", round(sum(d), 1))
print(d)
PY
