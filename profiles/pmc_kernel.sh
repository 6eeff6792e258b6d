#!/bin/bash
# PMC counters of one kernel, per dispatch, over one serial step (counters in their own passes, no trace domains):
#   gpurun -- 'bash profiles/pmc_kernel.sh k_gicp_linearize tag "SQ_WAVES SQ_INSTS_VALU ..." ["second set" ...]'
export GFS_BENCH_NO_SUPERVISOR=1  # the profiler must see the process that launches the kernels
set -u
K=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for set in "$@"; do
  n=$((n+1))
  rm -rf $OUT/${TAG}_pmc$n
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $OUT/${TAG}_pmc$n -- \
    python $R/bench.py --steps 1 --warmup 1 --serial --lanes 1 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 0 > $OUT/${TAG}_pmc$n.log 2>&1
  python - "$OUT/${TAG}_pmc$n" "$K" <<'PY'
import csv, glob, sys, collections
per = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(per)
print(sys.argv[2], "dispatches", len(ids))
half = ids[len(ids) // 2:]  # the timed step (the first half is the warm-up step)
for k, i in enumerate(half[:4]):
    print(" dispatch", k, {a: int(b) for a, b in per[i].items()})
tot = collections.Counter()
for i in half:
    tot.update(per[i])
print(" step total", {a: int(b) for a, b in tot.items()})
PY
done
