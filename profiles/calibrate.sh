#!/bin/bash
# Calibration of the HBM traffic counters against kernels with known byte counts (gfs_test_traffic):
#   gpurun -- 'bash profiles/calibrate.sh r03'      -> gpurun_out/<tag>_calibration.json (copy it to profiles/)
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/${TAG}_cal_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_cal_$c -- python $R/profiles/calibrate.py > $OUT/${TAG}_cal_$c.log 2>&1
done
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
known = json.loads([l for l in open(f"{out}/{tag}_cal_FETCH_SIZE.log") if l.startswith("{")][-1])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(f"{out}/{tag}_cal_{c}/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "k_cal_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for (name, k), r in zip(known.items(), rows):   # the cases run in order, one dispatch each
        assert k["kernel"] in r["Kernel_Name"], (name, r["Kernel_Name"])
        res.setdefault(name, dict(asked_bytes=k["bytes"], table_bytes=k["table_bytes"]))[c + "_kb"] = float(r["Counter_Value"])
for name, v in res.items():
    v["fetch_bytes_over_asked"] = round(v.get("FETCH_SIZE_kb", 0) * 1024 / v["asked_bytes"], 4)
    v["write_bytes_over_asked"] = round(v.get("WRITE_SIZE_kb", 0) * 1024 / v["asked_bytes"], 4)
json.dump(res, open(f"{out}/{tag}_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
