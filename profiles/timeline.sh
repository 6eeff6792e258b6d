#!/bin/bash
# Kernel timeline of the overlapped headline step: wall span, time with >= 1 / >= 2 / ... kernels in flight, per-kernel totals.
#   gpurun -- 'bash profiles/timeline.sh tag [bench args]'
export GFS_BENCH_NO_SUPERVISOR=1  # the profiler must see the process that launches the kernels
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_tl -- \
  python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 2 "$@" > $OUT/${TAG}_tl.log 2>&1
python - "$OUT/${TAG}_tl" <<'PY'
import csv, glob, re, sys, collections
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(kb?_\w+|__amd_rocclr_\w+)", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40]))
ev.sort()
# the timed region = the last 6/10 of the launches by time: take the last 55 % of the span
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t0 + int((t1 - t0) * 0.45)
ev = [e for e in ev if e[0] >= lo]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, n in ev:
    pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort()
cov = collections.Counter(); alone = collections.Counter(); cur = 0; last = t0; live = collections.Counter()
for t, d, n in pts:
    cov[min(cur, 8)] += t - last
    if cur == 1:
        alone[next(k for k, v in live.items() if v > 0)] += t - last
    last = t; cur += d; live[n] += d
span = t1 - t0
print("span_ms", round(span / 1e6, 2), "in flight:", {k: round(v / span, 3) for k, v in sorted(cov.items())})
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in ev:
    tot[n] += e - s; cnt[n] += 1
print("sum_kernel_ms", round(sum(tot.values()) / 1e6, 2))
print("alone (only kernel in flight), ms:", {k: round(v / 1e6, 2) for k, v in alone.most_common(12)})
for n, v in tot.most_common(16):
    print(f"  {n:40s} {v/1e6:8.2f} ms {cnt[n]:6d} launches  avg {v/cnt[n]/1e3:8.1f} us")
PY
tail -c 300 $OUT/${TAG}_tl.log
