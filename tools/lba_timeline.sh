#!/bin/bash
# kernel timeline of ONE LocalBundleAdjustment window (configs[4]): per kernel its duration and the gap in front of it
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/lba_one.py <<PY
import sys; sys.path.insert(0, "$R")
from geoflowslam_amd import api, synth
w = synth.lba_window(5, n_free=20, n_fixed=5, n_points=3000)
opt = api.Optimizer(max_poses=32, max_points=4096, max_edges=65536)
for _ in range(6): r = opt.LocalBundleAdjustment(w)
print("iterations", r["iterations_run"])
PY
rm -rf $OUT/lba_tl
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/lba_tl -- python /tmp/lba_one.py > $OUT/lba_tl.log 2>&1
python3 - "$OUT/lba_tl" <<'PY'
import csv, glob, sys, collections, re
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_lba[a-z_0-9]*)", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy_" + r.get("Direction", "")[:12]))
ev.sort()
# the last window: from the last k_lba_init on
starts = [i for i, e in enumerate(ev) if e[2] == "k_lba_init"]
i0 = starts[-1]
# include the H2D copy in front of it
while i0 > 0 and ev[i0 - 1][2].startswith("copy") and ev[i0][0] - ev[i0 - 1][1] < 200000: i0 -= 1
win = ev[i0:]
t0 = win[0][0]
dur = collections.Counter(); gap = collections.Counter(); cnt = collections.Counter()
prev_end = None
for s, e, n in win:
    dur[n] += e - s; cnt[n] += 1
    if prev_end is not None: gap[n] += max(0, s - prev_end)
    prev_end = e
print("window wall (first start -> last end): %.1f us, kernels+copies busy %.1f us, gaps %.1f us, events %d" % ((win[-1][1] - t0) / 1e3, sum(dur.values()) / 1e3, sum(gap.values()) / 1e3, len(win)))
for n in sorted(dur, key=lambda k: -(dur[k] + gap[k])):
    print("  %-26s n=%3d  busy %7.1f us (%.1f each)  gap in front %7.1f us (%.1f each)" % (n, cnt[n], dur[n] / 1e3, dur[n] / 1e3 / cnt[n], gap[n] / 1e3, gap[n] / 1e3 / cnt[n]))
PY
