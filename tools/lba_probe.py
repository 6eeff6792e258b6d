"""Latency of one LocalBundleAdjustment window (BASELINE.json configs[4]) through gfs_lba_solve, and of 64 windows batched."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geoflowslam_amd import api, synth
w = synth.lba_window(5, n_free=20, n_fixed=5, n_points=3000)
opt = api.Optimizer(max_poses=32, max_points=4096, max_edges=65536)
for _ in range(5): r = opt.LocalBundleAdjustment(w)
ts = []
for _ in range(40):
    t = time.perf_counter(); r = opt.LocalBundleAdjustment(w); ts.append(time.perf_counter() - t)
print("one window: median %.3f ms, min %.3f ms, iterations %d" % (np.median(ts) * 1e3, np.min(ts) * 1e3, r["iterations_run"]))
if hasattr(api, "BatchOptimizer"):
    wins = [synth.lba_window(5 + k % 16, n_free=20, n_fixed=5, n_points=3000) for k in range(64)]
    bat = api.BatchOptimizer(64, max_poses=32, max_points=4096, max_edges=65536)
    for _ in range(2): bat.LocalBundleAdjustment(wins)
    ts = []
    for _ in range(6):
        t = time.perf_counter(); bat.LocalBundleAdjustment(wins); ts.append(time.perf_counter() - t)
    print("64 windows batched: median %.2f ms = %.0f windows/s" % (np.median(ts) * 1e3, 64 / np.median(ts)))
    one = api.BatchOptimizer(1, max_poses=32, max_points=4096, max_edges=65536)
    for _ in range(5): rb = one.LocalBundleAdjustment([w])
    ts = []
    for _ in range(40):
        t = time.perf_counter(); rb = one.LocalBundleAdjustment([w]); ts.append(time.perf_counter() - t)
    same = np.array_equal(rb[0]["poses"], r["poses"]) and np.array_equal(rb[0]["points"], r["points"]) if "poses" in r else None
    print("one window through the batched entry: median %.3f ms, min %.3f ms, same bits as gfs_lba_solve: %s" % (np.median(ts) * 1e3, np.min(ts) * 1e3, same))
