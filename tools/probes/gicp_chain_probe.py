"""Per-kernel time of ONE gfs_gicp_align_batch_device call (HIP events around every launch: the kernels run one after the other) for a few
batch sizes, and the wall time of the call without the events: what the chain of a small batch is made of."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from geoflowslam_amd import api, synth
from test_gpu_gms import _Hip
hip = _Hip()
SP = 20480
pairs = [synth.frame_pair(1000 + k, 640, 480, 4) for k in range(8)]
for B in [int(x) for x in (sys.argv[1:] or ["1", "32"])]:
    c0 = np.zeros((B, SP, 4), np.float32); c1 = np.zeros((B, SP, 4), np.float32); n0 = np.zeros(B, np.int32); n1 = np.zeros(B, np.int32)
    for b in range(B):
        p = pairs[b % len(pairs)]
        c0[b, :len(p["cloud0"])], c1[b, :len(p["cloud1"])], n0[b], n1[b] = p["cloud0"], p["cloud1"], len(p["cloud0"]), len(p["cloud1"])
    d = [hip.to_device(x) for x in (c0, n0, c1, n1)]
    reg = api.RegistrationGICP(max_points=SP, max_batch=B)
    for _ in range(5):
        reg.align_batch_device(d[0], d[1], d[2], d[3], B, SP, raw=True)
    t0 = time.perf_counter(); N = 30
    for _ in range(N):
        reg.align_batch_device(d[0], d[1], d[2], d[3], B, SP, raw=True)
    wall = (time.perf_counter() - t0) / N
    t0 = time.perf_counter()
    for _ in range(N):
        reg.align_next_batch_device(d[2], d[3], B, SP, raw=True)
    wall_next = (time.perf_counter() - t0) / N
    api.profile_reset(); api.profile_enable(True)
    for _ in range(10):
        reg.align_batch_device(d[0], d[1], d[2], d[3], B, SP, raw=True)
    rep = api.profile_report(); api.profile_enable(False)
    tot = sum(v[0] for v in rep.values()) / 10
    print(f"== B = {B}: wall {wall*1e3:.3f} ms a call (both clouds), {wall_next*1e3:.3f} ms streaming (one cloud); sum of kernels {tot:.3f} ms; coop {reg.coop_stats()}")
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0]):
        print(f"   {k:26s} {v[0]/10*1e3:8.1f} us/call {v[1]/10:5.1f} launches  avg {v[0]/max(v[1],1)*1e3:7.1f} us")
