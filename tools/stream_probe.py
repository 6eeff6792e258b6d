"""Per-stage wall time and kernel time of the single-stream chain (bench_stream.py) on one synthetic VGA pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from geoflowslam_amd import api, synth
import bench_stream as bs
W, H = 640, 480
fp = synth.frame_pair(1000, W, H, 4)
frames = [(fp["gray0"], fp["depth0"]), (fp["gray1"], fp["depth1"])]
K = synth.intrinsics(W, H)
be = bs.GpuBackend(api, W, H, 1000, 8, 20480)
lat, st, states = bs.run_stream(be, frames, K, W, H, 4, 40, warm=6)
print(bs.summarize(lat, st))
api.profile_reset(); api.profile_enable(True)
lat, st, states = bs.run_stream(be, frames, K, W, H, 4, 20, warm=2)
rep = api.profile_report(); api.profile_enable(False)
n = 20 + 2 + 1
for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"{k:28s} {v[0]/n*1e3:8.1f} us/frame  {v[1]/n:6.1f} launches/frame")
