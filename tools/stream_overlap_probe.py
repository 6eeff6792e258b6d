"""Single stream: the sequential chain and the chain with the ORB extraction beside cloud + registration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geoflowslam_amd import api, synth
import bench_stream as bs
W, H = 640, 480
fp = synth.frame_pair(1000, W, H, 4)
frames = [(fp["gray0"], fp["depth0"]), (fp["gray1"], fp["depth1"])]
K = synth.intrinsics(W, H)
be = bs.GpuBackend(api, W, H, 1000, 8, 20480)
la, sa, xa = bs.run_stream(be, frames, K, W, H, 4, 60, warm=6)
lb, sb, xb = bs.run_stream(be, frames, K, W, H, 4, 60, warm=6, overlap=True)
print("sequential", bs.summarize(la, sa))
print("overlapped", bs.summarize(lb, sb))
print("same results", all(a["matches"] == b["matches"] and a["inliers"] == b["inliers"] and np.array_equal(a["T"], b["T"]) and np.array_equal(a["match"], b["match"]) for a, b in zip(xa[-40:], xb[-40:])))
