"""Where a single-stream frame's wall time goes: inside the C ABI calls (ctypes call to return) vs the Python around them."""
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
L = api.lib()
acc = {}
def wrap(name):
    f = getattr(L, name)
    def g(*a):
        t = time.perf_counter(); r = f(*a); acc.setdefault(name, []).append(time.perf_counter() - t); return r
    return g
class Proxy:
    def __init__(self, L): self._L = L; self._c = {}
    def __getattr__(self, n):
        if n not in self._c: self._c[n] = wrap(n)
        return self._c[n]
api._lib = Proxy(L)
lat, st, states = bs.run_stream(be, frames, K, W, H, 4, 40, warm=6)
print(bs.summarize(lat, st))
tot = 0
for k, v in sorted(acc.items(), key=lambda kv: -np.median(kv[1]) * len(kv[1])):
    per_frame = np.median(v) * len(v) / 47
    tot += per_frame
    print(f"{k:40s} median {np.median(v)*1e6:8.1f} us  calls/frame {len(v)/47:5.2f}")
print("inside the C ABI per frame (us):", round(tot * 1e6, 1))
