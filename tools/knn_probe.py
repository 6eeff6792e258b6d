"""How many queries of a synthetic VGA / 720p cloud go to the deferred k-NN passes, and how large the isolated ones' probes are."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geoflowslam_amd import api, synth
for (w, h, st, cap) in ((640, 480, 4, 20480), (1280, 720, 5, 36864)):
    reg = api.RegistrationGICP(max_points=cap)
    for seed in (1000, 1001, 1002, 1003):
        fp = synth.frame_pair(seed, w, h, st)
        reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"])
        for which in (0, 1):
            out, dk = reg.knn_stats(0, which)
            r = np.ceil(np.sqrt(np.minimum(dk, 1e6)) / 0.1)
            print(w, seed, which, "points", out[0], "far", out[1], "far2", out[2], "rings", np.sort(r)[::-1][:12].astype(int).tolist(), "unbounded", int((dk > 1e300).sum()))
