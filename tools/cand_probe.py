import numpy as np, sys
sys.path.insert(0,'/root/repo')
from geoflowslam_amd import api, synth
ext = api.ORBextractor(1000, 1.2, 8, 20, 7, max_rows=480, max_cols=640, max_batch=1)
for seed in (1000, 1001, 1100, 1300):
    p = synth.frame_pair(seed, 640, 480, 4)
    g = p["gray0"] if "gray0" in p else p[list(p.keys())[0]]
    ext(g)
    print(seed, [len(ext.candidates(l)[0]) for l in range(8)])
