import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['GFS_GICP_TILE_STATS'] = '1'
import numpy as np
from geoflowslam_amd import api, synth
reg = api.RegistrationGICP(max_points=20480)
for seed in (1000,1001,1005):
    fp = synth.frame_pair(seed)
    r = reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"])
    print(seed, "iters", r["iterations"], "tile stats", reg.tile_stats().tolist())
