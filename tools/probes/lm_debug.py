import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from geoflowslam_amd import api, synth
fp = synth.frame_pair(1)
reg = api.RegistrationGICP(max_points=20480)
for mi in (2, 3):
    cfg = api.gicp_default_config(); cfg.max_iterations = mi
    r = reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"], cfg=cfg)
    print("   max_it", mi, r["n_linearize"], r["n_error_evals"], r["num_inliers"], r["error"], float(r["T"][0, 3]), reg.coop_stats()["last_workgroups"])
