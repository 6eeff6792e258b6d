import sys; sys.path.insert(0, "/root/repo")
from geoflowslam_amd import api, synth
pf = synth.pose_frame(3, n_obs=330)
po = api.PoseOptimizer(max_obs=1024, max_batch=1)
for _ in range(3): r = po.PoseOptimization(pf)
print(r["iterations_run"], r["n_inliers"])
