"""Variant build -DGFS_LIN_UTIL (tools/variant.sh build util gicp -DGFS_LIN_UTIL): what the waves of the linearisation's 1-NN walk do.
slots of gfs_gicp_tile_stats: [2] lanes that take a neighbouring row, summed over the rounds of rows; [3] rounds of rows (per wave);
[4] lanes at work, summed over the steps of the walks; [5] steps (per wave); [6] searches (lanes); [7] waves that search."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['GFS_GICP_TILE_STATS'] = '1'
os.environ['GFS_GICP_COOP'] = '0'
import numpy as np
from geoflowslam_amd import api, synth
reg = api.RegistrationGICP(max_points=20480)
for seed in (1000, 1001, 1005):
    fp = synth.frame_pair(seed)
    r = reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"])
    ts = [int(x) for x in reg.tile_stats().tolist()]
    print(seed, "n_lin", r["n_linearize"], "src pts", r["n_source_ds"], ts)
    if ts[7]:
        print(f"   searches {ts[6]} in {ts[7]} waves ({ts[6]/ts[7]:.1f} lanes a wave); steps of the walk a wave {ts[5]/ts[7]:.1f}, lanes at work a step "
              f"{ts[4]/max(ts[5],1):.1f} of 64 = {ts[4]/max(ts[5],1)/64:.2f}; candidates a search {4*ts[4]/ts[6]:.1f}; rounds of neighbouring rows a wave "
              f"{ts[3]/ts[7]:.1f}, lanes with a row in a round {ts[2]/max(ts[3],1):.1f}")
