"""Variant build -DGFS_KNN_UTIL (tools/variant.sh build knnutil gicp -DGFS_KNN_UTIL): what the waves of k_knn_cov's first pass do.
slots of gfs_gicp_tile_stats: [0] lanes at work summed over the steps of the own-row walk, [1] those steps (per wave); [2], [3] the same
for the two scans of the neighbouring rows; [4] queries, [5] waves; [6] points in the queries' own rows (what a full scan would visit)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['GFS_GICP_TILE_STATS'] = '1'
import numpy as np
from geoflowslam_amd import api, synth
reg = api.RegistrationGICP(max_points=20480)
for seed in (1000, 1001, 1005):
    fp = synth.frame_pair(seed)
    r = reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"])
    ts = [int(x) for x in reg.tile_stats().tolist()]
    print(seed, ts)
    if ts[5]:
        print(f"   {ts[4]} queries in {ts[5]} waves; own row: {ts[1]/ts[5]:.1f} steps a wave, {ts[0]/max(ts[1],1):.1f} lanes at work = {4*ts[0]/ts[4]:.1f} candidates a query "
              f"(its row holds {ts[6]/ts[4]:.1f}); neighbouring rows: {ts[3]/ts[5]:.1f} steps a wave, {ts[2]/max(ts[3],1):.1f} lanes at work = {4*ts[2]/ts[4]:.1f} candidates a query")
