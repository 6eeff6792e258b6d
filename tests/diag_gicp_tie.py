"""Replays GICP case `index` of tests/fuzz_gpu.py (seed `seed`) on the GPU and on the oracle, lists the points whose 10th and 11th
neighbours are exactly equidistant, moves the raw point(s) behind every such 11th neighbour by 20 micrometres and replays:
    python tests/diag_gicp_tie.py <seed> <index>
Round 2: fuzz seed 4, cases 72 and 3598: 1.5e-4 / 3.2e-5 as generated (one tie each, runs that stop at the iteration cap),
6e-17 / 9e-17 with the tie broken."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geoflowslam_amd import api, synth
from oracle import oracle as O
seed0, ci = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng([1234 + seed0, 2, ci])
s = int(rng.integers(0, 1 << 30))
w, h = int(rng.choice([96, 128, 160, 200])), int(rng.choice([72, 96, 120, 150]))
tr_, rd_ = float(rng.uniform(0.0, 0.15)), float(rng.uniform(0, 5))
c0, c1, T = synth.cloud_pair(s, w, h, trans=tr_, rot_deg=rd_)
if rng.integers(0, 4) == 0: c0 = c0[:int(len(c0) * rng.uniform(0.2, 1.0))]
if rng.integers(0, 4) == 0: c1 = c1[::int(rng.integers(1, 4))]
reg = api.RegistrationGICP(max_points=65536)
def run(c0, c1, tag):
    r = reg.RegisterPointClouds(c0, c1); ro = O.gicp_align(c0, c1)
    rel = np.linalg.norm(r["T"] - ro["T"]) / np.linalg.norm(ro["T"])
    print(tag, "rel", rel, "it", r["iterations"], ro["iterations"], "conv", r["converged"], ro["converged"], "inl", r["num_inliers"], ro["num_inliers"], flush=True)
run(c0, c1, "as is     ")
clouds = [c0.copy(), c1.copy()]
for which in (0, 1):
    po, co, _ = O.gicp_preprocess(clouds[which])
    idx, sq = O.knn(po, po, 11)
    tie = np.nonzero(sq[:, 9] == sq[:, 10])[0]
    print("cloud", which, "points", len(po), "tie points", tie.tolist())
    for t in tie:
        j = int(idx[t, 10])  # the 11th neighbour: move the raw points of its voxel by 20 micrometres
        key = np.floor(po[j, :3] / 0.02).astype(np.int64)
        raw = clouds[which]
        sel = np.nonzero((np.floor(raw[:, :3].astype(np.float64) / 0.02).astype(np.int64) == key).all(1))[0]
        print("  tie at point", int(t), "11th neighbour", j, "raw points in its voxel", sel.tolist())
        raw[sel, 0] += np.float32(2e-5)
run(clouds[0], clouds[1], "ties moved")
