"""VERDICT r3, item 7(b) -- a CPU experiment (oracle only, no GPU): at an exact distance tie between the k-th and the (k+1)-th
neighbour, which candidate does the oracle's KdTree (ann/kdtree.hpp:194-233 visiting order, ann/knn_result.hpp:80-101 keeps the
first one pushed) keep -- the lowest index of the down-sampled cloud ("lowest original voxel index wins"), or the first one in the
device's cell order?  Neither: round 4 ran 12 000 noise-free raster clouds (173 M points, 78 tie points where a choice had to be
made): the KdTree's pick is the lowest voxel index in 38 of them (49 %), the first in cell order in 41 (53 %).  Its visiting order
follows the split planes, not any index; a different tie rule on the device would not remove the documented deviation.
    python tools/tie_rule_experiment.py <number of cloud pairs>"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from geoflowslam_amd import synth
from oracle import oracle as O
n_pts = n_tie = low = cellf = 0
N = int(sys.argv[1])
for seed in range(N):
    rng = np.random.default_rng([4321, 2, seed])
    sd = int(rng.integers(0, 1 << 30))
    w, h = int(rng.choice([96, 128, 160, 200])), int(rng.choice([72, 96, 120, 150]))
    c0, c1, _ = synth.cloud_pair(sd, w, h, trans=float(rng.uniform(0.0, 0.15)), rot_deg=float(rng.uniform(0, 5)))
    for cl in (c0, c1):
        po, _, _ = O.gicp_preprocess(cl)
        n = len(po); n_pts += n
        idx, sq = O.knn(po, po, 16)
        tied = np.nonzero(sq[:, 9] == sq[:, 10])[0]
        if not len(tied): continue
        u = po[:, :3] * 10.0
        cc = np.floor(u).astype(np.int64)
        sub = np.clip(((u[:, 0] - cc[:, 0]) * 16).astype(np.int64), 0, 15)
        keyc = ((cc[:, 2] + (1 << 18)) << 45) | ((cc[:, 1] + (1 << 19)) << 25) | (cc[:, 0] * 16 + sub + (1 << 24))
        rank = np.empty(n, np.int64); rank[np.argsort(keyc, kind="stable")] = np.arange(n)
        for t in tied:
            kd = sq[t, 9]
            cand = idx[t][sq[t] == kd]; inside = idx[t][sq[t] < kd]
            slots = 10 - len(inside)
            if sq[t, 15] == kd: continue   # more tied candidates than looked at
            n_tie += 1
            chosen = set(idx[t][:10]) - set(inside)   # the first 10 of the k=16 search = what k=10 keeps? (checked below)
            i10, _ = O.knn(po, po[t:t+1], 10)
            chosen = set(i10[0]) - set(inside)
            low += set(sorted(cand)[:slots]) == chosen
            cellf += set(sorted(cand, key=lambda j: rank[j])[:slots]) == chosen
print(dict(clouds=2 * N, points=n_pts, tie_points=n_tie, oracle_keeps_lowest_voxel_index=low, oracle_keeps_first_in_cell_order=cellf))
