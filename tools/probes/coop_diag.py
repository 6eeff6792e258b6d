"""Which pairs of the ragged 128-pair batch differ between the cooperative LM kernel and the launch-per-step rounds, and where."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from geoflowslam_amd import api
from test_gpu_batched import _ragged_cloud_pairs, _same
from test_gpu_gms import _Hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
triples = _ragged_cloud_pairs(B, 160, 120, 3000)
SP = (max(max(len(a), len(b)) for a, b, _ in triples) + 1023) // 1024 * 1024
os.environ["GFS_GICP_COOP"] = "0"
r1 = api.RegistrationGICP(max_points=SP, max_batch=1)
rB = api.RegistrationGICP(max_points=SP, max_batch=B)
del os.environ["GFS_GICP_COOP"]
c1 = api.RegistrationGICP(max_points=SP, max_batch=1)
cB = api.RegistrationGICP(max_points=SP, max_batch=B)
c0 = np.zeros((B, SP, 4), np.float32); cc1 = np.zeros((B, SP, 4), np.float32)
n0 = np.zeros(B, np.int32); n1 = np.zeros(B, np.int32); init = np.zeros((B, 4, 4))
for b, (a, s, T0) in enumerate(triples):
    c0[b, :len(a)], cc1[b, :len(s)], n0[b], n1[b], init[b] = a, s, len(a), len(s), T0
hip = _Hip()
d = [hip.to_device(x) for x in (c0, n0, cc1, n1)]
for rep in range(3):
    gr = rB.align_batch_device(d[0], d[1], d[2], d[3], B, SP, init_T=init)
    gc = cB.align_batch_device(d[0], d[1], d[2], d[3], B, SP, init_T=init)
    bad = [b for b in range(B) if not _same(gr[b], gc[b])]
    print("rep", rep, "batch rounds vs batch coop: differing pairs", bad, cB.coop_stats())
    for b in bad[:4]:
        print("  ", b, "n_src", gr[b]["n_source_ds"], "iters", gr[b]["iterations"], gc[b]["iterations"], "n_err", gr[b]["n_error_evals"], gc[b]["n_error_evals"],
              "dT", float(np.abs(gr[b]["T"] - gc[b]["T"]).max()), "err", gr[b]["error"], gc[b]["error"])
bad1 = []
for b, (a, s, T0) in enumerate(triples):
    x, y = r1.RegisterPointClouds(a, s, T0), c1.RegisterPointClouds(a, s, T0)
    if not _same(x, y):
        bad1.append(b)
        print("single", b, "n_src", x["n_source_ds"], "iters", x["iterations"], y["iterations"], "n_err", x["n_error_evals"], y["n_error_evals"],
              "dT", float(np.abs(x["T"] - y["T"]).max()), "err", x["error"], y["error"])
    if not _same(x, gr[b]):
        print("rounds single != rounds batch", b)
print("single rounds vs single coop: differing", bad1, c1.coop_stats())
