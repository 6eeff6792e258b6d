"""CPU tests pinning the oracle's small_gicp restatement (no GPU).  Ideas follow the reference's vendored
small_gicp tests, whose data files are not vendored: brute-force kNN cross-check
(Thirdparty/small_gicp/src/test/kdtree_synthetic_test.cpp:97-135), noisy-init convergence
(registration_test.cpp:60-86), voxel-count equivalence with an exact voxel grid (downsampling_test.cpp)."""
import numpy as np
import pytest
from scipy.linalg import expm

from geoflowslam_amd import synth


def _hat(t):
    w, v = t[:3], t[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = v
    return M


def test_se3_exp_vs_expm(oracle):
    rng = np.random.default_rng(0)
    for scale in (1e-7, 1e-3, 0.3, 2.0):
        tw = rng.normal(size=6) * scale
        assert np.abs(oracle.se3_exp(tw) - expm(_hat(tw))).max() < 1e-12


def test_eig3_direct_vs_lapack(oracle):
    rng = np.random.default_rng(1)
    for _ in range(50):
        A = rng.normal(size=(3, 3))
        A = A @ A.T * rng.uniform(1e-6, 10)
        ev, V = oracle.eig3_direct(A)
        w, U = np.linalg.eigh(A)
        assert np.abs(ev - w).max() < 1e-9 * max(1, abs(w).max())
        assert np.abs(np.abs(V.T @ U) - np.eye(3)).max() < 1e-6
    ev, V = oracle.eig3_direct(np.eye(3) * 2.5)  # all eigenvalues equal -> identity vectors (Eigen)
    assert np.allclose(ev, 2.5) and np.allclose(V, np.eye(3))


@pytest.mark.parametrize("dist", ["uniform", "normal", "plane", "clusters"])
def test_knn_vs_bruteforce(oracle, dist):
    rng = np.random.default_rng(2)
    n = 3000
    if dist == "uniform":
        p = rng.uniform(-5, 5, (n, 3))
    elif dist == "normal":
        p = rng.normal(0, 2, (n, 3))
    elif dist == "plane":
        p = np.c_[rng.uniform(-5, 5, (n, 2)), rng.normal(0, 1e-3, n)]
    else:
        p = rng.normal(0, 0.05, (n, 3)) + rng.integers(-3, 4, (n, 3))
    pts = np.c_[p, np.ones(n)]
    q = np.c_[rng.uniform(-5, 5, (60, 3)), np.ones(60)]
    for k in (1, 10):
        idx, sq = oracle.knn(pts, q, k)
        d = ((pts[None, :, :3] - q[:, None, :3]) ** 2).sum(-1)
        order = np.argsort(d, 1)[:, :k]
        assert (idx == order).all()
        assert np.allclose(sq, np.take_along_axis(d, order, 1), rtol=0, atol=1e-12)


def test_voxel_downsample_counts(oracle):
    # exact voxel-grid cardinality, plus at most one extra point per 1024-block boundary (reference quirk)
    fp = synth.frame_pair(3, 320, 240, 2)
    cloud = fp["cloud0"]
    pts, covs, nrm = oracle.gicp_preprocess(cloud)
    vox = np.unique(np.floor(cloud[:, :3].astype(np.float64) / 0.02).astype(np.int64), axis=0)
    assert len(vox) <= len(pts) <= len(vox) + len(cloud) // 1024 + 1
    # every output point lies inside an occupied voxel; covariances are V diag(1e-3,1,1) V^T
    keys = set(map(tuple, vox.tolist()))
    assert all(tuple(k) in keys for k in np.floor(pts[:, :3] / 0.02).astype(np.int64).tolist())
    w = np.linalg.eigvalsh(covs[:, :3, :3])
    assert np.allclose(w, [1e-3, 1, 1], atol=1e-9)
    assert np.allclose(np.linalg.norm(nrm[:, :3], axis=1), 1, atol=1e-9)
    assert ((pts[:, :3] * nrm[:, :3]).sum(1) <= 1e-12).all()  # normals flipped toward the origin


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gicp_recovers_motion(oracle, seed):
    fp = synth.frame_pair(20 + seed)
    r = oracle.gicp_align(fp["cloud0"], fp["cloud1"])
    assert r["converged"] and r["num_inliers"] > 200  # Tracking::PredictStateICP gate (src/Tracking.cc:3394)
    dT = np.linalg.inv(fp["T_01"]) @ r["T"]
    ang = np.degrees(np.arccos(np.clip((np.trace(dT[:3, :3]) - 1) / 2, -1, 1)))
    assert ang < 0.1 and np.linalg.norm(dT[:3, 3]) < 5e-3
    # noisy-init convergence (registration_test.cpp:60-86): perturbed initial guesses reach the same optimum
    rng = np.random.default_rng(seed)
    tw = np.r_[rng.uniform(-1, 1, 3) * np.radians(1.0), rng.uniform(-0.02, 0.02, 3)]
    r2 = oracle.gicp_align(fp["cloud0"], fp["cloud1"], fp["T_01"] @ expm(_hat(tw)))
    assert np.linalg.norm(r2["T"] - r["T"]) < 2e-3


def test_gicp_result_fields(oracle):
    fp = synth.frame_pair(5, 320, 240, 2)
    r = oracle.gicp_align(fp["cloud0"], fp["cloud1"])
    assert np.allclose(r["H"], r["H"].T) and (np.linalg.eigvalsh(r["H"]) > 0).all()
    assert r["iterations"] == r["n_linearize"] - 1 and r["n_error_evals"] >= r["n_linearize"]
    assert abs(np.linalg.det(r["T"][:3, :3]) - 1) < 1e-9 and (r["T"][3] == [0, 0, 0, 1]).all()
    # identical clouds + identity init: converges immediately with zero motion
    r0 = oracle.gicp_align(fp["cloud0"], fp["cloud0"])
    assert r0["converged"] and np.allclose(r0["T"], np.eye(4), atol=1e-9) and r0["iterations"] == 0
    # far-apart clouds: no correspondences within 0.1 m -> zero inliers, pose unchanged
    far = fp["cloud0"].copy()
    far[:, 2] += 50
    rf = oracle.gicp_align(fp["cloud0"], far)
    assert rf["num_inliers"] == 0 and np.allclose(rf["T"], np.eye(4))
