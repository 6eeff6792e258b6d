"""SURVEY.md section 8(f) rank 4 on the DEVICE: the HIP optical-flow tracker and the HIP F check against the oracle's variants that
carry OpenCV's own arithmetic — the scalar float accumulation of `LKTrackerInvoker` (oracle/klt_oracle.cpp,
gfso_klt_set_accumulation(1); a four-lane model of the vectorised builds = 2) and the Jacobi-SVD + solveCubic internals of the
7-point solver (oracle/fmat_oracle.cpp, gfso_fmat_set_solver(1)).  tests/test_f4_reference_arithmetic.py bounds exact-sum oracle vs
OpenCV-order oracle on the CPU and tests/test_gpu_klt.py / test_gpu_fmat.py prove HIP == exact-sum oracle; this module asserts the
composed statement directly: HIP within the same tolerances of OpenCV's order.  Reference call sites: src/ORBmatcher.cc:2186-2297
(fbKltTracking), :2399-2405 (findFundamentalMat)."""
import numpy as np
import pytest

from geoflowslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [1, 2], ids=["opencv_scalar_float_loop", "four_lane_float_model"])
def test_hip_fb_klt_tracking_against_opencvs_float_accumulation(gpu_api, oracle, mode):
    """gfs_klt_fb_track vs the oracle run with OpenCV's float accumulation order: the tolerances of
    tests/test_f4_reference_arithmetic.py:92-98 (status flips <= max(2, points / 5000); max 0.05 px, p99 1e-3 px, median 1e-4 px)."""
    points, flips, deltas = 0, [], []
    ext = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7, max_rows=480, max_cols=640)
    for seed in range(4):
        fp = synth.frame_pair(seed, 640, 480)
        _, k, _ = ext(fp["gray0"])
        kps = np.stack([k["x"], k["y"]], 1).astype(np.float32)
        for win in (15, 21, 35):
            trk = gpu_api.KltTracker(640, 480, win, max_batch=1, max_points=2048)
            g0, g1 = trk.buildOpticalFlowPyramid(fp["gray0"]), trk.buildOpticalFlowPyramid(fp["gray1"])
            pri, ok, _ = trk.fbKltTracking(g0, g1, 3, 15.0, 0.5, kps, kps.copy())
            o0, o1 = oracle.klt_build_pyramid(fp["gray0"], win), oracle.klt_build_pyramid(fp["gray1"], win)
            try:
                oracle.klt_set_accumulation(mode)
                prio, oko, _ = oracle.fb_klt_tracking(o0, o1, 640, 480, win, 3, 15.0, 0.5, kps, kps.copy())
            finally:
                oracle.klt_set_accumulation(0)
            points += len(kps)
            flips += [(seed, win, int(i)) for i in np.nonzero(ok != oko)[0]]
            both = np.nonzero(ok & oko)[0]
            deltas.append(np.abs(pri[both] - prio[both]).max(1))
    d = np.concatenate(deltas)
    assert points > 9000
    assert len(flips) <= max(2, points // 5000), flips
    assert d.max() <= 0.05 and np.quantile(d, 0.99) <= 1e-3 and np.median(d) <= 1e-4, (d.max(), np.quantile(d, 0.99), np.median(d))


def test_hip_find_fundamental_ransac_against_opencvs_solver_internals(gpu_api, oracle):
    """gfs_find_fundamental_ransac vs the oracle run with OpenCV's Jacobi SVD + solveCubic: same iteration budget consequences —
    consensus SIZES equal in every RANSAC problem, the sets differ only on ties between two models of one subset
    (tests/test_f4_reference_arithmetic.py:101-112)."""
    fm = gpu_api.FundamentalMatcher(max_points=2048, max_batch=1)
    cases, differs = 0, []
    for i in range(240):
        rng = np.random.default_rng([5, i])
        n = int(rng.choice([15, 30, 100, 400, 1000, 1500]))
        p1, p2 = synth.two_view_points(int(rng.integers(0, 1 << 30)), n=n, outlier_frac=float(rng.uniform(0.05, 0.7)),
                                       noise=float(rng.uniform(0.05, 1.0)))[:2]
        thr = float(rng.choice([0.5, 1.0, 3.0]))
        mask, _, cnt = fm.findFundamentalMat(p1, p2, thr, 0.99)
        try:
            oracle.fmat_set_solver(1)
            mo, _, co, _ = oracle.fundamental_ransac(p1, p2, thr, 0.99)
        finally:
            oracle.fmat_set_solver(0)
        cases += 1
        assert cnt == co, (i, n, thr, cnt, co)
        if not np.array_equal(mask, mo):
            differs.append(dict(case=i, n=n, threshold=thr, differing_flags=int((mask != mo).sum())))
    assert len(differs) <= max(1, cases // 200), differs
