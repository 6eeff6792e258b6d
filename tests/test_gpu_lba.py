"""GPU parity tests for LocalBundleAdjustment (MI355X): normal-equation blocks within 1e-10 relative of the CPU
oracle, optimised poses / points within 1e-5 relative Frobenius (BASELINE.json north_star tolerance), identical
outlier classification, LM iteration counts equal."""
import numpy as np
import pytest

from geoflowslam_amd import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def test_linearize_blocks_match_oracle(gpu_api, oracle):
    w = synth.lba_window(0, n_free=20, n_fixed=5, n_points=3000)  # BASELINE.json configs[4]
    opt = gpu_api.Optimizer()
    L = opt.linearize(w)
    Lo = oracle.lba_linearize(w)
    for k in ("Hpp", "Hll", "Hpl", "bp", "bl", "edge_chi2"):
        assert _rel(L[k], Lo[k]) < 1e-10, (k, _rel(L[k], Lo[k]))
    assert abs(L["chi2"] - Lo["chi2"]) < 1e-9 * Lo["chi2"]


@pytest.mark.parametrize("cfg", [dict(seed=0, n_free=20, n_fixed=5, n_points=3000),
                                 dict(seed=1, n_free=8, n_fixed=3, n_points=600),
                                 dict(seed=2, n_free=3, n_fixed=1, n_points=80, mono_frac=1.0),
                                 dict(seed=3, n_free=30, n_fixed=2, n_points=400),
                                 dict(seed=5, n_free=31, n_fixed=2, n_points=500),    # first size factored in HBM instead of LDS
                                 dict(seed=6, n_free=48, n_fixed=6, n_points=1500),
                                 dict(seed=7, n_free=80, n_fixed=4, n_points=800)])
def test_solve_matches_oracle(gpu_api, oracle, cfg):
    w = synth.lba_window(**cfg)
    opt = gpu_api.Optimizer(max_poses=96, max_points=4096, max_edges=200000)
    r = opt.LocalBundleAdjustment(w)
    ro = oracle.lba_solve(w)
    assert r["iterations_run"] == ro["iterations_run"]
    for i in range(w["n_poses"]):
        assert _rel(r["pose_q"][i], ro["pose_q"][i]) < 1e-5 and _rel(r["pose_t"][i], ro["pose_t"][i]) < 1e-5
    assert _rel(r["points"], ro["points"]) < 1e-5
    assert _rel(r["final_chi2"], ro["final_chi2"]) < 1e-6
    thr = np.where(w["edge_stereo"] == 1, 7.815, 5.991)
    near = np.abs(ro["edge_chi2"] - thr) < 1e-4 * thr
    assert ((r["edge_chi2"] > thr) == (ro["edge_chi2"] > thr))[~near].all()
    assert (r["edge_depth_positive"] == ro["edge_depth_positive"]).all()


def test_edge_cases(gpu_api, oracle):
    opt = gpu_api.Optimizer()
    w = synth.lba_window(4, n_free=3, n_fixed=2, n_points=50)
    # shuffled (not landmark-major) edge order gives the same answer
    perm = np.random.default_rng(0).permutation(w["n_edges"])
    ws = dict(w)
    for k in ("edge_pose", "edge_point", "edge_obs", "edge_inv_sigma2", "edge_stereo"):
        ws[k] = w[k][perm]
    r, rs = opt.LocalBundleAdjustment(w), opt.LocalBundleAdjustment(ws)
    assert _rel(rs["points"], r["points"]) < 1e-9 and _rel(rs["edge_chi2"], r["edge_chi2"][perm]) < 1e-6
    # zero iterations: estimates untouched
    w0 = dict(w); w0["iterations"] = 0
    r0 = opt.LocalBundleAdjustment(w0)
    assert r0["iterations_run"] == 0 and np.allclose(r0["points"], w["points"])
    # every pose fixed: only landmarks move
    wf = dict(w); wf["pose_fixed"] = np.ones_like(w["pose_fixed"])
    rf, of = opt.LocalBundleAdjustment(wf), oracle.lba_solve(wf)
    assert (rf["pose_t"] == w["pose_t"]).all() and _rel(rf["points"], of["points"]) < 1e-5
    # stop flag already raised: the reference returns before optimising (src/Optimizer.cc:1955-1956)
    assert opt.LocalBundleAdjustment(w, stop_flag=np.ones(1, np.int32)) is None
    # stop flag present but not raised: normal result
    rn = opt.LocalBundleAdjustment(w, stop_flag=np.zeros(1, np.int32))
    assert _rel(rn["points"], r["points"]) == 0


def test_stop_flag_raised_while_solving(gpu_api, oracle):
    """setForceStopFlag semantics (src/Optimizer.cc:1679): raising the flag during optimize() ends it at the next check
    (top of an iteration / end of an LM trial); the estimate is then a state some earlier iteration reached."""
    import threading
    import time
    w = synth.lba_window(0, n_free=20, n_fixed=5, n_points=3000)
    opt = gpu_api.Optimizer()
    full = opt.LocalBundleAdjustment(w)
    flag = np.zeros(1, np.int32)

    def raiser():
        time.sleep(0.0015)
        flag[0] = 1

    t = threading.Thread(target=raiser)
    t.start()
    r = opt.LocalBundleAdjustment(w, stop_flag=flag)
    t.join()
    assert r is not None and 0 <= r["iterations_run"] <= full["iterations_run"]
    assert np.isfinite(r["points"]).all() and np.isfinite(r["pose_t"]).all()
    if r["iterations_run"] < full["iterations_run"]:  # stopped early: the cost is not above the starting cost
        chi0 = oracle.lba_linearize(w)["chi2"]
        assert r["final_chi2"] <= chi0 * (1 + 1e-12)


def test_stopped_solve_is_exactly_the_state_after_its_last_iteration(gpu_api):
    """gfs_lba_solve's host loop runs one LM iteration ahead of the flags it has read (round 6).  When the caller's stop flag goes up while
    an iteration is running ahead, that iteration is discarded: the result must be, bit for bit, what optimize(k) leaves for the k
    iterations the stopped call reports -- estimates, per-edge chi2, final chi2, lambda.  The flag is raised at a sweep of delays so that
    it lands in different iterations (and some calls see it only after they are done)."""
    import threading
    import time
    w = synth.lba_window(0, n_free=20, n_fixed=5, n_points=3000)
    opt = gpu_api.Optimizer()
    full = opt.LocalBundleAdjustment(w)
    by_iterations = {}
    seen = set()
    for delay_us in (400, 600, 800, 1000, 1200, 1400, 1600, 1800, 2000, 2300, 700, 1100, 1500, 1900):
        flag = np.zeros(1, np.int32)

        def raiser():
            time.sleep(delay_us * 1e-6)
            flag[0] = 1

        t = threading.Thread(target=raiser)
        t.start()
        r = opt.LocalBundleAdjustment(w, stop_flag=flag)
        t.join()
        assert r is not None
        k = r["iterations_run"]
        seen.add(k)
        if k not in by_iterations:
            wk = dict(w)
            wk["iterations"] = k
            by_iterations[k] = opt.LocalBundleAdjustment(wk)  # no flag: the loop runs ahead, nothing is discarded
        ref = by_iterations[k]
        if k == full["iterations_run"] and ref["iterations_run"] != k:
            continue  # (the full run ended by its own termination rule before `iterations`)
        if k == 0:  # the flag was up before the first iteration: nothing was evaluated (g2o computes no errors either), the estimates stand
            assert np.array_equal(r["points"], ref["points"]) and np.array_equal(r["pose_t"], ref["pose_t"]), delay_us
            continue
        for key in ("points", "pose_t", "pose_q", "edge_chi2"):
            assert np.array_equal(r[key], ref[key]), (delay_us, k, key)
        assert r["final_chi2"] == ref["final_chi2"] and r["final_lambda"] == ref["final_lambda"] and r["iterations_run"] == ref["iterations_run"], (delay_us, k)
    # (every stop with 0 < k < the full count went through the discard: the next iteration is always running ahead when the flags are read)
    assert any(0 < k < full["iterations_run"] for k in seen), seen


def test_stop_flag_raised_while_a_batch_is_solving(gpu_api, oracle):
    """the same for gfs_lba_solve_batch: every window closes its running iteration and returns what it has"""
    import threading
    import time
    wins = [synth.lba_window(60 + k, n_free=20, n_fixed=5, n_points=2500) for k in range(6)]
    bat = gpu_api.BatchOptimizer(max_windows=len(wins), max_poses=32, max_points=4096, max_edges=65536)
    full = bat.LocalBundleAdjustment(wins)
    flag = np.zeros(1, np.int32)

    def raiser():
        time.sleep(0.002)
        flag[0] = 1

    t = threading.Thread(target=raiser)
    t.start()
    got = bat.LocalBundleAdjustment(wins, stop_flag=flag)
    t.join()
    for w, r, f in zip(wins, got, full):
        assert 0 <= r["iterations_run"] <= f["iterations_run"]
        assert np.isfinite(r["points"]).all() and np.isfinite(r["pose_t"]).all() and np.isfinite(r["pose_q"]).all()
        if r["iterations_run"] < f["iterations_run"]:
            assert r["final_chi2"] <= oracle.lba_linearize(w)["chi2"] * (1 + 1e-12)
    # the handle is usable afterwards
    again = bat.LocalBundleAdjustment(wins)
    for a, f in zip(again, full):
        assert np.array_equal(a["points"], f["points"]) and a["iterations_run"] == f["iterations_run"]


def _with_second_camera_edges(w, seed, frac=0.15):
    """the window with a second edge between some (key-frame, point) pairs, the way a two-camera rig adds a right-camera
    observation of a point the left camera sees too (src/Optimizer.cc:1859-1925): a monocular observation ~0.7 px away"""
    rng = np.random.default_rng(seed)
    E = w["n_edges"]
    pick = np.sort(rng.choice(E, int(frac * E), replace=False))
    w2 = dict(w)
    obs2 = w["edge_obs"][pick].copy()
    obs2[:, :2] += rng.normal(0, 0.7, (len(pick), 2))
    obs2[:, 2] = 0
    # point-major order like the reference builds it: the second edge right after the first
    order = np.argsort(np.r_[np.arange(E), pick + 0.5], kind="stable")
    for k, extra in (("edge_pose", w["edge_pose"][pick]), ("edge_point", w["edge_point"][pick]), ("edge_obs", obs2),
                     ("edge_inv_sigma2", w["edge_inv_sigma2"][pick]), ("edge_stereo", np.zeros(len(pick), w["edge_stereo"].dtype))):
        w2[k] = np.ascontiguousarray(np.concatenate([w[k], extra])[order])
    w2["n_edges"] = E + len(pick)
    return w2


@pytest.mark.parametrize("cfg", [dict(seed=11, n_free=20, n_fixed=5, n_points=3000), dict(seed=12, n_free=6, n_fixed=2, n_points=300),
                                 dict(seed=13, n_free=30, n_fixed=3, n_points=500)])
def test_several_edges_between_one_pose_and_one_point(gpu_api, oracle, cfg):
    """g2o takes any number of edges between two vertices (their Hpl contributions share one block,
    Thirdparty/g2o/g2o/core/block_solver.hpp:143-295): blocks, solution and flags against the oracle; alone and in a batch."""
    w = _with_second_camera_edges(synth.lba_window(**cfg), cfg["seed"])
    opt = gpu_api.Optimizer(max_poses=64, max_points=4096, max_edges=100000)
    L, Lo = opt.linearize(w), oracle.lba_linearize(w)
    for k in ("Hpp", "Hll", "Hpl", "bp", "bl", "edge_chi2"):
        assert _rel(L[k], Lo[k]) < 1e-10, (k, _rel(L[k], Lo[k]))
    r, ro = opt.LocalBundleAdjustment(w), oracle.lba_solve(w)
    assert r["iterations_run"] == ro["iterations_run"]
    assert _rel(r["pose_q"], ro["pose_q"]) < 1e-5 and _rel(r["pose_t"], ro["pose_t"]) < 1e-5 and _rel(r["points"], ro["points"]) < 1e-5
    assert _rel(r["final_chi2"], ro["final_chi2"]) < 1e-6
    assert (r["edge_depth_positive"] == ro["edge_depth_positive"]).all()
    bat = gpu_api.BatchOptimizer(max_windows=2, max_poses=64, max_points=4096, max_edges=100000)
    rb = bat.LocalBundleAdjustment([w, synth.lba_window(1, n_free=8, n_fixed=3, n_points=600)])[0]
    for k in ("pose_q", "pose_t", "points", "edge_chi2"):
        assert np.array_equal(rb[k], r[k]), k


def test_lba_batch_with_degenerate_windows(gpu_api, oracle):
    """A batch mixing ordinary windows with the degenerate ones: every pose fixed (only landmarks move), zero iterations (estimates
    untouched), a single free pose, a tiny window, a window too wide for the LDS solve — each equal, bit for bit, to the same window
    solved alone."""
    base = synth.lba_window(4, n_free=3, n_fixed=2, n_points=50)
    all_fixed = dict(base, pose_fixed=np.ones_like(base["pose_fixed"]))
    zero_it = dict(base, iterations=0)
    wins = [synth.lba_window(31, n_free=12, n_fixed=3, n_points=800), all_fixed, zero_it, synth.lba_window(32, n_free=1, n_fixed=2, n_points=40),
            synth.lba_window(33, n_free=2, n_fixed=1, n_points=12), synth.lba_window(34, n_free=24, n_fixed=2, n_points=300),
            synth.lba_window(35, n_free=40, n_fixed=3, n_points=500)]  # (> 30 free poses: reduced system factored in HBM, not LDS)
    bat = gpu_api.BatchOptimizer(max_windows=len(wins), max_poses=48, max_points=1024, max_edges=65536)
    got = bat.LocalBundleAdjustment(wins)
    one = gpu_api.Optimizer(max_poses=48, max_points=1024, max_edges=65536)
    for w, r in zip(wins, got):
        r1 = one.LocalBundleAdjustment(w)
        for k in ("pose_q", "pose_t", "points", "edge_chi2", "edge_depth_positive"):
            assert np.array_equal(r[k], r1[k]), k
        assert r["iterations_run"] == r1["iterations_run"] and r["final_chi2"] == r1["final_chi2"] and r["final_lambda"] == r1["final_lambda"]
    assert got[2]["iterations_run"] == 0 and np.allclose(got[2]["points"], base["points"])
    # ... and a batch of nothing but zero-iteration windows still evaluates the errors
    only0 = bat.LocalBundleAdjustment([zero_it, zero_it])
    r1 = one.LocalBundleAdjustment(zero_it)
    for r in only0:
        assert r["iterations_run"] == 0 and np.array_equal(r["edge_chi2"], r1["edge_chi2"]) and r["final_chi2"] == r1["final_chi2"]
    assert (got[1]["pose_t"] == base["pose_t"]).all()
    of = oracle.lba_solve(all_fixed)
    assert _rel(got[1]["points"], of["points"]) < 1e-5


def test_lba_batched_windows_match_oracle_and_single(gpu_api, oracle):
    """gfs_lba_solve_batch: windows of different sizes (3 ... 20 free key-frames, 80 ... 3000 points, different iteration
    counts and numbers of rejected trials) solved together; every window against the oracle and, bit for bit, against the same
    window solved alone."""
    cfgs = [(1, 20, 5, 3000), (2, 4, 2, 120), (3, 3, 2, 80), (4, 12, 4, 1200), (5, 8, 3, 600), (6, 20, 5, 2500), (7, 5, 2, 300),
            (8, 16, 3, 2000)] * 2
    wins = [synth.lba_window(100 + k if k >= 8 else s, n_free=f, n_fixed=x, n_points=n) for k, (s, f, x, n) in enumerate(cfgs)]
    wins[3]["iterations"] = 4          # optimize(4) next to optimize(10)
    bat = gpu_api.BatchOptimizer(max_windows=len(wins), max_poses=32, max_points=4096, max_edges=65536)
    got = bat.LocalBundleAdjustment(wins)
    one = gpu_api.Optimizer(max_poses=32, max_points=4096, max_edges=65536)
    its = set()
    for w, r in zip(wins, got):
        ro = oracle.lba_solve(w)
        assert r["iterations_run"] == ro["iterations_run"]
        assert np.linalg.norm(r["points"] - ro["points"]) <= 1e-5 * np.linalg.norm(ro["points"])
        assert np.abs(r["pose_t"] - ro["pose_t"]).max() <= 1e-6
        assert np.array_equal(r["edge_depth_positive"], ro["edge_depth_positive"])
        r1 = one.LocalBundleAdjustment(w)
        for k in ("pose_q", "pose_t", "points", "edge_chi2", "edge_depth_positive"):
            assert np.array_equal(r[k], r1[k]), k
        assert r["iterations_run"] == r1["iterations_run"] and r["final_chi2"] == r1["final_chi2"] and r["final_lambda"] == r1["final_lambda"]
        its.add(r["iterations_run"])
    assert len(its) >= 2  # the windows really finish at different times
    # the stop flag raised before the call: nothing is optimised (src/Optimizer.cc:1955-1956)
    with pytest.raises(gpu_api.GfsError):
        bat.LocalBundleAdjustment(wins[:2], stop_flag=np.ones(1, np.int32))


def test_batch_mixing_one_block_and_several_block_schur_windows(gpu_api):
    """A batch that mixes windows whose reduced system is one 128 x 128 block of pose pairs (<= 21 free poses) with wider ones
    (several blocks): each window must be solved by the Schur-product instance gfs_lba_solve launches for it alone
    (k_lba_schur_mfma<true> / <false>), so the batch stays bit-identical to the single solves."""
    wins = [synth.lba_window(70 + i, n_free=nf, n_fixed=3, n_points=npt) for i, (nf, npt) in
            enumerate([(4, 300), (40, 900), (20, 800), (26, 700), (21, 500), (22, 500), (60, 1200), (3, 200)])]
    bat = gpu_api.BatchOptimizer(max_windows=8, max_poses=80, max_points=2048, max_edges=200000)
    opt = gpu_api.Optimizer(max_poses=80, max_points=2048, max_edges=200000)
    got = bat.LocalBundleAdjustment(wins)
    for w, r in zip(wins, got):
        r1 = opt.LocalBundleAdjustment(w)
        assert r["iterations_run"] == r1["iterations_run"]
        for key in ("pose_q", "pose_t", "points", "edge_chi2", "edge_depth_positive"):
            assert np.array_equal(r[key], r1[key]), (w["n_poses"], key)
