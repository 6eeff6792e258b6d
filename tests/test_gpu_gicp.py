"""GPU parity tests for GICP (MI355X).  Floating point: T_target_source within 1e-5 relative Frobenius of the
CPU oracle (BASELINE.json north_star tolerance); converged / num_inliers equal (inliers +-0.1 % where the
reference's own voxel-split ordering is nondeterministic)."""
import numpy as np
import pytest

from geoflowslam_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _voxel_keys(x, y, z):
    return (x.astype(np.uint64) | (y.astype(np.uint64) << np.uint64(21)) | (z.astype(np.uint64) << np.uint64(42)))


def _sort_cases():
    rng = np.random.default_rng(7)
    cases = []
    for n in (0, 1, 2, 3, 16, 17, 18, 64, 65, 500, 1023, 1024, 1025, 2047, 2048, 5000, 19200, 36864, 40000):
        for mode in range(6):
            if mode == 0:      # depth-camera like: most voxels unique, ~10 % shared
                x, y, z = rng.integers(0, 200, n), rng.integers(0, 150, n), rng.integers(0, 4, n)
            elif mode == 1:    # heavy ties
                x, y, z = rng.integers(0, 5, n), rng.integers(0, 3, n), np.zeros(n, np.int64)
            elif mode == 2:    # already sorted with ties
                x, y, z = np.arange(n) // 3 % 1024, np.arange(n) // 3 // 1024, np.zeros(n, np.int64)
            elif mode == 3:    # reversed
                x, y, z = (n - np.arange(n)) // 2 % 512, (n - np.arange(n)) // 2 // 512, np.zeros(n, np.int64)
            elif mode == 4:    # wide extent: the compacted keys need more than 31 bits (64-bit leaf path)
                x, y, z = rng.integers(0, 1 << 15, n), rng.integers(0, 1 << 12, n), rng.integers(0, 1 << 10, n)
                if n > 4:
                    x[:n // 3] = x[n // 3:2 * (n // 3)]
                    y[:n // 3] = y[n // 3:2 * (n // 3)]
                    z[:n // 3] = z[n // 3:2 * (n // 3)]
            else:              # all equal
                x, y, z = np.full(n, 7), np.full(n, 9), np.full(n, 11)
            k = _voxel_keys(np.asarray(x) + 1000, np.asarray(y) + 2000, np.asarray(z) + 3000)
            if mode in (0, 4) and n > 10:  # some out-of-range points: key = all ones
                k[rng.integers(0, n, max(1, n // 50))] = np.uint64(0xFFFFFFFFFFFFFFFF)
            cases.append((f"n{n}-m{mode}", k))
    return cases


# handle capacities that select the three kernels of the n >= 1024 levels (csrc/gicp.hip voxel_qsort_top): the cloud resident in LDS,
# the keys in LDS with the point indices in global memory (720p clouds), the ping-pong kernel through HBM
SORT_CAPS = [19456, 37888, 40960]


@pytest.mark.parametrize("cap", SORT_CAPS)
def test_voxel_sort_permutation_is_the_references(gpu_api, oracle, cap):
    """The preprocessing's voxel sort must produce small_gicp quick_sort_omp's permutation (util/sort_omp.hpp:58-85; the sort is
    not stable and the 1024-block splits of voxelgrid_sampling_omp depend on it): 3-way quicksort levels, libstdc++ introsort
    leaves, final insertion sort — and the heap-sort fallback, reached with McIlroy's adversarial input for std::sort."""
    reg = gpu_api.RegistrationGICP(max_points=cap)
    cases = [c for c in _sort_cases() if len(c[1]) <= cap]
    for n in (100, 700, 1000, 1023):
        a = oracle.antiqsort_keys(n).astype(np.int64)
        for div in (1, 2, 3):
            cases.append((f"antiqsort{n}/{div}", _voxel_keys(a // div + 5, np.full(n, 3), np.full(n, 1))))
    # leaves inside a big sort: the adversarial block sits between two key ranges of a 20000-element cloud
    a = oracle.antiqsort_keys(1000).astype(np.int64) // 2
    rng = np.random.default_rng(1)
    lo, hi = rng.integers(0, 1000, 9000), rng.integers(3000, 4000, 10000)
    mix = np.concatenate([lo, a + 1500, hi])
    cases.append(("antiqsort-embedded", _voxel_keys(mix[rng.permutation(len(mix))], np.full(len(mix), 2), np.full(len(mix), 2))))
    cases.append(("antiqsort-embedded-inorder", _voxel_keys(mix, np.full(len(mix), 2), np.full(len(mix), 2))))
    for name, k in cases:
        if len(k) > cap:
            continue
        got = reg.voxel_sort_perm(k)
        want, _ = oracle.quick_sort_perm(k)
        assert np.array_equal(got, want), (name, int((got != want).sum()), len(k))


@pytest.mark.parametrize("cap", SORT_CAPS)
def test_voxel_sort_permutation_against_the_compiled_reference(gpu_api, oracle, cap):
    """The same, against the REFERENCE'S OWN quick_sort_omp (util/sort_omp.hpp compiled from /root/reference into oracle/_ref by
    oracle/ref_build.sh; the prebuilt library travels to the GPU box) instead of the restatement."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    from test_oracle_ref import sort_cases
    reg = gpu_api.RegistrationGICP(max_points=cap)
    for name, k in [c for c in sort_cases() if len(c[1]) <= cap]:
        got = reg.voxel_sort_perm(k)
        want, _ = oracle.ref_quick_sort_perm(k, 4)
        assert np.array_equal(got, want), (name, int((got != want).sum()), len(k))


def test_wave_std_sort_is_libstdcxx_std_sort(gpu_api, oracle):
    """csrc/wave_std_sort.hpp (one wave, LDS) against std::sort itself — the oracle's quick_sort_omp hands ranges below 1024
    elements to std::sort (util/sort_omp.hpp:61) —: the permutation, not just the order, on keys with many ties, sorted / reversed
    input, and McIlroy's adversary (depth limit -> heap sort)."""
    rng = np.random.default_rng(11)
    cases = []
    for n in (0, 1, 2, 3, 15, 16, 17, 18, 33, 64, 65, 100, 127, 128, 129, 255, 300, 511, 700, 1000, 1023):
        cases.append(rng.integers(0, 1 << 30, n))            # distinct
        cases.append(rng.integers(0, max(n // 4, 1), n))     # ties
        cases.append(rng.integers(0, 3, n))                  # heavy ties
        cases.append(np.sort(rng.integers(0, max(n // 2, 1), n)))
        cases.append(np.sort(rng.integers(0, max(n // 2, 1), n))[::-1])
        cases.append(np.full(n, 5))
    for n in (100, 500, 1000, 1023):
        a = oracle.antiqsort_keys(n).astype(np.int64)
        cases += [a, a // 2, a // 5]
    for _ in range(200):  # the quadtree's lists: (size << 12 | x) with small sizes and a dozen x values
        n = int(rng.integers(2, 400))
        cases.append((rng.integers(2, 12, n) << 12) | (rng.integers(0, 10, n) * 37))
    for k in cases:
        k = np.asarray(k, np.uint32)
        got = gpu_api.wave_std_sort_perm(k)
        want, _ = oracle.quick_sort_perm(k.astype(np.uint64))
        assert np.array_equal(got, np.asarray(want, np.int64)), (len(k), int((got != want).sum()))


def test_preprocess_stage_matches_oracle(gpu_api, oracle):
    """Voxel means bit-identical to the (unmodified, reference-order) oracle — same permutation of equal voxel keys, same
    1024-block splits, same summation order — and covariances equal to 1e-9."""
    for seed, res in ((2, (640, 480, 4)), (202, (640, 480, 4)), (12, (1280, 720, 5))):
        fp = synth.frame_pair(seed, *res)
        reg = gpu_api.RegistrationGICP(max_points=61440)
        reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"])
        for which, key in ((0, "cloud0"), (1, "cloud1")):
            pts, covs = reg.preprocessed(0, which)
            po, co, _ = oracle.gicp_preprocess(fp[key])
            assert len(pts) == len(po)
            ig = np.lexsort((pts[:, 2], pts[:, 1], pts[:, 0]))
            io = np.lexsort((po[:, 2], po[:, 1], po[:, 0]))
            assert (pts[ig] == po[io]).all(), "voxel means differ"
            # exact kNN is structure independent except for exact distance ties at the k-th neighbour
            # (KdTree visiting order vs cell order): exclude those points, they must be very rare
            _, sq = oracle.knn(po[io], po[io], 11)
            tie = sq[:, 9] == sq[:, 10]
            assert tie.sum() <= 5
            diff = np.abs(covs[ig] - co[io][:, :3, :3]).reshape(len(pts), -1).max(1)
            assert diff[~tie].max() < 1e-9, "covariances differ"


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_gicp_pose_parity_vga(gpu_api, oracle, seed):
    fp = synth.frame_pair(seed)
    reg = gpu_api.RegistrationGICP(max_points=20480)
    r = reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"])
    ro = oracle.gicp_align(fp["cloud0"], fp["cloud1"])
    assert _rel(r["T"], ro["T"]) < TOL, (_rel(r["T"], ro["T"]), r, ro)
    assert r["converged"] == ro["converged"] and r["iterations"] == ro["iterations"]
    assert abs(r["num_inliers"] - ro["num_inliers"]) <= max(1, int(1e-3 * ro["num_inliers"]))
    assert r["n_target_ds"] == ro["n_target_ds"] and r["n_source_ds"] == ro["n_source_ds"]
    # 1e-5 on H / error: an exact distance tie at a 10th neighbour (seed 2 has one) picks another point than the KdTree order
    assert _rel(r["H"], ro["H"]) < 1e-5 and abs(r["error"] - ro["error"]) < 1e-5 * abs(ro["error"])


def test_gicp_pose_parity_sweep_40_seeds(gpu_api, oracle):
    """Every one of 40 synthetic VGA pairs meets the north_star bar against the UNMODIFIED (reference-order) oracle; seed 202 was
    at 5.4e-5 in round 1 when the voxel sort was a stable radix sort.  What is left is rounding noise plus exact k-th-neighbour
    distance ties (1e-7)."""
    import concurrent.futures as cf
    seeds = list(range(200, 240))
    pairs = [synth.frame_pair(s, 640, 480, 4) for s in seeds]
    with cf.ThreadPoolExecutor(max_workers=16) as ex:
        want = list(ex.map(lambda fp: oracle.gicp_align(fp["cloud0"], fp["cloud1"]), pairs))
    reg = gpu_api.RegistrationGICP(max_points=20480)
    worst = 0.0
    for s, fp, ro in zip(seeds, pairs, want):
        r = reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"])
        e = _rel(r["T"], ro["T"])
        worst = max(worst, e)
        assert e < 1e-6, (s, e)
        assert r["converged"] == ro["converged"] and r["iterations"] == ro["iterations"], s
        assert r["num_inliers"] == ro["num_inliers"] and r["n_source_ds"] == ro["n_source_ds"], s
        assert _rel(r["H"], ro["H"]) < 1e-5 and abs(r["error"] - ro["error"]) <= 1e-5 * abs(ro["error"]), s
    print("worst rel. Frobenius over 40 seeds:", worst)


def test_gicp_with_init_and_small_config(gpu_api, oracle):
    fp = synth.frame_pair(9, 320, 240, 2)
    T0 = np.eye(4)
    T0[:3, 3] = [0.01, -0.005, 0.008]
    reg = gpu_api.RegistrationGICP(max_points=20480)
    r = reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"], T0)
    ro = oracle.gicp_align(fp["cloud0"], fp["cloud1"], T0)
    assert _rel(r["T"], ro["T"]) < TOL and r["converged"] == ro["converged"]


def test_gicp_edge_cases(gpu_api, oracle):
    fp = synth.frame_pair(5, 320, 240, 2)
    reg = gpu_api.RegistrationGICP(max_points=20480)
    # identical clouds: zero motion, converged at iteration 0
    r0 = reg.RegisterPointClouds(fp["cloud0"], fp["cloud0"])
    assert r0["converged"] and np.allclose(r0["T"], np.eye(4), atol=1e-9) and r0["iterations"] == 0
    # no correspondences within 0.1 m
    far = fp["cloud0"].copy()
    far[:, 2] += 50
    rf = reg.RegisterPointClouds(fp["cloud0"], far)
    of = oracle.gicp_align(fp["cloud0"], far)
    assert rf["num_inliers"] == of["num_inliers"] == 0 and np.allclose(rf["T"], np.eye(4))
    assert rf["converged"] == of["converged"] and rf["iterations"] == of["iterations"]
    # tiny clouds (<= 10 points): the reference only warns (registration.hpp:34-39)
    tiny = fp["cloud0"][:8]
    rt = reg.RegisterPointClouds(tiny, tiny)
    ot = oracle.gicp_align(tiny, tiny)
    assert _rel(rt["T"], ot["T"]) < TOL and rt["num_inliers"] == ot["num_inliers"]
    # empty source
    re = reg.RegisterPointClouds(fp["cloud0"], np.zeros((0, 4), np.float32))
    assert re["num_inliers"] == 0 and np.allclose(re["T"], np.eye(4))


def test_wide_extent_cloud_on_an_lds_sized_handle(gpu_api, oracle):
    """A handle whose capacity fits the LDS-resident voxel-sort kernel does not launch the other two sort kernels up front
    (csrc/gicp.hip voxel_qsort_top): a cloud whose voxel keys do not compact to 31 bits is counted by the LDS kernel, gicp_run
    sees the count at its first poll and runs the call again with all kernels.  The result must be the oracle's either way."""
    fp = synth.frame_pair(9, 320, 240, 4)
    rng = np.random.default_rng(3)
    far = np.zeros((60, 4), np.float32)  # a few isolated returns far out: 80 m x 40 m x 20 m of 2 cm voxels = 12 + 11 + 10 bits
    far[:, 0] = rng.uniform(-40, 40, 60)
    far[:, 1] = rng.uniform(-20, 20, 60)
    far[:, 2] = rng.uniform(0.5, 20, 60)
    far[:, 3] = 1
    c0 = np.concatenate([fp["cloud0"], far]).astype(np.float32)
    c1 = np.concatenate([fp["cloud1"], far[::-1]]).astype(np.float32)
    reg = gpu_api.RegistrationGICP(max_points=19200)
    ro = oracle.gicp_align(c0, c1)
    for _ in range(2):  # (the second call starts from the state the redo left behind)
        r = reg.RegisterPointClouds(c0, c1)
        assert _rel(r["T"], ro["T"]) < TOL and r["converged"] == ro["converged"] and r["iterations"] == ro["iterations"]
        assert r["num_inliers"] == ro["num_inliers"] and r["n_source_ds"] == ro["n_source_ds"] and r["n_target_ds"] == ro["n_target_ds"]
    # ... and a compact pair on the same handle afterwards (no redo) is still right
    r = reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"])
    rn = oracle.gicp_align(fp["cloud0"], fp["cloud1"])
    assert _rel(r["T"], rn["T"]) < TOL and r["iterations"] == rn["iterations"] and r["n_source_ds"] == rn["n_source_ds"]


@pytest.mark.parametrize("cap", [36864, 40960])
def test_gicp_720p_cloud(gpu_api, oracle, cap):
    fp = synth.frame_pair(12, 1280, 720, 5)
    assert len(fp["cloud0"]) > 30000
    reg = gpu_api.RegistrationGICP(max_points=cap)
    r = reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"])
    ro = oracle.gicp_align(fp["cloud0"], fp["cloud1"])
    assert _rel(r["T"], ro["T"]) < TOL and r["converged"] == ro["converged"]


def test_stable_voxel_order_knob(gpu_api, oracle, monkeypatch):
    """GFS_GICP_VOXEL_ORDER=stable keeps round 1's stable radix order (equal voxel keys by point index): equal to the oracle's
    stable-order switch to rounding noise, and a documented deviation from the reference (seed 202: 5.4e-5)."""
    monkeypatch.setenv("GFS_GICP_VOXEL_ORDER", "stable")
    fp = synth.frame_pair(202, 640, 480, 4)
    reg = gpu_api.RegistrationGICP(max_points=32768)
    oracle.gicp_set_stable_voxel_order(1)
    try:
        r = reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"])
        ro = oracle.gicp_align(fp["cloud0"], fp["cloud1"])
    finally:
        oracle.gicp_set_stable_voxel_order(0)
    assert r["iterations"] == ro["iterations"] and r["num_inliers"] == ro["num_inliers"] and r["converged"] == ro["converged"]
    assert np.linalg.norm(r["T"] - ro["T"]) <= 1e-7 * np.linalg.norm(ro["T"])


def _same(a, b):
    return (np.array_equal(a["T"], b["T"]) and a["converged"] == b["converged"] and a["iterations"] == b["iterations"] and
            a["num_inliers"] == b["num_inliers"] and np.array_equal(a["H"], b["H"]) and np.array_equal(a["b"], b["b"]) and
            a["error"] == b["error"] and a["n_target_ds"] == b["n_target_ds"] and a["n_source_ds"] == b["n_source_ds"])


def test_streaming_form_is_bit_identical(gpu_api):
    """gfs_gicp_align_next: the previous call's preprocessed source serves as the target (Tracking::PredictStateICP's chain
    last frame -> current frame) — same bits as preprocessing both clouds again, for a chain of frames of different sizes."""
    sc = synth.Scene(31)
    rng = np.random.default_rng(2)
    clouds = []
    T = np.eye(4)
    for k in range(5):
        T = T @ synth.random_motion(rng)
        _, d = sc.render(320, 240, T if k else None, k)
        clouds.append(synth.depth_to_cloud(d, 2 + (k % 2)))      # alternating strides: clouds of different sizes
    plain = gpu_api.RegistrationGICP(max_points=32768)
    chain = gpu_api.RegistrationGICP(max_points=32768)
    r = chain.RegisterPointClouds(clouds[0], clouds[1])
    assert _same(r, plain.RegisterPointClouds(clouds[0], clouds[1]))
    for k in range(2, 5):
        r = chain.RegisterNext(clouds[k])
        assert _same(r, plain.RegisterPointClouds(clouds[k - 1], clouds[k])), k
        pts_c, cov_c = chain.preprocessed(0, 0)
        pts_p, cov_p = plain.preprocessed(0, 0)
        assert np.array_equal(pts_c, pts_p) and np.array_equal(cov_c, cov_p)
    # a plain call in between re-bases the chain on its source
    r = chain.RegisterPointClouds(clouds[1], clouds[0])
    assert _same(chain.RegisterNext(clouds[3]), plain.RegisterPointClouds(clouds[0], clouds[3]))
    # with an initial guess and an empty new cloud
    T0 = np.linalg.inv(synth.random_motion(rng))
    assert _same(chain.RegisterNext(clouds[4], T0), plain.RegisterPointClouds(clouds[3], clouds[4], T0))
    e = chain.RegisterNext(clouds[0][:0])
    assert e["num_inliers"] == 0 and e["n_source_ds"] == 0
    # a fresh handle has nothing cached; changed preprocessing parameters invalidate the cache
    with pytest.raises(gpu_api.GfsError):
        gpu_api.RegistrationGICP(max_points=32768).RegisterNext(clouds[1])
    cfg = gpu_api.gicp_default_config()
    cfg.downsampling_resolution = 0.05
    with pytest.raises(gpu_api.GfsError):
        chain.RegisterNext(clouds[1], cfg=cfg)


def test_streaming_batch_device(gpu_api):
    from test_gpu_gms import _Hip
    B, SP = 3, 9216
    frames = []
    for b in range(B):
        sc = synth.Scene(40 + b)
        rng = np.random.default_rng(b)
        T = np.eye(4)
        seq = []
        for k in range(3):
            T = T @ synth.random_motion(rng)
            seq.append(synth.depth_to_cloud(sc.render(320, 240, T if k else None, k)[1], 3))
        frames.append(seq)
    hip = _Hip()

    def dev(k):
        c = np.zeros((B, SP, 4), np.float32)
        n = np.zeros(B, np.int32)
        for b in range(B):
            c[b, :len(frames[b][k])] = frames[b][k]
            n[b] = len(frames[b][k])
        return hip.to_device(c), hip.to_device(n)

    d = [dev(k) for k in range(3)]
    chain = gpu_api.RegistrationGICP(max_points=SP, max_batch=B)
    plain = gpu_api.RegistrationGICP(max_points=SP, max_batch=B)
    chain.align_batch_device(d[0][0], d[0][1], d[1][0], d[1][1], B, SP)
    got = chain.align_next_batch_device(d[2][0], d[2][1], B, SP)
    want = plain.align_batch_device(d[1][0], d[1][1], d[2][0], d[2][1], B, SP)
    for b in range(B):
        assert _same(got[b], want[b]), b
    got = chain.align_next_batch_device(d[0][0], d[0][1], B, SP)           # third link of the chain: slots swap back
    want = plain.align_batch_device(d[2][0], d[2][1], d[0][0], d[0][1], B, SP)
    for b in range(B):
        assert _same(got[b], want[b]), b
    with pytest.raises(gpu_api.GfsError):
        chain.align_next_batch_device(d[0][0], d[0][1], B - 1, SP)           # batch size differs from the cached call
    hip.free()


def test_cooperative_lm_kernel_gives_the_bits_of_the_launch_per_step_rounds(gpu_api, monkeypatch):
    """k_gicp_lm_coop (the LM loop of a few pairs in ONE launch: several workgroups a pair, a per-pair barrier, the last arriver
    folds) against the launch-per-step rounds (k_gicp_linearize / solve / error / decide + host polls): every output bit for
    bit -- single pairs (the whole loop runs there), a small batch (the same), a batch of 24 whose tail goes there after the
    full rounds; ragged iteration counts, an initial guess far off (many rejected trials), an empty and a tiny cloud."""
    from test_gpu_gms import _Hip
    rng = np.random.default_rng(5)
    pairs = []
    for k in range(24):
        fp = synth.frame_pair(300 + k, 160, 120, stride=1 + (k % 2))
        pairs.append((fp["cloud0"], fp["cloud1"]))
    pairs[3] = (pairs[3][0], pairs[3][1][:7])          # a tiny source cloud
    pairs[5] = (pairs[5][0], pairs[5][1][:0])          # an empty one
    inits = [None] * 24
    inits[2] = np.linalg.inv(synth.random_motion(rng)) @ np.linalg.inv(synth.random_motion(rng))
    far = np.eye(4)
    far[:3, 3] = [0.08, -0.05, 0.06]
    inits[7] = far
    SP = 20480
    monkeypatch.setenv("GFS_GICP_COOP", "0")
    rounds1 = gpu_api.RegistrationGICP(max_points=SP)
    roundsB = gpu_api.RegistrationGICP(max_points=SP, max_batch=24)
    monkeypatch.delenv("GFS_GICP_COOP")
    monkeypatch.setenv("GFS_GICP_COOP_TAIL", "2")  # (the tail of a larger batch: off by default, measured slower next to busy lanes)
    coop1 = gpu_api.RegistrationGICP(max_points=SP)
    coopB = gpu_api.RegistrationGICP(max_points=SP, max_batch=24)
    assert rounds1.coop_stats()["budget"] == 0 and coop1.coop_stats()["budget"] >= 75
    # one pair at a time, and the streaming entry
    its = []
    for k in range(12):
        a, b = coop1.RegisterPointClouds(*pairs[k], inits[k]), rounds1.RegisterPointClouds(*pairs[k], inits[k])
        assert _same(a, b) and a["n_linearize"] == b["n_linearize"] and a["n_error_evals"] == b["n_error_evals"], k
        its.append(a["n_error_evals"])
    assert coop1.coop_stats()["launches"] == 12 and not coop1.coop_stats()["failed"] and rounds1.coop_stats()["launches"] == 0
    assert len(set(its)) >= 3 and max(its) >= 6, its  # ragged: short and long loops, rejected trials
    a, b = coop1.RegisterNext(pairs[13][1]), rounds1.RegisterNext(pairs[13][1])
    assert _same(a, b)
    # batches through the device entry: 4 pairs (whole loop in the kernel), 24 pairs (rounds, then the tail)
    hip = _Hip()

    def dev(which, idx):
        c = np.zeros((len(idx), SP, 4), np.float32)
        n = np.zeros(len(idx), np.int32)
        for j, k in enumerate(idx):
            c[j, :len(pairs[k][which])] = pairs[k][which]
            n[j] = len(pairs[k][which])
        return hip.to_device(c), hip.to_device(n)

    for idx in ([0, 2, 5, 7], list(range(24))):
        t, s_ = dev(0, idx), dev(1, idx)
        T0 = np.stack([np.eye(4) if inits[k] is None else inits[k] for k in idx])
        before = coopB.coop_stats()["launches"]
        got = coopB.align_batch_device(t[0], t[1], s_[0], s_[1], len(idx), SP, init_T=T0)
        want = roundsB.align_batch_device(t[0], t[1], s_[0], s_[1], len(idx), SP, init_T=T0)
        for j in range(len(idx)):
            assert _same(got[j], want[j]) and got[j]["n_error_evals"] == want[j]["n_error_evals"], (len(idx), j)
        if len(idx) * 80 <= coopB.coop_stats()["budget"] or len(idx) > 8:  # the whole loop / the tail behind the full rounds
            assert coopB.coop_stats()["launches"] == before + 1, (len(idx), coopB.coop_stats())
        # ... and a pair inside a batch = the pair alone
        for j, k in enumerate(idx[:4]):
            assert _same(got[j], coop1.RegisterPointClouds(*pairs[k], inits[k])), (len(idx), j)
    assert not coopB.coop_stats()["failed"]
    hip.free()


def test_step_in_the_last_workgroup_knob_gives_the_same_bits(gpu_api, monkeypatch):
    """GFS_GICP_FUSE_STEP=1: a pair's scalar step taken by the workgroup of the pass that finishes last (one launch a round; measured
    slower, not the default) -- the same bits as the step kernel, for a ragged batch with an empty and a tiny cloud."""
    from test_gpu_gms import _Hip
    pairs = []
    for k in range(12):
        fp = synth.frame_pair(500 + k, 160, 120, stride=1 + (k % 2))
        pairs.append((fp["cloud0"], fp["cloud1"]))
    pairs[3] = (pairs[3][0], pairs[3][1][:7])
    pairs[5] = (pairs[5][0], pairs[5][1][:0])
    SP, B = 20480, 12
    c0 = np.zeros((B, SP, 4), np.float32); c1 = np.zeros((B, SP, 4), np.float32); n0 = np.zeros(B, np.int32); n1 = np.zeros(B, np.int32)
    for b, (a, s_) in enumerate(pairs):
        c0[b, :len(a)], c1[b, :len(s_)], n0[b], n1[b] = a, s_, len(a), len(s_)
    hip = _Hip()
    d = [hip.to_device(x) for x in (c0, n0, c1, n1)]
    monkeypatch.setenv("GFS_GICP_COOP", "0")
    want = gpu_api.RegistrationGICP(max_points=SP, max_batch=B).align_batch_device(d[0], d[1], d[2], d[3], B, SP)
    monkeypatch.setenv("GFS_GICP_FUSE_STEP", "1")
    got = gpu_api.RegistrationGICP(max_points=SP, max_batch=B).align_batch_device(d[0], d[1], d[2], d[3], B, SP)
    for b in range(B):
        assert _same(got[b], want[b]) and got[b]["n_error_evals"] == want[b]["n_error_evals"], b
    hip.free()


def _gicp_same(r, ro, bar=1e-5):
    return (r["converged"] == ro["converged"] and r["iterations"] == ro["iterations"] and r["num_inliers"] == ro["num_inliers"]
            and r["n_target_ds"] == ro["n_target_ds"] and r["n_source_ds"] == ro["n_source_ds"] and _rel(r["T"], ro["T"]) < bar)


def test_random_pairs_meet_the_bar_or_have_a_proved_kth_distance_tie(gpu_api, oracle):
    """2 000 random noise-free cloud pairs (drawn like tests/fuzz_gpu.py section 2: sizes, truncations, motions up to 0.15 m / 5 deg).
    Every pair either agrees with the oracle -- converged flag, iteration count, inlier count, down-sampled sizes equal, pose within
    the 1e-5 bar of BASELINE.json -- or the ONE documented cause is established for it and then removed: the voxel means are
    bit-identical, some point has an EXACT distance tie between its 10th and 11th neighbour (ann/knn_result.hpp:80-101 keeps the first
    one pushed, i.e. the KdTree's visiting order, which the reference itself does not reproduce between runs: DESIGN.md section 2),
    and after moving the raw points behind every tied 11th neighbour by 20 micrometres the same pair agrees."""
    import concurrent.futures as cf
    reg = gpu_api.RegistrationGICP(max_points=65536)
    ties, worst = [], 0.0

    def draw(ci):  # the pair and the oracle's answer: on worker threads (the oracle is a C library, the GIL is released), ahead of the GPU
        rng = np.random.default_rng([4321, 2, ci])
        s = int(rng.integers(0, 1 << 30))
        w, h = int(rng.choice([96, 128, 160, 200])), int(rng.choice([72, 96, 120, 150]))
        c0, c1, _ = synth.cloud_pair(s, w, h, trans=float(rng.uniform(0.0, 0.15)), rot_deg=float(rng.uniform(0, 5)))
        if rng.integers(0, 4) == 0:
            c0 = c0[:int(len(c0) * rng.uniform(0.2, 1.0))]
        if rng.integers(0, 4) == 0:
            c1 = c1[::int(rng.integers(1, 4))]
        return c0, c1, oracle.gicp_align(c0, c1)

    pool = cf.ThreadPoolExecutor(max_workers=24)
    drawn = {}
    for ci in range(2000):
        for k in range(ci, min(ci + 96, 2000)):  # a bounded window of pairs in flight
            if k not in drawn:
                drawn[k] = pool.submit(draw, k)
        c0, c1, ro = drawn.pop(ci).result()
        r = reg.RegisterPointClouds(c0, c1)
        if _gicp_same(r, ro):
            worst = max(worst, _rel(r["T"], ro["T"]))
            continue
        clouds = [c0.copy(), c1.copy()]
        for attempt in range(3):  # breaking one tie can, rarely, create another
            n_ties = 0
            for which in (0, 1):
                pts, _ = reg.preprocessed(0, which)
                po, _, _ = oracle.gicp_preprocess(clouds[which])
                ig, io = np.lexsort((pts[:, 2], pts[:, 1], pts[:, 0])), np.lexsort((po[:, 2], po[:, 1], po[:, 0]))
                assert len(pts) == len(po) and (pts[ig] == po[io]).all(), ("voxel means differ", ci)
                idx, sq = oracle.knn(po, po, 11)
                for t in np.nonzero(sq[:, 9] == sq[:, 10])[0]:
                    n_ties += 1
                    key = np.floor(po[int(idx[t, 10]), :3] / 0.02).astype(np.int64)
                    raw = clouds[which]
                    sel = (np.floor(raw[:, :3].astype(np.float64) / 0.02).astype(np.int64) == key).all(1)
                    raw[sel, 0] += np.float32(2e-5)
            assert n_ties > 0, ("pair over the bar without a k-th-distance tie", ci, _rel(r["T"], ro["T"]), r["iterations"], ro["iterations"])
            r2, ro2 = reg.RegisterPointClouds(clouds[0], clouds[1]), oracle.gicp_align(clouds[0], clouds[1])
            if _gicp_same(r2, ro2):
                break
        else:
            raise AssertionError(("still over the bar with every tie broken", ci))
        ties.append((ci, _rel(r["T"], ro["T"]), n_ties))
    pool.shutdown()
    assert len(ties) <= 40, ties  # round 2 measured 7 of 13 000
    assert worst < 1e-5


def test_staged_tile_gives_the_same_bits(gpu_api, monkeypatch):
    """GFS_GICP_LIN_TILE=1: k_gicp_linearize stages a workgroup's tile of the target cloud in LDS (not the default: measured slower
    than the x-ordered sweep straight from HBM, profiles/README.md).  Same search, same order of visits: bit-identical results."""
    reg = gpu_api.RegistrationGICP(max_points=20480)
    for seed in (3, 1001, 1005):
        fp = synth.frame_pair(seed)
        monkeypatch.setenv("GFS_GICP_LIN_TILE", "0")
        a = reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"])
        monkeypatch.setenv("GFS_GICP_LIN_TILE", "1")
        b = reg.RegisterPointClouds(fp["cloud0"], fp["cloud1"])
        assert np.array_equal(a["T"], b["T"]) and a["iterations"] == b["iterations"] and a["num_inliers"] == b["num_inliers"]
        assert np.array_equal(a["H"], b["H"]) and a["error"] == b["error"]


def test_key_merge_selects_what_the_exact_passes_select(gpu_api, monkeypatch):
    """k_knn_cov keeps its candidates as (distance | point index) keys merged by a min / max network (csrc/gicp.hip TopKey11) and
    hands a query to the exact deferred passes when the k-th and (k+1)-th key agree above the index bits.  GFS_GICP_KNN_EXACT=1
    sends EVERY query through those exact passes: the covariances must be the same bits either way -- on depth-camera clouds and on a
    lattice, where nearly every point has an exact four-way tie at its 10th neighbour (4 at d, 4 at sqrt(2) d, then 4 at 2 d)."""
    clouds = [synth.frame_pair(s, 320, 240, 4)["cloud0"] for s in (3, 11)]
    gx, gy = np.meshgrid(np.arange(70), np.arange(50))
    lattice = np.stack([gx.ravel() * 0.02 - 0.69, gy.ravel() * 0.02 - 0.49, np.full(gx.size, 1.5) + 0.003 * (gx.ravel() % 3), np.ones(gx.size)], 1)
    clouds.append(lattice.astype(np.float32))
    reg = gpu_api.RegistrationGICP(max_points=20480)
    for c in clouds:
        monkeypatch.setenv("GFS_GICP_KNN_EXACT", "0")
        reg.RegisterPointClouds(c, c)
        pa, ca = reg.preprocessed(0, 0)
        monkeypatch.setenv("GFS_GICP_KNN_EXACT", "1")
        reg.RegisterPointClouds(c, c)
        pb, cb = reg.preprocessed(0, 0)
        assert np.array_equal(pa, pb) and len(pa) > 1000
        assert np.array_equal(ca.view(np.uint64), cb.view(np.uint64)), int((ca != cb).any(axis=(1, 2)).sum())
