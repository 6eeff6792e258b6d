"""GPU parity tests (MI355X): the HIP path, called through the C ABI, must be BIT-EXACT against the CPU oracle
for pyramid planes, FAST candidates, keypoints (x, y, size, angle, response, octave), 256-bit descriptors and
brute-force match pairs.  Tolerance: none (integer / byte / index work and exactly-rounded float32)."""
import numpy as np
import pytest

from geoflowslam_amd import synth

pytestmark = pytest.mark.gpu


def _compare_full(ext, orc, img, lap=(0, 0)):
    m, k, d = ext(img, lap)
    mo, ko, do = orc.extract(img, lap)
    for l in range(ext.nlevels):
        assert (ext.level(l) == orc.level(l)).all(), f"pyramid level {l} differs"
    for l in range(ext.nlevels):
        x, y, s = ext.candidates(l)
        xo, yo, so = orc.candidates(l)
        assert len(x) == len(xo), f"level {l}: {len(x)} candidates vs oracle {len(xo)}"
        assert (x == xo).all() and (y == yo).all() and (s == so).all(), f"FAST candidates differ on level {l}"
    for l in range(ext.nlevels):
        if len(orc.level_keypoints(l)):
            assert (ext.level(l, blurred=True) == orc.blurred(l)).all(), f"blurred level {l} differs"
    assert m == mo and len(k) == len(ko)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert (k[f] == ko[f]).all(), f"keypoint field {f} differs"
    assert (k["angle"].view(np.uint32) == ko["angle"].view(np.uint32)).all(), "angles differ (bitwise)"
    assert (d == do).all(), "descriptors differ"
    return m, k, d


@pytest.mark.parametrize("seed", [0, 1])
def test_orb_vga_noise_image(gpu_api, oracle, seed):
    ext = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7, max_rows=480, max_cols=640)
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    _compare_full(ext, orc, synth.noise_image(seed, 640, 480))


def test_orb_rendered_pair_and_match(gpu_api, oracle):
    fp = synth.frame_pair(11)
    ext = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7)
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    _, k0, d0 = _compare_full(ext, orc, fp["gray0"])
    _, k1, d1 = _compare_full(ext, orc, fp["gray1"])
    mt = gpu_api.ORBmatcher()
    ti, di = mt.match(d0, d1)
    to, do = oracle.bf_match(d0, d1)
    assert len(ti) == len(d0) and (ti == to).all() and (di == do).all()


def test_orb_720p_2000(gpu_api, oracle):
    ext = gpu_api.ORBextractor(2000, 1.2, 8, 20, 7, max_rows=720, max_cols=1280)
    orc = oracle.OrbOracle(2000, 1.2, 8, 20, 7)
    _compare_full(ext, orc, synth.noise_image(5, 1280, 720))


def test_orb_other_params_and_strided_input(gpu_api, oracle):
    # iniTh 25 / minTh 7 (the G1 yaml), 5 levels, odd size, non-continuous input (honour `stride`)
    big = synth.noise_image(9, 700, 500)
    view = big[7:7 + 411, 13:13 + 577]
    assert not view.flags["C_CONTIGUOUS"]
    ext = gpu_api.ORBextractor(600, 1.2, 5, 25, 7, max_rows=411, max_cols=577)
    orc = oracle.OrbOracle(600, 1.2, 5, 25, 7)
    m, k, d = ext(view)
    mo, ko, do = orc.extract(np.ascontiguousarray(view))
    assert m == mo and (k == ko).all() and (d == do).all()


def test_orb_blur_variant_and_lapping(gpu_api, oracle):
    img = synth.noise_image(21, 640, 480)
    ext = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7, blur_taps_variant=1)
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7, blur_variant=1)
    _compare_full(ext, orc, img, (0, 300))


def test_orb_edge_cases(gpu_api, oracle):
    ext = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7)
    assert ext(np.zeros((0, 0), np.uint8))[0] == -1  # empty image -> -1 (src/ORBextractor.cc:1150)
    m, k, d = ext(np.full((480, 640), 90, np.uint8))  # flat image: nothing
    assert m == 0 and len(k) == 0 and len(d) == 0
    # low-contrast image: only the minThFAST fallback fires
    img = (synth.noise_image(2, 640, 480).astype(np.int32) - 128) // 6 + 128
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    _compare_full(ext, orc, img.astype(np.uint8))


def test_orb_batch_matches_single(gpu_api, oracle):
    imgs = [synth.noise_image(30 + i, 640, 480) for i in range(5)]
    ext = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=5)
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    res = ext.extract_batch(imgs)
    for im, (m, k, d) in zip(imgs, res):
        mo, ko, do = orc.extract(im)
        assert m == mo and (k == ko).all() and (d == do).all()


@pytest.mark.parametrize("kind", ["binary_noise", "checker3", "ternary_noise"])
def test_orb_images_where_every_pixel_passes_the_pretests(gpu_api, oracle, kind):
    """k_fast_cells keeps ONE list entry a pixel with the polarities its antipodal-pair pre-test allows.  On these images most pixels
    pass the pre-test, many of them for BOTH polarities (top darker, bottom brighter, left darker, right brighter): the list is as
    full as it can get, and a pixel that is no dark corner must still be tried as a bright one."""
    rng = np.random.default_rng(5)
    W, H = 640, 480
    if kind == "binary_noise":
        img = (rng.integers(0, 2, (H, W)) * 255).astype(np.uint8)
    elif kind == "ternary_noise":
        img = rng.choice(np.array([0, 128, 255], np.uint8), size=(H, W))
    else:  # period-6 checkerboard in x and y, jittered: the four compass pixels of the ring (distance 3) alternate around the centre
        yy, xx = np.mgrid[0:H, 0:W]
        img = ((((xx // 3) + (yy // 3)) % 2) * 200 + 20 + rng.integers(0, 30, (H, W))).astype(np.uint8)
    ext = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7)
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    _compare_full(ext, orc, img)


def test_orb_every_batch_size_maps_every_cell_once(gpu_api, oracle):
    """k_fast_cells maps workgroup ids to (frame, cell) so that a frame's cells share an XCD (fast_map in csrc/orb.hip): with fewer
    than 8 frames a frame's cells are cut into 8 / B' parts, with more the frames go round-robin over the XCDs and the last group of 8
    is ragged.  Every batch size 1 .. 17 (and 24, 33) must give every frame the oracle's key points -- a cell mapped twice or not at
    all changes the candidates."""
    W, H = 320, 240
    pool = [synth.noise_image(300 + i, W, H) for i in range(6)]
    orc = oracle.OrbOracle(500, 1.2, 6, 20, 7)
    want = [orc.extract(im) for im in pool]
    ext = gpu_api.ORBextractor(500, 1.2, 6, 20, 7, max_rows=H, max_cols=W, max_batch=33)
    for B in list(range(1, 18)) + [24, 33]:
        imgs = [pool[(3 * b + B) % len(pool)] for b in range(B)]
        for b, (m, k, d) in enumerate(ext.extract_batch(imgs)):
            mo, ko, do = want[(3 * b + B) % len(pool)]
            assert m == mo and (k == ko).all() and (d == do).all(), (B, b)


def test_bf_match_random_and_ties(gpu_api, oracle):
    rng = np.random.default_rng(0)
    mt = gpu_api.ORBmatcher(max_query=5000, max_train=5000)
    for nq, nt in [(1, 1), (63, 65), (1000, 1000), (2000, 1999), (257, 4097), (5000, 3)]:
        q = rng.integers(0, 256, (nq, 32)).astype(np.uint8)
        t = rng.integers(0, 256, (nt, 32)).astype(np.uint8)
        if nt > 10:
            t[nt // 2] = t[3]  # duplicate train rows: lowest index must win
            q[0] = t[3]
        ti, di = mt.match(q, t)
        to, do = oracle.bf_match(q, t)
        assert (ti == to).all() and (di == do).all()
        if nt > 10:
            assert ti[0] == 3 and di[0] == 0
    # empty train set -> no matches; empty query -> no matches
    assert len(mt.match(rng.integers(0, 256, (5, 32)).astype(np.uint8), np.zeros((0, 32), np.uint8))[0]) == 0
    assert len(mt.match(np.zeros((0, 32), np.uint8), rng.integers(0, 256, (5, 32)).astype(np.uint8))[0]) == 0
    # all-equal distances: every query picks train 0
    z = np.zeros((70, 32), np.uint8)
    assert (mt.match(z, z)[0] == 0).all()


@pytest.mark.parametrize("seed", range(6))
def test_device_octree_matches_oracle(gpu_api, oracle, seed):
    """k_octree (device DistributeOctTree) against the oracle's std::list restatement: same keypoints, same order."""
    rng = np.random.default_rng(100 + seed)
    for trial in range(25):
        w, h = [(608, 448), (501, 368), (1248, 688), (338, 246), (147, 102)][trial % 5]
        n = int(rng.integers(1, [12, 300, 3000, 9000][trial % 4]))
        centers = rng.uniform(0, 1, (10, 2)) * [w, h]
        pts = centers[rng.integers(0, 10, n)] + rng.normal(0, 20, (n, 2))
        pts = np.unique(np.clip(np.rint(pts), [3, 3], [w - 4, h - 4]).astype(np.int32), axis=0)
        pts = pts[np.lexsort((pts[:, 0], pts[:, 1]))]
        score = rng.integers(7, 25, len(pts)).astype(np.int32)  # narrow range: many response ties
        quota = int(rng.choice([5, 60, 217, 434]))
        exp = oracle.distribute_octree(pts[:, 0], pts[:, 1], score, 16, 16 + w, 16, 16 + h, quota)
        ox, oy, os_ = gpu_api.octree_device(pts[:, 0], pts[:, 1], score, 16, 16 + w, 16, 16 + h, quota)
        assert len(ox) == len(exp)
        assert (ox == pts[exp, 0]).all() and (oy == pts[exp, 1]).all() and (os_ == score[exp]).all()


def test_host_octree_path_still_matches(gpu_api, oracle, monkeypatch):
    """GFS_ORB_OCTREE=host selects the round-1 host quadtree (kept as the fallback for geometries k_octree rejects)."""
    monkeypatch.setenv("GFS_ORB_OCTREE", "host")
    ext = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7)
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    _compare_full(ext, orc, synth.noise_image(3, 640, 480))


@pytest.mark.parametrize("knob", [None, "GFS_ORB_PYR_XTAB_HBM", "GFS_ORB_PYR_HBM", ("GFS_ORB_PYR_LDS_KB", "60"), ("GFS_ORB_PYR_LDS_KB", "24")])
def test_pyramid_kernel_paths_give_the_same_frames(gpu_api, oracle, knob, monkeypatch):
    """k_pyr_area_lds with its x tables in LDS (default) / in memory, finer strip cuts, and k_pyr_area_hbm: one result.
    The knobs are read when the extractor is created."""
    if knob is not None:
        name, value = knob if isinstance(knob, tuple) else (knob, "1")
        monkeypatch.setenv(name, value)
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    for w, h, seed in ((640, 480, 3), (577, 411, 4), (322, 242, 5)):
        ext = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7, max_rows=h, max_cols=w)
        _compare_full(ext, orc, synth.noise_image(seed, w, h))


@pytest.mark.parametrize("scale,levels", [(1.1, 8), (1.3, 6), (1.5, 5), (1.9, 4), (2.5, 3)])
def test_pyramid_scale_factors(gpu_api, oracle, scale, levels):
    """INTER_AREA tables of other scale factors: two to four taps a pixel (above 2 a fourth tap: the x tables stay in memory).
    (Integer factors take another cv::resize path and are refused by gfs_orb_create.)"""
    orc = oracle.OrbOracle(800, scale, levels, 20, 7)
    for w, h, seed in ((640, 480, 6), (801, 603, 7)):
        ext = gpu_api.ORBextractor(800, scale, levels, 20, 7, max_rows=h, max_cols=w)
        _compare_full(ext, orc, synth.noise_image(seed, w, h))
