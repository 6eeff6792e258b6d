"""Second, structurally different implementations (numpy, vectorised) of the OpenCV 4.5.4 primitives the CPU oracle restates
in C++ (oracle/orb_oracle.cpp; semantics in SURVEY.md App. A), compared bit for bit over randomised inputs.  Parity for this
path is unpinned by the reference (no vectors, not buildable here): these cross-checks catch slips of one restatement against
another; tests/golden/OPENCV_CHECK.md says how to run the committed fixtures against a real OpenCV.

ids (referenced from OPENCV_CHECK.md): A2 INTER_AREA, A4 GaussianBlur 7x7 fixed point, A5 fastAtan2, A3 FAST score / NMS."""
import numpy as np
import pytest


# ---------------------------------------------------------------------------------------------------------- A2: INTER_AREA
def _area_weights(ssize, dsize):
    """Dense [dsize, ssize] float32 weight matrix of cv::resize(INTER_AREA) along one axis (computeResizeAreaTab): geometric
    overlap of the destination cell [d s, (d+1) s) with every source pixel, divided by the cell width (clipped at the image
    end); partial overlaps below 1e-3 are dropped, interior pixels get exactly 1 / cellWidth."""
    scale = 1.0 / (dsize / ssize)  # cv::resize: inv_scale = dsize / ssize, scale = 1 / inv_scale (both double)
    d = np.arange(dsize, dtype=np.float64)
    f1 = d * scale
    f2 = f1 + scale
    cw = np.minimum(scale, ssize - f1)
    s1 = np.ceil(f1).astype(np.int64)
    s2 = np.minimum(np.floor(f2).astype(np.int64), ssize - 1)
    s1 = np.minimum(s1, s2)
    W = np.zeros((dsize, ssize), np.float32)
    k = np.arange(ssize)[None, :]
    inner = (k >= s1[:, None]) & (k < s2[:, None])
    W[inner] = np.broadcast_to((1.0 / cw).astype(np.float32)[:, None], W.shape)[inner]
    left = (s1 - f1) > 1e-3
    W[np.flatnonzero(left), (s1 - 1)[left]] = ((s1 - f1) / cw).astype(np.float32)[left]
    right = (f2 - s2) > 1e-3
    W[np.flatnonzero(right), s2[right]] = (np.minimum(np.minimum(f2 - s2, 1.0), cw) / cw).astype(np.float32)[right]
    return W


def _resize_area_numpy(src, drows, dcols):
    """ResizeArea_Invoker<uchar, float>: per source row buf = sum over taps (ascending source column) of S * alpha in float,
    rows folded as sum = beta * buf for the first source row of a destination row and sum += beta * buf after, cvRound."""
    srows, scols = src.shape
    Wx, Wy = _area_weights(scols, dcols), _area_weights(srows, drows)
    S = src.astype(np.float32)
    buf = np.zeros((srows, dcols), np.float32)
    for sx in range(scols):  # ascending source index = tab order; zero weights add an exact 0
        col = np.flatnonzero(Wx[:, sx])
        if len(col):
            buf[:, col] = buf[:, col] + S[:, sx:sx + 1] * Wx[col, sx][None, :]
    out = np.zeros((drows, dcols), np.float32)
    started = np.zeros(drows, bool)
    for sy in range(srows):
        for dy in np.flatnonzero(Wy[:, sy]):
            term = Wy[dy, sy] * buf[sy]
            out[dy] = term if not started[dy] else out[dy] + term
            started[dy] = True
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)  # rint = round half to even = cvRound


@pytest.mark.parametrize("shape,dst", [((480, 640), (400, 533)), ((400, 533), (333, 444)), ((134, 179), (112, 149)),
                                        ((720, 1280), (600, 1067)), ((97, 131), (81, 109)), ((60, 50), (59, 49)),
                                        ((64, 64), (32, 32)), ((31, 47), (13, 20))])
def test_A2_inter_area_matches_independent_numpy(oracle, shape, dst):
    rng = np.random.default_rng(shape[0] * 7 + dst[1])
    src = rng.integers(0, 256, shape).astype(np.uint8)
    src[::7, ::5] = 255
    src[3::11, 2::9] = 0
    assert np.array_equal(oracle.resize_area(src, *dst), _resize_area_numpy(src, *dst))


def test_A2_area_weights_are_the_overlaps():
    for ssize, dsize in ((640, 533), (533, 444), (1280, 1067), (179, 149), (50, 49)):
        W = _area_weights(ssize, dsize).astype(np.float64)
        scale = ssize / dsize
        d = np.arange(dsize)[:, None]
        k = np.arange(ssize)[None, :]
        ov = np.clip(np.minimum((d + 1) * scale, k + 1) - np.maximum(d * scale, k), 0, None) / np.minimum(scale, ssize - d * scale)
        assert np.abs(W - ov).max() < 1.5e-3 / scale + 1e-6  # only sub-1e-3 slivers are dropped
        assert np.abs(W.sum(1) - 1).max() < 2e-3


# ------------------------------------------------------------------------------------------------ A4: GaussianBlur 7x7, sigma 2
@pytest.mark.parametrize("variant,taps", [(0, [18, 34, 48, 56, 48, 34, 18]), (1, [18, 34, 49, 55, 49, 34, 18])])
def test_A4_blur_matches_direct_2d_fixed_point(oracle, variant, taps):
    """Direct 2-D form of the separable fixed-point filter: every product is an integer, so (sum_ij k_i k_j p_ij + 2^15) >> 16
    must equal the two-pass Q8.8 -> Q16.16 result exactly; BORDER_REFLECT_101."""
    k = np.array(taps, np.int64)
    K2 = np.outer(k, k)
    rng = np.random.default_rng(variant)
    for shape in ((30, 41), (7, 7), (64, 48), (120, 160)):
        img = rng.integers(0, 256, shape).astype(np.uint8)
        img[::5] = 255
        pad = np.pad(img.astype(np.int64), 3, mode="reflect")
        acc = np.zeros(shape, np.int64)
        for i in range(7):
            for j in range(7):
                acc += K2[i, j] * pad[i:i + shape[0], j:j + shape[1]]
        exp = np.minimum((acc + 32768) >> 16, 255).astype(np.uint8)
        assert np.array_equal(oracle.gaussian_blur7(img, variant), exp)


# ------------------------------------------------------------------------------------------------------------ A5: fastAtan2
def _fast_atan2_numpy(y, x):
    f = np.float32
    s = f(180 / np.pi)
    p1, p3, p5, p7 = f(0.9997878412794807) * s, f(-0.3258083974640975) * s, f(0.1555786518463281) * s, f(-0.04432655554792128) * s
    y, x = np.asarray(y, np.float32), np.asarray(x, np.float32)
    ax, ay = np.abs(x), np.abs(y)
    eps = f(2.2204460492503131e-16)
    swap = ax < ay
    num, den = np.where(swap, ax, ay), np.where(swap, ay, ax) + eps
    c = num / den
    c2 = c * c
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    a = np.where(swap, f(90) - a, a)
    a = np.where(x < 0, f(180) - a, a)
    a = np.where(y < 0, f(360) - a, a)
    return a.astype(np.float32)


def test_A5_fast_atan2_matches_independent_numpy(oracle):
    g = np.arange(-40, 41, dtype=np.float32)
    yy, xx = np.meshgrid(g * 37, g * 53, indexing="ij")            # the integer moments IC_Angle feeds it, scaled
    rng = np.random.default_rng(0)
    ys = np.concatenate([yy.ravel(), rng.integers(-200000, 200000, 20000).astype(np.float32), [0, 0, 1, -1, 0]])
    xs = np.concatenate([xx.ravel(), rng.integers(-200000, 200000, 20000).astype(np.float32), [0, 1, 0, 0, -1]])
    want = _fast_atan2_numpy(ys, xs)
    got = np.array([oracle.fast_atan2(float(a), float(b)) for a, b in zip(ys, xs)], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ------------------------------------------------------------------------------------------------------- A3: FAST on more data
def test_A3_fast_more_images(oracle):
    from geoflowslam_amd import synth
    from test_oracle_orb import _fast_bruteforce
    for seed, thr, shape in ((11, 20, (33, 57)), (12, 7, (50, 31)), (13, 60, (40, 40))):
        img = synth.noise_image(seed, shape[1], shape[0])
        if seed == 13:
            img = (img // 64 * 64).astype(np.uint8)  # plateaus: equal scores under the strict '>' NMS
        x, y, s = oracle.fast9_16(img, thr)
        assert list(zip(x.tolist(), y.tolist(), s.tolist())) == _fast_bruteforce(img, thr)


# -------------------------------------------------------------------------------- a1: the extractor's tables, GPU against oracle
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(1000, 1.2, 8, 20, 7), (2000, 1.2, 8, 20, 7), (500, 1.5, 5, 25, 9), (1250, 1.1, 12, 20, 7)])
def test_a1_orb_tables_equal_the_oracle(gpu_api, oracle, cfg):
    """gfs_orb_get_tables against ORBextractor's constructor (src/ORBextractor.cc:428-478): scale / inverse scale / sigma^2 /
    inverse sigma^2 per level (float, cumulative products), features per level (cvRound quotas), umax."""
    t = gpu_api.ORBextractor(*cfg).tables()
    o = oracle.OrbOracle(*cfg).tables()
    for k in ("scale", "inv_scale", "sigma2", "inv_sigma2"):
        assert np.array_equal(t[k].view(np.uint32), o[k].view(np.uint32)), k
    assert np.array_equal(t["feats"], o["feats"]) and np.array_equal(t["umax"], o["umax"])
    assert int(t["feats"].sum()) == cfg[0]
