"""GPU parity tests for the optical-flow front end (MI355X; k_klt_level0 / k_klt_pyrdown / k_klt_scharr, k_klt_track, k_klt_fb
through the C ABI): pyramids, tracked positions, status flags and error measures are BIT-EXACT against the CPU oracle
(oracle/klt_oracle.cpp) — every sum of the method is an exact integer sum, so no tolerance is needed."""
import ctypes as C

import numpy as np
import pytest

from geoflowslam_amd import synth

pytestmark = pytest.mark.gpu
f32 = np.float32


def _points(rng, n, w, h, margin=4.0):
    return np.stack([rng.uniform(margin, w - margin, n), rng.uniform(margin, h - margin, n)], 1).astype(f32)


def _bits(a):
    return np.ascontiguousarray(a, f32).view(np.uint32)


@pytest.mark.parametrize("cfg", [dict(w=640, h=480, win=35), dict(w=321, h=243, win=21), dict(w=640, h=480, win=40),
                                 dict(w=1280, h=720, win=30), dict(w=131, h=97, win=15, lvl=2), dict(w=640, h=480, win=63)])
def test_pyramid_is_bit_exact(gpu_api, oracle, cfg):
    w, h, win, lvl = cfg["w"], cfg["h"], cfg["win"], cfg.get("lvl", 3)
    imgs = [synth.noise_image(7 + k, w, h) for k in range(3)]
    imgs[2] = synth.klt_texture_pair(3, w, h)[0]
    trk = gpu_api.KltTracker(w, h, win, max_level=lvl, max_batch=3, max_points=16)
    lay, olay = trk.layout(), oracle.klt_layout(w, h, win, lvl)
    for a, b in zip(lay, olay):
        assert np.array_equal(a, b)
    pyr = trk.buildOpticalFlowPyramid(imgs)
    for f in range(3):
        gi, gd = pyr.download(f)
        oi, od = oracle.klt_build_pyramid(imgs[f], win, lvl)
        assert np.array_equal(gi, oi)
        assert np.array_equal(gd, od)
    # a strided host image (cv::Mat ROI) gives the same pyramid
    big = np.zeros((h, w + 13), np.uint8)
    big[:, :w] = imgs[0]
    ptrs = (C.c_void_p * 1)(big.ctypes.data)
    gpu_api._check(gpu_api.lib().gfs_klt_build_pyramid(trk.h, pyr.h, ptrs, w + 13, 1), "gfs_klt_build_pyramid")
    assert np.array_equal(pyr.download(0)[0], oracle.klt_build_pyramid(imgs[0], win, lvl)[0])


@pytest.mark.parametrize("cfg", [dict(win=35, flags=12, lvl=3), dict(win=35, flags=0, lvl=3), dict(win=15, flags=4, lvl=3),
                                 dict(win=30, flags=8, lvl=1), dict(win=40, flags=12, lvl=6), dict(win=21, flags=0, lvl=0),
                                 dict(win=9, flags=0, lvl=3), dict(win=63, flags=12, lvl=3), dict(win=4, flags=12, lvl=3)])
def test_calc_optical_flow_pyr_lk_is_bit_exact(gpu_api, oracle, cfg):
    w, h, win = 640, 480, cfg["win"]
    i0, i1, flow = synth.klt_texture_pair(21 + win, w, h, shift=(4.3, -3.6), rot_deg=0.6)
    rng = np.random.default_rng(win)
    pts = _points(rng, 700, w, h, 1.0)
    pts[:6] = [[-50.0, 10.0], [w + 70.0, 50.0], [0.2, 0.3], [w - 0.6, h - 0.4], [w + win * 0.4, 100.0], [200.0, -win * 0.45]]
    init = (flow(pts) + rng.normal(0, 1.5, pts.shape)).astype(f32)
    init[10:14] += 2000.0                                  # estimates far outside the image
    trk = gpu_api.KltTracker(w, h, win, max_batch=1, max_points=1024)
    p0, p1 = trk.buildOpticalFlowPyramid(i0), trk.buildOpticalFlowPyramid(i1)
    o0, o1 = oracle.klt_build_pyramid(i0, win), oracle.klt_build_pyramid(i1, win)
    g = trk.calcOpticalFlowPyrLK(p0, p1, pts, init, max_level=cfg["lvl"], flags=cfg["flags"])
    o = oracle.klt_track(o0, o1, w, h, win, pts, init, max_level=cfg["lvl"], flags=cfg["flags"])
    assert np.array_equal(g[1], o[1])
    assert np.array_equal(_bits(g[0]), _bits(o[0]))
    assert np.array_equal(_bits(g[2]), _bits(o[2]))
    assert g[1][:2].sum() == 0
    if win >= 9:
        ok = g[1] > 0
        assert ok.mean() > 0.9 and np.median(np.linalg.norm(g[0] - flow(pts), axis=1)[ok]) < 0.1


def test_termination_parameters_and_flat_images(gpu_api, oracle):
    w, h, win = 320, 240, 21
    i0, i1, _ = synth.klt_texture_pair(5, w, h, shift=(2.0, 1.0))
    flat = np.full((h, w), 99, np.uint8)
    trk = gpu_api.KltTracker(w, h, win, max_batch=1, max_points=256)
    p0, p1, pf = (trk.buildOpticalFlowPyramid(x) for x in (i0, i1, flat))
    o0, o1, of = (oracle.klt_build_pyramid(x, win) for x in (i0, i1, flat))
    pts = _points(np.random.default_rng(0), 200, w, h, 2.0)
    for kw in (dict(max_iter=0), dict(max_iter=1), dict(max_iter=3, eps=0.0), dict(max_iter=100, eps=1e-4), dict(eps=0.5),
               dict(min_eig_thr=0.05), dict(max_iter=1000, eps=50.0)):
        g = trk.calcOpticalFlowPyrLK(p0, p1, pts, **kw)
        o = oracle.klt_track(o0, o1, w, h, win, pts, **kw)
        assert np.array_equal(g[1], o[1]) and np.array_equal(_bits(g[0]), _bits(o[0])) and np.array_equal(_bits(g[2]), _bits(o[2])), kw
    g = trk.calcOpticalFlowPyrLK(pf, p1, pts, flags=gpu_api.KLT_GET_MIN_EIGENVALS)
    o = oracle.klt_track(of, o1, w, h, win, pts, flags=oracle.KLT_GET_MIN_EIGENVALS)
    assert g[1].sum() == 0 and np.array_equal(g[1], o[1]) and np.array_equal(_bits(g[0]), _bits(o[0]))


@pytest.mark.parametrize("cfg", [dict(seed=3, win=35), dict(seed=4, win=30, lvl=6), dict(seed=5, win=15, lvl=0), dict(seed=6, win=40)])
def test_fb_klt_tracking_is_bit_exact_on_rendered_frames(gpu_api, oracle, cfg):
    """The reference's use: key points of the last frame tracked into the current frame of a moving camera."""
    w, h, win, lvl = 640, 480, cfg["win"], cfg.get("lvl", 3)
    fp = synth.frame_pair(cfg["seed"], w, h, 8)
    ext = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7, max_rows=h, max_cols=w)
    _, k0, _ = ext(fp["gray0"])
    kps = np.stack([k0["x"], k0["y"]], 1).astype(f32)
    pri = kps + f32(0.75)
    trk = gpu_api.KltTracker(w, h, win, max_batch=1, max_points=2048)
    p0, p1 = trk.buildOpticalFlowPyramid(fp["gray0"]), trk.buildOpticalFlowPyramid(fp["gray1"])
    o0, o1 = oracle.klt_build_pyramid(fp["gray0"], win), oracle.klt_build_pyramid(fp["gray1"], win)
    g = trk.fbKltTracking(p0, p1, lvl, 15.0, 0.5, kps, pri)
    o = oracle.fb_klt_tracking(o0, o1, w, h, win, lvl, 15.0, 0.5, kps, pri)
    assert g[2] == o[2] and np.array_equal(g[1], o[1])
    assert np.array_equal(_bits(g[0]), _bits(o[0]))
    assert g[2] > 0.5 * len(kps)


def test_fb_batch_ragged_and_device_entry(gpu_api, oracle):
    from test_gpu_gms import _Hip
    w, h, win, B, S = 320, 240, 21, 4, 512
    rng = np.random.default_rng(1)
    pairs = [synth.klt_texture_pair(30 + b, w, h, shift=(rng.uniform(-5, 5), rng.uniform(-5, 5)), rot_deg=rng.uniform(-1, 1))
             for b in range(B)]
    trk = gpu_api.KltTracker(w, h, win, max_batch=B, max_points=S)
    prev = trk.buildOpticalFlowPyramid([p[0] for p in pairs])
    cur = trk.buildOpticalFlowPyramid([p[1] for p in pairs])
    n = [400, 0, 1, 512]
    kps = [_points(rng, k, w, h, 3.0) for k in n]
    pri = [(k + rng.normal(0, 0.7, k.shape)).astype(f32) for k in kps]
    G = trk.fbKltTracking(prev, cur, 3, 15.0, 0.5, kps, pri)
    for b in range(B):
        o0, o1 = oracle.klt_build_pyramid(pairs[b][0], win), oracle.klt_build_pyramid(pairs[b][1], win)
        o = oracle.fb_klt_tracking(o0, o1, w, h, win, 3, 15.0, 0.5, kps[b], pri[b])
        assert G[b][2] == o[2] and np.array_equal(G[b][1], o[1]) and np.array_equal(_bits(G[b][0]), _bits(o[0])), b
    # the device-resident entry point gives the same answers
    hip = _Hip()
    K = np.zeros((B, S, 2), f32)
    P = np.zeros((B, S, 2), f32)
    for b in range(B):
        K[b, :n[b]] = kps[b]
        P[b, :n[b]] = pri[b]
    d_n, d_k, d_p = hip.to_device(np.array(n, np.int32)), hip.to_device(K), hip.to_device(P)
    d_s, d_g = hip.to_device(np.zeros((B, S), np.uint8)), hip.to_device(np.zeros(B, np.int32))
    trk.fb_track_device(prev, cur, B, S, d_n, d_k, d_p, d_s, d_g)
    Pd, Sd, Gd = hip.to_host(d_p, (B, S, 2), f32), hip.to_host(d_s, (B, S), np.uint8), hip.to_host(d_g, (B,), np.int32)
    for b in range(B):
        assert Gd[b] == G[b][2] and np.array_equal(Sd[b, :n[b]].astype(bool), G[b][1]) and np.array_equal(_bits(Pd[b, :n[b]]), _bits(G[b][0]))
    hip.free()


def test_full_size_batch_recovers_the_flow(gpu_api):
    """BASELINE-sized frames (VGA, 1000 points, window 35), a batch of 16 pairs: size-independent properties only."""
    w, h, win, B = 640, 480, 35, 16
    rng = np.random.default_rng(9)
    pairs = [synth.klt_texture_pair(60 + b, w, h, shift=(rng.uniform(-6, 6), rng.uniform(-6, 6)), rot_deg=rng.uniform(-1, 1)) for b in range(B)]
    trk = gpu_api.KltTracker(w, h, win, max_batch=B, max_points=1024)
    prev = trk.buildOpticalFlowPyramid([p[0] for p in pairs])
    cur = trk.buildOpticalFlowPyramid([p[1] for p in pairs])
    kps = [_points(rng, 1000, w, h, 12.0) for _ in range(B)]
    G = trk.fbKltTracking(prev, cur, 3, 15.0, 0.5, kps, [k.copy() for k in kps])
    G2 = trk.fbKltTracking(prev, cur, 3, 15.0, 0.5, kps, [k.copy() for k in kps])
    for b in range(B):
        pri, ok, good = G[b]
        d = np.linalg.norm(pri - pairs[b][2](kps[b]), axis=1)
        assert good == ok.sum() and ok.mean() > 0.85 and np.median(d[ok]) < 0.15 and (d[ok] > 1.0).mean() < 0.06
        assert np.array_equal(_bits(pri), _bits(G2[b][0])) and np.array_equal(ok, G2[b][1])      # run-to-run identical
    # tracking a frame against itself leaves every textured point where it is
    S = trk.fbKltTracking(prev, prev, 3, 15.0, 0.5, kps[:1] * B, [k.copy() for k in kps[:1] * B])
    assert S[0][1].mean() > 0.97 and np.abs(S[0][0] - kps[0])[S[0][1]].max() < 0.02


def test_argument_checks(gpu_api):
    with pytest.raises(gpu_api.GfsError):
        gpu_api.KltTracker(640, 480, 2)
    with pytest.raises(gpu_api.GfsError):
        gpu_api.KltTracker(640, 480, 64)
    trk = gpu_api.KltTracker(160, 120, 15, max_batch=1, max_points=32)
    other = gpu_api.KltTracker(160, 120, 15, max_batch=1, max_points=32)
    img = synth.noise_image(0, 160, 120)
    p = trk.buildOpticalFlowPyramid(img)
    q = other.buildOpticalFlowPyramid(img)
    pts = np.full((40, 2), 50, f32)
    with pytest.raises(gpu_api.GfsError):
        trk.fbKltTracking(p, p, 3, 15.0, 0.5, pts, pts)          # more points than max_points
    with pytest.raises(gpu_api.GfsError):
        trk.fbKltTracking(p, q, 3, 15.0, 0.5, pts[:8], pts[:8])  # pyramid of another tracker
    with pytest.raises(gpu_api.GfsError):
        trk.buildOpticalFlowPyramid([img, img])                  # batch larger than max_batch
    with pytest.raises(gpu_api.GfsError):
        trk.calcOpticalFlowPyrLK(p, p, pts[:8], flags=1)
    empty = trk.fbKltTracking(p, p, 3, 15.0, 0.5, pts[:0], pts[:0])
    assert empty[2] == 0 and len(empty[1]) == 0


def test_huge_and_infinite_coordinates(gpu_api, oracle):
    """Key points / priors far outside any image (1e30, inf): dropped like any out-of-range point, nothing hangs."""
    w, h, win = 160, 120, 15
    i0, i1, _ = synth.klt_texture_pair(3, w, h, shift=(1.0, 0.5))
    trk = gpu_api.KltTracker(w, h, win, max_batch=1, max_points=64)
    p0, p1 = trk.buildOpticalFlowPyramid(i0), trk.buildOpticalFlowPyramid(i1)
    o0, o1 = oracle.klt_build_pyramid(i0, win), oracle.klt_build_pyramid(i1, win)
    kps = _points(np.random.default_rng(0), 24, w, h, 10.0)
    pri = kps.copy()
    kps[0], kps[1], kps[2] = (1e30, 5.0), (-1e30, 7.0), (np.inf, 9.0)
    pri[3], pri[4], pri[5] = (1e30, 1e30), (20.0, -np.inf), (-3e38, 3e38)
    g = trk.fbKltTracking(p0, p1, 3, 15.0, 0.5, kps, pri)
    o = oracle.fb_klt_tracking(o0, o1, w, h, win, 3, 15.0, 0.5, kps, pri)
    assert g[2] == o[2] and np.array_equal(g[1], o[1]) and not g[1][:6].any() and g[1][6:].all()
    ok = g[1]
    assert np.array_equal(_bits(g[0][ok]), _bits(o[0][ok]))
