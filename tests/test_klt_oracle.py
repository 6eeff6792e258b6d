"""CPU checks of the optical-flow restatement (oracle/klt_oracle.cpp; reference call sites src/Frame.cc:373,
src/ORBmatcher.cc:2186-2297).  The C oracle is compared bit for bit with an independent numpy transcription of
cv::buildOpticalFlowPyramid / LKTrackerInvoker, and against the properties the algorithm must have (a known flow is
recovered, the gates fire where they should)."""
import numpy as np
import pytest

from geoflowslam_amd import synth
from oracle import oracle as O

f32 = np.float32


# ---------------------------------------------------------------- independent numpy transcription
def _np_pyramid(img, win, max_level):
    lw, lh, off = O.klt_layout(img.shape[1], img.shape[0], win, max_level)
    pimg = np.zeros(int(off[-1]), np.uint8)
    pder = np.zeros((int(off[-1]), 2), np.int16)
    cur = img.astype(np.int64)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    for l in range(len(lw)):
        h, w = cur.shape
        assert (w, h) == (lw[l], lh[l])
        pimg[off[l]:off[l + 1]] = np.pad(cur, win, mode="reflect").astype(np.uint8).ravel()
        P = np.pad(cur, 1, mode="reflect")
        t0 = 3 * (P[:-2] + P[2:]) + 10 * P[1:-1]
        t1 = P[2:] - P[:-2]
        dx = t0[:, 2:] - t0[:, :-2]
        dy = 3 * (t1[:, 2:] + t1[:, :-2]) + 10 * t1[:, 1:-1]
        d = np.zeros((h + 2 * win, w + 2 * win, 2), np.int16)
        d[win:win + h, win:win + w, 0] = dx
        d[win:win + h, win:win + w, 1] = dy
        pder[off[l]:off[l + 1]] = d.reshape(-1, 2)
        if l + 1 < len(lw):
            nw, nh = (w + 1) // 2, (h + 1) // 2
            P2 = np.pad(cur, 2, mode="reflect")
            acc = np.zeros((nh, nw), np.int64)
            for r in range(5):
                for c in range(5):
                    acc += k[r] * k[c] * P2[r:r + 2 * nh:2, c:c + 2 * nw:2]
            cur = (acc + 128) >> 8
    return pimg, pder


def _weights(a, b):
    w00 = int(np.rint((f32(1) - a) * (f32(1) - b) * f32(16384)))
    w01 = int(np.rint(a * (f32(1) - b) * f32(16384)))
    w10 = int(np.rint((f32(1) - a) * b * f32(16384)))
    return w00, w01, w10, 16384 - w00 - w01 - w10


def _blend(A, y0, x0, win, wts, shift):
    w00, w01, w10, w11 = wts
    s = (A[y0:y0 + win, x0:x0 + win] * w00 + A[y0:y0 + win, x0 + 1:x0 + win + 1] * w01 + A[y0 + 1:y0 + win + 1, x0:x0 + win] * w10 +
         A[y0 + 1:y0 + win + 1, x0 + 1:x0 + win + 1] * w11)
    return (s + (1 << (shift - 1))) >> shift


def _np_track(prev_pyr, next_pyr, width, height, win, prev_pts, next_pts, max_level, pyr_max_level=3, max_iter=30, eps=0.01, flags=0,
              thr=1e-4):
    lw, lh, off = O.klt_layout(width, height, win, pyr_max_level)
    max_level = min(max_level, len(lw) - 1)
    n = len(prev_pts)
    status = np.ones(n, np.uint8)
    err = np.zeros(n, f32)
    nxt = np.array(next_pts, f32) if (flags & 4) else np.zeros((n, 2), f32)
    half = f32((win - 1) * 0.5)
    eps2 = min(max(eps, 0.0), 10.0) ** 2
    SC = f32(1.0 / (1 << 20))
    for level in range(max_level, -1, -1):
        w, h = int(lw[level]), int(lh[level])
        pw, ph = w + 2 * win, h + 2 * win
        I = prev_pyr[0][off[level]:off[level + 1]].reshape(ph, pw).astype(np.int64)
        dI = prev_pyr[1][off[level]:off[level + 1]].reshape(ph, pw, 2).astype(np.int64)
        J = next_pyr[0][off[level]:off[level + 1]].reshape(ph, pw).astype(np.int64)
        sc = f32(1.0 / (1 << level))
        for i in range(n):
            px, py = f32(prev_pts[i][0]) * sc, f32(prev_pts[i][1]) * sc
            if level == max_level:
                nx, ny = (nxt[i, 0] * sc, nxt[i, 1] * sc) if (flags & 4) else (px, py)
            else:
                nx, ny = nxt[i, 0] * f32(2), nxt[i, 1] * f32(2)
            nxt[i] = (nx, ny)
            px, py = px - half, py - half
            ipx, ipy = int(np.floor(px)), int(np.floor(py))
            if ipx < -win or ipx >= w or ipy < -win or ipy >= h:
                if level == 0:
                    status[i], err[i] = 0, 0
                continue
            wts = _weights(px - f32(ipx), py - f32(ipy))
            Iw = _blend(I, ipy + win, ipx + win, win, wts, 9)
            Ix = _blend(dI[..., 0], ipy + win, ipx + win, win, wts, 14)
            Iy = _blend(dI[..., 1], ipy + win, ipx + win, win, wts, 14)
            A11, A12, A22 = f32(int((Ix * Ix).sum())) * SC, f32(int((Ix * Iy).sum())) * SC, f32(int((Iy * Iy).sum())) * SC
            D = A11 * A22 - A12 * A12
            me = (A22 + A11 - np.sqrt((A11 - A22) * (A11 - A22) + f32(4) * A12 * A12)) / f32(2 * win * win)
            if flags & 8:
                err[i] = me
            if float(me) < thr or D < np.finfo(f32).eps:
                if level == 0:
                    status[i] = 0
                continue
            D = f32(1) / D
            nx, ny = nx - half, ny - half
            pdx = pdy = f32(0)
            for j in range(min(max(max_iter, 0), 100)):
                inx, iny = int(np.floor(nx)), int(np.floor(ny))
                if inx < -win or inx >= w or iny < -win or iny >= h:
                    if level == 0:
                        status[i] = 0
                    break
                wts = _weights(nx - f32(inx), ny - f32(iny))
                diff = _blend(J, iny + win, inx + win, win, wts, 9) - Iw
                b1, b2 = f32(int((diff * Ix).sum())) * SC, f32(int((diff * Iy).sum())) * SC
                dx, dy = (A12 * b2 - A22 * b1) * D, (A12 * b1 - A11 * b2) * D
                nx, ny = nx + dx, ny + dy
                nxt[i] = (nx + half, ny + half)
                if float(dx) * float(dx) + float(dy) * float(dy) <= eps2:
                    break
                if j > 0 and float(abs(dx + pdx)) < 0.01 and float(abs(dy + pdy)) < 0.01:
                    nxt[i] = (nxt[i, 0] - dx * f32(0.5), nxt[i, 1] - dy * f32(0.5))
                    break
                pdx, pdy = dx, dy
            if status[i] and level == 0 and not (flags & 8):
                ex, ey = nxt[i, 0] - half, nxt[i, 1] - half
                iex, iey = int(np.floor(ex)), int(np.floor(ey))
                if iex < -win or iex >= w or iey < -win or iey >= h:
                    status[i] = 0
                    continue
                wts = _weights(ex - f32(iex), ey - f32(iey))
                diff = _blend(J, iey + win, iex + win, win, wts, 9) - Iw
                err[i] = f32(int(np.abs(diff).sum())) * f32(1) / f32(32 * win * win)
    return nxt, status, err


def _points(rng, n, w, h, margin=4.0):
    return np.stack([rng.uniform(margin, w - margin, n), rng.uniform(margin, h - margin, n)], 1).astype(f32)


# ---------------------------------------------------------------- layout + pyramid
def test_layout_follows_build_optical_flow_pyramid():
    lw, lh, off = O.klt_layout(640, 480, 35, 3)
    assert list(lw) == [640, 320, 160, 80] and list(lh) == [480, 240, 120, 60]
    assert off[1] == (640 + 70) * (480 + 70) and off[-1] == sum((a + 70) * (b + 70) for a, b in zip(lw, lh))
    assert len(O.klt_layout(640, 480, 35, 6)[0]) == 4     # 40 x 30 is not larger than the window: stop at 80 x 60
    assert len(O.klt_layout(640, 480, 63, 3)[0]) == 3     # level 3 would be 80 x 60 <= 63
    assert len(O.klt_layout(640, 480, 15, 0)[0]) == 1
    lw, lh, _ = O.klt_layout(321, 243, 21, 3)
    assert list(lw) == [321, 161, 81, 41] and list(lh) == [243, 122, 61, 31]


@pytest.mark.parametrize("shape,win,lvl", [((120, 160), 15, 3), ((97, 131), 21, 3), ((60, 75), 9, 2)])
def test_pyramid_equals_numpy_transcription(shape, win, lvl):
    img = synth.noise_image(3, shape[1], shape[0])
    a_img, a_der = O.klt_build_pyramid(img, win, lvl)
    b_img, b_der = _np_pyramid(img, win, lvl)
    assert np.array_equal(a_img, b_img)
    assert np.array_equal(a_der, b_der)


def test_pyramid_properties():
    win = 11
    img = synth.noise_image(5, 96, 80)
    pimg, pder = O.klt_build_pyramid(img, win, 2)
    lw, lh, off = O.klt_layout(96, 80, win, 2)
    L0 = pimg[:off[1]].reshape(80 + 2 * win, 96 + 2 * win)
    assert np.array_equal(L0[win:-win, win:-win], img)
    assert np.array_equal(L0[win - 3, win:-win], img[3])            # reflect-101: row -3 mirrors row 3 (the edge row is not repeated)
    assert np.array_equal(L0[win:-win, -win + 2], img[:, -4])       # column w + 2 mirrors column w - 4
    D0 = pder[:off[1]].reshape(80 + 2 * win, 96 + 2 * win, 2)
    assert not D0[:win].any() and not D0[:, :win].any() and not D0[-win:].any() and not D0[:, -win:].any()
    flat = np.full((80, 96), 77, np.uint8)
    fimg, fder = O.klt_build_pyramid(flat, win, 2)
    assert (fimg == 77).all() and not fder.any()
    ramp = np.tile(np.arange(96, dtype=np.uint8) * 2, (80, 1))       # d/dx = 2 per pixel -> Scharr dx = 2 * 2 * 16
    _, rder = O.klt_build_pyramid(ramp, win, 0)
    R = rder.reshape(80 + 2 * win, 96 + 2 * win, 2)[win:-win, win:-win]
    assert (R[:, 1:-1, 0] == 64).all() and not R[..., 1].any()


# ---------------------------------------------------------------- tracker
@pytest.mark.parametrize("cfg", [dict(win=15, flags=0, lvl=3), dict(win=21, flags=12, lvl=2), dict(win=9, flags=8, lvl=0),
                                 dict(win=12, flags=4, lvl=3)])
def test_tracker_equals_numpy_transcription(cfg):
    w, h, win = 160, 120, cfg["win"]
    i0, i1, flow = synth.klt_texture_pair(11, w, h, shift=(2.3, -1.6), rot_deg=0.7)
    p0, p1 = O.klt_build_pyramid(i0, win), O.klt_build_pyramid(i1, win)
    rng = np.random.default_rng(2)
    pts = _points(rng, 60, w, h, 1.0)
    pts[:4] = [[-30.0, 10.0], [w + 20.0, 50.0], [0.2, 0.3], [w - 0.6, h - 0.4]]   # outside / on the border
    init = (pts + rng.normal(0, 1.0, pts.shape)).astype(f32)
    a = O.klt_track(p0, p1, w, h, win, pts, init, max_level=cfg["lvl"], flags=cfg["flags"])
    b = _np_track(p0, p1, w, h, win, pts, init, cfg["lvl"], flags=cfg["flags"])
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert a[1][:2].sum() == 0 and a[1].sum() > 40


def test_known_flow_is_recovered():
    w, h, win = 320, 240, 21
    i0, i1, flow = synth.klt_texture_pair(4, w, h, shift=(6.4, -4.2), rot_deg=0.4)
    p0, p1 = O.klt_build_pyramid(i0, win), O.klt_build_pyramid(i1, win)
    pts = _points(np.random.default_rng(0), 200, w, h, 25.0)
    nxt, st, err = O.klt_track(p0, p1, w, h, win, pts)
    d = np.linalg.norm(nxt - flow(pts), axis=1)
    assert st.mean() > 0.98 and np.median(d[st > 0]) < 0.08 and np.percentile(d[st > 0], 95) < 0.3
    assert (err[st > 0] < 3.0).all()                                   # L1 residual per pixel of a correct match is small
    # a single level cannot bridge a 7.6 px motion with a 21 px window as reliably as the pyramid does
    nxt0, st0, _ = O.klt_track(p0, p1, w, h, win, pts, max_level=0)
    d0 = np.linalg.norm(nxt0 - flow(pts), axis=1)
    assert np.median(d0) > np.median(d)
    # ... unless it is started from a good prior (OPTFLOW_USE_INITIAL_FLOW)
    nxt1, st1, me = O.klt_track(p0, p1, w, h, win, pts, flow(pts) + 0.5, max_level=0, flags=O.KLT_USE_INITIAL_FLOW | O.KLT_GET_MIN_EIGENVALS)
    d1 = np.linalg.norm(nxt1 - flow(pts), axis=1)
    assert np.median(d1[st1 > 0]) < 0.08 and (me[st1 > 0] >= 1e-4).all()


def test_gates():
    w, h, win = 160, 120, 15
    flat = np.full((h, w), 120, np.uint8)
    pf = O.klt_build_pyramid(flat, win)
    pts = _points(np.random.default_rng(1), 20, w, h, 10)
    nxt, st, err = O.klt_track(pf, pf, w, h, win, pts, flags=O.KLT_GET_MIN_EIGENVALS)
    assert st.sum() == 0 and (err == 0).all() and np.array_equal(nxt, pts)   # minEig gate; the estimate stays where it started
    i0, i1, _ = synth.klt_texture_pair(2, w, h, shift=(0.0, 0.0))
    p0, p1 = O.klt_build_pyramid(i0, win), O.klt_build_pyramid(i1, win)
    far = np.array([[w + 3.0 * win, 10.0], [10.0, -3.0 * win]], f32)
    _, st, err = O.klt_track(p0, p1, w, h, win, far)
    assert st.sum() == 0 and (err == 0).all()
    # an initial estimate far outside the next image: the point is dropped when the window leaves the padded image
    _, st, _ = O.klt_track(p0, p1, w, h, win, pts[:3], pts[:3] + 500.0, max_level=0, flags=O.KLT_USE_INITIAL_FLOW)
    assert st.sum() == 0
    out = O.klt_track(p0, p1, w, h, win, np.zeros((0, 2), f32))
    assert len(out[0]) == 0 and len(out[1]) == 0
    a = O.klt_track(p0, p1, w, h, win, pts, max_iter=0)                     # no Newton step at all
    assert np.array_equal(a[0], pts)
    b = O.klt_track(p0, p1, w, h, win, pts, max_level=7)                    # maxLevel is clamped to the pyramid's
    c = O.klt_track(p0, p1, w, h, win, pts, max_level=3)
    assert np.array_equal(b[0], c[0])


# ---------------------------------------------------------------- fbKltTracking
def test_fb_tracking_follows_the_reference_flow_of_control():
    w, h, win = 320, 240, 21
    i0, i1, flow = synth.klt_texture_pair(8, w, h, shift=(3.1, 2.2), rot_deg=0.3)
    i1 = i1.copy()
    i1[60:140, 100:200] = synth.noise_image(1, 100, 80)                      # an occluder: forward-backward must reject it
    p0, p1 = O.klt_build_pyramid(i0, win), O.klt_build_pyramid(i1, win)
    pts = _points(np.random.default_rng(3), 300, w, h, 6.0)
    pri, ok, good = O.fb_klt_tracking(p0, p1, w, h, win, 3, 15.0, 0.5, pts, pts.copy())
    assert good == ok.sum() and 150 < good < 300
    tgt = flow(pts)
    occluded = (tgt[:, 0] > 105) & (tgt[:, 0] < 195) & (tgt[:, 1] > 65) & (tgt[:, 1] < 135)
    assert ok[occluded].mean() < 0.2 and ok[~occluded].mean() > 0.85
    d = np.linalg.norm(pri - tgt, axis=1)
    assert np.median(d[ok]) < 0.08 and (d[ok] > 1.0).mean() < 0.02    # the sine texture is locally periodic: rare consistent aliases
    # by hand from the two calcOpticalFlowPyrLK calls (src/ORBmatcher.cc:2224-2291)
    flags = O.KLT_USE_INITIAL_FLOW | O.KLT_GET_MIN_EIGENVALS
    fw, st, er = O.klt_track(p0, p1, w, h, win, pts, pts.copy(), max_level=3, flags=flags, eps=float(f32(0.01)))
    assert np.array_equal(fw, pri)
    keep = (st > 0) & ~(er > 15.0) & (fw[:, 0] >= 1) & (fw[:, 0] < w - 1) & (fw[:, 1] >= 1) & (fw[:, 1] < h - 1)
    bw, stb, _ = O.klt_track(p1, p0, w, h, win, fw[keep], pts[keep].copy(), max_level=0, flags=flags, eps=float(f32(0.01)))
    dist = np.sqrt(((pts[keep] - bw).astype(np.float64) ** 2).sum(1))
    ok2 = np.zeros(len(pts), bool)
    ok2[np.flatnonzero(keep)] = (stb > 0) & ~(dist > 0.5)
    assert np.array_equal(ok, ok2)
    # the level count asked for is clamped to the pyramid's (nbpyrlvl_2d_2d = 6 with 4 levels, src/ORBmatcher.cc:2444)
    pri6, ok6, _ = O.fb_klt_tracking(p0, p1, w, h, win, 6, 15.0, 0.5, pts, pts.copy())
    assert np.array_equal(pri6, pri) and np.array_equal(ok6, ok)
    e = O.fb_klt_tracking(p0, p1, w, h, win, 3, 15.0, 0.5, pts[:0], pts[:0])
    assert e[2] == 0 and len(e[1]) == 0


def test_fb_tracking_uses_the_priors():
    w, h, win = 320, 240, 15
    i0, i1, flow = synth.klt_texture_pair(9, w, h, shift=(14.0, -11.0))
    p0, p1 = O.klt_build_pyramid(i0, win), O.klt_build_pyramid(i1, win)
    pts = _points(np.random.default_rng(5), 150, w, h, 30.0)
    _, ok_plain, _ = O.fb_klt_tracking(p0, p1, w, h, win, 0, 15.0, 0.5, pts, pts.copy())
    pri, ok_prior, _ = O.fb_klt_tracking(p0, p1, w, h, win, 0, 15.0, 0.5, pts, flow(pts).astype(f32) + 0.4)
    assert ok_prior.mean() > 0.9 > ok_plain.mean()
    assert np.median(np.linalg.norm(pri - flow(pts), axis=1)[ok_prior]) < 0.08
