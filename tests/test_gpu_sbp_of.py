"""ORBmatcher::SearchByProjectionWithOF (reference src/ORBmatcher.cc:2303-2497) as host adaptor code around the device KLT / F
kernels (gfs_host::SearchByProjectionWithOF, geoflowslam_amd/host/gfs_adaptors.hpp): prior projection, the occupancy mask with
cv::circle discs, the 3-D -> 2-D hand-over, the two tracked lists.  Checked against a Python transcription of the reference
function that calls the CPU oracle's fbKltTracking / findFundamentalMat."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from geoflowslam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "tests", "host", "_sbp_of_test.so")


@pytest.fixture(scope="module")
def harness(api):
    src = os.path.join(ROOT, "tests", "host", "sbp_of_test.cpp")
    hdr = os.path.join(ROOT, "geoflowslam_amd", "host", "gfs_adaptors.hpp")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        libdir = os.path.join(ROOT, "geoflowslam_amd")
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-o", _SO, src, "-L" + libdir, "-lgfs_hip",
                        "-Wl,-rpath," + libdir], check=True)
    L = C.CDLL(_SO)
    L.sbp_of_test.argtypes = ([C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
                              + [C.c_void_p] * 4 + [C.c_float, C.c_int] + [C.c_void_p] * 7)
    L.fill_circle_test.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
    return L


def _circle(img, x, y, radius):
    """cv::circle(img, Point2f, radius, 255, FILLED): OpenCV's midpoint recurrence with filled spans (drawing.cpp Circle())."""
    rows, cols = img.shape
    cx, cy = int(np.rint(np.float32(x))), int(np.rint(np.float32(y)))
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        for yy, x0, x1 in ((cy - dy, cx - dx, cx + dx), (cy + dy, cx - dx, cx + dx), (cy - dx, cx - dy, cx + dy), (cy + dx, cx - dy, cx + dy)):
            if 0 <= yy < rows:
                img[yy, max(x0, 0):min(x1, cols - 1) + 1] = 255
        dy += 1
        err += plus
        plus += 2
        m = -1 if err > 0 else 0
        err -= minus & m
        dx += m
        minus -= m & 2


def test_fill_circle_is_a_disc(harness):
    for r in (0, 1, 2, 3, 5, 10, 30):
        for (x, y) in ((40.0, 37.0), (40.5, 36.5), (2.2, 3.7), (78.9, 59.1)):
            a = np.zeros((60, 80), np.uint8)
            b = np.zeros((60, 80), np.uint8)
            harness.fill_circle_test(a.ctypes.data, 60, 80, x, y, r)
            _circle(b, x, y, r)
            assert np.array_equal(a, b)
            yy, xx = np.mgrid[0:60, 0:80]
            d2 = (xx - int(np.rint(np.float32(x)))) ** 2 + (yy - int(np.rint(np.float32(y)))) ** 2
            assert (a[d2 <= max(r - 1, 0) ** 2] == 255).all() and (a[d2 > (r + 1) ** 2] == 0).all()


def _reference_transcription(O, g0, g1, win, keys_last, has_mp, bad, outl, xw, keys_cur, q, t, K8, F_THR, DIST):
    """The reference function line by line (float32 arithmetic), on top of the oracle's KLT / RANSAC restatements."""
    f32 = np.float32
    H, W = g1.shape
    mask = np.zeros((H, W), np.uint8)
    for k in keys_cur:
        if 0 < k["x"] < W and 0 < k["y"] < H:
            mask[int(k["y"]), int(k["x"])] = 255
    fx, fy, cx, cy, minx, maxx, miny, maxy = [f32(v) for v in K8]
    qx, qy, qz, qw = [f32(v) for v in q]
    v3id, v3k, v3p, v2id, v2k, v2p = [], [], [], [], [], []

    def push2d(i):
        v2k.append((keys_last[i]["x"], keys_last[i]["y"]))
        v2p.append((keys_last[i]["x"], keys_last[i]["y"]))
        v2id.append(i)

    for i in range(len(keys_last)):
        if not has_mp[i]:
            push2d(i)
            continue
        if bad[i] or outl[i]:
            continue
        X = xw[i].astype(f32)
        tx, ty, tz = f32(2) * (qy * X[2] - qz * X[1]), f32(2) * (qz * X[0] - qx * X[2]), f32(2) * (qx * X[1] - qy * X[0])
        xc = X[0] + qw * tx + (qy * tz - qz * ty) + f32(t[0])
        yc = X[1] + qw * ty + (qz * tx - qx * tz) + f32(t[1])
        zc = X[2] + qw * tz + (qx * ty - qy * tx) + f32(t[2])
        with np.errstate(divide="ignore"):
            invz = f32(1.0 / np.float64(zc))
        u, v = fx * xc * invz + cx, fy * yc * invz + cy
        if invz < 0 or u < minx or u > maxx or v < miny or v > maxy:
            push2d(i)
            continue
        v3k.append((keys_last[i]["x"], keys_last[i]["y"]))
        v3p.append((u, v))
        v3id.append(i)
    p0, p1 = O.klt_build_pyramid(g0, win), O.klt_build_pyramid(g1, win)

    def track(kps, pri, lvl, fthr):
        kps, pri = np.array(kps, f32).reshape(-1, 2), np.array(pri, f32).reshape(-1, 2)
        pri, ok, _ = O.fb_klt_tracking(p0, p1, W, H, win, lvl, 15.0, 0.5, kps, pri)
        ok = ok.copy()
        idx = np.flatnonzero(ok)
        if len(idx) > 8:
            m, _, _, _ = O.fundamental_ransac(kps[idx], pri[idx], float(fthr), 0.99)
            ok[idx[~m]] = False
        return pri, ok

    good, t3, t2 = 0, [], []
    if v3id:
        pri, ok = track(v3k, v3p, 3, F_THR)
        for j, i in enumerate(v3id):
            if ok[j]:
                x, y = pri[j]
                if mask[int(y), int(x)] == 255:
                    continue
                t3.append((i, x, y))
                good += 1
                _circle(mask, x, y, DIST)
            else:
                push2d(i)
    if v2id:
        pri, ok = track(v2k, v2p, 6, f32(F_THR) * f32(0.5))
        for j, i in enumerate(v2id):
            if ok[j]:
                x, y = pri[j]
                if mask[int(y), int(x)] == 255:
                    continue
                t2.append((i, x, y))
                _circle(mask, x, y, DIST)
                good += 1
    return good, t3, t2, mask


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 8])
def test_search_by_projection_with_of(harness, gpu_api, oracle, seed):
    W, H, WIN = 320, 240, 21
    fp = synth.frame_pair(seed, W, H, 4)
    orb = oracle.OrbOracle(600, 1.2, 6, 20, 7)
    _, k0, _ = orb.extract(fp["gray0"])
    _, k1, _ = orb.extract(fp["gray1"])
    k1 = k1[::3]                                 # the current frame already has some key points (they seed the mask)
    rng = np.random.default_rng(seed)
    n = len(k0)
    fx, fy, cx, cy = synth.intrinsics(W, H)
    z = fp["depth0"][k0["y"].astype(int), k0["x"].astype(int)]
    has_mp = ((z > 0) & (rng.random(n) < 0.8)).astype(np.uint8)      # key points with depth mostly own a map point
    bad = (rng.random(n) < 0.03).astype(np.uint8)
    outl = (rng.random(n) < 0.05).astype(np.uint8)
    xw = np.zeros((n, 3), np.float32)
    xw[:, 0] = (k0["x"] - np.float32(cx)) * z / np.float32(fx)
    xw[:, 1] = (k0["y"] - np.float32(cy)) * z / np.float32(fy)
    xw[:, 2] = z                                  # frame 0 is the world frame
    xw[rng.random(n) < 0.04] *= -1                # a few points behind the camera / outside the image: 2-D fallback
    Tcw = np.linalg.inv(fp["T_01"])               # world (frame 0) -> current camera
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(Tcw[:3, :3]).as_quat().astype(np.float32)
    t = Tcw[:3, 3].astype(np.float32)
    K8 = np.array([fx, fy, cx, cy, 0, W, 0, H], np.float32)
    F_THR, DIST = 1.0, 10
    want_good, w3, w2, wmask = _reference_transcription(oracle, fp["gray0"], fp["gray1"], WIN, k0, has_mp, bad, outl, xw, k1, q, t, K8,
                                                        F_THR, DIST)
    mask = np.zeros((H, W), np.uint8)
    o3, o2 = np.zeros(n, gpu_api.KP_DTYPE), np.zeros(n, gpu_api.KP_DTYPE)
    i3, i2 = np.zeros(n, np.int32), np.zeros(n, np.int32)
    n3, n2 = C.c_int32(), C.c_int32()
    g0, g1 = np.ascontiguousarray(fp["gray0"]), np.ascontiguousarray(fp["gray1"])
    k0c, k1c = np.ascontiguousarray(k0), np.ascontiguousarray(k1)
    good = harness.sbp_of_test(g0.ctypes.data, g1.ctypes.data, W, H, WIN, n, k0c.ctypes.data, has_mp.ctypes.data, bad.ctypes.data,
                               outl.ctypes.data, xw.ctypes.data, len(k1c), k1c.ctypes.data, q.ctypes.data, t.ctypes.data, K8.ctypes.data,
                               F_THR, DIST, mask.ctypes.data, o3.ctypes.data, i3.ctypes.data, C.addressof(n3), o2.ctypes.data,
                               i2.ctypes.data, C.addressof(n2))
    assert good == want_good and n3.value == len(w3) and n2.value == len(w2)
    assert want_good > 50 and len(w3) > 20 and len(w2) > 3
    for (got_k, got_i, want) in ((o3, i3, w3), (o2, i2, w2)):
        for j, (i, x, y) in enumerate(want):
            assert got_i[j] == i and got_k[j]["x"] == x and got_k[j]["y"] == y           # bit-exact positions
            assert got_k[j]["octave"] == k0[i]["octave"] and got_k[j]["angle"] == k0[i]["angle"]   # "the other properties as before"
    assert np.array_equal(mask, wmask)
