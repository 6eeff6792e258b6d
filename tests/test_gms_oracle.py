"""CPU checks of the GMS restatement (oracle/gms_oracle.cpp; reference Thirdparty/GMS/include/gms_matcher.h, vendored).
The oracle is checked against an independent numpy transcription of gms_matcher::run(1) with the dense 400 x 400 vote matrix."""
import numpy as np
import pytest

from geoflowslam_amd import synth
from oracle import oracle as O


def _numpy_gms(kp1, size1, kp2, size2, q, t):
    f32 = np.float32
    p1 = np.stack([kp1["x"].astype(f32) / f32(size1[0]), kp1["y"].astype(f32) / f32(size1[1])], 1).astype(f32)
    p2 = np.stack([kp2["x"].astype(f32) / f32(size2[0]), kp2["y"].astype(f32) / f32(size2[1])], 1).astype(f32)
    n = len(q)
    mask = np.zeros(n, bool)
    pair_r = np.zeros(n, np.int64)

    def nb9(idx):
        out = [-1] * 9
        ix, iy = idx % 20, idx // 20
        for yi in (-1, 0, 1):
            for xi in (-1, 0, 1):
                xx, yy = ix + xi, iy + yi
                if 0 <= xx < 20 and 0 <= yy < 20:
                    out[xi + 4 + yi * 3] = xx + yy * 20
        return out

    for gtype in (1, 2, 3, 4):
        stats = np.zeros((400, 400), np.int64)
        npts = np.zeros(400, np.int64)
        pair_l = np.zeros(n, np.int64)
        for i in range(n):
            lx, ly = p1[q[i]]
            vx, vy = f32(lx * f32(20)), f32(ly * f32(20))
            x = int(np.floor(vx)) if gtype in (1, 3) else int(np.floor(float(vx) + 0.5))
            y = int(np.floor(vy)) if gtype in (1, 2) else int(np.floor(float(vy) + 0.5))
            l = -1 if (x >= 20 or y >= 20) else x + y * 20
            pair_l[i] = l
            if gtype == 1:
                rx, ry = p2[t[i]]
                pair_r[i] = int(np.floor(f32(rx * f32(20)))) + int(np.floor(f32(ry * f32(20)))) * 20
            r = pair_r[i]
            if l < 0 or r < 0 or l >= 400 or r >= 400:
                continue
            stats[l, r] += 1
            npts[l] += 1
        cell = np.full(400, -1, np.int64)
        for i in range(400):
            if stats[i].sum() == 0:
                continue
            cell[i] = int(np.argmax(stats[i]))  # first maximum
            nl, nr = nb9(i), nb9(cell[i])
            score, thresh, numpair = 0, 0.0, 0
            for j in range(9):
                if nl[j] == -1 or nr[j] == -1:
                    continue
                score += stats[nl[j], nr[j]]
                thresh += npts[nl[j]]
                numpair += 1
            if score < 6 * np.sqrt(thresh / numpair):
                cell[i] = -2
        for i in range(n):
            if pair_l[i] >= 0 and cell[pair_l[i]] == pair_r[i]:
                mask[i] = True
    return mask, int(mask.sum())


def _orb_matches(seed, w=640, h=480):
    fp = synth.frame_pair(seed, w, h, 8)
    orc = O.OrbOracle(1000, 1.2, 8, 20, 7)
    _, k0, d0 = orc.extract(fp["gray0"])
    _, k1, d1 = orc.extract(fp["gray1"])
    ti, _ = O.bf_match(d0, d1)
    return k0, k1, np.arange(len(ti), dtype=np.int32), ti.astype(np.int32)


@pytest.mark.parametrize("seed", [3, 4])
def test_matches_dense_matrix_transcription(seed):
    k0, k1, q, t = _orb_matches(seed)
    m, n = O.gms_inlier_mask(k0, (640, 480), k1, (640, 480), q, t)
    mb, nb = _numpy_gms(k0, (640, 480), k1, (640, 480), q, t)
    assert n == nb and np.array_equal(m, mb)
    assert 50 < n < len(q)  # a real filter: keeps the motion-consistent part of the brute-force matches


def test_border_key_points_and_random_matches():
    """Key-points in the last 2.5 % of the image take the `return -1` branch of the shifted grids; random matches get no support."""
    rng = np.random.default_rng(0)
    kp = np.zeros(600, dtype=[("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    kp["x"] = rng.uniform(0, 639.99, 600).astype(np.float32)
    kp["y"] = rng.uniform(0, 479.99, 600).astype(np.float32)
    kp["x"][:40] = rng.uniform(630, 639.9, 40).astype(np.float32)
    kp["y"][40:80] = rng.uniform(470, 479.9, 40).astype(np.float32)
    kp2 = kp.copy()
    kp2["x"] = np.clip(kp["x"] + 3.0, 0, 639.9).astype(np.float32)
    q = np.arange(600, dtype=np.int32)
    t = q.copy()
    t[300:] = rng.integers(0, 600, 300)  # half of the matches are random
    m, n = O.gms_inlier_mask(kp, (640, 480), kp2, (640, 480), q, t)
    mb, nb = _numpy_gms(kp, (640, 480), kp2, (640, 480), q, t)
    assert n == nb and np.array_equal(m, mb)
    assert m[:300].mean() > 0.25 and m[:300].mean() > 3 * max(m[300:].mean(), 0.01)


def test_empty():
    kp = np.zeros(0, dtype=[("x", "<f4"), ("y", "<f4")])
    m, n = O.gms_inlier_mask(kp, (640, 480), kp, (640, 480), np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert n == 0 and len(m) == 0
