"""CPU checks of the ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) restatement (oracle/sbp_oracle.cpp;
reference src/ORBmatcher.cc:1853-2063).  PARITY UNPINNED (the reference needs OpenCV / Eigen / Sophus builds that are not
available here): property tests against a brute-force numpy restatement of the same loop and of the reference's quirks."""
import numpy as np
import pytest

from geoflowslam_amd import synth
from oracle import oracle as O


def _popcnt(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def _bruteforce(p):
    """Straight Python transcription of the loop (float32 arithmetic through numpy scalars) without the grid: every current
    key-point is tested against the window, candidates are visited in (cell x, cell y, index) order like GetFeaturesInArea."""
    f32 = np.float32
    kps = p["cur_kps_un"]
    N = len(kps)

    def act(q, v):
        q = q.astype(f32); v = v.astype(f32)
        uv = np.cross(q[:3], v).astype(f32)
        uv = (uv + uv).astype(f32)
        return ((v + q[3] * uv).astype(f32) + np.cross(q[:3], uv).astype(f32)).astype(f32)

    gw, gh = f32(p["grid_w_inv"]), f32(p["grid_h_inv"])
    cell = []
    for i in range(N):
        px = int(np.round((kps["x"][i] - p["min_x"]) * gw)) if True else 0
        py = int(np.round((kps["y"][i] - p["min_y"]) * gh))
        # np.round is half-to-even, std::round half-away: redo exactly
        fxv, fyv = float((kps["x"][i] - p["min_x"]) * gw), float((kps["y"][i] - p["min_y"]) * gh)
        px, py = int(np.floor(abs(fxv) + 0.5) * np.sign(fxv)), int(np.floor(abs(fyv) + 0.5) * np.sign(fyv))
        cell.append((px, py) if (0 <= px < 64 and 0 <= py < 48) else None)
    qc = p["Tcw_q"].astype(f32)
    qi = np.array([-qc[0], -qc[1], -qc[2], qc[3]], f32)
    qi = (qi / np.sqrt((qi[0] * qi[0] + qi[2] * qi[2]) + (qi[1] * qi[1] + qi[3] * qi[3]))).astype(f32)
    twc = act(qi, (p["Tcw_t"].astype(f32) * f32(-1)))
    tlc = (act(p["Tlw_q"], twc) + p["Tlw_t"].astype(f32)).astype(f32)
    fwd = bool(tlc[2] > p["b"]) and not p["mono"]
    bwd = bool(-tlc[2] > p["b"]) and not p["mono"]
    state = np.full(N, -1, np.int64)
    hist = [[] for _ in range(30)]
    nm = 0
    for l in range(len(p["last_xw"])):
        xc = (act(qc, p["last_xw"][l]) + p["Tcw_t"].astype(f32)).astype(f32)
        invz = f32(1.0 / float(xc[2]))
        if invz < 0:
            continue
        u = f32(f32(f32(p["fx"] * xc[0]) / xc[2]) + p["cx"])
        v = f32(f32(f32(p["fy"] * xc[1]) / xc[2]) + p["cy"])
        if u < p["min_x"] or u > p["max_x"] or v < p["min_y"] or v > p["max_y"]:
            continue
        octv = int(p["last_octave"][l])
        r = f32(p["th"] * p["scale_factors"][octv])
        lo, hi = (octv, -1) if fwd else ((0, octv) if bwd else (octv - 1, octv + 1))
        x0 = max(0, int(np.floor(f32(f32(f32(u - p["min_x"]) - r) * gw))))
        x1 = min(63, int(np.ceil(f32(f32(f32(u - p["min_x"]) + r) * gw))))
        y0 = max(0, int(np.floor(f32(f32(f32(v - p["min_y"]) - r) * gh))))
        y1 = min(47, int(np.ceil(f32(f32(f32(v - p["min_y"]) + r) * gh))))
        if x0 >= 64 or x1 < 0 or y0 >= 48 or y1 < 0:
            continue
        check = lo > 0 or hi >= 0
        cands = []
        for i2 in range(N):
            c = cell[i2]
            if c is None or not (x0 <= c[0] <= x1 and y0 <= c[1] <= y1):
                continue
            if check and (kps["octave"][i2] < lo or (hi >= 0 and kps["octave"][i2] > hi)):
                continue
            if not (abs(f32(kps["x"][i2] - u)) < r and abs(f32(kps["y"][i2] - v)) < r):
                continue
            cands.append((c[0], c[1], i2))
        if not cands:
            continue
        cands.sort()
        best, bi = 256, -1
        for _, _, i2 in cands:
            blocked = bool(p["cur_has_mp_obs"][i2]) or (state[i2] >= 0 and p["last_mp_has_obs"][state[i2]])
            if blocked:
                continue
            if p["cur_u_right"][i2] > 0:
                ur = f32(u - f32(p["bf"] * invz))
                if abs(f32(ur - p["cur_u_right"][i2])) > r:
                    continue
            d = _popcnt(p["last_desc"][l], p["cur_desc"][i2])
            if d < best:
                best, bi = d, i2
        if best <= 100:
            state[bi] = l
            nm += 1
            if p["check_orientation"]:
                rot = f32(p["last_angle"][l] - kps["angle"][bi])
                if rot < 0:
                    rot = f32(rot + f32(360))
                x = float(f32(rot * f32(1.0 / 30)))
                b = int(np.floor(x + 0.5))
                hist[0 if b == 30 else b].append(bi)
    if p["check_orientation"]:
        sizes = [len(h) for h in hist]
        m1 = m2 = m3 = 0
        i1 = i2_ = i3 = -1
        for i, s in enumerate(sizes):
            if s > m1:
                m3, m2, m1, i3, i2_, i1 = m2, m1, s, i2_, i1, i
            elif s > m2:
                m3, m2, i3, i2_ = m2, s, i2_, i
            elif s > m3:
                m3, i3 = s, i
        if m2 < np.float32(0.1) * np.float32(m1):
            i2_ = i3 = -1
        elif m3 < np.float32(0.1) * np.float32(m1):
            i3 = -1
        for i in range(30):
            if i not in (i1, i2_, i3):
                for idx in hist[i]:
                    state[idx] = -2
                    nm -= 1
    return state.astype(np.int32), nm


@pytest.mark.parametrize("cfg", [dict(seed=1, n_points=300, n_extra_cur=80),
                                 dict(seed=2, n_points=250, n_extra_cur=60, dup_frac=0.3, zero_obs_frac=0.3, preassigned_frac=0.1, th=15.0, mono=True),
                                 dict(seed=3, n_points=200, n_extra_cur=50, motion=0.3),
                                 dict(seed=4, n_points=200, n_extra_cur=50, check_orientation=False, dup_frac=0.2)])
def test_matches_python_transcription(cfg):
    p = synth.sbp_pair(**cfg)
    m, n = O.search_by_projection(p)
    mb, nb = _bruteforce(p)
    assert n == nb
    assert np.array_equal(m, mb)


def test_finds_the_true_correspondences():
    p = synth.sbp_pair(5, n_points=800, n_extra_cur=200)
    m, n = O.search_by_projection(p)
    assert n > 400 and n == int((m >= 0).sum())  # no duplicates / zero-observation points: nothing is counted twice
    hit = m >= 0
    d = np.array([_popcnt(p["last_desc"][m[i]], p["cur_desc"][i]) for i in np.nonzero(hit)[0]])
    assert d.max() <= 100 and np.median(d) < 20


def test_overwritten_assignments_are_counted_twice():
    """A key-point taken by a map point without observations can be taken again by a later one: nmatches counts both."""
    p = synth.sbp_pair(6, n_points=400, n_extra_cur=50, dup_frac=0.5, zero_obs_frac=1.0, check_orientation=False)
    m, n = O.search_by_projection(p)
    assert n > int((m >= 0).sum())


def test_empty_inputs():
    p = synth.sbp_pair(7, n_points=50, n_extra_cur=10)
    e = dict(p, last_xw=np.zeros((0, 3), np.float32), last_desc=np.zeros((0, 32), np.uint8), last_octave=np.zeros(0, np.int32),
             last_angle=np.zeros(0, np.float32), last_mp_has_obs=np.zeros(0, np.uint8))
    m, n = O.search_by_projection(e)
    assert n == 0 and (m == -1).all()
    e2 = dict(p, cur_kps_un=p["cur_kps_un"][:0], cur_u_right=np.zeros(0, np.float32), cur_desc=np.zeros((0, 32), np.uint8),
              cur_has_mp_obs=np.zeros(0, np.uint8))
    m, n = O.search_by_projection(e2)
    assert n == 0 and len(m) == 0


def _bruteforce_map(p):
    """Python transcription of ORBmatcher::SearchByProjection(F, vpMapPoints, th, ...) (src/ORBmatcher.cc:43-206)."""
    f32 = np.float32
    kps = p["cur_kps_un"]
    N = len(kps)
    gw, gh = f32(p["grid_w_inv"]), f32(p["grid_h_inv"])
    cell = []
    for i in range(N):
        fxv, fyv = float((kps["x"][i] - p["min_x"]) * gw), float((kps["y"][i] - p["min_y"]) * gh)
        px, py = int(np.floor(abs(fxv) + 0.5) * np.sign(fxv)), int(np.floor(abs(fyv) + 0.5) * np.sign(fyv))
        cell.append((px, py) if (0 <= px < 64 and 0 <= py < 48) else None)
    state = np.full(N, -1, np.int64)
    nm = 0
    for l in range(len(p["mp_proj"])):
        level = int(p["mp_level"][l])
        r = f32(2.5) if float(p["mp_view_cos"][l]) > 0.998 else f32(4.0)
        if float(p["th"]) != 1.0:
            r = f32(r * p["th"])
        x, y, xr = (f32(t) for t in p["mp_proj"][l])
        rad = f32(r * p["scale_factors"][level])
        x0 = max(0, int(np.floor(f32(f32(f32(x - p["min_x"]) - rad) * gw))))
        x1 = min(63, int(np.ceil(f32(f32(f32(x - p["min_x"]) + rad) * gw))))
        y0 = max(0, int(np.floor(f32(f32(f32(y - p["min_y"]) - rad) * gh))))
        y1 = min(47, int(np.ceil(f32(f32(f32(y - p["min_y"]) + rad) * gh))))
        if x0 >= 64 or x1 < 0 or y0 >= 48 or y1 < 0:
            continue
        cands = []
        for i2 in range(N):
            c = cell[i2]
            if c is None or not (x0 <= c[0] <= x1 and y0 <= c[1] <= y1):
                continue
            if kps["octave"][i2] < level - 1 or kps["octave"][i2] > level:
                continue
            if not (abs(f32(kps["x"][i2] - x)) < rad and abs(f32(kps["y"][i2] - y)) < rad):
                continue
            cands.append((c[0], c[1], i2))
        if not cands:
            continue
        cands.sort()
        bd, bl, bd2, bl2, bi = 256, -1, 256, -1, -1
        for _, _, i2 in cands:
            if bool(p["cur_has_mp_obs"][i2]) or (state[i2] >= 0 and p["mp_has_obs"][state[i2]]):
                continue
            if p["cur_u_right"][i2] > 0 and abs(f32(xr - p["cur_u_right"][i2])) > rad:
                continue
            d = _popcnt(p["mp_desc"][l], p["cur_desc"][i2])
            if d < bd:
                bd2, bd, bl2, bl, bi = bd, d, bl, int(kps["octave"][i2]), i2
            elif d < bd2:
                bl2, bd2 = int(kps["octave"][i2]), d
        if bd <= 100:
            if bl == bl2 and f32(bd) > f32(p["nn_ratio"] * f32(bd2)):
                continue
            state[bi] = l
            nm += 1
    return state.astype(np.int32), nm


@pytest.mark.parametrize("cfg", [dict(seed=1, n_points=300, n_extra_cur=80),
                                 dict(seed=2, n_points=250, n_extra_cur=60, th=3.0, dup_frac=0.3, zero_obs_frac=0.3, preassigned_frac=0.1),
                                 dict(seed=3, n_points=250, n_extra_cur=60, th=5.0, nn_ratio=0.6, desc_flip_bits=80)])
def test_map_variant_matches_python_transcription(cfg):
    p = synth.sbp_map_frame(**cfg)
    m, n = O.search_by_projection_map(p)
    mb, nb = _bruteforce_map(p)
    assert n == nb and np.array_equal(m, mb)


def test_map_variant_ratio_test_rejects_ambiguous_matches():
    """A twin key-point (same level, almost the same descriptor) next to every key-point makes best and second best
    comparable: with a strict mfNNratio the match is dropped (bestDist > mfNNratio * bestDist2), with ratio 1 it is kept."""
    p = synth.sbp_map_frame(4, n_points=200, n_extra_cur=0, th=5.0)
    k = p["cur_kps_un"]
    twin = k.copy()
    twin["x"] += np.float32(0.5)
    d2 = p["cur_desc"].copy()
    d2[:, 0] ^= 1  # one bit away from the original
    q = dict(p, cur_kps_un=np.concatenate([k, twin]), cur_u_right=np.concatenate([p["cur_u_right"]] * 2),
             cur_desc=np.concatenate([p["cur_desc"], d2]), cur_has_mp_obs=np.concatenate([p["cur_has_mp_obs"]] * 2))
    _, n_loose = O.search_by_projection_map(dict(q, nn_ratio=np.float32(1.0)))
    _, n_tight = O.search_by_projection_map(dict(q, nn_ratio=np.float32(0.3)))
    assert n_loose > 50 and n_tight < n_loose // 4
