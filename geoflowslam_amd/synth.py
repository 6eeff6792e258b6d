"""Seeded synthetic RGB-D inputs for tests and bench (SURVEY.md §8(d) "Synthetic inputs").

A tiny analytic ray-caster: a room (floor + two walls) and a few axis-aligned boxes, textured with a
procedural multi-scale checker so that FAST corners exist at every pyramid level and so that two views
of the same scene are photo-consistent (ORB matches and GICP have a true answer).  Harness code only —
nothing here is part of the hot path.
"""
import numpy as np


def intrinsics(width, height):
    """fx=fy=607 for VGA, 910 for 720p (script/run_orbslam/RGBD-Inertial/config/g1_op_icp_lidar_indoor1.yaml:25-34)."""
    f = 607.0 * width / 640.0
    return f, f, (width - 1) / 2.0, (height - 1) / 2.0


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def random_motion(rng, trans=0.03, rot_deg=1.5):
    """SE(3) T_world_cam of the second view: t ~ U(-3,3) cm, r ~ U(-1.5,1.5) deg per axis."""
    T = np.eye(4)
    T[:3, :3] = _rot(*(np.deg2rad(rot_deg) * rng.uniform(-1, 1, 3)))
    T[:3, 3] = trans * rng.uniform(-1, 1, 3)
    return T


class Scene:
    def __init__(self, seed):
        rng = np.random.default_rng(seed)
        self.seed = int(seed)
        # planes: (normal, offset) with n.p = d ; camera looks along +z, y down
        self.planes = [
            (np.array([0.0, 1.0, 0.0]), 1.3 + 0.3 * rng.random()),   # floor (y = +h below the camera)
            (np.array([0.0, 0.0, 1.0]), 4.0 + 1.8 * rng.random()),   # front wall
            (np.array([1.0, 0.0, 0.0]), -(2.0 + 1.0 * rng.random())),  # left wall x = -a
        ]
        self.boxes = []
        for _ in range(5):
            c = np.array([rng.uniform(-1.5, 2.0), rng.uniform(0.2, 1.0), rng.uniform(1.5, 3.8)])
            h = np.array([rng.uniform(0.15, 0.5), rng.uniform(0.15, 0.5), rng.uniform(0.15, 0.4)])
            self.boxes.append((c - h, c + h))
        nsurf = 3 + 6 * 5
        self.freq = rng.uniform(2.5, 9.0, (nsurf, 3, 2)).astype(np.float32)
        self.phase = rng.uniform(0, 6.28, (nsurf, 3, 2)).astype(np.float32)
        self.amp = rng.uniform(15, 45, (nsurf, 3)).astype(np.float32)
        self.base = rng.uniform(70, 180, nsurf).astype(np.float32)

    def render(self, width, height, T_wc=None, noise_seed=0, depth_noise=0.002, invalid_frac=0.02):
        """-> gray u8 [H,W], depth f32 [H,W] (metres, 0 = invalid)."""
        fx, fy, cx, cy = intrinsics(width, height)
        T = np.eye(4) if T_wc is None else T_wc
        R, t = T[:3, :3].astype(np.float32), T[:3, 3].astype(np.float32)
        u, v = np.meshgrid(np.arange(width, dtype=np.float32), np.arange(height, dtype=np.float32))
        dc = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1).reshape(-1, 3)
        d = dc @ R.T
        o = t
        n = d.shape[0]
        best = np.full(n, np.inf, np.float32)
        sid = np.zeros(n, np.int32)
        uv = np.zeros((n, 2), np.float32)

        def consider(tt, mask, s, ucoord, vcoord):
            nonlocal best, sid, uv
            m = mask & (tt > 0.05) & (tt < best)
            best = np.where(m, tt, best)
            sid = np.where(m, s, sid)
            uv[m, 0] = ucoord[m]
            uv[m, 1] = vcoord[m]

        s = 0
        for nrm, off in self.planes:
            denom = d @ nrm.astype(np.float32)
            with np.errstate(divide="ignore", invalid="ignore"):
                tt = (off - o @ nrm.astype(np.float32)) / denom
            p = o + d * tt[:, None]
            ax = [i for i in range(3) if abs(nrm[i]) < 0.5]
            consider(tt, np.isfinite(tt), s, p[:, ax[0]], p[:, ax[1]])
            s += 1
        for lo, hi in self.boxes:
            lo = lo.astype(np.float32)
            hi = hi.astype(np.float32)
            for axis in range(3):
                for side, val in ((0, lo[axis]), (1, hi[axis])):
                    with np.errstate(divide="ignore", invalid="ignore"):
                        tt = (val - o[axis]) / d[:, axis]
                    p = o + d * tt[:, None]
                    oth = [i for i in range(3) if i != axis]
                    inside = (np.isfinite(tt) & (p[:, oth[0]] >= lo[oth[0]]) & (p[:, oth[0]] <= hi[oth[0]])
                              & (p[:, oth[1]] >= lo[oth[1]]) & (p[:, oth[1]] <= hi[oth[1]]))
                    consider(tt, inside, s, p[:, oth[0]], p[:, oth[1]])
                    s += 1
        hit = np.isfinite(best)
        f, ph, am = self.freq[sid], self.phase[sid], self.amp[sid]
        val = self.base[sid].copy()
        for k in range(3):
            val += am[:, k] * np.sign(np.sin(f[:, k, 0] * (2.0 ** k) * uv[:, 0] + ph[:, k, 0])
                                      * np.sin(f[:, k, 1] * (2.0 ** k) * uv[:, 1] + ph[:, k, 1]))
        rng = np.random.default_rng((self.seed * 7919 + noise_seed) & 0x7FFFFFFF)
        val = np.where(hit, val, 20.0) + rng.integers(-2, 3, n)
        gray = np.clip(np.rint(val), 0, 255).astype(np.uint8).reshape(height, width)
        z = (best * dc[:, 2]).astype(np.float32)  # depth along the optical axis (dc.z == 1)
        z = z + rng.normal(0, depth_noise, n).astype(np.float32)
        z = np.where(hit & (z > 0.3) & (z < 9.5), z, 0).astype(np.float32)
        z[rng.random(n) < invalid_frac] = 0
        return gray, z.reshape(height, width)


def depth_to_cloud(depth, stride, width=None, height=None):
    """Frame::ConvertDepthToPointCloud (reference src/Frame.cc:590-623): pixels on a stride grid with
    0 < d < 10 -> (x, y, z, 1) float32 in the camera frame. Returns [N,4] float32."""
    h, w = depth.shape
    fx, fy, cx, cy = intrinsics(w, h)
    vs, us = np.meshgrid(np.arange(0, h, stride), np.arange(0, w, stride), indexing="ij")
    d = depth[vs, us]
    m = (d > 0) & (d < 10)
    d = d[m].astype(np.float32)
    x = ((us[m].astype(np.float32) - np.float32(cx)) * d / np.float32(fx)).astype(np.float32)
    y = ((vs[m].astype(np.float32) - np.float32(cy)) * d / np.float32(fy)).astype(np.float32)
    return np.stack([x, y, d, np.ones_like(d)], -1).astype(np.float32)


def frame_pair(seed, width=640, height=480, stride=4):
    """-> dict(gray0, depth0, gray1, depth1, cloud0, cloud1, T_01) where T_01 maps frame-1 points into
    frame 0 (= T_target_source for GICP with target = previous frame, source = current frame,
    reference src/Tracking.cc:3375-3382)."""
    sc = Scene(seed)
    rng = np.random.default_rng(seed + 0x6F5)
    T1 = random_motion(rng)
    g0, d0 = sc.render(width, height, None, 0)
    g1, d1 = sc.render(width, height, T1, 1)
    return dict(gray0=g0, depth0=d0, gray1=g1, depth1=d1, cloud0=depth_to_cloud(d0, stride),
                cloud1=depth_to_cloud(d1, stride), T_01=T1)


def noise_image(seed, width, height):
    """Unstructured test image: smooth background + rectangles + noise (lots of corners)."""
    rng = np.random.default_rng(seed)
    img = rng.uniform(60, 190) + 25 * np.sin(np.linspace(0, 6, width))[None, :] * np.cos(np.linspace(0, 4, height))[:, None]
    for _ in range(max(40, width * height // 800)):
        x0, y0 = rng.integers(0, width), rng.integers(0, height)
        w, h = rng.integers(4, 60), rng.integers(4, 60)
        img[y0:y0 + h, x0:x0 + w] += rng.uniform(30, 120) * rng.choice([-1, 1])
    img += rng.integers(-2, 3, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _quat_from_R(R):
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(R).as_quat()  # x, y, z, w
    return q if q[3] >= 0 else -q


def lba_window(seed, n_free=20, n_fixed=5, n_points=3000, mono_frac=0.1, outlier_frac=0.03):
    """Synthetic LocalBundleAdjustment window (BASELINE.json configs[4]): key-frames on a 2 m arc
    looking at `n_points` points in a 6x4x3 m box; every point is observed by every key-frame whose frustum contains it;
    RGB-D ("stereo", 3-D) edges, a fraction without depth (mono, 2-D); pixel noise sigma = sqrt(sigma2[octave]);
    3 % gross outliers (20 px); key-frame poses perturbed by 1 cm / 0.3 deg, points by 2 cm.
    Returns a dict of flat arrays matching gfs_lba_problem (include/gfs_abi.h) plus the ground truth."""
    rng = np.random.default_rng(seed)
    fx = fy = np.float64(np.float32(607.0))
    cx, cy = np.float64(np.float32(319.5)), np.float64(np.float32(239.5))
    bf = np.float64(np.float32(0.0745 * 607.0))
    n_poses = n_free + n_fixed
    pts = np.c_[rng.uniform(-3, 3, n_points), rng.uniform(-2, 2, n_points), rng.uniform(3.0, 6.0, n_points)]
    Rs, ts = [], []
    for i in range(n_poses):
        a = (i / max(n_poses - 1, 1) - 0.5) * 0.8  # arc angle
        c = np.array([2.0 * np.sin(a), 0.05 * rng.normal(), 2.0 - 2.0 * np.cos(a)])  # camera centre (world)
        Rwc = _rot(0.02 * rng.normal(), -a + 0.02 * rng.normal(), 0.02 * rng.normal())
        Rcw = Rwc.T
        Rs.append(Rcw)
        ts.append(-Rcw @ c)
    sigma2 = np.float64(np.float32(1.2) ** (2 * np.arange(8))).astype(np.float64)
    inv_sigma2 = np.float64(np.float32(1.0) / np.float32(1.2) ** (2 * np.arange(8)))
    e_pose, e_point, e_obs, e_is2, e_stereo = [], [], [], [], []
    for j in range(n_points):  # point-major edge order like the reference (src/Optimizer.cc:1816-1952)
        for i in range(n_poses):
            xc = Rs[i] @ pts[j] + ts[i]
            if xc[2] < 0.3:
                continue
            u, v = fx * xc[0] / xc[2] + cx, fy * xc[1] / xc[2] + cy
            if not (0 <= u < 640 and 0 <= v < 480):
                continue
            octv = int(rng.choice(8, p=np.array([217, 181, 151, 126, 105, 87, 73, 60]) / 1000.0))
            s = np.sqrt(sigma2[octv])
            noise = rng.normal(0, s, 3)
            if rng.random() < outlier_frac:
                noise[:2] += rng.choice([-1, 1], 2) * 20.0
            stereo = rng.random() >= mono_frac
            ur = u - bf / xc[2]
            e_pose.append(i)
            e_point.append(j)
            e_obs.append([np.float32(u + noise[0]), np.float32(v + noise[1]), np.float32(ur + noise[2]) if stereo else -1.0])
            e_is2.append(inv_sigma2[octv])
            e_stereo.append(1 if stereo else 0)
    fixed = np.zeros(n_poses, np.uint8)
    fixed[n_free:] = 1
    fixed[0] = 1 if n_fixed == 0 else fixed[0]
    q_gt = np.array([_quat_from_R(R) for R in Rs])
    t_gt = np.array(ts)
    q0, t0 = q_gt.copy(), t_gt.copy()
    for i in range(n_poses):
        if fixed[i]:
            continue
        dR = _rot(*(np.deg2rad(0.3) * rng.normal(size=3)))
        q0[i] = _quat_from_R(dR @ Rs[i])
        t0[i] = dR @ ts[i] + 0.01 * rng.normal(size=3)
    # the reference stores poses / points as float and widens them (src/Optimizer.cc:1692-1694, 1821)
    q0 = q0.astype(np.float32).astype(np.float64)
    t0 = t0.astype(np.float32).astype(np.float64)
    p0 = (pts + 0.02 * rng.normal(size=pts.shape)).astype(np.float32).astype(np.float64)
    return dict(n_poses=n_poses, n_points=n_points, n_edges=len(e_pose), pose_q=np.ascontiguousarray(q0),
                pose_t=np.ascontiguousarray(t0), pose_fixed=fixed, points=np.ascontiguousarray(p0),
                edge_pose=np.array(e_pose, np.int32), edge_point=np.array(e_point, np.int32),
                edge_obs=np.array(e_obs, np.float64), edge_inv_sigma2=np.array(e_is2, np.float64),
                edge_stereo=np.array(e_stereo, np.uint8), fx=fx, fy=fy, cx=cx, cy=cy, bf=bf,
                huber_mono=float(np.float32(np.sqrt(5.991))), huber_stereo=float(np.float32(np.sqrt(7.815))),
                iterations=10, gt_q=q_gt, gt_t=t_gt, gt_points=pts)


def pose_frame(seed, n_obs=300, mono_frac=0.15, outlier_frac=0.1, rot_deg=1.0, trans=0.03, outlier_px=25.0, noise_scale=1.0):
    """Synthetic Optimizer::PoseOptimization problem (reference src/Optimizer.cc:763-1098): one frame looking at `n_obs`
    map points in a 6x4x3 m box, RGB-D ("stereo", 3-D) observations with a fraction without depth (mono, 2-D), pixel noise
    sigma = sqrt(sigma2[octave]), `outlier_frac` gross outliers, initial pose off by ~rot_deg / ~trans metres.
    Returns the flat arrays of gfs_pose_problem (include/gfs_abi.h) plus the ground truth pose (q_gt, t_gt)."""
    rng = np.random.default_rng(seed)
    fx = fy = np.float64(np.float32(607.0))
    cx, cy = np.float64(np.float32(319.5)), np.float64(np.float32(239.5))
    bf = np.float64(np.float32(0.0745 * 607.0))
    Rcw = _rot(0.05 * rng.normal(), 0.2 * rng.normal(), 0.05 * rng.normal())
    tcw = np.array([0.3 * rng.normal(), 0.1 * rng.normal(), 0.2 * rng.normal()])
    sigma2 = np.float64(np.float32(1.2) ** (2 * np.arange(8)))
    inv_sigma2 = (np.float32(1.0) / np.float32(1.2) ** (2 * np.arange(8))).astype(np.float32)
    xw, obs, w, st, is_out = [], [], [], [], []
    while len(xw) < n_obs:
        xc = np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(1.0, 6.0)])
        u, v = fx * xc[0] / xc[2] + cx, fy * xc[1] / xc[2] + cy
        if not (0 <= u < 640 and 0 <= v < 480):
            continue
        octv = int(rng.choice(8, p=np.array([217, 181, 151, 126, 105, 87, 73, 60]) / 1000.0))
        noise = noise_scale * rng.normal(0, np.sqrt(sigma2[octv]), 3)
        out = rng.random() < outlier_frac
        if out:
            noise[:2] += rng.choice([-1, 1], 2) * outlier_px
        stereo = rng.random() >= mono_frac
        ur = u - bf / xc[2]
        xw.append((Rcw.T @ (xc - tcw)).astype(np.float32).astype(np.float64))  # MapPoint world positions are floats
        obs.append([np.float32(u + noise[0]), np.float32(v + noise[1]), np.float32(ur + noise[2]) if stereo else -1.0])
        w.append(inv_sigma2[octv])
        st.append(1 if stereo else 0)
        is_out.append(out)
    dR = _rot(*(np.deg2rad(rot_deg) * rng.normal(size=3)))
    q0 = _quat_from_R(dR @ Rcw).astype(np.float32).astype(np.float64)  # Sophus::SE3f pose widened to double
    t0 = (dR @ tcw + trans * rng.normal(size=3)).astype(np.float32).astype(np.float64)
    return dict(q=q0, t=t0, n_obs=n_obs, xw=np.array(xw), obs=np.array(obs, np.float64), inv_sigma2=np.array(w, np.float32),
                stereo=np.array(st, np.uint8), fx=fx, fy=fy, cx=cx, cy=cy, bf=bf, n_rounds=4, its=10,
                q_gt=_quat_from_R(Rcw), t_gt=tcw, is_outlier=np.array(is_out))


def sbp_pair(seed, n_points=900, n_extra_cur=250, motion=0.04, rot_deg=0.8, th=7.0, mono=False, desc_flip_bits=18,
             dup_frac=0.0, zero_obs_frac=0.0, preassigned_frac=0.0, n_levels=8, check_orientation=True):
    """Synthetic input of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (reference
    src/ORBmatcher.cc:1853-2063), flattened as gfs_sbp_problem (include/gfs_abi.h).

    `n_points` map points seen by the last frame; the current frame (pose = last pose composed with a small motion) sees
    most of them again (key-point = projection + sub-pixel noise, octave within +-1, descriptor = map-point descriptor with
    `desc_flip_bits` random bit flips, angle = last angle + small rotation) plus `n_extra_cur` unrelated key-points.
    dup_frac: fraction of map points duplicated (two map points competing for the same key-point -> the order-dependent
    skip of :1914-1915); zero_obs_frac: map points with Observations() == 0 (their assignment may be overwritten);
    preassigned_frac: current key-points that already hold a map point with observations."""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    fx = fy = f32(607.0)
    cx, cy = f32(319.5), f32(239.5)
    bf = f32(0.0745 * 607.0)
    b = f32(0.0745)
    W, H = 640, 480
    min_x, max_x, min_y, max_y = f32(0), f32(W), f32(0), f32(H)
    scale = np.cumprod(np.r_[1.0, np.full(n_levels - 1, 1.2)]).astype(np.float32)
    Rlw = _rot(0.03 * rng.normal(), 0.2 * rng.normal(), 0.03 * rng.normal())
    tlw = np.array([0.2 * rng.normal(), 0.05 * rng.normal(), 0.1 * rng.normal()])
    dR = _rot(*(np.deg2rad(rot_deg) * rng.normal(size=3)))
    Rcw, tcw = dR @ Rlw, dR @ tlw + motion * rng.normal(size=3) + np.array([0, 0, -motion])
    kp_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                         ("octave", "<i4"), ("class_id", "<i4")])
    last_xw, last_desc, last_oct, last_ang, last_obs = [], [], [], [], []
    cur = []  # (x, y, octave, angle, u_right, desc)
    roll = np.deg2rad(rot_deg) * rng.normal()
    while len(last_xw) < n_points:
        xl = np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(1.0, 7.0)])  # in the last camera
        ul, vl = fx * xl[0] / xl[2] + cx, fy * xl[1] / xl[2] + cy
        if not (0 <= ul < W and 0 <= vl < H):
            continue
        xw = (Rlw.T @ (xl - tlw)).astype(np.float32)
        octv = int(rng.choice(n_levels, p=np.array([217, 181, 151, 126, 105, 87, 73, 60][:n_levels]) / sum([217, 181, 151, 126, 105, 87, 73, 60][:n_levels])))
        ang = f32(rng.uniform(0, 360))
        desc = rng.integers(0, 256, 32, dtype=np.uint8)
        reps = 2 if rng.random() < dup_frac else 1
        for _ in range(reps):
            last_xw.append(xw + (0 if _ == 0 else rng.normal(0, 0.002, 3).astype(np.float32)))
            d = desc.copy()
            if _ > 0:
                flip = rng.choice(256, 6, replace=False)
                np.bitwise_xor.at(d, flip // 8, (1 << (flip % 8)).astype(np.uint8))
            last_desc.append(d)
            last_oct.append(octv)
            last_ang.append(ang)
            last_obs.append(0 if rng.random() < zero_obs_frac else 1)
        xc = Rcw @ xw.astype(np.float64) + tcw
        if xc[2] <= 0.2:
            continue
        uc, vc = fx * xc[0] / xc[2] + cx, fy * xc[1] / xc[2] + cy
        if not (0 <= uc < W and 0 <= vc < H) or rng.random() < 0.15:
            continue
        d = desc.copy()
        flip = rng.choice(256, int(rng.integers(0, desc_flip_bits + 1)), replace=False)
        np.bitwise_xor.at(d, flip // 8, (1 << (flip % 8)).astype(np.uint8))
        o2 = int(np.clip(octv + rng.integers(-1, 2), 0, n_levels - 1))
        a2 = f32((ang + np.rad2deg(roll) + rng.normal(0, 3.0)) % 360.0)
        if rng.random() < 0.06:
            a2 = f32(rng.uniform(0, 360))  # inconsistent rotation -> removed by the histogram check
        x2, y2 = f32(uc + rng.normal(0, 0.7)), f32(vc + rng.normal(0, 0.7))
        ur = f32(x2 - bf / xc[2] + rng.normal(0, 0.5)) if (not mono and rng.random() < 0.85) else f32(-1)
        cur.append((x2, y2, o2, a2, ur, d))
    for _ in range(n_extra_cur):
        cur.append((f32(rng.uniform(0, W)), f32(rng.uniform(0, H)), int(rng.integers(0, n_levels)), f32(rng.uniform(0, 360)),
                    f32(rng.uniform(1, W)) if rng.random() < 0.5 else f32(-1), rng.integers(0, 256, 32, dtype=np.uint8)))
    order = rng.permutation(len(cur))
    cur = [cur[i] for i in order]
    kps = np.zeros(len(cur), kp_dtype)
    kps["x"] = [c[0] for c in cur]
    kps["y"] = [c[1] for c in cur]
    kps["octave"] = [c[2] for c in cur]
    kps["angle"] = [c[3] for c in cur]
    kps["size"] = 31.0
    kps["class_id"] = -1
    return dict(last_xw=np.array(last_xw, np.float32)[:n_points], last_desc=np.array(last_desc, np.uint8)[:n_points],
                last_octave=np.array(last_oct, np.int32)[:n_points], last_angle=np.array(last_ang, np.float32)[:n_points],
                last_mp_has_obs=np.array(last_obs, np.uint8)[:n_points], cur_kps_un=kps,
                cur_u_right=np.array([c[4] for c in cur], np.float32), cur_desc=np.array([c[5] for c in cur], np.uint8).reshape(-1, 32),
                cur_has_mp_obs=(rng.random(len(cur)) < preassigned_frac).astype(np.uint8),
                Tcw_q=_quat_from_R(Rcw).astype(np.float32), Tcw_t=tcw.astype(np.float32),
                Tlw_q=_quat_from_R(Rlw).astype(np.float32), Tlw_t=tlw.astype(np.float32),
                fx=fx, fy=fy, cx=cx, cy=cy, bf=bf, b=b, min_x=min_x, max_x=max_x, min_y=min_y, max_y=max_y,
                grid_w_inv=f32(64) / (max_x - min_x), grid_h_inv=f32(48) / (max_y - min_y), scale_factors=scale,
                th=f32(th), mono=int(mono), check_orientation=int(check_orientation))


def sbp_map_frame(seed, th=1.0, nn_ratio=0.8, **kw):
    """Synthetic input of ORBmatcher::SearchByProjection(F, vpMapPoints, th, ...) (reference src/ORBmatcher.cc:43-206) as
    gfs_sbp_map_problem: the scene of sbp_pair with the projections Frame::isInFrustum would leave on the map points
    (mTrackProjX / Y / XR, predicted level, viewing cosine), in double like the float pipeline's inputs rounded to float."""
    p = sbp_pair(seed, **kw)
    rng = np.random.default_rng(seed + 7919)
    R = _rot_from_quat(p["Tcw_q"].astype(np.float64))
    xc = p["last_xw"].astype(np.float64) @ R.T + p["Tcw_t"].astype(np.float64)
    keep = (xc[:, 2] > 0.2)
    u = p["fx"] * xc[:, 0] / xc[:, 2] + p["cx"]
    v = p["fy"] * xc[:, 1] / xc[:, 2] + p["cy"]
    keep &= (u >= 0) & (u < 640) & (v >= 0) & (v < 480)
    ur = u - p["bf"] / xc[:, 2]
    view_cos = np.where(rng.random(len(u)) < 0.5, 0.9995, rng.uniform(0.5, 0.998, len(u))).astype(np.float32)
    return dict(mp_proj=np.stack([u, v, ur], 1)[keep].astype(np.float32), mp_level=p["last_octave"][keep],
                mp_view_cos=view_cos[keep], mp_desc=p["last_desc"][keep], mp_has_obs=p["last_mp_has_obs"][keep],
                cur_kps_un=p["cur_kps_un"], cur_u_right=p["cur_u_right"], cur_desc=p["cur_desc"], cur_has_mp_obs=p["cur_has_mp_obs"],
                min_x=p["min_x"], min_y=p["min_y"], grid_w_inv=p["grid_w_inv"], grid_h_inv=p["grid_h_inv"],
                scale_factors=p["scale_factors"], th=np.float32(th), nn_ratio=np.float32(nn_ratio))


def _rot_from_quat(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def klt_texture_pair(seed, width=640, height=480, shift=(3.3, -2.1), rot_deg=0.0, gain=1.0, noise=1):
    """Two views of one analytic texture: image 1 samples the texture at R(p - c) + c + shift, so a point p in image 0 is seen at
    p' = R^T(p - c - shift) + c ... in image 1 (returned as the callable `flow`).  -> (img0, img1, flow)."""
    rng = np.random.default_rng(seed)
    nwave = 24
    k = rng.uniform(0.02, 0.45, (nwave, 2)) * rng.choice([-1, 1], (nwave, 2))
    ph = rng.uniform(0, 6.28, nwave)
    amp = rng.uniform(6, 22, nwave)
    base = rng.uniform(90, 150)
    th = np.deg2rad(rot_deg)
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    c = np.array([width / 2.0, height / 2.0])

    def tex(x, y):
        v = np.full(x.shape, base)
        for i in range(nwave):
            v += amp[i] * np.sin(k[i, 0] * x + k[i, 1] * y + ph[i])
        return v

    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    img0 = tex(u, v)
    # image 1 pixel q shows the texture point R (q - c) + c + shift
    q = np.stack([u - c[0], v - c[1]], -1) @ R.T
    img1 = gain * tex(q[..., 0] + c[0] + shift[0], q[..., 1] + c[1] + shift[1])
    nrng = np.random.default_rng(seed + 17)
    if noise:
        img0 = img0 + nrng.integers(-noise, noise + 1, img0.shape)
        img1 = img1 + nrng.integers(-noise, noise + 1, img1.shape)

    def flow(p):
        """Position in image 1 of the texture point seen at p ([n, 2]) in image 0."""
        p = np.asarray(p, np.float64)
        return (p - c - np.asarray(shift)) @ R + c

    to_u8 = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return to_u8(img0), to_u8(img1), flow


def two_view_points(seed, n=400, outlier_frac=0.25, noise=0.3, width=640, height=480, trans=0.15, rot_deg=3.0):
    """Image points of n random 3-D points in two views related by a general motion (so that a fundamental matrix exists), pixel
    noise `noise`, a fraction replaced by gross mismatches.  -> (pts1 f32 [n, 2], pts2 f32 [n, 2], inlier bool [n], F_true [3, 3])."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = intrinsics(width, height)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    X = np.c_[rng.uniform(-2.5, 2.5, n), rng.uniform(-1.8, 1.8, n), rng.uniform(2.0, 7.0, n)]
    R = _rot(*np.deg2rad(rng.uniform(-rot_deg, rot_deg, 3)))
    t = rng.uniform(-trans, trans, 3)
    t[0] += trans  # never a pure rotation
    x1 = (K @ X.T).T
    x2 = (K @ (R @ X.T + t[:, None])).T
    p1 = x1[:, :2] / x1[:, 2:3] + rng.normal(0, noise, (n, 2))
    p2 = x2[:, :2] / x2[:, 2:3] + rng.normal(0, noise, (n, 2))
    out = rng.random(n) < outlier_frac
    p2[out] = np.c_[rng.uniform(0, width, out.sum()), rng.uniform(0, height, out.sum())]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.linalg.inv(K)
    F = Ki.T @ tx @ R @ Ki
    return p1.astype(np.float32), p2.astype(np.float32), ~out, F / F[2, 2]


def cloud_pair(seed, width=160, height=120, trans=0.03, rot_deg=1.5):
    """Clouds only (no images): the depth maps are rendered at width x height and unprojected at stride 1, which gives the point
    density of a (4 width) x (4 height) depth image sampled at stride 4 (the intrinsics scale with the width) at 1/16 of the
    rendering cost.  -> (cloud0, cloud1, T_01)."""
    sc = Scene(seed)
    rng = np.random.default_rng(seed + 0x6F5)
    T1 = random_motion(rng, trans, rot_deg)
    d0 = sc.render(width, height, None, 0)[1]
    d1 = sc.render(width, height, T1, 1)[1]
    return depth_to_cloud(d0, 1), depth_to_cloud(d1, 1), T1
