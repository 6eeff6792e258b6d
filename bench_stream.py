"""bench_stream.py -- the single-stream leg of bench.py: ONE live RGB-D stream, one frame at a time, the way
System::TrackRGBD (reference src/System.cc:600) -> Tracking::GrabImageRGBD -> Frame::Frame -> Tracking::Track drives the path:

    ORB extraction of the new image                      Frame::ExtractORB            src/Frame.cc:ORBextractor::operator()
 -> mvuRight / mvDepth from the depth map                Frame::ComputeStereoFromRGBD src/Frame.cc:1314-1332   } one stage: the RGB-D tail
 -> depth map -> point cloud                             Frame::ConvertDepthToPointCloud src/Frame.cc:590-623  } of the Frame constructor
 -> GICP of the new cloud against the previous one       RegistrationGICP::RegisterPointClouds src/RegistrationGICP.cc:5-20
 -> windowed matching against the last frame's points    ORBmatcher::SearchByProjection(Frame&, const Frame&, th = 15, false) src/ORBmatcher.cc:1853-2063
 -> motion-only bundle adjustment                        Optimizer::PoseOptimization  src/Optimizer.cc:763-1098

The chain is written once over a small backend interface; bench.py runs it on the GPU through the C ABI (geoflowslam_amd.api;
the GICP step uses the streaming entry gfs_gicp_align_next: the previous frame's preprocessed cloud is kept on the device) and, as
the CPU figure beside it, on the oracle (which, like the reference, preprocesses both clouds in every call).  Latency is wall
time per frame on the calling thread, every host<->device copy and synchronisation included.
"""
import time

import numpy as np

TH_RGBD = 15.0  # Tracking::TrackWithMotionModel: th = 7 for stereo, 15 otherwise
BF = 40.0       # baseline x fx of the TUM RGB-D settings files (Examples/RGB-D/TUM1.yaml)


def _quat_from_R(R):
    """Unit quaternion (x, y, z, w), w >= 0, of a rotation matrix: the largest of (trace, diagonal) picks the branch (what
    scipy's Rotation.from_matrix(...).as_quat() computes; written out because that call costs ~0.1 ms, a twentieth of a frame)."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = (float(v) for v in np.asarray(R, np.float64).reshape(9))
    tr = m00 + m11 + m22
    if tr >= m00 and tr >= m11 and tr >= m22:
        q = [m21 - m12, m02 - m20, m10 - m01, 1.0 + tr]
    elif m00 >= m11 and m00 >= m22:
        q = [1.0 - tr + 2.0 * m00, m01 + m10, m02 + m20, m21 - m12]
    elif m11 >= m22:
        q = [m01 + m10, 1.0 - tr + 2.0 * m11, m12 + m21, m02 - m20]
    else:
        q = [m02 + m20, m12 + m21, 1.0 - tr + 2.0 * m22, m10 - m01]
    n = (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) ** 0.5
    s = (1.0 if q[3] >= 0 else -1.0) / n
    return np.array([q[0] * s, q[1] * s, q[2] * s, q[3] * s])


class GpuBackend:
    def __init__(self, api, W, H, nfeatures, nlevels, max_points, device=0, pose_sums=None):
        """pose_sums: None = the library's default (g2o's edge order: the reference's outlier flags bit for bit), "tree" = the opt-in
        fixed-shape tree sums (gfs_pose_set_sum_order)."""
        self.api, self.device = api, device
        self.ext = api.ORBextractor(nfeatures, 1.2, nlevels, 20, 7, max_rows=H, max_cols=W, max_batch=1, device=device)
        self.frm = api.Frame(max_rows=H, max_cols=W, max_keypoints=self.ext.cap, device=device)
        self.reg = api.RegistrationGICP(max_points=max_points, max_batch=1, device=device)
        self.pm = api.ProjectionMatcher(max_last=self.ext.cap, max_cur=self.ext.cap, max_batch=1, device=device)
        self.po = api.PoseOptimizer(max_obs=self.ext.cap, max_batch=1, device=device, sums=pose_sums)
        t = self.ext.tables()
        self.scale, self.inv_sigma2 = np.asarray(t["scale"], np.float32), np.asarray(t["inv_sigma2"], np.float32)
        self.have_target = False
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(1)
        self.no_kps = np.zeros(0, self.ext(np.zeros((H, W), np.uint8))[1].dtype)

    def set_pose_sums(self, sums):
        """Replace the PoseOptimization handle by one with the given sum order (None = the library's default)."""
        self.po.close()
        self.po = self.api.PoseOptimizer(max_obs=self.ext.cap, max_batch=1, device=self.device, sums=sums)

    def orb(self, gray):
        _, k, d = self.ext(gray)
        return k, d

    def frame_rgbd(self, kps, depth, ds, K):
        # the RGB-D tail of the Frame constructor as ONE call: the depth map crosses PCIe once, the cloud stays on the device
        # (a host copy only until the registration has its first target)
        ur, zd, cloud, dev = self.frm.FrameRGBD(kps, depth, BF, ds, *K, host_cloud=not self.have_target)
        return ur, zd, (cloud if cloud is not None else dev)

    def front_overlapped(self, gray, depth, ds, K):
        """ORB extraction BESIDE depth -> cloud -> registration: the two do not depend on each other (the registration needs the
        depth map only), so the extractor runs on a worker thread with its own handle and stream while this thread uploads the
        depth map, builds the cloud and registers it; the stereo coordinates follow when the key-points are there, from the depth
        map still on the device.  Same calls, same results as the sequential chain -- only the order on the time line differs."""
        fut = self.pool.submit(self.orb, gray)
        _, _, _, dev = self.frm.FrameRGBD(self.no_kps, depth, BF, ds, *K, host_cloud=False)
        T = self.reg.align_next_batch_device(dev[0], dev[1], 1, dev[2])[0]["T"]
        kps, desc = fut.result()
        ur, zd, _, _ = self.frm.FrameRGBD(kps, None, BF, 0, *K, host_cloud=False, shape=depth.shape)
        return kps, desc, ur, zd, dev, T

    def gicp(self, prev_cloud, cloud):
        if not self.have_target:  # first pair of the stream: both clouds; afterwards only the new one is preprocessed
            self.have_target = True
            return self.reg.RegisterPointClouds(prev_cloud, cloud)["T"]
        d_cloud, d_n, stride, _ = cloud  # the device-resident cloud of frame_rgbd
        return self.reg.align_next_batch_device(d_cloud, d_n, 1, stride)[0]["T"]

    def sbp(self, prob):
        return self.pm.SearchByProjection(prob)

    def pose(self, prob):
        return self.po.PoseOptimization(prob)


class OracleBackend:
    """The CPU restatement with the reference's threading (ORB: OpenMP over the levels; GICP: 4 threads)."""

    def __init__(self, O, W, H, nfeatures, nlevels):
        self.O = O
        self.orbx = O.OrbOracle(nfeatures, 1.2, nlevels, 20, 7)
        self.orbx.set_threads(8)
        O.gicp_set_threads(4)
        t = self.orbx.tables()
        self.scale, self.inv_sigma2 = np.asarray(t["scale"], np.float32), np.asarray(t["inv_sigma2"], np.float32)

    def close(self):
        self.orbx.set_threads(1)
        self.O.gicp_set_threads(1)

    def orb(self, gray):
        _, k, d = self.orbx.extract(gray)
        return k, d

    def frame_rgbd(self, kps, depth, ds, K):
        ur, zd = self.O.stereo_from_rgbd(kps, depth, BF)
        return ur, zd, self.O.depth_to_cloud(depth, ds, *K)

    def gicp(self, prev_cloud, cloud):
        return self.O.gicp_align(prev_cloud, cloud)["T"]

    def sbp(self, prob):
        return self.O.search_by_projection(prob)

    def pose(self, prob):
        return self.O.pose_optimization(prob)


def track_frame(be, last, gray, depth, K, W, H, ds, stages, overlap=False):
    """One frame through the chain.  `last` = the previous frame's state (None for the first frame).  Returns the new state.
    overlap: the extractor runs beside cloud + registration (GpuBackend.front_overlapped) once the registration has a target."""
    fx, fy, cx, cy = K
    t = time.perf_counter()
    if overlap and last is not None and getattr(be, "have_target", False):
        kps, desc, ur, zd, cloud, T = be.front_overlapped(gray, depth, ds, K)
        T = np.asarray(T, np.float64).reshape(4, 4)
        t = _lap(stages, "orb_beside_cloud_and_gicp", t)
        cur = dict(kps=kps, desc=desc, ur=np.asarray(ur, np.float32), z=np.asarray(zd, np.float32), cloud=cloud, matches=0, inliers=0, T=np.eye(4))
    else:
        kps, desc = be.orb(gray)
        t = _lap(stages, "orb", t)
        ur, zd, cloud = be.frame_rgbd(kps, depth, ds, K)
        t = _lap(stages, "frame_rgbd", t)
        cur = dict(kps=kps, desc=desc, ur=np.asarray(ur, np.float32), z=np.asarray(zd, np.float32), cloud=cloud, matches=0, inliers=0, T=np.eye(4))
        if last is None:
            return cur
        T = np.asarray(be.gicp(last["cloud"], cloud), np.float64).reshape(4, 4)  # x_last = T x_cur
        t = _lap(stages, "gicp", t)
    Tcl = np.linalg.inv(T)  # the current camera's pose with the last camera as the world
    qcl = _quat_from_R(Tcl[:3, :3])
    # the last frame's map points: its key points with a depth, unprojected (Frame::UnprojectStereo), world = last camera
    lk, lz = last["kps"], last["z"]
    has = lz > 0
    xw = np.empty((len(lz), 3), np.float32)  # (double-precision expressions rounded once, like np.stack(...).astype(float32))
    xw[:, 0] = (lk["x"] - cx) * lz / fx
    xw[:, 1] = (lk["y"] - cy) * lz / fy
    xw[:, 2] = lz
    if has.all():  # every key point of the last frame has a depth: nothing to compact
        l_desc, l_oct, l_ang = last["desc"], lk["octave"].astype(np.int32), lk["angle"].astype(np.float32)
    else:
        xw, l_desc, l_oct, l_ang = xw[has], last["desc"][has], lk["octave"][has].astype(np.int32), lk["angle"][has].astype(np.float32)
    f32 = np.float32
    prob = dict(last_xw=xw, last_desc=l_desc, last_octave=l_oct, last_angle=l_ang,
                last_mp_has_obs=np.ones(len(xw), np.uint8), cur_kps_un=kps, cur_u_right=cur["ur"], cur_desc=desc,
                cur_has_mp_obs=np.zeros(len(kps), np.uint8), Tcw_q=qcl.astype(np.float32), Tcw_t=Tcl[:3, 3].astype(np.float32),
                Tlw_q=np.array([0, 0, 0, 1], np.float32), Tlw_t=np.zeros(3, np.float32), fx=f32(fx), fy=f32(fy), cx=f32(cx), cy=f32(cy), bf=f32(BF),
                b=f32(BF / fx), min_x=f32(0), max_x=f32(W), min_y=f32(0), max_y=f32(H), grid_w_inv=f32(64) / f32(W), grid_h_inv=f32(48) / f32(H),
                scale_factors=be.scale, th=f32(TH_RGBD), mono=0, check_orientation=1)
    match, nm = be.sbp(prob)
    t = _lap(stages, "search_by_projection", t)
    sel = np.nonzero(match >= 0)[0]
    mp = match[sel]
    obs = np.stack([kps["x"][sel], kps["y"][sel], cur["ur"][sel]], 1).astype(np.float64)
    pp = dict(q=qcl.astype(np.float64), t=Tcl[:3, 3].astype(np.float64), xw=xw[mp].astype(np.float64), obs=obs,
              inv_sigma2=be.inv_sigma2[kps["octave"][sel]], stereo=(cur["ur"][sel] >= 0).astype(np.uint8), fx=fx, fy=fy, cx=cx, cy=cy, bf=BF)
    r = be.pose(pp)
    _lap(stages, "pose_optimization", t)
    cur.update(matches=int(nm), inliers=int(r["n_inliers"]), T=T, pose_q=np.asarray(r["q"]), pose_t=np.asarray(r["t"]), match=match)
    return cur


def _lap(stages, name, t0):
    t1 = time.perf_counter()
    stages.setdefault(name, []).append(t1 - t0)
    return t1


def run_stream(be, frames, K, W, H, ds, n_frames, warm=3, overlap=False):
    """frames: list of (gray, depth); the stream walks it back and forth.  Returns (per-frame seconds, per-stage seconds, states)."""
    seq = list(range(len(frames))) + list(range(len(frames) - 2, 0, -1)) if len(frames) > 2 else [0, 1]
    last, lat, stages, states = None, [], {}, []
    for i in range(n_frames + warm + 1):
        gray, depth = frames[seq[i % len(seq)]]
        st = {}
        t0 = time.perf_counter()
        last = track_frame(be, last, gray, depth, K, W, H, ds, st, overlap)
        dt = time.perf_counter() - t0
        if i > warm:  # frame 0 has no predecessor; the next `warm` frames load code objects and ramp the clocks
            lat.append(dt)
            for k, v in st.items():
                stages.setdefault(k, []).extend(v)
            states.append(last)
    return np.asarray(lat), stages, states


def summarize(lat, stages):
    return dict(frames=int(len(lat)), median_ms=round(float(np.median(lat)) * 1e3, 3), p99_ms=round(float(np.quantile(lat, 0.99)) * 1e3, 3),
                mean_ms=round(float(lat.mean()) * 1e3, 3), frames_per_s=round(float(1.0 / np.median(lat)), 1),
                stages_median_ms={k: round(float(np.median(v)) * 1e3, 3) for k, v in stages.items()})
