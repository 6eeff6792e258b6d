"""Frame helpers: depth -> cloud and RGB-D stereo coordinates.  CPU: oracle vs the numpy
restatement used by the harness.  GPU: HIP path bit-exact against the oracle (float32, op-by-op rounding)."""
import numpy as np
import pytest

from geoflowslam_amd import synth


def _depth(seed, w=640, h=480):
    return synth.Scene(seed).render(w, h, None, 0)[1]


def test_oracle_depth_to_cloud_matches_numpy(oracle):
    d = _depth(1)
    fx, fy, cx, cy = (np.float32(v) for v in synth.intrinsics(640, 480))
    for ds in (1, 3, 4):
        c = oracle.depth_to_cloud(d, ds, fx, fy, cx, cy)
        ref = synth.depth_to_cloud(d, ds)
        assert c.shape == ref.shape and (c.view(np.uint32) == ref.view(np.uint32)).all()
    assert len(oracle.depth_to_cloud(np.zeros((0, 0), np.float32), 4, fx, fy, cx, cy)) == 0
    dd = d.copy(); dd[::2] = 0; dd[1, 1] = 10.0; dd[1, 2] = 11.0  # zeros and >= 10 m are dropped
    assert len(oracle.depth_to_cloud(dd, 1, fx, fy, cx, cy)) == int(((dd > 0) & (dd < 10)).sum())


@pytest.mark.gpu
def test_gpu_depth_to_cloud_and_stereo(gpu_api, oracle):
    fr = gpu_api.Frame(max_rows=720, max_cols=1280)
    for (w, h, ds, seed) in ((640, 480, 4, 1), (640, 480, 3, 2), (1280, 720, 5, 3), (160, 120, 1, 4)):
        d = _depth(seed, w, h)
        fx, fy, cx, cy = (float(np.float32(v)) for v in synth.intrinsics(w, h))
        c = fr.ConvertDepthToPointCloud(d, ds, fx, fy, cx, cy)
        co = oracle.depth_to_cloud(d, ds, fx, fy, cx, cy)
        assert c.shape == co.shape and (c.view(np.uint32) == co.view(np.uint32)).all()
    assert len(fr.ConvertDepthToPointCloud(np.zeros((0, 0), np.float32), 4, 607, 607, 319.5, 239.5)) == 0
    view = _depth(5, 700, 500)[7:7 + 411, 13:13 + 577]  # non-continuous input (honour the row stride)
    c = fr.ConvertDepthToPointCloud(view, 3, 607.0, 607.0, 288.0, 205.0)
    co = oracle.depth_to_cloud(np.ascontiguousarray(view), 3, 607.0, 607.0, 288.0, 205.0)
    assert (c.view(np.uint32) == co.view(np.uint32)).all()
    # stereo coordinates on real ORB keypoints
    fp = synth.frame_pair(6)
    ext = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7)
    _, kps, _ = ext(fp["gray0"])
    bf = float(np.float32(0.0745 * 607.0))
    ur, vd = fr.ComputeStereoFromRGBD(kps, fp["depth0"], bf)
    uo, vo = oracle.stereo_from_rgbd(kps, fp["depth0"], bf)
    assert (ur.view(np.uint32) == uo.view(np.uint32)).all() and (vd.view(np.uint32) == vo.view(np.uint32)).all()
    assert (vd == -1).any() or (fp["depth0"] > 0).all()


@pytest.mark.gpu
def test_gpu_u16_depth_conversion_is_convert_to(gpu_api, oracle):
    """gfs_depth_convert_u16_batch_device = imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor) (src/Tracking.cc:1622-1623):
    float(raw) * float(factor), one rounding (cvtScale16u32f computes src * a + b in float with b = 0); then the cloud of the
    converted map equals the oracle's cloud of the host-converted map, bit for bit."""
    from test_gpu_gms import _Hip
    hip = _Hip()
    try:
        fr = gpu_api.Frame(max_rows=480, max_cols=640)
        rng = np.random.default_rng(3)
        for B, h, w, factor in ((3, 480, 640, 1.0 / 5000.0), (1, 37, 52, 0.001), (2, 120, 160, 1.0)):
            raw = rng.integers(0, 65536, (B, h, w), dtype=np.uint16)
            raw[:, ::7] = 0
            d_raw = hip.to_device(raw)
            d_f = hip.to_device(np.zeros((B, h, w), np.float32))
            fr.depth_convert_u16_batch_device(d_raw, B, h, w, factor, d_f)
            got = hip.to_host(d_f, (B, h, w), np.float32)
            want = raw.astype(np.float32) * np.float32(factor)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        # ... and on into the cloud
        fx, fy, cx, cy = (float(np.float32(v)) for v in synth.intrinsics(640, 480))
        raw = np.clip(np.rint(_depth(9, 640, 480) * 5000.0), 0, 65535).astype(np.uint16)[None]
        d_raw, d_f = hip.to_device(raw), hip.to_device(np.zeros((1, 480, 640), np.float32))
        fr.depth_convert_u16_batch_device(d_raw, 1, 480, 640, 1.0 / 5000.0, d_f)
        cap = 32768
        d_out, d_cnt = hip.to_device(np.zeros((cap, 4), np.float32)), hip.to_device(np.zeros(1, np.int32))
        fr.depth_to_cloud_batch_device(d_f, 1, 480, 640, 4, fx, fy, cx, cy, d_out, cap, d_cnt)
        n = int(hip.to_host(d_cnt, 1, np.int32)[0])
        co = oracle.depth_to_cloud(raw[0].astype(np.float32) * np.float32(1.0 / 5000.0), 4, fx, fy, cx, cy)
        assert n == len(co) and np.array_equal(hip.to_host(d_out, (cap, 4), np.float32)[:n].view(np.uint32), co.view(np.uint32))
    finally:
        hip.free()


@pytest.mark.gpu
def test_gpu_frame_rgbd_is_the_two_calls_and_feeds_the_registration(gpu_api, oracle):
    """gfs_frame_rgbd = ComputeStereoFromRGBD + ConvertDepthToPointCloud behind ONE depth upload (the RGB-D tail of the Frame
    constructor, src/Frame.cc:1314-1332, 590-623): mvuRight / mvDepth / cloud bit-exact against the oracle, with and without the host
    copy of the cloud; the device-resident cloud it returns is what gfs_gicp_align_next_batch_device consumes (same pose, bit for bit,
    as the host-pointer entry on the same clouds)."""
    fp = synth.frame_pair(6)
    fx, fy, cx, cy = (float(np.float32(v)) for v in synth.intrinsics(640, 480))
    bf = float(np.float32(0.0745 * 607.0))
    ext = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7)
    fr = gpu_api.Frame(max_rows=480, max_cols=640)
    reg_a, reg_b = gpu_api.RegistrationGICP(max_points=20480, max_batch=1), gpu_api.RegistrationGICP(max_points=20480, max_batch=1)
    clouds = []
    for key_g, key_d in (("gray0", "depth0"), ("gray1", "depth1")):
        _, kps, _ = ext(fp[key_g])
        ur, vd, cloud, dev = fr.FrameRGBD(kps, fp[key_d], bf, 4, fx, fy, cx, cy, host_cloud=True)
        uo, vo = oracle.stereo_from_rgbd(kps, fp[key_d], bf)
        co = oracle.depth_to_cloud(fp[key_d], 4, fx, fy, cx, cy)
        assert np.array_equal(ur.view(np.uint32), uo.view(np.uint32)) and np.array_equal(vd.view(np.uint32), vo.view(np.uint32))
        assert cloud.shape == co.shape and np.array_equal(cloud.view(np.uint32), co.view(np.uint32)) and dev[3] == len(co)
        ur2, vd2, none, dev2 = fr.FrameRGBD(kps, fp[key_d], bf, 4, fx, fy, cx, cy, host_cloud=False)
        assert none is None and dev2[3] == len(co) and np.array_equal(ur2.view(np.uint32), uo.view(np.uint32))
        # the cloud first (no key-points yet), the stereo coordinates later from the depth map still on the device: what a caller does
        # that runs the ORB extraction beside the registration (bench_stream.GpuBackend.front_overlapped)
        e0, e1, none, dev3 = fr.FrameRGBD(kps[:0], fp[key_d], bf, 4, fx, fy, cx, cy, host_cloud=False)
        assert len(e0) == 0 and dev3[3] == len(co)
        ur3, vd3, none, dev4 = fr.FrameRGBD(kps, None, bf, 0, fx, fy, cx, cy, host_cloud=False, shape=fp[key_d].shape)
        assert np.array_equal(ur3.view(np.uint32), uo.view(np.uint32)) and np.array_equal(vd3.view(np.uint32), vo.view(np.uint32)) and dev4[3] == 0
        dev2 = dev3
        clouds.append((co, dev2))
    # frame 0 against itself seeds both handles; then frame 1 once through the device-resident cloud, once through host pointers
    reg_a.RegisterPointClouds(clouds[0][0], clouds[0][0])
    reg_b.RegisterPointClouds(clouds[0][0], clouds[0][0])
    d_cloud, d_n, stride, _ = clouds[1][1]  # (the stereo-only call after it left the cloud and its count where they were)
    ra = reg_a.align_next_batch_device(d_cloud, d_n, 1, stride)[0]
    rb = reg_b.RegisterNext(clouds[1][0])
    assert np.array_equal(ra["T"], rb["T"]) and ra["iterations"] == rb["iterations"] and ra["num_inliers"] == rb["num_inliers"]
    ro = oracle.gicp_align(clouds[0][0], clouds[1][0])
    assert np.linalg.norm(ra["T"] - ro["T"]) <= 1e-6 * np.linalg.norm(ro["T"])
