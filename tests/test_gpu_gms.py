"""GPU parity tests for the GMS filter (MI355X, k_gms through gfs_gms_inlier_mask and the device-resident chain
ORB -> BF match -> GMS): inlier masks and counts bit-exact against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

from geoflowslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [dict(seed=3), dict(seed=4), dict(seed=5, w=1280, h=720, nf=2000)])
def test_host_entry_matches_oracle(gpu_api, oracle, cfg):
    w, h, nf = cfg.get("w", 640), cfg.get("h", 480), cfg.get("nf", 1000)
    fp = synth.frame_pair(cfg["seed"], w, h, 8)
    ext = gpu_api.ORBextractor(nf, 1.2, 8, 20, 7, max_rows=h, max_cols=w)
    _, k0, d0 = ext(fp["gray0"])
    _, k1, d1 = ext(fp["gray1"])
    ti, _ = gpu_api.ORBmatcher(max_query=4096, max_train=4096).match(d0, d1)
    q = np.arange(len(ti), dtype=np.int32)
    gm = gpu_api.GmsMatcher(max_keypoints=4096, max_batch=2)
    m, n = gm.GetInlierMask(k0, (w, h), k1, (w, h), q, ti)
    mo, no = oracle.gms_inlier_mask(k0, (w, h), k1, (w, h), q, ti)
    assert n == no and np.array_equal(m, mo)
    assert n > 30


def test_border_and_random_matches(gpu_api, oracle):
    rng = np.random.default_rng(0)
    kp = np.zeros(600, gpu_api.KP_DTYPE)
    kp["x"] = rng.uniform(0, 639.99, 600).astype(np.float32)
    kp["y"] = rng.uniform(0, 479.99, 600).astype(np.float32)
    kp["x"][:40] = rng.uniform(630, 639.9, 40).astype(np.float32)
    kp["y"][40:80] = rng.uniform(470, 479.9, 40).astype(np.float32)
    kp2 = kp.copy()
    kp2["x"] = np.clip(kp["x"] + 3.0, 0, 639.9).astype(np.float32)
    q = np.arange(600, dtype=np.int32)
    t = q.copy()
    t[300:] = rng.integers(0, 600, 300)
    gm = gpu_api.GmsMatcher(max_keypoints=1024, max_batch=1)
    m, n = gm.GetInlierMask(kp, (640, 480), kp2, (640, 480), q, t)
    mo, no = oracle.gms_inlier_mask(kp, (640, 480), kp2, (640, 480), q, t)
    assert n == no and np.array_equal(m, mo)
    m, n = gm.GetInlierMask(kp[:0], (640, 480), kp2, (640, 480), q[:0], t[:0])
    assert n == 0 and len(m) == 0
    with pytest.raises(gpu_api.GfsError):
        gm.GetInlierMask(kp, (640, 480), kp2, (640, 480), q, t + 1000)  # trainIdx out of range


class _Hip:
    """Device buffers through the HIP runtime the library itself uses (torch ships its own copy of libamdhip64; loading it
    after libgfs_hip.so in the same process would give torch no devices)."""

    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.ptrs = []

    def to_device(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), max(a.nbytes, 4)) == 0
        assert self.lib.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        self.ptrs.append(p)
        return p.value

    def to_host(self, ptr, shape, dtype):
        out = np.zeros(shape, dtype)
        assert self.lib.hipDeviceSynchronize() == 0
        assert self.lib.hipMemcpy(out.ctypes.data, C.c_void_p(ptr), out.nbytes, 2) == 0
        return out

    def stream(self):
        """A HIP stream: the *_batch_device entry points return without waiting, a chain of them must share one stream."""
        st = C.c_void_p()
        assert self.lib.hipStreamCreate(C.byref(st)) == 0
        return st.value

    def free(self):
        for p in self.ptrs:
            self.lib.hipFree(p)


def test_device_chain_matches_oracle(gpu_api, oracle):
    """ORB (device batch) -> BF match (device batch) -> GMS (device batch) without leaving HBM."""
    B, W, H = 3, 640, 480
    pairs = [synth.frame_pair(20 + b, W, H, 8) for b in range(B)]
    hip = _Hip()
    g0 = hip.to_device(np.stack([p["gray0"] for p in pairs]))
    g1 = hip.to_device(np.stack([p["gray1"] for p in pairs]))
    e0 = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7, max_rows=H, max_cols=W, max_batch=B)
    e1 = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7, max_rows=H, max_cols=W, max_batch=B)
    st = hip.stream()
    e0.extract_batch_device(g0, B, H, W, (0, 0), st)
    e1.extract_batch_device(g1, B, H, W, (0, 0), st)
    r0, r1 = e0.device_results(), e1.device_results()
    cap = e0.cap
    mt = gpu_api.ORBmatcher(max_query=cap, max_train=cap, max_batch=B)
    idx = hip.to_device(np.zeros(B * cap, np.int32))
    dist = hip.to_device(np.zeros(B * cap, np.int32))
    mt.match_batch_device(r0["desc"], r0["counts"], r1["desc"], r1["counts"], B, cap, idx, dist, st)
    mask = hip.to_device(np.zeros(B * cap, np.uint8))
    cnt = hip.to_device(np.zeros(B, np.int32))
    gm = gpu_api.GmsMatcher(max_keypoints=cap, max_batch=B)
    gm.inlier_mask_batch_device(r0["kps"], r0["counts"], r1["kps"], r1["counts"], B, cap, idx, W, H, mask, cnt, st)
    mask_h = hip.to_host(mask, (B, cap), np.uint8)
    cnt_h = hip.to_host(cnt, B, np.int32)
    idx_h = hip.to_host(idx, (B, cap), np.int32)
    try:
        for b in range(B):
            _, k0, d0 = e0.fetch(b)
            _, k1, d1 = e1.fetch(b)
            n0 = len(k0)
            ti = idx_h[b, :n0]
            mo, no = oracle.gms_inlier_mask(k0, (W, H), k1, (W, H), np.arange(n0, dtype=np.int32), ti)
            assert cnt_h[b] == no and np.array_equal(mask_h[b, :n0].astype(bool), mo)
            assert no > 30
    finally:
        hip.free()
