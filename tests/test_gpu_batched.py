"""GPU parity tests of the BATCHED device entry points at the batch sizes bench.py and BASELINE.json's configs use
(configs[2]: 32 x 720p; configs[3] / the bench lane: 64 and 128 x VGA): every element of the batch against the CPU oracle and,
bit for bit, against the same work done one at a time (B = 1) — the XCD-aware pair -> workgroup map, the per-pair
Levenberg-Marquardt state machine with ragged convergence and the k-NN deferral lists all depend on B."""
import concurrent.futures as cf

import numpy as np
import pytest

from geoflowslam_amd import synth
from test_gpu_gms import _Hip

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _same(a, b):
    return (np.array_equal(a["T"], b["T"]) and a["converged"] == b["converged"] and a["iterations"] == b["iterations"] and
            a["num_inliers"] == b["num_inliers"] and np.array_equal(a["H"], b["H"]) and np.array_equal(a["b"], b["b"]) and
            a["error"] == b["error"] and a["n_target_ds"] == b["n_target_ds"] and a["n_source_ds"] == b["n_source_ds"])


def _ragged_cloud_pairs(B, w, h, seed0):
    """B distinct (target, source, init) triples: ordinary small motions, large motions that need up to all 20 iterations or
    never converge, clouds cut to different lengths, an empty source, an empty target, clouds 50 m apart (no correspondence),
    a tiny cloud, a non-identity initial guess."""
    rng = np.random.default_rng(seed0)
    motions = [(0.03, 1.5)] * 5 + [(0.3, 10.0), (0.4, 12.0), (0.6, 8.0), (0.05, 3.0), (0.8, 20.0)]

    def make(b):
        tr, rd = motions[b % len(motions)]
        return synth.cloud_pair(seed0 + b, w, h, trans=tr, rot_deg=rd)

    with cf.ThreadPoolExecutor(max_workers=16) as ex:
        raw = list(ex.map(make, range(B)))
    out = []
    for b, (c0, c1, T01) in enumerate(raw):
        init = np.eye(4)
        if b % 7 == 3:
            c0 = c0[:int(len(c0) * rng.uniform(0.3, 0.9))]
        if b % 11 == 5:
            c1 = c1[::2]
        if b == 2:
            c1 = c1[:0]
        if b == 4:
            c0 = c0[:0]
        if b == 6:
            c1 = c1.copy()
            c1[:, 2] += 50.0
        if b == 9:
            c0, c1 = c0[:9], c1[:9]
        if b % 5 == 1:
            init = T01 @ synth.random_motion(rng, 0.01, 0.5)  # a prior near the true motion, as the tracker supplies
        out.append((c0, c1, init))
    return out


@pytest.mark.parametrize("B,w,h,name", [(128, 160, 120, "vga-lane"), (64, 160, 120, "c4-shard"), (32, 256, 144, "c3-720p")])
def test_gicp_batch_device_matches_oracle_and_single(gpu_api, oracle, B, w, h, name):
    triples = _ragged_cloud_pairs(B, w, h, {"vga-lane": 3000, "c4-shard": 5000, "c3-720p": 7000}[name])
    SP = (max(max(len(a), len(b)) for a, b, _ in triples) + 1023) // 1024 * 1024
    c0 = np.zeros((B, SP, 4), np.float32)
    c1 = np.zeros((B, SP, 4), np.float32)
    n0 = np.zeros(B, np.int32)
    n1 = np.zeros(B, np.int32)
    init = np.zeros((B, 4, 4))
    for b, (a, s, T0) in enumerate(triples):
        c0[b, :len(a)], c1[b, :len(s)], n0[b], n1[b], init[b] = a, s, len(a), len(s), T0
    hip = _Hip()
    try:
        reg = gpu_api.RegistrationGICP(max_points=SP, max_batch=B)
        got = reg.align_batch_device(hip.to_device(c0), hip.to_device(n0), hip.to_device(c1), hip.to_device(n1), B, SP, init_T=init)
        with cf.ThreadPoolExecutor(max_workers=32) as ex:
            want = list(ex.map(lambda t: oracle.gicp_align(t[0], t[1], t[2]), triples))
        single = gpu_api.RegistrationGICP(max_points=SP, max_batch=1)
        iters = []
        for b, ((a, s, T0), r, ro) in enumerate(zip(triples, got, want)):
            assert r["converged"] == ro["converged"] and r["iterations"] == ro["iterations"], (b, r, ro)
            assert r["num_inliers"] == ro["num_inliers"], (b, r["num_inliers"], ro["num_inliers"])
            assert r["n_target_ds"] == ro["n_target_ds"] and r["n_source_ds"] == ro["n_source_ds"], b
            assert _rel(r["T"], ro["T"]) < 1e-6, (b, _rel(r["T"], ro["T"]))
            if ro["num_inliers"]:
                assert _rel(r["H"], ro["H"]) < 1e-5 and abs(r["error"] - ro["error"]) <= 1e-5 * abs(ro["error"]), b
            assert _same(r, single.RegisterPointClouds(a, s, T0)), b  # the batch changes nothing, bit for bit
            iters.append(ro["iterations"])
        assert len(set(iters)) >= 4 and max(iters) >= 19  # the batch really is ragged
    finally:
        hip.free()


def test_orb_match_gms_batch64_device_matches_oracle(gpu_api, oracle):
    """configs[3]'s per-GPU shard: 64 distinct VGA frame pairs through gfs_orb_extract_batch_device ->
    gfs_bf_match_hamming_batch_device -> gfs_gms_inlier_mask_batch_device; key points, descriptors, match pairs and the GMS
    mask of every pair bit-exact against the oracle."""
    B, W, H = 64, 640, 480
    with cf.ThreadPoolExecutor(max_workers=32) as ex:
        pairs = list(ex.map(lambda s: synth.frame_pair(s, W, H, 16), range(9000, 9000 + B)))
    pairs[5]["gray1"] = np.full((H, W), 90, np.uint8)  # a frame without a single corner
    pairs[7]["gray0"] = np.full((H, W), 17, np.uint8)
    hip = _Hip()
    try:
        g0 = hip.to_device(np.stack([p["gray0"] for p in pairs]))
        g1 = hip.to_device(np.stack([p["gray1"] for p in pairs]))
        e0 = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7, max_rows=H, max_cols=W, max_batch=B)
        e1 = gpu_api.ORBextractor(1000, 1.2, 8, 20, 7, max_rows=H, max_cols=W, max_batch=B)
        st = hip.stream()  # the device entry points return without waiting: the chain shares one stream
        e0.extract_batch_device(g0, B, H, W, (0, 0), st)
        e1.extract_batch_device(g1, B, H, W, (0, 0), st)
        r0, r1 = e0.device_results(), e1.device_results()
        cap = e0.cap
        mt = gpu_api.ORBmatcher(max_query=cap, max_train=cap, max_batch=B)
        idx, dist = hip.to_device(np.zeros(B * cap, np.int32)), hip.to_device(np.zeros(B * cap, np.int32))
        mt.match_batch_device(r0["desc"], r0["counts"], r1["desc"], r1["counts"], B, cap, idx, dist, st)
        mask, cnt = hip.to_device(np.zeros(B * cap, np.uint8)), hip.to_device(np.zeros(B, np.int32))
        gm = gpu_api.GmsMatcher(max_keypoints=cap, max_batch=B)
        gm.inlier_mask_batch_device(r0["kps"], r0["counts"], r1["kps"], r1["counts"], B, cap, idx, W, H, mask, cnt, st)
        mask_h, cnt_h = hip.to_host(mask, (B, cap), np.uint8), hip.to_host(cnt, B, np.int32)
        idx_h, dist_h = hip.to_host(idx, (B, cap), np.int32), hip.to_host(dist, (B, cap), np.int32)

        def ref(b):
            orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
            m0, k0, d0 = orc.extract(pairs[b]["gray0"])
            m1, k1, d1 = orc.extract(pairs[b]["gray1"])
            ti, di = oracle.bf_match(d0, d1)
            mo, no = oracle.gms_inlier_mask(k0, (W, H), k1, (W, H), np.arange(len(ti), dtype=np.int32), ti) if len(ti) else (np.zeros(0, bool), 0)
            return k0, d0, k1, d1, ti, di, mo, no

        with cf.ThreadPoolExecutor(max_workers=32) as ex:
            refs = list(ex.map(ref, range(B)))
        for b, (k0, d0, k1, d1, ti, di, mo, no) in enumerate(refs):
            _, gk0, gd0 = e0.fetch(b)
            _, gk1, gd1 = e1.fetch(b)
            assert len(gk0) == len(k0) and len(gk1) == len(k1), b
            assert (gk0 == k0).all() and (gd0 == d0).all() and (gk1 == k1).all() and (gd1 == d1).all(), b
            nq = len(ti)
            assert np.array_equal(idx_h[b, :nq], ti) and np.array_equal(dist_h[b, :nq], di), b
            assert cnt_h[b] == no and np.array_equal(mask_h[b, :nq].astype(bool), mo), b
    finally:
        hip.free()
