"""GPU parity tests for ORBmatcher::SearchByProjection (frame to frame; MI355X, k_sbp through gfs_search_by_projection):
bit-exact key-point assignments and match counts against the CPU oracle, including the order-dependent cases (competing map
points, overwritable zero-observation points, pre-assigned key-points), mono / stereo level windows, forward / backward
motion, ragged batches and empty inputs."""
import numpy as np
import pytest

from geoflowslam_amd import synth

pytestmark = pytest.mark.gpu

CASES = [dict(seed=1), dict(seed=2, dup_frac=0.3, zero_obs_frac=0.3, preassigned_frac=0.1, th=15.0, mono=True),
         dict(seed=3, motion=0.3), dict(seed=4, check_orientation=False, dup_frac=0.2),
         dict(seed=5, n_points=1900, n_extra_cur=400), dict(seed=6, n_points=400, dup_frac=0.5, zero_obs_frac=1.0),
         dict(seed=7, motion=-0.3, rot_deg=2.0), dict(seed=8, th=3.0, desc_flip_bits=60)]


@pytest.mark.parametrize("cfg", CASES)
def test_single_pair_matches_oracle(gpu_api, oracle, cfg):
    p = synth.sbp_pair(**cfg)
    pm = gpu_api.ProjectionMatcher(max_last=2048, max_cur=2560, max_batch=2)
    m, n = pm.SearchByProjection(p)
    mo, no = oracle.search_by_projection(p)
    assert n == no
    assert np.array_equal(m, mo)


def test_ragged_batch_and_empty_inputs(gpu_api, oracle):
    pairs = [synth.sbp_pair(20 + i, n_points=n, n_extra_cur=e) for i, (n, e) in enumerate([(300, 50), (40, 5), (900, 300), (1, 0)])]
    p = pairs[0]
    pairs.append(dict(p, last_xw=np.zeros((0, 3), np.float32), last_desc=np.zeros((0, 32), np.uint8),
                      last_octave=np.zeros(0, np.int32), last_angle=np.zeros(0, np.float32), last_mp_has_obs=np.zeros(0, np.uint8)))
    pairs.append(dict(p, cur_kps_un=p["cur_kps_un"][:0], cur_u_right=np.zeros(0, np.float32), cur_desc=np.zeros((0, 32), np.uint8),
                      cur_has_mp_obs=np.zeros(0, np.uint8)))
    pm = gpu_api.ProjectionMatcher(max_last=1024, max_cur=1536, max_batch=8)
    res = pm.SearchByProjection(pairs)
    for q, (m, n) in zip(pairs, res):
        mo, no = oracle.search_by_projection(q)
        assert n == no and np.array_equal(m, mo)
    assert res[4][1] == 0 and (res[4][0] == -1).all() and len(res[5][0]) == 0


def test_more_candidates_than_the_list_holds(gpu_api, oracle):
    """A dense cluster of key-points inside one search window (> 64 candidates) takes the re-enumeration path."""
    p = synth.sbp_pair(30, n_points=60, n_extra_cur=0, th=15.0, mono=True)
    rng = np.random.default_rng(0)
    k = p["cur_kps_un"].copy()
    n = len(k)
    extra = np.zeros(200, k.dtype)
    extra["x"] = k["x"][0] + rng.uniform(-10, 10, 200).astype(np.float32)
    extra["y"] = k["y"][0] + rng.uniform(-10, 10, 200).astype(np.float32)
    extra["octave"] = k["octave"][0]
    extra["angle"] = rng.uniform(0, 360, 200).astype(np.float32)
    q = dict(p, cur_kps_un=np.concatenate([k, extra]), cur_u_right=np.concatenate([p["cur_u_right"], np.full(200, -1, np.float32)]),
             cur_desc=np.concatenate([p["cur_desc"], rng.integers(0, 256, (200, 32), dtype=np.uint8)]),
             cur_has_mp_obs=np.concatenate([p["cur_has_mp_obs"], np.zeros(200, np.uint8)]))
    pm = gpu_api.ProjectionMatcher(max_last=128, max_cur=512, max_batch=1)
    m, nm = pm.SearchByProjection(q)
    mo, no = oracle.search_by_projection(q)
    assert nm == no and np.array_equal(m, mo)


def test_capacity_and_argument_errors(gpu_api):
    pm = gpu_api.ProjectionMatcher(max_last=64, max_cur=64, max_batch=1)
    with pytest.raises(gpu_api.GfsError):
        pm.SearchByProjection(synth.sbp_pair(1, n_points=100, n_extra_cur=10))
    with pytest.raises(gpu_api.GfsError):
        gpu_api.ProjectionMatcher(max_last=64, max_cur=100000, max_batch=1)
    bad = synth.sbp_pair(2, n_points=30, n_extra_cur=5)
    bad["last_octave"] = bad["last_octave"].copy()
    bad["last_octave"][0] = 99
    with pytest.raises(gpu_api.GfsError):
        pm.SearchByProjection(bad)


MAP_CASES = [dict(seed=1), dict(seed=2, th=3.0, dup_frac=0.3, zero_obs_frac=0.3, preassigned_frac=0.1),
             dict(seed=3, th=5.0, nn_ratio=0.6, desc_flip_bits=80), dict(seed=4, n_points=1900, n_extra_cur=400, th=3.0),
             dict(seed=5, th=1.0, nn_ratio=1.0, zero_obs_frac=1.0, dup_frac=0.5)]


@pytest.mark.parametrize("cfg", MAP_CASES)
def test_map_variant_matches_oracle(gpu_api, oracle, cfg):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th, ...) (src/ORBmatcher.cc:43-206): bit-exact assignments and count."""
    p = synth.sbp_map_frame(**cfg)
    pm = gpu_api.ProjectionMatcher(max_last=2048, max_cur=2560, max_batch=2)
    m, n = pm.SearchByProjectionMap(p)
    mo, no = oracle.search_by_projection_map(p)
    assert n == no and np.array_equal(m, mo)


def test_map_variant_batch_and_dense_window(gpu_api, oracle):
    frames = [synth.sbp_map_frame(40 + i, n_points=n, n_extra_cur=e, th=t) for i, (n, e, t) in enumerate([(300, 50, 1.0), (5, 0, 3.0), (700, 200, 5.0)])]
    # a dense cluster (> 64 candidates in one window) exercises the re-enumeration path with its second-best bookkeeping
    p = synth.sbp_map_frame(50, n_points=60, n_extra_cur=0, th=5.0)
    rng = np.random.default_rng(1)
    k = p["cur_kps_un"]
    extra = np.zeros(200, k.dtype)
    extra["x"] = p["mp_proj"][0, 0] + rng.uniform(-8, 8, 200).astype(np.float32)
    extra["y"] = p["mp_proj"][0, 1] + rng.uniform(-8, 8, 200).astype(np.float32)
    extra["octave"] = p["mp_level"][0]
    frames.append(dict(p, cur_kps_un=np.concatenate([k, extra]), cur_u_right=np.concatenate([p["cur_u_right"], np.full(200, -1, np.float32)]),
                       cur_desc=np.concatenate([p["cur_desc"], rng.integers(0, 256, (200, 32), dtype=np.uint8)]),
                       cur_has_mp_obs=np.concatenate([p["cur_has_mp_obs"], np.zeros(200, np.uint8)])))
    pm = gpu_api.ProjectionMatcher(max_last=1024, max_cur=1536, max_batch=4)
    for q, (m, n) in zip(frames, pm.SearchByProjectionMap(frames)):
        mo, no = oracle.search_by_projection_map(q)
        assert n == no and np.array_equal(m, mo)


def _dense_contested_map_frame(seed):
    """Many map points projecting into one small region, key-points of mixed octaves with descriptors drawn from a small pool: the
    best and the second-best candidate of a map point often sit on different levels and are contested by lower-index map points,
    so proposals get WITHDRAWN when a blocker takes a second best away (the ratio test flips) -- ADVICE round 4, sbp.hip mode 1."""
    rng = np.random.default_rng(seed)
    p = synth.sbp_map_frame(60 + seed % 3, n_points=40, n_extra_cur=0, th=5.0, nn_ratio=0.8)
    nmp, ncur = int(rng.integers(120, 400)), int(rng.integers(80, 300))
    cx, cy = rng.uniform(120, 520), rng.uniform(120, 360)
    half = rng.uniform(14, 40)
    pool = rng.integers(0, 256, (6, 32), dtype=np.uint8)

    def descs(n, flips):
        d = pool[rng.integers(0, len(pool), n)].copy()
        bits = np.unpackbits(d, axis=1)
        for r in range(n):
            idx = rng.choice(256, size=int(rng.integers(0, flips + 1)), replace=False)
            bits[r, idx] ^= 1
        return np.packbits(bits, axis=1)

    uv = np.stack([cx + rng.uniform(-half, half, nmp), cy + rng.uniform(-half, half, nmp)], 1)
    mp_proj = np.concatenate([uv, (uv[:, :1] - rng.uniform(5, 40, (nmp, 1)))], 1).astype(np.float32)
    k = np.zeros(ncur, p["cur_kps_un"].dtype)
    k["x"] = (cx + rng.uniform(-half, half, ncur)).astype(np.float32)
    k["y"] = (cy + rng.uniform(-half, half, ncur)).astype(np.float32)
    k["octave"] = rng.integers(1, 4, ncur)
    k["size"], k["angle"] = 31.0, rng.uniform(0, 360, ncur).astype(np.float32)
    return dict(p, mp_proj=mp_proj, mp_level=rng.integers(2, 4, nmp).astype(p["mp_level"].dtype),
                mp_view_cos=np.where(rng.random(nmp) < 0.5, 0.9995, 0.9).astype(np.float32), mp_desc=descs(nmp, 30),
                mp_has_obs=(rng.random(nmp) < 0.85).astype(p["mp_has_obs"].dtype), cur_kps_un=k,
                cur_u_right=np.full(ncur, -1, np.float32), cur_desc=descs(ncur, 30),
                cur_has_mp_obs=(rng.random(ncur) < 0.05).astype(p["cur_has_mp_obs"].dtype))


def test_map_variant_withdrawn_proposals_dense(gpu_api, oracle):
    pm = gpu_api.ProjectionMatcher(max_last=512, max_cur=512, max_batch=8)
    bad, total_matches = [], 0
    for s0 in range(0, 160, 8):
        frames = [_dense_contested_map_frame(s) for s in range(s0, s0 + 8)]
        for rep in range(2):  # twice: the old scheme was also non-deterministic
            for s, q, (m, n) in zip(range(s0, s0 + 8), frames, pm.SearchByProjectionMap(frames)):
                mo, no = oracle.search_by_projection_map(q)
                total_matches += no
                if n != no or not np.array_equal(m, mo):
                    bad.append((s, rep, n, no, int((m != mo).sum())))
    assert total_matches > 160 * 2 * 20, total_matches  # the cases do produce contested assignments
    assert not bad, bad
