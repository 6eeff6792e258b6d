"""The C++ host mirror of the reference's classes (namespace gfs_host, geoflowslam_amd/host/gfs_adaptors.hpp) driven from C++
(tests/host/host_mirror_test.cpp, built here with g++ against libgfs_hip.so): ORBextractor::operator() with its tables and
lapping area, ORBmatcher::match / DescriptorDistance, RegistrationGICP::RegisterPointClouds / RegisterNext, GmsMatcher,
ProjectionMatcher, PoseOptimizer — against
the CPU oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from geoflowslam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "tests", "host", "_host_mirror_test.so")


@pytest.fixture(scope="module")
def harness(gpu_api):
    src = os.path.join(ROOT, "tests", "host", "host_mirror_test.cpp")
    hdr = os.path.join(ROOT, "geoflowslam_amd", "host", "gfs_adaptors.hpp")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        libdir = os.path.join(ROOT, "geoflowslam_amd")
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-o", _SO, src, "-L" + libdir, "-lgfs_hip",
                        "-Wl,-rpath," + libdir], check=True)
    return C.CDLL(_SO)


def _p(a):
    return C.c_void_p(a.ctypes.data)


def test_orb_extractor_operator_and_tables(harness, gpu_api, oracle):
    for seed, (w, h), nf, lap in ((1, (640, 480), 1000, (0, 0)), (2, (752, 480), 1500, (0, 300)), (3, (320, 240), 500, (0, 0))):
        img = synth.frame_pair(seed, w, h, 8)["gray0"]
        cap = 4 * nf
        kps = np.zeros(cap, oracle.KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        tabs = np.zeros((4, 8), np.float32)
        mono = harness.hm_orb(_p(img), h, w, w, nf, C.c_float(1.2), 8, 20, 7, lap[0], lap[1], _p(kps), _p(desc), cap, C.byref(n), _p(tabs))
        orc = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
        mo, ko, do = orc.extract(img, lap)
        assert mono == mo and n.value == len(ko)
        assert (kps[:n.value] == ko).all() and (desc[:n.value] == do).all()
        to = orc.tables()
        for k, name in enumerate(("scale", "inv_scale", "sigma2", "inv_sigma2")):
            assert np.array_equal(tabs[k].view(np.uint32), np.asarray(to[name], np.float32).view(np.uint32)), name
    # empty image: operator() returns -1 and no key-point
    n = C.c_int(7)
    assert harness.hm_orb(None, 0, 0, 0, 1000, C.c_float(1.2), 8, 20, 7, 0, 0, None, None, 0, C.byref(n), None) == -1 and n.value == 0


def test_matcher_and_gms(harness, gpu_api, oracle):
    fp = synth.frame_pair(5, 640, 480, 8)
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    _, k0, d0 = orc.extract(fp["gray0"])
    _, k1, d1 = orc.extract(fp["gray1"])
    nq = len(d0)
    qi, ti, dist, dd = np.zeros(nq, np.int32), np.zeros(nq, np.int32), np.zeros(nq, np.float32), C.c_int(-1)
    n = harness.hm_match(_p(d0), nq, _p(d1), len(d1), _p(qi), _p(ti), _p(dist), C.byref(dd))
    to, do = oracle.bf_match(d0, d1)
    assert n == len(to) and np.array_equal(qi[:n], np.arange(n)) and np.array_equal(ti[:n], to) and np.array_equal(dist[:n], do.astype(np.float32))
    assert dd.value == int(np.unpackbits(d0[0] ^ d1[0]).sum())
    mask = np.zeros(max(n, 1), np.uint8)
    k0c, k1c = np.ascontiguousarray(k0), np.ascontiguousarray(k1)
    nin = harness.hm_gms(_p(k0c), len(k0c), 640, 480, _p(k1c), len(k1c), 640, 480, _p(qi), _p(ti), n, _p(mask))
    mo, no = oracle.gms_inlier_mask(k0c, (640, 480), k1c, (640, 480), qi[:n], ti[:n])
    assert nin == no and np.array_equal(mask[:n].astype(bool), mo)


def test_registration_gicp_and_streaming_form(harness, gpu_api, oracle):
    a, b, _ = synth.cloud_pair(21, 160, 120, trans=0.03, rot_deg=1.5)
    _, c, _ = synth.cloud_pair(22, 160, 120, trans=0.03, rot_deg=1.5)
    init = np.eye(4).T.reshape(-1).copy()  # column-major identity
    R = gpu_api.GicpResult if hasattr(gpu_api, "GicpResult") else None
    assert R is not None
    r_ab, r_bc = R(), R()
    rc = harness.hm_gicp(_p(a), len(a), _p(b), len(b), _p(c), len(c), _p(init), C.byref(r_ab), C.byref(r_bc))
    assert rc == 0
    for r, (t, s) in ((r_ab, (a, b)), (r_bc, (b, c))):
        ro = oracle.gicp_align(t, s)
        T = np.array(r.T[:]).reshape(4, 4).T
        assert bool(r.converged) == ro["converged"] and int(r.iterations) == ro["iterations"] and int(r.num_inliers) == ro["num_inliers"]
        assert np.linalg.norm(T - ro["T"]) / np.linalg.norm(ro["T"]) < 1e-6


def test_projection_matcher_and_pose_optimizer(harness, gpu_api, oracle):
    from geoflowslam_amd import api as A
    q = synth.sbp_pair(41, n_points=700, n_extra_cur=200, dup_frac=0.2, zero_obs_frac=0.3)
    P, keep = A.sbp_struct(q)
    out = np.full(max(P.n_cur, 1), -9, np.int32)
    n = harness.hm_sbp(C.byref(P), _p(out))
    mo, no = oracle.search_by_projection(q)
    assert n == no and np.array_equal(out[:P.n_cur], mo)
    f = synth.pose_frame(42, n_obs=400, outlier_frac=0.15)
    PP, SS, keep2, nobs = A.pose_structs(f)
    ninl = harness.hm_pose(C.byref(PP), C.byref(SS))
    r, ro = A.pose_result(SS, keep2, nobs), oracle.pose_optimization(f)
    assert ninl == ro["n_inliers"] and np.array_equal(r["outlier"], ro["outlier"]) and r["rounds_run"] == ro["rounds_run"]
    assert np.linalg.norm(r["q"] - ro["q"]) < 1e-5 * np.linalg.norm(ro["q"]) and np.linalg.norm(r["t"] - ro["t"]) < 1e-5 * max(np.linalg.norm(ro["t"]), 1e-30)
