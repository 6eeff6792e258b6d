"""Optimizer::LocalBundleAdjustment as real adaptor code (geoflowslam_amd/host/gfs_adaptors.hpp: gather of the local window,
float -> double flattening in the reference's vertex / edge creation order, stop flag, chi2 / depth classification, write-back):
driven through plain-struct stand-ins for KeyFrame / MapPoint / Map (tests/host/lba_adaptor_test.cpp).  The CPU tests solve with
the oracle; the GPU test with gfs_lba_solve."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from geoflowslam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "tests", "host", "_lba_adaptor_test.so")


@pytest.fixture(scope="module")
def harness(api):
    src = os.path.join(ROOT, "tests", "host", "lba_adaptor_test.cpp")
    hdr = os.path.join(ROOT, "geoflowslam_amd", "host", "gfs_adaptors.hpp")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        libdir = os.path.join(ROOT, "geoflowslam_amd")
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-o", _SO, src, "-L" + libdir, "-lgfs_hip", "-ldl", "-lpthread",
                        "-Wl,-rpath," + libdir], check=True)
    L = C.CDLL(_SO)
    L.lba_adaptor_test.argtypes = ([C.c_char_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_double] * 5 + [C.c_int] * 3
                                   + [C.c_void_p] * 10)
    return L


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, np.float32).astype(np.float64))


def _window32(seed, **kw):
    """A synthetic window whose inputs are exactly representable in float (what KeyFrame / MapPoint hold)."""
    w = synth.lba_window(seed, **kw)
    for k in ("pose_q", "pose_t", "points", "edge_obs", "edge_inv_sigma2"):
        w[k] = _f32(w[k])
    for k in ("fx", "fy", "cx", "cy", "bf"):
        w[k] = float(np.float32(w[k]))
    # a key-point has a stereo coordinate iff mvuRight >= 0 (that is what the reference tests, :1837 / :1866)
    w["edge_stereo"] = (np.asarray(w["edge_stereo"]) > 0) & (np.asarray(w["edge_obs"])[:, 2] >= 0)
    w["edge_stereo"] = w["edge_stereo"].astype(np.uint8)
    return w


def _run(L, w, solver_lib, init_kf_pose=-1, bad_point=-1, stop_flag=0):
    npz, npt, ne = int(w["n_poses"]), int(w["n_points"]), int(w["n_edges"])
    out = dict(pose_q=np.zeros((npz, 4), np.float32), pose_t=np.zeros((npz, 3), np.float32), points=np.zeros((npt, 3), np.float32),
               erased=np.zeros((max(ne, 1), 2), np.int32), counts=np.zeros(8, np.int32), hubers=np.zeros(2),
               inv_sigma2=np.zeros(max(ne, 1)), obs=np.zeros((max(ne, 1), 3)), edge_pose=np.zeros(max(ne, 1), np.int32),
               edge_point=np.zeros(max(ne, 1), np.int32))
    arrs = [np.ascontiguousarray(w["pose_q"], np.float64), np.ascontiguousarray(w["pose_t"], np.float64),
            np.ascontiguousarray(w["pose_fixed"], np.uint8), np.ascontiguousarray(w["points"], np.float64),
            np.ascontiguousarray(w["edge_pose"], np.int32), np.ascontiguousarray(w["edge_point"], np.int32),
            np.ascontiguousarray(w["edge_obs"], np.float64), np.ascontiguousarray(w["edge_inv_sigma2"], np.float64),
            np.ascontiguousarray(w["edge_stereo"], np.uint8)]
    rc = L.lba_adaptor_test(solver_lib.encode() if solver_lib else None, npz, npt, ne, *[a.ctypes.data for a in arrs],
                            w["fx"], w["fy"], w["cx"], w["cy"], w["bf"], init_kf_pose, bad_point, stop_flag,
                            *[out[k].ctypes.data for k in ("pose_q", "pose_t", "points", "erased", "counts", "hubers", "inv_sigma2", "obs",
                                                          "edge_pose", "edge_point")])
    out["rc"] = rc
    return out


def _local_only(w, local_pose=None):
    """Only MapPoints seen in a LOCAL key-frame enter the window (src/Optimizer.cc:1609-1634); points seen by fixed key-frames
    alone, and their observations, stay out.  -> (window restricted to them, original index of every kept point)"""
    w = dict(w)
    ep, el = np.asarray(w["edge_pose"]), np.asarray(w["edge_point"])
    local = np.zeros(int(w["n_points"]), bool)
    local_pose = (np.asarray(w["pose_fixed"]) == 0) if local_pose is None else local_pose
    local[el[local_pose[ep]]] = True
    keep_e = local[el]
    remap = np.cumsum(local) - 1
    for k in ("edge_pose", "edge_obs", "edge_inv_sigma2", "edge_stereo"):
        w[k] = np.asarray(w[k])[keep_e]
    w["edge_point"] = remap[el[keep_e]].astype(np.int32)
    w["points"] = np.asarray(w["points"])[local]
    w["n_points"], w["n_edges"] = int(local.sum()), int(keep_e.sum())
    return w, np.flatnonzero(local)


def _expected(oracle, w, init_kf_pose=-1):
    w = dict(w)
    local_pose = np.asarray(w["pose_fixed"]) == 0   # the initial key-frame stays a LOCAL key-frame, it is only held fixed
    if init_kf_pose >= 0:
        pf = np.array(w["pose_fixed"], np.uint8).copy()
        pf[init_kf_pose] = 1
        w["pose_fixed"] = pf
    w["huber_mono"] = float(np.float32(np.sqrt(5.991)))
    w["huber_stereo"] = float(np.float32(np.sqrt(7.815)))
    wl, kept = _local_only(w, local_pose)
    ro = oracle.lba_solve(wl)
    # back to the original point numbering (points outside the window keep their position, their edges are never classified)
    full = dict(ro)
    full["points"] = np.asarray(w["points"], np.float64).copy()
    full["points"][kept] = ro["points"]
    wl["edge_point"] = kept[np.asarray(wl["edge_point"])]
    wl["n_points_all"] = int(w["n_points"])
    return wl, full


def _check(out, w, ro, skip_point=-1):
    ne = int(w["n_edges"])
    assert out["rc"] >= 0
    free = np.array(w["pose_fixed"]) == 0
    # poses and points are written back through float: compare with the double solution rounded the same way
    q, qo = out["pose_q"].astype(np.float64), ro["pose_q"]
    sign = np.sign((q * qo).sum(1, keepdims=True))
    assert np.abs(q * sign - qo)[free].max() < 2e-7 and np.abs(out["pose_t"] - ro["pose_t"])[free].max() < 1e-6
    pts_ok = np.abs(out["points"] - ro["points"]).max(1) < 2e-6
    if skip_point >= 0:
        pts_ok[skip_point] = True
    assert pts_ok.all()
    # erased observations == edges with chi2 over the gate or behind the camera (:1961-1999)
    gate = np.where(np.array(w["edge_stereo"]) > 0, 7.815, 5.991)
    bad = (ro["edge_chi2"] > gate) | (ro["edge_depth_positive"] == 0)
    if skip_point >= 0:
        bad &= np.array(w["edge_point"]) != skip_point
    want = sorted(zip(np.array(w["edge_pose"])[bad].tolist(), np.array(w["edge_point"])[bad].tolist()))
    got = sorted(map(tuple, out["erased"][:out["counts"][4]].tolist()))
    assert got == want and len(want) > 0
    return ne


def test_adaptor_flattening_and_casts(harness, oracle):
    """What the numeric core receives: float values widened to double, the float-rounded Huber deltas, edges grouped by MapPoint
    in the order the local key-frames list them, the current key-frame as pose 0, local before fixed key-frames."""
    w = _window32(3, n_free=4, n_fixed=2, n_points=150)
    out = _run(harness, w, oracle._LIB_PATH if hasattr(oracle, "_LIB_PATH") else os.path.join(ROOT, "oracle", "libgfs_oracle.so"))
    wl, _ = _local_only(w)
    ne = int(wl["n_edges"])
    assert ne < int(w["n_edges"])          # the synthetic window does hold points seen by fixed key-frames only
    assert out["rc"] == ne * 1000 + int(w["n_poses"])
    assert out["hubers"][0] == float(np.float32(np.sqrt(5.991))) and out["hubers"][1] == float(np.float32(np.sqrt(7.815)))
    assert out["hubers"][0] != np.sqrt(5.991)  # `const float thHuberMono = sqrt(5.991)`: the rounding is part of the reference
    assert sorted(out["inv_sigma2"][:ne].tolist()) == sorted(np.asarray(wl["edge_inv_sigma2"]).tolist())
    assert (out["inv_sigma2"][:ne] == out["inv_sigma2"][:ne].astype(np.float32)).all()
    assert (out["obs"][:ne] == out["obs"][:ne].astype(np.float32)).all()
    pts = out["edge_point"][:ne]
    assert (np.diff(pts) >= 0).all() and pts[0] == 0 and pts[-1] == int(wl["n_points"]) - 1   # one MapPoint after the other
    nfree = int((np.array(w["pose_fixed"]) == 0).sum())
    assert out["counts"][1] == nfree and out["counts"][0] == int(w["n_poses"]) - nfree and out["counts"][2] == ne
    assert out["counts"][7] == -7  # num_MPs is not written (the reference never does either)


@pytest.mark.parametrize("seed,init_kf", [(1, -1), (2, 1)])
def test_adaptor_end_to_end_with_oracle_solver(harness, oracle, seed, init_kf):
    """Gather -> solve (CPU oracle) -> classification -> write-back; also with a local key-frame that is the map's initial
    key-frame (setFixed(pKFi->mnId == pMap->GetInitKFid()), num_fixedKF + 1)."""
    w = _window32(seed, n_free=5, n_fixed=3, n_points=300)
    free_idx = np.flatnonzero(np.array(w["pose_fixed"]) == 0)
    init_pose = int(free_idx[init_kf]) if init_kf >= 0 else -1
    w2, ro = _expected(oracle, w, init_pose)
    out = _run(harness, w, os.path.join(ROOT, "oracle", "libgfs_oracle.so"), init_kf_pose=init_pose)
    _check(out, w2, ro)
    nfixed_listed = int((np.array(w["pose_fixed"]) != 0).sum())
    assert out["counts"][0] == nfixed_listed + (1 if init_kf >= 0 else 0)
    assert out["counts"][1] == len(free_idx)                  # num_OptKF counts every local key-frame
    assert out["counts"][3] == 1                               # pMap->IncreaseChangeIndex()
    assert out["counts"][5] == len(free_idx) and out["counts"][6] == int(w2["n_points"])   # SetPose / UpdateNormalAndDepth calls


def test_adaptor_stop_flag_and_no_fixed_keyframe(harness, oracle):
    w = _window32(5, n_free=3, n_fixed=2, n_points=80)
    lib = os.path.join(ROOT, "oracle", "libgfs_oracle.so")
    out = _run(harness, w, lib, stop_flag=1)       # *pbStopFlag set: return before the optimisation, nothing written back
    assert out["rc"] == 0 and out["counts"][3] == 0 and out["counts"][5] == 0 and out["counts"][2] == int(_local_only(w)[0]["n_edges"])
    assert np.array_equal(out["points"], np.asarray(w["points"], np.float32))
    w0 = dict(w)
    keep = np.array(w["pose_fixed"])[np.array(w["edge_pose"])] == 0   # no observation from a fixed key-frame: 0 fixed KFs -> abort
    for k in ("edge_pose", "edge_point", "edge_obs", "edge_inv_sigma2", "edge_stereo"):
        w0[k] = np.asarray(w[k])[keep]
    w0["n_edges"] = int(keep.sum())
    out = _run(harness, w0, lib)
    assert out["rc"] == 0 and out["counts"][0] == 0 and out["counts"][3] == 0


def test_adaptor_refuses_second_camera_observations(harness, oracle):
    """A key-frame of a two-camera rig (mpCamera2 != nullptr) whose RIGHT image observed a map point would need
    EdgeSE3ProjectXYZToBody (src/Optimizer.cc:1906-1949): not implemented -- the adaptor must refuse the window loudly (an exception,
    -1 from the harness) before anything is optimised or written back, not drop the observation."""
    w = _window32(6, n_free=3, n_fixed=2, n_points=60)
    out = _run(harness, w, os.path.join(ROOT, "oracle", "libgfs_oracle.so"), init_kf_pose=-2)
    assert out["rc"] == -1
    assert not out["points"].any() and not out["pose_t"].any()  # the outputs were never filled


@pytest.mark.gpu
def test_adaptor_end_to_end_on_gpu(harness, gpu_api, oracle):
    w = _window32(7, n_free=6, n_fixed=3, n_points=500)
    w2, ro = _expected(oracle, w)
    out = _run(harness, w, None)
    _check(out, w2, ro)
    assert out["counts"][3] == 1


@pytest.mark.gpu
def test_adaptor_stop_flag_raised_while_the_adjustment_runs(harness, gpu_api, oracle):
    """mbAbortBA raised by another thread DURING LocalBundleAdjustment: the solver reads the caller's bool live (top of every
    iteration and after every trial step, g2o's setForceStopFlag, src/Optimizer.cc:1679), stops early, and -- like the reference, which
    checks the flag only before optimize() (:1955-1956) -- the state reached so far is classified and written back."""
    w = _window32(11, n_free=20, n_fixed=5, n_points=3000)
    full = _run(harness, w, None, stop_flag=10_000_000)   # raised far too late: the complete optimisation
    assert full["rc"] > 0 and full["hubers"][1] >= 3
    early = _run(harness, w, None, stop_flag=150)          # raised 150 us in: a window of this size needs ~2 ms
    assert early["rc"] > 0 and early["counts"][3] == 1     # written back all the same
    assert 0 <= early["hubers"][1] < full["hubers"][1], (early["hubers"][1], full["hubers"][1])
