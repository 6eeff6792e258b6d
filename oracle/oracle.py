"""ctypes binding of the CPU oracle (oracle/libgfs_oracle.so).

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module (see oracle/gfs_oracle.h).  PARITY UNPINNED, except the voxel sort and the k-NN result
container, which tests/test_oracle_ref.py pins against the reference's own headers compiled into oracle/_ref (ref_lib()).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"),
     ("class_id", "<i4")]
)


class GicpCfg(C.Structure):
    _fields_ = [
        ("num_threads", C.c_int32),
        ("downsampling_resolution", C.c_double),
        ("max_correspondence_distance", C.c_double),
        ("rotation_eps", C.c_double),
        ("translation_eps", C.c_double),
        ("max_iterations", C.c_int32),
        ("num_neighbors", C.c_int32),
    ]


class GicpResult(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16),
        ("converged", C.c_int32),
        ("iterations", C.c_uint64),
        ("num_inliers", C.c_uint64),
        ("H", C.c_double * 36),
        ("b", C.c_double * 6),
        ("error", C.c_double),
        ("n_target_ds", C.c_int32),
        ("n_source_ds", C.c_int32),
        ("n_linearize", C.c_int32),
        ("n_error_evals", C.c_int32),
    ]


class LbaProblem(C.Structure):
    _fields_ = [
        ("n_poses", C.c_int32),
        ("n_points", C.c_int32),
        ("n_edges", C.c_int32),
        ("pose_q", C.c_void_p),
        ("pose_t", C.c_void_p),
        ("pose_fixed", C.c_void_p),
        ("points", C.c_void_p),
        ("edge_pose", C.c_void_p),
        ("edge_point", C.c_void_p),
        ("edge_obs", C.c_void_p),
        ("edge_inv_sigma2", C.c_void_p),
        ("edge_stereo", C.c_void_p),
        ("fx", C.c_double),
        ("fy", C.c_double),
        ("cx", C.c_double),
        ("cy", C.c_double),
        ("bf", C.c_double),
        ("huber_mono", C.c_double),
        ("huber_stereo", C.c_double),
        ("iterations", C.c_int32),
    ]


class LbaSolution(C.Structure):
    _fields_ = [
        ("pose_q", C.c_void_p),
        ("pose_t", C.c_void_p),
        ("points", C.c_void_p),
        ("edge_chi2", C.c_void_p),
        ("edge_depth_positive", C.c_void_p),
        ("iterations_run", C.c_int32),
        ("final_chi2", C.c_double),
        ("final_lambda", C.c_double),
    ]


def build():
    subprocess.run(["make", "-s", "-C", _HERE, "libgfs_oracle.so"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgfs_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.gfso_orb_create.restype = C.c_void_p
        L.gfso_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.gfso_orb_destroy.argtypes = [C.c_void_p]
        L.gfso_orb_get_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.gfso_orb_extract.restype = C.c_int
        L.gfso_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.gfso_orb_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.gfso_orb_get_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.gfso_orb_get_blurred.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.gfso_orb_num_candidates.argtypes = [C.c_void_p, C.c_int]
        L.gfso_orb_get_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gfso_orb_num_level_keypoints.argtypes = [C.c_void_p, C.c_int]
        L.gfso_orb_get_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.gfso_resize_area_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.gfso_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int]
        L.gfso_fast_atan2.restype = C.c_float
        L.gfso_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.gfso_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.gfso_distribute_octree.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.gfso_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.gfso_bf_match_hamming.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        if hasattr(L, "gfso_gicp_align"):
            L.gfso_gicp_default_cfg.argtypes = [C.POINTER(GicpCfg)]
            L.gfso_gicp_align.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(GicpCfg),
                                          C.POINTER(GicpResult)]
            L.gfso_gicp_preprocess.restype = C.c_int
            L.gfso_gicp_preprocess.argtypes = [C.c_void_p, C.c_int, C.POINTER(GicpCfg), C.c_void_p, C.c_void_p,
                                               C.c_void_p]
            L.gfso_knn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            L.gfso_eig3_direct.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            L.gfso_se3_exp.argtypes = [C.c_void_p, C.c_void_p]
        if hasattr(L, "gfso_lba_solve"):
            L.gfso_lba_solve.restype = C.c_int
            L.gfso_lba_solve.argtypes = [C.POINTER(LbaProblem), C.POINTER(LbaSolution)]
            L.gfso_lba_linearize.restype = C.c_double
            L.gfso_lba_linearize.argtypes = [C.POINTER(LbaProblem)] + [C.c_void_p] * 6
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OrbOracle:
    """Mirror of ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:46-118)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, blur_variant=0):
        self.L = lib()
        self.nlevels = nlevels
        self.h = self.L.gfso_orb_create(nfeatures, scale_factor, nlevels, ini_th, min_th, blur_variant)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.gfso_orb_destroy(self.h)
            self.h = None

    def set_threads(self, n):
        self.L.gfso_orb_set_threads(C.c_void_p(self.h), int(n))

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        feats = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.gfso_orb_get_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2), _p(feats), _p(umax))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, feats=feats, umax=umax)

    def extract(self, img, lapping=(0, 0), cap=None):
        """-> (mono_index, keypoints[KP_DTYPE], descriptors[N,32] u8)"""
        img = np.ascontiguousarray(img, np.uint8)
        rows, cols = img.shape
        n = C.c_int(0)
        r = self.L.gfso_orb_extract(self.h, _p(img), rows, cols, cols, lapping[0], lapping[1], None, None, 0,
                                    C.byref(n))
        if r < 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        N = n.value
        kps = np.zeros(N, KP_DTYPE)
        desc = np.zeros((N, 32), np.uint8)
        r = self.L.gfso_orb_extract(self.h, _p(img), rows, cols, cols, lapping[0], lapping[1], _p(kps), _p(desc), N,
                                    C.byref(n))
        return r, kps, desc

    def level_size(self, l):
        r, c = C.c_int(), C.c_int()
        self.L.gfso_orb_level_size(self.h, l, C.byref(r), C.byref(c))
        return r.value, c.value

    def level(self, l):
        r, c = self.level_size(l)
        a = np.zeros((r, c), np.uint8)
        self.L.gfso_orb_get_level(self.h, l, _p(a))
        return a

    def blurred(self, l):
        r, c = self.level_size(l)
        a = np.zeros((r, c), np.uint8)
        self.L.gfso_orb_get_blurred(self.h, l, _p(a))
        return a

    def candidates(self, l):
        n = self.L.gfso_orb_num_candidates(self.h, l)
        x, y, s = (np.zeros(n, np.int32) for _ in range(3))
        if n:
            self.L.gfso_orb_get_candidates(self.h, l, _p(x), _p(y), _p(s))
        return x, y, s

    def level_keypoints(self, l):
        n = self.L.gfso_orb_num_level_keypoints(self.h, l)
        k = np.zeros(n, KP_DTYPE)
        if n:
            self.L.gfso_orb_get_level_keypoints(self.h, l, _p(k))
        return k


def resize_area(src, drows, dcols):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((drows, dcols), np.uint8)
    lib().gfso_resize_area_u8(_p(src), src.shape[0], src.shape[1], src.shape[1], _p(dst), drows, dcols, dcols)
    return dst


def fast9_16(img, threshold, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    x, y, s = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().gfso_fast9_16(_p(img), img.shape[0], img.shape[1], img.shape[1], threshold, int(nonmax), _p(x), _p(y),
                            _p(s), cap)
    return x[:n], y[:n], s[:n]


def fast_atan2(y, x):
    return float(lib().gfso_fast_atan2(float(y), float(x)))


def gaussian_blur7(img, variant=0):
    img = np.ascontiguousarray(img, np.uint8)
    dst = np.zeros_like(img)
    lib().gfso_gaussian_blur7(_p(img), img.shape[0], img.shape[1], img.shape[1], _p(dst), img.shape[1], variant)
    return dst


def distribute_octree(x, y, resp, min_x, max_x, min_y, max_y, n_features):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    resp = np.ascontiguousarray(resp, np.float32)
    out = np.zeros(max(len(x), 1), np.int32)
    n = lib().gfso_distribute_octree(_p(x), _p(y), _p(resp), len(x), min_x, max_x, min_y, max_y, n_features, _p(out),
                                     len(out))
    return out[:n]


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return int(lib().gfso_descriptor_distance(_p(a), _p(b)))


def bf_match(q, t, nthreads=1):
    """-> (train_idx[nq], dist[nq]) or empty arrays when the train set is empty."""
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    ti = np.zeros(len(q), np.int32)
    di = np.zeros(len(q), np.int32)
    n = lib().gfso_bf_match_hamming(_p(q), len(q), _p(t), len(t), _p(ti), _p(di), nthreads)
    return ti[:n], di[:n]


def gicp_set_stable_voxel_order(on):
    lib().gfso_gicp_set_stable_voxel_order(int(on))


def quick_sort_perm(keys):
    """small_gicp quick_sort_omp on (key, index) pairs -> the permutation (input index of every sorted element)."""
    k = np.ascontiguousarray(keys, np.uint64).copy()
    idx = np.arange(len(k), dtype=np.uint64)
    lib().gfso_quick_sort_pairs(_p(k), _p(idx), len(k))
    return idx.astype(np.int64), k


_REF = False


def ref_lib():
    """oracle/_ref/libgfs_ref_small_gicp.so: the REFERENCE's own sort_omp.hpp / knn_result.hpp compiled from /root/reference by
    oracle/ref_build.sh (the only hot-path sources that build without OpenCV / Eigen).  None when it has not been built."""
    global _REF
    if _REF is False:
        path = os.path.join(_HERE, "_ref", "libgfs_ref_small_gicp.so")
        _REF = C.CDLL(path) if os.path.exists(path) else None
    return _REF


def ref_quick_sort_perm(keys, num_threads=4):
    """The reference's quick_sort_omp (util/sort_omp.hpp:58-103) as voxelgrid_sampling_omp calls it -> (permutation, sorted keys)."""
    k = np.ascontiguousarray(keys, np.uint64).copy()
    idx = np.arange(len(k), dtype=np.uint64)
    ref_lib().gfsref_quick_sort_omp(_p(k), _p(idx), len(k), int(num_threads))
    return idx.astype(np.int64), k


def knn_push_stream(k, index, distance, ref=False, static_one=False):
    """(index, distance) pushes into KnnResult of capacity k: the restated container, or the reference's (ref=True)."""
    index = np.ascontiguousarray(index, np.uint64)
    distance = np.ascontiguousarray(distance, np.float64)
    io, do = np.zeros(k, np.uint64), np.zeros(k, np.float64)
    if ref:
        n = ref_lib().gfsref_knn_push_stream(k, int(static_one), _p(index), _p(distance), len(index), _p(io), _p(do))
    else:
        n = lib().gfso_knn_push_stream(k, _p(index), _p(distance), len(index), _p(io), _p(do))
    return n, io, do


def antiqsort_keys(n):
    out = np.zeros(n, np.int32)
    lib().gfso_antiqsort_keys(n, _p(out))
    return out


def gicp_set_threads(n):
    lib().gfso_gicp_set_threads(int(n))


def gicp_default_cfg():
    c = GicpCfg()
    lib().gfso_gicp_default_cfg(C.byref(c))
    return c


def gicp_align(target, source, init_T=None, cfg=None):
    target = np.ascontiguousarray(target, np.float32).reshape(-1, 4)
    source = np.ascontiguousarray(source, np.float32).reshape(-1, 4)
    T0 = np.eye(4) if init_T is None else np.asarray(init_T, np.float64)
    T0c = np.ascontiguousarray(T0.T.reshape(-1))  # column-major
    cfg = cfg or gicp_default_cfg()
    res = GicpResult()
    lib().gfso_gicp_align(_p(target), len(target), _p(source), len(source), _p(T0c), C.byref(cfg), C.byref(res))
    return dict(
        T=np.array(res.T).reshape(4, 4).T.copy(), converged=bool(res.converged), iterations=int(res.iterations),
        num_inliers=int(res.num_inliers), H=np.array(res.H).reshape(6, 6).T.copy(), b=np.array(res.b),
        error=float(res.error), n_target_ds=res.n_target_ds, n_source_ds=res.n_source_ds,
        n_linearize=res.n_linearize, n_error_evals=res.n_error_evals)


def gicp_preprocess(points, cfg=None):
    points = np.ascontiguousarray(points, np.float32).reshape(-1, 4)
    cfg = cfg or gicp_default_cfg()
    n = len(points)
    pts = np.zeros((n, 4))
    covs = np.zeros((n, 16))
    nrm = np.zeros((n, 4))
    m = lib().gfso_gicp_preprocess(_p(points), n, C.byref(cfg), _p(pts), _p(covs), _p(nrm))
    return pts[:m], covs[:m].reshape(m, 4, 4).transpose(0, 2, 1).copy(), nrm[:m]


def knn(points, queries, k):
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 4)
    queries = np.ascontiguousarray(queries, np.float64).reshape(-1, 4)
    idx = np.zeros((len(queries), k), np.int64)
    sqd = np.zeros((len(queries), k))
    lib().gfso_knn(_p(points), len(points), _p(queries), len(queries), k, _p(idx), _p(sqd))
    return idx, sqd


def eig3_direct(m):
    m = np.ascontiguousarray(np.asarray(m, np.float64).T)
    ev = np.zeros(3)
    evec = np.zeros((3, 3))
    lib().gfso_eig3_direct(_p(m), _p(ev), _p(evec))
    return ev, evec.T.copy()


def se3_exp(twist):
    tw = np.ascontiguousarray(twist, np.float64)
    T = np.zeros(16)
    lib().gfso_se3_exp(_p(tw), _p(T))
    return T.reshape(4, 4).T.copy()


def _lba_struct(cls, prob):
    P = cls()
    keep = {}
    for name, dt in (("pose_q", np.float64), ("pose_t", np.float64), ("pose_fixed", np.uint8), ("points", np.float64),
                     ("edge_pose", np.int32), ("edge_point", np.int32), ("edge_obs", np.float64),
                     ("edge_inv_sigma2", np.float64), ("edge_stereo", np.uint8)):
        keep[name] = np.ascontiguousarray(prob[name], dt)
        setattr(P, name, keep[name].ctypes.data)
    for name in ("n_poses", "n_points", "n_edges", "iterations"):
        setattr(P, name, int(prob[name]))
    for name in ("fx", "fy", "cx", "cy", "bf", "huber_mono", "huber_stereo"):
        setattr(P, name, float(prob[name]))
    return P, keep


def lba_solve(prob):
    """-> dict(pose_q, pose_t, points, edge_chi2, edge_depth_positive, iterations_run, final_chi2, final_lambda)"""
    P, keep = _lba_struct(LbaProblem, prob)
    out = dict(pose_q=np.zeros((P.n_poses, 4)), pose_t=np.zeros((P.n_poses, 3)), points=np.zeros((P.n_points, 3)),
               edge_chi2=np.zeros(P.n_edges), edge_depth_positive=np.zeros(P.n_edges, np.uint8))
    S = LbaSolution()
    for k, v in out.items():
        setattr(S, k, v.ctypes.data)
    lib().gfso_lba_solve(C.byref(P), C.byref(S))
    out.update(iterations_run=S.iterations_run, final_chi2=S.final_chi2, final_lambda=S.final_lambda)
    return out


def lba_linearize(prob):
    """One buildSystem at the initial estimates -> dict(Hpp [nf,6,6], Hll [np,3,3], Hpl [ne,6,3], bp, bl, edge_chi2, chi2)"""
    P, keep = _lba_struct(LbaProblem, prob)
    nf = int((np.asarray(prob["pose_fixed"]) == 0).sum())
    Hpp = np.zeros((nf, 36)); Hll = np.zeros((P.n_points, 9)); Hpl = np.zeros((P.n_edges, 18))
    bp = np.zeros((nf, 6)); bl = np.zeros((P.n_points, 3)); chi = np.zeros(P.n_edges)
    tot = lib().gfso_lba_linearize(C.byref(P), _p(Hpp), _p(Hll), _p(Hpl), _p(bp), _p(bl), _p(chi))
    return dict(Hpp=Hpp.reshape(nf, 6, 6).transpose(0, 2, 1).copy(), Hll=Hll.reshape(-1, 3, 3).transpose(0, 2, 1).copy(),
                Hpl=Hpl.reshape(-1, 3, 6).transpose(0, 2, 1).copy(), bp=bp, bl=bl, edge_chi2=chi, chi2=float(tot))


class PoseProblem(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3), ("n_obs", C.c_int32), ("xw", C.c_void_p), ("obs", C.c_void_p),
                ("inv_sigma2", C.c_void_p), ("stereo", C.c_void_p), ("fx", C.c_double), ("fy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double), ("n_rounds", C.c_int32), ("its", C.c_int32)]


class PoseSolution(C.Structure):
    _fields_ = [("outlier", C.c_void_p), ("chi2", C.c_void_p), ("q", C.c_double * 4), ("t", C.c_double * 3),
                ("avg_reproj_error", C.c_float), ("n_inliers", C.c_int32), ("rounds_run", C.c_int32),
                ("iterations_run", C.c_int32)]


def pose_optimization(prob):
    """Optimizer::PoseOptimization restatement (oracle/pose_oracle.cpp) on one frame dict:
    q, t, xw [n,3], obs [n,3], inv_sigma2 [n], stereo [n], fx, fy, cx, cy, bf (+ n_rounds = 4, its = 10)."""
    P, S = PoseProblem(), PoseSolution()
    xw = np.ascontiguousarray(prob["xw"], np.float64).reshape(-1, 3)
    obs = np.ascontiguousarray(prob["obs"], np.float64).reshape(-1, 3)
    w = np.ascontiguousarray(prob["inv_sigma2"], np.float32)
    st = np.ascontiguousarray(prob["stereo"], np.uint8)
    n = len(xw)
    outlier = np.zeros(max(n, 1), np.uint8)
    chi2 = np.zeros(max(n, 1), np.float64)
    P.q[:] = [float(v) for v in prob["q"]]
    P.t[:] = [float(v) for v in prob["t"]]
    P.n_obs = n
    P.xw, P.obs, P.inv_sigma2, P.stereo = xw.ctypes.data, obs.ctypes.data, w.ctypes.data, st.ctypes.data
    for name in ("fx", "fy", "cx", "cy", "bf"):
        setattr(P, name, float(prob[name]))
    P.n_rounds = int(prob.get("n_rounds", 4))
    P.its = int(prob.get("its", 10))
    S.outlier, S.chi2 = outlier.ctypes.data, chi2.ctypes.data
    L = lib()
    L.gfso_pose_optimization.restype = C.c_int
    L.gfso_pose_optimization.argtypes = [C.POINTER(PoseProblem), C.POINTER(PoseSolution)]
    L.gfso_pose_optimization(C.byref(P), C.byref(S))
    return dict(outlier=outlier[:n].astype(bool), chi2=chi2[:n].copy(), q=np.array(S.q[:]), t=np.array(S.t[:]),
                avg_reproj_error=float(S.avg_reproj_error), n_inliers=int(S.n_inliers), rounds_run=int(S.rounds_run),
                iterations_run=int(S.iterations_run))


class SbpProblem(C.Structure):
    _fields_ = [("n_last", C.c_int32), ("last_xw", C.c_void_p), ("last_desc", C.c_void_p), ("last_octave", C.c_void_p),
                ("last_angle", C.c_void_p), ("last_mp_has_obs", C.c_void_p), ("n_cur", C.c_int32), ("cur_xy", C.c_void_p),
                ("cur_octave", C.c_void_p), ("cur_angle", C.c_void_p), ("cur_u_right", C.c_void_p), ("cur_desc", C.c_void_p),
                ("cur_has_mp_obs", C.c_void_p), ("Tcw_q", C.c_float * 4), ("Tcw_t", C.c_float * 3), ("Tlw_q", C.c_float * 4),
                ("Tlw_t", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("bf", C.c_float), ("b", C.c_float), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float),
                ("max_y", C.c_float), ("grid_w_inv", C.c_float), ("grid_h_inv", C.c_float), ("scale_factors", C.c_void_p),
                ("n_levels", C.c_int32), ("th", C.c_float), ("mono", C.c_int32), ("check_orientation", C.c_int32)]


def search_by_projection(prob):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) restatement (oracle/sbp_oracle.cpp).
    prob: the keys of gfs_sbp_problem; cur_kps_un is a structured array with x, y, angle, octave fields.
    Returns (cur_match int32 [n_cur], nmatches)."""
    P = SbpProblem()
    kps = prob["cur_kps_un"]
    keep = dict(last_xw=np.ascontiguousarray(prob["last_xw"], np.float32).reshape(-1, 3),
                last_desc=np.ascontiguousarray(prob["last_desc"], np.uint8).reshape(-1, 32),
                last_octave=np.ascontiguousarray(prob["last_octave"], np.int32),
                last_angle=np.ascontiguousarray(prob["last_angle"], np.float32),
                last_mp_has_obs=np.ascontiguousarray(prob["last_mp_has_obs"], np.uint8),
                cur_xy=np.ascontiguousarray(np.stack([kps["x"], kps["y"]], 1) if len(kps) else np.zeros((0, 2)), np.float32),
                cur_octave=np.ascontiguousarray(kps["octave"], np.int32), cur_angle=np.ascontiguousarray(kps["angle"], np.float32),
                cur_u_right=np.ascontiguousarray(prob["cur_u_right"], np.float32),
                cur_desc=np.ascontiguousarray(prob["cur_desc"], np.uint8).reshape(-1, 32),
                cur_has_mp_obs=np.ascontiguousarray(prob["cur_has_mp_obs"], np.uint8),
                scale_factors=np.ascontiguousarray(prob["scale_factors"], np.float32))
    P.n_last, P.n_cur = len(keep["last_xw"]), len(keep["cur_xy"])
    for name, a in keep.items():
        setattr(P, name, a.ctypes.data)
    for name in ("Tcw_q", "Tcw_t", "Tlw_q", "Tlw_t"):
        getattr(P, name)[:] = [float(np.float32(v)) for v in prob[name]]
    for name in ("fx", "fy", "cx", "cy", "bf", "b", "min_x", "max_x", "min_y", "max_y", "grid_w_inv", "grid_h_inv", "th"):
        setattr(P, name, float(np.float32(prob[name])))
    P.n_levels = len(keep["scale_factors"])
    P.mono = int(prob.get("mono", 0))
    P.check_orientation = int(prob.get("check_orientation", 1))
    out = np.full(max(P.n_cur, 1), -9, np.int32)
    L = lib()
    L.gfso_search_by_projection.restype = C.c_int
    L.gfso_search_by_projection.argtypes = [C.POINTER(SbpProblem), C.c_void_p]
    n = L.gfso_search_by_projection(C.byref(P), out.ctypes.data)
    return out[:P.n_cur].copy(), int(n)


class SbpMapProblem(C.Structure):
    _fields_ = [("n_mp", C.c_int32), ("mp_proj", C.c_void_p), ("mp_level", C.c_void_p), ("mp_view_cos", C.c_void_p),
                ("mp_desc", C.c_void_p), ("mp_has_obs", C.c_void_p), ("n_cur", C.c_int32), ("cur_xy", C.c_void_p),
                ("cur_octave", C.c_void_p), ("cur_u_right", C.c_void_p), ("cur_desc", C.c_void_p), ("cur_has_mp_obs", C.c_void_p),
                ("min_x", C.c_float), ("min_y", C.c_float), ("grid_w_inv", C.c_float), ("grid_h_inv", C.c_float),
                ("scale_factors", C.c_void_p), ("n_levels", C.c_int32), ("th", C.c_float), ("nn_ratio", C.c_float)]


def search_by_projection_map(prob):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th, ...) restatement (oracle/sbp_oracle.cpp) -> (cur_match, nmatches)."""
    P = SbpMapProblem()
    kps = prob["cur_kps_un"]
    keep = dict(mp_proj=np.ascontiguousarray(prob["mp_proj"], np.float32).reshape(-1, 3),
                mp_level=np.ascontiguousarray(prob["mp_level"], np.int32),
                mp_view_cos=np.ascontiguousarray(prob["mp_view_cos"], np.float32),
                mp_desc=np.ascontiguousarray(prob["mp_desc"], np.uint8).reshape(-1, 32),
                mp_has_obs=np.ascontiguousarray(prob["mp_has_obs"], np.uint8),
                cur_xy=np.ascontiguousarray(np.stack([kps["x"], kps["y"]], 1) if len(kps) else np.zeros((0, 2)), np.float32),
                cur_octave=np.ascontiguousarray(kps["octave"], np.int32),
                cur_u_right=np.ascontiguousarray(prob["cur_u_right"], np.float32),
                cur_desc=np.ascontiguousarray(prob["cur_desc"], np.uint8).reshape(-1, 32),
                cur_has_mp_obs=np.ascontiguousarray(prob["cur_has_mp_obs"], np.uint8),
                scale_factors=np.ascontiguousarray(prob["scale_factors"], np.float32))
    P.n_mp, P.n_cur = len(keep["mp_proj"]), len(keep["cur_xy"])
    for name, a in keep.items():
        setattr(P, name, a.ctypes.data)
    for name in ("min_x", "min_y", "grid_w_inv", "grid_h_inv", "th", "nn_ratio"):
        setattr(P, name, float(np.float32(prob[name])))
    P.n_levels = len(keep["scale_factors"])
    out = np.full(max(P.n_cur, 1), -9, np.int32)
    L = lib()
    L.gfso_search_by_projection_map.restype = C.c_int
    L.gfso_search_by_projection_map.argtypes = [C.POINTER(SbpMapProblem), C.c_void_p]
    n = L.gfso_search_by_projection_map(C.byref(P), out.ctypes.data)
    return out[:P.n_cur].copy(), int(n)


def gms_inlier_mask(kp1, size1, kp2, size2, query_idx, train_idx):
    """gms_matcher(...).GetInlierMask(mask, false, false) restatement (oracle/gms_oracle.cpp); size = (width, height).
    kp1 / kp2: structured key-point arrays with x, y fields.  Returns (mask bool [n_matches], n_inliers)."""
    xy1 = np.ascontiguousarray(np.stack([kp1["x"], kp1["y"]], 1) if len(kp1) else np.zeros((0, 2)), np.float32)
    xy2 = np.ascontiguousarray(np.stack([kp2["x"], kp2["y"]], 1) if len(kp2) else np.zeros((0, 2)), np.float32)
    q = np.ascontiguousarray(query_idx, np.int32)
    t = np.ascontiguousarray(train_idx, np.int32)
    mask = np.zeros(max(len(q), 1), np.uint8)
    L = lib()
    L.gfso_gms_inlier_mask.restype = C.c_int
    L.gfso_gms_inlier_mask.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p]
    n = L.gfso_gms_inlier_mask(xy1.ctypes.data, len(xy1), int(size1[0]), int(size1[1]), xy2.ctypes.data, len(xy2), int(size2[0]),
                               int(size2[1]), q.ctypes.data, t.ctypes.data, len(q), mask.ctypes.data)
    return mask[:len(q)].astype(bool), int(n)


def depth_to_cloud(depth, downsample, fx, fy, cx, cy):
    depth = np.ascontiguousarray(depth, np.float32)
    rows, cols = depth.shape if depth.ndim == 2 else (0, 0)
    out = np.zeros((max(rows * cols, 1), 4), np.float32)
    L = lib()
    L.gfso_depth_to_cloud.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                      C.c_float, C.c_void_p, C.c_int]
    n = L.gfso_depth_to_cloud(_p(depth), rows, cols, cols, downsample, fx, fy, cx, cy, _p(out), len(out))
    return out[:n].copy()


def stereo_from_rgbd(kps, depth, bf, kps_un_x=None):
    depth = np.ascontiguousarray(depth, np.float32)
    kps = np.ascontiguousarray(kps)
    n = len(kps)
    ur = np.zeros(max(n, 1), np.float32)
    vd = np.zeros(max(n, 1), np.float32)
    L = lib()
    L.gfso_stereo_from_rgbd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    unx = np.ascontiguousarray(kps_un_x, np.float32) if kps_un_x is not None else None
    L.gfso_stereo_from_rgbd(_p(kps), _p(unx) if unx is not None else None, n, _p(depth), depth.shape[1], bf, _p(ur), _p(vd))
    return ur[:n], vd[:n]


KLT_USE_INITIAL_FLOW, KLT_GET_MIN_EIGENVALS = 4, 8


def klt_set_accumulation(mode):
    """0 exact integer sums (default), 1 OpenCV's scalar float loop, 2 four-lane float model (oracle/klt_oracle.cpp header)."""
    lib().gfso_klt_set_accumulation(int(mode))


def klt_layout(width, height, win, max_level=3):
    """Level sizes and offsets of cv::buildOpticalFlowPyramid(img, pyr, Size(win, win), max_level) in the shared storage layout.
    Returns (lw, lh, off) with len(off) == levels + 1."""
    lw = np.zeros(8, np.int32)
    lh = np.zeros(8, np.int32)
    off = np.zeros(9, np.int64)
    L = lib()
    L.gfso_klt_layout.restype = C.c_int
    L.gfso_klt_layout.argtypes = [C.c_int] * 4 + [C.c_void_p] * 3
    n = L.gfso_klt_layout(width, height, win, max_level, _p(lw), _p(lh), _p(off))
    return lw[:n].copy(), lh[:n].copy(), off[:n + 1].copy()


def klt_build_pyramid(img, win, max_level=3):
    """cv::buildOpticalFlowPyramid restatement.  Returns (pyr_img u8 [total], pyr_deriv i16 [total, 2])."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    _, _, off = klt_layout(w, h, win, max_level)
    pimg = np.zeros(int(off[-1]), np.uint8)
    pder = np.zeros((int(off[-1]), 2), np.int16)
    L = lib()
    L.gfso_klt_build_pyramid.restype = C.c_int
    L.gfso_klt_build_pyramid.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 2
    L.gfso_klt_build_pyramid(_p(img), w, h, w, win, max_level, _p(pimg), _p(pder))
    return pimg, pder


def klt_track(prev_pyr, next_pyr, width, height, win, prev_pts, next_pts=None, max_level=3, pyr_max_level=3, max_iter=30, eps=0.01,
              flags=0, min_eig_thr=1e-4):
    """cv::calcOpticalFlowPyrLK restatement on pyramids from klt_build_pyramid.  Returns (next_pts [n, 2], status [n], err [n])."""
    prev_pts = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
    n = len(prev_pts)
    nxt = np.zeros((max(n, 1), 2), np.float32)
    if next_pts is not None and n:
        nxt[:n] = np.asarray(next_pts, np.float32).reshape(-1, 2)
    st = np.zeros(max(n, 1), np.uint8)
    er = np.zeros(max(n, 1), np.float32)
    L = lib()
    L.gfso_klt_track.restype = C.c_int
    L.gfso_klt_track.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p] * 4 + [C.c_int, C.c_double, C.c_int, C.c_double]
    rc = L.gfso_klt_track(_p(prev_pyr[0]), _p(prev_pyr[1]), _p(next_pyr[0]), width, height, win, pyr_max_level, max_level, n,
                          _p(prev_pts), _p(nxt), _p(st), _p(er), max_iter, float(eps), flags, float(min_eig_thr))
    if rc != 0:
        raise RuntimeError("gfso_klt_track rc=%d" % rc)
    return nxt[:n].copy(), st[:n].copy(), er[:n].copy()


def fb_klt_tracking(prev_pyr, cur_pyr, width, height, win, nbpyrlvl, ferr, fmax_fbklt_dist, kps, priors, pyr_max_level=3):
    """ORBmatcher::fbKltTracking restatement.  Returns (priors_out [n, 2], kpstatus bool [n], n_good)."""
    kps = np.ascontiguousarray(kps, np.float32).reshape(-1, 2)
    n = len(kps)
    pri = np.zeros((max(n, 1), 2), np.float32)
    if n:
        pri[:n] = np.asarray(priors, np.float32).reshape(-1, 2)
    st = np.zeros(max(n, 1), np.uint8)
    L = lib()
    L.gfso_fb_klt_tracking.restype = C.c_int
    L.gfso_fb_klt_tracking.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float, C.c_float, C.c_int] + [C.c_void_p] * 3
    good = L.gfso_fb_klt_tracking(_p(prev_pyr[0]), _p(prev_pyr[1]), _p(cur_pyr[0]), _p(cur_pyr[1]), width, height, win,
                                  pyr_max_level, nbpyrlvl, ferr, fmax_fbklt_dist, n, _p(kps), _p(pri), _p(st))
    return pri[:n].copy(), st[:n].astype(bool), int(good)


def fmat_set_solver(mode):
    """0 Gauss-Jordan null space + bisection (default), 1 OpenCV's Jacobi SVD + cv::solveCubic (oracle/fmat_oracle.cpp header)."""
    lib().gfso_fmat_set_solver(int(mode))


def fundamental_ransac(pts1, pts2, threshold=3.0, confidence=0.99, max_iters=1000):
    """cv::findFundamentalMat(pts1, pts2, cv::FM_RANSAC, threshold, confidence, mask) restatement (oracle/fmat_oracle.cpp): RANSAC for n >= 15, LMedS for 8 <= n < 15 like the cv:: wrapper.
    Returns (mask bool [n], F [3, 3] or None, n_inliers, iterations_run)."""
    p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    n = len(p1)
    mask = np.zeros(max(n, 1), np.uint8)
    F = np.zeros(9)
    it = C.c_int(0)
    L = lib()
    L.gfso_fundamental_ransac.restype = C.c_int
    L.gfso_fundamental_ransac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_int)]
    rc = L.gfso_fundamental_ransac(_p(p1), _p(p2), n, threshold, confidence, max_iters, _p(mask), _p(F), C.byref(it))
    if rc == -2:
        raise ValueError("fewer than 8 points")
    return mask[:n].astype(bool), (F.reshape(3, 3).copy() if rc > 0 and F.any() else None), int(rc), int(it.value)
