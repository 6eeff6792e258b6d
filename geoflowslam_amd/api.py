"""Python mirror of the reference's four C++ seams on top of the C ABI (include/gfs_abi.h, libgfs_hip.so).

The product is the shared library; this module is the thin host-side binding used by tests and bench.py.
It mirrors the reference interfaces by name and argument meaning:
    ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)(image, lapping) -> (monoIndex, keypoints, descriptors)
        reference include/ORBextractor.h:53-64
    ORBmatcher.DescriptorDistance(a, b) / ORBmatcher.match(d1, d2)     reference include/ORBmatcher.h:41, src/ORBmatcher.cc:755-756
    RegistrationGICP.RegisterPointClouds(target, source, init_T)       reference include/RegistrationGICP.h:25-28
    Optimizer.LocalBundleAdjustment(problem)                           reference include/Optimizer.h:62-65
There is no CPU fallback here: if the library is missing, or no gfx950 device is present, calls raise GfsError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgfs_hip.so")
_lib = None

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"),
     ("class_id", "<i4")]
)


class GfsError(RuntimeError):
    pass


class OrbConfig(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("max_rows", C.c_int32),
                ("max_cols", C.c_int32), ("max_batch", C.c_int32), ("device", C.c_int32),
                ("blur_taps_variant", C.c_int32)]


class GicpConfig(C.Structure):
    _fields_ = [("num_threads", C.c_int32), ("downsampling_resolution", C.c_double),
                ("max_correspondence_distance", C.c_double), ("rotation_eps", C.c_double),
                ("translation_eps", C.c_double), ("max_iterations", C.c_int32), ("num_neighbors", C.c_int32)]


class GicpResult(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("converged", C.c_int32), ("iterations", C.c_uint64),
                ("num_inliers", C.c_uint64), ("H", C.c_double * 36), ("b", C.c_double * 6), ("error", C.c_double),
                ("n_target_ds", C.c_int32), ("n_source_ds", C.c_int32), ("n_linearize", C.c_int32),
                ("n_error_evals", C.c_int32)]


class LbaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32), ("pose_q", C.c_void_p),
                ("pose_t", C.c_void_p), ("pose_fixed", C.c_void_p), ("points", C.c_void_p), ("edge_pose", C.c_void_p),
                ("edge_point", C.c_void_p), ("edge_obs", C.c_void_p), ("edge_inv_sigma2", C.c_void_p),
                ("edge_stereo", C.c_void_p), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("bf", C.c_double), ("huber_mono", C.c_double), ("huber_stereo", C.c_double),
                ("iterations", C.c_int32)]


class LbaSolution(C.Structure):
    _fields_ = [("pose_q", C.c_void_p), ("pose_t", C.c_void_p), ("points", C.c_void_p), ("edge_chi2", C.c_void_p),
                ("edge_depth_positive", C.c_void_p), ("iterations_run", C.c_int32), ("final_chi2", C.c_double),
                ("final_lambda", C.c_double)]


class PoseProblem(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3), ("n_obs", C.c_int32), ("xw", C.c_void_p), ("obs", C.c_void_p),
                ("inv_sigma2", C.c_void_p), ("stereo", C.c_void_p), ("fx", C.c_double), ("fy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double), ("n_rounds", C.c_int32), ("its", C.c_int32)]


class PoseSolution(C.Structure):
    _fields_ = [("outlier", C.c_void_p), ("chi2", C.c_void_p), ("q", C.c_double * 4), ("t", C.c_double * 3),
                ("avg_reproj_error", C.c_float), ("n_inliers", C.c_int32), ("rounds_run", C.c_int32),
                ("iterations_run", C.c_int32)]


def pose_structs(prob):
    """ctypes views of one PoseOptimization problem dict (shared with oracle/oracle.py: same struct layout)."""
    P, S = PoseProblem(), PoseSolution()
    n = int(prob["n_obs"]) if "n_obs" in prob else len(prob["xw"])
    keep = dict(xw=np.ascontiguousarray(prob["xw"], np.float64).reshape(-1, 3), obs=np.ascontiguousarray(prob["obs"], np.float64).reshape(-1, 3),
                inv_sigma2=np.ascontiguousarray(prob["inv_sigma2"], np.float32), stereo=np.ascontiguousarray(prob["stereo"], np.uint8),
                outlier=np.zeros(max(n, 1), np.uint8), chi2=np.zeros(max(n, 1), np.float64))
    P.q[:] = [float(v) for v in prob["q"]]
    P.t[:] = [float(v) for v in prob["t"]]
    P.n_obs = n
    for name in ("xw", "obs", "inv_sigma2", "stereo"):
        setattr(P, name, keep[name].ctypes.data)
    for name in ("fx", "fy", "cx", "cy", "bf"):
        setattr(P, name, float(prob[name]))
    P.n_rounds = int(prob.get("n_rounds", 4))
    P.its = int(prob.get("its", 10))
    S.outlier = keep["outlier"].ctypes.data
    S.chi2 = keep["chi2"].ctypes.data
    return P, S, keep, n


def pose_result(S, keep, n):
    return dict(outlier=keep["outlier"][:n].astype(bool), chi2=keep["chi2"][:n].copy(), q=np.array(S.q[:]), t=np.array(S.t[:]),
                avg_reproj_error=float(S.avg_reproj_error), n_inliers=int(S.n_inliers), rounds_run=int(S.rounds_run),
                iterations_run=int(S.iterations_run))


class GmsProblem(C.Structure):
    _fields_ = [("n1", C.c_int32), ("n2", C.c_int32), ("kp1", C.c_void_p), ("kp2", C.c_void_p), ("width1", C.c_int32),
                ("height1", C.c_int32), ("width2", C.c_int32), ("height2", C.c_int32), ("n_matches", C.c_int32),
                ("query_idx", C.c_void_p), ("train_idx", C.c_void_p)]


class SbpProblem(C.Structure):
    _fields_ = [("n_last", C.c_int32), ("last_xw", C.c_void_p), ("last_desc", C.c_void_p), ("last_octave", C.c_void_p),
                ("last_angle", C.c_void_p), ("last_mp_has_obs", C.c_void_p), ("n_cur", C.c_int32), ("cur_kps_un", C.c_void_p),
                ("cur_u_right", C.c_void_p), ("cur_desc", C.c_void_p), ("cur_has_mp_obs", C.c_void_p),
                ("Tcw_q", C.c_float * 4), ("Tcw_t", C.c_float * 3), ("Tlw_q", C.c_float * 4), ("Tlw_t", C.c_float * 3),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float), ("b", C.c_float),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("grid_w_inv", C.c_float), ("grid_h_inv", C.c_float), ("scale_factors", C.c_void_p), ("n_levels", C.c_int32),
                ("th", C.c_float), ("mono", C.c_int32), ("check_orientation", C.c_int32)]


class SbpMapProblem(C.Structure):
    _fields_ = [("n_mp", C.c_int32), ("mp_proj", C.c_void_p), ("mp_level", C.c_void_p), ("mp_view_cos", C.c_void_p),
                ("mp_desc", C.c_void_p), ("mp_has_obs", C.c_void_p), ("n_cur", C.c_int32), ("cur_kps_un", C.c_void_p),
                ("cur_u_right", C.c_void_p), ("cur_desc", C.c_void_p), ("cur_has_mp_obs", C.c_void_p), ("min_x", C.c_float),
                ("min_y", C.c_float), ("grid_w_inv", C.c_float), ("grid_h_inv", C.c_float), ("scale_factors", C.c_void_p),
                ("n_levels", C.c_int32), ("th", C.c_float), ("nn_ratio", C.c_float)]


def sbp_map_struct(prob):
    P = SbpMapProblem()
    keep = dict(mp_proj=np.ascontiguousarray(prob["mp_proj"], np.float32).reshape(-1, 3),
                mp_level=np.ascontiguousarray(prob["mp_level"], np.int32),
                mp_view_cos=np.ascontiguousarray(prob["mp_view_cos"], np.float32),
                mp_desc=np.ascontiguousarray(prob["mp_desc"], np.uint8).reshape(-1, 32),
                mp_has_obs=np.ascontiguousarray(prob["mp_has_obs"], np.uint8),
                cur_kps_un=np.ascontiguousarray(prob["cur_kps_un"], KP_DTYPE),
                cur_u_right=np.ascontiguousarray(prob["cur_u_right"], np.float32),
                cur_desc=np.ascontiguousarray(prob["cur_desc"], np.uint8).reshape(-1, 32),
                cur_has_mp_obs=np.ascontiguousarray(prob["cur_has_mp_obs"], np.uint8),
                scale_factors=np.ascontiguousarray(prob["scale_factors"], np.float32))
    P.n_mp, P.n_cur = len(keep["mp_proj"]), len(keep["cur_kps_un"])
    for name, a in keep.items():
        setattr(P, name, a.ctypes.data)
    for name in ("min_x", "min_y", "grid_w_inv", "grid_h_inv", "th", "nn_ratio"):
        setattr(P, name, float(np.float32(prob[name])))
    P.n_levels = len(keep["scale_factors"])
    return P, keep


def sbp_struct(prob):
    """ctypes view of one SearchByProjection problem dict (keys as in gfs_sbp_problem; cur_kps_un = KP_DTYPE array)."""
    P = SbpProblem()
    keep = dict(last_xw=np.ascontiguousarray(prob["last_xw"], np.float32).reshape(-1, 3),
                last_desc=np.ascontiguousarray(prob["last_desc"], np.uint8).reshape(-1, 32),
                last_octave=np.ascontiguousarray(prob["last_octave"], np.int32),
                last_angle=np.ascontiguousarray(prob["last_angle"], np.float32),
                last_mp_has_obs=np.ascontiguousarray(prob["last_mp_has_obs"], np.uint8),
                cur_kps_un=np.ascontiguousarray(prob["cur_kps_un"], KP_DTYPE),
                cur_u_right=np.ascontiguousarray(prob["cur_u_right"], np.float32),
                cur_desc=np.ascontiguousarray(prob["cur_desc"], np.uint8).reshape(-1, 32),
                cur_has_mp_obs=np.ascontiguousarray(prob["cur_has_mp_obs"], np.uint8),
                scale_factors=np.ascontiguousarray(prob["scale_factors"], np.float32))
    P.n_last, P.n_cur = len(keep["last_xw"]), len(keep["cur_kps_un"])
    for name, a in keep.items():
        setattr(P, name, a.ctypes.data)
    for name in ("Tcw_q", "Tcw_t", "Tlw_q", "Tlw_t"):
        getattr(P, name)[:] = [float(np.float32(v)) for v in prob[name]]
    for name in ("fx", "fy", "cx", "cy", "bf", "b", "min_x", "max_x", "min_y", "max_y", "grid_w_inv", "grid_h_inv", "th"):
        setattr(P, name, float(np.float32(prob[name])))
    P.n_levels = len(keep["scale_factors"])
    P.mono = int(prob.get("mono", 0))
    P.check_orientation = int(prob.get("check_orientation", 1))
    return P, keep


# every symbol include/gfs_abi.h declares (tests/test_abi.py checks the built library exports all of them)
ABI_SYMBOLS = [
    "gfs_abi_version", "gfs_last_error", "gfs_device_count",
    "gfs_orb_default_config", "gfs_orb_create", "gfs_orb_destroy", "gfs_orb_get_tables", "gfs_orb_max_keypoints",
    "gfs_orb_extract", "gfs_orb_extract_batch", "gfs_orb_extract_batch_device", "gfs_orb_device_results",
    "gfs_orb_fetch", "gfs_orb_level_size", "gfs_orb_fetch_level", "gfs_orb_fetch_candidates", "gfs_orb_octree_host",
    "gfs_orb_octree_device", "gfs_test_sort_replica", "gfs_test_heap_sort_replica", "gfs_test_glibc_math", "gfs_test_traffic",
    "gfs_test_orb_blur_tiles",
    "gfs_hamming256", "gfs_matcher_create", "gfs_matcher_destroy", "gfs_bf_match_hamming",
    "gfs_bf_match_hamming_batch_device",
    "gfs_gicp_default_config", "gfs_gicp_create", "gfs_gicp_destroy", "gfs_gicp_align", "gfs_gicp_align_batch_device",
    "gfs_gicp_fetch_preprocessed", "gfs_gicp_tile_stats", "gfs_gicp_knn_stats", "gfs_gicp_coop_stats", "gfs_frame_rgbd", "gfs_gicp_align_next", "gfs_gicp_align_next_batch_device", "gfs_test_voxel_sort", "gfs_test_wave_std_sort",
    "gfs_lba_create", "gfs_lba_destroy", "gfs_lba_solve", "gfs_lba_solve_bool", "gfs_lba_linearize", "gfs_lba_batch_create", "gfs_lba_batch_destroy",
    "gfs_lba_solve_batch",
    "gfs_frame_create", "gfs_frame_destroy", "gfs_depth_to_cloud", "gfs_depth_to_cloud_batch_device", "gfs_depth_convert_u16_batch_device", "gfs_stereo_from_rgbd",
    "gfs_stereo_from_rgbd_batch_device",
    "gfs_pose_create", "gfs_pose_destroy", "gfs_pose_optimize", "gfs_pose_set_sum_order",
    "gfs_gms_create", "gfs_gms_destroy", "gfs_gms_inlier_mask", "gfs_gms_inlier_mask_batch_device",
    "gfs_sbp_create", "gfs_sbp_destroy", "gfs_search_by_projection", "gfs_search_by_projection_map",
    "gfs_klt_create", "gfs_klt_destroy", "gfs_klt_layout", "gfs_klt_pyramid_create", "gfs_klt_pyramid_destroy",
    "gfs_klt_build_pyramid", "gfs_klt_build_pyramid_device", "gfs_klt_pyramid_download", "gfs_klt_track", "gfs_klt_fb_track",
    "gfs_klt_fb_track_device",
    "gfs_fmat_create", "gfs_fmat_destroy", "gfs_find_fundamental_ransac", "gfs_find_fundamental_ransac_device",
    "gfs_klt_compact_tracks_device", "gfs_klt_apply_mask_device",
    "gfs_timer_create", "gfs_timer_destroy", "gfs_timer_start", "gfs_timer_stop", "gfs_timer_elapsed_ms",
    "gfs_profile_enable", "gfs_profile_report", "gfs_profile_reset",
]


def lib():
    """Load libgfs_hip.so (built in-tree by __graft_entry__.build()). Raises GfsError if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise GfsError(f"{_LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
        L = C.CDLL(_LIB_PATH)
        L.gfs_last_error.restype = C.c_char_p
        vp, ip, i = C.c_void_p, C.POINTER(C.c_int), C.c_int
        L.gfs_orb_default_config.argtypes = [C.POINTER(OrbConfig)]
        L.gfs_orb_create.argtypes = [C.POINTER(OrbConfig), C.POINTER(vp)]
        L.gfs_orb_destroy.argtypes = [vp]
        L.gfs_orb_get_tables.argtypes = [vp] * 7
        L.gfs_orb_max_keypoints.argtypes = [vp]
        L.gfs_orb_extract.argtypes = [vp, vp, i, i, i, i, i, vp, vp, i, ip]
        L.gfs_orb_extract_batch.argtypes = [vp, vp, i, i, i, i, i, i, vp, vp, i, vp, vp]
        L.gfs_orb_extract_batch_device.argtypes = [vp, vp, i, i, i, i, i, vp]
        L.gfs_orb_device_results.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), ip]
        L.gfs_orb_fetch.argtypes = [vp, i, vp, vp, i, ip, ip]
        L.gfs_orb_level_size.argtypes = [vp, i, ip, ip]
        L.gfs_orb_fetch_level.argtypes = [vp, i, i, i, vp]
        L.gfs_orb_fetch_candidates.argtypes = [vp, i, i, vp, vp, vp, i]
        L.gfs_orb_octree_host.argtypes = [vp, vp, vp, i, i, i, i, i, i, vp, i]
        L.gfs_orb_octree_device.argtypes = [i, vp, vp, vp, i, i, i, i, i, i, vp, vp, vp, i]
        L.gfs_test_glibc_math.argtypes = [i, vp, i, vp, vp, vp]
        L.gfs_test_traffic.argtypes = [i, i, C.c_longlong, C.c_longlong, i, C.POINTER(C.c_longlong)]
        L.gfs_hamming256.argtypes = [vp, vp]
        L.gfs_matcher_create.argtypes = [i, i, i, i, C.POINTER(vp)]
        L.gfs_matcher_destroy.argtypes = [vp]
        L.gfs_bf_match_hamming.argtypes = [vp, vp, i, vp, i, vp, vp]
        L.gfs_bf_match_hamming_batch_device.argtypes = [vp, vp, vp, vp, vp, i, i, vp, vp, vp]
        if hasattr(L, "gfs_gicp_create"):
            L.gfs_gicp_default_config.argtypes = [C.POINTER(GicpConfig)]
            L.gfs_gicp_create.argtypes = [i, i, i, C.POINTER(vp)]
            L.gfs_gicp_destroy.argtypes = [vp]
            L.gfs_gicp_align.argtypes = [vp, vp, i, vp, i, vp, C.POINTER(GicpConfig), C.POINTER(GicpResult)]
            L.gfs_gicp_align_batch_device.argtypes = [vp, vp, vp, vp, vp, i, i, vp, C.POINTER(GicpConfig), vp, vp]
            L.gfs_gicp_fetch_preprocessed.argtypes = [vp, i, i, vp, vp, i, ip]
            L.gfs_gicp_tile_stats.argtypes = [vp, vp, i]
            L.gfs_gicp_knn_stats.argtypes = [vp, i, i, vp, vp, i]
            L.gfs_gicp_coop_stats.argtypes = [vp, vp]
            L.gfs_test_voxel_sort.argtypes = [vp, vp, i, vp]
            L.gfs_test_wave_std_sort.argtypes = [i, vp, i, vp]
            L.gfs_gicp_align_next.argtypes = [vp, vp, i, vp, C.POINTER(GicpConfig), C.POINTER(GicpResult)]
            L.gfs_gicp_align_next_batch_device.argtypes = [vp, vp, vp, i, i, vp, C.POINTER(GicpConfig), vp, vp]
        if hasattr(L, "gfs_lba_create"):
            L.gfs_lba_create.argtypes = [i, i, i, i, C.POINTER(vp)]
            L.gfs_lba_destroy.argtypes = [vp]
            L.gfs_lba_solve.argtypes = [vp, C.POINTER(LbaProblem), C.POINTER(LbaSolution), vp]
            L.gfs_lba_solve_bool.argtypes = [vp, C.POINTER(LbaProblem), C.POINTER(LbaSolution), vp]
            L.gfs_lba_linearize.argtypes = [vp, C.POINTER(LbaProblem), vp, vp, vp, vp, vp, vp, C.POINTER(C.c_double)]
            L.gfs_lba_batch_create.argtypes = [i, i, i, i, i, C.POINTER(vp)]
            L.gfs_lba_batch_destroy.argtypes = [vp]
            L.gfs_lba_solve_batch.argtypes = [vp, vp, vp, i, vp]
        if hasattr(L, "gfs_frame_create"):
            f = C.c_float
            L.gfs_frame_create.argtypes = [i, i, i, i, C.POINTER(vp)]
            L.gfs_frame_destroy.argtypes = [vp]
            L.gfs_depth_to_cloud.argtypes = [vp, vp, i, i, i, i, f, f, f, f, vp, i, ip]
            L.gfs_depth_to_cloud_batch_device.argtypes = [vp, vp, i, i, i, i, f, f, f, f, vp, i, vp, vp]
            L.gfs_depth_convert_u16_batch_device.argtypes = [vp, vp, i, i, i, f, vp, vp]
            L.gfs_stereo_from_rgbd.argtypes = [vp, vp, vp, i, vp, i, i, i, f, vp, vp]
            L.gfs_frame_rgbd.argtypes = [vp, vp, vp, i, vp, i, i, i, f, i, f, f, f, f, vp, vp, vp, i, ip, vp, vp, ip]
            L.gfs_stereo_from_rgbd_batch_device.argtypes = [vp, vp, vp, vp, i, i, vp, i, i, f, vp, vp, vp]
        if hasattr(L, "gfs_klt_create"):
            f, d = C.c_float, C.c_double
            L.gfs_klt_create.argtypes = [i, i, i, i, i, i, i, C.POINTER(vp)]
            L.gfs_klt_destroy.argtypes = [vp]
            L.gfs_klt_layout.argtypes = [vp, vp, vp, vp]
            L.gfs_klt_pyramid_create.argtypes = [vp, C.POINTER(vp)]
            L.gfs_klt_pyramid_destroy.argtypes = [vp]
            L.gfs_klt_build_pyramid.argtypes = [vp, vp, vp, i, i]
            L.gfs_klt_build_pyramid_device.argtypes = [vp, vp, vp, i, i, vp]
            L.gfs_klt_pyramid_download.argtypes = [vp, vp, i, vp, vp]
            L.gfs_klt_track.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, vp, i, i, d, i, d]
            L.gfs_klt_fb_track.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, vp, i, f, f]
            L.gfs_klt_fb_track_device.argtypes = [vp, vp, vp, i, i, vp, vp, vp, vp, vp, i, f, f, vp]
            L.gfs_klt_compact_tracks_device.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp]
            L.gfs_klt_apply_mask_device.argtypes = [vp, i, i, vp, vp, vp, vp, vp]
        if hasattr(L, "gfs_fmat_create"):
            L.gfs_fmat_create.argtypes = [i, i, i, C.POINTER(vp)]
            L.gfs_fmat_destroy.argtypes = [vp]
            L.gfs_find_fundamental_ransac.argtypes = [vp, i, vp, vp, vp, C.c_double, C.c_double, i, vp, vp, vp]
            L.gfs_find_fundamental_ransac_device.argtypes = [vp, i, i, vp, vp, vp, C.c_double, C.c_double, i, vp, vp, vp]
        L.gfs_timer_create.argtypes = [i, C.POINTER(vp)]
        L.gfs_timer_destroy.argtypes = [vp]
        L.gfs_timer_start.argtypes = [vp, vp]
        L.gfs_timer_stop.argtypes = [vp, vp]
        L.gfs_timer_elapsed_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.gfs_profile_report.argtypes = [vp, vp, vp, i]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _check(rc, what):
    if rc < 0:
        raise GfsError(f"{what} failed ({rc}): {lib().gfs_last_error().decode()}")
    return rc


def device_count():
    return lib().gfs_device_count()


class ORBextractor:
    """ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:46-118) backed by gfs_orb_*."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, max_rows=480,
                 max_cols=640, max_batch=1, device=0, blur_taps_variant=0):
        L = lib()
        cfg = OrbConfig()
        L.gfs_orb_default_config(C.byref(cfg))
        cfg.nfeatures, cfg.scale_factor, cfg.nlevels = nfeatures, scaleFactor, nlevels
        cfg.ini_th_fast, cfg.min_th_fast = iniThFAST, minThFAST
        cfg.max_rows, cfg.max_cols, cfg.max_batch, cfg.device = max_rows, max_cols, max_batch, device
        cfg.blur_taps_variant = blur_taps_variant
        self.nlevels = nlevels
        self.h = C.c_void_p()
        _check(L.gfs_orb_create(C.byref(cfg), C.byref(self.h)), "gfs_orb_create")
        self.cap = L.gfs_orb_max_keypoints(self.h)

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gfs_orb_destroy(self.h)
        self.h = None

    __del__ = close

    def GetLevels(self):
        return self.nlevels

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        feats = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        _check(lib().gfs_orb_get_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2), _p(feats), _p(umax)), "get_tables")
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, feats=feats, umax=umax)

    def GetScaleFactors(self):
        return self.tables()["scale"]

    def GetInverseScaleFactors(self):
        return self.tables()["inv_scale"]

    def GetScaleSigmaSquares(self):
        return self.tables()["sigma2"]

    def GetInverseScaleSigmaSquares(self):
        return self.tables()["inv_sigma2"]

    def __call__(self, image, vLappingArea=(0, 0)):
        """operator(): -> (monoIndex or -1, keypoints[KP_DTYPE], descriptors [N,32] u8)"""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (src/ORBextractor.cc:1153)"
        stride = image.strides[0]
        assert image.strides[1] == 1
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int(0)
        r = lib().gfs_orb_extract(self.h, C.c_void_p(image.ctypes.data), image.shape[0], image.shape[1], stride,
                                  vLappingArea[0], vLappingArea[1], _p(kps), _p(desc), self.cap, C.byref(n))
        if r < -1:
            raise GfsError(f"gfs_orb_extract failed ({r + 100}): {lib().gfs_last_error().decode()}")
        return r, kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images, vLappingArea=(0, 0)):
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        B = len(imgs)
        rows, cols = imgs[0].shape
        ptrs = (C.c_void_p * B)(*[im.ctypes.data for im in imgs])
        kps = np.zeros((B, self.cap), KP_DTYPE)
        desc = np.zeros((B, self.cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        mono = np.zeros(B, np.int32)
        _check(lib().gfs_orb_extract_batch(self.h, ptrs, B, rows, cols, cols, vLappingArea[0], vLappingArea[1], _p(kps),
                                           _p(desc), self.cap, _p(n), _p(mono)), "gfs_orb_extract_batch")
        return [(int(mono[b]), kps[b, :n[b]].copy(), desc[b, :n[b]].copy()) for b in range(B)]

    def extract_batch_device(self, dev_ptr, B, rows, cols, vLappingArea=(0, 0), stream=None):
        _check(lib().gfs_orb_extract_batch_device(self.h, C.c_void_p(dev_ptr), B, rows, cols, vLappingArea[0],
                                                  vLappingArea[1], C.c_void_p(stream) if stream else None),
               "gfs_orb_extract_batch_device")

    def device_results(self):
        k, d, c, m = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        cap = C.c_int()
        _check(lib().gfs_orb_device_results(self.h, C.byref(k), C.byref(d), C.byref(c), C.byref(m), C.byref(cap)),
               "gfs_orb_device_results")
        return dict(kps=k.value, desc=d.value, counts=c.value, mono=m.value, cap=cap.value)

    def fetch(self, b):
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n, mono = C.c_int(), C.c_int()
        _check(lib().gfs_orb_fetch(self.h, b, _p(kps), _p(desc), self.cap, C.byref(n), C.byref(mono)), "gfs_orb_fetch")
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    def level_size(self, l):
        r, c = C.c_int(), C.c_int()
        _check(lib().gfs_orb_level_size(self.h, l, C.byref(r), C.byref(c)), "gfs_orb_level_size")
        return r.value, c.value

    def level(self, l, b=0, blurred=False):
        r, c = self.level_size(l)
        a = np.zeros((r, c), np.uint8)
        _check(lib().gfs_orb_fetch_level(self.h, b, l, int(blurred), _p(a)), "gfs_orb_fetch_level")
        return a

    def candidates(self, l, b=0):
        n = _check(lib().gfs_orb_fetch_candidates(self.h, b, l, None, None, None, 0), "gfs_orb_fetch_candidates")
        x, y, s = (np.zeros(max(n, 1), np.int32) for _ in range(3))
        lib().gfs_orb_fetch_candidates(self.h, b, l, _p(x), _p(y), _p(s), n)
        return x[:n], y[:n], s[:n]


def wave_std_sort_perm(keys, device=0):
    """csrc/wave_std_sort.hpp on the GPU: the permutation std::sort leaves (test hook)."""
    k = np.ascontiguousarray(keys, np.uint32)
    perm = np.zeros(max(len(k), 1), np.uint16)
    _check(lib().gfs_test_wave_std_sort(device, _p(k), len(k), _p(perm)), "gfs_test_wave_std_sort")
    return perm[:len(k)].astype(np.int64)


def octree_host(x, y, score, min_x, max_x, min_y, max_y, n_features):
    """The library's host DistributeOctTree (no GPU needed)."""
    x = np.ascontiguousarray(x, np.int32)
    y = np.ascontiguousarray(y, np.int32)
    score = np.ascontiguousarray(score, np.int32)
    out = np.zeros(max(len(x), 1), np.int32)
    n = lib().gfs_orb_octree_host(_p(x), _p(y), _p(score), len(x), min_x, max_x, min_y, max_y, n_features, _p(out),
                                  len(out))
    return out[:n]


def octree_device(x, y, score, min_x, max_x, min_y, max_y, n_features, device=0):
    """The library's device DistributeOctTree (k_octree) -> kept (x, y, score) arrays in list order."""
    x = np.ascontiguousarray(x, np.int32)
    y = np.ascontiguousarray(y, np.int32)
    score = np.ascontiguousarray(score, np.int32)
    cap = n_features + 64
    ox, oy, os_ = (np.zeros(cap, np.int32) for _ in range(3))
    n = _check(lib().gfs_orb_octree_device(device, _p(x), _p(y), _p(score), len(x), min_x, max_x, min_y, max_y, n_features,
                                           _p(ox), _p(oy), _p(os_), cap), "gfs_orb_octree_device")
    return ox[:n], oy[:n], os_[:n]


class ORBmatcher:
    """The brute-force Hamming part of ORB_SLAM3::ORBmatcher (reference src/ORBmatcher.cc:744-778, 2536-2550)."""

    def __init__(self, max_query=4096, max_train=4096, max_batch=1, device=0):
        self.h = C.c_void_p()
        _check(lib().gfs_matcher_create(device, max_query, max_train, max_batch, C.byref(self.h)), "gfs_matcher_create")

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gfs_matcher_destroy(self.h)
        self.h = None

    __del__ = close

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return int(lib().gfs_hamming256(_p(a), _p(b)))

    def match(self, query, train):
        """cv::BFMatcher(NORM_HAMMING).match(query, train) -> (trainIdx[nq], distance[nq]); empty if no train rows."""
        q = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
        ti = np.zeros(len(q), np.int32)
        di = np.zeros(len(q), np.int32)
        n = _check(lib().gfs_bf_match_hamming(self.h, _p(q), len(q), _p(t), len(t), _p(ti), _p(di)),
                   "gfs_bf_match_hamming")
        return ti[:n], di[:n]

    def match_batch_device(self, d_query, d_nq, d_train, d_nt, B, stride_rows, d_idx, d_dist, stream=None):
        _check(lib().gfs_bf_match_hamming_batch_device(self.h, C.c_void_p(d_query), C.c_void_p(d_nq), C.c_void_p(d_train),
                                                       C.c_void_p(d_nt), B, stride_rows, C.c_void_p(d_idx),
                                                       C.c_void_p(d_dist), C.c_void_p(stream) if stream else None),
               "gfs_bf_match_hamming_batch_device")


def gicp_default_config():
    c = GicpConfig()
    lib().gfs_gicp_default_config(C.byref(c))
    return c


def _result_dict(res):
    return dict(
        T=np.array(res.T).reshape(4, 4).T.copy(), converged=bool(res.converged), iterations=int(res.iterations),
        num_inliers=int(res.num_inliers), H=np.array(res.H).reshape(6, 6).T.copy(), b=np.array(res.b),
        error=float(res.error), n_target_ds=res.n_target_ds, n_source_ds=res.n_source_ds,
        n_linearize=res.n_linearize, n_error_evals=res.n_error_evals)


class RegistrationGICP:
    """RegistrationGICP (reference include/RegistrationGICP.h:19-31, src/RegistrationGICP.cc:5-20)."""

    def __init__(self, max_points=40960, max_batch=1, device=0):
        self.h = C.c_void_p()
        self.max_points = max_points
        _check(lib().gfs_gicp_create(device, max_points, max_batch, C.byref(self.h)), "gfs_gicp_create")

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gfs_gicp_destroy(self.h)
        self.h = None

    __del__ = close

    def RegisterPointClouds(self, target_points, source_points, init_T_target_source=None, cfg=None):
        t = np.ascontiguousarray(target_points, np.float32).reshape(-1, 4)
        s = np.ascontiguousarray(source_points, np.float32).reshape(-1, 4)
        T0 = np.eye(4) if init_T_target_source is None else np.asarray(init_T_target_source, np.float64)
        T0c = np.ascontiguousarray(T0.T.reshape(-1))
        cfg = cfg or gicp_default_config()
        res = GicpResult()
        _check(lib().gfs_gicp_align(self.h, _p(t), len(t), _p(s), len(s), _p(T0c), C.byref(cfg), C.byref(res)),
               "gfs_gicp_align")
        return _result_dict(res)

    def voxel_sort_perm(self, keys):
        """Test hook: permutation of the preprocessing's voxel sort (small_gicp quick_sort_omp replica) for caller keys."""
        k = np.ascontiguousarray(keys, np.uint64)
        perm = np.zeros(max(len(k), 1), np.uint32)
        _check(lib().gfs_test_voxel_sort(self.h, _p(k), len(k), _p(perm)), "gfs_test_voxel_sort")
        return perm[:len(k)].astype(np.int64)

    def RegisterNext(self, source_points, init_T_target_source=None, cfg=None):
        """Streaming form: the target is the source cloud of the previous call on this object (kept preprocessed in HBM), as in
        Tracking::PredictStateICP; bit-identical to RegisterPointClouds(previous source, source_points)."""
        s = np.ascontiguousarray(source_points, np.float32).reshape(-1, 4)
        T0 = np.eye(4) if init_T_target_source is None else np.asarray(init_T_target_source, np.float64)
        T0c = np.ascontiguousarray(T0.T.reshape(-1))
        cfg = cfg or gicp_default_config()
        res = GicpResult()
        _check(lib().gfs_gicp_align_next(self.h, _p(s), len(s), _p(T0c), C.byref(cfg), C.byref(res)), "gfs_gicp_align_next")
        return _result_dict(res)

    def align_next_batch_device(self, d_source, d_ns, B, stride_pts, init_T=None, cfg=None, stream=None, raw=False):
        cfg = cfg or gicp_default_config()
        out = (GicpResult * B)()
        T0 = None
        if init_T is not None:
            T0 = np.ascontiguousarray(np.asarray(init_T, np.float64).transpose(0, 2, 1).reshape(B, 16))
        _check(lib().gfs_gicp_align_next_batch_device(self.h, C.c_void_p(d_source), C.c_void_p(d_ns), B, stride_pts, _p(T0),
                                                      C.byref(cfg), out, C.c_void_p(stream) if stream else None),
               "gfs_gicp_align_next_batch_device")
        return out if raw else [_result_dict(r) for r in out]

    def align_batch_device(self, d_target, d_nt, d_source, d_ns, B, stride_pts, init_T=None, cfg=None, stream=None,
                           raw=False):
        """raw=True returns the ctypes array of gfs_gicp_result (no per-pair Python conversion on the hot path)."""
        cfg = cfg or gicp_default_config()
        out = (GicpResult * B)()
        T0 = None
        if init_T is not None:
            T0 = np.ascontiguousarray(np.asarray(init_T, np.float64).transpose(0, 2, 1).reshape(B, 16))
        _check(lib().gfs_gicp_align_batch_device(self.h, C.c_void_p(d_target), C.c_void_p(d_nt), C.c_void_p(d_source),
                                                 C.c_void_p(d_ns), B, stride_pts, _p(T0), C.byref(cfg), out,
                                                 C.c_void_p(stream) if stream else None), "gfs_gicp_align_batch_device")
        if raw:
            return out
        return [_result_dict(r) for r in out]

    def tile_stats(self, reset=True):
        """Workgroups of the linearisation kernel by outcome of the LDS tile staging (index 0 = staged), see gfs_gicp_tile_stats."""
        out = np.zeros(8, np.uint64)
        _check(lib().gfs_gicp_tile_stats(self.h, _p(out), int(reset)), "gfs_gicp_tile_stats")
        return out

    def coop_stats(self):
        """dict(launches, last_workgroups, failed, budget) of the cooperative LM kernel on this handle: gfs_gicp_coop_stats."""
        out = np.zeros(4, np.int32)
        _check(lib().gfs_gicp_coop_stats(self.h, _p(out)), "gfs_gicp_coop_stats")
        return dict(launches=int(out[0]), last_workgroups=int(out[1]), failed=bool(out[2]), budget=int(out[3]))

    def knn_stats(self, b, which, cap=256):
        """(points, deferred to the r = 2 pass, deferred to the isolated-point pass), bounds of the latter: gfs_gicp_knn_stats."""
        out = np.zeros(3, np.int32)
        dk = np.zeros(cap)
        _check(lib().gfs_gicp_knn_stats(self.h, b, which, _p(out), _p(dk), cap), "gfs_gicp_knn_stats")
        return out, dk[:min(int(out[2]), cap)]

    def preprocessed(self, b, which, cap=None):
        cap = cap or self.max_points
        pts = np.zeros((cap, 4))
        covs = np.zeros((cap, 9))
        m = C.c_int()
        _check(lib().gfs_gicp_fetch_preprocessed(self.h, b, which, _p(pts), _p(covs), cap, C.byref(m)),
               "gfs_gicp_fetch_preprocessed")
        return pts[:m.value], covs[:m.value].reshape(-1, 3, 3).transpose(0, 2, 1).copy()


class Timer:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        _check(lib().gfs_timer_create(device, C.byref(self.h)), "gfs_timer_create")

    def start(self, stream=None):
        _check(lib().gfs_timer_start(self.h, C.c_void_p(stream) if stream else None), "gfs_timer_start")

    def stop(self, stream=None):
        _check(lib().gfs_timer_stop(self.h, C.c_void_p(stream) if stream else None), "gfs_timer_stop")

    def elapsed_ms(self):
        ms = C.c_float()
        _check(lib().gfs_timer_elapsed_ms(self.h, C.byref(ms)), "gfs_timer_elapsed_ms")
        return ms.value

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gfs_timer_destroy(self.h)
        self.h = None


def profile_enable(on=True):
    lib().gfs_profile_enable(int(on))


def profile_reset():
    lib().gfs_profile_reset()


def profile_report():
    cap = 64
    names = (C.c_char * 64 * cap)()
    tot = np.zeros(cap)
    cnt = np.zeros(cap, np.int64)
    n = lib().gfs_profile_report(names, _p(tot), _p(cnt), cap)
    return {names[i].value.decode(): (float(tot[i]), int(cnt[i])) for i in range(min(n, cap))}


def _lba_problem(prob):
    P = LbaProblem()
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


class Optimizer:
    """The numeric core of ORB_SLAM3::Optimizer::LocalBundleAdjustment (reference include/Optimizer.h:62-65,
    src/Optimizer.cc:1588-2040) on a flattened problem (see gfs_lba_problem in include/gfs_abi.h)."""

    def __init__(self, max_poses=64, max_points=8192, max_edges=131072, device=0):
        self.h = C.c_void_p()
        _check(lib().gfs_lba_create(device, max_poses, max_points, max_edges, C.byref(self.h)), "gfs_lba_create")

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gfs_lba_destroy(self.h)
        self.h = None

    __del__ = close

    def LocalBundleAdjustment(self, prob, stop_flag=None):
        P, keep = _lba_problem(prob)
        out = dict(pose_q=np.zeros((P.n_poses, 4)), pose_t=np.zeros((P.n_poses, 3)), points=np.zeros((P.n_points, 3)),
                   edge_chi2=np.zeros(P.n_edges), edge_depth_positive=np.zeros(P.n_edges, np.uint8))
        S = LbaSolution()
        for k, v in out.items():
            setattr(S, k, v.ctypes.data)
        stop = stop_flag.ctypes.data_as(C.c_void_p) if stop_flag is not None else None
        rc = lib().gfs_lba_solve(self.h, C.byref(P), C.byref(S), stop)
        if rc == -6:  # GFS_ERR_STOPPED: the reference returns without touching the map (src/Optimizer.cc:1955-1956)
            return None
        _check(rc, "gfs_lba_solve")
        out.update(iterations_run=S.iterations_run, final_chi2=S.final_chi2, final_lambda=S.final_lambda)
        return out

    def linearize(self, prob):
        P, keep = _lba_problem(prob)
        nf = int((np.asarray(prob["pose_fixed"]) == 0).sum())
        Hpp = np.zeros((nf, 36)); Hll = np.zeros((P.n_points, 9)); Hpl = np.zeros((P.n_edges, 18))
        bp = np.zeros((nf, 6)); bl = np.zeros((P.n_points, 3)); chi = np.zeros(P.n_edges)
        tot = C.c_double()
        _check(lib().gfs_lba_linearize(self.h, C.byref(P), _p(Hpp), _p(Hll), _p(Hpl), _p(bp), _p(bl), _p(chi),
                                       C.byref(tot)), "gfs_lba_linearize")
        return dict(Hpp=Hpp.reshape(nf, 6, 6).transpose(0, 2, 1).copy(), Hll=Hll.reshape(-1, 3, 3).transpose(0, 2, 1).copy(),
                    Hpl=Hpl.reshape(-1, 3, 6).transpose(0, 2, 1).copy(), bp=bp, bl=bl, edge_chi2=chi, chi2=tot.value)


class BatchOptimizer:
    """n independent LocalBundleAdjustment windows solved together (gfs_lba_solve_batch)."""

    def __init__(self, max_windows=64, max_poses=32, max_points=4096, max_edges=65536, device=0):
        self.h = C.c_void_p()
        _check(lib().gfs_lba_batch_create(device, max_windows, max_poses, max_points, max_edges, C.byref(self.h)), "gfs_lba_batch_create")

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gfs_lba_batch_destroy(self.h)
        self.h = None

    __del__ = close

    def prepare(self, probs):
        """ctypes views (kept alive by the returned object) so that repeated solves pay no Python conversion"""
        n = len(probs)
        P = (LbaProblem * n)()
        S = (LbaSolution * n)()
        keep, outs = [], []
        for k, prob in enumerate(probs):
            pk, kp = _lba_problem(prob)
            P[k] = pk
            keep.append(kp)
            out = dict(pose_q=np.zeros((pk.n_poses, 4)), pose_t=np.zeros((pk.n_poses, 3)), points=np.zeros((pk.n_points, 3)),
                       edge_chi2=np.zeros(pk.n_edges), edge_depth_positive=np.zeros(pk.n_edges, np.uint8))
            for name, v in out.items():
                setattr(S[k], name, v.ctypes.data)
            outs.append(out)
        keep.append(outs)  # S points into these arrays: whoever holds `keep` keeps them alive
        return P, S, outs, keep

    def solve_prepared(self, P, S, n, stop_flag=None):
        stop = stop_flag.ctypes.data_as(C.c_void_p) if stop_flag is not None else None
        _check(lib().gfs_lba_solve_batch(self.h, P, S, n, stop), "gfs_lba_solve_batch")

    def LocalBundleAdjustment(self, probs, stop_flag=None):
        P, S, outs, keep = self.prepare(probs)
        self.solve_prepared(P, S, len(probs), stop_flag)
        for k, out in enumerate(outs):
            out.update(iterations_run=S[k].iterations_run, final_chi2=S[k].final_chi2, final_lambda=S[k].final_lambda)
        return outs


class GmsMatcher:
    """gms_matcher(kp1, size1, kp2, size2, matches).GetInlierMask(mask, false, false) (reference
    Thirdparty/GMS/include/gms_matcher.h; call sites src/ORBmatcher.cc:761-762, 812-813, 893-894)."""

    def __init__(self, max_keypoints=4096, max_batch=64, device=0):
        self.h = C.c_void_p()
        _check(lib().gfs_gms_create(device, max_keypoints, max_batch, C.byref(self.h)), "gfs_gms_create")

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gfs_gms_destroy(self.h)
        self.h = None

    __del__ = close

    def GetInlierMask(self, kp1, size1, kp2, size2, query_idx, train_idx):
        """size = (width, height) like cv::Size -> (mask bool [n_matches], n_inliers)"""
        kp1 = np.ascontiguousarray(kp1, KP_DTYPE)
        kp2 = np.ascontiguousarray(kp2, KP_DTYPE)
        q = np.ascontiguousarray(query_idx, np.int32)
        t = np.ascontiguousarray(train_idx, np.int32)
        P = GmsProblem(len(kp1), len(kp2), kp1.ctypes.data, kp2.ctypes.data, int(size1[0]), int(size1[1]), int(size2[0]),
                       int(size2[1]), len(q), q.ctypes.data, t.ctypes.data)
        mask = np.zeros(max(len(q), 1), np.uint8)
        ptrs = (C.c_void_p * 1)(mask.ctypes.data)
        n = np.zeros(1, np.int32)
        _check(lib().gfs_gms_inlier_mask(self.h, C.byref(P), 1, ptrs, _p(n)), "gfs_gms_inlier_mask")
        return mask[:len(q)].astype(bool), int(n[0])

    def inlier_mask_batch_device(self, d_kps1, d_n1, d_kps2, d_n2, B, kp_stride, d_train_idx, width, height, d_mask, d_counts,
                                 stream=None):
        _check(lib().gfs_gms_inlier_mask_batch_device(self.h, C.c_void_p(d_kps1), C.c_void_p(d_n1), C.c_void_p(d_kps2),
                                                      C.c_void_p(d_n2), B, kp_stride, C.c_void_p(d_train_idx), width, height,
                                                      C.c_void_p(d_mask), C.c_void_p(d_counts),
                                                      C.c_void_p(stream) if stream else None), "gfs_gms_inlier_mask_batch_device")


KLT_USE_INITIAL_FLOW, KLT_GET_MIN_EIGENVALS = 4, 8


class KltPyramid:
    """The optical-flow pyramids (cv::buildOpticalFlowPyramid output, images + derivatives) of a batch of frames, resident in HBM."""

    def __init__(self, tracker):
        self.tracker = tracker
        self.h = C.c_void_p()
        _check(lib().gfs_klt_pyramid_create(tracker.h, C.byref(self.h)), "gfs_klt_pyramid_create")

    def close(self):
        if getattr(self, "h", None) and _lib is not None and getattr(self.tracker, "h", None):
            _lib.gfs_klt_pyramid_destroy(self.h)
        self.h = None

    __del__ = close

    def download(self, f=0):
        """-> (img u8 [total], deriv i16 [total, 2]) of frame f in the layout of KltTracker.layout()."""
        total = int(self.tracker.layout()[2][-1])
        img = np.zeros(total, np.uint8)
        der = np.zeros((total, 2), np.int16)
        _check(lib().gfs_klt_pyramid_download(self.tracker.h, self.h, f, _p(img), _p(der)), "gfs_klt_pyramid_download")
        return img, der


class KltTracker:
    """The optical-flow front end of the reference: cv::buildOpticalFlowPyramid (src/Frame.cc:373), cv::calcOpticalFlowPyrLK and
    ORBmatcher::fbKltTracking (src/ORBmatcher.cc:2186-2297).  win = LKWindowSize, max_level = 3 as in Frame.cc:371."""

    def __init__(self, width, height, win, max_level=3, max_batch=1, max_points=4096, device=0):
        self.width, self.height, self.win = width, height, win
        self.h = C.c_void_p()
        _check(lib().gfs_klt_create(device, width, height, win, max_level, max_batch, max_points, C.byref(self.h)), "gfs_klt_create")

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gfs_klt_destroy(self.h)
        self.h = None

    __del__ = close

    def layout(self):
        lw, lh, off = np.zeros(8, np.int32), np.zeros(8, np.int32), np.zeros(9, np.int64)
        n = lib().gfs_klt_layout(self.h, _p(lw), _p(lh), _p(off))
        if n <= 0:
            raise GfsError("gfs_klt_layout failed")
        return lw[:n].copy(), lh[:n].copy(), off[:n + 1].copy()

    def buildOpticalFlowPyramid(self, images, pyramid=None):
        """images: one [H, W] u8 array or a list of them -> KltPyramid (reused when given)."""
        if isinstance(images, np.ndarray) and images.ndim == 2:
            images = [images]
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        for im in imgs:
            if im.shape != (self.height, self.width):
                raise GfsError(f"image shape {im.shape} != {(self.height, self.width)}")
        pyr = pyramid if pyramid is not None else KltPyramid(self)
        ptrs = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
        _check(lib().gfs_klt_build_pyramid(self.h, pyr.h, ptrs, self.width, len(imgs)), "gfs_klt_build_pyramid")
        return pyr

    def build_pyramid_device(self, d_images, stride, B, pyramid, stream=None):
        _check(lib().gfs_klt_build_pyramid_device(self.h, pyramid.h, C.c_void_p(d_images), stride, B,
                                                  C.c_void_p(stream) if stream else None), "gfs_klt_build_pyramid_device")

    @staticmethod
    def _lists(pts):
        if isinstance(pts, np.ndarray):
            pts = [pts]
        return [np.ascontiguousarray(p, np.float32).reshape(-1, 2) for p in pts]

    def calcOpticalFlowPyrLK(self, prev, nxt, prev_pts, next_pts=None, max_level=3, max_iter=30, eps=0.01, flags=0, min_eig_thr=1e-4):
        """Batched cv::calcOpticalFlowPyrLK: prev_pts = [n, 2] array or one per pair -> list of (next_pts, status, err) (a single
        tuple when a single array was given)."""
        single = isinstance(prev_pts, np.ndarray)
        P = self._lists(prev_pts)
        B = len(P)
        N = [np.zeros((max(len(p), 1), 2), np.float32) for p in P]
        if next_pts is not None:
            for dst, src in zip(N, self._lists(next_pts)):
                dst[:len(src)] = src
        S = [np.zeros(max(len(p), 1), np.uint8) for p in P]
        E = [np.zeros(max(len(p), 1), np.float32) for p in P]
        n = np.array([len(p) for p in P], np.int32)
        arr = lambda L: (C.c_void_p * B)(*[a.ctypes.data for a in L])
        _check(lib().gfs_klt_track(self.h, prev.h, nxt.h, B, _p(n), arr(P), arr(N), arr(S), arr(E), max_level, max_iter, float(eps),
                                   flags, float(min_eig_thr)), "gfs_klt_track")
        out = [(N[b][:n[b]].copy(), S[b][:n[b]].copy(), E[b][:n[b]].copy()) for b in range(B)]
        return out[0] if single else out

    def fbKltTracking(self, prev, cur, nbpyrlvl, ferr, fmax_fbklt_dist, kps, priors):
        """Batched ORBmatcher::fbKltTracking -> list of (priors_out, kpstatus bool, n_good) (a single tuple for a single array)."""
        single = isinstance(kps, np.ndarray)
        K = self._lists(kps)
        B = len(K)
        Pr = [np.zeros((max(len(k), 1), 2), np.float32) for k in K]
        for dst, src in zip(Pr, self._lists(priors)):
            dst[:len(src)] = src
        S = [np.zeros(max(len(k), 1), np.uint8) for k in K]
        n = np.array([len(k) for k in K], np.int32)
        good = np.zeros(B, np.int32)
        arr = lambda L: (C.c_void_p * B)(*[a.ctypes.data for a in L])
        _check(lib().gfs_klt_fb_track(self.h, prev.h, cur.h, B, _p(n), arr(K), arr(Pr), arr(S), _p(good), nbpyrlvl, ferr,
                                      fmax_fbklt_dist), "gfs_klt_fb_track")
        out = [(Pr[b][:n[b]].copy(), S[b][:n[b]].astype(bool), int(good[b])) for b in range(B)]
        return out[0] if single else out

    def compact_tracks_device(self, B, pt_stride, d_n, d_kps, d_priors, d_kpstatus, d_a, d_b, d_index, d_m, stream=None):
        _check(lib().gfs_klt_compact_tracks_device(self.h, B, pt_stride, C.c_void_p(d_n), C.c_void_p(d_kps), C.c_void_p(d_priors),
                                                   C.c_void_p(d_kpstatus), C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_index),
                                                   C.c_void_p(d_m), C.c_void_p(stream) if stream else None),
               "gfs_klt_compact_tracks_device")

    def apply_mask_device(self, B, pt_stride, d_m, d_index, d_mask, d_kpstatus, stream=None):
        _check(lib().gfs_klt_apply_mask_device(self.h, B, pt_stride, C.c_void_p(d_m), C.c_void_p(d_index), C.c_void_p(d_mask),
                                               C.c_void_p(d_kpstatus), C.c_void_p(stream) if stream else None),
               "gfs_klt_apply_mask_device")

    def fb_track_device(self, prev, cur, B, pt_stride, d_n, d_kps, d_priors, d_kpstatus, d_n_good, nbpyrlvl=3, ferr=15.0,
                        fmax_fbklt_dist=0.5, stream=None):
        _check(lib().gfs_klt_fb_track_device(self.h, prev.h, cur.h, B, pt_stride, C.c_void_p(d_n), C.c_void_p(d_kps),
                                             C.c_void_p(d_priors), C.c_void_p(d_kpstatus), C.c_void_p(d_n_good), nbpyrlvl, ferr,
                                             fmax_fbklt_dist, C.c_void_p(stream) if stream else None), "gfs_klt_fb_track_device")


class FundamentalMatcher:
    """cv::findFundamentalMat(pts1, pts2, cv::FM_RANSAC, threshold, confidence, mask) (reference call sites src/ORBmatcher.cc:236,
    2399, 2463; src/Tracking.cc:1974): RANSAC for 15 or more points, LMedS for 8 .. 14 like the cv:: wrapper."""

    def __init__(self, max_points=4096, max_batch=1, device=0):
        self.h = C.c_void_p()
        _check(lib().gfs_fmat_create(device, max_points, max_batch, C.byref(self.h)), "gfs_fmat_create")

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gfs_fmat_destroy(self.h)
        self.h = None

    __del__ = close

    def findFundamentalMat(self, pts1, pts2, threshold=3.0, confidence=0.99, max_iters=1000):
        """pts1 / pts2: [n, 2] arrays or lists of them -> (mask bool [n], F [3, 3] or None, n_inliers) per problem."""
        single = isinstance(pts1, np.ndarray)
        P1 = [np.ascontiguousarray(p, np.float32).reshape(-1, 2) for p in ([pts1] if single else pts1)]
        P2 = [np.ascontiguousarray(p, np.float32).reshape(-1, 2) for p in ([pts2] if single else pts2)]
        B = len(P1)
        n = np.array([len(p) for p in P1], np.int32)
        M = [np.zeros(max(len(p), 1), np.uint8) for p in P1]
        F = np.zeros((B, 9))
        cnt = np.zeros(B, np.int32)
        arr = lambda L: (C.c_void_p * B)(*[a.ctypes.data for a in L])
        _check(lib().gfs_find_fundamental_ransac(self.h, B, _p(n), arr(P1), arr(P2), float(threshold), float(confidence), max_iters,
                                                 arr(M), _p(F), _p(cnt)), "gfs_find_fundamental_ransac")
        out = [(M[b][:n[b]].astype(bool), F[b].reshape(3, 3).copy() if cnt[b] > 0 and F[b].any() else None, int(cnt[b])) for b in range(B)]
        return out[0] if single else out

    def find_device(self, B, stride, d_n, d_pts1, d_pts2, d_mask, threshold=3.0, confidence=0.99, max_iters=1000):
        """Device-resident batch -> (F [B, 3, 3], n_inliers [B]); the masks are written to d_mask [B][stride]."""
        F = np.zeros((B, 9))
        cnt = np.zeros(B, np.int32)
        _check(lib().gfs_find_fundamental_ransac_device(self.h, B, stride, C.c_void_p(d_n), C.c_void_p(d_pts1), C.c_void_p(d_pts2),
                                                        float(threshold), float(confidence), max_iters, C.c_void_p(d_mask), _p(F),
                                                        _p(cnt)), "gfs_find_fundamental_ransac_device")
        return F.reshape(B, 3, 3), cnt


class ProjectionMatcher:
    """ORB_SLAM3::ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (reference include/ORBmatcher.h,
    src/ORBmatcher.cc:1853-2063) on flattened single-camera frame pairs (gfs_sbp_problem in include/gfs_abi.h)."""

    def __init__(self, max_last=2048, max_cur=2048, max_batch=64, device=0):
        self.h = C.c_void_p()
        _check(lib().gfs_sbp_create(device, max_last, max_cur, max_batch, C.byref(self.h)), "gfs_sbp_create")

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gfs_sbp_destroy(self.h)
        self.h = None

    __del__ = close

    def SearchByProjection(self, pairs):
        """pairs: one problem dict or a list -> (cur_match int32 [n_cur], nmatches) per pair."""
        single = isinstance(pairs, dict)
        probs = [pairs] if single else list(pairs)
        B = len(probs)
        PP = (SbpProblem * B)()
        keeps, outs = [], []
        ptrs = (C.c_void_p * B)()
        for f, prob in enumerate(probs):
            P, keep = sbp_struct(prob)
            PP[f] = P
            keeps.append(keep)
            outs.append(np.full(max(P.n_cur, 1), -9, np.int32))
            ptrs[f] = outs[f].ctypes.data
        nm = np.zeros(B, np.int32)
        _check(lib().gfs_search_by_projection(self.h, PP, B, ptrs, _p(nm)), "gfs_search_by_projection")
        res = [(outs[f][:PP[f].n_cur].copy(), int(nm[f])) for f in range(B)]
        return res[0] if single else res


    def SearchByProjectionMap(self, frames):
        """ORBmatcher::SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints) (src/ORBmatcher.cc:43-206):
        one problem dict (keys of gfs_sbp_map_problem) or a list -> (cur_match int32 [n_cur], nmatches) per frame."""
        single = isinstance(frames, dict)
        probs = [frames] if single else list(frames)
        B = len(probs)
        PP = (SbpMapProblem * B)()
        keeps, outs = [], []
        ptrs = (C.c_void_p * B)()
        for f, prob in enumerate(probs):
            P, keep = sbp_map_struct(prob)
            PP[f] = P
            keeps.append(keep)
            outs.append(np.full(max(P.n_cur, 1), -9, np.int32))
            ptrs[f] = outs[f].ctypes.data
        nm = np.zeros(B, np.int32)
        _check(lib().gfs_search_by_projection_map(self.h, PP, B, ptrs, _p(nm)), "gfs_search_by_projection_map")
        res = [(outs[f][:PP[f].n_cur].copy(), int(nm[f])) for f in range(B)]
        return res[0] if single else res


class PoseOptimizer:
    """ORB_SLAM3::Optimizer::PoseOptimization (reference include/Optimizer.h, src/Optimizer.cc:763-1098), conventional-SLAM
    branch, on flattened frames (gfs_pose_problem in include/gfs_abi.h).  A batch of frames is one kernel launch."""

    SUMS_TREE, SUMS_EDGE_ORDER = 0, 1  # GFS_POSE_SUMS_* (include/gfs_abi.h)

    def __init__(self, max_obs=4096, max_batch=64, device=0, sums=None):
        """sums: None = the library's default (g2o's edge order on one lane: the bits of the sequential code, the reference's outlier
        flags), "edge_order" = the same, said explicitly, or "tree" (opt-in: sums over the edges by a fixed-shape tree, about half the
        latency of a single frame, flags equal up to chi2-threshold ties)."""
        self.h = C.c_void_p()
        _check(lib().gfs_pose_create(device, max_obs, max_batch, C.byref(self.h)), "gfs_pose_create")
        if sums is not None:
            _check(lib().gfs_pose_set_sum_order(self.h, {"tree": 0, "edge_order": 1}[sums]), "gfs_pose_set_sum_order")

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gfs_pose_destroy(self.h)
        self.h = None

    __del__ = close

    def PoseOptimization(self, frames):
        """frames: one problem dict or a list of them -> result dict(s) (outlier, chi2, q, t, avg_reproj_error, n_inliers)."""
        single = isinstance(frames, dict)
        probs = [frames] if single else list(frames)
        B = len(probs)
        PP, SS = (PoseProblem * B)(), (PoseSolution * B)()
        keeps = []
        for f, prob in enumerate(probs):
            P, S, keep, n = pose_structs(prob)
            PP[f], SS[f] = P, S
            keeps.append((keep, n))
        _check(lib().gfs_pose_optimize(self.h, PP, B, SS), "gfs_pose_optimize")
        res = [pose_result(SS[f], *keeps[f]) for f in range(B)]
        return res[0] if single else res


class Frame:
    """The two Frame members next to the hot path (reference src/Frame.cc:590-623 and 1314-1332)."""

    def __init__(self, max_rows=720, max_cols=1280, max_keypoints=8192, device=0):
        self.h = C.c_void_p()
        _check(lib().gfs_frame_create(device, max_rows, max_cols, max_keypoints, C.byref(self.h)), "gfs_frame_create")

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.gfs_frame_destroy(self.h)
        self.h = None

    __del__ = close

    def ConvertDepthToPointCloud(self, depth, downSample, fx, fy, cx, cy):
        depth = np.asarray(depth, np.float32)
        if depth.size == 0:
            return np.zeros((0, 4), np.float32)
        rows, cols = depth.shape
        stride = depth.strides[0] // 4
        out = np.zeros((rows * cols, 4), np.float32)
        n = C.c_int()
        _check(lib().gfs_depth_to_cloud(self.h, C.c_void_p(depth.ctypes.data), rows, cols, stride, downSample, fx, fy, cx, cy,
                                        _p(out), len(out), C.byref(n)), "gfs_depth_to_cloud")
        return out[:n.value].copy()

    def ComputeStereoFromRGBD(self, kps, depth, bf, kps_un_x=None):
        depth = np.ascontiguousarray(depth, np.float32)
        kps = np.ascontiguousarray(kps)
        n = len(kps)
        ur = np.zeros(max(n, 1), np.float32)
        vd = np.zeros(max(n, 1), np.float32)
        unx = np.ascontiguousarray(kps_un_x, np.float32) if kps_un_x is not None else None
        _check(lib().gfs_stereo_from_rgbd(self.h, _p(kps), _p(unx), n, _p(depth), depth.shape[0], depth.shape[1], depth.shape[1],
                                          bf, _p(ur), _p(vd)), "gfs_stereo_from_rgbd")
        return ur[:n], vd[:n]

    def FrameRGBD(self, kps, depth, bf, downSample, fx, fy, cx, cy, kps_un_x=None, host_cloud=True, shape=None):
        """The RGB-D tail of the Frame constructor (ComputeStereoFromRGBD + ConvertDepthToPointCloud, src/Frame.cc:1314-1332,
        590-623) in one call: gfs_frame_rgbd.  Returns (mvuRight, mvDepth, cloud or None, (dev_cloud, dev_count, stride, n)).
        depth=None (with shape=(rows, cols)): the depth map of the previous call, still on the device; downSample=0: no cloud."""
        if depth is not None:
            depth = np.ascontiguousarray(depth, np.float32)
            rows, cols = depth.shape
        else:
            rows, cols = shape
        kps = np.ascontiguousarray(kps)
        n = len(kps)
        ur = np.empty(max(n, 1), np.float32)
        vd = np.empty(max(n, 1), np.float32)
        unx = np.ascontiguousarray(kps_un_x, np.float32) if kps_un_x is not None else None
        want_cloud = host_cloud and downSample > 0
        out = np.empty((rows * cols // (downSample * downSample) + rows + cols, 4), np.float32) if want_cloud else None
        nc, stride = C.c_int(), C.c_int()
        dc, dn = C.c_void_p(), C.c_void_p()
        _check(lib().gfs_frame_rgbd(self.h, _p(kps), _p(unx), n, _p(depth), rows, cols, cols, bf, downSample, fx, fy, cx, cy, _p(ur),
                                    _p(vd), _p(out), len(out) if out is not None else 0, C.byref(nc), C.byref(dc), C.byref(dn),
                                    C.byref(stride)), "gfs_frame_rgbd")
        return ur[:n], vd[:n], (out[:nc.value] if out is not None else None), (dc.value, dn.value, stride.value, nc.value)

    def depth_convert_u16_batch_device(self, d_u16, B, rows, cols, factor, d_f32, stream=None):
        """imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor) for CV_16U depth maps already in HBM (src/Tracking.cc:1622-1623)"""
        _check(lib().gfs_depth_convert_u16_batch_device(self.h, C.c_void_p(d_u16), B, rows, cols, float(factor), C.c_void_p(d_f32),
                                                        C.c_void_p(stream) if stream else None), "gfs_depth_convert_u16_batch_device")

    def depth_to_cloud_batch_device(self, d_depth, B, rows, cols, ds, fx, fy, cx, cy, d_out, stride_pts, d_counts, stream=None):
        _check(lib().gfs_depth_to_cloud_batch_device(self.h, C.c_void_p(d_depth), B, rows, cols, ds, fx, fy, cx, cy,
                                                     C.c_void_p(d_out), stride_pts, C.c_void_p(d_counts),
                                                     C.c_void_p(stream) if stream else None), "gfs_depth_to_cloud_batch_device")
