#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ from the CPU oracle (oracle/).

The reference (HorizonRobotics/GeoFlowSlam) ships no golden vectors for this path and cannot be built here
(OpenCV/Eigen absent), so these fixtures pin the ORACLE's behaviour at a point in time: the CPU
suite checks the oracle still reproduces them, the GPU suite checks the HIP path against them without needing
the oracle build.  Inputs are seeded synthetic data (geoflowslam_amd/synth.py).  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from geoflowslam_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    # 1. ORB: 160x120 rendered frame, 300 features, 3 levels
    fp = synth.frame_pair(77, 160, 120, 2)
    orb = O.OrbOracle(300, 1.2, 3, 20, 7)
    m0, k0, d0 = orb.extract(fp["gray0"])
    m1, k1, d1 = orb.extract(fp["gray1"])
    np.savez_compressed(os.path.join(OUT, "orb_160x120.npz"), gray0=fp["gray0"], gray1=fp["gray1"], mono0=m0, kps0=k0,
                        desc0=d0, mono1=m1, kps1=k1, desc1=d1, params=np.array([300, 3, 20, 7]))
    # 2. BF match of the two descriptor sets + a synthetic tie case
    ti, di = O.bf_match(d0, d1)
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (40, 32)).astype(np.uint8)
    t = rng.integers(0, 256, (50, 32)).astype(np.uint8)
    t[30] = t[4]
    q[0] = t[4]
    ti2, di2 = O.bf_match(q, t)
    np.savez_compressed(os.path.join(OUT, "bf_match.npz"), train_idx=ti, dist=di, q=q, t=t, train_idx2=ti2, dist2=di2)
    # 3. GICP on the two depth clouds of the same pair (~4.7k points each)
    r = O.gicp_align(fp["cloud0"], fp["cloud1"])
    np.savez_compressed(os.path.join(OUT, "gicp_160x120.npz"), cloud0=fp["cloud0"], cloud1=fp["cloud1"], T=r["T"],
                        converged=r["converged"], iterations=r["iterations"], num_inliers=r["num_inliers"],
                        H=r["H"], b=r["b"], error=r["error"], n_ds=np.array([r["n_target_ds"], r["n_source_ds"]]))
    # 4. LBA: 4 free + 2 fixed key-frames, 80 points
    w = synth.lba_window(9, n_free=4, n_fixed=2, n_points=80)
    s = O.lba_solve(w)
    L = O.lba_linearize(w)
    keys = ("n_poses", "n_points", "n_edges", "pose_q", "pose_t", "pose_fixed", "points", "edge_pose", "edge_point",
            "edge_obs", "edge_inv_sigma2", "edge_stereo", "fx", "fy", "cx", "cy", "bf", "huber_mono", "huber_stereo",
            "iterations")
    np.savez_compressed(os.path.join(OUT, "lba_small.npz"), **{k: w[k] for k in keys}, out_pose_q=s["pose_q"],
                        out_pose_t=s["pose_t"], out_points=s["points"], out_edge_chi2=s["edge_chi2"],
                        out_iterations=s["iterations_run"], out_chi2=s["final_chi2"], lin_bp=L["bp"], lin_bl=L["bl"],
                        lin_Hpp=L["Hpp"], lin_chi2=L["chi2"])
    # 5. optical flow: the ORB key points of frame 0 tracked into frame 1 (window 15, 3 pyramid levels: 40 x 30 is the last
    #    level larger than the window), forward-backward checked; plus a plain calcOpticalFlowPyrLK call with the L1 error
    import zlib
    kps = np.stack([k0["x"], k0["y"]], 1).astype(np.float32)
    p0, p1 = O.klt_build_pyramid(fp["gray0"], 15), O.klt_build_pyramid(fp["gray1"], 15)
    pri, ok, good = O.fb_klt_tracking(p0, p1, 160, 120, 15, 3, 15.0, 0.5, kps, kps + np.float32(0.5))
    nxt, st, er = O.klt_track(p0, p1, 160, 120, 15, kps, max_level=2, flags=0)
    np.savez_compressed(os.path.join(OUT, "klt_160x120.npz"), kps=kps, priors=pri, status=ok, good=good, next=nxt, st=st, err=er,
                        pyr_crc=np.array([zlib.crc32(p0[0].tobytes()), zlib.crc32(p0[1].tobytes()), zlib.crc32(p1[0].tobytes()),
                                          zlib.crc32(p1[1].tobytes())], np.int64))
    # 6. fundamental-matrix RANSAC on synthetic two-view matches (200 points, 30 % gross mismatches)
    q1, q2, _, _ = synth.two_view_points(77, 200, 0.3)
    fmask, fF, fcnt, fit = O.fundamental_ransac(q1, q2, 2.0, 0.99)
    np.savez_compressed(os.path.join(OUT, "fmat_200.npz"), pts1=q1, pts2=q2, mask=fmask, F=fF, count=fcnt, iterations=fit)
    # 7. the SURVEY 8(f) rows next to the path: windowed matching (both overloads), motion-only BA, the GMS filter of the ORB matches
    sp = synth.sbp_pair(3, n_points=200, n_extra_cur=40)
    sm, sn = O.search_by_projection(sp)
    mp = synth.sbp_map_frame(4, n_points=200, n_extra_cur=40)
    mm, mn = O.search_by_projection_map(mp)
    pf = synth.pose_frame(3, n_obs=150)
    pr = O.pose_optimization(pf)
    gq = np.arange(len(ti), dtype=np.int32)
    gmask, gcnt = O.gms_inlier_mask(k0, (160, 120), k1, (160, 120), gq, ti)
    np.savez_compressed(os.path.join(OUT, "next_rows.npz"), **{"sbp_" + k: np.asarray(v) for k, v in sp.items()}, sbp_out=sm, sbp_n=sn,
                        **{"map_" + k: np.asarray(v) for k, v in mp.items()}, map_out=mm, map_n=mn,
                        **{"pose_" + k: np.asarray(v) for k, v in pf.items()}, poseout_outlier=pr["outlier"], poseout_q=pr["q"], poseout_t=pr["t"],
                        poseout_inliers=pr["n_inliers"], poseout_iterations=pr["iterations_run"], gms_mask=gmask, gms_count=gcnt)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
