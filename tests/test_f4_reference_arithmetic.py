"""how far are the order-independent variants the HIP path implements (exact integer sums in
calcOpticalFlowPyrLK; Gauss-Jordan null space + bisection in the 7-point solver of findFundamentalMat) from OpenCV's own
arithmetic?  The oracle carries both (oracle/klt_oracle.cpp gfso_klt_set_accumulation, oracle/fmat_oracle.cpp
gfso_fmat_set_solver); this module tracks / estimates the same inputs both ways and bounds the differences.  Reference call sites:
src/ORBmatcher.cc:2186-2297 (fbKltTracking), :2399-2405 (findFundamentalMat).

    python tests/test_f4_reference_arithmetic.py        # writes profiles/r03_f4_deviation.json (every differing case listed)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from geoflowslam_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def measure_klt(n_pairs=6, windows=(15, 21, 35)):
    """fbKltTracking of ~1000 ORB key points per VGA pair in the three accumulation modes.  Per float mode: points, status flips,
    |delta position| statistics over the points both variants keep, and the list of points that differ by more than 1e-3 px."""
    out = {1: dict(points=0, status_differs=[], deltas=[], over_1e3=[]), 2: dict(points=0, status_differs=[], deltas=[], over_1e3=[])}
    for seed in range(n_pairs):
        fp = synth.frame_pair(seed, 640, 480)
        _, k, _ = O.OrbOracle(1000, 1.2, 8, 20, 7).extract(fp["gray0"])
        kps = np.stack([k["x"], k["y"]], 1).astype(np.float32)
        for win in windows:
            p0, p1 = O.klt_build_pyramid(fp["gray0"], win), O.klt_build_pyramid(fp["gray1"], win)
            res = {}
            try:
                for mode in (0, 1, 2):
                    O.klt_set_accumulation(mode)
                    res[mode] = O.fb_klt_tracking(p0, p1, 640, 480, win, 3, 15.0, 0.5, kps, kps.copy())
            finally:
                O.klt_set_accumulation(0)
            pri0, ok0, _ = res[0]
            for m in (1, 2):
                pri, ok, _ = res[m]
                o = out[m]
                o["points"] += len(kps)
                for i in np.nonzero(ok != ok0)[0]:
                    o["status_differs"].append(dict(seed=seed, win=win, point=int(i), exact=bool(ok0[i]), float_order=bool(ok[i])))
                both = np.nonzero(ok & ok0)[0]
                d = np.abs(pri[both] - pri0[both]).max(1) if len(both) else np.zeros(0)
                o["deltas"].append(d)
                for i, v in zip(both, d):
                    if v > 1e-3:
                        o["over_1e3"].append(dict(seed=seed, win=win, point=int(i), delta_px=float(v)))
    for m in (1, 2):
        d = np.concatenate(out[m].pop("deltas"))
        out[m].update(max_px=float(d.max()), median_px=float(np.median(d)), p99_px=float(np.quantile(d, 0.99)), p999_px=float(np.quantile(d, 0.999)),
                      identical_fraction=float((d == 0).mean()))
    return {"opencv_scalar_float_loop": out[1], "four_lane_float_model": out[2]}


def measure_fmat(n_cases=1500):
    """findFundamentalMat(FM_RANSAC) both ways: 8 ... 1500 points, 5 ... 70 % gross outliers, thresholds 0.5 / 1 / 3 px."""
    ransac = dict(cases=0, mask_differs=[], iterations_differ=0)
    lmeds = dict(cases=0, cases_median_among_fitted=0, mask_differs=[], mask_differs_median_among_fitted=0)
    for i in range(n_cases):
        rng = np.random.default_rng([5, i])
        n = int(rng.choice([8, 10, 13, 14, 15, 30, 100, 400, 1000, 1500]))
        p1, p2 = synth.two_view_points(int(rng.integers(0, 1 << 30)), n=n, outlier_frac=float(rng.uniform(0.05, 0.7)), noise=float(rng.uniform(0.05, 1.0)))[:2]
        thr = float(rng.choice([0.5, 1.0, 3.0]))
        try:
            O.fmat_set_solver(0)
            m0, _, c0, it0 = O.fundamental_ransac(p1, p2, thr, 0.99)
            O.fmat_set_solver(1)
            m1, _, c1, it1 = O.fundamental_ransac(p1, p2, thr, 0.99)
        finally:
            O.fmat_set_solver(0)
        differs = c0 != c1 or not np.array_equal(m0, m1)
        if n >= 15:
            ransac["cases"] += 1
            ransac["iterations_differ"] += it0 != it1
            if differs:
                ransac["mask_differs"].append(dict(case=i, n=n, threshold=thr, inliers=[c0, c1], differing_flags=int((m0 != m1).sum())))
        else:
            noise = n // 2 < 7  # the median error belongs to one of the 7 points the model was fitted to: pure rounding noise
            lmeds["cases"] += 1
            lmeds["cases_median_among_fitted"] += noise
            if differs:
                lmeds["mask_differs"].append(dict(case=i, n=n, inliers=[c0, c1], differing_flags=int((m0 != m1).sum())))
                lmeds["mask_differs_median_among_fitted"] += noise
    return {"ransac_n_ge_15": ransac, "lmeds_n_8_to_14": lmeds}


def test_exact_sums_against_opencvs_float_accumulation():
    r = measure_klt(n_pairs=4, windows=(15, 21, 35))
    for name, o in r.items():
        assert o["points"] > 9000
        assert len(o["status_differs"]) <= max(2, o["points"] // 5000), (name, o["status_differs"])  # measured: 1 of 18 093
        assert o["max_px"] <= 0.05 and o["p99_px"] <= 1e-3 and o["median_px"] <= 1e-4, (name, {k: v for k, v in o.items() if k.endswith("_px")})


def test_seven_point_solver_against_opencvs_internals():
    r = measure_fmat(700)
    ra, lm = r["ransac_n_ge_15"], r["lmeds_n_8_to_14"]
    assert ra["cases"] >= 350
    # RANSAC: same iteration counts and consensus SIZES everywhere; the sets themselves differ only where two models of one subset
    # tie on the inlier count (OpenCV keeps the first, and the two root finders order a subset's models differently): measured 1 of 891
    assert ra["iterations_differ"] == 0 and all(d["inliers"][0] == d["inliers"][1] for d in ra["mask_differs"]), ra
    assert len(ra["mask_differs"]) <= max(1, ra["cases"] // 200), ra["mask_differs"]
    # LMedS (8 ... 14 points): with the median index n / 2 < 7 the criterion is the error of an exactly fitted point, i.e. rounding noise:
    # any two arithmetic variants (and any two OpenCV builds) disagree there; only n = 14 has a meaningful median
    assert lm["mask_differs_median_among_fitted"] == len(lm["mask_differs"]), lm["mask_differs"][:5]


if __name__ == "__main__":
    rep = {"klt": measure_klt(), "fmat": measure_fmat()}
    path = os.path.join(ROOT, "profiles", "r03_f4_deviation.json")
    json.dump(rep, open(path, "w"), indent=1)
    k, f = rep["klt"]["opencv_scalar_float_loop"], rep["fmat"]
    print("klt vs OpenCV scalar loop:", {x: k[x] for x in ("points", "max_px", "median_px", "p99_px", "p999_px", "identical_fraction")}, "status flips", len(k["status_differs"]))
    print("fmat RANSAC:", f["ransac_n_ge_15"]["cases"], "cases,", len(f["ransac_n_ge_15"]["mask_differs"]), "differ; LMedS:", f["lmeds_n_8_to_14"]["cases"], "cases,",
          len(f["lmeds_n_8_to_14"]["mask_differs"]), "differ, all with the median among the fitted points:", f["lmeds_n_8_to_14"]["mask_differs_median_among_fitted"])
