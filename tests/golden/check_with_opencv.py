#!/usr/bin/env python3
"""Runs the committed fixtures and the SURVEY.md App. A assumptions against a REAL OpenCV (cv2), which this build image does not
have.  Usage, on any machine with `pip install opencv-python==4.5.4.60 numpy` (the pinned semantics; newer 4.x also works for
everything except A4's taps on < 4.5.1):

    python tests/golden/check_with_opencv.py            # prints PASS / FAIL per assumption id

It needs only numpy + cv2 + the .npz files next to it: nothing from this repository is imported, so a maintainer of the
reference can run it inside the reference's own environment.  A FAIL names the App. A assumption that does not hold for that
OpenCV build — the oracle (oracle/orb_oracle.cpp, oracle/klt_oracle.cpp) has to follow the library, not the other way round."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    try:
        import cv2
    except ImportError:
        print("cv2 is not installed: nothing checked (this script is for a machine that has OpenCV)")
        return 2
    print("OpenCV", cv2.__version__)
    ok = True

    def report(tag, cond, note=""):
        nonlocal ok
        ok &= bool(cond)
        print(("PASS " if cond else "FAIL ") + tag + (" - " + note if note else ""))

    d = np.load(os.path.join(HERE, "orb_160x120.npz"))
    g0 = d["gray0"]
    # A2: INTER_AREA pyramid level (160x120 -> 133x100), the oracle's level 1 is not stored; check the primitive on a ramp
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (120, 160)).astype(np.uint8)
    lvl = cv2.resize(src, (133, 100), interpolation=cv2.INTER_AREA)
    exp = np.load(os.path.join(HERE, "opencv_assumptions.npz"))
    report("A2 resize INTER_AREA 160x120 -> 133x100 (float weights, round-half-even)", np.array_equal(lvl, exp["area_133x100"]))
    # A4: GaussianBlur 7x7 sigma 2 REFLECT_101 on 8U (fixed point, taps 18 34 48 56 48 34 18 for >= 4.5.1)
    bl = cv2.GaussianBlur(src, (7, 7), 2, 2, cv2.BORDER_REFLECT_101)
    report("A4 GaussianBlur taps {18,34,48,56,48,34,18} (OpenCV >= 4.5.1)", np.array_equal(bl, exp["blur_v0"]),
           "if this fails and the next passes, build the oracle with blur_taps_variant = 1")
    report("A4' GaussianBlur taps {18,34,49,55,49,34,18} (OpenCV 4.0 - 4.5.0)", np.array_equal(bl, exp["blur_v1"]))
    # A5 / A3: fastAtan2 on the moment lattice
    ys, xs = exp["atan_y"], exp["atan_x"]
    got = np.array([cv2.fastAtan2(float(a), float(b)) for a, b in zip(ys, xs)], np.float32)
    report("A3 fastAtan2 (degree-7 polynomial, float)", np.array_equal(got.view(np.uint32), exp["atan_deg"].view(np.uint32)),
           "max |diff| %.3g deg" % float(np.abs(got - exp["atan_deg"]).max()))
    # A1: FAST 9_16 with NMS: positions, scores, raster order
    fast = cv2.FastFeatureDetector_create(threshold=20, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
    kp = fast.detect(src[:60, :80].copy(), None)
    got = np.array([(int(k.pt[0]), int(k.pt[1]), int(k.response)) for k in kp], np.int32).reshape(-1, 3)
    report("A1 cv::FAST 9_16 threshold 20 + NMS (x, y, score, order)", np.array_equal(got, exp["fast_xys"]))
    # the whole extractor through cv2 is not available (ORBextractor is the reference's class); A6: BFMatcher ties
    m = np.load(os.path.join(HERE, "bf_match.npz"))
    bf = cv2.BFMatcher(cv2.NORM_HAMMING)
    mm = bf.match(m["q"], m["t"])
    report("A6 BFMatcher(NORM_HAMMING).match: lowest train index on ties", [x.trainIdx for x in mm] == m["train_idx2"].tolist()
           and [int(x.distance) for x in mm] == m["dist2"].tolist())
    # A8: cv::circle(mask, Point2f, r, 255, FILLED) (the occupancy mask of SearchByProjectionWithOF)
    okc = True
    for r, (x, y) in ((10, (40.0, 37.0)), (10, (40.5, 36.5)), (3, (2.2, 3.7)), (30, (78.9, 59.1)), (0, (5.0, 5.0))):
        a = np.zeros((60, 80), np.uint8)
        cv2.circle(a, (int(np.rint(np.float32(x))), int(np.rint(np.float32(y)))), r, 255, cv2.FILLED)
        okc &= np.array_equal(a, exp["circle_r%d_%d" % (r, int(x))])
    report("A8 cv::circle filled disc (midpoint recurrence, spans)", okc)
    # optical flow: pyramid + tracker agree to ~1e-3 px (float summation order is build dependent)
    k = np.load(os.path.join(HERE, "klt_160x120.npz"))
    p0 = cv2.buildOpticalFlowPyramid(g0, (15, 15), 3)[1]
    p1 = cv2.buildOpticalFlowPyramid(d["gray1"], (15, 15), 3)[1]
    nxt, st, err = cv2.calcOpticalFlowPyrLK(p0, p1, k["kps"].reshape(-1, 1, 2), None, winSize=(15, 15), maxLevel=2,
                                            criteria=(cv2.TERM_CRITERIA_COUNT | cv2.TERM_CRITERIA_EPS, 30, 0.01))
    same = st.ravel() == k["st"]
    both = (st.ravel() > 0) & (k["st"] > 0)
    dev = np.abs(nxt.reshape(-1, 2) - k["next"])[both].max() if both.any() else 0.0
    report("LK status equal, positions within 0.02 px of the exact-sum variant", same.mean() > 0.99 and dev < 0.02,
           "status equal %.1f %%, max position difference %.4f px" % (100 * same.mean(), dev))
    print("ALL PASS" if ok else "SOME ASSUMPTIONS DO NOT HOLD FOR THIS OPENCV")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
