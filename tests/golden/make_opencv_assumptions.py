#!/usr/bin/env python3
"""Generates tests/golden/opencv_assumptions.npz: what the CPU ORACLE produces for the inputs check_with_opencv.py feeds to a
real OpenCV (the inputs are regenerated there from the same seeds).  Run from the repo root."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(0)
src = rng.integers(0, 256, (120, 160)).astype(np.uint8)
g = np.arange(-40, 41, dtype=np.float32)
yy, xx = np.meshgrid(g * 37, g * 53, indexing="ij")
ys, xs = yy.ravel(), xx.ravel()
x, y, s = O.fast9_16(np.ascontiguousarray(src[:60, :80]), 20)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_sbp_of import _circle  # noqa: E402  (the transcription of drawing.cpp Circle() the adaptor is tested against)
circles = {}
for r, (cx_, cy_) in ((10, (40.0, 37.0)), (10, (40.5, 36.5)), (3, (2.2, 3.7)), (30, (78.9, 59.1)), (0, (5.0, 5.0))):
    a = np.zeros((60, 80), np.uint8)
    _circle(a, cx_, cy_, r)
    circles["circle_r%d_%d" % (r, int(cx_))] = a
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "opencv_assumptions.npz"), **circles,
                    area_133x100=O.resize_area(src, 100, 133), blur_v0=O.gaussian_blur7(src, 0), blur_v1=O.gaussian_blur7(src, 1),
                    atan_y=ys, atan_x=xs, atan_deg=np.array([O.fast_atan2(float(a), float(b)) for a, b in zip(ys, xs)], np.float32),
                    fast_xys=np.stack([x, y, s], 1).astype(np.int32))
print("written")
