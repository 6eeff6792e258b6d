"""The device side of csrc/glibc_math.hpp against the host's libm (through Python's math module): sin / cos / pow(x, 3) bit for bit over the
argument ranges the optimizers produce and well beyond (tests/test_glibc_math.py checks the same header compiled for the host)."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device(gpu_api, x):
    L = gpu_api.lib()
    x = np.ascontiguousarray(x, np.float64)
    s, c, p = (np.empty_like(x) for _ in range(3))
    rc = L.gfs_test_glibc_math(0, x.ctypes.data, len(x), s.ctypes.data, c.ctypes.data, p.ctypes.data)
    assert rc == 0, L.gfs_last_error()
    return s, c, p


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def test_device_sin_cos_pow3_have_glibcs_bits(gpu_api):
    rng = np.random.default_rng(5)
    parts = [rng.uniform(-2.42, 2.42, 2_000_000), rng.uniform(-0.2, 0.2, 1_000_000),
             rng.uniform(-1, 1, 1_000_000) * np.exp2(-rng.integers(0, 40, 1_000_000).astype(np.float64)),
             np.array([0.0, -0.0, 1.0, -1.0, 0.126, 0.125, 0.855469, 0.8554687, 2.426265, 2.4262, 1e-5, 1e-8, 2.0 ** -26, 2.0 ** -27, 0.5, 3.0])]
    x = np.concatenate(parts)
    s, c, p = _device(gpu_api, x)
    # math.* call the C library directly (numpy's ufuncs may use their own SIMD kernels)
    assert np.array_equal(_bits(s), _bits(np.array([math.sin(v) for v in x])))
    assert np.array_equal(_bits(c), _bits(np.array([math.cos(v) for v in x])))
    assert np.array_equal(_bits(p), _bits(np.array([math.pow(v, 3.0) for v in x])))


def test_pow3_far_from_one(gpu_api):
    rng = np.random.default_rng(6)
    x = np.concatenate([rng.uniform(-1, 1, 500_000) * np.exp2(rng.integers(-300, 170, 500_000).astype(np.float64)),
                        np.array([1e300, -1e300, 1e-300, 1e-110, -1e-110, 5e-324, 2.2250738585072014e-308, 1e103, np.inf, -np.inf])])
    _, _, p = _device(gpu_api, x)
    with np.errstate(over="ignore", under="ignore"):
        want = np.array([math.pow(v, 3.0) if abs(v) < 1e103 else (math.copysign(math.inf, v)) for v in x])
    bad = np.nonzero(_bits(p) != _bits(want))[0]
    assert len(bad) == 0, [(float(x[i]).hex(), float(p[i]).hex(), float(want[i]).hex()) for i in bad[:8]]
