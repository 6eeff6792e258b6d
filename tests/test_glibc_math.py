"""geoflowslam_amd/csrc/glibc_math.hpp (sin / cos / pow(x, 3) with glibc 2.35's arithmetic, used by the device optimizers for
g2o's SE3Quat::exp, Thirdparty/g2o/g2o/types/se3quat.h:223-257, and its Levenberg step control,
core/optimization_algorithm_levenberg.cpp:127) compiled for the HOST and compared bit for bit with this machine's libm -- the
library the CPU restatement (and the reference) calls.  The committed tables must be what tools/extract_glibc_tables.py reads
out of that libm."""
import os
import platform
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _glibc_235_with_fma():
    if platform.machine() != "x86_64" or platform.libc_ver()[0] != "glibc":
        return False
    try:
        return "fma" in open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0].split()
    except Exception:
        return False


needs_host_libm = pytest.mark.skipif(not _glibc_235_with_fma(), reason="needs an x86-64 glibc host with FMA (the libm the oracle calls)")


@needs_host_libm
def test_restated_sin_cos_pow3_equal_the_host_libm(tmp_path):
    exe = tmp_path / "glibc_math_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", os.path.join(ROOT, "tests", "host", "glibc_math_check.cpp"),
                           "-o", str(exe)])
    out = subprocess.run([str(exe), "20000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("sin 0 cos 0 pow3 0 of 20000000")


@needs_host_libm
def test_committed_tables_are_the_hosts(tmp_path):
    if platform.libc_ver()[1] != "2.35":
        pytest.skip("tables were read from glibc 2.35")
    inc = os.path.join(ROOT, "geoflowslam_amd", "csrc", "glibc_tables.inc")
    fresh = tmp_path / "glibc_tables.inc"  # (never the tracked file: touching it would make every object of the library stale)
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "extract_glibc_tables.py"), "--out", str(fresh)], stdout=subprocess.DEVNULL)
    assert open(fresh).read() == open(inc).read()
