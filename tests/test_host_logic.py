"""CPU tests of the product's host-side logic (no GPU): the index-based quadtree must reproduce the oracle's
std::list restatement of DistributeOctTree (reference src/ORBextractor.cc:567-768) exactly, including order."""
import numpy as np
import pytest


@pytest.mark.parametrize("seed", range(12))
def test_octree_host_matches_oracle(api, oracle, seed):
    rng = np.random.default_rng(seed)
    w, h = [(608, 448), (501, 368), (1248, 688), (338, 246), (147, 102)][seed % 5]
    n = int(rng.integers(1, 6000))
    # clustered candidates like real FAST output, unique pixel positions
    centers = rng.uniform(0, 1, (12, 2)) * [w, h]
    pts = centers[rng.integers(0, 12, n)] + rng.normal(0, 25, (n, 2))
    pts = np.unique(np.clip(np.rint(pts), [3, 3], [w - 4, h - 4]).astype(np.int32), axis=0)
    order = np.lexsort((pts[:, 0], pts[:, 1]))
    pts = pts[order]
    score = rng.integers(7, 120, len(pts)).astype(np.int32)
    quota = int(rng.choice([5, 60, 217, 434]))
    got = api.octree_host(pts[:, 0], pts[:, 1], score, 16, 16 + w, 16, 16 + h, quota)
    exp = oracle.distribute_octree(pts[:, 0], pts[:, 1], score, 16, 16 + w, 16, 16 + h, quota)
    assert got.tolist() == exp.tolist()


def test_octree_host_degenerate(api, oracle):
    for x, y, s in [([5], [5], [9]), ([5, 5], [5, 5], [9, 9]), ([3, 600, 3, 600], [3, 3, 440, 440], [1, 2, 3, 4])]:
        got = api.octree_host(x, y, s, 16, 624, 16, 464, 100)
        exp = oracle.distribute_octree(x, y, s, 16, 624, 16, 464, 100)
        assert got.tolist() == exp.tolist()
    assert len(api.octree_host([], [], [], 16, 624, 16, 464, 100)) == 0


def test_hamming256_host(api, oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.integers(0, 256, 32).astype(np.uint8)
        b = rng.integers(0, 256, 32).astype(np.uint8)
        ref = int(np.unpackbits(a ^ b).sum())
        assert api.ORBmatcher.DescriptorDistance(a, b) == ref == oracle.descriptor_distance(a, b)


def _sort_pair(fn_prod, fn_oracle, s, x):
    import ctypes as C
    n = len(s)
    p = np.arange(n, dtype=np.int32)
    a = [s.astype(np.int32).copy(), x.astype(np.int32).copy(), p.copy()]
    b = [s.astype(np.int32).copy(), x.astype(np.int32).copy(), p.copy()]
    fn_prod(*[v.ctypes.data_as(C.c_void_p) for v in a], n)
    fn_oracle(*[v.ctypes.data_as(C.c_void_p) for v in b], n)
    return all((u == v).all() for u, v in zip(a, b))


def test_std_sort_replica_matches_libstdcxx(api, oracle):
    """The device quadtree re-implements libstdc++'s std::sort (introsort + final insertion sort + heap fallback)
    because the reference's tie order depends on it (src/ORBextractor.cc:552-565, 697-698).  Same permutation as
    std::sort / std::partial_sort on tie-heavy inputs."""
    P, O = api.lib(), oracle.lib()
    rng = np.random.default_rng(0)
    for trial in range(600):
        n = int(rng.integers(0, 700))
        mode = trial % 4
        if mode == 0:
            s, x = rng.integers(2, 6, n), rng.integers(0, 8, n) * 10
        elif mode == 1:
            s, x = rng.integers(2, 200, n), rng.integers(0, 600, n)
        elif mode == 2:
            s, x = np.full(n, 3), np.full(n, 7)
        else:
            s, x = np.sort(rng.integers(2, 30, n))[::-1].copy(), rng.integers(0, 3, n)
        assert _sort_pair(P.gfs_test_sort_replica, O.gfso_std_sort_pairs, s, x)
        assert _sort_pair(P.gfs_test_heap_sort_replica, O.gfso_std_partial_sort_pairs, s, x)


def test_blur_tile_order_groups_the_tiles_over_one_line_on_one_xcd(api):
    """k_blur7's launch order (OrbGeometry::build, round 6): every 64 x 32 tile of every level exactly once; MI355X hands workgroup i
    to XCD i % 8, so the 2 x 2 tiles over the same 128-byte lines and halo rows -- (2k, 2m), (2k + 1, 2m), (2k, 2m + 1), (2k + 1, 2m + 1) --
    must sit at positions p, p + 8, p + 16, p + 24, pairs at p, p + 8; only a tail of fewer than eight groups may be left unaligned."""
    import ctypes as C
    L = api.lib()
    L.gfs_test_orb_blur_tiles.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int]
    L.gfs_test_orb_blur_tiles.restype = C.c_int
    for rows, cols, nl, sf in ((480, 640, 8, 1.2), (720, 1280, 8, 1.2), (240, 320, 6, 1.2), (203, 517, 3, 1.5), (128, 160, 2, 1.2)):
        out = np.zeros((8192, 3), np.int32)
        n = L.gfs_test_orb_blur_tiles(rows, cols, nl, sf, out.ctypes.data_as(C.c_void_p), len(out))
        assert 0 < n <= len(out)
        tiles = [tuple(int(v) for v in t) for t in out[:n]]
        # exactly the tiles of every level (cvRound(size / scale) as ComputePyramid sizes the levels)
        want = set()
        for l in range(nl):
            inv = np.float32(1.0) / np.float32(sf) ** l if l else np.float32(1.0)
            lc, lr = int(np.rint(np.float32(cols) * np.float32(1.0 / float(np.float32(sf) ** l)))), int(np.rint(np.float32(rows) * np.float32(1.0 / float(np.float32(sf) ** l))))
            want |= {(l, tx, ty) for ty in range((lr + 31) // 32) for tx in range((lc + 63) // 64)}
        assert len(tiles) == len(set(tiles)) == n and len(set(t[0] for t in tiles)) == nl
        assert abs(len(want) - n) <= 2 * nl, (len(want), n)  # (level sizes: this test's rounding may differ from cvRound by a pixel)
        pos = {t: i for i, t in enumerate(tiles)}
        grouped = loose = 0
        for (l, tx, ty), p in pos.items():
            if tx % 2 or ty % 2:
                continue
            block = [t for t in ((l, tx + 1, ty), (l, tx, ty + 1), (l, tx + 1, ty + 1)) if t in pos]
            same_xcd = all(pos[t] % 8 == p % 8 for t in block)
            tight = all(0 < pos[t] - p <= 24 and (pos[t] - p) % 8 == 0 for t in block)
            if block:
                grouped += int(same_xcd and tight)
                loose += int(not (same_xcd and tight))
        assert loose <= 14, (rows, cols, loose)  # the tails: fewer than eight blocks, then fewer than eight pairs
        assert grouped >= 1 or n < 8, (rows, cols, grouped)
        if n >= 64:
            assert grouped >= 0.8 * (grouped + loose), (rows, cols, grouped, loose)


def test_reference_dropins_compile():
    """geoflowslam_amd/host/gfs_reference_dropins.hpp -- the code INTEGRATION.md tells a maintainer to add to the reference tree --
    goes through a compiler: against the reference's real include/ORBextractor.h and small_gicp registration_result.hpp when
    /root/reference is on this machine (a reduced declaration of the two otherwise), over declaration-only OpenCV / Eigen / Sophus
    stand-ins (tests/host/stubs/).  Syntax, override signatures and struct layouts only: nothing is linked or run."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(root, "tests", "host", "stubs")]
    ref = "/root/reference"
    variants = [[]]
    if os.path.exists(os.path.join(ref, "include", "ORBextractor.h")):
        variants.append(["-DGFS_HAVE_REFERENCE_TREE", "-I" + os.path.join(ref, "include"),
                         "-I" + os.path.join(ref, "Thirdparty", "small_gicp", "include")])
    for extra in variants:
        out = subprocess.run(cmd + extra + [os.path.join(root, "tests", "host", "reference_dropins_check.cpp")], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-3000:]


def test_pyramid_strip_geometry(tmp_path):
    """The strip cuts k_pyr_area_lds is launched with (OrbGeometry::build, csrc/orb_host.cpp), over image sizes from 160 x 120 to
    4000 x 3000, scale factors 1.1 ... 2.5 and 2 / 4 / 8 levels: every row of every level is produced by a strip, every row a strip
    produces reads source rows the same strip holds, the strips fit their LDS buffers and the kernel's row tables, the y tables are
    streamable, and the cell / blur-tile descriptors carry their level's geometry (tests/host/orb_geometry_check.cpp)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "orb_geometry_check")
    src = [os.path.join(root, "tests", "host", "orb_geometry_check.cpp"), os.path.join(root, "geoflowslam_amd", "csrc", "orb_host.cpp")]
    inc = ["-I" + os.path.join(root, "geoflowslam_amd", "csrc"), "-I" + os.path.join(root, "include")]
    out = subprocess.run(["g++", "-std=c++17", "-O1"] + inc + src + ["-o", exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True)
    assert run.returncode == 0 and "violations: 0" in run.stdout, run.stdout[-3000:]
    assert "640x480 sf=1.20 nl=8 cut: 6 strips" in run.stdout  # (the cut the bench runs with: the x tables in LDS)
