"""CPU tests: the built C-ABI library loads without a GPU and exports every symbol include/gfs_abi.h declares;
compute entry points fail loudly (no CPU fallback); frame sharding across ranks (gloo, world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(api):
    declared = []
    for name in ("gfs_abi.h", "gfs_abi_test.h"):  # the boundary, and the test hooks kept apart from it
        hdr = open(os.path.join(ROOT, "include", name)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        syms = set(re.findall(r"\b(gfs_[a-z0-9_]+)\s*\(", hdr))
        assert all(s.startswith("gfs_test_") for s in syms) == (name == "gfs_abi_test.h"), name
        assert not any(s.startswith("gfs_test_") for s in syms) or name == "gfs_abi_test.h", "test hook in the product header"
        declared += sorted(syms)
    declared = sorted(set(declared))
    assert len(declared) >= 35
    L = api.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, f"libgfs_hip.so does not export: {missing}"
    assert sorted(api.ABI_SYMBOLS) == declared, "geoflowslam_amd.api.ABI_SYMBOLS is out of sync with include/gfs_abi.h"
    assert L.gfs_abi_version() == 1


def test_no_cpu_fallback_without_gpu(api):
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(api.GfsError, match="no usable HIP device|no CPU fallback"):
        api.ORBextractor()
    with pytest.raises(api.GfsError):
        api.ORBmatcher()
    with pytest.raises(api.GfsError):
        api.RegistrationGICP()
    with pytest.raises(api.GfsError):
        api.Optimizer()
    for cls in (api.Frame, api.ProjectionMatcher, api.PoseOptimizer, api.GmsMatcher, api.FundamentalMatcher):
        with pytest.raises(api.GfsError):
            cls()
    with pytest.raises(api.GfsError):
        api.KltTracker(640, 480, 35)


def test_product_does_not_reference_the_oracle():
    """Nothing under geoflowslam_amd/ may import, link or call oracle/ (oracle/gfs_oracle.h)."""
    pkg = os.path.join(ROOT, "geoflowslam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", ".inc")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "gfso_" not in txt and "libgfs_oracle" not in txt and "from oracle" not in txt, f
    out = subprocess.run(["ldd", os.path.join(pkg, "libgfs_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_shard_range_partition():
    from geoflowslam_amd.shard import pair_halo, shard_range
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(512, 3, 8) == (192, 256) and pair_halo(192) == 191 and pair_halo(0) == 0


_WORKER = r"""
import os, sys, hashlib
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from geoflowslam_amd import api
from geoflowslam_amd.shard import shard_range, gather_counts
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n_items = 37
b, e = shard_range(n_items, rank, world)
rng = np.random.default_rng(123)
desc = rng.integers(0, 256, (n_items + 1, 32)).astype(np.uint8)   # frame i's "descriptor"
# each rank processes its own pairs (i-1, i) with the host Hamming helper; no data-path collective
local = np.array([api.ORBmatcher.DescriptorDistance(desc[i], desc[i + 1]) for i in range(b, e)], np.int64)
total = gather_counts(len(local), dist)
out = [None] * world
dist.all_gather_object(out, (b, local.tolist()))
t = torch.tensor([float(rank + 1)]); dist.all_reduce(t, op=dist.ReduceOp.MAX)   # max-over-ranks timing pattern
if rank == 0:
    merged = [v for _, vals in sorted(out) for v in vals]
    ref = [int(np.unpackbits(desc[i] ^ desc[i + 1]).sum()) for i in range(n_items)]
    assert total == n_items and merged == ref and t.item() == world, (total, t.item())
    print("OK")
dist.destroy_process_group()
"""


def test_two_rank_sharding_gloo(api, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), WORLD_SIZE="2"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0]


def _run_bench(args, env_extra=None, timeout=300):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_spawns_its_own_ranks(api):
    if api.device_count() > 0:
        pytest.skip("checks the launcher up to the point where a GPU is required")
    """`python bench.py --gpus 2` (no torchrun) must itself start 2 ranks that form a process group; here (no GPU) every rank
    stops at the device check, after rendezvous, reporting the world size and its block of the global batch (--strong)."""
    cp = _run_bench(["--gpus", "2", "--dist-backend", "gloo", "--batch", "3", "--distinct", "1", "--gen-procs", "1", "--strong", "--verify", "0"])
    assert cp.returncode != 0
    err = cp.stderr
    assert "rank 0 of 2" in err and "rank 1 of 2" in err, err
    assert "process group of 2 ranks is up" in err
    assert "block of 2 pairs from pair 0" in err and "block of 1 pairs from pair 2" in err, err  # shard_range(3, r, 2)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    cp = _run_bench(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=60)
    assert cp.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in cp.stderr


def test_bench_has_no_local_import_shadowing_a_module_import():
    """`import x` inside main() makes x a local of the whole function: an earlier use of the module-level x then raises
    UnboundLocalError, but only on the legs that run on a GPU box (this cost a collection run in round 2)."""
    import ast
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    tree = ast.parse(src)
    top = {(a.asname or a.name).split(".")[0] for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) for a in n.names}
    bad = [(f.name, (a.asname or a.name), n.lineno) for f in ast.walk(tree) if isinstance(f, ast.FunctionDef)
           for n in ast.walk(f) if isinstance(n, (ast.Import, ast.ImportFrom)) for a in n.names
           if (a.asname or a.name).split(".")[0] in top]
    assert not bad, bad
