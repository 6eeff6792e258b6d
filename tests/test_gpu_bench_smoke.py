"""bench.py end to end on the GPU box with small legs: GPUTEST goes red before BENCH can (round 2 and round 4 both lost their
driver bench to a bug that only ran on a GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "GFS_BENCH_CHILD", "GFS_BENCH_NO_SUPERVISOR")}
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
    return cp, lines


@pytest.mark.gpu
def test_bench_small_run_prints_one_line_with_roofline_and_cpu_baseline(gpu_api):
    cp, lines = _run(["--steps", "2", "--warmup", "1", "--batch", "16", "--distinct", "4", "--prime", "1", "--cpu-sample", "8", "--verify", "4"], 1500)
    assert cp.returncode == 0, cp.stderr[-3000:]
    assert len(lines) == 1, cp.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "frames/s"
    r, c = d["roofline"], d["cpu_baseline"]
    assert r and "error" not in r and r["bound"] == "hbm" and r["frac"] > 0 and r["achieved"] > 0 and r["peak"] == 8000.0, r
    # round 6: the traffic of the roofline's kernel is measured by the run itself (two rocprofv3 --pmc child runs, the last leg); the
    # committed figure, if there is one for this batch size, rides along
    live = r.get("traffic_live")
    assert live and "error" not in live, live
    assert live["traffic"] is not None and live["traffic"] > 0 and live["launches_counted"] >= 1, live
    assert r["traffic"] == live["traffic"] and r["traffic_source"].startswith("measured by this run"), r["traffic_source"]
    assert c and "error" not in c and c["value"] > 0 and c["kind"] == "port" and c["cores"] >= 1, c
    assert "side_legs_incomplete" not in d and "skipped_legs" not in d, d.get("side_legs_incomplete") or d.get("skipped_legs")
    assert d["verify"]["verified_pairs"] == d["verify"]["checked_pairs"] == 4, d["verify"]
    for leg in ("optical_flow", "orb_only", "lba", "c3", "c4_shard", "single_stream", "h2d_inclusive"):
        assert leg in d, leg
        assert "error" not in d[leg], (leg, d[leg])
    assert "side_figures_error" not in d, d.get("side_figures_error")


@pytest.mark.gpu
def test_bench_survives_a_side_leg_that_raises(gpu_api, tmp_path):
    """A broken profiles/ directory (GFS_BENCH_PROFILES_DIR points at partial files) must cost `traffic`, nothing else."""
    (tmp_path / "r99z_pmc_traffic_serial.json").write_text(json.dumps({"_batch_pairs": 16, "k_gicp_linearize": {"fetch_kb_per_step": 5.0}}))
    os.environ["GFS_BENCH_PROFILES_DIR"] = str(tmp_path)
    try:
        cp, lines = _run(["--steps", "2", "--warmup", "1", "--batch", "16", "--distinct", "4", "--prime", "1", "--no-cpu-baseline", "--no-extras", "--no-klt", "--verify", "0"], 600)
    finally:
        del os.environ["GFS_BENCH_PROFILES_DIR"]
    assert cp.returncode == 0, cp.stderr[-3000:]
    d = json.loads(lines[-1])
    assert len(lines) == 1 and d["roofline"]["frac"] > 0 and d["roofline"]["traffic"] is None


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu(gpu_api):
    """The N > 1 path of bench.py (what the driver's SCALE run launches, there one rank per GPU over RCCL): here two ranks share GPU 0
    over gloo -- rendezvous, per-rank scenes, barrier + max-over-ranks timing, the `strong` object (configs[3]: one batch cut over the
    ranks) and ONE JSON line from rank 0 with rc 0."""
    cp, lines = _run(["--gpus", "2", "--dist-backend", "gloo", "--all-ranks-device0", "--batch", "32", "--lanes", "2", "--steps", "3", "--warmup", "1",
                      "--prime", "1", "--gen-procs", "8", "--verify", "0"], 900)
    assert cp.returncode == 0, cp.stderr[-3000:]
    assert len(lines) == 1, cp.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak" and d["config"]["global_batch_pairs"] == 64
    assert d["roofline"] and "error" not in d["roofline"] and d["roofline"]["frac"] > 0
    s = d["strong"]
    assert s["scaling"] == "strong" and s["value"] > 0 and s["global_batch_pairs"] == 32 and s["pairs_per_gpu"] == 16
    assert d["cpu_baseline"] is None  # rank 0 at N = 1 only
