#!/usr/bin/env python3
"""bench.py — front-end frames/sec (ORB + BF match + GMS + GICP) on synthetic 640x480 RGB-D, BASELINE.json config C2.

One "step" = one pass of the hot path over one batch of B independent frame pairs already resident in HBM:
    ORB extract (1000 feats, 8 levels) of the B current frames
 -> brute-force Hamming match of the B previous-frame descriptor sets against them + GMS filter (SearchWithGMS)
 -> GICP (voxel 0.02, max-corr 0.1, LM <= 20x10) of the B (previous, current) cloud pairs, both clouds preprocessed.
`value` = frames/s = N * B * steps / max-over-ranks wall time.  Multi-GPU: one process per GPU, batches are
independent (no RCCL collective on the data path); torch.distributed is used only for the barrier / max reduction.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, HIP-event timed on its launch stream)
and "cpu_baseline" (the CPU oracle timed on a bounded sample of the same workload; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)

# SURVEY.md §8(d) algorithmic bytes of each kernel for ONE STEP (whole batch); bench divides by launches per step
def algorithmic_bytes_per_step(kernel, ctx):
    P0, P, K = ctx["P0"], ctx["P"], ctx["K"]  # level-0 pixels, pyramid pixels, keypoints per frame
    B = ctx["B"]
    table = {
        "k_fast_cells": B * (P + 4 * ctx["cands"]),                 # P read + 4 B per packed candidate
        "k_pyr_area_lds": B * (P0 + (P - P0)),                      # P0 read + levels written (a level is re-read from LDS)
        "k_pyr_area_hbm": B * (P0 + (P - P0) + (P - ctx["p_last"])),  # P0 read + levels written + levels re-read
        "k_blur7": B * 2 * P,
        "k_orient_brief": B * K * (31 * 31 + 37 * 37 + 60),         # K x (31x31 + 37x37) read + 60 B out
        "k_bf_hamming": B * (32 * 2 * K + 8 * K),
        "k_gms": B * (2 * 28 * K + 5 * K),                          # both key-point sets + match indices in, mask out
        # 320 B per source point per linearisation.  Since round 6 a pending trial's error is evaluated in the same pass (it needs the
        # point under the trial pose and the previous correspondence, which the linearisation loads anyway): only the error-ONLY passes
        # -- a pair's last trial, one per pair -- are counted on top (136 B a point); the errors that ride along are not
        "k_gicp_linearize": 320.0 * ctx["lin_points"] + 136.0 * ctx.get("last_err_points", 0),
        "k_knn_cov": (10 * 32 + 160) * ctx["ds_points"],            # 10-NN gather + covariance write, both clouds
        "k_radix_sort": 2 * 12 * ctx["ds_points"],                    # the cell sort: (8 B key + 4 B index) read + written
        "k_cell_sort_lds": 2 * 12 * ctx["ds_points"],
        "k_voxel_qsort_top_reg": 2 * 12 * ctx["in_points"],            # the voxel sort (quick_sort_omp replica): keys + indices in, out
        "k_voxel_qsort_top_lds": 2 * 12 * ctx["in_points"],            # (everything between the read and the write stays in LDS)
        "k_voxel_qsort_top": 2 * 12 * ctx["in_points"],
        "k_voxel_qsort_leaf": 2 * 12 * ctx["in_points"],
        "k_voxel_reduce": (16 + 12) * ctx["in_points"] + 32 * ctx["ds_points"],
        "k_voxel_keys": (16 + 12) * ctx["in_points"],
        "k_cell_build": (32 + 12 + 32) * ctx["ds_points"],
        "k_grid_fill": 12 * ctx["ds_points"],
        "k_cand_pack": B * 8 * ctx["cands"],
    }
    return table.get(kernel)


def select_committed_traffic(profiles_dir, kernel, launches_per_step, batch_pairs, workload="c2"):
    """`roofline.traffic` from a COMMITTED rocprofv3 PMC summary (profiles/r*_pmc_traffic_serial.json; written on another box by
    profiles/collect.sh + summarize.py).  It is not something this run measured, so it is optional by construction: the newest file
    that (a) parses, (b) was collected at this batch size and (c) holds BOTH the FETCH_SIZE and the WRITE_SIZE pass for `kernel` is
    used; anything else gives traffic = null with the reason in `note`.  Never raises."""
    import glob
    none = dict(traffic=None, traffic_raw_counters=None, source=None, note=None)
    try:
        if workload != "c2":
            return dict(none, note="no committed PMC profile for this workload")
        files = sorted(glob.glob(os.path.join(profiles_dir, "r*_pmc_traffic_serial.json")), reverse=True)  # newest round first
        skipped = []
        for f in files:
            try:
                allk = json.load(open(f))
            except Exception as e:
                skipped.append(f"{os.path.basename(f)}: unreadable ({type(e).__name__})")
                continue
            if not isinstance(allk, dict) or allk.get("_batch_pairs") != batch_pairs:
                skipped.append(f"{os.path.basename(f)}: other batch size")
                continue
            t = allk.get(kernel)
            if not isinstance(t, dict):
                skipped.append(f"{os.path.basename(f)}: kernel not in file")
                continue
            fk, wk = t.get("fetch_kb_per_step"), t.get("write_kb_per_step")
            if not isinstance(fk, (int, float)) or not isinstance(wk, (int, float)) or not launches_per_step:
                skipped.append(f"{os.path.basename(f)}: incomplete (a PMC pass is missing)")
                continue
            # calibration (profiles/r03_calibration.json, kernels with known byte counts): FETCH_SIZE reports HALF the bytes of wide
            # streaming reads (0.500), the full 64-byte line for a 32-byte gather that misses the caches (2.02 x the 32 bytes asked
            # for), nothing for gathers served by the L2; WRITE_SIZE is exact (1.000)
            return dict(traffic=int((2.0 * fk + wk) * 1024 / launches_per_step),
                        traffic_raw_counters=int((fk + wk) * 1024 / launches_per_step),
                        source=f"committed profiles/{os.path.basename(f)} (collected on another box by profiles/collect.sh), NOT measured by this run",
                        note="rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes, same batch, 1 lane, serial), (2 x FETCH_SIZE + WRITE_SIZE) "
                             "KB*1024 per step / launches per step; the factor 2 is the measured under-count of streaming reads "
                             "(profiles/r03_calibration.json) and over-counts the share of cache-missing gathers, which the counter reports in "
                             "full: an upper bound of the HBM-side bytes" + (f"; skipped: {'; '.join(skipped)}" if skipped else ""))
        return dict(none, note="no complete committed PMC profile for this kernel and batch size" + (f" ({'; '.join(skipped[:4])})" if skipped else ""))
    except Exception as e:
        return dict(none, note=f"traffic unavailable: {type(e).__name__}: {e}")


def measure_traffic_live(kernel, batch_pairs, nlanes, timeout_s, alg_bytes_per_launch=None):
    """`roofline.traffic` measured by THIS run: FETCH_SIZE and WRITE_SIZE of `kernel`, per launch, from two child runs of this script
    under rocprofv3 (one counter a pass: they do not fit one, and no trace domain beside them).  Same batch size and lane shape as the
    timed run, one warm-up + one timed step + the two profiling steps, 64 distinct scenes tiled over the batch (a child renders its own).
    (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024, as select_committed_traffic() forms it.  The child's scenes converge after other numbers of
    iterations than the parent's 512 (fewer, larger launches), so the figure is carried over per ALGORITHMIC byte: the child's counter
    bytes per step over the algorithmic bytes per step of ITS OWN roofline object, times the parent's algorithmic bytes per launch.
    Never raises; a child is killed with its process group when it outlives its share of timeout_s."""
    import csv, glob, shutil, signal, subprocess, tempfile
    out = dict(traffic=None, traffic_raw_counters=None, source=None)
    try:
        rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
        if not rp:
            return dict(out, note="rocprofv3 not found")
        kb, launches, child_alg_step = {}, {}, None
        child_steps = 1 + 1 + 2  # warm-up + timed + the roofline leg's two profiling steps: every one the same pass over the batch
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix=f"gfs_pmc_{counter}_", dir="/tmp")
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GFS_BENCH_CHILD")}
            env.update(TMPDIR="/tmp", GFS_BENCH_NO_SUPERVISOR="1", GFS_BENCH_LIVE_TRAFFIC="0")
            # (only the roofline's kernel is instrumented: every other dispatch runs at full speed)
            cmd = [rp, "--pmc", counter, "--kernel-include-regex", kernel, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "1",
                   "--warmup", "1", "--batch", str(batch_pairs), "--lanes", str(nlanes), "--distinct", "64", "--no-cpu-baseline", "--no-extras",
                   "--no-klt", "--verify", "0", "--prime", "0"]
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, start_new_session=True, text=True)
            try:
                child_out, _ = p.communicate(timeout=timeout_s / 2)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except Exception:
                    p.kill()
                p.wait()
                shutil.rmtree(d, ignore_errors=True)
                return dict(out, note=f"the {counter} pass did not finish in {timeout_s / 2:.0f} s (killed)")
            try:  # the child's own line: algorithmic bytes of the kernel per step of ITS batch
                cr = json.loads([ln_ for ln_ in child_out.splitlines() if ln_.startswith("{")][-1])["roofline"]
                if cr["kernel"] == kernel:
                    child_alg_step = float(cr["algorithmic_bytes_per_launch"]) * float(cr["launches_per_step"])
            except Exception:
                pass
            tot, ids = 0.0, set()
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        tot += float(r["Counter_Value"])
                        ids.add(r["Dispatch_Id"])
            shutil.rmtree(d, ignore_errors=True)
            if not ids:
                return dict(out, note=f"the {counter} pass (rc {p.returncode}) left no rows for {kernel}")
            kb[counter], launches[counter] = tot, len(ids)
        f_l, w_l = kb["FETCH_SIZE"] / launches["FETCH_SIZE"], kb["WRITE_SIZE"] / launches["WRITE_SIZE"]
        res = dict(fetch_kb_per_launch_child=round(f_l, 1), write_kb_per_launch_child=round(w_l, 1), launches_counted=launches["FETCH_SIZE"],
                   child_launches_per_step=launches["FETCH_SIZE"] / child_steps)
        if child_alg_step and alg_bytes_per_launch:
            up_step = (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024 / child_steps
            raw_step = (kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024 / child_steps
            res.update(traffic=int(up_step / child_alg_step * alg_bytes_per_launch), traffic_raw_counters=int(raw_step / child_alg_step * alg_bytes_per_launch),
                       traffic_per_algorithmic_byte=round(up_step / child_alg_step, 4), raw_counters_per_algorithmic_byte=round(raw_step / child_alg_step, 4),
                       scaled="per algorithmic byte: the child's counter bytes per step / its algorithmic bytes per step x this run's algorithmic bytes per launch")
        else:  # (no usable line of the child: its own launches as they are)
            res.update(traffic=int((2.0 * f_l + w_l) * 1024), traffic_raw_counters=int((f_l + w_l) * 1024), scaled="per launch of the child run")
        return dict(res,
                    source=f"measured by this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE on {kernel} (two child runs of bench.py after the timed "
                           f"region, batch {batch_pairs}, {nlanes} lanes, 64 distinct scenes tiled), 2 x FETCH_SIZE + WRITE_SIZE",
                    note="the factor 2 is the measured under-count of streaming reads (profiles/r03_calibration.json) and over-counts the share of "
                         "cache-missing gathers: an upper bound of the HBM-side bytes")
    except Exception as e:
        return dict(out, note=f"live traffic unavailable: {type(e).__name__}: {e}")


def supervise(argv, deadline_s):
    """N = 1 only: run the bench in a CHILD process that prints the JSON line again after every side leg, keep the newest one and
    print exactly ONE line.  A side leg that crashes the process (GPU fault, segfault in a library) or hangs past the deadline then
    costs its own object, not the measured headline; rc is 0 whenever the timed headline was produced."""
    import subprocess
    env = dict(os.environ, GFS_BENCH_CHILD="1")
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    last = [None]

    def forward(signum, frame):
        # The driver (or an outer `timeout`) stopping the parent must not leave the child running on the GPU -- and must not cost the
        # headline either: the newest line the child produced is printed before leaving, rc 0 when there is one.
        p.kill()
        if last[0] is None:
            sys.exit(128 + signum)
        try:
            o = json.loads(last[0])
            o["side_legs_incomplete"] = f"stopped by signal {signum} after the last completed leg"
            if o.get("verify_requested") and "verify" not in o:
                o["verified"] = False
            print(json.dumps(o), flush=True)
        except Exception:
            print(last[0], flush=True)
        os._exit(0)

    import signal
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sg, forward)

    def reader():
        for ln in p.stdout:
            ln = ln.strip()
            if ln.startswith("{"):
                last[0] = ln
            elif ln:
                print(ln, file=sys.stderr)

    th = threading.Thread(target=reader, daemon=True)
    th.start()
    died = None
    try:
        rc = p.wait(timeout=deadline_s)
        if rc != 0:
            died = f"bench child exited with rc {rc} after the last completed leg"
    except subprocess.TimeoutExpired:
        p.kill()  # the exact PID this function started
        p.wait()
        died = f"bench child passed the {deadline_s:.0f} s deadline in a side leg and was stopped"
    th.join(timeout=10)
    if last[0] is None:
        sys.exit(f"bench.py: no headline was produced ({died or 'child printed nothing'})")
    if died:
        try:
            o = json.loads(last[0])
            o["side_legs_incomplete"] = died
            if o.get("verify_requested") and "verify" not in o:
                o["verified"] = False  # a consumer that keys on `value` alone can see that the oracle check did not finish
            last[0] = json.dumps(o)
        except Exception:
            pass
    print(last[0], flush=True)
    sys.exit(0)


def _gen_one(a):
    from geoflowslam_amd import synth
    seed, width, height, stride = a
    try:  # one BLAS thread per worker process: the pool already uses every core
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            return synth.frame_pair(seed, width, height, stride)
    except ImportError:
        pass
    p = synth.frame_pair(seed, width, height, stride)
    # depth maps are kept as float32 (what Tracking::GrabImageRGBD hands to the Frame constructor)
    return p


def gen_pairs(seeds, width, height, stride, procs):
    """Distinct synthetic scenes, rendered by worker PROCESSES (the ray caster is numpy-bound; called before any HIP state
    exists in this process, so forking is safe)."""
    seeds = list(seeds)
    if procs <= 1 or len(seeds) <= 2:
        return [_gen_one((s, width, height, stride)) for s in seeds]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(procs, len(seeds))) as pool:
        return pool.map(_gen_one, [(s, width, height, stride) for s in seeds], chunksize=1)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-execute this script as N ranks (one per GPU, rank 0's stdout is the
    JSON line), exactly as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` would."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    sys.exit(max(abs(rc) for rc in rcs))


def main():
    # The lanes drive 2 HIP streams each; with the default of 4 hardware queues they would share queues and serialise.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frame pairs per GPU per step")
    ap.add_argument("--distinct", type=int, default=0,
                    help="distinct synthetic scenes per GPU, tiled to --batch (default 0 = every pair of the batch is its own scene)")
    ap.add_argument("--strong", action="store_true",
                    help="BASELINE.json configs[3]: ONE global batch of --batch pairs cut into contiguous blocks across the ranks "
                         "(geoflowslam_amd.shard.shard_range; 512 -> 64 per GPU at 8 GPUs), scaling = strong; default: every rank "
                         "processes its own --batch pairs (weak)")
    ap.add_argument("--verify", type=int, default=16,
                    help="after the timed region, check this many pairs of the TIMED batch against the CPU oracle (key points, descriptors, "
                         "matches, GMS mask bit-exact; GICP pose <= 1e-5): reported as verified_pairs (0 = off)")
    ap.add_argument("--gen-procs", type=int, default=0, help="worker processes for the scene rendering (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-join", action="store_true",
                    help="join all lanes after every pass (default: every (lane, half) chain runs its K passes back to back)")
    ap.add_argument("--prime", type=int, default=10, help="untimed set-up passes before the W warm-up steps (code objects, clocks)")
    ap.add_argument("--no-klt", action="store_true", help="skip the optical-flow side measurement")
    ap.add_argument("--no-extras", action="store_true", help="skip the ORB-only and LBA side measurements")
    ap.add_argument("--gicp-stream", action="store_true",
                    help="experiment, not the headline metric: GICP through gfs_gicp_align_next (the previous call's preprocessed "
                         "source cloud is the target, as in a live stream) instead of preprocessing both clouds per pair")
    ap.add_argument("--serial", action="store_true", help="run ORB+match and GICP back to back on one stream")
    ap.add_argument("--lanes", type=int, default=4, help="independent slices of the batch processed concurrently per GPU")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for 1-GPU dry runs)")
    ap.add_argument("--all-ranks-device0", action="store_true", help="dry-run aid: every rank uses GPU 0 (needs --dist-backend gloo)")
    ap.add_argument("--cpu-sample", type=int, default=32, help="frame pairs in the CPU-baseline sample")
    ap.add_argument("--workload", choices=["c2", "c3"], default="c2",
                    help="c2 = BASELINE.json configs[1] (640x480, 1000 features, ~19k-pt clouds; the metric's configuration); "
                         "c3 = configs[2] (1280x720, 2000 features, ~37k-pt clouds; use --batch 32)")
    ap.add_argument("--leg-budget-s", type=float, default=420.0,
                    help="side legs (everything after roofline + cpu_baseline) are skipped once the run is this old: the line must reach the driver")
    ap.add_argument("--deadline-s", type=float, default=1500.0, help="N = 1: the supervising parent stops a hung child after this long and prints the newest line")
    ap.add_argument("--live-traffic-timeout-s", type=float, default=300.0,
                    help="both rocprofv3 --pmc child runs of the roofline.traffic leg together (GFS_BENCH_LIVE_TRAFFIC=0 skips the leg)")
    ap.add_argument("--no-supervisor", action="store_true", help="N = 1: run in this process (no child), side legs guarded by try/except only")
    args = ap.parse_args()
    t_start = time.perf_counter()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)  # does not return
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.no_supervisor and "GFS_BENCH_CHILD" not in os.environ \
            and "GFS_BENCH_NO_SUPERVISOR" not in os.environ:
        supervise(sys.argv[1:], args.deadline_s)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without a launcher)")

    # ---- synthetic inputs first (numpy only: the renderer forks worker processes, which must happen before HIP is initialised)
    from geoflowslam_amd import synth
    from geoflowslam_amd.shard import shard_range
    W, H, STRIDE, NF, NL = (640, 480, 4, 1000, 8) if args.workload == "c2" else (1280, 720, 5, 2000, 8)
    if args.strong:
        g0_, g1_ = shard_range(args.batch, rank, world)  # this rank's block of the global batch
        B, first_pair = g1_ - g0_, g0_
    else:
        B, first_pair = args.batch, rank * args.batch
    assert B > 0, "more ranks than pairs"
    nd = B if args.distinct <= 0 else max(1, min(args.distinct, B))
    seed0 = 1000 + first_pair
    ncpu = os.cpu_count() or 1
    gen_procs = args.gen_procs or max(1, min(96, ncpu // max(world, 1)))
    t_gen = time.perf_counter()
    pairs = gen_pairs(range(seed0, seed0 + nd), W, H, STRIDE, gen_procs)
    t_gen = time.perf_counter() - t_gen

    import torch
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.all_ranks_device0:
            local_rank = 0
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)
    if not torch.cuda.is_available():
        sys.exit(f"bench.py needs an MI355X: rank {rank} of {world} has no GPU"
                 + (f" (process group of {dist.get_world_size()} ranks is up, block of {B} pairs from pair {first_pair})" if world > 1 else ""))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from geoflowslam_amd import api
    sel = [i % nd for i in range(B)]
    npts = max(max(len(p["cloud0"]), len(p["cloud1"])) for p in pairs)
    SP = (npts + 1023) // 1024 * 1024

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    gray0 = to_dev(np.stack([pairs[i]["gray0"] for i in sel]))
    gray1 = to_dev(np.stack([pairs[i]["gray1"] for i in sel]))
    c0 = np.zeros((B, SP, 4), np.float32)
    c1 = np.zeros((B, SP, 4), np.float32)
    n0 = np.zeros(B, np.int32)
    n1 = np.zeros(B, np.int32)
    for b, i in enumerate(sel):
        a, bb = pairs[i]["cloud0"], pairs[i]["cloud1"]
        c0[b, :len(a)] = a
        c1[b, :len(bb)] = bb
        n0[b], n1[b] = len(a), len(bb)
    d_c0, d_c1, d_n0, d_n1 = to_dev(c0), to_dev(c1), to_dev(n0), to_dev(n1)

    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    class Lane:
        """One independent slice of the batch with its own handles, HIP streams and host threads.  Several lanes in
        flight keep the GPU busy while another lane's host work (quadtree, LM convergence polling) runs."""

        def __init__(self, b0, b1):
            self.b0, self.n = b0, b1 - b0
            n = self.n
            self.ext = api.ORBextractor(NF, 1.2, NL, 20, 7, max_rows=H, max_cols=W, max_batch=n, device=local_rank)
            self.cap = self.ext.cap
            self.mt = api.ORBmatcher(max_query=self.cap, max_train=self.cap, max_batch=n, device=local_rank)
            self.gms = api.GmsMatcher(max_keypoints=self.cap, max_batch=n, device=local_rank)
            self.reg = api.RegistrationGICP(max_points=SP, max_batch=n, device=local_rank)
            self.s1 = torch.cuda.Stream(device=dev)
            self.s2 = torch.cuda.Stream(device=dev)  # (raising the GICP stream's priority measured 15 % slower)
            self.g0, self.g1 = gray0[b0:b1], gray1[b0:b1]
            self.c0, self.c1, self.n0, self.n1 = d_c0[b0:b1], d_c1[b0:b1], d_n0[b0:b1], d_n1[b0:b1]
            # previous-frame features (the "keyframe" side of SearchWithGMS): computed once, kept in HBM
            self.ext.extract_batch_device(self.g0.data_ptr(), n, H, W, (0, 0), self.s1.cuda_stream)
            self.res = self.ext.device_results()
            self.s1.synchronize()
            self.prev_desc = torch.empty(n * self.cap * 32, dtype=torch.uint8, device=dev)
            self.prev_cnt = torch.empty(n, dtype=torch.int32, device=dev)
            self.prev_kps = torch.empty(n * self.cap * 28, dtype=torch.uint8, device=dev)  # cv::KeyPoint layout
            assert hip.hipMemcpy(self.prev_kps.data_ptr(), self.res["kps"], n * self.cap * 28, 3) == 0
            assert hip.hipMemcpy(self.prev_desc.data_ptr(), self.res["desc"], n * self.cap * 32, 3) == 0
            assert hip.hipMemcpy(self.prev_cnt.data_ptr(), self.res["counts"], n * 4, 3) == 0
            self.m_idx = torch.empty(n * self.cap, dtype=torch.int32, device=dev)
            self.m_dist = torch.empty(n * self.cap, dtype=torch.int32, device=dev)
            self.m_mask = torch.empty(n * self.cap, dtype=torch.uint8, device=dev)
            self.m_inl = torch.empty(n, dtype=torch.int32, device=dev)
            self.gicp_out = None

        def orb_and_match(self):
            sp = self.s1.cuda_stream
            self.ext.extract_batch_device(self.g1.data_ptr(), self.n, H, W, (0, 0), sp)
            self.mt.match_batch_device(self.prev_desc.data_ptr(), self.prev_cnt.data_ptr(), self.res["desc"], self.res["counts"],
                                       self.n, self.cap, self.m_idx.data_ptr(), self.m_dist.data_ptr(), sp)
            # ... followed by the GMS filter like every BFMatcher::match of the reference (SearchWithGMS, src/ORBmatcher.cc:744-778)
            self.gms.inlier_mask_batch_device(self.prev_kps.data_ptr(), self.prev_cnt.data_ptr(), self.res["kps"], self.res["counts"],
                                              self.n, self.cap, self.m_idx.data_ptr(), W, H, self.m_mask.data_ptr(),
                                              self.m_inl.data_ptr(), sp)

        def gicp(self):
            if args.gicp_stream and self.gicp_out is not None:
                self.flip = not getattr(self, "flip", False)
                c, nn = (self.c0, self.n0) if self.flip else (self.c1, self.n1)
                self.gicp_out = self.reg.align_next_batch_device(c.data_ptr(), nn.data_ptr(), self.n, SP, None, None,
                                                                 self.s2.cuda_stream, raw=True)
                return
            self.gicp_out = self.reg.align_batch_device(self.c0.data_ptr(), self.n0.data_ptr(), self.c1.data_ptr(),
                                                        self.n1.data_ptr(), self.n, SP, None, None, self.s2.cuda_stream, raw=True)

    nlanes = max(1, min(args.lanes, B))
    lanes = [Lane(*shard_range(B, l, nlanes)) for l in range(nlanes)]
    pool = ThreadPoolExecutor(max_workers=2 * nlanes)
    last = {}
    ext = lanes[0].ext

    def step_serial():  # strictly one module at a time: clean per-kernel timings (profiling pass, --serial)
        for ln in lanes:
            ln.orb_and_match()
            torch.cuda.synchronize()
            ln.gicp()
            torch.cuda.synchronize()

    def step_overlapped():
        # ORB and GICP of a slice are independent, and so are the slices: every lane runs its two halves on two host
        # threads / two HIP streams (ctypes releases the GIL); no data-path synchronisation between lanes.
        futs = [pool.submit(f) for ln in lanes for f in (ln.orb_and_match, ln.gicp)]
        for f in futs:
            f.result()

    step = step_serial if args.serial else step_overlapped

    def run_steps(k, lns=None):
        """k passes of the hot path over the batch.  The two halves of a lane never exchange data, and neither do the lanes, so
        (unless --step-join) every (lane, half) chain runs its k passes back to back on its own host thread and HIP stream: a
        chain does not wait at the end of each pass for the slowest one (whose Levenberg-Marquardt tail leaves the GPU half
        empty).  Every pass is still executed k times in full; the caller brackets the k passes with barriers."""
        if lns is None and (args.serial or args.step_join):
            for _ in range(k):
                step()
            return

        def chain(f):
            for _ in range(k):
                f()

        futs = [pool.submit(chain, f) for ln in (lanes if lns is None else lns) for f in (ln.orb_and_match, ln.gicp)]
        for f in futs:
            f.result()

    def gicp_results():
        return [dict(n_linearize=r.n_linearize, n_error_evals=r.n_error_evals, n_source_ds=r.n_source_ds,
                     n_target_ds=r.n_target_ds, converged=bool(r.converged)) for ln in lanes for r in ln.gicp_out]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not part of the W + K contract: on a fresh box the first passes pay for loading the code objects and ramping the
    # clocks (the narrow one-workgroup-per-cloud kernels run 2-5x slower then); run the pipeline a few times before warm-up
    run_steps(args.prime)
    barrier()
    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_pairs = args.batch if args.strong else world * B
    fps = total_pairs * args.steps / dt
    # ---- N > 1 without --strong: the weak figure above is the contract's line; BASELINE.json configs[3] (ONE batch of --batch pairs cut
    #      over the ranks) is measured right after it, same K / W, and reported beside it as `strong`
    strong = None
    if world > 1 and not args.strong:
        s0_, s1_ = shard_range(args.batch, rank, world)
        nb_ = s1_ - s0_  # this rank's block of the global batch (it takes the block from its own scene set: the blocks are independent)
        if nb_ > 0:
            nl_s = max(1, min(args.lanes, nb_ // 32))  # >= 32 pairs per lane: c4_shard measured 2 lanes of 32 ahead of 4 of 16
            lanes_s = [Lane(*shard_range(nb_, l, nl_s)) for l in range(nl_s)]
            run_steps(max(2, args.prime // 2), lanes_s)
            barrier()
            run_steps(args.warmup, lanes_s)
            barrier()
            t0s = time.perf_counter()
            run_steps(args.steps, lanes_s)
            barrier()
            dts_ = time.perf_counter() - t0s
        else:
            barrier(); barrier(); barrier()
            dts_ = 0.0
        ts_ = torch.tensor([dts_], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(ts_, op=dist.ReduceOp.MAX)
        dts_ = float(ts_.item())
        strong = dict(value=round(args.batch * args.steps / dts_, 2), unit="frames/s", scaling="strong", ms_per_step=round(dts_ / args.steps * 1e3, 3),
                      global_batch_pairs=args.batch, pairs_per_gpu=nb_, lanes_per_gpu=nl_s if nb_ > 0 else 0,
                      note=f"BASELINE.json configs[3]: ONE batch of {args.batch} pairs cut into contiguous blocks over {world} GPUs (shard_range), same K and W, "
                           "barrier + max over ranks; no collective on the data path")
        if nb_ > 0:
            del lanes_s
    free_b, total_b = torch.cuda.mem_get_info(dev)  # everything the path holds in HBM (workspaces are sized once, at handle creation)
    hbm_used_gb = round((total_b - free_b) / 2**30, 1)

    def assemble():
        g = g_head
        out = {
            "metric": "front-end frames/sec (ORB+match+GICP) on 640x480 RGBD",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": (dist.get_world_size() if world > 1 else 1), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "u8/int32 (ORB, match) + f64 (GICP)", "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[1]: 640x480 RGBD frame pair, ORB extract (1000 feats, 8 levels) "
                                    "+ BF Hamming match + GMS filter + GICP on ~19k-pt clouds (stride-4 depth grid)") if args.workload == "c2" else
                                   ("BASELINE.json configs[2] (NOT the metric's configuration): 1280x720 RGBD frame pair, ORB extract (2000 feats, "
                                    "8 levels) + BF Hamming match + GMS filter + GICP on ~37k-pt clouds (stride-5 depth grid)"),
                       "batch_pairs_per_gpu": B, "lanes_per_gpu": nlanes, "hbm_in_use_gb": hbm_used_gb,
                       "passes": "joined after every pass" if (args.step_join or args.serial) else "K passes per lane chain, joined once", "distinct_scenes_per_gpu": nd, "scene_seeds": f"{seed0}..{seed0 + nd - 1} (rank 0)",
                       "scene_render_s": round(t_gen, 1), "global_batch_pairs": total_pairs,
                       "parallelism": (f"one global batch of {args.batch} pairs cut into contiguous blocks over {world} rank(s), no collective"
                                       if args.strong else f"frames sharded x{world}, no collective"),
                       "gicp_mean_outer_iterations": round(float(np.mean([r["n_linearize"] for r in g])), 2),
                       "gicp_mean_error_evals": round(float(np.mean([r["n_error_evals"] for r in g])), 2),
                       "gicp_converged_frac": round(float(np.mean([r["converged"] for r in g])), 3)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if args.gicp_stream:
            out["config"]["workload"] += " [EXPERIMENT --gicp-stream: target preprocessing reused from the previous call]"
        verify_wanted = rank == 0 and args.verify > 0 and not args.no_cpu_baseline and not args.gicp_stream
        if verify_wanted:
            out["verify_requested"] = int(args.verify)
        if verify:
            out["verified_pairs"] = verify.get("verified_pairs")
            out["verify"] = verify
            out["verified"] = bool(verify.get("checked_pairs")) and verify.get("verified_pairs") == verify.get("checked_pairs")
        elif verify_wanted and final_line[0]:
            out["verified"] = False  # the check against the oracle was asked for and did not produce a result
        if strong:
            out["strong"] = strong
        if h2d:
            out["h2d_inclusive"] = h2d
        if klt:
            out["optical_flow"] = klt
        out.update(extras)
        # the three side figures a reader looks for first, at the top level (the full objects stay where they are)
        try:
            if isinstance(extras.get("c4_shard"), dict) and "efficiency_vs_batch_512" in extras["c4_shard"]:
                out["c4_shard_efficiency_vs_batch_512"] = extras["c4_shard"]["efficiency_vs_batch_512"]
            if isinstance(extras.get("single_stream"), dict) and "median_ms" in extras["single_stream"]:
                out["single_stream_median_ms"] = extras["single_stream"]["median_ms"]
            if isinstance(extras.get("lba"), dict) and "ms_per_window" in extras["lba"]:
                out["lba_ms_per_window"] = extras["lba"]["ms_per_window"]
        except Exception:
            pass
        if skipped_legs:
            out["skipped_legs"] = dict(legs=list(skipped_legs), reason=f"run older than --leg-budget-s {args.leg_budget_s:.0f} s")
        if cpu and "value" in cpu:
            out["gpu_over_cpu"] = round(fps / cpu["value"], 2)
            out["gpu_over_cpu_all_cores"] = round(fps / cpu["all_cores"]["value"], 2)
        return out

    # ---- from here on everything is a side leg around a headline that is already measured: every leg is guarded, the line is
    #      (re)assembled by emit() after each one (the N = 1 supervisor keeps the newest), legs are skipped when the run is too old
    cpu = klt = verify = h2d = None
    extras = {}
    skipped_legs = []
    g_head = gicp_results() if rank == 0 else []

    def over_budget(leg):
        if time.perf_counter() - t_start > args.leg_budget_s:
            skipped_legs.append(leg)
            return True
        return False

    final_line = [False]

    def emit(final=False):
        if rank != 0 or not (final or "GFS_BENCH_CHILD" in os.environ):
            return
        final_line[0] = final
        try:
            line = json.dumps(assemble(), default=str)
        except Exception as e:  # even a bug in the assembly must not cost the measured headline
            line = json.dumps({"metric": "front-end frames/sec (ORB+match+GICP) on 640x480 RGBD", "value": round(fps, 2), "unit": "frames/s",
                               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                               "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
                               "dtype": "u8/int32 (ORB, match) + f64 (GICP)", "data": "synthetic", "roofline": None, "cpu_baseline": None,
                               "assemble_error": f"{type(e).__name__}: {e}"})
        print(line, flush=True)

    # ---- dominant-kernel roofline: per-kernel HIP-event timing on the launch stream (extra, untimed steps, run one
    #      module at a time so that a kernel's events do not include waiting for kernels of other streams)
    roofline = None
    kern = {}
    emit()  # the headline alone, before any side leg

    def guarded(leg):
        """The headline was measured before any side leg runs: a failing leg is recorded as {"error": ...} in its own object and the
        JSON line is still printed with rc 0."""
        try:
            return leg()
        except Exception as e:
            import traceback
            return dict(error=f"{type(e).__name__}: {e}", where=traceback.format_exc(limit=3).strip().splitlines()[-3:])

    def committed_traffic(name, lps):
        """HBM-side bytes per launch of kernel `name` from the newest COMPLETE committed PMC profile (profiles/r*_pmc_traffic_serial.json, written on
        another box by profiles/collect.sh): NOT measured by this run, optional, never allowed to cost the line."""
        return select_committed_traffic(os.environ.get("GFS_BENCH_PROFILES_DIR", os.path.join(ROOT, "profiles")), name, lps, B, args.workload)

    def roofline_leg():
        nonlocal kern
        api.profile_reset()
        api.profile_enable(True)
        nprof = 2
        for _ in range(nprof):
            step_serial()
        torch.cuda.synchronize()
        kern = api.profile_report()
        api.profile_enable(False)
        g = gicp_results()
        counts = torch.cat([ln.prev_cnt.cpu() for ln in lanes])
        lin_pts = sum(r["n_linearize"] * r["n_source_ds"] for r in g)
        err_pts = sum(r["n_error_evals"] * r["n_source_ds"] for r in g)
        lv = [ext.level_size(l) for l in range(NL)]
        P = sum(r * c for r, c in lv)
        ctx = dict(B=B, P0=W * H, P=P, p_last=lv[-1][0] * lv[-1][1], K=float(counts.float().mean()),
                   cands=float(np.mean([sum(len(ext.candidates(l, b)[0]) for l in range(NL)) for b in range(min(lanes[0].n, 4))])),
                   lin_points=lin_pts, err_points=err_pts, last_err_points=sum(r["n_source_ds"] for r in g if r["n_error_evals"] > 0),
                   ds_points=sum(r["n_source_ds"] + r["n_target_ds"] for r in g), in_points=int(n0.sum() + n1.sum()))
        tot = sum(v[0] for v in kern.values())
        # the dominant kernel = the one with the largest share of kernel time that has an algorithmic byte model (SURVEY.md 8d has none for
        # the latency-bound helpers -- the heap-sort replay, the quadtree: at tiny batches one of them can lead; it is named beside)
        by_time = sorted(kern, key=lambda k: -kern[k][0])
        name = next((k for k in by_time if algorithmic_bytes_per_step(k, ctx)), by_time[0])
        dominant_overall = by_time[0]
        ms, launches = kern[name]
        avg_s = ms / launches / 1e3
        lps = launches / nprof
        ab_step = algorithmic_bytes_per_step(name, ctx)
        ab = ab_step / lps if ab_step else None
        ach = ab / avg_s / 1e9 if ab else None
        tr = committed_traffic(name, lps)
        traffic, traffic_raw, traffic_note, traffic_source = tr["traffic"], tr["traffic_raw_counters"], tr["note"], tr["source"]
        return dict(bound="hbm", kernel=name, achieved=round(ach, 2) if ach else None, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(ach / HBM_PEAK_GBS, 5) if ach else None, traffic=traffic, traffic_is_upper_bound=traffic is not None,
                        traffic_raw_counters=traffic_raw, traffic_source=traffic_source, traffic_note=traffic_note,
                        avg_launch_us=round(avg_s * 1e6, 2), launches_per_step=lps,
                        algorithmic_bytes_per_launch=int(ab) if ab else None,
                        share_of_gpu_kernel_time=round(ms / tot, 3), largest_kernel_by_time=dominant_overall,
                        kernels_ms_per_step={k: round(v[0] / nprof, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0])})

    if rank == 0:
        roofline = guarded(roofline_leg)
        emit()

    # ---- verification of the TIMED batch's outputs against the CPU oracle (a sample; after the timed region)
    def verify_leg():
        from oracle import oracle as O
        O.lib()
        step_overlapped()  # the same pass as the timed ones, outputs left in HBM
        torch.cuda.synchronize()
        picks = sorted(set(int(round(x)) for x in np.linspace(0, B - 1, min(args.verify, B))))
        lane_of = {}
        for li, ln in enumerate(lanes):
            for k in range(ln.n):
                lane_of[ln.b0 + k] = (ln, k)

        def check(b):
            ln, k = lane_of[b]
            p = pairs[sel[b]]
            orc = O.OrbOracle(NF, 1.2, NL, 20, 7)
            _, k0, d0 = orc.extract(p["gray0"])
            _, k1, d1 = orc.extract(p["gray1"])
            _, gk1, gd1 = ln.ext.fetch(k)
            ok_orb = len(gk1) == len(k1) and bool((gk1 == k1).all()) and bool((gd1 == d1).all())
            ti, di = O.bf_match(d0, d1)
            nq = len(ti)
            gi = ln.m_idx.view(ln.n, ln.cap)[k, :nq].cpu().numpy()
            gd = ln.m_dist.view(ln.n, ln.cap)[k, :nq].cpu().numpy()
            ok_match = bool(np.array_equal(gi, ti) and np.array_equal(gd, di))
            mo, no = O.gms_inlier_mask(k0, (W, H), k1, (W, H), np.arange(nq, dtype=np.int32), ti) if nq else (np.zeros(0, bool), 0)
            gm = ln.m_mask.view(ln.n, ln.cap)[k, :nq].cpu().numpy().astype(bool)
            ok_gms = bool(np.array_equal(gm, mo)) and int(ln.m_inl[k].item()) == int(no)
            ro = O.gicp_align(p["cloud0"], p["cloud1"])
            r = ln.gicp_out[k]
            T = np.array(r.T).reshape(4, 4).T
            e = float(np.linalg.norm(T - ro["T"]) / np.linalg.norm(ro["T"]))
            ok_gicp = e <= 1e-5 and int(r.iterations) == ro["iterations"] and bool(r.converged) == ro["converged"]
            return b, ok_orb, ok_match, ok_gms, ok_gicp, e

        with ThreadPoolExecutor(max_workers=min(len(picks), 16)) as ex:
            res = list(ex.map(check, picks))
        bad = [dict(pair=b, orb=o, match=m, gms=gm_, gicp=gi_, gicp_rel_err=e) for b, o, m, gm_, gi_, e in res if not (o and m and gm_ and gi_)]
        return dict(verified_pairs=len(res) - len(bad), checked_pairs=len(res),
                      checks="ORB key points + descriptors, BF matches, GMS mask bit-exact; GICP pose <= 1e-5 rel. Frobenius, iterations and converged equal",
                      max_gicp_rel_err=max(e for *_, e in res), failures=bad)

    # (run right behind the roofline leg and outside the leg budget: a headline without its oracle check is worth less than any extra)
    if rank == 0 and args.verify > 0 and not args.no_cpu_baseline and not args.gicp_stream:
        verify = guarded(verify_leg)
        emit()


    # ---- CPU baseline: the oracle (CPU restatement; the reference itself cannot be built: OpenCV/Eigen absent)
    def cpu_leg():
        from oracle import oracle as O
        O.lib()
        ncores = os.cpu_count() or 1
        orb0 = O.OrbOracle(NF, 1.2, NL, 20, 7)
        prev_full = [orb0.extract(p["gray0"]) for p in pairs]
        prev_cpu = [r[2] for r in prev_full]
        prev_kps_cpu = [r[1] for r in prev_full]
        # (a) the reference's own threading: one frame at a time, ORB with OpenMP over the 8 levels
        #     (src/ORBextractor.cc:775-777), GICP with 4 threads (src/RegistrationGICP.cc:10), match on all cores
        orb0.set_threads(8)
        O.gicp_set_threads(4)
        nseq = max(8, min(args.cpu_sample, 64))
        t1 = time.perf_counter()
        for i in range(nseq):
            p = pairs[i % nd]
            _, k_cur, d_cur = orb0.extract(p["gray1"])
            ti_cpu, _ = O.bf_match(prev_cpu[i % nd], d_cur, nthreads=min(ncores, 16))
            O.gms_inlier_mask(prev_kps_cpu[i % nd], (W, H), k_cur, (W, H), np.arange(len(ti_cpu), dtype=np.int32), ti_cpu)
            O.gicp_align(p["cloud0"], p["cloud1"])
        dt_ref = time.perf_counter() - t1
        orb0.set_threads(1)
        O.gicp_set_threads(1)
        # (b) all host cores: independent frame pairs on worker threads, each running the single-threaded oracle
        tl = threading.local()

        def one(i):
            p = pairs[i % nd]
            if not hasattr(tl, "orb"):
                tl.orb = O.OrbOracle(NF, 1.2, NL, 20, 7)
            _, k_cur, d_cur = tl.orb.extract(p["gray1"])
            ti_cpu, _ = O.bf_match(prev_cpu[i % nd], d_cur)
            O.gms_inlier_mask(prev_kps_cpu[i % nd], (W, H), k_cur, (W, H), np.arange(len(ti_cpu), dtype=np.int32), ti_cpu)
            O.gicp_align(p["cloud0"], p["cloud1"])

        nall = max(2 * ncores, 64)
        t1 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=ncores) as ex:
            list(ex.map(one, range(nall)))
        dt_all = time.perf_counter() - t1
        return dict(value=round(nseq / dt_ref, 3), unit="frames/s", cores=8, kind="port",
                   sample=f"{nseq} VGA frame pairs processed one at a time with the reference's threading: ORB 1000 feats "
                          f"(OpenMP over 8 levels) + BF match 1000x1000 + GMS + GICP ~19k-pt clouds (4 threads, as hard-coded)",
                   all_cores=dict(value=round(nall / dt_all, 3), cores=ncores,
                                  sample=f"{nall} pairs on {ncores} worker threads, single-threaded oracle per pair"),
                   note="CPU restatement of the reference algorithm (reference not buildable here: OpenCV/Eigen/PCL absent)")

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = guarded(cpu_leg)
        emit()

    # ---- the optical-flow stream next to the path: reported beside the metric, never part of `value`
    if rank == 0 and world == 1 and not args.no_klt and not over_budget("optical_flow"):
        try:
            WIN = 35
            cap = lanes[0].cap
            trk = api.KltTracker(W, H, WIN, max_level=3, max_batch=B, max_points=cap, device=local_rank)
            pyr_prev, pyr_cur = api.KltPyramid(trk), api.KltPyramid(trk)
            kp_all = torch.cat([ln.prev_kps.view(torch.float32).view(ln.n, cap, 7)[:, :, :2] for ln in lanes]).contiguous()
            cnt_all = torch.cat([ln.prev_cnt for ln in lanes]).contiguous()
            pri = kp_all.clone()
            st = torch.zeros(B * cap, dtype=torch.uint8, device=dev)
            good = torch.zeros(B, dtype=torch.int32, device=dev)
            trk.build_pyramid_device(gray0.data_ptr(), W, B, pyr_prev)

            def klt_step():  # per new frame: its pyramid (the previous frame's is kept) + forward/backward tracking of the key points
                trk.build_pyramid_device(gray1.data_ptr(), W, B, pyr_cur)
                pri.copy_(kp_all)
                torch.cuda.synchronize()
                trk.fb_track_device(pyr_prev, pyr_cur, B, cap, cnt_all.data_ptr(), kp_all.data_ptr(), pri.data_ptr(), st.data_ptr(),
                                    good.data_ptr(), 3, 15.0, 0.5)

            for _ in range(2):
                klt_step()
            torch.cuda.synchronize()
            nk = 10
            t1 = time.perf_counter()
            for _ in range(nk):
                klt_step()
            torch.cuda.synchronize()
            dtk = (time.perf_counter() - t1) / nk
            api.profile_reset()
            api.profile_enable(True)
            for _ in range(3):
                klt_step()
            torch.cuda.synchronize()
            krep = api.profile_report()
            api.profile_enable(False)
            npts_k, ngood_k = int(cnt_all.sum().item()), int(good.sum().item())
            # ... followed by the F check of the tracks (cv::findFundamentalMat FM_RANSAC) and the status update, all device-resident
            fmt = api.FundamentalMatcher(max_points=cap, max_batch=B, device=local_rank)
            t_a, t_b = torch.zeros_like(kp_all), torch.zeros_like(kp_all)
            t_idx = torch.zeros(B * cap, dtype=torch.int32, device=dev)
            t_m = torch.zeros(B, dtype=torch.int32, device=dev)
            t_mask = torch.zeros(B * cap, dtype=torch.uint8, device=dev)

            def chain_step():
                klt_step()
                trk.compact_tracks_device(B, cap, cnt_all.data_ptr(), kp_all.data_ptr(), pri.data_ptr(), st.data_ptr(), t_a.data_ptr(),
                                          t_b.data_ptr(), t_idx.data_ptr(), t_m.data_ptr())
                _, cnt_f = fmt.find_device(B, cap, t_m.data_ptr(), t_a.data_ptr(), t_b.data_ptr(), t_mask.data_ptr(), 1.0, 0.99)
                trk.apply_mask_device(B, cap, t_m.data_ptr(), t_idx.data_ptr(), t_mask.data_ptr(), st.data_ptr())
                return cnt_f

            chain_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                cnt_f = chain_step()
            torch.cuda.synchronize()
            dtc = (time.perf_counter() - t1) / 5
            klt = dict(metric="optical-flow frame pairs/s (buildOpticalFlowPyramid + fbKltTracking, window 35, 4 levels)",
                       value=round(B / dtk, 1), unit="pairs/s", ms_per_batch=round(dtk * 1e3, 3), batch_pairs=B,
                       points_per_pair=round(npts_k / B, 1), tracked_frac=round(ngood_k / max(npts_k, 1), 3),
                       kernels_ms_per_batch={k: round(v[0] / 3, 4) for k, v in sorted(krep.items(), key=lambda kv: -kv[1][0]) if "klt" in k})
            klt["with_f_check"] = dict(value=round(B / dtc, 1), unit="pairs/s", ms_per_batch=round(dtc * 1e3, 3),
                                       f_inlier_frac=round(float(cnt_f.sum()) / max(ngood_k, 1), 3),
                                       note="+ compaction of the tracks, findFundamentalMat(FM_RANSAC, 1 px, 0.99), status update")
            if not args.no_cpu_baseline:
                from oracle import oracle as O
                kp_h = kp_all[:4].cpu().numpy()
                cn_h = cnt_all[:4].cpu().numpy()
                t1 = time.perf_counter()
                o_prev = [O.klt_build_pyramid(pairs[sel[b]]["gray0"], WIN) for b in range(4)]
                for b in range(4):
                    o_cur = O.klt_build_pyramid(pairs[sel[b]]["gray1"], WIN)
                    O.fb_klt_tracking(o_prev[b], o_cur, W, H, WIN, 3, 15.0, 0.5, kp_h[b, :cn_h[b]], kp_h[b, :cn_h[b]].copy())
                klt["cpu_oracle"] = dict(value=round(4 / (time.perf_counter() - t1), 3), unit="pairs/s", cores=1,
                                         sample="4 VGA pairs, single-threaded oracle (5 pyramids + 4 forward/backward passes)")
        except Exception as e:  # a side figure must never cost the headline line
            klt = dict(error=f"{type(e).__name__}: {e}")

    # ---- the other SURVEY.md 8(d) figures, beside the metric (rank 0, N = 1): ORB only (configs[0]) and LBA windows (configs[4])
    emit()
    if rank == 0 and world == 1 and not args.no_extras and not over_budget("orb_only/lba/c3/c4_shard"):
        try:
            # all lanes' extractors at once, each on its own stream (the way the headline step runs them), and one lane alone
            def _orb_pass(lns, reps):
                for _ in range(reps):
                    for ln_ in lns:
                        ln_.ext.extract_batch_device(ln_.g1.data_ptr(), ln_.n, H, W, (0, 0), ln_.s1.cuda_stream)
                torch.cuda.synchronize()

            _orb_pass(lanes, 2)
            t1 = time.perf_counter()
            _orb_pass(lanes, 10)
            dto = (time.perf_counter() - t1) / 10
            ln0 = lanes[0]
            _orb_pass([ln0], 2)
            t1 = time.perf_counter()
            _orb_pass([ln0], 10)
            dto1 = (time.perf_counter() - t1) / 10
            nfr = sum(ln_.n for ln_ in lanes)
            extras["orb_only"] = dict(metric="ORB extraction frames/s (640x480, 1000 features, 8 levels; BASELINE.json configs[0] workload)",
                                      value=round(nfr / dto, 1), unit="frames/s", ms_per_batch=round(dto * 1e3, 3), batch_frames=nfr,
                                      lanes=len(lanes),
                                      one_lane=dict(value=round(ln0.n / dto1, 1), unit="frames/s", ms_per_batch=round(dto1 * 1e3, 3), batch_frames=ln0.n))
            w5 = synth.lba_window(0, n_free=20, n_fixed=5, n_points=3000)
            opt = api.Optimizer(max_poses=32, max_points=4096, max_edges=65536, device=local_rank)
            r5 = opt.LocalBundleAdjustment(w5)
            tl = []
            for _ in range(9):  # median: one window is 2.4 ms, a single hiccup on a shared host would dominate a mean
                t1 = time.perf_counter()
                r5 = opt.LocalBundleAdjustment(w5)
                tl.append(time.perf_counter() - t1)
            dtl = float(np.median(tl))
            # roofline of the LBA build (errors + landmark blocks + pose blocks = g2o computeActiveErrors + buildSystem): SURVEY.md
            # 8(d) counts 7.0 MB per build of this window (E x (read 2 poses' worth of state + obs, write Hpl 144 B) + Hll / Hpp / b)
            api.profile_reset()
            api.profile_enable(True)
            opt.LocalBundleAdjustment(w5)
            torch.cuda.synchronize()
            lrep = api.profile_report()
            api.profile_enable(False)
            build = [lrep[k] for k in ("k_lba_errors", "k_lba_build_landmarks", "k_lba_build_poses") if k in lrep]
            nbuild = lrep["k_lba_build_landmarks"][1] if "k_lba_build_landmarks" in lrep else 0
            build_ms = sum(v[0] for v in build)
            lba_roof = None
            if nbuild and build_ms > 0:
                lba_bytes = 7.0e6 * float(w5["n_edges"]) / 36000.0  # SURVEY's figure is for ~36k edges; this window has n_edges
                ach = lba_bytes * nbuild / (build_ms * 1e-3) / 1e9
                lba_roof = dict(bound="hbm", kernel="k_lba_errors + k_lba_build_landmarks + k_lba_build_poses (one build)",
                                achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                                algorithmic_bytes_per_build=int(lba_bytes), builds=int(nbuild), avg_build_us=round(build_ms / nbuild * 1e3, 2),
                                kernels_ms_per_window={k: round(v[0], 4) for k, v in sorted(lrep.items(), key=lambda kv: -kv[1][0]) if "lba" in k},
                                note="one window = one workgroup-chain: launch-latency bound, see DESIGN.md")
            # ... and 64 independent windows solved together (gfs_lba_solve_batch: replicas, the LBA of one map does not shard)
            lba_batch = None
            try:
                NW, ND = 64, 16
                wl = [synth.lba_window(k, n_free=20, n_fixed=5, n_points=3000) for k in range(ND)]
                wl = [wl[k % ND] for k in range(NW)]
                bat = api.BatchOptimizer(max_windows=NW, max_poses=32, max_points=4096, max_edges=65536, device=local_rank)
                Pb, Sb, outs_b, keep_b = bat.prepare(wl)  # (the result arrays must outlive the calls)
                bat.solve_prepared(Pb, Sb, NW)
                t1 = time.perf_counter()
                for _ in range(3):
                    bat.solve_prepared(Pb, Sb, NW)
                dtb = (time.perf_counter() - t1) / 3
                # the build kernels' roofline with the chip filled (same bytes per window as lba.roofline above)
                api.profile_reset()
                api.profile_enable(True)
                bat.solve_prepared(Pb, Sb, NW)
                torch.cuda.synchronize()
                brep = api.profile_report()
                api.profile_enable(False)
                bb = [brep[k] for k in ("kb_lba_errors", "kb_lba_build_landmarks", "kb_lba_build_poses") if k in brep]
                nb_b = brep["kb_lba_build_landmarks"][1] if "kb_lba_build_landmarks" in brep else 0
                batch_roof = None
                if nb_b and sum(v[0] for v in bb) > 0:
                    bms = sum(v[0] for v in bb)
                    # every launch covers the windows still iterating; count a window's build once per LM iteration it ran
                    builds = float(sum(int(S_.iterations_run) for S_ in Sb))
                    bytes_b = sum(7.0e6 * float(w_["n_edges"]) / 36000.0 * int(S_.iterations_run) for w_, S_ in zip(wl, Sb))
                    ach_b = bytes_b / (bms * 1e-3) / 1e9
                    batch_roof = dict(bound="hbm", kernel="kb_lba_errors + kb_lba_build_landmarks + kb_lba_build_poses", achieved=round(ach_b, 2),
                                      peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach_b / HBM_PEAK_GBS, 5), window_builds=int(builds),
                                      build_kernels_ms=round(bms, 3),
                                      kernels_ms_per_batch={k: round(v[0], 3) for k, v in sorted(brep.items(), key=lambda kv: -kv[1][0]) if "lba" in k})
                # several handles on as many host threads: the host preparation / scatter of one batch overlaps the device work of the
                # others (a server with several maps)
                hs = [(bat, Pb, Sb, keep_b)]
                flight = {}
                for NH in (2, 4):
                    while len(hs) < NH:
                        bt = api.BatchOptimizer(max_windows=NW, max_poses=32, max_points=4096, max_edges=65536, device=local_rank)
                        k0 = len(hs)
                        P_, S_, _o, keep_ = bt.prepare(wl[k0:] + wl[:k0])
                        bt.solve_prepared(P_, S_, NW)
                        hs.append((bt, P_, S_, keep_))
                    reps = 4

                    def _loop(bt, P_, S_, keep_):
                        for _ in range(reps):
                            bt.solve_prepared(P_, S_, NW)

                    th = [threading.Thread(target=_loop, args=a) for a in hs[:NH]]
                    t1 = time.perf_counter()
                    for t_ in th:
                        t_.start()
                    for t_ in th:
                        t_.join()
                    flight[NH] = round(NH * reps * NW / (time.perf_counter() - t1), 1)
                lba_batch = dict(value=flight[4], unit="windows/s", windows=NW, handles_in_flight=4, distinct_windows=ND,
                                 two_handles=dict(value=flight[2], unit="windows/s"),
                                 one_handle=dict(value=round(NW / dtb, 1), unit="windows/s", ms_per_batch=round(dtb * 1e3, 2)),
                                 roofline=batch_roof,
                                 note="host preparation, upload, solve and download of 64 windows per call, every call included; per window "
                                      "bit-identical to gfs_lba_solve")
                for h_ in hs[1:]:
                    h_[0].close()
                bat.close()
            except Exception as e:
                lba_batch = dict(error=f"{type(e).__name__}: {e}")
            extras["lba"] = dict(roofline=lba_roof, batched=lba_batch, metric="LocalBundleAdjustment windows/s (20 free + 5 fixed key-frames x 3000 points; BASELINE.json configs[4])",
                                 value=round(1.0 / dtl, 1), unit="windows/s", ms_per_window=round(dtl * 1e3, 3), edges=int(w5["n_edges"]),
                                 lm_iterations=int(r5["iterations_run"]))
            if not args.no_cpu_baseline:
                from oracle import oracle as O
                orb1 = O.OrbOracle(NF, 1.2, NL, 20, 7)
                orb1.set_threads(8)
                t1 = time.perf_counter()
                for i in range(8):
                    orb1.extract(pairs[i % nd]["gray1"])
                extras["orb_only"]["cpu_oracle"] = dict(value=round(8 / (time.perf_counter() - t1), 2), unit="frames/s", cores=8,
                                                        sample="8 VGA frames, OpenMP over the 8 levels as in the reference")
                t1 = time.perf_counter()
                O.lba_solve(w5)
                extras["lba"]["cpu_oracle"] = dict(value=round(1.0 / (time.perf_counter() - t1), 2), unit="windows/s", cores=1,
                                                   sample="the same window once, single-threaded oracle")
        except Exception as e:
            extras["side_figures_error"] = f"{type(e).__name__}: {e}"
        if args.workload == "c2":
            # BASELINE.json configs[2] (1280x720, 2000 features, ~37k-pt clouds, batch 32) as a side figure: the same script, its own
            # process (this one is idle meanwhile), 32 distinct scenes
            try:
                import subprocess
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "c3", "--batch", "32", "--lanes", "2", "--steps", "10",
                                     "--warmup", "3", "--no-klt", "--no-extras", "--no-cpu-baseline", "--verify", "0"],
                                    capture_output=True, text=True, timeout=600, env=env)
                line = [ln_ for ln_ in cp.stdout.splitlines() if ln_.startswith("{")]
                c3 = json.loads(line[-1])
                extras["c3"] = dict(metric="front-end frames/s on 1280x720 RGBD, 2000 features, ~37k-pt clouds, batch 32 (BASELINE.json configs[2])",
                                    value=c3["value"], unit="frames/s", ms_per_step=c3["ms_per_step"], batch_pairs=32,
                                    distinct_scenes=c3["config"]["distinct_scenes_per_gpu"],
                                    gicp_mean_outer_iterations=c3["config"]["gicp_mean_outer_iterations"],
                                    dominant_kernel=c3["roofline"]["kernel"], dominant_kernel_frac=c3["roofline"]["frac"],
                                    roofline=c3["roofline"])
            except Exception as e:
                extras["c3"] = dict(error=f"{type(e).__name__}: {e}")
            # BASELINE.json configs[3] seen from ONE GPU: the 64-pair block a GPU receives when the batch of 512 is cut over 8 GPUs
            # (`--strong` at N = 8), measured here at N = 1; efficiency = its rate over the rate at B = 512 (the headline)
            try:
                shard = {}
                for nl_ in (4, 2):
                    cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--batch", "64", "--lanes", str(nl_), "--steps", str(args.steps),
                                         "--warmup", str(args.warmup), "--no-klt", "--no-extras", "--no-cpu-baseline", "--verify", "0"],
                                        capture_output=True, text=True, timeout=600, env=env)
                    line = [ln_ for ln_ in cp.stdout.splitlines() if ln_.startswith("{")]
                    c4 = json.loads(line[-1])
                    shard[nl_] = dict(value=c4["value"], ms_per_step=c4["ms_per_step"], dominant_kernel=c4["roofline"]["kernel"],
                                      dominant_kernel_frac=c4["roofline"]["frac"])
                best = max(shard, key=lambda k_: shard[k_]["value"])
                extras["c4_shard"] = dict(metric="front-end frames/s of ONE GPU's share of BASELINE.json configs[3] (512 VGA pairs over 8 GPUs = 64 pairs per GPU)",
                                          value=shard[best]["value"], unit="frames/s", batch_pairs=64, lanes=best, ms_per_step=shard[best]["ms_per_step"],
                                          efficiency_vs_batch_512=round(shard[best]["value"] / fps, 3),
                                          projected_8gpu_strong=round(8 * shard[best]["value"], 1),
                                          by_lanes={str(k_): v_ for k_, v_ in shard.items()},
                                          note="projection = 8 x this rate (the blocks are independent, no collective); the driver's SCALE run measures the real thing")
            except Exception as e:
                extras["c4_shard"] = dict(error=f"{type(e).__name__}: {e}")

    # ---- ONE live stream, a frame at a time, as System::TrackRGBD (src/System.cc:600) drives the path: ORB -> stereo from RGB-D ->
    #      depth -> cloud -> GICP against the previous cloud (gfs_gicp_align_next) -> SearchByProjection -> PoseOptimization
    emit()
    if rank == 0 and world == 1 and not args.no_extras and args.workload == "c2" and not over_budget("single_stream"):
        try:
            import bench_stream as bs
            Kc = synth.intrinsics(W, H)
            frames_s = [(pairs[0]["gray0"], pairs[0]["depth0"]), (pairs[0]["gray1"], pairs[0]["depth1"])]
            gbe = bs.GpuBackend(api, W, H, NF, NL, SP, device=local_rank)
            lat_g, st_g, states_g = bs.run_stream(gbe, frames_s, Kc, W, H, STRIDE, 120, warm=6)
            ss = bs.summarize(lat_g, st_g)
            # the same chain with the ORB extraction running BESIDE depth -> cloud -> GICP (they do not depend on each other): what a
            # Tracking::GrabImageRGBD written for the device would do; same calls, same results, reported beside the sequential figure
            lat_v, st_v, states_v = bs.run_stream(gbe, frames_s, Kc, W, H, STRIDE, 120, warm=6, overlap=True)
            ov = bs.summarize(lat_v, st_v)
            ov["same_results_as_sequential"] = bool(all(a_["matches"] == b_["matches"] and a_["inliers"] == b_["inliers"] and np.array_equal(a_["T"], b_["T"])
                                                        and np.array_equal(a_["match"], b_["match"]) for a_, b_ in zip(states_g[-40:], states_v[-40:])))
            ss["orb_beside_registration"] = ov
            # PoseOptimization runs in the library's default above: g2o's edge order, the reference's outlier flags bit for bit.  The opt-in
            # fixed-shape tree sums (gfs_pose_set_sum_order(GFS_POSE_SUMS_TREE)) are timed beside it and labelled as what they are.
            ss["pose_sum_order"] = "edge_order (library default: bit-identical to the oracle's g2o restatement)"
            try:
                gbe.set_pose_sums("tree")
                gbe.have_target = False
                lat_t, st_t, _ = bs.run_stream(gbe, frames_s, Kc, W, H, STRIDE, 60, warm=6)
                tsum = bs.summarize(lat_t, st_t)
                ss["with_tree_sums_opt_in"] = dict(median_ms=tsum["median_ms"], pose_optimization_ms=tsum["stages_median_ms"].get("pose_optimization"),
                                                   note="GFS_POSE_SUMS_TREE: pose within 1e-7, outlier flags equal up to chi2-threshold ties; not the default")
            finally:
                gbe.set_pose_sums(None)
            ss.update(metric="single-stream front-end latency per frame (B = 1, sequential): ORB + the RGB-D tail of the Frame constructor (stereo-from-RGBD + depth->cloud: "
                             "gfs_frame_rgbd, the cloud stays on the device) + GICP (streaming entry) + SearchByProjection + PoseOptimization, host pointers in, "
                             "results out, every copy and sync included",
                      matches_per_frame=int(np.median([s_["matches"] for s_ in states_g])), pose_inliers_per_frame=int(np.median([s_["inliers"] for s_ in states_g])))
            if not args.no_cpu_baseline:
                from oracle import oracle as O
                obe = bs.OracleBackend(O, W, H, NF, NL)
                try:
                    lat_o, st_o, states_o = bs.run_stream(obe, frames_s, Kc, W, H, STRIDE, 12, warm=1)
                finally:
                    obe.close()
                so = bs.summarize(lat_o, st_o)
                so.update(cores=8, kind="port", note="the oracle chain with the reference's threading (ORB: OpenMP over 8 levels, GICP: 4 threads, both clouds "
                                                     "preprocessed per call like RegistrationGICP::RegisterPointClouds)")
                ss["cpu_oracle"] = so
                # the two chains must tell the same story, frame by frame (frame k of the CPU run = frame k of the GPU run: same sequence)
                ga, oa = states_g[5:5 + len(states_o)], states_o  # GPU warm = 6, CPU warm = 1: align the sequence positions
                same = len(oa) > 0 and all(a_["matches"] == b_["matches"] and a_["inliers"] == b_["inliers"]
                                          and np.linalg.norm(a_["T"] - b_["T"]) <= 1e-5 * np.linalg.norm(b_["T"]) and np.array_equal(a_["match"], b_["match"])
                                          for a_, b_ in zip(ga, oa))
                ss["agrees_with_oracle"] = bool(same)
                ss["speedup_vs_cpu_oracle"] = round(so["median_ms"] / ss["median_ms"], 1)
            extras["single_stream"] = ss
        except Exception as e:
            extras["single_stream"] = dict(error=f"{type(e).__name__}: {e}")

    # ---- PCIe-inclusive figure (N = 1): images and depth maps start in pinned HOST memory every pass, the clouds are built on the
    #      device (Frame::ConvertDepthToPointCloud), the per-pair results come back to pinned host memory
    if rank == 0 and world == 1 and not args.no_extras and args.workload == "c2" and not args.gicp_stream and not over_budget("h2d_inclusive"):
        try:
            fx, fy, cx_, cy_ = synth.intrinsics(W, H)
            h_gray = torch.from_numpy(np.stack([pairs[i]["gray1"] for i in sel])).pin_memory()
            h_d0 = torch.from_numpy(np.stack([pairs[i]["depth0"] for i in sel])).pin_memory()
            h_d1 = torch.from_numpy(np.stack([pairs[i]["depth1"] for i in sel])).pin_memory()
            h_res = [dict(idx=torch.empty(ln.n * ln.cap, dtype=torch.int32).pin_memory(), mask=torch.empty(ln.n * ln.cap, dtype=torch.uint8).pin_memory(),
                          cnt=torch.empty(ln.n, dtype=torch.int32).pin_memory()) for ln in lanes]
            for ln in lanes:
                ln.frm = api.Frame(max_rows=H, max_cols=W, device=local_rank)
                ln.dd0 = torch.empty((ln.n, H, W), dtype=torch.float32, device=dev)
                ln.dd1 = torch.empty((ln.n, H, W), dtype=torch.float32, device=dev)
                ln.gg = torch.empty((ln.n, H, W), dtype=torch.uint8, device=dev)
                ln.cc0, ln.cc1 = torch.zeros_like(ln.c0), torch.zeros_like(ln.c1)
                ln.nn0, ln.nn1 = torch.zeros_like(ln.n0), torch.zeros_like(ln.n1)

            def orb_h2d(ln, hr):
                with torch.cuda.stream(ln.s1):
                    ln.gg.copy_(h_gray[ln.b0:ln.b0 + ln.n], non_blocking=True)
                sp = ln.s1.cuda_stream
                ln.ext.extract_batch_device(ln.gg.data_ptr(), ln.n, H, W, (0, 0), sp)
                ln.mt.match_batch_device(ln.prev_desc.data_ptr(), ln.prev_cnt.data_ptr(), ln.res["desc"], ln.res["counts"], ln.n, ln.cap,
                                         ln.m_idx.data_ptr(), ln.m_dist.data_ptr(), sp)
                ln.gms.inlier_mask_batch_device(ln.prev_kps.data_ptr(), ln.prev_cnt.data_ptr(), ln.res["kps"], ln.res["counts"], ln.n, ln.cap,
                                                ln.m_idx.data_ptr(), W, H, ln.m_mask.data_ptr(), ln.m_inl.data_ptr(), sp)
                with torch.cuda.stream(ln.s1):
                    hr["idx"].copy_(ln.m_idx, non_blocking=True)
                    hr["mask"].copy_(ln.m_mask, non_blocking=True)
                    hr["cnt"].copy_(ln.m_inl, non_blocking=True)
                ln.s1.synchronize()

            def gicp_h2d(ln):
                with torch.cuda.stream(ln.s2):
                    ln.dd0.copy_(h_d0[ln.b0:ln.b0 + ln.n], non_blocking=True)
                    ln.dd1.copy_(h_d1[ln.b0:ln.b0 + ln.n], non_blocking=True)
                sp = ln.s2.cuda_stream
                ln.frm.depth_to_cloud_batch_device(ln.dd0.data_ptr(), ln.n, H, W, STRIDE, fx, fy, cx_, cy_, ln.cc0.data_ptr(), SP, ln.nn0.data_ptr(), sp)
                ln.frm.depth_to_cloud_batch_device(ln.dd1.data_ptr(), ln.n, H, W, STRIDE, fx, fy, cx_, cy_, ln.cc1.data_ptr(), SP, ln.nn1.data_ptr(), sp)
                ln.gicp_out = ln.reg.align_batch_device(ln.cc0.data_ptr(), ln.nn0.data_ptr(), ln.cc1.data_ptr(), ln.nn1.data_ptr(), ln.n, SP,
                                                        None, None, sp, raw=True)  # (returns the poses on the host)

            def h2d_steps(k):
                def chain(f, *a):
                    for _ in range(k):
                        f(*a)
                futs = [pool.submit(chain, orb_h2d, ln, hr) for ln, hr in zip(lanes, h_res)] + [pool.submit(chain, gicp_h2d, ln) for ln in lanes]
                for f in futs:
                    f.result()

            h2d_steps(2)
            torch.cuda.synchronize()
            kh = max(3, args.steps // 4)
            t1 = time.perf_counter()
            h2d_steps(kh)
            torch.cuda.synchronize()
            dth = (time.perf_counter() - t1) / kh
            same = all(int(a) == int(b_) for ln in lanes for a, b_ in zip(ln.nn1.cpu().tolist(), ln.n1.cpu().tolist()))
            h2d = dict(value=round(B / dth, 1), unit="frames/s", ms_per_step=round(dth * 1e3, 3),
                       bytes_in_per_pair=W * H * (1 + 4 + 4), clouds_match_host_built=bool(same),
                       note="per pass: gray image + both depth maps pinned host -> HBM, depth -> cloud on device (gfs_depth_to_cloud_batch_device), "
                            "ORB + match + GMS + GICP, match indices / GMS mask / poses back to pinned host memory; never part of `value`")
            # the same with the depth maps as the sensor delivers them (CV_16U, 1 / 5000 m): 2 bytes a pixel over PCIe, converted on the
            # device (imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor), src/Tracking.cc:1622-1623)
            try:
                FACT = 1.0 / 5000.0
                q16 = lambda key: torch.from_numpy(np.stack([np.clip(np.rint(pairs[i][key] * 5000.0), 0, 65535).astype(np.uint16) for i in sel])).pin_memory()
                h_q0, h_q1 = q16("depth0"), q16("depth1")
                for ln in lanes:
                    ln.qq0 = torch.empty((ln.n, H, W), dtype=torch.uint16, device=dev)
                    ln.qq1 = torch.empty((ln.n, H, W), dtype=torch.uint16, device=dev)

                def gicp_h2d_u16(ln):
                    with torch.cuda.stream(ln.s2):
                        ln.qq0.copy_(h_q0[ln.b0:ln.b0 + ln.n], non_blocking=True)
                        ln.qq1.copy_(h_q1[ln.b0:ln.b0 + ln.n], non_blocking=True)
                    sp = ln.s2.cuda_stream
                    ln.frm.depth_convert_u16_batch_device(ln.qq0.data_ptr(), ln.n, H, W, FACT, ln.dd0.data_ptr(), sp)
                    ln.frm.depth_convert_u16_batch_device(ln.qq1.data_ptr(), ln.n, H, W, FACT, ln.dd1.data_ptr(), sp)
                    ln.frm.depth_to_cloud_batch_device(ln.dd0.data_ptr(), ln.n, H, W, STRIDE, fx, fy, cx_, cy_, ln.cc0.data_ptr(), SP, ln.nn0.data_ptr(), sp)
                    ln.frm.depth_to_cloud_batch_device(ln.dd1.data_ptr(), ln.n, H, W, STRIDE, fx, fy, cx_, cy_, ln.cc1.data_ptr(), SP, ln.nn1.data_ptr(), sp)
                    ln.gicp_out = ln.reg.align_batch_device(ln.cc0.data_ptr(), ln.nn0.data_ptr(), ln.cc1.data_ptr(), ln.nn1.data_ptr(), ln.n, SP,
                                                            None, None, sp, raw=True)

                def u16_steps(k):
                    def chain(f, *a):
                        for _ in range(k):
                            f(*a)
                    futs = [pool.submit(chain, orb_h2d, ln, hr) for ln, hr in zip(lanes, h_res)] + [pool.submit(chain, gicp_h2d_u16, ln) for ln in lanes]
                    for f in futs:
                        f.result()

                u16_steps(2)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                u16_steps(kh)
                torch.cuda.synchronize()
                dtu = (time.perf_counter() - t1) / kh
                ln0_ = lanes[0]
                want = [int((((q := h_q1[ln0_.b0 + j].numpy().astype(np.float32) * np.float32(FACT))[::STRIDE, ::STRIDE] > 0) & (q[::STRIDE, ::STRIDE] < 10)).sum())
                        for j in range(min(4, ln0_.n))]
                h2d["u16_depth"] = dict(value=round(B / dtu, 1), unit="frames/s", ms_per_step=round(dtu * 1e3, 3), bytes_in_per_pair=W * H * (1 + 2 + 2),
                                        cloud_sizes_match_host=bool(want == ln0_.nn1.cpu().tolist()[:len(want)]),
                                        note="depth maps as CV_16U (1/5000 m) over PCIe, gfs_depth_convert_u16_batch_device on arrival")
                for ln in lanes:
                    del ln.qq0, ln.qq1
            except Exception as e:
                h2d["u16_depth"] = dict(error=f"{type(e).__name__}: {e}")
            # what a LIVE stream moves: ONE new frame per pair and pass -- its gray image and its CV_16U depth map (0.92 MB) --, the
            # previous frame's features and preprocessed cloud stay in HBM (gfs_gicp_align_next: the last source becomes the target)
            try:
                for ln in lanes:
                    ln.qq = torch.empty((ln.n, H, W), dtype=torch.uint16, device=dev)
                    ln.flip = False
                    ln.reg.align_batch_device(ln.c0.data_ptr(), ln.n0.data_ptr(), ln.c1.data_ptr(), ln.n1.data_ptr(), ln.n, SP, None, None,
                                              ln.s2.cuda_stream, raw=True)  # opens the stream: frame 1 is the target of the first streamed call

                def gicp_h2d_stream(ln):
                    ln.flip = not ln.flip
                    hq = h_q0 if ln.flip else h_q1  # the stream walks back and forth between the pair's two frames
                    with torch.cuda.stream(ln.s2):
                        ln.qq.copy_(hq[ln.b0:ln.b0 + ln.n], non_blocking=True)
                    sp = ln.s2.cuda_stream
                    ln.frm.depth_convert_u16_batch_device(ln.qq.data_ptr(), ln.n, H, W, FACT, ln.dd0.data_ptr(), sp)
                    ln.frm.depth_to_cloud_batch_device(ln.dd0.data_ptr(), ln.n, H, W, STRIDE, fx, fy, cx_, cy_, ln.cc0.data_ptr(), SP, ln.nn0.data_ptr(), sp)
                    ln.gicp_out = ln.reg.align_next_batch_device(ln.cc0.data_ptr(), ln.nn0.data_ptr(), ln.n, SP, None, None, sp, raw=True)

                def stream_steps(k):
                    def chain(f, *a):
                        for _ in range(k):
                            f(*a)
                    futs = [pool.submit(chain, orb_h2d, ln, hr) for ln, hr in zip(lanes, h_res)] + [pool.submit(chain, gicp_h2d_stream, ln) for ln in lanes]
                    for f in futs:
                        f.result()

                stream_steps(2)
                torch.cuda.synchronize()
                ks = 2 * max(2, kh // 2)  # an even number of passes: as many forward as backward steps
                t1 = time.perf_counter()
                stream_steps(ks)
                torch.cuda.synchronize()
                dts = (time.perf_counter() - t1) / ks
                conv = float(np.mean([bool(r_.converged) for ln in lanes for r_ in ln.gicp_out]))
                h2d["streaming"] = dict(value=round(B / dts, 1), unit="frames/s", ms_per_step=round(dts * 1e3, 3), bytes_in_per_frame=W * H * (1 + 2),
                                        gicp_converged_frac=round(conv, 3),
                                        note="per pass and pair ONE new frame crosses PCIe (gray u8 + depth CV_16U); ORB + match + GMS against the resident "
                                             "previous features, depth convert + cloud on the device, gfs_gicp_align_next_batch_device (only the new cloud "
                                             "is preprocessed); match indices / GMS mask / poses back to pinned host memory")
                for ln in lanes:
                    del ln.qq
            except Exception as e:
                h2d["streaming"] = dict(error=f"{type(e).__name__}: {e}")
            for ln in lanes:
                del ln.dd0, ln.dd1, ln.gg, ln.cc0, ln.cc1
        except Exception as e:
            h2d = dict(error=f"{type(e).__name__}: {e}")
    # ---- HBM-side traffic of the roofline's kernel, MEASURED BY THIS RUN: two child runs of this script (one step of the same batch and
    #      lane shape, 64 distinct scenes tiled) under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` -- counters in their own passes,
    #      no trace domains (MI355X_MICROARCH.md).  The last leg: whatever happens here costs nothing that was measured before.
    if (rank == 0 and world == 1 and not args.no_extras and args.workload == "c2" and isinstance(roofline, dict) and roofline.get("kernel")
            and os.environ.get("GFS_BENCH_LIVE_TRAFFIC", "1") != "0" and not over_budget("traffic_live")):
        try:
            live = measure_traffic_live(roofline["kernel"], B, nlanes, args.live_traffic_timeout_s, roofline.get("algorithmic_bytes_per_launch"))
            roofline["traffic_live"] = live
            if live.get("traffic") is not None:
                roofline["traffic_committed"] = dict(traffic=roofline.get("traffic"), traffic_raw_counters=roofline.get("traffic_raw_counters"),
                                                     source=roofline.get("traffic_source"))
                roofline["traffic"], roofline["traffic_raw_counters"] = live["traffic"], live["traffic_raw_counters"]
                roofline["traffic_source"] = live["source"]
                roofline["traffic_is_upper_bound"] = True
        except Exception as e:
            roofline["traffic_live"] = dict(error=f"{type(e).__name__}: {e}")
    emit(final=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
