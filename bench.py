#!/usr/bin/env python3
"""bench.py — front-end frames/sec (ORB + BF match + GICP) on synthetic 640x480 RGB-D, BASELINE.json config C2.

One "step" = one pass of the hot path over one batch of B independent frame pairs already resident in HBM:
    ORB extract (1000 feats, 8 levels) of the B current frames
 -> brute-force Hamming match of the B previous-frame descriptor sets against them
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

# SURVEY.md §8(d) algorithmic bytes per unit of work for each kernel
def algorithmic_bytes(kernel, ctx):
    P0, P, K = ctx["P0"], ctx["P"], ctx["K"]  # level-0 pixels, pyramid pixels, keypoints per frame
    B = ctx["B"]
    if kernel == "k_fast_cells":      # P read + 12 B x candidates (here 4 B packed)  -> per frame
        return B * (P + 4 * ctx["cands"])
    if kernel == "k_pyr_area":        # whole chain per frame: P0 read + (P - P0) write + re-read of levels 0..L-2
        return B * (P0 + (P - P0) + (P - ctx["p_last"])) / ctx["nlevels_m1"]  # per launch (one level)
    if kernel == "k_blur7":
        return B * 2 * P
    if kernel == "k_orient_brief":    # K x (31x31 + 37x37) read + 60 B out
        return B * K * (31 * 31 + 37 * 37 + 60)
    if kernel == "k_bf_hamming":
        return B * (32 * 2 * K + 8 * K)
    if kernel == "k_gicp_linearize":  # 320 B per source point per linearisation
        return 320.0 * ctx["lin_points_per_launch"]
    if kernel == "k_gicp_error":
        return 136.0 * ctx["err_points_per_launch"]
    if kernel == "k_knn_cov":         # kNN gather 10 x 32 B + 160 B write, per down-sampled point, both clouds
        return (10 * 32 + 160) * ctx["ds_points"]
    if kernel == "k_radix_sort":      # 2 x (key + idx) per pass-free sort, N points
        return 2 * 12 * ctx["sort_points_per_launch"]
    if kernel == "k_voxel_reduce":
        return (16 + 12) * ctx["in_points"] + 32 * ctx["ds_points"]
    if kernel == "k_voxel_keys":
        return (16 + 12) * ctx["in_points"]
    if kernel == "k_cell_build":
        return (32 + 12 + 32) * ctx["ds_points"]
    if kernel == "k_cand_pack":
        return B * 8 * ctx["cands"]
    return None


def gen_pairs(n_distinct, seed0, width, height, stride):
    from geoflowslam_amd import synth
    with ThreadPoolExecutor(max_workers=min(n_distinct, os.cpu_count() or 1)) as ex:
        return list(ex.map(lambda s: synth.frame_pair(s, width, height, stride), range(seed0, seed0 + n_distinct)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="frame pairs per GPU per step")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic scenes per GPU (tiled to --batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true", help="run ORB+match and GICP back to back on one stream")
    ap.add_argument("--cpu-sample", type=int, default=32, help="frame pairs in the CPU-baseline sample")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from geoflowslam_amd import api, synth
    W, H, STRIDE, NF, NL = 640, 480, 4, 1000, 8
    B = args.batch
    nd = max(1, min(args.distinct, B))
    pairs = gen_pairs(nd, 1000 + 100 * rank, W, H, STRIDE)
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

    ext = api.ORBextractor(NF, 1.2, NL, 20, 7, max_rows=H, max_cols=W, max_batch=B, device=local_rank)
    mt = api.ORBmatcher(max_query=ext.cap, max_train=ext.cap, max_batch=B, device=local_rank)
    reg = api.RegistrationGICP(max_points=SP, max_batch=B, device=local_rank)
    stream = torch.cuda.Stream(device=dev)
    stream2 = torch.cuda.Stream(device=dev)
    sp, sp2 = stream.cuda_stream, stream2.cuda_stream
    cap = ext.cap
    pool = ThreadPoolExecutor(max_workers=2)

    # previous-frame features (the "keyframe" side of SearchWithGMS): computed once, kept in HBM
    ext.extract_batch_device(gray0.data_ptr(), B, H, W, (0, 0), sp)
    res = ext.device_results()
    stream.synchronize()
    import ctypes as C
    prev_desc = torch.empty(B * cap * 32, dtype=torch.uint8, device=dev)
    prev_cnt = torch.empty(B, dtype=torch.int32, device=dev)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(prev_desc.data_ptr(), res["desc"], B * cap * 32, 3) == 0
    assert hip.hipMemcpy(prev_cnt.data_ptr(), res["counts"], B * 4, 3) == 0
    m_idx = torch.empty(B * cap, dtype=torch.int32, device=dev)
    m_dist = torch.empty(B * cap, dtype=torch.int32, device=dev)
    last = {}

    def orb_and_match():
        ext.extract_batch_device(gray1.data_ptr(), B, H, W, (0, 0), sp)
        mt.match_batch_device(prev_desc.data_ptr(), prev_cnt.data_ptr(), res["desc"], res["counts"], B, cap,
                              m_idx.data_ptr(), m_dist.data_ptr(), sp)

    def gicp():
        last["gicp"] = reg.align_batch_device(d_c0.data_ptr(), d_n0.data_ptr(), d_c1.data_ptr(), d_n1.data_ptr(), B, SP,
                                              None, None, sp2)

    def step():
        # ORB (+ its host quadtree) and GICP of the same batch are independent: two host threads, two HIP streams
        # (ctypes releases the GIL), so the quadtree hides under the GICP kernels.
        if args.serial:
            orb_and_match()
            gicp()
        else:
            f1, f2 = pool.submit(orb_and_match), pool.submit(gicp)
            f1.result()
            f2.result()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    fps = world * B * args.steps / dt

    # ---- dominant-kernel roofline: per-kernel HIP-event timing on the launch stream (extra, untimed steps)
    roofline = None
    kern = {}
    if rank == 0:
        api.profile_reset()
        api.profile_enable(True)
        nprof = 2
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        kern = api.profile_report()
        api.profile_enable(False)
        g = last["gicp"]
        counts = torch.empty(B, dtype=torch.int32)
        assert hip.hipMemcpy(counts.data_ptr(), res["counts"], B * 4, 2) == 0
        lin_pts = sum(r["n_linearize"] * r["n_source_ds"] for r in g)
        err_pts = sum(r["n_error_evals"] * r["n_source_ds"] for r in g)
        nl_lin = max(1, kern.get("k_gicp_linearize", (0, 1))[1] // nprof)
        nl_err = max(1, kern.get("k_gicp_error", (0, 1))[1] // nprof)
        lv = [ext.level_size(l) for l in range(NL)]
        P = sum(r * c for r, c in lv)
        ctx = dict(B=B, P0=W * H, P=P, p_last=lv[-1][0] * lv[-1][1], nlevels_m1=NL - 1, K=float(counts.float().mean()),
                   cands=float(np.mean([sum(len(ext.candidates(l, b)[0]) for l in range(NL)) for b in range(min(B, 4))])),
                   lin_points_per_launch=lin_pts / nl_lin, err_points_per_launch=err_pts / nl_err,
                   ds_points=sum(r["n_source_ds"] + r["n_target_ds"] for r in g), in_points=int(n0.sum() + n1.sum()),
                   sort_points_per_launch=(int(n0.sum() + n1.sum()) + sum(r["n_source_ds"] + r["n_target_ds"] for r in g)) / 2)
        tot = sum(v[0] for v in kern.values())
        name = max(kern, key=lambda k: kern[k][0])
        ms, launches = kern[name]
        avg_s = ms / launches / 1e3
        ab = algorithmic_bytes(name, ctx)
        ach = ab / avg_s / 1e9 if ab else None
        roofline = dict(bound="hbm", kernel=name, achieved=round(ach, 2) if ach else None, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(ach / HBM_PEAK_GBS, 5) if ach else None, traffic=None,
                        avg_launch_us=round(avg_s * 1e6, 2), launches_per_step=launches / nprof,
                        algorithmic_bytes_per_launch=int(ab) if ab else None,
                        share_of_gpu_kernel_time=round(ms / tot, 3),
                        kernels_ms_per_step={k: round(v[0] / nprof, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0])})

    # ---- CPU baseline: the oracle (CPU restatement; the reference itself cannot be built: OpenCV/Eigen absent)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        O.lib()
        ncores = os.cpu_count() or 1
        orb0 = O.OrbOracle(NF, 1.2, NL, 20, 7)
        prev_cpu = [orb0.extract(p["gray0"])[2] for p in pairs]
        # (a) the reference's own threading: one frame at a time, ORB with OpenMP over the 8 levels
        #     (src/ORBextractor.cc:775-777), GICP with 4 threads (src/RegistrationGICP.cc:10), match on all cores
        orb0.set_threads(8)
        O.gicp_set_threads(4)
        nseq = max(8, min(args.cpu_sample, 64))
        t1 = time.perf_counter()
        for i in range(nseq):
            p = pairs[i % nd]
            _, _, d_cur = orb0.extract(p["gray1"])
            O.bf_match(prev_cpu[i % nd], d_cur, nthreads=min(ncores, 16))
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
            _, _, d_cur = tl.orb.extract(p["gray1"])
            O.bf_match(prev_cpu[i % nd], d_cur)
            O.gicp_align(p["cloud0"], p["cloud1"])

        nall = max(2 * ncores, 64)
        t1 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=ncores) as ex:
            list(ex.map(one, range(nall)))
        dt_all = time.perf_counter() - t1
        cpu = dict(value=round(nseq / dt_ref, 3), unit="frames/s", cores=8, kind="port",
                   sample=f"{nseq} VGA frame pairs processed one at a time with the reference's threading: ORB 1000 feats "
                          f"(OpenMP over 8 levels) + BF match 1000x1000 + GICP ~19k-pt clouds (4 threads, as hard-coded)",
                   all_cores=dict(value=round(nall / dt_all, 3), cores=ncores,
                                  sample=f"{nall} pairs on {ncores} worker threads, single-threaded oracle per pair"),
                   note="CPU restatement of the reference algorithm (reference not buildable here: OpenCV/Eigen/PCL absent)")

    if rank == 0:
        g = last["gicp"]
        out = {
            "metric": "front-end frames/sec (ORB+match+GICP) on 640x480 RGBD",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32 (ORB, match) + f64 (GICP)", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: 640x480 RGBD frame pair, ORB extract (1000 feats, 8 levels) "
                                   "+ BF Hamming match + GICP on ~19k-pt clouds (stride-4 depth grid)",
                       "batch_pairs_per_gpu": B, "distinct_scenes_per_gpu": nd, "parallelism": f"frames sharded x{world}, no collective",
                       "gicp_mean_outer_iterations": round(float(np.mean([r["n_linearize"] for r in g])), 2),
                       "gicp_mean_error_evals": round(float(np.mean([r["n_error_evals"] for r in g])), 2),
                       "gicp_converged_frac": round(float(np.mean([r["converged"] for r in g])), 3)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if cpu:
            out["gpu_over_cpu"] = round(fps / cpu["value"], 2)
            out["gpu_over_cpu_all_cores"] = round(fps / cpu["all_cores"]["value"], 2)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
