"""Randomised parity fuzz on an MI355X against the CPU oracle (not collected by pytest; run it by hand):
    python tests/fuzz_gpu.py [seconds] [seed]
voxel-sort permutations (random / tied / raster-like / adversarial keys, 1 .. 50 000 elements) bit-exact, GICP on random cloud
pairs (sizes, truncations, motions) with equal iteration / inlier counts and the pose within the 1e-5 bar (the largest error is
printed), ORB on odd image sizes / feature counts / level counts bit-exact, LocalBundleAdjustment windows of random size with
and without second-camera edges, mixed LBA batches and ragged GICP batches bit for bit against single calls.  Exit code 1 on any failure.  Round 2: 900 k sorts, 13 000 GICP pairs (without a tie at the 10th neighbour the pose agrees to
1e-16; with one — noise-free raster clouds have one or two per cloud — 7 pairs over the 1e-5 bar), 2 200 ORB frames, 3 000 LBA windows, ~1 200 each of
SearchByProjection / PoseOptimization / BF match / fbKltTracking / findFundamentalMat, 1 055 mixed LBA batches and 2 966 ragged GICP batches (device entry) against single calls: no failure; the total
LM iteration count of PoseOptimization differs by one in ~5 % of the frames (poses equal to 1e-10)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from geoflowslam_amd import api, synth
from oracle import oracle as O
fails = []
T0 = time.time()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(1234 + seed0)


def case_rng(section, index):  # a case's draws depend on (seed, section, index) only: any failure can be replayed
    return np.random.default_rng([1234 + seed0, section, index])


SECTIONS = [int(v) for v in os.environ.get("FUZZ_SECTIONS", "1,2,3,4,5,6,7").split(",")]  # e.g. FUZZ_SECTIONS=2: cloud pairs only
ONLY = None  # "section:index" as the third argument replays one case
if len(sys.argv) > 3:
    ONLY = tuple(int(v) for v in sys.argv[3].split(":"))

def vk(x, y, z):
    return (x.astype(np.uint64) | (y.astype(np.uint64) << np.uint64(21)) | (z.astype(np.uint64) << np.uint64(42)))

# ---- 1. voxel sort
reg = api.RegistrationGICP(max_points=65536)
nsort = 0
while time.time() - T0 < budget * 0.25 and ONLY is None and 1 in SECTIONS:
    rng = case_rng(1, nsort)
    n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 1100), rng.integers(1100, 6000), rng.integers(6000, 50000)]))
    mode = rng.integers(0, 7)
    if mode == 0:
        x, y, z = rng.integers(0, 300, n), rng.integers(0, 200, n), rng.integers(0, 6, n)
    elif mode == 1:
        m = rng.integers(1, 40); x, y, z = rng.integers(0, m, n), rng.integers(0, 3, n), np.zeros(n, np.int64)
    elif mode == 2:   # raster-like: sorted runs with local jitter (depth image order)
        base = np.arange(n) // rng.integers(1, 5); j = rng.integers(-2, 3, n)
        x, y, z = (base + j) % 700 + 5, (base // 700) + 5, rng.integers(0, 2, n)
    elif mode == 3:   # sawtooth / organ pipe
        t = np.arange(n); p = rng.integers(2, 200)
        x, y, z = np.minimum(t % p, p - t % p) + 3, np.full(n, 4), np.full(n, 5)
    elif mode == 4:   # adversarial block embedded
        a = O.antiqsort_keys(min(n, int(rng.integers(20, 1024)))).astype(np.int64) // int(rng.integers(1, 4))
        rest = rng.integers(0, 5000, max(n - len(a), 0))
        mix = np.concatenate([rest[:len(rest) // 2], a + 6000, rest[len(rest) // 2:] + 9000])
        if rng.integers(0, 2): mix = mix[rng.permutation(len(mix))]
        x, y, z = mix, np.full(len(mix), 2), np.full(len(mix), 2)
    elif mode == 5:   # wide extent (64-bit path) with ties
        x, y, z = rng.integers(0, 1 << 16, n), rng.integers(0, 1 << 13, n), rng.integers(0, 1 << 9, n)
        if n > 6:
            c = n // 3; x[:c] = x[c:2 * c]; y[:c] = y[c:2 * c]; z[:c] = z[c:2 * c]
    else:             # few distinct + sorted descending
        x = np.sort(rng.integers(0, 50, n))[::-1].copy(); y = np.zeros(n, np.int64); z = np.zeros(n, np.int64)
    k = vk(np.asarray(x) + 1000, np.asarray(y) + 2000, np.asarray(z) + 3000)
    if rng.integers(0, 3) == 0 and len(k) > 10:
        k[rng.integers(0, len(k), max(1, len(k) // 40))] = np.uint64(0xFFFFFFFFFFFFFFFF)
    got = reg.voxel_sort_perm(k); want, _ = O.quick_sort_perm(k)
    nsort += 1
    if not np.array_equal(got, want):
        fails.append(("sort", "case 1:%d" % (nsort - 1), int(mode), len(k), int((got != want).sum())))
print("sort cases", nsort, "fails", len(fails), flush=True)

# ---- 2. GICP pairs of random size / motion
ngicp = 0
worst = 0.0
tie_cases = []
while (time.time() - T0 < budget * (0.5 if SECTIONS != [2] else 1.0) and ONLY is None and 2 in SECTIONS) or (ONLY is not None and ONLY[0] == 2 and ngicp == 0):
    ci = ngicp if ONLY is None else ONLY[1]
    rng = case_rng(2, ci)
    s = int(rng.integers(0, 1 << 30))
    w, h = int(rng.choice([96, 128, 160, 200])), int(rng.choice([72, 96, 120, 150]))
    tr_, rd_ = float(rng.uniform(0.0, 0.15)), float(rng.uniform(0, 5))
    c0, c1, T = synth.cloud_pair(s, w, h, trans=tr_, rot_deg=rd_)
    if rng.integers(0, 4) == 0: c0 = c0[:int(len(c0) * rng.uniform(0.2, 1.0))]
    if rng.integers(0, 4) == 0: c1 = c1[::int(rng.integers(1, 4))]
    r = reg.RegisterPointClouds(c0, c1); ro = O.gicp_align(c0, c1)
    ngicp += 1
    rel = np.linalg.norm(r["T"] - ro["T"]) / np.linalg.norm(ro["T"])
    worst = max(worst, float(rel))
    if not (r["converged"] == ro["converged"] and r["iterations"] == ro["iterations"] and r["num_inliers"] == ro["num_inliers"]
            and r["n_target_ds"] == ro["n_target_ds"] and r["n_source_ds"] == ro["n_source_ds"] and rel < 1e-5):
        # the one known cause: an exact distance tie at the 10th neighbour of some point -- established here,
        # not assumed: the voxel means must be identical, a tie must exist, and the error must stay small
        nt = 0
        same = True
        for which, cl in ((0, c0), (1, c1)):
            pts, _ = reg.preprocessed(0, which)
            po, _, _ = O.gicp_preprocess(cl)
            ig = np.lexsort((pts[:, 2], pts[:, 1], pts[:, 0])); io = np.lexsort((po[:, 2], po[:, 1], po[:, 0]))
            same = same and len(pts) == len(po) and bool((pts[ig] == po[io]).all())
            if same and len(po) > 11:
                _, sq = O.knn(po[io], po[io], 11)
                nt += int((sq[:, 9] == sq[:, 10]).sum())
        if same and nt > 0 and (rel < 5e-5 or not ro["converged"]) and abs(int(r["num_inliers"]) - int(ro["num_inliers"])) <= 8 and abs(int(r["iterations"]) - int(ro["iterations"])) <= 1:
            tie_cases.append(("case 2:%d" % ci, float(rel), nt))
        else:
            fails.append(("gicp", "case 2:%d" % ci, s, w, h, tr_, rd_, len(c0), len(c1), rel, r["iterations"], ro["iterations"], r["num_inliers"], ro["num_inliers"], nt, same))
print("gicp cases", ngicp, "largest pose error", worst, "pairs over the bar through a k-th-distance tie", tie_cases, "fails", len(fails), flush=True)

# ---- 3. ORB on odd sizes
norb = 0
i3 = 0
while time.time() - T0 < budget * 0.7 and ONLY is None and 3 in SECTIONS:
    rng = case_rng(3, i3)
    i3 += 1
    W, H = int(rng.integers(200, 900)), int(rng.integers(160, 640))
    nf = int(rng.choice([300, 500, 1000, 1500, 2000])); nl = int(rng.choice([4, 6, 8])); sf = float(rng.choice([1.2, 1.25, 1.5]))
    s = int(rng.integers(0, 1 << 30))
    img = synth.frame_pair(s, W, H, 4)["gray0"]
    if rng.integers(0, 5) == 0: img = (img // 8 * 8).astype(np.uint8)  # flat-ish
    try:
        ext = api.ORBextractor(nf, sf, nl, 20, 7, max_rows=H, max_cols=W)
        _, k, d = ext(img)
        orc = O.OrbOracle(nf, sf, nl, 20, 7); _, ko, do = orc.extract(img)
        norb += 1
        if len(k) != len(ko) or not (k == ko).all() or not (d == do).all():
            fails.append(("orb", "case 3:%d" % (i3 - 1), s, W, H, nf, nl, sf, len(k), len(ko)))
        ext.close() if hasattr(ext, "close") else None
    except Exception as e:
        if "unsupported" not in repr(e):
            fails.append(("orb-exc", s, W, H, nf, nl, sf, repr(e)[:200]))
print("orb cases", norb, "fails", len(fails), flush=True)

# ---- 4. LBA random windows (+ duplicate edges)
import test_gpu_lba as TL
opt = api.Optimizer(max_poses=128, max_points=4096, max_edges=300000)
nlba = 0
i4 = 0
while time.time() - T0 < budget and ONLY is None and 4 in SECTIONS:
    rng = case_rng(4, i4)
    i4 += 1
    s = int(rng.integers(0, 1 << 30))
    nfree, nfix, npts = int(rng.integers(1, 70)), int(rng.integers(1, 6)), int(rng.integers(10, 1500))
    w = synth.lba_window(s, n_free=nfree, n_fixed=nfix, n_points=npts, mono_frac=float(rng.choice([0.0, 0.1, 1.0])))
    if rng.integers(0, 3) == 0: w = TL._with_second_camera_edges(w, s, frac=float(rng.uniform(0.01, 0.3)))
    if rng.integers(0, 5) == 0: w["iterations"] = int(rng.integers(1, 6))
    try:
        r = opt.LocalBundleAdjustment(w); ro = O.lba_solve(w)
        nlba += 1
        rel = lambda a, b: np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)
        if not (r["iterations_run"] == ro["iterations_run"] and rel(r["pose_q"], ro["pose_q"]) < 1e-5 and rel(r["pose_t"], ro["pose_t"]) < 1e-5
                and rel(r["points"], ro["points"]) < 1e-5 and rel(r["final_chi2"], ro["final_chi2"]) < 1e-6):
            fails.append(("lba", "case 4:%d" % (i4 - 1), s, nfree, nfix, npts, w["n_edges"], r["iterations_run"], ro["iterations_run"], rel(r["points"], ro["points"]),
                          rel(r["final_chi2"], ro["final_chi2"])))
    except Exception as e:
        fails.append(("lba-exc", s, nfree, nfix, npts, repr(e)[:200]))
print("lba cases", nlba, "fails", len(fails), flush=True)
# ---- 5. the "next" rows: SearchByProjection, PoseOptimization, BF match, fbKltTracking, findFundamentalMat (a fifth of the budget more)
T1 = time.time()
extra = budget * 0.25
nx = dict(sbp=0, pose=0, match=0, klt=0, fmat=0)
pm = api.ProjectionMatcher(max_last=2048, max_cur=2560, max_batch=1)
po = api.PoseOptimizer(max_obs=2048, max_batch=1, sums="edge_order")  # (the bit-for-bit comparison below is a statement about g2o's sum order)
po_tree = api.PoseOptimizer(max_obs=2048, max_batch=1)
mt = api.ORBmatcher()
fm = api.FundamentalMatcher(max_points=2048, max_batch=1)
ext = api.ORBextractor(1000, 1.2, 8, 20, 7, max_rows=480, max_cols=640)
trk = api.KltTracker(640, 480, 35, max_batch=1, max_points=2048)
i5 = 0
while time.time() - T1 < extra and ONLY is None and 5 in SECTIONS:
    rng = case_rng(5, i5)
    i5 += 1
    s = int(rng.integers(0, 1 << 30))
    which = int(rng.integers(0, 5))
    try:
        if which == 0:
            q = synth.sbp_pair(s, n_points=int(rng.integers(1, 1800)), n_extra_cur=int(rng.integers(0, 500)), motion=float(rng.uniform(-0.3, 0.3)),
                               rot_deg=float(rng.uniform(0, 2)), th=float(rng.choice([3.0, 7.0, 15.0])), mono=bool(rng.integers(0, 2)),
                               dup_frac=float(rng.uniform(0, 0.5)), zero_obs_frac=float(rng.uniform(0, 1)), preassigned_frac=float(rng.uniform(0, 0.2)))
            m, n = pm.SearchByProjection(q); mo, no = O.search_by_projection(q)
            nx["sbp"] += 1
            if n != no or not np.array_equal(m, mo): fails.append(("sbp", "case 5:%d" % (i5 - 1), s, n, no))
        elif which == 1:
            q = synth.pose_frame(s, n_obs=int(rng.integers(0, 1500)), mono_frac=float(rng.choice([0.0, 0.15, 1.0])), outlier_frac=float(rng.uniform(0, 0.4)),
                                 rot_deg=float(rng.uniform(0, 3)), trans=float(rng.uniform(0, 0.1)))
            r, ro = po.PoseOptimization(q), O.pose_optimization(q)
            nx["pose"] += 1
            rel = lambda a, b: np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)
            # (the total LM iteration count may differ by one or two: at a converged state the sign of rho of a null step is rounding
            # noise; counted, not failed)
            nx["pose_iteration_counts_differ"] = nx.get("pose_iteration_counts_differ", 0) + (r["iterations_run"] != ro["iterations_run"])
            nx["pose_not_bit_identical"] = nx.get("pose_not_bit_identical", 0) + (not (np.array_equal(np.asarray(r["q"]).view(np.uint64), np.asarray(ro["q"]).view(np.uint64)) and np.array_equal(np.asarray(r["t"]).view(np.uint64), np.asarray(ro["t"]).view(np.uint64))))
            if not (np.array_equal(r["outlier"], ro["outlier"]) and r["n_inliers"] == ro["n_inliers"] and r["rounds_run"] == ro["rounds_run"]
                    and r["iterations_run"] == ro["iterations_run"] and rel(r["q"], ro["q"]) < 1e-5 and rel(r["t"], ro["t"]) < 1e-5):
                fails.append(("pose", "case 5:%d" % (i5 - 1), s, q["n_obs"], r["n_inliers"], ro["n_inliers"], r["iterations_run"], ro["iterations_run"]))
            # ... and the handle's default: fixed-shape tree sums, un-pivoted solve -- pose within the bar, every flipped flag a proved
            # chi2-threshold tie (tests/test_gpu_pose.py has the rule)
            rt = po_tree.PoseOptimization(q)
            flip = np.flatnonzero(rt["outlier"] != ro["outlier"])
            tie = all(min(abs(float(ro["chi2"][e]) - 5.991), abs(float(ro["chi2"][e]) - 7.815)) <= 1e-6 * 7.815 for e in flip)
            nx["pose_tree_frames_with_flips"] = nx.get("pose_tree_frames_with_flips", 0) + (len(flip) > 0)
            if not (tie and abs(rt["n_inliers"] - ro["n_inliers"]) <= len(flip) and rt["rounds_run"] == ro["rounds_run"]
                    and (q["n_obs"] < 3 or (rel(rt["q"], ro["q"]) < 1e-5 and rel(rt["t"], ro["t"]) < 1e-5))):
                fails.append(("pose-tree", "case 5:%d" % (i5 - 1), s, q["n_obs"], rt["n_inliers"], ro["n_inliers"], len(flip)))
        elif which == 2:
            nq, nt = int(rng.integers(0, 3000)), int(rng.integers(0, 3000))
            dq = rng.integers(0, 256, (nq, 32), dtype=np.uint8); dt = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
            if nq and nt and rng.integers(0, 2): dt[rng.integers(0, nt, nt // 3)] = dq[rng.integers(0, nq, nt // 3)]  # exact duplicates: distance ties
            ti, di = mt.match(dq, dt); to, do = O.bf_match(dq, dt)
            nx["match"] += 1
            if not (np.array_equal(ti, to) and np.array_equal(di, do)): fails.append(("match", "case 5:%d" % (i5 - 1), s, nq, nt))
        elif which == 3:
            fp = synth.frame_pair(s, 640, 480, 8)
            _, k0, _ = ext(fp["gray0"])
            kps = np.stack([k0["x"], k0["y"]], 1).astype(np.float32)
            pri = kps + np.float32(rng.uniform(-2, 2))
            lvl = int(rng.choice([0, 3, 6]))
            p0, p1 = trk.buildOpticalFlowPyramid(fp["gray0"]), trk.buildOpticalFlowPyramid(fp["gray1"])
            o0, o1 = O.klt_build_pyramid(fp["gray0"], 35), O.klt_build_pyramid(fp["gray1"], 35)
            g = trk.fbKltTracking(p0, p1, lvl, 15.0, 0.5, kps, pri); o = O.fb_klt_tracking(o0, o1, 640, 480, 35, lvl, 15.0, 0.5, kps, pri)
            nx["klt"] += 1
            if not (g[2] == o[2] and np.array_equal(g[1], o[1]) and np.array_equal(g[0].view(np.uint32), o[0].view(np.uint32))):
                fails.append(("klt", "case 5:%d" % (i5 - 1), s, lvl, g[2], o[2]))
        else:
            n = int(rng.integers(8, 1500))
            a, b, _, _ = synth.two_view_points(s, n, float(rng.uniform(0, 0.7)), float(rng.uniform(0, 1.0)))
            thr, conf = float(rng.choice([0.5, 1.5, 3.0])), float(rng.choice([0.99, 0.999]))
            g = fm.findFundamentalMat(a, b, thr, conf); o = O.fundamental_ransac(a, b, thr, conf)
            nx["fmat"] += 1
            same_F = (g[1] is None) == (o[1] is None) and (g[1] is None or np.array_equal(g[1].view(np.uint64), o[1].view(np.uint64)))
            if not (g[2] == o[2] and np.array_equal(g[0], o[0]) and same_F): fails.append(("fmat", "case 5:%d" % (i5 - 1), s, n, thr, conf, g[2], o[2]))
    except Exception as e:
        fails.append(("next-exc", "case 5:%d" % (i5 - 1), which, s, repr(e)[:200]))
print("next rows", nx, "fails", len(fails), flush=True)
# ---- 6. mixed LBA batches against the same windows solved alone (bit for bit)
T2 = time.time()
nb6 = 0
i6 = 0
if 6 in SECTIONS and ONLY is None:
    bat = api.BatchOptimizer(max_windows=8, max_poses=64, max_points=1024, max_edges=100000)
    one = api.Optimizer(max_poses=64, max_points=1024, max_edges=100000)
    while time.time() - T2 < budget * 0.15:
        rng = case_rng(6, i6)
        i6 += 1
        wins = []
        for _ in range(int(rng.integers(1, 9))):
            w = synth.lba_window(int(rng.integers(0, 1 << 30)), n_free=int(rng.integers(1, 46)), n_fixed=int(rng.integers(1, 5)),
                                 n_points=int(rng.integers(5, 800)), mono_frac=float(rng.choice([0.0, 0.1, 1.0])))
            kind = int(rng.integers(0, 8))
            if kind == 0: w = TL._with_second_camera_edges(w, int(rng.integers(0, 1000)), frac=float(rng.uniform(0.02, 0.3)))
            if kind == 1: w["iterations"] = int(rng.integers(0, 4))
            if kind == 2: w = dict(w, pose_fixed=np.ones_like(w["pose_fixed"]))
            wins.append(w)
        try:
            got = bat.LocalBundleAdjustment(wins)
            nb6 += 1
            for k, (w, r) in enumerate(zip(wins, got)):
                r1 = one.LocalBundleAdjustment(w)
                okk = all(np.array_equal(r[key], r1[key]) for key in ("pose_q", "pose_t", "points", "edge_chi2", "edge_depth_positive"))
                if not (okk and r["iterations_run"] == r1["iterations_run"] and r["final_chi2"] == r1["final_chi2"]):
                    fails.append(("lba-batch", "case 6:%d" % (i6 - 1), k, len(wins), w["n_poses"], w["n_points"], w["n_edges"], w.get("iterations", 10),
                                  r["iterations_run"], r1["iterations_run"]))
        except Exception as e:
            fails.append(("lba-batch-exc", "case 6:%d" % (i6 - 1), repr(e)[:200]))
print("lba batches", nb6, "fails", len(fails), flush=True)
# ---- 7. ragged GICP batches through the device entry against the same pairs one at a time (bit for bit)
T3 = time.time()
nb7 = 0
i7 = 0
if 7 in SECTIONS and ONLY is None:
    from test_gpu_gms import _Hip
    hip = _Hip()
    single = api.RegistrationGICP(max_points=16384, max_batch=1)
    while time.time() - T3 < budget * 0.15:
        rng = case_rng(7, i7)
        i7 += 1
        B = int(rng.integers(2, 17))
        triples = []
        for b in range(B):
            c0, c1, T01 = synth.cloud_pair(int(rng.integers(0, 1 << 30)), int(rng.choice([96, 128, 160])), int(rng.choice([72, 96])),
                                           trans=float(rng.uniform(0, 0.3)), rot_deg=float(rng.uniform(0, 8)))
            kind = int(rng.integers(0, 10))
            if kind == 0: c0 = c0[:0]
            if kind == 1: c1 = c1[:0]
            if kind == 2: c0 = c0[:int(len(c0) * rng.uniform(0.05, 0.9))]
            if kind == 3: c1 = c1[::int(rng.integers(2, 5))]
            if kind == 4: c0, c1 = c0[:7], c1[:9]
            init = T01 @ synth.random_motion(rng, 0.01, 0.5) if kind == 5 else np.eye(4)
            triples.append((c0, c1, init))
        SP = (max(max(len(a), len(b_)) for a, b_, _ in triples) + 1023) // 1024 * 1024
        a0, a1 = np.zeros((B, SP, 4), np.float32), np.zeros((B, SP, 4), np.float32)
        n0, n1, init = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros((B, 4, 4))
        for b, (a, s_, T0_) in enumerate(triples):
            a0[b, :len(a)], a1[b, :len(s_)], n0[b], n1[b], init[b] = a, s_, len(a), len(s_), T0_
        try:
            regb = api.RegistrationGICP(max_points=SP, max_batch=B)
            ptrs = [hip.to_device(x) for x in (a0, n0, a1, n1)]
            got = regb.align_batch_device(ptrs[0], ptrs[1], ptrs[2], ptrs[3], B, SP, init_T=init)
            nb7 += 1
            for b, ((a, s_, T0_), r) in enumerate(zip(triples, got)):
                r1 = single.RegisterPointClouds(a, s_, T0_)
                if not (np.array_equal(r["T"], r1["T"]) and r["iterations"] == r1["iterations"] and r["num_inliers"] == r1["num_inliers"]
                        and np.array_equal(r["H"], r1["H"]) and r["error"] == r1["error"]):
                    fails.append(("gicp-batch", "case 7:%d" % (i7 - 1), b, B, len(a), len(s_), r["iterations"], r1["iterations"]))
            regb.close() if hasattr(regb, "close") else None
        except Exception as e:
            fails.append(("gicp-batch-exc", "case 7:%d" % (i7 - 1), repr(e)[:200]))
        hip.free()
        hip = _Hip()
print("gicp batches", nb7, "fails", len(fails), flush=True)
for f in fails[:40]: print("FAIL", f)
sys.exit(1 if fails else 0)
