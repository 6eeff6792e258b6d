"""Per-queue view of a rocprofv3 --kernel-trace of bench.py (a small batch): for every hardware queue (one per HIP stream of a lane's
ORB / GICP chain) the kernels of the timed region in order -- busy time, and the gaps between the end of a kernel and the start of the
next one on the SAME queue, split by what precedes the gap.  A latency-bound chain shows ~1.5 us gaps (the dependent-launch boundary);
larger gaps are the host (a launch not queued yet: polls, memcpy round trips, lock contention) or a full chip."""
import csv, glob, re, sys, collections
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(kb?_\w+|__amd_rocclr_\w+)", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40], r["Queue_Id"], r["Thread_Id"]))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t0 + int((t1 - t0) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.5))
ev = [e for e in ev if e[0] >= lo]
t0, t1 = ev[0][0], max(e[1] for e in ev)
print(f"span {1e-6*(t1-t0):.2f} ms, {len(ev)} dispatches")
byq = collections.defaultdict(list)
for e in ev:
    byq[e[3]].append(e)
for q, L in sorted(byq.items()):
    busy = sum(e[1] - e[0] for e in L)
    gaps = collections.Counter(); gapn = collections.Counter(); big = []
    for a, b in zip(L, L[1:]):
        g = b[0] - a[1]
        if g < 0: g = 0
        gaps[a[2]] += g; gapn[a[2]] += 1
        if g > 20000: big.append((g, a[2], b[2]))
    tot_gap = sum(gaps.values())
    names = collections.Counter(e[2] for e in L)
    kind = "gicp" if any("gicp" in n for n in names) else "orb" if any("fast" in n for n in names) else "?"
    print(f"queue {q} ({kind}, threads {sorted(set(e[4] for e in L))}): {len(L)} kernels, busy {busy*1e-6:.2f} ms, gaps {tot_gap*1e-6:.2f} ms "
          f"(median {(sorted((b[0]-a[1]) for a,b in zip(L,L[1:])) or [0])[(len(L)-1)//2]/1e3:.1f} us), span {(L[-1][1]-L[0][0])*1e-6:.2f} ms")
    for n, g in gaps.most_common(8):
        print(f"     gap behind {n:28s} {g*1e-6:7.3f} ms over {gapn[n]:4d} ({g/gapn[n]/1e3:6.1f} us each)")
    tot = collections.Counter(); cnt = collections.Counter()
    for e in L:
        tot[e[2]] += e[1] - e[0]; cnt[e[2]] += 1
    print("     busy:", ", ".join(f"{n} {v*1e-6:.2f}/{cnt[n]}" for n, v in tot.most_common(10)))
