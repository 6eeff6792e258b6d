export GFS_BENCH_NO_SUPERVISOR=1  # the profiler must see the process that launches the kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 64 128 256 512; do
rm -rf $R/gpurun_out/ocb_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ocb_trace -- python $R/bench.py --batch $B --steps 1 --warmup 1 --serial --lanes 1 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 0 > /dev/null 2>&1
python - $R/gpurun_out/ocb_trace $B <<'PY'
import csv, glob, sys
for name in ("k_octree","k_fast_cells"):
    d=[]
    for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"]: d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(sys.argv[2], name, [round(x,1) for x in d])
PY
done
