# per-kernel totals of one serial step for a list of GFS_GICP_CELL values:  gpurun -- 'bash tools/cell_probe.sh 0.1 0.07 0.05'
export GFS_BENCH_NO_SUPERVISOR=1  # the profiler must see the process that launches the kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in "$@"; do
rm -rf $R/gpurun_out/cell_trace
GFS_GICP_CELL=$C timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cell_trace -- python $R/bench.py --steps 1 --warmup 1 --serial --lanes 1 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 0 > /dev/null 2>&1
python - $R/gpurun_out/cell_trace $C <<'PY'
import csv, glob, sys, re
rows={}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(k_\w+)', r["Kernel_Name"]); n=m.group(1) if m else r["Kernel_Name"][:30]
        rows[n]=rows.get(n,0)+(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
keys=["k_knn_cov","k_knn_cov_far","k_gicp_linearize","k_gicp_error","k_cell_build","k_grid_fill","k_radix_sort"]
print(sys.argv[2], {k: round(rows.get(k,0)/2/1e3,3) for k in keys}, "gicp_sum_ms", round(sum(v for k,v in rows.items() if k in keys)/2/1e3,3))
PY
done
