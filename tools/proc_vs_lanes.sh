cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-supervisor --batch $1 --lanes $2 --steps 60 --warmup 5 --no-klt --no-extras --no-cpu-baseline --verify 0 2>/dev/null | python -c 'import sys,json
for ln in sys.stdin:
    if ln.startswith("{"): d=json.loads(ln); print(d["value"], d["ms_per_step"])'; }
echo "1 proc, 64 pairs, 2 lanes: $(run 64 2)"
echo "2 procs x (32 pairs, 1 lane):"; (run 32 1 & run 32 1 & wait)
echo "4 procs x (16 pairs, 1 lane):"; (run 16 1 & run 16 1 & run 16 1 & run 16 1 & wait)
echo "1 proc, 64 pairs, 4 lanes: $(run 64 4)"
echo "2 procs x (256 pairs, 2 lanes):"; (run 256 2 & run 256 2 & wait)
echo "1 proc, 512 pairs, 4 lanes: $(run 512 4)"
