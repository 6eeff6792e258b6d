cd $GRAFT_REPO_ROOT
for l in 1 2 3 4; do
  echo "c3 lanes $l: $(timeout 300 python bench.py --workload c3 --batch 32 --lanes $l --steps 10 --warmup 3 --no-klt --no-extras --no-cpu-baseline --verify 0 2>/dev/null | python -c 'import sys,json
for ln in sys.stdin:
    if ln.startswith("{"): d=json.loads(ln); print(d["value"], d["ms_per_step"])')"
done
for l in 1 2 3 4; do
  echo "c4 lanes $l: $(timeout 300 python bench.py --batch 64 --lanes $l --steps 40 --warmup 5 --no-klt --no-extras --no-cpu-baseline --verify 0 2>/dev/null | python -c 'import sys,json
for ln in sys.stdin:
    if ln.startswith("{"): d=json.loads(ln); print(d["value"], d["ms_per_step"])')"
done
