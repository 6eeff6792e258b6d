cd $GRAFT_REPO_ROOT
for l in 2 3 4 5 6 8; do
  echo "lanes $l: $(timeout 300 python bench.py --lanes $l --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-klt --verify 0 2>/dev/null | python -c 'import sys,json
for ln in sys.stdin:
    if ln.startswith("{"): d=json.loads(ln); print(d["value"], d["ms_per_step"])')"
done
