R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
b() { timeout 300 python $R/bench.py "$@" --no-cpu-baseline --no-extras --no-klt --verify 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'; }
for mode in exact stable exact stable; do
  export GFS_GICP_VOXEL_ORDER=$mode
  echo "$mode c4 $(b --batch 64 --lanes 2 --steps 40 --warmup 5) c3 $(b --workload c3 --batch 32 --lanes 2 --steps 40 --warmup 5) c2 $(b --steps 20 --warmup 3)"
done
