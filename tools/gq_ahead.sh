#!/bin/bash
# round 6: LM rounds queued ahead of the host's poll (GFS_GICP_AHEAD), and the own-row walk of k_knn_cov against the full-row scan
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_gicp.py tests/test_gpu_batched.py 2>&1 | tail -3
q() { timeout 600 python bench.py --no-cpu-baseline --no-extras --no-klt --verify 0 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"])'; }
for a in 1 2 3 4 6; do
  echo "== ahead $a: c4 2 lanes $(GFS_GICP_AHEAD=$a q --batch 64 --lanes 2) | 4 lanes $(GFS_GICP_AHEAD=$a q --batch 64 --lanes 4) | headline $(GFS_GICP_AHEAD=$a q)"
done
echo "== serial, walk"; bash tools/gq.sh serial | python3 -c 'import sys,json
d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["value"], d["frac"], {n:k[n] for n in k if "knn" in n or "gicp" in n})'
bash tools/variant.sh run knnscan -- bash -c 'bash tools/gq.sh serial | python3 -c "import sys,json
d=json.loads(sys.stdin.read()); k=d[\"kernels\"]; print(d[\"value\"], d[\"frac\"], {n:k[n] for n in k if \"knn\" in n or \"gicp\" in n})"'
timeout 600 python tools/probes/gicp_chain_probe.py 1 32 2>&1 | grep "==" 
