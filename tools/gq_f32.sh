#!/bin/bash
# round 6: the single-precision 1-NN walk of the linearisation (nn_search27_f32) -- parity, then the kernel and the headline with it on / off
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_gicp.py -k "f32 or cooperative or streaming or pose_parity_vga or edge_cases or with_init or staged_tile or 720p" 2>&1 | tail -5
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_batched.py -k "gicp_batch" 2>&1 | tail -3
for v in 0 1; do
  echo "== serial GFS_GICP_NN_F32=$v"; bash tools/gq.sh serial GFS_GICP_NN_F32=$v | python3 -c 'import sys,json
d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["value"], d["frac"], {n:k[n] for n in k if "gicp" in n or "grid_fill" in n})'
done
for v in 0 1 0 1; do echo "== headline GFS_GICP_NN_F32=$v: $(GFS_GICP_NN_F32=$v bash tools/gq.sh quick | cut -c1-120)"; done
GFS_GICP_TILE_STATS=1 timeout 300 python tools/tile_stats_probe.py 2>&1 | tail -5
timeout 600 python tools/probes/gicp_chain_probe.py 1 32 2>&1 | grep "==\|coop\|linearize\|step\|heap\|leaf\|top_lds\|knn\|cell\|grid\|reduce\|keys" 
timeout 300 python tools/stream_probe.py | head -24
