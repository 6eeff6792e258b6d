#!/bin/bash
# round 6: the cooperative LM kernel -- parity tests that touch the LM loop, then the single-stream and 64-pair-block figures
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_gicp.py -k "cooperative or streaming or pose_parity_vga or edge_cases or with_init" 2>&1 | tail -8
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_batched.py -k "gicp_batch" 2>&1 | tail -5
echo "== single stream"
timeout 600 python tools/stream_probe.py 2>&1 | head -8
q() { timeout 600 python bench.py --no-cpu-baseline --no-extras --no-klt --verify 0 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"])'; }
for t in 0 1 2 4 8; do
  echo "== c4 shard (64 pairs, 2 lanes) tail factor $t (0 = coop off)"
  if [ $t = 0 ]; then GFS_GICP_COOP=0 q --batch 64 --lanes 2; else GFS_GICP_COOP_TAIL=$t q --batch 64 --lanes 2; fi
done
echo "== c4 4 lanes tail 2"; GFS_GICP_COOP_TAIL=2 q --batch 64 --lanes 4
echo "== headline coop off"; GFS_GICP_COOP=0 q
echo "== headline tail 2"; GFS_GICP_COOP_TAIL=2 q
echo "== headline tail 1"; GFS_GICP_COOP_TAIL=1 q
