#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
echo "nproc $(nproc)"; lscpu | grep -i "model name\|^CPU(s)\|Thread" | head -4
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_gicp.py -k "cooperative or streaming or pose_parity_vga" 2>&1 | tail -3
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_batched.py -k "gicp_batch" 2>&1 | tail -3
q() { timeout 600 python bench.py --no-cpu-baseline --no-extras --no-klt --verify 0 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"])'; }
export GFS_GICP_COOP_TAIL=0.0001
for b in 16 32 64; do for l in 1 2 4; do echo "== batch $b lanes $l: $(q --batch $b --lanes $l)"; done; done
echo "== batch 64 lanes 2 serial: $(q --batch 64 --lanes 2 --serial)"
echo "== batch 32 lanes 1 step-join: $(q --batch 32 --lanes 1 --step-join)"
