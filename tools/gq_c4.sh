#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_gicp.py -k "cooperative or streaming or pose_parity_vga or edge_cases or with_init or staged_tile" 2>&1 | tail -3
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_batched.py -k "gicp_batch" 2>&1 | tail -3
q() { timeout 600 python bench.py --no-cpu-baseline --no-extras --no-klt --verify 0 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"])'; }
for b in 32 64; do for l in 1 2 4; do echo "== batch $b lanes $l: $(q --batch $b --lanes $l)"; done; done
echo "== headline: $(q)"
echo "== headline again: $(q)"
timeout 600 python tools/probes/gicp_chain_probe.py 1 32 | grep "==\|coop\|linearize\|step"
timeout 300 python tools/stream_probe.py | head -3
