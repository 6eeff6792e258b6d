#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_gicp.py tests/test_gpu_batched.py tests/test_gpu_host_mirror.py 2>&1 | tail -3
q() { timeout 600 python bench.py --no-cpu-baseline --no-extras --no-klt --verify 0 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"])'; }
for i in 1 2; do echo "== c4 2 lanes $(q --batch 64 --lanes 2) | headline $(q)"; done
timeout 600 python tools/probes/gicp_chain_probe.py 1 32 2>&1 | grep "==" 
timeout 300 python tools/stream_probe.py | head -1
