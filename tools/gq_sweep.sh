#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_gicp.py -k "wide_extent or streaming or edge_cases or pose_parity_vga" 2>&1 | tail -2
q() { timeout 600 python bench.py --no-cpu-baseline --no-extras --no-klt --verify 0 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"])'; }
for l in 2 3 4 5 6 8; do echo "== lanes $l: $(q --lanes $l)"; done
for l in 4 6 8; do echo "== lanes $l, 16 hw queues: $(GPU_MAX_HW_QUEUES=16 q --lanes $l)"; done
FUZZ_SECTIONS=5,6,7 timeout 700 python tests/fuzz_gpu.py 600 7071 2>&1 | grep -v Warning | tail -12
