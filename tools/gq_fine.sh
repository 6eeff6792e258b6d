#!/bin/bash
# round 6: sub-cell bits of the cell sort key (the slack of the x-ordered walk), and what the waves of the walk do
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
bash tools/variant.sh run util -- timeout 300 python tools/probes/lin_util_probe.py 2>&1 | grep -v "^$"
bash tools/variant.sh run base fine3 fine5 fine6 -- bash -c 'timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_gicp.py -k "pose_parity_vga or streaming_form or edge_cases" 2>&1 | tail -1; bash tools/gq.sh serial | python3 -c "import sys,json
d=json.loads(sys.stdin.read()); k=d[\"kernels\"]; print(d[\"value\"], d[\"frac\"], {n:k[n] for n in k if \"gicp\" in n or \"knn\" in n or \"cell\" in n or \"grid\" in n})"'
