#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_orb_match.py tests/test_gpu_host_mirror.py tests/test_gpu_batched.py 2>&1 | tail -3
bash tools/gq.sh serial | python3 -c 'import sys,json
d=json.loads(sys.stdin.read()); k=d["kernels"]; print(d["value"], {n:k[n] for n in k if n in ("k_blur7","k_orient_brief","k_pyr_area_lds","k_fast_cells","k_octree")})'
bash profiles/pmc_kernel.sh k_orient_brief ob "FETCH_SIZE" 2>&1 | grep -v "^$"; GFS_ORB_OB_XCD=0 bash profiles/pmc_kernel.sh k_orient_brief ob0 "FETCH_SIZE" 2>&1 | grep -v "^$"; echo "== xcd off: $(GFS_ORB_OB_XCD=0 bash tools/gq.sh serial | cut -c1-400)"
echo "== headline: $(bash tools/gq.sh quick | cut -c1-100)"
