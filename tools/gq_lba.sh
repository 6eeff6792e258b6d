#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_lba.py tests/test_lba_adaptor.py tests/test_gpu_batched.py -k "lba or Lba or LBA" 2>&1 | tail -3
for v in 0 1 0 1 0 1; do echo "== GFS_LBA_SPECULATE=$v"; GFS_LBA_SPECULATE=$v timeout 300 python tools/lba_probe.py 2>&1 | head -1; done
