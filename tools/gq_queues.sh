#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
q() { timeout 600 python bench.py --no-cpu-baseline --no-extras --no-klt --verify 0 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"])'; }
for t in 1 0; do for hq in 8 16; do
  echo "== torch streams $t, hw queues $hq: c4 2 lanes $(GFS_BENCH_TORCH_STREAMS=$t GPU_MAX_HW_QUEUES=$hq q --batch 64 --lanes 2) | 4 lanes $(GFS_BENCH_TORCH_STREAMS=$t GPU_MAX_HW_QUEUES=$hq q --batch 64 --lanes 4) | headline $(GFS_BENCH_TORCH_STREAMS=$t GPU_MAX_HW_QUEUES=$hq q)"
done; done
bash tools/gq_tl.sh c4l4raw --batch 64 --lanes 4 2>&1 | grep "^queue\|bench under"
