#!/bin/bash
# upper bound of any speed-up of the heap replay: the variant -DVQS_NO_POPS skips the pops (WRONG results, timing only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
q() { timeout 600 python bench.py --no-cpu-baseline --no-extras --no-klt --verify 0 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], {k:v for k,v in d["roofline"]["kernels_ms_per_step"].items() if "qsort" in k})'; }
bash tools/variant.sh run base nopops base nopops -- bash -c "$(declare -f q); echo \"headline \$(q)\"; echo \"c4 \$(q --batch 64 --lanes 2)\""
