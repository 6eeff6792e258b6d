#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
q() { timeout 600 python bench.py --no-cpu-baseline --no-extras --no-klt --verify 0 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"])'; }
for hq in 4 8 16 24; do for l in 2 3 4; do echo "== HWQ $hq batch 64 lanes $l: $(GPU_MAX_HW_QUEUES=$hq q --batch 64 --lanes $l)"; done; done
echo "== headline HWQ 8: $(q)"
echo "== headline HWQ 16: $(GPU_MAX_HW_QUEUES=16 q)"
