R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for i in 1 2 3; do python tools/stream_probe.py 2>&1 | head -1 | cut -c1-60; done
for i in 1 2; do timeout 600 python bench.py --batch 64 --lanes 2 --steps 40 --warmup 3 --no-cpu-baseline --no-extras --no-klt --verify 0 --prime 5 2>&1 | tail -1 | grep -o '"value": [0-9.]*' | head -1; done
python tools/lba_probe.py 2>&1 | tail -2
