R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python bench.py --no-cpu-baseline --no-klt 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'roofline', d['roofline']['frac'])
for k in ('orb_only', 'single_stream', 'c3', 'c4_shard', 'lba'):
    e = d.get('extras', d).get(k) if isinstance(d.get('extras', d), dict) else None
    if e: print(k, {kk: e[kk] for kk in list(e)[:6] if not isinstance(e[kk], (dict, list))})
"
