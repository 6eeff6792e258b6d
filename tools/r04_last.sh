R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_orb_match.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160
