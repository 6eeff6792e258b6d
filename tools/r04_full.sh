R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/r04_full_tests.log 2>&1; tail -5 $OUT/r04_full_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
