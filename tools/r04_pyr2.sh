R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python - <<'P'
from geoflowslam_amd import api
for sc, nl in ((2.0, 4), (2.5, 3), (1.5, 5)):
    try:
        api.ORBextractor(800, sc, nl, 20, 7, max_rows=480, max_cols=640); print(sc, "ok")
    except Exception as e:
        print(sc, "ERR", e)
P
timeout 1200 python -m pytest tests/test_gpu_orb_match.py -q -m gpu 2>&1 | tail -8
