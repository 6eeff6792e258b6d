R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python tools/stream_overhead.py 2>&1 | head -8
python - <<'P'
import sys; sys.path.insert(0,'.')
import numpy as np
from geoflowslam_amd import api, synth
from oracle import oracle as O
import bench_stream as bs
W,H=640,480
fp = synth.frame_pair(1000, W, H, 4); frames=[(fp["gray0"],fp["depth0"]),(fp["gray1"],fp["depth1"])]
K = synth.intrinsics(W,H)
g = bs.GpuBackend(api, W, H, 1000, 8, 20480); o = bs.OracleBackend(O, W, H, 1000, 8)
lg,sg,stg = bs.run_stream(g, frames, K, W, H, 4, 12, warm=1)
lo,so,sto = bs.run_stream(o, frames, K, W, H, 4, 12, warm=1); o.close()
same = all(a["matches"]==b["matches"] and a["inliers"]==b["inliers"] and np.linalg.norm(a["T"]-b["T"])<=1e-5*np.linalg.norm(b["T"]) and np.array_equal(a["match"],b["match"]) for a,b in zip(stg,sto))
print("agrees_with_oracle", same, bs.summarize(lo,so)["stages_median_ms"])
P
