#!/bin/bash
# k_fast_cells phase clocks.  Step 1 (here, CPU container):  bash tools/fast_timing.sh build   -> scratchless gpurun_out/.. no: geoflowslam_amd/libgfs_hip_timing.so
# Step 2 (GPU box):  gpurun -- 'bash tools/fast_timing.sh run'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "${1:-run}" = build ]; then
  cd $R/geoflowslam_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=16 -ffp-contract=off -DGFS_FAST_TIMING -c orb.hip -o /tmp/orb_t.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgfs_hip_timing.so gfs_common.o match.o /tmp/orb_t.o gicp.o lba.o frame.o pose.o sbp.o gms.o klt.o fmat.o orb_host.o
  exit $?
fi
cd $R
cp geoflowslam_amd/libgfs_hip.so /tmp/libgfs_hip.keep
cp geoflowslam_amd/libgfs_hip_timing.so geoflowslam_amd/libgfs_hip.so
timeout 600 python bench.py --no-supervisor --steps 1 --warmup 0 --prime 0 --no-cpu-baseline --no-extras --no-klt --verify 0 --lanes 1 --serial --batch 64 2>&1 | grep FASTT > gpurun_out/fastt.log
cp /tmp/libgfs_hip.keep geoflowslam_amd/libgfs_hip.so
python3 - <<'PY'
import re,collections
acc=collections.defaultdict(lambda: collections.Counter()); n=collections.Counter()
for l in open('gpurun_out/fastt.log'):
    d={k:int(v) for k,v in re.findall(r'(\w+)=(-?\d+)',l)}
    n[d['lvl']]+=1; acc[d['lvl']].update(d)
tot=collections.Counter()
for lvl in sorted(acc):
    a=acc[lvl]; print(lvl, n[lvl], {k:round(v/n[lvl]) for k,v in a.items() if k!='lvl'}); tot.update(a)
N=max(1,sum(n.values())); print('all',N,{k:round(v/N) for k,v in tot.items() if k!='lvl'})
PY
