#!/bin/bash
# A/B variants of one translation unit without touching the product library.
#   here (CPU container):  bash tools/variant.sh build <name> <unit: orb|gicp|lba|...> [-DFLAG ...]   -> geoflowslam_amd/variants/<name>.so
#   GPU box:               gpurun -- 'bash tools/variant.sh run <name> [name ...] -- <command>'   runs <command> once per variant with the
#                          variant in place of libgfs_hip.so (restored afterwards); "base" = the product library
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mode=$1; shift
if [ "$mode" = build ]; then
  name=$1; unit=$2; shift 2
  mkdir -p $R/geoflowslam_amd/variants
  cd $R/geoflowslam_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=16 -ffp-contract=off "$@" -c $unit.hip -o /tmp/${unit}_$name.o || exit 1
  objs=""
  for o in gfs_common match orb gicp lba frame pose sbp gms klt fmat orb_host; do
    if [ $o = $unit ]; then objs="$objs /tmp/${unit}_$name.o"; else objs="$objs $o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$name.so $objs
  exit $?
fi
names=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
shift
cd $R
cp geoflowslam_amd/libgfs_hip.so /tmp/libgfs_hip.keep
for n in "${names[@]}"; do
  if [ $n = base ]; then cp /tmp/libgfs_hip.keep geoflowslam_amd/libgfs_hip.so; else cp geoflowslam_amd/variants/$n.so geoflowslam_amd/libgfs_hip.so; fi
  echo "=== variant $n"
  "$@"
done
cp /tmp/libgfs_hip.keep geoflowslam_amd/libgfs_hip.so
