#!/bin/bash
# Builds oracle/_ref/libgfs_ref_small_gicp.so from the reference's own headers where they lie under /root/reference
# (oracle/ref_small_gicp.cpp is only a C wrapper).  Flags follow the reference build (CMakeLists.txt:35-40: -O3, OpenMP).
# Everything else of the hot path needs OpenCV / Eigen, which are not in this image: unbuildable here.
set -e
cd "$(dirname "$0")"
REF=/root/reference/Thirdparty/small_gicp/include
[ -f "$REF/small_gicp/util/sort_omp.hpp" ] || { echo "ref_build: no reference tree, keeping the prebuilt oracle/_ref"; exit 0; }
mkdir -p _ref
g++ -O3 -std=c++17 -fPIC -fopenmp -shared -I"$REF" ref_small_gicp.cpp -o _ref/libgfs_ref_small_gicp.so
echo "ref_build: oracle/_ref/libgfs_ref_small_gicp.so"
