/* gfs_abi_test.h — test hooks of libgfs_hip.so.  NOT part of the drop-in boundary (include/gfs_abi.h): nothing a maintainer of the
 * reference binds.  They expose internal replicas of third-party behaviour (libstdc++ std::sort, small_gicp's quick_sort_omp, glibc's
 * sin / cos / pow) and the counter-calibration kernels so that tests/ and profiles/calibrate.sh can check them in isolation.
 */
#ifndef GFS_ABI_TEST_H_
#define GFS_ABI_TEST_H_

#include "gfs_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host test hooks for the libstdc++ std::sort replica used by the device quadtree (sorts (size, x) pairs in place). */
int gfs_test_sort_replica(int32_t* size_key, int32_t* x_key, int32_t* payload, int n);
int gfs_test_heap_sort_replica(int32_t* size_key, int32_t* x_key, int32_t* payload, int n);

/* Host test hook: the order in which k_blur7 takes the 64 x 32 tiles of a frame's pyramid (rows x cols, nlevels levels at
 * scale_factor): level_tx_ty[3 i ..] = (level, tile column, tile row) of list position i, at most cap entries; returns the number of
 * tiles.  The 2 x 2 tiles over the same 128-byte lines sit at positions p, p + 8, p + 16, p + 24 (one XCD's L2 serves them). */
int gfs_test_orb_blur_tiles(int rows, int cols, int nlevels, float scale_factor, int32_t* level_tx_ty, int cap);

/* GPU test hook: sin(x), cos(x), pow(x, 3.0) of n doubles evaluated on the device with the restated glibc 2.35 arithmetic
 * (csrc/glibc_math.hpp) that the pose / window / registration optimizers use for SE3Quat::exp
 * (Thirdparty/g2o/g2o/types/se3quat.h:223-257) and the Levenberg step control (core/optimization_algorithm_levenberg.cpp:127). */
int gfs_test_glibc_math(int device, const double* x, int n, double* sin_out, double* cos_out, double* pow3_out);
/* GPU test hook for calibrating the HBM counters (profiles/calibrate.sh): a kernel with a KNOWN byte count -- mode 0 streaming read,
 * 1 per-lane gathers of 32-byte records out of a table of `table` records, 2 streaming write; n records (mode 1: n threads x per_thread
 * gathers).  *bytes_out = the bytes the kernel asked for. */
int gfs_test_traffic(int device, int mode, long long n, long long table, int per_thread, long long* bytes_out);

/* GPU test hook: the voxel sort of the preprocessing — the device replica of small_gicp's quick_sort_omp
 * (util/sort_omp.hpp:58-85: 3-way quicksort above 1024 elements, libstdc++ std::sort below), whose permutation of
 * equal keys decides the 1024-block splits of voxelgrid_sampling_omp (util/downsampling_omp.hpp:57-90) — on n <=
 * max_points caller keys (3 x 21-bit voxel fields, or all ones = invalid).  perm_out[i] = input index of the i-th
 * element of the sorted sequence. */
int gfs_test_voxel_sort(gfs_gicp* h, const unsigned long long* keys, int n, unsigned* perm_out);
/* GPU test hook: the one-wave std::sort replica (csrc/wave_std_sort.hpp: the voxel sort's leaves, sort_omp.hpp:61, and the quadtree's
 * (size, x) list, ORBextractor.cc:697-698) on n <= 1024 caller keys; perm_out[i] = original index of the element left at position i. */
int gfs_test_wave_std_sort(int device, const unsigned* keys, int n, unsigned short* perm_out);

#ifdef __cplusplus
}
#endif
#endif /* GFS_ABI_TEST_H_ */
