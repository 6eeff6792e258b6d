// oracle/_ref: the two translation-unit-free pieces of the REFERENCE that compile here from their own sources
// (everything else on the hot path includes OpenCV and/or Eigen, which this image does not have):
//   /root/reference/Thirdparty/small_gicp/include/small_gicp/util/sort_omp.hpp   quick_sort_omp (the voxel sort of
//       voxelgrid_sampling_omp, util/downsampling_omp.hpp:56, run with its own OpenMP tasks)
//   /root/reference/Thirdparty/small_gicp/include/small_gicp/ann/knn_result.hpp  KnnResult::push (the k-NN result container of
//       ann/kdtree.hpp:194-233)
// This file is only the C wrapper; the headers are compiled where they lie (oracle/ref_build.sh), nothing of the reference is
// copied into the repository.  TEST INFRASTRUCTURE: used by tests/test_oracle_ref.py to pin the oracle's restatement of these
// two pieces (and, through it, the device voxel sort) against the reference's own code.
#include <small_gicp/ann/knn_result.hpp>
#include <small_gicp/util/sort_omp.hpp>

#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

extern "C" {

// exactly the call of util/downsampling_omp.hpp:56: (key, point index) pairs, comparator on the key only
void gfsref_quick_sort_omp(std::uint64_t* keys_io, std::uint64_t* idx_io, int n, int num_threads) {
  std::vector<std::pair<std::uint64_t, size_t>> coord_pt(n);
  for (int i = 0; i < n; i++) coord_pt[i] = {keys_io[i], (size_t)idx_io[i]};
  small_gicp::quick_sort_omp(coord_pt.begin(), coord_pt.end(), [](const auto& lhs, const auto& rhs) { return lhs.first < rhs.first; }, num_threads);
  for (int i = 0; i < n; i++) {
    keys_io[i] = coord_pt[i].first;
    idx_io[i] = coord_pt[i].second;
  }
}

// a stream of (index, distance) pushes into KnnResult<-1> (capacity k) or KnnResult<1>; returns num_found
int gfsref_knn_push_stream(int k, int static_one, const std::uint64_t* index, const double* distance, int n, std::uint64_t* idx_out,
                           double* dist_out) {
  std::vector<size_t> idx(k);
  std::vector<double> d(k);
  int found;
  if (static_one) {
    small_gicp::KnnResult<1> r(idx.data(), d.data());
    for (int i = 0; i < n; i++) r.push((size_t)index[i], distance[i]);
    found = (int)r.num_found();
  } else {
    small_gicp::KnnResult<-1> r(idx.data(), d.data(), k);
    for (int i = 0; i < n; i++) r.push((size_t)index[i], distance[i]);
    found = (int)r.num_found();
  }
  for (int i = 0; i < k; i++) {
    idx_out[i] = idx[i];
    dist_out[i] = d[i];
  }
  return found;
}

}  // extern "C"
