// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see gfs_oracle.h header). PARITY UNPINNED.
//
// CPU restatement of the brute-force Hamming matching used by ORBmatcher::SearchWithGMS /
// SearchForInitializationWithGMS / SearchForTriangulationWithGMS (reference src/ORBmatcher.cc:755-756,
// 805-806, 888-889: cv::BFMatcher(NORM_HAMMING).match) and of ORBmatcher::DescriptorDistance
// (src/ORBmatcher.cc:2536-2550).
#include <climits>
#include <cstdint>
#include <cstring>

#include "gfs_oracle.h"

extern "C" {

// src/ORBmatcher.cc:2536-2550 — SWAR popcount over 8 x 32-bit words.
int gfso_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t wa, wb;
    std::memcpy(&wa, a + 4 * i, 4);
    std::memcpy(&wb, b + 4 * i, 4);
    unsigned int v = wa ^ wb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// cv::BFMatcher::knnMatchImpl with k=1, no mask, crossCheck=false (OpenCV 4.5.4
// modules/features2d/src/matchers.cpp -> batchDistance(..., NORM_HAMMING, K=1)): for every query row the
// train row with minimum distance, first (lowest index) minimum wins (strict '<' update from INT_MAX).
int gfso_bf_match_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* train_idx, int32_t* dist,
                          int nthreads) {
  if (nt <= 0 || nq <= 0) return 0;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int i = 0; i < nq; i++) {
    int best = INT_MAX, bestj = -1;
    uint64_t qa[4];
    std::memcpy(qa, q + (size_t)i * 32, 32);
    for (int j = 0; j < nt; j++) {
      uint64_t tb[4];
      std::memcpy(tb, t + (size_t)j * 32, 32);
      int d = __builtin_popcountll(qa[0] ^ tb[0]) + __builtin_popcountll(qa[1] ^ tb[1]) +
              __builtin_popcountll(qa[2] ^ tb[2]) + __builtin_popcountll(qa[3] ^ tb[3]);
      if (d < best) {
        best = d;
        bestj = j;
      }
    }
    train_idx[i] = bestj;
    dist[i] = best;
  }
  return nq;
}

}  // extern "C"
