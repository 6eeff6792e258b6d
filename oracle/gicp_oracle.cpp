// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see gfs_oracle.h header). PARITY UNPINNED, except quick_sort_impl and KnnResult below:
// pinned against the reference's own util/sort_omp.hpp and ann/knn_result.hpp compiled into oracle/_ref (tests/test_oracle_ref.py).
//
// CPU restatement of RegistrationGICP::RegisterPointClouds (reference src/RegistrationGICP.cc:5-20) and of the
// small_gicp code it reaches (Thirdparty/small_gicp/include/small_gicp/..., cited per function), all in double,
// single-threaded and deterministic.  Eigen pieces (3x3 inverse, computeDirect, LDLT, Quaternion ->
// matrix) are restated from Eigen 3.4.0.  std::sort / std::partition / std::nth_element are libstdc++'s.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <omp.h>

#include <vector>

#include "gfs_oracle.h"

namespace {

using V4 = std::array<double, 4>;
struct M3 {
  double m[9];  // column-major: m[r + 3*c]
  double& operator()(int r, int c) { return m[r + 3 * c]; }
  double operator()(int r, int c) const { return m[r + 3 * c]; }
};

struct Cloud {
  std::vector<V4> points;   // (x, y, z, 1)
  std::vector<V4> normals;  // (nx, ny, nz, 0)
  std::vector<M3> covs;     // 3x3 block of the 4x4 covariance (rest is zero)
};

// ---- Eigen 3.4 restatements -------------------------------------------------------------------------
M3 inverse3(const M3& a) {  // Eigen::Matrix3d::inverse(): cofactors / determinant (Inverse.h compute_inverse_size3)
  M3 r;
  auto cof = [&](int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return a(i1, j1) * a(i2, j2) - a(i1, j2) * a(i2, j1);
  };
  const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
  const double det = c00 * a(0, 0) + c10 * a(1, 0) + c20 * a(2, 0);
  const double invdet = 1.0 / det;
  r(0, 0) = c00 * invdet;
  r(0, 1) = c10 * invdet;
  r(0, 2) = c20 * invdet;
  r(1, 0) = cof(0, 1) * invdet;
  r(1, 1) = cof(1, 1) * invdet;
  r(1, 2) = cof(2, 1) * invdet;
  r(2, 0) = cof(0, 2) * invdet;
  r(2, 1) = cof(1, 2) * invdet;
  r(2, 2) = cof(2, 2) * invdet;
  return r;
}

inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
inline double sqn3(const double* a) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }

// SelfAdjointEigenSolver<Matrix3d>::computeDirect (Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h,
// direct_selfadjoint_eigenvalues<SolverType,3,false>). Uses the lower triangle. evecs column-major.
void eig3_direct(const M3& mat, double evals[3], M3& evecs) {
  const double shift = (mat(0, 0) + mat(1, 1) + mat(2, 2)) / 3.0;
  M3 s;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) s(r, c) = r >= c ? mat(r, c) : mat(c, r);
  s(0, 0) -= shift;
  s(1, 1) -= shift;
  s(2, 2) -= shift;
  double scale = 0;
  for (int i = 0; i < 9; i++) scale = std::max(scale, std::fabs(s.m[i]));
  if (scale > 0)
    for (int i = 0; i < 9; i++) s.m[i] /= scale;
  // computeRoots
  {
    const double s_inv3 = 1.0 / 3.0, s_sqrt3 = std::sqrt(3.0);
    const double c0 = s(0, 0) * s(1, 1) * s(2, 2) + 2.0 * s(1, 0) * s(2, 0) * s(2, 1) - s(0, 0) * s(2, 1) * s(2, 1) -
                      s(1, 1) * s(2, 0) * s(2, 0) - s(2, 2) * s(1, 0) * s(1, 0);
    const double c1 = s(0, 0) * s(1, 1) - s(1, 0) * s(1, 0) + s(0, 0) * s(2, 2) - s(2, 0) * s(2, 0) + s(1, 1) * s(2, 2) -
                      s(2, 1) * s(2, 1);
    const double c2 = s(0, 0) + s(1, 1) + s(2, 2);
    const double c2_over_3 = c2 * s_inv3;
    double a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
    a_over_3 = std::max(a_over_3, 0.0);
    const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
    double q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
    q = std::max(q, 0.0);
    const double rho = std::sqrt(a_over_3);
    const double theta = std::atan2(std::sqrt(q), half_b) * s_inv3;
    const double cos_theta = std::cos(theta), sin_theta = std::sin(theta);
    evals[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    evals[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    evals[2] = c2_over_3 + 2.0 * rho * cos_theta;
  }
  const double eps = std::numeric_limits<double>::epsilon();
  auto extract_kernel = [](M3& m, double* res, double* representative) {
    int i0 = 0;
    double best = std::fabs(m(0, 0));
    for (int i = 1; i < 3; i++)
      if (std::fabs(m(i, i)) > best) {
        best = std::fabs(m(i, i));
        i0 = i;
      }
    double rep[3] = {m(0, i0), m(1, i0), m(2, i0)};
    representative[0] = rep[0];
    representative[1] = rep[1];
    representative[2] = rep[2];
    double c0[3], c1[3];
    const int j1 = (i0 + 1) % 3, j2 = (i0 + 2) % 3;
    const double col1[3] = {m(0, j1), m(1, j1), m(2, j1)}, col2[3] = {m(0, j2), m(1, j2), m(2, j2)};
    cross3(rep, col1, c0);
    cross3(rep, col2, c1);
    const double n0 = sqn3(c0), n1 = sqn3(c1);
    if (n0 > n1) {
      const double d = std::sqrt(n0);
      for (int i = 0; i < 3; i++) res[i] = c0[i] / d;
    } else {
      const double d = std::sqrt(n1);
      for (int i = 0; i < 3; i++) res[i] = c1[i] / d;
    }
  };
  if ((evals[2] - evals[0]) <= eps) {
    for (int i = 0; i < 9; i++) evecs.m[i] = 0;
    evecs(0, 0) = evecs(1, 1) = evecs(2, 2) = 1;
  } else {
    M3 tmp = s;
    double d0 = evals[2] - evals[1];
    double d1 = evals[1] - evals[0];
    int k = 0, l = 2;
    if (d0 > d1) {
      std::swap(k, l);
      d0 = d1;
    }
    double colk[3], coll[3];
    tmp(0, 0) -= evals[k];
    tmp(1, 1) -= evals[k];
    tmp(2, 2) -= evals[k];
    extract_kernel(tmp, colk, coll);
    if (d0 <= 2 * eps * d1) {
      const double dot = colk[0] * coll[0] + colk[1] * coll[1] + colk[2] * coll[2];
      for (int i = 0; i < 3; i++) coll[i] -= dot * coll[i];
      const double nn = std::sqrt(sqn3(coll));
      for (int i = 0; i < 3; i++) coll[i] /= nn;
    } else {
      tmp = s;
      tmp(0, 0) -= evals[l];
      tmp(1, 1) -= evals[l];
      tmp(2, 2) -= evals[l];
      double dummy[3];
      extract_kernel(tmp, coll, dummy);
    }
    for (int i = 0; i < 3; i++) {
      evecs(i, k) = colk[i];
      evecs(i, l) = coll[i];
    }
    double c2v[3] = {evecs(0, 2), evecs(1, 2), evecs(2, 2)}, c0v[3] = {evecs(0, 0), evecs(1, 0), evecs(2, 0)}, c1v[3];
    cross3(c2v, c0v, c1v);
    const double nn = std::sqrt(sqn3(c1v));
    for (int i = 0; i < 3; i++) evecs(i, 1) = c1v[i] / nn;
  }
  for (int i = 0; i < 3; i++) evals[i] = evals[i] * scale + shift;
}

// Solve (A) x = rhs for a symmetric 6x6 A through a diagonally-pivoted LDL^T (Eigen::LDLT semantics).
void ldlt_solve6(const double A_in[36], const double rhs[6], double x[6]) {
  double A[36];
  std::memcpy(A, A_in, sizeof(A));
  int perm[6];
  for (int i = 0; i < 6; i++) perm[i] = i;
  for (int k = 0; k < 6; k++) {
    int p = k;
    double best = std::fabs(A[k + 6 * k]);
    for (int i = k + 1; i < 6; i++)
      if (std::fabs(A[i + 6 * i]) > best) {
        best = std::fabs(A[i + 6 * i]);
        p = i;
      }
    if (p != k) {
      for (int j = 0; j < 6; j++) std::swap(A[k + 6 * j], A[p + 6 * j]);
      for (int j = 0; j < 6; j++) std::swap(A[j + 6 * k], A[j + 6 * p]);
      std::swap(perm[k], perm[p]);
    }
    const double d = A[k + 6 * k];
    if (d == 0) continue;
    for (int i = k + 1; i < 6; i++) A[i + 6 * k] /= d;
    for (int j = k + 1; j < 6; j++)
      for (int i = j; i < 6; i++) {
        A[i + 6 * j] -= A[i + 6 * k] * d * A[j + 6 * k];
        A[j + 6 * i] = A[i + 6 * j];
      }
  }
  double y[6];
  for (int i = 0; i < 6; i++) y[i] = rhs[perm[i]];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < i; j++) y[i] -= A[i + 6 * j] * y[j];
  for (int i = 0; i < 6; i++) y[i] = A[i + 6 * i] != 0 ? y[i] / A[i + 6 * i] : 0;
  for (int i = 5; i >= 0; i--)
    for (int j = i + 1; j < 6; j++) y[i] -= A[j + 6 * i] * y[j];
  for (int i = 0; i < 6; i++) x[perm[i]] = y[i];
}

// ---- util/lie.hpp:54-103 ----
struct Iso {
  double R[9];  // column-major
  double t[3];
};
Iso iso_identity() {
  Iso T{};
  T.R[0] = T.R[4] = T.R[8] = 1;
  return T;
}
Iso iso_mul(const Iso& a, const Iso& b) {
  Iso r;
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < 3; i++) r.R[i + 3 * c] = a.R[i] * b.R[3 * c] + a.R[i + 3] * b.R[1 + 3 * c] + a.R[i + 6] * b.R[2 + 3 * c];
  for (int i = 0; i < 3; i++) r.t[i] = a.R[i] * b.t[0] + a.R[i + 3] * b.t[1] + a.R[i + 6] * b.t[2] + a.t[i];
  return r;
}
void quat_to_R(double w, double x, double y, double z, double* R) {  // Eigen QuaternionBase::toRotationMatrix
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz);
  R[3] = txy - twz;
  R[6] = txz + twy;
  R[1] = txy + twz;
  R[4] = 1 - (txx + tzz);
  R[7] = tyz - twx;
  R[2] = txz - twy;
  R[5] = tyz + twx;
  R[8] = 1 - (txx + tyy);
}
Iso se3_exp(const double a[6]) {
  const double* omega = a;
  const double theta_sq = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
  const double theta = std::sqrt(theta_sq);
  double imag_factor, real_factor;  // so3_exp, lie.hpp:54-74
  if (theta_sq < 1e-10) {
    const double theta_quad = theta_sq * theta_sq;
    imag_factor = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real_factor = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    const double half_theta = 0.5 * theta;
    imag_factor = std::sin(half_theta) / theta;
    real_factor = std::cos(half_theta);
  }
  Iso se3 = iso_identity();
  quat_to_R(real_factor, imag_factor * omega[0], imag_factor * omega[1], imag_factor * omega[2], se3.R);
  const double* v = a + 3;
  if (theta < 1e-10) {
    for (int i = 0; i < 3; i++) se3.t[i] = se3.R[i] * v[0] + se3.R[i + 3] * v[1] + se3.R[i + 6] * v[2];
  } else {
    // V = I + (1-cos)/theta^2 * Omega + (theta - sin)/(theta^3) * Omega^2
    double O[9] = {0, omega[2], -omega[1], -omega[2], 0, omega[0], omega[1], -omega[0], 0};  // column-major skew
    double O2[9];
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) O2[r + 3 * c] = O[r] * O[3 * c] + O[r + 3] * O[1 + 3 * c] + O[r + 6] * O[2 + 3 * c];
    const double k1 = (1.0 - std::cos(theta)) / theta_sq, k2 = (theta - std::sin(theta)) / (theta_sq * theta);
    double V[9];
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + k1 * O[i] + k2 * O2[i];
    for (int i = 0; i < 3; i++) se3.t[i] = V[i] * v[0] + V[i + 3] * v[1] + V[i + 6] * v[2];
  }
  return se3;
}

// ---- util/sort_omp.hpp:58-85 quick_sort_omp_impl (tasks run here sequentially: same permutation) ----
using KeyIdx = std::pair<std::uint64_t, size_t>;
struct KeyLess {
  bool operator()(const KeyIdx& a, const KeyIdx& b) const { return a.first < b.first; }
};
void quick_sort_impl(KeyIdx* first, KeyIdx* last) {
  const KeyLess comp;
  const std::ptrdiff_t n = last - first;
  if (n < 1024) {
    std::sort(first, last, comp);
    return;
  }
  const auto median3 = [&](const KeyIdx& a, const KeyIdx& b, const KeyIdx& c) {
    return comp(a, b) ? (comp(b, c) ? b : (comp(a, c) ? c : a)) : (comp(a, c) ? a : (comp(b, c) ? c : b));
  };
  const int offset = (int)(n / 8);
  const KeyIdx m1 = median3(*first, *(first + offset), *(first + offset * 2));
  const KeyIdx m2 = median3(*(first + offset * 3), *(first + offset * 4), *(first + offset * 5));
  const KeyIdx m3 = median3(*(first + offset * 6), *(first + offset * 7), *(last - 1));
  const KeyIdx pivot = median3(m1, m2, m3);
  KeyIdx* middle1 = std::partition(first, last, [&](const KeyIdx& val) { return comp(val, pivot); });
  KeyIdx* middle2 = std::partition(middle1, last, [&](const KeyIdx& val) { return !comp(pivot, val); });
  quick_sort_impl(first, middle1);
  quick_sort_impl(middle2, last);
}

// ---- util/downsampling_omp.hpp:26-95 voxelgrid_sampling_omp (+ util/fast_floor.hpp:12-15) ----
// stable_order != 0: deviate from the reference by ordering equal keys by point index (used by stage-level
// parity tests of the GPU path, whose radix sort is stable); the SET of voxels and block splits is unchanged.
void voxelgrid_sampling(const std::vector<V4>& pts, double leaf_size, int stable_order, std::vector<V4>& out) {
  out.clear();
  const size_t N = pts.size();
  if (N == 0) return;
  const double inv_leaf_size = 1.0 / leaf_size;
  constexpr std::uint64_t invalid_coord = std::numeric_limits<std::uint64_t>::max();
  constexpr int coord_bit_size = 21;
  constexpr size_t coord_bit_mask = (1 << 21) - 1;
  constexpr int coord_offset = 1 << (coord_bit_size - 1);
  std::vector<KeyIdx> coord_pt(N);
  for (size_t i = 0; i < N; i++) {
    int coord[4];
    bool bad = false;
    for (int k = 0; k < 4; k++) {
      const double v = pts[i][k] * inv_leaf_size;
      const int nc = (int)v;                           // fast_floor: cast<int>() ...
      coord[k] = nc - (v < (double)nc ? 1 : 0) + coord_offset;  // ... minus (pt < ncoord)
      if (coord[k] < 0 || coord[k] > (int)coord_bit_mask) bad = true;
    }
    if (bad) {
      coord_pt[i] = {invalid_coord, i};
      continue;
    }
    const std::uint64_t bits = (static_cast<std::uint64_t>(coord[0] & coord_bit_mask) << (coord_bit_size * 0)) |
                               (static_cast<std::uint64_t>(coord[1] & coord_bit_mask) << (coord_bit_size * 1)) |
                               (static_cast<std::uint64_t>(coord[2] & coord_bit_mask) << (coord_bit_size * 2));
    coord_pt[i] = {bits, i};
  }
  if (stable_order)
    std::stable_sort(coord_pt.begin(), coord_pt.end(), KeyLess());
  else
    quick_sort_impl(coord_pt.data(), coord_pt.data() + N);
  const int block_size = 1024;
  for (size_t block_begin = 0; block_begin < N; block_begin += block_size) {  // blocks in order (deterministic)
    const size_t block_end = std::min<size_t>(N, block_begin + block_size);
    V4 sum_pt = pts[coord_pt[block_begin].second];
    for (size_t i = block_begin + 1; i != block_end; i++) {
      if (coord_pt[i].first == invalid_coord) continue;
      if (coord_pt[i - 1].first != coord_pt[i].first) {
        out.push_back({sum_pt[0] / sum_pt[3], sum_pt[1] / sum_pt[3], sum_pt[2] / sum_pt[3], sum_pt[3] / sum_pt[3]});
        sum_pt = {0, 0, 0, 0};
      }
      const V4& p = pts[coord_pt[i].second];
      for (int k = 0; k < 4; k++) sum_pt[k] += p[k];
    }
    out.push_back({sum_pt[0] / sum_pt[3], sum_pt[1] / sum_pt[3], sum_pt[2] / sum_pt[3], sum_pt[3] / sum_pt[3]});
  }
}

// ---- ann/kdtree.hpp:54-233, ann/projection.hpp:17-50, ann/knn_result.hpp:29-117 ----
struct KdNode {
  uint32_t first = 0, last = 0;
  int axis = 0;
  double thresh = 0;
  uint32_t left = 0xffffffffu, right = 0xffffffffu;
};
struct KdTree {
  const std::vector<V4>* points = nullptr;
  std::vector<size_t> indices;
  std::vector<KdNode> nodes;
  uint32_t root = 0;
  static constexpr int max_leaf_size = 20, max_scan_count = 128;

  int find_axis(size_t* first, size_t* last) const {
    const size_t N = last - first;
    double sum_pt[4] = {0, 0, 0, 0}, sum_sq[4] = {0, 0, 0, 0};
    const size_t step = N < (size_t)max_scan_count ? 1 : N / max_scan_count;
    const size_t num_steps = N / step;
    for (size_t i = 0; i < num_steps; i++) {
      const V4& pt = (*points)[*(first + step * i)];
      for (int k = 0; k < 4; k++) {
        sum_pt[k] += pt[k];
        sum_sq[k] += pt[k] * pt[k];
      }
    }
    double var[4];
    for (int k = 0; k < 4; k++) {
      const double mean = sum_pt[k] / sum_pt[3];
      var[k] = sum_sq[k] - mean * sum_pt[k];
    }
    return var[0] > var[1] ? (var[0] > var[2] ? 0 : 2) : (var[1] > var[2] ? 1 : 2);
  }
  uint32_t create_node(size_t& node_count, size_t* global_first, size_t* first, size_t* last) {
    const size_t N = last - first;
    const uint32_t node_index = (uint32_t)node_count++;
    if (N <= (size_t)max_leaf_size) {
      nodes[node_index].first = (uint32_t)(first - global_first);
      nodes[node_index].last = (uint32_t)(last - global_first);
      return node_index;
    }
    const int axis = find_axis(first, last);
    size_t* median_itr = first + N / 2;
    const std::vector<V4>& P = *points;
    std::nth_element(first, median_itr, last, [&](size_t i, size_t j) { return P[i][axis] < P[j][axis]; });
    nodes[node_index].axis = axis;
    nodes[node_index].thresh = P[*median_itr][axis];
    const uint32_t l = create_node(node_count, global_first, first, median_itr);
    const uint32_t r = create_node(node_count, global_first, median_itr, last);
    nodes[node_index].left = l;
    nodes[node_index].right = r;
    return node_index;
  }
  void build(const std::vector<V4>& pts) {
    points = &pts;
    indices.resize(pts.size());
    std::iota(indices.begin(), indices.end(), 0);
    nodes.assign(std::max<size_t>(pts.size(), 1), KdNode());
    if (pts.empty()) return;
    size_t node_count = 0;
    root = create_node(node_count, indices.data(), indices.data(), indices.data() + indices.size());
    nodes.resize(node_count);
  }
};

struct KnnResult {  // knn_result.hpp (dynamic capacity; N == 1 specialisation has the same observable behaviour)
  int capacity, num_found = 0;
  size_t* indices;
  double* distances;
  KnnResult(size_t* idx, double* d, int k) : capacity(k), indices(idx), distances(d) {
    std::fill(indices, indices + k, std::numeric_limits<size_t>::max());
    std::fill(distances, distances + k, std::numeric_limits<double>::max());
  }
  double worst_distance() const { return distances[capacity - 1]; }
  void push(size_t index, double distance) {
    if (distance >= worst_distance()) return;
    if (capacity == 1) {
      indices[0] = index;
      distances[0] = distance;
    } else {
      int insert_loc = std::min<int>(num_found, capacity - 1);
      for (; insert_loc > 0 && distance < distances[insert_loc - 1]; insert_loc--) {
        indices[insert_loc] = indices[insert_loc - 1];
        distances[insert_loc] = distances[insert_loc - 1];
      }
      indices[insert_loc] = index;
      distances[insert_loc] = distance;
    }
    num_found = std::min<int>(num_found + 1, capacity);
  }
};

bool knn_search(const KdTree& t, const V4& query, uint32_t node_index, KnnResult& result) {  // kdtree.hpp:194-233
  const KdNode& node = t.nodes[node_index];
  if (node.left == 0xffffffffu) {
    for (size_t i = node.first; i < node.last; i++) {
      const V4& p = (*t.points)[t.indices[i]];
      double sq = 0;
      for (int k = 0; k < 4; k++) sq += (p[k] - query[k]) * (p[k] - query[k]);
      result.push(t.indices[i], sq);
    }
    return !(result.worst_distance() < 0.0);  // KnnSetting::fulfilled with epsilon = 0
  }
  const double val = query[node.axis];
  const double diff = val - node.thresh;
  const double cut_sq_dist = diff * diff;
  uint32_t best_child, other_child;
  if (diff < 0.0) {
    best_child = node.left;
    other_child = node.right;
  } else {
    best_child = node.right;
    other_child = node.left;
  }
  if (!knn_search(t, query, best_child, result)) return false;
  if (result.worst_distance() > cut_sq_dist) return knn_search(t, query, other_child, result);
  return true;
}
size_t knn(const KdTree& t, const V4& q, int k, size_t* idx, double* sqd) {
  KnnResult r(idx, sqd, k);
  if (t.points->empty()) return 0;
  knn_search(t, q, t.root, r);
  return (size_t)r.num_found;
}

// ---- util/normal_estimation.hpp:13-92 (NormalCovarianceSetter) ----
void estimate_normals_covariances(Cloud& cloud, const KdTree& tree, int num_neighbors, int num_threads) {
  const size_t N = cloud.points.size();
  cloud.normals.assign(N, V4{0, 0, 0, 0});
  cloud.covs.assign(N, M3{});
#pragma omp parallel for num_threads(num_threads) if (num_threads > 1)  // util/normal_estimation_omp.hpp:21-26
  for (size_t pi = 0; pi < N; pi++) {
    std::vector<size_t> k_indices(num_neighbors);
    std::vector<double> k_sq_dists(num_neighbors);
    const size_t n = knn(tree, cloud.points[pi], num_neighbors, k_indices.data(), k_sq_dists.data());
    if (n < 5) {  // set_invalid: normal = 0, cov = diag(1,1,1,0)
      M3 c{};
      c(0, 0) = c(1, 1) = c(2, 2) = 1.0;
      cloud.covs[pi] = c;
      continue;
    }
    double sum_points[4] = {0, 0, 0, 0}, sum_cross[16] = {0};
    for (size_t i = 0; i < n; i++) {
      const V4& pt = cloud.points[k_indices[i]];
      for (int r = 0; r < 4; r++) {
        sum_points[r] += pt[r];
        for (int c = 0; c < 4; c++) sum_cross[r + 4 * c] += pt[r] * pt[c];
      }
    }
    double mean[4];
    for (int r = 0; r < 4; r++) mean[r] = sum_points[r] / n;
    M3 cov;
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) cov(r, c) = (sum_cross[r + 4 * c] - mean[r] * sum_points[c]) / n;
    double evals[3];
    M3 V;
    eig3_direct(cov, evals, V);
    // NormalSetter::set
    double nrm[3] = {V(0, 0), V(1, 0), V(2, 0)};
    const double nn = std::sqrt(sqn3(nrm));
    for (int i = 0; i < 3; i++) nrm[i] /= nn;
    const V4& p = cloud.points[pi];
    const double dot = p[0] * nrm[0] + p[1] * nrm[1] + p[2] * nrm[2];
    const double sgn = dot > 0 ? -1.0 : 1.0;
    cloud.normals[pi] = {sgn * nrm[0], sgn * nrm[1], sgn * nrm[2], 0.0};
    // CovarianceSetter::set: V * diag(1e-3, 1, 1) * V^T
    const double values[3] = {1e-3, 1.0, 1.0};
    M3 VD;
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) VD(r, c) = V(r, c) * values[c];
    M3 C;
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) C(r, c) = VD(r, 0) * V(c, 0) + VD(r, 1) * V(c, 1) + VD(r, 2) * V(c, 2);
    cloud.covs[pi] = C;
  }
}

// preprocess_points, registration_helper.cpp:22-34
void preprocess(const float* xyzw, int n, const gfso_gicp_cfg& cfg, int stable_order, Cloud& out, KdTree& tree, int num_threads = 1) {
  std::vector<V4> pts(n);
  for (int i = 0; i < n; i++) pts[i] = {(double)xyzw[4 * i], (double)xyzw[4 * i + 1], (double)xyzw[4 * i + 2], 1.0};  // point_cloud.hpp:26-31
  voxelgrid_sampling(pts, cfg.downsampling_resolution, stable_order, out.points);
  tree.build(out.points);
  estimate_normals_covariances(out, tree, cfg.num_neighbors, num_threads);
}

// ---- factors/gicp_factor.hpp:35-89 ----
struct Factor {
  size_t target_index = std::numeric_limits<size_t>::max();
  M3 mahalanobis{};
};

inline void transform_pt(const Iso& T, const V4& p, double* o) {
  for (int i = 0; i < 3; i++) o[i] = T.R[i] * p[0] + T.R[i + 3] * p[1] + T.R[i + 6] * p[2] + T.t[i] * p[3];
}

bool linearize(const Cloud& target, const Cloud& source, const KdTree& tree, const Iso& T, size_t si, double max_dist_sq,
               Factor& f, double H[36], double b[6], double* e) {
  f.target_index = std::numeric_limits<size_t>::max();
  double tp[3];
  transform_pt(T, source.points[si], tp);
  const V4 q = {tp[0], tp[1], tp[2], 1.0};
  size_t k_index;
  double k_sq_dist;
  if (!knn(tree, q, 1, &k_index, &k_sq_dist) || k_sq_dist > max_dist_sq) return false;  // rejector.hpp:24
  f.target_index = k_index;
  // RCR = C_t + T * C_s * T^T (3x3 block)
  const M3& Cs = source.covs[si];
  const M3& Ct = target.covs[k_index];
  M3 RC, RCR;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) RC(r, c) = T.R[r] * Cs(0, c) + T.R[r + 3] * Cs(1, c) + T.R[r + 6] * Cs(2, c);
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) RCR(r, c) = Ct(r, c) + (RC(r, 0) * T.R[c] + RC(r, 1) * T.R[c + 3] + RC(r, 2) * T.R[c + 6]);
  f.mahalanobis = inverse3(RCR);
  const M3& M = f.mahalanobis;
  const V4& tq = target.points[k_index];
  const double res[3] = {tq[0] - tp[0], tq[1] - tp[1], tq[2] - tp[2]};
  // J = [ R * skew(p) | -R ]   (3x6, the 4th row of the reference's 4x6 J is zero)
  const V4& p = source.points[si];
  const double S[9] = {0, p[2], -p[1], -p[2], 0, p[0], p[1], -p[0], 0};  // column-major skew(p)
  double J[18];
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) {
      J[r + 3 * c] = T.R[r] * S[3 * c] + T.R[r + 3] * S[1 + 3 * c] + T.R[r + 6] * S[2 + 3 * c];
      J[r + 3 * (c + 3)] = -T.R[r + 3 * c];
    }
  double MJ[18];
  for (int c = 0; c < 6; c++)
    for (int r = 0; r < 3; r++) MJ[r + 3 * c] = M(r, 0) * J[3 * c] + M(r, 1) * J[1 + 3 * c] + M(r, 2) * J[2 + 3 * c];
  for (int c = 0; c < 6; c++)
    for (int r = 0; r < 6; r++) H[r + 6 * c] = J[3 * r] * MJ[3 * c] + J[1 + 3 * r] * MJ[1 + 3 * c] + J[2 + 3 * r] * MJ[2 + 3 * c];
  double Mr[3];
  for (int r = 0; r < 3; r++) Mr[r] = M(r, 0) * res[0] + M(r, 1) * res[1] + M(r, 2) * res[2];
  for (int r = 0; r < 6; r++) b[r] = J[3 * r] * Mr[0] + J[1 + 3 * r] * Mr[1] + J[2 + 3 * r] * Mr[2];
  *e = 0.5 * (res[0] * Mr[0] + res[1] * Mr[1] + res[2] * Mr[2]);
  return true;
}

double factor_error(const Cloud& target, const Cloud& source, const Iso& T, size_t si, const Factor& f) {
  if (f.target_index == std::numeric_limits<size_t>::max()) return 0.0;
  double tp[3];
  transform_pt(T, source.points[si], tp);
  const V4& tq = target.points[f.target_index];
  const double res[3] = {tq[0] - tp[0], tq[1] - tp[1], tq[2] - tp[2]};
  const M3& M = f.mahalanobis;
  double Mr[3];
  for (int r = 0; r < 3; r++) Mr[r] = M(r, 0) * res[0] + M(r, 1) * res[1] + M(r, 2) * res[2];
  return 0.5 * (res[0] * Mr[0] + res[1] * Mr[1] + res[2] * Mr[2]);
}

Iso iso_from_colmajor(const double* T16) {
  Iso T;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) T.R[r + 3 * c] = T16[r + 4 * c];
  for (int r = 0; r < 3; r++) T.t[r] = T16[r + 12];
  return T;
}
void iso_to_colmajor(const Iso& T, double* T16) {
  std::memset(T16, 0, 16 * sizeof(double));
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) T16[r + 4 * c] = T.R[r + 3 * c];
  for (int r = 0; r < 3; r++) T16[r + 12] = T.t[r];
  T16[15] = 1.0;
}

}  // namespace

extern "C" {

void gfso_gicp_default_cfg(gfso_gicp_cfg* c) {
  c->num_threads = 4;                    // src/RegistrationGICP.cc:10
  c->downsampling_resolution = 0.02;     // :11
  c->max_correspondence_distance = 0.1;  // :12-13
  c->rotation_eps = 0.1 * M_PI / 180.0;  // registration_helper.hpp RegistrationSetting
  c->translation_eps = 1e-3;
  c->max_iterations = 20;
  c->num_neighbors = 10;  // registration_helper.cpp:60-61
}

static int g_stable_order = 0;
static int g_omp_threads = 1;  // 1 = deterministic; the reference hard-codes 4 (src/RegistrationGICP.cc:10) — used for timing
void gfso_gicp_set_stable_voxel_order(int on) { g_stable_order = on; }
// Test-input generator: M. D. McIlroy, "A Killer Adversary for Quicksort" (1999), run against libstdc++'s std::sort: the
// returned key sequence (a permutation of 0 .. n-1) drives that std::sort into its depth limit, i.e. into the heap-sort
// fallback of __introsort_loop, which the device replica (voxel_qsort.hpp) must reproduce as well.
void gfso_antiqsort_keys(int n, int32_t* out) {
  std::vector<int> val((size_t)std::max(n, 0)), ptr(val.size());
  const int gas = n - 1;
  int nsolid = 0, candidate = 0;
  for (int i = 0; i < n; i++) {
    ptr[i] = i;
    val[i] = gas;
  }
  std::sort(ptr.begin(), ptr.end(), [&](int x, int y) {
    if (val[x] == gas && val[y] == gas) {
      if (x == candidate)
        val[x] = nsolid++;
      else
        val[y] = nsolid++;
    }
    if (val[x] == gas)
      candidate = x;
    else if (val[y] == gas)
      candidate = y;
    return val[x] < val[y];
  });
  for (int i = 0; i < n; i++) out[i] = val[i];
}
void gfso_quick_sort_pairs(uint64_t* keys_io, uint64_t* idx_io, int n) {
  std::vector<KeyIdx> v((size_t)std::max(n, 0));
  for (int i = 0; i < n; i++) v[i] = {keys_io[i], (size_t)idx_io[i]};
  quick_sort_impl(v.data(), v.data() + n);
  for (int i = 0; i < n; i++) {
    keys_io[i] = v[i].first;
    idx_io[i] = v[i].second;
  }
}
// a stream of pushes into the restated KnnResult (pinned against the reference's own container by tests/test_oracle_ref.py)
int gfso_knn_push_stream(int k, const uint64_t* index, const double* distance, int n, uint64_t* idx_out, double* dist_out) {
  std::vector<size_t> idx((size_t)k);
  std::vector<double> d((size_t)k);
  KnnResult r(idx.data(), d.data(), k);
  for (int i = 0; i < n; i++) r.push((size_t)index[i], distance[i]);
  for (int i = 0; i < k; i++) {
    idx_out[i] = idx[i];
    dist_out[i] = d[i];
  }
  return r.num_found;
}
void gfso_gicp_set_threads(int n) { g_omp_threads = n < 1 ? 1 : n; }

void gfso_gicp_align(const float* target_xyzw, int nt, const float* source_xyzw, int ns, const double init_T[16],
                     const gfso_gicp_cfg* cfg, gfso_gicp_result* out) {
  // small_gicp::align<float,4>, registration_helper.cpp:57-69
  Cloud target, source;
  KdTree target_tree, source_tree;
  const int NT = g_omp_threads;
  preprocess(target_xyzw, nt, *cfg, g_stable_order, target, target_tree, NT);
  preprocess(source_xyzw, ns, *cfg, g_stable_order, source, source_tree, NT);
  // Registration<GICPFactor, ParallelReductionOMP>::align, registration.hpp:33-43 +
  // LevenbergMarquardtOptimizer::optimize, optimizer.hpp:83-147
  const double max_dist_sq = cfg->max_correspondence_distance * cfg->max_correspondence_distance;
  std::vector<Factor> factors(source.points.size());
  const int max_inner_iterations = 10;
  const double init_lambda = 1e-3, lambda_factor = 10.0;
  double lambda = init_lambda;
  Iso T = iso_from_colmajor(init_T);
  bool converged = false;
  size_t iterations = 0;
  double H[36] = {0}, b[6] = {0}, e = 0;
  int n_lin = 0, n_err = 0;
  double Hres[36] = {0}, bres[6] = {0}, eres = 0;
  for (int i = 0; i < cfg->max_iterations && !converged; i++) {
    // ParallelReductionOMP::linearize, reduction_omp.hpp:21-55 (summed here in source order)
    std::memset(H, 0, sizeof(H));
    std::memset(b, 0, sizeof(b));
    e = 0;
    if (NT == 1) {
      for (size_t s = 0; s < factors.size(); s++) {
        double Hi[36], bi[6], ei;
        if (!linearize(target, source, target_tree, T, s, max_dist_sq, factors[s], Hi, bi, &ei)) continue;
        for (int k = 0; k < 36; k++) H[k] += Hi[k];
        for (int k = 0; k < 6; k++) b[k] += bi[k];
        e += ei;
      }
    } else {  // per-thread sums folded thread 0..NT-1, reduction_omp.hpp:27-55
      std::vector<std::array<double, 43>> part(NT);
      for (auto& a : part) a.fill(0.0);
#pragma omp parallel num_threads(NT)
      {
        const int tid = omp_get_thread_num();
#pragma omp for schedule(guided, 8)
        for (std::int64_t s = 0; s < (std::int64_t)factors.size(); s++) {
          double Hi[36], bi[6], ei;
          if (!linearize(target, source, target_tree, T, (size_t)s, max_dist_sq, factors[s], Hi, bi, &ei)) continue;
          for (int k = 0; k < 36; k++) part[tid][k] += Hi[k];
          for (int k = 0; k < 6; k++) part[tid][36 + k] += bi[k];
          part[tid][42] += ei;
        }
      }
      for (int t2 = 0; t2 < NT; t2++) {
        for (int k = 0; k < 36; k++) H[k] += part[t2][k];
        for (int k = 0; k < 6; k++) b[k] += part[t2][36 + k];
        e += part[t2][42];
      }
    }
    n_lin++;
    bool success = false;
    for (int j = 0; j < max_inner_iterations; j++) {
      double A[36], rhs[6], delta[6];
      for (int k = 0; k < 36; k++) A[k] = H[k] + (k % 7 == 0 ? lambda : 0.0);
      for (int k = 0; k < 6; k++) rhs[k] = -b[k];
      ldlt_solve6(A, rhs, delta);
      const Iso new_T = iso_mul(T, se3_exp(delta));
      double new_e = 0;
#pragma omp parallel for num_threads(NT) schedule(guided, 8) reduction(+ : new_e) if (NT > 1)  // reduction_omp.hpp:58-66
      for (std::int64_t s = 0; s < (std::int64_t)factors.size(); s++) new_e += factor_error(target, source, new_T, (size_t)s, factors[s]);
      n_err++;
      if (new_e <= e) {
        const double dr = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
        const double dt = std::sqrt(delta[3] * delta[3] + delta[4] * delta[4] + delta[5] * delta[5]);
        converged = dr <= cfg->rotation_eps && dt <= cfg->translation_eps;  // termination_criteria.hpp:17
        T = new_T;
        lambda /= lambda_factor;
        success = true;
        break;
      } else {
        lambda *= lambda_factor;
      }
    }
    iterations = (size_t)i;
    std::memcpy(Hres, H, sizeof(H));
    std::memcpy(bres, b, sizeof(b));
    eres = e;
    if (!success) break;
  }
  size_t inliers = 0;
  for (const Factor& f : factors) inliers += f.target_index != std::numeric_limits<size_t>::max();
  iso_to_colmajor(T, out->T);
  out->converged = converged;
  out->iterations = iterations;
  out->num_inliers = inliers;
  std::memcpy(out->H, Hres, sizeof(Hres));
  std::memcpy(out->b, bres, sizeof(bres));
  out->error = eres;
  out->n_target_ds = (int)target.points.size();
  out->n_source_ds = (int)source.points.size();
  out->n_linearize = n_lin;
  out->n_error_evals = n_err;
}

int gfso_gicp_preprocess(const float* xyzw, int n, const gfso_gicp_cfg* cfg, double* pts, double* covs, double* normals) {
  Cloud c;
  KdTree t;
  preprocess(xyzw, n, *cfg, g_stable_order, c, t);
  for (size_t i = 0; i < c.points.size(); i++) {
    if (pts) std::memcpy(pts + 4 * i, c.points[i].data(), 32);
    if (normals) std::memcpy(normals + 4 * i, c.normals[i].data(), 32);
    if (covs) {
      std::memset(covs + 16 * i, 0, 128);
      for (int cc = 0; cc < 3; cc++)
        for (int r = 0; r < 3; r++) covs[16 * i + r + 4 * cc] = c.covs[i](r, cc);
    }
  }
  return (int)c.points.size();
}

void gfso_knn(const double* pts, int n, const double* queries, int nq, int k, int64_t* idx, double* sqd) {
  std::vector<V4> P(n);
  for (int i = 0; i < n; i++) P[i] = {pts[4 * i], pts[4 * i + 1], pts[4 * i + 2], pts[4 * i + 3]};
  KdTree t;
  t.build(P);
  std::vector<size_t> ki(k);
  for (int q = 0; q < nq; q++) {
    const V4 qq = {queries[4 * q], queries[4 * q + 1], queries[4 * q + 2], queries[4 * q + 3]};
    const size_t found = knn(t, qq, k, ki.data(), sqd + (size_t)q * k);
    for (int j = 0; j < k; j++) idx[(size_t)q * k + j] = j < (int)found ? (int64_t)ki[j] : -1;
  }
}

void gfso_eig3_direct(const double* m, double* evals, double* evecs) {
  M3 a, v;
  std::memcpy(a.m, m, sizeof(a.m));
  eig3_direct(a, evals, v);
  std::memcpy(evecs, v.m, sizeof(v.m));
}

void gfso_se3_exp(const double* twist6, double* T16) { iso_to_colmajor(se3_exp(twist6), T16); }

}  // extern "C"
