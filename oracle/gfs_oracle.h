/* ============================================================================================
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of the GeoFlow-SLAM per-frame hot path, used ONLY by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / timed CPU baseline.
 * Nothing under geoflowslam_amd/ may include, link or call anything declared here.
 *
 * PARITY UNPINNED, with one exception: the reference (HorizonRobotics/GeoFlowSlam) ships no tests, golden vectors or
 * fixtures for this path and cannot be compiled here (OpenCV / Eigen / PCL are absent) -- except
 * small_gicp's util/sort_omp.hpp (quick_sort_omp, the voxel sort) and ann/knn_result.hpp (KnnResult), which build from their own
 * sources: oracle/ref_build.sh compiles them from /root/reference into oracle/_ref/, and tests/test_oracle_ref.py pins the
 * restatement of those two pieces against them.  Everything else of this restatement is pinned only by (a) the reference sources it cites line by line,
 * (b) hand-derived known-answer tests (tests/test_oracle_*.py) and (c) the documented semantics of
 * the third-party primitives it restates:
 *     OpenCV 4.5.4 (Ubuntu 22.04 libopencv-dev; reference says ">=3.0", CMakeLists.txt:65)
 *         cv::FAST (TYPE_9_16), cv::resize(INTER_AREA), cv::GaussianBlur fixed-point 8U path,
 *         cv::fastAtan2, cvRound/cvFloor/cvCeil, cv::BFMatcher(NORM_HAMMING)
 *     Eigen 3.4.0 (reference says ">=3.1.0", CMakeLists.txt:74)
 *         SelfAdjointEigenSolver<Matrix3d>::computeDirect, Matrix3d::inverse, LDLT 6x6
 *     glibc 2.35 cosf/sinf, libstdc++ std::sort / std::nth_element (used directly)
 * ============================================================================================ */
#ifndef GFS_ORACLE_H_
#define GFS_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} gfso_keypoint;

/* ---------------- ORB extractor (src/ORBextractor.cc) ---------------- */
typedef struct gfso_orb gfso_orb;

/* blur_variant: 0 = OpenCV >= 4.5.1 taps {18,34,48,56,48,34,18}; 1 = 4.0..4.5.0 taps {18,34,49,55,49,34,18} */
gfso_orb* gfso_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast,
                          int blur_variant);
void gfso_orb_destroy(gfso_orb*);
/* OpenMP threads for timing runs (over levels / keypoints like the reference's ENABLE_OMP build); default 1 */
void gfso_orb_set_threads(gfso_orb*, int n);
/* ctor tables (src/ORBextractor.cc:421-479) */
void gfso_orb_get_tables(const gfso_orb*, float* scale /*nlevels*/, float* inv_scale, float* sigma2,
                         float* inv_sigma2, int32_t* feats_per_level, int32_t* umax /*16*/);
/* ORBextractor::operator() (src/ORBextractor.cc:1145-1225). Returns monoIndex, or -1 on empty image.
 * kps/desc may be NULL to only fill the intermediates. cap = capacity of kps (desc = cap*32). */
int gfso_orb_extract(gfso_orb*, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1,
                     gfso_keypoint* kps, uint8_t* desc, int cap, int* n_out);
/* intermediates of the last gfso_orb_extract call */
void gfso_orb_level_size(const gfso_orb*, int level, int* rows, int* cols);
void gfso_orb_get_level(const gfso_orb*, int level, uint8_t* dst /*rows*cols, unpadded*/);
void gfso_orb_get_blurred(const gfso_orb*, int level, uint8_t* dst /*rows*cols*/);
int gfso_orb_num_candidates(const gfso_orb*, int level);
/* candidates in the order handed to DistributeOctTree; coords relative to (16,16) like the reference */
void gfso_orb_get_candidates(const gfso_orb*, int level, int32_t* x, int32_t* y, int32_t* score);
int gfso_orb_num_level_keypoints(const gfso_orb*, int level);
/* per-level keypoints after octree + orientation, level coordinates (before pt *= scale) */
void gfso_orb_get_level_keypoints(const gfso_orb*, int level, gfso_keypoint* kps);

/* stand-alone primitives for known-answer tests */
void gfso_resize_area_u8(const uint8_t* src, int srows, int scols, int sstride, uint8_t* dst, int drows, int dcols,
                         int dstride);
int gfso_fast9_16(const uint8_t* img, int rows, int cols, int stride, int threshold, int nonmax, int32_t* x,
                  int32_t* y, int32_t* score, int cap);
float gfso_fast_atan2(float y, float x);
void gfso_gaussian_blur7(const uint8_t* src, int rows, int cols, int stride, uint8_t* dst, int dstride,
                         int blur_variant);
/* DistributeOctTree (src/ORBextractor.cc:567-768) on integer-valued candidates. Returns #kept;
 * out_idx[i] = index into the input of the i-th result, in std::list order. */
int gfso_distribute_octree(const float* x, const float* y, const float* response, int n, int min_x, int max_x,
                           int min_y, int max_y, int n_features, int32_t* out_idx, int cap);

/* ---------------- ORBmatcher (src/ORBmatcher.cc) ---------------- */
/* ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:2536-2550 */
int gfso_descriptor_distance(const uint8_t* a, const uint8_t* b);
/* cv::BFMatcher(NORM_HAMMING).match at src/ORBmatcher.cc:755-756 (no cross-check): for each query row
 * the lowest-index train row of minimum distance. nt == 0 -> returns 0 matches. Returns #matches (= nq). */
int gfso_bf_match_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* train_idx, int32_t* dist,
                          int nthreads);

/* ---------------- RegistrationGICP (src/RegistrationGICP.cc + Thirdparty/small_gicp) ---------------- */
typedef struct {
  int32_t num_threads;                /* 4   src/RegistrationGICP.cc:10 */
  double downsampling_resolution;     /* 0.02  :11 */
  double max_correspondence_distance; /* 0.1   :12-13 */
  double rotation_eps;                /* 0.1*pi/180  registration_helper.hpp */
  double translation_eps;             /* 1e-3 */
  int32_t max_iterations;             /* 20 */
  int32_t num_neighbors;              /* 10  registration_helper.cpp:60 */
} gfso_gicp_cfg;

typedef struct {
  double T[16]; /* column-major 4x4 T_target_source */
  int32_t converged;
  uint64_t iterations, num_inliers;
  double H[36]; /* column-major (symmetric) */
  double b[6];
  double error;
  /* diagnostics (not in small_gicp::RegistrationResult) */
  int32_t n_target_ds, n_source_ds, n_linearize, n_error_evals;
} gfso_gicp_result;

void gfso_gicp_default_cfg(gfso_gicp_cfg*);
/* on != 0: equal voxel keys are ordered by point index instead of by the reference's quick_sort_omp permutation
 * (util/sort_omp.hpp:58-85). Only which points fall on either side of a 1024-block split changes. Default 0. */
void gfso_gicp_set_stable_voxel_order(int on);
/* quick_sort_omp (util/sort_omp.hpp:58-85) of (key, index) pairs by key, through libstdc++'s std::partition / std::sort:
 * idx_io holds the payload (0..n-1 on entry), both arrays are permuted in place */
void gfso_quick_sort_pairs(uint64_t* keys_io, uint64_t* idx_io, int n);
int gfso_knn_push_stream(int k, const uint64_t* index, const double* distance, int n, uint64_t* idx_out, double* dist_out);
/* adversarial input for libstdc++'s std::sort (McIlroy's antiqsort): a permutation of 0..n-1 that reaches the heap-sort fallback */
void gfso_antiqsort_keys(int n, int32_t* out);
/* OpenMP threads for timing runs (the reference hard-codes 4, src/RegistrationGICP.cc:10); default 1 = deterministic */
void gfso_gicp_set_threads(int n);
/* RegistrationGICP::RegisterPointClouds, src/RegistrationGICP.cc:5-20 */
void gfso_gicp_align(const float* target_xyzw, int nt, const float* source_xyzw, int ns, const double init_T[16],
                     const gfso_gicp_cfg* cfg, gfso_gicp_result* out);
/* preprocess_points (registration_helper.cpp:22-34): returns #downsampled points; outputs optional */
int gfso_gicp_preprocess(const float* xyzw, int n, const gfso_gicp_cfg* cfg, double* pts /*n*4*/,
                         double* covs /*n*16 col-major*/, double* normals /*n*4*/);
/* exact kNN through the restated KdTree (ann/kdtree.hpp) over a double xyz1 cloud */
void gfso_knn(const double* pts, int n, const double* queries, int nq, int k, int64_t* idx, double* sqd);
/* Eigen SelfAdjointEigenSolver<Matrix3d>::computeDirect restatement (column-major in/out) */
void gfso_eig3_direct(const double* m, double* evals, double* evecs);
void gfso_se3_exp(const double* twist6, double* T16);

/* ---------------- Frame helpers (src/Frame.cc:590-623, 1314-1332) ---------------- */
int gfso_depth_to_cloud(const float* depth, int rows, int cols, int stride_elems, int downsample, float fx, float fy, float cx,
                        float cy, float* out_xyzw, int cap);
void gfso_stereo_from_rgbd(const gfso_keypoint* kps, const float* kps_un_x, int n, const float* depth, int stride_elems,
                           float bf, float* u_right, float* depth_out);

/* ---------------- Optimizer::LocalBundleAdjustment (src/Optimizer.cc:1588-2040 + Thirdparty/g2o) -------- */
typedef struct {
  int32_t n_poses, n_points, n_edges;
  const double* pose_q;      /* n_poses*4  (x,y,z,w) unit quaternion of Tcw */
  const double* pose_t;      /* n_poses*3 */
  const uint8_t* pose_fixed; /* n_poses */
  const double* points;      /* n_points*3 */
  const int32_t* edge_pose;  /* n_edges */
  const int32_t* edge_point; /* n_edges */
  const double* edge_obs;    /* n_edges*3 (u, v, u_right; u_right unused for mono) */
  const double* edge_inv_sigma2;
  const uint8_t* edge_stereo; /* 1 = EdgeStereoSE3ProjectXYZ, 0 = EdgeSE3ProjectXYZ */
  double fx, fy, cx, cy;
  double bf;
  double huber_mono, huber_stereo; /* sqrt(5.991), sqrt(7.815) */
  int32_t iterations;              /* 10 */
} gfso_lba_problem;

typedef struct {
  double* pose_q; /* n_poses*4 */
  double* pose_t; /* n_poses*3 */
  double* points; /* n_points*3 */
  double* edge_chi2;
  uint8_t* edge_depth_positive;
  int32_t iterations_run;
  double final_chi2, final_lambda;
} gfso_lba_solution;

int gfso_lba_solve(const gfso_lba_problem*, gfso_lba_solution*);
/* one buildSystem (block_solver.hpp:502-558): Hpp (n_free*36 diag blocks), Hll (n_points*9), b (6*n_free+3*n_points),
 * Hpl per edge (18, pose-rows x point-cols, zero for fixed poses); returns active robust chi2 */
double gfso_lba_linearize(const gfso_lba_problem*, double* Hpp, double* Hll, double* Hpl, double* bp, double* bl,
                          double* edge_chi2);

/* ---- Optimizer::PoseOptimization (src/Optimizer.cc:763-1098): motion-only BA of one frame ---- */
typedef struct {
  double q[4], t[3];          /* Tcw as handed to g2o::SE3Quat (x,y,z,w | translation), doubles cast from the Sophus floats */
  int32_t n_obs;              /* key-points with a MapPoint, in key-point index order */
  const double* xw;           /* n_obs*3: pMP->GetWorldPos().cast<double>() */
  const double* obs;          /* n_obs*3: kpUn.pt.x, kpUn.pt.y, mvuRight (any value when mono) */
  const float* inv_sigma2;    /* n_obs: mvInvLevelSigma2[kpUn.octave] */
  const uint8_t* stereo;      /* n_obs: mvuRight[i] >= 0 */
  double fx, fy, cx, cy, bf;  /* floats of the Frame widened to double */
  int32_t n_rounds;           /* 4; its[] = 10 each */
  int32_t its;                /* 10 */
} gfso_pose_problem;

typedef struct {
  uint8_t* outlier;     /* n_obs: pFrame->mvbOutlier */
  double* chi2;         /* n_obs: e->chi2() read by the last classification */
  double q[4], t[3];    /* vSE3_recov->estimate() (the reference computes it but never writes it back: SURVEY F12) */
  float avg_reproj_error; /* value handed to SetFrame2FrameReprojError / SetFrame2MapReprojError by the last round */
  int32_t n_inliers;    /* return value: nInitialCorrespondences - nBad */
  int32_t rounds_run, iterations_run; /* LM iterations summed over the rounds */
} gfso_pose_solution;

int gfso_pose_optimization(const gfso_pose_problem*, gfso_pose_solution*);

/* ---- ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (src/ORBmatcher.cc:1853-2063), Nleft == -1 ---- */
typedef struct {
  int32_t n_last;               /* LastFrame key-points with a MapPoint that are not outliers, in index order */
  const float* last_xw;         /* n_last*3 pMP->GetWorldPos() */
  const uint8_t* last_desc;     /* n_last*32 pMP->GetDescriptor() */
  const int32_t* last_octave;   /* LastFrame.mvKeys[i].octave */
  const float* last_angle;      /* LastFrame.mvKeysUn[i].angle */
  const uint8_t* last_mp_has_obs; /* pMP->Observations() > 0 */
  int32_t n_cur;
  const float* cur_xy;          /* n_cur*2 CurrentFrame.mvKeysUn[i].pt */
  const int32_t* cur_octave;
  const float* cur_angle;
  const float* cur_u_right;     /* mvuRight */
  const uint8_t* cur_desc;      /* n_cur*32 mDescriptors */
  const uint8_t* cur_has_mp_obs; /* mvpMapPoints[i] != NULL && Observations() > 0 on entry */
  float Tcw_q[4], Tcw_t[3], Tlw_q[4], Tlw_t[3]; /* Sophus::SE3f unit quaternions (x,y,z,w) + translations */
  float fx, fy, cx, cy, bf, b;
  float min_x, max_x, min_y, max_y, grid_w_inv, grid_h_inv;
  const float* scale_factors;   /* mvScaleFactors */
  int32_t n_levels;
  float th;
  int32_t mono, check_orientation;
} gfso_sbp_problem;
/* cur_match[n_cur]: >= 0 CurrentFrame.mvpMapPoints[i] := map point of that last-list entry; -1 left as it was;
 * -2 reset to NULL by the rotation-consistency check.  Returns nmatches. */
int gfso_search_by_projection(const gfso_sbp_problem*, int32_t* cur_match);

/* ---- ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints) (src/ORBmatcher.cc:43-206),
 *      Nleft == -1: the map points that pass the filters of :53-58, with the projection Frame::isInFrustum left on them ---- */
typedef struct {
  int32_t n_mp;
  const float* mp_proj;        /* n_mp*3 mTrackProjX, mTrackProjY, mTrackProjXR */
  const int32_t* mp_level;     /* mnTrackScaleLevel */
  const float* mp_view_cos;    /* mTrackViewCos */
  const uint8_t* mp_desc;      /* n_mp*32 */
  const uint8_t* mp_has_obs;   /* Observations() > 0 */
  int32_t n_cur;
  const float* cur_xy;
  const int32_t* cur_octave;
  const float* cur_u_right;
  const uint8_t* cur_desc;
  const uint8_t* cur_has_mp_obs;
  float min_x, min_y, grid_w_inv, grid_h_inv;
  const float* scale_factors;
  int32_t n_levels;
  float th, nn_ratio;
} gfso_sbp_map_problem;
/* cur_match[n_cur]: >= 0 F.mvpMapPoints[i] := that list entry, -1 untouched.  Returns nmatches. */
int gfso_search_by_projection_map(const gfso_sbp_map_problem*, int32_t* cur_match);

/* ---- gms_matcher(vkp1, size1, vkp2, size2, vDMatches).GetInlierMask(mask, false, false)
 *      (Thirdparty/GMS/include/gms_matcher.h:43-60, 289-301, 356-466; call sites src/ORBmatcher.cc:761-762, 812-813, 893-894) ---- */
int gfso_gms_inlier_mask(const float* kp1_xy, int n1, int width1, int height1, const float* kp2_xy, int n2, int width2,
                         int height2, const int32_t* query_idx, const int32_t* train_idx, int n_matches, uint8_t* inlier);

/* ---- optical-flow front end: cv::buildOpticalFlowPyramid (src/Frame.cc:373,505,1415), cv::calcOpticalFlowPyrLK and
 *      ORBmatcher::fbKltTracking (src/ORBmatcher.cc:2186-2297) == Tracking::fbKltTracking (src/Tracking.cc:3262-3366).
 *      Pyramid storage (shared with the HIP path so whole buffers can be compared): level l is an image of
 *      (lw[l] + 2*win) x (lh[l] + 2*win) bytes at byte offset off[l] of pyr_img (reflect-101 border), and the same number of
 *      short2 (Scharr dx, dy; zero border) at element offset off[l] of pyr_deriv.  See klt_oracle.cpp for the one deliberate
 *      difference to OpenCV (exact integer sums instead of build-dependent float summation). ---- */
#define GFSO_KLT_MAX_LEVELS 8
#define GFSO_KLT_USE_INITIAL_FLOW 4   /* cv::OPTFLOW_USE_INITIAL_FLOW */
#define GFSO_KLT_GET_MIN_EIGENVALS 8  /* cv::OPTFLOW_LK_GET_MIN_EIGENVALS */
/* Returns the number of levels buildOpticalFlowPyramid(…, maxLevel) produces; lw/lh [levels], off [levels + 1] (may be NULL). */
/* 0: exact integer sums (default, what the HIP path implements); 1: OpenCV 4.5.4's scalar float loop; 2: four-lane float model */
void gfso_klt_set_accumulation(int mode);
int gfso_klt_layout(int w, int h, int win, int max_level, int32_t* lw, int32_t* lh, int64_t* off);
int gfso_klt_build_pyramid(const uint8_t* img, int w, int h, int stride, int win, int max_level, uint8_t* pyr_img,
                           int16_t* pyr_deriv);
/* calcOpticalFlowPyrLK(prevPyr, nextPyr, prevPts, nextPts, status, err, Size(win, win), maxLevel,
 * TermCriteria(COUNT + EPS, max_iter, eps), flags, minEigThreshold); pyramids built with pyr_max_level. */
int gfso_klt_track(const uint8_t* prev_img, const int16_t* prev_deriv, const uint8_t* next_img, int w, int h, int win,
                   int pyr_max_level, int max_level, int n, const float* prev_pts, float* next_pts, uint8_t* status, float* err,
                   int max_iter, double eps, int flags, double min_eig_thr);
/* fbKltTracking: priors in/out (vpriorkps), kpstatus[n] out; returns the number of points that survive both passes. */
int gfso_fb_klt_tracking(const uint8_t* prev_img, const int16_t* prev_deriv, const uint8_t* cur_img, const int16_t* cur_deriv, int w,
                         int h, int win, int pyr_max_level, int nbpyrlvl, float ferr, float fmax_fbklt_dist, int n, const float* kps,
                         float* priors, uint8_t* kpstatus);

/* ---- cv::findFundamentalMat(pts1, pts2, cv::FM_RANSAC, threshold, confidence, mask) for n >= 15 points (call sites
 *      src/ORBmatcher.cc:236, 2399, 2463; src/Tracking.cc:1974): 7-point RANSAC with cv::RNG((uint64)-1).  See fmat_oracle.cpp for the
 *      two deliberate differences (null-space basis and cubic solver in basic arithmetic only).  pts: n x 2 floats.  Returns the
 *      inlier count of the best model (0 = no model, -2 = fewer than 15 points: OpenCV runs LMedS there, not restated). ---- */
/* 0: Gauss-Jordan null space + bisection (default, what the HIP path implements); 1: OpenCV's Jacobi SVD + cv::solveCubic */
void gfso_fmat_set_solver(int mode);
int gfso_fundamental_ransac(const float* pts1, const float* pts2, int n, double threshold, double confidence, int max_iters,
                            uint8_t* mask, double* F_out, int* iterations_run);

#ifdef __cplusplus
}
#endif
#endif
