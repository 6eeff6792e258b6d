/* gfs_abi.h — C ABI of the MI355X-native GeoFlow-SLAM front-end hot path (libgfs_hip.so).
 *
 * This is the drop-in boundary.  The reference has no FFI layer: the seams are four
 * ordinary C++ methods (sections 1-4), followed by the neighbouring rows of SURVEY.md §8(f) (sections 5-10).  Each entry point below names the reference interface it replaces
 * (file:line in HorizonRobotics/GeoFlowSlam).  Signatures are plain C: pointers, sizes, POD structs.
 * No torch / OpenCV / Eigen types cross this boundary.  INTEGRATION.md shows the reference-side
 * adaptor (what a maintainer adds to src/ORBextractor.cc etc. to call these).
 *
 * Conventions
 *   - every function returns GFS_OK (0) or a negative gfs_status unless documented otherwise;
 *     gfs_last_error() returns a thread-local human-readable message for the last failure.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute entry point fails with
 *     GFS_ERR_NO_DEVICE.
 *   - "host" pointers are ordinary process memory; "dev" pointers are HIP device pointers on the
 *     handle's device.  `stream` is a hipStream_t passed as void* (NULL = the handle's own stream).
 *   - matrices are column-major (Eigen default), quaternions are (x, y, z, w).
 */
#ifndef GFS_ABI_H_
#define GFS_ABI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFS_ABI_VERSION 1

typedef enum {
  GFS_OK = 0,
  GFS_ERR_INVALID_ARG = -1,
  GFS_ERR_NO_DEVICE = -2,   /* no gfx950 GPU / HIP runtime unusable */
  GFS_ERR_HIP = -3,         /* a HIP call failed; see gfs_last_error() */
  GFS_ERR_CAPACITY = -4,    /* caller buffer or handle capacity too small */
  GFS_ERR_UNSUPPORTED = -5, /* configuration outside what the kernels implement */
  GFS_ERR_STOPPED = -6      /* stop flag was raised (LBA) */
} gfs_status;

int gfs_abi_version(void);
const char* gfs_last_error(void);
/* number of usable gfx950 devices (0 if none / HIP unavailable) */
int gfs_device_count(void);

/* ============================================================================================
 * 1. ORB extraction — replaces ORB_SLAM3::ORBextractor
 *      ctor        include/ORBextractor.h:53-54, src/ORBextractor.cc:421-479
 *      operator()  include/ORBextractor.h:61-64, src/ORBextractor.cc:1145-1225
 *      getters     include/ORBextractor.h:66-80
 *    called from Frame::ExtractORB (src/Frame.cc:768-777).
 * ============================================================================================ */

/* Field order and types of cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id): 28 bytes. */
typedef struct {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} gfs_keypoint;

typedef struct gfs_orb gfs_orb; /* opaque; re-entrant: concurrent calls on one handle serialise internally (F9) */

typedef struct {
  int32_t nfeatures;   /* ORBextractor.nFeatures   (1000) */
  float scale_factor;  /* ORBextractor.scaleFactor (1.2; must give non-integer level ratios) */
  int32_t nlevels;     /* ORBextractor.nLevels     (8)   */
  int32_t ini_th_fast; /* ORBextractor.iniThFAST   (20)  */
  int32_t min_th_fast; /* ORBextractor.minThFAST   (7)   */
  int32_t max_rows, max_cols; /* largest image the handle must accept */
  int32_t max_batch;   /* largest batch for the *_batch entry points */
  int32_t device;      /* HIP device ordinal */
  int32_t blur_taps_variant; /* 0: OpenCV >= 4.5.1 {18,34,48,56,48,34,18}; 1: 4.0-4.5.0 {18,34,49,55,49,34,18} */
} gfs_orb_config;

void gfs_orb_default_config(gfs_orb_config* cfg);
int gfs_orb_create(const gfs_orb_config* cfg, gfs_orb** out);
void gfs_orb_destroy(gfs_orb* h);

/* GetLevels / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
 * (include/ORBextractor.h:66-80) plus the per-level feature quota and the umax table. Any pointer may be NULL. */
int gfs_orb_get_tables(const gfs_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                       int32_t* features_per_level, int32_t* umax16);
/* upper bound of keypoints one image can produce (sum of quotas + 3 per level: the octree may overshoot). */
int gfs_orb_max_keypoints(const gfs_orb* h);

/* operator()(image, mask (ignored), keypoints, descriptors, vLappingArea): host image (CV_8UC1, `stride`
 * bytes per row) -> kps[cap], desc[cap*32], *n = number of keypoints.
 * RETURNS monoIndex (>= 0; == *n on the RGB-D path where lapping = {0,0}), -1 for an empty image
 * (src/ORBextractor.cc:1150), or a gfs_status < -1 ... NOTE: to keep -1 unambiguous, errors are
 * reported as (GFS_ERR_* - 100).  Do NOT test the return value with `< 0` (-1 is the reference's own answer for an empty image): use
 * the two macros below. */
#define GFS_ORB_EXTRACT_FAILED(rc) ((rc) < -1)                            /* an error of the library, not the reference's -1 */
#define GFS_ORB_EXTRACT_STATUS(rc) ((rc) < -1 ? (rc) + 100 : GFS_OK)      /* the gfs_status behind a failed call */
int gfs_orb_extract(gfs_orb* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1,
                    gfs_keypoint* kps, uint8_t* desc, int cap, int* n);

/* Batched operator() over B independent host images of identical size: image b at imgs[b].
 * kps: [B][cap], desc: [B][cap][32], n/mono_index: [B]. */
int gfs_orb_extract_batch(gfs_orb* h, const uint8_t* const* imgs, int B, int rows, int cols, int stride, int lap0,
                          int lap1, gfs_keypoint* kps, uint8_t* desc, int cap, int* n, int* mono_index);

/* Device-resident batch: dev_imgs is [B][rows][cols] dense u8 already in HBM.  Results stay in HBM in
 * handle-owned buffers (see gfs_orb_device_results) until the next call on this handle. */
int gfs_orb_extract_batch_device(gfs_orb* h, const void* dev_imgs, int B, int rows, int cols, int lap0, int lap1,
                                 void* stream);
/* Device result views of the last *_device call: kps [B][cap] gfs_keypoint, desc [B][cap][32] u8,
 * counts [B] int32 (n), mono [B] int32. */
int gfs_orb_device_results(gfs_orb* h, void** dev_kps, void** dev_desc, void** dev_counts, void** dev_mono, int* cap);
/* Copy the last device results of image b to host buffers. */
int gfs_orb_fetch(gfs_orb* h, int b, gfs_keypoint* kps, uint8_t* desc, int cap, int* n, int* mono_index);

/* Introspection used by the parity tests (mvImagePyramid is a public member of the reference class,
 * include/ORBextractor.h:82): copies level `level` of image b of the last call (un-padded, rows*cols). */
int gfs_orb_level_size(const gfs_orb* h, int level, int* rows, int* cols);
int gfs_orb_fetch_level(gfs_orb* h, int b, int level, int blurred, uint8_t* dst);
/* FAST candidates handed to the octree for (image b, level): x, y relative to (16,16), score. Returns count. */
int gfs_orb_fetch_candidates(gfs_orb* h, int b, int level, int32_t* x, int32_t* y, int32_t* score, int cap);
/* Host-logic test hook (no GPU needed): the library's DistributeOctTree (src/ORBextractor.cc:567-768) on caller
 * candidates (integer x, y < 4096 relative to (min_x, min_y); score 0..255). Returns the number kept;
 * out_idx[i] = input index of the i-th kept keypoint in the reference's std::list order. */
int gfs_orb_octree_host(const int32_t* x, const int32_t* y, const int32_t* score, int n, int min_x, int max_x,
                        int min_y, int max_y, int n_features, int32_t* out_idx, int cap);
/* GPU test hook: the device DistributeOctTree kernel on caller candidates (min_x = min_y = 16). Writes the kept
 * candidates (x, y, score) in list order; returns their number. */
int gfs_orb_octree_device(int device, const int32_t* x, const int32_t* y, const int32_t* score, int n, int min_x, int max_x,
                          int min_y, int max_y, int n_features, int32_t* out_x, int32_t* out_y, int32_t* out_score, int cap);
/* (the gfs_test_* hooks of the library's internal replicas -- sort, glibc math, traffic calibration -- are in gfs_abi_test.h) */

/* ============================================================================================
 * 2. Brute-force Hamming matching — replaces
 *      ORBmatcher::DescriptorDistance                       include/ORBmatcher.h:41, src/ORBmatcher.cc:2536-2550
 *      cv::BFMatcher(NORM_HAMMING).match(d1, d2, matches)   src/ORBmatcher.cc:755-756, 805-806, 888-889
 * ============================================================================================ */

/* 256-bit Hamming distance of two 32-byte descriptors (pure host helper, bit-identical to DescriptorDistance). */
int gfs_hamming256(const uint8_t* a, const uint8_t* b);

typedef struct gfs_matcher gfs_matcher;
int gfs_matcher_create(int device, int max_query, int max_train, int max_batch, gfs_matcher** out);
void gfs_matcher_destroy(gfs_matcher* h);

/* matcher.match(query, train): for every query row i the train row j of minimum Hamming distance, lowest j on
 * ties -> train_idx[i], dist[i] (DMatch{queryIdx=i, trainIdx=train_idx[i], imgIdx=0, distance=(float)dist[i]}).
 * Returns the number of matches written: nq, or 0 when the train set is empty (no matches, as OpenCV). */
int gfs_bf_match_hamming(gfs_matcher* h, const uint8_t* query, int nq, const uint8_t* train, int nt,
                         int32_t* train_idx, int32_t* dist);

/* Device-resident batch of B independent pairs.  dev_query/dev_train: [B][stride_rows][32] u8;
 * dev_nq/dev_nt: [B] int32 row counts (device memory); outputs dev_train_idx/dev_dist: [B][stride_rows] int32
 * (entries >= nq[b] untouched; train_idx = -1, dist = INT32_MAX when nt[b] == 0). */
int gfs_bf_match_hamming_batch_device(gfs_matcher* h, const void* dev_query, const void* dev_nq, const void* dev_train,
                                      const void* dev_nt, int B, int stride_rows, void* dev_train_idx, void* dev_dist,
                                      void* stream);

/* ============================================================================================
 * 3. GICP registration — replaces RegistrationGICP::RegisterPointClouds
 *      include/RegistrationGICP.h:25-28, src/RegistrationGICP.cc:5-20
 *    (small_gicp::align<float,4> with GICPFactor + LevenbergMarquardtOptimizer,
 *     Thirdparty/small_gicp/src/small_gicp/registration/registration_helper.cpp:57-122)
 *    called from Tracking::PredictStateICP (src/Tracking.cc:3380-3382).
 * ============================================================================================ */
typedef struct {
  int32_t num_threads;                /* kept for signature parity (src/RegistrationGICP.cc:10); ignored on GPU */
  double downsampling_resolution;     /* 0.02 */
  double max_correspondence_distance; /* 0.1  */
  double rotation_eps;                /* 0.1 * pi / 180 */
  double translation_eps;             /* 1e-3 */
  int32_t max_iterations;             /* 20 */
  int32_t num_neighbors;              /* 10 */
} gfs_gicp_config;

/* Field-for-field small_gicp::RegistrationResult (registration/registration_result.hpp:10-30). */
typedef struct {
  double T_target_source[16]; /* column-major 4x4 */
  int32_t converged;
  uint64_t iterations;
  uint64_t num_inliers;
  double H[36]; /* column-major 6x6 */
  double b[6];
  double error;
  /* extra diagnostics */
  int32_t n_target_downsampled, n_source_downsampled, n_linearize, n_error_evals;
} gfs_gicp_result;

typedef struct gfs_gicp gfs_gicp;
void gfs_gicp_default_config(gfs_gicp_config* cfg);
int gfs_gicp_create(int device, int max_points, int max_batch, gfs_gicp** out);
void gfs_gicp_destroy(gfs_gicp* h);

/* RegisterPointClouds(target_points, source_points, init_T_target_source): clouds are arrays of
 * Eigen::Vector4f (x, y, z, w; w ignored and forced to 1 like points/point_cloud.hpp:29). */
int gfs_gicp_align(gfs_gicp* h, const float* target_xyzw, int nt, const float* source_xyzw, int ns,
                   const double init_T_target_source[16], const gfs_gicp_config* cfg, gfs_gicp_result* out);

/* Device-resident batch of B independent (target, source) pairs.  dev_target/dev_source: [B][stride_pts][4] f32,
 * dev_nt/dev_ns: [B] int32 (device), init_T: host [B][16] (NULL = identity), out: host [B]. */
int gfs_gicp_align_batch_device(gfs_gicp* h, const void* dev_target, const void* dev_nt, const void* dev_source,
                                const void* dev_ns, int B, int stride_pts, const double* init_T,
                                const gfs_gicp_config* cfg, gfs_gicp_result* out, void* stream);

/* Streaming form (the gfs_gicp_cache of SURVEY.md 8b) for Tracking::PredictStateICP, where the target of every call is the
 * source of the previous one (src/Tracking.cc:3375-3382: target = mLastFrame.source_points, source = mCurrentFrame's): the
 * handle keeps the preprocessed clouds (voxel means, covariances, search grid) of its last call's SOURCES in HBM; these
 * calls register new source clouds against them and preprocess only the new clouds.  The result is bit-identical to
 * gfs_gicp_align*(previous sources, new sources) — preprocessing is deterministic, so reusing it changes nothing but the time.
 * GFS_ERR_INVALID_ARG unless the previous call on this handle had the same batch size, stride and preprocessing parameters. */
int gfs_gicp_align_next(gfs_gicp* h, const float* source_xyzw, int ns, const double init_T_target_source[16],
                        const gfs_gicp_config* cfg, gfs_gicp_result* out);
int gfs_gicp_align_next_batch_device(gfs_gicp* h, const void* dev_source, const void* dev_ns, int B, int stride_pts,
                                     const double* init_T, const gfs_gicp_config* cfg, gfs_gicp_result* out, void* stream);

/* Introspection for parity tests: preprocessing output (voxel means + covariances) of cloud `which`
 * (0 = target, 1 = source) of pair b of the last call. pts: [m][4] f64, covs: [m][9] f64 (3x3 col-major). */
int gfs_gicp_fetch_preprocessed(gfs_gicp* h, int b, int which, double* pts, double* covs, int cap, int* m);
/* Diagnostics: workgroups of the linearisation kernel since the last reset, by outcome of staging their tile of the target cloud in
 * LDS: out8[0] staged; [1] no dense grid; [2] no usable point; [3] / [4] / [5] too many rows / points / cell boundaries for the
 * tile (those workgroups search the cloud in HBM: same results); [6] tiling switched off.  Counted only by handles created with
 * GFS_GICP_TILE_STATS=1 in the environment (one atomic per workgroup on one address is not free). */
int gfs_gicp_tile_stats(gfs_gicp* h, unsigned long long* out8, int reset);
/* Diagnostics (tools/knn_probe.py): out = {down-sampled points, queries deferred to the r = 2 pass, queries deferred to the
 * isolated-point pass} of cloud (b, which) of the last call; dk (may be NULL): the squared-distance bounds of the latter. */
int gfs_gicp_knn_stats(gfs_gicp* h, int b, int which, int out[3], double* dk, int cap);
/* Diagnostics: how the Levenberg-Marquardt loops of this handle's calls were driven.  out = {launches of the cooperative kernel
 * (the loop of a few pairs in ONE launch: a single live stream, the tail of a batch) so far, workgroups of the last such launch,
 * 1 if a launch ever failed to become resident and the handle fell back to a launch per step for good, the workgroup budget}. */
int gfs_gicp_coop_stats(gfs_gicp* h, int out[4]);

/* ============================================================================================
 * 4. Local bundle adjustment — replaces the numeric core of Optimizer::LocalBundleAdjustment
 *      include/Optimizer.h:62-65, src/Optimizer.cc:1588-2040 (graph build 1662-1953, optimize(10) 1958-1959,
 *      chi2 classification 1961-1999) with g2o BlockSolver<6,3> + Levenberg (Thirdparty/g2o).
 *    The pointer-graph gather / write-back (src/Optimizer.cc:1592-1660, 2003-2039) stays in the C++ adaptor.
 * ============================================================================================ */
typedef struct {
  int32_t n_poses, n_points, n_edges;
  const double* pose_q;      /* [n_poses][4] unit quaternion (x,y,z,w) of Tcw */
  const double* pose_t;      /* [n_poses][3] */
  const uint8_t* pose_fixed; /* [n_poses] 1 = fixed (setFixed(true), src/Optimizer.cc:1697,1715) */
  const double* points;      /* [n_points][3] world coordinates */
  const int32_t* edge_pose;  /* [n_edges] */
  const int32_t* edge_point; /* [n_edges]; several edges may join the same (pose, point) pair, as in g2o */
  const double* edge_obs;    /* [n_edges][3] (u, v, u_right); u_right ignored for mono edges */
  const double* edge_inv_sigma2; /* [n_edges] information = inv_sigma2 * I */
  const uint8_t* edge_stereo;    /* [n_edges] 1 = EdgeStereoSE3ProjectXYZ, 0 = EdgeSE3ProjectXYZ */
  double fx, fy, cx, cy, bf;
  double huber_mono, huber_stereo; /* sqrt(5.991), sqrt(7.815) (src/Optimizer.cc:1728-1729) */
  int32_t iterations;              /* 10 (src/Optimizer.cc:1959) */
} gfs_lba_problem;

typedef struct {
  double* pose_q;               /* [n_poses][4] */
  double* pose_t;               /* [n_poses][3] */
  double* points;               /* [n_points][3] */
  double* edge_chi2;            /* [n_edges] e->chi2() at the solution */
  uint8_t* edge_depth_positive; /* [n_edges] e->isDepthPositive() */
  int32_t iterations_run;
  double final_chi2, final_lambda;
} gfs_lba_solution;

typedef struct gfs_lba gfs_lba;
int gfs_lba_create(int device, int max_poses, int max_points, int max_edges, gfs_lba** out);
void gfs_lba_destroy(gfs_lba* h);
/* optimizer.optimize(iterations) with the stop flag polled like setForceStopFlag (src/Optimizer.cc:1679). */
int gfs_lba_solve(gfs_lba* h, const gfs_lba_problem* p, gfs_lba_solution* s, volatile const int* stop);
/* The same with the reference's own flag type: `stop` points at a C++ bool (one byte), e.g. pbStopFlag = &mbAbortBA of
 * LocalMapping (src/LocalMapping.cc: mbAbortBA is raised by the tracking thread while the adjustment runs).  It is read live,
 * at the top of every iteration and after every trial step, like g2o's setForceStopFlag (src/Optimizer.cc:1679). */
int gfs_lba_solve_bool(gfs_lba* h, const gfs_lba_problem* p, gfs_lba_solution* s, const volatile unsigned char* stop);
/* One BlockSolver::buildSystem (core/block_solver.hpp:502-558): Hpp [n_free][36] (col-major 6x6 diagonal blocks,
 * free poses in ascending index order), Hll [n_points][9], Hpl per edge [n_edges][18] (6x3 col-major, zero for
 * edges on fixed poses), bp [n_free][6], bl [n_points][3], edge_chi2 [n_edges]. Returns robust chi2 in *chi2. */
int gfs_lba_linearize(gfs_lba* h, const gfs_lba_problem* p, double* Hpp, double* Hll, double* Hpl, double* bp,
                      double* bl, double* edge_chi2, double* chi2);
/* n independent windows solved together (replicas: the LBA of one map does not shard; a server bundle-adjusting many maps, or
 * several candidate windows, fills the GPU this way).  Per window the result is bit-identical to gfs_lba_solve; the phase
 * kernels of all windows share their launches.  `stop` (may be NULL) is polled between LM trials: when raised, every window
 * closes its running iteration like SparseOptimizer::terminate() and returns what it has. */
typedef struct gfs_lba_batch gfs_lba_batch;
int gfs_lba_batch_create(int device, int max_windows, int max_poses, int max_points, int max_edges, gfs_lba_batch** out);
void gfs_lba_batch_destroy(gfs_lba_batch* h);
int gfs_lba_solve_batch(gfs_lba_batch* h, const gfs_lba_problem* problems, gfs_lba_solution* solutions, int n,
                        volatile const int* stop);

/* ============================================================================================
 * 5. Frame helpers next to the hot path — keep GICP input and RGB-D "stereo" coordinates on device
 *      Frame::ConvertDepthToPointCloud(downSample, ...)   src/Frame.cc:590-623
 *      Frame::ComputeStereoFromRGBD(imDepth)              src/Frame.cc:1314-1332
 * ============================================================================================ */
typedef struct gfs_frame gfs_frame;
int gfs_frame_create(int device, int max_rows, int max_cols, int max_keypoints, gfs_frame** out);
void gfs_frame_destroy(gfs_frame* h);
/* depth: CV_32F metres, `stride_elems` floats per row.  Emits (x, y, z, 1) for every pixel of the `downsample` grid with
 * 0 < depth < 10 in raster order (the reference's push_back order): x = (u - cx) * depth / fx, y = (v - cy) * depth / fy.
 * An empty depth image yields *n = 0 (the reference logs an error and returns). */
int gfs_depth_to_cloud(gfs_frame* h, const float* depth, int rows, int cols, int stride_elems, int downsample, float fx,
                       float fy, float cx, float cy, float* out_xyzw, int cap, int* n);
/* dev_depth [B][rows][cols] f32 -> dev_out_xyzw [B][stride_pts][4] f32 + dev_counts [B] int32: exactly the cloud layout
 * gfs_gicp_align_batch_device consumes. */
int gfs_depth_to_cloud_batch_device(gfs_frame* h, const void* dev_depth, int B, int rows, int cols, int downsample, float fx,
                                    float fy, float cx, float cy, void* dev_out_xyzw, int stride_pts, void* dev_counts,
                                    void* stream);
/* imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor) for a CV_16U sensor image (src/Tracking.cc:1622-1623), on the device: the
 * depth maps then cross PCIe as 2 bytes a pixel.  dev_depth_u16 [B][rows][cols] u16 (8-byte aligned) -> dev_depth_f32 (16-byte
 * aligned), float(raw) * factor in single precision like cv::Mat::convertTo. */
int gfs_depth_convert_u16_batch_device(gfs_frame* h, const void* dev_depth_u16, int B, int rows, int cols, float factor,
                                       void* dev_depth_f32, void* stream);
/* mvDepth[i] = d = imDepth.at<float>(kp.pt.y, kp.pt.x) (float -> int truncation), mvuRight[i] = kpUn.pt.x - bf / d when
 * d > 0, else both -1.  kps_un_x may be NULL (no distortion: mvKeysUn == mvKeys). */
int gfs_stereo_from_rgbd(gfs_frame* h, const gfs_keypoint* kps, const float* kps_un_x, int n, const float* depth, int rows,
                         int cols, int stride_elems, float bf, float* u_right, float* depth_out);
int gfs_stereo_from_rgbd_batch_device(gfs_frame* h, const void* dev_kps, const void* dev_kps_un_x, const void* dev_counts,
                                      int B, int kp_stride, const void* dev_depth, int rows, int cols, float bf,
                                      void* dev_u_right, void* dev_depth_out, void* stream);
/* The RGB-D tail of the Frame constructor in one call: ComputeStereoFromRGBD(imDepth) (src/Frame.cc:1314-1332) followed by
 * ConvertDepthToPointCloud (:590-623).  The depth map is uploaded once and the cloud stays on the device: *dev_cloud /
 * *dev_count / *cloud_stride are the arguments gfs_gicp_align_batch_device / gfs_gicp_align_next_batch_device take (valid until
 * the next call on this handle).  out_xyzw may be NULL (no host copy of the cloud); *n_cloud receives the point count.
 * depth == NULL: use the depth map the previous gfs_frame_rgbd call on this handle uploaded (same rows x cols); downsample <= 0:
 * no cloud in this call.  Together: the cloud first (n = 0), the registration, and the stereo coordinates when the key-points of
 * an ORB extraction that ran beside the registration have arrived -- one depth upload either way. */
int gfs_frame_rgbd(gfs_frame* h, const gfs_keypoint* kps, const float* kps_un_x, int n, const float* depth, int rows, int cols,
                   int stride_elems, float bf, int downsample, float fx, float fy, float cx, float cy, float* u_right,
                   float* depth_out, float* out_xyzw, int cap, int* n_cloud, void** dev_cloud, void** dev_count, int* cloud_stride);

/* ============================================================================================
 * 6. Optimizer::PoseOptimization — motion-only bundle adjustment of a frame
 *      int Optimizer::PoseOptimization(Frame*, bool, bool, int)      src/Optimizer.cc:763-1098
 *    conventional-SLAM branch (pFrame->mpCamera2 == nullptr): EdgeSE3ProjectXYZOnlyPose (src/OptimizableTypes.cpp:49-63) and
 *    g2o::EdgeStereoSE3ProjectXYZOnlyPose (Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:339-404), Huber kernels,
 *    4 rounds x optimize(10) with BlockSolver_6_3 + LinearSolverDense + OptimizationAlgorithmLevenberg.
 *    The two-camera rigid-body branch (:883-951, EdgeSE3ProjectXYZOnlyPoseToBody) is not implemented.
 * ============================================================================================ */
typedef struct {
  double q[4], t[3];          /* Tcw = pFrame->GetPose(): unit quaternion (x, y, z, w) and translation cast to double (:784-786) */
  int32_t n_obs;              /* key-points that have a MapPoint, in key-point index order (= edge creation order) */
  const double* xw;           /* [n_obs][3] pMP->GetWorldPos().cast<double>() */
  const double* obs;          /* [n_obs][3] kpUn.pt.x, kpUn.pt.y, mvuRight[i] (third value ignored for mono edges) */
  const float* inv_sigma2;    /* [n_obs] mvInvLevelSigma2[kpUn.octave] */
  const uint8_t* stereo;      /* [n_obs] mvuRight[i] >= 0 */
  double fx, fy, cx, cy, bf;  /* pFrame->fx ... mbf (floats widened to double, :865-869) */
  int32_t n_rounds;           /* 4 */
  int32_t its;                /* 10 (its[] = {10, 10, 10, 10}) */
} gfs_pose_problem;

typedef struct {
  uint8_t* outlier;       /* [n_obs] pFrame->mvbOutlier after the last round */
  double* chi2;           /* [n_obs] e->chi2() as read by the last classification */
  double q[4], t[3];      /* vSE3_recov->estimate(): the reference computes it and then drops it (SetPose is commented out, :1078-1097) */
  float avg_reproj_error; /* value of the last SetFrame2FrameReprojError / SetFrame2MapReprojError call */
  int32_t n_inliers;      /* return value: nInitialCorrespondences - nBad (0 when there are fewer than 3 correspondences) */
  int32_t rounds_run, iterations_run;
} gfs_pose_solution;

typedef struct gfs_pose gfs_pose;
int gfs_pose_create(int device, int max_obs, int max_batch, gfs_pose** out);
void gfs_pose_destroy(gfs_pose* h);
/* B independent frames (host pointers), one workgroup per frame. */
int gfs_pose_optimize(gfs_pose* h, const gfs_pose_problem* problems, int B, gfs_pose_solution* solutions);
/* How the sums over a frame's edges (activeRobustChi2, the 6x6 normal equations, the inliers' mean chi2) are added up.
 *   GFS_POSE_SUMS_EDGE_ORDER (default since round 6): g2o's order, edge after edge on one lane (core/sparse_optimizer.cpp:104-122,
 *     core/base_unary_edge.hpp:43-72): every double as the sequential code computes it -- mvbOutlier, the return value, the LM
 *     iteration counts and the pose are those of the reference's single-threaded solve, bit for bit against the oracle restatement.
 *   GFS_POSE_SUMS_TREE (opt-in): a tree of fixed shape -- deterministic, the same bits for a frame alone or inside any batch, about
 *     half the latency of a single frame; against g2o's edge-by-edge sums the last bits differ.  GUARANTEED (tests/test_gpu_pose.py):
 *     pose within 1e-7 relative; an outlier flag can differ only for an edge whose chi2 sits within 7.8e-6 of 5.991 / 7.815, the LM
 *     iteration count by at most 2.  A caller that needs the reference's flags exactly keeps the default. */
#define GFS_POSE_SUMS_TREE 0
#define GFS_POSE_SUMS_EDGE_ORDER 1
int gfs_pose_set_sum_order(gfs_pose* h, int order);

/* ============================================================================================
 * 7. ORBmatcher::SearchByProjection, frame to frame — the windowed matcher of TrackWithMotionModel
 *      int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th, bool bMono)
 *                                                                             src/ORBmatcher.cc:1853-2063
 *    with Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea (src/Frame.cc:734-761, 1073-1084, 1007-1071),
 *    ORBmatcher::DescriptorDistance (:2536-2550, TH_HIGH = 100) and ComputeThreeMaxima (:2500-2532, HISTO_LENGTH = 30).
 *    Single-camera frames only (Nleft == -1); the two-camera branch (:1951-2035) is not implemented.
 * ============================================================================================ */
typedef struct {
  /* LastFrame: the key-points i with mvpMapPoints[i] != NULL && !mvbOutlier[i], in index order (the loop of :1872-1875) */
  int32_t n_last;
  const float* last_xw;           /* [n_last][3] pMP->GetWorldPos() */
  const uint8_t* last_desc;       /* [n_last][32] pMP->GetDescriptor() */
  const int32_t* last_octave;     /* [n_last] LastFrame.mvKeys[i].octave */
  const float* last_angle;        /* [n_last] LastFrame.mvKeysUn[i].angle */
  const uint8_t* last_mp_has_obs; /* [n_last] pMP->Observations() > 0 */
  /* CurrentFrame */
  int32_t n_cur;
  const gfs_keypoint* cur_kps_un; /* [n_cur] mvKeysUn (cv::KeyPoint layout) */
  const float* cur_u_right;       /* [n_cur] mvuRight */
  const uint8_t* cur_desc;        /* [n_cur][32] mDescriptors */
  const uint8_t* cur_has_mp_obs;  /* [n_cur] mvpMapPoints[i] != NULL && ->Observations() > 0 on entry */
  float Tcw_q[4], Tcw_t[3];       /* CurrentFrame.GetPose(): Sophus::SE3f unit quaternion (x, y, z, w), translation */
  float Tlw_q[4], Tlw_t[3];       /* LastFrame.GetPose() */
  float fx, fy, cx, cy;           /* CurrentFrame.mpCamera (Pinhole) */
  float bf, b;                    /* CurrentFrame.mbf, mb */
  float min_x, max_x, min_y, max_y; /* Frame::mnMinX ... mnMaxY */
  float grid_w_inv, grid_h_inv;   /* Frame::mfGridElementWidthInv / HeightInv (64 x 48 grid) */
  const float* scale_factors;     /* CurrentFrame.mvScaleFactors */
  int32_t n_levels;               /* <= 16 */
  float th;                       /* window size factor (15 mono, 7 stereo in Tracking::TrackWithMotionModel) */
  int32_t mono;                   /* bMono */
  int32_t check_orientation;      /* ORBmatcher::mbCheckOrientation */
} gfs_sbp_problem;

typedef struct gfs_sbp gfs_sbp;
/* max_last <= 8192 map points, max_cur <= 4096 key-points per frame */
int gfs_sbp_create(int device, int max_last, int max_cur, int max_batch, gfs_sbp** out);
void gfs_sbp_destroy(gfs_sbp* h);
/* B independent frame pairs (host pointers).  cur_match[f][i] (n_cur entries): >= 0 = CurrentFrame.mvpMapPoints[i] now holds
 * the map point of last-list entry cur_match[f][i]; -1 = left as it was; -2 = reset to NULL by the rotation-consistency
 * check.  nmatches[f] = the function's return value (it counts overwritten assignments twice, like the reference). */
int gfs_search_by_projection(gfs_sbp* h, const gfs_sbp_problem* problems, int B, int32_t* const* cur_match, int32_t* nmatches);

/*      int ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, float th, bool bFarPoints,
 *                                         float thFarPoints)                    src/ORBmatcher.cc:43-206
 *    (TrackLocalMap / relocalisation): the caller lists the map points that pass the filters of :53-58 (mbTrackInView, not
 *    beyond thFarPoints, !isBad()) with the projection Frame::isInFrustum left on them; the window depends on the viewing
 *    angle (RadiusByViewingCos :250-255), best / second-best ratio test (mfNNratio) when both sit on the same pyramid level. */
typedef struct {
  int32_t n_mp;
  const float* mp_proj;          /* [n_mp][3] pMP->mTrackProjX, mTrackProjY, mTrackProjXR */
  const int32_t* mp_level;       /* [n_mp] mnTrackScaleLevel */
  const float* mp_view_cos;      /* [n_mp] mTrackViewCos */
  const uint8_t* mp_desc;        /* [n_mp][32] GetDescriptor() */
  const uint8_t* mp_has_obs;     /* [n_mp] Observations() > 0 */
  int32_t n_cur;
  const gfs_keypoint* cur_kps_un; /* F.mvKeysUn */
  const float* cur_u_right;      /* F.mvuRight */
  const uint8_t* cur_desc;       /* F.mDescriptors */
  const uint8_t* cur_has_mp_obs; /* F.mvpMapPoints[i] != NULL && ->Observations() > 0 on entry */
  float min_x, min_y, grid_w_inv, grid_h_inv;
  const float* scale_factors;    /* F.mvScaleFactors */
  int32_t n_levels;
  float th;                      /* window factor (1, 3, 5 ... in Tracking::SearchLocalPoints) */
  float nn_ratio;                /* ORBmatcher::mfNNratio */
} gfs_sbp_map_problem;
/* cur_match[f][i]: >= 0 = F.mvpMapPoints[i] := map point of list entry cur_match[f][i]; -1 = left as it was. */
int gfs_search_by_projection_map(gfs_sbp* h, const gfs_sbp_map_problem* problems, int B, int32_t* const* cur_match,
                                 int32_t* nmatches);

/* ============================================================================================
 * 8. GMS filter of the brute-force matches (the second half of ORBmatcher::SearchWithGMS / SearchForInitializationWithGMS)
 *      gms_matcher gms(kp1, frameSize, kp2, frameSize, matches_all); nmatches = gms.GetInlierMask(vbInliers, false, false);
 *                                                                     src/ORBmatcher.cc:761-762, 812-813, 893-894
 *      Thirdparty/GMS/include/gms_matcher.h:43-60 (constructor), 289-301 (GetInlierMask), 356-455 (run, no scale / rotation)
 * ============================================================================================ */
typedef struct {
  int32_t n1, n2;
  const gfs_keypoint* kp1;   /* [n1] vkp1 (cv::KeyPoint layout; only pt is used) */
  const gfs_keypoint* kp2;   /* [n2] vkp2 */
  int32_t width1, height1, width2, height2; /* size1, size2 */
  int32_t n_matches;
  const int32_t* query_idx;  /* [n_matches] vDMatches[i].queryIdx */
  const int32_t* train_idx;  /* [n_matches] vDMatches[i].trainIdx */
} gfs_gms_problem;

typedef struct gfs_gms gfs_gms;
int gfs_gms_create(int device, int max_keypoints /* <= 8192 */, int max_batch, gfs_gms** out);
void gfs_gms_destroy(gfs_gms* h);
/* B pairs (host pointers): inlier[f][i] = vbInliers[i] (0 / 1), n_inliers[f] = the return value. */
int gfs_gms_inlier_mask(gfs_gms* h, const gfs_gms_problem* problems, int B, uint8_t* const* inlier, int32_t* n_inliers);
/* Device-resident chain ORB -> BF match -> GMS: dev_kps1 / dev_kps2 [B][kp_stride] gfs_keypoint and dev_n1 / dev_n2 [B] as
 * returned by gfs_orb_device_results, dev_train_idx [B][kp_stride] as written by gfs_bf_match_hamming_batch_device (match i
 * of pair b = (i, dev_train_idx[b][i]), i < n1[b]; no matches when n2[b] == 0).  dev_mask [B][kp_stride] u8, dev_counts [B]. */
int gfs_gms_inlier_mask_batch_device(gfs_gms* h, const void* dev_kps1, const void* dev_n1, const void* dev_kps2, const void* dev_n2,
                                     int B, int kp_stride, const void* dev_train_idx, int width, int height, void* dev_mask,
                                     void* dev_counts, void* stream);

/* ============================================================================================
 * 9. Optical-flow front end (the "GeoFlow" stream): replaces
 *      cv::buildOpticalFlowPyramid(image, mImGray, Size(w, w), 3)             src/Frame.cc:373, 505, 1415
 *      cv::calcOpticalFlowPyrLK(prevPyr, nextPyr, ...)                         src/ORBmatcher.cc:2224, 2271; src/Tracking.cc:3298, 3342
 *      ORBmatcher::fbKltTracking / Tracking::fbKltTracking                     src/ORBmatcher.cc:2186-2297; src/Tracking.cc:3262-3366
 *    (the projection of the priors, the mask bookkeeping and cv::findFundamentalMat of SearchByProjectionWithOF,
 *    src/ORBmatcher.cc:2304-2497, stay with the caller).  A gfs_klt_pyramid holds the pyramids of up to max_batch frames in HBM:
 *    level l of a frame is an image of (lw[l] + 2*win) x (lh[l] + 2*win) bytes at byte offset off[l] (BORDER_REFLECT_101 border
 *    of win pixels) plus as many short2 (Scharr dx, dy; zero border); a level is built while both sides exceed the window, as
 *    OpenCV does.  A frame's pyramid is built once and used as `cur` for one pair and as `prev` for the next.
 *    Sums of the 2x2 system are exact integer sums rounded once: bit-equal to oracle/klt_oracle.cpp.
 * ============================================================================================ */
#define GFS_KLT_MAX_LEVELS 8
#define GFS_KLT_USE_INITIAL_FLOW 4   /* cv::OPTFLOW_USE_INITIAL_FLOW */
#define GFS_KLT_GET_MIN_EIGENVALS 8  /* cv::OPTFLOW_LK_GET_MIN_EIGENVALS */
typedef struct gfs_klt gfs_klt;
typedef struct gfs_klt_pyramid gfs_klt_pyramid;
/* win = LKWindowSize (3 .. 63), max_level = the maxLevel given to buildOpticalFlowPyramid (the reference: 3). */
int gfs_klt_create(int device, int width, int height, int win, int max_level, int max_batch, int max_points, gfs_klt** out);
void gfs_klt_destroy(gfs_klt* h);
/* Number of levels held; lw / lh [levels] and off [levels + 1] may be NULL.  off[levels] = bytes of one frame's images. */
int gfs_klt_layout(const gfs_klt* h, int32_t* lw, int32_t* lh, int64_t* off);
int gfs_klt_pyramid_create(gfs_klt* h, gfs_klt_pyramid** out);
void gfs_klt_pyramid_destroy(gfs_klt_pyramid* p);
/* cv::buildOpticalFlowPyramid for B frames: images[f] = host pointer to height rows of `stride` bytes. */
int gfs_klt_build_pyramid(gfs_klt* h, gfs_klt_pyramid* pyr, const uint8_t* const* images, int stride, int B);
/* Same from device memory: dev_images = [B][height][stride] bytes; asynchronous on `stream` when it is not NULL. */
int gfs_klt_build_pyramid_device(gfs_klt* h, gfs_klt_pyramid* pyr, const void* dev_images, int stride, int B, void* stream);
/* Copies frame f's pyramid to the host: img [off[levels]] bytes, deriv [off[levels] * 2] int16 (parity tests). */
int gfs_klt_pyramid_download(gfs_klt* h, const gfs_klt_pyramid* pyr, int f, uint8_t* img, int16_t* deriv);
/* cv::calcOpticalFlowPyrLK(prev, next, prev_pts, next_pts, status, err, Size(win, win), max_level,
 * TermCriteria(COUNT + EPS, max_iter, eps), flags, min_eig_thr) for B pairs; n_points[f] <= max_points;
 * prev_pts[f] / next_pts[f]: n x 2 floats (next_pts is read when GFS_KLT_USE_INITIAL_FLOW is set). */
int gfs_klt_track(gfs_klt* h, const gfs_klt_pyramid* prev, const gfs_klt_pyramid* next, int B, const int32_t* n_points,
                  const float* const* prev_pts, float* const* next_pts, uint8_t* const* status, float* const* err, int max_level,
                  int max_iter, double eps, int flags, double min_eig_thr);
/* fbKltTracking(prev, cur, win, nbpyrlvl, ferr, fmax_fbklt_dist, kps, priors, kpstatus) for B pairs, both passes and the gates
 * in one kernel: priors[f] in/out (vpriorkps), kpstatus[f][i] 0 / 1, n_good[f] = points that survive both passes. */
int gfs_klt_fb_track(gfs_klt* h, const gfs_klt_pyramid* prev, const gfs_klt_pyramid* cur, int B, const int32_t* n_points,
                     const float* const* kps, float* const* priors, uint8_t* const* kpstatus, int32_t* n_good, int nbpyrlvl,
                     float ferr, float fmax_fbklt_dist);
/* Device-resident form: dev_kps / dev_priors [B][pt_stride][2] float, dev_n [B] int32, dev_kpstatus [B][pt_stride] u8,
 * dev_n_good [B] int32; asynchronous on `stream` when it is not NULL. */
int gfs_klt_fb_track_device(gfs_klt* h, const gfs_klt_pyramid* prev, const gfs_klt_pyramid* cur, int B, int pt_stride,
                            const void* dev_n, const void* dev_kps, void* dev_priors, void* dev_kpstatus, void* dev_n_good,
                            int nbpyrlvl, float ferr, float fmax_fbklt_dist, void* stream);

/* The bookkeeping around the F check of SearchByProjectionWithOF on the device (src/ORBmatcher.cc:2386-2406): the tracks that survived
 * fbKltTracking, in order, as the two point lists of the check plus their original indices (dev_out_a / dev_out_b
 * [B][pt_stride][2] float, dev_out_index [B][pt_stride] int32, dev_out_n [B] int32) ... */
int gfs_klt_compact_tracks_device(gfs_klt* h, int B, int pt_stride, const void* dev_n, const void* dev_kps, const void* dev_priors,
                                  const void* dev_kpstatus, void* dev_out_a, void* dev_out_b, void* dev_out_index, void* dev_out_n,
                                  void* stream);
/* ... and, after gfs_find_fundamental_ransac_device, kpstatus[index[j]] = 0 for every track j whose mask is 0. */
int gfs_klt_apply_mask_device(gfs_klt* h, int B, int pt_stride, const void* dev_m, const void* dev_index, const void* dev_mask,
                              void* dev_kpstatus, void* stream);

/* ============================================================================================
 * 10. cv::findFundamentalMat(points1, points2, cv::FM_RANSAC, threshold, confidence, mask) — the F check of the optical-flow
 *     matcher (src/ORBmatcher.cc:236, 2399-2405, 2463-2469) and of Tracking::EstimatePoseByOF (src/Tracking.cc:1973-1974):
 *     OpenCV's 7-point RANSAC (cv::RNG((uint64)-1), getSubset / checkSubset, symmetric epipolar distance against threshold^2,
 *     adaptive iteration budget, maxIters 1000) for n >= 15 points; for 8 .. 14 points the LMedS registrator the cv:: wrapper
 *     switches to (least median of the errors, inliers within 2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(median); `threshold` unused).
 *     Fewer than 8 points: GFS_ERR_UNSUPPORTED (the reference only calls it with more than 8).  The null space of the 7-point system and the roots
 *     of its cubic are computed with +, -, *, /, sqrt only (oracle/fmat_oracle.cpp): same inlier sets as the oracle
 *     bit for bit; against OpenCV the models agree to rounding noise, the order of the up to three models of a subset can differ.
 * ============================================================================================ */
typedef struct gfs_fmat gfs_fmat;
int gfs_fmat_create(int device, int max_points, int max_batch, gfs_fmat** out);
void gfs_fmat_destroy(gfs_fmat* h);
/* B independent problems (host pointers): pts1[b] / pts2[b] = n_points[b] x 2 floats (cv::Point2f), mask[b][i] = status[i] (0 / 1),
 * n_inliers[b] = size of the consensus set (0 = no model), F (may be NULL) = [B][9] row-major 3x3 of the best model.
 * threshold <= 0 -> 3, confidence outside (0, 1) -> 0.99, like the cv:: wrapper. */
int gfs_find_fundamental_ransac(gfs_fmat* h, int B, const int32_t* n_points, const float* const* pts1, const float* const* pts2,
                                double threshold, double confidence, int max_iters, uint8_t* const* mask, double* F,
                                int32_t* n_inliers);
/* Device-resident form: dev_pts1 / dev_pts2 [B][stride][2] float, dev_n [B] int32, dev_mask [B][stride] u8 (device); F and n_inliers
 * on the host.  Problems with 8 or fewer points are passed through (mask all ones, n_inliers = n): SearchByProjectionWithOF only runs
 * the check for more than 8 (src/ORBmatcher.cc:2397, 2461).  Synchronous (the acceptance rule is replayed on the host). */
int gfs_find_fundamental_ransac_device(gfs_fmat* h, int B, int stride, const void* dev_n, const void* dev_pts1, const void* dev_pts2,
                                       double threshold, double confidence, int max_iters, void* dev_mask, double* F,
                                       int32_t* n_inliers);

/* ============================================================================================
 * Timing helper for the harness: HIP events on a given stream (bench.py measures the dominant kernel
 * with these rather than torch events, which only see torch's current stream).
 * ============================================================================================ */
typedef struct gfs_timer gfs_timer;
int gfs_timer_create(int device, gfs_timer** out);
void gfs_timer_destroy(gfs_timer* t);
int gfs_timer_start(gfs_timer* t, void* stream);
int gfs_timer_stop(gfs_timer* t, void* stream);
/* blocks until the stop event completed; milliseconds between start and stop */
int gfs_timer_elapsed_ms(gfs_timer* t, float* ms);

/* Per-kernel accumulated device time (HIP events around each launch) — enable for profiling runs only. */
int gfs_profile_enable(int on);
/* Fills up to cap entries; returns number of distinct kernels. name buffers are 64 bytes each. */
int gfs_profile_report(char (*names)[64], double* total_ms, int64_t* launches, int cap);
int gfs_profile_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* GFS_ABI_H_ */
