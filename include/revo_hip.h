/*
 * revo_hip.h -- C ABI of the MI355X-native REVO hot path (librevo_hip.so).
 *
 * The reference (fabianschenk/REVO) has no FFI/plugin layer: its hot path is a
 * set of C++ classes (ImgPyramidRGBD, TrackerNew, Optimizer) linked into one
 * executable.  This header is the drop-in boundary for that method surface:
 * plain pointers and sizes, POD structs, no Eigen / cv / torch types.  Every
 * entry point cites the reference interface it replaces (paths relative to the
 * reference tree).  The header-only C++ adapters in revo_amd/cpp/ re-create the
 * reference class names on top of it.
 *
 * Conventions
 *   - return value 0 = REVO_OK, negative = error; revo_last_error() gives text
 *     (thread-local).  The reference has no error codes (log + exit(0) /
 *     assert / Sophus abort()); the adapters translate.
 *   - R is a 3x3 rotation in COLUMN-major order (Eigen::Matrix3f storage),
 *     T a 3-vector; together they map CURRENT-frame points into the
 *     KEYFRAME (tracker.cpp:286-288, optimizer.cpp:93).
 *   - 4x4 poses are column-major (Eigen::Matrix4f storage).
 *   - images are row-major with a byte stride (cv::Mat layout).
 *   - level 0 = full resolution (PYR_MAX_LVL), level PYR_MIN_LVL = coarsest.
 *   - every call does hipSetDevice(ctx device) itself; a pyramid may be created
 *     on one host thread and consumed on another (iowrapperRGBD.cpp:279 vs
 *     system.cpp:188), but one handle must not be used concurrently.
 */
#ifndef REVO_HIP_H
#define REVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REVO_MAX_LEVELS 6 /* optimizer.h:37 PYRAMID_LEVELS */

/* ---- error codes ------------------------------------------------------- */
enum {
  REVO_OK = 0,
  REVO_ERR_INVALID_ARG = -1,
  REVO_ERR_HIP = -2,          /* a HIP runtime call failed / no device        */
  REVO_ERR_NOT_KEYFRAME = -3, /* imgpyramidrgbd.h:113-116 "optimizationStructure not built" */
  REVO_ERR_NOT_ORTHOGONAL = -4, /* Sophus SO3(R) ENSURE, so3.hpp:419-424       */
  REVO_ERR_CAPACITY = -5,
  REVO_ERR_LEVEL = -6         /* assert(lvl < size) in imgpyramidrgbd.h:59-94  */
};

/* TrackerNew::TrackerStatus, tracker.h:61-66 */
enum {
  REVO_TRACKER_STATE_OK = 0,
  REVO_TRACKER_STATE_LOST = 1,
  REVO_TRACKER_STATE_NEW_KF = 2,
  REVO_TRACKER_STATE_UNKNOWN = 3
};

/* ---- settings (POD mirrors of the reference's settings classes) -------- */

/* ImgPyramidSettings, camerapyr.h:27-89 (+ Camera, camerapyr.h:90-111). */
typedef struct revo_pyr_settings {
  int32_t width, height;             /* camerapyr.h:49-52 (640x480)          */
  float fx, fy, cx, cy;              /* camerapyr.h:54-61 (level-0 K)        */
  int32_t pyr_min_lvl;               /* coarsest level, camerapyr.h:45 (2)   */
  int32_t pyr_max_lvl;               /* finest level,   camerapyr.h:46 (0); must be 0 */
  int32_t canny_threshold1;          /* camerapyr.h:40 (150)                 */
  int32_t canny_threshold2;          /* camerapyr.h:41 (100)                 */
  float depth_min, depth_max;        /* camerapyr.h:43-44 (0.1, 5.2)         */
  int32_t use_edge_hist;             /* camerapyr.h:62 (1)                   */
  float n_percentage;                /* camerapyr.h:63 (0.3)                 */
  /* distPatchSizes, imgpyramidrgbd.cpp:50: {20,10,5}.  The reference indexes
   * this 3-entry vector with the level, so levels >= 3 are undefined there;
   * here 0 means "no histogram / no fill-in at this level". */
  int32_t hist_patch[REVO_MAX_LEVELS];
} revo_pyr_settings;
/* Geometry accepted by revo_ctx_create (anything else is REVO_ERR_INVALID_ARG with a message): width a
 * multiple of 4*2^(levels-1) and <= 2048, height a multiple of 2^(levels-1) and <= 2048, at most 2048 tiles of 32 x 32
 * pixels (1920x1080 and 1280x1024 fit), at most REVO_MAX_LEVELS levels, every level a multiple of 16 pixels (640x480: up to
 * 5 levels, 1280x960: 6), hist_patch[l] dividing into <= 128 tiles per row.  The reference itself truncates any size
 * (camerapyr.h:98-103). */

/* OptimizerSettings, optimizer.h:42-112 (only the fields the hot path reads). */
typedef struct revo_opt_settings {
  float lambda_success_fac;                  /* optimizer.h:53 (0.5)  */
  float lambda_fail_fac;                     /* optimizer.h:54 (2.0)  */
  float lambda_initial[REVO_MAX_LEVELS];     /* optimizer.h:63 (0)    */
  float step_size_min[REVO_MAX_LEVELS];      /* optimizer.h:55 (1e-16)*/
  float convergence_eps[REVO_MAX_LEVELS];    /* optimizer.h:65 (0.999)*/
  int32_t max_its_per_lvl[REVO_MAX_LEVELS];  /* optimizer.h:56 (100)  */
  float edge_distance_lvl[REVO_MAX_LEVELS];  /* optimizer.h:59 {30,20,10,5,5,5} */
  float huber_edge;                          /* optimizer.h:75 (0.3)  */
  int32_t use_edge_filter;                   /* tracker.h:46 (1)      */
} revo_opt_settings;

/* TrackerSettings, tracker.h:31-55 (+ TrackerNew::histogramLevel, tracker.cpp:229). */
typedef struct revo_tracker_settings {
  int32_t check_tracking_results;   /* tracker.h:45 (1) */
  int32_t check_init_values;        /* tracker.h:43 (1) */
  int32_t n_frames_hist_voting;     /* tracker.h:47 (3) */
  int32_t histogram_level;          /* tracker.cpp:229 (2) */
} revo_tracker_settings;

/* Optimizer::ResidualInfo, optimizer.h:118-140. */
typedef struct revo_residual_info {
  int32_t good_pts_edges;
  int32_t bad_pts_edges;
  float sum_error_unweighted;
  float sum_error_weighted;
} revo_residual_info;

/* Values of config/dataset_tum1.yaml + config/revo_settings.yaml +
 * OptimizerSettings() defaults. */
void revo_pyr_settings_default(revo_pyr_settings* s);
void revo_opt_settings_default(revo_opt_settings* s);
void revo_tracker_settings_default(revo_tracker_settings* s);

/* ---- handles ------------------------------------------------------------ */
typedef struct revo_ctx revo_ctx;     /* CameraPyr + TrackerNew + Optimizer state */
typedef struct revo_pyr revo_pyr;     /* one ImgPyramidRGBD                       */
typedef struct revo_batch revo_batch; /* B independent frame-pairs, device resident */

const char* revo_last_error(void);
/* "x.y.z gfx950" */
const char* revo_version(void);

/* REVO::REVO -> new CameraPyr(settingsPyr) (camerapyr.h:117-164), new
 * TrackerNew(settingsTracker, settingsPyr) (tracker.cpp:225-235) which owns the
 * Optimizer (optimizer.cpp:44-61).  device = HIP device ordinal. */
int revo_ctx_create(int device, const revo_pyr_settings* pyr,
                    const revo_opt_settings* opt,
                    const revo_tracker_settings* trk, revo_ctx** out);
void revo_ctx_destroy(revo_ctx* ctx);
/* TrackerNew(const TrackerSettings&, const ImgPyramidSettings&) (tracker.cpp:225-235)
 * is constructed AFTER CameraPyr and the IO thread in the reference
 * (system.cpp:96,107): applies tracker/optimizer settings to an existing ctx. */
int revo_ctx_set_tracker(revo_ctx* ctx, const revo_opt_settings* opt,
                         const revo_tracker_settings* trk);

/* TrackerNew::histogramLevel (tracker.h:67, tracker.cpp:229). */
int revo_ctx_histogram_level(const revo_ctx* ctx);

/* Camera(fx,fy,cx,cy,w,h,scale) for level lvl, camerapyr.h:98-103,139-144:
 * out6 = {fx,fy,cx,cy,width,height}. */
int revo_ctx_camera(const revo_ctx* ctx, int lvl, float out6[6]);

/* ---- ImgPyramidRGBD ------------------------------------------------------ */

/* ImgPyramidRGBD(settings, camPyr, fullResRgb [BGR8], fullResDepth [f32 metres],
 * timestamp), imgpyramidrgbd.cpp:43-96.  Inputs are copied before returning
 * (the reference clones them, cpp:51,54), so the caller may reuse its buffers.
 * Strides in bytes. */
int revo_pyramid_create(revo_ctx* ctx, const uint8_t* bgr, size_t bgr_stride,
                        const float* depth_m, size_t depth_stride,
                        double timestamp, revo_pyr** out);
/* Same, fusing iowrapperRGBD.cpp:326-327: depth = u16 * (float)(1/scale). */
int revo_pyramid_create_u16(revo_ctx* ctx, const uint8_t* bgr, size_t bgr_stride,
                            const uint16_t* depth_raw, size_t depth_stride,
                            double depth_scale_factor, double timestamp,
                            revo_pyr** out);
/* ~ImgPyramidRGBD, imgpyramidrgbd.cpp:32-41 */
void revo_pyramid_destroy(revo_pyr* pyr);
/* ImgPyramidRGBD::makeKeyframe, imgpyramidrgbd.cpp:231-252: exact Euclidean
 * distance transform + (-dDT/dx, -dDT/dy, DT, 0) float4 table per level. */
int revo_pyramid_make_keyframe(revo_pyr* pyr);
int revo_pyramid_is_keyframe(const revo_pyr* pyr);
double revo_pyramid_timestamp(const revo_pyr* pyr); /* imgpyramidrgbd.h:97-100 */

/* Accessor planes (imgpyramidrgbd.h:45-117). */
typedef enum revo_plane {
  REVO_PLANE_GRAY = 0,       /* returnGray(lvl)            u8  W*H            */
  REVO_PLANE_DEPTH = 1,      /* returnDepth(lvl)           f32 W*H            */
  REVO_PLANE_EDGES = 2,      /* returnEdges(lvl)           u8  W*H {0,255}    */
  REVO_PLANE_EDGES_ORIG = 3, /* returnOrigEdges(lvl)       u8  W*H {0,255}    */
  REVO_PLANE_DT = 4,         /* returnDistTransform(lvl)   f32 W*H (keyframe) */
  REVO_PLANE_GRADTABLE = 5,  /* returnOptimizationStructure(lvl) f32 4*W*H    */
  REVO_PLANE_EDGES3D = 6,    /* return3DEdges(lvl)         f32 4*N col-major  */
  REVO_PLANE_HIST = 7,       /* histPyr[lvl]               u8  (H/P)*(W/P)    */
  REVO_PLANE_EDGES3D_TILED = 8 /* the same N points as EDGES3D in the order the tracker reads them (32x32-pixel tiles in raster
                                  order, row-major inside a tile): what the per-frame build writes; EDGES3D is derived on demand */
} revo_plane;

/* Lazy device->host read of one accessor plane.  cap_bytes = size of host_dst;
 * *count = number of ELEMENTS written (pixels; 3-D points for EDGES3D).
 * host_dst may be NULL to query *count only. */
int revo_pyramid_read(revo_pyr* pyr, revo_plane what, int lvl, void* host_dst,
                      size_t cap_bytes, size_t* count);

/* ImgPyramidRGBD::generateColoredPcl(lvl, clrPcl, densePcl) (imgpyramidrgbd.cpp:279-327), the
 * cloud REVO::start hands to the viewer / PLY export per keyframe (system.cpp:165,235):
 * 8 floats per point (X,Y,Z,1, r,g,b,1 with colours in [0,1]), x-outer / y-inner order;
 * dense != 0: every pixel with a usable depth, else edge pixels only.  dst8 == NULL only
 * counts.  Single-frame pyramids only (batch views keep no colour image). */
int revo_pyramid_colored_pcl(revo_pyr* p, int lvl, int dense, float* dst8, size_t cap_points, size_t* count);

/* ---- Optimizer ------------------------------------------------------------ */

/* float Optimizer::trackFrames(ref, curr, R, T, lvl, resInfo),
 * optimizer.cpp:235-311: the LM loop of ONE pyramid level, run on the device.
 * R,T in/out; *err = last accepted mean weighted residual. */
int revo_optimizer_track_level(revo_ctx* ctx, const revo_pyr* ref,
                               const revo_pyr* curr, float R_colmajor[9],
                               float T[3], int lvl, revo_residual_info* info,
                               float* err);

/* Optimizer::calcErrorAndBuffers + calculateWarpUpdate at a fixed pose
 * (optimizer.cpp:74-234), exposed for parity tests: A (6x6 row==col major,
 * symmetric), b (6) and error are the LGS6 members after finish()
 * (LGSX.h:320-326). */
int revo_optimizer_eval(revo_ctx* ctx, const revo_pyr* ref, const revo_pyr* curr,
                        const float R_colmajor[9], const float T[3], int lvl,
                        revo_residual_info* info, float* err, float A[36],
                        float b[6]);

/* Eigen::Matrix<float,6,1> inc = A.ldlt().solve(b) after A(i,i) *= 1 + LM_lambda (optimizer.cpp:258-262), as the
 * tracker kernel computes it (float LDL^T pivoted on the largest |diagonal|, pseudo-inverse of D), for n systems:
 * in = n x 43 floats {A[36] symmetric, b[6], lambda}, x6 = n x 6.  Exposed for the parity tests of the solver. */
int revo_optimizer_solve6(revo_ctx* ctx, int n, const float* A36_b6_lambda, float* x6);

/* ---- TrackerNew ------------------------------------------------------------ */

/* TrackerStatus TrackerNew::trackFrames(R, T, error, refFrame, currFrame),
 * tracker.cpp:294-353: init check (265-283, 357-393) + coarse-to-fine loop.
 * iters_per_lvl (may be NULL) receives the number of residual evaluations
 * (calls of calcErrorAndBuffers) per level -- the E_l of SURVEY 8(d). */
int revo_tracker_track_frames(revo_ctx* ctx, const revo_pyr* ref,
                              const revo_pyr* curr, float R_colmajor[9],
                              float T[3], float* err, int* status,
                              revo_residual_info* info,
                              int32_t iters_per_lvl[REVO_MAX_LEVELS]);

/* TrackerStatus TrackerNew::assessTrackingQuality(estimatedPose, currFrame),
 * tracker.cpp:118-201.  hist4/overlaps4 (may be NULL) receive the counts. */
int revo_tracker_assess_quality(revo_ctx* ctx, const float T_w_curr_colmajor[16],
                                const revo_pyr* curr, int* status,
                                int32_t hist4[4], int32_t overlaps4[4]);
/* void TrackerNew::addOldPclAndPose(pcl, worldPose, timeStamp),
 * tracker.cpp:209-223: pcl = src->return3DEdges(lvl) (kept on the device). */
int revo_tracker_add_old_pcl(revo_ctx* ctx, const revo_pyr* src, int lvl,
                             const float T_w_colmajor[16], double timestamp);
/* The same with the cloud in HOST memory, exactly the reference's signature
 * addOldPclAndPose(const Eigen::MatrixXf& pcl, ...): pcl = 4 x n column-major floats (X,Y,Z,1 per point), copied
 * (the reference copies the matrix, tracker.cpp:219). */
int revo_tracker_add_old_pcl_host(revo_ctx* ctx, const float* pcl_4xn_colmajor, size_t n,
                                  const float T_w_colmajor[16], double timestamp);
/* void TrackerNew::clearUpPastLists(), tracker.cpp:248-257 */
int revo_tracker_clear_past(revo_ctx* ctx);
int revo_tracker_past_size(const revo_ctx* ctx);

/* ---- batched independent frame-pairs (new) -------------------- */

/* One record per pair, 96 bytes. */
typedef struct revo_pair_result {
  float R[9];       /* column-major, curr -> ref */
  float T[3];
  float err;        /* last level's mean weighted residual */
  int32_t good, bad;
  int32_t status;   /* tracker.cpp:351-352 */
  int32_t evals[REVO_MAX_LEVELS]; /* residual evaluations per level */
  int32_t flags;    /* bit0: init pose reset to identity (tracker.cpp:277-282);
                       bit1: non-orthogonal input R (the single-pair calls return REVO_ERR_NOT_ORTHOGONAL);
                       bit2: evaluation cap hit (6000 residual evaluations; the reference's bound is 100 outer
                             iterations x unbounded retries) -- pose is the last accepted one;
                       bit3: the workgroups of this pair could not exchange their partial sums in time (not
                             co-resident, e.g. the device is shared): R, T are NOT valid (the single-pair
                             calls return REVO_ERR_HIP) */
  int32_t n_pts0;   /* N_0 of the current frame */
} revo_pair_result;

/* A batch owns device storage for 2*n_pairs pyramids (ref, curr). */
int revo_batch_create(revo_ctx* ctx, int n_pairs, revo_batch** out);
void revo_batch_destroy(revo_batch* b);
/* Device-resident inputs: d_bgr [2*n_pairs][H][W][3] u8, d_depth
 * [2*n_pairs][H][W] f32 metres; frame 2*i = reference (keyframe) of pair i,
 * frame 2*i+1 = current.  h_init_RT: n_pairs x 12 floats (R col-major, T) on
 * the HOST or NULL for identity.  d_results: n_pairs revo_pair_result records
 * in DEVICE memory.  stream: hipStream_t (NULL = the batch's own stream).
 * Enqueues: pyramid build of all frames, keyframe promotion of the refs,
 * TrackerNew::trackFrames of every pair.  Asynchronous. */
int revo_batch_track(revo_batch* b, const uint8_t* d_bgr, const float* d_depth,
                     const float* h_init_RT, revo_pair_result* d_results,
                     void* stream);
/* Stage-wise variants used by bench.py for per-kernel timing. */
int revo_batch_build(revo_batch* b, const uint8_t* d_bgr, const float* d_depth,
                     void* stream);
/* revo_batch_build without the copy of the depth input: level 0 of the depth pyramid IS d_depth (the reference's
 * level 0 is the input image as well, imgpyramidrgbd.cpp:62-64).  d_depth must stay valid and unchanged until the
 * batch is built again or destroyed -- accessors, the tracker's point lists and the keyframe promotion read it. */
int revo_batch_build_borrow(revo_batch* b, const uint8_t* d_bgr, const float* d_depth,
                            void* stream);
/* revo_batch_build for raw uint16 depth [2*n_pairs][H][W] (depth = raw * (float)(1/scale),
 * iowrapperRGBD.cpp:326-327, fused into the first build kernel). */
int revo_batch_build_u16(revo_batch* b, const uint8_t* d_bgr, const uint16_t* d_depth_raw,
                         double depth_scale_factor, void* stream);
int revo_batch_track_only(revo_batch* b, const float* h_init_RT,
                          revo_pair_result* d_results, void* stream);
/* Runs, on `stream`, whatever part of the last build was left to the batch's first consumer (default: the depth levels >= 1 of the
 * pyramid, imgpyramidrgbd.h:218-249, which only the edge lists read; the 3-D edge lists,
 * imgpyramidrgbd.cpp:199-226, and the keyframes' distance transforms, imgpyramidrgbd.cpp:231-252 -- the build stream is the
 * critical one of a pipelined caller).  revo_batch_track_only does this itself on ITS stream; call this first to run that work
 * on another stream (the library orders every later consumer and the next build of the batch behind it) or to keep it outside a
 * timed tracker launch.  On a stream other than the build's it waits for the build.  No-op when nothing is pending. */
int revo_batch_prepare(revo_batch* b, void* stream);
/* Waits for `stream` (NULL = the batch's own), runs whatever the last build still left pending, waits for the batch's last
 * tracker grid WHEREVER it ran, and decodes the flags of that grid's records: a record with bit 3 makes the call return
 * REVO_ERR_HIP.  Lifetime contract: the d_results buffer of the last revo_batch_track_only / revo_batch_track must stay
 * valid (not freed, not reused for something else) until this call or the batch's next tracker launch -- the call reads it.
 * A later revo_batch_build* does not end that obligation (a pipelined caller builds step k+1 before it syncs step k). */
int revo_batch_sync(revo_batch* b, void* stream);
/* Pyramid view of frame f of the batch (owned by the batch). */
int revo_batch_frame(revo_batch* b, int frame, revo_pyr** out);
/* Time one launch of the dominant (tracker) kernel with HIP events on its own
 * stream: returns the mean duration in ms over `reps` launches. */
int revo_batch_time_tracker(revo_batch* b, const float* h_init_RT,
                            revo_pair_result* d_results, void* stream, int reps,
                            float* ms_mean);

/* Measurement aid (bench.py's per-kernel roofline table): every kernel of revo_batch_build_borrow + revo_batch_prepare run
 * ALONE on the batch's own stream with HIP events between the launches; us[i] = mean duration of stage i over `reps` passes
 * (event to event: kernel + a few us of dispatch), name[i] = the kernel ("hysteresis" = k_hyst, or the banded kernels where a
 * level takes that path).  Leaves the batch built. */
#define REVO_MAX_STAGES 24
typedef struct revo_stage_times {
  int32_t n;
  float us[REVO_MAX_STAGES];
  char name[REVO_MAX_STAGES][32];
} revo_stage_times;
int revo_batch_profile_build(revo_batch* b, const uint8_t* d_bgr, const float* d_depth, int reps, revo_stage_times* out);

/* ---- host-buffer batches: what a producer like IOWrapperRGBD::readNextFrame hands over ------------ */

/* One frame-pair in HOST memory, as the reference's producer thread holds it after cv::imread
 * (iowrapperRGBD.cpp:301-333): BGR8 rows + depth rows (raw uint16 as on disk, or float32 metres after the
 * reference's convertTo), byte strides like cv::Mat::step.  R_init / T_init: initial pose curr -> ref
 * (tracker.cpp:286-288), used when use_init != 0 (else identity). */
typedef struct revo_pair_in {
  const uint8_t* ref_bgr;   size_t ref_bgr_stride;
  const void*    ref_depth; size_t ref_depth_stride;
  const uint8_t* cur_bgr;   size_t cur_bgr_stride;
  const void*    cur_depth; size_t cur_depth_stride;
  float R_init[9], T_init[3];
  int32_t use_init;
} revo_pair_in;
typedef revo_pair_result revo_pair_out;
typedef struct revo_pairs_job revo_pairs_job;

/* n independent frame-pairs from host buffers: upload (H2D on its own stream), both pyramids,
 * keyframe promotion of the reference frame and TrackerNew::trackFrames per pair, results back to the host.
 * depth_is_u16 != 0: the depth rows are raw uint16 and depth = raw * (float)(1 / depth_scale_factor) runs inside
 * the device build (iowrapperRGBD.cpp:326-327).
 *   revo_track_pairs_submit returns once the inputs have been consumed (the caller may reuse its buffers -- the
 *   reference's producer does, iowrapperRGBD.h:163), while the device work of this and earlier jobs continues:
 *   consecutive submits overlap job k+1's PCIe transfer with job k's kernels.  Host memory that is page-locked
 *   (hipHostMalloc / hipHostRegister / torch pin_memory) is read by DMA at PCIe speed; pageable memory works, slower.
 *   revo_track_pairs_wait blocks for the job's results (out: n records) and releases it.  A record with flag bit 3
 *   makes it return REVO_ERR_HIP.  At most 3 jobs may be in flight per context.
 *   revo_track_pairs = submit + wait. */
int revo_track_pairs_submit(revo_ctx* ctx, int n, const revo_pair_in* pairs, int depth_is_u16,
                            double depth_scale_factor, revo_pairs_job** job);
int revo_track_pairs_wait(revo_pairs_job* job, revo_pair_out* out);
int revo_track_pairs(revo_ctx* ctx, int n, const revo_pair_in* pairs, int depth_is_u16,
                     double depth_scale_factor, revo_pair_out* out);

/* ---- the pipelined batch mode as one handle (new in round 5) ------------------------------------------------
 *
 * The reference owns its producer / consumer pipeline: REVO::start drains the queue an IO thread fills
 * (system/system.cpp:96,128-284, io/iowrapperRGBD.cpp:279-288).  The batched mode's counterpart is a rotation of
 * `depth` device-resident batches over FOUR streams the handle owns -- build | edge lists + keyframe EDT | two
 * alternating tracker streams (the resident gate keeps two tracker grids in flight) -- so that step t+3 is built
 * while step t+2 is prepared and steps t+1 and t are tracked.  This is the shape `bench.py` measures; the handle
 * exists so that an integrator gets it without rebuilding the choreography (HIP multiplexes streams onto a few
 * hardware queues, and one stream more, or the same streams created in another order, costs a third of the
 * throughput: DESIGN.md 3.0).  revo_pipeline_create probes its streams (a 150 us kernel on one, a time stamp on
 * another) and replaces those that share a hardware queue; revo_pipeline_info reports the outcome.
 *
 *   submit(t)  enqueues build, deferred work and tracker grid of step t and returns at once: a ticket and the
 *              stream the grid runs on.  Work the caller enqueues on THAT stream before the next submit that uses
 *              the same tracker stream (the next-but-one with two tracker streams, i.e. depth >= 3; the next one
 *              otherwise) runs behind the grid and before the stream's next grid (the result collective, a copy):
 *              the "after the grid" slot.  Do not create a stream of your own for it: a fifth active stream ends up
 *              behind one of the four in a hardware queue.
 *   wait(t)    blocks until step t and its after-grid work are complete; with host_results the n records of the
 *              step are returned from pinned memory (a record with flag bit 3 makes it return REVO_ERR_HIP).
 *
 * d_bgr / d_depth: device-resident inputs of the step, [2*n_pairs][H][W][3] u8 and [2*n_pairs][H][W] depth, frame 2i =
 * reference (keyframe) of pair i, frame 2i+1 = current (revo_batch_track).  depth_kind 0: f32 metres, BORROWED
 * (level 0 of the depth pyramid is the caller's plane: it must stay valid and unchanged for `depth` further
 * submits or until revo_pipeline_wait of the step); 1: f32 metres, copied; 2: raw u16 with depth_scale_factor
 * (iowrapperRGBD.cpp:326-327).  input_ready_event: a hipEvent_t recorded behind whatever produces the inputs, or
 * NULL when they are already complete.  h_init_RT: n_pairs x 12 floats on the host (R column-major, T) or NULL.
 * d_results: n_pairs records in device memory, or NULL for a buffer of the handle's own (host_results).  Results
 * are the same bits as revo_batch_track on one batch: the pipeline only changes when kernels run.
 * One handle is driven by one host thread at a time. */
typedef struct revo_pipeline revo_pipeline;
typedef struct revo_pipeline_info_t {
  int32_t batches;            /* batches in rotation (= depth)                                            */
  int32_t pairs_per_step;
  int32_t tracker_streams;    /* 1 or 2                                                                    */
  int32_t distinct_hw_queues; /* how many of the handle's streams sit on pairwise distinct hardware queues
                                 (4 = none alias; -1 = not probed: REVO_PIPE_PROBE=0)                      */
  int32_t streams_replaced;   /* candidate streams discarded because they aliased one already kept         */
  int32_t probes_run;
  void* streams[4];           /* hipStream_t: tracker 0, tracker 1, build, auxiliary                       */
  uint64_t steps_submitted;
} revo_pipeline_info_t;
/* depth: batches in rotation, 0 = the default (4), 1 = one batch on one stream (nothing overlaps), 2 = build
 * next to one tracker stream.  host_results != 0: every step's records are copied to pinned host memory behind
 * its grid and revo_pipeline_wait returns them; then a slot's step must be waited for before the slot is
 * submitted again (`depth` steps later), else REVO_ERR_CAPACITY. */
int revo_pipeline_create(revo_ctx* ctx, int n_pairs, int depth, int host_results, revo_pipeline** out);
void revo_pipeline_destroy(revo_pipeline* p);
int revo_pipeline_submit(revo_pipeline* p, const uint8_t* d_bgr, const void* d_depth, int depth_kind,
                         double depth_scale_factor, const float* h_init_RT, revo_pair_result* d_results,
                         void* input_ready_event, uint64_t* ticket, void** after_grid_stream);
int revo_pipeline_wait(revo_pipeline* p, uint64_t ticket, revo_pair_result* h_results);
/* Waits for everything submitted so far (all four streams). */
int revo_pipeline_drain(revo_pipeline* p);
int revo_pipeline_info(const revo_pipeline* p, revo_pipeline_info_t* out);
/* The batch that holds step `ticket` (accessors through revo_batch_frame); valid until its slot is submitted again. */
int revo_pipeline_batch(revo_pipeline* p, uint64_t ticket, revo_batch** out);
/* Live timing of the dominant kernel inside the pipelined steps: every every_n-th submit carries a HIP event pair
 * around its tracker grid, on the grid's stream (0 = off; resets the statistics).  revo_pipeline_tracker_ms: mean
 * duration and number of the launches harvested so far (a launch is harvested when its slot is reused, waited
 * for, or drained). */
int revo_pipeline_time_tracker(revo_pipeline* p, int every_n);
int revo_pipeline_tracker_ms(revo_pipeline* p, float* mean_ms, int* launches);

/* ---- Multi-GPU: the result gather behind the C ABI --------------------------------------------
 * The reference is a single-process CPU program (main.cpp:22-47) and has no collective; the batched mode shards
 * independent frame-pairs over the GPUs of a node, ONE PROCESS PER GPU, and its only exchange is an all-gather of
 * the 96-byte pair records over RCCL / xGMI.  A communicator is created like an MPI-style NCCL program does it:
 * rank 0 calls revo_comm_unique_id and ships the 128 bytes to the other ranks by whatever the host has (MPI, a
 * file, a socket, torch.distributed's store); every rank then calls revo_comm_create (ncclCommInitRank on the
 * context's device: a collective).  RCCL is loaded at run time (dlopen: $REVO_RCCL_LIB, a copy the process
 * already carries -- PyTorch bundles one --, librccl.so.1); a box without RCCL still loads the library and only
 * these calls fail (REVO_ERR_HIP, revo_last_error says why). */
typedef struct revo_comm revo_comm;
#define REVO_COMM_ID_BYTES 128
/* REVO_OK iff an RCCL library is loadable; path (optional) = what was loaded, version = ncclGetVersion. */
int revo_comm_available(char* path_out, size_t path_cap, int* version_out);
int revo_comm_unique_id(uint8_t id[REVO_COMM_ID_BYTES]);
int revo_comm_create(revo_ctx* ctx, const uint8_t id[REVO_COMM_ID_BYTES], int world_size, int rank, revo_comm** out);
void revo_comm_destroy(revo_comm* c);
int revo_comm_world(const revo_comm* c, int* world_size, int* rank);
/* ncclAllGather of n_records records per rank, enqueued on `stream` (hipStream_t): d_recv holds
 * world_size * n_records records, rank-major.  Nothing is reduced; the records travel as bytes. */
int revo_comm_allgather_records(revo_comm* c, const revo_pair_result* d_send, revo_pair_result* d_recv,
                                int n_records, void* stream);
/* The pipeline enqueues the collective itself, in the after-grid slot: steps are grouped into windows of `every`
 * (1..8) consecutive steps; the grids of a window write their records into a send buffer of the handle's, and
 * behind the window's LAST grid one all-gather moves them to window slot (k % ring) of d_gathered -- device memory,
 * ring * world_size * every * n_pairs records, laid out [ring][rank][step in window][pair].  revo_pipeline_wait
 * (ticket of the window's last step) covers the collective; a slot is overwritten `ring` (2..16) windows later.
 * With a communicator attached revo_pipeline_submit's d_results may be NULL (non-NULL: the step's own records are
 * copied there as well).  comm = NULL detaches.  Attach / detach drain the pipeline.  Every rank must submit the
 * same number of steps.  revo_pipeline_flush_comm gathers an incomplete last window (*steps_valid of its `every`
 * steps carry records of this run; 0 = the last window was complete and nothing was enqueued) into *slot; it is a
 * collective too (all ranks, same point) and the next submit starts a new window. */
int revo_pipeline_set_comm(revo_pipeline* p, revo_comm* comm, int every, revo_pair_result* d_gathered, int ring);
int revo_pipeline_flush_comm(revo_pipeline* p, int* steps_valid, int* slot);

/* ---- REVO::start sequencing (system/system.cpp:84-305) ----------------------- */

/* The reference runs two threads: IOWrapperRGBD::generateImgPyramid builds pyramids into a
 * queue (iowrapperRGBD.cpp:257-300), REVO::start consumes the oldest one per loop body
 * (system.cpp:128-284).  revo_vo_submit is the producer side (asynchronous: the build of
 * frame N+1 overlaps the tracking of frame N on the device), revo_vo_track_next the consumer:
 * first frame -> keyframe; later frames: trackFrames against the keyframe,
 * assessTrackingQuality, optional promotion of the PREVIOUS frame to keyframe + re-track
 * (system.cpp:203-241), constant-velocity initialisation of the next frame (267-271). */
typedef struct revo_vo revo_vo;
int revo_vo_create(revo_ctx* ctx, revo_vo** out);
void revo_vo_destroy(revo_vo* vo);
int revo_vo_submit(revo_vo* vo, const uint8_t* bgr, size_t bgr_stride,
                   const float* depth_m, size_t depth_stride, double timestamp);
/* Same with raw uint16 depth (iowrapperRGBD.cpp:326-327 fused into the device build). */
int revo_vo_submit_u16(revo_vo* vo, const uint8_t* bgr, size_t bgr_stride,
                       const uint16_t* depth_raw, size_t depth_stride,
                       double depth_scale_factor, double timestamp);
/* pose_colmajor: 4x4 curr->world as REVO::writePose would emit it (system.cpp:275);
 * *new_keyframe: 1 if this frame created a keyframe.  REVO_ERR_INVALID_ARG if the queue is empty. */
int revo_vo_track_next(revo_vo* vo, float pose_colmajor[16], int* new_keyframe,
                       double* timestamp);
int revo_vo_queued(const revo_vo* vo);
/* Two-thread use (the reference's IO thread + consumer loop, system.cpp:96): revo_vo_set_max_queue(n > 0)
 * (re-)opens the stream and makes revo_vo_submit* block while n pyramids wait (n <= 0: never block); the producer ends the stream with revo_vo_close; the
 * consumer calls revo_vo_wait_frame (1: a frame is queued, 0: closed and drained) before revo_vo_track_next. */
int revo_vo_set_max_queue(revo_vo* vo, int max_queue);
int revo_vo_close(revo_vo* vo);
int revo_vo_wait_frame(revo_vo* vo);
int revo_vo_num_keyframes(const revo_vo* vo);
/* The current keyframe (kfPyr) and its pose in the world (getTransKFtoWorld), as REVO::start hands
 * them to the map drawer after a keyframe change (system.cpp:165-167,235-237).  The handle is
 * borrowed: valid until the next revo_vo_track_next / revo_vo_destroy. */
int revo_vo_keyframe(const revo_vo* v, revo_pyr** kf_out, float T_w_kf[16]);

#ifdef __cplusplus
}
#endif
#endif /* REVO_HIP_H */
