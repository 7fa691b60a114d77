/*
 * revo_oracle.h -- CPU ORACLE for the REVO hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm
 * (fabianschenk/REVO: datastructures/imgpyramidrgbd.*, system/optimizer.*,
 * system/tracker.*, utils/LGSX.h, the used slice of thirdparty/Sophus, and the
 * OpenCV-3 / Eigen-3.3 routines those call).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (revo_amd/, librevo_hip.so) never does.
 *
 * PARITY PINNING STATUS (see DESIGN.md "Oracle"):
 *   - SE3 exp / compose: pinned against thirdparty/Sophus/py (sympy) golden
 *     matrices generated in the authoring container (tests/golden/) and the
 *     reference's own Sophus test tangents (test/core/test_se3.cpp:30-43).
 *   - cvtColor / pyrDown / Canny / distanceTransform / Eigen LDLT: the reference
 *     holds no tests or fixtures for these and OpenCV/Eigen are absent here, so
 *     pixel-exact agreement with a real OpenCV-3 build is "parity unpinned";
 *     they are instead checked against independent known answers
 *     (scipy.ndimage exact EDT, brute force, integer numpy Sobel/Gauss,
 *     numpy.linalg.solve, numeric Jacobians).
 *   - the reference itself cannot be built in this image (needs OpenCV 3 and
 *     Eigen 3.3, neither present): there is no oracle/_ref.
 */
#ifndef REVO_ORACLE_H
#define REVO_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/revo_hip.h" /* POD settings structs only */

#ifdef __cplusplus
extern "C" {
#endif

/* 0 = faithful (float, sequential order of the reference); 1 = accumulate the
 * long sums in double (used to quantify rounding, never as "the reference"). */
void ro_set_accum_double(int on);
/* test hook: +-1 ulp of seeded noise on every entry of the finished normal equations (0 = off) */
void ro_set_ab_ulp_noise(unsigned seed);

/* ---- image primitives --------------------------------------------------- */
void ro_bgr2gray(const uint8_t* bgr, size_t stride, int w, int h, uint8_t* gray);
void ro_pyrdown_u8(const uint8_t* src, int w, int h, uint8_t* dst);
void ro_depth_subsample(const float* in, int w, int h, float* out);
void ro_sobel3(const uint8_t* gray, int w, int h, int16_t* dx, int16_t* dy);
void ro_canny(const uint8_t* gray, int w, int h, double thr1, double thr2,
              uint8_t* edges);
float ro_dist_histogram(const uint8_t* edges, int w, int h, int patch,
                        uint8_t* hist);
void ro_fill_in_edges(const uint8_t* hist_lvl, int hist_w,
                      const uint8_t* top_edges, int top_w, int top_h, int patch,
                      int patch_low, uint8_t* edges_mod, int mod_w);
int ro_edges3d(const uint8_t* edges, const float* depth, int w, int h, float fx,
               float fy, float cx, float cy, float dmin, float dmax,
               float* out4);
void ro_edt(const uint8_t* edges, int w, int h, float* dt);
void ro_grad_table(const float* dt, int w, int h, float* table4);
void ro_u16_to_depth(const uint16_t* raw, size_t stride, int w, int h,
                     double scale_factor, float* depth);

/* ---- small dense algebra (Eigen / Sophus restatements) ------------------- */
void ro_ldlt6_solve(const float A[36], const float b[6], float x[6]);
void ro_quat_from_R(const float R_cm[9], float q_wxyz[4]);
void ro_quat_to_R(const float q_wxyz[4], float R_cm[9]);
/* Sophus::SE3f::exp(a): out q (wxyz), t */
void ro_se3_exp(const float a[6], float q_wxyz[4], float t[3]);
/* SE3 * SE3 */
void ro_se3_mul(const float qa[4], const float ta[3], const float qb[4],
                const float tb[3], float qo[4], float to[3]);
int ro_is_orthogonal(const float R_cm[9]);
void ro_mat4_inverse(const float M_cm[16], float out_cm[16]);
void ro_mat4_mul(const float A_cm[16], const float B_cm[16], float out_cm[16]);

/* ---- ImgPyramidRGBD ------------------------------------------------------ */
typedef struct ro_pyramid ro_pyramid;
ro_pyramid* ro_pyramid_create(const revo_pyr_settings* s, const uint8_t* bgr,
                              size_t bgr_stride, const float* depth,
                              size_t depth_stride, double timestamp);
void ro_pyramid_destroy(ro_pyramid* p);
void ro_pyramid_make_keyframe(ro_pyramid* p);
int ro_pyramid_is_keyframe(const ro_pyramid* p);
/* same plane ids as revo_plane; returns element count (points for EDGES3D) */
size_t ro_pyramid_read(const ro_pyramid* p, int what, int lvl, void* dst,
                       size_t cap_bytes);
void ro_pyramid_camera(const ro_pyramid* p, int lvl, float out6[6]);
/* generateColoredPcl (imgpyramidrgbd.cpp:279-327): 8 floats per point; returns the count */
size_t ro_pyramid_colored_pcl(const ro_pyramid* p, int lvl, int dense, float* out8, size_t cap_points);

/* ---- Optimizer / TrackerNew ---------------------------------------------- */
typedef struct ro_tracker ro_tracker;
ro_tracker* ro_tracker_create(const revo_pyr_settings* ps,
                              const revo_opt_settings* os,
                              const revo_tracker_settings* ts);
void ro_tracker_destroy(ro_tracker* t);

/* calcErrorAndBuffers (+ calculateWarpUpdate when A/b != NULL) at a fixed pose */
float ro_optimizer_eval(ro_tracker* t, const ro_pyramid* ref,
                        const ro_pyramid* curr, const float R_cm[9],
                        const float T[3], int lvl, revo_residual_info* info,
                        float A[36], float b[6], float* ls_error);
/* Optimizer::trackFrames; returns last_residual; evals (may be NULL) counts
 * calcErrorAndBuffers calls.  Returns NAN and sets *aborted if the Sophus
 * orthogonality ENSURE would abort. */
float ro_optimizer_track_level(ro_tracker* t, const ro_pyramid* ref,
                               const ro_pyramid* curr, float R_cm[9], float T[3],
                               int lvl, revo_residual_info* info, int* evals,
                               int* aborted);
float ro_tracker_eval_cost(ro_tracker* t, const float R_cm[9], const float T[3],
                           int lvl, const ro_pyramid* curr,
                           const ro_pyramid* ref);
/* TrackerNew::trackFrames */
int ro_tracker_track_frames(ro_tracker* t, const ro_pyramid* ref,
                            const ro_pyramid* curr, float R_cm[9], float T[3],
                            float* err, revo_residual_info* info,
                            int32_t evals[REVO_MAX_LEVELS], int* flags);
int ro_tracker_assess_quality(ro_tracker* t, const float T_w_curr_cm[16],
                              const ro_pyramid* curr, int32_t hist4[4],
                              int32_t overlaps4[4]);
void ro_tracker_add_old_pcl(ro_tracker* t, const ro_pyramid* src, int lvl,
                            const float T_w_cm[16], double ts);
void ro_tracker_clear_past(ro_tracker* t);
int ro_tracker_past_size(const ro_tracker* t);

/* ---- REVO::start sequencing (system.cpp:84-305), one frame per call ------ */
typedef struct ro_vo ro_vo;
ro_vo* ro_vo_create(const revo_pyr_settings* ps, const revo_opt_settings* os,
                    const revo_tracker_settings* ts);
void ro_vo_destroy(ro_vo* v);
/* Feeds one frame; writes the absolute pose (4x4 col-major, curr->world) that
 * writePose() would emit.  Returns 1 if a new keyframe was created. */
int ro_vo_push(ro_vo* v, const uint8_t* bgr, size_t bgr_stride,
               const float* depth, size_t depth_stride, double ts,
               float pose_cm[16]);
int ro_vo_num_keyframes(const ro_vo* v);
/* Same sequencing but the pyramid build is skipped in the timing split:
 * cumulative seconds spent in (pyramid build, makeKeyframe, tracking+vote). */
void ro_vo_times(const ro_vo* v, double out3[3]);
/* REVO::start with the reference's IO thread (system.cpp:96): producer pinned to cpu_io, consumer (the caller) to
 * cpu_main (< 0: not pinned); returns the wall time of the stream. */
double ro_vo_run_pipelined(ro_vo* v, int n_frames, const uint8_t* bgr, const float* depth, const double* ts, int cpu_io,
                           int cpu_main, int queue_cap, float* poses_out);
/* n independent frame-pairs with the same two threads; seconds_out[passes] */
void ro_bench_pairs_pipelined(const revo_pyr_settings* ps, const revo_opt_settings* os, const revo_tracker_settings* ts,
                              const uint8_t* bgr, const float* depth, int n_pairs, int cpu_io, int cpu_main, int passes,
                              double* seconds_out);

/* bench.py's all-core CPU baseline: n_threads pthreads, one frame-pair per thread at a time (2 pyramids +
 * makeKeyframe + trackFrames from identity) for `seconds`; frames are packed [ref0,curr0,ref1,...].  Returns
 * the pairs completed. */
long ro_bench_pairs_mt(const revo_pyr_settings* ps, const revo_opt_settings* os, const revo_tracker_settings* ts,
                       const uint8_t* bgr, const float* depth, int n_pairs, int n_threads, double seconds,
                       double* elapsed_out);

#ifdef __cplusplus
}
#endif
#endif
