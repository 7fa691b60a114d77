// revo_dev.h -- shared host/device declarations of librevo_hip.so (gfx950 only).
//
// HBM layout.  A FrameSet holds B frames.  Every per-level plane is stored
// level-major with the frame as the slow index inside the level:
//     plane[l] + f * P_l      (P_l = w_l * h_l elements)
// so one launch covers all frames (blockIdx.z = frame) and, where levels are
// independent, all levels (blockIdx.x decodes level + tile).  A single
// ImgPyramidRGBD is a FrameSet with B = 1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/revo_hip.h"

#define REVO_L REVO_MAX_LEVELS

struct LevelGeom {
  int w, h;            // Camera(...,scale) width/height, camerapyr.h:98-103
  int npix;            // w*h
  float fx, fy, cx, cy;
  int patch;           // distPatchSizes[l] (0 = none), imgpyramidrgbd.cpp:50
  int hist_w, hist_h;  // w/patch, h/patch
  int nchunk;          // row-chunks per column for the ordered compaction
  int chunk_rows;
  // launch decode helpers (prefix sums over levels)
  int wpr;             // 32-pixel bitmap words per row (Canny candidate / strong / edge bitmaps)
  int nms_block_base;  // first block of this level in the k_canny_nms launch
  int has_orig;        // fillInEdges may change this level: edgesOrigPyr is a separate plane (imgpyramidrgbd.cpp:185-195)
  int pix_base;        // sum of npix of finer levels
  int edt_rows;        // rows one k_edt_rows workgroup handles (about EDT_ROW_PX pixels)
  int edt_block_base;  // first block of this level in the k_edt_rows launch
  int strip_base;      // number of 64-column strips of finer levels (compaction launch decode)
  int cc_base;         // sum of w*nchunk of finer levels
  int band_rows, nbands, band_base;  // banded hysteresis: rows per band (a multiple of the patch size and of 4), bands, bands of finer levels
};

struct PyrGeom {
  int frame0;  // first frame of this launch (blockIdx.z counts from it): lets a batch be split across streams
  int n_levels;
  int total_nms_blocks, total_pix, total_edt_blocks, total_strips, total_cc;
  int total_bands, any_banded;  // hysteresis bands of all levels; 1 if some level has more than one
  int hyst_force;      // -1: banded hysteresis where a level does not fit one workgroup (default); 1 / 0: always / never (REVO_HYST_BANDED)
  int hyst_heavy_runs; // MIXED hysteresis (levels that fit one workgroup): a level-0 frame with at least this many weak runs takes the
                       // banded path, several workgroups, inside the same launch (REVO_HYST_HEAVY_RUNS; 0 = off: one workgroup per frame)
  int total_tiles;     // 32 x 32-pixel tiles of all levels (wpr x nchunk per level): the tracker's tile-ordered edge list
  int pts_staged;      // per launch: 1 = the depths of the edge pixels of levels < n_levels - 1 were staged by the depth half of
                       // k_pyrdown (k_edge_prefix in front of it): k_pts_tiles reads them from FramePlanes::stage, not from the depth planes
  float depth_min, depth_max;
  int canny_low, canny_high;  // squared L2 thresholds (cv::Canny, L2gradient=true)
  int use_edge_hist;
  float n_percentage;
  double fill_thr[REVO_L];    // PATCH_SIZE_2*0.05, imgpyramidrgbd.cpp:133
  LevelGeom lv[REVO_L];
};

// Base pointers of one FrameSet (frame stride = lv[l].npix elements unless noted).
struct FramePlanes {
  uint8_t* gray[REVO_L];
  float* depth[REVO_L];
  uint2* cs[REVO_L];           // Canny bitmaps: per row and 32 pixels {candidate bits, strong bits} (frame stride h*wpr)
  uint8_t* edges[REVO_L];      // edgesPyr     {0,255}
  uint8_t* edges_orig[REVO_L]; // edgesOrigPyr {0,255}: written only for levels with has_orig (elsewhere it IS edges)
  int* scratch[REVO_L];        // CCL parent keys during the build; column g^2 during makeKeyframe
  float4* pts[REVO_L];         // edges3DPyr: (X,Y,Z,1), capacity npix per frame -- the reference's order, built on demand
  float4* pts_trk[REVO_L];     // the same points tile-ordered (hot path: what the tracker and the vote read)
  float* dt[REVO_L];           // dtPyr
  float4* table[REVO_L];       // optimizationStructure
  uint8_t* hist[REVO_L];       // histPyr (frame stride hist_w*hist_h)
  int* chunk[REVO_L];          // compaction counts -> offsets (frame stride w*nchunk)
  unsigned* cmask[REVO_L];     // per (column, 32-row chunk): bit y = edge pixel with valid depth
  uint8_t* vb[REVO_L];         // depth validity, one bit per pixel (byte = 8 pixels of a row), written by the pyrDown that
                               // reads the level anyway; the coarsest level has none (frame stride npix/8)
  int* npts;                   // [B][REVO_L]
  int* hist_nz;                // [B][REVO_L]
  int* strip_tot;              // [B][total_strips]: edge points per 64-column strip (the compaction's cross-strip offsets)
  uint32_t* ebits[REVO_L];     // banded hysteresis: the edge bitmap between its kernels (frame stride h*wpr)
  int* need_full;              // [B][REVO_L]: a band could not label its runs: k_hyst takes the whole (level, frame)
  int* hyst_heavy;             // [B]: mixed hysteresis: 1 = level 0 of the frame went through the bands (its band records are valid)
  int* tile_base;              // [B][total_tiles]: first list position of every 32 x 32 tile (exclusive scan per level)
  // staged edge depths (round 6; batches whose depth half of the pyramid runs behind Canny): the depth of every EDGE pixel of a
  // level, compact, tiles in raster order and row-major inside a tile (the order of the tile-ordered list, valid depth or not)
  float* stage[REVO_L];        // [B][npix] (capacity; ~10 % used)
  unsigned short* epre[REVO_L];// [B][h * wpr]: edge pixels of the rows above (inside the 32-row tile) in word column wc
  int* stage_base;             // [B][total_tiles]: first staging position of every tile (exclusive scan per level)
};

// One frame-pair for the tracker kernel.
struct PairDesc {
  const float4* pts[REVO_L];    // current frame's 3-D edge lists
  const int* npts;              // -> int[REVO_L] of the current frame
  const float* dt[REVO_L];      // keyframe's distance transform (gradients are formed on the fly)
  float R[9];                   // initial R (column-major), curr -> ref
  float T[3];
};

struct TrackParams {
  int pyr_min_lvl, pyr_max_lvl;   // coarse-to-fine range, tracker.cpp:324
  int lvl_begin, lvl_end;         // levels to run (lvl_begin >= lvl_end), inclusive
  int check_init;                 // tracker.cpp:268
  int eval_only;                  // 1: one calcErrorAndBuffers+calculateWarpUpdate at (R,T), lvl_begin
  int kspec[REVO_L];              // per level: LM candidates evaluated per pass, 1 full + (kspec-1) error-only retries (1..TRACK_KMAX)
  int redundant_n;                // levels with at most this many points are evaluated by every cluster member (no exchange)
  float lambda_success_fac, lambda_fail_fac;
  float lambda_initial[REVO_L], step_size_min[REVO_L], convergence_eps[REVO_L];
  int max_its[REVO_L];
  float edge_distance[REVO_L];
  float huber_edge;
  int use_edge_filter;
  struct { float fx, fy, cx, cy; int w, h; } cam[REVO_L];
};

// eval_only output (parity tests): LGS6 after finish() + ResidualInfo
struct EvalOut {
  float A[36];
  float b[6];
  float error;      // ls.error
  float mean_err;   // return value of calcErrorAndBuffers
  float sum_w, sum_u;
  int good, bad;
};

#define REVO_MAX_WIDTH 2048  // EDT row staged in LDS as int32
#define NMS_ROWS 6                  // output rows per k_canny_nms4 thread (4 pixels wide)
#define EDT_ROW_PX 1280             // pixels per k_edt_rows workgroup (whole rows)
#ifndef REVO_HYST_LDS_MAX
#define REVO_HYST_LDS_MAX 158720    // dynamic LDS of k_hyst: the level's edge bitmap (+ candidate bitmap when both fit)
#endif
#ifndef TRACK_THREADS
#define TRACK_THREADS 512
#endif
#ifndef TRACK_MAX_CLUSTER
#define TRACK_MAX_CLUSTER 32       // workgroups per frame-pair (one XCD holds 32 CUs)
#endif
#define TRACK_KMAX 4               // speculative LM candidates per pass
#define TRACK_NVAL 48              // values a workgroup publishes per pass: 32 normal-equation + 16 error slots

// ---- launchers (defined in the kernel translation units) -------------------
void launch_gray_depth(const PyrGeom& g, const FramePlanes& p, const uint8_t* d_bgr, const float* d_depth_f32,
                       const uint16_t* d_depth_u16, float u16_alpha, int B, hipStream_t s);
// parts: 3 = gray + depth in one launch, 1 = the gray half (Canny's input), 2 = the depth half (depth level + the source level's validity bits)
// stage_edges (parts = 2 only, launch_edge_prefix in front): the depth half also stages the depths of the source level's edge pixels
void launch_pyrdown(const PyrGeom& g, const FramePlanes& p, int lvl, int B, hipStream_t s, int parts = 3, bool stage_edges = false);
// per tile and row of levels < n_levels - 1: how many EDGE pixels lie in front (staging positions of launch_pyrdown(..., true))
void launch_edge_prefix(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s);
void launch_canny_nms(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s);
void launch_hyst(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s);
void launch_fill(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s);
void launch_compact(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s);      // reference-ordered list (accessor)
// tile-ordered list + npts (hot path); which: bit 0 = k_tile_count, bit 1 = k_pts_tiles (the stage profiler times them apart)
void launch_tile_points(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s, int which = 3);
// keyframe promotion of frames f0, f0+fstride, ... (count frames)
void launch_pyrdown_bgr(const uint8_t* src, int w, int h, uint8_t* dst, hipStream_t s);
void launch_colored_pcl(const PyrGeom& g, const FramePlanes& p, int frame, int lvl, int dense, const uint8_t* bgr_lvl,
                        int* chunk, unsigned* cmask, int* total, int cap, float* out8, hipStream_t s);
void launch_keyframe(const PyrGeom& g, const FramePlanes& p, int f0, int fstride, int count, hipStream_t s, int which = 3);  // bit 0 = k_edt_cols, bit 1 = k_edt_rows
// epoch_io: per-mailbox epoch counter kept by the owner of d_mail (zero it together with the mailbox)
// both tracker launchers return the grid size (workgroups) they enqueued; d_resident: the device's census counter (every
// workgroup adds 1 when it starts), nullptr = no census
int launch_track(const PairDesc* d_descs, const TrackParams& prm, revo_pair_result* d_out, EvalOut* d_eval,
                 int n_pairs, unsigned long long* d_mail, unsigned* epoch_io, int cluster, unsigned* d_resident, hipStream_t s);
// waits on the device until the census counter d_resident[0] has reached `want`; a gate that gives up counts itself in d_resident[1]
void launch_track_gate(unsigned* d_resident, unsigned want, hipStream_t s);
// seq_ptr (pinned host memory, may be null): receives seq_val after the result record has been written
int launch_track_one(const PairDesc& desc, const TrackParams& prm, revo_pair_result* out, EvalOut* eval_out,
                     unsigned long long* d_mail, unsigned* epoch_io, int cluster, unsigned* seq_ptr, unsigned seq_val,
                     unsigned* d_resident, hipStream_t s);
void launch_solve6(const float* d_Ab /*n x 43: A 36, b 6, lambda*/, int n, float* d_x /*n x 6*/, hipStream_t s);
int track_blocks_per_cu();  // occupancy query of k_track (advisory)
void launch_grad_table(const PyrGeom& g, const FramePlanes& p, int f0, int fstride, int count, hipStream_t s);
// past clouds of the quality vote (tracker.cpp:138-176): pose of cloud c relative to the current frame
// (R column-major 9 + T 3), its points and its on-device count
struct VoteArgs { float RT[3][12]; const float4* pts[3]; const int* n[3]; };
void launch_vote(const PyrGeom& g, const FramePlanes& curr, int curr_frame, int lvl, int n_clouds, const VoteArgs& va,
                 int* d_marks /*npix, all-zero in/out*/, int* d_hist8 /*all-zero in/out*/, unsigned* d_done /*zero in/out*/,
                 int* h_out8 /*pinned host: hist[4], overlaps[4], then the sequence word*/, unsigned seq_val, int use_orig_edges,
                 hipStream_t s);
void launch_copy_cloud(float4* dst, const float4* src, int* dst_n, const int* src_n, hipStream_t s);
