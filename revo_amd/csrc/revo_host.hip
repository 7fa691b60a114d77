// revo_host.hip -- host side of librevo_hip.so: the C ABI of include/revo_hip.h.
//
// Owns HBM (one blob per FrameSet, pooled), the per-context HIP stream, and the
// sequencing of the kernels in revo_pyramid.hip / revo_track.hip.  No torch, no
// Eigen, no OpenCV types cross this boundary.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <atomic>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "revo_dev.h"
#include "revo_mat4.h"

// ------------------------------------------------------------------ errors --
static thread_local std::string g_err;
extern "C" const char* revo_last_error(void) { return g_err.c_str(); }
extern "C" const char* revo_version(void) { return "0.1.0 gfx950"; }

static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
// used by the other translation units of the library (revo_pipeline.hip): one error string per thread
extern "C" void revo_set_error_(const char* msg) { g_err = msg ? msg : ""; }
#define HIPCHECK(expr)                                                                      \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess)                                                                  \
      return fail(REVO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
  } while (0)

// ---------------------------------------------------------------- defaults --
extern "C" void revo_pyr_settings_default(revo_pyr_settings* s) {  // config/dataset_tum1.yaml
  memset(s, 0, sizeof(*s));
  s->width = 640; s->height = 480;
  s->fx = 517.306408f; s->fy = 516.469215f; s->cx = 318.643040f; s->cy = 255.313989f;
  s->pyr_min_lvl = 2; s->pyr_max_lvl = 0;
  s->canny_threshold1 = 150; s->canny_threshold2 = 100;
  s->depth_min = 0.1f; s->depth_max = 5.2f;
  s->use_edge_hist = 1; s->n_percentage = 0.3f;
  s->hist_patch[0] = 20; s->hist_patch[1] = 10; s->hist_patch[2] = 5;  // imgpyramidrgbd.cpp:50
}
extern "C" void revo_opt_settings_default(revo_opt_settings* s) {  // optimizer.h:46-85
  memset(s, 0, sizeof(*s));
  s->lambda_success_fac = 0.5f; s->lambda_fail_fac = 2.0f;
  const float ed[6] = {30, 20, 10, 5, 5, 5};
  for (int i = 0; i < REVO_L; ++i) {
    s->lambda_initial[i] = 0.f; s->step_size_min[i] = 1e-16f; s->convergence_eps[i] = 0.999f;
    s->max_its_per_lvl[i] = 100; s->edge_distance_lvl[i] = ed[i];
  }
  s->huber_edge = 0.3f;
  s->use_edge_filter = 1;  // config/revo_settings.yaml:11
}
extern "C" void revo_tracker_settings_default(revo_tracker_settings* s) {  // config/revo_settings.yaml:9-12
  s->check_tracking_results = 1; s->check_init_values = 1; s->n_frames_hist_voting = 3;
  s->histogram_level = 2;  // tracker.cpp:229
}

// ----------------------------------------------------------------- objects --
struct FrameSet {
  int B = 0;
  void* blob = nullptr;
  size_t blob_bytes = 0;
  FramePlanes p{};
  float* own_depth0 = nullptr;  // the set's own level-0 depth plane (p.depth[0] may point at a borrowed input instead)
  uint8_t* d_bgr = nullptr;   // input staging (single-frame API)
  float* d_depth = nullptr;   // input staging; aliased as u16 for the u16 entry point
  // asynchronous creation (single-frame API): pinned host staging, "built" and "released" events
  uint8_t* h_bgr = nullptr;
  float* h_depth = nullptr;
  hipEvent_t ev_ready = nullptr;  // recorded on the build stream after the last build kernel
  hipStream_t ready_stream = nullptr;  // ... that stream (a consumer on it is ordered already)
  hipEvent_t ev_aux = nullptr;    // (batches) the keyframe EDT on the batch's side stream: not owned by the set
  bool has_aux = false;
  // (batches) the keyframes' EDT has been DEFERRED to whoever needs it first -- normally the tracker launch of the batch, on
  // the tracker's stream: the build stream is the critical one of the pipelined step and the tracker streams have slack
  std::mutex edt_mu;
  std::atomic<bool> edt_pending{false};  // set by a batch build, cleared under edt_mu by whoever runs the EDT
  int edt_count = 0;              // keyframes: frames 0, 2, 4, ...
  bool pts_pending = false;       // (REVO_DEFER >= 2) the tile-ordered edge lists of ALL frames were left to the same consumer too
  bool depth_pending = false;     // ... and the build ran only the gray half of pyrDown: the depth half goes in front of the lists
  bool hyst_pending = false;      // (REVO_DEFER = 3) ... and hysteresis + fill-in: the build stopped behind the Canny NMS
  bool fill_pending = false;      // (REVO_DEFER = 2, REVO_DEFER_FILL) ... fill-in alone: the build stopped behind the hysteresis
  // whoever ran the deferred EDT recorded this on ITS stream: consumers (and the next build into these planes) on any other
  // stream order themselves behind it (`edt_pending == false` alone says "enqueued somewhere", not "visible here")
  hipEvent_t ev_edt = nullptr;
  hipStream_t edt_stream = nullptr;
  bool has_edt = false;
  hipEvent_t ev_free = nullptr;   // recorded on the tracker stream when the set goes back to the pool
  hipEvent_t ev_free2 = nullptr;  // ... and on the vote stream (the quality vote and the cloud copy read the set there)
  hipEvent_t ev_h2d = nullptr;    // (single-frame API) the copy out of page-locked caller rows has finished
  bool has_ready = false, has_free = false, has_free2 = false;
};

struct Past {  // one entry of mPastPcl / mPastWorldPoses / mPastTimeStamps (tracker.h:92-95)
  float4* d_pts; int* d_n; int n; float T_w[16]; double ts;
  size_t cap;  // points the buffer holds
};

// Experiment knobs (environment variables), read ONCE PER CONTEXT when it is created (they used to be
// process-wide statics, so two contexts of one process could not differ): the defaults are what ships.
struct Knobs {
  int track_depth;     // REVO_TRACK_DEPTH (1..4, default 2; REVO_TRACK_SERIAL=1 forces 1): tracker grids the resident gate keeps in flight
  int cluster_one;     // REVO_TRACK_CLUSTER_ONE (0 = automatic): workgroups of the single-pair launch
  int redundant_one;   // REVO_TRACK_REDUNDANT_ONE: levels up to this many points are evaluated redundantly by the single-pair launch
  int h2d_mode;        // REVO_H2D_STREAMS: 0 = colour and depth planes on two copy streams, 1 = one stream, 2 = swapped
  int direct_h2d;      // REVO_DIRECT_H2D (default 1): revo_pyramid_create reads page-locked caller rows in place (no staging copy)
  int h2d_max_run_mb;  // REVO_H2D_MAX_RUN_MB: host-buffer batches merge adjacent frames into copies of at most this many MB
  bool kspec_one_set;  // REVO_TRACK_KSPEC_ONE (or no REVO_TRACK_KSPEC at all): the single-pair launches' own speculation depths
  int kspec_one[REVO_L];
  int upload_blocks;   // REVO_UPLOAD_BLOCKS: workgroups of the upload kernel for one contiguous plane
  int h2d_kernel;      // REVO_H2D_KERNEL (default 0): 1 = the in-place frame upload is a copy KERNEL reading the page-locked rows, not hipMemcpyAsync (2: on the build stream)
  int h2d_wait_poll;   // REVO_H2D_WAIT_POLL (default 1): the in-place frame upload is waited for by polling the event, not by hipEventSynchronize
  int stage_edge_depths;  // REVO_STAGE_EDGE_DEPTHS (default 0): pipelined batches stage the edge pixels' depths in the depth pass
};

struct revo_ctx {
  // handles (pyramids, batches, VO drivers) keep their context alive: revo_ctx_destroy only drops
  // the owner's reference, the last handle to go frees the device state
  std::atomic<int> refs{1};
  int device;
  Knobs knobs;
  revo_pyr_settings ps; revo_opt_settings os; revo_tracker_settings ts;
  PyrGeom geom;
  TrackParams tp;
  hipStream_t stream;        // tracker / consumer stream
  hipStream_t build_stream;  // pyramid builds of the single-frame API (overlap with tracking, like the IO thread)
  std::mutex mu;
  std::vector<FrameSet*> pool;  // free single-frame FrameSets
  int framesets_created = 0;    // single-frame FrameSets that exist (free or in use)
  std::vector<Past> past_pool;  // recycled past-cloud buffers (no hipMalloc per frame)
  // single-pair tracker scratch
  // the kernel reads the descriptor from its argument segment and writes the result straight into
  // pinned host memory: one launch + one sync per trackFrames, no copies on the stream
  PairDesc* h_desc;
  revo_pair_result* h_res;      // [3]: slots 0/1 = the VO driver's look-ahead launches, slot 2 = the public single-pair calls
  EvalOut* h_eval;
  unsigned* h_seq;              // [4] pinned: sequence words the kernels write after their results (slots 0/1: the VO driver's look-ahead launches, 2: the public single-pair calls)
  unsigned seq_next = 1;
  unsigned long long* d_mail;   // cluster mailbox of the single-pair path
  unsigned mail_epoch = 0;      // next free epoch window of d_mail (launch_track_one)
  int num_cus, blocks_per_cu;
  // vote
  std::deque<Past> past;
  int* d_marks; int* d_hist8; int* h_hist8; unsigned* d_vote_done;
  // host-buffer batches (revo_track_pairs_*): up to 3 jobs in flight, slots recycled per (n, depth type)
  std::vector<struct revo_pairs_job*> jobs;
  hipStream_t copy_stream = nullptr, copy_stream2 = nullptr, pair_streams[2] = {nullptr, nullptr};
  // The quality vote and the past-cloud copies run on a stream of their own (REVO_VOTE_STREAM=0: on the tracker stream, as up
  // to round 5): nothing the NEXT frame's tracker reads comes out of them, so in the sequential loop tracker N+1 follows
  // tracker N directly instead of queueing behind vote N and copy N-1 (profiles/r06_single_stream_timeline.txt: ~35 us of 240).
  hipStream_t vote_stream = nullptr;
  hipEvent_t ev_vote = nullptr;
  hipStream_t frame_copy_stream = nullptr;  // single-frame API: H2D straight out of page-locked caller rows (pyramid_create_common)
  unsigned long long jobs_submitted = 0;
  // coloured point cloud (generateColoredPcl), allocated on first use
  char* d_pcl = nullptr; float* d_pcl_out; uint8_t* d_pcl_clr[2]; int* d_pcl_chunk; unsigned* d_pcl_mask; int* d_pcl_total;
};

struct revo_pyr {
  revo_ctx* ctx;
  FrameSet* fs;
  int frame;
  bool owns_fs;
  bool is_kf;
  double ts;
  bool table_built;
  bool ref_list_built;  // edges3DPyr in the reference's order (the hot path only writes the tile-ordered list)
  bool dt_ready = false;  // the distance transforms are already enqueued on the tracker stream (revo_pyramid_prepare_keyframe_)
};

struct revo_batch {
  revo_ctx* ctx;
  int n_pairs;
  FrameSet* fs;
  std::vector<revo_pyr> views;
  PairDesc* h_descs; PairDesc* d_descs;
  unsigned long long* d_mail;
  unsigned mail_epoch = 0;
  bool identity_uploaded = false;  // d_descs already holds the identity initial poses (nothing to upload)
  int cluster;
  int defer = 2;                   // REVO_DEFER / REVO_EDT_DEFER, read when the batch is created (env_defer_level)
  int split_depth = 1;             // REVO_SPLIT_DEPTH (default 1): with the lists deferred, the depth half of the pyramid is deferred with them
  int defer_fill = 0;              // REVO_DEFER_FILL: with the lists deferred, fill-in is deferred with them
  hipStream_t stream;
  hipStream_t side = nullptr;                      // the EDT of the keyframes runs here, next to the edge lists
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev0, ev1, ev_upload;
  // the batch's last tracker grid: a later launch of the SAME batch on another stream shares its mailbox and descriptors,
  // and the next build rewrites what it reads -- both order themselves behind this event
  hipEvent_t ev_trk = nullptr;
  hipStream_t trk_stream = nullptr;
  bool has_trk = false;
  hipEvent_t tev0 = nullptr, tev1 = nullptr;       // (revo_batch_time_next_grid_) recorded directly around the NEXT tracker grid, then cleared
  const revo_pair_result* last_results = nullptr;  // device records of the last track launch (revo_batch_sync decodes their flags)
  revo_pair_result* h_flags = nullptr;             // pinned scratch for that
};

// experiment knobs (environment): clamped integers, defaults are what ships
static int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  const int v = atoi(e);
  return v < lo ? lo : (v > hi ? hi : v);
}

// ---------------------------------------------------------------- geometry --
static int build_geom(const revo_pyr_settings& s, PyrGeom* g, std::string* why) {
  memset(g, 0, sizeof(*g));
  const int L = s.pyr_min_lvl - s.pyr_max_lvl + 1;  // camerapyr.h:68-71
  if (s.pyr_max_lvl != 0) { *why = "pyr_max_lvl must be 0 (the reference indexes per-level vectors by level)"; return -1; }
  if (L < 1 || L > REVO_L) { *why = "1..6 pyramid levels supported"; return -1; }
  // the reference truncates whatever it is given (camerapyr.h:98-103); here: up to 2048 x 2048 (the EDT's LDS rows, the
  // 64 x 32-row chunks of the column walks) with at most 2048 tiles of 32 x 32 pixels per level (k_tile_count)
  if (s.width <= 0 || s.height <= 0 || s.width > REVO_MAX_WIDTH || s.height > 2048 ||
      ((s.width + 31) / 32) * ((s.height + 31) / 32) > 2048) {
    *why = "image size must be within 2048 x 2048 and 2048 tiles of 32 x 32 pixels (1920 x 1080 fits, 2048 x 1536 does not)"; return -1;
  }
  if (s.width % (4 << (L - 1)) || s.height % (1 << (L - 1))) {
    *why = "width must be a multiple of 4*2^(levels-1) and height of 2^(levels-1)"; return -1;
  }
  g->n_levels = L;
  { const char* e = getenv("REVO_HYST_BANDED"); g->hyst_force = (e && *e) ? (*e != '0' ? 1 : 0) : -1; }
  // mixed hysteresis (round 6): where a level fits one workgroup, the frames whose level 0 has at least this many weak runs (the
  // launch lasts as long as the heaviest of them) are cut into bands inside the same launch.  Bit-exact and OFF by default (0):
  // the band / seam / output passes of the heavy frames cost more than the single workgroup they replace
  g->hyst_heavy_runs = env_int("REVO_HYST_HEAVY_RUNS", 0, 0, 1 << 30);  // measured and not kept as the default: profiles/r06_ab_mixed_hyst.txt
  g->depth_min = s.depth_min; g->depth_max = s.depth_max;
  // cv::Canny with L2gradient: low/high swapped if needed, squared (imgpyramidrgbd.cpp:184)
  double lo = s.canny_threshold1, hi = s.canny_threshold2;
  if (lo > hi) std::swap(lo, hi);
  lo = std::min(32767.0, lo); hi = std::min(32767.0, hi);
  if (lo > 0) lo *= lo;
  if (hi > 0) hi *= hi;
  g->canny_low = (int)std::floor(lo); g->canny_high = (int)std::floor(hi);
  g->use_edge_hist = s.use_edge_hist; g->n_percentage = s.n_percentage;
  int tile = 0, pix = 0, row = 0, col = 0, cc = 0, tiles32 = 0, bands = 0;
  bool bands_fit = true;
  for (int l = 0; l < L; ++l) {
    LevelGeom& v = g->lv[l];
    const float scale = 1.0f / (float)std::pow(2, l);  // camerapyr.h:142
    if (l == 0) { v.fx = s.fx; v.fy = s.fy; v.cx = s.cx; v.cy = s.cy; v.w = s.width; v.h = s.height; }
    else {
      v.fx = s.fx * scale; v.fy = s.fy * scale; v.cx = s.cx * scale; v.cy = s.cy * scale;
      v.w = (int)((float)s.width * scale); v.h = (int)((float)s.height * scale);
    }
    v.npix = v.w * v.h;
    if (v.npix % 16) { *why = "every level must have a multiple of 16 pixels"; return -1; }
    v.patch = s.hist_patch[l] > 0 ? s.hist_patch[l] : 0;
    if (v.patch > 0) {
      v.hist_w = v.w / v.patch; v.hist_h = v.h / v.patch;
      if (v.hist_w < 1 || v.hist_h < 1 || v.hist_w > 128) { *why = "hist_patch out of range for this level size"; return -1; }
    }
    g->fill_thr[l] = (double)(v.patch * v.patch) * 0.05;  // imgpyramidrgbd.cpp:133
    v.chunk_rows = 32; v.nchunk = (v.h + 31) / 32;
    v.wpr = (v.w + 31) / 32;
    v.nms_block_base = tile; tile += (8 * v.wpr * ((v.h + NMS_ROWS - 1) / NMS_ROWS) + 255) / 256;
    // (a level whose edge bitmap does not fit one workgroup's LDS takes the banded hysteresis; its last resort keeps the bitmap
    // in the scratch plane: launch_hyst)
    v.pix_base = pix; pix += v.npix;
    v.edt_rows = std::max(1, EDT_ROW_PX / v.w);
    v.edt_block_base = row; row += (v.h + v.edt_rows - 1) / v.edt_rows;
    v.strip_base = col; col += (v.w + 63) / 64;
    v.cc_base = cc; cc += v.w * v.nchunk;
    tiles32 += v.wpr * v.nchunk;
    // banded hysteresis: bands of about 2400 bitmap words, heights a multiple of the histogram patch and of 4 rows
    {
      int unit = v.patch > 0 ? v.patch : 4;
      while (unit % 4) unit *= 2;
      int rows = (env_int("REVO_HYST_BAND_WORDS", 2400, 64, 1 << 20) / v.wpr) / unit * unit;
      if (rows < unit) rows = unit;
      if (rows >= v.h) rows = v.h;
      if ((v.h + rows - 1) / rows > 32) rows = ((v.h + 31) / 32 + unit - 1) / unit * unit;  // k_hyst_seam holds at most 32 bands' flags
      v.band_rows = rows;
      v.nbands = (v.h + rows - 1) / rows;
      v.band_base = bands; bands += v.nbands;
      if (v.nbands > 1) g->any_banded = 1;
      // a band's bitmaps (weak words, 16-bit id bases, edge rows + halo) must leave room for tables in the 64 KB of a band workgroup
      const size_t nwb = (size_t)rows * v.wpr;
      if (nwb + (nwb + 2) / 2 + (size_t)(rows + 2) * v.wpr + 2 + 256 > 16384) bands_fit = false;
    }
  }
  g->total_tiles = tiles32;
  g->total_bands = bands_fit ? bands : 0;  // 0: no banded hysteresis for this geometry (k_hyst alone, as in round 2)
  for (int l = 1; l < L; ++l)  // fillInEdges' gate (imgpyramidrgbd.cpp:188-195 + the patch sizes that exist)
    g->lv[l].has_orig = (s.use_edge_hist && g->lv[l].patch > 0 && g->lv[l - 1].patch > 0) ? 1 : 0;
  g->total_nms_blocks = tile; g->total_pix = pix; g->total_edt_blocks = row; g->total_strips = col; g->total_cc = cc;
  return 0;
}

static void build_track_params(const revo_ctx* c, TrackParams* t) {
  memset(t, 0, sizeof(*t));
  t->pyr_min_lvl = c->ps.pyr_min_lvl; t->pyr_max_lvl = c->ps.pyr_max_lvl;
  t->lvl_begin = c->ps.pyr_min_lvl; t->lvl_end = c->ps.pyr_max_lvl;
  t->check_init = c->ts.check_init_values;
  t->lambda_success_fac = c->os.lambda_success_fac; t->lambda_fail_fac = c->os.lambda_fail_fac;
  for (int i = 0; i < REVO_L; ++i) {
    t->lambda_initial[i] = c->os.lambda_initial[i]; t->step_size_min[i] = c->os.step_size_min[i];
    t->convergence_eps[i] = c->os.convergence_eps[i]; t->max_its[i] = c->os.max_its_per_lvl[i];
    t->edge_distance[i] = c->os.edge_distance_lvl[i];
  }
  t->huber_edge = c->os.huber_edge; t->use_edge_filter = c->os.use_edge_filter;
  // speculation depth per level: REVO_TRACK_KSPEC = one digit for all levels or one digit per level (finest first)
  {
    const char* e = getenv("REVO_TRACK_KSPEC");
    const size_t n = e ? strlen(e) : 0;
    for (int i = 0; i < REVO_L; ++i) {
      // default: the two finest levels evaluate one retry next to the candidate (their points are what a pass costs),
      // the coarse ones three (profiles/r02_single_stream_sweep.txt)
      int k = i < 2 ? 2 : TRACK_KMAX;
      if (n == 1) k = e[0] - '0'; else if (n > 1) k = e[std::min<size_t>(i, n - 1)] - '0';
      t->kspec[i] = std::max(1, std::min(TRACK_KMAX, k));
    }
  }
  t->redundant_n = env_int("REVO_TRACK_REDUNDANT_BATCH", 400, 0, 1 << 30);
  for (int l = 0; l < c->geom.n_levels; ++l) {
    const LevelGeom& v = c->geom.lv[l];
    t->cam[l].fx = v.fx; t->cam[l].fy = v.fy; t->cam[l].cx = v.cx; t->cam[l].cy = v.cy; t->cam[l].w = v.w; t->cam[l].h = v.h;
  }
}

// Workgroups per frame-pair.  Every member of every cluster must be resident at once (they exchange partial sums inside
// the launch).  A batch grid takes HALF of what the device can hold for this kernel: the resident gate keeps the tracker
// grids of two consecutive batches in flight (the second fills the CUs the first one's finished pairs free), and two
// half-chip grids are resident together, whereas a 75 % grid (round 2) leaves the next one waiting with workgroups that
// hold CUs and spin (measured, 32 pairs, three batches in rotation: cluster 6 -> 68.5 k, 4 -> 78.2 k, 3 -> 79.2 k frames/s;
// a launch alone: 0.39 / 0.45 / 0.50 ms).  The in-kernel wait is bounded as a second line of defence.
// Tracker grids the resident gate keeps in flight per device (REVO_TRACK_DEPTH, default 2; 1 = one at a time, the round-2
// behaviour; up to 4 for experiments with small clusters: workgroup-time per pair falls with the cluster size -- the
// exchange and the decision are paid by every member -- but a grid of small clusters only fills a fraction of the chip).
static int env_track_depth() {
  const char* e = getenv("REVO_TRACK_SERIAL");
  if (e && *e && *e != '0') return 1;
  return env_int("REVO_TRACK_DEPTH", 2, 1, 4);
}
static int pick_cluster(const revo_ctx* c, int n_pairs) {
  const int resident = c->num_cus * c->blocks_per_cu;
  // a batch grid takes 1/depth of what the device holds: `depth` grids are resident together
  // (depth 1 = one grid at a time: the round-2 shape, 75 % of the chip)
  const int depth = c->knobs.track_depth;
  const int share = n_pairs == 1 || depth == 1 ? (int)(0.75 * resident) : resident / depth;
  int cl = share / std::max(1, n_pairs);
  // a single pair: its members share one XCD (blockIdx % 8), i.e. 32 CUs -- 16 workgroups leave half of them
  // to the build kernels of the next frame
  if (n_pairs == 1) cl = std::min(cl, 16);
  if (cl > TRACK_MAX_CLUSTER) cl = TRACK_MAX_CLUSTER;
  {  // tuning knobs: REVO_TRACK_CLUSTER_ONE (per context), REVO_TRACK_CLUSTER (read when a batch is created)
    const int want = n_pairs == 1 ? c->knobs.cluster_one : env_int("REVO_TRACK_CLUSTER", 0, 0, TRACK_MAX_CLUSTER);
    const int hard = std::min(TRACK_MAX_CLUSTER, resident / std::max(1, n_pairs));
    if (want >= 1) cl = std::min(want, std::max(1, hard));
  }
  if (cl < 1) cl = 1;
  return cl;
}

// Tracker launches of one device are ordered by a RESIDENT GATE.  A cluster's workgroups exchange partial sums inside
// the launch, so two tracker grids must never be PARTIALLY resident at the same time (each would wait for members the
// other keeps off the CUs until the bounded spin gives up with flag 8).  Round 2 serialised the grids completely (launch
// n waited for launch n-1 to finish) -- and a launch lasts as long as its slowest pair (455 us against a mean of 294 us
// per pair), so a third of the tracker's CU time idled behind the tail.  Now launch n
//   * waits for launch n-2 to COMPLETE (at most two tracker grids are ever in flight), and
//   * waits, through a one-wave gate kernel in front of it, until every workgroup of launch n-1 has STARTED
//     (k_track counts its workgroups into a per-device census counter as they start).  A started workgroup holds its CU
//     until it exits, so from then on launch n-1 is complete on the chip and makes progress whatever else arrives;
//     launch n's workgroups fill the CUs that n-1's finished pairs free, and are themselves all resident once n-1 has
//     drained (192 <= 256 CUs; build kernels finish in finite time).  By induction the older of the two grids is always
//     fully resident: no cyclic wait.
// REVO_TRACK_DEPTH=n (1..4, default 2; read per context) sets how many grids may be in flight (launch n waits for launch
// n-depth to complete); REVO_TRACK_SERIAL=1 = depth 1 = the round-2 behaviour.  The chain itself is per DEVICE (the census
// counter is) and keeps the events of the last TRACK_MAX_DEPTH launches, so contexts with different depths may share a device:
// each launch applies its own context's depth.  (Other PROCESSES sharing the GPU are outside this library's reach: the
// bounded spin + flag 8 + REVO_ERR_HIP remain the answer there.)
#define TRACK_MAX_DEPTH 4
struct TrackChain {
  std::mutex mu;
  hipEvent_t ev[TRACK_MAX_DEPTH] = {};   // completion of the last TRACK_MAX_DEPTH launches (ring, slot = launch index % 4)
  hipStream_t st[TRACK_MAX_DEPTH] = {};
  bool has[TRACK_MAX_DEPTH] = {};
  unsigned long long n = 0;      // launches so far
  int depth_seen = 0;            // the depth of the first launch; `mixed` once a launch with another depth arrived
  bool mixed = false;
  unsigned* d_resident = nullptr;
  unsigned started_total = 0;    // workgroups of all launches enqueued so far (the census value once they have all started)
};
static TrackChain g_chain[64];
// launch(): enqueues the tracker grid on s and returns its workgroup count
template <typename F>
static int chained_track_launch(int device, int depth, hipStream_t s, F&& launch) {
  TrackChain& ch = g_chain[device & 63];
  std::lock_guard<std::mutex> lk(ch.mu);
  depth = depth < 1 ? 1 : (depth > TRACK_MAX_DEPTH ? TRACK_MAX_DEPTH : depth);
  if (!ch.ev[0]) {
    for (int i = 0; i < TRACK_MAX_DEPTH; ++i) HIPCHECK(hipEventCreateWithFlags(&ch.ev[i], hipEventDisableTiming));
    HIPCHECK(hipMalloc((void**)&ch.d_resident, 2 * sizeof(unsigned)));  // [0] census, [1] gates that timed out
    // (hipMemset on device memory is asynchronous to the host and runs on the NULL stream; the tracker streams are non-blocking
    // streams, so the first grid could count itself into the census BEFORE the zeroing lands: finish it here, once per device)
    HIPCHECK(hipMemset(ch.d_resident, 0, 2 * sizeof(unsigned)));
    HIPCHECK(hipDeviceSynchronize());
  }
  // launch n-depth must be COMPLETE (at most `depth` grids in flight; with one depth in use every older launch completed
  // before it by induction -- only when contexts with DIFFERENT depths share the device are the older ring slots waited for
  // as well), launch n-1 must be fully RESIDENT: only the newest grid is ever partially on the chip, every older one holds
  // all its CUs and finishes whatever arrives -- no cyclic wait at any depth.  With depth 1 the two coincide: one grid at a time.
  if (!ch.depth_seen) ch.depth_seen = depth;
  if (depth != ch.depth_seen) ch.mixed = true;
  for (int back = depth; back <= (ch.mixed ? TRACK_MAX_DEPTH : depth); ++back) {
    if ((unsigned long long)back > ch.n) break;
    const int slot = (int)((ch.n - back) % TRACK_MAX_DEPTH);
    if (ch.has[slot] && ch.st[slot] != s) HIPCHECK(hipStreamWaitEvent(s, ch.ev[slot], 0));
  }
  const int prev = (int)((ch.n + TRACK_MAX_DEPTH - 1) % TRACK_MAX_DEPTH), cur = (int)(ch.n % TRACK_MAX_DEPTH);
  if (depth > 1 && ch.n > 0 && ch.has[prev] && ch.st[prev] != s) launch_track_gate(ch.d_resident, ch.started_total, s);
  const int n_wg = launch(ch.d_resident);
  HIPCHECK(hipGetLastError());              // (a refused launch adds nothing to the census: the next gate must not wait for it)
  ch.started_total += (unsigned)n_wg;
  HIPCHECK(hipEventRecord(ch.ev[cur], s));
  ch.has[cur] = true;
  ch.st[cur] = s;
  ch.n += 1;
  return REVO_OK;
}
// how many resident gates of this device gave up waiting so far (0 in a healthy process); -1: no tracker launch yet
extern "C" int revo_debug_gate_timeouts_(int device) {
  TrackChain& ch = g_chain[device & 63];
  std::lock_guard<std::mutex> lk(ch.mu);
  if (!ch.d_resident) return -1;
  unsigned v = 0;
  if (hipMemcpy(&v, ch.d_resident + 1, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -2;
  return (int)v;
}
// diagnostics: out3 = {census (workgroups that have started), gates that timed out, workgroups of all launches enqueued so far}.
// After hipDeviceSynchronize census == enqueued in a healthy process.
extern "C" int revo_debug_census_(int device, unsigned out3[3]) {
  TrackChain& ch = g_chain[device & 63];
  std::lock_guard<std::mutex> lk(ch.mu);
  if (!ch.d_resident || !out3) return -1;
  if (hipMemcpy(out3, ch.d_resident, 2 * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -2;
  out3[2] = ch.started_total;
  return 0;
}
static size_t mail_bytes(int n_pairs, int cluster) { return sizeof(unsigned long long) * (size_t)n_pairs * 2 * cluster * TRACK_NVAL; }

// --------------------------------------------------------------- FrameSets --
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int frameset_create(revo_ctx* c, int B, bool with_staging, FrameSet** out) {
  const PyrGeom& g = c->geom;
  FrameSet* fs = new FrameSet();
  fs->B = B;
  // pass 1: size, pass 2: carve
  for (int pass = 0; pass < 2; ++pass) {
    size_t off = 0;
    auto take = [&](size_t bytes) -> void* {
      void* p = pass ? (void*)((char*)fs->blob + off) : nullptr;
      off = align_up(off + bytes, 256);
      return p;
    };
    for (int l = 0; l < g.n_levels; ++l) {
      const LevelGeom& v = g.lv[l];
      const size_t n = (size_t)v.npix * B;
      fs->p.gray[l] = (uint8_t*)take(n);
      fs->p.depth[l] = (float*)take(n * 4);
      fs->p.cs[l] = (uint2*)take((size_t)v.h * v.wpr * 8 * B);
      fs->p.edges[l] = (uint8_t*)take(n);
      fs->p.edges_orig[l] = (uint8_t*)take(n);
      fs->p.scratch[l] = (int*)take(n * 4);
      fs->p.pts[l] = (float4*)take(n * 16);
      fs->p.pts_trk[l] = (float4*)take(n * 16);
      fs->p.dt[l] = (float*)take(n * 4);
      fs->p.table[l] = (float4*)take(n * 16);
      fs->p.hist[l] = (uint8_t*)take((size_t)std::max(1, v.hist_w * v.hist_h) * B);
      fs->p.chunk[l] = (int*)take((size_t)v.w * v.nchunk * B * 4);
      fs->p.cmask[l] = (unsigned*)take((size_t)v.w * v.nchunk * B * 4);
      fs->p.vb[l] = (uint8_t*)take((n + 7) / 8);
      fs->p.ebits[l] = (uint32_t*)take((size_t)v.h * v.wpr * 4 * B);
      if (B > 1 && l < g.n_levels - 1) {  // (batches only: the single-frame API builds its depth pyramid in front of Canny)
        fs->p.stage[l] = (float*)take(n * 4);
        fs->p.epre[l] = (unsigned short*)take((size_t)v.h * v.wpr * 2 * B);
      }
    }
    fs->own_depth0 = fs->p.depth[0];
    fs->p.npts = (int*)take(sizeof(int) * REVO_L * B);
    fs->p.hist_nz = (int*)take(sizeof(int) * REVO_L * B);
    fs->p.strip_tot = (int*)take(sizeof(int) * (size_t)g.total_strips * B);
    fs->p.tile_base = (int*)take(sizeof(int) * (size_t)g.total_tiles * B);
    fs->p.need_full = (int*)take(sizeof(int) * REVO_L * B);
    fs->p.hyst_heavy = (int*)take(sizeof(int) * B);
    fs->p.stage_base = (int*)take(sizeof(int) * (size_t)g.total_tiles * B);
    if (with_staging) {
      fs->d_bgr = (uint8_t*)take((size_t)g.lv[0].npix * 3 * B);
      fs->d_depth = (float*)take((size_t)g.lv[0].npix * 4 * B);
    }
    if (!pass) {
      fs->blob_bytes = off;
      hipError_t e = hipMalloc(&fs->blob, off);
      if (e != hipSuccess) { delete fs; return fail(REVO_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    }
  }
  if (with_staging) {
    HIPCHECK(hipHostMalloc((void**)&fs->h_bgr, (size_t)g.lv[0].npix * 3 * B));
    HIPCHECK(hipHostMalloc((void**)&fs->h_depth, (size_t)g.lv[0].npix * 4 * B));
  }
  HIPCHECK(hipEventCreateWithFlags(&fs->ev_ready, hipEventDisableTiming));
  HIPCHECK(hipEventCreateWithFlags(&fs->ev_free, hipEventDisableTiming));
  HIPCHECK(hipEventCreateWithFlags(&fs->ev_free2, hipEventDisableTiming));
  HIPCHECK(hipEventCreateWithFlags(&fs->ev_edt, hipEventDisableTiming));
  // counts start at zero so an accessor on a not-yet-built pyramid is well defined
  hipMemsetAsync(fs->p.npts, 0, sizeof(int) * REVO_L * B, c->stream);
  HIPCHECK(hipStreamSynchronize(c->stream));
  *out = fs;
  return REVO_OK;
}
static void frameset_destroy(FrameSet* fs) {
  if (!fs) return;
  if (fs->blob) hipFree(fs->blob);
  if (fs->h_bgr) hipHostFree(fs->h_bgr);
  if (fs->h_depth) hipHostFree(fs->h_depth);
  if (fs->ev_ready) hipEventDestroy(fs->ev_ready);
  if (fs->ev_free) hipEventDestroy(fs->ev_free);
  if (fs->ev_free2) hipEventDestroy(fs->ev_free2);
  if (fs->ev_edt) hipEventDestroy(fs->ev_edt);
  if (fs->ev_h2d) hipEventDestroy(fs->ev_h2d);
  delete fs;
}

// enqueue the full per-frame build (imgpyramidrgbd.cpp:43-96) for all B frames.
// borrow_depth (f32 input only): the level-0 depth plane IS the input buffer -- the reference's level 0 is the input
// image too (imgpyramidrgbd.cpp:62-64), and copying 64 x 1.2 MB per batch was 208 of the 236 MB the first kernel moved.
// The caller guarantees the buffer stays valid and unchanged for as long as the pyramids are used.
// with_points = false: stops after fillInEdges; the caller enqueues launch_tile_points itself (batches run it next to the EDT)
static void enqueue_build(revo_ctx* c, FrameSet* fs, const uint8_t* d_bgr, const float* d_depth_f32,
                          const uint16_t* d_depth_u16, float alpha, hipStream_t s, bool borrow_depth = false,
                          bool with_points = true, int frame0 = 0, int nframes = -1, bool with_hyst = true, bool gray_only = false,
                          bool with_fill = true) {
  PyrGeom g = c->geom;
  g.frame0 = frame0;
  const int B = nframes < 0 ? fs->B : nframes;
  fs->p.depth[0] = (borrow_depth && d_depth_f32) ? const_cast<float*>(d_depth_f32) : fs->own_depth0;
  launch_gray_depth(g, fs->p, d_bgr, d_depth_f32, d_depth_u16, alpha, B, s);
  // gray_only (batches that leave their edge lists to the first consumer): the depth half of the pyramid is left to it too
  for (int l = 1; l < g.n_levels; ++l) launch_pyrdown(g, fs->p, l, B, s, gray_only ? 1 : 3);
  launch_canny_nms(g, fs->p, B, s);
  if (!with_hyst) return;  // (batches with REVO_DEFER = 3: the rest is left to the first consumer, run_pending_edt)
  launch_hyst(g, fs->p, B, s);
  if (with_fill) launch_fill(g, fs->p, B, s);  // (pipelined batches may leave it to the first consumer: only the edge lists and the EDT read its result)
  if (with_points) launch_tile_points(g, fs->p, B, s);  // the tracker's (tile-ordered) edge list; the reference's order is built on demand
}

// The tail of a batch build: the edge lists of all frames (two kernels) and the EDT of the keyframes (two kernels) only
// depend on the edge maps, not on each other, and all four are latency-bound with idle CUs around them: the EDT runs on a
// side stream next to the lists (fork after fillInEdges, join before the batch's stream goes on).
static int enqueue_batch_tail(revo_batch* b, hipStream_t s);
static int batch_wait_tracker(revo_batch* b, hipStream_t s);

// ------------------------------------------------------------------ context --
static void ctx_free(revo_ctx* c);
extern "C" int revo_ctx_create(int device, const revo_pyr_settings* pyr, const revo_opt_settings* opt,
                               const revo_tracker_settings* trk, revo_ctx** out) {
  if (!pyr || !out) return fail(REVO_ERR_INVALID_ARG, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(REVO_ERR_HIP, "no HIP device: librevo_hip has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(REVO_ERR_INVALID_ARG, "device ordinal out of range");
  HIPCHECK(hipSetDevice(device));
  revo_ctx* c = new revo_ctx();
  c->device = device;
  c->ps = *pyr;
  c->knobs.track_depth = env_track_depth();
  c->knobs.cluster_one = env_int("REVO_TRACK_CLUSTER_ONE", 0, 0, TRACK_MAX_CLUSTER);
  c->knobs.redundant_one = env_int("REVO_TRACK_REDUNDANT_ONE", 1024, 0, 1 << 30);
  c->knobs.h2d_mode = env_int("REVO_H2D_STREAMS", 0, 0, 2);
  c->knobs.direct_h2d = env_int("REVO_DIRECT_H2D", 1, 0, 1);
  c->knobs.stage_edge_depths = env_int("REVO_STAGE_EDGE_DEPTHS", 0, 0, 1);  // built, bit-exact, 80 MB less traffic per step and 2.6 % SLOWER: off (profiles/r06_ab_stage_edge_depths.txt)
  {
    const char* e1 = getenv("REVO_TRACK_KSPEC_ONE");
    const char* eb = getenv("REVO_TRACK_KSPEC");
    const size_t n1 = e1 ? strlen(e1) : 0;
    c->knobs.kspec_one_set = n1 > 0 || !(eb && *eb);
    for (int i = 0; i < REVO_L; ++i) {
      int k = i < 2 ? 3 : TRACK_KMAX;
      if (n1 == 1) k = e1[0] - '0'; else if (n1 > 1) k = e1[std::min<size_t>(i, n1 - 1)] - '0';
      c->knobs.kspec_one[i] = std::max(1, std::min(TRACK_KMAX, k));
    }
  }
  c->knobs.upload_blocks = env_int("REVO_UPLOAD_BLOCKS", 64, 1, 4096);
  c->knobs.h2d_kernel = env_int("REVO_H2D_KERNEL", 0, 0, 2);
  c->knobs.h2d_wait_poll = env_int("REVO_H2D_WAIT_POLL", 1, 0, 1);
  c->knobs.h2d_max_run_mb = env_int("REVO_H2D_MAX_RUN_MB", 64, 1, 4096);  // (profiles/r06_h2d_run_sizes.txt: 2 / 8 / 24 / 64 MB / unbounded)
  if (opt) c->os = *opt; else revo_opt_settings_default(&c->os);
  if (trk) c->ts = *trk; else revo_tracker_settings_default(&c->ts);
  std::string why;
  if (build_geom(c->ps, &c->geom, &why)) { delete c; return fail(REVO_ERR_INVALID_ARG, why); }
  build_track_params(c, &c->tp);
  struct Guard { revo_ctx* c; ~Guard() { if (c) ctx_free(c); } } guard{c};  // a HIP failure below frees what exists so far
  HIPCHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHECK(hipStreamCreateWithFlags(&c->build_stream, hipStreamNonBlocking));
  if (env_int("REVO_VOTE_STREAM", 1, 0, 1)) {
    HIPCHECK(hipStreamCreateWithFlags(&c->vote_stream, hipStreamNonBlocking));
    HIPCHECK(hipEventCreateWithFlags(&c->ev_vote, hipEventDisableTiming));
  }
  HIPCHECK(hipHostMalloc((void**)&c->h_desc, sizeof(PairDesc)));
  HIPCHECK(hipHostMalloc((void**)&c->h_res, sizeof(revo_pair_result) * 3));
  HIPCHECK(hipHostMalloc((void**)&c->h_seq, sizeof(unsigned) * 4));
  memset(c->h_seq, 0, sizeof(unsigned) * 4);
  HIPCHECK(hipHostMalloc((void**)&c->h_eval, sizeof(EvalOut)));
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, device));
  c->num_cus = prop.multiProcessorCount;
  c->blocks_per_cu = std::min(track_blocks_per_cu(), 2);
  HIPCHECK(hipMalloc((void**)&c->d_mail, mail_bytes(1, TRACK_MAX_CLUSTER)));
  HIPCHECK(hipMemset(c->d_mail, 0, mail_bytes(1, TRACK_MAX_CLUSTER)));
  size_t maxpix = 0;
  for (int l = 0; l < c->geom.n_levels; ++l) maxpix = std::max(maxpix, (size_t)c->geom.lv[l].npix);
  HIPCHECK(hipMalloc((void**)&c->d_marks, sizeof(int) * maxpix));
  HIPCHECK(hipMalloc((void**)&c->d_hist8, sizeof(int) * 8));
  HIPCHECK(hipMalloc((void**)&c->d_vote_done, sizeof(unsigned)));
  // the vote kernels leave their scratch clean: zero it once
  HIPCHECK(hipMemset(c->d_marks, 0, sizeof(int) * maxpix));
  HIPCHECK(hipMemset(c->d_hist8, 0, sizeof(int) * 8));
  HIPCHECK(hipMemset(c->d_vote_done, 0, sizeof(unsigned)));
  HIPCHECK(hipHostMalloc((void**)&c->h_hist8, sizeof(int) * 16));  // hist[4], overlaps[4], sequence word
  memset(c->h_hist8, 0, sizeof(int) * 16);
  HIPCHECK(hipDeviceSynchronize());  // the zeroing above runs on the NULL stream; the context's streams are non-blocking
  guard.c = nullptr;
  *out = c;
  return REVO_OK;
}

static void ctx_free(revo_ctx* c) {
  hipSetDevice(c->device);
  if (c->build_stream) hipStreamSynchronize(c->build_stream);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->vote_stream) { (void)hipStreamSynchronize(c->vote_stream); (void)hipStreamDestroy(c->vote_stream); }
  if (c->ev_vote) (void)hipEventDestroy(c->ev_vote);
  for (int k = 0; k < 2; ++k) if (c->pair_streams[k]) (void)hipStreamDestroy(c->pair_streams[k]);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->copy_stream2) (void)hipStreamDestroy(c->copy_stream2);
  if (c->frame_copy_stream) { (void)hipStreamSynchronize(c->frame_copy_stream); (void)hipStreamDestroy(c->frame_copy_stream); }
  for (auto& p : c->past) { hipFree(p.d_pts); hipFree(p.d_n); }
  for (auto& p : c->past_pool) { hipFree(p.d_pts); hipFree(p.d_n); }
  if (c->build_stream) hipStreamDestroy(c->build_stream);
  for (FrameSet* fs : c->pool) frameset_destroy(fs);
  hipHostFree(c->h_desc); hipHostFree(c->h_res); hipHostFree(c->h_eval); hipHostFree(c->h_seq); hipFree(c->d_mail);
  hipFree(c->d_marks); hipFree(c->d_hist8); hipHostFree(c->h_hist8); hipFree(c->d_vote_done);
  hipFree(c->d_pcl);
  if (c->stream) hipStreamDestroy(c->stream);
  (void)hipGetLastError();  // a partially built context frees null handles on purpose
  delete c;
}
static void ctx_ref(revo_ctx* c) { c->refs.fetch_add(1); }
static void ctx_unref(revo_ctx* c) { if (c->refs.fetch_sub(1) == 1) ctx_free(c); }
static void pairs_jobs_release(revo_ctx* c);
extern "C" void revo_ctx_destroy(revo_ctx* c) {
  if (!c) return;
  pairs_jobs_release(c);  // the job slots hold batches, and batches hold the context
  ctx_unref(c);
}
// Single-frame FrameSets (20 MB of HBM + 2 MB of pinned staging each) are pooled, but the pool only grew on demand: the
// first full-length run of a sequential driver paid a hipMalloc + hipHostMalloc per new frame in flight (bench r02:
// first run 5x slower than the rest).  A driver that knows its queue depth reserves them up front.
extern "C" int revo_ctx_reserve_framesets_(revo_ctx* c, int total) {
  if (!c) return fail(REVO_ERR_INVALID_ARG, "null context");
  HIPCHECK(hipSetDevice(c->device));
  for (;;) {
    {
      std::lock_guard<std::mutex> lk(c->mu);
      if (c->framesets_created >= total) return REVO_OK;
      c->framesets_created += 1;
    }
    FrameSet* fs = nullptr;
    const int rc = frameset_create(c, 1, true, &fs);
    std::lock_guard<std::mutex> lk(c->mu);
    if (rc) { c->framesets_created -= 1; return rc; }
    c->pool.push_back(fs);
  }
}
extern "C" int revo_ctx_device_(const revo_ctx* c) { return c ? c->device : -1; }
// used by revo_vo.hip / revo_pipeline.hip
extern "C" void revo_ctx_retain_(revo_ctx* c) { if (c) ctx_ref(c); }
extern "C" void revo_ctx_release_(revo_ctx* c) { if (c) ctx_unref(c); }

extern "C" int revo_ctx_set_tracker(revo_ctx* c, const revo_opt_settings* opt, const revo_tracker_settings* trk) {
  if (!c) return fail(REVO_ERR_INVALID_ARG, "null context");
  std::lock_guard<std::mutex> lk(c->mu);
  if (opt) c->os = *opt;
  if (trk) c->ts = *trk;
  build_track_params(c, &c->tp);
  return REVO_OK;
}

extern "C" int revo_ctx_histogram_level(const revo_ctx* c) { return c ? c->ts.histogram_level : -1; }

extern "C" int revo_ctx_camera(const revo_ctx* c, int lvl, float out6[6]) {
  if (!c || !out6) return fail(REVO_ERR_INVALID_ARG, "null argument");
  if (lvl < 0 || lvl >= c->geom.n_levels) return fail(REVO_ERR_LEVEL, "level out of range");
  const LevelGeom& v = c->geom.lv[lvl];
  out6[0] = v.fx; out6[1] = v.fy; out6[2] = v.cx; out6[3] = v.cy; out6[4] = (float)v.w; out6[5] = (float)v.h;
  return REVO_OK;
}

// ----------------------------------------------------------------- pyramids --
// Order the consumer stream after the (asynchronous) build of a single-frame pyramid.
// Runs a deferred keyframe EDT of the set on stream s (which is first ordered behind the build).
static int run_pending_edt(revo_ctx* c, FrameSet* fs, hipStream_t s) {
  std::lock_guard<std::mutex> lk(fs->edt_mu);
  if (!fs->edt_pending) {
    // already enqueued by an earlier consumer: a consumer on ANOTHER stream must still wait for it
    if (fs->has_edt && fs->edt_stream != s) HIPCHECK(hipStreamWaitEvent(s, fs->ev_edt, 0));
    return REVO_OK;
  }
  if (fs->has_ready) HIPCHECK(hipStreamWaitEvent(s, fs->ev_ready, 0));
  if (fs->hyst_pending) { launch_hyst(c->geom, fs->p, fs->B, s); launch_fill(c->geom, fs->p, fs->B, s); fs->hyst_pending = false; fs->fill_pending = false; }
  if (fs->fill_pending) { launch_fill(c->geom, fs->p, fs->B, s); fs->fill_pending = false; }
  // (the edge lists and the EDT do not depend on each other; running the EDT first -- next to the memory-bound first kernels of
  // the following build instead of its VALU-bound NMS -- was measured equal: profiles/r04_ab_aux_order.txt)
  if (fs->pts_pending) {
    // ... with the depth half of the pyramid in front of them when the build left it out (nothing on the build stream reads the
    // coarser depth levels or the validity bits: they feed the edge lists only).  Only then: with REVO_SPLIT_DEPTH=0 the build ran
    // the fused kernel and the depth half must not run a second time.
    PyrGeom gp = c->geom;
    if (fs->depth_pending) {
      // ... and the edges are final by now, so the depth pass -- which streams every depth of a level anyway -- also leaves the
      // depths of the level's EDGE pixels behind, compact and in list order (k_edge_prefix tells it where): the edge-list
      // write then reads ~10 MB of staged depths instead of every 128-byte line of the depth planes that holds an edge
      // pixel (~100 MB per 64-frame batch).  Measured (profiles/r06_ab_stage_edge_depths.txt): k_pts_tiles
      // moves 61 MB instead of 141 (1.09x its algorithmic bytes), the step 72 MB less -- and is 2.6 % SLOWER: 2.3 M scattered
      // 4-byte stores and the prefix kernel cost the depth pass more time than the sparse line reads they replace, and no kernel
      // of this chain is bandwidth-bound.  Bit-exact, OFF by default (REVO_STAGE_EDGE_DEPTHS=1 turns it on).
      const bool stage = c->knobs.stage_edge_depths && fs->p.stage[0] != nullptr && c->geom.n_levels > 1;
      if (stage) launch_edge_prefix(c->geom, fs->p, fs->B, s);
      for (int l = 1; l < c->geom.n_levels; ++l) launch_pyrdown(c->geom, fs->p, l, fs->B, s, 2, stage);
      gp.pts_staged = stage ? 1 : 0;
      fs->depth_pending = false;
    }
    launch_tile_points(gp, fs->p, fs->B, s);
    fs->pts_pending = false;
  }
  launch_keyframe(c->geom, fs->p, 0, 2, fs->edt_count, s);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipEventRecord(fs->ev_edt, s));
  fs->has_edt = true;
  fs->edt_stream = s;
  fs->edt_pending = false;
  return REVO_OK;
}
// the next build into a set's planes: the deferred EDT of the previous build may still be reading its edge maps on another stream
static int wait_edt_before_rebuild(FrameSet* fs, hipStream_t s) {
  std::lock_guard<std::mutex> lk(fs->edt_mu);
  if (fs->has_edt && fs->edt_stream != s) HIPCHECK(hipStreamWaitEvent(s, fs->ev_edt, 0));
  return REVO_OK;
}
// The in-place upload of a frame (revo_pyramid_create*, rows in page-locked host memory) as a KERNEL (REVO_H2D_KERNEL=1; the default
// is hipMemcpyAsync): 16 bytes per lane straight out of the caller's rows over PCIe into the set's input plane; same bytes, same
// stream, same event.  Why it exists: about one hipMemcpyAsync in 36 000 does not RETURN for 6-13 ms (profiles/r06_slow_run_probe.txt,
// per-section maxima of 54 000 frame submissions: every other HIP call of a submission stayed below 1.3 ms; inside bench.py a
// build's kernel launches were seen to take 9 ms once as well).  The IO thread sits in
// that call, the queue of four pyramids runs dry, the consumer waits: THE slow run of the sequential stream's 60-frame measurement
// (one run in 15-25 at 60-80 % of the median in every round since the second; on a long stream it is 7 ms in ~4 s, 0.2 %).  With the
// kernel there is no such stall -- 0 slow runs in 900, the longest gap between two poses 1.0 ms -- but the median drops from
// 4.06-4.20 k to 3.55-3.87 k frames/s (the shader's reads over PCIe run next to the tracker and the build; on the build stream or on
// a high-priority stream it is the same): a latency-critical consumer can switch it on, throughput keeps the DMA engine.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_upload_rows(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t src_stride,
                                                     size_t row_bytes, int rows) {
  // row_bytes is a multiple of 4 for every plane type at every width the library accepts only when w % 4 == 0: handled per byte tail
  const size_t vec = row_bytes / 16, tail0 = vec * 16;
  for (int y = blockIdx.y; y < rows; y += gridDim.y) {
    const uint8_t* s = src + (size_t)y * src_stride;
    uint8_t* d = dst + (size_t)y * row_bytes;
    const bool aligned = (((uintptr_t)s | (uintptr_t)d) & 15) == 0;
    if (aligned) {
      for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < vec; i += (size_t)gridDim.x * blockDim.x)
        ((v4u*)d)[i] = __builtin_nontemporal_load((const v4u*)s + i);
      for (size_t i = tail0 + blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
    } else {
      for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
    }
  }
}
static hipError_t upload_rows(void* dst, const void* src, size_t src_stride, size_t row_bytes, int rows, int max_blocks, hipStream_t s) {
  void* dsrc = nullptr;
  hipError_t e = hipHostGetDevicePointer(&dsrc, const_cast<void*>(src), 0);  // (registered memory: the device's view of it)
  if (e != hipSuccess) { (void)hipGetLastError(); return e; }
  if (src_stride == row_bytes) {  // one contiguous block: treat it as a single long row
    const size_t bytes = row_bytes * rows;
    const int blocks = (int)std::min<size_t>((size_t)max_blocks, (bytes / 16 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_upload_rows, dim3(blocks, 1), dim3(256), 0, s, (uint8_t*)dst, (const uint8_t*)dsrc, bytes, bytes, 1);
  } else {
    hipLaunchKernelGGL(k_upload_rows, dim3(2, std::min(rows, 128)), dim3(256), 0, s, (uint8_t*)dst, (const uint8_t*)dsrc, src_stride, row_bytes, rows);
  }
  return hipGetLastError();
}
// (debug: the longest time each section of a frame submission has taken so far -- profiles/slow_run_probe.py prints them)
static std::atomic<unsigned long long> g_sec_max_ns[12];
extern "C" void revo_debug_section_max_(unsigned long long out[12], int reset) {
  for (int i = 0; i < 12; ++i) { out[i] = g_sec_max_ns[i].load(); if (reset) g_sec_max_ns[i].store(0); }
}
extern "C" void revo_debug_section_note_(int i, unsigned long long ns) {
  if (i < 0 || i >= 12) return;
  unsigned long long cur = g_sec_max_ns[i].load(std::memory_order_relaxed);
  while (ns > cur && !g_sec_max_ns[i].compare_exchange_weak(cur, ns)) {}
}
struct SecTimer {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(int i) {
    const auto n = std::chrono::steady_clock::now();
    revo_debug_section_note_(i, (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count());
    t = n;
  }
};
// Waits for an event by POLLING it (the in-place frame upload, ~50-100 us): the IO thread gets its answer a few microseconds after the
// copy has finished instead of after hipEventSynchronize's park-and-wake; the sequential stream runs 1-3 % faster with it
// (profiles/r06_slow_run_probe.txt, REVO_H2D_WAIT_POLL=0 / 1 alternating on one box: 4.05-4.19 k against 4.09-4.24 k frames/s).
static int wait_event_polled(hipEvent_t ev) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; ++spin) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return REVO_OK;
    if (e != hipErrorNotReady) { (void)hipGetLastError(); return fail(REVO_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e)); }
    (void)hipGetLastError();
    for (int k = 0; k < 32; ++k) __builtin_ia32_pause();
    if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;  // something is wrong: block
  }
  HIPCHECK(hipEventSynchronize(ev));
  return REVO_OK;
}
static int wait_ready_on(revo_ctx* c, const revo_pyr* p, hipStream_t s) {
  // single-frame pyramids: built on the build stream; batch views: built on the batch's / the caller's stream
  // (revo_batch_build records the event) -- either way the consumer stream is ordered behind the build
  if (p->fs->has_ready) HIPCHECK(hipStreamWaitEvent(s, p->fs->ev_ready, 0));
  if (p->fs->has_aux) HIPCHECK(hipStreamWaitEvent(s, p->fs->ev_aux, 0));
  // an accessor / single-pair call on a batch view: runs the deferred EDT if it is still pending, waits for it otherwise
  return run_pending_edt(c, p->fs, s);
}
static int wait_ready(revo_ctx* c, const revo_pyr* p) { return wait_ready_on(c, p, c->stream); }
// The vote stream's work on a pyramid `p` is enqueued.  A single-frame pyramid's planes are recycled behind ev_free2
// (revo_pyramid_destroy); a batch view's planes belong to the caller, whose ordering promises are all about the tracker
// stream -- so that stream waits here (batch views are not what the sequential loop runs on).
static int vote_enqueued(revo_ctx* c, const revo_pyr* p) {
  if (!c->vote_stream || p->owns_fs) return REVO_OK;
  HIPCHECK(hipEventRecord(c->ev_vote, c->vote_stream));
  HIPCHECK(hipStreamWaitEvent(c->stream, c->ev_vote, 0));
  return REVO_OK;
}

static int pyramid_create_common(revo_ctx* c, const uint8_t* bgr, size_t bgr_stride, const void* depth,
                                 size_t depth_stride, bool is_u16, double scale, double ts, revo_pyr** out) {
  if (!c || !bgr || !depth || !out) return fail(REVO_ERR_INVALID_ARG, "null argument");
  HIPCHECK(hipSetDevice(c->device));
  const int w = c->geom.lv[0].w, h = c->geom.lv[0].h;
  if (bgr_stride < (size_t)w * 3 || depth_stride < (size_t)w * (is_u16 ? 2 : 4))
    return fail(REVO_ERR_INVALID_ARG, "stride smaller than a row");
  FrameSet* fs = nullptr;
  SecTimer sec;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->pool.empty()) { fs = c->pool.back(); c->pool.pop_back(); }
  }
  sec.lap(0);  // the context's lock + the pool
  if (!fs) {
    int rc = frameset_create(c, 1, true, &fs);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    c->framesets_created += 1;
  }
  // The reference clones its inputs (imgpyramidrgbd.cpp:51,54): the caller may reuse its buffers when this returns.
  // Everything after the clone is asynchronous on the build stream, so building frame N+1 overlaps tracking frame N exactly
  // like the reference's IO thread (iowrapperRGBD.cpp:279, system.cpp:96).
  //   * rows in PAGE-LOCKED host memory (hipHostMalloc / hipHostRegister / torch pin_memory -- what a decoder thread that owns
  //     its buffers uses): the DMA engine reads them in place on a copy stream (next to the previous frame's build kernels),
  //     and the call returns when that copy has finished -- the device-side plane IS the clone.  No host-side copy at all
  //     (round 4: the 2.1 MB memcpy into the staging area was 0.18 ms of the IO thread's 0.26 ms per frame).
  //   * pageable rows: copied into the set's pinned staging area first, as before.
  const size_t brow = (size_t)w * 3, drow = (size_t)w * (is_u16 ? 2 : 4);
  hipStream_t bs = c->build_stream;
  auto page_locked = [](const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
  };
  const bool direct = c->knobs.direct_h2d && page_locked(bgr) && page_locked(depth) &&
                      page_locked(bgr + (size_t)(h - 1) * bgr_stride + brow - 1) &&
                      page_locked((const char*)depth + (size_t)(h - 1) * depth_stride + drow - 1);
  sec.lap(1);  // pointer attributes (and a new frame set, if the pool was empty)
  if (direct) {
    {
      std::lock_guard<std::mutex> lk(c->mu);
      if (!c->frame_copy_stream) {
        // (experiment knob REVO_COPY_STREAM_PRIO: 1 = a high-priority stream, which HIP serves from a hardware-queue pool of its own)
        int lo = 0, hi = 0;
        if (env_int("REVO_COPY_STREAM_PRIO", 0, 0, 1) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo)
          HIPCHECK(hipStreamCreateWithPriority(&c->frame_copy_stream, hipStreamNonBlocking, hi));
        else
          HIPCHECK(hipStreamCreateWithFlags(&c->frame_copy_stream, hipStreamNonBlocking));
      }
    }
    if (!fs->ev_h2d) HIPCHECK(hipEventCreateWithFlags(&fs->ev_h2d, hipEventDisableTiming));
    hipStream_t cs = c->knobs.h2d_kernel == 2 ? bs : c->frame_copy_stream;  // (2: experiment -- the upload kernel in front of the build, same stream)
    if (fs->has_ready) HIPCHECK(hipStreamWaitEvent(cs, fs->ev_ready, 0));  // the previous build out of this set's input planes is done
    if (fs->has_free) HIPCHECK(hipStreamWaitEvent(cs, fs->ev_free, 0));    // ... and its last consumer (depth level 0 is read in place)
    if (fs->has_free2) HIPCHECK(hipStreamWaitEvent(cs, fs->ev_free2, 0));
    sec.lap(6);  // the copy stream's waits for the set's previous users
    // (memory the device cannot address through hipHostGetDevicePointer goes the old way)
    if (c->knobs.h2d_kernel && upload_rows(fs->d_bgr, bgr, bgr_stride, brow, h, c->knobs.upload_blocks, cs) == hipSuccess) {}
    else if (bgr_stride == brow) HIPCHECK(hipMemcpyAsync(fs->d_bgr, bgr, brow * h, hipMemcpyHostToDevice, cs));
    else HIPCHECK(hipMemcpy2DAsync(fs->d_bgr, brow, bgr, bgr_stride, brow, h, hipMemcpyHostToDevice, cs));
    sec.lap(7);  // upload, colour
    if (c->knobs.h2d_kernel && upload_rows(fs->d_depth, depth, depth_stride, drow, h, c->knobs.upload_blocks, cs) == hipSuccess) {}
    else if (depth_stride == drow) HIPCHECK(hipMemcpyAsync(fs->d_depth, depth, drow * h, hipMemcpyHostToDevice, cs));
    else HIPCHECK(hipMemcpy2DAsync(fs->d_depth, drow, depth, depth_stride, drow, h, hipMemcpyHostToDevice, cs));
    sec.lap(8);  // upload, depth
    HIPCHECK(hipEventRecord(fs->ev_h2d, cs));
    sec.lap(9);  // hipEventRecord
    HIPCHECK(hipStreamWaitEvent(bs, fs->ev_h2d, 0));
    sec.lap(2);  // the build stream's wait for the copies
    // the clone exists: the caller's buffers are free again
    if (c->knobs.h2d_wait_poll) { int rc = wait_event_polled(fs->ev_h2d); if (rc) return rc; }
    else HIPCHECK(hipEventSynchronize(fs->ev_h2d));
    sec.lap(3);  // waiting for them
  } else {
    if (fs->has_ready) HIPCHECK(hipEventSynchronize(fs->ev_ready));  // previous upload out of this staging is done
    for (int y = 0; y < h; ++y) memcpy(fs->h_bgr + (size_t)y * brow, bgr + (size_t)y * bgr_stride, brow);
    for (int y = 0; y < h; ++y) memcpy((char*)fs->h_depth + (size_t)y * drow, (const char*)depth + (size_t)y * depth_stride, drow);
    if (fs->has_free) HIPCHECK(hipStreamWaitEvent(bs, fs->ev_free, 0));  // last consumer of the recycled set is done
    if (fs->has_free2) HIPCHECK(hipStreamWaitEvent(bs, fs->ev_free2, 0));
    HIPCHECK(hipMemcpyAsync(fs->d_bgr, fs->h_bgr, brow * h, hipMemcpyHostToDevice, bs));
    HIPCHECK(hipMemcpyAsync(fs->d_depth, fs->h_depth, drow * h, hipMemcpyHostToDevice, bs));
  }
  const float alpha = is_u16 ? (float)(1.0f / scale) : 0.0f;  // iowrapperRGBD.cpp:327
  // (the f32 staging plane is the set's own memory: level 0 reads it in place)
  enqueue_build(c, fs, fs->d_bgr, is_u16 ? nullptr : fs->d_depth, is_u16 ? (const uint16_t*)fs->d_depth : nullptr, alpha, bs, true);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipEventRecord(fs->ev_ready, bs));
  sec.lap(4);  // enqueueing the build
  fs->has_ready = true;
  fs->ready_stream = bs;
  revo_pyr* p = new revo_pyr{c, fs, 0, true, false, ts, false, false};
  ctx_ref(c);
  *out = p;
  return REVO_OK;
}

extern "C" int revo_pyramid_create(revo_ctx* ctx, const uint8_t* bgr, size_t bgr_stride, const float* depth_m,
                                   size_t depth_stride, double timestamp, revo_pyr** out) {
  return pyramid_create_common(ctx, bgr, bgr_stride, depth_m, depth_stride, false, 1.0, timestamp, out);
}
extern "C" int revo_pyramid_create_u16(revo_ctx* ctx, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth_raw,
                                       size_t depth_stride, double depth_scale_factor, double timestamp, revo_pyr** out) {
  if (!(depth_scale_factor > 0)) return fail(REVO_ERR_INVALID_ARG, "depth_scale_factor must be > 0");
  return pyramid_create_common(ctx, bgr, bgr_stride, depth_raw, depth_stride, true, depth_scale_factor, timestamp, out);
}

extern "C" void revo_pyramid_destroy(revo_pyr* p) {
  if (!p) return;
  if (p->owns_fs) {
    // reuse is ordered by events: the next build into this set waits for everything the
    // consumer stream has enqueued against it so far
    hipSetDevice(p->ctx->device);
    std::lock_guard<std::mutex> lk(p->ctx->mu);
    hipEventRecord(p->fs->ev_free, p->ctx->stream);
    p->fs->has_free = true;
    if (p->ctx->vote_stream) { hipEventRecord(p->fs->ev_free2, p->ctx->vote_stream); p->fs->has_free2 = true; }
    p->ctx->pool.push_back(p->fs);
  }
  revo_ctx* c = p->owns_fs ? p->ctx : nullptr;
  delete p;
  if (c) ctx_unref(c);
}

extern "C" int revo_pyramid_make_keyframe(revo_pyr* p) {
  if (!p) return fail(REVO_ERR_INVALID_ARG, "null pyramid");
  HIPCHECK(hipSetDevice(p->ctx->device));
  { int rc = wait_ready(p->ctx, p); if (rc) return rc; }
  if (!p->dt_ready) launch_keyframe(p->ctx->geom, p->fs->p, p->frame, 1, 1, p->ctx->stream);
  HIPCHECK(hipGetLastError());
  p->dt_ready = false;  // (a second makeKeyframe computes again, like the reference)
  p->is_kf = true;
  p->table_built = false;
  return REVO_OK;
}
// The distance transforms of a frame that is LIKELY to become the keyframe (revo_vo.hip: the previous vote was close to a change),
// enqueued while the vote that decides it is still running; revo_pyramid_make_keyframe then finds them in place.  The frame is
// not a keyframe until that call; if it never comes the planes are simply never read (single-frame pyramids only).
extern "C" int revo_pyramid_prepare_keyframe_(revo_pyr* p) {
  if (!p) return fail(REVO_ERR_INVALID_ARG, "null pyramid");
  if (!p->owns_fs || p->is_kf || p->dt_ready) return REVO_OK;
  HIPCHECK(hipSetDevice(p->ctx->device));
  { int rc = wait_ready(p->ctx, p); if (rc) return rc; }
  launch_keyframe(p->ctx->geom, p->fs->p, p->frame, 1, 1, p->ctx->stream);
  HIPCHECK(hipGetLastError());
  p->dt_ready = true;
  return REVO_OK;
}
extern "C" int revo_pyramid_is_keyframe(const revo_pyr* p) { return p && p->is_kf; }
extern "C" double revo_pyramid_timestamp(const revo_pyr* p) { return p ? p->ts : 0.0; }

extern "C" int revo_pyramid_read(revo_pyr* p, revo_plane what, int lvl, void* dst, size_t cap, size_t* count) {
  if (!p) return fail(REVO_ERR_INVALID_ARG, "null pyramid");
  revo_ctx* c = p->ctx;
  HIPCHECK(hipSetDevice(c->device));
  if (lvl < 0 || lvl >= c->geom.n_levels) return fail(REVO_ERR_LEVEL, "level out of range");
  const LevelGeom& v = c->geom.lv[lvl];
  const FramePlanes& P = p->fs->p;
  const size_t f = (size_t)p->frame;
  { int rc = wait_ready(c, p); if (rc) return rc; }
  HIPCHECK(hipStreamSynchronize(c->stream));
  const void* src = nullptr;
  size_t n = v.npix, esz = 1;
  switch (what) {
    case REVO_PLANE_GRAY: src = P.gray[lvl] + f * v.npix; break;
    case REVO_PLANE_DEPTH: src = P.depth[lvl] + f * v.npix; esz = 4; break;
    case REVO_PLANE_EDGES: src = P.edges[lvl] + f * v.npix; break;
    case REVO_PLANE_EDGES_ORIG:  // returnOrigEdges, imgpyramidrgbd.h:67-75
      src = (v.has_orig ? P.edges_orig[lvl] : P.edges[lvl]) + f * v.npix;  // no fill-in at this level: the clone equals edgesPyr
      break;
    case REVO_PLANE_DT:
      if (!p->is_kf) return fail(REVO_ERR_NOT_KEYFRAME, "distance transform not built: call makeKeyframe");
      src = P.dt[lvl] + f * v.npix; esz = 4; break;
    case REVO_PLANE_GRADTABLE:
      if (!p->is_kf) return fail(REVO_ERR_NOT_KEYFRAME, "optimizationStructure not built");
      if (!p->table_built) {  // lazily materialised: the tracker itself samples the DT plane
        launch_grad_table(c->geom, P, p->frame, 1, 1, c->stream);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(c->stream));
        p->table_built = true;
      }
      src = P.table[lvl] + f * v.npix; esz = 16; break;
    case REVO_PLANE_EDGES3D: {
      if (!p->ref_list_built) {  // the reference's column-major order (imgpyramidrgbd.cpp:199-226): materialised for the accessor
        PyrGeom g1 = c->geom;
        g1.frame0 = p->frame;
        launch_compact(g1, P, 1, c->stream);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(c->stream));
        p->ref_list_built = true;
      }
      int np = 0;
      HIPCHECK(hipMemcpy(&np, P.npts + f * REVO_L + lvl, sizeof(int), hipMemcpyDeviceToHost));
      n = (size_t)np; src = P.pts[lvl] + f * v.npix; esz = 16; break;
    }
    case REVO_PLANE_EDGES3D_TILED: {
      int np = 0;
      HIPCHECK(hipMemcpy(&np, P.npts + f * REVO_L + lvl, sizeof(int), hipMemcpyDeviceToHost));
      n = (size_t)np; src = P.pts_trk[lvl] + f * v.npix; esz = 16; break;
    }
    case REVO_PLANE_HIST:
      if (v.patch <= 0) { n = 0; src = P.hist[lvl]; break; }
      n = (size_t)v.hist_w * v.hist_h; src = P.hist[lvl] + f * n; break;
    default: return fail(REVO_ERR_INVALID_ARG, "unknown plane");
  }
  if (count) *count = n;
  if (dst) {
    if (n * esz > cap) return fail(REVO_ERR_CAPACITY, "host buffer too small");
    if (n) HIPCHECK(hipMemcpy(dst, src, n * esz, hipMemcpyDeviceToHost));
  }
  return REVO_OK;
}

// ImgPyramidRGBD::generateColoredPcl(lvl, clrPcl, densePcl), imgpyramidrgbd.cpp:279-327.
extern "C" int revo_pyramid_colored_pcl(revo_pyr* p, int lvl, int dense, float* dst8, size_t cap_points, size_t* count) {
  if (!p) return fail(REVO_ERR_INVALID_ARG, "null pyramid");
  revo_ctx* c = p->ctx;
  HIPCHECK(hipSetDevice(c->device));
  if (lvl < 0 || lvl >= c->geom.n_levels) return fail(REVO_ERR_LEVEL, "level out of range");
  if (!p->owns_fs || !p->fs->d_bgr)
    return fail(REVO_ERR_INVALID_ARG, "batch views keep no colour image (rgbFullSize): use revo_pyramid_create");
  { int rc = wait_ready(c, p); if (rc) return rc; }
  std::lock_guard<std::mutex> lk(c->mu);
  const PyrGeom& g = c->geom;
  const size_t n0 = g.lv[0].npix, slots = (size_t)g.lv[0].w * g.lv[0].nchunk;
  if (!c->d_pcl) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_out = take(n0 * 32), o_a = take(n0 / 4 * 3 + 3), o_b = take(n0 / 16 * 3 + 3), o_ch = take(slots * 4),
                 o_m = take(slots * 4), o_t = take(4);
    HIPCHECK(hipMalloc((void**)&c->d_pcl, off));
    c->d_pcl_out = (float*)(c->d_pcl + o_out);
    c->d_pcl_clr[0] = (uint8_t*)(c->d_pcl + o_a); c->d_pcl_clr[1] = (uint8_t*)(c->d_pcl + o_b);
    c->d_pcl_chunk = (int*)(c->d_pcl + o_ch); c->d_pcl_mask = (unsigned*)(c->d_pcl + o_m); c->d_pcl_total = (int*)(c->d_pcl + o_t);
  }
  hipStream_t s = c->stream;
  const uint8_t* clr = p->fs->d_bgr;  // the full-resolution clone (imgpyramidrgbd.cpp:51), pyrDown'ed lvl times
  for (int l = 0; l < lvl; ++l) {
    uint8_t* d = c->d_pcl_clr[l & 1];
    launch_pyrdown_bgr(clr, g.lv[l].w, g.lv[l].h, d, s);
    clr = d;
  }
  const int cap = (int)std::min<size_t>(dst8 ? cap_points : 0, (size_t)g.lv[lvl].npix);
  launch_colored_pcl(g, p->fs->p, p->frame, lvl, dense ? 1 : 0, clr, c->d_pcl_chunk, c->d_pcl_mask, c->d_pcl_total,
                     dst8 ? cap : g.lv[lvl].npix, c->d_pcl_out, s);
  HIPCHECK(hipGetLastError());
  int total = 0;
  HIPCHECK(hipMemcpyAsync(&total, c->d_pcl_total, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  if (count) *count = (size_t)total;
  if (dst8) {
    if ((size_t)total > cap_points) return fail(REVO_ERR_CAPACITY, "host buffer too small");
    if (total) HIPCHECK(hipMemcpy(dst8, c->d_pcl_out, (size_t)total * 32, hipMemcpyDeviceToHost));
  }
  return REVO_OK;
}

// ------------------------------------------------------------------ tracking --
static void fill_desc(PairDesc* d, const revo_pyr* ref, const revo_pyr* curr, const float* R, const float* T) {
  const PyrGeom& g = ref->ctx->geom;
  memset(d, 0, sizeof(*d));
  for (int l = 0; l < g.n_levels; ++l) {
    d->pts[l] = curr->fs->p.pts_trk[l] + (size_t)curr->frame * g.lv[l].npix;
    d->dt[l] = ref->fs->p.dt[l] + (size_t)ref->frame * g.lv[l].npix;
  }
  d->npts = curr->fs->p.npts + (size_t)curr->frame * REVO_L;
  if (R) memcpy(d->R, R, sizeof(float) * 9); else { d->R[0] = d->R[4] = d->R[8] = 1.f; }
  if (T) memcpy(d->T, T, sizeof(float) * 3);
}

static int check_pair(const revo_ctx* c, const revo_pyr* ref, const revo_pyr* curr) {
  if (!c || !ref || !curr) return fail(REVO_ERR_INVALID_ARG, "null argument");
  if (ref->ctx != c || curr->ctx != c) return fail(REVO_ERR_INVALID_ARG, "pyramid belongs to another context");
  if (!ref->is_kf) return fail(REVO_ERR_NOT_KEYFRAME, "optimizationStructure not built! (reference frame is not a keyframe)");
  return REVO_OK;
}

// Waits for a sequence word a kernel writes (system scope) after its results in pinned host memory: a short spin
// (the kernel is usually about to finish: a stream synchronise costs a sleep / wake-up of the calling thread),
// then the stream.
// (debug: how often the spin below ran out and what the longest wait was -- profiles/slow_run_probe.py prints them)
static std::atomic<unsigned long long> g_wait_fallbacks{0}, g_wait_max_ns{0}, g_wait_calls{0};
extern "C" void revo_debug_wait_stats_(unsigned long long out[3]) {
  out[0] = g_wait_calls.load(); out[1] = g_wait_fallbacks.load(); out[2] = g_wait_max_ns.load();
}
static int wait_seq(revo_ctx* c, volatile unsigned* word, unsigned want) {
  static const int spin_limit = env_int("REVO_WAIT_SPINS", 200000, 1000, 2000000000);
  const auto t0 = std::chrono::steady_clock::now();
  g_wait_calls.fetch_add(1, std::memory_order_relaxed);
  auto note = [&]() {
    const unsigned long long ns = (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long cur = g_wait_max_ns.load(std::memory_order_relaxed);
    while (ns > cur && !g_wait_max_ns.compare_exchange_weak(cur, ns)) {}
  };
  for (int spin = 0; spin < spin_limit; ++spin) {
    if (*word == want) { std::atomic_thread_fence(std::memory_order_acquire); note(); return REVO_OK; }
    __builtin_ia32_pause();
  }
  g_wait_fallbacks.fetch_add(1, std::memory_order_relaxed);
  HIPCHECK(hipStreamSynchronize(c->stream));
  note();
  if (*word != want) return fail(REVO_ERR_HIP, "a kernel finished without publishing its result");
  return REVO_OK;
}

// One single-pair tracker launch into result slot `slot` (0 / 1: the VO driver's look-ahead, 2: the public single-pair
// calls, which may run on a context whose VO driver keeps a speculative launch outstanding); *seq_out identifies it for track_wait.
static int track_launch(revo_ctx* c, const revo_pyr* ref, const revo_pyr* curr, const float* R, const float* T,
                        const TrackParams& tp, int slot, unsigned* seq_out) {
  fill_desc(c->h_desc, ref, curr, R, T);
  { int rc = wait_ready(c, ref); if (rc) return rc; rc = wait_ready(c, curr); if (rc) return rc; }
  TrackParams tp1 = tp;
  tp1.redundant_n = c->knobs.redundant_one;
  // a single pair has the chip to itself: one more retry next to the candidate at the two finest levels costs its 16 workgroups
  // little and saves passes (sequential stream 4.23 -> 4.29 k frames/s, profiles/r06_single_stream_sweep.txt).  REVO_TRACK_KSPEC_ONE:
  // one digit per level (finest first) or one for all; an explicit REVO_TRACK_KSPEC applies to single pairs too.
  if (c->knobs.kspec_one_set)
    for (int i = 0; i < REVO_L; ++i) tp1.kspec[i] = c->knobs.kspec_one[i];
  const unsigned seq = c->seq_next++;
  if (c->seq_next == 0) c->seq_next = 1;
  const int rc = chained_track_launch(c->device, c->knobs.track_depth, c->stream, [&](unsigned* d_resident) {
    return launch_track_one(*c->h_desc, tp1, c->h_res + slot, c->h_eval, c->d_mail, &c->mail_epoch, pick_cluster(c, 1), c->h_seq + slot,
                            seq, d_resident, c->stream);
  });
  if (rc) return rc;
  *seq_out = seq;
  return REVO_OK;
}
static int track_wait(revo_ctx* c, int slot, unsigned seq) { return wait_seq(c, c->h_seq + slot, seq); }

enum { SINGLE_SLOT = 2 };
static int run_single(revo_ctx* c, const revo_pyr* ref, const revo_pyr* curr, const float* R, const float* T,
                      const TrackParams& tp) {
  unsigned seq = 0;
  int rc = track_launch(c, ref, curr, R, T, tp, SINGLE_SLOT, &seq);
  if (rc) return rc;
  if (tp.eval_only) {  // the EvalOut record is written by many lanes: wait for the stream, not for a word
    HIPCHECK(hipStreamSynchronize(c->stream));
    return REVO_OK;
  }
  return track_wait(c, SINGLE_SLOT, seq);
}

static int decode_track(const revo_pair_result& r, float R[9], float T[3], float* err, int* status, revo_residual_info* info,
                        int32_t* iters) {
  if (r.flags & 2) return fail(REVO_ERR_NOT_ORTHOGONAL, "R is not orthogonal (Sophus::SO3 precondition)");
  if (r.flags & 8) return fail(REVO_ERR_HIP, "tracker: the workgroups of the pair could not exchange partial sums in time (device shared?)");
  memcpy(R, r.R, sizeof(float) * 9); memcpy(T, r.T, sizeof(float) * 3);
  if (err) *err = r.err;
  if (status) *status = r.status;
  if (info) { info->good_pts_edges = r.good; info->bad_pts_edges = r.bad; info->sum_error_unweighted = 0.f; info->sum_error_weighted = 0.f; }
  if (iters) memcpy(iters, r.evals, sizeof(int32_t) * REVO_L);
  return REVO_OK;
}

// ---- split calls for the VO driver's look-ahead (revo_vo.hip); not part of the public ABI
extern "C" int revo_track_launch_(revo_ctx* c, const revo_pyr* ref, const revo_pyr* curr, const float R[9], const float T[3], int slot,
                                  unsigned* seq_out) {
  int rc = check_pair(c, ref, curr);
  if (rc) return rc;
  HIPCHECK(hipSetDevice(c->device));
  std::lock_guard<std::mutex> lk(c->mu);
  TrackParams tp = c->tp;
  tp.eval_only = 0;
  return track_launch(c, ref, curr, R, T, tp, slot & 1, seq_out);
}
extern "C" int revo_track_wait_(revo_ctx* c, int slot, unsigned seq, float R[9], float T[3], float* err, int* status) {
  if (!c) return fail(REVO_ERR_INVALID_ARG, "null context");
  int rc = track_wait(c, slot & 1, seq);
  if (rc) return rc;
  return decode_track(c->h_res[slot & 1], R, T, err, status, nullptr, nullptr);
}

extern "C" int revo_optimizer_track_level(revo_ctx* c, const revo_pyr* ref, const revo_pyr* curr, float R[9], float T[3],
                                          int lvl, revo_residual_info* info, float* err) {
  int rc = check_pair(c, ref, curr);
  if (rc) return rc;
  if (!R || !T) return fail(REVO_ERR_INVALID_ARG, "null pose");
  if (lvl < 0 || lvl >= c->geom.n_levels) return fail(REVO_ERR_LEVEL, "level out of range");
  HIPCHECK(hipSetDevice(c->device));
  std::lock_guard<std::mutex> lk(c->mu);
  TrackParams tp = c->tp;
  tp.lvl_begin = tp.lvl_end = lvl; tp.check_init = 0; tp.eval_only = 0;
  rc = run_single(c, ref, curr, R, T, tp);
  if (rc) return rc;
  return decode_track(c->h_res[SINGLE_SLOT], R, T, err, nullptr, info, nullptr);
}

extern "C" int revo_optimizer_eval(revo_ctx* c, const revo_pyr* ref, const revo_pyr* curr, const float R[9],
                                   const float T[3], int lvl, revo_residual_info* info, float* err, float A[36],
                                   float b[6]) {
  int rc = check_pair(c, ref, curr);
  if (rc) return rc;
  if (!R || !T) return fail(REVO_ERR_INVALID_ARG, "null pose");
  if (lvl < 0 || lvl >= c->geom.n_levels) return fail(REVO_ERR_LEVEL, "level out of range");
  HIPCHECK(hipSetDevice(c->device));
  std::lock_guard<std::mutex> lk(c->mu);
  TrackParams tp = c->tp;
  tp.lvl_begin = tp.lvl_end = lvl; tp.check_init = 0; tp.eval_only = 1;
  rc = run_single(c, ref, curr, R, T, tp);
  if (rc) return rc;
  const EvalOut& e = *c->h_eval;
  if (info) { info->good_pts_edges = e.good; info->bad_pts_edges = e.bad; info->sum_error_unweighted = e.sum_u; info->sum_error_weighted = e.sum_w; }
  if (err) *err = e.mean_err;
  if (A) memcpy(A, e.A, sizeof(float) * 36);
  if (b) memcpy(b, e.b, sizeof(float) * 6);
  return REVO_OK;
}

// inc = A.ldlt().solve(b) with A(i,i) *= 1 + lambda (optimizer.cpp:258-262) for n systems, on the device
extern "C" int revo_optimizer_solve6(revo_ctx* c, int n, const float* A36_b6_lambda, float* x6) {
  if (!c || !A36_b6_lambda || !x6 || n <= 0) return fail(REVO_ERR_INVALID_ARG, "bad argument");
  HIPCHECK(hipSetDevice(c->device));
  std::lock_guard<std::mutex> lk(c->mu);
  float *d_in = nullptr, *d_out = nullptr;
  HIPCHECK(hipMalloc((void**)&d_in, sizeof(float) * 43 * (size_t)n));
  hipError_t e = hipMalloc((void**)&d_out, sizeof(float) * 6 * (size_t)n);
  if (e == hipSuccess) e = hipMemcpyAsync(d_in, A36_b6_lambda, sizeof(float) * 43 * (size_t)n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) { launch_solve6(d_in, n, d_out, c->stream); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpyAsync(x6, d_out, sizeof(float) * 6 * (size_t)n, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  hipFree(d_in); hipFree(d_out);
  if (e != hipSuccess) return fail(REVO_ERR_HIP, std::string("revo_optimizer_solve6: ") + hipGetErrorString(e));
  return REVO_OK;
}

// profile builds (-DREVO_TRACK_PROFILE): the phase cycle counters the last single-pair launch left in the EvalOut record
extern "C" int revo_debug_track_profile_(revo_ctx* c, float out13[13]) {
  if (!c || !out13) return REVO_ERR_INVALID_ARG;
  memcpy(out13, c->h_eval->A, sizeof(float) * 13);
  return REVO_OK;
}

extern "C" int revo_tracker_track_frames(revo_ctx* c, const revo_pyr* ref, const revo_pyr* curr, float R[9], float T[3],
                                         float* err, int* status, revo_residual_info* info, int32_t iters[REVO_MAX_LEVELS]) {
  int rc = check_pair(c, ref, curr);
  if (rc) return rc;
  if (!R || !T) return fail(REVO_ERR_INVALID_ARG, "null pose");
  HIPCHECK(hipSetDevice(c->device));
  std::lock_guard<std::mutex> lk(c->mu);
  TrackParams tp = c->tp;
  tp.eval_only = 0;
  rc = run_single(c, ref, curr, R, T, tp);
  if (rc) return rc;
  return decode_track(c->h_res[SINGLE_SLOT], R, T, err, status, info, iters);
}

// assessTrackingQuality (tracker.cpp:118-201) in two halves: enqueue the vote kernels / read the counts
static int assess_launch(revo_ctx* c, const float T_w_curr[16], const revo_pyr* curr, int* nframes_out, unsigned* seq_out) {
  *nframes_out = -1;  // -1: nothing to vote on (tracker.cpp:121)
  if (c->past.empty() || !c->ts.check_tracking_results) return REVO_OK;
  const int hl = c->ts.histogram_level;
  if (hl < 0 || hl >= c->geom.n_levels) return fail(REVO_ERR_LEVEL, "histogram_level outside the pyramid");
  hipStream_t vs = c->vote_stream ? c->vote_stream : c->stream;
  { int rc = wait_ready_on(c, curr, vs); if (rc) return rc; }
  float inv[16];
  mat4_inverse(T_w_curr, inv);
  int nframes = 0;
  VoteArgs va{};
  for (int fr = 0; fr < c->ts.n_frames_hist_voting && fr < (int)c->past.size() && fr < 3; ++fr) {
    float tf[16];
    mat4_mul(inv, c->past[fr].T_w, tf);
    float* RT = va.RT[fr];
    for (int cc = 0; cc < 3; ++cc) for (int r = 0; r < 3; ++r) RT[cc * 3 + r] = tf[cc * 4 + r];
    RT[9] = tf[12]; RT[10] = tf[13]; RT[11] = tf[14];
    va.pts[fr] = c->past[fr].d_pts;
    va.n[fr] = c->past[fr].d_n;
    ++nframes;
  }
  const int use_orig = c->geom.lv[hl].has_orig;  // returnOrigEdges(lvl), imgpyramidrgbd.h:67-75 (elsewhere the clone equals edgesPyr)
  const unsigned seq = c->seq_next++;
  if (c->seq_next == 0) c->seq_next = 1;
  launch_vote(c->geom, curr->fs->p, curr->frame, hl, nframes, va, c->d_marks, c->d_hist8, c->d_vote_done, c->h_hist8, seq,
              use_orig, vs);
  HIPCHECK(hipGetLastError());
  { int rc = vote_enqueued(c, curr); if (rc) return rc; }
  *nframes_out = nframes;
  *seq_out = seq;
  return REVO_OK;
}
static int assess_wait(revo_ctx* c, int nframes, unsigned seq, int* status, int32_t hist4[4], int32_t overlaps4[4]) {
  if (status) *status = REVO_TRACKER_STATE_OK;
  if (nframes < 0) return REVO_OK;
  { int rc = wait_seq(c, (volatile unsigned*)(c->h_hist8 + 8), seq); if (rc) return rc; }
  const int* hist = c->h_hist8;
  const int* ov = c->h_hist8 + 4;
  const float wts[4] = {0.f, 1.f, 1.25f, 1.5f};  // tracker.cpp:231-234
  float overlapMeasure = 0.0f;
  const int hsize = 1 + nframes;
  for (int k = 1; k < hsize; ++k) overlapMeasure += (ov[k] * wts[k]);
  if (hist4) memcpy(hist4, hist, sizeof(int32_t) * 4);
  if (overlaps4) memcpy(overlaps4, ov, sizeof(int32_t) * 4);
  if (status) *status = (overlapMeasure >= ov[0] || hsize < 4) ? REVO_TRACKER_STATE_OK : REVO_TRACKER_STATE_NEW_KF;  // tracker.cpp:184
  return REVO_OK;
}

extern "C" int revo_tracker_assess_quality(revo_ctx* c, const float T_w_curr[16], const revo_pyr* curr, int* status,
                                           int32_t hist4[4], int32_t overlaps4[4]) {
  if (!c || !T_w_curr || !curr) return fail(REVO_ERR_INVALID_ARG, "null argument");
  if (curr->ctx != c) return fail(REVO_ERR_INVALID_ARG, "pyramid belongs to another context");
  if (hist4) memset(hist4, 0, sizeof(int32_t) * 4);
  if (overlaps4) memset(overlaps4, 0, sizeof(int32_t) * 4);
  if (status) *status = REVO_TRACKER_STATE_OK;
  HIPCHECK(hipSetDevice(c->device));
  std::lock_guard<std::mutex> lk(c->mu);
  int nframes = -1;
  unsigned seq = 0;
  int rc = assess_launch(c, T_w_curr, curr, &nframes, &seq);
  if (rc) return rc;
  return assess_wait(c, nframes, seq, status, hist4, overlaps4);
}
// split form for the VO driver's look-ahead (revo_vo.hip); not part of the public ABI
// 1: the vote runs beside the tracker stream (revo_vo.hip then launches the look-ahead tracker FIRST)
extern "C" int revo_vote_overlaps_(const revo_ctx* c) { return c && c->vote_stream ? 1 : 0; }
extern "C" int revo_assess_launch_(revo_ctx* c, const float T_w_curr[16], const revo_pyr* curr, int* nframes_out, unsigned* seq_out) {
  if (!c || !T_w_curr || !curr || curr->ctx != c) return fail(REVO_ERR_INVALID_ARG, "bad argument");
  HIPCHECK(hipSetDevice(c->device));
  std::lock_guard<std::mutex> lk(c->mu);
  return assess_launch(c, T_w_curr, curr, nframes_out, seq_out);
}
// ratio_out (may be null): overlapMeasure / overlaps[0] of this vote -- how far the frame is from asking for a new keyframe (< 1 does,
// tracker.cpp:184); +inf while fewer than three past clouds vote (the answer is OK whatever the numbers) or nothing voted
extern "C" int revo_assess_wait_(revo_ctx* c, int nframes, unsigned seq, int* status, float* ratio_out) {
  if (!c) return fail(REVO_ERR_INVALID_ARG, "null context");
  int32_t ov[4] = {0, 0, 0, 0};
  const int rc = assess_wait(c, nframes, seq, status, nullptr, ov);
  if (ratio_out) {
    *ratio_out = INFINITY;
    if (rc == REVO_OK && nframes >= 3 && ov[0] > 0) {
      const float wts[4] = {0.f, 1.f, 1.25f, 1.5f};
      float m = 0.0f;
      for (int k = 1; k < 1 + nframes && k < 4; ++k) m += ov[k] * wts[k];
      *ratio_out = m / (float)ov[0];
    }
  }
  return rc;
}

// a past-cloud buffer for at least `need` points: recycled if one is large enough (no hipMalloc per frame in
// the steady state), sized for what is stored -- the histogram level's cloud, not level 0's
static int past_take(revo_ctx* c, size_t need, Past* out) {
  for (size_t k = 0; k < c->past_pool.size(); ++k)
    if (c->past_pool[k].cap >= need) { *out = c->past_pool[k]; c->past_pool.erase(c->past_pool.begin() + k); return REVO_OK; }
  Past p{};
  p.cap = std::max<size_t>(need, 1024);
  HIPCHECK(hipMalloc((void**)&p.d_pts, sizeof(float4) * p.cap));
  hipError_t e = hipMalloc((void**)&p.d_n, sizeof(int));
  if (e != hipSuccess) { (void)hipFree(p.d_pts); return fail(REVO_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  *out = p;
  return REVO_OK;
}

extern "C" int revo_tracker_add_old_pcl(revo_ctx* c, const revo_pyr* src, int lvl, const float T_w[16], double ts) {
  if (!c || !src || !T_w) return fail(REVO_ERR_INVALID_ARG, "null argument");
  if (lvl < 0 || lvl >= c->geom.n_levels) return fail(REVO_ERR_LEVEL, "level out of range");
  HIPCHECK(hipSetDevice(c->device));
  std::lock_guard<std::mutex> lk(c->mu);
  // the reference copies the Eigen matrix (tracker.cpp:219); the copy stays in HBM, in a
  // recycled buffer sized for the level
  hipStream_t vs = c->vote_stream ? c->vote_stream : c->stream;  // the votes that read the copy run there
  { int rc = wait_ready_on(c, src, vs); if (rc) return rc; }
  Past p{};
  const size_t f = (size_t)src->frame;
  { int rc = past_take(c, (size_t)c->geom.lv[lvl].npix, &p); if (rc) return rc; }
  // the count stays on the device: one kernel copies the n valid points and n (stream ordered)
  launch_copy_cloud(p.d_pts, src->fs->p.pts_trk[lvl] + f * c->geom.lv[lvl].npix, p.d_n, src->fs->p.npts + f * REVO_L + lvl,
                    vs);
  HIPCHECK(hipGetLastError());
  { int rc = vote_enqueued(c, src); if (rc) return rc; }
  p.n = -1;
  memcpy(p.T_w, T_w, sizeof(float) * 16);
  p.ts = ts;
  c->past.push_back(p);
  return REVO_OK;
}

// addOldPclAndPose(const Eigen::MatrixXf& pcl, const Eigen::Matrix4f& worldPose, double timeStamp), tracker.cpp:209-223,
// with the cloud in host memory (4 x n column-major = n packed float4)
extern "C" int revo_tracker_add_old_pcl_host(revo_ctx* c, const float* pcl, size_t n, const float T_w[16], double ts) {
  if (!c || !T_w || (!pcl && n)) return fail(REVO_ERR_INVALID_ARG, "null argument");
  if (n > (size_t)c->geom.lv[0].npix) return fail(REVO_ERR_CAPACITY, "more points than pixels");
  HIPCHECK(hipSetDevice(c->device));
  std::lock_guard<std::mutex> lk(c->mu);
  Past p{};
  { int rc = past_take(c, n, &p); if (rc) return rc; }
  const int ni = (int)n;
  // pageable source: the runtime has read it when the call returns (the caller's matrix may die right after)
  hipStream_t vs = c->vote_stream ? c->vote_stream : c->stream;
  if (n) HIPCHECK(hipMemcpyAsync(p.d_pts, pcl, sizeof(float4) * n, hipMemcpyHostToDevice, vs));
  HIPCHECK(hipMemcpyAsync(p.d_n, &ni, sizeof(int), hipMemcpyHostToDevice, vs));
  HIPCHECK(hipStreamSynchronize(vs));
  p.n = ni;
  memcpy(p.T_w, T_w, sizeof(float) * 16);
  p.ts = ts;
  c->past.push_back(p);
  return REVO_OK;
}

extern "C" int revo_tracker_clear_past(revo_ctx* c) {  // tracker.cpp:248-257
  if (!c) return fail(REVO_ERR_INVALID_ARG, "null context");
  HIPCHECK(hipSetDevice(c->device));
  std::lock_guard<std::mutex> lk(c->mu);
  while ((int)c->past.size() > c->ts.n_frames_hist_voting) {  // stream-ordered reuse of the buffers
    c->past_pool.push_back(c->past.front());
    c->past.pop_front();
  }
  return REVO_OK;
}
// a new REVO::start builds a new TrackerNew (system.cpp:107): empty lists.  Internal (revo_vo.hip).
extern "C" void revo_tracker_reset_past_(revo_ctx* c) {
  if (!c) return;
  std::lock_guard<std::mutex> lk(c->mu);
  while (!c->past.empty()) { c->past_pool.push_back(c->past.front()); c->past.pop_front(); }
}
extern "C" int revo_tracker_past_size(const revo_ctx* c) { return c ? (int)c->past.size() : 0; }

// -------------------------------------------------------------------- batch --
static int env_defer_level() {  // read when a batch is created
  if (!env_int("REVO_EDT_DEFER", 1, 0, 1)) return 0;
  // default 2 (round 4: 87.1 k -> 95.6 k frames/s together with a stream of their own for the deferred kernels and four batches
  // in rotation, profiles/r04_ab_defer_levels.txt; 3 was measured too: 92.8 k)
  return env_int("REVO_DEFER", 2, 0, 3);
}
extern "C" int revo_batch_create(revo_ctx* c, int n_pairs, revo_batch** out) {
  if (!c || !out || n_pairs <= 0) return fail(REVO_ERR_INVALID_ARG, "bad argument");
  HIPCHECK(hipSetDevice(c->device));
  revo_batch* b = new revo_batch();
  b->ctx = c; b->n_pairs = n_pairs;
  ctx_ref(c);
  int rc = frameset_create(c, 2 * n_pairs, false, &b->fs);
  if (rc) { delete b; ctx_unref(c); return rc; }
  struct Guard { revo_batch* b; ~Guard() { if (b) revo_batch_destroy(b); } } guard{b};  // frees a partial batch on failure
  HIPCHECK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  HIPCHECK(hipEventCreate(&b->ev0)); HIPCHECK(hipEventCreate(&b->ev1));
  HIPCHECK(hipEventCreateWithFlags(&b->ev_upload, hipEventDisableTiming));
  HIPCHECK(hipEventCreateWithFlags(&b->ev_trk, hipEventDisableTiming));
  // REVO_BUILD_FORK=1: the keyframes' EDT on a side stream next to the edge-list kernels.  Off by default: when the two
  // really run concurrently (their streams on different hardware queues) the pipelined step collapses (GPU_MAX_HW_QUEUES=8:
  // 78.5 k -> 47.4 k frames/s), and with HIP's default four queues, where they mostly share a queue, it gains nothing
  // (77.1 k with the fork, 78.7 k without).
  if (env_int("REVO_BUILD_FORK", 0, 0, 1)) {
    HIPCHECK(hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
    HIPCHECK(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
  }
  b->cluster = pick_cluster(c, n_pairs);
  b->defer = env_defer_level();
  b->split_depth = env_int("REVO_SPLIT_DEPTH", 1, 0, 1);
  b->defer_fill = env_int("REVO_DEFER_FILL", 0, 0, 1);
  HIPCHECK(hipMalloc((void**)&b->d_mail, mail_bytes(n_pairs, b->cluster)));
  HIPCHECK(hipMemset(b->d_mail, 0, mail_bytes(n_pairs, b->cluster)));
  for (int f = 0; f < 2 * n_pairs; ++f) b->views.push_back(revo_pyr{c, b->fs, f, false, (f % 2) == 0, 0.0, false, false});
  HIPCHECK(hipHostMalloc((void**)&b->h_descs, sizeof(PairDesc) * n_pairs));
  HIPCHECK(hipHostMalloc((void**)&b->h_flags, sizeof(revo_pair_result) * n_pairs));
  HIPCHECK(hipMalloc((void**)&b->d_descs, sizeof(PairDesc) * n_pairs));
  for (int i = 0; i < n_pairs; ++i) fill_desc(&b->h_descs[i], &b->views[2 * i], &b->views[2 * i + 1], nullptr, nullptr);
  HIPCHECK(hipStreamSynchronize(c->stream));  // frameset_create's memset
  HIPCHECK(hipStreamSynchronize(nullptr));    // the mailbox zeroing (NULL stream; the batch's streams are non-blocking)
  guard.b = nullptr;
  *out = b;
  return REVO_OK;
}
extern "C" void revo_batch_destroy(revo_batch* b) {
  if (!b) return;
  hipSetDevice(b->ctx->device);
  if (b->stream) hipStreamSynchronize(b->stream);
  if (b->side) { hipStreamSynchronize(b->side); hipStreamDestroy(b->side); }
  if (b->ev_fork) hipEventDestroy(b->ev_fork);
  if (b->ev_join) hipEventDestroy(b->ev_join);
  hipHostFree(b->h_descs); hipHostFree(b->h_flags); hipFree(b->d_descs); hipFree(b->d_mail);
  if (b->ev0) hipEventDestroy(b->ev0);
  if (b->ev1) hipEventDestroy(b->ev1);
  if (b->ev_upload) hipEventDestroy(b->ev_upload);
  if (b->ev_trk) hipEventDestroy(b->ev_trk);
  if (b->stream) hipStreamDestroy(b->stream);
  frameset_destroy(b->fs);
  revo_ctx* c = b->ctx;
  delete b;
  ctx_unref(c);
}

static int batch_wait_tracker(revo_batch* b, hipStream_t s) {
  if (b->has_trk && b->trk_stream != s) HIPCHECK(hipStreamWaitEvent(s, b->ev_trk, 0));
  return REVO_OK;
}
// A tracker grid on a stream other than the build's: ordered behind the build whatever was deferred (with
// REVO_DEFER=0 / REVO_EDT_DEFER=0 / REVO_BUILD_FORK=1 nothing is pending, so run_pending_edt orders nothing -- and the edge
// lists, counts and DT planes the grid reads are written by the build stream).  Cheap and idempotent.
static int batch_wait_build(revo_batch* b, hipStream_t s) {
  // (with work deferred to the first consumer, run_pending_edt does the ordering: it either runs that work on s behind the
  // "built" event or waits for the "prepared" event of whoever ran it -- no second barrier packet on the tracker's stream)
  if (b->defer >= 1 && !b->side) return REVO_OK;
  if (b->fs->has_ready && b->fs->ready_stream != s) HIPCHECK(hipStreamWaitEvent(s, b->fs->ev_ready, 0));
  return REVO_OK;
}
static int batch_mark_tracker(revo_batch* b, hipStream_t s) {
  HIPCHECK(hipEventRecord(b->ev_trk, s));
  b->has_trk = true;
  b->trk_stream = s;
  return REVO_OK;
}

// the depth half of the pyramid travels with the deferred edge lists (REVO_DEFER >= 2, no side stream; REVO_SPLIT_DEPTH=0 keeps it
// in the build: an experiment knob read when the batch is created)
static bool batch_splits_depth(const revo_batch* b) { return !b->side && b->defer >= 2 && b->split_depth; }
// fillInEdges travels with the deferred work too (REVO_DEFER = 2, REVO_DEFER_FILL: read when the batch is created): nothing on the
// build stream reads its result -- the next kernels of that stream belong to the NEXT build
static bool batch_defers_fill(const revo_batch* b) { return !b->side && b->defer == 2 && b->defer_fill; }
static int enqueue_batch_tail(revo_batch* b, hipStream_t s) {
  const PyrGeom& g = b->ctx->geom;
  if (b->side) {
    // The batch's stream does NOT wait for the EDT: what needs it -- the tracker of this batch, the accessors of its views,
    // the next build into the same planes -- waits for ev_join itself, so the next batch's first build kernels (on the
    // caller's stream) can start while this batch's EDT is still running.
    HIPCHECK(hipEventRecord(b->ev_fork, s));
    HIPCHECK(hipStreamWaitEvent(b->side, b->ev_fork, 0));
    launch_keyframe(g, b->fs->p, 0, 2, b->n_pairs, b->side);  // frame 2i = keyframe of pair i
    HIPCHECK(hipEventRecord(b->ev_join, b->side));
    b->fs->ev_aux = b->ev_join; b->fs->has_aux = true;
    launch_tile_points(g, b->fs->p, b->fs->B, s);
  } else if (b->defer >= 1) {
    // Work left to the batch's first consumer (run_pending_edt: the tracker launch on ITS stream, revo_batch_prepare on a
    // stream of the caller's choice, an accessor, revo_batch_sync) -- the build stream is the critical chain of the
    // pipelined step.  REVO_DEFER = 1: the keyframes' EDT (97 us less on the build stream; round 3);  2: the edge lists of all
    // frames too (they only depend on the edge maps and nothing on the build stream reads them);  3: hysteresis + fill-in as
    // well (the build ends behind the Canny NMS).  REVO_EDT_DEFER=0 / REVO_DEFER=0: nothing is deferred.
    const int lvl = b->defer;
    if (lvl < 2) launch_tile_points(g, b->fs->p, b->fs->B, s);
    std::lock_guard<std::mutex> lk(b->fs->edt_mu);
    b->fs->hyst_pending = lvl >= 3;
    b->fs->fill_pending = batch_defers_fill(b);
    b->fs->pts_pending = lvl >= 2;
    b->fs->depth_pending = batch_splits_depth(b);
    b->fs->edt_pending = true; b->fs->edt_count = b->n_pairs;
  } else {
    launch_tile_points(g, b->fs->p, b->fs->B, s);
    launch_keyframe(g, b->fs->p, 0, 2, b->n_pairs, s);
  }
  return REVO_OK;
}

static int batch_build_f32(revo_batch* b, const uint8_t* d_bgr, const float* d_depth, void* stream, bool borrow) {
  if (!b || !d_bgr || !d_depth) return fail(REVO_ERR_INVALID_ARG, "null argument");
  HIPCHECK(hipSetDevice(b->ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : b->stream;
  // (Splitting the batch into slices of pairs on 2 / 4 streams was measured in both rounds: no gain in round 1,
  // and with the round-2 kernels the step goes from 0.64 ms to 0.73 / 0.84 ms next to a tracker -- the build
  // kernels are throughput-limited, smaller launches only add tails.)
  // (last_results stays: a caller that pipelines track_only(k) -> build(k) -> sync still gets the flag-8 check of launch k;
  // the result buffer must stay valid until revo_batch_sync or the next revo_batch_track_only)
  if (b->fs->has_aux) HIPCHECK(hipStreamWaitEvent(s, b->ev_join, 0));  // the previous build's EDT still reads these planes
  { int rc = wait_edt_before_rebuild(b->fs, s); if (rc) return rc; }     // ... or its deferred EDT, on whichever stream ran it
  { int rc = batch_wait_tracker(b, s); if (rc) return rc; }              // ... and its tracker grid still reads lists and DT planes
  // (Measured and not kept: the two halves of the batch as two concurrent kernel chains -- 78.4 k -> 70.5 k frames/s.)
  enqueue_build(b->ctx, b->fs, d_bgr, d_depth, nullptr, 0.f, s, borrow, false, 0, -1, b->side || b->defer < 3, batch_splits_depth(b),
                !batch_defers_fill(b));
  { int rc = enqueue_batch_tail(b, s); if (rc) return rc; }
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipEventRecord(b->fs->ev_ready, s));  // accessors / single-pair calls on the batch's views wait for this
  b->fs->has_ready = true;
  b->fs->ready_stream = s;
  for (auto& v : b->views) { v.table_built = false; v.ref_list_built = false; }
  return REVO_OK;
}
extern "C" int revo_batch_build(revo_batch* b, const uint8_t* d_bgr, const float* d_depth, void* stream) {
  return batch_build_f32(b, d_bgr, d_depth, stream, false);
}
extern "C" int revo_batch_build_borrow(revo_batch* b, const uint8_t* d_bgr, const float* d_depth, void* stream) {
  return batch_build_f32(b, d_bgr, d_depth, stream, true);
}

static int batch_upload_init(revo_batch* b, const float* h_init_RT, hipStream_t s) {
  if (!h_init_RT && b->identity_uploaded) return REVO_OK;  // same descriptors as last time: no copy on the stream
  b->identity_uploaded = (h_init_RT == nullptr);
  // the previous upload reads h_descs asynchronously: let it finish before rewriting the poses
  HIPCHECK(hipEventSynchronize(b->ev_upload));
  for (int i = 0; i < b->n_pairs; ++i) {
    PairDesc& d = b->h_descs[i];
    if (h_init_RT) { memcpy(d.R, h_init_RT + 12 * i, sizeof(float) * 9); memcpy(d.T, h_init_RT + 12 * i + 9, sizeof(float) * 3); }
    else { memset(d.R, 0, sizeof(d.R)); d.R[0] = d.R[4] = d.R[8] = 1.f; memset(d.T, 0, sizeof(d.T)); }
  }
  HIPCHECK(hipMemcpyAsync(b->d_descs, b->h_descs, sizeof(PairDesc) * b->n_pairs, hipMemcpyHostToDevice, s));
  HIPCHECK(hipEventRecord(b->ev_upload, s));
  return REVO_OK;
}

extern "C" int revo_batch_prepare(revo_batch* b, void* stream) {
  if (!b) return fail(REVO_ERR_INVALID_ARG, "null argument");
  HIPCHECK(hipSetDevice(b->ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : b->stream;
  if (b->fs->has_aux) HIPCHECK(hipStreamWaitEvent(s, b->ev_join, 0));
  return run_pending_edt(b->ctx, b->fs, s);
}

extern "C" int revo_batch_track_only(revo_batch* b, const float* h_init_RT, revo_pair_result* d_results, void* stream) {
  if (!b || !d_results) return fail(REVO_ERR_INVALID_ARG, "null argument");
  HIPCHECK(hipSetDevice(b->ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : b->stream;
  // an earlier grid of this batch on another stream still reads the descriptors and the mailbox
  { int rc0 = batch_wait_tracker(b, s); if (rc0) return rc0; }
  int rc = batch_upload_init(b, h_init_RT, s);
  if (rc) return rc;
  TrackParams tp = b->ctx->tp;
  tp.eval_only = 0;
  { int rc1 = batch_wait_build(b, s); if (rc1) return rc1; }
  if (b->fs->has_aux) HIPCHECK(hipStreamWaitEvent(s, b->ev_join, 0));  // the keyframes' EDT (side stream of the build)
  { int rc2 = run_pending_edt(b->ctx, b->fs, s); if (rc2) return rc2; }   // ... or deferred to this launch
  b->last_results = d_results;
  rc = chained_track_launch(b->ctx->device, b->ctx->knobs.track_depth, s, [&](unsigned* d_resident) {
    // (live timing, ADVICE r05: the pair sits INSIDE the chain -- behind the waits for older grids, the pose upload and the
    // resident gate, directly around the grid)
    if (b->tev0) (void)hipEventRecord(b->tev0, s);
    const int n_wg = launch_track(b->d_descs, tp, d_results, nullptr, b->n_pairs, b->d_mail, &b->mail_epoch, b->cluster, d_resident, s);
    if (b->tev1) (void)hipEventRecord(b->tev1, s);
    b->tev0 = b->tev1 = nullptr;
    return n_wg;
  });
  if (rc) return rc;
  return batch_mark_tracker(b, s);
}
// the next revo_batch_track_only of this batch records ev0 / ev1 (hipEvent_t) directly around its grid
extern "C" void revo_batch_time_next_grid_(revo_batch* b, void* ev0, void* ev1) {
  if (b) { b->tev0 = (hipEvent_t)ev0; b->tev1 = (hipEvent_t)ev1; }
}

// Same as revo_batch_build for raw 16-bit depth (the reference's on-disk format): the conversion
// depth = raw * (float)(1/scale) of iowrapperRGBD.cpp:326-327 runs inside the first build kernel.
extern "C" int revo_batch_build_u16(revo_batch* b, const uint8_t* d_bgr, const uint16_t* d_depth_raw,
                                    double depth_scale_factor, void* stream) {
  if (!b || !d_bgr || !d_depth_raw) return fail(REVO_ERR_INVALID_ARG, "null argument");
  if (!(depth_scale_factor > 0.0)) return fail(REVO_ERR_INVALID_ARG, "depth_scale_factor must be positive");
  HIPCHECK(hipSetDevice(b->ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : b->stream;
  if (b->fs->has_aux) HIPCHECK(hipStreamWaitEvent(s, b->ev_join, 0));
  { int rc = wait_edt_before_rebuild(b->fs, s); if (rc) return rc; }
  { int rc = batch_wait_tracker(b, s); if (rc) return rc; }
  enqueue_build(b->ctx, b->fs, d_bgr, nullptr, d_depth_raw, (float)(1.0f / depth_scale_factor), s, false, false, 0, -1,
                b->side || b->defer < 3, batch_splits_depth(b), !batch_defers_fill(b));
  { int rc = enqueue_batch_tail(b, s); if (rc) return rc; }
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipEventRecord(b->fs->ev_ready, s));  // accessors / single-pair calls on the batch's views wait for this
  b->fs->has_ready = true;
  b->fs->ready_stream = s;
  for (auto& v : b->views) { v.table_built = false; v.ref_list_built = false; }
  return REVO_OK;
}

extern "C" int revo_batch_track(revo_batch* b, const uint8_t* d_bgr, const float* d_depth, const float* h_init_RT,
                                revo_pair_result* d_results, void* stream) {
  int rc = revo_batch_build(b, d_bgr, d_depth, stream);
  if (rc) return rc;
  return revo_batch_track_only(b, h_init_RT, d_results, stream);
}

extern "C" int revo_batch_sync(revo_batch* b, void* stream) {
  if (!b) return fail(REVO_ERR_INVALID_ARG, "null batch");
  HIPCHECK(hipSetDevice(b->ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : b->stream;
  HIPCHECK(hipStreamSynchronize(s));
  if (b->side) HIPCHECK(hipStreamSynchronize(b->side));  // (the keyframes' EDT of the last build)
  if (b->fs->edt_pending) { int rc2 = run_pending_edt(b->ctx, b->fs, s); if (rc2) return rc2; HIPCHECK(hipStreamSynchronize(s)); }
  // A record with bit 3 carries no pose (its workgroups could not exchange partial sums: device shared with
  // another process): that is an error of the call, not something to find by decoding flags.
  if (b->last_results) {
    //  the grid that wrote these records may have run on ANOTHER stream than s: finish it before reading
    { int rc3 = batch_wait_tracker(b, s); if (rc3) return rc3; }
    HIPCHECK(hipMemcpyAsync(b->h_flags, b->last_results, sizeof(revo_pair_result) * b->n_pairs, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    b->last_results = nullptr;  // decoded once: the caller may free or reuse that buffer after this call
    for (int i = 0; i < b->n_pairs; ++i)
      if (b->h_flags[i].flags & 8)
        return fail(REVO_ERR_HIP, "tracker: pair " + std::to_string(i) + ": the workgroups of the pair could not exchange "
                    "partial sums in time (device shared with another process?) -- its pose is not valid");
  }
  return REVO_OK;
}

// ------------------------------------------------------- host-buffer batches --
struct revo_pairs_job {
  revo_ctx* ctx = nullptr;
  int n = 0, is_u16 = 0;
  bool busy = false;
  revo_batch* batch = nullptr;
  uint8_t* d_bgr = nullptr;   // [2n][H][W][3]
  void* d_depth = nullptr;    // [2n][H][W] f32 or u16
  revo_pair_result* d_res = nullptr;
  revo_pair_result* h_res = nullptr;  // pinned
  float* h_init = nullptr;            // n x 12
  hipEvent_t ev_h2d = nullptr, ev_h2d2 = nullptr, ev_done = nullptr;
  hipStream_t stream = nullptr;       // compute stream of the job (alternates, so job k+1's build overlaps job k's tracker)
};

static void pairs_job_free(revo_pairs_job* j) {
  if (!j) return;
  if (j->batch) revo_batch_destroy(j->batch);
  (void)hipFree(j->d_bgr); (void)hipFree(j->d_depth); (void)hipFree(j->d_res);
  (void)hipHostFree(j->h_res);
  delete[] j->h_init;
  if (j->ev_h2d) (void)hipEventDestroy(j->ev_h2d);
  if (j->ev_h2d2) (void)hipEventDestroy(j->ev_h2d2);
  if (j->ev_done) (void)hipEventDestroy(j->ev_done);
  delete j;
}

static void pairs_jobs_release(revo_ctx* c) {
  std::vector<revo_pairs_job*> jobs;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    jobs.swap(c->jobs);
  }
  (void)hipSetDevice(c->device);
  for (revo_pairs_job* j : jobs) {
    if (j->busy && j->ev_done) (void)hipEventSynchronize(j->ev_done);
    pairs_job_free(j);
  }
}

extern "C" int revo_track_pairs_submit(revo_ctx* c, int n, const revo_pair_in* pairs, int depth_is_u16,
                                       double depth_scale_factor, revo_pairs_job** job_out) {
  if (!c || !pairs || !job_out || n <= 0) return fail(REVO_ERR_INVALID_ARG, "bad argument");
  if (depth_is_u16 && !(depth_scale_factor > 0.0)) return fail(REVO_ERR_INVALID_ARG, "depth_scale_factor must be positive");
  HIPCHECK(hipSetDevice(c->device));
  const int w = c->geom.lv[0].w, h = c->geom.lv[0].h;
  const size_t npix = (size_t)w * h, dsz = depth_is_u16 ? 2 : 4;
  for (int i = 0; i < n; ++i) {
    const revo_pair_in& p = pairs[i];
    if (!p.ref_bgr || !p.ref_depth || !p.cur_bgr || !p.cur_depth) return fail(REVO_ERR_INVALID_ARG, "null image pointer");
    if (p.ref_bgr_stride < (size_t)w * 3 || p.cur_bgr_stride < (size_t)w * 3 || p.ref_depth_stride < w * dsz || p.cur_depth_stride < w * dsz)
      return fail(REVO_ERR_INVALID_ARG, "stride smaller than a row");
  }
  revo_pairs_job* j = nullptr;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->copy_stream) {
      HIPCHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
      HIPCHECK(hipStreamCreateWithFlags(&c->copy_stream2, hipStreamNonBlocking));
      HIPCHECK(hipStreamCreateWithFlags(&c->pair_streams[0], hipStreamNonBlocking));
      HIPCHECK(hipStreamCreateWithFlags(&c->pair_streams[1], hipStreamNonBlocking));
    }
    int in_flight = 0;
    for (revo_pairs_job* q : c->jobs) in_flight += q->busy ? 1 : 0;
    if (in_flight >= 3) return fail(REVO_ERR_CAPACITY, "three jobs are in flight: call revo_track_pairs_wait first");
    for (revo_pairs_job* q : c->jobs)
      if (!q->busy && q->n == n && q->is_u16 == (depth_is_u16 ? 1 : 0)) { j = q; break; }
    if (!j) {
      for (size_t k = 0; k < c->jobs.size(); ++k)  // a free slot of another shape makes room (HBM is finite)
        if (!c->jobs[k]->busy && c->jobs.size() >= 3) { pairs_job_free(c->jobs[k]); c->jobs.erase(c->jobs.begin() + k); break; }
      j = new revo_pairs_job();
      j->ctx = c; j->n = n; j->is_u16 = depth_is_u16 ? 1 : 0;
      struct Guard { revo_pairs_job* j; ~Guard() { if (j) pairs_job_free(j); } } guard{j};
      int rc = revo_batch_create(c, n, &j->batch);
      if (rc) return rc;
      HIPCHECK(hipMalloc((void**)&j->d_bgr, 2 * (size_t)n * npix * 3));
      HIPCHECK(hipMalloc(&j->d_depth, 2 * (size_t)n * npix * dsz));
      HIPCHECK(hipMalloc((void**)&j->d_res, sizeof(revo_pair_result) * n));
      HIPCHECK(hipHostMalloc((void**)&j->h_res, sizeof(revo_pair_result) * n));
      j->h_init = new float[12 * (size_t)n];
      HIPCHECK(hipEventCreateWithFlags(&j->ev_h2d, hipEventDisableTiming));
      HIPCHECK(hipEventCreateWithFlags(&j->ev_h2d2, hipEventDisableTiming));
      HIPCHECK(hipEventCreateWithFlags(&j->ev_done, hipEventDisableTiming));
      guard.j = nullptr;
      c->jobs.push_back(j);
    }
    j->busy = true;
    j->stream = c->pair_streams[c->jobs_submitted++ & 1];
  }
  // H2D straight from the caller's rows (no host-side staging copy): one copy per plane (strided only when the
  // rows are padded), colour planes and depth planes on two streams so that two DMA engines work
  // HIP multiplexes its streams over a few hardware queues, and a copy stream that shares a queue with a compute stream
  // waits behind that stream's kernels: that, not the link, is why the f32 path reaches 20-24 GB/s where the u16 path
  // reaches 40 (profiles/r03_host_buffer_streams.txt: GPU_MAX_HW_QUEUES=8 lifts f32 to 45 GB/s -- and costs the pipelined
  // device-buffer batches a third of their throughput, so the library does not ask for it; swapping the planes' streams
  // helped in one order of events and not in another: the mapping depends on the process's stream history).
  // REVO_H2D_STREAMS: 1 = one copy stream for both planes, 2 = planes swapped (experiments).
  const int h2d_mode = c->knobs.h2d_mode;
  hipStream_t cs = c->copy_stream, cs2 = h2d_mode == 1 ? c->copy_stream : c->copy_stream2;  // cs: colour, cs2: depth
  if (h2d_mode == 2) std::swap(cs, cs2);
  // Any failure below must not leave the slot busy for ever (three of those and the context only ever answers
  // REVO_ERR_CAPACITY), nor return while the DMA engines may still be reading the caller's buffers.
  struct SlotGuard {
    revo_ctx* c; revo_pairs_job* j; hipStream_t a, b;
    ~SlotGuard() {
      if (!j) return;
      (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
      if (j->stream) (void)hipStreamSynchronize(j->stream);
      std::lock_guard<std::mutex> lk(c->mu);
      j->busy = false;
    }
  } slot_guard{c, j, cs, cs2};
  bool any_init = false;
  // Frames that lie back to back in host memory (a decoder that fills one job-sized slab per plane type: [2n][H][W][3] colours,
  // [2n][H][W] depths) travel in ONE copy per run of adjacent frames instead of one per frame: the DMA engines reach the link's
  // rate only with large transfers (profiles/r03_h2d_chunk_rates.txt: 0.6 MB pieces 43 GB/s on two streams, 1.2 MB 53 GB/s; a
  // whole job's planes are 59 + 39 MB).  Frames that are not adjacent, or whose rows are padded, go one by one as before.
  // Runs are cut at 64 MB: measured (profiles/r06_h2d_run_sizes.txt) u16 depth 39 -> 49 GB/s, f32 43 -> 47 GB/s with the cut,
  // but 39 GB/s when a job's 78 MB f32 depth slab travels as ONE copy (it then monopolises its engine past the colour copy).
  struct Run { char* dst; const char* src; size_t bytes; };
  const size_t max_run = (size_t)c->knobs.h2d_max_run_mb << 20;  // REVO_H2D_MAX_RUN_MB: upper bound of one merged copy
  auto flush = [&](Run& r, hipStream_t st) -> hipError_t {
    if (!r.bytes) return hipSuccess;
    const hipError_t e = hipMemcpyAsync(r.dst, r.src, r.bytes, hipMemcpyHostToDevice, st);
    r.bytes = 0;
    return e;
  };
  auto upload = [&](Run& r, void* dst, const void* src, size_t src_stride, size_t row_bytes, hipStream_t st) -> hipError_t {
    if (src_stride != row_bytes) {  // padded rows: a strided copy of its own
      const hipError_t e = flush(r, st);
      if (e != hipSuccess) return e;
      return hipMemcpy2DAsync(dst, row_bytes, src, src_stride, row_bytes, h, hipMemcpyHostToDevice, st);
    }
    const size_t bytes = row_bytes * h;
    if (r.bytes && r.bytes + bytes <= max_run && r.src + r.bytes == (const char*)src && r.dst + r.bytes == (char*)dst) { r.bytes += bytes; return hipSuccess; }
    const hipError_t e = flush(r, st);
    if (e != hipSuccess) return e;
    r.dst = (char*)dst; r.src = (const char*)src; r.bytes = bytes;
    return hipSuccess;
  };
  Run run_c{nullptr, nullptr, 0}, run_d{nullptr, nullptr, 0};
  for (int i = 0; i < n; ++i) {
    const revo_pair_in& p = pairs[i];
    const uint8_t* bs[2] = {p.ref_bgr, p.cur_bgr};
    const size_t bst[2] = {p.ref_bgr_stride, p.cur_bgr_stride};
    const void* ds[2] = {p.ref_depth, p.cur_depth};
    const size_t dst[2] = {p.ref_depth_stride, p.cur_depth_stride};
    for (int k = 0; k < 2; ++k) {
      const size_t f = 2 * (size_t)i + k;
      HIPCHECK(upload(run_c, j->d_bgr + f * npix * 3, bs[k], bst[k], (size_t)w * 3, cs));
      HIPCHECK(upload(run_d, (char*)j->d_depth + f * npix * dsz, ds[k], dst[k], w * dsz, cs2));
    }
    float* q = j->h_init + 12 * (size_t)i;
    if (p.use_init) { memcpy(q, p.R_init, sizeof(float) * 9); memcpy(q + 9, p.T_init, sizeof(float) * 3); any_init = true; }
    else { const float I[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}; memcpy(q, I, sizeof(I)); }
  }
  HIPCHECK(flush(run_c, cs));
  HIPCHECK(flush(run_d, cs2));
  HIPCHECK(hipEventRecord(j->ev_h2d2, cs2));
  HIPCHECK(hipStreamWaitEvent(cs, j->ev_h2d2, 0));
  HIPCHECK(hipEventRecord(j->ev_h2d, cs));
  hipStream_t s = j->stream;
  HIPCHECK(hipStreamWaitEvent(s, j->ev_h2d, 0));
  int rc = depth_is_u16 ? revo_batch_build_u16(j->batch, j->d_bgr, (const uint16_t*)j->d_depth, depth_scale_factor, s)
                        : revo_batch_build_borrow(j->batch, j->d_bgr, (const float*)j->d_depth, s);  // the job's own staging
  if (!rc) rc = revo_batch_track_only(j->batch, any_init ? j->h_init : nullptr, j->d_res, s);
  if (rc) return rc;
  HIPCHECK(hipMemcpyAsync(j->h_res, j->d_res, sizeof(revo_pair_result) * n, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipEventRecord(j->ev_done, s));
  // the caller's buffers are free again once the DMA has read them; the kernels keep running
  HIPCHECK(hipEventSynchronize(j->ev_h2d));
  slot_guard.j = nullptr;
  *job_out = j;
  return REVO_OK;
}

extern "C" int revo_track_pairs_wait(revo_pairs_job* j, revo_pair_out* out) {
  if (!j || !j->busy) return fail(REVO_ERR_INVALID_ARG, "not a job in flight");
  HIPCHECK(hipSetDevice(j->ctx->device));
  hipError_t e = hipEventSynchronize(j->ev_done);
  int rc = REVO_OK;
  if (e != hipSuccess) rc = fail(REVO_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
  if (!rc) {
    if (out) memcpy(out, j->h_res, sizeof(revo_pair_result) * j->n);
    for (int i = 0; i < j->n && !rc; ++i)
      if (j->h_res[i].flags & 8)
        rc = fail(REVO_ERR_HIP, "tracker: pair " + std::to_string(i) + ": the workgroups of the pair could not exchange partial sums "
                  "in time (device shared with another process?) -- its pose is not valid");
  }
  std::lock_guard<std::mutex> lk(j->ctx->mu);
  j->busy = false;
  return rc;
}

extern "C" int revo_track_pairs(revo_ctx* c, int n, const revo_pair_in* pairs, int depth_is_u16, double depth_scale_factor,
                                revo_pair_out* out) {
  revo_pairs_job* j = nullptr;
  const int rc = revo_track_pairs_submit(c, n, pairs, depth_is_u16, depth_scale_factor, &j);
  if (rc) return rc;
  return revo_track_pairs_wait(j, out);
}

// Every kernel of the batch build ALONE, HIP events between the launches on one stream, mean of `reps` passes: the per-kernel
// table of bench.py's roofline (VERDICT r04 #6b: the kernel furthest below its roofline must be visible in the driver's record).
// Leaves the batch built exactly like revo_batch_build_borrow + revo_batch_prepare.
extern "C" int revo_batch_profile_build(revo_batch* b, const uint8_t* d_bgr, const float* d_depth, int reps, revo_stage_times* out) {
  if (!b || !d_bgr || !d_depth || !out || reps <= 0) return fail(REVO_ERR_INVALID_ARG, "bad argument");
  revo_ctx* c = b->ctx;
  HIPCHECK(hipSetDevice(c->device));
  hipStream_t s = b->stream;
  if (b->fs->has_aux) HIPCHECK(hipStreamWaitEvent(s, b->ev_join, 0));
  { int rc = wait_edt_before_rebuild(b->fs, s); if (rc) return rc; }
  { int rc = batch_wait_tracker(b, s); if (rc) return rc; }
  memset(out, 0, sizeof(*out));
  PyrGeom g = c->geom;
  g.frame0 = 0;
  FrameSet* fs = b->fs;
  const int B = fs->B, L = g.n_levels;
  fs->p.depth[0] = const_cast<float*>(d_depth);
  struct Stage { const char* name; int a, w; };
  std::vector<Stage> st;
  // (the order a pipelined batch runs them in: the gray half of the pyramid, Canny, fill-in on the build stream; the depth half,
  // the edge lists and the keyframes' EDT on the first consumer's)
  st.push_back({"k_gray_depth", 0, 0});
  for (int l = 1; l < L; ++l) st.push_back({"k_pyrdown_gray", l, 1});
  st.push_back({"k_canny_nms4", 0, 0});
  st.push_back({"hysteresis", 0, 0});
  st.push_back({"k_fill", 0, 0});
  // (as a pipelined batch runs it: the depth half stages the edge pixels' depths when the knob is on)
  const bool stage = c->knobs.stage_edge_depths && fs->p.stage[0] != nullptr && L > 1;
  PyrGeom gp = g;
  gp.pts_staged = stage ? 1 : 0;
  if (stage) st.push_back({"k_edge_prefix", 0, 0});
  for (int l = 1; l < L; ++l) st.push_back({"k_pyrdown_depth", l, 2});
  st.push_back({"k_tile_count", 0, 1});
  st.push_back({"k_pts_tiles", 0, 2});
  st.push_back({"k_edt_cols", 0, 1});
  st.push_back({"k_edt_rows", 0, 2});
  const int n = (int)st.size();
  if (n > REVO_MAX_STAGES) return fail(REVO_ERR_CAPACITY, "too many stages");
  struct Events {  // destroyed on every exit path (a HIPCHECK inside the loop used to leak them)
    std::vector<hipEvent_t> v;
    ~Events() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); }
  } evs;
  evs.v.assign(n + 1, nullptr);
  std::vector<hipEvent_t>& ev = evs.v;
  for (auto& e : ev) HIPCHECK(hipEventCreate(&e));
  std::vector<double> sum(n, 0.0);
  for (int r = 0; r < reps; ++r) {
    HIPCHECK(hipEventRecord(ev[0], s));
    for (int i = 0; i < n; ++i) {
      const std::string nm = st[i].name;
      if (nm == "k_gray_depth") launch_gray_depth(g, fs->p, d_bgr, d_depth, nullptr, 0.f, B, s);
      else if (nm == "k_pyrdown_gray") launch_pyrdown(g, fs->p, st[i].a, B, s, 1);
      else if (nm == "k_pyrdown_depth") launch_pyrdown(g, fs->p, st[i].a, B, s, 2, stage);
      else if (nm == "k_edge_prefix") launch_edge_prefix(g, fs->p, B, s);
      else if (nm == "k_canny_nms4") launch_canny_nms(g, fs->p, B, s);
      else if (nm == "hysteresis") launch_hyst(g, fs->p, B, s);
      else if (nm == "k_fill") launch_fill(g, fs->p, B, s);
      else if (nm == "k_tile_count" || nm == "k_pts_tiles") launch_tile_points(gp, fs->p, B, s, st[i].w);
      else launch_keyframe(g, fs->p, 0, 2, b->n_pairs, s, st[i].w);
      HIPCHECK(hipGetLastError());
      HIPCHECK(hipEventRecord(ev[i + 1], s));
    }
    HIPCHECK(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) {
      float ms = 0.f;
      HIPCHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
      sum[i] += ms * 1e3;
    }
  }
  out->n = n;
  for (int i = 0; i < n; ++i) {
    out->us[i] = (float)(sum[i] / reps);
    if (std::string(st[i].name).rfind("k_pyrdown", 0) == 0) snprintf(out->name[i], sizeof(out->name[i]), "%s[%d]", st[i].name, st[i].a);
    else snprintf(out->name[i], sizeof(out->name[i]), "%s", st[i].name);
  }
  {  // the state a build + prepare leaves behind
    std::lock_guard<std::mutex> lk(fs->edt_mu);
    fs->edt_pending = false; fs->pts_pending = false; fs->depth_pending = false; fs->hyst_pending = false; fs->fill_pending = false;
    HIPCHECK(hipEventRecord(fs->ev_edt, s));
    fs->has_edt = true; fs->edt_stream = s;
  }
  HIPCHECK(hipEventRecord(fs->ev_ready, s));
  fs->has_ready = true; fs->ready_stream = s;
  fs->has_aux = false;
  for (auto& v : b->views) { v.table_built = false; v.ref_list_built = false; }
  return REVO_OK;
}

extern "C" int revo_batch_frame(revo_batch* b, int frame, revo_pyr** out) {
  if (!b || !out || frame < 0 || frame >= 2 * b->n_pairs) return fail(REVO_ERR_INVALID_ARG, "bad frame index");
  *out = &b->views[frame];
  return REVO_OK;
}

extern "C" int revo_batch_time_tracker(revo_batch* b, const float* h_init_RT, revo_pair_result* d_results, void* stream,
                                       int reps, float* ms_mean) {
  if (!b || !d_results || !ms_mean || reps <= 0) return fail(REVO_ERR_INVALID_ARG, "bad argument");
  HIPCHECK(hipSetDevice(b->ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : b->stream;
  // an earlier grid of this batch on another stream still reads the descriptors and the mailbox
  { int rc0 = batch_wait_tracker(b, s); if (rc0) return rc0; }
  int rc = batch_upload_init(b, h_init_RT, s);
  if (rc) return rc;
  TrackParams tp = b->ctx->tp;
  tp.eval_only = 0;
  { int rc1 = batch_wait_build(b, s); if (rc1) return rc1; }
  if (b->fs->has_aux) HIPCHECK(hipStreamWaitEvent(s, b->ev_join, 0));
  { int rc2 = run_pending_edt(b->ctx, b->fs, s); if (rc2) return rc2; }
  float total = 0.f;
  for (int r = 0; r < reps; ++r) {  // events bracket exactly one kernel on its own stream
    HIPCHECK(hipEventRecord(b->ev0, s));
    rc = chained_track_launch(b->ctx->device, b->ctx->knobs.track_depth, s, [&](unsigned* d_resident) {
      return launch_track(b->d_descs, tp, d_results, nullptr, b->n_pairs, b->d_mail, &b->mail_epoch, b->cluster, d_resident, s);
    });
    if (rc) return rc;
    HIPCHECK(hipEventRecord(b->ev1, s));
    HIPCHECK(hipEventSynchronize(b->ev1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    total += ms;
  }
  *ms_mean = total / (float)reps;
  return batch_mark_tracker(b, s);
}
