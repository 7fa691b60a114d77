// revo_vo.hip -- REVO::start sequencing (system/system.cpp:84-305) on top of the C ABI.
// Host-only code: the device work is what revo_pyramid_* / revo_tracker_* enqueue.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <condition_variable>
#include <mutex>

#include "../../include/revo_hip.h"
#include "revo_mat4.h"

extern "C" void revo_ctx_retain_(revo_ctx*);
extern "C" void revo_ctx_release_(revo_ctx*);
extern "C" void revo_tracker_reset_past_(revo_ctx*);
extern "C" int revo_ctx_reserve_framesets_(revo_ctx*, int total);
// split single-pair calls (revo_host.hip): launch now, read the result later
extern "C" int revo_track_launch_(revo_ctx*, const revo_pyr* ref, const revo_pyr* curr, const float R[9], const float T[3], int slot,
                                  unsigned* seq_out);
extern "C" int revo_track_wait_(revo_ctx*, int slot, unsigned seq, float R[9], float T[3], float* err, int* status);
extern "C" int revo_assess_launch_(revo_ctx*, const float T_w_curr[16], const revo_pyr* curr, int* nframes_out, unsigned* seq_out);
extern "C" int revo_assess_wait_(revo_ctx*, int nframes, unsigned seq, int* status, float* ratio_out);
extern "C" int revo_vote_overlaps_(const revo_ctx*);
extern "C" int revo_pyramid_prepare_keyframe_(revo_pyr*);
extern "C" void revo_debug_section_note_(int i, unsigned long long ns);

namespace {
struct M4 {  // column-major 4x4, Eigen::Matrix4f storage
  float m[16];
  static M4 identity() { M4 o; memset(o.m, 0, sizeof(o.m)); o.m[0] = o.m[5] = o.m[10] = o.m[15] = 1.f; return o; }
};
M4 mul(const M4& A, const M4& B) { M4 o; mat4_mul(A.m, B.m, o.m); return o; }
M4 inverse(const M4& A) { M4 o; mat4_inverse(A.m, o.m); return o; }  // Eigen Matrix4f::inverse()
M4 from_RT(const float* R, const float* T) {  // transformFromRT
  M4 o = M4::identity();
  for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) o.m[c * 4 + r] = R[c * 3 + r];
  o.m[12] = T[0]; o.m[13] = T[1]; o.m[14] = T[2];
  return o;
}
void to_RT(const M4& M, float* R, float* T) {
  for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) R[c * 3 + r] = M.m[c * 4 + r];
  T[0] = M.m[12]; T[1] = M.m[13]; T[2] = M.m[14];
}
struct Pose { M4 T_kf_curr, T_w_kf; M4 world() const { return mul(T_w_kf, T_kf_curr); } };  // REVO::Pose, system.h:89-152
struct Frame { revo_pyr* pyr; double ts; M4 T_w_f; };
}  // namespace

struct revo_vo {
  revo_ctx* ctx;
  std::deque<Frame> queue;  // mPyrQueue, iowrapperRGBD.h:166-167
  std::mutex qmu;           // mtx of the reference's queue: submit may run on an IO thread (system.cpp:96)
  std::condition_variable qcv;  // queue became non-empty / got room / was closed
  int max_queue = 0;        // > 0: submit blocks while this many pyramids wait (back-pressure on the IO thread)
  bool closed = false;      // the producer announced the end of the stream (iowrapperRGBD's hasMoreImages() == false)
  Frame kf{nullptr, 0, M4::identity()}, prev{nullptr, 0, M4::identity()};
  Pose last, before_last;
  M4 T_NM1_N = M4::identity();
  float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, T[3] = {0, 0, 0};
  int no_frames = 0, n_keyframes = 0;
  bool just_added_kf = false;
  int hist_level = 2;
  // look-ahead: the tracker of the NEXT queued frame is launched while this frame's quality vote is still running,
  // on the assumption that the vote keeps the keyframe (it does for all but a few frames per hundred); a keyframe
  // change simply ignores that launch.  Same calls, same arguments, same results as the one-at-a-time order.
  bool spec_valid = false;
  revo_pyr* spec_pyr = nullptr;
  int spec_slot = 0;
  unsigned spec_seq = 0;
  int slot = 0;  // result slot of the next non-speculative launch
  // ... unless the PREVIOUS vote was close to asking for a new keyframe (overlap ratio below kf_guard; the ratio decays smoothly
  // from ~2.5 behind a keyframe to 1, where the change happens): then the look-ahead is held back until this frame's vote is in --
  // ~35 us later when the keyframe stays, and no tracker against the old keyframe (190 us on the stream the re-track needs) when
  // it changes.  Launch timing only: every launch that counts has the same arguments either way.  REVO_KF_GUARD=0 switches it off.
  float last_ratio = INFINITY;
  float kf_guard = 1.08f;
};

extern "C" int revo_vo_create(revo_ctx* ctx, revo_vo** out) {
  if (!ctx || !out) return REVO_ERR_INVALID_ARG;
  revo_vo* v = new revo_vo();
  v->ctx = ctx;
  revo_ctx_retain_(ctx);
  v->hist_level = revo_ctx_histogram_level(ctx);
  { const char* e = getenv("REVO_KF_GUARD"); if (e && *e) v->kf_guard = (float)atof(e); }
  // REVO::start makes a fresh TrackerNew (system.cpp:107): a driver never votes against the clouds
  // another driver on the same context left behind
  revo_tracker_reset_past_(ctx);
  *out = v;
  return REVO_OK;
}
extern "C" void revo_vo_destroy(revo_vo* v) {
  if (!v) return;
  for (auto& f : v->queue) revo_pyramid_destroy(f.pyr);
  if (v->kf.pyr && v->kf.pyr != v->prev.pyr) revo_pyramid_destroy(v->kf.pyr);
  if (v->prev.pyr) revo_pyramid_destroy(v->prev.pyr);
  revo_ctx_release_(v->ctx);
  delete v;
}
extern "C" int revo_vo_queued(const revo_vo* v) {
  if (!v) return 0;
  std::lock_guard<std::mutex> lk(const_cast<revo_vo*>(v)->qmu);
  return (int)v->queue.size();
}
extern "C" int revo_vo_num_keyframes(const revo_vo* v) { return v ? v->n_keyframes : 0; }
// Hand-off between the IO thread and the consumer loop inside the library (the reference: mPyrQueue + its
// mutex, hasMoreImages()).  max_queue > 0 makes submit block while that many pyramids wait.
extern "C" int revo_vo_set_max_queue(revo_vo* v, int max_queue) {
  if (!v) return REVO_ERR_INVALID_ARG;
  { std::lock_guard<std::mutex> lk(v->qmu); v->max_queue = max_queue; v->closed = false; }  // (re-)opens the stream
  v->qcv.notify_all();
  // the frames a run holds at once: the queue, the one being submitted, the one being tracked, the previous one, the keyframe
  // and one being recycled -- reserved now, not grown frame by frame during the first run
  if (max_queue > 0) return revo_ctx_reserve_framesets_(v->ctx, max_queue + 5);
  return REVO_OK;
}
// the producer has no more frames: revo_vo_wait_frame returns 0 once the queue has drained
extern "C" int revo_vo_close(revo_vo* v) {
  if (!v) return REVO_ERR_INVALID_ARG;
  { std::lock_guard<std::mutex> lk(v->qmu); v->closed = true; }
  v->qcv.notify_all();
  return REVO_OK;
}
// blocks until a frame is queued (returns 1) or the stream is closed and drained (returns 0)
extern "C" int revo_vo_wait_frame(revo_vo* v) {
  if (!v) return REVO_ERR_INVALID_ARG;
  std::unique_lock<std::mutex> lk(v->qmu);
  v->qcv.wait(lk, [&] { return !v->queue.empty() || v->closed; });
  return v->queue.empty() ? 0 : 1;
}

// kfPyr and kfPyr->getTransKFtoWorld() (system.cpp:165-167,235-237): what the viewer / model export consume
extern "C" int revo_vo_keyframe(const revo_vo* v, revo_pyr** kf_out, float T_w_kf[16]) {
  if (!v || !v->kf.pyr) return REVO_ERR_INVALID_ARG;
  if (kf_out) *kf_out = v->kf.pyr;
  if (T_w_kf) memcpy(T_w_kf, v->kf.T_w_f.m, sizeof(float) * 16);
  return REVO_OK;
}

// IOWrapperRGBD::generateImgPyramidFromFiles: new ImgPyramidRGBD(...) -> queue (iowrapperRGBD.cpp:279-288)
extern "C" int revo_vo_submit(revo_vo* v, const uint8_t* bgr, size_t bgr_stride, const float* depth, size_t depth_stride,
                              double ts) {
  if (!v) return REVO_ERR_INVALID_ARG;
  Frame f{nullptr, ts, M4::identity()};
  const int rc = revo_pyramid_create(v->ctx, bgr, bgr_stride, depth, depth_stride, ts, &f.pyr);
  if (rc) return rc;
  {
    const auto tq = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> lk(v->qmu);
    v->qcv.wait(lk, [&] { return v->max_queue <= 0 || (int)v->queue.size() < v->max_queue; });
    v->queue.push_back(f);
    revo_debug_section_note_(5, (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tq).count());
  }
  v->qcv.notify_all();
  return REVO_OK;
}

extern "C" int revo_vo_submit_u16(revo_vo* v, const uint8_t* bgr, size_t bgr_stride, const uint16_t* depth_raw,
                                  size_t depth_stride, double depth_scale_factor, double ts) {
  if (!v) return REVO_ERR_INVALID_ARG;
  Frame f{nullptr, ts, M4::identity()};
  const int rc = revo_pyramid_create_u16(v->ctx, bgr, bgr_stride, depth_raw, depth_stride, depth_scale_factor, ts, &f.pyr);
  if (rc) return rc;
  {
    std::unique_lock<std::mutex> lk(v->qmu);
    v->qcv.wait(lk, [&] { return v->max_queue <= 0 || (int)v->queue.size() < v->max_queue; });
    v->queue.push_back(f);
  }
  v->qcv.notify_all();
  return REVO_OK;
}

// one body of the while loop of REVO::start (system.cpp:128-284)
extern "C" int revo_vo_track_next(revo_vo* v, float pose_out[16], int* new_kf_out, double* ts_out) {
  if (!v) return REVO_ERR_INVALID_ARG;
  Frame curr;
  {
    std::lock_guard<std::mutex> lk(v->qmu);
    if (v->queue.empty()) return REVO_ERR_INVALID_ARG;
    curr = v->queue.front();
    v->queue.pop_front();
  }
  v->qcv.notify_all();
  int rc, new_kf = 0, status = 0;
  const M4 I = M4::identity();
  if (ts_out) *ts_out = curr.ts;
  if (v->no_frames == 0) {  // system.cpp:151-175
    v->kf = v->prev = curr;
    if ((rc = revo_pyramid_make_keyframe(v->kf.pyr))) return rc;
    v->kf.T_w_f = I;
    v->prev.T_w_f = I;
    v->last = Pose{I, I};
    ++v->n_keyframes;
    ++v->no_frames;
    v->just_added_kf = true;
    if ((rc = revo_tracker_add_old_pcl(v->ctx, v->kf.pyr, v->hist_level, I.m, curr.ts))) return rc;
    if (pose_out) memcpy(pose_out, I.m, sizeof(I.m));
    if (new_kf_out) *new_kf_out = 1;
    return REVO_OK;
  }
  ++v->no_frames;
  float err = 0.f;
  // trackFrames(kf, curr): already in flight if the previous call looked ahead
  unsigned seq = 0;
  int slot = v->slot;
  if (v->spec_valid && v->spec_pyr == curr.pyr) {
    slot = v->spec_slot; seq = v->spec_seq;
  } else {
    if ((rc = revo_track_launch_(v->ctx, v->kf.pyr, curr.pyr, v->R, v->T, slot, &seq))) return rc;
  }
  v->spec_valid = false;
  if ((rc = revo_track_wait_(v->ctx, slot, seq, v->R, v->T, &err, &status))) return rc;
  M4 T_KF_N = from_RT(v->R, v->T);
  M4 currPoseInWorld = mul(v->kf.T_w_f, T_KF_N);
  int nvote = -1;
  unsigned vseq = 0;
  // The vote has a stream of its own (revo_host.hip, vote_stream): the look-ahead tracker goes out FIRST, so that it starts one
  // host hop behind the tracker that just finished instead of behind the vote's two launches as well.  With REVO_VOTE_STREAM=0
  // both share the tracker stream and the vote must stay in front (its answer is waited for below).
  const bool spec_first = revo_vote_overlaps_(v->ctx) != 0;
  // (the previous vote was close to a keyframe change and this frame's vote counts -- the one right behind a new keyframe is
  // ignored, system.cpp:203: hold the look-ahead until the vote is in)
  const bool hold = spec_first && !v->just_added_kf && v->last_ratio < v->kf_guard;
  if ((!spec_first || hold) && (rc = revo_assess_launch_(v->ctx, currPoseInWorld.m, curr.pyr, &nvote, &vseq))) return rc;
  // what the loop body does when the vote says OK (system.cpp:243-271), computed now so that the next frame's
  // tracker can start behind the vote kernels instead of behind a host round trip
  const Pose ok_last{T_KF_N, v->kf.T_w_f};
  const M4 ok_w1 = ok_last.world();
  const M4 ok_T_NM1_N = mul(inverse(v->last.world()), ok_w1);
  float ok_R[9], ok_T[3];
  to_RT(mul(ok_last.T_kf_curr, ok_T_NM1_N), ok_R, ok_T);
  auto look_ahead = [&]() {
    revo_pyr* nxt = nullptr;
    {
      std::lock_guard<std::mutex> lk(v->qmu);
      if (!v->queue.empty()) nxt = v->queue.front().pyr;
    }
    if (nxt) {
      const int sslot = slot ^ 1;
      unsigned sseq = 0;
      if (revo_track_launch_(v->ctx, v->kf.pyr, nxt, ok_R, ok_T, sslot, &sseq) == REVO_OK) {
        v->spec_valid = true; v->spec_pyr = nxt; v->spec_slot = sslot; v->spec_seq = sseq;
      }
    }
  };
  if (!hold) look_ahead();
  // held back: the tracker stream is idle while the vote runs -- the distance transforms of the frame that would become the keyframe
  // (the previous one, system.cpp:205-215) go there now; makeKeyframe below finds them done
  else if (v->prev.pyr && v->prev.pyr != v->kf.pyr) (void)revo_pyramid_prepare_keyframe_(v->prev.pyr);
  if (spec_first && !hold && (rc = revo_assess_launch_(v->ctx, currPoseInWorld.m, curr.pyr, &nvote, &vseq))) return rc;
  float ratio = INFINITY;
  if ((rc = revo_assess_wait_(v->ctx, nvote, vseq, &status, &ratio))) return rc;
  v->last_ratio = ratio;
  const bool change_kf = status == REVO_TRACKER_STATE_NEW_KF && !v->just_added_kf;
  if (hold && !change_kf) look_ahead();  // the keyframe stays: the held-back tracker goes out now
  if (change_kf) {  // system.cpp:203-241
    v->spec_valid = false;  // the look-ahead tracked against the old keyframe: ignored
    revo_pyr* old_kf = v->kf.pyr;
    v->kf = v->prev;
    v->kf.T_w_f = v->last.world();  // kfPyr->setTwf(mPoseGraph.back().getCurrToWorld())
    if ((rc = revo_pyramid_make_keyframe(v->kf.pyr))) return rc;
    v->last = Pose{I, v->kf.T_w_f};  // mPoseGraph.back().setKfFrame(kfPyr)
    ++v->n_keyframes;
    if ((rc = revo_tracker_clear_past(v->ctx))) return rc;
    to_RT(v->T_NM1_N, v->R, v->T);
    if ((rc = revo_tracker_track_frames(v->ctx, v->kf.pyr, curr.pyr, v->R, v->T, &err, &status, nullptr, nullptr))) return rc;
    T_KF_N = from_RT(v->R, v->T);
    currPoseInWorld = mul(v->kf.T_w_f, T_KF_N);
    if (spec_first) {
      // the next frame's tracker goes out before the second vote (nobody acts on its answer: just_added_kf) -- its initialisation
      // is what the end of this body computes from the same matrices in the same order (system.cpp:267-271)
      const Pose nl{T_KF_N, v->kf.T_w_f};
      to_RT(mul(nl.T_kf_curr, mul(inverse(v->last.world()), nl.world())), ok_R, ok_T);
      look_ahead();
    }
    if ((rc = revo_tracker_assess_quality(v->ctx, currPoseInWorld.m, curr.pyr, &status, nullptr, nullptr))) return rc;
    v->just_added_kf = true;
    new_kf = 1;
    if (old_kf != v->kf.pyr) revo_pyramid_destroy(old_kf);
  } else {
    v->just_added_kf = false;
  }
  v->slot = v->spec_valid ? (v->spec_slot ^ 1) : 0;
  v->before_last = v->last;
  v->last = Pose{T_KF_N, v->kf.T_w_f};  // mPoseGraph.push_back(Pose(T_KF_N, ts, kfPyr))
  if ((rc = revo_tracker_add_old_pcl(v->ctx, curr.pyr, v->hist_level, currPoseInWorld.m, curr.ts))) return rc;
  // T_NM1_N = graph[size-2].T_N_W() * graph.back().T_W_N(); T_init = back().T_kf_N() * T_NM1_N (system.cpp:267-271)
  const M4 w1 = v->last.world();
  v->T_NM1_N = mul(inverse(v->before_last.world()), w1);
  to_RT(mul(v->last.T_kf_curr, v->T_NM1_N), v->R, v->T);
  if (pose_out) memcpy(pose_out, w1.m, sizeof(w1.m));
  if (new_kf_out) *new_kf_out = new_kf;
  if (v->prev.pyr != v->kf.pyr) revo_pyramid_destroy(v->prev.pyr);  // prevPyr = currPyr
  v->prev = curr;
  return REVO_OK;
}
