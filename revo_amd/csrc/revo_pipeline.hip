// revo_pipeline.hip -- the pipelined batch mode as ONE handle of the C ABI (include/revo_hip.h: revo_pipeline_*).
//
// The reference owns its producer/consumer pipeline (system/system.cpp:96,128-284: the IO thread fills mPyrQueue,
// REVO::start drains it; io/iowrapperRGBD.cpp:279-288).  The batched mode's equivalent is a rotation of `depth`
// resident batches over FOUR device streams:
//
//     build stream     : pyramids of step t            (gray, pyrDown, Canny NMS, hysteresis, fill-in)
//     auxiliary stream : what that build leaves to its first consumer (edge lists of all frames + the keyframes' EDT)
//     tracker stream 0/1: the tracker grids of consecutive steps alternate (the library's resident gate keeps two in flight)
//
// Up to round 4 this choreography lived in bench.py; an integrator who rebuilt it with one stream more or in another
// creation order silently lost a third of the throughput, because HIP multiplexes its streams onto a few hardware queues
// and two streams that share a queue serialise.  The handle therefore owns its streams, CHECKS at
// creation that they sit on distinct hardware queues (a 150 us probe kernel on one stream, a time stamp on another: the
// stamp lands after the probe's end iff the two share a queue) and replaces the ones that alias, and it hands the caller a
// defined slot for "work behind the grid" (the result collective, a D2H) on the grid's own stream instead of a fifth one.
//
// Host-only code on top of the batch entry points (revo_batch_build_* / _prepare / _track_only do the cross-stream
// ordering themselves: per-FrameSet "built" / "prepared" events, per-batch tracker event); the only kernel is the probe.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/revo_hip.h"

extern "C" void revo_ctx_retain_(revo_ctx*);
extern "C" void revo_ctx_release_(revo_ctx*);
extern "C" int revo_ctx_device_(const revo_ctx*);
extern "C" void revo_set_error_(const char* msg);
extern "C" void revo_batch_time_next_grid_(revo_batch*, void* ev0, void* ev1);

namespace {

int fail(int code, const std::string& msg) {
  revo_set_error_(msg.c_str());
  return code;
}
#define PCHECK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess)                                                                  \
      return fail(REVO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
  } while (0)

int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  const int v = atoi(e);
  return v < lo ? lo : (v > hi ? hi : v);
}

// out[0] = start, out[1] = end of this one-lane kernel on the device's constant-rate clock; spin_ticks > 0: stay that long
__global__ void __launch_bounds__(64) k_pipe_probe(unsigned long long* __restrict__ out, unsigned long long spin_ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  out[0] = t0;
  if (spin_ticks) {
    for (int guard = 0; guard < (1 << 22); ++guard) {  // bounded whatever the clock does
      if ((unsigned long long)wall_clock64() - t0 >= spin_ticks) break;
      __builtin_amdgcn_s_sleep(16);
    }
  }
  out[1] = (unsigned long long)wall_clock64();
}

struct Slot {  // one batch of the rotation
  revo_batch* batch = nullptr;
  hipEvent_t ev_done = nullptr;          // behind the grid AND whatever the caller enqueued in the after-grid slot
  unsigned long long ticket = 0;         // the step this batch holds (0 = none yet)
  int trk = 0;                           // tracker stream of that step
  bool done_recorded = true;             // ev_done has been recorded for `ticket`
  bool waited = true;                    // revo_pipeline_wait has returned this step's records (host_results)
  revo_pair_result* d_res = nullptr;     // library-owned records (when the caller passes none)
  revo_pair_result* h_res = nullptr;     // pinned copy (host_results)
  hipEvent_t t0 = nullptr, t1 = nullptr; // timing pair around this slot's grid
  bool timed = false;
};

}  // namespace

struct revo_pipeline {
  revo_ctx* ctx = nullptr;
  int device = 0;
  int n_pairs = 0, nb = 0, ntrk = 1;
  int host_results = 0;
  std::vector<Slot> slots;
  hipStream_t s_trk[2] = {nullptr, nullptr}, s_build = nullptr, s_aux = nullptr;
  unsigned long long submitted = 0;
  // probe outcome
  int distinct_queues = 0, streams_replaced = 0, probes = 0;
  // every stream this handle ever created, kept or discarded as aliasing, from the moment it exists: destroyed with the
  // pipeline (destroying an aliasing one earlier would hand its queue slot to the next candidate) -- an error path in between
  // leaks nothing
  std::vector<hipStream_t> owned;
  // the result collective (revo_pipeline_set_comm): windows of `every` steps, `ring` windows in rotation
  revo_comm* comm = nullptr;
  int every = 0, ring = 0, world = 1;
  revo_pair_result* d_send = nullptr;      // [ring][every][n_pairs]: the grids of a window write their records here
  revo_pair_result* d_gathered = nullptr;  // caller's: [ring][world][every][n_pairs]
  std::vector<hipEvent_t> ev_step;         // [ring * every]: behind the grid of a window's step (the collective waits for the steps on the other tracker stream)
  std::vector<hipEvent_t> ev_coll;         // [ring]: behind the collective that read send window w
  std::vector<char> coll_pending;          // [ring]
  unsigned long long collectives = 0;
  // live timing of the tracker grid (bench.py's roofline leg): every `time_every`-th submit carries an event pair
  int time_every = 0;
  double timed_ms = 0.0;
  int timed_n = 0;
  std::mutex mu;
};

namespace {

// true iff a kernel on `b` cannot start before an earlier-enqueued kernel on `a` has finished (same hardware queue)
int streams_alias(hipStream_t a, hipStream_t b, unsigned long long* d_buf, unsigned long long spin_ticks, bool* alias) {
  hipLaunchKernelGGL(k_pipe_probe, dim3(1), dim3(64), 0, a, d_buf, spin_ticks);
  hipLaunchKernelGGL(k_pipe_probe, dim3(1), dim3(64), 0, b, d_buf + 2, 0ull);
  PCHECK(hipGetLastError());
  PCHECK(hipStreamSynchronize(a));
  PCHECK(hipStreamSynchronize(b));
  unsigned long long h[4];
  PCHECK(hipMemcpy(h, d_buf, sizeof(h), hipMemcpyDeviceToHost));
  *alias = h[2] >= h[1];  // the stamp on b was taken after the probe on a had ended
  return REVO_OK;
}

// Pick streams on pairwise distinct hardware queues: candidates are created one at a time and kept only if they alias none
// of the streams kept so far (HIP hands its hardware queues to new streams in turn, so a replacement usually lands on the
// next queue).  With fewer hardware queues than `want` (GPU_MAX_HW_QUEUES < 4) the best set found is used and reported.
int pick_streams(revo_pipeline* p, int want, std::vector<hipStream_t>* out) {
  const bool probe = env_int("REVO_PIPE_PROBE", 1, 0, 1) != 0;
  unsigned long long* d_buf = nullptr;
  struct BufGuard { unsigned long long** p; ~BufGuard() { if (*p) (void)hipFree(*p); } } buf_guard{&d_buf};
  int rate_khz = 0;
  if (probe) {
    PCHECK(hipMalloc((void**)&d_buf, 4 * sizeof(unsigned long long)));
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, p->device) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
  }
  const unsigned long long spin = (unsigned long long)rate_khz * 150ull / 1000ull;  // 150 us
  int tries = 0;
  std::vector<hipStream_t> kept;
  std::vector<hipStream_t> aliased;  // in creation order: fallback when no distinct set exists
  while ((int)kept.size() < want && tries < want + 8) {
    ++tries;
    hipStream_t s = nullptr;
    PCHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    p->owned.push_back(s);
    bool bad = false;
    if (probe) {
      // first use of a stream binds it to its hardware queue: touch it before probing
      hipLaunchKernelGGL(k_pipe_probe, dim3(1), dim3(64), 0, s, d_buf, 0ull);
      PCHECK(hipStreamSynchronize(s));
      for (hipStream_t k : kept) {
        bool a1 = false, a2 = false;
        int rc = streams_alias(k, s, d_buf, spin, &a1);
        if (rc) return rc;
        ++p->probes;
        if (!a1) {  // (both directions must overlap: one-sided overlap would be launch-order luck)
          rc = streams_alias(s, k, d_buf, spin, &a2);
          if (rc) return rc;
          ++p->probes;
        }
        if (a1 || a2) { bad = true; break; }
      }
    }
    if (bad) aliased.push_back(s); else kept.push_back(s);
  }
  p->distinct_queues = (int)kept.size();
  p->streams_replaced = (int)aliased.size();
  if ((int)kept.size() < want)  // (say it once, loudly: the pipeline still works, its stages only stop overlapping)
    fprintf(stderr, "revo_pipeline: only %d of the %d streams sit on distinct hardware queues (GPU_MAX_HW_QUEUES < 4?): stages that share a "
                    "queue run one after the other\n", (int)kept.size(), want);
  // not enough distinct queues: fill up with aliasing streams (still correct, only slower) and say so in revo_pipeline_info
  while ((int)kept.size() < want && !aliased.empty()) { kept.push_back(aliased.front()); aliased.erase(aliased.begin()); }
  if (!probe) p->distinct_queues = -1;  // unknown
  *out = kept;
  return REVO_OK;
}

// The done event of the step a slot holds: recorded lazily, right before its tracker stream is used again (or when somebody
// waits for the step), so that it also covers what the caller enqueued behind the grid.
int finalize_slot(revo_pipeline* p, Slot& sl) {
  if (sl.ticket && !sl.done_recorded) {
    PCHECK(hipEventRecord(sl.ev_done, p->s_trk[sl.trk]));
    sl.done_recorded = true;
  }
  return REVO_OK;
}

int harvest_timing(revo_pipeline* p, Slot& sl) {
  if (!sl.timed) return REVO_OK;
  PCHECK(hipEventSynchronize(sl.t1));
  float ms = 0.f;
  PCHECK(hipEventElapsedTime(&ms, sl.t0, sl.t1));
  p->timed_ms += ms;
  p->timed_n += 1;
  sl.timed = false;
  return REVO_OK;
}

}  // namespace

extern "C" int revo_pipeline_create(revo_ctx* ctx, int n_pairs, int depth, int host_results, revo_pipeline** out) {
  if (!ctx || !out || n_pairs <= 0) return fail(REVO_ERR_INVALID_ARG, "bad argument");
  if (depth == 0) depth = 4;  // build(t+3) | edge lists + EDT(t+2) | tracker grids t+1 and t  (profiles/r04_ab_around_default.txt)
  if (depth < 1 || depth > 8) return fail(REVO_ERR_INVALID_ARG, "pipeline depth must be 1..8 (0 = default 4)");
  const int device = revo_ctx_device_(ctx);
  PCHECK(hipSetDevice(device));
  revo_pipeline* p = new revo_pipeline();
  p->ctx = ctx; p->device = device; p->n_pairs = n_pairs; p->nb = depth; p->host_results = host_results ? 1 : 0;
  revo_ctx_retain_(ctx);
  struct Guard { revo_pipeline* p; ~Guard() { if (p) revo_pipeline_destroy(p); } } guard{p};
  // the tracker streams first (the order the measured shape was created in), then build, then auxiliary
  if (depth == 1) {
    PCHECK(hipStreamCreateWithFlags(&p->s_trk[0], hipStreamNonBlocking));
    p->owned.push_back(p->s_trk[0]);
    p->s_trk[1] = p->s_build = p->s_aux = p->s_trk[0];
    p->ntrk = 1; p->distinct_queues = 1;
  } else {
    p->ntrk = depth >= 3 ? 2 : 1;
    const int want = p->ntrk + 2;
    std::vector<hipStream_t> st;
    int rc = pick_streams(p, want, &st);
    if (rc) return rc;
    if ((int)st.size() < want) return fail(REVO_ERR_HIP, "could not create the pipeline's streams");
    p->s_trk[0] = st[0];
    p->s_trk[1] = p->ntrk == 2 ? st[1] : st[0];
    p->s_build = st[p->ntrk];
    p->s_aux = st[p->ntrk + 1];
  }
  p->slots.resize(depth);
  for (Slot& sl : p->slots) {
    int rc = revo_batch_create(ctx, n_pairs, &sl.batch);
    if (rc) return rc;
    PCHECK(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
    PCHECK(hipEventCreate(&sl.t0));
    PCHECK(hipEventCreate(&sl.t1));
    PCHECK(hipMalloc((void**)&sl.d_res, sizeof(revo_pair_result) * n_pairs));
    PCHECK(hipMemset(sl.d_res, 0, sizeof(revo_pair_result) * n_pairs));
    if (p->host_results) PCHECK(hipHostMalloc((void**)&sl.h_res, sizeof(revo_pair_result) * n_pairs));
  }
  PCHECK(hipStreamSynchronize(nullptr));  // the zeroing above (NULL stream; the pipeline's streams are non-blocking)
  guard.p = nullptr;
  *out = p;
  return REVO_OK;
}

extern "C" void revo_pipeline_destroy(revo_pipeline* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  hipStream_t all[4] = {p->s_trk[0], p->s_trk[1], p->s_build, p->s_aux};
  for (int i = 0; i < 4; ++i) if (all[i]) (void)hipStreamSynchronize(all[i]);
  for (Slot& sl : p->slots) {
    if (sl.batch) revo_batch_destroy(sl.batch);
    if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    if (sl.t0) (void)hipEventDestroy(sl.t0);
    if (sl.t1) (void)hipEventDestroy(sl.t1);
    (void)hipFree(sl.d_res);
    if (sl.h_res) (void)hipHostFree(sl.h_res);
  }
  for (hipEvent_t e : p->ev_step) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->ev_coll) if (e) (void)hipEventDestroy(e);
  if (p->d_send) (void)hipFree(p->d_send);
  for (hipStream_t s : p->owned) (void)hipStreamDestroy(s);
  (void)hipGetLastError();
  revo_ctx* c = p->ctx;
  delete p;
  revo_ctx_release_(c);
}

extern "C" int revo_pipeline_submit(revo_pipeline* p, const uint8_t* d_bgr, const void* d_depth, int depth_kind,
                                    double depth_scale_factor, const float* h_init_RT, revo_pair_result* d_results,
                                    void* input_ready_event, uint64_t* ticket_out, void** after_grid_stream) {
  if (!p || !d_bgr || !d_depth) return fail(REVO_ERR_INVALID_ARG, "null argument");
  if (depth_kind < 0 || depth_kind > 2) return fail(REVO_ERR_INVALID_ARG, "depth_kind: 0 = f32 borrowed, 1 = f32 copied, 2 = u16 raw");
  std::lock_guard<std::mutex> lk(p->mu);
  //  without host results, without a communicator and without a caller buffer the records would land in a buffer no
  // call exposes: refuse instead of losing them silently
  if (!d_results && !p->host_results && !p->comm)
    return fail(REVO_ERR_INVALID_ARG, "d_results is NULL and the pipeline neither copies records to the host (host_results) nor gathers them "
                                      "(revo_pipeline_set_comm): the step's records would be unreachable");
  PCHECK(hipSetDevice(p->device));
  const unsigned long long t = p->submitted;
  Slot& sl = p->slots[t % p->nb];
  const int trk = (int)(t % p->ntrk);
  hipStream_t s_tr = p->s_trk[trk];
  if (p->host_results && !sl.waited)
    return fail(REVO_ERR_CAPACITY, "the step that holds this slot's records has not been waited for: call revo_pipeline_wait(ticket " +
                                       std::to_string(sl.ticket) + ") before submitting step " + std::to_string(t + 1));
  // the tracker stream is about to be reused: the step that ran on it last ends HERE (its after-grid slot is closed)
  for (Slot& o : p->slots)
    if (o.ticket && !o.done_recorded && o.trk == trk) { int rc = finalize_slot(p, o); if (rc) return rc; }
  { int rc = harvest_timing(p, sl); if (rc) return rc; }
  if (input_ready_event) PCHECK(hipStreamWaitEvent(p->s_build, (hipEvent_t)input_ready_event, 0));
  // (the batch's build orders itself behind the batch's previous tracker grid and deferred work: revo_host.hip)
  int rc;
  if (depth_kind == 2) {
    rc = revo_batch_build_u16(sl.batch, d_bgr, (const uint16_t*)d_depth, depth_scale_factor, p->s_build);
  } else if (depth_kind == 1) {
    rc = revo_batch_build(sl.batch, d_bgr, (const float*)d_depth, p->s_build);
  } else {
    rc = revo_batch_build_borrow(sl.batch, d_bgr, (const float*)d_depth, p->s_build);
  }
  if (rc) return rc;
  // what the build left to its first consumer runs on the auxiliary stream, behind the set's "built" event
  if (p->s_aux != p->s_build) { rc = revo_batch_prepare(sl.batch, p->s_aux); if (rc) return rc; }
  revo_pair_result* d_out = d_results ? d_results : sl.d_res;
  // with a communicator the grid writes straight into its window of the send buffer (step j of window k -> slot k % ring)
  const unsigned long long win = p->comm ? t / (unsigned long long)p->every : 0ull;
  const int wj = p->comm ? (int)(t % (unsigned long long)p->every) : 0, ws = p->comm ? (int)(win % (unsigned long long)p->ring) : 0;
  if (p->comm) {
    d_out = p->d_send + ((size_t)ws * p->every + wj) * p->n_pairs;
    // the collective that read this send window `ring` windows ago may still be in flight on the other tracker stream
    if (p->coll_pending[ws]) PCHECK(hipStreamWaitEvent(s_tr, p->ev_coll[ws], 0));
  }
  const bool time_it = p->time_every > 0 && ((t + 1) % (unsigned long long)p->time_every) == 0;
  // the event pair is recorded inside the tracker chain, directly around the grid: behind the waits for older grids, the
  // pose upload and the resident gate (recorded here, around the call, it timed gate + waits + grid)
  if (time_it) revo_batch_time_next_grid_(sl.batch, (void*)sl.t0, (void*)sl.t1);
  rc = revo_batch_track_only(sl.batch, h_init_RT, d_out, s_tr);
  if (rc) { if (time_it) revo_batch_time_next_grid_(sl.batch, nullptr, nullptr); return rc; }
  if (time_it) sl.timed = true;
  if (p->comm) {
    if (d_results) PCHECK(hipMemcpyAsync(d_results, d_out, sizeof(revo_pair_result) * p->n_pairs, hipMemcpyDeviceToDevice, s_tr));
    if (wj == 0) p->coll_pending[ws] = 0;  // a new window starts in this slot
    if (wj + 1 < p->every) {
      PCHECK(hipEventRecord(p->ev_step[(size_t)ws * p->every + wj], s_tr));
    } else {
      // the window is complete with this grid: its earlier steps ran on the other tracker stream (alternating), order them
      // in front, then ONE all-gather of the window's records in this step's after-grid slot
      if (p->ntrk > 1)
        for (int j = 0; j < wj; ++j)
          if ((int)((t - (unsigned long long)(wj - j)) % (unsigned long long)p->ntrk) != trk)
            PCHECK(hipStreamWaitEvent(s_tr, p->ev_step[(size_t)ws * p->every + j], 0));
      rc = revo_comm_allgather_records(p->comm, p->d_send + (size_t)ws * p->every * p->n_pairs,
                                       p->d_gathered + (size_t)ws * p->world * p->every * p->n_pairs, p->every * p->n_pairs, (void*)s_tr);
      if (rc) return rc;
      PCHECK(hipEventRecord(p->ev_coll[ws], s_tr));
      p->coll_pending[ws] = 1;
      p->collectives += 1;
    }
  }
  if (p->host_results)
    PCHECK(hipMemcpyAsync(sl.h_res, d_out, sizeof(revo_pair_result) * p->n_pairs, hipMemcpyDeviceToHost, s_tr));
  p->submitted = t + 1;
  sl.ticket = t + 1;
  sl.trk = trk;
  sl.done_recorded = false;
  sl.waited = false;
  if (ticket_out) *ticket_out = t + 1;
  if (after_grid_stream) *after_grid_stream = (void*)s_tr;
  return REVO_OK;
}

extern "C" int revo_pipeline_wait(revo_pipeline* p, uint64_t ticket, revo_pair_result* h_out) {
  if (!p || ticket == 0) return fail(REVO_ERR_INVALID_ARG, "bad argument");
  std::lock_guard<std::mutex> lk(p->mu);
  PCHECK(hipSetDevice(p->device));
  if (ticket > p->submitted) return fail(REVO_ERR_INVALID_ARG, "no such step");
  Slot& sl = p->slots[(ticket - 1) % p->nb];
  if (sl.ticket != ticket) {
    // The slot has moved on.  With host results the records are gone: an error.  Otherwise wait for the tracker stream the step
    // ran on (streams are in order: everything the step and its after-grid slot enqueued lies in front of what is there now);
    // the newer step that holds the slot is NOT finalized by this -- its after-grid slot stays open.
    if (h_out || p->host_results)
      return fail(REVO_ERR_INVALID_ARG, "step " + std::to_string(ticket) + ": its records have been overwritten by a later step");
    PCHECK(hipStreamSynchronize(p->s_trk[(ticket - 1) % (unsigned long long)p->ntrk]));
    return REVO_OK;
  }
  { int rc = finalize_slot(p, sl); if (rc) return rc; }
  PCHECK(hipEventSynchronize(sl.ev_done));
  { int rc = harvest_timing(p, sl); if (rc) return rc; }
  sl.waited = true;
  if (p->host_results) {
    if (h_out) memcpy(h_out, sl.h_res, sizeof(revo_pair_result) * p->n_pairs);
    for (int i = 0; i < p->n_pairs; ++i)
      if (sl.h_res[i].flags & 8)
        return fail(REVO_ERR_HIP, "tracker: step " + std::to_string(ticket) + " pair " + std::to_string(i) +
                                      ": the workgroups of the pair could not exchange partial sums in time (device shared with "
                                      "another process?) -- its pose is not valid");
  } else if (h_out) {
    return fail(REVO_ERR_INVALID_ARG, "this pipeline was created without host_results: read the records from d_results");
  }
  return REVO_OK;
}

extern "C" int revo_pipeline_drain(revo_pipeline* p) {
  if (!p) return fail(REVO_ERR_INVALID_ARG, "null pipeline");
  std::lock_guard<std::mutex> lk(p->mu);
  PCHECK(hipSetDevice(p->device));
  for (Slot& sl : p->slots) { int rc = finalize_slot(p, sl); if (rc) return rc; }
  hipStream_t all[4] = {p->s_build, p->s_aux, p->s_trk[0], p->s_trk[1]};
  for (int i = 0; i < 4; ++i) PCHECK(hipStreamSynchronize(all[i]));
  for (Slot& sl : p->slots) { int rc = harvest_timing(p, sl); if (rc) return rc; }
  return REVO_OK;
}

extern "C" int revo_pipeline_info(const revo_pipeline* p, revo_pipeline_info_t* out) {
  if (!p || !out) return fail(REVO_ERR_INVALID_ARG, "null argument");
  memset(out, 0, sizeof(*out));
  out->batches = p->nb;
  out->pairs_per_step = p->n_pairs;
  out->tracker_streams = p->ntrk;
  out->distinct_hw_queues = p->distinct_queues;
  out->streams_replaced = p->streams_replaced;
  out->probes_run = p->probes;
  out->streams[0] = (void*)p->s_trk[0]; out->streams[1] = (void*)p->s_trk[1];
  out->streams[2] = (void*)p->s_build; out->streams[3] = (void*)p->s_aux;
  out->steps_submitted = p->submitted;
  return REVO_OK;
}

extern "C" int revo_pipeline_batch(revo_pipeline* p, uint64_t ticket, revo_batch** out) {
  if (!p || !out || ticket == 0 || ticket > p->submitted) return fail(REVO_ERR_INVALID_ARG, "bad argument");
  Slot& sl = p->slots[(ticket - 1) % p->nb];
  if (sl.ticket != ticket) return fail(REVO_ERR_INVALID_ARG, "the batch of that step has been rebuilt by a later step");
  *out = sl.batch;
  return REVO_OK;
}

extern "C" int revo_pipeline_time_tracker(revo_pipeline* p, int every_n) {
  if (!p || every_n < 0) return fail(REVO_ERR_INVALID_ARG, "bad argument");
  std::lock_guard<std::mutex> lk(p->mu);
  p->time_every = every_n;
  p->timed_ms = 0.0;
  p->timed_n = 0;
  return REVO_OK;
}

extern "C" int revo_pipeline_tracker_ms(revo_pipeline* p, float* mean_ms, int* launches) {
  if (!p) return fail(REVO_ERR_INVALID_ARG, "null pipeline");
  std::lock_guard<std::mutex> lk(p->mu);
  if (mean_ms) *mean_ms = p->timed_n ? (float)(p->timed_ms / p->timed_n) : 0.f;
  if (launches) *launches = p->timed_n;
  return REVO_OK;
}

// ---- the result collective inside the pipeline -------------------------------------------------------------
extern "C" int revo_pipeline_set_comm(revo_pipeline* p, revo_comm* comm, int every, revo_pair_result* d_gathered, int ring) {
  if (!p) return fail(REVO_ERR_INVALID_ARG, "null pipeline");
  std::lock_guard<std::mutex> lk(p->mu);
  PCHECK(hipSetDevice(p->device));
  if (p->submitted % (unsigned long long)(p->every > 0 ? p->every : 1) != 0)
    return fail(REVO_ERR_INVALID_ARG, "a window of the current communicator is incomplete: revo_pipeline_flush_comm first");
  // detach / re-attach only with nothing in flight
  hipStream_t all[4] = {p->s_build, p->s_aux, p->s_trk[0], p->s_trk[1]};
  for (int i = 0; i < 4; ++i) PCHECK(hipStreamSynchronize(all[i]));
  for (hipEvent_t e : p->ev_step) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->ev_coll) if (e) (void)hipEventDestroy(e);
  p->ev_step.clear(); p->ev_coll.clear(); p->coll_pending.clear();
  if (p->d_send) { (void)hipFree(p->d_send); p->d_send = nullptr; }
  p->comm = nullptr; p->every = 0; p->ring = 0; p->world = 1; p->d_gathered = nullptr;
  if (!comm) return REVO_OK;
  if (every < 1 || every > 8 || ring < 2 || ring > 16 || !d_gathered)
    return fail(REVO_ERR_INVALID_ARG, "set_comm: every must be 1..8, ring 2..16, d_gathered non-null");
  if (p->submitted != 0 && p->submitted % (unsigned long long)every != 0)
    return fail(REVO_ERR_INVALID_ARG, "set_comm: attach at a multiple of `every` submitted steps");
  int world = 1, rank = 0;
  { int rc = revo_comm_world(comm, &world, &rank); if (rc) return rc; }
  PCHECK(hipMalloc((void**)&p->d_send, sizeof(revo_pair_result) * (size_t)ring * every * p->n_pairs));
  PCHECK(hipMemset(p->d_send, 0, sizeof(revo_pair_result) * (size_t)ring * every * p->n_pairs));
  PCHECK(hipStreamSynchronize(nullptr));
  p->ev_step.assign((size_t)ring * every, nullptr);
  p->ev_coll.assign((size_t)ring, nullptr);
  p->coll_pending.assign((size_t)ring, 0);
  for (hipEvent_t& e : p->ev_step) PCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (hipEvent_t& e : p->ev_coll) PCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  p->comm = comm; p->every = every; p->ring = ring; p->world = world; p->d_gathered = d_gathered;
  return REVO_OK;
}

// Gathers an INCOMPLETE last window (submitted % every steps; the unused steps of the window carry stale records) in the
// after-grid slot of the last submitted step.  A collective: every rank calls it at the same point.  *steps_out: how many steps
// of the window are valid (0: the last window was complete, nothing was enqueued), *slot_out: its slot in d_gathered.
extern "C" int revo_pipeline_flush_comm(revo_pipeline* p, int* steps_out, int* slot_out) {
  if (!p) return fail(REVO_ERR_INVALID_ARG, "null pipeline");
  std::lock_guard<std::mutex> lk(p->mu);
  if (!p->comm) return fail(REVO_ERR_INVALID_ARG, "no communicator attached");
  PCHECK(hipSetDevice(p->device));
  const int have = (int)(p->submitted % (unsigned long long)p->every);
  const unsigned long long win = p->submitted / (unsigned long long)p->every;
  const int ws = (int)(win % (unsigned long long)p->ring);
  if (steps_out) *steps_out = have;
  if (slot_out) *slot_out = ws;
  if (!have) return REVO_OK;
  const unsigned long long t_last = p->submitted - 1;
  const int trk = (int)(t_last % (unsigned long long)p->ntrk);
  hipStream_t s_tr = p->s_trk[trk];
  for (int j = 0; j < have; ++j)
    if ((int)((win * p->every + j) % (unsigned long long)p->ntrk) != trk)
      PCHECK(hipStreamWaitEvent(s_tr, p->ev_step[(size_t)ws * p->every + j], 0));
  int rc = revo_comm_allgather_records(p->comm, p->d_send + (size_t)ws * p->every * p->n_pairs,
                                       p->d_gathered + (size_t)ws * p->world * p->every * p->n_pairs, p->every * p->n_pairs, (void*)s_tr);
  if (rc) return rc;
  PCHECK(hipEventRecord(p->ev_coll[ws], s_tr));
  p->coll_pending[ws] = 1;
  p->collectives += 1;
  // the window is closed: the next submit starts a fresh one (the step counter moves to the next multiple of `every`)
  p->submitted = (win + 1) * (unsigned long long)p->every;
  return REVO_OK;
}
