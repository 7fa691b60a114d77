// A plain C++ host of the pipelined batch mode: include/revo_hip.h + the HIP runtime, nothing else -- what an integrator who
// replaces the reference's IO thread + REVO::start loop (system/system.cpp:96,128-284) for batches of independent pairs links.
// usage: pipeline_host W H N_PAIRS bgr.bin depth.bin   (frames [2N][H][W][3] u8 and [2N][H][W] f32, written by the pytest side)
// Runs the batch alone (revo_batch_track) and then `steps` pipelined steps of the same input through revo_pipeline_*, with a
// device-to-device copy of the records in the after-grid slot, and compares everything bit for bit.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "revo_hip.h"

#define OK(call)                                                                              \
  do {                                                                                        \
    const int rc__ = (call);                                                                  \
    if (rc__ != 0) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc__, revo_last_error()); return 1; } \
  } while (0)
#define HIP(call)                                                                             \
  do {                                                                                        \
    const hipError_t e__ = (call);                                                            \
    if (e__ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e__)); return 1; } \
  } while (0)

static std::vector<char> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 6) { std::fprintf(stderr, "usage: %s w h n_pairs bgr.bin depth.bin\n", argv[0]); return 2; }
  const int w = std::atoi(argv[1]), h = std::atoi(argv[2]), n = std::atoi(argv[3]);
  const std::vector<char> bgr = slurp(argv[4]), dep = slurp(argv[5]);
  const size_t npix = (size_t)w * h;
  if (bgr.size() != 2 * (size_t)n * npix * 3 || dep.size() != 2 * (size_t)n * npix * 4) { std::fprintf(stderr, "bad input size\n"); return 2; }
  revo_pyr_settings ps;
  revo_pyr_settings_default(&ps);
  const float sx = w / 640.0f, sy = h / 480.0f;
  ps.width = w; ps.height = h; ps.fx *= sx; ps.fy *= sy; ps.cx *= sx; ps.cy *= sy;
  if (w != 640) { ps.hist_patch[0] = 10; ps.hist_patch[1] = 5; ps.hist_patch[2] = 0; }
  revo_ctx* ctx = nullptr;
  OK(revo_ctx_create(0, &ps, nullptr, nullptr, &ctx));
  uint8_t* d_bgr = nullptr; float* d_dep = nullptr; revo_pair_result *d_ref = nullptr, *d_out = nullptr, *d_copy = nullptr;
  const int steps = 11;
  HIP(hipMalloc((void**)&d_bgr, bgr.size()));
  HIP(hipMalloc((void**)&d_dep, dep.size()));
  HIP(hipMalloc((void**)&d_ref, sizeof(revo_pair_result) * n));
  HIP(hipMalloc((void**)&d_out, sizeof(revo_pair_result) * n * steps));
  HIP(hipMalloc((void**)&d_copy, sizeof(revo_pair_result) * n * steps));
  HIP(hipMemcpy(d_bgr, bgr.data(), bgr.size(), hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_dep, dep.data(), dep.size(), hipMemcpyHostToDevice));
  HIP(hipMemset(d_copy, 0, sizeof(revo_pair_result) * n * steps));
  // the batch alone
  revo_batch* b = nullptr;
  OK(revo_batch_create(ctx, n, &b));
  OK(revo_batch_track(b, d_bgr, d_dep, nullptr, d_ref, nullptr));
  OK(revo_batch_sync(b, nullptr));
  std::vector<revo_pair_result> ref(n), got((size_t)n * steps), cp((size_t)n * steps);
  HIP(hipMemcpy(ref.data(), d_ref, sizeof(revo_pair_result) * n, hipMemcpyDeviceToHost));
  revo_batch_destroy(b);
  // the pipeline: default depth, device records, a copy of the records behind every grid on the grid's own stream
  revo_pipeline* p = nullptr;
  OK(revo_pipeline_create(ctx, n, 0, 0, &p));
  revo_pipeline_info_t info;
  OK(revo_pipeline_info(p, &info));
  std::printf("pipeline: %d batches, %d tracker streams, %d distinct hardware queues, %d streams replaced, %d probes\n", info.batches,
              info.tracker_streams, info.distinct_hw_queues, info.streams_replaced, info.probes_run);
  uint64_t last = 0;
  for (int t = 0; t < steps; ++t) {
    uint64_t ticket = 0;
    void* s = nullptr;
    OK(revo_pipeline_submit(p, d_bgr, d_dep, 0, 1.0, nullptr, d_out + (size_t)t * n, nullptr, &ticket, &s));
    HIP(hipMemcpyAsync(d_copy + (size_t)t * n, d_out + (size_t)t * n, sizeof(revo_pair_result) * n, hipMemcpyDeviceToDevice, (hipStream_t)s));
    last = ticket;
  }
  OK(revo_pipeline_wait(p, last, nullptr));
  OK(revo_pipeline_drain(p));
  HIP(hipMemcpy(got.data(), d_out, sizeof(revo_pair_result) * n * steps, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(cp.data(), d_copy, sizeof(revo_pair_result) * n * steps, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int t = 0; t < steps; ++t)
    for (int i = 0; i < n; ++i) {
      if (std::memcmp(&got[(size_t)t * n + i], &ref[i], sizeof(revo_pair_result)) != 0) ++bad;
      if (std::memcmp(&cp[(size_t)t * n + i], &ref[i], sizeof(revo_pair_result)) != 0) ++bad;
      if (ref[i].flags & (2 | 4 | 8)) ++bad;
    }
  revo_pipeline_destroy(p);
  // host_results: records through pinned memory, one ticket at a time
  OK(revo_pipeline_create(ctx, n, 2, 1, &p));
  for (int t = 0; t < 3; ++t) {
    uint64_t ticket = 0;
    OK(revo_pipeline_submit(p, d_bgr, d_dep, 1, 1.0, nullptr, nullptr, nullptr, &ticket, nullptr));
    std::vector<revo_pair_result> hr(n);
    OK(revo_pipeline_wait(p, ticket, hr.data()));
    for (int i = 0; i < n; ++i)
      if (std::memcmp(&hr[i], &ref[i], sizeof(revo_pair_result)) != 0) ++bad;
  }
  revo_pipeline_destroy(p);
  // the result collective behind the C ABI: a world-size-1 RCCL communicator, windows of 2 steps gathered by the
  // pipeline itself in the after-grid slot, 7 steps (3 complete windows + a flushed half one), ring of 2 window slots
  {
    char path[256]; int ver = 0;
    OK(revo_comm_available(path, sizeof(path), &ver));
    uint8_t id[REVO_COMM_ID_BYTES];
    OK(revo_comm_unique_id(id));
    revo_comm* comm = nullptr;
    OK(revo_comm_create(ctx, id, 1, 0, &comm));
    const int every = 2, ring = 2, csteps = 7;
    revo_pair_result* d_gath = nullptr;
    HIP(hipMalloc((void**)&d_gath, sizeof(revo_pair_result) * ring * 1 * every * n));
    OK(revo_pipeline_create(ctx, n, 0, 0, &p));
    OK(revo_pipeline_set_comm(p, comm, every, d_gath, ring));
    std::vector<revo_pair_result> win((size_t)every * n);
    int windows_checked = 0;
    for (int t = 0; t < csteps; ++t) {
      uint64_t ticket = 0;
      OK(revo_pipeline_submit(p, d_bgr, d_dep, 0, 1.0, nullptr, nullptr, nullptr, &ticket, nullptr));
      if ((t + 1) % every == 0) {  // the window's last step: its wait covers the collective
        OK(revo_pipeline_wait(p, ticket, nullptr));
        const int slot = (t / every) % ring;
        HIP(hipMemcpy(win.data(), d_gath + (size_t)slot * every * n, sizeof(revo_pair_result) * every * n, hipMemcpyDeviceToHost));
        for (int j = 0; j < every; ++j)
          for (int i = 0; i < n; ++i)
            if (std::memcmp(&win[(size_t)j * n + i], &ref[i], sizeof(revo_pair_result)) != 0) ++bad;
        ++windows_checked;
      }
    }
    int valid = -1, slot = -1;
    OK(revo_pipeline_flush_comm(p, &valid, &slot));
    OK(revo_pipeline_drain(p));
    if (valid != csteps % every || slot != (csteps / every) % ring) ++bad;
    HIP(hipMemcpy(win.data(), d_gath + (size_t)slot * every * n, sizeof(revo_pair_result) * every * n, hipMemcpyDeviceToHost));
    for (int j = 0; j < valid; ++j)
      for (int i = 0; i < n; ++i)
        if (std::memcmp(&win[(size_t)j * n + i], &ref[i], sizeof(revo_pair_result)) != 0) ++bad;
    std::printf("native collective: %s (nccl %d), %d windows + %d flushed step(s) gathered\n", path, ver, windows_checked, valid);
    revo_pipeline_destroy(p);
    revo_comm_destroy(comm);
    (void)hipFree(d_gath);
  }
  revo_ctx_destroy(ctx);
  (void)hipFree(d_bgr); (void)hipFree(d_dep); (void)hipFree(d_ref); (void)hipFree(d_out); (void)hipFree(d_copy);
  std::printf("pose0 T %.9g %.9g %.9g evals %d %d %d\n", ref[0].T[0], ref[0].T[1], ref[0].T[2], ref[0].evals[0], ref[0].evals[1], ref[0].evals[2]);
  if (bad) { std::printf("MISMATCHES %d\n", bad); return 1; }
  std::printf("PIPELINE_HOST_OK %d steps x %d pairs\n", steps, n);
  return 0;
}
