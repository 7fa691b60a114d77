/* selftest.c -- sanitizer driver of the CPU oracle (TEST INFRASTRUCTURE, like everything under oracle/).
 * Built by `make -C oracle sanitize` with -fsanitize=address,undefined and run by
 * tests/test_oracle_primitives.py: two pyramids of a procedural scene, keyframe promotion, trackFrames, the coloured
 * cloud, three frames of the REVO::start sequencing and the odd / tiny geometries the image primitives must survive.
 * Any out-of-bounds access, use of uninitialised heap or undefined arithmetic aborts the process. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "revo_oracle.h"

static void scene(int w, int h, float shift, uint8_t* bgr, float* depth) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const float u = (float)x + shift, v = (float)y + 0.5f * shift;
      const int box = ((int)floorf(u / 23.0f) + (int)floorf(v / 17.0f)) & 1;
      const int ring = ((int)(sqrtf((u - w * 0.4f) * (u - w * 0.4f) + (v - h * 0.5f) * (v - h * 0.5f)) / 11.0f)) & 1;
      const int g = box ? 200 : (ring ? 120 : 40);
      uint8_t* p = bgr + ((size_t)y * w + x) * 3;
      p[0] = (uint8_t)g; p[1] = (uint8_t)g; p[2] = (uint8_t)(g > 100 ? g - 20 : g + 10);
      float d = 1.0f + 0.002f * u + (box ? 0.3f : 0.0f);
      if (((x * 7 + y * 13) % 97) == 0) d = 0.0f;          /* holes */
      if (((x * 3 + y * 5) % 211) == 0) d = NAN;           /* invalid */
      depth[(size_t)y * w + x] = d;
    }
}

int main(void) {
  const int W = 160, H = 96;
  revo_pyr_settings ps; memset(&ps, 0, sizeof(ps));
  ps.width = W; ps.height = H; ps.fx = 130.f; ps.fy = 129.f; ps.cx = 79.5f; ps.cy = 47.5f;
  ps.pyr_min_lvl = 2; ps.pyr_max_lvl = 0; ps.canny_threshold1 = 150; ps.canny_threshold2 = 100;
  ps.depth_min = 0.1f; ps.depth_max = 5.2f; ps.use_edge_hist = 1; ps.n_percentage = 0.3f;
  ps.hist_patch[0] = 20; ps.hist_patch[1] = 10; ps.hist_patch[2] = 5;
  revo_opt_settings os; memset(&os, 0, sizeof(os));
  os.lambda_success_fac = 0.5f; os.lambda_fail_fac = 2.0f; os.huber_edge = 0.3f; os.use_edge_filter = 1;
  const float ed[6] = {30, 20, 10, 5, 5, 5};
  for (int i = 0; i < REVO_MAX_LEVELS; ++i) {
    os.lambda_initial[i] = 0.f; os.step_size_min[i] = 1e-16f; os.convergence_eps[i] = 0.999f;
    os.max_its_per_lvl[i] = 100; os.edge_distance_lvl[i] = ed[i];
  }
  revo_tracker_settings ts; memset(&ts, 0, sizeof(ts));
  ts.check_tracking_results = 1; ts.check_init_values = 1; ts.n_frames_hist_voting = 3; ts.histogram_level = 2;

  uint8_t* bgr = (uint8_t*)malloc((size_t)W * H * 3);
  float* depth = (float*)malloc(sizeof(float) * W * H);
  scene(W, H, 0.f, bgr, depth);
  ro_pyramid* ref = ro_pyramid_create(&ps, bgr, (size_t)W * 3, depth, (size_t)W * 4, 0.0);
  scene(W, H, 1.5f, bgr, depth);
  ro_pyramid* cur = ro_pyramid_create(&ps, bgr, (size_t)W * 3, depth, (size_t)W * 4, 1.0);
  ro_pyramid_make_keyframe(ref);
  ro_tracker* t = ro_tracker_create(&ps, &os, &ts);
  float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, T[3] = {0, 0, 0}, err = 0.f;
  revo_residual_info info; int32_t evals[REVO_MAX_LEVELS]; int flags = 0;
  const int st = ro_tracker_track_frames(t, ref, cur, R, T, &err, &info, evals, &flags);
  float* pcl = (float*)malloc(sizeof(float) * 8 * (size_t)W * H);
  const size_t npcl = ro_pyramid_colored_pcl(ref, 1, 0, pcl, (size_t)W * H);
  const size_t ndense = ro_pyramid_colored_pcl(ref, 2, 1, pcl, (size_t)W * H);
  float* tab = (float*)malloc(sizeof(float) * 4 * (size_t)W * H);
  for (int l = 0; l < 3; ++l)
    for (int what = 0; what < 8; ++what) (void)ro_pyramid_read(ref, what, l, tab, sizeof(float) * 4 * (size_t)W * H);
  printf("track: status %d flags %d err %.5f evals %d %d %d, sparse cloud %zu, dense cloud %zu\n", st, flags, err, evals[0], evals[1],
         evals[2], npcl, ndense);

  ro_vo* vo = ro_vo_create(&ps, &os, &ts);
  for (int f = 0; f < 3; ++f) {
    float pose[16];
    scene(W, H, 0.7f * (float)f, bgr, depth);
    (void)ro_vo_push(vo, bgr, (size_t)W * 3, depth, (size_t)W * 4, (double)f, pose);
  }
  printf("vo: %d keyframes\n", ro_vo_num_keyframes(vo));
  ro_vo_destroy(vo);

  /* image primitives on odd and tiny sizes (borders, reflect-101, replicate) */
  for (int w = 1; w <= 9; w += 2)
    for (int h = 1; h <= 7; h += 3) {
      uint8_t g[9 * 7], e[9 * 7], d[5 * 4];
      int16_t dx[9 * 7], dy[9 * 7];
      float dt[9 * 7];
      for (int i = 0; i < w * h; ++i) g[i] = (uint8_t)((i * 37) & 255);
      ro_sobel3(g, w, h, dx, dy);
      ro_canny(g, w, h, 150, 100, e);
      ro_edt(e, w, h, dt);
      if (w >= 2 && h >= 2) ro_pyrdown_u8(g, w, h, d);
    }

  ro_tracker_destroy(t);
  ro_pyramid_destroy(ref); ro_pyramid_destroy(cur);
  free(bgr); free(depth); free(pcl); free(tab);
  puts("SELFTEST OK");
  return 0;
}
