#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* pthread_setaffinity_np (CPU baseline by the book) */
#endif
/*
 * revo_oracle.c -- CPU ORACLE (test infrastructure only; see revo_oracle.h).
 *
 * Plain-C restatement of the REVO hot path.  Every function cites the reference
 * file:line (relative to the reference tree) it follows.  Library routines the
 * reference calls (OpenCV 3.x cvtColor/pyrDown/Canny/distanceTransform, Eigen
 * 3.3 LDLT / Quaternion, Sophus SE3) are restated from their published
 * algorithms; see the header for what is and is not pinned.
 *
 * Build with -ffp-contract=off: the reference is built without FMA
 * (CMakeLists.txt:6-9 has -mavx2 but no -mfma), so a*b+c rounds twice.
 */
#include "revo_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static int g_accum_double = 0;
void ro_set_accum_double(int on) { g_accum_double = on; }
/* Test hook (tests/test_oracle_tracker.py): every entry of the finished normal equations A/n, b/n moves by -1, 0 or +1 ulp
 * (seeded, deterministic) -- what ANY other summation order, a fused multiply-add or a different compiler does to them.
 * Shows how much of the accept / reject sequence of optimizer.cpp:258-304 is decided by the last bit of LGS6. */
static unsigned g_ab_noise = 0;  /* 0 = off; otherwise the state of the generator */
void ro_set_ab_ulp_noise(unsigned seed) { g_ab_noise = seed; }
static float ab_noise(float v) {
  g_ab_noise = g_ab_noise * 1664525u + 1013904223u;
  const unsigned r = (g_ab_noise >> 16) % 3u;
  if (r == 0 || !isfinite(v)) return v;
  return nextafterf(v, r == 1 ? INFINITY : -INFINITY);
}
/* LM trace (test/analysis aid): one entry per residual evaluation after a level's first,
 * lvl*2 + accepted.  Not thread-safe: single-thread analysis only. */
static int g_lm_trace_on = 0, g_lm_trace_n = 0;
static unsigned char g_lm_trace[16384];
void ro_lm_trace(int on) { g_lm_trace_on = on; g_lm_trace_n = 0; }
int ro_lm_trace_get(unsigned char* dst, int cap) {
  const int n = g_lm_trace_n < cap ? g_lm_trace_n : cap;
  memcpy(dst, g_lm_trace, (size_t)n);
  return g_lm_trace_n;
}
static void lm_trace_push(int lvl, int accepted) {
  if (g_lm_trace_on && g_lm_trace_n < (int)sizeof(g_lm_trace)) g_lm_trace[g_lm_trace_n++] = (unsigned char)(lvl * 2 + accepted);
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ======================================================================== */
/* Image primitives                                                          */
/* ======================================================================== */

/* cv::cvtColor(rgb, gray, CV_BGRA2GRAY) on a 3-channel BGR8 Mat,
 * imgpyramidrgbd.cpp:53.  OpenCV 3.x RGB2Gray<uchar>: 14-bit fixed point,
 * B2Y=1868, G2Y=9617, R2Y=4899, rounding constant 1<<13. */
void ro_bgr2gray(const uint8_t* bgr, size_t stride, int w, int h, uint8_t* gray) {
  for (int y = 0; y < h; ++y) {
    const uint8_t* row = bgr + (size_t)y * stride;
    for (int x = 0; x < w; ++x) {
      const int b = row[3 * x + 0], g = row[3 * x + 1], r = row[3 * x + 2];
      gray[(size_t)y * w + x] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
    }
  }
}

static inline int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

/* cv::pyrDown(gray, downResGray), imgpyramidrgbd.cpp:82.  OpenCV: separable
 * [1 4 6 4 1] x [1 4 6 4 1], BORDER_REFLECT_101, integer accumulate,
 * dst = (sum + 128) >> 8, output (x,y) centred on source (2x,2y).
 * w,h must be even (reference sizes come from Camera width*scale). */
void ro_pyrdown_u8(const uint8_t* src, int w, int h, uint8_t* dst) {
  /* separable form (horizontal sums of the 5 source rows an output row needs, then the vertical
   * combination): the same integers as the 25-tap definition, at the cost OpenCV's routine has */
  const int dw = w / 2, dh = h / 2;
  int* hrow = (int*)malloc(sizeof(int) * 5 * (size_t)dw);
  for (int y = 0; y < dh; ++y) {
    for (int j = 0; j < 5; ++j) {
      const uint8_t* r = src + (size_t)reflect101(2 * y + j - 2, h) * w;
      int* hr = hrow + (size_t)j * dw;
      for (int x = 0; x < dw; ++x) {
        const int c = 2 * x;
        if (c >= 2 && c + 2 < w) hr[x] = r[c - 2] + r[c + 2] + 4 * (r[c - 1] + r[c + 1]) + 6 * r[c];
        else hr[x] = r[reflect101(c - 2, w)] + r[reflect101(c + 2, w)] + 4 * (r[reflect101(c - 1, w)] + r[reflect101(c + 1, w)]) + 6 * r[c];
      }
    }
    for (int x = 0; x < dw; ++x) {
      const int sum = hrow[x] + hrow[4 * (size_t)dw + x] + 4 * (hrow[(size_t)dw + x] + hrow[3 * (size_t)dw + x]) + 6 * hrow[2 * (size_t)dw + x];
      dst[(size_t)y * dw + x] = (uint8_t)((sum + 128) >> 8);
    }
  }
  free(hrow);
}

/* ImgPyramidRGBD::FilterSubsampleWithHoles, imgpyramidrgbd.h:218-249. */
void ro_depth_subsample(const float* in, int w, int h, float* out) {
  const int ow = w / 2, oh = h / 2;
  for (int y = 0; y < oh; ++y)
    for (int x = 0; x < ow; ++x) {
      const int sx = x * 2, sy = y * 2;
      float pixel_out = 0.0f, pixel_in, no_good = 0.0f;
      pixel_in = in[(sx + 0) + (size_t)(sy + 0) * w];
      if (pixel_in > 0.0f) { pixel_out += pixel_in; no_good++; }
      pixel_in = in[(sx + 1) + (size_t)(sy + 0) * w];
      if (pixel_in > 0.0f) { pixel_out += pixel_in; no_good++; }
      pixel_in = in[(sx + 0) + (size_t)(sy + 1) * w];
      if (pixel_in > 0.0f) { pixel_out += pixel_in; no_good++; }
      pixel_in = in[(sx + 1) + (size_t)(sy + 1) * w];
      if (pixel_in > 0.0f) { pixel_out += pixel_in; no_good++; }
      if (no_good > 0) pixel_out /= no_good;
      out[x + (size_t)y * ow] = pixel_out;
    }
}

/* iowrapperRGBD.cpp:326-327: depth.convertTo(depth, CV_32FC1, 1.0f/scale):
 * OpenCV cvtScale<ushort,float>: dst = src * (float)alpha + (float)beta. */
void ro_u16_to_depth(const uint16_t* raw, size_t stride, int w, int h,
                     double scale_factor, float* depth) {
  const float alpha = (float)(1.0f / scale_factor);
  for (int y = 0; y < h; ++y) {
    const uint16_t* row = (const uint16_t*)((const uint8_t*)raw + (size_t)y * stride);
    for (int x = 0; x < w; ++x) depth[(size_t)y * w + x] = (float)row[x] * alpha + 0.0f;
  }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cv::Sobel(src, d, CV_16S, 1,0 / 0,1, 3, 1, 0, BORDER_REPLICATE) as called
 * inside cv::Canny (imgpyramidrgbd.cpp:184). */
void ro_sobel3(const uint8_t* g, int w, int h, int16_t* dx, int16_t* dy) {
  for (int y = 0; y < h; ++y) {
    const uint8_t* rm = g + (size_t)clampi(y - 1, 0, h - 1) * w;
    const uint8_t* r0 = g + (size_t)y * w;
    const uint8_t* rp = g + (size_t)clampi(y + 1, 0, h - 1) * w;
    int16_t* ox = dx + (size_t)y * w;
    int16_t* oy = dy + (size_t)y * w;
    /* the two border columns with BORDER_REPLICATE, the interior without clamps (vectorisable) */
    for (int x = 0; x < w; x += (w > 1 ? w - 1 : 1)) {
      const int xm = clampi(x - 1, 0, w - 1), xp = clampi(x + 1, 0, w - 1);
      ox[x] = (int16_t)((rm[xp] + 2 * r0[xp] + rp[xp]) - (rm[xm] + 2 * r0[xm] + rp[xm]));
      oy[x] = (int16_t)((rp[xm] + 2 * rp[x] + rp[xp]) - (rm[xm] + 2 * rm[x] + rm[xp]));
    }
    for (int x = 1; x < w - 1; ++x) {
      ox[x] = (int16_t)((rm[x + 1] + 2 * r0[x + 1] + rp[x + 1]) - (rm[x - 1] + 2 * r0[x - 1] + rp[x - 1]));
      oy[x] = (int16_t)((rp[x - 1] + 2 * rp[x] + rp[x + 1]) - (rm[x - 1] + 2 * rm[x] + rm[x + 1]));
    }
  }
}

/* cv::Canny(gray, edges, thr1, thr2, 3, L2gradient=true), imgpyramidrgbd.cpp:184.
 * OpenCV 3.x (non-IPP path): thresholds swapped so low<=high, squared for L2;
 * magnitude = dx^2+dy^2 (int32), zero outside the image; non-maximum
 * suppression by sector with TG22 = round(tan(22.5deg)*2^15); hysteresis by an
 * explicit stack over the 8-neighbourhood.  Output 255 for map==2. */
void ro_canny(const uint8_t* gray, int w, int h, double thr1, double thr2,
              uint8_t* dst) {
  double low_t = thr1, high_t = thr2;
  if (low_t > high_t) { double t = low_t; low_t = high_t; high_t = t; }
  low_t = low_t < 32767.0 ? low_t : 32767.0;
  high_t = high_t < 32767.0 ? high_t : 32767.0;
  if (low_t > 0) low_t *= low_t;
  if (high_t > 0) high_t *= high_t;
  const int low = (int)floor(low_t), high = (int)floor(high_t);

  int16_t* dx = (int16_t*)malloc(sizeof(int16_t) * (size_t)w * h);
  int16_t* dy = (int16_t*)malloc(sizeof(int16_t) * (size_t)w * h);
  ro_sobel3(gray, w, h, dx, dy);

  const int mw = w + 2, mh = h + 2;
  int* mag = (int*)calloc((size_t)mw * mh, sizeof(int)); /* zero border */
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const int gx = dx[(size_t)y * w + x], gy = dy[(size_t)y * w + x];
      mag[(size_t)(y + 1) * mw + (x + 1)] = gx * gx + gy * gy;
    }
  /* map: 0 = might be edge, 1 = not an edge, 2 = edge; border = 1 */
  uint8_t* map = (uint8_t*)malloc((size_t)mw * mh);
  memset(map, 1, (size_t)mw * mh);
  uint8_t** stack = (uint8_t**)malloc(sizeof(uint8_t*) * ((size_t)w * h + 1));
  size_t sp = 0;

  const int CANNY_SHIFT = 15;
  const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << CANNY_SHIFT) + 0.5);
  for (int y = 0; y < h; ++y) {
    const int* _mag = mag + (size_t)(y + 1) * mw + 1;
    uint8_t* _map = map + (size_t)(y + 1) * mw + 1;
    int prev_flag = 0;
    for (int j = 0; j < w; ++j) {
      const int m = _mag[j];
      int is_max = 0;
      if (m > low) {
        const int xs = dx[(size_t)y * w + j], ys = dy[(size_t)y * w + j];
        const int x = abs(xs);
        const int yv = abs(ys) << CANNY_SHIFT;
        const int tg22x = x * TG22;
        if (yv < tg22x) {
          if (m > _mag[j - 1] && m >= _mag[j + 1]) is_max = 1;
        } else {
          const int tg67x = tg22x + (x << (CANNY_SHIFT + 1));
          if (yv > tg67x) {
            if (m > _mag[j - mw] && m >= _mag[j + mw]) is_max = 1;
          } else {
            const int s = (xs ^ ys) < 0 ? -1 : 1;
            if (m > _mag[j - mw - s] && m > _mag[j + mw + s]) is_max = 1;
          }
        }
      }
      if (!is_max) {
        prev_flag = 0;
        _map[j] = 1;
        continue;
      }
      if (!prev_flag && m > high && _map[j - mw] != 2) {
        _map[j] = 2;
        stack[sp++] = _map + j;
        prev_flag = 1;
      } else {
        _map[j] = 0;
      }
    }
  }
  /* hysteresis: 8-neighbour flood from the strong pixels */
  while (sp > 0) {
    uint8_t* m = stack[--sp];
    static const int dxs[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
    static const int dys[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
    for (int k = 0; k < 8; ++k) {
      uint8_t* n = m + dys[k] * mw + dxs[k];
      if (!*n) { *n = 2; stack[sp++] = n; }
    }
  }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      dst[(size_t)y * w + x] = (uint8_t)(map[(size_t)(y + 1) * mw + (x + 1)] == 2 ? 255 : 0);
  free(stack); free(map); free(mag); free(dx); free(dy);
}

/* ImgPyramidRGBD::generateDistHistogram, imgpyramidrgbd.cpp:146-172.
 * hist is (h/patch) x (w/patch) u8 (wraps at 256 like the reference's ++ on a
 * u8).  Pixels beyond the last full patch are skipped (the reference would
 * index out of bounds).  Returns countNonZero / total. */
float ro_dist_histogram(const uint8_t* edges, int w, int h, int patch, uint8_t* hist) {
  const int hw = w / patch, hh = h / patch;
  memset(hist, 0, (size_t)hw * hh);
  for (int yy = 0; yy < h; ++yy)
    for (int xx = 0; xx < w; ++xx)
      if (edges[(size_t)yy * w + xx] > 0) {
        const int ty = yy / patch, tx = xx / patch;
        if (ty < hh && tx < hw) hist[(size_t)ty * hw + tx]++;
      }
  int nz = 0;
  for (int i = 0; i < hw * hh; ++i) nz += hist[i] != 0;
  return (float)nz / (float)(hw * hh);
}

/* ImgPyramidRGBD::fillInEdges, imgpyramidrgbd.cpp:111-145.  Finer-level pixels whose tile (yy/PATCH_SIZE_LOW,
 * xx/PATCH_SIZE_LOW) lies beyond the coarse level's histogram are skipped: the reference indexes the cv::Mat out of
 * bounds there (only when a level size is not a multiple of its patch size; never at 640x480).  Found by
 * `make -C oracle sanitize`; librevo_hip's k_fill skips the same pixels. */
void ro_fill_in_edges(const uint8_t* dist, int hist_w, const uint8_t* top_edges,
                      int top_w, int top_h, int patch, int patch_low,
                      uint8_t* edges_mod, int mod_w) {
  const int patch2 = patch * patch;
  const int hist_h = (top_h / 2) / patch;
  for (int yy = 0; yy < top_h; ++yy)
    for (int xx = 0; xx < top_w; ++xx)
      if ((yy % 2 == 1) && (xx % 2 == 1) && yy / patch_low < hist_h && xx / patch_low < hist_w &&
          (double)dist[(size_t)(yy / patch_low) * hist_w + (xx / patch_low)] < patch2 * 0.05) {
        if (top_edges[(size_t)yy * top_w + xx] > 0)
          edges_mod[(size_t)(yy / 2) * mod_w + (xx / 2)] = 255;
      }
}

/* addLevelEdge's scan, imgpyramidrgbd.cpp:199-226: x outer / y inner. */
int ro_edges3d(const uint8_t* edges, const float* depth, int w, int h, float fx,
               float fy, float cx, float cy, float dmin, float dmax, float* out4) {
  int n = 0;
  for (int xx = 0; xx < w; ++xx)
    for (int yy = 0; yy < h; ++yy) {
      const float Z = depth[(size_t)yy * w + xx];
      if (isfinite(Z) && Z > dmin && Z < dmax) {
        if (edges[(size_t)yy * w + xx] > 0) {
          const float X = Z * (xx - cx) / fx;
          const float Y = Z * (yy - cy) / fy;
          out4[4 * n + 0] = X; out4[4 * n + 1] = Y; out4[4 * n + 2] = Z; out4[4 * n + 3] = 1.0f;
          ++n;
        }
      }
    }
  return n;
}

/* cv::distanceTransform(255-edges, dt, CV_DIST_L2, CV_DIST_MASK_PRECISE),
 * imgpyramidrgbd.cpp:241.  OpenCV trueDistTrans (Felzenszwalb-Huttenlocher):
 * column pass gives the exact vertical distance (squared, as float; 1e15f in a
 * column without any zero pixel), row pass takes the lower envelope of
 * parabolas, result sqrt(...).  All finite squared distances are integers
 * < 2^24, exact in float, so the result equals sqrtf of the exact squared
 * Euclidean distance; we compute that with the same two-pass structure but in
 * integers, and reproduce the 1e15f sentinel for an image without edges. */
void ro_edt(const uint8_t* edges, int w, int h, float* dt) {
  const int INF = 1 << 29;
  int* g2 = (int*)malloc(sizeof(int) * (size_t)w * h);
  /* column pass as two row sweeps (top-down, bottom-up) with one running distance per column: the same
   * integers as a per-column walk, but sequential in memory (a per-column walk over the row-major image is
   * cache-hostile and would make this CPU baseline slower than the OpenCV routine it stands for) */
  int* run = (int*)malloc(sizeof(int) * (size_t)w);
  for (int x = 0; x < w; ++x) run[x] = INF;
  for (int y = 0; y < h; ++y) { /* nearest edge above */
    const uint8_t* e = edges + (size_t)y * w;
    int* g = g2 + (size_t)y * w;
    for (int x = 0; x < w; ++x) {
      int d = run[x];
      if (e[x] > 0) d = 0; else if (d < INF) d++;
      run[x] = d;
      g[x] = d;
    }
  }
  for (int x = 0; x < w; ++x) run[x] = INF;
  for (int y = h - 1; y >= 0; --y) { /* nearest edge below, then square */
    const uint8_t* e = edges + (size_t)y * w;
    int* g = g2 + (size_t)y * w;
    for (int x = 0; x < w; ++x) {
      int d = run[x];
      if (e[x] > 0) d = 0; else if (d < INF) d++;
      run[x] = d;
      const int m = d < g[x] ? d : g[x];
      g[x] = m >= INF ? INF : m * m;
    }
  }
  free(run);
  /* row pass: lower envelope of parabolas (F-H scan).  The intersection abscissae are kept as exact
   * rationals num/den (den > 0) and compared by cross-multiplication in int64: exact, and no division
   * per candidate (OpenCV avoids it with reciprocal tables). */
  int* v = (int*)malloc(sizeof(int) * (size_t)w);
  long long* zn = (long long*)malloc(sizeof(long long) * ((size_t)w + 1));
  int* zd = (int*)malloc(sizeof(int) * ((size_t)w + 1));
  for (int y = 0; y < h; ++y) {
    const int* f = g2 + (size_t)y * w;
    int k = -1;
    for (int q = 0; q < w; ++q) {
      if (f[q] >= INF) continue;
      const long long fq = (long long)f[q] + (long long)q * q;
      while (k > 0) {
        const int p = v[k];
        const long long num = fq - ((long long)f[p] + (long long)p * p);
        const int den = 2 * (q - p);
        if (num * zd[k] <= zn[k] * den) --k; else break; /* s <= z[k]: parabola p is hidden */
      }
      if (k < 0) { k = 0; v[0] = q; zn[0] = 0; zd[0] = 1; /* z[0] = -inf: never compared */ }
      else {
        const int p = v[k];
        ++k;
        v[k] = q;
        zn[k] = fq - ((long long)f[p] + (long long)p * p);
        zd[k] = 2 * (q - p);
      }
    }
    float* out = dt + (size_t)y * w;
    if (k < 0) { /* no edge anywhere in the image (all columns empty) */
      for (int q = 0; q < w; ++q) out[q] = sqrtf(1e15f);
      continue;
    }
    const int kmax = k;
    k = 0;
    for (int q = 0; q < w; ++q) {
      while (k < kmax && zn[k + 1] < (long long)q * zd[k + 1]) ++k;
      const int p = v[k];
      const int d2 = (q - p) * (q - p) + f[p];
      out[q] = sqrtf((float)d2);
    }
  }
  free(zn); free(zd);
  free(v); free(g2);
}

/* ImgPyramidRGBD::buildOptimizationStructure, imgpyramidrgbd.cpp:255-276.
 * The reference leaves rows 0 and h-1 and the 4th lane uninitialised; here
 * they are zero. */
void ro_grad_table(const float* dt, int w, int h, float* table4) {
  memset(table4, 0, sizeof(float) * 4 * (size_t)w * h);
  const size_t first = (size_t)w, last = (size_t)w * (h - 1);
  for (size_t i = first; i < last; ++i) {
    table4[4 * i + 0] = 0.5f * (dt[i - 1] - dt[i + 1]);
    table4[4 * i + 1] = 0.5f * (dt[i - w] - dt[i + w]);
    table4[4 * i + 2] = dt[i];
  }
}

/* ======================================================================== */
/* Eigen / Sophus restatements                                                */
/* ======================================================================== */
#define M3(R, r, c) ((R)[(c) * 3 + (r)]) /* column-major 3x3 */
#define M4(M, r, c) ((M)[(c) * 4 + (r)]) /* column-major 4x4 */

/* Eigen 3.3 LDLT<Matrix6f>::compute + solve (optimizer.cpp:262): robust
 * Cholesky with diagonal pivoting (largest |diagonal|), lower triangle only,
 * pseudo-inverse of D with tolerance numeric_limits<float>::min(). */
void ro_ldlt6_solve(const float Ain[36], const float bin[6], float x[6]) {
  const int n = 6;
  float m[6][6];
  int tr[6];
  float temp[6];
  for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) m[r][c] = Ain[c * 6 + r];
  for (int k = 0; k < n; ++k) {
    int big = k; float bigv = fabsf(m[k][k]);
    for (int i = k + 1; i < n; ++i) if (fabsf(m[i][i]) > bigv) { bigv = fabsf(m[i][i]); big = i; }
    tr[k] = big;
    if (k != big) {
      for (int c = 0; c < k; ++c) { float t = m[k][c]; m[k][c] = m[big][c]; m[big][c] = t; }
      for (int r = big + 1; r < n; ++r) { float t = m[r][k]; m[r][k] = m[r][big]; m[r][big] = t; }
      { float t = m[k][k]; m[k][k] = m[big][big]; m[big][big] = t; }
      for (int i = k + 1; i < big; ++i) { float t = m[i][k]; m[i][k] = m[big][i]; m[big][i] = t; }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int c = 0; c < k; ++c) temp[c] = m[c][c] * m[k][c];
      float acc = 0.0f;
      for (int c = 0; c < k; ++c) acc += m[k][c] * temp[c];
      m[k][k] -= acc;
      for (int r = 0; r < rs; ++r) {
        float a2 = 0.0f;
        for (int c = 0; c < k; ++c) a2 += m[k + 1 + r][c] * temp[c];
        m[k + 1 + r][k] -= a2;
      }
    }
    const float akk = m[k][k];
    const int valid = fabsf(akk) > 0.0f;
    if (k == 0 && !valid) { for (int j = 0; j < n; ++j) tr[j] = j; break; }
    if (rs > 0 && valid) for (int r = 0; r < rs; ++r) m[k + 1 + r][k] /= akk;
  }
  float d[6];
  for (int i = 0; i < n; ++i) d[i] = bin[i];
  for (int k = 0; k < n; ++k) if (tr[k] != k) { float t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < n; ++i) { float s = d[i]; for (int j = 0; j < i; ++j) s -= m[i][j] * d[j]; d[i] = s; }
  const float tol = 1.17549435e-38f;
  for (int i = 0; i < n; ++i) { if (fabsf(m[i][i]) > tol) d[i] /= m[i][i]; else d[i] = 0.0f; }
  for (int i = n - 1; i >= 0; --i) { float s = d[i]; for (int j = i + 1; j < n; ++j) s -= m[j][i] * d[j]; d[i] = s; }
  for (int k = n - 1; k >= 0; --k) if (tr[k] != k) { float t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < n; ++i) x[i] = d[i];
}

/* Eigen Quaternionf(Matrix3f) (quaternionbase_assign_impl<Other,3,3>), used by
 * Sophus SO3(R) so3.hpp:419 and REVO::writePose system.cpp:78.  q = (w,x,y,z). */
void ro_quat_from_R(const float R[9], float q[4]) {
  float t = M3(R, 0, 0) + M3(R, 1, 1) + M3(R, 2, 2);
  if (t > 0.0f) {
    t = sqrtf(t + 1.0f);
    q[0] = 0.5f * t;
    t = 0.5f / t;
    q[1] = (M3(R, 2, 1) - M3(R, 1, 2)) * t;
    q[2] = (M3(R, 0, 2) - M3(R, 2, 0)) * t;
    q[3] = (M3(R, 1, 0) - M3(R, 0, 1)) * t;
  } else {
    int i = 0;
    if (M3(R, 1, 1) > M3(R, 0, 0)) i = 1;
    if (M3(R, 2, 2) > M3(R, i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrtf(M3(R, i, i) - M3(R, j, j) - M3(R, k, k) + 1.0f);
    q[1 + i] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (M3(R, k, j) - M3(R, j, k)) * t;
    q[1 + j] = (M3(R, j, i) + M3(R, i, j)) * t;
    q[1 + k] = (M3(R, k, i) + M3(R, i, k)) * t;
  }
}

/* Eigen QuaternionBase::toRotationMatrix (SO3::matrix, so3.hpp:280-282). */
void ro_quat_to_R(const float q[4], float R[9]) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  M3(R, 0, 0) = 1.0f - (tyy + tzz); M3(R, 0, 1) = txy - twz; M3(R, 0, 2) = txz + twy;
  M3(R, 1, 0) = txy + twz; M3(R, 1, 1) = 1.0f - (txx + tzz); M3(R, 1, 2) = tyz - twx;
  M3(R, 2, 0) = txz - twy; M3(R, 2, 1) = tyz + twx; M3(R, 2, 2) = 1.0f - (txx + tyy);
}

static void quat_mul(const float a[4], const float b[4], float o[4]) {
  const float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  const float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const float y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  const float z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}

/* Eigen QuaternionBase::_transformVector (SO3 * point). */
static void quat_rotate(const float q[4], const float v[3], float o[3]) {
  float uv[3] = {q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  const float c[3] = {q[2] * uv[2] - q[3] * uv[1], q[3] * uv[0] - q[1] * uv[2], q[1] * uv[1] - q[2] * uv[0]};
  o[0] = v[0] + q[0] * uv[0] + c[0];
  o[1] = v[1] + q[0] * uv[1] + c[1];
  o[2] = v[2] + q[0] * uv[2] + c[2];
}

/* Sophus::SE3f::exp, se3.hpp:723-745 with SO3::expAndTheta so3.hpp:531-565;
 * Constants<float>::epsilon() = 1e-5 (common.hpp:150-158). */
void ro_se3_exp(const float a[6], float q[4], float t[3]) {
  const float ox = a[3], oy = a[4], oz = a[5];
  const float theta_sq = ox * ox + oy * oy + oz * oz;
  const float theta = sqrtf(theta_sq);
  const float half_theta = 0.5f * theta;
  float imag_factor, real_factor;
  if (theta < 1e-5f) {
    const float theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * theta_po4;
    real_factor = 1.0f - (float)(1.0 / 8.0) * theta_sq + (float)(1.0 / 384.0) * theta_po4;
  } else {
    const float s = sinf(half_theta);
    imag_factor = s / theta;
    real_factor = cosf(half_theta);
  }
  q[0] = real_factor; q[1] = imag_factor * ox; q[2] = imag_factor * oy; q[3] = imag_factor * oz;
  /* Omega = hat(omega), Omega_sq = Omega*Omega (row-major here, local) */
  const float O[3][3] = {{0, -oz, oy}, {oz, 0, -ox}, {-oy, ox, 0}};
  float O2[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      O2[r][c] = O[r][0] * O[0][c] + O[r][1] * O[1][c] + O[r][2] * O[2][c];
  float V[3][3];
  if (theta < 1e-5f) {
    float Rm[9];
    ro_quat_to_R(q, Rm);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) V[r][c] = M3(Rm, r, c);
  } else {
    const float ca = (1.0f - cosf(theta)) / theta_sq;
    const float cb = (theta - sinf(theta)) / (theta_sq * theta);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        V[r][c] = ((r == c ? 1.0f : 0.0f) + ca * O[r][c]) + cb * O2[r][c];
  }
  for (int r = 0; r < 3; ++r) t[r] = V[r][0] * a[0] + V[r][1] * a[1] + V[r][2] * a[2];
}

/* SE3 * SE3: se3.hpp:317-321 + SO3::operator*= so3.hpp:335-352 (renormalise
 * with 2/(1+|q|^2)). */
void ro_se3_mul(const float qa[4], const float ta[3], const float qb[4],
                const float tb[3], float qo[4], float to[3]) {
  float rt[3];
  quat_rotate(qa, tb, rt);
  to[0] = ta[0] + rt[0]; to[1] = ta[1] + rt[1]; to[2] = ta[2] + rt[2];
  float q[4];
  quat_mul(qa, qb, q);
  const float sn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (sn != 1.0f) {
    const float s = 2.0f / (1.0f + sn);
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
  }
  qo[0] = q[0]; qo[1] = q[1]; qo[2] = q[2]; qo[3] = q[3];
}

/* Sophus isOrthogonal (rotation_matrix.hpp:14-24) && det > 0 (so3.hpp:419-424). */
int ro_is_orthogonal(const float R[9]) {
  float n2 = 0.0f;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      float v = M3(R, r, 0) * M3(R, c, 0) + M3(R, r, 1) * M3(R, c, 1) + M3(R, r, 2) * M3(R, c, 2);
      v -= (r == c) ? 1.0f : 0.0f;
      n2 += v * v;
    }
  const float det = M3(R, 0, 0) * (M3(R, 1, 1) * M3(R, 2, 2) - M3(R, 1, 2) * M3(R, 2, 1)) -
                    M3(R, 0, 1) * (M3(R, 1, 0) * M3(R, 2, 2) - M3(R, 1, 2) * M3(R, 2, 0)) +
                    M3(R, 0, 2) * (M3(R, 1, 0) * M3(R, 2, 1) - M3(R, 1, 1) * M3(R, 2, 0));
  return sqrtf(n2) < 1e-5f && det > 0.0f;
}

/* Eigen Matrix4f::inverse() (tracker.cpp:142, imgpyramidrgbd.h:129): general
 * cofactor inverse in float. */
void ro_mat4_inverse(const float m[16], float inv[16]) {
  float o[16];
  o[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  o[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  o[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  o[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  o[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  o[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  o[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  o[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  o[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  o[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  o[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  o[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  o[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  o[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  o[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  o[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const float det = m[0] * o[0] + m[1] * o[4] + m[2] * o[8] + m[3] * o[12];
  const float idet = 1.0f / det;
  for (int i = 0; i < 16; ++i) inv[i] = o[i] * idet;
}

void ro_mat4_mul(const float A[16], const float B[16], float out[16]) {
  float o[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      o[c * 4 + r] = M4(A, r, 0) * M4(B, 0, c) + M4(A, r, 1) * M4(B, 1, c) +
                     M4(A, r, 2) * M4(B, 2, c) + M4(A, r, 3) * M4(B, 3, c);
  memcpy(out, o, sizeof(o));
}

static void mat4_identity(float M[16]) {
  memset(M, 0, sizeof(float) * 16);
  M[0] = M[5] = M[10] = M[15] = 1.0f;
}
static void mat4_from_RT(const float R[9], const float T[3], float M[16]) {
  mat4_identity(M);
  for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) M4(M, r, c) = M3(R, r, c);
  M4(M, 0, 3) = T[0]; M4(M, 1, 3) = T[1]; M4(M, 2, 3) = T[2];
}
static void mat4_to_RT(const float M[16], float R[9], float T[3]) {
  for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) M3(R, r, c) = M4(M, r, c);
  T[0] = M4(M, 0, 3); T[1] = M4(M, 1, 3); T[2] = M4(M, 2, 3);
}

/* ======================================================================== */
/* ImgPyramidRGBD                                                             */
/* ======================================================================== */
typedef struct { float fx, fy, cx, cy; int width, height; } ro_camera;

struct ro_pyramid {
  uint8_t* bgr_full;  /* rgbFullSize = fullResRgb.clone(), imgpyramidrgbd.cpp:51 */
  revo_pyr_settings s;
  int n_levels;
  double ts;
  ro_camera cam[REVO_MAX_LEVELS + 1];
  uint8_t* gray[REVO_MAX_LEVELS];
  float* depth[REVO_MAX_LEVELS];
  uint8_t* edges[REVO_MAX_LEVELS];
  uint8_t* edges_orig[REVO_MAX_LEVELS];
  uint8_t* hist[REVO_MAX_LEVELS];
  int hist_w[REVO_MAX_LEVELS], hist_h[REVO_MAX_LEVELS];
  float* pts[REVO_MAX_LEVELS];
  int npts[REVO_MAX_LEVELS];
  float* dt[REVO_MAX_LEVELS];
  float* table[REVO_MAX_LEVELS];
  int is_kf;
  float T_w_f[16];
};

/* Camera(fx,fy,cx,cy,w,h,scale), camerapyr.h:98-103; CameraPyr camerapyr.h:139-144 */
static void make_cameras(const revo_pyr_settings* s, int n_levels, ro_camera* cam) {
  cam[0].fx = s->fx; cam[0].fy = s->fy; cam[0].cx = s->cx; cam[0].cy = s->cy;
  cam[0].width = s->width; cam[0].height = s->height;
  for (int lvl = 1; lvl <= n_levels && lvl <= REVO_MAX_LEVELS; ++lvl) {
    const float scale = 1.0f / (float)pow(2, lvl);
    cam[lvl].fx = s->fx * scale; cam[lvl].fy = s->fy * scale;
    cam[lvl].cx = s->cx * scale; cam[lvl].cy = s->cy * scale;
    cam[lvl].width = (int)((float)s->width * scale);
    cam[lvl].height = (int)((float)s->height * scale);
  }
}

/* addLevelEdge, imgpyramidrgbd.cpp:173-229 (gray/depth ownership transferred). */
static void add_level_edge(ro_pyramid* p, int lvl, uint8_t* gray, float* depth) {
  const ro_camera* cam = &p->cam[lvl];
  const int w = cam->width, h = cam->height;
  p->gray[lvl] = gray;
  p->depth[lvl] = depth;
  uint8_t* edges = (uint8_t*)malloc((size_t)w * h);
  ro_canny(gray, w, h, p->s.canny_threshold1, p->s.canny_threshold2, edges);
  p->edges_orig[lvl] = (uint8_t*)malloc((size_t)w * h);
  memcpy(p->edges_orig[lvl], edges, (size_t)w * h);
  p->edges[lvl] = edges;
  const int patch = p->s.hist_patch[lvl];
  if (patch > 0) {
    p->hist_w[lvl] = w / patch; p->hist_h[lvl] = h / patch;
    p->hist[lvl] = (uint8_t*)malloc((size_t)p->hist_w[lvl] * p->hist_h[lvl] + 1);
    const float frac = ro_dist_histogram(edges, w, h, patch, p->hist[lvl]);
    if (p->s.use_edge_hist && lvl > 0 && p->s.hist_patch[lvl - 1] > 0) {
      if (frac < p->s.n_percentage)
        ro_fill_in_edges(p->hist[lvl], p->hist_w[lvl], p->edges[lvl - 1], p->cam[lvl - 1].width,
                         p->cam[lvl - 1].height, patch, p->s.hist_patch[lvl - 1], edges, w);
    }
  }
  float* list = (float*)malloc(sizeof(float) * 4 * (size_t)w * h + 16);
  p->npts[lvl] = ro_edges3d(edges, depth, w, h, cam->fx, cam->fy, cam->cx, cam->cy,
                            p->s.depth_min, p->s.depth_max, list);
  p->pts[lvl] = list;
}

/* ImgPyramidRGBD::ImgPyramidRGBD(...), imgpyramidrgbd.cpp:43-96. */
ro_pyramid* ro_pyramid_create(const revo_pyr_settings* s, const uint8_t* bgr,
                              size_t bgr_stride, const float* depth_in,
                              size_t depth_stride, double timestamp) {
  ro_pyramid* p = (ro_pyramid*)calloc(1, sizeof(ro_pyramid));
  p->s = *s;
  p->n_levels = s->pyr_min_lvl - s->pyr_max_lvl + 1;
  p->ts = timestamp;
  make_cameras(s, p->n_levels, p->cam);
  mat4_identity(p->T_w_f);
  const int w = s->width, h = s->height;
  p->bgr_full = (uint8_t*)malloc((size_t)w * h * 3);
  for (int y = 0; y < h; ++y) memcpy(p->bgr_full + (size_t)y * w * 3, bgr + (size_t)y * bgr_stride, (size_t)w * 3);
  uint8_t* gray = (uint8_t*)malloc((size_t)w * h);
  ro_bgr2gray(bgr, bgr_stride, w, h, gray);
  float* depth = (float*)malloc(sizeof(float) * (size_t)w * h);
  for (int y = 0; y < h; ++y)
    memcpy(depth + (size_t)y * w, (const uint8_t*)depth_in + (size_t)y * depth_stride, sizeof(float) * (size_t)w);
  add_level_edge(p, 0, gray, depth);
  for (int lvl = 1; lvl < p->n_levels; ++lvl) {
    const int cw = p->cam[lvl].width, ch = p->cam[lvl].height;
    uint8_t* dgray = (uint8_t*)malloc((size_t)cw * ch);
    ro_pyrdown_u8(p->gray[lvl - 1], p->cam[lvl - 1].width, p->cam[lvl - 1].height, dgray);
    float* ddepth = (float*)malloc(sizeof(float) * (size_t)cw * ch);
    ro_depth_subsample(p->depth[lvl - 1], p->cam[lvl - 1].width, p->cam[lvl - 1].height, ddepth);
    add_level_edge(p, lvl, dgray, ddepth);
  }
  return p;
}

void ro_pyramid_destroy(ro_pyramid* p) {
  if (!p) return;
  free(p->bgr_full);
  for (int l = 0; l < REVO_MAX_LEVELS; ++l) {
    free(p->gray[l]); free(p->depth[l]); free(p->edges[l]); free(p->edges_orig[l]);
    free(p->hist[l]); free(p->pts[l]); free(p->dt[l]); free(p->table[l]);
  }
  free(p);
}

/* ImgPyramidRGBD::makeKeyframe, imgpyramidrgbd.cpp:231-252. */
void ro_pyramid_make_keyframe(ro_pyramid* p) {
  for (int lvl = 0; lvl < p->n_levels; ++lvl) {
    const int w = p->cam[lvl].width, h = p->cam[lvl].height;
    free(p->dt[lvl]); free(p->table[lvl]);
    p->dt[lvl] = (float*)malloc(sizeof(float) * (size_t)w * h);
    ro_edt(p->edges[lvl], w, h, p->dt[lvl]);
    p->table[lvl] = (float*)malloc(sizeof(float) * 4 * (size_t)w * h);
    ro_grad_table(p->dt[lvl], w, h, p->table[lvl]);
  }
  p->is_kf = 1;
}
int ro_pyramid_is_keyframe(const ro_pyramid* p) { return p->is_kf; }

void ro_pyramid_camera(const ro_pyramid* p, int lvl, float out6[6]) {
  out6[0] = p->cam[lvl].fx; out6[1] = p->cam[lvl].fy; out6[2] = p->cam[lvl].cx;
  out6[3] = p->cam[lvl].cy; out6[4] = (float)p->cam[lvl].width; out6[5] = (float)p->cam[lvl].height;
}

size_t ro_pyramid_read(const ro_pyramid* p, int what, int lvl, void* dst, size_t cap) {
  if (lvl < 0 || lvl >= p->n_levels) return 0;
  const size_t px = (size_t)p->cam[lvl].width * p->cam[lvl].height;
  const void* src = NULL; size_t count = 0, esz = 1;
  switch (what) {
    case REVO_PLANE_GRAY: src = p->gray[lvl]; count = px; esz = 1; break;
    case REVO_PLANE_DEPTH: src = p->depth[lvl]; count = px; esz = 4; break;
    case REVO_PLANE_EDGES: src = p->edges[lvl]; count = px; esz = 1; break;
    case REVO_PLANE_EDGES_ORIG: /* returnOrigEdges, imgpyramidrgbd.h:67-75 */
      src = (p->s.use_edge_hist && lvl > p->s.pyr_max_lvl) ? p->edges_orig[lvl] : p->edges[lvl];
      count = px; esz = 1; break;
    case REVO_PLANE_DT: src = p->dt[lvl]; count = p->dt[lvl] ? px : 0; esz = 4; break;
    case REVO_PLANE_GRADTABLE: src = p->table[lvl]; count = p->table[lvl] ? px : 0; esz = 16; break;
    case REVO_PLANE_EDGES3D: src = p->pts[lvl]; count = (size_t)p->npts[lvl]; esz = 16; break;
    case REVO_PLANE_HIST: src = p->hist[lvl]; count = p->hist[lvl] ? (size_t)p->hist_w[lvl] * p->hist_h[lvl] : 0; esz = 1; break;
    default: return 0;
  }
  if (dst && src && count * esz <= cap) memcpy(dst, src, count * esz);
  return count;
}

/* cv::pyrDown on a 3-channel BGR8 image (imgpyramidrgbd.cpp:290-295): the same 5x5 kernel per channel. */
static void pyrdown_bgr(const uint8_t* src, int w, int h, uint8_t* dst) {
  const int dw = w / 2, dh = h / 2;
  static const int k[5] = {1, 4, 6, 4, 1};
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x)
      for (int c = 0; c < 3; ++c) {
        int sum = 0;
        for (int j = -2; j <= 2; ++j) {
          const int sy = reflect101(2 * y + j, h);
          int rowsum = 0;
          for (int i = -2; i <= 2; ++i) rowsum += k[i + 2] * src[((size_t)sy * w + reflect101(2 * x + i, w)) * 3 + c];
          sum += k[j + 2] * rowsum;
        }
        dst[((size_t)y * dw + x) * 3 + c] = (uint8_t)((sum + 128) >> 8);
      }
}

/* ImgPyramidRGBD::generateColoredPcl(lvl, clrPcl, densePcl), imgpyramidrgbd.cpp:279-327:
 * out8 receives (X,Y,Z,1,r,g,b,1) per point, x outer / y inner; returns the point count.
 * (The reference handles lvl 0..2 explicitly; here the colour image is pyrDown'ed lvl times.) */
size_t ro_pyramid_colored_pcl(const ro_pyramid* p, int lvl, int dense, float* out8, size_t cap_points) {
  if (lvl < 0 || lvl >= p->n_levels) return 0;
  int w = p->cam[0].width, h = p->cam[0].height;
  uint8_t* rgb = (uint8_t*)malloc((size_t)w * h * 3);
  memcpy(rgb, p->bgr_full, (size_t)w * h * 3);
  for (int l = 0; l < lvl; ++l) {
    uint8_t* d = (uint8_t*)malloc((size_t)(w / 2) * (h / 2) * 3);
    pyrdown_bgr(rgb, w, h, d);
    free(rgb); rgb = d; w /= 2; h /= 2;
  }
  const ro_camera cam = p->cam[lvl];
  const uint8_t* edges = p->edges[lvl];
  const float* depth = p->depth[lvl];
  size_t n = 0;
  for (int xx = 0; xx < w; ++xx)
    for (int yy = 0; yy < h; ++yy) {
      const float Z = depth[(size_t)yy * w + xx];
      if (isfinite(Z) && Z > p->s.depth_min && Z < p->s.depth_max) {
        if (dense || edges[(size_t)yy * w + xx] > 0) {
          if (out8 && n < cap_points) {
            const uint8_t* clr = rgb + ((size_t)yy * w + xx) * 3;
            float* v = out8 + 8 * n;
            v[0] = Z * (xx - cam.cx) / cam.fx; v[1] = Z * (yy - cam.cy) / cam.fy; v[2] = Z; v[3] = 1.0f;
            v[4] = clr[2] / 255.0f; v[5] = clr[1] / 255.0f; v[6] = clr[0] / 255.0f; v[7] = 1.0f;
          }
          ++n;
        }
      }
    }
  free(rgb);
  return n;
}

/* ======================================================================== */
/* Optimizer                                                                  */
/* ======================================================================== */
typedef struct {
  float A[36]; /* Matrix6x6, symmetric */
  float b[6];
  float error;
  size_t num_constraints;
} ro_lgs6;

typedef struct { float pcl[1]; } ro_dummy;

typedef struct ro_past {
  float* pcl; int n; float T_w[16]; double ts;
} ro_past;

struct ro_tracker {
  revo_pyr_settings ps;
  revo_opt_settings os;
  revo_tracker_settings ts;
  /* Optimizer scratch, optimizer.cpp:49-58 */
  float *buf_res, *buf_dx, *buf_dy, *buf_x, *buf_y, *buf_z, *buf_w;
  size_t buf_cap;
  ro_past* past; int n_past, cap_past;
  float hist_weights[4];
};

ro_tracker* ro_tracker_create(const revo_pyr_settings* ps, const revo_opt_settings* os,
                              const revo_tracker_settings* ts) {
  ro_tracker* t = (ro_tracker*)calloc(1, sizeof(ro_tracker));
  t->ps = *ps; t->os = *os; t->ts = *ts;
  t->buf_cap = (size_t)ps->width * ps->height;
  float** bufs[7] = {&t->buf_res, &t->buf_dx, &t->buf_dy, &t->buf_x, &t->buf_y, &t->buf_z, &t->buf_w};
  for (int i = 0; i < 7; ++i) *bufs[i] = (float*)malloc(sizeof(float) * t->buf_cap);
  /* tracker.cpp:231-234 */
  t->hist_weights[0] = 0.0f; t->hist_weights[1] = 1.0f; t->hist_weights[2] = 1.25f; t->hist_weights[3] = 1.5f;
  return t;
}
void ro_tracker_destroy(ro_tracker* t) {
  if (!t) return;
  free(t->buf_res); free(t->buf_dx); free(t->buf_dy); free(t->buf_x); free(t->buf_y); free(t->buf_z); free(t->buf_w);
  for (int i = 0; i < t->n_past; ++i) free(t->past[i].pcl);
  free(t->past);
  free(t);
}

/* Optimizer::calcErrorAndBuffers, optimizer.cpp:74-191 with
 * getInterpolatedElement43 optimizer.h:173-185 and getWeightOfEvoR
 * optimizer.h:156-160. */
static float calc_error_and_buffers(ro_tracker* t, const ro_pyramid* ref, const ro_pyramid* curr,
                                    const float R[9], const float T[3], revo_residual_info* ri, int lvl) {
  ri->good_pts_edges = ri->bad_pts_edges = 0;
  ri->sum_error_unweighted = ri->sum_error_weighted = 0.0f;
  const ro_camera cam = ref->cam[lvl];
  const int w = cam.width, h = cam.height;
  const float* pcl = curr->pts[lvl];
  const int n = curr->npts[lvl];
  const float* tab = ref->table[lvl];
  double sw = 0.0, su = 0.0;
  for (int c = 0; c < n; ++c) {
    const float p0 = pcl[4 * c + 0], p1 = pcl[4 * c + 1], p2 = pcl[4 * c + 2];
    float Wxp[3];
    for (int r = 0; r < 3; ++r)
      Wxp[r] = ((M3(R, r, 0) * p0 + M3(R, r, 1) * p1) + M3(R, r, 2) * p2) + T[r];
    const float u_new = Wxp[0] / Wxp[2] * cam.fx + cam.cx;
    const float v_new = Wxp[1] / Wxp[2] * cam.fy + cam.cy;
    if (!(u_new > 1 && v_new > 1 && u_new < w - 2 && v_new < h - 2)) { ri->bad_pts_edges++; continue; }
    const int ix = (int)u_new, iy = (int)v_new;
    const float dx = u_new - ix, dy = v_new - iy;
    const float dxdy = dx * dy;
    const float* bp = tab + 4 * ((size_t)ix + (size_t)iy * w);
    const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
    float res[3];
    for (int k = 0; k < 3; ++k)
      res[k] = ((w11 * bp[4 * (1 + w) + k] + w01 * bp[4 * w + k]) + w10 * bp[4 + k]) + w00 * bp[k];
    const float residual = res[2];
    if ((residual > t->os.edge_distance_lvl[lvl]) && t->os.use_edge_filter) { ri->bad_pts_edges++; continue; }
    const float w_r = (residual <= t->os.huber_edge ? 1 : t->os.huber_edge / residual);
    const int e = ri->good_pts_edges;
    t->buf_x[e] = Wxp[0]; t->buf_y[e] = Wxp[1]; t->buf_z[e] = Wxp[2];
    t->buf_dx[e] = cam.fx * res[0];
    t->buf_dy[e] = cam.fy * res[1];
    t->buf_res[e] = residual;
    t->buf_w[e] = w_r;
    const float res_2 = residual * residual;
    if (g_accum_double) { sw += (double)(w_r * res_2); su += (double)res_2; }
    else { ri->sum_error_weighted += (w_r * res_2); ri->sum_error_unweighted += res_2; }
    ri->good_pts_edges++;
  }
  if (g_accum_double) { ri->sum_error_weighted = (float)sw; ri->sum_error_unweighted = (float)su; }
  return ri->sum_error_weighted / (ri->good_pts_edges);
}

/* Optimizer::calculateWarpUpdate, optimizer.cpp:192-234 with LGS6
 * initialize/update/finish, LGSX.h:196-204,392-398,320-326. */
static void calculate_warp_update(ro_tracker* t, ro_lgs6* ls, int good) {
  memset(ls, 0, sizeof(*ls));
  double Ad[36] = {0}, bd[6] = {0}, ed = 0.0;
  for (int i = 0; i < good; ++i) {
    const float px = t->buf_x[i], py = t->buf_y[i], pz = t->buf_z[i];
    const float r = t->buf_res[i], gx = t->buf_dx[i], gy = t->buf_dy[i];
    float v[6];
    const float z = 1.0f / pz;
    const float z_sqr = 1.0f / (pz * pz);
    v[0] = z * gx + 0;
    v[1] = 0 + z * gy;
    v[2] = (-px * z_sqr) * gx + (-py * z_sqr) * gy;
    v[3] = (float)((-px * py * z_sqr) * gx + (-(1.0 + py * py * z_sqr)) * gy);
    v[4] = (float)((1.0 + px * px * z_sqr) * gx + (px * py * z_sqr) * gy);
    v[5] = (-py * z) * gx + (px * z) * gy;
    const float weight = t->buf_w[i];
    if (g_accum_double) {
      for (int a = 0; a < 6; ++a) {
        for (int c = 0; c < 6; ++c) Ad[c * 6 + a] += (double)(v[a] * v[c] * weight);
        bd[a] -= (double)(v[a] * (r * weight));
      }
      ed += (double)(r * r * weight);
    } else {
      for (int a = 0; a < 6; ++a) {
        for (int c = 0; c < 6; ++c) ls->A[c * 6 + a] += v[a] * v[c] * weight;
        ls->b[a] -= v[a] * (r * weight);
      }
      ls->error += r * r * weight;
    }
    ls->num_constraints += 1;
  }
  if (g_accum_double) {
    for (int a = 0; a < 36; ++a) ls->A[a] = (float)Ad[a];
    for (int a = 0; a < 6; ++a) ls->b[a] = (float)bd[a];
    ls->error = (float)ed;
  }
  const float nc = (float)ls->num_constraints;
  for (int a = 0; a < 36; ++a) ls->A[a] /= nc;
  for (int a = 0; a < 6; ++a) ls->b[a] /= nc;
  ls->error /= nc;
  if (g_ab_noise) {  /* symmetric: the upper triangle's noise is mirrored */
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c) { const float v = ab_noise(ls->A[c * 6 + r]); ls->A[c * 6 + r] = v; ls->A[r * 6 + c] = v; }
    for (int a = 0; a < 6; ++a) ls->b[a] = ab_noise(ls->b[a]);
  }
}

static int check_pair(const ro_pyramid* ref, const ro_pyramid* curr, int lvl) {
  return ref && curr && ref->is_kf && lvl >= 0 && lvl < ref->n_levels && lvl < curr->n_levels;
}

float ro_optimizer_eval(ro_tracker* t, const ro_pyramid* ref, const ro_pyramid* curr,
                        const float R[9], const float T[3], int lvl, revo_residual_info* info,
                        float A[36], float b[6], float* ls_error) {
  if (!check_pair(ref, curr, lvl)) return NAN;
  revo_residual_info ri;
  const float e = calc_error_and_buffers(t, ref, curr, R, T, &ri, lvl);
  if (info) *info = ri;
  if (A || b || ls_error) {
    ro_lgs6 ls;
    calculate_warp_update(t, &ls, ri.good_pts_edges);
    if (A) memcpy(A, ls.A, sizeof(ls.A));
    if (b) memcpy(b, ls.b, sizeof(ls.b));
    if (ls_error) *ls_error = ls.error;
  }
  return e;
}

/* Optimizer::trackFrames, optimizer.cpp:235-311. */
float ro_optimizer_track_level(ro_tracker* t, const ro_pyramid* ref, const ro_pyramid* curr,
                               float R[9], float T[3], int lvl, revo_residual_info* resInfo,
                               int* evals, int* aborted) {
  if (aborted) *aborted = 0;
  if (evals) *evals = 0;
  if (!check_pair(ref, curr, lvl)) return NAN;
  /* Sophus::SE3f referenceToFrame(R,T): so3.hpp:419-424 */
  if (!ro_is_orthogonal(R)) { if (aborted) *aborted = 1; return NAN; }
  float q[4], tr[3] = {T[0], T[1], T[2]};
  ro_quat_from_R(R, q);
  ro_lgs6 ls;
  int n_eval = 0;
  float lastErr = calc_error_and_buffers(t, ref, curr, R, T, resInfo, lvl); ++n_eval;
  float last_residual = lastErr;
  float LM_lambda = t->os.lambda_initial[lvl];
  for (int iteration = 0; iteration < t->os.max_its_per_lvl[lvl]; iteration++) {
    calculate_warp_update(t, &ls, resInfo->good_pts_edges);
    int incTry = 0;
    while (1) {
      float b[6], A[36], inc[6];
      for (int i = 0; i < 6; ++i) b[i] = -ls.b[i];
      memcpy(A, ls.A, sizeof(A));
      for (int i = 0; i < 6; ++i) A[i * 6 + i] *= 1 + LM_lambda;
      ro_ldlt6_solve(A, b, inc);
      incTry++;
      float qe[4], te[3], qn[4], tn[3], Rn[9];
      ro_se3_exp(inc, qe, te);
      ro_se3_mul(qe, te, q, tr, qn, tn);
      ro_quat_to_R(qn, Rn);
      const float error = calc_error_and_buffers(t, ref, curr, Rn, tn, resInfo, lvl); ++n_eval;
      lm_trace_push(lvl, error < lastErr);
      if (error < lastErr) {
        memcpy(q, qn, sizeof(q)); memcpy(tr, tn, sizeof(tr));
        if (error / lastErr > t->os.convergence_eps[lvl]) iteration = t->os.max_its_per_lvl[lvl];
        last_residual = lastErr = error;
        if (LM_lambda <= 0.2f) LM_lambda = 0.0f;
        else LM_lambda *= t->os.lambda_success_fac;
        break;
      } else {
        const float d = inc[0] * inc[0] + inc[1] * inc[1] + inc[2] * inc[2] + inc[3] * inc[3] + inc[4] * inc[4] + inc[5] * inc[5];
        if (!(d > t->os.step_size_min[lvl])) { iteration = t->os.max_its_per_lvl[lvl]; break; }
        if (LM_lambda == 0.0f) LM_lambda = 0.2f;
        else LM_lambda = (float)((double)LM_lambda * pow((double)t->os.lambda_fail_fac, (double)incTry));
      }
    }
  }
  ro_quat_to_R(q, R);
  T[0] = tr[0]; T[1] = tr[1]; T[2] = tr[2];
  if (evals) *evals = n_eval;
  return last_residual;
}

/* ======================================================================== */
/* TrackerNew                                                                 */
/* ======================================================================== */

/* TrackerNew::evalCostFunction, tracker.cpp:357-393 (without the imwrite). */
float ro_tracker_eval_cost(ro_tracker* t, const float R[9], const float T[3], int minLvl,
                           const ro_pyramid* curr, const ro_pyramid* ref) {
  const ro_camera cam = curr->cam[minLvl];
  const float fx = cam.fx, fy = cam.fy, cx = cam.cx, cy = cam.cy;
  float totalCost = 0;
  double tc = 0.0;
  const float* pcl = curr->pts[minLvl];
  const int n = curr->npts[minLvl];
  const float* dtm = ref->dt[minLvl];
  const int W = cam.width, H = cam.height;
  for (int ir = 0; ir < n; ++ir) {
    const float p0 = pcl[4 * ir], p1 = pcl[4 * ir + 1], p2 = pcl[4 * ir + 2];
    float np[3];
    for (int r = 0; r < 3; ++r)
      np[r] = ((M3(R, r, 0) * p0 + M3(R, r, 1) * p1) + M3(R, r, 2) * p2) + T[r];
    np[0] = fx * np[0] / np[2] + cx;
    np[1] = fy * np[1] / np[2] + cy;
    if (np[0] >= 0 && np[0] < W && np[1] >= 0 && np[1] < H) {
      const float residual = dtm[(size_t)floorf(np[1]) * W + (size_t)floorf(np[0])];
      if (residual > t->os.edge_distance_lvl[minLvl] && t->os.use_edge_filter) continue;
      if (g_accum_double) tc += (double)residual; else totalCost += residual;
    }
  }
  return g_accum_double ? (float)tc : totalCost;
}

/* TrackerNew::trackFrames, tracker.cpp:294-353 with checkInitializationValues
 * tracker.cpp:265-283.  flags bit0: init reset to identity; bit1: abort. */
int ro_tracker_track_frames(ro_tracker* t, const ro_pyramid* ref, const ro_pyramid* curr,
                            float R[9], float T[3], float* err, revo_residual_info* info,
                            int32_t evals[REVO_MAX_LEVELS], int* flags) {
  int fl = 0;
  if (evals) memset(evals, 0, sizeof(int32_t) * REVO_MAX_LEVELS);
  if (t->ts.check_init_values) {
    const int minLvl = curr->s.pyr_min_lvl;
    const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z3[3] = {0, 0, 0};
    const float costEye = ro_tracker_eval_cost(t, I3, Z3, minLvl, curr, ref);
    const float costInit = ro_tracker_eval_cost(t, R, T, minLvl, curr, ref);
    if (costEye < costInit) {
      memcpy(R, I3, sizeof(I3)); memcpy(T, Z3, sizeof(Z3));
      fl |= 1;
    }
  }
  float error = INFINITY;
  revo_residual_info ri; memset(&ri, 0, sizeof(ri));
  for (int lvl = t->ps.pyr_min_lvl; lvl >= t->ps.pyr_max_lvl; --lvl) {
    memset(&ri, 0, sizeof(ri));
    int ne = 0, ab = 0;
    error = ro_optimizer_track_level(t, ref, curr, R, T, lvl, &ri, &ne, &ab);
    if (evals) evals[lvl] = ne;
    if (ab) { fl |= 2; break; }
  }
  if (err) *err = error;
  if (info) *info = ri;
  if (flags) *flags = fl;
  if ((double)ri.good_pts_edges / (double)ri.bad_pts_edges < 4) return REVO_TRACKER_STATE_NEW_KF;
  return REVO_TRACKER_STATE_OK;
}

/* TrackerNew::addOldPclAndPose, tracker.cpp:209-223 */
void ro_tracker_add_old_pcl(ro_tracker* t, const ro_pyramid* src, int lvl, const float T_w[16], double ts) {
  if (t->n_past == t->cap_past) {
    t->cap_past = t->cap_past ? t->cap_past * 2 : 16;
    t->past = (ro_past*)realloc(t->past, sizeof(ro_past) * (size_t)t->cap_past);
  }
  ro_past* p = &t->past[t->n_past++];
  p->n = src->npts[lvl];
  p->pcl = (float*)malloc(sizeof(float) * 4 * (size_t)(p->n + 1));
  memcpy(p->pcl, src->pts[lvl], sizeof(float) * 4 * (size_t)p->n);
  memcpy(p->T_w, T_w, sizeof(float) * 16);
  p->ts = ts;
}
/* TrackerNew::clearUpPastLists, tracker.cpp:248-257 */
void ro_tracker_clear_past(ro_tracker* t) {
  while (t->n_past > t->ts.n_frames_hist_voting) {
    free(t->past[0].pcl);
    memmove(t->past, t->past + 1, sizeof(ro_past) * (size_t)(t->n_past - 1));
    t->n_past--;
  }
}
int ro_tracker_past_size(const ro_tracker* t) { return t->n_past; }

/* TrackerNew::assessTrackingQuality, tracker.cpp:118-201 (without imwrite). */
int ro_tracker_assess_quality(ro_tracker* t, const float T_w_curr[16], const ro_pyramid* curr,
                              int32_t hist4[4], int32_t overlaps4[4]) {
  int histogram[4] = {0, 0, 0, 0}, overlaps[4] = {0, 0, 0, 0};
  if (hist4) memset(hist4, 0, sizeof(int32_t) * 4);
  if (overlaps4) memset(overlaps4, 0, sizeof(int32_t) * 4);
  if (t->n_past == 0 || !t->ts.check_tracking_results) return REVO_TRACKER_STATE_OK;
  const int hl = t->ts.histogram_level;
  const ro_camera cam = curr->cam[hl];
  const int W = cam.width, H = cam.height;
  const uint8_t* currEdges = (curr->s.use_edge_hist && hl > curr->s.pyr_max_lvl) ? curr->edges_orig[hl] : curr->edges[hl];
  const float* currDepth = curr->depth[hl];
  uint8_t* M = (uint8_t*)calloc((size_t)W * H, 1);
  uint8_t* Mi = (uint8_t*)calloc((size_t)W * H, 1);
  int hsize = 1;
  float inv[16];
  ro_mat4_inverse(T_w_curr, inv);
  for (int frame = 0; frame < t->ts.n_frames_hist_voting && frame < t->n_past && frame < 3; ++frame) {
    hsize++;
    float tf[16];
    ro_mat4_mul(inv, t->past[frame].T_w, tf);
    float R[9], T[3];
    mat4_to_RT(tf, R, T);
    memset(Mi, 0, (size_t)W * H);
    const float* pts = t->past[frame].pcl;
    for (int ir = 0; ir < t->past[frame].n; ++ir) {
      const float p0 = pts[4 * ir], p1 = pts[4 * ir + 1], p2 = pts[4 * ir + 2];
      float np[3];
      for (int r = 0; r < 3; ++r)
        np[r] = ((M3(R, r, 0) * p0 + M3(R, r, 1) * p1) + M3(R, r, 2) * p2) + T[r];
      np[0] = cam.fx * np[0] / np[2] + cam.cx;
      np[1] = cam.fy * np[1] / np[2] + cam.cy;
      if (np[0] >= 0 && np[0] < W && np[1] >= 0 && np[1] < H)
        Mi[(size_t)floorf(np[1]) * W + (size_t)floorf(np[0])] = 1;
    }
    for (size_t i = 0; i < (size_t)W * H; ++i) M[i] = (uint8_t)(M[i] + Mi[i]);
  }
  for (int xx = 0; xx < W; ++xx)
    for (int yy = 0; yy < H; ++yy) {
      const float Z = currDepth[(size_t)yy * W + xx];
      if (isfinite(Z) && Z > curr->s.depth_min && Z < curr->s.depth_max) {
        const int val = M[(size_t)yy * W + xx];
        histogram[val]++;
        if (currEdges[(size_t)yy * W + xx] > 0) overlaps[val]++;
      }
    }
  free(M); free(Mi);
  float overlapMeasure = 0.0f;
  for (int hLvl = 0; hLvl < hsize; ++hLvl)
    if (hLvl > 0) overlapMeasure += (overlaps[hLvl] * t->hist_weights[hLvl]);
  if (hist4) for (int i = 0; i < 4; ++i) hist4[i] = histogram[i];
  if (overlaps4) for (int i = 0; i < 4; ++i) overlaps4[i] = overlaps[i];
  if (overlapMeasure >= overlaps[0] || hsize < 4) return REVO_TRACKER_STATE_OK;
  return REVO_TRACKER_STATE_NEW_KF;
}

/* ======================================================================== */
/* REVO::start sequencing, system.cpp:84-305 (one loop body per push)         */
/* ======================================================================== */
typedef struct { float T_kf_curr[16]; float T_w_kf[16]; } ro_pose;

struct ro_vo {
  revo_pyr_settings ps; revo_opt_settings os; revo_tracker_settings ts;
  ro_tracker* tracker;
  ro_pyramid *kf, *prev;       /* kf may alias prev */
  ro_pose last, before_last;   /* mPoseGraph.back() and .at(size-2) */
  int n_poses;
  float T_NM1_N[16];
  float R[9], T[3];
  int no_frames, n_keyframes, just_added_kf;
  double t_pyr, t_kf, t_track;
};

ro_vo* ro_vo_create(const revo_pyr_settings* ps, const revo_opt_settings* os, const revo_tracker_settings* ts) {
  ro_vo* v = (ro_vo*)calloc(1, sizeof(ro_vo));
  v->ps = *ps; v->os = *os; v->ts = *ts;
  v->tracker = ro_tracker_create(ps, os, ts);
  mat4_identity(v->T_NM1_N);
  const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(v->R, I3, sizeof(I3));
  return v;
}
void ro_vo_destroy(ro_vo* v) {
  if (!v) return;
  if (v->kf && v->kf != v->prev) ro_pyramid_destroy(v->kf);
  ro_pyramid_destroy(v->prev);
  ro_tracker_destroy(v->tracker);
  free(v);
}
int ro_vo_num_keyframes(const ro_vo* v) { return v->n_keyframes; }
void ro_vo_times(const ro_vo* v, double out3[3]) { out3[0] = v->t_pyr; out3[1] = v->t_kf; out3[2] = v->t_track; }

static void pose_world(const ro_pose* p, float out[16]) { ro_mat4_mul(p->T_w_kf, p->T_kf_curr, out); }

static int ro_vo_track(ro_vo* v, ro_pyramid* curr, double ts, float pose_out[16]);
int ro_vo_push(ro_vo* v, const uint8_t* bgr, size_t bgr_stride, const float* depth,
               size_t depth_stride, double ts, float pose_out[16]) {
  const double t0 = now_s();
  ro_pyramid* curr = ro_pyramid_create(&v->ps, bgr, bgr_stride, depth, depth_stride, ts);
  v->t_pyr += now_s() - t0;
  return ro_vo_track(v, curr, ts, pose_out);
}

/* one body of the consumer loop of REVO::start (system.cpp:128-284) on a pyramid the producer built */
static int ro_vo_track(ro_vo* v, ro_pyramid* curr, double ts, float pose_out[16]) {
  double t0;
  int new_kf = 0;
  const int hl = v->ts.histogram_level;
  if (v->no_frames == 0) { /* system.cpp:151-175 */
    v->kf = curr; v->prev = curr;
    t0 = now_s();
    ro_pyramid_make_keyframe(v->kf);
    v->t_kf += now_s() - t0;
    mat4_identity(v->kf->T_w_f);
    mat4_identity(v->last.T_kf_curr); mat4_identity(v->last.T_w_kf);
    v->n_poses = 1;
    ++v->n_keyframes;
    mat4_identity(pose_out);
    ++v->no_frames;
    v->just_added_kf = 1;
    float I4[16]; mat4_identity(I4);
    ro_tracker_add_old_pcl(v->tracker, v->kf, hl, I4, ts);
    return 1;
  }
  ++v->no_frames;
  t0 = now_s();
  float error;
  revo_residual_info ri;
  ro_tracker_track_frames(v->tracker, v->kf, curr, v->R, v->T, &error, &ri, NULL, NULL);
  float T_KF_N[16], currPoseInWorld[16];
  mat4_from_RT(v->R, v->T, T_KF_N);
  ro_mat4_mul(v->kf->T_w_f, T_KF_N, currPoseInWorld);
  int status = ro_tracker_assess_quality(v->tracker, currPoseInWorld, curr, NULL, NULL);
  v->t_track += now_s() - t0;
  ro_pyramid* kf_for_pose = v->kf;
  if (status == REVO_TRACKER_STATE_NEW_KF && !v->just_added_kf) { /* system.cpp:203-241 */
    ro_pyramid* old_kf = v->kf;
    v->kf = v->prev;
    float back_world[16];
    pose_world(&v->last, back_world);
    memcpy(v->kf->T_w_f, back_world, sizeof(back_world)); /* setTwf */
    t0 = now_s();
    ro_pyramid_make_keyframe(v->kf);
    v->t_kf += now_s() - t0;
    /* mPoseGraph.back().setKfFrame(kfPyr) */
    memcpy(v->last.T_w_kf, v->kf->T_w_f, sizeof(float) * 16);
    mat4_identity(v->last.T_kf_curr);
    v->n_keyframes++;
    ro_tracker_clear_past(v->tracker);
    mat4_to_RT(v->T_NM1_N, v->R, v->T);
    t0 = now_s();
    ro_tracker_track_frames(v->tracker, v->kf, curr, v->R, v->T, &error, &ri, NULL, NULL);
    mat4_from_RT(v->R, v->T, T_KF_N);
    ro_mat4_mul(v->kf->T_w_f, T_KF_N, currPoseInWorld);
    status = ro_tracker_assess_quality(v->tracker, currPoseInWorld, curr, NULL, NULL);
    v->t_track += now_s() - t0;
    v->just_added_kf = 1;
    new_kf = 1;
    if (old_kf != v->kf) ro_pyramid_destroy(old_kf);
    kf_for_pose = v->kf;
  } else {
    v->just_added_kf = 0;
  }
  /* mPoseGraph.push_back(Pose(T_KF_N, ts, kfPyr)) */
  v->before_last = v->last;
  memcpy(v->last.T_kf_curr, T_KF_N, sizeof(T_KF_N));
  memcpy(v->last.T_w_kf, kf_for_pose->T_w_f, sizeof(float) * 16);
  v->n_poses++;
  ro_tracker_add_old_pcl(v->tracker, curr, hl, currPoseInWorld, ts);
  /* T_NM1_N = graph[size-2].T_N_W() * graph.back().T_W_N(), system.cpp:267 */
  float w2[16], w2inv[16], w1[16];
  pose_world(&v->before_last, w2);
  ro_mat4_inverse(w2, w2inv);
  pose_world(&v->last, w1);
  ro_mat4_mul(w2inv, w1, v->T_NM1_N);
  float T_init[16];
  ro_mat4_mul(v->last.T_kf_curr, v->T_NM1_N, T_init);
  mat4_to_RT(T_init, v->R, v->T);
  memcpy(pose_out, w1, sizeof(w1)); /* absPose = back().getCurrToWorld() */
  /* prevPyr = currPyr */
  if (v->prev != v->kf) ro_pyramid_destroy(v->prev);
  v->prev = curr;
  return new_kf;
}

/* ======================================================================== */
/* Multi-threaded CPU baseline (bench.py "cpu_baseline_all_cores")           */
/* One frame-pair per thread at a time: 2 pyramids + makeKeyframe +          */
/* trackFrames from identity, exactly the per-pair work of the GPU batch.    */
/* Plain pthreads so that the number is not limited by the Python GIL.       */
/* ======================================================================== */
#include <pthread.h>
typedef struct {
  const revo_pyr_settings* ps; const revo_opt_settings* os; const revo_tracker_settings* ts;
  const uint8_t* bgr; const float* depth; /* [2*n_pairs] frames: ref, curr, ref, curr, ... */
  int n_pairs, first; double seconds; long done;
} ro_mt_job;

static void* ro_mt_worker(void* arg) {
  ro_mt_job* j = (ro_mt_job*)arg;
  const int w = j->ps->width, h = j->ps->height;
  const size_t fb = (size_t)w * h * 3, fd = (size_t)w * h;
  ro_tracker* t = ro_tracker_create(j->ps, j->os, j->ts);
  const double t0 = now_s();
  int i = j->first % j->n_pairs;
  while (now_s() - t0 < j->seconds) {
    ro_pyramid* ref = ro_pyramid_create(j->ps, j->bgr + (size_t)(2 * i) * fb, (size_t)w * 3, j->depth + (size_t)(2 * i) * fd, (size_t)w * 4, 0.0);
    ro_pyramid* cur = ro_pyramid_create(j->ps, j->bgr + (size_t)(2 * i + 1) * fb, (size_t)w * 3, j->depth + (size_t)(2 * i + 1) * fd, (size_t)w * 4, 0.0);
    ro_pyramid_make_keyframe(ref);
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, T[3] = {0, 0, 0}, err = 0.f;
    int flags = 0;
    ro_tracker_track_frames(t, ref, cur, R, T, &err, NULL, NULL, &flags);
    ro_pyramid_destroy(ref);
    ro_pyramid_destroy(cur);
    j->done += 1;
    i = (i + 1) % j->n_pairs;
  }
  ro_tracker_destroy(t);
  return NULL;
}

long ro_bench_pairs_mt(const revo_pyr_settings* ps, const revo_opt_settings* os, const revo_tracker_settings* ts,
                       const uint8_t* bgr, const float* depth, int n_pairs, int n_threads, double seconds,
                       double* elapsed_out) {
  if (n_threads < 1) n_threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
  ro_mt_job* jobs = (ro_mt_job*)calloc((size_t)n_threads, sizeof(ro_mt_job));
  const double t0 = now_s();
  for (int k = 0; k < n_threads; ++k) {
    jobs[k] = (ro_mt_job){ps, os, ts, bgr, depth, n_pairs, k, seconds, 0};
    pthread_create(&th[k], NULL, ro_mt_worker, &jobs[k]);
  }
  long total = 0;
  for (int k = 0; k < n_threads; ++k) { pthread_join(th[k], NULL); total += jobs[k].done; }
  if (elapsed_out) *elapsed_out = now_s() - t0;
  free(th); free(jobs);
  return total;
}


/* ======================================================================== */
/* The reference's threading: the IO thread builds   */
/* pyramids into a bounded queue (system.cpp:96, iowrapperRGBD.cpp:279-288), */
/* the main thread promotes keyframes and tracks.  Each thread is pinned to   */
/* one CPU (cpu id < 0: not pinned).                                          */
/* ======================================================================== */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <sched.h>
typedef struct {
  pthread_mutex_t mu; pthread_cond_t cv;
  ro_pyramid** ring; int cap, head, count, closed;
} ro_queue;
static void q_init(ro_queue* q, int cap) {
  pthread_mutex_init(&q->mu, NULL); pthread_cond_init(&q->cv, NULL);
  q->ring = (ro_pyramid**)calloc((size_t)cap, sizeof(ro_pyramid*)); q->cap = cap; q->head = q->count = q->closed = 0;
}
static void q_free(ro_queue* q) { free(q->ring); pthread_mutex_destroy(&q->mu); pthread_cond_destroy(&q->cv); }
static void q_push(ro_queue* q, ro_pyramid* p) {
  pthread_mutex_lock(&q->mu);
  while (q->count == q->cap) pthread_cond_wait(&q->cv, &q->mu);
  q->ring[(q->head + q->count++) % q->cap] = p;
  pthread_cond_broadcast(&q->cv);
  pthread_mutex_unlock(&q->mu);
}
static ro_pyramid* q_pop(ro_queue* q) { /* NULL: closed and drained */
  pthread_mutex_lock(&q->mu);
  while (q->count == 0 && !q->closed) pthread_cond_wait(&q->cv, &q->mu);
  ro_pyramid* p = NULL;
  if (q->count) { p = q->ring[q->head]; q->head = (q->head + 1) % q->cap; --q->count; }
  pthread_cond_broadcast(&q->cv);
  pthread_mutex_unlock(&q->mu);
  return p;
}
static void q_close(ro_queue* q) {
  pthread_mutex_lock(&q->mu); q->closed = 1; pthread_cond_broadcast(&q->cv); pthread_mutex_unlock(&q->mu);
}
static void pin_self(int cpu) {
  if (cpu < 0) return;
  cpu_set_t set; CPU_ZERO(&set); CPU_SET(cpu, &set);
  pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
}

typedef struct {
  const revo_pyr_settings* ps; const uint8_t* bgr; const float* depth; const double* ts;
  int n_frames, cpu; ro_queue* q;
} ro_producer;
static void* ro_producer_main(void* arg) {
  ro_producer* p = (ro_producer*)arg;
  pin_self(p->cpu);
  const int w = p->ps->width, h = p->ps->height;
  const size_t fb = (size_t)w * h * 3, fd = (size_t)w * h;
  for (int i = 0; i < p->n_frames; ++i)
    q_push(p->q, ro_pyramid_create(p->ps, p->bgr + (size_t)i * fb, (size_t)w * 3, p->depth + (size_t)i * fd, (size_t)w * 4,
                                   p->ts ? p->ts[i] : (double)i));
  q_close(p->q);
  return NULL;
}

/* REVO::start with its IO thread: frames [n][H][W][3] / [n][H][W]; poses_out n x 16 (may be NULL).  Returns the
 * wall time of the whole stream. */
double ro_vo_run_pipelined(ro_vo* v, int n_frames, const uint8_t* bgr, const float* depth, const double* ts, int cpu_io,
                           int cpu_main, int queue_cap, float* poses_out) {
  ro_queue q; q_init(&q, queue_cap > 0 ? queue_cap : 4);
  ro_producer prod = {&v->ps, bgr, depth, ts, n_frames, cpu_io, &q};
  cpu_set_t old; const int have_old = pthread_getaffinity_np(pthread_self(), sizeof(old), &old) == 0;
  const double t0 = now_s();
  pthread_t th; pthread_create(&th, NULL, ro_producer_main, &prod);
  pin_self(cpu_main);
  int i = 0;
  for (ro_pyramid* p; (p = q_pop(&q)) != NULL; ++i) {
    float pose[16];
    ro_vo_track(v, p, ts ? ts[i] : (double)i, pose);
    if (poses_out) memcpy(poses_out + 16 * (size_t)i, pose, sizeof(pose));
  }
  pthread_join(th, NULL);
  const double dt = now_s() - t0;
  if (have_old) pthread_setaffinity_np(pthread_self(), sizeof(old), &old);
  q_free(&q);
  return dt;
}

/* The batch workload (n independent frame-pairs: frames ref, curr, ref, curr, ...) with the same two threads: the IO
 * thread builds the pyramids, the main thread runs makeKeyframe + trackFrames per pair.  seconds_out[passes]. */
void ro_bench_pairs_pipelined(const revo_pyr_settings* ps, const revo_opt_settings* os, const revo_tracker_settings* ts,
                              const uint8_t* bgr, const float* depth, int n_pairs, int cpu_io, int cpu_main, int passes,
                              double* seconds_out) {
  cpu_set_t old; const int have_old = pthread_getaffinity_np(pthread_self(), sizeof(old), &old) == 0;
  ro_tracker* t = ro_tracker_create(ps, os, ts);
  for (int pass = 0; pass < passes; ++pass) {
    ro_queue q; q_init(&q, 8);
    ro_producer prod = {ps, bgr, depth, NULL, 2 * n_pairs, cpu_io, &q};
    const double t0 = now_s();
    pthread_t th; pthread_create(&th, NULL, ro_producer_main, &prod);
    pin_self(cpu_main);
    for (;;) {
      ro_pyramid* ref = q_pop(&q);
      if (!ref) break;
      ro_pyramid* cur = q_pop(&q);
      ro_pyramid_make_keyframe(ref);
      float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, T[3] = {0, 0, 0}, err = 0.f;
      int flags = 0;
      ro_tracker_track_frames(t, ref, cur, R, T, &err, NULL, NULL, &flags);
      ro_pyramid_destroy(ref);
      ro_pyramid_destroy(cur);
    }
    pthread_join(th, NULL);
    seconds_out[pass] = now_s() - t0;
    q_free(&q);
  }
  ro_tracker_destroy(t);
  if (have_old) pthread_setaffinity_np(pthread_self(), sizeof(old), &old);
}
