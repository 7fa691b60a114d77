// revo_pyramid.hip -- gfx950 kernels for the per-frame pyramid build
// (ImgPyramidRGBD ctor, imgpyramidrgbd.cpp:43-96,173-229) and the keyframe
// promotion (makeKeyframe, imgpyramidrgbd.cpp:231-276).
//
// Integer/byte streaming and small stencils: coalesced dword/dwordx4 row accesses, an LDS
// tile for the 3x3 stencil + union-find hysteresis, LDS staging where a block's output is a
// contiguous range (ordered compaction), blockIdx.z = frame, blockIdx.x decodes level + tile
// so one launch covers every level of every frame in the batch.  No MFMA: there is no
// contraction anywhere on this path.  What binds each kernel (HBM for k_gray_depth, VALU issue for
// k_canny_nms, dependent-load latency for the hysteresis passes, ...) is tabulated in DESIGN.md 3.
//
// Exactness: every integer stage is bit-exact by construction; float stages
// keep the reference's operation order and are compiled with -ffp-contract=off.
#include "revo_dev.h"

namespace {

__device__ __forceinline__ int level_of(const PyrGeom& g, int v, int LevelGeom::*base) {
  int l = 0;
#pragma unroll
  for (int k = 1; k < REVO_L; ++k)
    if (k < g.n_levels && v >= g.lv[k].*base) l = k;
  return l;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

// ---------------------------------------------------------------------------
// a2: cv::cvtColor BGR->GRAY (imgpyramidrgbd.cpp:53) fused with the depth clone
// (cpp:54) or the u16 -> metres conversion of iowrapperRGBD.cpp:326-327.
// 4 pixels per thread: 3 dword loads of BGR, 1 dword store of gray, float4 depth.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gray_depth(const uint8_t* __restrict__ bgr, const float* __restrict__ depth_f32,
                                                    const uint16_t* __restrict__ depth_u16, float alpha,
                                                    uint8_t* __restrict__ gray, float* __restrict__ depth_out, int npix, int frame0) {
  const int f = frame0 + blockIdx.z;
  const int g4 = blockIdx.x * 256 + threadIdx.x;
  if (g4 * 4 >= npix) return;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(bgr + (size_t)f * npix * 3) + (size_t)g4 * 3;
  const uint32_t a = src[0], b = src[1], c = src[2];
  // bytes: a = B0 G0 R0 B1 | b = G1 R1 B2 G2 | c = R2 B3 G3 R3
  const int B0 = a & 255, G0 = (a >> 8) & 255, R0 = (a >> 16) & 255, B1 = a >> 24;
  const int G1 = b & 255, R1 = (b >> 8) & 255, B2 = (b >> 16) & 255, G2 = b >> 24;
  const int R2 = c & 255, B3 = (c >> 8) & 255, G3 = (c >> 16) & 255, R3 = c >> 24;
  const uint32_t y0 = (uint32_t)(B0 * 1868 + G0 * 9617 + R0 * 4899 + 8192) >> 14;
  const uint32_t y1 = (uint32_t)(B1 * 1868 + G1 * 9617 + R1 * 4899 + 8192) >> 14;
  const uint32_t y2 = (uint32_t)(B2 * 1868 + G2 * 9617 + R2 * 4899 + 8192) >> 14;
  const uint32_t y3 = (uint32_t)(B3 * 1868 + G3 * 9617 + R3 * 4899 + 8192) >> 14;
  reinterpret_cast<uint32_t*>(gray + (size_t)f * npix)[g4] = y0 | (y1 << 8) | (y2 << 16) | (y3 << 24);
  float4 d;
  if (depth_u16) {
    const uint2 r = reinterpret_cast<const uint2*>(depth_u16 + (size_t)f * npix)[g4];
    d.x = (float)(r.x & 0xffff) * alpha + 0.0f;
    d.y = (float)(r.x >> 16) * alpha + 0.0f;
    d.z = (float)(r.y & 0xffff) * alpha + 0.0f;
    d.w = (float)(r.y >> 16) * alpha + 0.0f;
  } else {
    d = reinterpret_cast<const float4*>(depth_f32 + (size_t)f * npix)[g4];
  }
  reinterpret_cast<float4*>(depth_out + (size_t)f * npix)[g4] = d;
}

// ---------------------------------------------------------------------------
// a3 + a4: cv::pyrDown (imgpyramidrgbd.cpp:82) + FilterSubsampleWithHoles
// (imgpyramidrgbd.h:218-249).  One thread makes a 4 x 2 block of outputs straight from aligned
// words of the source rows (7 rows x 4 words; L1/L2 serve the reuse between neighbouring threads):
// no LDS, no barrier, no per-byte address arithmetic.  (The first version staged a 67x19 byte tile
// in LDS one byte per thread-iteration with a div/mod each: ~250 VALU per output pixel.)
// Separable [1 4 6 4 1], BORDER_REFLECT_101, (sum + 128) >> 8.  Source width is a multiple of 8.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pyrdown(const uint8_t* __restrict__ src, int sw, int sh, uint8_t* __restrict__ dst,
                                                 int dw, int dh, const float* __restrict__ dsrc, float* __restrict__ ddst, int frame0) {
  const int f = frame0 + blockIdx.z;
  src += (size_t)f * sw * sh;
  dst += (size_t)f * dw * dh;
  dsrc += (size_t)f * sw * sh;
  ddst += (size_t)f * dw * dh;
  const int gw = dw >> 2, gh = (dh + 1) >> 1;  // groups of 4 outputs per row, pairs of output rows
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= gw * gh) return;
  const int gx = i % gw, gy = i / gw;
  const int oy = 2 * gy;
  const int nrows = (oy + 1 < dh) ? 2 : 1;
  const int swords = sw >> 2;
  const uint32_t* srcw = reinterpret_cast<const uint32_t*>(src);
  const int w1i = 2 * gx;  // word holding source pixel 8*gx
  int hs[7][4];            // horizontal sums of source rows 2*oy-2 .. 2*oy+4 at the 4 output columns
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    if (r < 2 * nrows + 3) {
      const uint32_t* row = srcw + (size_t)reflect101(2 * oy - 2 + r, sh) * swords;
      uint32_t w0 = row[max(w1i - 1, 0)];
      const uint32_t w1 = row[w1i], w2 = row[w1i + 1];
      uint32_t w3 = row[min(w1i + 2, swords - 1)];
      // BORDER_REFLECT_101: pixels -2, -1 are pixels 2, 1; pixel sw is pixel sw-2
      if (gx == 0) w0 = (((w1 >> 16) & 0xffu) << 16) | (((w1 >> 8) & 0xffu) << 24);
      if (gx == gw - 1) w3 = (w2 >> 16) & 0xffu;
      // b[k] = source pixel 8*gx - 2 + k, k = 0..10
      const int b[11] = {(int)((w0 >> 16) & 0xffu), (int)(w0 >> 24),
                         (int)(w1 & 0xffu), (int)((w1 >> 8) & 0xffu), (int)((w1 >> 16) & 0xffu), (int)(w1 >> 24),
                         (int)(w2 & 0xffu), (int)((w2 >> 8) & 0xffu), (int)((w2 >> 16) & 0xffu), (int)(w2 >> 24),
                         (int)(w3 & 0xffu)};
#pragma unroll
      for (int j = 0; j < 4; ++j) hs[r][j] = b[2 * j] + b[2 * j + 4] + 4 * (b[2 * j + 1] + b[2 * j + 3]) + 6 * b[2 * j + 2];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) hs[r][j] = 0;
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (q >= nrows) break;
    uint32_t packed = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int v = hs[2 * q][j] + hs[2 * q + 4][j] + 4 * (hs[2 * q + 1][j] + hs[2 * q + 3][j]) + 6 * hs[2 * q + 2][j];
      packed |= (uint32_t)((v + 128) >> 8) << (8 * j);
    }
    reinterpret_cast<uint32_t*>(dst + (size_t)(oy + q) * dw)[gx] = packed;
    // depth: mean of the positive samples of each 2x2 block, reference order
    const float* d0 = dsrc + (size_t)(2 * (oy + q)) * sw + 8 * gx;
    const float4 a0 = *reinterpret_cast<const float4*>(d0), a1 = *reinterpret_cast<const float4*>(d0 + 4);
    const float4 c0 = *reinterpret_cast<const float4*>(d0 + sw), c1 = *reinterpret_cast<const float4*>(d0 + sw + 4);
    const float top[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float bot[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float out = 0.0f, cnt = 0.0f;
      if (top[2 * j] > 0.0f) { out += top[2 * j]; cnt += 1.0f; }
      if (top[2 * j + 1] > 0.0f) { out += top[2 * j + 1]; cnt += 1.0f; }
      if (bot[2 * j] > 0.0f) { out += bot[2 * j]; cnt += 1.0f; }
      if (bot[2 * j + 1] > 0.0f) { out += bot[2 * j + 1]; cnt += 1.0f; }
      if (cnt > 0.0f) out = __fdiv_rn(out, cnt);
      o[j] = out;
    }
    *reinterpret_cast<float4*>(ddst + (size_t)(oy + q) * dw + 4 * gx) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ---------------------------------------------------------------------------
// Union-find helpers for the Canny hysteresis.  Keys are pixel indices; the
// parent of a pixel always has a smaller key, roots point to themselves.
// Lock-free union by atomicMin (Komura-style); placement/order independent.
// ---------------------------------------------------------------------------
// Loads that race with concurrent unions: agent scope is enough (L2-served); HIP's bare
// __atomic_load_n would be SYSTEM scope, i.e. a cache-bypassing fabric read per hop.
template <typename P>
__device__ __forceinline__ int uf_load(P* L, int i) { return __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <typename P>
__device__ __forceinline__ int uf_find(P* L, int x) {
  int p = uf_load(L, x);
  while (p != x) { x = p; p = uf_load(L, x); }
  return x;
}
// Global variant with path halving: x's parent is replaced by its grandparent through
// atomicMin (parents only ever decrease, and a grandparent is in the same set), which keeps
// the root chains of giant components (textured walls) short while unions are in flight.
__device__ __forceinline__ int uf_find_halving(int* L, int x) {
  int p = uf_load(L, x);
  while (p != x) {
    const int gp = uf_load(L, p);
    if (gp != p) atomicMin(&L[x], gp);
    x = p;
    p = gp;
  }
  return x;
}
// The two root walks run in lockstep: the two loads of a step are independent, so a union costs
// max(depth_a, depth_b) round trips instead of depth_a + depth_b.
__device__ __forceinline__ void uf_unite_global(int* L, int a, int b) {
  int pa = uf_load(L, a), pb = uf_load(L, b);
  for (;;) {
    while (pa != a || pb != b) {
      const int ga = (pa != a) ? uf_load(L, pa) : pa;
      const int gb = (pb != b) ? uf_load(L, pb) : pb;
      if (pa != a) {
        if (ga != pa) atomicMin(&L[a], ga);  // path halving
        a = pa; pa = ga;
      }
      if (pb != b) {
        if (gb != pb) atomicMin(&L[b], gb);
        b = pb; pb = gb;
      }
    }
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }  // a < b: hang b under a
    const int old = atomicMin(&L[b], a);
    if (old == b) return;
    b = old;  // b was re-parented meanwhile: keep that link by uniting with it too
    pa = uf_load(L, a);
    pb = uf_load(L, b);
  }
}
// After a kernel boundary the forest is final: plain (cacheable) loads.
__device__ __forceinline__ int uf_find_final(const int* __restrict__ L, int x) {
  int p = L[x];
  while (p != x) { x = p; p = L[x]; }
  return x;
}
template <typename P>
__device__ __forceinline__ void uf_unite(P* L, int a, int b) {
  for (;;) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }  // a < b: hang b under a
    const int old = atomicMin(&L[b], a);
    if (old == b) return;
    b = old;  // b was re-parented meanwhile: keep that link by uniting with it too
  }
}

// ---------------------------------------------------------------------------
// a5 (first half): cv::Canny's Sobel 3x3 (BORDER_REPLICATE) + L2 magnitude +
// non-maximum suppression (imgpyramidrgbd.cpp:184), 64x16 tile + halo in LDS.
// Writes the map {0 none, 1 weak, 2 strong} and, per tile, resolves the
// 8-connectivity of the candidates with a union-find in LDS; the global parent
// array gets each pixel's tile-local root (as a global pixel index).
// ---------------------------------------------------------------------------
// LDS geometry (byte column bc = x - (x0 - 4), so the tile's pixels sit at bc = 4..67 and every
// 4-pixel group is word aligned):
//   s_gw  [20][19] words : gray rows y0-2 .. y0+17, bytes bc = 0..71 (replicated at the image border)
//   s_mag [18][76] ints  : |grad|^2 rows y0-1 .. y0+16, columns bc = 0..71 (0 outside the image)
//   s_dxy [18][76] ints  : dx | dy << 16
#define NMS_GW 19
#define NMS_MS 76
__global__ void __launch_bounds__(NMS_THREADS) k_canny_nms(PyrGeom g, FramePlanes pl) {
  __shared__ uint32_t s_gw[NMS_TILE_H + 4][NMS_GW];
  __shared__ __attribute__((aligned(16))) int s_mag[NMS_TILE_H + 2][NMS_MS];
  __shared__ __attribute__((aligned(16))) int s_dxy[NMS_TILE_H + 2][NMS_MS];
  __shared__ int s_lab[NMS_TILE_H * NMS_TILE_W];
  const int f = g.frame0 + blockIdx.z;
  const int l = level_of(g, blockIdx.x, &LevelGeom::tile_base);
  const LevelGeom& lv = g.lv[l];
  const int t = blockIdx.x - lv.tile_base;
  const int x0 = (t % lv.tiles_x) * NMS_TILE_W, y0 = (t / lv.tiles_x) * NMS_TILE_H;
  const int w = lv.w, h = lv.h;
  const uint8_t* gray = pl.gray[l] + (size_t)f * lv.npix;
  const int tid = threadIdx.x;

  // phase A: gray tile as aligned words (w is a multiple of 4, x0 of 64: a word is entirely inside
  // or entirely outside the image; outside = BORDER_REPLICATE of the edge pixel)
  for (int i = tid; i < (NMS_TILE_H + 4) * 18; i += NMS_THREADS) {
    const int r = i / 18, wc = i - r * 18;
    const int gy = clampi(y0 - 2 + r, 0, h - 1);
    const int gx = x0 - 4 + 4 * wc;
    uint32_t v;
    if (gx < 0) v = 0x01010101u * gray[(size_t)gy * w];
    else if (gx >= w) v = 0x01010101u * gray[(size_t)gy * w + w - 1];
    else v = *reinterpret_cast<const uint32_t*>(gray + (size_t)gy * w + gx);
    s_gw[r][wc] = v;
  }
  __syncthreads();
  // phase B: Sobel 3x3 + L2 magnitude, 4 adjacent positions per task from 9 word reads
  for (int i = tid; i < (NMS_TILE_H + 2) * 18; i += NMS_THREADS) {
    const int r = i / 18, c = i - r * 18;
    const int iy = y0 - 1 + r;
    // bytes bc = 4c-1 .. 4c+4 of gray rows r, r+1, r+2
    uint32_t lo[3], mid[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      lo[k] = c > 0 ? s_gw[r + k][c - 1] : 0u;
      mid[k] = s_gw[r + k][c];
      hi[k] = c < 17 ? s_gw[r + k][c + 1] : 0u;
    }
    int mg[4], dq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // neighbours of byte j of `mid`: left = byte j-1 (or byte 3 of lo), right = byte j+1 (or byte 0 of hi)
      int L3[3], C3[3], R3[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        L3[k] = j == 0 ? (int)(lo[k] >> 24) : (int)((mid[k] >> (8 * (j - 1))) & 255u);
        C3[k] = (int)((mid[k] >> (8 * j)) & 255u);
        R3[k] = j == 3 ? (int)(hi[k] & 255u) : (int)((mid[k] >> (8 * (j + 1))) & 255u);
      }
      const int dx = (R3[0] + 2 * R3[1] + R3[2]) - (L3[0] + 2 * L3[1] + L3[2]);
      const int dy = (L3[2] + 2 * C3[2] + R3[2]) - (L3[0] + 2 * C3[0] + R3[0]);
      const int ix = x0 - 4 + 4 * c + j;
      const bool inside = ix >= 0 && ix < w && iy >= 0 && iy < h;
      mg[j] = inside ? dx * dx + dy * dy : 0;
      dq[j] = inside ? (int)(((uint32_t)dx & 0xffffu) | ((uint32_t)dy << 16)) : 0;
    }
    *reinterpret_cast<int4*>(&s_mag[r][4 * c]) = make_int4(mg[0], mg[1], mg[2], mg[3]);
    *reinterpret_cast<int4*>(&s_dxy[r][4 * c]) = make_int4(dq[0], dq[1], dq[2], dq[3]);
  }
  __syncthreads();

  // phase C: non-maximum suppression, 4 adjacent pixels per thread and pass (a pass = 16 tile rows)
  const int TG22 = 13573;  // (int)(0.41421356...*(1<<15) + 0.5)
  const int lx0 = (tid % 16) * 4;
  const int bc0 = lx0 + 4;  // byte column of the first pixel
  int cand[NMS_PASSES][4];
#pragma unroll
  for (int ps = 0; ps < NMS_PASSES; ++ps) {
    const int ly = ps * NMS_ROWS_PER_PASS + tid / 16;
    int mrow[3][6];  // magnitudes of rows ly, ly+1, ly+2 (mag-row coords), columns bc0-1 .. bc0+4
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int4 m4 = *reinterpret_cast<const int4*>(&s_mag[ly + k][bc0]);
      mrow[k][0] = s_mag[ly + k][bc0 - 1];
      mrow[k][1] = m4.x; mrow[k][2] = m4.y; mrow[k][3] = m4.z; mrow[k][4] = m4.w;
      mrow[k][5] = s_mag[ly + k][bc0 + 4];
    }
    const int4 d4 = *reinterpret_cast<const int4*>(&s_dxy[ly + 1][bc0]);
    const int dxy4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int lx = lx0 + k;
      const int m = mrow[1][k + 1];
      int val = 0;
      if (m > g.canny_low) {
        const int dxy = dxy4[k];
        const int xs = (int)(int16_t)(dxy & 0xffff), ys = dxy >> 16;
        const int ax = abs(xs), ay = abs(ys) << 15;
        const int tg22x = ax * TG22;
        bool is_max;
        if (ay < tg22x) {
          is_max = m > mrow[1][k] && m >= mrow[1][k + 2];
        } else {
          const int tg67x = tg22x + (ax << 16);
          if (ay > tg67x) {
            is_max = m > mrow[0][k + 1] && m >= mrow[2][k + 1];
          } else {
            const bool neg = (xs ^ ys) < 0;  // s = -1: up-right / down-left; s = +1: up-left / down-right
            const int mu = neg ? mrow[0][k + 2] : mrow[0][k];
            const int md = neg ? mrow[2][k] : mrow[2][k + 2];
            is_max = m > mu && m > md;
          }
        }
        if (is_max) val = (m > g.canny_high) ? 2 : 1;
      }
      const bool inside = (x0 + lx < w) && (y0 + ly < h);
      if (!inside) val = 0;
      cand[ps][k] = val;
    }
    // Run-based initialisation: a row of the tile is handled by 16 consecutive lanes, so its 64-bit
    // candidate mask is an OR-butterfly over those lanes; every candidate starts with the FIRST
    // pixel of its horizontal run as parent.  Horizontal connectivity then needs no unions at all
    // and the vertical links below produce chains bounded by the rows of the tile.
    unsigned long long rm = (unsigned long long)((cand[ps][0] ? 1 : 0) | (cand[ps][1] ? 2 : 0) | (cand[ps][2] ? 4 : 0) |
                                                 (cand[ps][3] ? 8 : 0)) << lx0;
    rm |= __shfl_xor(rm, 1);
    rm |= __shfl_xor(rm, 2);
    rm |= __shfl_xor(rm, 4);
    rm |= __shfl_xor(rm, 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lx0 + k;
      const unsigned long long zeros_below = ~rm & ((1ull << c) - 1ull);
      const int start = zeros_below ? 64 - __clzll((long long)zeros_below) : 0;
      s_lab[ly * NMS_TILE_W + c] = cand[ps][k] ? ly * NMS_TILE_W + start : -1;
    }
  }
  __syncthreads();
  // vertical links: N if it is a candidate (then NW / NE belong to N's run), otherwise NW and NE
#pragma unroll
  for (int ps = 0; ps < NMS_PASSES; ++ps) {
    const int ly = ps * NMS_ROWS_PER_PASS + tid / 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!cand[ps][k] || ly == 0) continue;
      const int lx = lx0 + k;
      const int me = ly * NMS_TILE_W + lx;
      if (s_lab[me - NMS_TILE_W] >= 0) {
        uf_unite(s_lab, me, me - NMS_TILE_W);
      } else {
        if (lx > 0 && s_lab[me - NMS_TILE_W - 1] >= 0) uf_unite(s_lab, me, me - NMS_TILE_W - 1);
        if (lx < NMS_TILE_W - 1 && s_lab[me - NMS_TILE_W + 1] >= 0) uf_unite(s_lab, me, me - NMS_TILE_W + 1);
      }
    }
  }
  __syncthreads();
  // tile-local roots; a strong pixel marks its tile root (s_mag is free now: reused as flags)
  int* s_strong = &s_mag[0][0];  // (NMS_TILE_H+2)*NMS_MS ints >= NMS_TILE_H*NMS_TILE_W
  int root[NMS_PASSES][4];
#pragma unroll
  for (int ps = 0; ps < NMS_PASSES; ++ps) {
    const int ly = ps * NMS_ROWS_PER_PASS + tid / 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      root[ps][k] = cand[ps][k] ? uf_find(s_lab, ly * NMS_TILE_W + lx0 + k) : -1;
      s_strong[ly * NMS_TILE_W + lx0 + k] = 0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < NMS_PASSES; ++ps)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (cand[ps][k] == 2) s_strong[root[ps][k]] = 1;  // benign same-value race
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < NMS_PASSES; ++ps) {
    const int ly = ps * NMS_ROWS_PER_PASS + tid / 16;
    if (x0 + lx0 < w && y0 + ly < h) {
      const size_t pix = (size_t)(y0 + ly) * w + x0 + lx0;
      uint32_t packed = 0;
      int4 lab;
      int* lp = &lab.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int key = -1;
        uint32_t byte = (uint32_t)cand[ps][k];  // 0 none, 1 weak, 2 strong
        if (cand[ps][k]) {
          const int me = ly * NMS_TILE_W + lx0 + k;
          const int r = root[ps][k];
          key = (y0 + r / NMS_TILE_W) * w + x0 + (r % NMS_TILE_W);
          if (r == me) byte |= 8u | (s_strong[me] ? 4u : 0u);  // bit 3: tile root, bit 2: its component holds a strong pixel
        }
        lp[k] = key;
        packed |= byte << (8 * k);
      }
      *reinterpret_cast<uint32_t*>(pl.nms[l] + (size_t)f * lv.npix + pix) = packed;
      // parents are only ever read for candidates (every consumer tests the map first): skipping the
      // -1 fill of the ~92 % candidate-free groups saves most of the 4 B/px label traffic
      if (packed) *reinterpret_cast<int4*>(pl.scratch[l] + (size_t)f * lv.npix + pix) = lab;
    }
  }
}

// The three global hysteresis passes handle 16 pixels per thread (one 16-byte load of the
// map): with one byte per thread they were pure launch/latency overhead (26 M threads).
// level = blockIdx.y (block-uniform, so the geometry is read with scalar loads); blocks past
// the end of a small level exit at once.
__device__ __forceinline__ bool ccl_chunk(const PyrGeom& g, int chunk, int* l_out, int* p0_out) {
  const int l = blockIdx.y;
  if (chunk * 16 >= g.lv[l].npix) return false;
  *l_out = l;
  *p0_out = chunk * 16;
  return true;
}

// a5 (second half, 1/3): unite candidates across tile borders (global memory).  One block per
// NMS tile, one thread per BORDER pixel of the tile (top row, left column, right column): the
// unions of a horizontal edge lying on a tile's top row run in parallel instead of 48 in a row
// inside one thread.
__global__ void __launch_bounds__(64 + 2 * NMS_TILE_H) k_ccl_border(PyrGeom g, FramePlanes pl) {
  const int f = g.frame0 + blockIdx.z;
  const int l = level_of(g, blockIdx.x, &LevelGeom::tile_base);
  const LevelGeom& lv = g.lv[l];
  const int t = blockIdx.x - lv.tile_base;
  const int x0 = (t % lv.tiles_x) * NMS_TILE_W, y0 = (t / lv.tiles_x) * NMS_TILE_H;
  const int w = lv.w, h = lv.h;
  const int tid = threadIdx.x;
  int lx, ly;
  if (tid < 64) { lx = tid; ly = 0; }
  else if (tid < 64 + NMS_TILE_H) { lx = 0; ly = tid - 64; }
  else if (tid < 64 + 2 * NMS_TILE_H) { lx = NMS_TILE_W - 1; ly = tid - 64 - NMS_TILE_H; }
  else return;
  if (tid >= 64 && ly == 0) return;  // the corners belong to the top row
  const int x = x0 + lx, y = y0 + ly;
  if (x >= w || y >= h) return;
  const uint8_t* nmsp = pl.nms[l] + (size_t)f * lv.npix;
  const int p = y * w + x;
  if ((nmsp[p] & 3) == 0) return;
  int* L = pl.scratch[l] + (size_t)f * lv.npix;
  // neighbours that live in another tile: the four map bytes are fetched together (they were four
  // dependent round trips behind short-circuit conditions), then the unions run
  const bool want_w = lx == 0 && x > 0;
  const bool want_nw = y > 0 && (lx == 0 || ly == 0) && x > 0;
  const bool want_n = y > 0 && ly == 0;
  const bool want_ne = y > 0 && (lx == NMS_TILE_W - 1 || ly == 0) && x < w - 1;
  const uint8_t b_w = want_w ? nmsp[p - 1] : (uint8_t)0;
  const uint8_t b_nw = want_nw ? nmsp[p - w - 1] : (uint8_t)0;
  const uint8_t b_n = want_n ? nmsp[p - w] : (uint8_t)0;
  const uint8_t b_ne = want_ne ? nmsp[p - w + 1] : (uint8_t)0;
  if (b_w & 3) uf_unite_global(L, p, p - 1);
  if (b_nw & 3) uf_unite_global(L, p, p - w - 1);
  if (b_n & 3) uf_unite_global(L, p, p - w);
  if (b_ne & 3) uf_unite_global(L, p, p - w + 1);
}

// a5 (2/3): only TILE ROOTS (bit 3) work here: each is re-pointed straight at its global
// root (so every pixel is <= 2 hops away afterwards) and hands its tile-level "holds a
// strong pixel" bit to that root.
__global__ void __launch_bounds__(256) k_ccl_flag(PyrGeom g, FramePlanes pl) {
  const int f = g.frame0 + blockIdx.z;
  int l, p0;
  if (!ccl_chunk(g, blockIdx.x * 256 + threadIdx.x, &l, &p0)) return;
  const LevelGeom& lv = g.lv[l];
  uint8_t* nms = pl.nms[l] + (size_t)f * lv.npix;
  const uint4 m4 = *reinterpret_cast<const uint4*>(nms + p0);
  if (((m4.x | m4.y | m4.z | m4.w) & 0x08080808u) == 0) return;
  const uint32_t mw[4] = {m4.x, m4.y, m4.z, m4.w};
  int* L = pl.scratch[l] + (size_t)f * lv.npix;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const uint32_t b = (mw[k >> 2] >> (8 * (k & 3))) & 0xffu;
    if (!(b & 8u)) continue;
    const int p = p0 + k;
    const int r = uf_find_final(L, p);
    if (r != p) L[p] = r;  // any concurrent reader sees the old parent or the root: both are ancestors
    if (b & 4u) nms[r] = (uint8_t)(nms[r] | 4u);  // every writer sets bit 2, the other bits are constant
  }
}

// a5 (3/3): edge = candidate whose component holds a strong pixel.  Writes edgesPyr and its clone
// edgesOrigPyr (imgpyramidrgbd.cpp:185-186).  One block per NMS tile, 4 pixels per thread (the
// mapping of k_canny_nms).  All candidates of a tile-local component share one tile root, and after
// k_ccl_flag a tile root points straight at its global root: so only the tile's ROOTS (a handful)
// chase pointers -- two batched loads each -- and leave their verdict in LDS; every other candidate
// reads the verdict of its tile root from LDS.  (Before: every candidate pixel chased two dependent
// global loads on its own.)  A parent that path halving moved out of the tile, or onto a non-root,
// takes the global walk.
__global__ void __launch_bounds__(NMS_THREADS) k_ccl_out(PyrGeom g, FramePlanes pl) {
  __shared__ uint32_t s_verdict[NMS_TILE_H * NMS_TILE_W / 4];  // bytes: 0xff unknown, 0 weak-only, 1 holds a strong pixel
  const int f = g.frame0 + blockIdx.z;
  const int l = level_of(g, blockIdx.x, &LevelGeom::tile_base);
  const LevelGeom& lv = g.lv[l];
  const int t = blockIdx.x - lv.tile_base;
  const int x0 = (t % lv.tiles_x) * NMS_TILE_W, y0 = (t / lv.tiles_x) * NMS_TILE_H;
  const int w = lv.w, h = lv.h;
  const int tid = threadIdx.x;
  uint8_t* verdict = reinterpret_cast<uint8_t*>(s_verdict);
  const uint8_t* nms = pl.nms[l] + (size_t)f * lv.npix;
  const int* L = pl.scratch[l] + (size_t)f * lv.npix;
#pragma unroll
  for (int ps = 0; ps < NMS_PASSES; ++ps) s_verdict[ps * NMS_THREADS + tid] = 0xffffffffu;
  uint32_t m[NMS_PASSES];
  int p0[NMS_PASSES];
#pragma unroll
  for (int ps = 0; ps < NMS_PASSES; ++ps) {
    const int ly = ps * NMS_ROWS_PER_PASS + tid / 16, lx0 = (tid % 16) * 4;
    const bool inside = (x0 + lx0 < w) && (y0 + ly < h);  // w is a multiple of 4: a group is inside or outside as a whole
    p0[ps] = (y0 + ly) * w + x0 + lx0;
    m[ps] = inside ? *reinterpret_cast<const uint32_t*>(nms + p0[ps]) : 0u;
  }
  // the parents of this thread's 4 pixels (one coalesced int4, only where there is a candidate): for a
  // tile root that IS its global root after k_ccl_flag, for the others their tile root -- fetched once,
  // before the root phase, so that a block spends 3 dependent round trips (map, parents, root's map
  // byte) instead of 5
  int4 lab[NMS_PASSES];
#pragma unroll
  for (int ps = 0; ps < NMS_PASSES; ++ps)
    lab[ps] = (m[ps] & 0x03030303u) ? *reinterpret_cast<const int4*>(L + p0[ps]) : make_int4(-1, -1, -1, -1);
  __syncthreads();
  // tile roots: the strong bit of their global root (batched over the 4 pixels of the thread)
#pragma unroll
  for (int ps = 0; ps < NMS_PASSES; ++ps) {
    if ((m[ps] & 0x08080808u) == 0) continue;
    const int ly = ps * NMS_ROWS_PER_PASS + tid / 16, lx0 = (tid % 16) * 4;
    const int tl[4] = {lab[ps].x, lab[ps].y, lab[ps].z, lab[ps].w};
    int r[4];
    uint8_t fl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = ((m[ps] >> (8 * k)) & 8u) ? tl[k] : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) fl[k] = r[k] >= 0 ? nms[r[k]] : (uint8_t)0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (r[k] >= 0) verdict[ly * NMS_TILE_W + lx0 + k] = (fl[k] & 4u) ? 1 : 0;
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < NMS_PASSES; ++ps) {
    const int ly = ps * NMS_ROWS_PER_PASS + tid / 16, lx0 = (tid % 16) * 4;
    if (!((x0 + lx0 < w) && (y0 + ly < h))) continue;
    uint32_t o = 0;
    if (m[ps] & 0x03030303u) {
      const int tl[4] = {lab[ps].x, lab[ps].y, lab[ps].z, lab[ps].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t b = (m[ps] >> (8 * k)) & 0xffu;
        if ((b & 3u) == 0) continue;
        int v;
        if (b & 8u) {
          v = verdict[ly * NMS_TILE_W + lx0 + k];  // a tile root: its own verdict
        } else {
          const int py = tl[k] / w - y0, px = tl[k] % w - x0;
          v = (py >= 0 && py < NMS_TILE_H && px >= 0 && px < NMS_TILE_W) ? verdict[py * NMS_TILE_W + px] : 0xff;
        }
        if (v == 0xff) {  // parent outside the tile or not a tile root (moved by path halving): walk
          const int r = uf_find_final(L, tl[k]);
          v = (nms[r] & 4) ? 1 : 0;
        }
        if (v) o |= 0xffu << (8 * k);
      }
    }
    *reinterpret_cast<uint32_t*>(pl.edges[l] + (size_t)f * lv.npix + p0[ps]) = o;
    *reinterpret_cast<uint32_t*>(pl.edges_orig[l] + (size_t)f * lv.npix + p0[ps]) = o;
  }
}

// ---------------------------------------------------------------------------
// a6: generateDistHistogram (imgpyramidrgbd.cpp:146-172): one block per
// (level, tile row); u8 counters wrap like the reference's ++ on uchar.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_hist(PyrGeom g, FramePlanes pl, int rows_total) {
  __shared__ int s_cnt[128];
  const int f = g.frame0 + blockIdx.z;
  // decode (level, tile row) over the levels that have a histogram
  int l = -1, ty = blockIdx.x;
  for (int k = 0; k < g.n_levels; ++k) {
    if (g.lv[k].patch <= 0) continue;
    if (ty < g.lv[k].hist_h) { l = k; break; }
    ty -= g.lv[k].hist_h;
  }
  if (l < 0) return;
  const LevelGeom lv = g.lv[l];
  const int tid = threadIdx.x;
  if (tid < 128) s_cnt[tid] = 0;
  __syncthreads();
  const uint8_t* edges = pl.edges[l] + (size_t)f * lv.npix;
  const int bw = lv.hist_w * lv.patch;
  if ((bw % 16) == 0 && (lv.w % 16) == 0) {
    const int chunks = bw / 16;
    for (int i = tid; i < lv.patch * chunks; i += 256) {
      const int y = ty * lv.patch + i / chunks, x0 = (i % chunks) * 16;
      const uint4 v = *reinterpret_cast<const uint4*>(edges + (size_t)y * lv.w + x0);
      if ((v.x | v.y | v.z | v.w) == 0) continue;
      const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
      // walk the 16 pixels once: the tile index advances at tile boundaries (one division per chunk, not
      // per pixel) and a tile gets ONE LDS atomic with its count instead of one per edge pixel
      int t = x0 / lv.patch, next = (t + 1) * lv.patch, c = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (x0 + k == next) {
          if (c) atomicAdd(&s_cnt[t], c);
          c = 0;
          ++t;
          next += lv.patch;
        }
        c += ((vw[k >> 2] >> (8 * (k & 3))) & 0xffu) ? 1 : 0;
      }
      if (c) atomicAdd(&s_cnt[t], c);
    }
  } else {
    for (int i = tid; i < lv.patch * bw; i += 256) {
      const int y = ty * lv.patch + i / bw, x = i % bw;
      if (edges[(size_t)y * lv.w + x] > 0) atomicAdd(&s_cnt[x / lv.patch], 1);
    }
  }
  __syncthreads();
  int nz = 0;
  if (tid < lv.hist_w) {
    const uint8_t v = (uint8_t)(s_cnt[tid] & 255);
    pl.hist[l][(size_t)f * lv.hist_w * lv.hist_h + (size_t)ty * lv.hist_w + tid] = v;
    nz = v != 0;
  }
  const unsigned long long m = __ballot(nz);
  if ((tid & 63) == 0 && m) atomicAdd(&pl.hist_nz[f * REVO_L + l], __popcll(m));
}

// a7: fillInEdges (imgpyramidrgbd.cpp:111-145, gate 188-195).  Level l reads
// the already-filled level l-1, so one 1024-thread block per frame walks the
// levels in order.
__global__ void __launch_bounds__(1024) k_fill(PyrGeom g, FramePlanes pl) {
  const int f = g.frame0 + blockIdx.z;
  for (int l = 1; l < g.n_levels; ++l) {
    const LevelGeom lv = g.lv[l], lf = g.lv[l - 1];
    if (!(g.use_edge_hist && lv.patch > 0 && lf.patch > 0)) continue;
    const float frac = (float)pl.hist_nz[f * REVO_L + l] / (float)(lv.hist_w * lv.hist_h);
    if (frac < g.n_percentage) {
      const uint8_t* top = pl.edges[l - 1] + (size_t)f * lf.npix;
      uint8_t* mod = pl.edges[l] + (size_t)f * lv.npix;
      const uint8_t* hist = pl.hist[l] + (size_t)f * lv.hist_w * lv.hist_h;
      const double thr = g.fill_thr[l];
      // finer pixel (yy,xx) odd,odd <-> coarse pixel (yy/2, xx/2)
      const int cw = lf.w / 2, ch = lf.h / 2;
      for (int i = threadIdx.x; i < cw * ch; i += 1024) {
        const int y = i / cw, x = i % cw;
        const int yy = 2 * y + 1, xx = 2 * x + 1;
        const int ty = yy / lf.patch, tx = xx / lf.patch;
        if (ty >= lv.hist_h || tx >= lv.hist_w) continue;
        if ((double)hist[(size_t)ty * lv.hist_w + tx] < thr && top[(size_t)yy * lf.w + xx] > 0)
          mod[(size_t)y * lv.w + x] = 255;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// a8: the 3-D edge list (imgpyramidrgbd.cpp:199-226) as an ORDERED stream
// compaction: x outer / y inner (the reference's visiting order), so the list
// is identical to the reference's, element for element.
//   count : one thread per (column, 32-row chunk), adjacent threads = adjacent
//           columns -> coalesced row reads
//   scan  : one block per (frame, level), exclusive scan in column-major chunk order
//   write : same walk, emits float4 (X,Y,Z,1) at the scanned offset
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool depth_ok(float Z, float dmin, float dmax) {
  return isfinite(Z) && Z > dmin && Z < dmax;  // imgpyramidrgbd.cpp:208
}

// One block = a strip of 64 columns x all row-chunks of the level (thread = column x chunk lane,
// a wave = 64 adjacent columns of one chunk, so edge / depth rows are read 64 B at a time).  The
// strip's slots (x-major, chunk-minor: the order of the scan) are one contiguous range of the chunk /
// mask arrays and its points one contiguous range of the list, so both go through LDS and reach
// HBM coalesced.  (Thread-per-slot with direct stores measured 10x write amplification on the slot
// arrays and 3x on the points: WRITE_SIZE 45 MB and 68 MB for 5 MB and 23 MB of data.)
#define CW_COLS 64
#define CW_LANES 16                    // chunk lanes per block (1024 threads); levels with more chunks loop
#define CW_MAXCHUNK 32                 // height <= 1024
#define CW_STAGE 4096                  // points staged in LDS per strip (64 KB); denser strips store directly
template <bool WRITE>
__global__ void __launch_bounds__(CW_COLS * CW_LANES) k_compact_walk(PyrGeom g, FramePlanes pl) {
  __shared__ int s_cnt[CW_COLS * CW_MAXCHUNK];        // count pass: counts; write pass: offsets
  __shared__ unsigned s_mask[CW_COLS * CW_MAXCHUNK];
  __shared__ float4 s_pts[WRITE ? CW_STAGE : 1];
  const int f = g.frame0 + blockIdx.z;
  const int l = level_of(g, blockIdx.x, &LevelGeom::strip_base);
  const LevelGeom& lv = g.lv[l];
  const int strip = blockIdx.x - lv.strip_base;
  const int tid = threadIdx.x, xl = tid & (CW_COLS - 1), cl = tid / CW_COLS;
  const int x = strip * CW_COLS + xl;
  const int ncols = min(CW_COLS, lv.w - strip * CW_COLS);
  const int nslots = ncols * lv.nchunk;
  const size_t slot0 = (size_t)f * lv.w * lv.nchunk + (size_t)strip * CW_COLS * lv.nchunk;  // first slot of the strip
  const float* depth = pl.depth[l] + (size_t)f * lv.npix;
  if (!WRITE) {
    // count: the depth plane is only touched where there is an edge (~8 % of the pixels); the
    // validity of the 32 rows is kept as a bit mask for the write pass
    const uint8_t* edges = pl.edges[l] + (size_t)f * lv.npix;
    if (x < lv.w) {
      for (int c = cl; c < lv.nchunk; c += CW_LANES) {
        const int yb = c * lv.chunk_rows, ye = min(lv.h, yb + lv.chunk_rows);
        unsigned em = 0;
#pragma unroll 8
        for (int y = yb; y < ye; ++y) em |= (edges[(size_t)y * lv.w + x] ? 1u : 0u) << (y - yb);
        // four set bits per trip: the depth loads of a trip are independent, so a column with k edge
        // pixels costs ceil(k/4) memory round trips instead of k
        unsigned vm = 0;
        for (unsigned m = em; m;) {
          int b[4];
          float Z[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            b[j] = m ? __ffs(m) - 1 : -1;
            m &= m - 1;  // 0 stays 0
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) Z[j] = b[j] >= 0 ? depth[(size_t)(yb + b[j]) * lv.w + x] : 0.0f;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (b[j] >= 0 && depth_ok(Z[j], g.depth_min, g.depth_max)) vm |= 1u << b[j];
        }
        s_mask[xl * lv.nchunk + c] = vm;
        s_cnt[xl * lv.nchunk + c] = __popc(vm);
      }
    }
    __syncthreads();
    for (int i = tid; i < nslots; i += CW_COLS * CW_LANES) {
      pl.cmask[l][slot0 + i] = s_mask[i];
      pl.chunk[l][slot0 + i] = s_cnt[i];
    }
  } else {
    for (int i = tid; i < nslots; i += CW_COLS * CW_LANES) {
      s_mask[i] = pl.cmask[l][slot0 + i];
      s_cnt[i] = pl.chunk[l][slot0 + i];
    }
    __syncthreads();
    const int base = s_cnt[0];
    const int total = s_cnt[nslots - 1] + __popc(s_mask[nslots - 1]) - base;
    const bool staged = total <= CW_STAGE;
    float4* out = pl.pts[l] + (size_t)f * lv.npix;
    if (x < lv.w) {
      for (int c = cl; c < lv.nchunk; c += CW_LANES) {
        const int yb = c * lv.chunk_rows;
        int o = s_cnt[xl * lv.nchunk + c] - (staged ? base : 0);
        float4* dst = staged ? s_pts : out;
        for (unsigned m = s_mask[xl * lv.nchunk + c]; m;) {  // four points per trip (independent depth loads)
          int y[4];
          float Z[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            y[j] = m ? yb + __ffs(m) - 1 : -1;
            m &= m - 1;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) Z[j] = y[j] >= 0 ? depth[(size_t)y[j] * lv.w + x] : 0.0f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (y[j] < 0) continue;
            const float X = __fdiv_rn(Z[j] * ((float)x - lv.cx), lv.fx);
            const float Y = __fdiv_rn(Z[j] * ((float)y[j] - lv.cy), lv.fy);
            dst[o++] = make_float4(X, Y, Z[j], 1.0f);
          }
        }
      }
    }
    if (staged) {
      __syncthreads();
      for (int i = tid; i < total; i += CW_COLS * CW_LANES) out[base + i] = s_pts[i];
    }
  }
}

// exclusive scan of a[0..n) by one 1024-thread block; returns the total (valid in every thread)
__device__ __forceinline__ int block_exclusive_scan(int* a, int n, int* s_part) {
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int b = min(n, tid * per), e = min(n, b + per);
  int sum = 0;
  for (int i = b; i < e; ++i) sum += a[i];
  s_part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
    const int v = (tid >= off) ? s_part[tid - off] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = s_part[tid] - sum;  // exclusive prefix of this thread's segment
  for (int i = b; i < e; ++i) { const int c = a[i]; a[i] = run; run += c; }
  return s_part[1023];
}

__global__ void __launch_bounds__(1024) k_compact_scan(PyrGeom g, FramePlanes pl) {
  __shared__ int s_part[1024];
  const int f = g.frame0 + blockIdx.z, l = blockIdx.x;
  const int n = g.lv[l].w * g.lv[l].nchunk;
  const int total = block_exclusive_scan(pl.chunk[l] + (size_t)f * n, n, s_part);
  if (threadIdx.x == 0) pl.npts[f * REVO_L + l] = total;
}

// ---------------------------------------------------------------------------
// generateColoredPcl (imgpyramidrgbd.cpp:279-327), the viewer / PLY-export cloud of a keyframe:
// the BGR image pyrDown'ed to the level (same 5x5 kernel per channel), then every pixel with a
// usable depth (dense) or every edge pixel with a usable depth (sparse) becomes
// (X,Y,Z,1, r/255, g/255, b/255, 1) in the reference's x-outer / y-inner order -- the same
// chunked count / scan / write compaction as the 3-D edge list.  Off the per-frame path
// (once per keyframe, only when a model is exported).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pyrdown_bgr(const uint8_t* __restrict__ src, int w, int h,
                                                     uint8_t* __restrict__ dst) {
  const int dw = w >> 1, dh = h >> 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= dw * dh) return;
  const int x = i % dw, y = i / dw;
  const int kk[5] = {1, 4, 6, 4, 1};
  int acc[3] = {0, 0, 0};
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const uint8_t* row = src + (size_t)reflect101(2 * y + j - 2, h) * w * 3;
    int rs[3] = {0, 0, 0};
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const uint8_t* px = row + (size_t)reflect101(2 * x + t - 2, w) * 3;
      rs[0] += kk[t] * px[0]; rs[1] += kk[t] * px[1]; rs[2] += kk[t] * px[2];
    }
    acc[0] += kk[j] * rs[0]; acc[1] += kk[j] * rs[1]; acc[2] += kk[j] * rs[2];
  }
  uint8_t* o = dst + (size_t)i * 3;
  o[0] = (uint8_t)((acc[0] + 128) >> 8); o[1] = (uint8_t)((acc[1] + 128) >> 8); o[2] = (uint8_t)((acc[2] + 128) >> 8);
}

template <bool WRITE>
__global__ void __launch_bounds__(256) k_pcl_walk(LevelGeom lv, float dmin, float dmax, const float* __restrict__ depth,
                                                  const uint8_t* __restrict__ edges, const uint8_t* __restrict__ bgr,
                                                  int dense, int* __restrict__ chunk, unsigned* __restrict__ cmask,
                                                  const int* __restrict__ total, int cap, float4* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= lv.w * lv.nchunk) return;
  const int x = i % lv.w, c = i / lv.w;
  const int yb = c * lv.chunk_rows, ye = min(lv.h, yb + lv.chunk_rows);
  const size_t slot = (size_t)x * lv.nchunk + c;
  if (!WRITE) {
    unsigned vm = 0;
    for (int y = yb; y < ye; ++y) {
      const size_t p = (size_t)y * lv.w + x;
      if ((dense || edges[p]) && depth_ok(depth[p], dmin, dmax)) vm |= 1u << (y - yb);
    }
    cmask[slot] = vm;
    chunk[slot] = __popc(vm);
  } else {
    if (*total > cap) return;  // the host reports REVO_ERR_CAPACITY; nothing is written past the buffer
    int o = chunk[slot];
    for (unsigned m = cmask[slot]; m; m &= m - 1) {
      const int y = yb + __ffs(m) - 1;
      const size_t p = (size_t)y * lv.w + x;
      const float Z = depth[p];
      const float X = __fdiv_rn(Z * ((float)x - lv.cx), lv.fx);
      const float Y = __fdiv_rn(Z * ((float)y - lv.cy), lv.fy);
      const uint8_t* px = bgr + p * 3;
      out[2 * o] = make_float4(X, Y, Z, 1.0f);
      out[2 * o + 1] = make_float4(__fdiv_rn((float)px[2], 255.0f), __fdiv_rn((float)px[1], 255.0f),
                                   __fdiv_rn((float)px[0], 255.0f), 1.0f);
      ++o;
    }
  }
}

__global__ void __launch_bounds__(1024) k_pcl_scan(int* chunk, int n, int* total) {
  __shared__ int s_part[1024];
  const int t = block_exclusive_scan(chunk, n, s_part);
  if (threadIdx.x == 0) *total = t;
}

// ---------------------------------------------------------------------------
// a9: cv::distanceTransform(255-edges, L2, PRECISE) (imgpyramidrgbd.cpp:241) as
// an exact two-pass EDT.  Columns: vertical distance to the nearest edge of the
// column (int); rows: d2(x) = min_x' (x-x')^2 + g2(x') searched outwards from x
// with early termination once (x-x')^2 >= best -- exact, and short because DT
// values are small wherever there are edges.  Result sqrtf(d2) is bit-identical
// to OpenCV's (all d2 < 2^24); no edge at all -> OpenCV's 1e15f sentinel.
// ---------------------------------------------------------------------------
#define EDT_INF (1 << 29)
// 1024 threads = (columns of the strip) x (row groups); every thread owns <= 32 rows of one column and
// keeps their edge bits in a 32-bit mask: nearest edge above/below = clz / ffs on the mask, or the
// LDS carry (last/first edge row of the other groups).  One edge read, one g^2 write per pixel.
#define EDT_THREADS 1024
__global__ void __launch_bounds__(EDT_THREADS) k_edt_cols(PyrGeom g, FramePlanes pl, int f0, int fstride) {
  __shared__ int s_first[EDT_THREADS];  // [group][column]: first edge row of the group's segment (or +INF)
  __shared__ int s_last[EDT_THREADS];   // last edge row of the segment (or -INF)
  const int f = f0 + blockIdx.z * fstride;
  // decode (level, strip); taller levels use 32 groups x 32 columns, the others 16 x 64
  int l = 0, sidx = blockIdx.x, ngroups = 16, ncols = 64;
  for (int k = 0; k < g.n_levels; ++k) {
    ngroups = g.lv[k].h > 512 ? 32 : 16;
    ncols = EDT_THREADS / ngroups;
    const int ns = (g.lv[k].w + ncols - 1) / ncols;
    if (sidx < ns) { l = k; break; }
    sidx -= ns;
  }
  const LevelGeom& lv = g.lv[l];
  const int col = threadIdx.x % ncols, grp = threadIdx.x / ncols;
  const int x = sidx * ncols + col;
  const int rpg = (lv.h + ngroups - 1) / ngroups;  // <= 32 (height <= 1024)
  const int yb = min(lv.h, grp * rpg), ye = min(lv.h, yb + rpg);
  const bool in = x < lv.w;
  const uint8_t* edges = pl.edges[l] + (size_t)f * lv.npix;
  unsigned em = 0;
  if (in) {
#pragma unroll 8
    for (int y = yb; y < ye; ++y) em |= (edges[(size_t)y * lv.w + x] ? 1u : 0u) << (y - yb);
  }
  s_first[grp * ncols + col] = em ? yb + __ffs(em) - 1 : EDT_INF;
  s_last[grp * ncols + col] = em ? yb + 31 - __clz(em) : -EDT_INF;
  __syncthreads();
  if (!in) return;
  int above = -EDT_INF, below = EDT_INF;  // nearest edge rows outside this segment
  for (int k = 0; k < grp; ++k) above = max(above, s_last[k * ncols + col]);
  for (int k = ngroups - 1; k > grp; --k) below = min(below, s_first[k * ncols + col]);
  int* g2 = pl.scratch[l] + (size_t)f * lv.npix;
  for (int y = yb; y < ye; ++y) {
    const int k = y - yb;
    const unsigned lo = em & (0xffffffffu >> (31 - k));  // bits 0..k: edges at or above y (inside the segment)
    const unsigned hi = em >> k;                          // bit 0 = row y: edges at or below y
    const int up = lo ? yb + 31 - __clz(lo) : above;
    const int dn = hi ? y + __ffs(hi) - 1 : below;
    const int d_up = (up <= -EDT_INF) ? EDT_INF : (y - up);
    const int d_dn = (dn >= EDT_INF) ? EDT_INF : (dn - y);
    const int m = min(d_up, d_dn);
    g2[(size_t)y * lv.w + x] = m >= EDT_INF ? EDT_INF : m * m;
  }
}

#define EDT_MAXW REVO_MAX_WIDTH
__global__ void __launch_bounds__(256) k_edt_rows(PyrGeom g, FramePlanes pl, int f0, int fstride) {
  __shared__ int s_g2[EDT_MAXW];
  const int f = f0 + blockIdx.z * fstride;
  const int l = level_of(g, blockIdx.x, &LevelGeom::row_base);
  const LevelGeom lv = g.lv[l];
  const int y = blockIdx.x - lv.row_base;
  const int w = lv.w;
  const int* g2 = pl.scratch[l] + (size_t)f * lv.npix + (size_t)y * w;
  for (int x = threadIdx.x; x < w; x += 256) s_g2[x] = g2[x];
  __syncthreads();
  float* dt = pl.dt[l] + (size_t)f * lv.npix + (size_t)y * w;
  for (int x = threadIdx.x; x < w; x += 256) {
    int best = s_g2[x];
    // four distances per trip: eight independent LDS reads, one dependent min chain; the loop control
    // (per-lane exit -> exec-mask bookkeeping on the scalar unit) was the bound with two (PMC: 35 M SALU
    // vs 32 M VALU per launch).  Distances past the exit bound cannot win, so the result is unchanged.
    for (int d = 1; d < w; d += 4) {
      if (d * d >= best) break;
      int m = best;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dj = d + j, ddj = dj * dj;
        const int a = (x - dj >= 0) ? s_g2[x - dj] : EDT_INF, b = (x + dj < w) ? s_g2[x + dj] : EDT_INF;
        m = min(m, ddj + min(a, b));
      }
      best = m;
    }
    dt[x] = best >= EDT_INF ? sqrtf(1e15f) : sqrtf((float)best);
  }
}

// a10: buildOptimizationStructure (imgpyramidrgbd.cpp:255-276): linear sweep
// over [w, w*(h-1)), (0.5(prev-next), 0.5(up-down), dt, 0); rows 0 and h-1 zero.
__global__ void __launch_bounds__(256) k_grad_table(PyrGeom g, FramePlanes pl, int f0, int fstride) {
  const int f = f0 + blockIdx.z * fstride;
  const int l = blockIdx.y;
  const LevelGeom& lv = g.lv[l];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= lv.npix) return;
  const float* dt = pl.dt[l] + (size_t)f * lv.npix;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i >= lv.w && i < lv.w * (lv.h - 1)) {
    o.x = 0.5f * (dt[i - 1] - dt[i + 1]);
    o.y = 0.5f * (dt[i - lv.w] - dt[i + lv.w]);
    o.z = dt[i];
  }
  pl.table[l][(size_t)f * lv.npix + i] = o;
}

// ---------------------------------------------------------------------------
// a17: assessTrackingQuality's counting maps (tracker.cpp:138-176).
// ---------------------------------------------------------------------------
// The relative poses and cloud pointers of the <= 3 past frames travel in the kernel-argument
// segment: no H2D copies in front of the vote (the sequential path is latency-bound).
__global__ void __launch_bounds__(256) k_vote_mark(VoteArgs a, float fx, float fy, float cx, float cy, int W, int H,
                                                   int* marks) {
  const int c = blockIdx.y;
  const int n = *a.n[c];
  const float* R = a.RT[c];
  const float* T = R + 9;
  const float4* pts = a.pts[c];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 p = pts[i];
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) q[r] = ((R[r] * p.x + R[3 + r] * p.y) + R[6 + r] * p.z) + T[r];
    const float u = __fdiv_rn(fx * q[0], q[2]) + cx;  // tracker.cpp:153-154 operation order
    const float v = __fdiv_rn(fy * q[1], q[2]) + cy;
    if (u >= 0 && u < (float)W && v >= 0 && v < (float)H)
      atomicOr(&marks[(int)floorf(v) * W + (int)floorf(u)], 1 << c);
  }
}

// Counts, then leaves everything it used clean for the next vote: the marks it read go back to 0
// and the last block to finish moves the 8 counters into pinned host memory and zeroes them (no
// memset, no D2H copy on the stream).
__global__ void __launch_bounds__(256) k_vote_hist(int* marks, const uint8_t* edges, const float* depth, int npix,
                                                   float dmin, float dmax, int* hist8, unsigned* done, int* host_out) {
  __shared__ int s_h[8];
  __shared__ bool s_last;
  if (threadIdx.x < 8) s_h[threadIdx.x] = 0;
  __syncthreads();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    const int m = marks[i];
    if (m) marks[i] = 0;
    const float Z = depth[i];
    if (depth_ok(Z, dmin, dmax)) {
      const int val = __popc(m);
      atomicAdd(&s_h[val], 1);
      if (edges[i] > 0) atomicAdd(&s_h[4 + val], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x < 8 && s_h[threadIdx.x]) atomicAdd(&hist8[threadIdx.x], s_h[threadIdx.x]);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(done, 1u) == gridDim.x - 1);
  __syncthreads();
  if (s_last) {
    if (threadIdx.x < 8) host_out[threadIdx.x] = atomicExch(&hist8[threadIdx.x], 0);
    if (threadIdx.x == 0) *done = 0u;
    __threadfence_system();
  }
}

// mPastPcl.push_back(cloud) (tracker.cpp:219): the n valid points and the count, one launch
__global__ void __launch_bounds__(256) k_copy_cloud(float4* __restrict__ dst, const float4* __restrict__ src,
                                                    int* __restrict__ dst_n, const int* __restrict__ src_n) {
  const int n = *src_n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) *dst_n = n;
}

}  // namespace

// ============================ launchers =====================================
void launch_gray_depth(const PyrGeom& g, const FramePlanes& p, const uint8_t* d_bgr, const float* d_depth_f32,
                       const uint16_t* d_depth_u16, float u16_alpha, int B, hipStream_t s) {
  const int npix = g.lv[0].npix;
  dim3 grid((npix / 4 + 255) / 256, 1, B);
  hipLaunchKernelGGL(k_gray_depth, grid, dim3(256), 0, s, d_bgr, d_depth_f32, d_depth_u16, u16_alpha, p.gray[0],
                     p.depth[0], npix, g.frame0);
}

void launch_pyrdown(const PyrGeom& g, const FramePlanes& p, int lvl, int B, hipStream_t s) {
  const LevelGeom& d = g.lv[lvl];
  const LevelGeom& sl = g.lv[lvl - 1];
  dim3 grid(((d.w / 4) * ((d.h + 1) / 2) + 255) / 256, 1, B);
  hipLaunchKernelGGL(k_pyrdown, grid, dim3(256), 0, s, p.gray[lvl - 1], sl.w, sl.h, p.gray[lvl], d.w, d.h,
                     p.depth[lvl - 1], p.depth[lvl], g.frame0);
}

void launch_canny_nms(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s) {
  hipLaunchKernelGGL(k_canny_nms, dim3(g.total_tiles, 1, B), dim3(NMS_THREADS), 0, s, g, p);
}

void launch_ccl(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s) {
  dim3 grid((g.lv[0].npix / 16 + 255) / 256, g.n_levels, B);
  hipLaunchKernelGGL(k_ccl_border, dim3(g.total_tiles, 1, B), dim3(64 + 2 * NMS_TILE_H), 0, s, g, p);
  hipLaunchKernelGGL(k_ccl_flag, grid, dim3(256), 0, s, g, p);
  hipLaunchKernelGGL(k_ccl_out, dim3(g.total_tiles, 1, B), dim3(NMS_THREADS), 0, s, g, p);
}

void launch_hist_fill(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s) {
  int rows = 0;
  for (int l = 0; l < g.n_levels; ++l)
    if (g.lv[l].patch > 0) rows += g.lv[l].hist_h;
  if (rows == 0) return;
  hipMemsetAsync(p.hist_nz + (size_t)g.frame0 * REVO_L, 0, sizeof(int) * REVO_L * B, s);
  hipLaunchKernelGGL(k_hist, dim3(rows, 1, B), dim3(256), 0, s, g, p, rows);
  if (g.use_edge_hist && g.n_levels > 1) hipLaunchKernelGGL(k_fill, dim3(1, 1, B), dim3(1024), 0, s, g, p);
}

void launch_compact(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s) {
  dim3 grid(g.total_strips, 1, B);
  hipLaunchKernelGGL(k_compact_walk<false>, grid, dim3(CW_COLS * CW_LANES), 0, s, g, p);
  hipLaunchKernelGGL(k_compact_scan, dim3(g.n_levels, 1, B), dim3(1024), 0, s, g, p);
  hipLaunchKernelGGL(k_compact_walk<true>, grid, dim3(CW_COLS * CW_LANES), 0, s, g, p);
}

void launch_pyrdown_bgr(const uint8_t* src, int w, int h, uint8_t* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_pyrdown_bgr, dim3(((w / 2) * (h / 2) + 255) / 256), dim3(256), 0, s, src, w, h, dst);
}

void launch_colored_pcl(const PyrGeom& g, const FramePlanes& p, int frame, int lvl, int dense, const uint8_t* bgr_lvl,
                        int* chunk, unsigned* cmask, int* total, int cap, float* out8, hipStream_t s) {
  const LevelGeom& lv = g.lv[lvl];
  const float* depth = p.depth[lvl] + (size_t)frame * lv.npix;
  const uint8_t* edges = p.edges[lvl] + (size_t)frame * lv.npix;
  const int n = lv.w * lv.nchunk;
  dim3 grid((n + 255) / 256);
  hipLaunchKernelGGL(k_pcl_walk<false>, grid, dim3(256), 0, s, lv, g.depth_min, g.depth_max, depth, edges, bgr_lvl, dense,
                     chunk, cmask, total, cap, (float4*)out8);
  hipLaunchKernelGGL(k_pcl_scan, dim3(1), dim3(1024), 0, s, chunk, n, total);
  hipLaunchKernelGGL(k_pcl_walk<true>, grid, dim3(256), 0, s, lv, g.depth_min, g.depth_max, depth, edges, bgr_lvl, dense,
                     chunk, cmask, total, cap, (float4*)out8);
}

void launch_keyframe(const PyrGeom& g, const FramePlanes& p, int f0, int fstride, int count, hipStream_t s) {
  int strips = 0;
  for (int l = 0; l < g.n_levels; ++l) {
    const int ncols = EDT_THREADS / (g.lv[l].h > 512 ? 32 : 16);
    strips += (g.lv[l].w + ncols - 1) / ncols;
  }
  hipLaunchKernelGGL(k_edt_cols, dim3(strips, 1, count), dim3(EDT_THREADS), 0, s, g, p, f0, fstride);
  hipLaunchKernelGGL(k_edt_rows, dim3(g.total_rows, 1, count), dim3(256), 0, s, g, p, f0, fstride);
}

// The float4 table is only materialised for the returnOptimizationStructure accessor: the
// tracker samples the DT plane and forms the same gradients on the fly.
void launch_grad_table(const PyrGeom& g, const FramePlanes& p, int f0, int fstride, int count, hipStream_t s) {
  hipLaunchKernelGGL(k_grad_table, dim3((g.lv[0].npix + 255) / 256, g.n_levels, count), dim3(256), 0, s, g, p, f0, fstride);
}

void launch_vote(const PyrGeom& g, const FramePlanes& curr, int curr_frame, int lvl, int n_clouds, const VoteArgs& va,
                 int* d_marks, int* d_hist8, unsigned* d_done, int* h_out8, int use_orig_edges, hipStream_t s) {
  // d_marks, d_hist8 and d_done are all-zero on entry (zeroed at allocation, left clean by k_vote_hist)
  const LevelGeom& lv = g.lv[lvl];
  if (n_clouds > 0)
    hipLaunchKernelGGL(k_vote_mark, dim3(32, n_clouds), dim3(256), 0, s, va, lv.fx, lv.fy, lv.cx, lv.cy, lv.w, lv.h, d_marks);
  const uint8_t* edges = (use_orig_edges ? curr.edges_orig[lvl] : curr.edges[lvl]) + (size_t)curr_frame * lv.npix;
  hipLaunchKernelGGL(k_vote_hist, dim3(16), dim3(256), 0, s, d_marks, edges,
                     curr.depth[lvl] + (size_t)curr_frame * lv.npix, lv.npix, g.depth_min, g.depth_max, d_hist8, d_done, h_out8);
}

void launch_copy_cloud(float4* dst, const float4* src, int* dst_n, const int* src_n, hipStream_t s) {
  hipLaunchKernelGGL(k_copy_cloud, dim3(16), dim3(256), 0, s, dst, src, dst_n, src_n);
}
