// revo_pyramid.hip -- gfx950 kernels for the per-frame pyramid build
// (ImgPyramidRGBD ctor, imgpyramidrgbd.cpp:43-96,173-229) and the keyframe
// promotion (makeKeyframe, imgpyramidrgbd.cpp:231-276).
//
// Integer/byte streaming and small stencils: coalesced dword/dwordx4 row accesses, the 3x3 stencil of
// Canny entirely in registers (bitmaps out; four pixels per thread so that the kernel keeps two waves per SIMD next to
// the resident trackers), hysteresis as a union-find over weak runs in the LDS of one workgroup per (level, frame) --
// or, for levels that do not fit, per band of rows with an exact seam pass --, the 3-D edge list written TILE-ORDERED
// for the tracker (32 x 32-pixel tiles: a wavefront's points share a handful of DT rows) with the reference's
// column-major order produced on demand, 32 x 32 bit tiles transposed across lanes wherever a column-wise pass needs a
// row-wise bitmap.  One launch covers every level of every frame in the batch; grids are 1-D and ordered so that the
// 8 XCDs (ids go round-robin over them) either SHARE a frame's cache lines (frame-fastest: strips / tiles of one frame
// on one L2) or SPREAD the expensive workgroups (level-major hysteresis).  No MFMA: there is no contraction anywhere
// on this path.  What binds each kernel is tabulated in DESIGN.md 3.
//
// Exactness: every integer stage is bit-exact by construction; float stages
// keep the reference's operation order and are compiled with -ffp-contract=off.
#include "revo_dev.h"
#include <type_traits>

namespace {

// experiment knob (compile time, profiles/build_var.sh): instruction-issue priority of the build stream's kernels -- the build
// stream is the busiest of the pipelined step's four and its kernels run next to tracker waves on every CU
#ifdef REVO_BUILD_PRIO
#define BUILD_PRIO() __builtin_amdgcn_s_setprio(REVO_BUILD_PRIO)
#else
#define BUILD_PRIO()
#endif

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
// 16 pixels per thread: three 16-byte loads of BGR, one 16-byte store of gray, four float4 of depth (the level
// sizes are multiples of 16 pixels).  With 4 pixels per thread the kernel was latency-bound once level 0 of the
// depth pyramid borrows the input plane (29 us for 79 MB).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t gray4(uint32_t a, uint32_t b, uint32_t c) {
  // bytes: a = B0 G0 R0 B1 | b = G1 R1 B2 G2 | c = R2 B3 G3 R3
  const int B0 = a & 255, G0 = (a >> 8) & 255, R0 = (a >> 16) & 255, B1 = a >> 24;
  const int G1 = b & 255, R1 = (b >> 8) & 255, B2 = (b >> 16) & 255, G2 = b >> 24;
  const int R2 = c & 255, B3 = (c >> 8) & 255, G3 = (c >> 16) & 255, R3 = c >> 24;
  const uint32_t y0 = (uint32_t)(B0 * 1868 + G0 * 9617 + R0 * 4899 + 8192) >> 14;
  const uint32_t y1 = (uint32_t)(B1 * 1868 + G1 * 9617 + R1 * 4899 + 8192) >> 14;
  const uint32_t y2 = (uint32_t)(B2 * 1868 + G2 * 9617 + R2 * 4899 + 8192) >> 14;
  const uint32_t y3 = (uint32_t)(B3 * 1868 + G3 * 9617 + R3 * 4899 + 8192) >> 14;
  return y0 | (y1 << 8) | (y2 << 16) | (y3 << 24);
}
__global__ void __launch_bounds__(256) k_gray_depth(const uint8_t* __restrict__ bgr, const float* __restrict__ depth_f32,
                                                    const uint16_t* __restrict__ depth_u16, float alpha,
                                                    uint8_t* __restrict__ gray, float* __restrict__ depth_out, int npix, int frame0) {
  BUILD_PRIO();
  const int f = frame0 + blockIdx.z;
  const int g16 = blockIdx.x * 256 + threadIdx.x;
  if (g16 * 16 >= npix) return;
  const uint4* src = reinterpret_cast<const uint4*>(bgr + (size_t)f * npix * 3) + (size_t)g16 * 3;
  const uint4 p = src[0], q = src[1], r = src[2];
  reinterpret_cast<uint4*>(gray + (size_t)f * npix)[g16] =
      make_uint4(gray4(p.x, p.y, p.z), gray4(p.w, q.x, q.y), gray4(q.z, q.w, r.x), gray4(r.y, r.z, r.w));
  if (depth_out == depth_f32) return;  // level 0 borrows the input plane: nothing to copy
  float4* dst = reinterpret_cast<float4*>(depth_out + (size_t)f * npix) + (size_t)g16 * 4;
  if (depth_u16) {
    const uint4* raw = reinterpret_cast<const uint4*>(depth_u16 + (size_t)f * npix) + (size_t)g16 * 2;
    const uint4 u = raw[0], v = raw[1];
    const uint32_t wds[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 d;
      d.x = (float)(wds[2 * k] & 0xffff) * alpha + 0.0f;
      d.y = (float)(wds[2 * k] >> 16) * alpha + 0.0f;
      d.z = (float)(wds[2 * k + 1] & 0xffff) * alpha + 0.0f;
      d.w = (float)(wds[2 * k + 1] >> 16) * alpha + 0.0f;
      dst[k] = d;
    }
  } else {
    const float4* in = reinterpret_cast<const float4*>(depth_f32 + (size_t)f * npix) + (size_t)g16 * 4;
    const float4 d0 = in[0], d1 = in[1], d2 = in[2], d3 = in[3];
    dst[0] = d0; dst[1] = d1; dst[2] = d2; dst[3] = d3;
  }
}

// ---------------------------------------------------------------------------
// a3 + a4: cv::pyrDown (imgpyramidrgbd.cpp:82) + FilterSubsampleWithHoles
// (imgpyramidrgbd.h:218-249).  One thread makes a 4 x 2 block of outputs straight from aligned
// words of the source rows (7 rows x 4 words; L1/L2 serve the reuse between neighbouring threads):
// no LDS, no barrier, no per-byte address arithmetic.  (The first version staged a 67x19 byte tile
// in LDS one byte per thread-iteration with a div/mod each: ~250 VALU per output pixel.)
// Separable [1 4 6 4 1], BORDER_REFLECT_101, (sum + 128) >> 8.  Source width is a multiple of 8.
// ---------------------------------------------------------------------------
// GRAY / DEPTH (round 5): the two halves are independent -- the gray half feeds Canny, i.e. the build stream's critical chain, the
// depth half (78.6 of the 98 MB the level-1 launch reads) only feeds the edge lists.  A batch build that leaves its edge lists
// to the first consumer (REVO_DEFER >= 2) launches the gray half alone and leaves the depth half to that consumer as well
// (run_pending_edt): same threads, same arithmetic, same bits, two launches.  The single-frame API keeps the fused launch.
// Staging of the source level's edge depths (depth half of a pipelined batch, edges final): see k_edge_prefix / k_pts_tiles.
struct EdgeStage {
  const uint2* cs;              // source level's bitmaps, .y = the final edge bitmap (frame stride sh * wpr)
  const unsigned short* epre;   // edge pixels of the rows above inside the tile, per (row, word column)
  const int* base;              // first staging position of the level's tiles (this level's slice of stage_base), frame stride tiles_total
  float* out;                   // staged depths (frame stride sw * sh)
  int wpr, tiles_total;         // words per row of the source level; tiles of all levels (frame stride of base)
};
template <bool GRAY, bool DEPTH, bool STAGE = false>
__global__ void __launch_bounds__(256) k_pyrdown(const uint8_t* __restrict__ src, int sw, int sh, uint8_t* __restrict__ dst,
                                                 int dw, int dh, const float* __restrict__ dsrc, float* __restrict__ ddst, int frame0,
                                                 uint8_t* __restrict__ vsrc, float dmin, float dmax, EdgeStage es) {
  BUILD_PRIO();
  static_assert(!STAGE || DEPTH, "staging rides on the depth half");
  const int f = frame0 + blockIdx.z;
  src += (size_t)f * sw * sh;
  dst += (size_t)f * dw * dh;
  dsrc += (size_t)f * sw * sh;
  ddst += (size_t)f * dw * dh;
  vsrc += (size_t)f * (sw >> 3) * sh;
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
    if (GRAY && r < 2 * nrows + 3) {
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
    if (GRAY) {
      uint32_t packed = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int v = hs[2 * q][j] + hs[2 * q + 4][j] + 4 * (hs[2 * q + 1][j] + hs[2 * q + 3][j]) + 6 * hs[2 * q + 2][j];
        packed |= (uint32_t)((v + 128) >> 8) << (8 * j);
      }
      reinterpret_cast<uint32_t*>(dst + (size_t)(oy + q) * dw)[gx] = packed;
    }
    if (!DEPTH) continue;
    // depth: mean of the positive samples of each 2x2 block, reference order
    const float* d0 = dsrc + (size_t)(2 * (oy + q)) * sw + 8 * gx;
    const float4 a0 = *reinterpret_cast<const float4*>(d0), a1 = *reinterpret_cast<const float4*>(d0 + 4);
    const float4 c0 = *reinterpret_cast<const float4*>(d0 + sw), c1 = *reinterpret_cast<const float4*>(d0 + sw + 4);
    const float top[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float bot[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    // by-product: the SOURCE level's depth-validity bits (imgpyramidrgbd.cpp:208) -- this kernel reads every depth of
    // that level anyway, and the edge-list count pass then needs no depth reads at all (they were sparse gathers
    // that pulled the whole plane through HBM: 150 MB per launch for 41 MB of algorithmic bytes)
    unsigned vt = 0, vbm = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      vt |= (isfinite(top[k]) && top[k] > dmin && top[k] < dmax) ? 1u << k : 0u;
      vbm |= (isfinite(bot[k]) && bot[k] > dmin && bot[k] < dmax) ? 1u << k : 0u;
    }
    vsrc[(size_t)(2 * (oy + q)) * (sw >> 3) + gx] = (uint8_t)vt;
    vsrc[(size_t)(2 * (oy + q) + 1) * (sw >> 3) + gx] = (uint8_t)vbm;
    if (STAGE) {
      // the eight pixels of each of the two source rows held here: those that are EDGE pixels leave their depth at
      // base(tile) + edge pixels of the tile's rows above (epre) + edge pixels in front of them in this row's word
      const int wc = (8 * gx) >> 5, sh8 = (8 * gx) & 31;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int y = 2 * (oy + q) + rr;
        const uint32_t E = es.cs[((size_t)f * sh + y) * es.wpr + wc].y;
        uint32_t eb = (E >> sh8) & 0xffu;
        if (eb) {
          const int tile = (y >> 5) * es.wpr + wc;
          int pos = es.base[(size_t)f * es.tiles_total + tile] + (int)es.epre[((size_t)f * sh + y) * es.wpr + wc] +
                    __popc(E & ((1u << sh8) - 1u));
          float* o = es.out + (size_t)f * sw * sh;
          const float* row = rr ? bot : top;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if ((eb >> k) & 1u) o[pos++] = row[k];
        }
      }
    }
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
// a5 (first half): cv::Canny's Sobel 3x3 (BORDER_REPLICATE) + L2 magnitude +
// non-maximum suppression (imgpyramidrgbd.cpp:184).
//
// One thread = 4 pixels x NMS_R rows, everything in registers: no LDS, no barrier, no
// cross-lane traffic except the final 8-lane OR that assembles a 32-pixel bitmap word.
// Per gray row the thread loads 3 aligned words (its 4 pixels + 4 on either side; the
// neighbours' words come out of L1) and forms, with packed 16-bit arithmetic on column
// PAIRS, the horizontal difference d = g[c+1] - g[c-1] and smooth s = g[c-1] + 2 g[c] + g[c+1]
// of 6 columns (x-1 .. x+4).  Three consecutive rows give dx = d0 + 2 d1 + d2 and
// dy = s2 - s0 (exact in int16: |.| <= 1020); one v_perm packs (dx,dy) of a column and one
// v_dot2 squares it: |grad|^2 = dx^2 + dy^2.  The magnitude rows stream through registers, so NMS
// sees its 3x3 neighbourhood without ever storing a magnitude.  (Round 1 staged gray, magnitudes
// and directions of a 64x16 tile in LDS and ran a union-find there: 173 VALU + 125 SALU
// lane-instructions per pixel; this form needs ~45.)
//
// Output: per row and 32 pixels one {candidate, strong} pair of bitmap words -- 1/16 of the bytes
// of the round-1 map + label planes.  Hysteresis works on the bitmaps (k_hyst).
// ---------------------------------------------------------------------------
typedef short s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2v as_s2(uint32_t v) { return __builtin_bit_cast(s2v, v); }
__device__ __forceinline__ uint32_t as_u32(s2v v) { return __builtin_bit_cast(uint32_t, v); }
// two bytes of the 8-byte value {hi:lo} zero-extended into the halves of a dword (v_perm_b32, selector 0x0c = 0x00)
#define PAIR(hi, lo, sel) as_s2(__builtin_amdgcn_perm((hi), (lo), (sel)))

// NMS of one pixel (cv::Canny): its 3x3 neighbourhood of |grad|^2 by value, its own gradient; sets bit k of cand / strong
__device__ __forceinline__ void nms_px(int k, int aL, int aM, int aR, int bL, int m, int bR, int cL, int cM, int cR, uint32_t dxy,
                                       int low, int high, uint32_t& cb, uint32_t& sb) {
  const int TG22 = 13573;  // (int)(0.41421356...*(1<<15) + 0.5)
  const s2v v = as_s2(dxy);
  const s2v z = {0, 0};
  const uint32_t ab = as_u32(__builtin_elementwise_max(v, z - v));  // |dx| | |dy| << 16
  const int ax = (int)(ab & 0xffffu);
  const int ay15 = (int)((ab >> 1) & 0x7fff8000u);                  // |dy| << 15
  const int t22 = ax * TG22;
  const bool horiz = ay15 < t22;
  const bool vert = ay15 > t22 + (ax << 16);
  const bool neg = (int)(dxy ^ (dxy << 16)) < 0;                    // sign(dx) != sign(dy)
  const int diag_a = neg ? aR : aL;
  const int diag_b = neg ? cL : cR;
  const int a = horiz ? bL : (vert ? aM : diag_a);
  const int b = horiz ? bR : (vert ? cM : diag_b);
  const bool ge = horiz || vert;                                    // m > a && m >= b on the axes, m > both on the diagonals
  const bool is_max = (m > a) & (ge ? (m >= b) : (m > b));
  const bool c = (m > low) & is_max;
  cb |= c ? (1u << k) : 0u;
  sb |= (c & (m > high)) ? (1u << k) : 0u;
}
#define NMS_R 6

// FOUR pixels per thread (columns -1 .. 4 in flight; an 8-pixel form of the same kernel, columns -1 .. 8, existed until round 4): ~60 registers instead of
// 104.  It runs ~14 % more instructions (the halo columns are shared by fewer outputs), but the build shares every CU
// with two resident tracker workgroups (2 x 168 VGPRs per SIMD lane slot): 176 registers are left, i.e. ONE wave per SIMD
// of the 8-pixel kernel (42 us alone, 82-89 us in the pipelined step) against two or three of this one.
struct HRow4 { s2v d[3], s[3]; };  // column pairs (-1,0) (1,2) (3,4)
__device__ __forceinline__ HRow4 hrow4(uint32_t prev, uint32_t m0, uint32_t next) {  // prev = g[-4..-1], m0 = g[0..3], next = g[4..7]
  const s2v E0 = PAIR(0u, prev, 0x0c030c02u), E1 = PAIR(0u, m0, 0x0c010c00u), E2 = PAIR(0u, m0, 0x0c030c02u), E3 = PAIR(0u, next, 0x0c010c00u);
  const s2v O0 = PAIR(m0, prev, 0x0c040c03u), O1 = PAIR(0u, m0, 0x0c020c01u), O2 = PAIR(next, m0, 0x0c040c03u);
  HRow4 r;
  r.d[0] = E1 - E0; r.d[1] = E2 - E1; r.d[2] = E3 - E2;
  r.s[0] = (E0 + E1) + (O0 + O0); r.s[1] = (E1 + E2) + (O1 + O1); r.s[2] = (E2 + E3) + (O2 + O2);
  return r;
}
struct MRow4 { int v0, v1, v2, v3, v4, v5; };   // |grad|^2 of columns -1..4
struct DRow4 { uint32_t v0, v1, v2, v3; };      // (dx | dy << 16) of columns 0..3
__device__ __forceinline__ void mag_row4(const HRow4& a, const HRow4& b, const HRow4& c, uint32_t cm_left, uint32_t cm_right, bool valid,
                                         MRow4& m, DRow4& dxy) {
  uint32_t lo[3], hi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const s2v dx = (a.d[k] + c.d[k]) + (b.d[k] + b.d[k]);
    const s2v dy = c.s[k] - a.s[k];
    lo[k] = __builtin_amdgcn_perm(as_u32(dy), as_u32(dx), 0x05040100u);
    hi[k] = __builtin_amdgcn_perm(as_u32(dy), as_u32(dx), 0x07060302u);
  }
  const uint32_t rv = valid ? ~0u : 0u;
#define MAG2(x) __builtin_amdgcn_sdot2(as_s2(x), as_s2(x), 0, false)
  m.v0 = MAG2(lo[0]) & (int)(cm_left & rv);
  m.v1 = MAG2(hi[0]) & (int)rv;
  m.v2 = MAG2(lo[1]) & (int)rv;
  m.v3 = MAG2(hi[1]) & (int)rv;
  m.v4 = MAG2(lo[2]) & (int)rv;
  m.v5 = MAG2(hi[2]) & (int)(cm_right & rv);
#undef MAG2
  dxy.v0 = hi[0]; dxy.v1 = lo[1]; dxy.v2 = hi[1]; dxy.v3 = lo[2];
}
__device__ __forceinline__ void nms_row4(const MRow4& A, const MRow4& B, const MRow4& C, const DRow4& d, int low, int high,
                                         uint32_t* cand, uint32_t* strong) {
  uint32_t cb = 0, sb = 0;
  nms_px(0, A.v0, A.v1, A.v2, B.v0, B.v1, B.v2, C.v0, C.v1, C.v2, d.v0, low, high, cb, sb);
  nms_px(1, A.v1, A.v2, A.v3, B.v1, B.v2, B.v3, C.v1, C.v2, C.v3, d.v1, low, high, cb, sb);
  nms_px(2, A.v2, A.v3, A.v4, B.v2, B.v3, B.v4, C.v2, C.v3, C.v4, d.v2, low, high, cb, sb);
  nms_px(3, A.v3, A.v4, A.v5, B.v3, B.v4, B.v5, C.v3, C.v4, C.v5, d.v3, low, high, cb, sb);
  *cand = cb;
  *strong = sb;
}
__global__ void __launch_bounds__(256) k_canny_nms4(PyrGeom g, FramePlanes pl) {
  BUILD_PRIO();
  const int f = g.frame0 + blockIdx.z;
  if (blockIdx.x == 0 && threadIdx.x < REVO_L) {  // per-frame words the banded hysteresis accumulates into / raises
    pl.need_full[f * REVO_L + threadIdx.x] = 0;
    pl.hist_nz[f * REVO_L + threadIdx.x] = 0;
  }
  const int l = level_of(g, blockIdx.x, &LevelGeom::nms_block_base);
  const LevelGeom& lv = g.lv[l];
  const int w = lv.w, h = lv.h;
  const int rt = 8 * lv.wpr;  // threads per row: eight threads make one 32-pixel bitmap word
  const int t = (blockIdx.x - lv.nms_block_base) * 256 + threadIdx.x;
  const int yb = t / rt, xg = t - yb * rt;
  const int y0 = yb * NMS_R;
  if (y0 >= h) return;  // whole groups of eight leave together (rt is a multiple of 8)
  const int x = xg * 4;
  const bool active = x < w;
  const uint8_t* gray = pl.gray[l] + (size_t)f * lv.npix;
  const bool has_prev = x > 0, has_next = x + 4 < w;
  const int xo = active ? x : 0;
  const int o_prev = (active && has_prev) ? xo - 4 : xo, o_next = (active && has_next) ? xo + 4 : xo;
  const uint32_t cm_left = (active && has_prev) ? ~0u : 0u, cm_right = (active && has_next) ? ~0u : 0u;
  struct Raw { uint32_t prev, m0, next; };
  auto load_raw = [&](int r) -> Raw {
    const int rr = clampi(r, 0, h - 1);  // BORDER_REPLICATE
    const uint8_t* row = gray + (size_t)rr * w;
    Raw q;
    q.m0 = *reinterpret_cast<const uint32_t*>(row + xo);
    q.prev = *reinterpret_cast<const uint32_t*>(row + o_prev);
    q.next = *reinterpret_cast<const uint32_t*>(row + o_next);
    return q;
  };
  auto make_hrow = [&](Raw q) -> HRow4 {
    if (!has_prev) q.prev = (q.m0 & 0xffu) * 0x01010101u;
    if (!has_next) q.next = (q.m0 >> 24) * 0x01010101u;
    return hrow4(q.prev, q.m0, q.next);
  };
  HRow4 H0, H1, H2;
  MRow4 M0, M1, M2;
  DRow4 D0, D1, D2;
  {
    const Raw r0 = load_raw(y0 - 2), r1 = load_raw(y0 - 1), r2 = load_raw(y0), r3 = load_raw(y0 + 1);
    H0 = make_hrow(r0); H1 = make_hrow(r1); H2 = make_hrow(r2);
    mag_row4(H0, H1, H2, cm_left, cm_right, active && y0 - 1 >= 0, M0, D0);   // magnitude row y0 - 1
    H0 = make_hrow(r3);
  }
  Raw ahead = load_raw(y0 + 2);
  mag_row4(H1, H2, H0, cm_left, cm_right, active, M1, D1);                    // magnitude row y0
  uint2* out = pl.cs[l] + ((size_t)f * h + y0) * lv.wpr + (xg >> 3);
  const int sh = 4 * (threadIdx.x & 7);
  auto emit = [&](int i, uint32_t cb, uint32_t sb) {
    // eight threads' nibbles -> one word (DPP: two quad permutes, then the other quad through row_half_mirror)
    uint32_t cw = cb << sh, sw = sb << sh;
    cw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cw, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    sw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sw, 0xB1, 0xf, 0xf, true);
    cw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cw, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    sw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sw, 0x4E, 0xf, 0xf, true);
    cw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cw, 0x141, 0xf, 0xf, true);  // row_half_mirror: lane i <-> 7 - i
    sw |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sw, 0x141, 0xf, 0xf, true);
    if ((threadIdx.x & 7) == 0 && y0 + i < h) out[(size_t)i * lv.wpr] = make_uint2(cw, sw);
  };
#define NMS_STEP4(i, Ha, Hb, Hc, Ma, Mb, Mc, Db, Dc)                                  \
  {                                                                                   \
    Hc = make_hrow(ahead);                                                            \
    if ((i) < NMS_R - 1) ahead = load_raw(y0 + (i) + 3);                              \
    mag_row4(Ha, Hb, Hc, cm_left, cm_right, active && y0 + (i) + 1 < h, Mc, Dc);      \
    uint32_t cb, sb;                                                                  \
    nms_row4(Ma, Mb, Mc, Db, g.canny_low, g.canny_high, &cb, &sb);                    \
    emit((i), cb, sb);                                                                \
  }
  NMS_STEP4(0, H2, H0, H1, M0, M1, M2, D1, D2)
  NMS_STEP4(1, H0, H1, H2, M1, M2, M0, D2, D0)
  NMS_STEP4(2, H1, H2, H0, M2, M0, M1, D0, D1)
  NMS_STEP4(3, H2, H0, H1, M0, M1, M2, D1, D2)
  NMS_STEP4(4, H0, H1, H2, M1, M2, M0, D2, D0)
  NMS_STEP4(5, H1, H2, H0, M2, M0, M1, D0, D1)
#undef NMS_STEP4
}

// ---------------------------------------------------------------------------
// a5 (second half): hysteresis.  cv::Canny keeps a candidate iff it is 8-connected, through
// candidates, to a strong one.  One workgroup per (frame, level) holds the level's candidate bitmap C
// and the growing edge bitmap E (seeded with the strong pixels) in LDS and iterates
//      E <- E  u  fill_C( C n dilate8(E) )
// to its fixpoint: every item (a column of 32-pixel words x a strip of rows) sweeps its strip down and
// up, taking whole horizontal runs of C in one step with the carry trick
//      fill = (((C + S) ^ C) & C) | S        (and its bit-reversed twin for the other direction),
// so strong pixels -- the vast majority of an edge -- never need propagation and a weak chain costs one
// iteration per strip / word it crosses.  Items read their neighbours' words while those are being
// updated: E only ever grows and the operator is monotone, so any interleaving reaches the same
// fixpoint; an iteration in which nothing changed proves it.  (Round 1: tile-local union-find in the NMS
// kernel + three global pointer-chasing passes over a byte map and 4-byte labels: 162 us and ~450 MB per
// 64-frame launch for these passes; this one reads and writes the level once.)
// The same workgroup then writes edgesPyr (+ the edgesOrigPyr clone where fillInEdges may change the level,
// imgpyramidrgbd.cpp:185-195) and generateDistHistogram's tile counts (imgpyramidrgbd.cpp:146-172; u8
// counters wrap like the reference's ++ on uchar) straight from the bitmap.
// ---------------------------------------------------------------------------
#ifndef HYST_THREADS
#define HYST_THREADS 1024
#endif
__device__ __forceinline__ uint32_t dil3(uint32_t a, uint32_t al, uint32_t ar) {
  // a | a << 1 | a >> 1 with the neighbour words' edge bits shifted in
  return a | __builtin_amdgcn_alignbit(a, al, 31) | __builtin_amdgcn_alignbit(ar, a, 1);
}
__device__ __forceinline__ uint32_t run_fill(uint32_t C, uint32_t S) {  // S subset of C: the runs of C that hold a bit of S
  const uint32_t up = ((C + S) ^ C) & C;
  const uint32_t rC = __builtin_bitreverse32(C), rS = __builtin_bitreverse32(S);
  const uint32_t dn = __builtin_bitreverse32(((rC + rS) ^ rC) & rC);
  return up | dn | S;
}

// (-DREVO_HYST_PROFILE: thread 0's clock after every phase, one printf per level-0 frame at the end.  The printf of workgroups
// that finish early disturbs the LATE phases of the ones still running -- device printf is a host call -- so only the phases of
// undisturbed frames are meaningful: profiles/r05_hyst_phase_profile.txt shows both.)
#ifdef REVO_HYST_PROFILE
#define HP(i) if (threadIdx.x == 0) hp[i] = clock64()
#define HA(i) if (threadIdx.x == 0) { const long long now_ = clock64(); ha[i] += now_ - hlast; hlast = now_; }
#else
#define HP(i)
#define HA(i)
#endif
// E_GLOBAL (only with C_IN_LDS = false): the level's edge bitmap does not fit a CU's LDS either (levels beyond ~1280 x 990):
// E lives in the (level, frame)'s scratch plane in HBM / L2 -- same layout, same code; the workgroup's own stores are visible
// to its own loads after a barrier.  This is the exact last resort of the banded path on very large levels (a band whose runs
// exceed its label space): slow, correct, rare.
template <bool C_IN_LDS, bool E_GLOBAL = false>
__device__ __forceinline__ void hyst_level(const PyrGeom& g, const FramePlanes& pl, const int l, const int f, uint32_t* s_mem) {
  static_assert(!(C_IN_LDS && E_GLOBAL), "the candidate bitmap is in LDS only when the edge bitmap is");
#ifdef REVO_HYST_PROFILE
  long long hp[12], ha[8], hlast = 0;
  for (int i = 0; i < 12; ++i) hp[i] = 0;
  for (int i = 0; i < 8; ++i) ha[i] = 0;
  int dbg_sweeps = 0, dbg_bands = 0;
#endif
  HP(0);
  // 1-D grid, level-major: workgroup ids go round-robin over the 8 XCDs, and with (level, frame) = (x, z) every
  // level-0 workgroup -- the expensive ones -- had an id that is a multiple of n_levels = 4: all of them on XCDs 0
  // and 4, where a co-running tracker leaves 8 CUs each (200 us per launch instead of 80).  Heaviest level first.
  const LevelGeom& lv = g.lv[l];
  const int w = lv.w, h = lv.h, wpr = lv.wpr;
  const int pitch = wpr;                         // one zero row above and below (+ one pad word in front / behind)
  const int nwords = h * wpr;
  const int e_words = (h + 2) * pitch + 2;
  // C_IN_LDS: [Cl: the weak candidates, h x wpr] [union-find tables ...] ... [E, at the top]; otherwise E alone.
  // The union-find tables of a level with very many weak runs (a low-contrast 640x480 frame has ~10k) grow over
  // E: E is not needed while runs are linked, and is rebuilt as S | (C & ~Cl) from the NMS words (L2) before the
  // flag pass.  That keeps such a level in ONE band (two sweeps over three bands cost those frames -- and with
  // them the whole launch -- twice the time); levels with fewer runs never re-read anything.
  uint32_t* Cl = s_mem;
  uint32_t* Ebase = C_IN_LDS ? s_mem + (REVO_HYST_LDS_MAX / 4 - e_words)
                  : (E_GLOBAL ? reinterpret_cast<uint32_t*>(pl.scratch[l] + (size_t)f * g.lv[l].npix) : s_mem);
  uint32_t* E = Ebase + 1;                       // E[(r + 1) * pitch + c] = E(r, c), r = -1 .. h
  const uint2* cs = pl.cs[l] + (size_t)f * h * wpr;
  const int tid = threadIdx.x;
  // the zero rows above / below the level and the two pad words; everything between is written by the load
  for (int i = tid; i < pitch + 1; i += HYST_THREADS) { Ebase[i] = 0u; Ebase[e_words - 1 - i] = 0u; }
  for (int i = tid; i < nwords; i += HYST_THREADS) {
    const uint2 v = cs[i];
    E[pitch + i] = v.y;
    if (C_IN_LDS) Cl[i] = v.x & ~v.y;  // the WEAK candidates (C = Cl | S, and E holds S or more at all times)
  }
  __syncthreads();
  HP(1);
  // ---- union-find over the runs of weak pixels (the fast path) ---------------------------------------
  // Strong pixels are edges already; a weak candidate (C & ~S) becomes one iff its 8-connected component of
  // WEAK pixels touches a strong pixel.  Nodes are the horizontal runs of weak pixels inside a 32-pixel word
  // (a few thousand per level): consecutive ids from a prefix sum over the words' run counts, a lock-free
  // union-find over them in LDS (links to the previous word's run and to the runs of the row above, Komura-style
  // atomicMin), roots of components touching a strong pixel get a flag, and every run reads its root's flag:
  // a handful of barriers whatever the length of the chains (flood filling the same bitmaps needed ~27 rounds
  // of ~6 us on the bench scenes).  Words are dealt to the threads round-robin, so a long weak line is shared
  // by neighbouring lanes instead of queueing in one.  If the level has more runs than LDS can label, or its
  // candidate bitmap is not in LDS (big levels), the flood fill below does the job.
  bool done = false;
#ifdef REVO_HYST_PROFILE
  int dbg_runs = -1, dbg_t = 0, dbg_a = 0;
#endif
  if (C_IN_LDS) {
    unsigned short* Bs = reinterpret_cast<unsigned short*>(Cl + nwords);          // runs before word i (nwords + 1 entries)
    uint32_t* parent = Cl + nwords + (nwords + 2) / 2;
    // runs that fit: parent + rec while linking (over E if need be), the parents next to E afterwards
    const int table_words = (int)(REVO_HYST_LDS_MAX / 4) - (nwords + (nwords + 2) / 2);
    const int cap = min(table_words / 2, table_words - e_words);
    const int cap_keep = (table_words - e_words) / 2;  // up to here the tables end below E
    __shared__ int s_wsum[HYST_THREADS / 64];
    __shared__ int s_total, s_promoted;
#ifdef REVO_HYST_PROFILE
    __shared__ int s_dbg[4];
    if (tid < 4) s_dbg[tid] = 0;
#endif
    auto weak = [&](int wi) -> uint32_t { return Cl[wi]; };
    auto starts = [](uint32_t wk) -> uint32_t { return wk & ~(wk << 1); };         // first pixel of every run
    const float inv_wpr = 1.0f / (float)wpr;
    const uint32_t FLAG = 0x80000000u, IDM = 0x7fffffffu;
    // A level whose runs do not fit the label space is cut into bands of rows that do; a band is closed exactly
    // (its promoted pixels join E, i.e. become seeds for the bands below, closed later in the same sweep) and the
    // bands are swept again only if a band promoted pixels in its first row, next to a band closed before it --
    // normally there is one band and one sweep.
#ifdef REVO_HYST_PROFILE
    hlast = clock64();
#endif
    for (int sweep = 0; sweep < 64 && !done; ++sweep) {
#ifdef REVO_HYST_PROFILE
      ++dbg_sweeps;
#endif
      // ids: exclusive prefix of the run counts (contiguous chunks -> wave scan -> 16 wave totals)
      {
        const int wpt = (nwords + HYST_THREADS - 1) / HYST_THREADS;
        const int w0 = min(nwords, tid * wpt), w1 = min(nwords, w0 + wpt);
        int mine = 0;
        for (int wi = w0; wi < w1; ++wi) mine += __popc(starts(weak(wi)));
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(incl, o);
          if ((tid & 63) >= o) incl += v;
        }
        __syncthreads();  // (s_wsum / s_total of the previous sweep are no longer read)
        if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
        if (tid == 0) s_promoted = 0;
        __syncthreads();
        int before = incl - mine;
        for (int k = 0; k < (tid >> 6); ++k) before += s_wsum[k];
        if (tid == HYST_THREADS - 1) { s_total = before + mine; Bs[nwords] = (unsigned short)min(65535, before + mine); }
        for (int wi = w0; wi < w1; ++wi) {
          Bs[wi] = (unsigned short)before;
          before += __popc(starts(weak(wi)));
        }
        __syncthreads();
      }
      const int nruns = s_total;
      HA(0);
#ifdef REVO_HYST_PROFILE
      if (sweep == 0) dbg_runs = nruns;
#endif
      if (nruns >= 65536 || cap < 64) break;  // ids are 16 bit: leave it to the flood fill
      if (nruns == 0) { done = true; break; }
      // bands of equal height whose runs all fit
      int nb = (nruns + cap - 1) / cap, band_rows = h;
      for (;; ++nb) {
        band_rows = (h + nb - 1) / nb;
        bool fits = true;
        for (int r0 = 0; r0 < h; r0 += band_rows)
          fits = fits && ((int)Bs[min(h, r0 + band_rows) * wpr] - (int)Bs[r0 * wpr] <= cap);
        if (fits || band_rows == 1) break;
      }
      if (band_rows == 1 && nb > h) break;  // a single row beyond the label space: flood fill
      for (int r0 = 0; r0 < h; r0 += band_rows) {
        const int wa = r0 * wpr, wb = min(h, r0 + band_rows) * wpr;
        const int id0 = Bs[wa], nr = (int)Bs[wb] - id0;
        if (nr == 0) continue;
        uint32_t* rec = parent + nr;  // run -> (word << 5 | first bit): the phases below deal RUNS to the threads
        // local id of the run that holds pixel `bit` of word wi (wi inside the band)
        auto id_of = [&](int wi, int bit) -> int { return (int)Bs[wi] - id0 + __popc(starts(weak(wi)) & ((2u << bit) - 1u)) - 1; };
        // Parents are KEYS, hash15(id) << 16 | id: a root hangs under the root with the smaller key, i.e. under a
        // pseudo-random one.  Linking by raster-order id turns a vertical chain of n runs that unite concurrently
        // into a linked list of n hops (200k cycles per level on the low-contrast bench frames); random linking keeps
        // the expected depth logarithmic, and keys still only ever decrease along a path, so atomicMin halving holds.
        auto key_of = [](int x) -> uint32_t { return ((((uint32_t)x * 40503u) >> 1) & 0x7fffu) << 16 | (uint32_t)x; };
        auto find = [&](int x) -> int {
          uint32_t p2 = parent[x] & IDM;
          while ((int)(p2 & 0xffffu) != x) {
            const uint32_t gp = parent[p2 & 0xffffu] & IDM;
            if (gp != p2) atomicMin(&parent[x], gp);  // path halving (keys only ever decrease along a path)
            x = (int)(p2 & 0xffffu); p2 = gp;
          }
          return x;
        };
        auto unite = [&](int a2, int b2) {
          for (;;) {
            a2 = find(a2); b2 = find(b2);
            if (a2 == b2) return;
            uint32_t ka = key_of(a2), kb = key_of(b2);
            if (ka > kb) { const int t2 = a2; a2 = b2; b2 = t2; const uint32_t t3 = ka; ka = kb; kb = t3; }
            const uint32_t old = atomicMin(&parent[b2], ka);
            if (old == kb) return;
            b2 = (int)(old & 0xffffu);  // b2 was re-parented meanwhile: keep that link by uniting with it too
          }
        };
        auto run_at = [&](int i, int* wi_out, uint32_t* run_out) {  // rec: word << 10 | first bit << 5 | length - 1
          const uint32_t rc = rec[i];
          *wi_out = (int)(rc >> 10);
          *run_out = (0xffffffffu >> (31u - (rc & 31u))) << ((rc >> 5) & 31u);
        };
        for (int wi = wa + tid; wi < wb; wi += HYST_THREADS) {
          int me = (int)Bs[wi] - id0;
          const uint32_t wk = weak(wi);
          for (uint32_t m = starts(wk); m; m &= m - 1, ++me) {
            const int bit = __ffs(m) - 1;
            const uint32_t t2 = ~(wk >> bit);  // first zero above the run's first pixel
            const int len = t2 ? __ffs(t2) - 1 : 32 - bit;
            parent[me] = key_of(me);
            rec[me] = ((uint32_t)wi << 10) | ((uint32_t)bit << 5) | (uint32_t)(len - 1);
          }
        }
        __syncthreads();
        HA(1);
#ifdef REVO_HYST_PROFILE
        ++dbg_bands;
#endif
        // links: the run continuing from the previous word; the runs of the row above (inside the band) that touch this one
        for (int me = tid; me < nr; me += HYST_THREADS) {
          int wi; uint32_t run;
          run_at(me, &wi, &run);
          int r = (int)(((float)wi + 0.5f) * inv_wpr);
          r += (r + 1) * wpr <= wi ? 1 : (r * wpr > wi ? -1 : 0);
          const int c = wi - r * wpr;
          if ((run & 1u) && c > 0 && (weak(wi - 1) >> 31)) unite(me, id_of(wi - 1, 31));
          if (r == r0) continue;
          for (uint32_t a2 = weak(wi - wpr) & (run | (run << 1) | (run >> 1)); a2;) {  // one union per run above
            const int ab = __ffs(a2) - 1;
            a2 &= ~(a2 & ~(a2 + (1u << ab)));
            unite(me, id_of(wi - wpr, ab));
          }
          if ((run & 1u) && c > 0 && (weak(wi - wpr - 1) >> 31)) unite(me, id_of(wi - wpr - 1, 31));
          if ((run >> 31) && c < wpr - 1 && (weak(wi - wpr + 1) & 1u)) unite(me, id_of(wi - wpr + 1, 0));
        }
        __syncthreads();
        HA(2);
        // every run straight under its root (concurrent path halving flattens a list in ~log steps)
        for (int me = tid; me < nr; me += HYST_THREADS) parent[me] = key_of(find(me));
        __syncthreads();
        HA(3);
        if (nr > cap_keep) {  // the run records grew over E: rebuild it (they are dead now)
          for (int i = tid; i < pitch + 1; i += HYST_THREADS) { Ebase[i] = 0u; Ebase[e_words - 1 - i] = 0u; }
          for (int i0 = tid; i0 < nwords; i0 += 5 * HYST_THREADS) {  // five loads in flight, not one round trip per word
            uint2 v[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] = cs[min(i0 + k * HYST_THREADS, nwords - 1)];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              const int i = i0 + k * HYST_THREADS;
              if (i < nwords) E[pitch + i] = v[k].y | (v[k].x & ~Cl[i]);
            }
          }
          __syncthreads();
        }
        HA(6);
        // components that touch an edge pixel (strong, or promoted in another band): flag the root.  Words (not
        // runs) are dealt to the threads here: one neighbourhood per word, its runs found from the bitmap
        for (int wi = wa + tid; wi < wb; wi += HYST_THREADS) {
          const uint32_t wk = weak(wi);
          if (!wk) continue;
          int r = (int)(((float)wi + 0.5f) * inv_wpr);
          r += (r + 1) * wpr <= wi ? 1 : (r * wpr > wi ? -1 : 0);
          const int c = wi - r * wpr;
          const uint32_t lm = c > 0 ? ~0u : 0u, rm = c < wpr - 1 ? ~0u : 0u;
          const uint32_t* Xc = E + pitch + wi;
          const uint32_t sd = dil3(Xc[-pitch], Xc[-pitch - 1] & lm, Xc[-pitch + 1] & rm) | dil3(Xc[pitch], Xc[pitch - 1] & lm, Xc[pitch + 1] & rm) |
                              __builtin_amdgcn_alignbit(Xc[0], Xc[-1] & lm, 31) | __builtin_amdgcn_alignbit(Xc[1] & rm, Xc[0], 1);
          if (!(wk & sd)) continue;
          const uint32_t touched = run_fill(wk, wk & sd);  // the runs of the word that touch an edge pixel
          int me = (int)Bs[wi] - id0;
          for (uint32_t m = starts(wk); m; m &= m - 1, ++me)
            if (touched & m & (0u - m)) {
              uint32_t* root = &parent[parent[me] & 0xffffu];  // parent[me] is the root's key since the flattening pass
#ifdef REVO_HYST_PROFILE
              atomicAdd(&s_dbg[0], 1);
              if (!(*root & FLAG)) atomicAdd(&s_dbg[1], 1);
#endif
              if (!(*root & FLAG)) atomicOr(root, FLAG);      // (a big component is flagged by hundreds of runs: read first)
            }
        }
        __syncthreads();
        HA(4);
#ifdef REVO_HYST_PROFILE
        dbg_t = s_dbg[0]; dbg_a = s_dbg[1];
#endif
        // a weak run is an edge iff its root is flagged: it leaves the weak bitmap (E = S | (C & ~Cl))
        bool any = false;
        for (int wi = wa + tid; wi < wb; wi += HYST_THREADS) {
          const uint32_t wk = weak(wi);
          if (!wk) continue;
          int me = (int)Bs[wi] - id0;
          uint32_t prom = 0;
          for (uint32_t m = starts(wk); m; m &= m - 1, ++me)
            if (parent[parent[me] & 0xffffu] & FLAG) prom |= m & (0u - m);
          if (!prom) continue;
          const uint32_t runs = run_fill(wk, prom);
          Cl[wi] = wk & ~runs;      // this thread owns the word in this pass: plain stores
          E[pitch + wi] |= runs;
          // a promotion can only reach a band that this sweep has ALREADY closed through the band's first row
          any = any || (r0 > 0 && wi < wa + wpr);
        }
        if (any) s_promoted = 1;
        __syncthreads();
        HA(5);
      }
      if (nb == 1 || !s_promoted) done = true;
    }
    __syncthreads();
  }
  if (!done) {
  // Items = strips of rpt rows x one word column; at most HYST_THREADS of them.  After the first sweep over
  // everything an item is revisited only when it or one of its 8 neighbours changed in the previous round,
  // and the active items are packed into the first threads, so that late rounds (a weak chain creeping
  // across a few strips) cost a barrier, not a sweep over the level.
    __shared__ unsigned short s_list[HYST_THREADS];
    __shared__ unsigned char s_act[2][HYST_THREADS];
    __shared__ int s_n;
  const int max_strips = HYST_THREADS / wpr;  // wpr <= 64 (width <= 2048)
  const int rpt = (h + max_strips - 1) / max_strips;
  const int nstrips = (h + rpt - 1) / rpt;
  const int n_items = nstrips * wpr;
  s_act[0][tid] = 0; s_act[1][tid] = 0;
  if (tid == 0) s_n = 0;
  __syncthreads();
  int n_list = n_items;
#ifndef REVO_HYST_MAXIT
#define REVO_HYST_MAXIT 8192  // only guards against a hang; the fixpoint ends the loop
#endif
  for (int it = 0; it < REVO_HYST_MAXIT; ++it) {
    const int wb = it & 1;             // s_act[wb]: items that change in this round
    bool changed = false;
    if (tid < n_list) {
      const int item = it == 0 ? tid : (int)s_list[tid];
      const int strip = item / wpr, c = item - strip * wpr;
      const int r0 = strip * rpt, r1 = min(h, r0 + rpt) - 1;
      uint32_t* Ec = E + pitch + c;  // Ec[r * pitch] = E(r, c); the words left of column 0 / right of the last one are masked
      const uint32_t lm = c > 0 ? ~0u : 0u, rm = c < wpr - 1 ? ~0u : 0u;
      auto Cat = [&](int r) -> uint32_t { return C_IN_LDS ? (Cl[r * wpr + c] | Ec[r * pitch]) : cs[r * wpr + c].x; };
      // down: the row above is already final for this sweep, the row below still has last round's bits
      uint32_t prev_d = dil3(Ec[(r0 - 1) * pitch], Ec[(r0 - 1) * pitch - 1] & lm, Ec[(r0 - 1) * pitch + 1] & rm);
      uint32_t cur = Ec[r0 * pitch], cl = Ec[r0 * pitch - 1] & lm, cr = Ec[r0 * pitch + 1] & rm;
      for (int r = r0; r <= r1; ++r) {
        const uint32_t nx = Ec[(r + 1) * pitch], nl = Ec[(r + 1) * pitch - 1] & lm, nr = Ec[(r + 1) * pitch + 1] & rm;
        const uint32_t Cw = Cat(r);
        const uint32_t hz = __builtin_amdgcn_alignbit(cur, cl, 31) | __builtin_amdgcn_alignbit(cr, cur, 1);
        const uint32_t F = run_fill(Cw, cur | (Cw & (prev_d | dil3(nx, nl, nr) | hz)));
        if (F != cur) { Ec[r * pitch] = F; changed = true; }
        prev_d = dil3(F, cl, cr);
        cur = nx; cl = nl; cr = nr;
      }
      // up
      prev_d = dil3(cur, cl, cr);  // row r1 + 1, just loaded
      cur = Ec[r1 * pitch]; cl = Ec[r1 * pitch - 1] & lm; cr = Ec[r1 * pitch + 1] & rm;
      for (int r = r1; r >= r0; --r) {
        const uint32_t nx = Ec[(r - 1) * pitch], nl = Ec[(r - 1) * pitch - 1] & lm, nr = Ec[(r - 1) * pitch + 1] & rm;
        const uint32_t Cw = Cat(r);
        const uint32_t hz = __builtin_amdgcn_alignbit(cur, cl, 31) | __builtin_amdgcn_alignbit(cr, cur, 1);
        const uint32_t F = run_fill(Cw, cur | (Cw & (prev_d | dil3(nx, nl, nr) | hz)));
        if (F != cur) { Ec[r * pitch] = F; changed = true; }
        prev_d = dil3(F, cl, cr);
        cur = nx; cl = nl; cr = nr;
      }
      if (changed) s_act[wb][item] = 1;
    }
    if (!__syncthreads_or(changed ? 1 : 0)) break;  // nothing changed anywhere: the fixpoint
    // next round's work list: the items with a changed item in their 3x3 neighbourhood
    bool act = false;
    if (tid < n_items) {
      const int strip = tid / wpr, c = tid - strip * wpr;
      const unsigned char* a = s_act[wb];
#pragma unroll
      for (int ds = -1; ds <= 1; ++ds)
#pragma unroll
        for (int dc = -1; dc <= 1; ++dc) {
          const int ss = strip + ds, cc = c + dc;
          if (ss >= 0 && ss < nstrips && cc >= 0 && cc < wpr) act = act || a[ss * wpr + cc] != 0;
        }
    }
    const unsigned long long m = __ballot(act);
    int base = 0;
    if ((tid & 63) == 0 && m) base = atomicAdd(&s_n, __popcll(m));
    base = __shfl(base, 0);
    if (act) s_list[base + __popcll(m & ((1ull << (tid & 63)) - 1ull))] = (unsigned short)tid;
    s_act[wb ^ 1][tid] = 0;  // the buffer the next round writes
    __syncthreads();
    n_list = s_n;
    __syncthreads();
    if (tid == 0) s_n = 0;
  }
  __syncthreads();
  }
  __syncthreads();
  HP(5);
  // edgesPyr / edgesOrigPyr: 16 pixels per store
  uint8_t* edges = pl.edges[l] + (size_t)f * lv.npix;
  uint8_t* orig = lv.has_orig ? pl.edges_orig[l] + (size_t)f * lv.npix : nullptr;
  const bool rows16 = (w & 15) == 0;  // then a group of 16 pixels lies in one row and one half of a bitmap word
  const float inv_w = 1.0f / (float)w;
  for (int i = tid; i < lv.npix / 16; i += HYST_THREADS) {
    const int p = i * 16;
    int y = (int)(((float)p + 0.5f) * inv_w);  // p / w without the integer division ...
    y += (y + 1) * w <= p ? 1 : (y * w > p ? -1 : 0);  // ... and exact whatever the rounding did
    const int x = p - y * w;
    uint32_t o[4];
    if (rows16) {
      const uint32_t bits = (E[(y + 1) * pitch + (x >> 5)] >> (x & 31)) & 0xffffu;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = ((((bits >> (4 * q)) & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu;
    } else {  // w is a multiple of 4, npix of 16: a group may wrap into the next row(s)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int xx = x + 4 * q, yy = y;
        while (xx >= w) { xx -= w; yy += 1; }
        const uint32_t bits = (E[(yy + 1) * pitch + (xx >> 5)] >> (xx & 31)) & 0xfu;
        o[q] = ((bits * 0x00204081u) & 0x01010101u) * 0xffu;
      }
    }
    const uint4 v = make_uint4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<uint4*>(edges + p) = v;
    if (orig) *reinterpret_cast<uint4*>(orig + p) = v;
  }
  // the edge bitmap itself replaces the strong words (cs[].y): the edge-list count pass reads 32 pixels per load from it
  {
    uint2* csw = pl.cs[l] + (size_t)f * h * wpr;
    for (int i = tid; i < nwords; i += HYST_THREADS) csw[i].y = E[pitch + i];
  }
  HP(6);
  // histPyr[l] and the number of non-empty tiles (imgpyramidrgbd.cpp:146-172)
  if (lv.patch > 0) {
    const int ntiles = lv.hist_w * lv.hist_h;
    int nz = 0;
    const float inv_hw = 1.0f / (float)lv.hist_w;
    for (int t0 = 0; t0 < ntiles; t0 += HYST_THREADS) {
      const int t = t0 + tid;
      if (t < ntiles) {
        int ty = (int)(((float)t + 0.5f) * inv_hw);
        ty += (ty + 1) * lv.hist_w <= t ? 1 : (ty * lv.hist_w > t ? -1 : 0);
        const int tx = t - ty * lv.hist_w;
        const int xa = tx * lv.patch, xb = xa + lv.patch;  // [xa, xb)
        const int wa = xa >> 5, wb = (xb - 1) >> 5;        // patch <= 64: at most 3 words
        const uint32_t ma = ~0u << (xa & 31), mb = (xb & 31) ? ~0u >> (32 - (xb & 31)) : ~0u;
        const uint32_t m0 = wa == wb ? ma & mb : ma, m1 = wb > wa + 1 ? ~0u : (wb > wa ? mb : 0u), m2 = wb > wa + 1 ? mb : 0u;
        const int o1 = wb > wa ? 1 : 0, o2 = wb > wa + 1 ? 2 : 0;
        const uint32_t* row = E + (ty * lv.patch + 1) * pitch + wa;
        int cnt = 0;
        for (int y = 0; y < lv.patch; ++y, row += pitch) cnt += __popc(row[0] & m0) + __popc(row[o1] & m1) + __popc(row[o2] & m2);
        const uint8_t v = (uint8_t)(cnt & 255);
        pl.hist[l][(size_t)f * ntiles + t] = v;
        nz += v != 0;
      }
    }
    __shared__ int s_nz;  // block-wide sum of nz
    if (tid == 0) s_nz = 0;
    __syncthreads();
    if (nz) atomicAdd(&s_nz, nz);
    __syncthreads();
    if (tid == 0) pl.hist_nz[f * REVO_L + l] = s_nz;
  }
#ifdef REVO_HYST_PROFILE
  HP(7);
  if (threadIdx.x == 0 && l == 0 && f < 64)
    printf("hyst f=%d: load %lld uf_total %lld [scan %lld record %lld link %lld compress %lld flag %lld resolve %lld] runs %d sweeps %d bands %d uf %d rebuild %lld out %lld hist %lld touched %d atomics %d\n", f,
           hp[1] - hp[0], hp[5] - hp[1], ha[0], ha[1], ha[2], ha[3], ha[4], ha[5], dbg_runs, dbg_sweeps, dbg_bands, (int)done, ha[6], hp[6] - hp[5], hp[7] - hp[6], dbg_t, dbg_a);
#endif
}
// n_items = levels x frames of the launch.  only_flagged = 0: one workgroup per (level, frame), level-major (workgroup ids go
// round-robin over the 8 XCDs: with (level, frame) = (x, z) every level-0 workgroup -- the expensive ones -- had an id that is
// a multiple of n_levels = 4, i.e. all of them sat on XCDs 0 and 4).  only_flagged = 1 (launch D of the banded path): a few
// workgroups share the (level, frame) pairs a band handed over -- normally none, and then the launch costs a flag sweep.
// (Measured in round 4 and not kept, profiles/r04_ab_hyst_grouping_priority.txt: two workgroups per frame -- level 0 / the
// other levels one after the other -- so that the launch asks for 128 whole-LDS CUs instead of 256: 84.4 k against 87.2 k
// frames/s; a 512-thread, 144 KB variant that fits next to a tracker workgroup: 72 k, the kernel alone is 35 % slower.)
template <bool C_IN_LDS, bool E_GLOBAL = false>
__global__ void __launch_bounds__(HYST_THREADS) k_hyst(PyrGeom g, FramePlanes pl, int only_flagged, int n_frames) {
  BUILD_PRIO();
  extern __shared__ uint32_t s_mem[];
  const int n_items = g.n_levels * n_frames;
  if (only_flagged) {  // one parallel sweep over the flags: normally nothing is flagged and the launch ends here
    __shared__ int s_any;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    for (int item = threadIdx.x; item < n_items; item += HYST_THREADS)
      if (pl.need_full[(g.frame0 + item % n_frames) * REVO_L + item / n_frames]) s_any = 1;
    __syncthreads();
    if (!s_any) return;
  }
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int l = item / n_frames;
    const int f = g.frame0 + item % n_frames;
    if (only_flagged && !pl.need_full[f * REVO_L + l]) continue;
    hyst_level<C_IN_LDS, E_GLOBAL>(g, pl, l, f, s_mem);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// a5 (second half), BANDED: the same hysteresis with several workgroups per (level, frame).
// One 1024-thread workgroup per (level, frame) made the launch as long as its heaviest frame (80 us for ~10 000 weak runs
// at 640x480, on 64 of 256 CUs), and a level whose bitmaps + union-find tables do not fit one CU's LDS (1280x960) fell
// back to flood filling.  Now a level is cut into bands of rows; cv::Canny's result is reached exactly in three steps:
//   A  k_hyst_band : per band, the union-find over the band's weak runs (as in k_hyst), seeded by the band's strong pixels
//                    AND the strong pixels of the two neighbouring rows; locally anchored components are promoted; the band
//                    leaves its edge bitmap, every run's root, the run records and one flag bit per root in HBM scratch
//   S  k_hyst_seam : per (level, frame), one small workgroup walks the seams: a weak run in the last row of a band and a
//                    weak run in the first row of the next that touch (8-neighbourhood) belong to one component, so their
//                    roots must end with the same flag; flags are OR-ed across touching runs until nothing changes (the
//                    fixpoint is the connectivity closure over all seams, whatever zig-zag a component takes)
//   P  k_hyst_out  : per band, runs whose root got its flag from the other side of a seam are promoted, then edgesPyr
//                    (+ edgesOrigPyr), the edge bitmap (cs[].y) and the band's histogram tiles are written from LDS
// A band whose runs exceed its label space marks the (level, frame) for k_hyst, which then runs for that pair alone
// (launch D: every other workgroup of it exits at once).
// ---------------------------------------------------------------------------
#ifndef HB_THREADS
#define HB_THREADS 1024
#endif
#ifndef HB_LDS_WORDS
#define HB_LDS_WORDS 16384   // 64 KB per band workgroup: two per CU
#endif
struct BandRef { int l, R0, R1; };
__device__ __forceinline__ BandRef band_ref(const PyrGeom& g, int bi) {
  int l = 0;
#pragma unroll
  for (int k = 1; k < REVO_L; ++k)
    if (k < g.n_levels && bi >= g.lv[k].band_base) l = k;
  const LevelGeom& lv = g.lv[l];
  const int b = bi - lv.band_base;
  BandRef r;
  r.l = l; r.R0 = b * lv.band_rows; r.R1 = min(lv.h, r.R0 + lv.band_rows);
  return r;
}
// persistent record of a band in pl.scratch[l] (ints), at the band's own pixels: f * npix + R0 * w
//   [0] runs, [1] set by S: some root of this band got its flag across a seam, [2] 1 = the band's union-find is valid
//   [4 .. 4+wpr)        runs before word c of the band's FIRST row          [4+wpr .. 4+2wpr) ... of its LAST row
//   then capb/32 words of root flags, capb/2 words of 16-bit roots (one per run), capb run records
// runs a band may label: ONE value per level (every band's record has the same layout): what a full band's tables hold in
// LDS, 16-bit ids, and what fits the smallest (= last) band's share of the scratch plane
__device__ __forceinline__ int band_capb(const LevelGeom& lv) {
  const int hb = lv.band_rows, nwb = hb * lv.wpr;
  const int table_words = HB_LDS_WORDS - (nwb + (nwb + 2) / 2);
  const int e_words = (hb + 2) * lv.wpr + 2;
  int cap = min(table_words / 2, table_words - e_words);
  cap = min(cap, 65535);
  const int h_last = lv.h - (lv.nbands - 1) * lv.band_rows;
  cap = min(cap, ((h_last * lv.w - 4 - 2 * lv.wpr) / 8) * 5);
  return max(cap, 0) & ~31;
}
struct BandRec { int* hdr; int* bs_first; int* bs_last; uint32_t* flags; unsigned short* root; uint32_t* rec; };
__device__ __forceinline__ BandRec band_rec(const FramePlanes& pl, const LevelGeom& lv, int l, int f, int R0, int capb) {
  int* base = pl.scratch[l] + (size_t)f * lv.npix + (size_t)R0 * lv.w;
  BandRec r;
  r.hdr = base; r.bs_first = base + 4; r.bs_last = base + 4 + lv.wpr;
  r.flags = reinterpret_cast<uint32_t*>(base + 4 + 2 * lv.wpr);
  r.root = reinterpret_cast<unsigned short*>(base + 4 + 2 * lv.wpr + capb / 32);
  r.rec = reinterpret_cast<uint32_t*>(base + 4 + 2 * lv.wpr + capb / 32 + capb / 2);
  return r;
}

// A: one band of one (level, frame) -- the body of k_hyst_band, also called by the band workgroups of k_hyst_mixed
__device__ __forceinline__ void hyst_band(const PyrGeom& g, const FramePlanes& pl, const BandRef br, const int f, uint32_t* s_mem) {
  __shared__ int s_wsum[HB_THREADS / 64];
  __shared__ int s_total;
  const int l = br.l;
  const LevelGeom& lv = g.lv[l];
  const int h = lv.h, wpr = lv.wpr, pitch = wpr;
  const int hb = br.R1 - br.R0, nwb = hb * wpr, e_words = (hb + 2) * pitch + 2;
  const int tid = threadIdx.x;
  uint32_t* Cl = s_mem;
  uint32_t* Ebase = s_mem + (HB_LDS_WORDS - e_words);
  uint32_t* E = Ebase + 1;  // E[(r + 1) * pitch + c] = E(R0 + r, c), r = -1 .. hb: rows -1 and hb hold the neighbours' STRONG pixels
  const uint2* cs = pl.cs[l] + ((size_t)f * h + br.R0) * wpr;  // the band's first row
  const int capb = band_capb(lv);
  const BandRec rec_out = band_rec(pl, lv, l, f, br.R0, capb);
  auto load_e = [&](bool with_cl) {  // E = S (+ the halo rows) [, Cl = C & ~S]; with_cl = false: E = S | (C & ~Cl) (rebuild)
    if (tid == 0) { Ebase[0] = 0u; Ebase[e_words - 1] = 0u; }
    for (int c = tid; c < wpr; c += HB_THREADS) {
      E[c] = br.R0 > 0 ? cs[c - wpr].y : 0u;                        // row R0 - 1
      E[(hb + 1) * pitch + c] = br.R1 < h ? cs[nwb + c].y : 0u;     // row R1
    }
    for (int i0 = tid; i0 < nwb; i0 += 4 * HB_THREADS) {  // four loads in flight
      uint2 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = cs[min(i0 + k * HB_THREADS, nwb - 1)];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * HB_THREADS;
        if (i < nwb) {
          if (with_cl) { E[pitch + i] = v[k].y; Cl[i] = v[k].x & ~v[k].y; }
          else E[pitch + i] = v[k].y | (v[k].x & ~Cl[i]);
        }
      }
    }
  };
  load_e(true);
  __syncthreads();
  unsigned short* Bs = reinterpret_cast<unsigned short*>(Cl + nwb);  // runs before word i (nwb + 1 entries)
  uint32_t* parent = Cl + nwb + (nwb + 2) / 2;
  const int table_words = HB_LDS_WORDS - (nwb + (nwb + 2) / 2);
  const int cap_keep = (table_words - e_words) / 2;  // up to here the tables end below E
  auto starts = [](uint32_t wk) -> uint32_t { return wk & ~(wk << 1); };
  const float inv_wpr = 1.0f / (float)wpr;
  const uint32_t FLAG = 0x80000000u, IDM = 0x7fffffffu;
  // ids: exclusive prefix of the run counts
  {
    const int wpt = (nwb + HB_THREADS - 1) / HB_THREADS;
    const int w0 = min(nwb, tid * wpt), w1 = min(nwb, w0 + wpt);
    int mine = 0;
    for (int wi = w0; wi < w1; ++wi) mine += __popc(starts(Cl[wi]));
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if ((tid & 63) >= o) incl += v;
    }
    if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    int before = incl - mine;
    for (int k = 0; k < (tid >> 6); ++k) before += s_wsum[k];
    if (tid == HB_THREADS - 1) { s_total = before + mine; Bs[nwb] = (unsigned short)min(65535, before + mine); }
    for (int wi = w0; wi < w1; ++wi) {
      Bs[wi] = (unsigned short)before;
      before += __popc(starts(Cl[wi]));
    }
    __syncthreads();
  }
  const int nr = s_total;
  if (nr > capb) {  // beyond this band's label space: k_hyst takes the whole (level, frame)
    if (tid == 0) { pl.need_full[f * REVO_L + l] = 1; rec_out.hdr[0] = 0; rec_out.hdr[1] = 0; rec_out.hdr[2] = 0; }
    return;
  }
  uint32_t* rec = parent + nr;  // run -> word << 10 | first bit << 5 | length - 1
  if (nr > 0) {
    auto id_of = [&](int wi, int bit) -> int { return (int)Bs[wi] + __popc(starts(Cl[wi]) & ((2u << bit) - 1u)) - 1; };
    auto key_of = [](int x) -> uint32_t { return ((((uint32_t)x * 40503u) >> 1) & 0x7fffu) << 16 | (uint32_t)x; };
    auto find = [&](int x) -> int {
      uint32_t p2 = parent[x] & IDM;
      while ((int)(p2 & 0xffffu) != x) {
        const uint32_t gp = parent[p2 & 0xffffu] & IDM;
        if (gp != p2) atomicMin(&parent[x], gp);
        x = (int)(p2 & 0xffffu); p2 = gp;
      }
      return x;
    };
    auto unite = [&](int a2, int b2) {
      for (;;) {
        a2 = find(a2); b2 = find(b2);
        if (a2 == b2) return;
        uint32_t ka = key_of(a2), kb = key_of(b2);
        if (ka > kb) { const int t2 = a2; a2 = b2; b2 = t2; const uint32_t t3 = ka; ka = kb; kb = t3; }
        const uint32_t old = atomicMin(&parent[b2], ka);
        if (old == kb) return;
        b2 = (int)(old & 0xffffu);
      }
    };
    for (int wi = tid; wi < nwb; wi += HB_THREADS) {
      int me = (int)Bs[wi];
      const uint32_t wk = Cl[wi];
      for (uint32_t m = starts(wk); m; m &= m - 1, ++me) {
        const int bit = __ffs(m) - 1;
        const uint32_t t2 = ~(wk >> bit);
        const int len = t2 ? __ffs(t2) - 1 : 32 - bit;
        parent[me] = key_of(me);
        rec[me] = ((uint32_t)wi << 10) | ((uint32_t)bit << 5) | (uint32_t)(len - 1);
      }
    }
    __syncthreads();
    // links: the run continuing from the previous word; the touching runs of the row above (inside the band)
    for (int me = tid; me < nr; me += HB_THREADS) {
      const uint32_t rc = rec[me];
      const int wi = (int)(rc >> 10);
      const uint32_t run = (0xffffffffu >> (31u - (rc & 31u))) << ((rc >> 5) & 31u);
      int r = (int)(((float)wi + 0.5f) * inv_wpr);
      r += (r + 1) * wpr <= wi ? 1 : (r * wpr > wi ? -1 : 0);
      const int c = wi - r * wpr;
      if ((run & 1u) && c > 0 && (Cl[wi - 1] >> 31)) unite(me, id_of(wi - 1, 31));
      if (r == 0) continue;
      for (uint32_t a2 = Cl[wi - wpr] & (run | (run << 1) | (run >> 1)); a2;) {
        const int ab = __ffs(a2) - 1;
        a2 &= ~(a2 & ~(a2 + (1u << ab)));
        unite(me, id_of(wi - wpr, ab));
      }
      if ((run & 1u) && c > 0 && (Cl[wi - wpr - 1] >> 31)) unite(me, id_of(wi - wpr - 1, 31));
      if ((run >> 31) && c < wpr - 1 && (Cl[wi - wpr + 1] & 1u)) unite(me, id_of(wi - wpr + 1, 0));
    }
    __syncthreads();
    for (int me = tid; me < nr; me += HB_THREADS) {
      parent[me] = key_of(find(me));   // every run straight under its root
      rec_out.rec[me] = rec[me];       // (the records go to the band's record NOW: they may lie where E is rebuilt)
    }
    __syncthreads();
    if (nr > cap_keep) {  // the tables grew over E: rebuild it
      load_e(false);
      __syncthreads();
    }
    // components that touch an edge pixel (strong, in the band or in the neighbouring rows): flag the root
    for (int wi = tid; wi < nwb; wi += HB_THREADS) {
      const uint32_t wk = Cl[wi];
      if (!wk) continue;
      int r = (int)(((float)wi + 0.5f) * inv_wpr);
      r += (r + 1) * wpr <= wi ? 1 : (r * wpr > wi ? -1 : 0);
      const int c = wi - r * wpr;
      const uint32_t lm = c > 0 ? ~0u : 0u, rm = c < wpr - 1 ? ~0u : 0u;
      const uint32_t* Xc = E + pitch + wi;
      const uint32_t sd = dil3(Xc[-pitch], Xc[-pitch - 1] & lm, Xc[-pitch + 1] & rm) | dil3(Xc[pitch], Xc[pitch - 1] & lm, Xc[pitch + 1] & rm) |
                          __builtin_amdgcn_alignbit(Xc[0], Xc[-1] & lm, 31) | __builtin_amdgcn_alignbit(Xc[1] & rm, Xc[0], 1);
      if (!(wk & sd)) continue;
      const uint32_t touched = run_fill(wk, wk & sd);
      int me = (int)Bs[wi];
      for (uint32_t m = starts(wk); m; m &= m - 1, ++me)
        if (touched & m & (0u - m)) {
          uint32_t* root = &parent[parent[me] & 0xffffu];
          if (!(*root & FLAG)) atomicOr(root, FLAG);
        }
    }
    __syncthreads();
    // a weak run is an edge iff its root is flagged
    for (int wi = tid; wi < nwb; wi += HB_THREADS) {
      const uint32_t wk = Cl[wi];
      if (!wk) continue;
      int me = (int)Bs[wi];
      uint32_t prom = 0;
      for (uint32_t m = starts(wk); m; m &= m - 1, ++me)
        if (parent[parent[me] & 0xffffu] & FLAG) prom |= m & (0u - m);
      if (prom) E[pitch + wi] |= run_fill(wk, prom);  // (Cl keeps the run: its id must stay what the records say)
    }
    __syncthreads();
  }
  // ---- the band's record: edge bitmap, run count, boundary-row id bases, root of every run, run records, root flags
  uint32_t* eb = pl.ebits[l] + ((size_t)f * h + br.R0) * wpr;
  for (int i = tid; i < nwb; i += HB_THREADS) eb[i] = E[pitch + i];
  if (tid == 0) { rec_out.hdr[0] = nr; rec_out.hdr[1] = 0; rec_out.hdr[2] = 1; }
  for (int c = tid; c < wpr; c += HB_THREADS) { rec_out.bs_first[c] = Bs[c]; rec_out.bs_last[c] = Bs[(hb - 1) * wpr + c]; }
  for (int i = tid; i < (nr + 31) / 32; i += HB_THREADS) {
    uint32_t bits = 0;
    for (int k = 0; k < 32; ++k) {
      const int me = 32 * i + k;
      if (me < nr && (int)(parent[me] & 0xffffu) == me && (parent[me] & FLAG)) bits |= 1u << k;
    }
    rec_out.flags[i] = bits;
  }
  for (int me = tid; me < nr; me += HB_THREADS) rec_out.root[me] = (unsigned short)(parent[me] & 0xffffu);
}
__global__ void __launch_bounds__(HB_THREADS) k_hyst_band(PyrGeom g, FramePlanes pl) {
  extern __shared__ uint32_t s_mem[];
  // 1-D grid, band-major (level 0's bands first): consecutive ids = the same band of consecutive frames
  const int nB = gridDim.x / g.total_bands;
  hyst_band(g, pl, band_ref(g, blockIdx.x / nB), g.frame0 + blockIdx.x % nB, s_mem);
}

// MIXED (round 6): the levels fit one workgroup, but the launch lasted as long as its heaviest level-0 frames -- a low-contrast
// 640x480 frame has ~10 000 weak runs and its link phase alone is 80 k of 190 k cycles, bound by the LDS of the ONE CU it runs on
// (random-access union-find: bank conflicts), while most frames take 45-75 k (profiles/r05_hyst_phase_profile.txt).  Here every
// workgroup of frame f's level 0 first counts the frame's weak runs (the NMS bitmaps are in L2: 77 KB per frame); a frame with
// at least g.hyst_heavy_runs of them is closed by its BAND workgroups (four CUs' LDS instead of one; their records in the
// scratch plane, k_hyst_seam / k_hyst_out behind them, restricted to such frames), every other frame and every coarser level by
// one workgroup as before -- all in ONE launch, the band workgroups first.  pl.hyst_heavy[f] tells the later kernels which
// frames took the bands.  Both paths are bit-exact (the banded one is the default of big levels), so the mix is.
__device__ __forceinline__ bool hyst_frame_is_heavy(const PyrGeom& g, const FramePlanes& pl, int f) {
  __shared__ int s_cnt[HYST_THREADS / 64];
  const LevelGeom& lv = g.lv[0];
  const int nwords = lv.h * lv.wpr;
  const uint2* cs = pl.cs[0] + (size_t)f * nwords;
  int mine = 0;
  for (int i = threadIdx.x; i < nwords; i += HYST_THREADS) {
    const uint2 v = cs[i];
    const uint32_t wk = v.x & ~v.y;
    mine += __popc(wk & ~(wk << 1));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o);
  __syncthreads();  // (s_cnt of an earlier item of this workgroup is no longer read)
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = mine;
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int k = 0; k < HYST_THREADS / 64; ++k) total += s_cnt[k];
  return total >= g.hyst_heavy_runs;
}
__global__ void __launch_bounds__(HYST_THREADS) k_hyst_mixed(PyrGeom g, FramePlanes pl, int n_frames) {
  static_assert(HB_THREADS == HYST_THREADS, "one workgroup shape for both paths");
  extern __shared__ uint32_t s_mem[];
  const int nb0 = g.lv[0].nbands;
  const int n_band_wg = nb0 * n_frames;
  if ((int)blockIdx.x < n_band_wg) {  // band-major like k_hyst_band: consecutive ids = the same band of consecutive frames
    const int f = g.frame0 + blockIdx.x % n_frames;
    if (!hyst_frame_is_heavy(g, pl, f)) return;
    hyst_band(g, pl, band_ref(g, blockIdx.x / n_frames), f, s_mem);
    return;
  }
  const int item = blockIdx.x - n_band_wg;  // level-major, heaviest level first
  const int l = item / n_frames, f = g.frame0 + item % n_frames;
  if (l == 0) {
    const bool heavy = hyst_frame_is_heavy(g, pl, f);
    if (threadIdx.x == 0) pl.hyst_heavy[f] = heavy ? 1 : 0;
    if (heavy) return;
  }
  hyst_level<true>(g, pl, l, f, s_mem);
}

// S: flags across the seams of one (level, frame).  One sweep over the seams lists the touching (run above, run below) pairs
// as (band, root above, root below) in LDS -- the only part that reads HBM --, then the flags are OR-ed along the pairs
// until nothing changes.
#define HS_THREADS 256
#define HS_MAX_BANDS 32
#define HS_MAX_PAIRS 6144
// mixed = 1 (k_hyst_mixed in front): the grid covers level 0 only, and only frames whose level 0 took the bands have records
__global__ void __launch_bounds__(HS_THREADS) k_hyst_seam(PyrGeom g, FramePlanes pl, int mixed) {
  extern __shared__ uint32_t s_fl[];  // the bands' root flags, back to back (capb / 32 words per band)
  __shared__ uint2 s_pair[HS_MAX_PAIRS];
  __shared__ int s_npairs;
  __shared__ int s_changed_band[HS_MAX_BANDS];
  const int nB = mixed ? gridDim.x : gridDim.x / g.n_levels;
  const int l = blockIdx.x / nB;
  const int f = g.frame0 + blockIdx.x % nB;
  const LevelGeom& lv = g.lv[l];
  const int nb = lv.nbands;
  if (nb <= 1 || pl.need_full[f * REVO_L + l]) return;
  if (mixed && !pl.hyst_heavy[f]) return;
  const int wpr = lv.wpr, h = lv.h;
  const int tid = threadIdx.x;
  const int capb = band_capb(lv), fw = capb / 32;
  const uint2* cs = pl.cs[l] + (size_t)f * h * wpr;
  for (int b = 0; b < nb; ++b) {
    const BandRec rb = band_rec(pl, lv, l, f, b * lv.band_rows, capb);
    const int n = (rb.hdr[0] + 31) / 32;
    for (int i = tid; i < fw; i += HS_THREADS) s_fl[b * fw + i] = i < n ? rb.flags[i] : 0u;
  }
  if (tid < HS_MAX_BANDS) s_changed_band[tid] = 0;
  if (tid == 0) s_npairs = 0;
  __syncthreads();
  auto starts = [](uint32_t wk) -> uint32_t { return wk & ~(wk << 1); };
  auto weak_at = [&](int y, int c) -> uint32_t { if (c < 0 || c >= wpr) return 0u; const uint2 v = cs[(size_t)y * wpr + c]; return v.x & ~v.y; };
  for (int item = tid; item < (nb - 1) * wpr; item += HS_THREADS) {
    const int sm = item / wpr, c = item - sm * wpr;
    const int yb = (sm + 1) * lv.band_rows, ya = yb - 1;      // last row of band sm, first row of band sm + 1
    const uint32_t Wa = weak_at(ya, c);
    const uint32_t Wl = weak_at(yb, c - 1), Wm = weak_at(yb, c), Wr = weak_at(yb, c + 1);
    if (!Wa || !(Wm | (Wl >> 31) | (Wr & 1u))) continue;
    const BandRec ra = band_rec(pl, lv, l, f, sm * lv.band_rows, capb);
    const BandRec rbb = band_rec(pl, lv, l, f, yb, capb);
    int ida = ra.bs_last[c];
    const int idm = rbb.bs_first[c], idl = c > 0 ? rbb.bs_first[c - 1] : 0, idr = c < wpr - 1 ? rbb.bs_first[c + 1] : 0;
    for (uint32_t m = starts(Wa); m; m &= m - 1, ++ida) {
      const int bit = __ffs(m) - 1;
      const uint32_t t2 = ~(Wa >> bit);
      const int len = t2 ? __ffs(t2) - 1 : 32 - bit;
      const uint32_t ma = (0xffffffffu >> (32 - len)) << bit;
      int roota = -1;
      // the runs of row yb this run touches: in its own word column, and the neighbours' last / first pixel
      for (int side = 0; side < 3; ++side) {
        int idb0; uint32_t Wb, hits;
        if (side == 0) { Wb = Wm; hits = Wm & (ma | (ma << 1) | (ma >> 1)); idb0 = idm; }
        else if (side == 1) { Wb = Wl; hits = (ma & 1u) ? (Wl & 0x80000000u) : 0u; idb0 = idl; }
        else { Wb = Wr; hits = (ma >> 31) ? (Wr & 1u) : 0u; idb0 = idr; }
        while (hits) {
          const int hb2 = __ffs(hits) - 1;
          const uint32_t upto = (2u << hb2) - 1u;                      // pixels 0 .. hb2
          const int k = __popc(starts(Wb) & upto) - 1;                 // index of the run that holds pixel hb2
          const uint32_t gap = ~Wb & upto;                             // holes up to hb2
          const int first = gap ? 32 - __clz(gap) : 0;                 // first pixel of that run
          const uint32_t t3 = ~(Wb >> first);
          const int lenb = t3 ? __ffs(t3) - 1 : 32 - first;
          hits &= ~((0xffffffffu >> (32 - lenb)) << first);
          if (roota < 0) roota = ra.root[ida];
          const int rootb = rbb.root[idb0 + k];
          const int slot = atomicAdd(&s_npairs, 1);
          if (slot < HS_MAX_PAIRS) s_pair[slot] = make_uint2((uint32_t)sm, (uint32_t)roota | ((uint32_t)rootb << 16));
        }
      }
    }
  }
  __syncthreads();
  const int np = s_npairs;
  if (np > HS_MAX_PAIRS) {  // (a seam with thousands of weak contacts: noise) -- k_hyst takes the whole (level, frame)
    if (tid == 0) pl.need_full[f * REVO_L + l] = 1;
    return;
  }
  for (int it = 0; it < 65536; ++it) {
    bool changed = false;
    for (int i = tid; i < np; i += HS_THREADS) {
      const uint2 pr = s_pair[i];
      const int sm = (int)pr.x, roota = (int)(pr.y & 0xffffu), rootb = (int)(pr.y >> 16);
      uint32_t* fa = s_fl + sm * fw;
      uint32_t* fb = s_fl + (sm + 1) * fw;
      const bool A = (fa[roota >> 5] >> (roota & 31)) & 1u, Bf = (fb[rootb >> 5] >> (rootb & 31)) & 1u;
      if (A != Bf) {
        if (!A) { atomicOr(&fa[roota >> 5], 1u << (roota & 31)); s_changed_band[sm] = 1; }
        else { atomicOr(&fb[rootb >> 5], 1u << (rootb & 31)); s_changed_band[sm + 1] = 1; }
        changed = true;
      }
    }
    if (!__syncthreads_or(changed ? 1 : 0)) break;
  }
  __syncthreads();
  for (int b = 0; b < nb; ++b) {
    if (!s_changed_band[b]) continue;
    const BandRec rb = band_rec(pl, lv, l, f, b * lv.band_rows, capb);
    const int n = (rb.hdr[0] + 31) / 32;
    for (int i = tid; i < n; i += HS_THREADS) rb.flags[i] = s_fl[b * fw + i];
    if (tid == 0) rb.hdr[1] = 1;
  }
}

// P: promotions that came across a seam, then the band's outputs
#define HO_THREADS 512
__global__ void __launch_bounds__(HO_THREADS) k_hyst_out(PyrGeom g, FramePlanes pl, int mixed) {
  extern __shared__ uint32_t s_mem[];
  const int nB = gridDim.x / (mixed ? g.lv[0].nbands : g.total_bands);  // mixed: the bands of level 0 only (they come first)
  const BandRef br = band_ref(g, blockIdx.x / nB);
  const int f = g.frame0 + blockIdx.x % nB;
  const int l = br.l;
  if (pl.need_full[f * REVO_L + l]) return;
  if (mixed && !pl.hyst_heavy[f]) return;
  const LevelGeom& lv = g.lv[l];
  const int w = lv.w, h = lv.h, wpr = lv.wpr, pitch = wpr;
  const int hb = br.R1 - br.R0, nwb = hb * wpr;
  const int tid = threadIdx.x;
  uint32_t* E = s_mem;                    // E[r * pitch + c], r = 0 .. hb - 1 (the band's own rows)
  uint32_t* eb = pl.ebits[l] + ((size_t)f * h + br.R0) * wpr;
  for (int i = tid; i < nwb; i += HO_THREADS) E[i] = eb[i];
  const BandRec rb = band_rec(pl, lv, l, f, br.R0, band_capb(lv));
  const bool promote = rb.hdr[1] != 0;
  __syncthreads();
  if (promote) {
    const int nr = rb.hdr[0];
    for (int me = tid; me < nr; me += HO_THREADS) {
      const int root = rb.root[me];
      if ((rb.flags[root >> 5] >> (root & 31)) & 1u) {
        const uint32_t rc = rb.rec[me];
        atomicOr(&E[rc >> 10], (0xffffffffu >> (31u - (rc & 31u))) << ((rc >> 5) & 31u));
      }
    }
    __syncthreads();
  }
  // edgesPyr / edgesOrigPyr: 16 pixels per store (band heights are multiples of 4 rows, widths of 4 pixels)
  uint8_t* edges = pl.edges[l] + (size_t)f * lv.npix + (size_t)br.R0 * w;
  uint8_t* orig = lv.has_orig ? pl.edges_orig[l] + (size_t)f * lv.npix + (size_t)br.R0 * w : nullptr;
  const bool rows16 = (w & 15) == 0;
  const float inv_w = 1.0f / (float)w;
  for (int i = tid; i < hb * w / 16; i += HO_THREADS) {
    const int p = i * 16;
    int y = (int)(((float)p + 0.5f) * inv_w);
    y += (y + 1) * w <= p ? 1 : (y * w > p ? -1 : 0);
    const int x = p - y * w;
    uint32_t o[4];
    if (rows16) {
      const uint32_t bits = (E[y * pitch + (x >> 5)] >> (x & 31)) & 0xffffu;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = ((((bits >> (4 * q)) & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int xx = x + 4 * q, yy = y;
        while (xx >= w) { xx -= w; yy += 1; }
        const uint32_t bits = (E[yy * pitch + (xx >> 5)] >> (xx & 31)) & 0xfu;
        o[q] = ((bits * 0x00204081u) & 0x01010101u) * 0xffu;
      }
    }
    const uint4 v = make_uint4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<uint4*>(edges + p) = v;
    if (orig) *reinterpret_cast<uint4*>(orig + p) = v;
  }
  {
    uint2* csw = pl.cs[l] + ((size_t)f * h + br.R0) * wpr;
    for (int i = tid; i < nwb; i += HO_THREADS) csw[i].y = E[i];
  }
  // the band's histogram tiles (band heights are multiples of the patch size; imgpyramidrgbd.cpp:146-172)
  if (lv.patch > 0) {
    const int ty0 = br.R0 / lv.patch, ty1 = min(lv.hist_h, br.R1 / lv.patch);
    const int ntiles = max(0, ty1 - ty0) * lv.hist_w;
    int nz = 0;
    for (int t = tid; t < ntiles; t += HO_THREADS) {
      const int ty = ty0 + t / lv.hist_w, tx = t % lv.hist_w;
      const int xa = tx * lv.patch, xb = xa + lv.patch;
      const int wa = xa >> 5, wb = (xb - 1) >> 5;
      const uint32_t ma = ~0u << (xa & 31), mb = (xb & 31) ? ~0u >> (32 - (xb & 31)) : ~0u;
      const uint32_t m0 = wa == wb ? ma & mb : ma, m1 = wb > wa + 1 ? ~0u : (wb > wa ? mb : 0u), m2 = wb > wa + 1 ? mb : 0u;
      const int o1 = wb > wa ? 1 : 0, o2 = wb > wa + 1 ? 2 : 0;
      const uint32_t* row = E + (ty * lv.patch - br.R0) * pitch + wa;
      int cnt = 0;
      for (int y = 0; y < lv.patch; ++y, row += pitch) cnt += __popc(row[0] & m0) + __popc(row[o1] & m1) + __popc(row[o2] & m2);
      const uint8_t v = (uint8_t)(cnt & 255);
      pl.hist[l][(size_t)f * lv.hist_w * lv.hist_h + (size_t)ty * lv.hist_w + tx] = v;
      nz += v != 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nz += __shfl_xor(nz, o);
    if ((tid & 63) == 0 && nz) atomicAdd(&pl.hist_nz[f * REVO_L + l], nz);  // (zeroed by k_canny_nms)
  }
}

// a7: fillInEdges (imgpyramidrgbd.cpp:111-145, gate 188-195).  Level l reads
// the already-filled level l-1.  only_level = 0: one 1024-thread block per frame walks the
// levels in order (640x480: the gate rarely opens and the launch costs 5 us).  only_level = l > 0: that level alone,
// gridDim.x workgroups per frame share its pixels (large images: one block per frame took 205 us of a 1.19 ms build at
// 1280x960, where the bench frames do open the gate; the launcher then enqueues the levels one after the other).
__global__ void __launch_bounds__(1024) k_fill(PyrGeom g, FramePlanes pl, int only_level) {
  BUILD_PRIO();
  const int f = g.frame0 + blockIdx.z;
  const int la = only_level ? only_level : 1, lb = only_level ? only_level + 1 : g.n_levels;
  for (int l = la; l < lb; ++l) {
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
      for (int i = blockIdx.x * 1024 + threadIdx.x; i < cw * ch; i += gridDim.x * 1024) {
        const int y = i / cw, x = i % cw;
        const int yy = 2 * y + 1, xx = 2 * x + 1;
        const int ty = yy / lf.patch, tx = xx / lf.patch;
        if (ty >= lv.hist_h || tx >= lv.hist_w) continue;
        if ((double)hist[(size_t)ty * lv.hist_w + tx] < thr && top[(size_t)yy * lf.w + xx] > 0) {
          mod[(size_t)y * lv.w + x] = 255;
          // ... and in the level's edge bitmap (k_hyst left it in cs[].y), which the edge-list count pass reads
          atomicOr(&(pl.cs[l] + (size_t)f * lv.h * lv.wpr)[(size_t)y * lv.wpr + (x >> 5)].y, 1u << (x & 31));
        }
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
//   write : same walk; offsets = the point counts of the level's earlier strips (one int per strip, left by the
//           count pass) + an exclusive scan of the strip's own slot counts in LDS; emits float4 (X,Y,Z,1).
//           (Round 1 ran a scan kernel per (frame, level) between the passes: 10 us alone, 28 us next to a tracker.)
// ---------------------------------------------------------------------------
// 32 x 32 bit tile held one row per lane (lane r of a 32-lane half-wave: bit c = pixel (r, c)) -> one column per lane
// (lane c: bit r): the recursive block swap, five exchanges with lane ^ k
__device__ __forceinline__ uint32_t tile_transpose32(uint32_t v, int r) {
#pragma unroll
  for (int k = 16; k >= 1; k >>= 1) {
    const uint32_t mk = k == 16 ? 0x0000ffffu : k == 8 ? 0x00ff00ffu : k == 4 ? 0x0f0f0f0fu : k == 2 ? 0x33333333u : 0x55555555u;
    const uint32_t t = (uint32_t)__shfl_xor((int)v, k);
    v = (r & k) ? ((v & ~mk) | ((t >> k) & mk)) : ((v & mk) | ((t & mk) << k));
  }
  return v;
}

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
#define CW_LANES 16                    // chunk lanes per block of the write pass (1024 threads); levels with more chunks loop
#ifndef CW_LANES_COUNT
#define CW_LANES_COUNT 8               // count pass: 512 threads x 48 VGPRs fit next to a resident tracker workgroup (176 free
#endif                                 // VGPRs per SIMD); 1024 x 48 do not, and the pass then waited for the tracker's CUs
#define CW_MAXCHUNK 64                 // height <= 2048 (accessor-only kernel since round 3: its LDS footprint no longer matters)
#define CW_STAGE 4096                  // points staged in LDS per strip (64 KB); denser strips store directly
template <bool WRITE>
__global__ void __launch_bounds__(CW_COLS * (WRITE ? CW_LANES : CW_LANES_COUNT)) k_compact_walk(PyrGeom g, FramePlanes pl) {
  constexpr int LANES = WRITE ? CW_LANES : CW_LANES_COUNT;
  __shared__ int s_cnt[CW_COLS * CW_MAXCHUNK];        // count pass: counts; write pass: offsets
  __shared__ unsigned s_mask[CW_COLS * CW_MAXCHUNK];
  __shared__ float4 s_pts[WRITE ? CW_STAGE : 1];
  // 1-D grid, frame fastest: all strips of a frame get ids that are congruent mod 8 (batches are multiples of 8 frames
  // wide in practice), i.e. land on ONE XCD and share its L2 -- they read the same lines of the frame's bitmaps, and
  // with (strip, frame) = (x, z) every line was fetched into up to eight L2s
  const int nB = gridDim.x / g.total_strips;
  const int sx = blockIdx.x / nB;
  const int f = g.frame0 + blockIdx.x % nB;
  const int l = level_of(g, sx, &LevelGeom::strip_base);
  const LevelGeom& lv = g.lv[l];
  const int strip = sx - lv.strip_base;
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
    // Levels with depth-validity bits and whole 32-pixel words per row (the edge bitmap is k_hyst's, k_fill adds its
    // pixels to it): edge AND validity as 32 x 32 bit tiles -- lane r of a half-wave loads the two words of row r (coalesced),
    // five exchange steps transpose the tile, lane c ends with the row mask of column c.  Two loads per (column, chunk)
    // instead of 64 byte loads.
    const bool bit_tiles = l < g.n_levels - 1 && (lv.w & 31) == 0 && lv.chunk_rows == 32;
    if (bit_tiles) {
      const uint2* csw = pl.cs[l] + (size_t)f * lv.h * lv.wpr;
      const uint32_t* vw = reinterpret_cast<const uint32_t*>(pl.vb[l] + (size_t)f * (lv.npix >> 3));
      const int lane = tid & 63, r = lane & 31;
      const int wc = strip * (CW_COLS / 32) + (lane >> 5);  // this half-wave's word column
      for (int c = cl; c < lv.nchunk; c += LANES) {         // (wave-uniform)
        const int y = c * 32 + r;
        uint32_t v = 0;
        if (y < lv.h && wc < lv.wpr) v = csw[(size_t)y * lv.wpr + wc].y & vw[(size_t)y * lv.wpr + wc];
        v = tile_transpose32(v, r);
        if (x < lv.w) {
          s_mask[xl * lv.nchunk + c] = v;   // lane = column, bit = row of the chunk
          s_cnt[xl * lv.nchunk + c] = __popc(v);
        }
      }
    } else
    if (x < lv.w) {
      for (int c = cl; c < lv.nchunk; c += LANES) {
        const int yb = c * lv.chunk_rows, ye = min(lv.h, yb + lv.chunk_rows);
        unsigned em = 0;
#pragma unroll 8
        for (int y = yb; y < ye; ++y) em |= (edges[(size_t)y * lv.w + x] ? 1u : 0u) << (y - yb);
        unsigned vm = 0;
        if (l < g.n_levels - 1) {  // depth validity comes as bits from the pyrDown that read this level
          const uint8_t* vb = pl.vb[l] + (size_t)f * (lv.npix >> 3);
          const int vpitch = lv.w >> 3;
          if (em) {
#pragma unroll 8
            for (int y = yb; y < ye; ++y) vm |= ((unsigned)(vb[(size_t)y * vpitch + (x >> 3)] >> (x & 7)) & 1u) << (y - yb);
            vm &= em;
          }
        } else
        // the coarsest level: four set bits per trip -- the depth loads of a trip are independent, so a column with k
        // edge pixels costs ceil(k/4) memory round trips instead of k
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
    int mine = 0;
    for (int i = tid; i < nslots; i += CW_COLS * LANES) {
      pl.cmask[l][slot0 + i] = s_mask[i];
      pl.chunk[l][slot0 + i] = s_cnt[i];
      mine += s_cnt[i];
    }
    // the strip's point count: all the write pass needs from the other strips (no scan kernel in between)
    __shared__ int s_tot;
    if (tid == 0) s_tot = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((tid & 63) == 0 && mine) atomicAdd(&s_tot, mine);
    __syncthreads();
    if (tid == 0) pl.strip_tot[(size_t)f * g.total_strips + sx] = s_tot;
  } else {
    for (int i = tid; i < nslots; i += CW_COLS * LANES) {
      s_mask[i] = pl.cmask[l][slot0 + i];
      s_cnt[i] = pl.chunk[l][slot0 + i];
    }
    __syncthreads();
    // offsets: the points of the level's earlier strips + an exclusive scan of this strip's slot counts
    // (slot order = list order: x-major, chunk-minor), in place in s_cnt
    int* s_wsum = reinterpret_cast<int*>(s_pts);  // 16 wave totals in the (still unused) staging area: the block's LDS stays at
                                                  // exactly 80 KB, two blocks per CU (64 more bytes halved the occupancy: 36 -> 55 us)
    // (width <= 2048: at most 32 strips per level -- one load per lane and a butterfly, in every wave)
    int base = (tid & 63) < strip ? pl.strip_tot[(size_t)f * g.total_strips + lv.strip_base + (tid & 63)] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) base += __shfl_xor(base, o);
    int total;
    {
      constexpr int NT = CW_COLS * CW_LANES;  // (the write pass; this branch is dead code in the count instantiation)
      static_assert(CW_COLS * CW_MAXCHUNK <= 4 * NT, "four slots per thread");
      int cs4[4], sum = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) { cs4[q] = 4 * tid + q < nslots ? s_cnt[4 * tid + q] : 0; sum += cs4[q]; }
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += v;
      }
      if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
      __syncthreads();
      // wave totals: lane k holds wave k's, two butterflies give this wave's prefix and the strip total
      const int wv = (tid & 63) < NT / 64 ? s_wsum[tid & 63] : 0;
      int pre = (tid & 63) < (tid >> 6) ? wv : 0;
      total = wv;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { pre += __shfl_xor(pre, o); total += __shfl_xor(total, o); }
      int run = base + pre + incl - sum;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (4 * tid + q < nslots) s_cnt[4 * tid + q] = run;
        run += cs4[q];
      }
      __syncthreads();
    }
    // (npts is NOT stored here: k_tile_count wrote the same count during the build, and this accessor-only walk may run
    // while a batch tracker on another stream reads it -- ADVICE r03)
    const bool staged = total <= CW_STAGE;
    float4* out = pl.pts[l] + (size_t)f * lv.npix;
    if (x < lv.w) {
      for (int c = cl; c < lv.nchunk; c += LANES) {
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
      for (int i = tid; i < total; i += CW_COLS * LANES) out[base + i] = s_pts[i];
    }
  }
}

// ---------------------------------------------------------------------------
// a8 for the tracker: the same points as the ordered list (imgpyramidrgbd.cpp:199-226), TILE-ORDERED.
// The reference's list is column-major (x outer, y inner): 64 consecutive entries -- one wavefront of the tracker --
// sit in ~1.7 image columns and span every row, so each of the tracker's gather instructions touched ~44 different
// cache lines (TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD) and a workgroup's 512 points walked the whole height of the
// keyframe's DT plane.  The tracker does not care about the order of its sum, only the accessor does: the hot path
// now writes the points grouped by 32 x 32-pixel tiles (tiles in raster order, row-major inside a tile) -- a
// wavefront's points share a handful of DT rows -- and the reference-ordered list is produced on demand by the walk
// above (revo_pyramid_read, EDGES3D).  Both lists hold the same points with the same bits.
//   k_tile_count : one workgroup per (level, frame); thread = row of a tile (32-pixel word of the edge bitmap AND the
//                  depth-validity bits; the coarsest level, which has no validity bits, tests its depths), tile totals,
//                  an exclusive scan over the level's tiles -> tile_base, the level's point count -> npts
//   k_pts_tiles  : half a wavefront per tile; the tile's 32 x 32 depths go through LDS (rows loaded 128 B at a time
//                  instead of one sparse gather per point), each lane emits the points of its row
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t tile_valid_word(const PyrGeom& g, const FramePlanes& pl, int l, int f, int y, int wc, uint32_t E) {
  // depth-validity bits of pixels 32*wc .. 32*wc+31 of row y (levels below the coarsest: written by k_pyrdown)
  const LevelGeom& lv = g.lv[l];
  const uint8_t* vb = pl.vb[l] + (size_t)f * (lv.npix >> 3) + (size_t)y * (lv.w >> 3);
  if ((lv.w & 31) == 0) return E & *reinterpret_cast<const uint32_t*>(vb + 4 * wc);
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (32 * wc + 8 * k < lv.w) v |= (uint32_t)vb[4 * wc + k] << (8 * k);
  return E & v;
}
__device__ __forceinline__ int level_tile_base(const PyrGeom& g, int l) {  // tiles of finer levels (32 x 32 px, raster order)
  int b = 0;
  for (int k = 0; k < l; ++k) b += g.lv[k].wpr * g.lv[k].nchunk;
  return b;
}

// Where the depth half of k_pyrdown stages the depths of a level's EDGE pixels (levels below the coarsest; batches whose depth
// pyramid runs behind Canny): tiles in raster order, rows of a tile top to bottom, pixels of a row left to right -- the order of
// the tile-ordered list, over ALL edge pixels (the depth test comes later: it needs the depths).  One workgroup per
// (level, frame), half a wavefront per tile, lane = row: the row's edge count, its prefix inside the tile (epre), the tile's
// total, an exclusive scan over the level's tiles (stage_base).  Bitmaps only: ~3 MB read, ~1.3 MB written per 64 frames.
__global__ void __launch_bounds__(1024) k_edge_prefix(PyrGeom g, FramePlanes pl) {
  __shared__ int s_tile[2048];
  __shared__ int s_wsum[16];
  const int nL = g.n_levels - 1;
  const int nB = gridDim.x / nL;
  const int l = blockIdx.x / nB;
  const int f = g.frame0 + blockIdx.x % nB;
  const LevelGeom& lv = g.lv[l];
  const int wpr = lv.wpr, h = lv.h, ntiles = wpr * lv.nchunk;
  const uint2* csw = pl.cs[l] + (size_t)f * h * wpr;
  unsigned short* epre = pl.epre[l] + (size_t)f * h * wpr;
  const int tid = threadIdx.x, r = tid & 31;
  for (int i = tid; i < ntiles; i += 1024) s_tile[i] = 0;
  __syncthreads();
  for (int t0 = tid >> 5; t0 < ntiles; t0 += 4 * 32) {  // four tiles per trip, like k_tile_count
    uint32_t E[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = t0 + 32 * q;
      const int c = t / wpr, wc = t - c * wpr, y = c * 32 + r;
      E[q] = (t < ntiles && y < h) ? csw[(size_t)y * wpr + wc].y : 0u;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = t0 + 32 * q;
      const int c = t / wpr, wc = t - c * wpr, y = c * 32 + r;
      const int cnt = __popc(E[q]);
      int incl = cnt;
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
        const int u = __shfl_up(incl, o, 32);
        if (r >= o) incl += u;
      }
      if (t < ntiles && y < h) epre[(size_t)y * wpr + wc] = (unsigned short)(incl - cnt);
      if (r == 31 && t < ntiles) s_tile[t] = incl;
    }
  }
  __syncthreads();
  const int i0 = 2 * tid, i1 = 2 * tid + 1;
  const int c0 = i0 < ntiles ? s_tile[i0] : 0, c1 = i1 < ntiles ? s_tile[i1] : 0;
  int incl = c0 + c1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o);
    if ((tid & 63) >= o) incl += u;
  }
  if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
  __syncthreads();
  int pre = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) pre += k < (tid >> 6) ? s_wsum[k] : 0;
  const int ex = pre + incl - (c0 + c1);
  int* sb = pl.stage_base + (size_t)f * g.total_tiles + level_tile_base(g, l);
  if (i0 < ntiles) sb[i0] = ex;
  if (i1 < ntiles) sb[i1] = ex + c0;
}

__global__ void __launch_bounds__(1024) k_tile_count(PyrGeom g, FramePlanes pl) {
  __shared__ int s_tile[2048];   // per tile: count, then exclusive base (levels up to 2048 x 1024: 64 x 32 tiles)
  __shared__ int s_wsum[16];
  const int nB = gridDim.x / g.n_levels;
  const int l = blockIdx.x / nB;
  const int f = g.frame0 + blockIdx.x % nB;
  const LevelGeom& lv = g.lv[l];
  const int wpr = lv.wpr, h = lv.h, ntiles = wpr * lv.nchunk;
  const bool has_vb = l < g.n_levels - 1;
  const uint2* csw = pl.cs[l] + (size_t)f * h * wpr;
  const float* depth = pl.depth[l] + (size_t)f * lv.npix;
  const int tid = threadIdx.x, r = tid & 31;
  for (int i = tid; i < ntiles; i += 1024) s_tile[i] = 0;
  __syncthreads();
  // half-waves take tiles round-robin; lane r = row r of the tile
  // (four tiles per trip: their loads are independent, a half-wave waits for memory once per trip)
  for (int t0 = tid >> 5; t0 < ntiles; t0 += 4 * 32) {
    uint32_t v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = t0 + 32 * q;
      const int c = t / wpr, wc = t - c * wpr;
      const int y = c * 32 + r;
      v[q] = 0;
      if (t < ntiles && y < h) {
        const uint32_t E = csw[(size_t)y * wpr + wc].y;
        if (has_vb) v[q] = tile_valid_word(g, pl, l, f, y, wc, E);
        else
          for (uint32_t m = E; m; m &= m - 1) {  // the coarsest level: a few thousand pixels per frame
            const int b = __ffs(m) - 1;
            if (depth_ok(depth[(size_t)y * lv.w + 32 * wc + b], g.depth_min, g.depth_max)) v[q] |= 1u << b;
          }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int cnt = __popc(v[q]);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 32);
      if (r == 0 && t0 + 32 * q < ntiles) s_tile[t0 + 32 * q] = cnt;
    }
  }
  __syncthreads();
  // exclusive scan over the level's tiles (<= 2 per thread)
  const int i0 = 2 * tid, i1 = 2 * tid + 1;
  const int c0 = i0 < ntiles ? s_tile[i0] : 0, c1 = i1 < ntiles ? s_tile[i1] : 0;
  int incl = c0 + c1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o);
    if ((tid & 63) >= o) incl += u;
  }
  if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
  __syncthreads();
  int pre = 0, total = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { const int wv = s_wsum[k]; pre += k < (tid >> 6) ? wv : 0; total += wv; }
  const int ex = pre + incl - (c0 + c1);
  int* tb = pl.tile_base + (size_t)f * g.total_tiles + level_tile_base(g, l);
  if (i0 < ntiles) tb[i0] = ex;
  if (i1 < ntiles) tb[i1] = ex + c0;
  if (tid == 0) pl.npts[f * REVO_L + l] = total;
}

#define PT_TILES 8                 // tiles (half-waves) per block of k_pts_tiles
// Half a wavefront per 32 x 32 tile.  Lane r reads row r's word of (edge bitmap AND validity bits), the rows' counts are
// scanned across the half-wave, every lane records WHERE its row's points are (row << 5 | column, a few instructions per
// point: a row of a horizontal edge holds up to 32 points, a row crossed by a vertical edge one), and then the
// back-projection is dealt out evenly: lane k takes list positions k, k + 32, ... of the tile, gathers that pixel's depth
// (consecutive positions are neighbours in a row: a gather touches a handful of lines), divides twice and stores --
// consecutive lanes write consecutive 16-byte entries.  No depth staging, ~32 VGPRs, 2 KB of LDS per tile: the kernel
// keeps its occupancy next to the trackers and the EDT it runs beside (the round-3 version that staged 4 KB depth tiles
// through LDS took 33 us alone but 120 us in the pipelined step).
// STAGED (a template parameter, not a run-time flag: its second 16 KB table would halve the plain kernel's workgroups per CU --
// measured: 40 -> 90 us in the pipelined step): the depths come from FramePlanes::stage (k_edge_prefix / k_pyrdown<.., STAGE>)
template <bool STAGED>
__global__ void __launch_bounds__(32 * PT_TILES) k_pts_tiles(PyrGeom g, FramePlanes pl) {
  __shared__ unsigned short s_src[PT_TILES][1024];  // list position inside the tile -> (row << 5 | column)
  __shared__ unsigned short s_esrc[STAGED ? PT_TILES : 1][STAGED ? 1024 : 1]; // ... -> position among the tile's EDGE pixels (staged depths)
  // 1-D grid, frame fastest (the tile groups of a frame share one XCD's L2)
  const int groups = (g.total_tiles + PT_TILES - 1) / PT_TILES;
  const int nB = gridDim.x / groups;
  const int f = g.frame0 + blockIdx.x % nB;
  const int tg = (blockIdx.x / nB) * PT_TILES + (threadIdx.x >> 5);
  if (tg >= g.total_tiles) return;
  int l = 0, t = tg;
  for (int k = 0; k < g.n_levels; ++k) {
    const int n = g.lv[k].wpr * g.lv[k].nchunk;
    if (t < n) { l = k; break; }
    t -= n;
  }
  const LevelGeom& lv = g.lv[l];
  const int wpr = lv.wpr, h = lv.h, w = lv.w;
  const int c = t / wpr, wc = t - c * wpr;
  const int r = threadIdx.x & 31;
  const int y0 = c * 32, x0 = wc * 32, y = y0 + r;
  const bool has_vb = l < g.n_levels - 1;
  const uint2* csw = pl.cs[l] + (size_t)f * h * wpr;
  const float* depth = pl.depth[l] + (size_t)f * lv.npix;
  const int tile_first = pl.tile_base[(size_t)f * g.total_tiles + tg];  // (requested early: needed last)
  const uint32_t E = y < h ? csw[(size_t)y * wpr + wc].y : 0u;
  uint32_t v = (has_vb && y < h) ? tile_valid_word(g, pl, l, f, y, wc, E) : E;
  if (!has_vb)   // the coarsest level has no validity bits: the depth test of imgpyramidrgbd.cpp:208, like k_tile_count
    for (uint32_t m = v; m; m &= m - 1) {
      const int b = __ffs(m) - 1;
      if (!depth_ok(depth[(size_t)y * w + x0 + b], g.depth_min, g.depth_max)) v &= ~(1u << b);
    }
  const int cnt = __popc(v);
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int u = __shfl_up(incl, o, 32);
    if (r >= o) incl += u;
  }
  const int total = __shfl(incl, 31, 32);
  if (total == 0) return;  // (half-wave uniform)
  unsigned short* src = s_src[threadIdx.x >> 5];
  // staged depths (g.pts_staged, levels with validity bits): the depth of the tile's k-th EDGE pixel (row-major, valid or not)
  // sits at stage[stage_base(tile) + k] -- a few consecutive lines per tile instead of every line of the plane that holds a point
  const bool staged = STAGED && has_vb;
  unsigned short* esrc = s_esrc[STAGED ? (threadIdx.x >> 5) : 0];
  {
    int o = incl - cnt;
    int ebase = 0;
    if (staged) {  // edge pixels of the rows above in this tile (the scan k_edge_prefix did: recomputed, it is five exchanges)
      const int ec = __popc(E);
      int ei = ec;
#pragma unroll
      for (int q = 1; q < 32; q <<= 1) {
        const int u = __shfl_up(ei, q, 32);
        if (r >= q) ei += u;
      }
      ebase = ei - ec;
    }
    for (uint32_t m = v; m; m &= m - 1, ++o) {
      const int b = __ffs(m) - 1;
      src[o] = (unsigned short)((r << 5) | b);
      if (staged) esrc[o] = (unsigned short)(ebase + __popc(E & ((1u << b) - 1u)));
    }
  }
  // (the half-wave reads what it wrote itself: program order within the wave is enough)
  float4* out = pl.pts_trk[l] + (size_t)f * lv.npix + tile_first;
  const float* stage_t = staged ? pl.stage[l] + (size_t)f * lv.npix + pl.stage_base[(size_t)f * g.total_tiles + tg] : nullptr;
  for (int k0 = r; k0 < total; k0 += 4 * 32) {  // four positions per lane and trip: their gathers are in flight together
    int rb[4];
    float Z[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rb[q] = k0 + 32 * q < total ? (int)src[k0 + 32 * q] : -1;
    if (staged) {
#pragma unroll
      for (int q = 0; q < 4; ++q) Z[q] = rb[q] >= 0 ? stage_t[esrc[k0 + 32 * q]] : 0.0f;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) Z[q] = rb[q] >= 0 ? depth[(size_t)(y0 + (rb[q] >> 5)) * w + x0 + (rb[q] & 31)] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (rb[q] < 0) continue;
      const float X = __fdiv_rn(Z[q] * ((float)(x0 + (rb[q] & 31)) - lv.cx), lv.fx);
      const float Y = __fdiv_rn(Z[q] * ((float)(y0 + (rb[q] >> 5)) - lv.cy), lv.fy);
      out[k0 + 32 * q] = make_float4(X, Y, Z[q], 1.0f);
    }
  }
}

// exclusive scan of a[0..n) by one 1024-thread block; returns the total (valid in every thread).  Per-thread
// segments, a shuffle scan inside each wave and the 16 wave totals through LDS: two barriers (the Hillis-Steele
// scan over 1024 LDS entries it replaces took twenty).  s_part: >= 16 ints.
__device__ __forceinline__ int block_exclusive_scan(int* a, int n, int* s_part) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n + 1023) / 1024;
  const int b = min(n, tid * per), e = min(n, b + per);
  int sum = 0;
  for (int i = b; i < e; ++i) sum += a[i];
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  int run = incl - sum, total = 0;  // exclusive prefix of this thread's segment
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int wv = s_part[k];
    run += k < wave ? wv : 0;
    total += wv;
  }
  for (int i = b; i < e; ++i) { const int c = a[i]; a[i] = run; run += c; }
  return total;
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
#define EDT_COL_GROUPS(h) ((h) > 1024 ? 64 : ((h) > 512 ? 32 : 16))  // row groups of <= 32 rows per column (heights <= 2048)
__global__ void __launch_bounds__(EDT_THREADS) k_edt_cols(PyrGeom g, FramePlanes pl, int f0, int fstride, int nframes) {
  __shared__ int s_first[EDT_THREADS];  // [group][column]: first edge row of the group's segment (or +INF)
  __shared__ int s_last[EDT_THREADS];   // last edge row of the segment (or -INF)
  // 1-D grid, frame fastest (the strips of a frame share one XCD's L2, see k_compact_walk)
  const int f = f0 + (blockIdx.x % nframes) * fstride;
  // decode (level, strip); taller levels use 32 groups x 32 columns (64 x 16 above 1024 rows), the others 16 x 64
  int l = 0, sidx = blockIdx.x / nframes, ngroups = 16, ncols = 64;
  for (int k = 0; k < g.n_levels; ++k) {
    ngroups = EDT_COL_GROUPS(g.lv[k].h);
    ncols = EDT_THREADS / ngroups;
    const int ns = (g.lv[k].w + ncols - 1) / ncols;
    if (sidx < ns) { l = k; break; }
    sidx -= ns;
  }
  const LevelGeom& lv = g.lv[l];
  const int col = threadIdx.x % ncols, grp = threadIdx.x / ncols;
  const int x = sidx * ncols + col;
  const int rpg = (lv.h + ngroups - 1) / ngroups;  // <= 32 (height <= 2048)
  const int yb = min(lv.h, grp * rpg), ye = min(lv.h, yb + rpg);
  const bool in = x < lv.w;
  const uint8_t* edges = pl.edges[l] + (size_t)f * lv.npix;
  unsigned em = 0;
  if ((lv.w & 31) == 0 && (ncols & 31) == 0) {
    // the level's edge bitmap (k_hyst / k_fill left it in cs[].y) as 32 x 32 bit tiles: lane r of a half-wave loads
    // the word of row yb + r, the transpose hands every column its rows -- one load instead of 32 byte loads
    const uint2* csw = pl.cs[l] + (size_t)f * lv.h * lv.wpr;
    const int r = threadIdx.x & 31;
    const int wc = (sidx * ncols + (col & ~31)) >> 5;
    uint32_t v = 0;
    if (yb + r < ye && wc < lv.wpr) v = csw[(size_t)(yb + r) * lv.wpr + wc].y;
    em = tile_transpose32(v, r);
    if (!in) em = 0;
  } else if (in) {
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
  uint16_t* gd = reinterpret_cast<uint16_t*>(pl.scratch[l]) + (size_t)f * lv.npix;  // vertical distance, 0xffff = no edge in the column
  // walking down the segment: the distance to the nearest edge above grows by one per row and drops to 0 on an edge; the one
  // below is ffs on what is left of the mask, or the carry from the groups below.  "No edge" is a distance beyond any height
  // (the rows of a level are <= 2048), so the sums below cannot wrap and everything >= 0xffff is written as 0xffff.
  const int FAR = 1 << 20;
  int du = (above <= -EDT_INF) ? FAR : (yb - above);      // of row yb - 1, plus one
  const int dbelow = (below >= EDT_INF) ? FAR : (below - yb);
  for (int y = yb; y < ye; ++y) {
    const int k = y - yb;
    const unsigned hi = em >> k;                          // bit 0 = row y: edges at or below y
    du = (hi & 1u) ? 0 : du;
    const int dn = hi ? __ffs(hi) - 1 : dbelow - k;
    const int m = min(du, dn);
    gd[(size_t)y * lv.w + x] = (uint16_t)(m >= 0xffff ? 0xffff : m);
    du += 1;
  }
}

// Rows: a workgroup takes whole rows worth ~EDT_ROW_PX pixels (2 rows of 640, 16 of 80: every lane has a pixel), squares the
// vertical distances into LDS rows padded with "no edge" on both sides as far as the search can reach, and every pixel searches
//      d2(x) = min_d  d^2 + min(g2[x - d], g2[x + d]),     four distances per trip, until d^2 >= best      (no bounds checks)
// sqrtf of an integer-valued float in [0, 2^24): v_sqrt_f32 (within 1 ulp) and the +-1 ulp correction of the compiler's own
// sqrtf expansion, without its scaling of tiny inputs and its inf / nan / zero classification (x = 0 falls through both
// corrections: the residual of the lower neighbour is NaN, that of the upper one is not positive).  Same instructions on the
// same values as sqrtf(x) for x >= 1: correctly rounded.
__device__ __forceinline__ float sqrt_rn_int(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const unsigned si = __float_as_uint(s);
  const float sd = __uint_as_float(si - 1u), su = __uint_as_float(si + 1u);
  const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
  float r = (rd <= 0.0f) ? sd : s;
  r = (ru > 0.0f) ? su : r;
  return r;
}
__global__ void __launch_bounds__(256) k_edt_rows(PyrGeom g, FramePlanes pl, int f0, int fstride) {
  extern __shared__ int s_g2[];
  const int f = f0 + blockIdx.z * fstride;
  const int l = level_of(g, blockIdx.x, &LevelGeom::edt_block_base);
  const LevelGeom lv = g.lv[l];
  const int w = lv.w;
  const int y0 = (blockIdx.x - lv.edt_block_base) * lv.edt_rows;
  const int nr = min(lv.edt_rows, lv.h - y0);
  const int pad = w + 4, pitch = w + 2 * pad;  // the search stops at d < w (+3 for the trip): never leaves the padding
  const uint16_t* gd = reinterpret_cast<const uint16_t*>(pl.scratch[l]) + (size_t)f * lv.npix + (size_t)y0 * w;
  // LDS rows: [pad "no edge"][w squared vertical distances][pad "no edge"].  Row by row, no index division (the flattened
  // fill with its divisions was 43 % of this kernel's VALU instructions): the pads are constant 16-byte stores (widths are
  // multiples of 4, so pad, pitch and every row start are too), the data four pixels per lane from one 8-byte load.
  const int4 inf4 = make_int4(EDT_INF, EDT_INF, EDT_INF, EDT_INF);
  const int padq = pad >> 2, wq = w >> 2;
  for (int r = 0; r < nr; ++r) {
    int4* row4 = reinterpret_cast<int4*>(s_g2 + r * pitch);
    for (int i = threadIdx.x; i < padq; i += 256) { row4[i] = inf4; row4[padq + wq + i] = inf4; }
    const uint2* src = reinterpret_cast<const uint2*>(gd + r * w);
    for (int i = threadIdx.x; i < wq; i += 256) {
      const uint2 q = src[i];
      const int d0 = (int)(q.x & 0xffffu), d1 = (int)(q.x >> 16), d2 = (int)(q.y & 0xffffu), d3 = (int)(q.y >> 16);
      row4[padq + i] = make_int4(d0 == 0xffff ? EDT_INF : __mul24(d0, d0), d1 == 0xffff ? EDT_INF : __mul24(d1, d1),
                                 d2 == 0xffff ? EDT_INF : __mul24(d2, d2), d3 == 0xffff ? EDT_INF : __mul24(d3, d3));
    }
  }
  __syncthreads();
  float* dt = pl.dt[l] + (size_t)f * lv.npix + (size_t)y0 * w;
  // pixel p = r * w + x, p += 256 per trip: (r, x) advance by (256 / w, 256 % w) with one carry -- one division per thread
  const int dq = 256 / w, dr = 256 - dq * w;
  int r = threadIdx.x / w, x = threadIdx.x - r * w;
  for (int p = threadIdx.x; p < nr * w; p += 256) {
    const int* c = s_g2 + r * pitch + pad + x;
    int best = c[0];
    for (int d = 1; d < w && d * d < best; d += 4) {
      int m = best;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dj = d + j;
        m = min(m, dj * dj + min(c[-dj], c[dj]));
      }
      best = m;
    }
    dt[p] = best >= EDT_INF ? sqrtf(1e15f) : sqrt_rn_int((float)best);
    x += dr; r += dq;
    if (x >= w) { x -= w; ++r; }
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
                                                   float dmin, float dmax, int* hist8, unsigned* done, int* host_out,
                                                   unsigned seq_val) {
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
    __syncthreads();
    if (threadIdx.x == 0) *(volatile int*)&host_out[8] = (int)seq_val;  // the host polls this word
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
  dim3 grid((npix / 16 + 255) / 256, 1, B);
  hipLaunchKernelGGL(k_gray_depth, grid, dim3(256), 0, s, d_bgr, d_depth_f32, d_depth_u16, u16_alpha, p.gray[0],
                     p.depth[0], npix, g.frame0);
}

void launch_pyrdown(const PyrGeom& g, const FramePlanes& p, int lvl, int B, hipStream_t s, int parts, bool stage_edges) {
  const LevelGeom& d = g.lv[lvl];
  const LevelGeom& sl = g.lv[lvl - 1];
  dim3 grid(((d.w / 4) * ((d.h + 1) / 2) + 255) / 256, 1, B);
  EdgeStage es{};
  if (stage_edges) {
    int tb = 0;
    for (int k = 0; k < lvl - 1; ++k) tb += g.lv[k].wpr * g.lv[k].nchunk;
    es.cs = p.cs[lvl - 1]; es.epre = p.epre[lvl - 1]; es.base = p.stage_base + tb; es.out = p.stage[lvl - 1];
    es.wpr = sl.wpr; es.tiles_total = g.total_tiles;
  }
#define PYRDOWN_ARGS p.gray[lvl - 1], sl.w, sl.h, p.gray[lvl], d.w, d.h, p.depth[lvl - 1], p.depth[lvl], g.frame0, p.vb[lvl - 1], g.depth_min, g.depth_max, es
  if (parts == 1) hipLaunchKernelGGL((k_pyrdown<true, false>), grid, dim3(256), 0, s, PYRDOWN_ARGS);
  else if (parts == 2 && stage_edges) hipLaunchKernelGGL((k_pyrdown<false, true, true>), grid, dim3(256), 0, s, PYRDOWN_ARGS);
  else if (parts == 2) hipLaunchKernelGGL((k_pyrdown<false, true>), grid, dim3(256), 0, s, PYRDOWN_ARGS);
  else hipLaunchKernelGGL((k_pyrdown<true, true>), grid, dim3(256), 0, s, PYRDOWN_ARGS);
#undef PYRDOWN_ARGS
}

void launch_edge_prefix(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s) {
  if (g.n_levels < 2) return;
  hipLaunchKernelGGL(k_edge_prefix, dim3((g.n_levels - 1) * B), dim3(1024), 0, s, g, p);
}

void launch_canny_nms(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s) {
  hipLaunchKernelGGL(k_canny_nms4, dim3(g.total_nms_blocks, 1, B), dim3(256), 0, s, g, p);
}

// hysteresis + edge planes + tile histograms: one workgroup per (level, frame)
void launch_hyst(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s) {
  size_t e_bytes = 0, ec_bytes = 0;
  for (int l = 0; l < g.n_levels; ++l) {
    const size_t nw = (size_t)g.lv[l].h * g.lv[l].wpr;
    const size_t e = ((size_t)(g.lv[l].h + 2) * g.lv[l].wpr + 2) * 4, c = (nw + (nw + 2) / 2) * 4;  // + candidate words + id bases
    e_bytes = e > e_bytes ? e : e_bytes;
    ec_bytes = e + c > ec_bytes ? e + c : ec_bytes;
  }
  static bool attr_set = false;
  if (!attr_set) {  // more than the default 64 KB of dynamic LDS
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hyst<true>), hipFuncAttributeMaxDynamicSharedMemorySize, REVO_HYST_LDS_MAX);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hyst<false>), hipFuncAttributeMaxDynamicSharedMemorySize, REVO_HYST_LDS_MAX);
    attr_set = true;
  }
  // Banded (several workgroups per level and frame) when a level does not fit the single-workgroup union-find -- 1280x960
  // then builds in 1.25 ms instead of 1.75 ms with the flood fill; at 640x480 the four kernels of the banded path take as
  // long as the one workgroup per (level, frame) (78 vs 79 us per 64 frames: measured, profiles/r03_hyst_banding.txt), so
  // that size keeps the single kernel.  REVO_HYST_BANDED=1 / 0 forces either.
  const int force = g.hyst_force;  // per context (revo_host.hip: build_geom reads REVO_HYST_BANDED when the context is created)
  const bool fits_single = ec_bytes + 4096 <= REVO_HYST_LDS_MAX;
  const bool banded = force >= 0 ? force == 1 : !fits_single;
  const int only_flagged = banded && g.total_bands > 0 ? 1 : 0;
  if (only_flagged) {
    static bool attr2 = false;
    if (!attr2) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hyst_band), hipFuncAttributeMaxDynamicSharedMemorySize, HB_LDS_WORDS * 4);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hyst_out), hipFuncAttributeMaxDynamicSharedMemorySize, HB_LDS_WORDS * 4);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hyst_seam), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      (void)hipGetLastError();  // (a refused attribute must not surface as the error of the next launch check)
      attr2 = true;
    }
    size_t seam_lds = 4, out_lds = 4;
    for (int l = 0; l < g.n_levels; ++l) {
      const LevelGeom& lv = g.lv[l];
      const size_t nwb = (size_t)lv.band_rows * lv.wpr;
      out_lds = std::max(out_lds, nwb * 4);
      if (lv.nbands > 1) seam_lds = std::max(seam_lds, (size_t)lv.nbands * (size_t)(HB_LDS_WORDS / 2 / 32 + 64) * 4);
    }
    hipLaunchKernelGGL(k_hyst_band, dim3(g.total_bands * B), dim3(HB_THREADS), HB_LDS_WORDS * 4, s, g, p);
    if (g.any_banded) hipLaunchKernelGGL(k_hyst_seam, dim3(g.n_levels * B), dim3(HS_THREADS), seam_lds, s, g, p, 0);
    hipLaunchKernelGGL(k_hyst_out, dim3(g.total_bands * B), dim3(HO_THREADS), out_lds, s, g, p, 0);
  }
  // MIXED: every level fits one workgroup, level 0 has several bands, the knob is on: heavy level-0 frames through the bands,
  // everything else through one workgroup, in one launch; then the seam and output passes of the heavy frames and the sweep for
  // frames a band handed over (its runs exceeded its label space)
  const bool mixed = !banded && fits_single && g.hyst_heavy_runs > 0 && g.total_bands > 0 && g.lv[0].nbands > 1;
  if (mixed) {
    static bool attr3 = false;
    if (!attr3) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hyst_mixed), hipFuncAttributeMaxDynamicSharedMemorySize, REVO_HYST_LDS_MAX);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hyst_out), hipFuncAttributeMaxDynamicSharedMemorySize, HB_LDS_WORDS * 4);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hyst_seam), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      (void)hipGetLastError();
      attr3 = true;
    }
    const LevelGeom& l0 = g.lv[0];
    const size_t seam_lds = std::max<size_t>(4, (size_t)l0.nbands * (size_t)(HB_LDS_WORDS / 2 / 32 + 64) * 4);
    const size_t out_lds = std::max<size_t>(4, (size_t)l0.band_rows * l0.wpr * 4);
    hipLaunchKernelGGL(k_hyst_mixed, dim3((l0.nbands + g.n_levels) * B), dim3(HYST_THREADS), REVO_HYST_LDS_MAX, s, g, p, B);
    hipLaunchKernelGGL(k_hyst_seam, dim3(B), dim3(HS_THREADS), seam_lds, s, g, p, 1);
    hipLaunchKernelGGL(k_hyst_out, dim3(l0.nbands * B), dim3(HO_THREADS), out_lds, s, g, p, 1);
    hipLaunchKernelGGL(k_hyst<true>, dim3(std::min(32, g.n_levels * B)), dim3(HYST_THREADS), REVO_HYST_LDS_MAX, s, g, p, 1, B);
    return;
  }
  const int n_wg = only_flagged ? std::min(32, g.n_levels * B) : g.n_levels * B;
  if (ec_bytes + 4096 <= REVO_HYST_LDS_MAX)  // candidate bitmap + union-find labels in LDS: all of it (one workgroup per CU anyway)
    hipLaunchKernelGGL(k_hyst<true>, dim3(n_wg), dim3(HYST_THREADS), REVO_HYST_LDS_MAX, s, g, p, only_flagged, B);
  else if (e_bytes <= REVO_HYST_LDS_MAX)  // big levels: only the edge bitmap lives in LDS, the (constant) candidate words are re-read through L1/L2
    hipLaunchKernelGGL(k_hyst<false>, dim3(n_wg), dim3(HYST_THREADS), e_bytes, s, g, p, only_flagged, B);
  else  // very big levels (beyond ~1280 x 990): the edge bitmap in the scratch plane -- the last resort behind the bands
    hipLaunchKernelGGL((k_hyst<false, true>), dim3(n_wg), dim3(HYST_THREADS), 16, s, g, p, only_flagged, B);
}

void launch_fill(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s) {
  bool any = false;
  for (int l = 1; l < g.n_levels; ++l) any = any || g.lv[l].has_orig;
  if (!any) return;
  if (g.lv[0].npix <= 640 * 480) {  // one block per frame, all levels: 5 us, one launch
    hipLaunchKernelGGL(k_fill, dim3(1, 1, B), dim3(1024), 0, s, g, p, 0);
    return;
  }
  for (int l = 1; l < g.n_levels; ++l) {  // large images: a launch per level, its pixels shared by up to 64 workgroups per frame
    if (!g.lv[l].has_orig) continue;
    const int n = (g.lv[l - 1].w / 2) * (g.lv[l - 1].h / 2);
    hipLaunchKernelGGL(k_fill, dim3(std::min(64, (n + 1023) / 1024), 1, B), dim3(1024), 0, s, g, p, l);
  }
}

// the hot path's edge list: tile-ordered, for the tracker (the reference-ordered list is launch_compact, on demand)
void launch_tile_points(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s, int which) {
  if (which & 1) hipLaunchKernelGGL(k_tile_count, dim3(g.n_levels * B), dim3(1024), 0, s, g, p);
  const int groups = (g.total_tiles + PT_TILES - 1) / PT_TILES;
  if ((which & 2) && g.pts_staged) hipLaunchKernelGGL(k_pts_tiles<true>, dim3(groups * B), dim3(32 * PT_TILES), 0, s, g, p);
  else if (which & 2) hipLaunchKernelGGL(k_pts_tiles<false>, dim3(groups * B), dim3(32 * PT_TILES), 0, s, g, p);
}

void launch_compact(const PyrGeom& g, const FramePlanes& p, int B, hipStream_t s) {
  dim3 grid(g.total_strips * B);
  hipLaunchKernelGGL(k_compact_walk<false>, grid, dim3(CW_COLS * CW_LANES_COUNT), 0, s, g, p);
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

void launch_keyframe(const PyrGeom& g, const FramePlanes& p, int f0, int fstride, int count, hipStream_t s, int which) {
  int strips = 0;
  for (int l = 0; l < g.n_levels; ++l) {
    const int ncols = EDT_THREADS / EDT_COL_GROUPS(g.lv[l].h);
    strips += (g.lv[l].w + ncols - 1) / ncols;
  }
  if (which & 1) hipLaunchKernelGGL(k_edt_cols, dim3(strips * count), dim3(EDT_THREADS), 0, s, g, p, f0, fstride, count);
  size_t rows_lds = 0;
  for (int l = 0; l < g.n_levels; ++l) rows_lds = std::max(rows_lds, (size_t)g.lv[l].edt_rows * (3 * g.lv[l].w + 8) * sizeof(int));
  if (which & 2) hipLaunchKernelGGL(k_edt_rows, dim3(g.total_edt_blocks, 1, count), dim3(256), rows_lds, s, g, p, f0, fstride);
}

// The float4 table is only materialised for the returnOptimizationStructure accessor: the
// tracker samples the DT plane and forms the same gradients on the fly.
void launch_grad_table(const PyrGeom& g, const FramePlanes& p, int f0, int fstride, int count, hipStream_t s) {
  hipLaunchKernelGGL(k_grad_table, dim3((g.lv[0].npix + 255) / 256, g.n_levels, count), dim3(256), 0, s, g, p, f0, fstride);
}

void launch_vote(const PyrGeom& g, const FramePlanes& curr, int curr_frame, int lvl, int n_clouds, const VoteArgs& va,
                 int* d_marks, int* d_hist8, unsigned* d_done, int* h_out8, unsigned seq_val, int use_orig_edges, hipStream_t s) {
  // d_marks, d_hist8 and d_done are all-zero on entry (zeroed at allocation, left clean by k_vote_hist)
  const LevelGeom& lv = g.lv[lvl];
  if (n_clouds > 0)
    hipLaunchKernelGGL(k_vote_mark, dim3(32, n_clouds), dim3(256), 0, s, va, lv.fx, lv.fy, lv.cx, lv.cy, lv.w, lv.h, d_marks);
  const uint8_t* edges = (use_orig_edges ? curr.edges_orig[lvl] : curr.edges[lvl]) + (size_t)curr_frame * lv.npix;
  hipLaunchKernelGGL(k_vote_hist, dim3(16), dim3(256), 0, s, d_marks, edges,
                     curr.depth[lvl] + (size_t)curr_frame * lv.npix, lv.npix, g.depth_min, g.depth_max, d_hist8, d_done, h_out8,
                     seq_val);
}

void launch_copy_cloud(float4* dst, const float4* src, int* dst_n, const int* src_n, hipStream_t s) {
  hipLaunchKernelGGL(k_copy_cloud, dim3(16), dim3(256), 0, s, dst, src, dst_n, src_n);
}
