// revo_track.hip -- the edge-alignment tracker as ONE persistent kernel (gfx950).
//
// Replaces TrackerNew::trackFrames (tracker.cpp:294-353), checkInitializationValues
// / evalCostFunction (tracker.cpp:265-283,357-393), Optimizer::trackFrames
// (optimizer.cpp:235-311), calcErrorAndBuffers + calculateWarpUpdate
// (optimizer.cpp:74-234) and LGS6 (LGSX.h:185-404).
//
// A CLUSTER of 512-thread workgroups per frame-pair (6 at batch 32 = 192 of the 256 CUs, 8 for a
// single pair) runs every pyramid level and every Levenberg-Marquardt iteration on the
// device: no host round trip between residual evaluations.  The members of a cluster
// split the point list, all-gather their 32 partial sums through 8-byte {epoch,value}
// granules (agent-scope relaxed atomics, the data is the flag; MI355X_MICROARCH.md
// "handoff" rows) and then take the SAME decision redundantly, so one exchange per
// evaluation suffices.  Blocks of one pair share blockIdx % 8 (same XCD, same L2).
// The reference's two hot loops (A: warp/project/bilinear
// gather/Huber, B: 6-vector Jacobian into the 6x6 system) are fused, so the 7
// scratch buffers of optimizer.h:146-152 never exist: each thread keeps the 21
// upper-triangle entries of J^T W J, the 6 of J^T W r, sum(w r^2), sum(r^2) and
// the good count in registers, a 64-lane "reduce-scatter" butterfly folds the
// 32 values of a wavefront with 32 shuffles (instead of 32 x 6), the 8
// per-wave partials meet in LDS and are summed in double in a fixed order
// (deterministic run to run), and wave 0 runs the damped 6x6 solve (row-parallel
// across lanes), SE3 exp and the accept/reject logic, publishing the next pose through LDS.
// The kernel is budgeted at 3 waves per SIMD worth of registers (146 VGPRs) so that the build
// kernels of the next batch can be resident next to it (see TRACK_MAXP below).
//
// The gradient/DT float4 table of the reference (imgpyramidrgbd.cpp:255-276) is NOT
// read here: the kernel samples the 4x smaller DT plane and forms the four
// corner gradients 0.5*(dt[i-1]-dt[i+1]), 0.5*(dt[i-w]-dt[i+w]) on the fly -- the
// same float operations, hence the same values -- which keeps 4 pairs per XCD inside
// the 4 MB L2 (first measurements: the table version was gather-latency bound at
// ~19 cycles/point/CU against ~1.2 cycles of ALU work).
//
// There is no dense contraction here (a 6-vector outer product per point), so
// no MFMA; the kernel is bound by the serial latency of one residual evaluation
// (6.6 us: loop 47 %, LM decision 30 %, cluster exchange 13 %) times the ~40 evaluations of a pair.
#include "revo_dev.h"

namespace {

enum { MODE_EVAL = 0, MODE_COST = 1, MODE_DONE = 2 };
enum { PH_COST_EYE = 0, PH_COST_INIT = 1, PH_LEVEL_FIRST = 2, PH_LM = 3, PH_EVAL_ONLY = 4 };
#define NWAVES (TRACK_THREADS / 64)
#define SPIN_LIMIT 400000      // bounded cluster wait (~0.5 s): never hang the GPU
#define MAX_TOTAL_EVALS 6000  // hang guard; the reference bound is 100 outer iterations x retries

struct Ctrl {  // published by wave 0, read by everyone after the barrier
  float R[9];
  float T[3];
  int level;
  int mode;
};

struct W0State {  // wave-0 private LM state, kept in LDS to keep VGPRs for the hot loop
  double Aacc[27];  // accepted normal equations: 21 upper-tri of A/n, then 6 of (sum w r v)/n
  float q[4], t[3];    // accepted pose (Sophus::SE3f referenceToFrame)
  float qn[4], tn[3];  // candidate pose
  float lastErr, last_residual, lambda, incsq, costEye;
  int iteration, incTry, phase, flags, total_evals;
  int good, bad;
  float sumw, sumu;
  int evals[REVO_L];
#ifdef REVO_TRACK_PROFILE
  long long prof[6];  // cycles: eval loop, barrier 1, LDS sum, cluster exchange, decision, barrier 2
#endif
};

// ---- small algebra (Eigen/Sophus semantics, float like the reference) -------
// Eigen's Quaternionf(Matrix3f): four algebraically equivalent branches chosen by the
// largest of (trace, m00, m11, m22).  Written branch-free with selects so the quaternion
// stays in registers (element writes under control flow sent it to scratch memory, i.e. a
// dozen vector-memory round trips on the serial path of every evaluation).
__device__ __forceinline__ void quat_from_R(const float* R, float* q) {  // R column-major; q = (w,x,y,z)
#define RM(r, c) R[(c)*3 + (r)]
  const float m00 = RM(0, 0), m11 = RM(1, 1), m22 = RM(2, 2);
  const float tr = m00 + m11 + m22;
  const bool b0 = tr > 0.0f;
  const bool i1 = m11 > m00;                     // Eigen: i = 0; if (m11 > m00) i = 1; if (m22 > m(i,i)) i = 2;
  const bool i2 = m22 > (i1 ? m11 : m00);
  const int sel = b0 ? 0 : (i2 ? 3 : (i1 ? 2 : 1));  // 0: trace, 1: i=0, 2: i=1, 3: i=2
  const float arg = sel == 0 ? tr + 1.0f
                  : sel == 1 ? (m00 - m11 - m22 + 1.0f)
                  : sel == 2 ? (m11 - m22 - m00 + 1.0f)
                             : (m22 - m00 - m11 + 1.0f);
  const float t = sqrtf(arg);
  const float h = 0.5f * t;
  const float s = __fdiv_rn(0.5f, t);
  const float d21 = (RM(2, 1) - RM(1, 2)) * s, d02 = (RM(0, 2) - RM(2, 0)) * s, d10 = (RM(1, 0) - RM(0, 1)) * s;
  const float s10 = (RM(1, 0) + RM(0, 1)) * s, s20 = (RM(2, 0) + RM(0, 2)) * s, s21 = (RM(2, 1) + RM(1, 2)) * s;
  const float qw = sel == 0 ? h : sel == 1 ? d21 : sel == 2 ? d02 : d10;
  const float qx = sel == 0 ? d21 : sel == 1 ? h : sel == 2 ? s10 : s20;
  const float qy = sel == 0 ? d02 : sel == 1 ? s10 : sel == 2 ? h : s21;
  const float qz = sel == 0 ? d10 : sel == 1 ? s20 : sel == 2 ? s21 : h;
  q[0] = qw; q[1] = qx; q[2] = qy; q[3] = qz;
#undef RM
}

__device__ __forceinline__ void quat_to_R(const float* q, float* R) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0f - (tyy + tzz); R[3] = txy - twz; R[6] = txz + twy;
  R[1] = txy + twz; R[4] = 1.0f - (txx + tzz); R[7] = tyz - twx;
  R[2] = txz - twy; R[5] = tyz + twx; R[8] = 1.0f - (txx + tyy);
}

__device__ __forceinline__ bool is_orthogonal(const float* R) {  // rotation_matrix.hpp:14-24, so3.hpp:419-424
  float n2 = 0.0f;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = R[r] * R[c] + R[3 + r] * R[3 + c] + R[6 + r] * R[6 + c];
      v -= (r == c) ? 1.0f : 0.0f;
      n2 += v * v;
    }
  const float det = R[0] * (R[4] * R[8] - R[7] * R[5]) - R[3] * (R[1] * R[8] - R[7] * R[2]) + R[6] * (R[1] * R[5] - R[4] * R[2]);
  return sqrtf(n2) < 1e-5f && det > 0.0f;
}

// sin / cos for the LM increments (|x| is ~1e-2 .. 1e-1): below 0.5 rad the truncated series are
// accurate to float rounding (next terms x^11/11! < 2e-11, x^10/10! < 3e-10 relative) and avoid the
// full-range argument reduction of sinf/cosf on the serial path; larger angles take the library path.
__device__ __forceinline__ float sin_lm(float x) {
  if (fabsf(x) > 0.5f) return sinf(x);
  const float x2 = x * x;
  return x * (1.0f + x2 * (-1.0f / 6.0f + x2 * (1.0f / 120.0f + x2 * (-1.0f / 5040.0f + x2 * (1.0f / 362880.0f)))));
}
__device__ __forceinline__ float cos_lm(float x) {
  if (fabsf(x) > 0.5f) return cosf(x);
  const float x2 = x * x;
  return 1.0f + x2 * (-0.5f + x2 * (1.0f / 24.0f + x2 * (-1.0f / 720.0f + x2 * (1.0f / 40320.0f))));
}

// Sophus::SE3f::exp(inc) * (q,t)  (se3.hpp:723-745, so3.hpp:531-565, se3.hpp:317-321, so3.hpp:335-352)
__device__ __forceinline__ void se3_exp_mul(const float* a, const float* q, const float* t, float* qo, float* to) {
  const float ox = a[3], oy = a[4], oz = a[5];
  const float theta_sq = ox * ox + oy * oy + oz * oz;
  const float theta = sqrtf(theta_sq);
  float imag, real;
  float V[9];  // row-major 3x3
  const float O[9] = {0.f, -oz, oy, oz, 0.f, -ox, -oy, ox, 0.f};
  float O2[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) O2[r * 3 + c] = O[r * 3] * O[c] + O[r * 3 + 1] * O[3 + c] + O[r * 3 + 2] * O[6 + c];
  if (theta < 1e-5f) {
    const float p4 = theta_sq * theta_sq;
    imag = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * p4;
    real = 1.0f - (float)(1.0 / 8.0) * theta_sq + (float)(1.0 / 384.0) * p4;
  } else {
    const float h = 0.5f * theta;
    imag = __fdiv_rn(sin_lm(h), theta);
    real = cos_lm(h);
  }
  const float qe[4] = {real, imag * ox, imag * oy, imag * oz};
  if (theta < 1e-5f) {
    float Rm[9];
    quat_to_R(qe, Rm);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) V[r * 3 + c] = Rm[c * 3 + r];
  } else {
    const float ca = __fdiv_rn(1.0f - cos_lm(theta), theta_sq);
    const float cb = __fdiv_rn(theta - sin_lm(theta), theta_sq * theta);
#pragma unroll
    for (int i = 0; i < 9; ++i) V[i] = (((i % 4) == 0 ? 1.0f : 0.0f) + ca * O[i]) + cb * O2[i];
  }
  float te[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) te[r] = V[r * 3] * a[0] + V[r * 3 + 1] * a[1] + V[r * 3 + 2] * a[2];
  // translation: te + qe (x) t   (Eigen _transformVector)
  float uv[3] = {qe[2] * t[2] - qe[3] * t[1], qe[3] * t[0] - qe[1] * t[2], qe[1] * t[1] - qe[2] * t[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  const float c3[3] = {qe[2] * uv[2] - qe[3] * uv[1], qe[3] * uv[0] - qe[1] * uv[2], qe[1] * uv[1] - qe[2] * uv[0]};
#pragma unroll
  for (int r = 0; r < 3; ++r) to[r] = te[r] + (t[r] + qe[0] * uv[r] + c3[r]);
  // rotation: qe * q, renormalised by 2/(1+|q|^2)
  float w = qe[0] * q[0] - qe[1] * q[1] - qe[2] * q[2] - qe[3] * q[3];
  float x = qe[0] * q[1] + qe[1] * q[0] + qe[2] * q[3] - qe[3] * q[2];
  float y = qe[0] * q[2] + qe[2] * q[0] + qe[3] * q[1] - qe[1] * q[3];
  float z = qe[0] * q[3] + qe[3] * q[0] + qe[1] * q[2] - qe[2] * q[1];
  const float sn = w * w + x * x + y * y + z * z;
  if (sn != 1.0f) {
    const float s = __fdiv_rn(2.0f, 1.0f + sn);
    w *= s; x *= s; y *= s; z *= s;
  }
  qo[0] = w; qo[1] = x; qo[2] = y; qo[3] = z;
}

// std::pow(lambdaFailFac, incTry) (optimizer.cpp:303: float base, int exponent -> double pow).  The
// library pow in double costs ~100 VGPRs on the decision path; for an integer exponent the same
// correctly-rounded value comes out of a double-double running product (Dekker two-product with
// fma: error ~2^-100 before the final rounding).  Exact for the default factor 2.
__device__ __forceinline__ double powi_dd(double b, int n) {
  double hi = 1.0, lo = 0.0;
  for (int i = 0; i < n; ++i) {
    const double p = hi * b;
    const double e = __builtin_fma(hi, b, -p);
    const double l = lo * b + e;
    const double h2 = p + l;
    lo = l - (h2 - p);
    hi = h2;
  }
  return hi + lo;
}

// Damped 6x6 solve A(1+lambda on the diagonal) x = b in double (LDL^T, no
// pivoting: A = J^T W J / n is positive semi-definite).  The reference solves
// the same system with Eigen's float LDLT (optimizer.cpp:258-262); a zero /
// invalid pivot contributes 0 like Eigen's pseudo-inverse of D.
// Aacc: 21 upper-triangle entries (row-major) then 6 rhs.
#define AIDX(i, j) ((i) * 6 - ((i) * ((i)-1)) / 2 + ((j) - (i)))  // upper triangle, i <= j
__device__ __forceinline__ double bcast_lane(double v, int src) {  // wave-uniform copy of lane src's value (v_readlane)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// Row-parallel form: lane i (< 6) owns row i of A and of L; the pivots and the row-j entries a
// step needs are broadcast with v_readlane.  Every element sees exactly the operations of the
// textbook serial loops, in the same order (k ascending), so the result is bit-identical to
// them -- but the serial chain is 6 steps instead of 21 and a lane keeps 12 doubles instead of 33.
// Must be called by all 64 lanes of a wave (lanes >= 6 shadow row 5); x is wave-uniform.
__device__ __forceinline__ void solve6(const double* Aacc, float lambda, float* x, int lane) {
  const double damp = (double)(1.0f + lambda);
  const int row = lane < 6 ? lane : 5;
  double Arow[6], Lrow[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const int lo = row < c ? row : c, hi = row < c ? c : row;
    Arow[c] = Aacc[AIDX(lo, hi)];
    Lrow[c] = 0.0;
  }
  double v = Aacc[21 + row];  // rhs of this row, becomes y then x
  double D[6], Dinv_mine = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double a = (row == j) ? Arow[j] * damp : Arow[j];
#pragma unroll
    for (int k = 0; k < j; ++k) a -= Lrow[k] * bcast_lane(Lrow[k], j) * D[k];
    const double dj = bcast_lane(a, j);
    D[j] = dj;
    const double dinv = (dj > 1e-300) ? 1.0 / dj : 0.0;
    if (row == j) Dinv_mine = dinv;
    if (row > j) Lrow[j] = a * dinv;
  }
  // forward substitution L y = b (k ascending per row), then y *= D^-1
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double yk = bcast_lane(v, k);
    if (row > k) v -= Lrow[k] * yk;
  }
  v *= Dinv_mine;
  // backward substitution L^T x = y: x_i = y_i - sum_{k>i} L[k][i] x_k, k ascending
  double xs[6];
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    const double term = Lrow[i] * v;  // lane k > i: L[k][i] * x_k (x_k is final on those lanes)
    double acc = bcast_lane(v, i);
#pragma unroll
    for (int k = i + 1; k < 6; ++k) acc -= bcast_lane(term, k);
    xs[i] = acc;
    if (row == i) v = acc;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = (float)xs[i];
}

// 64-lane reduce-scatter butterfly: 32 per-lane values -> lane L ends with the
// wave total of value (L >> 1).  32 shuffles instead of 32 x 6.
template <int HALF, int MASK>
__device__ __forceinline__ void butterfly_step(float* v, int lane) {
  const bool up = (lane & MASK) != 0;
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const float send = up ? v[i] : v[HALF + i];
    const float keep = up ? v[HALF + i] : v[i];
    v[i] = keep + __shfl_xor(send, MASK);
  }
}

// ---- cluster all-gather of the 32 per-workgroup partials -------------------------
typedef unsigned long long u64;
// mail layout per pair: [2 (epoch parity)][cluster][32] granules of {epoch<<32 | float bits}
__device__ __forceinline__ bool cluster_allgather(u64* __restrict__ mail_pair, int cluster, int member, unsigned epoch,
                                                  float mine, int lane, double* tot_out) {
  u64* slot = mail_pair + (size_t)(epoch & 1u) * cluster * 32;
  if (lane < 32)
    __hip_atomic_store(&slot[member * 32 + lane], ((u64)epoch << 32) | (u64)__float_as_uint(mine), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  double tot = 0.0;
  for (unsigned spins = 0;; ++spins) {
    bool all = true;
    tot = 0.0;
    if (lane < 32) {
      // all member granules in flight at once (a rolled loop over `cluster` waited for every load before
      // issuing the next: 8 serial L2 round trips per poll, the exchange cost grew linearly with the cluster)
      u64 g[TRACK_MAX_CLUSTER];
#pragma unroll
      for (int j = 0; j < TRACK_MAX_CLUSTER; ++j)
        g[j] = (j < cluster) ? __hip_atomic_load(&slot[j * 32 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
      for (int j = 0; j < TRACK_MAX_CLUSTER; ++j) {
        if (j < cluster) {
          all = all && ((unsigned)(g[j] >> 32) == epoch);
          tot += (double)__uint_as_float((unsigned)g[j]);  // fixed order j = 0..cluster-1: same bits in every member
        }
      }
    }
    if (__all(all)) break;
    if (spins > SPIN_LIMIT) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  *tot_out = tot;
  return true;
}

// explicit global address space: the pointers come out of the descriptor (generic), and
// flat loads would tie up both vmcnt and lgkmcnt
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(1)))* gf32p;
typedef const f4v __attribute__((address_space(1)))* gf4p;

// 12 DT samples around (ix,iy): rows iy-1 (2), iy (4), iy+1 (4), iy+2 (2)
struct DtPatch { float a0, a1, b0, b1, b2, b3, c0, c1, c2, c3, d0, d1; };
template <typename PTR>
__device__ __forceinline__ DtPatch load_patch(PTR dt, int w, int ix, int iy) {
  PTR p = dt + iy * w + ix;
  DtPatch q;
  q.a0 = p[-w]; q.a1 = p[-w + 1];
  q.b0 = p[-1]; q.b1 = p[0]; q.b2 = p[1]; q.b3 = p[2];
  q.c0 = p[w - 1]; q.c1 = p[w]; q.c2 = p[w + 1]; q.c3 = p[w + 2];
  q.d0 = p[2 * w]; q.d1 = p[2 * w + 1];
  return q;
}

struct PtState { float X, Y, Z, dx, dy; int ix, iy; bool valid; };

__device__ __forceinline__ PtState project_point(const f4v p, const float* R, const float* T, float fx, float fy, float cx,
                                                 float cy, float wlim, float hlim, bool in_range) {
  PtState s;
  s.X = ((R[0] * p.x + R[3] * p.y) + R[6] * p.z) + T[0];
  s.Y = ((R[1] * p.x + R[4] * p.y) + R[7] * p.z) + T[1];
  s.Z = ((R[2] * p.x + R[5] * p.y) + R[8] * p.z) + T[2];
  const float u = __fdiv_rn(s.X, s.Z) * fx + cx;
  const float v = __fdiv_rn(s.Y, s.Z) * fy + cy;
  s.valid = in_range && (u > 1.0f && v > 1.0f && u < wlim && v < hlim);  // optimizer.cpp:100 (NaN-safe form)
  s.ix = s.valid ? (int)u : 1;
  s.iy = s.valid ? (int)v : 1;
  s.dx = u - (float)s.ix;
  s.dy = v - (float)s.iy;
  if (!s.valid) { s.X = 0.0f; s.Y = 0.0f; s.Z = 1.0f; s.dx = 0.0f; s.dy = 0.0f; }
  return s;
}

// calcErrorAndBuffers' interpolation + filter + Huber (optimizer.cpp:106-133, optimizer.h:156-185)
// fused with calculateWarpUpdate's Jacobian (optimizer.cpp:218-228) and LGS6::update.
__device__ __forceinline__ void accumulate_point(const PtState& s, const DtPatch& q, float fx, float fy, float ed, bool filt,
                                                 float huber, float* acc) {
  // the reference's table entries at the four corners: (0.5(prev-next), 0.5(up-down), dt)
  const float gx00 = 0.5f * (q.b0 - q.b2), gy00 = 0.5f * (q.a0 - q.c1), d00 = q.b1;
  const float gx10 = 0.5f * (q.b1 - q.b3), gy10 = 0.5f * (q.a1 - q.c2), d10 = q.b2;
  const float gx01 = 0.5f * (q.c0 - q.c2), gy01 = 0.5f * (q.b1 - q.d0), d01 = q.c1;
  const float gx11 = 0.5f * (q.c1 - q.c3), gy11 = 0.5f * (q.b2 - q.d1), d11 = q.c2;
  const float dxdy = s.dx * s.dy;
  const float w11 = dxdy, w01 = s.dy - dxdy, w10 = s.dx - dxdy, w00 = ((1.0f - s.dx) - s.dy) + dxdy;
  float r0 = ((w11 * gx11 + w01 * gx01) + w10 * gx10) + w00 * gx00;
  float r1 = ((w11 * gy11 + w01 * gy01) + w10 * gy10) + w00 * gy00;
  float res = ((w11 * d11 + w01 * d01) + w10 * d10) + w00 * d00;
  const bool good = s.valid && !(res > ed && filt);  // optimizer.cpp:108
  if (!good) { r0 = 0.0f; r1 = 0.0f; res = 0.0f; }
  const float wr = (res <= huber) ? 1.0f : __fdiv_rn(huber, res);
  const float gx = fx * r0, gy = fy * r1;
  const float z = __fdiv_rn(1.0f, s.Z);
  const float zs = z * z;  // reference: 1/(pz*pz), optimizer.cpp:213; differs by <= 1 ulp
  float jv[6];
  jv[0] = z * gx;
  jv[1] = z * gy;
  jv[2] = (-s.X * zs) * gx + (-s.Y * zs) * gy;
  jv[3] = (-s.X * s.Y * zs) * gx + (-(1.0f + s.Y * s.Y * zs)) * gy;
  jv[4] = (1.0f + s.X * s.X * zs) * gx + (s.X * s.Y * zs) * gy;
  jv[5] = (-s.Y * z) * gx + (s.X * z) * gy;
  float wv[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) wv[a] = wr * jv[a];
  {
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int c = a; c < 6; ++c) { acc[k] = fmaf(wv[a], jv[c], acc[k]); ++k; }
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) acc[21 + a] = fmaf(wv[a], res, acc[21 + a]);
  const float r2 = res * res;
  acc[27] = fmaf(wr, r2, acc[27]);
  acc[28] += r2;
  acc[29] += good ? 1.0f : 0.0f;
}

// Register budget.  k_track shares every CU with the pyramid-build kernels of the next batch
// (bench.py / revo_batch_*: build k+1 overlaps track k).  With the compiler's default budget for a
// 512-thread block (256 VGPRs) the kernel took 246, i.e. 2 waves x 246 = 96 % of each SIMD's
// register file: a 32-VGPR k_canny_nms wave could not become resident next to it and the build
// ran 2.3x slower while a tracker was in flight (profiles/r01_overlap_*.txt).  168 VGPRs (the
// 3-waves-per-SIMD budget) is reached without spills once the points kept in registers are cut to
// 4 per thread, the 6x6 solve is row-parallel and pow() is an integer power; that leaves 176
// VGPRs per SIMD to the co-runners.  (128 needs spills on the decision path: slower overall.)
#ifndef TRACK_MAXP
#define TRACK_MAXP 4                  // points of a level kept in registers per thread; the rest streams from L2
#endif
#ifndef TRACK_WAVES_PER_EU
#define TRACK_WAVES_PER_EU 3
#endif

// Two points in flight: both projections, then all 24 DT gathers, then the math.
template <typename PTR>
__device__ __forceinline__ void eval_pair(const f4v p0, const f4v p1, bool has0, bool has1, PTR dtm, int w, const float* R,
                                          const float* T, float fx, float fy, float cx, float cy, float wlim, float hlim,
                                          float ed, bool filt, float huber, float* acc) {
  const PtState s0 = project_point(p0, R, T, fx, fy, cx, cy, wlim, hlim, has0);
  const PtState s1 = project_point(p1, R, T, fx, fy, cx, cy, wlim, hlim, has1);
  const DtPatch q0 = load_patch(dtm, w, s0.ix, s0.iy);
  const DtPatch q1 = load_patch(dtm, w, s1.ix, s1.iy);
  accumulate_point(s0, q0, fx, fy, ed, filt, huber, acc);
  accumulate_point(s1, q1, fx, fy, ed, filt, huber, acc);
}

template <typename PTR>
__device__ __forceinline__ void eval_one(const f4v p0, PTR dtm, int w, const float* R, const float* T, float fx, float fy,
                                         float cx, float cy, float wlim, float hlim, float ed, bool filt, float huber,
                                         float* acc) {
  const PtState s0 = project_point(p0, R, T, fx, fy, cx, cy, wlim, hlim, true);
  const DtPatch q0 = load_patch(dtm, w, s0.ix, s0.iy);
  accumulate_point(s0, q0, fx, fy, ed, filt, huber, acc);
}

// TrackerNew::evalCostFunction's per-point term, tracker.cpp:371-389
template <typename PTR>
__device__ __forceinline__ float cost_point(const f4v p, PTR dtm, int w, int h, const float* R, const float* T, float fx,
                                            float fy, float cx, float cy, float ed, bool filt) {
  const float X = ((R[0] * p.x + R[3] * p.y) + R[6] * p.z) + T[0];
  const float Y = ((R[1] * p.x + R[4] * p.y) + R[7] * p.z) + T[1];
  const float Z = ((R[2] * p.x + R[5] * p.y) + R[8] * p.z) + T[2];
  const float u = __fdiv_rn(fx * X, Z) + cx;
  const float v = __fdiv_rn(fy * Y, Z) + cy;
  float c = 0.0f;
  if (u >= 0 && u < (float)w && v >= 0 && v < (float)h) {
    const float r = dtm[(int)floorf(v) * w + (int)floorf(u)];
    if (!(r > ed && filt)) c = r;
  }
  return c;
}

// One residual evaluation over this thread's share of the level: the first TRACK_MAXP points
// come from registers (loaded once per level), any overflow streams from HBM/L2.
template <typename PTR>
__device__ __forceinline__ void eval_level(const f4v* preg, gf4p pts, int first, int stride, int N, PTR dtm, int w,
                                           const float* R, const float* T, float fx, float fy, float cx, float cy,
                                           float wlim, float hlim, float ed, bool filt, float huber, float* acc) {
#pragma unroll
  for (int k = 0; k < TRACK_MAXP; ++k) {
    // (measured alternatives: two predicated points in flight 0.544 ms vs 0.505 ms for this form;
    //  skipping the cluster exchange on small levels by evaluating them redundantly: 0.501 ms, not kept)
    if (first + k * stride < N) eval_one(preg[k], dtm, w, R, T, fx, fy, cx, cy, wlim, hlim, ed, filt, huber, acc);
  }
  for (int i = first + TRACK_MAXP * stride; i < N; i += stride)
    eval_one(pts[i], dtm, w, R, T, fx, fy, cx, cy, wlim, hlim, ed, filt, huber, acc);
}

template <typename PTR>
__device__ __forceinline__ float cost_level(const f4v* preg, gf4p pts, int first, int stride, int N, PTR dtm, int w, int h,
                                            const float* R, const float* T, float fx, float fy, float cx, float cy, float ed,
                                            bool filt) {
  float cost = 0.0f;
#pragma unroll
  for (int k = 0; k < TRACK_MAXP; ++k)
    if (first + k * stride < N) cost += cost_point(preg[k], dtm, w, h, R, T, fx, fy, cx, cy, ed, filt);
  for (int i = first + TRACK_MAXP * stride; i < N; i += stride)
    cost += cost_point(pts[i], dtm, w, h, R, T, fx, fy, cx, cy, ed, filt);
  return cost;
}

// ---- the kernel ---------------------------------------------------------------
#define TRACK_OCC __attribute__((amdgpu_waves_per_eu(TRACK_WAVES_PER_EU, TRACK_WAVES_PER_EU)))
// ONE: the single pair of the sequential API arrives by value in the kernel-argument segment (no
// H2D copy of the descriptor in front of the launch); otherwise descs[] lives in HBM (batches).
// epoch_base: mailbox epochs keep counting across launches, so the mailbox is never re-zeroed.
template <bool ONE>
__global__ void __launch_bounds__(TRACK_THREADS) TRACK_OCC k_track(const PairDesc one, const PairDesc* __restrict__ descs,
                                                                   TrackParams prm, revo_pair_result* __restrict__ out,
                                                                   EvalOut* __restrict__ eval_out, u64* __restrict__ mail,
                                                                   int n_pairs, int cluster, unsigned epoch_base) {
  __shared__ Ctrl s_ctrl;
  __shared__ W0State s;
  __shared__ float s_part[NWAVES][32];
  // XCD-affine mapping: all members of a pair share blockIdx % 8 (speed only, never correctness)
  const int b = blockIdx.x;
  const int pair = (b / (8 * cluster)) * 8 + (b % 8);
  const int member = (b / 8) % cluster;
  if (pair >= n_pairs) return;
  const PairDesc& d = ONE ? one : descs[pair];
  u64* mail_pair = mail + (size_t)pair * 2 * cluster * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int stride = cluster * TRACK_THREADS;
  const int first = member * TRACK_THREADS + tid;
  unsigned epoch = epoch_base;
  int cur_level = -1;
  f4v preg[TRACK_MAXP > 0 ? TRACK_MAXP : 1];
#pragma unroll
  for (int k = 0; k < TRACK_MAXP; ++k) preg[k] = f4v{0.f, 0.f, 1.f, 1.f};

  if (wave == 0) {  // ---- initial control word
    float R0[9], T0[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R0[i] = d.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) T0[i] = d.T[i];
    int mode = MODE_EVAL, phase = PH_LEVEL_FIRST, level = prm.lvl_begin, flags = 0;
    float Rp[9], Tp[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rp[i] = R0[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) Tp[i] = T0[i];
    if (prm.eval_only) {
      phase = PH_EVAL_ONLY;
    } else if (prm.check_init) {  // tracker.cpp:314 runs BEFORE Sophus::SE3f(R,T) (optimizer.cpp:241)
      mode = MODE_COST; phase = PH_COST_EYE; level = prm.pyr_min_lvl;
#pragma unroll
      for (int i = 0; i < 9; ++i) Rp[i] = (i % 4 == 0) ? 1.0f : 0.0f;
      Tp[0] = Tp[1] = Tp[2] = 0.0f;
    }
    if (lane == 0) {
      float q0[4];
      quat_from_R(R0, q0);
#pragma unroll
      for (int i = 0; i < 4; ++i) s.q[i] = q0[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) s.t[i] = T0[i];
      s.lastErr = s.last_residual = __builtin_nanf("");
      s.lambda = 0.f; s.incsq = 0.f; s.costEye = 0.f;
      s.iteration = 0; s.incTry = 0; s.phase = phase; s.flags = flags; s.total_evals = 0;
      s.good = 0; s.bad = 0; s.sumw = 0.f; s.sumu = 0.f;
#pragma unroll
      for (int i = 0; i < REVO_L; ++i) s.evals[i] = 0;
#ifdef REVO_TRACK_PROFILE
      for (int i = 0; i < 6; ++i) s.prof[i] = 0;
#endif
#pragma unroll
      for (int i = 0; i < 9; ++i) s_ctrl.R[i] = Rp[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) s_ctrl.T[i] = Tp[i];
      if (!prm.eval_only && !prm.check_init && !is_orthogonal(R0)) {  // Sophus::SE3f(R,T) would abort
        s.flags = 2;
        mode = MODE_DONE;
      }
      s_ctrl.level = level;
      s_ctrl.mode = mode;
    }
  }
  __syncthreads();

  for (;;) {
    const int mode = s_ctrl.mode;
    if (mode == MODE_DONE) break;
    const int l = s_ctrl.level;
    float R[9], T[3];  // wave-uniform: keep them in SGPRs
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_ctrl.R[i])));
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_ctrl.T[i])));
    const float fx = prm.cam[l].fx, fy = prm.cam[l].fy, cx = prm.cam[l].cx, cy = prm.cam[l].cy;
    const int w = prm.cam[l].w, h = prm.cam[l].h;
    gf4p pts = (gf4p)d.pts[l];
    gf32p dtm = (gf32p)d.dt[l];
    const int N = d.npts[l];
    ++epoch;
#ifdef REVO_TRACK_PROFILE
    const long long tp0 = clock64();
#endif

    if (l != cur_level) {
      // Level entry (block-uniform): this thread's first points go to registers for every evaluation
      // of the level -- only the pose changes between them (optimizer.cpp:250-305).  (Staging the
      // coarse DT planes in LDS was measured at 0 % gain -- they sit in L2 -- and cost 76.8 KB of
      // LDS per CU that the co-running build kernels need.)
      cur_level = l;
#pragma unroll
      for (int k = 0; k < TRACK_MAXP; ++k) {
        const int i = first + k * stride;
        preg[k] = (i < N) ? pts[i] : f4v{0.f, 0.f, 1.f, 1.f};
      }
    }
    const float ed = prm.edge_distance[l];
    const bool filt = prm.use_edge_filter != 0;
    if (mode == MODE_COST) {
      // TrackerNew::evalCostFunction, tracker.cpp:357-393: nearest-pixel DT lookup
      float cost = cost_level(preg, pts, first, stride, N, dtm, w, h, R, T, fx, fy, cx, cy, ed, filt);
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) cost += __shfl_xor(cost, m);
      if (lane < 32) s_part[wave][lane] = (lane == 0) ? cost : 0.0f;
    } else {
      const float huber = prm.huber_edge;
      const float wlim = (float)(w - 2), hlim = (float)(h - 2);
      float acc[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) acc[k] = 0.0f;
      eval_level(preg, pts, first, stride, N, dtm, w, R, T, fx, fy, cx, cy, wlim, hlim, ed, filt, huber, acc);
      butterfly_step<16, 32>(acc, lane);
      butterfly_step<8, 16>(acc, lane);
      butterfly_step<4, 8>(acc, lane);
      butterfly_step<2, 4>(acc, lane);
      butterfly_step<1, 2>(acc, lane);
      acc[0] += __shfl_xor(acc[0], 1);
      if ((lane & 1) == 0) s_part[wave][lane >> 1] = acc[0];
    }
#ifdef REVO_TRACK_PROFILE
    const long long tp1 = clock64();
#endif
    __syncthreads();
#ifdef REVO_TRACK_PROFILE
    const long long tp2 = clock64();
    long long tp3 = tp2, tp4 = tp2, tp5 = tp2;
#endif

    if (wave == 0) {  // ---- wave 0: totals, cluster exchange, LM decision, next pose (all lanes uniform)
      double tot = 0.0;
      if (lane < 32) {
#pragma unroll
        for (int wv = 0; wv < NWAVES; ++wv) tot += (double)s_part[wv][lane];
      }
#ifdef REVO_TRACK_PROFILE
      tp3 = clock64();
#endif
      bool exchange_ok = true;
      if (cluster > 1) exchange_ok = cluster_allgather(mail_pair, cluster, member, epoch, (float)tot, lane, &tot);
#ifdef REVO_TRACK_PROFILE
      tp4 = clock64();
#endif
      int phase = s.phase;
      int next_mode = MODE_EVAL, next_level = l;
      bool new_candidate = false;  // solve + exp on the accepted state
      bool level_done = false;
      float q[4], t[3];
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = s.q[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) t[i] = s.t[i];
      float lambda = s.lambda;
      int iteration = s.iteration, incTry = s.incTry;
      int total_evals = s.total_evals + 1;
      int flags = s.flags;

      if (!exchange_ok) {  // a cluster member never showed up: give up loudly instead of hanging
        flags |= 8;
        next_mode = MODE_DONE;
      } else if (mode == MODE_COST) {
        const float cost = (float)__shfl(tot, 0);
        if (phase == PH_COST_EYE) {
          if (lane == 0) s.costEye = cost;
          phase = PH_COST_INIT;
          next_mode = MODE_COST;
          if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) s_ctrl.R[i] = d.R[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) s_ctrl.T[i] = d.T[i];
          }
        } else {  // PH_COST_INIT: tracker.cpp:277-282
          float Rs[9], Ts[3];
#pragma unroll
          for (int i = 0; i < 9; ++i) Rs[i] = d.R[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) Ts[i] = d.T[i];
          if (s.costEye < cost) {
#pragma unroll
            for (int i = 0; i < 9; ++i) Rs[i] = (i % 4 == 0) ? 1.0f : 0.0f;
            Ts[0] = Ts[1] = Ts[2] = 0.0f;
            flags |= 1;
          }
          quat_from_R(Rs, q);
          t[0] = Ts[0]; t[1] = Ts[1]; t[2] = Ts[2];
          phase = PH_LEVEL_FIRST;
          next_mode = MODE_EVAL;
          next_level = prm.lvl_begin;
          if (!is_orthogonal(Rs)) { flags |= 2; next_mode = MODE_DONE; }  // Sophus::SE3f(R,T), optimizer.cpp:241
          if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) s_ctrl.R[i] = Rs[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) s_ctrl.T[i] = Ts[i];
          }
        }
      } else {
        const double n_d = __shfl(tot, 29);
        const double sumw_d = __shfl(tot, 27);
        const double sumu_d = __shfl(tot, 28);
        const int good = (int)n_d;
        const float err = __fdiv_rn((float)sumw_d, (float)good);  // optimizer.cpp:190
        if (lane == 0) {
          s.good = good; s.bad = N - good; s.sumw = (float)sumw_d; s.sumu = (float)sumu_d;
          s.evals[l] += 1;
        }
        bool acceptA = false;
        if (phase == PH_EVAL_ONLY) {
          // LGS6 after finish(): A/n, b = -(sum w r v)/n, error = sum w r^2 / n
          const double an = tot / n_d;
          if (lane < 27) s.Aacc[lane] = an;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          if (lane == 0 && member == 0) {
            EvalOut& eo = eval_out[pair];
            int k = 0;
            for (int r = 0; r < 6; ++r)
              for (int c = r; c < 6; ++c) { const float v = (float)s.Aacc[k++]; eo.A[r * 6 + c] = v; eo.A[c * 6 + r] = v; }
            for (int a = 0; a < 6; ++a) eo.b[a] = -(float)s.Aacc[21 + a];
            eo.error = (float)(sumw_d / n_d);
            eo.mean_err = err; eo.sum_w = (float)sumw_d; eo.sum_u = (float)sumu_d;
            eo.good = good; eo.bad = N - good;
          }
          next_mode = MODE_DONE;
        } else if (phase == PH_LEVEL_FIRST) {  // optimizer.cpp:243-250
          acceptA = true;
          if (lane == 0) { s.lastErr = err; s.last_residual = err; }
          lambda = prm.lambda_initial[l];
          iteration = 0;
          if (iteration < prm.max_its[l]) { new_candidate = true; incTry = 0; } else level_done = true;
          phase = PH_LM;
        } else {  // PH_LM: accept / reject, optimizer.cpp:273-304
          const float lastErr = s.lastErr;
          if (err < lastErr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = s.qn[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) t[i] = s.tn[i];
            acceptA = true;
            if (__fdiv_rn(err, lastErr) > prm.convergence_eps[l]) iteration = prm.max_its[l];
            if (lane == 0) { s.lastErr = err; s.last_residual = err; }
            lambda = (lambda <= 0.2f) ? 0.0f : lambda * prm.lambda_success_fac;
            iteration += 1;
            if (iteration < prm.max_its[l]) { new_candidate = true; incTry = 0; } else level_done = true;
          } else {
            if (!(s.incsq > prm.step_size_min[l])) {
              level_done = true;
            } else {
              lambda = (lambda == 0.0f) ? 0.2f : (float)((double)lambda * powi_dd((double)prm.lambda_fail_fac, incTry));
              new_candidate = true;  // same outer iteration, incTry keeps counting
            }
          }
        }
        if (total_evals > MAX_TOTAL_EVALS && !level_done && next_mode != MODE_DONE) { level_done = true; flags |= 4; }
        if (acceptA) {
          const double an = tot / n_d;  // LGS6::finish, LGSX.h:320-326
          if (lane < 27) s.Aacc[lane] = an;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (level_done) {  // optimizer.cpp:308-309 -> next level or done (tracker.cpp:324-340)
          float Rn[9];
          quat_to_R(q, Rn);
          if (l > prm.lvl_end && !(flags & 4)) {
            next_level = l - 1;
            phase = PH_LEVEL_FIRST;
            float q2[4];
            quat_from_R(Rn, q2);  // Sophus::SE3f(R,T) of the next level
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = q2[i];
          } else {
            next_mode = MODE_DONE;
          }
          if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) s_ctrl.R[i] = Rn[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) s_ctrl.T[i] = t[i];
          }
        } else if (new_candidate) {  // optimizer.cpp:258-269
          float inc[6], qn[4], tn[3], Rn[9];
          solve6(s.Aacc, lambda, inc, lane);
          incTry += 1;
          const float incsq = inc[0] * inc[0] + inc[1] * inc[1] + inc[2] * inc[2] + inc[3] * inc[3] + inc[4] * inc[4] + inc[5] * inc[5];
          se3_exp_mul(inc, q, t, qn, tn);
          quat_to_R(qn, Rn);
          if (lane == 0) {
            s.incsq = incsq;
#pragma unroll
            for (int i = 0; i < 4; ++i) s.qn[i] = qn[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) s.tn[i] = tn[i];
#pragma unroll
            for (int i = 0; i < 9; ++i) s_ctrl.R[i] = Rn[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) s_ctrl.T[i] = tn[i];
          }
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s.q[i] = q[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) s.t[i] = t[i];
        s.lambda = lambda; s.iteration = iteration; s.incTry = incTry; s.phase = phase;
        s.flags = flags; s.total_evals = total_evals;
        s_ctrl.level = next_level;
        s_ctrl.mode = next_mode;
      }
#ifdef REVO_TRACK_PROFILE
      tp5 = clock64();
#endif
    }
    __syncthreads();
#ifdef REVO_TRACK_PROFILE
    if (tid == 0) {
      const long long tp6 = clock64();
      s.prof[0] += tp1 - tp0; s.prof[1] += tp2 - tp1; s.prof[2] += tp3 - tp2; s.prof[3] += tp4 - tp3; s.prof[4] += tp5 - tp4; s.prof[5] += tp6 - tp5;
    }
#endif
  }

  if (tid == 0 && member == 0 && !prm.eval_only) {
    revo_pair_result r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.R[i] = s_ctrl.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) r.T[i] = s_ctrl.T[i];
    r.err = s.last_residual;
    r.good = s.good;
    r.bad = s.bad;
    // tracker.cpp:351-352
    r.status = ((double)s.good / (double)s.bad < 4.0) ? REVO_TRACKER_STATE_NEW_KF : REVO_TRACKER_STATE_OK;
#pragma unroll
    for (int i = 0; i < REVO_L; ++i) r.evals[i] = s.evals[i];
#ifdef REVO_TRACK_PROFILE
    for (int i = 0; i < 6; ++i) r.evals[i] = (int)(s.prof[i] / 16);  // profile build: phase cycles / 16
#endif
    r.flags = s.flags;
    r.n_pts0 = d.npts[0];
    out[pair] = r;
  }
}

}  // namespace

int track_blocks_per_cu() {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_track<false>, TRACK_THREADS, 0) != hipSuccess) nb = 1;
  return nb < 1 ? 1 : nb;
}

// Mailbox granules carry {epoch, value}; a launch uses at most MAX_TOTAL_EVALS + a few epochs, so every
// launch gets a fresh window of TRACK_EPOCH_WINDOW epochs and stale granules of earlier launches can
// never match.  The mailbox is zeroed when it is allocated and when the 32-bit counter would wrap.
#define TRACK_EPOCH_WINDOW 8192u
static unsigned next_epoch_base(unsigned* epoch_io, unsigned long long* d_mail, size_t mail_bytes, hipStream_t s) {
  if (*epoch_io > 0xffffffffu - 2 * TRACK_EPOCH_WINDOW) {
    hipMemsetAsync(d_mail, 0, mail_bytes, s);
    *epoch_io = 0;
  }
  const unsigned base = *epoch_io;
  *epoch_io += TRACK_EPOCH_WINDOW;
  return base;
}

void launch_track(const PairDesc* d_descs, const TrackParams& prm, revo_pair_result* d_out, EvalOut* d_eval, int n_pairs,
                  unsigned long long* d_mail, unsigned* epoch_io, int cluster, hipStream_t s) {
  static_assert(MAX_TOTAL_EVALS + 64 < TRACK_EPOCH_WINDOW, "epoch window too small");
  const unsigned base = next_epoch_base(epoch_io, d_mail, sizeof(unsigned long long) * (size_t)n_pairs * 2 * cluster * 32, s);
  const int groups = (n_pairs + 7) / 8;
  hipLaunchKernelGGL(k_track<false>, dim3(groups * 8 * cluster), dim3(TRACK_THREADS), 0, s, PairDesc{}, d_descs, prm, d_out,
                     d_eval, (u64*)d_mail, n_pairs, cluster, base);
}

// one pair, descriptor by value; out / eval_out may be device-visible pinned host memory
void launch_track_one(const PairDesc& desc, const TrackParams& prm, revo_pair_result* out, EvalOut* eval_out,
                      unsigned long long* d_mail, unsigned* epoch_io, int cluster, hipStream_t s) {
  const unsigned base = next_epoch_base(epoch_io, d_mail, sizeof(unsigned long long) * 2 * (size_t)cluster * 32, s);
  hipLaunchKernelGGL(k_track<true>, dim3(8 * cluster), dim3(TRACK_THREADS), 0, s, desc, (const PairDesc*)nullptr, prm, out,
                     eval_out, (u64*)d_mail, 1, cluster, base);
}
