// revo_track.hip -- the edge-alignment tracker as ONE persistent kernel (gfx950).
//
// Replaces TrackerNew::trackFrames (tracker.cpp:294-353), checkInitializationValues
// / evalCostFunction (tracker.cpp:265-283,357-393), Optimizer::trackFrames
// (optimizer.cpp:235-311), calcErrorAndBuffers + calculateWarpUpdate
// (optimizer.cpp:74-234) and LGS6 (LGSX.h:185-404).
//
// A CLUSTER of 512-thread workgroups per frame-pair runs every pyramid level and every
// Levenberg-Marquardt iteration on the device: no host round trip between residual
// evaluations.  The members of a cluster split the point list, all-gather their partial sums
// through 8-byte {epoch,value} granules (agent-scope relaxed atomics, the data is the flag;
// MI355X_MICROARCH.md "handoff" rows) and then take the SAME decision redundantly, so one
// exchange per pass suffices.  Blocks of one pair share blockIdx % 8 (same XCD, same L2).
//
// The serial chain of the reference's LM loop is what bounds the kernel (a pass costs a few
// microseconds of latency, not of arithmetic), so the chain is shortened, not just sped up:
//   * SPECULATIVE CANDIDATES.  ~60 % of the reference's residual evaluations are rejected retries
//     (optimizer.cpp:291-304: same normal equations, larger damping).  All retries of an outer
//     iteration are known in advance from (A, b, lambda, incTry), so one pass evaluates candidate 0
//     in full (residual + Jacobian + normal equations, optimizer.cpp:74-234 fused) and the next
//     KSPEC-1 retries error-only (4 DT samples instead of 12, no Jacobian).  The decision then
//     consumes the results in the reference's order: 43 passes per pair become ~27, the pose and
//     the evaluation counts are those of the reference's sequence.
//   * The init check (tracker.cpp:265-283) evaluates both costs in one pass.
//   * The damped 6x6 solves + SE3 exp of the KSPEC candidates run on KSPEC wavefronts at once.
//   * Levels with few points are evaluated by every member redundantly (identical bits), which
//     removes the cluster exchange from their passes.
//
// The reference's two hot loops (A: warp/project/bilinear gather/Huber, B: 6-vector Jacobian into
// the 6x6 system) are fused, so the 7 scratch buffers of optimizer.h:146-152 never exist: each
// thread keeps the 21 upper-triangle entries of J^T W J, the 6 of J^T W r and per candidate
// sum(w r^2), sum(r^2) and the good count in registers; 64-lane "reduce-scatter" butterflies fold
// them, the per-wave partials meet in LDS and are summed in double in a fixed order
// (deterministic run to run).
//
// The solve is Eigen's LDLT<Matrix6f> restated (optimizer.cpp:262): float, pivoted on the largest
// |diagonal|, left-looking, pseudo-inverse of D -- row-parallel across lanes, every element seeing the
// serial algorithm's operations in the serial order.
//
// The points arrive TILE-ORDERED (revo_pyramid.hip: k_pts_tiles), not in the reference's column-major list order: the 64
// points of a wavefront then project into a small neighbourhood of the keyframe's DT plane, and a gather instruction costs
// the vector L1 one tag lookup per distinct cache line among its lanes -- with the column-major list that lookup rate, not
// latency or HBM, bounded the evaluation (44 lines per gather; 27 now).  The order of a sum is not part of the interface.
//
// Launches of one device are ordered by a resident gate (revo_host.hip): every workgroup counts itself into a census
// word when it starts, and the next grid may start once this one is completely resident -- two grids in flight, the
// older one always whole on the chip.
//
// The gradient/DT float4 table of the reference (imgpyramidrgbd.cpp:255-276) is NOT read here: the
// kernel samples the 4x smaller DT plane and forms the corner gradients 0.5*(dt[i-1]-dt[i+1]),
// 0.5*(dt[i-w]-dt[i+w]) on the fly -- the same float operations, hence the same values.
//
// There is no dense contraction here (a 6-vector outer product per point), so no MFMA.
#include "revo_dev.h"
#include "revo_div.h"

namespace {

enum { MODE_EVAL = 0, MODE_COST = 1, MODE_DONE = 2 };
enum { PH_FIRST = 0, PH_LM = 1, PH_REFILL = 2, PH_EVAL_ONLY = 3 };
#define NWAVES (TRACK_THREADS / 64)
#define KMAX TRACK_KMAX
#define NVAL TRACK_NVAL          // granules a workgroup publishes per pass: 32 float slots + 8 double slots as two words each
#define CSLOT 27                 // float slots: 0..26 normal equations (21 + 6), CSLOT + j = good-point count of candidate j
#define NSUM 40                  // per-wave partials in LDS, all as double: 32 float slots, then 8 double slots (DSLOT below;
                                 // MODE_COST: slots 0 / 2 = the two costs)
#define SPIN_LIMIT 400000      // bounded cluster wait (~0.5 s): never hang the GPU
#define MAX_TOTAL_EVALS 6000  // hang guard; the reference bound is 100 outer iterations x retries
static_assert(CSLOT + KMAX <= 32 && NVAL == 48 && NSUM == 40 && 1 + KMAX <= 8, "slot layout");

struct Cand {  // one pose to evaluate (R, T: what the evaluating waves read); for LM candidates also what the decision needs
  float R[9], T[3];    // R column-major; T is also the translation of Sophus::SE3f new_referenceToFrame (optimizer.cpp:266)
  float q[4];          // its quaternion
  float incsq;         // inc.dot(inc) (optimizer.cpp:296)
  float pad[3];
};
struct PassCtl { int mode, level, ncand, phase; };
struct SolverState {   // the LM state between passes: one PRIVATE copy per solver wave (they all take the same decisions)
  float q[4], t[3];    // accepted pose (Sophus::SE3f referenceToFrame)
  float lastErr;       // == last_residual (optimizer.cpp:276)
  float lambda;
  int iteration, incTry, total_evals;
  float lamc[KMAX];    // dampings of the candidates in flight; candidate j was solved with incTry + 1 + j (optimizer.cpp:263)
  int flags, good, bad, pad;
};

// wave-uniform control flow: a condition every lane computes alike, as a scalar the compiler can branch on without exec masks
#define UNI(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
__device__ __forceinline__ int rfl_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- small algebra (Eigen/Sophus semantics, float like the reference) -------
// Eigen's Quaternionf(Matrix3f): four algebraically equivalent branches chosen by the
// largest of (trace, m00, m11, m22).  Written branch-free with selects so the quaternion
// stays in registers.
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

// Sophus::SE3f::exp(inc) * (q,t)  (se3.hpp:723-745, so3.hpp:531-565, se3.hpp:317-321, so3.hpp:335-352).
// The coefficients sin(theta/2)/theta, cos(theta/2), (1-cos theta)/theta^2, (theta-sin theta)/theta^3 are evaluated as
// series in theta^2 below 0.5 rad: exact to float rounding there, and without the square root, the three
// divisions and the catastrophic cancellation of the closed forms ((1-cos theta)/theta^2 in float loses three
// digits at theta = 0.01) -- on the serial path of every LM pass this is ~0.5 us.  Sophus' own small-angle
// branch (theta < 1e-5: Taylor quaternion, V = R) is kept as it is.
// Every value here is wave-uniform (the solver wave's lanes all hold the same numbers): the branches are scalar (UNI).
// V = I + ca Omega + cb Omega^2 is written out with the structural zeros of Omega folded by hand (round 6): the products
// with 0 and the sums with +-0 of the generic 3x3 products return their other operand exactly, so every entry keeps the
// value the matrix form gives it (only the SIGN of an exact zero can differ); 27 operations instead of 81 on the
// serial path of every pass.
__device__ __forceinline__ void se3_exp_mul(const float* a, const float* q, const float* t, float* qo, float* to) {
  const float ox = a[3], oy = a[4], oz = a[5];
  const float t2 = ox * ox + oy * oy + oz * oz;
  float imag, real, ca, cb;
  if (UNI(t2 < 0.25f)) {
    imag = fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 1.0f / 185794560.0f, -1.0f / 645120.0f), 1.0f / 3840.0f), -1.0f / 48.0f), 0.5f);
    real = fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, -1.0f / 3715891200.0f, 1.0f / 10321920.0f), -1.0f / 46080.0f), 1.0f / 384.0f), -0.125f), 1.0f);
    ca = fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 1.0f / 3628800.0f, -1.0f / 40320.0f), 1.0f / 720.0f), -1.0f / 24.0f), 0.5f);
    cb = fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 1.0f / 39916800.0f, -1.0f / 362880.0f), 1.0f / 5040.0f), -1.0f / 120.0f), 1.0f / 6.0f);
  } else {  // a diverging step (also NaN)
    const float theta = sqrtf(t2), h = 0.5f * theta;
    imag = __fdiv_rn(sinf(h), theta);
    real = cosf(h);
    ca = __fdiv_rn(1.0f - cosf(theta), t2);
    cb = __fdiv_rn(theta - sinf(theta), t2 * theta);
  }
  const float qe[4] = {real, imag * ox, imag * oy, imag * oz};
  float te[3];
  if (UNI(t2 < 1e-10f)) {  // theta < Constants<float>::epsilon() = 1e-5 (common.hpp:150-158): V = matrix(qe)
    float Rm[9];
    quat_to_R(qe, Rm);  // column-major: V[r][c] = Rm[c * 3 + r]
#pragma unroll
    for (int r = 0; r < 3; ++r) te[r] = Rm[r] * a[0] + Rm[3 + r] * a[1] + Rm[6 + r] * a[2];
  } else {
    // Omega = [0 -oz oy; oz 0 -ox; -oy ox 0];  Omega^2: diagonal -(oy^2 + oz^2) ..., off-diagonal ox oy ...
    const float xx = ox * ox, yy = oy * oy, zz = oz * oz, xy = oy * ox, xz = oz * ox, yz = oz * oy;
    const float d0 = (-zz) + (-yy), d1 = (-zz) + (-xx), d2 = (-yy) + (-xx);
    const float V00 = 1.0f + cb * d0, V11 = 1.0f + cb * d1, V22 = 1.0f + cb * d2;
    const float px = ca * ox, py = ca * oy, pz = ca * oz, sxy = cb * xy, sxz = cb * xz, syz = cb * yz;
    const float V01 = (-pz) + sxy, V10 = pz + sxy;
    const float V02 = py + sxz, V20 = (-py) + sxz;
    const float V12 = (-px) + syz, V21 = px + syz;
    te[0] = V00 * a[0] + V01 * a[1] + V02 * a[2];
    te[1] = V10 * a[0] + V11 * a[1] + V12 * a[2];
    te[2] = V20 * a[0] + V21 * a[1] + V22 * a[2];
  }
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
  if (UNI(sn != 1.0f)) {
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
// LM_lambda after a rejected candidate (optimizer.cpp:300-303)
__device__ __forceinline__ float lambda_after_reject(float lambda, float fail_fac, int incTry) {
  if (lambda == 0.0f) return 0.2f;
  if (fail_fac == 2.0f) return ldexpf(lambda, incTry);  // (float)((double)lambda * 2^incTry), exactly (the default factor)
  return (float)((double)lambda * powi_dd((double)fail_fac, incTry));
}

// ---- Eigen LDLT<Matrix6f>::compute + solve (optimizer.cpp:262), float -------------------------
// ldlt_inplace<Lower>::unblocked pivots on the largest |diagonal| of the ORIGINAL diagonal entries (it is
// left-looking: a diagonal entry is only updated in the step that eliminates it), so the transposition
// sequence is known up front and the factorisation is an unpivoted left-looking LDL^T of P A P^T.
#define AIDX(i, j) ((i) * 6 - ((i) * ((i)-1)) / 2 + ((j) - (i)))  // upper triangle, i <= j
__device__ __forceinline__ float rl(float v, int src) {  // wave-uniform copy of lane src's value (v_readlane)
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), src));
}
// abv: lane k (< 27) of the calling wave holds entry k of the normalised normal equations (21 upper-triangle
// entries of A, then 6 of the rhs).  Must be called by all 64 lanes of a wave (lanes >= 6 shadow row 5); x is
// wave-uniform.
__device__ __forceinline__ float rl_dyn(float v, int src) {  // src: wave-uniform, not a constant
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), src));
}
__device__ __forceinline__ void solve6_ldlt(float abv_in, float lambda, float* x, int lane) {
  const float damp = 1.0f + lambda;  // A(i,i) *= 1 + LM_lambda, optimizer.cpp:261
  // the damped system, entry for entry: lanes AIDX(i,i) = 0, 6, 11, 15, 18, 20 hold the diagonal
  const float abv = (lane < 21 && ((0x148841u >> lane) & 1u)) ? abv_in * damp : abv_in;
  // transpositions on the damped diagonal: big = first maximum of |diag| among positions >= k
  float dv[6];
  int idx[6];  // idx[i] = original row/column at position i of P A P^T
#pragma unroll
  for (int i = 0; i < 6; ++i) { dv[i] = fabsf(rl(abv, AIDX(i, i))); idx[i] = i; }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int big = k;
    float bigv = dv[k];
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
      if (dv[i] > bigv) { bigv = dv[i]; big = i; }
    // swap positions k and big (static positions, wave-uniform selects)
    const float dk = dv[k];
    const int ik = idx[k];
    int ib = ik;
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
      if (i == big) { ib = idx[i]; idx[i] = ik; dv[i] = dk; }
    idx[k] = ib;
    dv[k] = bigv;
  }
  // row `row` of the permuted, damped matrix (lower triangle is what the algorithm touches)
  const int row = lane < 6 ? lane : 5;
  int myorig = idx[0];
#pragma unroll
  for (int i = 1; i < 6; ++i) myorig = (row == i) ? idx[i] : myorig;
  float m[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const int oc = idx[c];
    const int lo = myorig < oc ? myorig : oc, hi = myorig < oc ? oc : myorig;
    m[c] = __shfl(abv, hi + ((lo * (11 - lo)) >> 1));  // AIDX(lo, hi)
  }
  float v = __shfl(abv, 21 + myorig);  // m_transpositions * rhs
  float D[6], w[6];  // w[c] = L(:,c) * D[c]: lane k holds temp[c] = D[c] * L(k,c) of step k (the same product, commuted)
  float mydiag = 0.0f;
  bool stop = false;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (k > 0 && !stop) {
      float a2 = 0.0f;
#pragma unroll
      for (int c = 0; c < k; ++c) a2 = a2 + m[c] * rl(w[c], k);  // temp = D(0..k) .* A10^T
      if (row >= k) m[k] = m[k] - a2;         // A(k,k) -= A10 temp ; A21 -= A20 temp
    }
    const float akk = rl(m[k], k);
    D[k] = akk;
    mydiag = (row == k) ? akk : mydiag;
    const bool valid = fabsf(akk) > 0.0f;
    if (k == 0 && !valid) stop = true;        // the whole diagonal is zero: Eigen leaves the matrix as it is
    if (!stop && valid && row > k) m[k] = __fdiv_rn(m[k], akk);
    w[k] = m[k] * akk;
  }
  // L y = P b (j ascending per row)
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float yj = rl(v, j);
    if (row > j) v = v - m[j] * yj;
  }
  // pseudo-inverse of D (tolerance = 1 / highest())
  v = (fabsf(mydiag) > 1.17549435e-38f) ? __fdiv_rn(v, mydiag) : 0.0f;
  // L^T z = y: z_i = y_i - sum_{j>i} L[j][i] z_j, j ascending; lane i ends with z_i
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    const float term = m[i] * v;  // lane j > i: L[j][i] * z_j (z_j is final on those lanes)
    float acc = rl(v, i);
#pragma unroll
    for (int j = i + 1; j < 6; ++j) acc = acc - rl(term, j);
    if (row == i) v = acc;
  }
  // P^T z: x[o] = z at the position that holds original index o
#pragma unroll
  for (int o = 0; o < 6; ++o) {
    const unsigned long long at = __builtin_amdgcn_ballot_w64(lane < 6 && myorig == o);
    x[o] = rl_dyn(v, (int)__builtin_ctzll(at));
  }
}

// ---- wave-level reduce-scatter ---------------------------------------------------------------
// V per-lane values -> every value's wave total.  The halving steps exchange with lane ^ MASK; the masks
// inside a 16-lane row are DPP lane permutations fused into VALU moves (no LDS crossbar, a few cycles), and
// they come FIRST, when there are many values; only the last one or two values cross rows (ds_bpermute).
template <int MASK>
__device__ __forceinline__ float lane_xor(float v) {
  const int s = (int)__float_as_uint(v);
  int r;
  if (MASK == 1) r = __builtin_amdgcn_update_dpp(0, s, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
  else if (MASK == 2) r = __builtin_amdgcn_update_dpp(0, s, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  else if (MASK == 4) {
    r = __builtin_amdgcn_update_dpp(0, s, 0x104, 0xf, 0x5, false);                   // row_shl:4 into banks 0, 2 (lane bit 2 clear)
    r = __builtin_amdgcn_update_dpp(r, s, 0x114, 0xf, 0xa, false);                   // row_shr:4 into banks 1, 3
  } else if (MASK == 8) r = __builtin_amdgcn_update_dpp(0, s, 0x128, 0xf, 0xf, true);  // row_ror:8
  else return __shfl_xor(v, MASK);
  return __uint_as_float((unsigned)r);
}
template <int HALF, int MASK>
__device__ __forceinline__ void butterfly_step(float* v, int lane) {
  const bool up = (lane & MASK) != 0;
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const float send = up ? v[i] : v[HALF + i];
    const float keep = up ? v[HALF + i] : v[i];
    v[i] = keep + lane_xor<MASK>(send);
  }
}
// 32 values: lane L ends with the wave total of value idx32(L) in v[0]
__device__ __forceinline__ void reduce32(float* v, int lane) {
  butterfly_step<16, 1>(v, lane);
  butterfly_step<8, 2>(v, lane);
  butterfly_step<4, 4>(v, lane);
  butterfly_step<2, 8>(v, lane);
  butterfly_step<1, 16>(v, lane);
  v[0] += lane_xor<32>(v[0]);
}
__device__ __forceinline__ int idx32(int lane) {
  return ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
}
// 8 DOUBLE values (the candidates' error sums): lane L ends with the wave total of value idx8(L) in v[0].  Same tree as
// above; a double crosses lanes as its two words.
template <int MASK>
__device__ __forceinline__ double lane_xor_d(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = __float_as_uint(lane_xor<MASK>(__uint_as_float((unsigned)u)));
  const unsigned hi = __float_as_uint(lane_xor<MASK>(__uint_as_float((unsigned)(u >> 32))));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int HALF, int MASK>
__device__ __forceinline__ void butterfly_step_d(double* v, int lane) {
  const bool up = (lane & MASK) != 0;
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const double send = up ? v[i] : v[HALF + i];
    const double keep = up ? v[HALF + i] : v[i];
    v[i] = keep + lane_xor_d<MASK>(send);
  }
}
__device__ __forceinline__ void reduce8d(double* v, int lane) {
  butterfly_step_d<4, 1>(v, lane);
  butterfly_step_d<2, 2>(v, lane);
  butterfly_step_d<1, 4>(v, lane);
  v[0] += lane_xor_d<8>(v[0]);
  v[0] += lane_xor_d<16>(v[0]);
  v[0] += lane_xor_d<32>(v[0]);
}
__device__ __forceinline__ int idx8(int lane) { return ((lane & 1) << 2) | (lane & 2) | ((lane & 4) >> 2); }

// ---- cluster all-gather of the per-workgroup partials -------------------------
typedef unsigned long long u64;
// mail layout per pair: [2 (epoch parity)][cluster][NVAL] granules of {epoch << 32 | 32 payload bits}.  Granules 0..31 carry
// the float slots (the workgroup's double sum rounded to float), granules 32 + 2k / 33 + 2k the low / high word of double
// slot k (the candidates' error sums travel unrounded).
// tot: lane v < 32 holds float slot v, lanes 32 + 2k and 33 + 2k both hold double slot k
__device__ __forceinline__ void cluster_publish(u64* __restrict__ mail_pair, int cluster, int member, unsigned epoch, double tot,
                                                int lane) {
  u64* slot = mail_pair + (size_t)(epoch & 1u) * cluster * NVAL;
  const u64 bits = (u64)__double_as_longlong(tot);
  const unsigned word = lane < 32 ? __float_as_uint((float)tot) : ((lane & 1) ? (unsigned)(bits >> 32) : (unsigned)bits);
  if (lane < NVAL)
    __hip_atomic_store(&slot[member * NVAL + lane], ((u64)epoch << 32) | (u64)word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every member granule in flight at once (a rolled loop would wait for each load before issuing the next:
// `cluster` serial L2 round trips per poll).  The sums are formed once, after the last granule has arrived.
// NC = the cluster size as a compile-time constant for the shapes the library picks (no per-member `j < cluster` predicate:
// with the generic sweep those were 56 loop-invariant conditions, hoisted in front of the pass loop as exec-mask SGPR pairs and
// spilled to VGPR lanes -- two v_readlane on the serial path for every one of them); NC = 0: any size up to TRACK_MAX_CLUSTER,
// the members in order behind scalar branches.
template <int NC>
__device__ __forceinline__ bool cluster_poll(const u64* __restrict__ slot, int cluster, unsigned epoch, int lane, double* tot_out) {
  const bool isd = lane >= 32, odd = (lane & 1) != 0;
  const int my = lane < NVAL ? lane : 0;
  double tot = 0.0;
  if (NC > 0) {
    u64 g[NC > 0 ? NC : 1];
    for (unsigned spins = 0;; ++spins) {
      bool all = true;
#pragma unroll
      for (int j = 0; j < NC; ++j) g[j] = __hip_atomic_load(&slot[j * NVAL + my], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int j = 0; j < NC; ++j) all = all && ((unsigned)(g[j] >> 32) == epoch);
      if (__all(all)) break;
      if (spins > SPIN_LIMIT) return false;
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {  // fixed order j = 0..cluster-1: same bits in every member
      const unsigned wd = (unsigned)g[j];
      const unsigned pw = __float_as_uint(lane_xor<1>(__uint_as_float(wd)));  // the other word of the pair
      const double dd = __longlong_as_double((long long)(((u64)(odd ? wd : pw) << 32) | (u64)(odd ? pw : wd)));
      tot += isd ? dd : (double)__uint_as_float(wd);
    }
  } else {  // any other size: member by member (a granule that carries the epoch stays as it is until everybody has gathered it)
    for (unsigned spins = 0;; ++spins) {
      bool all = true;
      for (int j = 0; j < cluster; ++j)
        all = all && ((unsigned)(__hip_atomic_load(&slot[j * NVAL + my], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == epoch);
      if (__all(all)) break;
      if (spins > SPIN_LIMIT) return false;
      __builtin_amdgcn_s_sleep(1);
    }
    for (int j = 0; j < cluster; ++j) {
      const unsigned wd = (unsigned)__hip_atomic_load(&slot[j * NVAL + my], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned pw = __float_as_uint(lane_xor<1>(__uint_as_float(wd)));
      const double dd = __longlong_as_double((long long)(((u64)(odd ? wd : pw) << 32) | (u64)(odd ? pw : wd)));
      tot += isd ? dd : (double)__uint_as_float(wd);
    }
  }
  *tot_out = tot;
  return true;
}
__device__ __forceinline__ bool cluster_gather(const u64* __restrict__ mail_pair, int cluster, unsigned epoch, int lane,
                                               double* tot_out) {
  const u64* slot = mail_pair + (size_t)(epoch & 1u) * cluster * NVAL;
  switch (cluster) {  // wave-uniform (a kernel argument)
    case 2: return cluster_poll<2>(slot, cluster, epoch, lane, tot_out);
    case 3: return cluster_poll<3>(slot, cluster, epoch, lane, tot_out);
    case 4: return cluster_poll<4>(slot, cluster, epoch, lane, tot_out);
    case 6: return cluster_poll<6>(slot, cluster, epoch, lane, tot_out);
    case 8: return cluster_poll<8>(slot, cluster, epoch, lane, tot_out);
    case 16: return cluster_poll<16>(slot, cluster, epoch, lane, tot_out);
    case 32: return cluster_poll<32>(slot, cluster, epoch, lane, tot_out);
    default: return cluster_poll<0>(slot, cluster, epoch, lane, tot_out);
  }
}

// explicit global address space: the pointers come out of the descriptor (generic), and
// flat loads would tie up both vmcnt and lgkmcnt
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(1)))* gf32p;
typedef const f4v __attribute__((address_space(1)))* gf4p;

// 12 DT samples around (ix,iy): rows iy-1 (2), iy (4), iy+1 (4), iy+2 (2)
struct DtPatch { float a0, a1, b0, b1, b2, b3, c0, c1, c2, c3, d0, d1; };
// 4-byte aligned vector types: the middle rows of the patch are ONE dwordx4 gather each (global loads only need
// dword alignment) -- a gather instruction costs the vector L1 one tag lookup per distinct cache line among the 64
// lanes, whatever its width, and that lookup rate is part of what bounds the evaluation of the fine levels
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef const f4u __attribute__((address_space(1)))* gf4up;
typedef const f2u __attribute__((address_space(1)))* gf2up;
__device__ __forceinline__ DtPatch load_patch(gf32p dt, int w, int ix, int iy) {
  gf32p p = dt + iy * w + ix;
  DtPatch q;
  const f2u a = *(gf2up)(p - w);
  const f4u b = *(gf4up)(p - 1);
  const f4u c = *(gf4up)(p + w - 1);
  const f2u d = *(gf2up)(p + 2 * w);
  q.a0 = a.x; q.a1 = a.y;
  q.b0 = b.x; q.b1 = b.y; q.b2 = b.z; q.b3 = b.w;
  q.c0 = c.x; q.c1 = c.y; q.c2 = c.z; q.c3 = c.w;
  q.d0 = d.x; q.d1 = d.y;
  return q;
}

struct PtState { float X, Y, Z, rz, dx, dy; int ix, iy; bool valid; };  // rz: refined 1/Z (revo_div.h)
struct Cam { float fx, fy, cx, cy, wlim, hlim; int w, h; };

__device__ __forceinline__ PtState project_point(const f4v p, const float* R, const float* T, const Cam& c, bool in_range) {
  PtState s;
  s.X = ((R[0] * p.x + R[3] * p.y) + R[6] * p.z) + T[0];
  s.Y = ((R[1] * p.x + R[4] * p.y) + R[7] * p.z) + T[1];
  s.Z = ((R[2] * p.x + R[5] * p.y) + R[8] * p.z) + T[2];
  // X/Z and Y/Z, correctly rounded, off ONE refined reciprocal (revo_div.h: the same bits as two IEEE divisions)
  s.rz = revo_recip_refined(s.Z);
  const float u = revo_div_with(s.X, s.Z, s.rz) * c.fx + c.cx;
  const float v = revo_div_with(s.Y, s.Z, s.rz) * c.fy + c.cy;
  s.valid = in_range && (u > 1.0f && v > 1.0f && u < c.wlim && v < c.hlim);  // optimizer.cpp:100 (NaN-safe form)
  s.ix = s.valid ? (int)u : 1;
  s.iy = s.valid ? (int)v : 1;
  s.dx = u - (float)s.ix;
  s.dy = v - (float)s.iy;
  if (!s.valid) { s.X = 0.0f; s.Y = 0.0f; s.Z = 1.0f; s.rz = 1.0f; s.dx = 0.0f; s.dy = 0.0f; }
  return s;
}

// The candidate's error terms.  The LM's accept / stop decisions compare exactly these sums (optimizer.cpp:129-133,273-278:
// `error < lastErr`, `error / lastErr > 0.999`), so they are the one place where the ORDER of a float sum would show: the
// per-point terms are the reference's float values (w_r * res_2 rounded to float, res_2), and the sums are carried in DOUBLE from
// the first addition on -- per thread, through the butterflies, LDS and the cluster exchange -- and rounded to float once, like the
// reference's accumulator read at the end (~1e-16 relative whatever the order, the cluster size or the speculation depth: a pose
// has ONE error however it was evaluated).  The good-point count rides in the float butterfly (exact: < 2^24).
// Double slots of a pass: 0 = sum w r^2 of candidate 0, 1 = its sum r^2 (ResidualInfo::sumErrorUnweighted: reported, never
// compared), 1 + j = sum w r^2 of the error-only retry j (j = 1..KMAX-1; nothing reads a retry's unweighted sum).
#define DSLOT(j) ((j) == 0 ? 0 : 1 + (j))
template <bool WITH_UNWEIGHTED>
__device__ __forceinline__ void accumulate_error(float res, float wr, bool good, double* ed, float* cnt) {
  const float r2 = res * res;
  ed[0] += (double)(wr * r2);
  if (WITH_UNWEIGHTED) ed[1] += (double)r2;
  *cnt += good ? 1.0f : 0.0f;
}

// calcErrorAndBuffers' interpolation + filter + Huber (optimizer.cpp:106-133, optimizer.h:156-185)
// fused with calculateWarpUpdate's Jacobian (optimizer.cpp:218-228) and LGS6::update.
__device__ __forceinline__ void accumulate_point(const PtState& s, const DtPatch& q, float fx, float fy, float edist, bool filt,
                                                 float huber, float* acc, double* ed) {
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
  const bool good = s.valid && !(res > edist && filt);  // optimizer.cpp:108
  if (!good) { r0 = 0.0f; r1 = 0.0f; res = 0.0f; }
  const float wr = (res <= huber) ? 1.0f : revo_div(huber, res);
  const float gx = fx * r0, gy = fy * r1;
  const float z = revo_div_with(1.0f, s.Z, s.rz);
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
  accumulate_error<true>(res, wr, good, ed, acc + CSLOT);
}

__device__ __forceinline__ void full_point(const f4v p, gf32p dtm, const float* R, const float* T, const Cam& c, float edist,
                                           bool filt, float huber, float* acc, double* ed) {
  const PtState s = project_point(p, R, T, c, true);
  const DtPatch q = load_patch(dtm, c.w, s.ix, s.iy);
  accumulate_point(s, q, c.fx, c.fy, edist, filt, huber, acc, ed);
}

// TrackerNew::evalCostFunction's per-point term, tracker.cpp:371-389
__device__ __forceinline__ float cost_point(const f4v p, gf32p dtm, const float* R, const float* T, const Cam& c, float ed,
                                            bool filt) {
  const float X = ((R[0] * p.x + R[3] * p.y) + R[6] * p.z) + T[0];
  const float Y = ((R[1] * p.x + R[4] * p.y) + R[7] * p.z) + T[1];
  const float Z = ((R[2] * p.x + R[5] * p.y) + R[8] * p.z) + T[2];
  const float rz = revo_recip_refined(Z);
  const float u = revo_div_with(c.fx * X, Z, rz) + c.cx;
  const float v = revo_div_with(c.fy * Y, Z, rz) + c.cy;
  float cost = 0.0f;
  if (u >= 0 && u < (float)c.w && v >= 0 && v < (float)c.h) {
    const float r = dtm[(int)floorf(v) * c.w + (int)floorf(u)];
    if (!(r > ed && filt)) cost = r;
  }
  return cost;
}

// Register budget.  k_track shares every CU with the pyramid-build kernels of the next batch
// (bench.py / revo_batch_*: build k+1 overlaps track k): 168 VGPRs (the 3-waves-per-SIMD budget)
// leaves 176 VGPRs per SIMD to the co-runners (profiles/r01_overlap_vgpr_study.txt).
#ifndef TRACK_MAXP
#define TRACK_MAXP 2                  // points of a level kept in registers per thread; the rest streams from L2
#endif
#ifndef TRACK_WAVES_PER_EU
#define TRACK_WAVES_PER_EU 3
#endif

// wave-uniform pose of a candidate: SGPRs
__device__ __forceinline__ void load_pose(const Cand& c, float* R, float* T) {
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(c.R[i])));
#pragma unroll
  for (int i = 0; i < 3; ++i) T[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(c.T[i])));
}

// Candidate 0 in full and NE error-only retries of the SAME point in one go.  A retry is a shorter step from the same pose
// (optimizer.cpp:291-304: same normal equations, larger damping); near convergence -- where the rejections are -- it lands on
// candidate 0's pixel or a direct neighbour, and then its four DT samples are already in the 12-sample patch: they are
// picked out of registers instead of costing the vector L1 another ~27 tag lookups per gather.  Lanes whose retry moved
// farther gather as before, under the exec mask (a gather is priced by the distinct lines of its ACTIVE lanes).  Same
// values from the same addresses, same accumulation order per candidate: the sums do not change by a bit.
template <int NE>
__device__ __forceinline__ void fused_point(const f4v p, gf32p dtm, const float* R0, const float* T0, const float (*R)[9],
                                            const float (*T)[3], const Cam& c, float edist, bool filt, float huber, float* acc,
                                            double* ed) {
  const PtState s0 = project_point(p, R0, T0, c, true);
  PtState s[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) s[j] = project_point(p, R[j], T[j], c, true);
  const DtPatch q = load_patch(dtm, c.w, s0.ix, s0.iy);
  float g00[NE], g10[NE], g01[NE], g11[NE];
  int ddx[NE], ddy[NE];
  bool hit[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    ddx[j] = s[j].ix - s0.ix;
    ddy[j] = s[j].iy - s0.iy;
    hit[j] = (ddy[j] == 0 && ddx[j] >= -1 && ddx[j] <= 1) || (ddx[j] == 0 && (ddy[j] == 1 || ddy[j] == -1));
    g00[j] = g10[j] = g01[j] = g11[j] = 0.0f;
    if (!hit[j]) {
      gf32p t = dtm + s[j].iy * c.w + s[j].ix;
      g00[j] = t[0]; g10[j] = t[1]; g01[j] = t[c.w]; g11[j] = t[c.w + 1];
    }
  }
  accumulate_point(s0, q, c.fx, c.fy, edist, filt, huber, acc, ed);
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    float d00 = g00[j], d10 = g10[j], d01 = g01[j], d11 = g11[j];
    if (hit[j]) {
      const bool same_row = ddy[j] == 0, up = ddy[j] < 0, left = ddx[j] < 0, mid = ddx[j] == 0;
      // rows iy0 / iy0+1 shifted by ddx, or rows iy0-1 / iy0, or rows iy0+1 / iy0+2 (ddx == 0)
      d00 = same_row ? (left ? q.b0 : (mid ? q.b1 : q.b2)) : (up ? q.a0 : q.c1);
      d10 = same_row ? (left ? q.b1 : (mid ? q.b2 : q.b3)) : (up ? q.a1 : q.c2);
      d01 = same_row ? (left ? q.c0 : (mid ? q.c1 : q.c2)) : (up ? q.b1 : q.d0);
      d11 = same_row ? (left ? q.c1 : (mid ? q.c2 : q.c3)) : (up ? q.b2 : q.d1);
    }
    const float dxdy = s[j].dx * s[j].dy;
    const float w11 = dxdy, w01 = s[j].dy - dxdy, w10 = s[j].dx - dxdy, w00 = ((1.0f - s[j].dx) - s[j].dy) + dxdy;
    float res = ((w11 * d11 + w01 * d01) + w10 * d10) + w00 * d00;
    const bool good = s[j].valid && !(res > edist && filt);
    if (!good) res = 0.0f;
    const float wr = (res <= huber) ? 1.0f : revo_div(huber, res);
    accumulate_error<false>(res, wr, good, ed + 2 + j, acc + CSLOT + 1 + j);
  }
}

template <int NE>
__device__ __forceinline__ void fused_block(const Cand* cand, const f4v* preg, gf4p pts, int first, int stride, int N, gf32p dtm,
                                            const Cam& cam, float edist, bool filt, float huber, float* acc, double* ed) {
  float R0[9], T0[3], R[NE][9], T[NE][3];
  load_pose(cand[0], R0, T0);
#pragma unroll
  for (int j = 0; j < NE; ++j) load_pose(cand[1 + j], R[j], T[j]);
#pragma unroll
  for (int k = 0; k < TRACK_MAXP; ++k)
    if (first + k * stride < N) fused_point<NE>(preg[k], dtm, R0, T0, R, T, cam, edist, filt, huber, acc, ed);
  for (int i = first + TRACK_MAXP * stride; i < N; i += stride) fused_point<NE>(pts[i], dtm, R0, T0, R, T, cam, edist, filt, huber, acc, ed);
}

// ---- the kernel ---------------------------------------------------------------
#define TRACK_OCC __attribute__((amdgpu_waves_per_eu(TRACK_WAVES_PER_EU, TRACK_WAVES_PER_EU)))
// ONE: the single pair of the sequential API arrives by value in the kernel-argument segment (no
// H2D copy of the descriptor in front of the launch); otherwise descs[] lives in HBM (batches).
// epoch_base: mailbox epochs keep counting across launches, so the mailbox is never re-zeroed.
//
// One PASS = evaluate the candidates of s_pass/s_cand (all waves) -> barrier -> the kspec "solver" waves each
// get the totals (LDS sums of the workgroup; through the mailbox when the level is split over the cluster),
// take the LM decision redundantly and each solve + exponentiate ONE candidate of the next pass -> barrier.
//
// The decision (round 6).  Every solver wave keeps its OWN copy of the LM state in LDS (SolverState: all of them take the same
// decision from the same totals, so the copies never differ, nobody waits for anybody and nothing is double-buffered); its
// conditions are wave-uniform scalars (UNI / readfirstlane), so the accept / reject ladder of optimizer.cpp:258-304 is scalar
// branches over a dozen values instead of exec-masked code that copies two 19-word states and a pose through every arm; what a
// pass hands to the next one (candidate poses, the single pose of a refill / level entry) is written to LDS where it is produced.
template <bool ONE>
__global__ void __launch_bounds__(TRACK_THREADS) TRACK_OCC k_track(const PairDesc one, const PairDesc* __restrict__ descs,
                                                                   TrackParams prm, revo_pair_result* __restrict__ out,
                                                                   EvalOut* __restrict__ eval_out, u64* __restrict__ mail,
                                                                   int n_pairs, int cluster, unsigned epoch_base,
                                                                   unsigned* seq_ptr, unsigned seq_val, float* prof_out,
                                                                   unsigned* resident) {
  __shared__ Cand s_cand[2][KMAX];
  __shared__ PassCtl s_pass[2];
  __shared__ SolverState s_sv[KMAX];
  __shared__ double s_part[NWAVES][NSUM];
  __shared__ int s_evals[REVO_L];  // residual evaluations per level, in the reference's count (wave 0 / lane 0 only)
#ifdef REVO_TRACK_PROFILE
  // [12 + l]: cycles spent in level l, [12 + L + l]: its passes, [12 + 2L + l]: evaluation, [12 + 3L + l]: barrier + sums +
  // exchange, [12 + 4L + l]: decision (+ closing barrier) of level l
  __shared__ long long s_prof[12 + 5 * REVO_L];
#define PROF_MARK(var) const long long var = clock64()
#else
#define PROF_MARK(var)
#endif
  // XCD-affine mapping: all members of a pair share blockIdx % 8 (speed only, never correctness)
  const int b = blockIdx.x;
  const int pair = (b / (8 * cluster)) * 8 + (b % 8);
  const int member = (b / 8) % cluster;
  // Residency census (revo_host.hip, "resident gate"): a workgroup that has started holds its CU until it exits, so once
  // the count reaches the grid size every cluster of this launch is complete and a LATER tracker grid may start filling
  // the CUs this one frees -- it can no longer keep members of this one off the chip.
  if (resident && threadIdx.x == 0) __hip_atomic_fetch_add(resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (pair >= n_pairs) return;
  const PairDesc& d = ONE ? one : descs[pair];
  u64* mail_pair = mail + (size_t)pair * 2 * cluster * NVAL;
  const int tid = threadIdx.x, lane = tid & 63, lane_inv = lane, wave = rfl_i(tid >> 6);
  int kmax = 1;  // solver waves: the deepest speculation of any level
#pragma unroll
  for (int i = 0; i < REVO_L; ++i) kmax = prm.kspec[i] > kmax ? prm.kspec[i] : kmax;
  kmax = kmax > KMAX ? KMAX : kmax;
  unsigned epoch = epoch_base;
  int cur_level = -1;
  f4v preg[TRACK_MAXP > 0 ? TRACK_MAXP : 1];
#pragma unroll
  for (int k = 0; k < TRACK_MAXP; ++k) preg[k] = f4v{0.f, 0.f, 1.f, 1.f};
  int Nl[REVO_L];  // point counts of the levels: read once (they sit in HBM behind a pointer)
#pragma unroll
  for (int i = 0; i < REVO_L; ++i) Nl[i] = d.npts[i];
  float abv = 0.0f;  // solver waves: lane k < 27 keeps entry k of the accepted A/n (21) and (sum w r v)/n (6)

  if (wave < kmax) {  // ---- pass 0: every solver wave sets up its own copy of the LM state
    float R0[9], T0[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R0[i] = d.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) T0[i] = d.T[i];
    PassCtl pc{MODE_EVAL, prm.lvl_begin, 1, PH_FIRST};
    int flags = 0;
    float q0[4];
    quat_from_R(R0, q0);
    if (prm.eval_only) {
      pc.phase = PH_EVAL_ONLY;
    } else if (prm.check_init) {  // tracker.cpp:314 runs BEFORE Sophus::SE3f(R,T) (optimizer.cpp:241)
      pc.mode = MODE_COST; pc.level = prm.pyr_min_lvl; pc.ncand = 2;
    } else if (!UNI(is_orthogonal(R0))) {  // Sophus::SE3f(R,T) would abort
      flags = 2;
      pc.mode = MODE_DONE;
    }
    if (lane == 0) {
      SolverState& s = s_sv[wave];
#pragma unroll
      for (int i = 0; i < 4; ++i) s.q[i] = q0[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) s.t[i] = T0[i];
      s.lastErr = __builtin_nanf("");
      s.lambda = 0.f;
      s.iteration = 0; s.incTry = 0; s.flags = flags; s.total_evals = 0;
      s.good = 0; s.bad = 0;
#pragma unroll
      for (int i = 0; i < KMAX; ++i) s.lamc[i] = 0.f;
      if (wave == 0) {
#pragma unroll
        for (int i = 0; i < REVO_L; ++i) s_evals[i] = 0;
        Cand& c0 = s_cand[0][0];
        Cand& c1 = s_cand[0][1];
        if (pc.mode == MODE_COST) {  // candidate 0 = identity, candidate 1 = the given initialisation (tracker.cpp:272-273)
#pragma unroll
          for (int i = 0; i < 9; ++i) { c0.R[i] = (i % 4 == 0) ? 1.0f : 0.0f; c1.R[i] = R0[i]; }
#pragma unroll
          for (int i = 0; i < 3; ++i) { c0.T[i] = 0.0f; c1.T[i] = T0[i]; }
        } else {
#pragma unroll
          for (int i = 0; i < 9; ++i) c0.R[i] = R0[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) c0.T[i] = T0[i];
        }
        s_pass[0] = pc;
#ifdef REVO_TRACK_PROFILE
        for (int i = 0; i < 12 + 5 * REVO_L; ++i) s_prof[i] = 0;
#endif
      }
    }
  }
  __syncthreads();

  int p = 0;
  for (;; ++p) {
    const int pb = p & 1, nb = pb ^ 1;
    const int mode = rfl_i(s_pass[pb].mode), l = rfl_i(s_pass[pb].level), ncand = rfl_i(s_pass[pb].ncand), phase = rfl_i(s_pass[pb].phase);
    if (mode == MODE_DONE) break;
    Cam cam;
    cam.fx = prm.cam[l].fx; cam.fy = prm.cam[l].fy; cam.cx = prm.cam[l].cx; cam.cy = prm.cam[l].cy;
    cam.w = prm.cam[l].w; cam.h = prm.cam[l].h;
    cam.wlim = (float)(cam.w - 2); cam.hlim = (float)(cam.h - 2);
    gf4p pts = (gf4p)d.pts[l];
    gf32p dtm = (gf32p)d.dt[l];
    int N = Nl[0];
#pragma unroll
    for (int i = 1; i < REVO_L; ++i) N = (l == i) ? Nl[i] : N;
    // few points: every member evaluates the whole level itself (same threads, same order, same bits),
    // so the pass needs no exchange
    const bool redundant = cluster == 1 || N <= prm.redundant_n;
    const int stride = redundant ? TRACK_THREADS : cluster * TRACK_THREADS;
    const int first = redundant ? tid : member * TRACK_THREADS + tid;
    // the exchange counter only moves on passes that exchange: the two parity slots of the mailbox are safe only if every
    // epoch is both published and gathered by every member (a redundant pass in between would let a fast member publish
    // epoch e + 2 into the slot a slow member still polls for e); all members compute `redundant` alike, so they stay in step
    if (!redundant) ++epoch;
#ifdef REVO_TRACK_PROFILE
    const long long tp0 = clock64();
#endif
    if (l != cur_level) {
      // Level entry (block-uniform): this thread's first points go to registers for every evaluation
      // of the level -- only the pose changes between them (optimizer.cpp:250-305).
      cur_level = l;
#pragma unroll
      for (int k = 0; k < TRACK_MAXP; ++k) {
        const int i = first + k * stride;
        preg[k] = (i < N) ? pts[i] : f4v{0.f, 0.f, 1.f, 1.f};
      }
    }
    const float edist = prm.edge_distance[l];
    const bool filt = prm.use_edge_filter != 0;
    // per thread: 27 normal-equation sums + the candidates' good counts (float, slots CSLOT..), and per candidate
    // sum w r^2 / sum r^2 in double
    float acc[32];
    double ed[8];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) ed[k] = 0.0;
    if (mode == MODE_COST) {
      // TrackerNew::evalCostFunction, tracker.cpp:357-393: nearest-pixel DT lookup, both poses in one pass
      float R[2][9], T[2][3];
      load_pose(s_cand[pb][0], R[0], T[0]);
      load_pose(s_cand[pb][1], R[1], T[1]);
#pragma unroll
      for (int k = 0; k < TRACK_MAXP; ++k)
        if (first + k * stride < N) {
          ed[0] += (double)cost_point(preg[k], dtm, R[0], T[0], cam, edist, filt);
          ed[2] += (double)cost_point(preg[k], dtm, R[1], T[1], cam, edist, filt);
        }
      for (int i = first + TRACK_MAXP * stride; i < N; i += stride) {
        const f4v pt = pts[i];
        ed[0] += (double)cost_point(pt, dtm, R[0], T[0], cam, edist, filt);
        ed[2] += (double)cost_point(pt, dtm, R[1], T[1], cam, edist, filt);
      }
    } else {
      const float huber = prm.huber_edge;
      if (ncand == 4) fused_block<3>(s_cand[pb], preg, pts, first, stride, N, dtm, cam, edist, filt, huber, acc, ed);
      else if (ncand == 3) fused_block<2>(s_cand[pb], preg, pts, first, stride, N, dtm, cam, edist, filt, huber, acc, ed);
      else if (ncand == 2) fused_block<1>(s_cand[pb], preg, pts, first, stride, N, dtm, cam, edist, filt, huber, acc, ed);
      else {  // a single candidate in full: residual + Jacobian + normal equations
        float R[9], T[3];
        load_pose(s_cand[pb][0], R, T);
#pragma unroll
        for (int k = 0; k < TRACK_MAXP; ++k)
          if (first + k * stride < N) full_point(preg[k], dtm, R, T, cam, edist, filt, huber, acc, ed);
        for (int i = first + TRACK_MAXP * stride; i < N; i += stride) full_point(pts[i], dtm, R, T, cam, edist, filt, huber, acc, ed);
      }
    }
    PROF_MARK(te1);
    {
      int lane_r = lane;  // (opaque per pass, like the solver section's: the butterflies' `lane & MASK` predicates are nine SGPR pairs)
      asm volatile("" : "+v"(lane_r));
      reduce32(acc, lane_r);
      reduce8d(ed, lane_r);
      if (lane_r < 32) s_part[wave][idx32(lane_r)] = (double)acc[0];
      if (lane_r < 8) s_part[wave][32 + idx8(lane_r)] = ed[0];
    }
#ifdef REVO_TRACK_PROFILE
    const long long tp1 = clock64();
    if (tid == 0) { s_prof[6] += te1 - tp0; s_prof[7] += tp1 - te1; }
#endif
    __syncthreads();
#ifdef REVO_TRACK_PROFILE
    const long long tp2 = clock64();
    long long tp3 = tp2;
#endif

    if (wave < kmax) {  // ---- the solver waves: totals, the LM decision (redundantly), one candidate each
      // The lane id, re-issued opaquely per pass: every `lane == i` / `row > k` predicate below is loop-invariant, IR-level LICM
      // hoists them all in front of the pass loop, and ~40 exec-mask SGPR pairs kept alive across the evaluation come back as
      // v_readlane pairs from spill registers on the serial path.  Derived from an opaque value they are one v_cmp where used.
      int lane = lane_inv;
      asm volatile("" : "+v"(lane));
      // lane v < 32: total of float slot v; lanes 32 + 2k, 33 + 2k: total of double slot k (both lanes of the pair)
      const int ri = lane < 32 ? lane : (lane < NVAL ? 32 + ((lane - 32) >> 1) : 0);
      double tot = 0.0;
      bool xok = true;
      if (redundant || wave == 0) {
#pragma unroll
        for (int wv = 0; wv < NWAVES; ++wv) tot += s_part[wv][ri];
      }
      if (!redundant) {
        if (wave == 0) cluster_publish(mail_pair, cluster, member, epoch, tot, lane);
        xok = cluster_gather(mail_pair, cluster, epoch, lane, &tot);
      }
      const float tf = (float)tot;  // the reference's float accumulators, read once (optimizer.cpp:190, LGSX.h:320-326)
#define TOT(v) rl(tf, (v))
#define TOT_SW(j) rl(tf, 32 + 2 * DSLOT(j))   // sum w r^2 of candidate j
#define TOT_N(j) rl(tf, CSLOT + (j))     // its good count
#ifdef REVO_TRACK_PROFILE
      tp3 = clock64();
#endif
      SolverState& sv = s_sv[wave];
      float lastErr = sv.lastErr, lambda = sv.lambda;
      int iteration = rfl_i(sv.iteration), incTry = rfl_i(sv.incTry), total = rfl_i(sv.total_evals), flags = rfl_i(sv.flags);
      int nmode = MODE_EVAL, nlevel = l, nncand = 1, nphase = PH_LM;
      bool gen = false;
      int consumed = 0;  // residual evaluations of the reference's sequence decided in this pass
      float q[4], t[3];  // the accepted pose after this decision (Sophus::SE3f referenceToFrame)

      if (!UNI(xok)) {  // a cluster member never showed up: give up loudly instead of hanging
        flags |= 8;
        nmode = MODE_DONE;
      } else if (mode == MODE_COST) {  // tracker.cpp:272-282
        const float costEye = TOT(32), costInit = TOT(36);  // double slots 0 and 2
        float Rs[9], Ts[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rs[i] = d.R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) Ts[i] = d.T[i];
        if (UNI(costEye < costInit)) {
#pragma unroll
          for (int i = 0; i < 9; ++i) Rs[i] = (i % 4 == 0) ? 1.0f : 0.0f;
          Ts[0] = Ts[1] = Ts[2] = 0.0f;
          flags |= 1;
        }
        quat_from_R(Rs, q);
        nlevel = prm.lvl_begin; nphase = PH_FIRST;
        if (!UNI(is_orthogonal(Rs))) { flags |= 2; nmode = MODE_DONE; }  // Sophus::SE3f(R,T), optimizer.cpp:241
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) sv.q[i] = q[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) sv.t[i] = Ts[i];
          if (wave == 0) {
            Cand& c = s_cand[nb][0];
#pragma unroll
            for (int i = 0; i < 9; ++i) c.R[i] = Rs[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) c.T[i] = Ts[i];
          }
        }
      } else if (phase == PH_EVAL_ONLY) {
        const float n_f = TOT_N(0);
        const float an = __fdiv_rn(tf, n_f);  // LGS6 after finish(): A/n, b = -(sum w r v)/n, error = sum w r^2 / n
        if (wave == 0 && member == 0) {
          EvalOut& eo = eval_out[pair];
          if (lane < 21) {
            int r = 0, k = lane;
            while (k >= 6 - r) { k -= 6 - r; ++r; }
            eo.A[r * 6 + r + k] = an;
            eo.A[(r + k) * 6 + r] = an;
          } else if (lane < 27) {
            eo.b[lane - 21] = -an;
          } else if (lane == 32) {
            eo.error = an; eo.mean_err = an; eo.sum_w = tf;
          } else if (lane == 34) {
            eo.sum_u = tf;
          } else if (lane == CSLOT) {
            eo.good = (int)tf; eo.bad = N - (int)tf;
          }
        }
        nmode = MODE_DONE;
      } else {
        bool take_ab = false, level_done = false;
        float n_ab = 1.0f;  // point count of the evaluation the new normal equations come from
        int acc_j = -1;     // the accepted candidate of this pass
        const int max_its = prm.max_its[l];
        if (phase == PH_FIRST) {  // optimizer.cpp:243-250
          const float n_f = TOT_N(0);
          lastErr = __fdiv_rn(TOT_SW(0), n_f);  // optimizer.cpp:190
          consumed = 1;
          lambda = prm.lambda_initial[l];
          iteration = 0; incTry = 0;
          take_ab = true; n_ab = n_f;
          if (0 < max_its) gen = true; else level_done = true;
        } else if (phase == PH_REFILL) {  // the normal equations at the pose an error-only candidate was accepted at
          take_ab = true; n_ab = TOT_N(0);
          gen = true;
        } else {  // PH_LM: consume the candidates in the reference's order, optimizer.cpp:258-304
          int end_j = -1;  // the level ends at this (rejected) candidate: step too small, or the evaluation cap
          bool capped = false;
          float errA = 0.0f;
          const float smin = prm.step_size_min[l];
#pragma unroll
          for (int j = 0; j < KMAX; ++j) {
            if (j < ncand && acc_j < 0 && end_j < 0) {
              const float err = __fdiv_rn(TOT_SW(j), TOT_N(j));
              if (UNI(err < lastErr)) { acc_j = j; errA = err; }                   // accepted
              else if (!UNI(s_cand[pb][j].incsq > smin)) end_j = j;                // rejected, and the step is too small to go on
              else if (total + j + 1 > MAX_TOTAL_EVALS) { end_j = j; capped = true; }
            }
          }
          consumed = acc_j >= 0 ? acc_j + 1 : (end_j >= 0 ? end_j + 1 : ncand);
          if (acc_j >= 0) {
            const float lamA = sv.lamc[acc_j];  // the damping this candidate was solved with
            if (UNI(__fdiv_rn(errA, lastErr) > prm.convergence_eps[l])) iteration = max_its;
            lastErr = errA;
            lambda = (lamA <= 0.2f) ? 0.0f : lamA * prm.lambda_success_fac;
            iteration += 1;
            incTry = 0;
            if (iteration < max_its) {
              if (acc_j == 0) { take_ab = true; n_ab = TOT_N(0); gen = true; }
              else {  // its Jacobian was not evaluated: one full pass at that pose
                nphase = PH_REFILL;
                if (wave == 0 && lane < 12) (&s_cand[nb][0].R[0])[lane] = (&s_cand[pb][acc_j].R[0])[lane];  // R[9], T[3]
              }
            } else {
              level_done = true;
            }
          } else if (end_j >= 0) {
            level_done = true;
            if (capped) flags |= 4;
          } else {  // every candidate of the pass was rejected: the chain goes on with the next dampings
            const int tr_last = incTry + ncand;  // incTry of the last candidate (optimizer.cpp:263)
            lambda = lambda_after_reject(sv.lamc[ncand - 1], prm.lambda_fail_fac, tr_last);
            incTry = tr_last;
            gen = true;
          }
        }
        total += consumed;
        // the accepted pose: the accepted candidate's, otherwise unchanged
        {
          const float* qs = acc_j >= 0 ? s_cand[pb][acc_j].q : sv.q;
          const float* ts = acc_j >= 0 ? s_cand[pb][acc_j].T : sv.t;
#pragma unroll
          for (int i = 0; i < 4; ++i) q[i] = qs[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) t[i] = ts[i];
        }
        if (take_ab) abv = __fdiv_rn(tf, n_ab);  // LGS6::finish (LGSX.h:320-326): A/n and (sum w r v)/n, lane k = entry k
        if (level_done) {  // optimizer.cpp:308-309 -> next level or done (tracker.cpp:324-340)
          float Rs[9];
          quat_to_R(q, Rs);
          if (wave == 0 && lane == 0) {
            Cand& c = s_cand[nb][0];
#pragma unroll
            for (int i = 0; i < 9; ++i) c.R[i] = Rs[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) c.T[i] = t[i];
          }
          gen = false;
          if (l > prm.lvl_end && !(flags & 4)) {
            nlevel = l - 1; nphase = PH_FIRST;
            quat_from_R(Rs, q);  // Sophus::SE3f(R,T) of the next level
          } else {
            nmode = MODE_DONE;
          }
        }
        if (lane == 0) {
          if (acc_j >= 0 || level_done) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sv.q[i] = q[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) sv.t[i] = t[i];
          }
          if (consumed) {  // ResidualInfo of the last evaluation the reference's sequence has seen
            const int g = (int)rl(tf, CSLOT - 1 + consumed);
            sv.good = g; sv.bad = N - g;
          }
        }
      }
      const int kspec = prm.kspec[l] < 1 ? 1 : (prm.kspec[l] > KMAX ? KMAX : prm.kspec[l]);  // gen stays on level l
      PROF_MARK(td1);
      if (gen) {  // the candidates of the next pass: optimizer.cpp:258-269 at the dampings the retries would see
        float lamc[KMAX];
        lamc[0] = lambda;
#pragma unroll
        for (int j = 1; j < KMAX; ++j) lamc[j] = lambda_after_reject(lamc[j - 1], prm.lambda_fail_fac, incTry + j);  // as if 0..j-1 had been rejected
        if (lane == 0) {
#pragma unroll
          for (int j = 0; j < KMAX; ++j) sv.lamc[j] = lamc[j];
        }
        if (wave < kspec) {
          float lam = lamc[0];
#pragma unroll
          for (int j = 1; j < KMAX; ++j) lam = (wave == j) ? lamc[j] : lam;
          float inc[6], qn[4], tn[3], Rn[9];
          solve6_ldlt(abv, lam, inc, lane);
          PROF_MARK(td2);
#ifdef REVO_TRACK_PROFILE
          if (tid == 0) { s_prof[8] += td1 - tp3; s_prof[9] += td2 - td1; s_prof[10] += 1; }
#endif
          const float incsq = inc[0] * inc[0] + inc[1] * inc[1] + inc[2] * inc[2] + inc[3] * inc[3] + inc[4] * inc[4] + inc[5] * inc[5];
          se3_exp_mul(inc, q, t, qn, tn);
          quat_to_R(qn, Rn);
          if (lane == 0) {
            Cand& c = s_cand[nb][wave];
#pragma unroll
            for (int i = 0; i < 9; ++i) c.R[i] = Rn[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) c.T[i] = tn[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) c.q[i] = qn[i];
            c.incsq = incsq;
          }
        }
        nncand = kspec; nphase = PH_LM;
      }
      if (lane == 0) {
        sv.lastErr = lastErr; sv.lambda = lambda;
        sv.iteration = iteration; sv.incTry = incTry; sv.total_evals = total; sv.flags = flags;
        if (wave == 0) {
          s_pass[nb] = PassCtl{nmode, nlevel, nncand, nphase};
          s_evals[l] += consumed;
        }
      }
#undef TOT
#undef TOT_SW
#undef TOT_N
    }
#ifdef REVO_TRACK_PROFILE
    const long long tp5 = clock64();
#endif
    __syncthreads();
#ifdef REVO_TRACK_PROFILE
    if (tid == 0) {
      const long long tp6 = clock64();
      s_prof[0] += tp1 - tp0; s_prof[1] += tp2 - tp1; s_prof[2] += tp3 - tp2; s_prof[3] += 0; s_prof[4] += tp5 - tp3;
      s_prof[5] += tp6 - tp5;
      s_prof[12 + l] += tp6 - tp0; s_prof[12 + REVO_L + l] += 1;
      s_prof[12 + 2 * REVO_L + l] += tp1 - tp0; s_prof[12 + 3 * REVO_L + l] += tp3 - tp1; s_prof[12 + 4 * REVO_L + l] += tp6 - tp3;
    }
#endif
  }

  if (tid == 0 && member == 0 && !prm.eval_only) {
    const SolverState& s = s_sv[0];
    const Cand& c = s_cand[p & 1][0];
    revo_pair_result r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.R[i] = c.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) r.T[i] = c.T[i];
    if (s.flags & (2 | 8)) {  // no pose was produced: hand the input back, the flags say why
#pragma unroll
      for (int i = 0; i < 9; ++i) r.R[i] = d.R[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) r.T[i] = d.T[i];
    }
    r.err = s.lastErr;  // last_residual (optimizer.cpp:276): always assigned together with lastErr
    r.good = s.good;
    r.bad = s.bad;
    // tracker.cpp:351-352
    r.status = ((double)s.good / (double)s.bad < 4.0) ? REVO_TRACKER_STATE_NEW_KF : REVO_TRACKER_STATE_OK;
#pragma unroll
    for (int i = 0; i < REVO_L; ++i) r.evals[i] = s_evals[i];
#ifdef REVO_TRACK_PROFILE
    for (int i = 0; i < 5; ++i) r.evals[i] = (int)(s_prof[i] / 16);  // profile build: phase cycles / 16 ...
    r.evals[5] = p;                                                   // ... and the number of passes
    if (eval_out) {  // the finer split travels in the (otherwise unused) EvalOut record
      EvalOut& eo = eval_out[pair];
      for (int i = 0; i < 12; ++i) eo.A[i] = (float)s_prof[i];
      eo.A[12] = (float)p;
      for (int i = 0; i < 2 * REVO_L; ++i) eo.A[13 + i] = (float)s_prof[12 + i];
    } else {  // batches have no EvalOut: the per-level split replaces the pose of the record (profile builds only)
      for (int i = 0; i < 4; ++i) { r.R[i] = (float)s_prof[12 + i]; r.R[4 + i] = (float)s_prof[12 + REVO_L + i]; }
      r.R[8] = (float)s_prof[6]; r.T[0] = (float)s_prof[7]; r.T[1] = (float)s_prof[8]; r.T[2] = (float)s_prof[9];
      r.err = (float)s_prof[10];
    }
    if (prof_out) {  // the whole table, per pair (revo_debug_batch_profile_)
      for (int i = 0; i < 12 + 5 * REVO_L; ++i) prof_out[(size_t)pair * 64 + i] = (float)s_prof[i];
      prof_out[(size_t)pair * 64 + 63] = (float)p;
    }
#endif
    r.flags = s.flags;
    r.n_pts0 = d.npts[0];
    out[pair] = r;
    if (ONE && seq_ptr) {  // the host polls this word in pinned memory instead of waiting for the stream to drain
      __threadfence_system();
      *(volatile unsigned*)seq_ptr = seq_val;
    }
  }
}

// The resident gate: one wave that waits (bounded, sleeping) until the census counter of the tracker launches has
// reached `want`, i.e. until every workgroup of the previous tracker grid of this device has started.
// resident[1] counts the gates that gave up (bounded wait, ~2 s): the next grid then starts anyway and the in-kernel bounded
// spin + flag 8 are what is left -- the counter makes that fall-through observable (revo_debug_gate_timeouts_).
__global__ void __launch_bounds__(64) k_track_gate(unsigned* __restrict__ resident, unsigned want) {
  if (threadIdx.x != 0) return;
  for (unsigned spins = 0; spins < 4u * SPIN_LIMIT; ++spins) {
    const unsigned have = __hip_atomic_load(resident, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(have - want) >= 0) return;
    __builtin_amdgcn_s_sleep(8);
  }
  __hip_atomic_fetch_add(resident + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A.ldlt().solve(b) alone (optimizer.cpp:258-262), for the parity tests of the solver
__global__ void __launch_bounds__(64) k_solve6(const float* __restrict__ Ab, int n, float* __restrict__ x_out) {
  const int lane = threadIdx.x;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const float* src = Ab + (size_t)i * 43;  // A (36, symmetric), b (6), lambda (1)
    float abv = 0.0f;
    if (lane < 21) {
      int r = 0, k = lane;
      while (k >= 6 - r) { k -= 6 - r; ++r; }
      abv = src[r * 6 + r + k];
    } else if (lane < 27) {
      abv = src[36 + lane - 21];
    }
    float x[6];
    solve6_ldlt(abv, src[42], x, lane);
    if (lane == 0)
      for (int k = 0; k < 6; ++k) x_out[(size_t)i * 6 + k] = x[k];
  }
}

}  // namespace

void launch_solve6(const float* d_Ab, int n, float* d_x, hipStream_t s) {
  hipLaunchKernelGGL(k_solve6, dim3(n < 256 ? n : 256), dim3(64), 0, s, d_Ab, n, d_x);
}

int track_blocks_per_cu() {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_track<false>, TRACK_THREADS, 0) != hipSuccess) nb = 1;
  return nb < 1 ? 1 : nb;
}

// Mailbox granules carry {epoch, value}; a launch uses at most MAX_TOTAL_EVALS + a few epochs, so every
// launch gets a fresh window of TRACK_EPOCH_WINDOW epochs and stale granules of earlier launches can
// never match.  The mailbox is zeroed when it is allocated and when the 32-bit counter would wrap.
#define TRACK_EPOCH_WINDOW 8192u
#ifdef REVO_TRACK_PROFILE
static float* g_prof_dev = nullptr;
static int g_prof_cap = 0;
#endif
static unsigned next_epoch_base(unsigned* epoch_io, unsigned long long* d_mail, size_t mail_bytes, hipStream_t s) {
  if (*epoch_io > 0xffffffffu - 2 * TRACK_EPOCH_WINDOW) {
    hipMemsetAsync(d_mail, 0, mail_bytes, s);
    *epoch_io = 0;
  }
  const unsigned base = *epoch_io;
  *epoch_io += TRACK_EPOCH_WINDOW;
  return base;
}

void launch_track_gate(unsigned* d_resident, unsigned want, hipStream_t s) {
  hipLaunchKernelGGL(k_track_gate, dim3(1), dim3(64), 0, s, d_resident, want);
}

int launch_track(const PairDesc* d_descs, const TrackParams& prm, revo_pair_result* d_out, EvalOut* d_eval, int n_pairs,
                 unsigned long long* d_mail, unsigned* epoch_io, int cluster, unsigned* d_resident, hipStream_t s) {
  static_assert(MAX_TOTAL_EVALS + 64 < TRACK_EPOCH_WINDOW, "epoch window too small");
  const unsigned base = next_epoch_base(epoch_io, d_mail, sizeof(unsigned long long) * (size_t)n_pairs * 2 * cluster * NVAL, s);
  const int groups = (n_pairs + 7) / 8;
  float* prof = nullptr;
#ifdef REVO_TRACK_PROFILE
  if (g_prof_cap < n_pairs) {
    if (g_prof_dev) (void)hipFree(g_prof_dev);
    g_prof_cap = 0;
    if (hipMalloc((void**)&g_prof_dev, sizeof(float) * 64 * (size_t)n_pairs) == hipSuccess) g_prof_cap = n_pairs;
  }
  prof = g_prof_cap >= n_pairs ? g_prof_dev : nullptr;
#endif
  hipLaunchKernelGGL(k_track<false>, dim3(groups * 8 * cluster), dim3(TRACK_THREADS), 0, s, PairDesc{}, d_descs, prm, d_out,
                     d_eval, (u64*)d_mail, n_pairs, cluster, base, (unsigned*)nullptr, 0u, prof, d_resident);
  return groups * 8 * cluster;
}
// profile builds: per pair 64 floats of cycle counters of the last batch launch (layout: k_track's s_prof, [63] = passes)
extern "C" int revo_debug_batch_profile_(float* out, int n_pairs) {
#ifdef REVO_TRACK_PROFILE
  if (!out || n_pairs > g_prof_cap) return REVO_ERR_INVALID_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return REVO_ERR_HIP;
  return hipMemcpy(out, g_prof_dev, sizeof(float) * 64 * (size_t)n_pairs, hipMemcpyDeviceToHost) == hipSuccess ? REVO_OK : REVO_ERR_HIP;
#else
  (void)out; (void)n_pairs;
  return REVO_ERR_INVALID_ARG;
#endif
}

// one pair, descriptor by value; out / eval_out may be device-visible pinned host memory
int launch_track_one(const PairDesc& desc, const TrackParams& prm, revo_pair_result* out, EvalOut* eval_out,
                     unsigned long long* d_mail, unsigned* epoch_io, int cluster, unsigned* seq_ptr, unsigned seq_val,
                     unsigned* d_resident, hipStream_t s) {
  const unsigned base = next_epoch_base(epoch_io, d_mail, sizeof(unsigned long long) * 2 * (size_t)cluster * NVAL, s);
  hipLaunchKernelGGL(k_track<true>, dim3(8 * cluster), dim3(TRACK_THREADS), 0, s, desc, (const PairDesc*)nullptr, prm, out,
                     eval_out, (u64*)d_mail, 1, cluster, base, seq_ptr, seq_val, (float*)nullptr, d_resident);
  return 8 * cluster;
}
