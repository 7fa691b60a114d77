// revo_div.h -- correctly rounded float division with a SHARED refined reciprocal.
//
// The compiler expands an IEEE float division (`__fdiv_rn`) into v_div_scale x2, v_rcp, five fma/mul, v_div_fmas and
// v_div_fixup: 11 instructions.  The tracker divides two numerators by the same z per projection (optimizer.cpp:96-97),
// needs 1/z again for the Jacobian (optimizer.cpp:213) and evaluates up to four poses per point: a quarter of the
// evaluation's instructions were divisions.  The sequence below is the SAME arithmetic (rcp, one Newton step on the
// reciprocal, quotient, two residual corrections) without the scaling and the fix-up, which only act when the operands
// or the quotient leave the normal range by more than 2^+-96 / become denormal or when an operand is 0, inf or NaN.
// For finite normal operands it returns the same bits as __fdiv_rn (tests/cpp/div_exact.hip checks 2^24 operand pairs
// of the tracker's ranges on the GPU); for z = 0 / NaN both give a non-finite value that fails the projection's
// bounds test (optimizer.cpp:100), which is all the tracker asks of such a point.
#pragma once

__device__ __forceinline__ float revo_recip_refined(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e0 = __builtin_fmaf(-d, r0, 1.0f);
  return __builtin_fmaf(e0, r0, r0);
}
// n / d, r1 = revo_recip_refined(d)
__device__ __forceinline__ float revo_div_with(float n, float d, float r1) {
  const float q0 = n * r1;
  const float e1 = __builtin_fmaf(-d, q0, n);
  const float q1 = __builtin_fmaf(e1, r1, q0);
  const float e2 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(e2, r1, q1);
}
__device__ __forceinline__ float revo_div(float n, float d) { return revo_div_with(n, d, revo_recip_refined(d)); }
