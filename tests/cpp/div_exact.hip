// revo_div.h vs __fdiv_rn, bit for bit, on the operand ranges of the tracker (and a few beyond): prints the number of
// mismatching pairs.  Built and run by tests/test_gpu_tracker2.py on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../revo_amd/csrc/revo_div.h"

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float unit(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }  // [0, 1)

__global__ void k_check(unsigned long long* bad, unsigned long long* first, int mode) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t a = mix(i * 2u + 1u), b = mix(i * 2u + 0x9e3779b9u);
  float n, d;
  if (mode == 0) {         // projection: |X|, |Y| up to 12 m, z in [0.05, 12] m, either sign of the numerator
    n = (unit(a) * 2.0f - 1.0f) * 12.0f;
    d = 0.05f + unit(b) * 11.95f;
  } else if (mode == 1) {  // 1 / z and huber / residual: residual (a distance in pixels) in (0.3, 400]
    n = (i & 1u) ? 1.0f : 0.3f;
    d = (i & 1u) ? 0.05f + unit(b) * 11.95f : 0.3f + unit(b) * 400.0f;
  } else if (mode == 2) {  // err = sum / count and friends: wide positive range, random mantissas
    n = __uint_as_float(0x30000000u + (a % 0x20000000u));
    d = __uint_as_float(0x30000000u + (b % 0x20000000u));
  } else {                 // negative depths (points behind the camera) and tiny numerators
    n = (unit(a) - 0.5f) * 1e-3f;
    d = -(0.05f + unit(b) * 11.95f);
  }
  const float want = __fdiv_rn(n, d);
  const float got = revo_div(n, d);
  if (__float_as_uint(want) != __float_as_uint(got)) {
    if (atomicAdd(bad, 1ull) == 0) { first[0] = __float_as_uint(n); first[1] = __float_as_uint(d); }
  }
}

int main() {
  unsigned long long *bad, *first;
  if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&first, 16) != hipSuccess) { printf("no device\n"); return 2; }
  unsigned long long total = 0;
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(bad, 0, 8); hipMemset(first, 0, 16);
    hipLaunchKernelGGL(k_check, dim3(1 << 16), dim3(256), 0, 0, bad, first, mode);
    unsigned long long h = 0, f[2] = {0, 0};
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 16, hipMemcpyDeviceToHost);
    printf("mode %d: %llu mismatches of %d (first: n=%08llx d=%08llx)\n", mode, h, 1 << 24, f[0], f[1]);
    total += h;
  }
  printf("TOTAL_MISMATCHES %llu\n", total);
  return total ? 1 : 0;
}
