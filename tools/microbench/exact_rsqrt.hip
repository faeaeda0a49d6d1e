// Can 1/sqrt(x) := RN(1 / RN(sqrt(x)))  -- two correctly rounded IEEE operations, i.e. `1.0f / sqrtf(x)` on a CPU --
// be reached on gfx950 from v_sqrt_f32 / v_rcp_f32 with a few fma corrections, for EVERY normal input?
// Measured (all 2 130 706 432 positive normal floats): NO.  v_sqrt_f32 + one or two Heron corrections is not the
// correctly rounded sqrt for 1.9 M / 1.7 M inputs (the corrections stall half an ulp away near ties; the compiler's
// IEEE sequence adds a residual-sign fix-up for that), and every 1/sqrt built on it misses ~1.5 M inputs.  Unlike the
// reciprocal (tools/microbench/exact_rcp.hip: v_rcp_f32 + ONE Newton step is exact for all inputs), rsqrt therefore
// stays the integer-seed fixed sequence of the contract.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

// correctly rounded sqrt candidates
__device__ float sqrt_a(float x) {  // hardware sqrt + one Heron/Newton correction with a hardware reciprocal
  float s = __builtin_amdgcn_sqrtf(x);
  float r = fmaf(-s, s, x);               // exact residual (one rounding)
  float h = 0.5f * __builtin_amdgcn_rcpf(s);
  return fmaf(r, h, s);
}
// rsqrt candidate A: sqrt_a, then the exact-rcp recipe with its own seed
__device__ float rsqrt_a(float x) {
  float s = sqrt_a(x);
  float y = __builtin_amdgcn_rcpf(s);
  return fmaf(y, fmaf(-s, y, 1.0f), y);
}
// rsqrt candidate B: reuse the reciprocal of the uncorrected sqrt as the seed (one transcendental fewer)
__device__ float rsqrt_b(float x) {
  float s0 = __builtin_amdgcn_sqrtf(x);
  float y0 = __builtin_amdgcn_rcpf(s0);
  float r = fmaf(-s0, s0, x);
  float s = fmaf(r, 0.5f * y0, s0);
  return fmaf(y0, fmaf(-s, y0, 1.0f), y0);
}
// rsqrt candidate C: v_rsq_f32 as the seed for both (s0 = x * rsq)
__device__ float rsqrt_c(float x) {
  float y0 = __builtin_amdgcn_rsqf(x);
  float s0 = x * y0;
  float r = fmaf(-s0, s0, x);
  float s = fmaf(r, 0.5f * y0, s0);
  return fmaf(y0, fmaf(-s, y0, 1.0f), y0);
}

__device__ float sqrt_b(float x) {  // two corrections
  float s = __builtin_amdgcn_sqrtf(x);
  float h = 0.5f * __builtin_amdgcn_rcpf(s);
  s = fmaf(fmaf(-s, s, x), h, s);
  return fmaf(fmaf(-s, s, x), h, s);
}
__device__ float rsqrt_d(float x) {  // sqrt_b, then one Newton step on the reciprocal seeded with 1/sqrt0
  float s = __builtin_amdgcn_sqrtf(x);
  float y0 = __builtin_amdgcn_rcpf(s);
  float h = 0.5f * y0;
  s = fmaf(fmaf(-s, s, x), h, s);
  s = fmaf(fmaf(-s, s, x), h, s);
  return fmaf(y0, fmaf(-s, y0, 1.0f), y0);
}
__device__ float rsqrt_e(float x) {  // same with two reciprocal steps
  float s = __builtin_amdgcn_sqrtf(x);
  float y = __builtin_amdgcn_rcpf(s);
  float h = 0.5f * y;
  s = fmaf(fmaf(-s, s, x), h, s);
  s = fmaf(fmaf(-s, s, x), h, s);
  y = fmaf(y, fmaf(-s, y, 1.0f), y);
  return fmaf(y, fmaf(-s, y, 1.0f), y);
}

__global__ void k(unsigned long long *bad, uint32_t lo, uint32_t hi) {
  uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long b[5] = {0, 0, 0, 0, 0};
  for (uint64_t u = (uint64_t)lo + blockIdx.x * blockDim.x + threadIdx.x; u < hi; u += stride) {
    float x = __uint_as_float((uint32_t)u);
    float want_s = sqrtf(x);              // IEEE (hipcc expands to the correctly rounded sequence)
    float want = 1.0f / want_s;           // IEEE division
    b[0] += __float_as_uint(sqrt_b(x)) != __float_as_uint(want_s);
    b[1] += __float_as_uint(rsqrt_d(x)) != __float_as_uint(want);
    b[2] += __float_as_uint(rsqrt_e(x)) != __float_as_uint(want);
    b[3] += __float_as_uint(rsqrt_c(x)) != __float_as_uint(want);
    // sanity of the reference itself: sqrtf against double
    b[4] += __float_as_uint(want_s) != __float_as_uint((float)sqrt((double)x));
  }
  for (int i = 0; i < 5; ++i)
    if (b[i]) atomicAdd(&bad[i], b[i]);
}

int main() {
  unsigned long long *d, h[5];
  (void)hipMalloc(&d, sizeof h);
  (void)hipMemset(d, 0, sizeof h);
  (void)hipDeviceSynchronize();
  const uint32_t lo = 0x00800000u, hi = 0x7F800000u;  // every positive normal float
  k<<<4096, 256>>>(d, lo, hi);
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char *names[5] = {"sqrt: v_sqrt + 2 corrections", "rsqrt D (sqrt_b, 1 rcp step)", "rsqrt E (sqrt_b, 2 rcp steps)", "rsqrt C (seed = v_rsq)", "reference sqrtf != (float)sqrt(double)"};
  printf("inputs %llu\n", (unsigned long long)(hi - lo));
  for (int i = 0; i < 5; ++i) printf("%-40s mismatches: %llu\n", names[i], h[i]);
  return 0;
}
