// Which cheap sequences on top of v_rcp_f32 / v_rsq_f32 give the CORRECTLY ROUNDED 1/x and 1/sqrt(x) for every
// binary32 input?  (A correctly rounded result is what a CPU oracle reproduces with plain IEEE arithmetic.)
// Exhaustive over all positive normal floats; reference = IEEE division / double-precision sqrt+divide on the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ float rcp_a(float x) {  // one Newton step on the hardware seed
  float y = __builtin_amdgcn_rcpf(x);
  float e = fmaf(-x, y, 1.0f);
  return fmaf(y, e, y);
}
__device__ float rcp_b(float x) {  // two steps
  float y = __builtin_amdgcn_rcpf(x);
  float e = fmaf(-x, y, 1.0f);
  y = fmaf(y, e, y);
  e = fmaf(-x, y, 1.0f);
  return fmaf(y, e, y);
}
__device__ float rsq_a(float x) {  // one Newton step: y + y*(0.5*(1 - x*y*y))
  float y = __builtin_amdgcn_rsqf(x);
  float e = fmaf(-(x * y), y, 1.0f);
  return fmaf(0.5f * y, e, y);
}
__device__ float rsq_b(float x) {  // residual computed exactly-ish: h = 0.5*y, e = 0.5 - x*y*h via two fma
  float y = __builtin_amdgcn_rsqf(x);
  float g = x * y;            // ~sqrt(x)
  float h = 0.5f * y;
  float r = fmaf(-g, h, 0.5f);
  return fmaf(y, r, y);
}
__device__ float rsq_c(float x) {  // two steps of rsq_b
  float y = __builtin_amdgcn_rsqf(x);
  float g = x * y, h = 0.5f * y;
  float r = fmaf(-g, h, 0.5f);
  y = fmaf(y, r, y);
  g = x * y; h = 0.5f * y;
  r = fmaf(-g, h, 0.5f);
  return fmaf(y, r, y);
}

__global__ void k(unsigned long long *bad, uint32_t lo, uint32_t hi) {
  uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long b[6] = {0, 0, 0, 0, 0, 0};
  for (uint64_t u = (uint64_t)lo + blockIdx.x * blockDim.x + threadIdx.x; u < hi; u += stride) {
    float x = __uint_as_float((uint32_t)u);
    float want_rcp = 1.0f / x;
    float want_rsq = (float)(1.0 / sqrt((double)x));
    float seed = __builtin_amdgcn_rcpf(x);
    b[0] += __float_as_uint(seed) != __float_as_uint(want_rcp);
    b[1] += __float_as_uint(rcp_a(x)) != __float_as_uint(want_rcp);
    b[2] += __float_as_uint(rcp_b(x)) != __float_as_uint(want_rcp);
    b[3] += __float_as_uint(rsq_a(x)) != __float_as_uint(want_rsq);
    b[4] += __float_as_uint(rsq_b(x)) != __float_as_uint(want_rsq);
    b[5] += __float_as_uint(rsq_c(x)) != __float_as_uint(want_rsq);
  }
  for (int i = 0; i < 6; ++i)
    if (b[i]) atomicAdd(&bad[i], b[i]);
}

int main() {
  unsigned long long *d, h[6];
  hipMalloc(&d, sizeof h);
  // all positive normal floats whose reciprocal is normal too: [2^-126, 2^126]
  const uint32_t lo = 0x00800000u, hi = 0x7E800000u;
  hipMemset(d, 0, sizeof h);
  k<<<4096, 256>>>(d, lo, hi);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char *names[6] = {"v_rcp_f32 alone", "rcp + 1 step", "rcp + 2 steps", "rsq + 1 step", "rsq + 1 step (fma residual)", "rsq + 2 steps"};
  printf("inputs %llu\n", (unsigned long long)(hi - lo));
  for (int i = 0; i < 6; ++i) printf("%-32s not correctly rounded: %llu\n", names[i], h[i]);
  return 0;
}
