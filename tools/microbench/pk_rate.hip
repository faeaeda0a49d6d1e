// Microbenchmark: issue rate of plain FP32 FMA vs packed FP32 FMA on gfx950 (decides whether shading
// two pixels per lane with v_pk_fma_f32 can pay).  Build: hipcc -O3 --offload-arch=gfx950 pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int CHAINS>
__global__ void k_scalar(float *out, float a, float b, int iters) {
  float x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) x[i] = __builtin_fmaf(x[i], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ void k_packed(float *out, float a, float b, int iters) {
  v2f x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = (v2f)(threadIdx.x * 1e-3f + i);
  const v2f va = (v2f)(a), vb = (v2f)(b);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) x[i] = __builtin_elementwise_fma(x[i], va, vb);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_it(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float *out;
  const int blocks = 256 * 8, threads = 256, iters = 4096;
  hipMalloc(&out, blocks * threads * sizeof(float));
  constexpr int C = 16;
  for (int waves_per_simd : {1, 2, 4, 8}) {
    int nb = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x k
    float ms_s = time_it([&] { hipLaunchKernelGGL((k_scalar<C>), dim3(nb), dim3(threads), 0, 0, out, 1.0001f, 0.5f, iters); });
    float ms_p = time_it([&] { hipLaunchKernelGGL((k_packed<C>), dim3(nb), dim3(threads), 0, 0, out, 1.0001f, 0.5f, iters); });
    double fma_s = (double)nb * threads * C * iters, fma_p = fma_s * 2;
    printf("waves/SIMD %d: scalar %.3f ms = %.1f TFLOP/s (%.2f cyc/wave-instr/SIMD @2.4GHz) | packed %.3f ms = %.1f TFLOP/s (%.2f cyc/wave-instr/SIMD)\n",
           waves_per_simd, ms_s, 2 * fma_s / ms_s / 1e9, ms_s * 1e-3 * 2.4e9 / ((double)waves_per_simd * C * iters),
           ms_p, 2 * fma_p / ms_p / 1e9, ms_p * 1e-3 * 2.4e9 / ((double)waves_per_simd * C * iters));
  }
  return 0;
}
