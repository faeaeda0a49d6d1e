// Microbenchmark: how fast can the chip START workgroups?  N workgroups of T threads whose body is (a) nothing, (b) one
// dependent global load + store, (c) as (b) + a 16-byte store per thread x P passes (a light raster tile's work) --
// for several LDS allocations per workgroup (which bound the workgroups resident per CU) and register footprints.
// Build: hipcc -O3 --offload-arch=gfx950 launch_rate.hip -o launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int BODY>
__global__ __launch_bounds__(256) void k_body(const unsigned *__restrict__ in, unsigned *__restrict__ out, float4 *__restrict__ pix, int passes) {
  extern __shared__ unsigned lds[];
  if (BODY == 0) return;
  const unsigned v = in[blockIdx.x];          // uniform address: one line
  if (BODY >= 2) {
    for (int k = 0; k < passes; ++k)
      __builtin_nontemporal_store(make_float4(0.f, 0.f, 0.f, (float)v).x, &pix[((size_t)blockIdx.x * passes + k) * 256 + threadIdx.x].x);
  }
  if (threadIdx.x == 0) out[blockIdx.x] = v + 1u;
  if (v == 0xFFFFFFFFu) lds[threadIdx.x] = v;  // (keeps the allocation alive)
}

template <int BODY>
float run(int n, int threads, size_t lds, const unsigned *in, unsigned *out, float4 *pix, int passes, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_body<BODY>, dim3(n), dim3(threads), lds, 0, in, out, pix, passes);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_body<BODY>, dim3(n), dim3(threads), lds, 0, in, out, pix, passes);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  const int N = 65536;
  unsigned *in, *out; float4 *pix;
  hipMalloc(&in, N * 4); hipMalloc(&out, N * 4); hipMalloc(&pix, (size_t)8192 * 4 * 256 * 16);
  hipMemset(in, 0, N * 4);
  printf("%-8s %-8s %-8s | %10s %10s %10s   (us; waves per us)\n", "wgs", "threads", "lds", "empty", "load+store", "light tile");
  for (int threads : {64, 256}) {
    for (int n : {2048, 8192, 32768}) {
      for (size_t lds : {(size_t)0, (size_t)20480, (size_t)28672, (size_t)40960}) {
        if (n > 8192 && lds > 0 && lds != 28672) continue;
        const float t0 = run<0>(n, threads, lds, in, out, pix, 4, 20);
        const float t1 = run<1>(n, threads, lds, in, out, pix, 4, 20);
        const float t2 = n <= 8192 ? run<2>(n, threads, lds, in, out, pix, 4, 20) : 0.f;
        const float w = (float)n * (threads / 64);
        printf("%-8d %-8d %-8zu | %7.1f %5.0f %7.1f %5.0f %7.1f %5.0f\n", n, threads, lds, t0, w / t0, t1, w / t1, t2, t2 > 0 ? w / t2 : 0.f);
      }
    }
  }
  return 0;
}
