// Microbenchmark / probe: typed buffer loads on gfx950.  Can the texture-address unit convert packed RGBA8 texels to
// floats on the way into the registers (buffer_load_format_xyzw through a V# with data format 8_8_8_8, numeric format
// USCALED), for records of 9 bytes (offsets 0, 4, 8: not dword aligned), and at what rate compared with a plain
// global_load_dwordx3 of the same record followed by v_cvt_f32_ubyteN?
// Build: hipcc -O3 --offload-arch=gfx950 format_load.hip -o format_load
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// LLVM intrinsics by name (no clang builtin exists for the typed loads)
__device__ v4f llvm_struct_buffer_load_format_v4f32(v4i rsrc, int vindex, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.struct.buffer.load.format.v4f32");
__device__ float llvm_struct_buffer_load_format_f32(v4i rsrc, int vindex, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.struct.buffer.load.format.f32");

// V# (gfx9 layout): base, stride (bytes) in word1[29:16], num_records, word3 = dst_sel xyzw | num_format << 12 | data_format << 15
__device__ __host__ inline uint32_t word3(uint32_t data_format, uint32_t num_format, bool one_channel) {
  const uint32_t sel = one_channel ? (4u | (0u << 3) | (0u << 6) | (1u << 9)) : (4u | (5u << 3) | (6u << 6) | (7u << 9));
  return sel | (num_format << 12) | (data_format << 15);
}

__device__ v4i make_rsrc(const void *base, uint32_t stride, uint32_t num_records, uint32_t w3) {
  const uint64_t a = (uint64_t)base;
  v4i r;
  r.x = (int)(uint32_t)a;
  r.y = (int)(((uint32_t)(a >> 32) & 0xFFFFu) | (stride << 16));
  r.z = (int)num_records;
  r.w = (int)w3;
  return r;
}

constexpr uint32_t kFmt8888 = 10u, kFmt8 = 1u, kUscaled = 2u;

__global__ void k_probe(const uint8_t *__restrict__ texels, uint32_t n, const uint32_t *__restrict__ idx, float *__restrict__ out) {
  const v4i r4 = make_rsrc(texels, 9u, n, word3(kFmt8888, kUscaled, false));
  const v4i r1 = make_rsrc(texels, 9u, n, word3(kFmt8, kUscaled, true));
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = (int)idx[t];
  const v4f a = llvm_struct_buffer_load_format_v4f32(r4, i, 0, 0, 0);
  const v4f b = llvm_struct_buffer_load_format_v4f32(r4, i, 4, 0, 0);
  const float c = llvm_struct_buffer_load_format_f32(r1, i, 8, 0, 0);
  float *o = out + (size_t)t * 9;
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
  o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  o[8] = c;
}

// rate: each lane sums ITER random texels (9 channels), typed loads vs dwordx3 + cvt
template <bool TYPED>
__global__ __launch_bounds__(256) void k_rate(const uint8_t *__restrict__ texels, uint32_t n, uint32_t mask, float *__restrict__ out, int iters) {
  const v4i r4 = make_rsrc(texels, 9u, n, word3(kFmt8888, kUscaled, false));
  const v4i r1 = make_rsrc(texels, 9u, n, word3(kFmt8, kUscaled, true));
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const uint32_t i = (s >> 8) & mask;
    if (TYPED) {
      const v4f a = llvm_struct_buffer_load_format_v4f32(r4, (int)i, 0, 0, 0);
      const v4f b = llvm_struct_buffer_load_format_v4f32(r4, (int)i, 4, 0, 0);
      const float c = llvm_struct_buffer_load_format_f32(r1, (int)i, 8, 0, 0);
      acc += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + c;
    } else {
      uint32_t w[3];
      __builtin_memcpy(w, texels + 9u * i, 12);
      acc += (((float)(w[0] & 255u) + (float)((w[0] >> 8) & 255u)) + ((float)((w[0] >> 16) & 255u) + (float)(w[0] >> 24))) +
             (((float)(w[1] & 255u) + (float)((w[1] >> 8) & 255u)) + ((float)((w[1] >> 16) & 255u) + (float)(w[1] >> 24))) +
             (float)(w[2] & 255u);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  const uint32_t n = 1u << 20;  // 9 MB of texels
  std::vector<uint8_t> h((size_t)n * 9 + 16);
  uint32_t s = 12345u;
  for (auto &b : h) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
  uint8_t *d_tex; uint32_t *d_idx; float *d_out;
  const uint32_t threads = 256 * 64;
  std::vector<uint32_t> idx(threads);
  for (auto &i : idx) { s = s * 1664525u + 1013904223u; i = (s >> 4) % n; }
  idx[0] = 0; idx[1] = n - 1; idx[2] = 1; idx[3] = 7;
  hipMalloc(&d_tex, h.size()); hipMalloc(&d_idx, threads * 4); hipMalloc(&d_out, (size_t)256 * 8 * 256 * 9 * 4);  // k_probe: threads x 9 floats; k_rate: one float per thread of 2048 workgroups
  hipMemcpy(d_tex, h.data(), h.size(), hipMemcpyHostToDevice);
  hipMemcpy(d_idx, idx.data(), threads * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(64), dim3(256), 0, 0, d_tex, n, d_idx, d_out);
  if (hipDeviceSynchronize() != hipSuccess) { printf("probe kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  std::vector<float> o((size_t)threads * 9);
  hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (uint32_t t = 0; t < threads; ++t)
    for (int k = 0; k < 9; ++k)
      if (o[(size_t)t * 9 + k] != (float)h[(size_t)idx[t] * 9 + k]) {
        if (bad < 5) printf("mismatch lane %u ch %d: got %g want %u\n", t, k, o[(size_t)t * 9 + k], h[(size_t)idx[t] * 9 + k]);
        ++bad;
      }
  printf("typed loads of 9-byte records (8_8_8_8 USCALED @0, @4; 8 USCALED @8): %zu mismatches of %zu values\n", bad, o.size());

  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (uint32_t mask : {0xFFu, 0xFFFFu, 0xFFFFFu}) {
    for (int typed = 0; typed < 2; ++typed) {
      const int iters = 2000, grid = 256 * 8;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        if (typed) hipLaunchKernelGGL((k_rate<true>), dim3(grid), dim3(256), 0, 0, d_tex, n, mask, d_out, iters);
        else hipLaunchKernelGGL((k_rate<false>), dim3(grid), dim3(256), 0, 0, d_tex, n, mask, d_out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double texels = (double)grid * 256 * iters;
      printf("footprint %8u texels (%6.1f KB)  %-28s %8.3f ms  %7.1f Gtexel/s  %6.2f texels/clk/CU @2.4GHz\n", mask + 1, (mask + 1) * 9.0 / 1024,
             typed ? "typed (3 format loads)" : "dwordx3 + 9 v_cvt_f32_ubyte", best, texels / best / 1e6, texels / (best * 1e-3) / 256 / 2.4e9);
    }
  }
  return 0;
}
