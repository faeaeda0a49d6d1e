// Microbenchmark / probe: which XCD does workgroup b of a launch run on?  (HW_REG_XCC_ID, gfx940+.)
// Build: hipcc -O3 --offload-arch=gfx950 xcd_map.hip -o xcd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_where(unsigned *out) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
    out[blockIdx.x * 2] = xcc;
    out[blockIdx.x * 2 + 1] = hw;
  }
}

int main() {
  for (int threads : {64, 256}) {
    for (int n : {64, 4096}) {
      unsigned *d;
      hipMalloc(&d, n * 8);
      hipLaunchKernelGGL(k_where, dim3(n), dim3(threads), 0, 0, d);
      std::vector<unsigned> h(n * 2);
      hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
      printf("workgroup size %d, %d workgroups: XCC of workgroups 0..31:", threads, n);
      for (int b = 0; b < 32; ++b) printf(" %u", h[b * 2] & 15u);
      int rr = 0;
      for (int b = 0; b < n; ++b) rr += (h[b * 2] & 15u) == (unsigned)(b % 8);
      printf("\n   workgroups with XCC == b %% 8: %d of %d\n", rr, n);
      hipFree(d);
    }
  }
  return 0;
}
