// Probe: how many SCALAR instructions per cycle does a gfx950 CU retire?  Index arithmetic that hipcc keeps on the scalar
// unit (s_mul_i32 for every strided buffer offset, 64-bit division expansions) is free only if that unit has slack.
//   hipcc --offload-arch=gfx950 -O2 tools/salu_probe.hip -o build_tools/salu_probe && build_tools/salu_probe
// Every wavefront runs 8192 s_mul_i32 / s_add_u32 (four independent chains); argv[1] workgroups of 4 wavefronts
// (256 = one wavefront per SIMD, 1024 = four).  cycles per instruction per CU = t * clock / (wavefronts per CU * 8192).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
template <bool VALU>
__global__ __launch_bounds__(256) void k(int* out, int a, int reps) {
  int s0 = a, s1 = a + 1, s2 = a + 2, s3 = a + 3;
  float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (VALU)
        asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      asm volatile("s_mul_i32 %0, %0, %4\n\ts_add_u32 %1, %1, %4\n\ts_mul_i32 %2, %2, %4\n\ts_add_u32 %3, %3, %4"
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(a));
    }
  }
  if (s0 + s1 + s2 + s3 == 123456789 || v0 + v1 + v2 + v3 == 1.5f) out[0] = s0;
}
template <bool VALU>
float run(int* d, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> t;
  for (int it = 0; it < 20; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<VALU>, dim3(grid), dim3(256), 0, 0, d, 3, 8192 / 64);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    if (it == 0) { hipError_t e = hipGetLastError(); if (e != hipSuccess) printf("launch error: %s\n", hipGetErrorString(e)); }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    t.push_back(ms * 1000.f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}
int main(int argc, char** argv) {
  int* d;
  hipMalloc(&d, 4096);
  for (int grid : {256, 512, 1024, 2048}) {
    const float ts = run<false>(d, grid), tv = run<true>(d, grid);
    printf("%5d workgroups (%d wavefronts per CU): 8192 scalar instructions per wavefront %7.1f us; with 8192 v_fma interleaved %7.1f us\n",
           grid, grid / 64, ts, tv);
  }
  return 0;
}
