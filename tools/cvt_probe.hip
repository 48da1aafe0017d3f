// Probe: rounding and saturation of v_cvt_pk_u8_f32 on gfx950 (is it usable as "truncate to u8 and pack" in the quantiser?)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/cvt_probe.hip -o gpurun_out/cvt_probe && gpurun_out/cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, unsigned* out, int n) {
  int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0xAABBCC00u);
}
int main() {
  float h[] = {0.0f, 0.25f, 0.49999997f, 0.5f, 0.75f, 0.99999994f, 1.0f, 1.5f, 2.5f, 3.5f, 126.5f, 127.49999f, 254.5f,
               254.99998f, 255.0f, 255.4f, 255.5f, 256.0f, 300.0f, 1e9f, -0.3f, -0.5f, -1.0f, -100.0f, NAN, INFINITY, -INFINITY};
  const int n = sizeof(h) / sizeof(h[0]);
  float* d; unsigned* o; unsigned r[64];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%-14.9g -> byte %3u   (upper bytes kept: %06x)\n", h[i], r[i] & 255u, r[i] >> 8);
  return 0;
}
