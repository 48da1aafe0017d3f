// Probe: do the DPP wave shifts (wave_shr:1 / wave_shl:1) of gfx950 move data exactly like __shfl_up / __shfl_down by 1?
//   hipcc --offload-arch=gfx950 -O2 tools/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, float* up, float* dn, float* up_ref, float* dn_ref) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float v = x[i];
  up[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
  dn[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
  up_ref[i] = __shfl_up(v, 1, 64);
  dn_ref[i] = __shfl_down(v, 1, 64);
}
int main() {
  const int n = 256;
  float h[n], r[4][n], *d[5];
  for (int i = 0; i < n; ++i) h[i] = 1.0f + i;
  for (int j = 0; j < 5; ++j) hipMalloc(&d[j], n * 4);
  hipMemcpy(d[0], h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, d[0], d[1], d[2], d[3], d[4]);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
  for (int j = 0; j < 4; ++j) hipMemcpy(r[j], d[j + 1], n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const int l = i & 63;
    // the shuffles return the lane's own value at the wave edge, the DPP shifts return 0 there
    if (l != 0 && r[0][i] != r[2][i]) ++bad;
    if (l != 63 && r[1][i] != r[3][i]) ++bad;
    if (l == 0 && r[0][i] != 0.0f) ++bad;
    if (l == 63 && r[1][i] != 0.0f) ++bad;
  }
  printf("dpp wave shifts: %s (%d mismatches) up[1]=%g dn[1]=%g up[0]=%g dn[63]=%g\n", bad ? "DIFFER" : "match", bad, r[0][1], r[1][1], r[0][0], r[1][63]);
  return bad != 0;
}
