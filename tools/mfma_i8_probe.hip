// Probe the operand layout of v_mfma_i32_16x16x64_i8 on gfx950 (hypothesis: lane l holds row/col l&15 and the 16
// consecutive k = 16*(l>>4) + 0..15; C/D: col = l&15, row = 4*(l>>4) + r).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(const int8_t* A /*16x64 row-major*/, const int8_t* B /*64x16 row-major: B[k][j]*/, int* D /*16x16*/) {
  const int l = threadIdx.x;
  const int i = l & 15, kb = (l >> 4) * 16;
  v4i a, b, c = {0, 0, 0, 0};
  int8_t ta[16], tb[16];
  for (int t = 0; t < 16; ++t) { ta[t] = A[i * 64 + kb + t]; tb[t] = B[(kb + t) * 16 + i]; }
  __builtin_memcpy(&a, ta, 16);
  __builtin_memcpy(&b, tb, 16);
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
int main() {
  int8_t hA[16 * 64], hB[64 * 16]; int hD[256], ref[256];
  srand(7);
  for (auto& v : hA) v = (int8_t)(rand() % 255 - 127);
  for (auto& v : hB) v = (int8_t)(rand() % 255 - 127);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int s = 0; for (int k = 0; k < 64; ++k) s += (int)hA[i * 64 + k] * (int)hB[k * 16 + j]; ref[i * 16 + j] = s; }
  int8_t *dA, *dB; int* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0; for (int t = 0; t < 256; ++t) bad += hD[t] != ref[t];
  printf("mfma_i32_16x16x64_i8 layout hypothesis: %s (%d mismatches) D[0]=%d ref[0]=%d D[17]=%d ref[17]=%d\n", bad ? "WRONG" : "CONFIRMED", bad, hD[0], ref[0], hD[17], ref[17]);
  return bad != 0;
}
