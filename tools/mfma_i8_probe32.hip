// Probe the operand layout of v_mfma_i32_32x32x32_i8 on gfx950.  Hypothesis: lane l holds, for A, row l&31 and the 16
// consecutive k = 16*(l>>5) + 0..15; for B, column l&31 and the same k; C/D (16 registers): column l&31,
// row = 8*(r/4) + 4*(l>>5) + (r%4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void probe(const int8_t* A /*32x32 row-major*/, const int8_t* B /*32x32 row-major: B[k][j]*/, int* D /*32x32*/) {
  const int l = threadIdx.x;
  const int i = l & 31, kb = (l >> 5) * 16;
  v4i a, b;
  v16i c;
  for (int r = 0; r < 16; ++r) c[r] = 0;
  int8_t ta[16], tb[16];
  for (int t = 0; t < 16; ++t) { ta[t] = A[i * 32 + kb + t]; tb[t] = B[(kb + t) * 32 + i]; }
  __builtin_memcpy(&a, ta, 16);
  __builtin_memcpy(&b, tb, 16);
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * (l >> 5) + (r % 4)) * 32 + (l & 31)] = c[r];
}
int main() {
  int8_t hA[32 * 32], hB[32 * 32]; int hD[1024], ref[1024];
  srand(7);
  for (auto& v : hA) v = (int8_t)(rand() % 255 - 127);
  for (auto& v : hB) v = (int8_t)(rand() % 255 - 127);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < 32; ++k) s += (int)hA[i * 32 + k] * (int)hB[k * 32 + j]; ref[i * 32 + j] = s; }
  int8_t *dA, *dB; int* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemset(dD, 0xff, sizeof hD);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0; for (int t = 0; t < 1024; ++t) bad += hD[t] != ref[t];
  printf("mfma_i32_32x32x32_i8 layout hypothesis: %s (%d mismatches) D[0]=%d ref[0]=%d D[33]=%d ref[33]=%d\n", bad ? "WRONG" : "CONFIRMED", bad, hD[0], ref[0], hD[33], ref[33]);
  return bad != 0;
}
