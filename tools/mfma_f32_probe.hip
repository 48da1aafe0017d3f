// Probe: is v_mfma_f32_32x32x2_f32 bitwise an fmaf chain, and in which k order?  (gfx950; build with hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(const float* A /*[32][K]*/, const float* B /*[K][32]*/, const float* C /*[32][32]*/, float* D, int K) {
  const int l = threadIdx.x, j = l & 31, h = l >> 5;
  v16f acc;
  for (int r = 0; r < 16; ++r) acc[r] = C[((r / 4) * 8 + h * 4 + r % 4) * 32 + j];
  for (int k0 = 0; k0 < K; k0 += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j * K + k0 + h], B[(k0 + h) * 32 + j], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r / 4) * 8 + h * 4 + r % 4) * 32 + j] = acc[r];
}
int main() {
  const int K = 28;
  float hA[32 * K], hB[K * 32], hC[1024], hD[1024];
  srand(5);
  auto rnd = []() { return (float)((rand() % 20001) - 10000) / 777.0f * ((rand() & 1) ? 1.0f : 1e-3f); };
  for (auto& v : hA) v = rnd();
  for (auto& v : hB) v = rnd();
  for (auto& v : hC) v = rnd();
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, K);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  int same_fwd = 0, same_rev = 0, same_pair = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      float f = hC[i * 32 + j], r = hC[i * 32 + j], p = hC[i * 32 + j];
      for (int k0 = 0; k0 < K; k0 += 2) {
        f = fmaf(hA[i * K + k0 + 1], hB[(k0 + 1) * 32 + j], fmaf(hA[i * K + k0], hB[k0 * 32 + j], f));          // k ascending
        r = fmaf(hA[i * K + k0], hB[k0 * 32 + j], fmaf(hA[i * K + k0 + 1], hB[(k0 + 1) * 32 + j], r));          // pair reversed
        p = (float)((double)p + ((double)hA[i * K + k0] * hB[k0 * 32 + j] + (double)hA[i * K + k0 + 1] * hB[(k0 + 1) * 32 + j]));  // exact pair sum, one rounding
      }
      same_fwd += memcmp(&f, &hD[i * 32 + j], 4) == 0;
      same_rev += memcmp(&r, &hD[i * 32 + j], 4) == 0;
      same_pair += memcmp(&p, &hD[i * 32 + j], 4) == 0;
    }
  printf("K=%d: of 1024 outputs bit-equal to  fmaf chain k ascending: %d   pairs reversed: %d   exact pair sum then one rounding: %d\n",
         K, same_fwd, same_rev, same_pair);
  return 0;
}
