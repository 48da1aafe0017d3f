// LDS-DMA probe (round 5): what `buffer_load_dwordx4 ... offen lds` does on gfx950 - destination = M0 + 16 * lane (also beyond
// 64 KB of a 160 KB allocation), lanes whose offset is out of the resource's range, and vmcnt accounting next to stores.
//   hipcc --offload-arch=gfx950 -O3 tools/ldsdma_probe.hip -o /tmp/ldsdma_probe && /tmp/ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(const float* x, float* y, int nbytes, unsigned lds_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (unsigned i = threadIdx.x; i < 1024 * 8 / 4; i += 512) reinterpret_cast<float*>(sm + lds_off)[i] = -1.0f;
  __syncthreads();
  v4i rs;
  const unsigned long long b = (unsigned long long)x;
  rs[0] = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
  rs[1] = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  rs[2] = __builtin_amdgcn_readfirstlane(nbytes);
  rs[3] = 0x00020000;
  const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)sm + lds_off + 1024u * wave);
  // lanes 60..63 of every wavefront ask for an offset out of range
  const unsigned voff = lane < 60 ? (wave * 64 + lane) * 16u : 0x80000000u;
  const unsigned soff = 0;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ldsb), "v"(voff), "s"(rs),
               "s"(soff)
               : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (unsigned i = threadIdx.x; i < 1024 * 8 / 4; i += 512) y[i] = reinterpret_cast<float*>(sm + lds_off)[i];
}
int main() {
  const int n = 8 * 64 * 4;
  std::vector<float> hx(n), hy(n);
  for (int i = 0; i < n; ++i) hx[i] = (float)i;
  float *x, *y;
  hipMalloc(&x, n * 4);
  hipMalloc(&y, n * 4);
  hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (unsigned off : {0u, 60u * 1024u, 100u * 1024u, 150u * 1024u}) {
    hipMemset(y, 0, n * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(512), off + 8192, 0, x, y, n * 4, off);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(hy.data(), y, n * 4, hipMemcpyDeviceToHost);
    int good = 0, zero = 0, kept = 0, other = 0;
    for (int i = 0; i < n; ++i) {
      const int lane = (i / 4) % 64;
      if (lane < 60) good += hy[i] == hx[i];
      else if (hy[i] == 0.0f) ++zero;
      else if (hy[i] == -1.0f) ++kept;
      else ++other;
    }
    printf("lds offset %6u: %s  in-range values right %d / %d; out-of-range lanes: zero-filled %d, untouched %d, other %d (of %d)\n",
           off, hipGetErrorString(e), good, 8 * 60 * 4, zero, kept, other, 8 * 4 * 4);
  }
  return 0;
}
