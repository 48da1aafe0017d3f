// Probe: what does a cold pass through straight-line code cost on gfx950?  Every kernel launch starts with an empty
// instruction cache; a kernel whose hot path is N unrolled instructions pays the fetch of N * 8 bytes before it runs at speed.
//   hipcc --offload-arch=gfx950 -O2 tools/icache_probe.hip -o build_tools/icache_probe && build_tools/icache_probe
// Each kernel runs the SAME 4096 v_fma per wavefront (four independent chains; argv[1] = workgroups of 4 wavefronts, 256 =
// one wavefront per SIMD, 1280 = five):
//   straight<N>: N instructions unrolled, repeated 4096/N times in a loop  (N = 64: the loop body stays cached after one pass)
// and reports per-launch time; (t[N=4096] - t[N=64]) / 32 KB is the cold fetch cost per KB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
template <int N>
__global__ __launch_bounds__(256) void straight(float* out, float a, float b, int reps) {
  float v = threadIdx.x, v1 = v + 1.f, v2 = v + 2.f, v3 = v + 3.f;     // four independent chains: issue-bound, not latency-bound
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i)
      asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                   : "+v"(v), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
  }
  if (v + v1 + v2 + v3 == 12345.678f) out[0] = v;
}
template <int N>
float run(float* d, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> t;
  for (int it = 0; it < 30; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(straight<N>, dim3(grid), dim3(256), 0, 0, d, 1.0001f, 0.5f, 4096 / N);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    t.push_back(ms * 1000.f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}
int main(int argc, char** argv) {
  float* d;
  hipMalloc(&d, 4096);
  const int grid = argc > 1 ? atoi(argv[1]) : 256;
  printf("4096 v_fma per wavefront (4 independent chains), %d workgroups of 4 wavefronts; median us per launch (event pair included)\n", grid);
  printf("  unrolled   64: %7.2f\n", run<64>(d, grid));
  printf("  unrolled  256: %7.2f\n", run<256>(d, grid));
  printf("  unrolled 1024: %7.2f\n", run<1024>(d, grid));
  printf("  unrolled 2048: %7.2f\n", run<2048>(d, grid));
  printf("  unrolled 4096: %7.2f\n", run<4096>(d, grid));
  printf("  unrolled   64: %7.2f (again)\n", run<64>(d, grid));
  return 0;
}
