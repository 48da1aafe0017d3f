// Access-pattern probe for the statistic-only pointwise pass (round 6): how fast can a wavefront-per-pixel-tile read of an
// NCHW tensor go when every load instruction touches two channel planes?  Patterns: per lane 4 / 8 / 16 bytes of ONE channel
// (32 lanes = 128 / 256 / 512 contiguous bytes per plane and instruction), 16 channels per half-wave, D slabs in flight.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/statload_probe.hip -o /tmp/slp && /tmp/slp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int V, int D>      // V floats per lane, D tiles in flight
__global__ __launch_bounds__(256) void probe(const float* __restrict__ x, int C, int HW, long tiles, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i32 = lane & 31;
  const long nw = (long)gridDim.x * 4, wid = (long)blockIdx.x * 4 + wave;
  const long t0 = tiles * wid / nw, t1 = tiles * (wid + 1) / nw;
  typedef float vf __attribute__((ext_vector_type(V)));
  float m = 0.f;
  const int tpx = 32 * V;                       // pixels per tile
  vf buf[D][16];
  auto issue = [&](long t, vf (&v)[16]) {
    if (t >= t1) return;
    const long j = t * tpx + (long)i32 * V;
    const long smp = j / HW, p = j - smp * HW;
    for (int kt = 0; kt < C / 32; ++kt) {
      const float* b = x + ((smp * C + kt * 32 + 16 * h) * (long)HW + p);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const vf*>(b + (long)i * HW);
    }
  };
#pragma unroll
  for (int d = 0; d < D - 1; ++d) issue(t0 + d, buf[d]);
  for (long t = t0; t < t1; t += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (t + d >= t1) break;
      issue(t + d + D - 1, buf[(d + D - 1) % D]);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < V; ++e) m = fmaxf(m, buf[d][i][e]);
    }
  }
  if (m == 12345.678f) out[0] = m;
}

template <int V, int D>
void run(const float* x, int n, int C, int HW, float* out, int wg_per_cu) {
  const long tiles = (long)n * HW / (32 * V);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = 256 * wg_per_cu;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<V, D>), dim3(grid), dim3(256), 0, 0, x, C, HW, tiles, out);
  float best = 1e9f;
  for (int r = 0; r < 10; ++r) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((probe<V, D>), dim3(grid), dim3(256), 0, 0, x, C, HW, tiles, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
  }
  const double mb = 4e-6 * n * C * (double)HW;
  printf("  %2d B per lane, %d tiles in flight, %d workgroups per CU: %7.1f us  %5.2f TB/s\n", 4 * V, D, wg_per_cu, best * 1e3, mb / (best * 1e3));
}

int main() {
  const int shapes[3][3] = {{128, 32, 112 * 112}, {128, 64, 56 * 56}, {128, 128, 56 * 56}};
  for (auto& s : shapes) {
    const int n = s[0], C = s[1], HW = s[2];
    float *x, *out;
    CK(hipMalloc(&x, (size_t)n * C * HW * 4)); CK(hipMalloc(&out, 4));
    CK(hipMemset(x, 0, (size_t)n * C * HW * 4));
    printf("(%d, %d, %d) = %.0f MB\n", n, C, HW, 4e-6 * n * C * (double)HW);
    for (int wg : {2, 3, 4, 8}) {
      run<1, 2>(x, n, C, HW, out, wg);
      run<1, 4>(x, n, C, HW, out, wg);
      run<2, 2>(x, n, C, HW, out, wg);
      run<2, 3>(x, n, C, HW, out, wg);
      run<4, 2>(x, n, C, HW, out, wg);
    }
    CK(hipFree(x)); CK(hipFree(out));
  }
  return 0;
}
