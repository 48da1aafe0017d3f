// Probe: issue cost of the instructions of the exact quantiser on gfx950 (cycles per wavefront instruction at 5 wavefronts per
// SIMD, four independent chains per wavefront):  v_fma_f32 / v_med3_f32 / v_cvt_f64_f32 / v_mul_f64 / v_cvt_f32_f64 /
// v_trunc_f32 / v_cvt_i32_f32 / v_pk_fma_f32.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_probe.hip -o build_tools/valu_probe && build_tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
#define REP4(S) S S S S
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float a, int reps) {
  float v0 = threadIdx.x, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
  double d0 = v0, d1 = v1, d2 = v2, d3 = v3;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {v1, v0}, p3 = {v3, v2}, pa = {a, a};
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a));
      if (OP == 1) asm volatile("v_med3_f32 %0, %0, %4, %4\n\tv_med3_f32 %1, %1, %4, %4\n\tv_med3_f32 %2, %2, %4, %4\n\tv_med3_f32 %3, %3, %4, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a));
      if (OP == 2) asm volatile("v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
      if (OP == 3) asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0));
      if (OP == 4) asm volatile("v_cvt_f32_f64 %0, %4\n\tv_cvt_f32_f64 %1, %5\n\tv_cvt_f32_f64 %2, %6\n\tv_cvt_f32_f64 %3, %7" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
      if (OP == 5) asm volatile("v_trunc_f32 %0, %0\n\tv_trunc_f32 %1, %1\n\tv_trunc_f32 %2, %2\n\tv_trunc_f32 %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      if (OP == 6) asm volatile("v_cvt_i32_f32 %0, %0\n\tv_cvt_i32_f32 %1, %1\n\tv_cvt_i32_f32 %2, %2\n\tv_cvt_i32_f32 %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      if (OP == 7) asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
      if (OP == 8) asm volatile("v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\tv_fma_f64 %3, %3, %4, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0));
    }
  }
  if (v0 + v1 + v2 + v3 + (float)(d0 + d1 + d2 + d3) + p0.x + p1.x + p2.y + p3.y == 12345.678f) out[0] = v0;
}
template <int OP>
float run(float* d, int grid, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> t;
  for (int it = 0; it < 20; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 1.0001f, reps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    t.push_back(ms * 1000.f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}
template <int OP>
void report(float* d, const char* name) {
  const int grid = 1280;                                    // 5 wavefronts per SIMD
  const float t1 = run<OP>(d, grid, 64), t2 = run<OP>(d, grid, 192);      // 4096 and 12288 instructions per wavefront
  const double per_simd = 5.0 * (12288 - 4096);
  printf("  %-16s %6.2f ns per wavefront instruction and SIMD  (= %.1f cycles at 2.4 GHz)\n", name, (t2 - t1) * 1e3 / per_simd,
         (t2 - t1) * 1e3 / per_simd * 2.4);
}
int main() {
  float* d;
  hipMalloc(&d, 4096);
  report<0>(d, "v_fma_f32");
  report<1>(d, "v_med3_f32");
  report<2>(d, "v_cvt_f64_f32");
  report<3>(d, "v_mul_f64");
  report<4>(d, "v_cvt_f32_f64");
  report<5>(d, "v_trunc_f32");
  report<6>(d, "v_cvt_i32_f32");
  report<7>(d, "v_pk_fma_f32");
  report<8>(d, "v_fma_f64");
  return 0;
}
