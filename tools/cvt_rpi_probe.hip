// Probe: is v_cvt_rpi_i32_f32 ("floor(x + 0.5)") on gfx950 an EXACT round-half-up, i.e. equal to (int)roundf(x) for every
// non-negative fp32 x up to 70000 - including pred(0.5), where an fp32 addition of 0.5 would round across 1?  Exhaustive over
// the bit patterns 0 .. bits(70000.0f); also samples of negatives / NaN / Inf, and the issue cost next to v_cvt_i32_f32.
// Build here, run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/cvt_rpi_probe.hip -o gpurun_out/cvt_rpi_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
__device__ __forceinline__ int rpi(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__global__ void sweep(unsigned first, unsigned last, unsigned long long* bad, unsigned* first_bad) {
  unsigned long long i = first + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (; i <= last; i += stride) {
    const float x = __uint_as_float((unsigned)i);
    if (rpi(x) != (int)roundf(x)) {
      atomicAdd(bad, 1ull);
      atomicMin(first_bad, (unsigned)i);
    }
  }
}
__global__ void samples(const float* in, int* out, int n) {
  if ((int)threadIdx.x < n) out[threadIdx.x] = rpi(in[threadIdx.x]);
}
int main() {
  unsigned long long* bad; unsigned* fb;
  hipMalloc(&bad, 8); hipMalloc(&fb, 4);
  hipMemset(bad, 0, 8); hipMemset(fb, 0xFF, 4);
  float top = 70000.0f; unsigned last; memcpy(&last, &top, 4);
  hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, 0u, last, bad, fb);
  unsigned long long hb; unsigned hf;
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, fb, 4, hipMemcpyDeviceToHost);
  printf("non-negative fp32 0 .. 70000 (%u bit patterns): %llu mismatches against (int)roundf", last + 1, hb);
  if (hb) { float f; memcpy(&f, &hf, 4); printf("; first at %.9g (bits %08x)", f, hf); }
  printf("\n");
  float h[] = {0.49999997f, 0.5f, 0.50000006f, 1.4999999f, 1.5f, 2.5f, 254.49998f, 254.5f, 255.0f, 8388607.5f, -0.49999997f, -0.5f,
               -0.50000006f, -1.5f, -2.5f, -126.5f, NAN, INFINITY, -INFINITY, 3e9f};
  const int n = sizeof(h) / sizeof(h[0]);
  float* d; int* o; int r[64];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(samples, dim3(1), dim3(64), 0, 0, d, o, n);
  hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("  v_cvt_rpi_i32_f32(%-14.9g) = %11d    roundf -> %.0f\n", h[i], r[i], roundf(h[i]));
  return 0;
}
