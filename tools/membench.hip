// Standalone HBM streaming micro-benchmark used to find the achievable ceilings on MI355X for the access patterns
// of the fake-quant kernels (read-only reduce, read+write apply, two-pass forward/backward order).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/membench.hip -o gpurun_out/membench   (runs on the GPU box)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e = (x);                                                              \
    if (e != hipSuccess) {                                                           \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);   \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

template <int BLOCK, int UNROLL, bool NT>
__global__ __launch_bounds__(BLOCK) void read_max(const f4* __restrict__ x, long nvec, float* out) {
  float m = 0.f;
  const long chunk = (long)BLOCK * UNROLL;
  const long nchunks = nvec / chunk;
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const f4* p = x + c * chunk;
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT)
        v[u] = __builtin_nontemporal_load(p + threadIdx.x + u * BLOCK);
      else
        v[u] = p[threadIdx.x + u * BLOCK];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
  }
  if (m == 12345.678f) out[0] = m;   // keep the loads alive
}

template <int BLOCK, int UNROLL, bool NTL, bool NTS, bool MATH, bool REVERSE>
__global__ __launch_bounds__(BLOCK) void copy_k(const f4* __restrict__ x, f4* __restrict__ y, long nvec,
                                                float denom, float scale, float hi) {
  const long chunk = (long)BLOCK * UNROLL;
  const long nchunks = nvec / chunk;
  for (long c0 = blockIdx.x; c0 < nchunks; c0 += gridDim.x) {
    const long c = REVERSE ? (nchunks - 1 - c0) : c0;
    const f4* p = x + c * chunk;
    f4* o = y + c * chunk;
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NTL)
        v[u] = __builtin_nontemporal_load(p + threadIdx.x + u * BLOCK);
      else
        v[u] = p[threadIdx.x + u * BLOCK];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      f4 r = v[u];
      if (MATH) {
        r.x = roundf(fminf(fmaxf(r.x, 0.f), hi) / denom) * scale;
        r.y = roundf(fminf(fmaxf(r.y, 0.f), hi) / denom) * scale;
        r.z = roundf(fminf(fmaxf(r.z, 0.f), hi) / denom) * scale;
        r.w = roundf(fminf(fmaxf(r.w, 0.f), hi) / denom) * scale;
      }
      if (NTS)
        __builtin_nontemporal_store(r, o + threadIdx.x + u * BLOCK);
      else
        o[threadIdx.x + u * BLOCK] = r;
    }
  }
}

struct Timer {
  hipEvent_t a, b;
  Timer() {
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
  }
};

template <class F>
float median_ms(F&& f, int iters = 15) {
  Timer t;
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < iters; ++i) {
    CK(hipEventRecord(t.a));
    f();
    CK(hipEventRecord(t.b));
    CK(hipEventSynchronize(t.b));
    float ms;
    CK(hipEventElapsedTime(&ms, t.a, t.b));
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

template <int BLOCK, int UNROLL, bool NT>
void run_read(const char* name, const f4* x, long nvec, float* out, int blocks_per_cu) {
  int grid = 256 * blocks_per_cu;
  float ms = median_ms([&] { hipLaunchKernelGGL((read_max<BLOCK, UNROLL, NT>), dim3(grid), dim3(BLOCK), 0, 0, x, nvec, out); });
  printf("  read  %-28s blk=%4d unroll=%2d bpc=%2d : %7.3f ms  %7.1f GB/s\n", name, BLOCK, UNROLL, blocks_per_cu, ms,
         nvec * 16.0 / ms / 1e6);
}

template <int BLOCK, int UNROLL, bool NTL, bool NTS, bool MATH>
void run_copy(const char* name, const f4* x, f4* y, long nvec, int blocks_per_cu) {
  int grid = 256 * blocks_per_cu;
  float ms = median_ms([&] {
    hipLaunchKernelGGL((copy_k<BLOCK, UNROLL, NTL, NTS, MATH, false>), dim3(grid), dim3(BLOCK), 0, 0, x, y, nvec, 0.0157f,
                       0.0157f, 4.0f);
  });
  printf("  copy  %-28s blk=%4d unroll=%2d bpc=%2d : %7.3f ms  %7.1f GB/s (r+w)\n", name, BLOCK, UNROLL, blocks_per_cu,
         ms, nvec * 32.0 / ms / 1e6);
}

int main(int argc, char** argv) {
  long mb = argc > 1 ? atol(argv[1]) : 411;
  long nvec = mb * 1000000L / 16;
  nvec = nvec / (1024 * 16) * (1024 * 16);
  f4 *x, *y;
  float* out;
  CK(hipMalloc(&x, nvec * 16));
  CK(hipMalloc(&y, nvec * 16));
  CK(hipMalloc(&out, 64));
  std::vector<float> h(nvec * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (i * 2654435761u % 1000) < 500 ? 0.f : (float)((i * 40503u) % 4096) / 1024.f;
  CK(hipMemcpy(x, h.data(), nvec * 16, hipMemcpyHostToDevice));
  printf("tensor %.1f MB\n", nvec * 16 / 1e6);

  printf("memcpyDtoD: ");
  {
    float ms = median_ms([&] { CK(hipMemcpyAsync(y, x, nvec * 16, hipMemcpyDeviceToDevice, 0)); });
    printf("%7.3f ms %7.1f GB/s (r+w)\n", ms, nvec * 32.0 / ms / 1e6);
  }
  for (int bpc : {4, 8, 16}) {
    run_read<256, 4, false>("plain", x, nvec, out, bpc);
    run_read<256, 8, false>("plain", x, nvec, out, bpc);
    run_read<256, 16, false>("plain", x, nvec, out, bpc);
    run_read<256, 8, true>("nontemporal", x, nvec, out, bpc);
  }
  run_read<512, 8, false>("plain", x, nvec, out, 4);
  run_read<1024, 4, false>("plain", x, nvec, out, 2);
  run_read<1024, 8, false>("plain", x, nvec, out, 2);
  run_read<64, 16, false>("plain", x, nvec, out, 32);
  run_read<128, 8, false>("plain", x, nvec, out, 16);

  for (int bpc : {4, 8, 16}) {
    run_copy<256, 4, false, false, false>("plain", x, y, nvec, bpc);
    run_copy<256, 8, false, false, false>("plain", x, y, nvec, bpc);
    run_copy<256, 8, false, true, false>("nt store", x, y, nvec, bpc);
    run_copy<256, 8, true, true, false>("nt load+store", x, y, nvec, bpc);
    run_copy<256, 8, false, false, true>("plain + quant math", x, y, nvec, bpc);
    run_copy<256, 8, false, true, true>("nt store + quant math", x, y, nvec, bpc);
  }
  run_copy<512, 8, false, true, true>("nt store + quant math", x, y, nvec, 4);
  run_copy<1024, 4, false, true, true>("nt store + quant math", x, y, nvec, 2);
  run_copy<256, 4, false, true, true>("nt store + quant math", x, y, nvec, 8);
  run_copy<256, 2, false, true, true>("nt store + quant math", x, y, nvec, 8);

  // two-pass: read pass then apply pass, forward/forward vs forward/backward chunk order (MALL reuse)
  for (int rev = 0; rev < 2; ++rev) {
    float ms = median_ms([&] {
      hipLaunchKernelGGL((read_max<256, 8, false>), dim3(2048), dim3(256), 0, 0, x, nvec, out);
      if (rev)
        hipLaunchKernelGGL((copy_k<256, 8, false, false, true, true>), dim3(2048), dim3(256), 0, 0, x, y, nvec, 0.0157f,
                           0.0157f, 4.0f);
      else
        hipLaunchKernelGGL((copy_k<256, 8, false, false, true, false>), dim3(2048), dim3(256), 0, 0, x, y, nvec, 0.0157f,
                           0.0157f, 4.0f);
    });
    printf("  two-pass %s: %7.3f ms  %7.1f GB/s (12 B/elem)\n", rev ? "fwd/BACKWARD" : "fwd/fwd     ", ms,
           nvec * 48.0 / ms / 1e6);
  }
  for (int rev = 0; rev < 2; ++rev) {
    float ms = median_ms([&] {
      hipLaunchKernelGGL((read_max<256, 8, false>), dim3(2048), dim3(256), 0, 0, x, nvec, out);
      if (rev)
        hipLaunchKernelGGL((copy_k<256, 8, false, true, true, true>), dim3(2048), dim3(256), 0, 0, x, y, nvec, 0.0157f,
                           0.0157f, 4.0f);
      else
        hipLaunchKernelGGL((copy_k<256, 8, false, true, true, false>), dim3(2048), dim3(256), 0, 0, x, y, nvec, 0.0157f,
                           0.0157f, 4.0f);
    });
    printf("  two-pass nt-store %s: %7.3f ms  %7.1f GB/s (12 B/elem)\n", rev ? "fwd/BACKWARD" : "fwd/fwd     ", ms,
           nvec * 48.0 / ms / 1e6);
  }
  return 0;
}
