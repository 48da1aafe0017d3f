// Probe for VERDICT r3 item 3 (one persistent kernel for the small-plane tail of the MobileNet step): what does a DEVICE-WIDE
// barrier between two layers cost, against the launch boundary it would replace?  G resident workgroups of 256 threads run
// L "layers" (each ~W us of arithmetic on registers, plus one cache line read from what another workgroup wrote in the previous
// layer: the dependency a layer boundary carries) separated by
//   mode 0: nothing (the floor: L x W);
//   mode 1: a grid barrier - one agent-scope atomic add per workgroup on a counter, thread 0 spins on it with agent-scope
//           acquire loads, __syncthreads (monotonic counter: layer l waits for l * G arrivals; BOUNDED spin, a stuck barrier
//           sets an error flag and leaves);
//   mode 2: L separate launches of one layer each (what the step does today), back to back on one stream.
// Prints us per layer boundary = (total - floor) / (L - 1).  G <= resident workgroups (256 CUs x 2 is safe for 256 threads).
//   hipcc --offload-arch=gfx950 -O2 tools/grid_barrier_probe.hip -o build_tools/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned work(unsigned v, int iters) {
  for (int i = 0; i < iters; ++i) v = v * 1664525u + 1013904223u;
  return v;
}

__global__ __launch_bounds__(256) void layers(unsigned* counter, unsigned* data, unsigned* err, int L, int iters, int mode,
                                              int layer0) {
  unsigned v = threadIdx.x + blockIdx.x * 977u;
  const unsigned G = gridDim.x;
  for (int l = 0; l < L; ++l) {
    const int layer = layer0 + l;
    // read what the "previous layer" of ANOTHER workgroup left (agent-scope load: it was written on another CU / XCD)
    const unsigned peer = (blockIdx.x * 37u + 11u * (unsigned)layer) % G;
    v += __hip_atomic_load(data + peer * 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = work(v, iters);
    if (threadIdx.x == 0) __hip_atomic_store(data + blockIdx.x * 32u, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mode == 1 && l + 1 < L) {
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)(l + 1) * G;
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
          if (++spins > (1u << 22)) {                 // ~a second: never on a healthy run
            err[0] = 1u;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
    }
  }
  if (v == 12345u) err[1] = v;
}

static float run(int G, int L, int iters, int mode, unsigned* counter, unsigned* data, unsigned* err, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int r = 0; r < reps + 3; ++r) {
    hipMemsetAsync(counter, 0, 4, 0);
    hipEventRecord(a, 0);
    if (mode == 2) {
      for (int l = 0; l < L; ++l) hipLaunchKernelGGL(layers, dim3(G), dim3(256), 0, 0, counter, data, err, 1, iters, 0, l);
    } else {
      hipLaunchKernelGGL(layers, dim3(G), dim3(256), 0, 0, counter, data, err, L, iters, mode, 0);
    }
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (r >= 3 && ms < best) best = ms;
  }
  return best * 1000.0f;
}

int main() {
  unsigned *counter, *data, *err;
  hipMalloc(&counter, 256);
  hipMalloc(&data, 1 << 20);
  hipMalloc(&err, 256);
  hipMemset(data, 0, 1 << 20);
  hipMemset(err, 0, 256);
  const int L = 6;
  for (int G : {256, 512})
    for (int iters : {400, 4000}) {                   // ~2 us and ~20 us layers
      const float floor_us = run(G, L, iters, 0, counter, data, err, 20);
      const float bar_us = run(G, L, iters, 1, counter, data, err, 20);
      const float launch_us = run(G, L, iters, 2, counter, data, err, 20);
      unsigned e[2];
      hipMemcpy(e, err, 8, hipMemcpyDeviceToHost);
      printf("G %3d workgroups, %d layers of %5.1f us: grid barrier %5.2f us per boundary, launch boundary %5.2f us per boundary"
             "%s\n", G, L, floor_us / L, (bar_us - floor_us) / (L - 1), (launch_us - floor_us) / (L - 1),
             e[0] ? "  (A BARRIER TIMED OUT)" : "");
    }
  return 0;
}
