// Probe: what does a per-sample statistic atomic cost at the END of a short kernel?  G one-wavefront workgroups each issue ONE
// atomicMax (no return) to one of A addresses, after a short delay loop; the kernel's duration (HIP events over 200 launches,
// minus the same kernel without the atomic) is the exposed tail.  Mappings:  mod = workgroup b -> address b % A (A % 8 == 0: every
// workgroup of an address sits on the same XCD, workgroups being dealt round-robin to the 8 XCDs);  div = b -> b / (G / A)
// (consecutive workgroups = 8 different XCDs share an address).  Build here, run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/atomic_probe.hip -o quantization/mxnet_amd/csrc/build/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p, int A, int G, int mode, int do_atomic, unsigned* sink, int stride = 32) {
  unsigned v = threadIdx.x + blockIdx.x;
  for (int i = 0; i < 200; ++i) v = v * 1664525u + 1013904223u;          // ~1 us of work
  if (threadIdx.x == 0) {
    const int a = mode == 0 ? blockIdx.x % A : blockIdx.x / (G / A);
    if (do_atomic == 1) atomicMax(p + a * stride, v | 1u);               // (addresses 128 bytes apart by default)
    if (do_atomic == 2) unsafeAtomicAdd(reinterpret_cast<float*>(p) + a * stride, 1.0f);
    if (do_atomic == 3) atomicAdd(reinterpret_cast<float*>(p) + a * 32, 1.0f);
  }
  if (v == 12345u) sink[0] = v;
}
int main() {
  unsigned *p, *sink;
  hipMalloc(&p, 1 << 22);
  hipMalloc(&sink, 64);
  hipMemset(p, 0, 1 << 22);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int Gs[] = {512, 2048, 8192};
  const int As[] = {1, 8, 128, 2048};
  for (int G : Gs)
    for (int A : As) {
      if (A > G) continue;
      for (int mode = 0; mode < 2; ++mode)
        for (int da = 0; da < 4; ++da) {
          for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(G), dim3(64), 0, 0, p, A, G, mode, da, sink);
          hipEventRecord(a, 0);
          for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(G), dim3(64), 0, 0, p, A, G, mode, da, sink);
          hipEventRecord(b, 0);
          hipEventSynchronize(b);
          float ms;
          hipEventElapsedTime(&ms, a, b);
          printf("G %5d  A %5d  per address %5d  map %s  op %-16s  %7.2f us per launch\n", G, A, G / A, mode ? "div" : "mod",
                 da == 0 ? "none" : da == 1 ? "atomicMax u32" : da == 2 ? "hw add f32" : "CAS add f32", ms * 1000.0f / 200.0f);
        }
    }
  // the statistic rows of the library: A = 128 samples, one atomicMax per workgroup; how far apart must the slots be?
  const int strides[] = {1, 2, 4, 8, 16, 32, 64, 1024};
  for (int G : {512, 2048, 8192})
    for (int stride : strides)
      for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(G), dim3(64), 0, 0, p, 128, G, mode, 1, sink, stride);
        hipEventRecord(a, 0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(G), dim3(64), 0, 0, p, 128, G, mode, 1, sink, stride);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("G %5d  A   128  slots %5d bytes apart  map %s  atomicMax u32  %7.2f us per launch\n", G, stride * 4,
               mode ? "div" : "mod", ms * 1000.0f / 200.0f);
      }
  return 0;
}
