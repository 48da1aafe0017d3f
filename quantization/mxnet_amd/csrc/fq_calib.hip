// libfakequant — calibration: K5 EMA, K6 global max, K7 histogram, K8 KL threshold search
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// K5: EMA of L scalars
// ---------------------------------------------------------------------------------------------------------------
__global__ void ema_kernel(float* __restrict__ state, const float* __restrict__ cur, int64_t n, float one_minus_m,
                           float m) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float a = one_minus_m * cur[i];
    const float b = state[i] * m;
    state[i] = a + b;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// K7: 2048-bin histogram, LDS-privatised (one copy per wavefront), zeros skipped, exact uint64 accumulation.
// Bound by the 4 B/elem read: a workgroup walks a contiguous range of 32 KiB chunks with all 8 x 16 B loads of a
// chunk in flight per lane BEFORE the first LDS atomic (the atomics are order-free, the compiler otherwise keeps one
// load outstanding per iteration: 3.6 TB/s), then bins the 32 values.  Bin = (int)(clip(v, 0, max) * bins/(max+1e-5))
// exactly as distribution_calibrate.py:39-42; an index equal to `bins` (max >= 256: fp32 max + 1e-5 == max) is clamped
// into the last bin (documented deviation).  Flush: one 64-bit global atomic per non-empty bin and workgroup.
// ---------------------------------------------------------------------------------------------------------------
template <bool VEC, bool NT, bool PIPE>
__global__ __launch_bounds__(kBlock) void histogram_kernel(const float* __restrict__ x, int64_t numel,
                                                           const float* __restrict__ max_dev, int bins,
                                                           unsigned long long* __restrict__ hist,
                                                           unsigned int* __restrict__ neg_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned int lh[];      // 4 * bins
  for (int i = threadIdx.x; i < 4 * bins; i += kBlock) lh[i] = 0u;
  __syncthreads();
  unsigned int* mine = lh + (threadIdx.x >> 6) * bins;
  const float mx = max_dev[0];
  const float scales = (float)bins / (mx + 1e-5f);                       // distribution_calibrate.py:41
  const int last = bins - 1;
  unsigned int neg = 0;
  auto put = [&](float v) {
    neg += (v < 0.0f) ? 1u : 0u;
    const float c = __builtin_amdgcn_fmed3f(v, 0.0f, mx);                // :39 (NaN -> 0, like fmin(fmax()))
    if (c != 0.0f) {                                                     // :40
      int idx = (int)(c * scales);                                       // :42 (truncation)
      idx = idx < last ? idx : last;
      atomicAdd(&mine[idx], 1u);
    }
  };
  auto put8 = [&](const f4 (&v)[kUnroll]) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      put(v[u].x);
      put(v[u].y);
      put(v[u].z);
      put(v[u].w);
    }
  };
  const int64_t chunks = (numel + kChunk - 1) / kChunk;
  const int64_t full = numel / kChunk;                                   // chunks [0, full) are whole
  const ChunkRange rg = block_range(chunks);
  const int64_t vend = VEC ? (rg.end < full ? rg.end : full) : rg.begin; // whole chunks of this workgroup: [begin, vend)
  if (VEC && rg.begin < vend) {
    const f4* p = reinterpret_cast<const f4*>(x) + rg.begin * (kChunk / kVec) + threadIdx.x;
    f4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) v[u] = ld4<NT>(p + u * kBlock);
    for (int64_t c = rg.begin; c < vend; ++c) {
      if (PIPE) {
        // next chunk's loads go out before this chunk's atomics (the last iteration re-reads its own chunk)
        const f4* pn = p + (c + 1 < vend ? (kChunk / kVec) : 0);
        f4 w[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) w[u] = ld4<NT>(pn + u * kBlock);
        FQ_PIN();
        put8(v);
        FQ_PIN();
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = w[u];
        p = pn;
      } else {
        put8(v);
        if (c + 1 < vend) {
          p += kChunk / kVec;
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) v[u] = ld4<NT>(p + u * kBlock);
        }
      }
    }
  }
  for (int64_t c = (VEC ? vend : rg.begin); c < rg.end; ++c) {           // ragged last chunk / unaligned tensors
    const int64_t base = c * (int64_t)kChunk;
    const int64_t rem = numel - base;
    const int cnt = (int)(rem < kChunk ? rem : kChunk);
    for (int i = threadIdx.x; i < cnt; i += kBlock) put(x[base + i]);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += kBlock) {
    const unsigned int c = lh[b] + lh[bins + b] + lh[2 * bins + b] + lh[3 * bins + b];
    if (c) atomicAdd(&hist[b], (unsigned long long)c);
  }
  if (neg_count != nullptr) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) neg += __shfl_xor(neg, off, 64);
    if ((threadIdx.x & 63) == 0 && neg) atomicAdd(neg_count, neg);
  }
}

__global__ void hist_to_float_kernel(const unsigned long long* __restrict__ h, float* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)h[i];
}

// ---------------------------------------------------------------------------------------------------------------
// K8: KL threshold search.  One THREAD per candidate bin count i, every sum in the reference's own order and
// precision (distribution_calibrate.py:136-171); the `levels` merged bins of each candidate live in LDS,
// laid out [level][lane] so a wavefront's accesses are conflict-free.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kKlBlock = 64;

__global__ __launch_bounds__(kKlBlock) void kl_divergence_kernel(const float* __restrict__ hist, int bins,
                                                                 int levels, int min_bins,
                                                                 double* __restrict__ div_out) {
  extern __shared__ __attribute__((aligned(16))) double q[];             // levels * kKlBlock
  const int layer = blockIdx.y;
  const int i = min_bins + blockIdx.x * kKlBlock + threadIdx.x;
  const float* __restrict__ d = hist + (int64_t)layer * bins;
  double* out = div_out + (int64_t)layer * bins;
  if (i >= bins) return;
  const int tid = threadIdx.x;
  // P (fp32): tail mass folded into bin i-1, sequential sums (python `sum` over an fp32 array)
  float tail = 0.0f;
  for (int j = i; j < bins; ++j) tail = tail + d[j];
  const float plast = d[i - 1] + tail;
  float s = 0.0f;
  for (int j = 0; j < i - 1; ++j) s = s + d[j];
  s = s + plast;
  // Q: merge i bins into `levels` (fp64 accumulate in j order)
  for (int l = 0; l < levels; ++l) q[l * kKlBlock + tid] = 0.0;
  const double di = (double)i;
  for (int j = 0; j < i; ++j) {
    const int fl = (int)((double)((long long)j * levels) / di);
    q[fl * kKlBlock + tid] += (double)d[j];
  }
  // expand with linear interpolation, mask where P == 0, sequential fp64 sum
  double qs = 0.0;
  for (int j = 0; j < i; ++j) {
    const double b = (double)((long long)j * levels) / di;
    const int fl = (int)b;
    int ce = (int)ceil(b);
    ce = ce > levels - 1 ? levels - 1 : ce;
    const double qf = q[fl * kKlBlock + tid];
    double qe = (q[ce * kKlBlock + tid] - qf) * (b - (double)fl) + qf;
    const float pj = ((j == i - 1) ? plast : d[j]) / s;
    qe = qe * ((pj != 0.0f) ? 1.0 : 0.0);
    qs = qs + qe;
  }
  double div = 0.0;
  for (int j = 0; j < i; ++j) {
    const double b = (double)((long long)j * levels) / di;
    const int fl = (int)b;
    int ce = (int)ceil(b);
    ce = ce > levels - 1 ? levels - 1 : ce;
    const double qf = q[fl * kKlBlock + tid];
    double qe = (q[ce * kKlBlock + tid] - qf) * (b - (double)fl) + qf;
    const float pj = ((j == i - 1) ? plast : d[j]) / s;
    qe = qe * ((pj != 0.0f) ? 1.0 : 0.0);
    qe = qe / qs;
    if (qe != 0.0) div = div + (double)pj * log((double)pj / qe);
  }
  out[i] = div;
}

__global__ void kl_argmin_kernel(const double* __restrict__ div, int bins, int min_bins, int32_t* __restrict__ best) {
  const int layer = blockIdx.x;
  if (threadIdx.x != 0) return;
  const double* dv = div + (int64_t)layer * bins;
  double m = INFINITY;
  int b = min_bins;
  for (int i = min_bins; i < bins; ++i)
    if (dv[i] < m) {          // strict: first minimum wins; NaN never selected (:167-169)
      m = dv[i];
      b = i;
    }
  best[layer] = b;
}


}  // namespace

extern "C" {

int fq_ema_update(float* state, const float* current, int64_t count, double momentum, fqStream_t stream) {
  FQ_REQUIRE(state && current, "fq_ema_update: null pointer");
  FQ_REQUIRE(count > 0, "fq_ema_update: count must be positive");
  // (1 - momentum) is formed in double like the python expression `(1 - momentum)` (convert.py:70), then cast
  // (momentum arrives as a double for that reason: 1 - 0.9f != fp32(1 - 0.9))
  const float omm = (float)(1.0 - momentum);
  hipLaunchKernelGGL(ema_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, state,
                     current, count, omm, (float)momentum);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_global_max(const float* x, int64_t numel, float* out, fqStream_t stream) {
  FQ_REQUIRE(x && out, "fq_global_max: null pointer");
  FQ_REQUIRE(numel > 0, "fq_global_max: empty tensor");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, st, out, (int64_t)1, -INFINITY);
  ProfScope prof(FQ_KERNEL_GLOBAL_MAX, 4.0 * (double)numel, st);
  const int grid = grid_for((numel + kChunk - 1) / kChunk);
  hipLaunchKernelGGL((minmax_kernel<false, false>), dim3(grid), dim3(kBlock), 0, st, x, numel,
                     aligned16(x) ? 1 : 0, (float*)nullptr, out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_histogram_accumulate(const float* x, int64_t numel, const float* max_dev, int bins, uint64_t* hist,
                            uint32_t* neg_count, fqStream_t stream) {
  FQ_REQUIRE(x && max_dev && hist, "fq_histogram_accumulate: null pointer");
  FQ_REQUIRE(numel > 0, "fq_histogram_accumulate: empty tensor");
  FQ_REQUIRE(bins > 0 && bins <= 8192, "fq_histogram_accumulate: bins=%d out of range (1..8192)", bins);
  // every workgroup ends with up to `bins` global 64-bit atomics: a few workgroups per CU (each with 32 KiB of loads in
  // flight) keep the stream busy while the flush stays a small fraction of the work
  static const int wg_per_cu = env_int("FQ_HIST_WG_PER_CU", 2);
  int64_t hg = (numel + kChunk - 1) / kChunk;
  if (hg > (int64_t)num_cu() * wg_per_cu) hg = (int64_t)num_cu() * wg_per_cu;
  const int grid = (int)(hg < 1 ? 1 : hg);
  ProfScope prof(FQ_KERNEL_HISTOGRAM, 4.0 * (double)numel, (hipStream_t)stream);
  const size_t lds = (size_t)4 * bins * sizeof(unsigned int);
  static const int hist_form = env_int("FQ_HIST_FORM", 1);          // bit 0: nontemporal loads, bit 1: pipelined loads
  // measured on (128,64,112,112), 2 workgroups per CU (profiles/r2_hist_variants.txt): plain 5.28, nontemporal 5.88,
  // pipelined 5.31, both 5.80 TB/s; 3-4 workgroups per CU are slower in every form
#define FQ_HIST(V, N, P)                                                                                              \
  hipLaunchKernelGGL((histogram_kernel<V, N, P>), dim3(grid), dim3(kBlock), lds, (hipStream_t)stream, x, numel,       \
                     max_dev, bins, (unsigned long long*)hist, (unsigned int*)neg_count)
  if (!aligned16(x)) FQ_HIST(false, false, false);
  else if (hist_form == 0) FQ_HIST(true, false, false);
  else if (hist_form == 1) FQ_HIST(true, true, false);
  else if (hist_form == 2) FQ_HIST(true, false, true);
  else FQ_HIST(true, true, true);
#undef FQ_HIST
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_hist_to_float(const uint64_t* hist, float* out, int64_t count, fqStream_t stream) {
  FQ_REQUIRE(hist && out && count > 0, "fq_hist_to_float: bad arguments");
  hipLaunchKernelGGL(hist_to_float_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const unsigned long long*)hist, out, count);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

size_t fq_kl_workspace_bytes(int64_t L, int bins) { return (size_t)L * (size_t)bins * sizeof(double) + 64; }

int fq_kl_search(const float* hist, int64_t L, int bins, int levels, int min_bins, int32_t* out_best, void* ws,
                 fqStream_t stream) {
  FQ_REQUIRE(hist && out_best && ws, "fq_kl_search: null pointer");
  FQ_REQUIRE(L > 0 && L < 65536, "fq_kl_search: L=%lld out of range", (long long)L);
  FQ_REQUIRE(min_bins >= levels, "min_bins should be greater than levels (%d vs. %d)", min_bins, levels);
  FQ_REQUIRE(levels >= 2 && (size_t)levels * kKlBlock * sizeof(double) <= 160 * 1024,
             "fq_kl_search: levels=%d does not fit the LDS staging (max %d)", levels,
             (int)(160 * 1024 / (kKlBlock * sizeof(double))));
  FQ_REQUIRE(bins > min_bins, "fq_kl_search: bins (%d) must exceed min_bins (%d)", bins, min_bins);
  hipStream_t st = (hipStream_t)stream;
  double* div = (double*)ws;
  const size_t lds = (size_t)levels * kKlBlock * sizeof(double);
  if (lds > 64 * 1024)
    FQ_HIP(hipFuncSetAttribute((const void*)kl_divergence_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
  const int cands = bins - min_bins;
  dim3 grid((unsigned)((cands + kKlBlock - 1) / kKlBlock), (unsigned)L);
  hipLaunchKernelGGL(kl_divergence_kernel, grid, dim3(kKlBlock), lds, st, hist, bins, levels, min_bins, div);
  FQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(kl_argmin_kernel, dim3((unsigned)L), dim3(64), 0, st, div, bins, min_bins, out_best);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // extern "C"
