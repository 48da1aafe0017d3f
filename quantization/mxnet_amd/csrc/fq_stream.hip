// libfakequant — streaming kernels: per-sample statistic, fake-quant apply, BatchNorm+activation+statistic, pooling, counters
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// K1: per-sample statistic.  x viewed as (n, inner).
// ---------------------------------------------------------------------------------------------------------------
template <bool USE_ABS, bool VEC, bool NTL, int U>
__global__ __launch_bounds__(kBlock) void absmax_per_sample_kernel(const float* __restrict__ x, int64_t inner,
                                                                   int chunks_per_sample, int64_t total_chunks,
                                                                   float* __restrict__ out_max) {
  constexpr int kCh = kBlock * kVec * U;
  __shared__ float red[4];
  const ChunkRange rg = block_range(total_chunks);
  int64_t cur_s = -1;
  float m = stat_init<USE_ABS>();
  for (int64_t c = rg.begin; c < rg.end; ++c) {
    const int64_t s = c / chunks_per_sample;
    if (s != cur_s) {
      if (cur_s >= 0) {
        m = block_max(m, red);
        if (threadIdx.x == 0) atomic_max_f32(out_max + cur_s, m);
      }
      cur_s = s;
      m = stat_init<USE_ABS>();
    }
    const int64_t off0 = (c - s * chunks_per_sample) * (int64_t)kCh;
    const float* base = x + s * inner + off0;
    const int64_t rem = inner - off0;
    if (VEC) {
      const f4* p = reinterpret_cast<const f4*>(base);
      if (rem >= kCh) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld4<NTL>(p + threadIdx.x + u * kBlock);
#pragma unroll
        for (int u = 0; u < U; ++u) m = fmaxf(m, stat4<USE_ABS>(v[u]));
      } else {
        const int nvec = (int)(rem / kVec);
        for (int i = threadIdx.x; i < nvec; i += kBlock) m = fmaxf(m, stat4<USE_ABS>(ld4<NTL>(p + i)));
      }
    } else {
      const int cnt = (int)(rem < kCh ? rem : kCh);
      for (int i = threadIdx.x; i < cnt; i += kBlock) m = fmaxf(m, stat_of<USE_ABS>(base[i]));
    }
  }
  if (cur_s >= 0) {
    m = block_max(m, red);
    if (threadIdx.x == 0) atomic_max_f32(out_max + cur_s, m);
  }
}

// K1b: mean of n floats (one thread; n is a batch size)
__global__ void batch_mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;                    // one whole wavefront
  const float m = batch_mean_dev(v, n);
  if (threadIdx.x == 0) out[0] = m;
}

__global__ void batch_mean_gathered_kernel(const float* __restrict__ packs, int world, int64_t stride,
                                           float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double acc = 0.0;
  long long total = 0;
  for (int w = 0; w < world; ++w) {
    const float* rec = packs + (int64_t)w * stride;
    const int c = (int)rec[0];
    for (int i = 0; i < c; ++i) acc += (double)rec[1 + i];
    total += c;
  }
  out[0] = (float)acc / (float)total;
}

__global__ void batch_mean_rows_kernel(const float* __restrict__ v, int64_t rows, int n, int64_t stride,
                                       float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) out[r] = batch_mean_seq(v + r * stride, n);
}

// K1c: the two ends of the calibration-step collective (dist.py): per layer the fp64 sum of its per-sample maxima (sample
// order) followed by the local sample count, and — after the ranks' records were summed — mean = fp32(sum) / fp32(count),
// i.e. the batch mean of K1b over the GLOBAL batch (fp64 partial sums of a few thousand fp32 values are exact unless their
// exponents spread over more than 2^20, the same condition batch_mean_dev accepts).
__global__ void stat_rows_sum_kernel(const float* __restrict__ v, int64_t rows, int n, int64_t stride,
                                     double* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += (double)v[r * stride + i];
    out[r] = acc;
  }
  if (r == rows) out[rows] = (double)n;
}

__global__ void mean_from_sums_kernel(const double* __restrict__ sums, int64_t rows, float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) out[r] = (float)sums[r] / (float)sums[rows];
}

// ---------------------------------------------------------------------------------------------------------------
// K2: apply.  ONLINE: threshold = mean of stat_in[0..n);  else threshold = thr[0].
//     STATS (offline only): also produce the per-sample statistic of x into stat_out (fused, same pass).
// ---------------------------------------------------------------------------------------------------------------
template <bool ONLINE, bool STATS, bool CODES, bool USE_ABS, bool VEC, bool NTL, bool NTS, int U>
__global__ __launch_bounds__(kBlock) void act_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int32_t* __restrict__ codes, int64_t inner,
                                                           int chunks_per_sample, int64_t total_chunks,
                                                           const float* __restrict__ stat_in, int n,
                                                           const float* __restrict__ thr, float levels,
                                                           int lo_neg_max, float eps, int reverse,
                                                           float* __restrict__ stat_out,
                                                           float* __restrict__ cur_max_out) {
  constexpr int kCh = kBlock * kVec * U;
  __shared__ float red[4];
  const float max_ = ONLINE ? batch_mean_dev(stat_in, n) : thr[0];
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  if (ONLINE && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;

  const ChunkRange rg = block_range(total_chunks);
  int64_t cur_s = -1;
  float m = stat_init<USE_ABS>();
  for (int64_t cc = rg.begin; cc < rg.end; ++cc) {
    const int64_t c = reverse ? (rg.end - 1 - (cc - rg.begin)) : cc;
    const int64_t s = c / chunks_per_sample;
    if (STATS && s != cur_s) {
      if (cur_s >= 0) {
        m = block_max(m, red);
        if (threadIdx.x == 0) atomic_max_f32(stat_out + cur_s, m);
      }
      cur_s = s;
      m = stat_init<USE_ABS>();
    }
    const int64_t off0 = (c - s * chunks_per_sample) * (int64_t)kCh;
    const int64_t gbase = s * inner + off0;
    const int64_t rem = inner - off0;
    if (VEC) {
      const f4* p = reinterpret_cast<const f4*>(x + gbase);
      f4* o = reinterpret_cast<f4*>(y + gbase);
      i4* oc = CODES ? reinterpret_cast<i4*>(codes + gbase) : nullptr;
      if (rem >= kCh) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld4<NTL>(p + threadIdx.x + u * kBlock);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (STATS) m = fmaxf(m, stat4<USE_ABS>(v[u]));
          const f4 k = fq_code4(v[u], q);
          if (CODES) oc[threadIdx.x + u * kBlock] = __builtin_convertvector(k, i4);
          st4<NTS>(o + threadIdx.x + u * kBlock, k * q.scale);
        }
      } else {
        const int nvec = (int)(rem / kVec);
        for (int i = threadIdx.x; i < nvec; i += kBlock) {
          const f4 v = ld4<NTL>(p + i);
          if (STATS) m = fmaxf(m, stat4<USE_ABS>(v));
          const f4 k = fq_code4(v, q);
          if (CODES) oc[i] = __builtin_convertvector(k, i4);
          st4<NTS>(o + i, k * q.scale);
        }
      }
    } else {
      const int cnt = (int)(rem < kCh ? rem : kCh);
      for (int i = threadIdx.x; i < cnt; i += kBlock) {
        const float v = x[gbase + i];
        if (STATS) m = fmaxf(m, stat_of<USE_ABS>(v));
        const float k = fq_code(v, q);
        if (CODES) codes[gbase + i] = (int)k;
        y[gbase + i] = k * q.scale;
      }
    }
  }
  if (STATS && cur_s >= 0) {
    m = block_max(m, red);
    if (threadIdx.x == 0) atomic_max_f32(stat_out + cur_s, m);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K2b: fused inference BatchNorm (as per-channel scale/shift) + activation + per-sample statistic of the OUTPUT.
// x is (n, c, hw).  One division per 16-byte access finds the channel; the walk to the next channel inside the access
// is incremental (hw need not be a multiple of 4: 7x7 planes).
// ---------------------------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ float bn_act1(float v, float sc, float sh) {
  float r = v * sc;
  r = r + sh;
  if (ACT == FQ_ACT_RELU) r = fmaxf(r, 0.0f);
  if (ACT == FQ_ACT_RELU6) r = fminf(fmaxf(r, 0.0f), 6.0f);
  return r;
}

// ---------------------------------------------------------------------------------------------------------------
// Histogram of a producer's OUTPUT in the producer's own pass (fq_bn_act_stat_hist / fq_add_act_stat_hist): while the KL
// calibration collects feature maps (distribution_calibrate.py:91-106) every quantised block's input is histogrammed once
// per batch - 4 B/elem read back right after the BatchNorm / residual pass wrote it.  From the second batch on the range is
// fixed (the first batch sets it, :97-101), so the producer bins what it stores.  Binning is K7's (fq_calib.hip), to the
// letter: clip to [0, max], zeros skipped, (int)(c * bins / (max + 1e-5)), index `bins` folded into the last bin; counts
// are integers, so where they are added up cannot matter.  LDS: one private copy per wavefront (4 * bins counters).
// ---------------------------------------------------------------------------------------------------------------
struct LdsHist {
  unsigned int* mine;
  float mx, scales;
  int last;
  unsigned int neg;
  __device__ __forceinline__ void init(unsigned int* lh, int bins, const float* __restrict__ max_dev) {
    for (int i = threadIdx.x; i < 4 * bins; i += kBlock) lh[i] = 0u;
    __syncthreads();
    mine = lh + (threadIdx.x >> 6) * bins;
    mx = max_dev[0];
    scales = (float)bins / (mx + 1e-5f);
    last = bins - 1;
    neg = 0u;
  }
  __device__ __forceinline__ void put(float v) {
    neg += (v < 0.0f) ? 1u : 0u;
    const float c = __builtin_amdgcn_fmed3f(v, 0.0f, mx);
    if (c != 0.0f) {
      int idx = (int)(c * scales);
      idx = idx < last ? idx : last;
      atomicAdd(&mine[idx], 1u);
    }
  }
  __device__ __forceinline__ void put4(const f4& q) {
    put(q.x);
    put(q.y);
    put(q.z);
    put(q.w);
  }
  __device__ __forceinline__ void flush(const unsigned int* lh, int bins, unsigned long long* __restrict__ hist,
                                        unsigned int* __restrict__ neg_count) {
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
};

// RES (round 6, fq_bn_add_act_stat): a residual operand of x's shape joins after BatchNorm and before the activation -
// act(fl(fl(x * scale) + shift) + res), the value fq_bn_act_stat (no activation) followed by fq_add_act_stat forms, in one
// pass: 12 B per element instead of 8 + 12 (the closing BatchNorm of a ResNet unit while quantisation is switched off: the
// KL calibration's collection forward, an fp32 evaluation)
template <int ACT, bool STATS, bool VEC, int U, bool HIST = false, bool RES = false>
__global__ __launch_bounds__(kBlock) void bn_act_stat_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int64_t inner, int hw, int chunks_per_sample,
                                                             int64_t total_chunks, const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ res,
                                                             float* __restrict__ stat_out,
                                                             const float* __restrict__ hist_max, int bins,
                                                             unsigned long long* __restrict__ hist,
                                                             unsigned int* __restrict__ neg_count) {
  constexpr int kCh = kBlock * kVec * U;
  __shared__ float red[4];
  extern __shared__ __attribute__((aligned(16))) unsigned int lds_hist[];   // HIST: 4 * bins counters
  LdsHist lh;
  if (HIST) lh.init(lds_hist, bins, hist_max);
  const ChunkRange rg = block_range(total_chunks);
  int64_t cur_s = -1;
  float m = 0.0f;
  for (int64_t c = rg.begin; c < rg.end; ++c) {
    const int64_t s = c / chunks_per_sample;
    if (STATS && s != cur_s) {
      if (cur_s >= 0) {
        m = block_max(m, red);
        if (threadIdx.x == 0) atomic_max_f32(stat_out + cur_s, m);
      }
      cur_s = s;
      m = 0.0f;
    }
    const int64_t off0 = (c - s * chunks_per_sample) * (int64_t)kCh;     // offset inside the sample
    const int64_t gbase = s * inner + off0;
    const int64_t rem = inner - off0;
    // (RES: BatchNorm without activation, + the residual, then the activation - each step rounded as the two passes round it)
    auto one = [&](float v, float sc, float sh, float r) __attribute__((always_inline)) {
      if (!RES) return bn_act1<ACT>(v, sc, sh);
      float t = bn_act1<FQ_ACT_NONE>(v, sc, sh) + r;
      if (ACT == FQ_ACT_RELU) t = fmaxf(t, 0.0f);
      if (ACT == FQ_ACT_RELU6) t = fminf(fmaxf(t, 0.0f), 6.0f);
      return t;
    };
    if (VEC) {
      const f4* p = reinterpret_cast<const f4*>(x + gbase);
      const f4* pr = reinterpret_cast<const f4*>((RES ? res : x) + gbase);
      f4* o = reinterpret_cast<f4*>(y + gbase);
      const int nvec = (int)((rem < kCh ? rem : kCh) / kVec);
      f4 v[U], rv[RES ? U : 1];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * kBlock;
        if (i < nvec) {
          v[u] = p[i];
          if (RES) rv[u] = pr[i];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * kBlock;
        if (i < nvec) {
          const unsigned e0 = (unsigned)(off0 + (int64_t)i * kVec);       // element index inside the sample (< 2^32)
          unsigned ch = e0 / (unsigned)hw;
          unsigned r = e0 - ch * (unsigned)hw;
          float sc = scale[ch], sh = shift[ch];
          const f4 rr = RES ? rv[RES ? u : 0] : (f4){0.f, 0.f, 0.f, 0.f};
          f4 q;
          q.x = one(v[u].x, sc, sh, rr.x);
          if (++r == (unsigned)hw) { r = 0; ++ch; sc = scale[ch]; sh = shift[ch]; }
          q.y = one(v[u].y, sc, sh, rr.y);
          if (++r == (unsigned)hw) { r = 0; ++ch; sc = scale[ch]; sh = shift[ch]; }
          q.z = one(v[u].z, sc, sh, rr.z);
          if (++r == (unsigned)hw) { r = 0; ++ch; sc = scale[ch]; sh = shift[ch]; }
          q.w = one(v[u].w, sc, sh, rr.w);
          if (STATS) m = fmaxf(m, stat4<true>(q));
          o[i] = q;
          if (HIST) lh.put4(q);
        }
      }
    } else {
      const int cnt = (int)(rem < kCh ? rem : kCh);
      for (int i = threadIdx.x; i < cnt; i += kBlock) {
        const unsigned e = (unsigned)(off0 + i);
        const unsigned ch = e / (unsigned)hw;
        const float q = one(x[gbase + i], scale[ch], shift[ch], RES ? res[gbase + i] : 0.0f);
        if (STATS) m = fmaxf(m, fabsf(q));
        y[gbase + i] = q;
        if (HIST) lh.put(q);
      }
    }
  }
  if (STATS && cur_s >= 0) {
    m = block_max(m, red);
    if (threadIdx.x == 0) atomic_max_f32(stat_out + cur_s, m);
  }
  if (HIST) lh.flush(lds_hist, bins, hist, neg_count);
}

// ---------------------------------------------------------------------------------------------------------------
// K2m: residual tail of a ResNet unit, `(x + residual).relu()` (gluon model zoo), with the per-sample statistic of the
// result for the quantised convolutions that consume it: y = act(a + b), stat_out[n] = max|y[n]|.  12 B/elem (two reads,
// one write) instead of add (12) + relu (8) + one statistic pass per consumer (4 each).  Flat streaming kernel over
// (n, inner) with the chunking of K1 / K2.
// ---------------------------------------------------------------------------------------------------------------
template <int ACT, bool STATS, bool VEC, int U, bool HIST = false>
__global__ __launch_bounds__(kBlock) void add_act_stat_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ y, int64_t inner,
                                                              int chunks_per_sample, int64_t total_chunks,
                                                              float* __restrict__ stat_out,
                                                              const float* __restrict__ hist_max, int bins,
                                                              unsigned long long* __restrict__ hist,
                                                              unsigned int* __restrict__ neg_count) {
  constexpr int kCh = kBlock * kVec * U;
  __shared__ float red[4];
  extern __shared__ __attribute__((aligned(16))) unsigned int lds_hist[];   // HIST: 4 * bins counters
  LdsHist lh;
  if (HIST) lh.init(lds_hist, bins, hist_max);
  const ChunkRange rg = block_range(total_chunks);
  int64_t cur_s = -1;
  float m = 0.0f;
  auto one = [&](float p, float r) __attribute__((always_inline)) {
    float v = p + r;
    if (ACT == FQ_ACT_RELU) v = fmaxf(v, 0.0f);
    if (ACT == FQ_ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
    return v;
  };
  for (int64_t c = rg.begin; c < rg.end; ++c) {
    const int64_t s = c / chunks_per_sample;
    if (STATS && s != cur_s) {
      if (cur_s >= 0) {
        m = block_max(m, red);
        if (threadIdx.x == 0) atomic_max_f32(stat_out + cur_s, m);
      }
      cur_s = s;
      m = 0.0f;
    }
    const int64_t off0 = (c - s * chunks_per_sample) * (int64_t)kCh;
    const int64_t gbase = s * inner + off0;
    const int64_t rem = inner - off0;
    if (VEC && rem >= kCh) {
      const f4* pa = reinterpret_cast<const f4*>(a + gbase);
      const f4* pb = reinterpret_cast<const f4*>(b + gbase);
      f4* o = reinterpret_cast<f4*>(y + gbase);
      f4 va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        va[u] = pa[threadIdx.x + u * kBlock];
        vb[u] = pb[threadIdx.x + u * kBlock];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        f4 q;
        q.x = one(va[u].x, vb[u].x);
        q.y = one(va[u].y, vb[u].y);
        q.z = one(va[u].z, vb[u].z);
        q.w = one(va[u].w, vb[u].w);
        if (STATS) m = fmaxf(m, stat4<true>(q));
        o[threadIdx.x + u * kBlock] = q;
        if (HIST) lh.put4(q);
      }
    } else {
      const int cnt = (int)(rem < kCh ? rem : kCh);
      for (int i = threadIdx.x; i < cnt; i += kBlock) {
        const float q = one(a[gbase + i], b[gbase + i]);
        if (STATS) m = fmaxf(m, fabsf(q));
        y[gbase + i] = q;
        if (HIST) lh.put(q);
      }
    }
  }
  if (STATS && cur_s >= 0) {
    m = block_max(m, red);
    if (threadIdx.x == 0) atomic_max_f32(stat_out + cur_s, m);
  }
  if (HIST) lh.flush(lds_hist, bins, hist, neg_count);
}

// ---------------------------------------------------------------------------------------------------------------
// K11: global average pooling (gluon GlobalAvgPool2D = F.Pooling(global_pool=True, pool_type='avg')) with the per-sample
// max|y| the following Dense layer's input quantiser needs (convert_dense.py:40-41) - one launch instead of the library
// reduction + memset + statistic pass.  y[n][c] = fp32(sum over the plane in fp64, in order) / fp32(hw): deterministic,
// within an ulp of any fp32 summation order.  A thread owns a plane (hw is 49 here: 25 MB in all, latency-bound).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gap_stat_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int64_t planes, int c, int hw, float* __restrict__ stat_out) {
  __shared__ float red[4];
  const int64_t pl = (int64_t)blockIdx.x * kBlock + threadIdx.x;       // host: c % kBlock == 0 or one sample per block
  float v = 0.0f;
  if (pl < planes) {
    const float* p = x + pl * hw;
    double acc = 0.0;
    for (int i = 0; i < hw; ++i) acc += (double)p[i];
    v = (float)acc / (float)hw;
    y[pl] = v;
  }
  if (stat_out != nullptr) {
    // all planes of a block belong to one sample when c % kBlock == 0 (host checks); otherwise per-thread atomics
    const int64_t first = (int64_t)blockIdx.x * kBlock;
    const bool one_sample = (c % kBlock) == 0;
    if (one_sample) {
      const float m = block_max(fabsf(v), red);
      if (threadIdx.x == 0 && first < planes) atomic_max_f32(stat_out + first / c, m);
    } else if (pl < planes) {
      atomic_max_f32(stat_out + pl / c, fabsf(v));
    }
  }
}

// The same result for the small planes this block actually sees (7x7, 8x8): a wavefront owns 64 consecutive planes = one
// contiguous run of 64 * hw floats, which it reads COALESCED into its quarter of an LDS tile (a lane reading its own plane
// straight from memory touches a different 128-byte line per lane and load: 30 us for 25.7 MB); lane l then adds up plane
// l from LDS in the same order as above.
//
// Statistic: the atomics are the expensive part of this short kernel.  Atomics to ONE 128-byte line are served one after the
// other, ~10 ns each, whichever XCD they come from (tools/atomic_probe.hip, profiles/r3_atomic_probe.txt), and the 128
// statistic slots of a batch are four lines: with one atomic per wavefront the 2048 of them kept this 9 us kernel alive for
// another 3 (sent when the wavefronts finish, i.e. all at the end; the long kernels spread theirs over their run time and
// show no such tail).  So the workgroup's four wavefronts meet in LDS when their 256 planes lie in one sample: 512 atomics.
// (Workgroup b works on slice (b % 8) * (grid / 8) + b / 8 - the workgroups of a sample on one XCD; measured neutral.)
template <int PPW>      // planes per wavefront: 64 or 32 (lanes >= PPW only load)
__global__ __launch_bounds__(kBlock) void gap_stat_lds_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              int64_t planes, int c, int hw, float* __restrict__ stat_out) {
  extern __shared__ __attribute__((aligned(16))) float gap_tile[];     // 4 wavefronts x PPW planes x hw
  __shared__ float wmax[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* mine = gap_tile + (size_t)wave * PPW * hw;
  const unsigned per_xcd = gridDim.x >> 3;                             // host: the grid is a multiple of 8
  const int64_t slice = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const int64_t wg0 = slice * 4 * PPW;                                 // first plane of this workgroup
  const int64_t p0 = wg0 + (int64_t)wave * PPW;                        // ... of this wavefront
  const int np = p0 >= planes ? 0 : (int)(planes - p0 < PPW ? planes - p0 : PPW);
  const float* src = x + p0 * hw;
  const int cnt = np * hw;
  if ((cnt & 3) == 0 && ((PPW * hw) & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
    // 16 bytes per lane and load, four loads in flight (a wavefront's run is PPW * hw floats from a 16-byte aligned start)
    const f4* s4 = reinterpret_cast<const f4*>(src);
    f4* m4 = reinterpret_cast<f4*>(mine);
    const int c4 = cnt >> 2;
    for (int i = lane; i < c4; i += 256) {
      f4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = s4[i + 64 * u < c4 ? i + 64 * u : i];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + 64 * u < c4) m4[i + 64 * u] = v[u];
    }
  } else {
    for (int i = lane; i < cnt; i += 64) mine[i] = src[i];             // wave-private tile: no barrier needed
  }
  float v = 0.0f;
  if (lane < np) {
    const float* p = mine + lane * hw;
    double acc = 0.0;
    for (int i = 0; i < hw; ++i) acc += (double)p[i];
    v = (float)acc / (float)hw;
    y[p0 + lane] = v;
  }
  if (stat_out == nullptr) return;
  const int64_t wg_last = (wg0 + 4 * PPW < planes ? wg0 + 4 * PPW : planes) - 1;
  if (wg0 < planes && wg0 / c == wg_last / c) {                          // the whole workgroup lies in one sample
    const float m = wave_max_nonneg(lane < np ? fabsf(v) : 0.0f);
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomic_max_f32(stat_out + wg0 / c, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
    return;
  }
  if (np == 0) return;
  const int64_t s0 = p0 / c, s1 = (p0 + np - 1) / c;
  if (s0 == s1) {
    const float m = wave_max_nonneg(lane < np ? fabsf(v) : 0.0f);
    if (lane == 0) atomic_max_f32(stat_out + s0, m);
  } else if (lane < np) {
    atomic_max_f32(stat_out + (p0 + lane) / c, fabsf(v));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K12: the evaluation counters of simulate_quantization.py:122-148 (pred = argmax(outputs, axis=1), first index on
// ties as MXNet's argmax; test_num_correct, label_counter[gt], correct_counter[gt]) in ONE launch: a wavefront per
// sample.  The tensor-library formulation is nine launch-bound kernels (~60 us per batch, 4 % of an evaluation step).
// counters = [n_correct, total, correct[classes], label[classes]] as floats: the increments are 1.0, exact below 2^24.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void eval_counters_kernel(const float* __restrict__ logits,
                                                               const long long* __restrict__ labels, int64_t n,
                                                               int classes, float* __restrict__ counters) {
  // counters[0] (correct) and counters[1] (total) are ONE address each for the whole batch: same-address global atomics
  // serialise in L2 (128 samples x 2 of them were most of this kernel's 11 us), so the workgroup's four samples are summed in
  // LDS first and one thread adds the two sums
  __shared__ unsigned wg_total, wg_correct;
  if (threadIdx.x == 0) {
    wg_total = 0u;
    wg_correct = 0u;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t smp_raw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const bool live = smp_raw < n;
  const int64_t smp = live ? smp_raw : n - 1;
  const float* row = logits + smp * classes;
  float best = 0.0f;
  int bidx = 0x7FFFFFFF;
  bool bnan = false;
  // better(a, b): NaN beats everything (as torch / numpy argmax), then the larger value, then the smaller index
  auto take = [&](float v, int i) {
    const bool vnan = v != v;
    const bool better = bidx == 0x7FFFFFFF || (vnan && !bnan) || (!bnan && !vnan && v > best) ||
                        (((vnan && bnan) || (!vnan && !bnan && v == best)) && i < bidx);
    if (better) {
      best = v;
      bidx = i;
      bnan = vnan;
    }
  };
  // eight independent loads in flight per lane (one load per iteration made this kernel sixteen dependent L2 round trips)
  for (int base = 0; base < classes; base += 64 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 64 + lane;
      v[u] = row[i < classes ? i : classes - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 64 + lane;
      if (i < classes) take(v[u], i);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bidx, off, 64);
    if (oi != 0x7FFFFFFF) take(ov, oi);
  }
  if (lane == 0 && live) {
    const long long gt = labels[smp];
    atomicAdd(&wg_total, 1u);
    if (gt >= 0 && gt < classes) {
      atomicAdd(counters + 2 + classes + gt, 1.0f);
      if ((long long)bidx == gt) {
        atomicAdd(&wg_correct, 1u);
        atomicAdd(counters + 2 + gt, 1.0f);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (wg_total) atomicAdd(counters + 1, (float)wg_total);          // small integers: exact in fp32
    if (wg_correct) atomicAdd(counters, (float)wg_correct);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K2p: BatchNorm + activation + 3x3 / stride 2 / pad 1 max pooling + per-sample statistic in one pass: the head of the
// ImageNet ResNets behind their (un-quantised, library) first convolution.  The three passes it replaces move 411 + 411
// (BN + ReLU), 411 + 103 (pooling) and 103 MB (the consumer's statistic) at batch 128; this one reads 411 and writes 103.
// A thread produces TWO adjacent outputs of one row from a 3 x 5 input window: per input row one 16-byte load (columns 4q ..
// 4q+3, coalesced along the row) and the left neighbour column 4q-1 (same cache line as the previous thread's load).
// Padding never wins a maximum (-inf); BatchNorm and the activation are applied to every input BEFORE the maximum, exactly
// as the separate passes do (a negative BatchNorm scale would otherwise turn the maximum around).
// ---------------------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(kBlock) void bn_act_maxpool_stat_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                     int C, int H, int W, int Ho, int Wo,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     float* __restrict__ stat_out) {
  __shared__ float red[4];
  const int smp = blockIdx.y;
  const int Wq = Wo >> 1;                                               // output pairs per row (host: W % 4 == 0)
  const int64_t items = (int64_t)C * Ho * Wq;
  const float* xs = x + (int64_t)smp * C * H * W;
  float* ys = y + (int64_t)smp * C * Ho * Wo;
  float m = 0.0f;
  for (int64_t it = (int64_t)blockIdx.x * kBlock + threadIdx.x; it < items; it += (int64_t)gridDim.x * kBlock) {
    const int wq = (int)(it % Wq);
    const int64_t pr = it / Wq;
    const int ho = (int)(pr % Ho);
    const int ch = (int)(pr / Ho);
    const float sc = scale[ch], sh = shift[ch];
    const float* xp = xs + (int64_t)ch * H * W + 4 * wq;
    float o0 = -INFINITY, o1 = -INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = 2 * ho - 1 + k;
      const bool rin = r >= 0 && r < H;
      const int rc = r < 0 ? 0 : (r < H ? r : H - 1);
      const f4 v = *reinterpret_cast<const f4*>(xp + (int64_t)rc * W);
      const float lraw = xp[(int64_t)rc * W - (wq > 0 ? 1 : 0)];        // column 4q-1 (q = 0: padding, masked below)
      auto f = [&](float t) {
        t = t * sc;
        t = t + sh;
        return act_rt(t, ACT);
      };
      const float a0 = f(v.x), a1 = f(v.y), a2 = f(v.z), a3 = f(v.w);
      const float l = wq > 0 ? f(lraw) : -INFINITY;
      const float r0 = fmaxf(fmaxf(l, a0), a1), r1 = fmaxf(fmaxf(a1, a2), a3);
      o0 = rin ? fmaxf(o0, r0) : o0;
      o1 = rin ? fmaxf(o1, r1) : o1;
    }
    *reinterpret_cast<float2*>(ys + ((int64_t)ch * Ho + ho) * Wo + 2 * wq) = make_float2(o0, o1);
    m = fmaxf(m, fmaxf(fabsf(o0), fabsf(o1)));
  }
  if (stat_out != nullptr) {
    m = block_max(m, red);
    if (threadIdx.x == 0) atomic_max_f32(stat_out + smp, m);
  }
}

}  // namespace

namespace fqi {

int launch_absmax(const float* x, int64_t n, int64_t inner, bool use_abs, float* out, hipStream_t st) {
  // caller has initialised `out` (0 for |x|, -inf otherwise)
  const bool small = use_small_chunks(n, inner);
  const Chunking ck = chunking(n, inner, small ? kSmallChunk : kChunk);
  const bool vec = (inner % kVec == 0) && aligned16(x);
  const int grid = grid_for(ck.total);
  const bool ntl = (stream_policy(FQ_KERNEL_STAT, n * inner) & kPolNtLoad) != 0;
  ProfScope prof(FQ_KERNEL_STAT, 4.0 * (double)n * (double)inner, st);
#define FQ_ABSMAX(A, V, L, UU)                                                                                \
  hipLaunchKernelGGL((absmax_per_sample_kernel<A, V, L, UU>), dim3(grid), dim3(kBlock), 0, st, x, inner,      \
                     ck.chunks_per_sample, ck.total, out)
#define FQ_ABSMAX_U(A, V, L)                                            \
  do {                                                                  \
    if (small) FQ_ABSMAX(A, V, L, kSmallUnroll); else FQ_ABSMAX(A, V, L, kUnroll); \
  } while (0)
  if (use_abs) {
    if (vec) {
      if (ntl) FQ_ABSMAX_U(true, true, true); else FQ_ABSMAX_U(true, true, false);
    } else {
      FQ_ABSMAX_U(true, false, false);
    }
  } else {
    if (vec) FQ_ABSMAX_U(false, true, false); else FQ_ABSMAX_U(false, false, false);
  }
#undef FQ_ABSMAX_U
#undef FQ_ABSMAX
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int init_stat(float* p, int64_t n, bool use_abs, hipStream_t st) {
  if (use_abs) {
    FQ_HIP(hipMemsetAsync(p, 0, n * sizeof(float), st));
  } else {
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, -INFINITY);
    FQ_LAUNCH_CHECK();
  }
  return FQ_OK;
}

}  // namespace fqi

namespace {

template <bool ONLINE, bool STATS, bool CODES>
int launch_apply(const float* x, float* y, int32_t* codes, int64_t n, int64_t inner, const float* stat_in,
                 const float* thr, float levels, unsigned flags, float* stat_out, float* cur_out, hipStream_t st) {
  const bool small = use_small_chunks(n, inner);
  const Chunking ck = chunking(n, inner, small ? kSmallChunk : kChunk);
  const bool vec = (inner % kVec == 0) && aligned16(x) && aligned16(y) && (!CODES || aligned16(codes));
  const bool use_abs = !(flags & FQ_ACT_NO_ABS);
  const int grid = grid_for(ck.total);
  const int lo_neg = (flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
  const float eps = (flags & FQ_ACT_NO_EPS) ? 0.0f : kEps;
  int pol = stream_policy(ONLINE ? FQ_KERNEL_APPLY_ONLINE : FQ_KERNEL_APPLY_OFFLINE, n * inner);
  if (x == y) pol &= ~kPolNtLoad;
  const int reverse = (pol & kPolReverse) ? 1 : 0;
  ProfScope prof(ONLINE ? FQ_KERNEL_APPLY_ONLINE : FQ_KERNEL_APPLY_OFFLINE, 8.0 * (double)n * (double)inner, st);
#define FQ_APPLY(A, V, L, S, UU)                                                                                 \
  hipLaunchKernelGGL((act_apply_kernel<ONLINE, STATS, CODES, A, V, L, S, UU>), dim3(grid), dim3(kBlock), 0, st,  \
                     x, y, codes, inner, ck.chunks_per_sample, ck.total, stat_in, (int)n, thr, levels, lo_neg,   \
                     eps, reverse, stat_out, cur_out)
#define FQ_APPLY_U(A, V, L, S)                                                      \
  do {                                                                              \
    if (small) FQ_APPLY(A, V, L, S, kSmallUnroll); else FQ_APPLY(A, V, L, S, kUnroll); \
  } while (0)
  if (use_abs && vec && !CODES) {
    switch (pol & (kPolNtLoad | kPolNtStore)) {
      case 0: FQ_APPLY_U(true, true, false, false); break;
      case kPolNtLoad: FQ_APPLY_U(true, true, true, false); break;
      case kPolNtStore: FQ_APPLY_U(true, true, false, true); break;
      default: FQ_APPLY_U(true, true, true, true); break;
    }
  } else if (use_abs) {
    if (vec) FQ_APPLY_U(true, true, false, false); else FQ_APPLY_U(true, false, false, false);
  } else {
    if (vec) FQ_APPLY_U(false, true, false, false); else FQ_APPLY_U(false, false, false, false);
  }
#undef FQ_APPLY_U
#undef FQ_APPLY
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // namespace

extern "C" {

size_t fq_act_workspace_bytes(int64_t n) { return (size_t)(n < 1 ? 1 : n) * sizeof(float) * 2 + 64; }

int fq_absmax_per_sample(const float* x, int64_t n, int64_t inner, unsigned flags, float* out_max,
                         fqStream_t stream) {
  FQ_REQUIRE(x && out_max, "fq_absmax_per_sample: null pointer");
  FQ_REQUIRE(n > 0 && inner > 0, "fq_absmax_per_sample: empty tensor (n=%lld inner=%lld)", (long long)n,
             (long long)inner);
  hipStream_t st = (hipStream_t)stream;
  const bool use_abs = !(flags & FQ_ACT_NO_ABS);
  if (int rc = init_stat(out_max, n, use_abs, st)) return rc;
  return launch_absmax(x, n, inner, use_abs, out_max, st);
}

int fq_batch_mean(const float* v, int64_t n, float* out, fqStream_t stream) {
  FQ_REQUIRE(v && out, "fq_batch_mean: null pointer");
  FQ_REQUIRE(n > 0 && n < (1ll << 31), "fq_batch_mean: bad n=%lld", (long long)n);
  hipLaunchKernelGGL(batch_mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, v, (int)n, out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_batch_mean_gathered(const float* packs, int world, int64_t stride, float* out, fqStream_t stream) {
  FQ_REQUIRE(packs && out, "fq_batch_mean_gathered: null pointer");
  FQ_REQUIRE(world > 0 && stride > 1, "fq_batch_mean_gathered: bad shape");
  hipLaunchKernelGGL(batch_mean_gathered_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, packs, world, stride, out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_batch_mean_rows(const float* v, int64_t rows, int64_t n, int64_t row_stride, float* out, fqStream_t stream) {
  FQ_REQUIRE(v && out, "fq_batch_mean_rows: null pointer");
  FQ_REQUIRE(rows > 0 && n > 0 && n < (1ll << 31) && row_stride >= n, "fq_batch_mean_rows: bad shape");
  hipLaunchKernelGGL(batch_mean_rows_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, (hipStream_t)stream, v,
                     rows, (int)n, row_stride, out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_stat_rows_sum(const float* v, int64_t rows, int64_t n, int64_t row_stride, double* out, fqStream_t stream) {
  FQ_REQUIRE(v && out, "fq_stat_rows_sum: null pointer");
  FQ_REQUIRE(rows > 0 && n >= 0 && n < (1ll << 31) && row_stride >= n, "fq_stat_rows_sum: bad shape");
  hipLaunchKernelGGL(stat_rows_sum_kernel, dim3((unsigned)((rows + 1 + 63) / 64)), dim3(64), 0, (hipStream_t)stream, v,
                     rows, (int)n, row_stride, out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_mean_from_sums(const double* sums, int64_t rows, float* out, fqStream_t stream) {
  FQ_REQUIRE(sums && out && rows > 0, "fq_mean_from_sums: bad arguments");
  hipLaunchKernelGGL(mean_from_sums_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, (hipStream_t)stream, sums,
                     rows, out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_fake_quant_online(const float* x, float* y, int64_t n, int64_t inner, int width, unsigned flags,
                         float* out_current_max, int32_t* codes, void* ws, fqStream_t stream) {
  FQ_REQUIRE(x && y && ws, "fq_fake_quant_online: null pointer");
  FQ_REQUIRE(n > 0 && inner > 0 && n < (1ll << 31), "fq_fake_quant_online: bad shape (n=%lld inner=%lld)",
             (long long)n, (long long)inner);
  FQ_REQUIRE(width >= 2 && width <= 16, "fq_fake_quant_online: width %d out of range", width);
  hipStream_t st = (hipStream_t)stream;
  float* stat = (float*)ws;
  const bool use_abs = !(flags & FQ_ACT_NO_ABS);
  if (int rc = init_stat(stat, n, use_abs, st)) return rc;
  if (int rc = launch_absmax(x, n, inner, use_abs, stat, st)) return rc;
  const float levels = act_levels(width, flags);
  if (codes)
    return launch_apply<true, false, true>(x, y, codes, n, inner, stat, nullptr, levels, flags, nullptr,
                                           out_current_max, st);
  return launch_apply<true, false, false>(x, y, nullptr, n, inner, stat, nullptr, levels, flags, nullptr,
                                          out_current_max, st);
}

int fq_fake_quant_online_prestat(const float* x, float* y, int64_t n, int64_t inner, const float* stat, int width,
                                 unsigned flags, float* out_current_max, int32_t* codes, fqStream_t stream) {
  FQ_REQUIRE(x && y && stat, "fq_fake_quant_online_prestat: null pointer");
  FQ_REQUIRE(n > 0 && inner > 0 && n < (1ll << 31), "fq_fake_quant_online_prestat: bad shape (n=%lld inner=%lld)",
             (long long)n, (long long)inner);
  FQ_REQUIRE(width >= 2 && width <= 16, "fq_fake_quant_online_prestat: width %d out of range", width);
  const float levels = act_levels(width, flags);
  if (codes)
    return launch_apply<true, false, true>(x, y, codes, n, inner, stat, nullptr, levels, flags, nullptr,
                                           out_current_max, (hipStream_t)stream);
  return launch_apply<true, false, false>(x, y, nullptr, n, inner, stat, nullptr, levels, flags, nullptr,
                                          out_current_max, (hipStream_t)stream);
}

int fq_bn_act_maxpool_stat(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w, const float* scale,
                           const float* shift, int act, float* stat_out, fqStream_t stream) {
  FQ_REQUIRE(x && y && scale && shift, "fq_bn_act_maxpool_stat: null pointer");
  FQ_REQUIRE(n > 0 && n < 65536 && c > 0 && h > 0 && w > 0 && w % 4 == 0 && c * h * w < (1ll << 31),
             "fq_bn_act_maxpool_stat: bad shape (n=%lld c=%lld h=%lld w=%lld; w must be a multiple of 4)", (long long)n,
             (long long)c, (long long)h, (long long)w);
  FQ_REQUIRE(aligned16(x) && aligned16(y), "fq_bn_act_maxpool_stat: x and y must be 16-byte aligned");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_bn_act_maxpool_stat: unknown activation %d", act);
  hipStream_t st = (hipStream_t)stream;
  const int Ho = (int)((h - 1) / 2 + 1), Wo = (int)(w / 2);
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  const int64_t items = c * Ho * (Wo / 2);
  int64_t bx = (items + kBlock - 1) / kBlock;
  const int64_t cap = ((int64_t)num_cu() * kMaxBlocksPerCU + n - 1) / n;     // all workgroups resident, few atomics per sample
  if (bx > cap) bx = cap < 1 ? 1 : cap;
  ProfScope prof(FQ_KERNEL_BN_ACT, 4.0 * ((double)n * c * h * w + (double)n * c * Ho * Wo), st);
  const dim3 grid((unsigned)bx, (unsigned)n);
#define FQ_BMP(A)                                                                                                  \
  hipLaunchKernelGGL((bn_act_maxpool_stat_kernel<A>), grid, dim3(kBlock), 0, st, x, y, (int)c, (int)h, (int)w, Ho, Wo, \
                     scale, shift, stat_out)
  if (act == FQ_ACT_RELU) FQ_BMP(FQ_ACT_RELU);
  else if (act == FQ_ACT_RELU6) FQ_BMP(FQ_ACT_RELU6);
  else FQ_BMP(FQ_ACT_NONE);
#undef FQ_BMP
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

// the grid of a producer that also bins its output: every workgroup ends with up to `bins` 64-bit atomics (K7)
static int hist_grid(int64_t work_items) {
  static const int wg_per_cu = env_int("FQ_HISTF_WG_PER_CU", 2);
  const int64_t cap = (int64_t)num_cu() * wg_per_cu;
  const int64_t g = work_items < cap ? work_items : cap;
  return (int)(g < 1 ? 1 : g);
}

static int bn_act_launch(const float* x, float* y, int64_t n, int64_t c, int64_t hw, const float* scale,
                         const float* shift, int act, float* stat_out, const float* hist_max, int bins, uint64_t* hist_,
                         uint32_t* neg_count_, fqStream_t stream, const float* res = nullptr) {
  unsigned long long* hist = (unsigned long long*)hist_;
  unsigned int* neg_count = (unsigned int*)neg_count_;
  FQ_REQUIRE(x && y && scale && shift, "fq_bn_act_stat: null pointer");
  FQ_REQUIRE(n > 0 && c > 0 && hw > 0 && c * hw < (1ll << 32) && hw < (1ll << 31),
             "fq_bn_act_stat: bad shape (n=%lld c=%lld hw=%lld)", (long long)n, (long long)c, (long long)hw);
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_bn_act_stat: unknown activation %d", act);
  hipStream_t st = (hipStream_t)stream;
  const int64_t inner = c * hw;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  const bool small = use_small_chunks(n, inner);
  const Chunking ck = chunking(n, inner, small ? kSmallChunk : kChunk);
  const bool vec = (inner % kVec == 0) && aligned16(x) && aligned16(y) && (res == nullptr || aligned16(res));
  const bool with_hist = hist != nullptr;
  const int grid = with_hist ? hist_grid(ck.total) : grid_for(ck.total);
  const size_t lds = with_hist ? (size_t)4 * bins * sizeof(unsigned int) : 0;
  ProfScope prof(FQ_KERNEL_BN_ACT, (res ? 12.0 : 8.0) * (double)n * (double)inner, st);
  // (the residual form exists with the statistic only: its one caller always wants it)
#define FQ_BN_H(A, S, V, UU, H)                                                                                  \
  do {                                                                                                           \
    if (res != nullptr) {                                                                                        \
      if (S)                                                                                                     \
        hipLaunchKernelGGL((bn_act_stat_kernel<A, true, V, UU, H, true>), dim3(grid), dim3(kBlock), lds, st, x, y, inner, \
                           (int)hw, ck.chunks_per_sample, ck.total, scale, shift, res, stat_out, hist_max, bins, hist,    \
                           neg_count);                                                                           \
    } else {                                                                                                     \
      hipLaunchKernelGGL((bn_act_stat_kernel<A, S, V, UU, H>), dim3(grid), dim3(kBlock), lds, st, x, y, inner, (int)hw, \
                         ck.chunks_per_sample, ck.total, scale, shift, res, stat_out, hist_max, bins, hist, neg_count);  \
    }                                                                                                            \
  } while (0)
#define FQ_BN(A, S, V, UU)                                                  \
  do {                                                                      \
    if (with_hist) { if (S) FQ_BN_H(A, true, V, UU, true); } else FQ_BN_H(A, S, V, UU, false); \
  } while (0)
#define FQ_BN_U(A, S, V)                                                 \
  do {                                                                   \
    if (small) FQ_BN(A, S, V, kSmallUnroll); else FQ_BN(A, S, V, kUnroll); \
  } while (0)
#define FQ_BN_V(A, S)                                   \
  do {                                                  \
    if (vec) FQ_BN_U(A, S, true); else FQ_BN_U(A, S, false); \
  } while (0)
#define FQ_BN_S(A)                                         \
  do {                                                     \
    if (stat_out) FQ_BN_V(A, true); else FQ_BN_V(A, false); \
  } while (0)
  if (act == FQ_ACT_RELU) FQ_BN_S(FQ_ACT_RELU);
  else if (act == FQ_ACT_RELU6) FQ_BN_S(FQ_ACT_RELU6);
  else FQ_BN_S(FQ_ACT_NONE);
#undef FQ_BN_S
#undef FQ_BN_V
#undef FQ_BN_U
#undef FQ_BN
#undef FQ_BN_H
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_bn_act_stat(const float* x, float* y, int64_t n, int64_t c, int64_t hw, const float* scale,
                   const float* shift, int act, float* stat_out, fqStream_t stream) {
  return bn_act_launch(x, y, n, c, hw, scale, shift, act, stat_out, nullptr, 0, nullptr, nullptr, stream);
}

int fq_bn_act_stat_hist(const float* x, float* y, int64_t n, int64_t c, int64_t hw, const float* scale,
                        const float* shift, int act, float* stat_out, const float* hist_max, int bins, uint64_t* hist,
                        uint32_t* neg_count, fqStream_t stream) {
  FQ_REQUIRE(stat_out && hist_max && hist, "fq_bn_act_stat_hist: null pointer (the statistic is part of this form)");
  FQ_REQUIRE(bins > 0 && bins <= 4096, "fq_bn_act_stat_hist: bins=%d out of range (1..4096: four private copies in 64 KiB of LDS)", bins);
  return bn_act_launch(x, y, n, c, hw, scale, shift, act, stat_out, hist_max, bins, hist, neg_count, stream);
}

int fq_bn_add_act_stat(const float* x, const float* residual, float* y, int64_t n, int64_t c, int64_t hw, const float* scale,
                       const float* shift, int act, float* stat_out, fqStream_t stream) {
  FQ_REQUIRE(residual && stat_out, "fq_bn_add_act_stat: null pointer (residual, stat_out: the statistic is part of this form)");
  return bn_act_launch(x, y, n, c, hw, scale, shift, act, stat_out, nullptr, 0, nullptr, nullptr, stream, residual);
}

int fq_bn_add_act_stat_hist(const float* x, const float* residual, float* y, int64_t n, int64_t c, int64_t hw,
                            const float* scale, const float* shift, int act, float* stat_out, const float* hist_max, int bins,
                            uint64_t* hist, uint32_t* neg_count, fqStream_t stream) {
  FQ_REQUIRE(residual && stat_out && hist_max && hist, "fq_bn_add_act_stat_hist: null pointer");
  FQ_REQUIRE(bins > 0 && bins <= 4096, "fq_bn_add_act_stat_hist: bins=%d out of range (1..4096)", bins);
  return bn_act_launch(x, y, n, c, hw, scale, shift, act, stat_out, hist_max, bins, hist, neg_count, stream, residual);
}

static int add_act_launch(const float* a, const float* b, float* y, int64_t n, int64_t inner, int act, float* stat_out,
                          const float* hist_max, int bins, uint64_t* hist_, uint32_t* neg_count_, fqStream_t stream) {
  unsigned long long* hist = (unsigned long long*)hist_;
  unsigned int* neg_count = (unsigned int*)neg_count_;
  FQ_REQUIRE(a && b && y, "fq_add_act_stat: null pointer");
  FQ_REQUIRE(n > 0 && inner > 0 && n < (1ll << 31), "fq_add_act_stat: bad shape (n=%lld inner=%lld)", (long long)n,
             (long long)inner);
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_add_act_stat: unknown activation %d", act);
  hipStream_t st = (hipStream_t)stream;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  const bool small = use_small_chunks(n, inner);
  const Chunking ck = chunking(n, inner, small ? kSmallChunk : kChunk);
  const bool vec = (inner % kVec == 0) && aligned16(a) && aligned16(b) && aligned16(y);
  const bool with_hist = hist != nullptr;
  const int grid = with_hist ? hist_grid(ck.total) : grid_for(ck.total);
  const size_t lds = with_hist ? (size_t)4 * bins * sizeof(unsigned int) : 0;
  ProfScope prof(FQ_KERNEL_BN_ACT, 12.0 * (double)n * (double)inner, st);
#define FQ_ADD_H(A, S, V, UU, H)                                                                                  \
  hipLaunchKernelGGL((add_act_stat_kernel<A, S, V, UU, H>), dim3(grid), dim3(kBlock), lds, st, a, b, y, inner,    \
                     ck.chunks_per_sample, ck.total, stat_out, hist_max, bins, hist, neg_count)
#define FQ_ADD(A, S, V, UU)                                                   \
  do {                                                                        \
    if (with_hist) { if (S) FQ_ADD_H(A, true, V, UU, true); } else FQ_ADD_H(A, S, V, UU, false); \
  } while (0)
#define FQ_ADD_U(A, S, V)                                                 \
  do {                                                                    \
    if (small) FQ_ADD(A, S, V, kSmallUnroll); else FQ_ADD(A, S, V, kUnroll); \
  } while (0)
#define FQ_ADD_V(A, S)                                    \
  do {                                                    \
    if (vec) FQ_ADD_U(A, S, true); else FQ_ADD_U(A, S, false); \
  } while (0)
#define FQ_ADD_S(A)                                          \
  do {                                                       \
    if (stat_out) FQ_ADD_V(A, true); else FQ_ADD_V(A, false); \
  } while (0)
  if (act == FQ_ACT_RELU) FQ_ADD_S(FQ_ACT_RELU);
  else if (act == FQ_ACT_RELU6) FQ_ADD_S(FQ_ACT_RELU6);
  else FQ_ADD_S(FQ_ACT_NONE);
#undef FQ_ADD_S
#undef FQ_ADD_V
#undef FQ_ADD_U
#undef FQ_ADD
#undef FQ_ADD_H
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_add_act_stat(const float* a, const float* b, float* y, int64_t n, int64_t inner, int act, float* stat_out,
                    fqStream_t stream) {
  return add_act_launch(a, b, y, n, inner, act, stat_out, nullptr, 0, nullptr, nullptr, stream);
}

int fq_add_act_stat_hist(const float* a, const float* b, float* y, int64_t n, int64_t inner, int act, float* stat_out,
                         const float* hist_max, int bins, uint64_t* hist, uint32_t* neg_count, fqStream_t stream) {
  FQ_REQUIRE(stat_out && hist_max && hist, "fq_add_act_stat_hist: null pointer (the statistic is part of this form)");
  FQ_REQUIRE(bins > 0 && bins <= 4096, "fq_add_act_stat_hist: bins=%d out of range (1..4096: four private copies in 64 KiB of LDS)", bins);
  return add_act_launch(a, b, y, n, inner, act, stat_out, hist_max, bins, hist, neg_count, stream);
}

int fq_fake_quant_offline(const float* x, float* y, int64_t n, int64_t inner, const float* threshold, int width,
                          unsigned flags, float* out_current_max, int32_t* codes, void* ws, fqStream_t stream) {
  FQ_REQUIRE(x && y && threshold, "fq_fake_quant_offline: null pointer");
  FQ_REQUIRE(n > 0 && inner > 0 && n < (1ll << 31), "fq_fake_quant_offline: bad shape (n=%lld inner=%lld)",
             (long long)n, (long long)inner);
  FQ_REQUIRE(width >= 2 && width <= 16, "fq_fake_quant_offline: width %d out of range", width);
  hipStream_t st = (hipStream_t)stream;
  const float levels = act_levels(width, flags);
  if (out_current_max == nullptr) {
    if (codes)
      return launch_apply<false, false, true>(x, y, codes, n, inner, nullptr, threshold, levels, flags, nullptr,
                                              nullptr, st);
    return launch_apply<false, false, false>(x, y, nullptr, n, inner, nullptr, threshold, levels, flags, nullptr,
                                             nullptr, st);
  }
  FQ_REQUIRE(ws, "fq_fake_quant_offline: workspace required when out_current_max is requested");
  float* stat = (float*)ws;
  const bool use_abs = !(flags & FQ_ACT_NO_ABS);
  if (int rc = init_stat(stat, n, use_abs, st)) return rc;
  int rc;
  if (codes)
    rc = launch_apply<false, true, true>(x, y, codes, n, inner, nullptr, threshold, levels, flags, stat, nullptr, st);
  else
    rc = launch_apply<false, true, false>(x, y, nullptr, n, inner, nullptr, threshold, levels, flags, stat, nullptr,
                                          st);
  if (rc) return rc;
  hipLaunchKernelGGL(batch_mean_kernel, dim3(1), dim3(64), 0, st, stat, (int)n, out_current_max);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_global_avg_pool_stat(const float* x, float* y, int64_t n, int64_t c, int64_t hw, int flags, float* stat_out,
                            fqStream_t stream) {
  FQ_REQUIRE(x && y, "fq_global_avg_pool_stat: null pointer");
  FQ_REQUIRE(n > 0 && c > 0 && hw > 0 && hw < (1ll << 31) && c < (1ll << 31), "fq_global_avg_pool_stat: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const bool prezeroed = (flags & FQ_STAT_PREZEROED) != 0;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  const int64_t planes = n * c;
  ProfScope prof(FQ_KERNEL_POOL, 4.0 * ((double)planes * hw + (double)planes), st);
  if (hw <= 64) {                                                       // 64 KB of LDS at most
    static const int ppw_env = env_int("FQ_GAP_PPW", 64);               // tuning: 32 = twice the wavefronts
    const int ppw = ppw_env == 32 ? 32 : 64;
    const int64_t per_wg = 4 * ppw;
    const dim3 grid((unsigned)(((planes + per_wg - 1) / per_wg + 7) / 8 * 8));     // whole rounds over the 8 XCDs
    const size_t lds = (size_t)per_wg * hw * sizeof(float);
    if (ppw == 64)
      hipLaunchKernelGGL(gap_stat_lds_kernel<64>, grid, dim3(kBlock), lds, st, x, y, planes, (int)c, (int)hw, stat_out);
    else
      hipLaunchKernelGGL(gap_stat_lds_kernel<32>, grid, dim3(kBlock), lds, st, x, y, planes, (int)c, (int)hw, stat_out);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
  }
  hipLaunchKernelGGL(gap_stat_kernel, dim3((unsigned)((planes + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, x, y, planes,
                     (int)c, (int)hw, stat_out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_eval_counters(const float* logits, const int64_t* labels, int64_t n, int64_t classes, float* counters,
                     fqStream_t stream) {
  FQ_REQUIRE(logits && labels && counters, "fq_eval_counters: null pointer");
  FQ_REQUIRE(n > 0 && classes > 0 && classes < (1ll << 30) && n < (1ll << 31), "fq_eval_counters: bad shape");
  const int64_t grid = (n + (kBlock / 64) - 1) / (kBlock / 64);
  hipLaunchKernelGGL(eval_counters_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, logits,
                     (const long long*)labels, n, (int)classes, counters);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // extern "C"
