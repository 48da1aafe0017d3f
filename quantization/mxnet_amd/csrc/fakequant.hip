// libfakequant — hand-written HIP kernels for MI355X (gfx950 / CDNA4) + the C ABI of include/fakequant.h.
//
// Contents (search for the tag):
//   K0  fill (event-overhead calibration)          K1  per-sample abs-max statistic        K1b batch means
//   K2  fake-quant apply (online / offline)        K2b BatchNorm + activation + statistic
//   K2c/K2d/K2e depthwise 3x3 with quantise-on-load: LDS tiles / 1 column per lane / 4 columns per lane
//   K2f pointwise 1x1 on int8 codes, two kernels (quantise+transpose, 16x16x64 MFMA GEMM)   K2g LDS-panel single launch
//   K2h streaming form (32x32x32 MFMA, activations as B operand, weights in LDS)
//   K2i chunked-weights streaming form             K2j tile form (split quantisation, weights streamed from L2)
//   K2s stem convolution 3x3 s2 (3 -> 32)          K3/K3b/K3c weight fake-quant, generic STE   K4 Winograd-domain weights
//   K5  EMA    K6 global min/max    K7 histogram   K8 KL threshold search    K9 int-code quantise / dequantise
//   K10 exact int8 x int8 -> int32 GEMM (nn.Conv2D(quantized=True))          K11 global average pool + statistic
//   K12 evaluation counters                        then: host-side launch helpers and the extern "C" entry points.
//
// Design rules applied:
//   * 64-wide wavefronts, 256-thread workgroups, 16 B per lane per access where the layout allows; streaming kernels keep
//     8 independent loads in flight per lane and cap the grid at 8 workgroups per CU with contiguous work ranges;
//   * per-sample statistics: lane-local max -> wavefront shuffle tree -> LDS -> ONE integer atomicMax per workgroup and
//     sample (|x| >= 0, so the fp32 bit pattern orders like an unsigned int; same-address global atomics serialise in L2);
//   * the batch statistic never leaves the device: every consumer re-derives mean -> scale in its prologue from the N
//     per-sample maxima (wave-parallel fp64 sum, accepted only when the exponent spread proves every order exact);
//   * arithmetic that decides an integer code is exactly the reference's clip -> IEEE fp32 divide -> roundf -> multiply by
//     the epsilon-free scale, computed as v_med3 clamp, (float)((double)c * RN_f64(1/d)) (proven equal to the fp32
//     quotient, see ieee_div_by) and trunc(Q + copysign(pred(0.5), Q)) (checked exhaustively); compiled with
//     -ffp-contract=off and without fast-math so nothing is fused or re-associated behind the oracle's back;
//   * the 1x1 convolutions multiply the integer CODES on the int8 matrix cores (exact int32 sums); everything else is
//     elementwise / reduction / small-stencil work bounded by HBM or by instruction issue (profiles/r1_pmc_sq.txt);
//   * hipcc's scheduler is kept honest in the hand-pipelined loops with FQ_PIN (asm memory clobber + sched_barrier) and
//     empty "+v" asm pins (it otherwise sinks arithmetic below prefetches or hoists every load of an unrolled loop).
//
// Reference lines each kernel replaces are cited at its C entry point in include/fakequant.h.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "fakequant.h"

namespace {

constexpr int kBlock = 256;                       // 4 wavefronts
constexpr int kVec = 4;                           // floats per lane per access (16 B)
constexpr int kUnroll = 8;                        // independent 16 B accesses in flight per lane
constexpr int kChunk = kBlock * kVec * kUnroll;   // 8192 floats = 32 KiB per workgroup step
constexpr int kMaxBlocksPerCU = 8;
constexpr float kEps = 1e-10f;                    // ste_func.py:39,41

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define FQ_HIP(expr)                                                                             \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) return fail(FQ_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));       \
  } while (0)
#define FQ_REQUIRE(cond, ...)                                 \
  do {                                                        \
    if (!(cond)) return fail(FQ_ERR_INVALID, __VA_ARGS__);    \
  } while (0)
#define FQ_LAUNCH_CHECK() FQ_HIP(hipGetLastError())

// ---- optional per-kernel event timing (fq_profile_*) -----------------------------------------------------------
struct ProfRec {
  int kid;
  double bytes;
  hipEvent_t a, b;
};
bool g_prof_on = false;
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;

struct ProfScope {
  bool on;
  ProfRec r;
  hipStream_t st;
  ProfScope(int kid, double bytes, hipStream_t s) : on(g_prof_on), st(s) {
    if (!on) return;
    r.kid = kid;
    r.bytes = bytes;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
      on = false;
      return;
    }
    (void)hipEventRecord(r.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
  }
};

int g_num_cu = 0;
int num_cu() {
  if (g_num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_num_cu = prop.multiProcessorCount;
    if (g_num_cu <= 0) g_num_cu = 256;
  }
  return g_num_cu;
}

inline int grid_for(int64_t work_items) {
  int64_t cap = (int64_t)num_cu() * kMaxBlocksPerCU;
  int64_t g = work_items < cap ? work_items : cap;
  return (int)(g < 1 ? 1 : g);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}

// max over the workgroup; result valid in thread 0.  `red` = 4 floats of LDS.
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();                       // protect `red` against the previous step's readers
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  return v;
}
__device__ __forceinline__ float block_min(float v, float* red) {
  v = wave_min(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) v = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
  return v;
}

// Order-preserving atomics on fp32 through integer atomics (no CAS loop).
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (v >= 0.0f)
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
  if (v >= 0.0f)
    atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else
    atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// Deterministic batch mean: fp64 accumulate in sample order, one rounding to fp32, fp32 divide (oracle: batch_mean).
__device__ __forceinline__ float batch_mean_seq(const float* __restrict__ v, int n) {
  double acc = 0.0;
  for (int i = 0; i < n; ++i) acc += (double)v[i];
  return (float)acc / (float)n;
}

// The same value computed by a whole (converged) wavefront: every kernel that takes the online statistic starts with
// this, and the serial loop above cost each workgroup ~6 us before its first useful load (tools/pw_trace.py).
// Lanes take strided partial sums and the wave tree-reduces them in fp64.  That changes the ORDER of the additions, so
// the result is only accepted when order provably cannot matter: all finite non-zero inputs are integer multiples of
// q = 2^(emin-23), so every partial sum of any subset is a multiple of q bounded by n * 2^(emax+1); when
// (emax - emin) + 24 + ceil(log2 n) <= 53 all of them are exactly representable in fp64, i.e. every addition in every
// order is exact.  Otherwise (statistics spread over > 2^20, Inf/NaN) the wave falls back to the serial loop.
__device__ __forceinline__ float batch_mean_dev(const float* __restrict__ v, int n) {
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  unsigned emin = 255u, emax = 0u;
  for (int i = lane; i < n; i += 64) {
    const float f = v[i];
    acc += (double)f;
    unsigned e = (__float_as_uint(f) >> 23) & 0xFFu;
    if ((__float_as_uint(f) & 0x7FFFFFFFu) != 0u) {
      e = e < 1u ? 1u : e;                       // denormals share the lsb of exponent field 1
      emin = e < emin ? e : emin;
      emax = e > emax ? e : emax;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    acc += __shfl_xor(acc, off, 64);
    const unsigned a = (unsigned)__shfl_xor((int)emin, off, 64), b = (unsigned)__shfl_xor((int)emax, off, 64);
    emin = a < emin ? a : emin;
    emax = b > emax ? b : emax;
  }
  const int logn = 32 - __clz(n > 1 ? n - 1 : 1);
  const bool exact = emax == 0u || (emax < 255u && (int)(emax - emin) + 24 + logn <= 53);
  if (!exact) return batch_mean_seq(v, n);
  return (float)acc / (float)n;
}

struct QParams {
  float lo, hi;      // clip bounds
  float denom;       // scale + eps
  float scale;       // multiply-back scale (no eps)
  double rden;       // RN_f64(1 / denom), see ieee_div_by()
};

__device__ __forceinline__ QParams make_qparams(float max_, float levels, bool lo_neg_max, float eps) {
  QParams q;
  q.hi = max_;
  q.lo = lo_neg_max ? -max_ : 0.0f;
  q.scale = max_ / levels;
  q.denom = q.scale + eps;
  q.rden = 1.0 / (double)q.denom;
  return q;
}

// Correctly rounded fp32 quotient c / d for a divisor that is the same for the whole kernel, in 3 instructions instead
// of the ~11 of the IEEE division expansion:  (float)((double)c * RN_f64(1/d))  ==  RN_f32(c / d)  for ALL fp32 c, d.
// Proof sketch: the double product carries a relative error <= 2^-52, while the exact quotient of two fp32 numbers is
// either an fp32 number or at least 2^-49 (relative) away from every fp32 rounding midpoint (c - M*d is a non-zero
// integer multiple of 2^(g+b) for a 25-bit midpoint M = N*2^g and d = D*2^b), so the product falls on the same side of
// every midpoint as the exact quotient and the final conversion rounds it to the same fp32 number.  0/0 and x/0 behave
// as in IEEE (rden = inf).  Checked exhaustively around every .5 tie in tests/test_gpu_parity.py.
__device__ __forceinline__ float ieee_div_by(float c, double rden) { return (float)((double)c * rden); }

// The integer stage and the dequantised value (ste_func.py:41): clip -> IEEE divide -> roundf -> multiply.
// roundf(Q) (half away from zero) == trunc(Q + copysign(pred(0.5), Q)) for every fp32 |Q| < 2^23: the only fp32 whose
// sum with 0.5 would round across an integer is pred(0.5), and pred(0.5) + pred(0.5) is exact (checked exhaustively for
// |Q| <= 70000; codes are <= 65535).  3 instructions instead of roundf's 6; NaN and Inf pass through as with roundf.
__device__ __forceinline__ float round_half_away(float Q) { return truncf(Q + __builtin_copysignf(0.49999997f, Q)); }

// clip(x, lo, hi) in one instruction: v_med3_f32 is the median of three, i.e. the clamp when lo <= hi (always: hi = max_ >= 0,
// lo = 0 or -max_); a NaN input yields lo, exactly as fminf(fmaxf(NaN, lo), hi) does.  (fmaxf/fminf cost three: the
// compiler first canonicalises x with a v_max.)
__device__ __forceinline__ float fq_clip(float x, const QParams& q) { return __builtin_amdgcn_fmed3f(x, q.lo, q.hi); }

__device__ __forceinline__ float fq_code(float x, const QParams& q) {
  float c = fq_clip(x, q);
  return round_half_away(ieee_div_by(c, q.rden));
}

// The integer code itself, for the kernels that keep codes (int8 paths): the same value as (int)fq_code(x, q) in
// fewer instructions — these kernels are instruction-bound, not HBM-bound (tools/pw_trace.py --ablate).
// roundf(Q) == trunc(Q + copysign(pred(0.5), Q)) for every fp32 |Q| < 2^23 (checked exhaustively for |Q| <= 70000:
// the only fp32 for which Q + 0.5 itself would round across an integer is pred(0.5), and pred(0.5) + pred(0.5) is exact).
__device__ __forceinline__ int fq_code_int(float x, const QParams& q) {
  const float c = fq_clip(x, q);
  const float Q = ieee_div_by(c, q.rden);
  return (int)(Q + __builtin_copysignf(0.49999997f, Q));
}

// Four codes -> one dword of int8.  `ubias` = 128 - zoff makes every code non-negative (unsigned codes are stored
// re-centred by zoff = 128, signed ones as they are), so the bytes can be merged without masks; the final XOR turns
// u = code + 128 back into the two's complement byte of code - zoff... i.e. (u ^ 0x80) == (u - 128) mod 256.
__device__ __forceinline__ int pack4_codes(int k0, int k1, int k2, int k3, int ubias) {
  unsigned u = (unsigned)(k0 + ubias);
  u |= (unsigned)(k1 + ubias) << 8;
  u |= (unsigned)(k2 + ubias) << 16;
  u |= (unsigned)(k3 + ubias) << 24;
  return (int)(u ^ 0x80808080u);
}

template <bool USE_ABS>
__device__ __forceinline__ float stat_of(float v) {
  return USE_ABS ? fabsf(v) : v;
}
template <bool USE_ABS>
__device__ __forceinline__ float stat_init() {
  return USE_ABS ? 0.0f : -INFINITY;
}

// ---------------------------------------------------------------------------------------------------------------
// K0: fill
// ---------------------------------------------------------------------------------------------------------------
__global__ void fill_kernel(float* p, int64_t n, float v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// Streaming helpers.  f4 is the native 16-byte vector (the nontemporal builtins want a native vector type).
// Policy bits (host-chosen, see stream_policy()):
//   kPolNtLoad   : x is dead after this pass -> nontemporal loads (do not displace other lines in L2 / Infinity Cache)
//   kPolNtStore  : nontemporal stores of y
//   kPolReverse  : walk each workgroup's chunk range backwards — the second pass of the online path starts with the
//                  chunks the statistic pass read LAST, which are the ones still resident in the 256 MiB Infinity
//                  Cache when the tensor is larger than it.
// ---------------------------------------------------------------------------------------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
constexpr int kPolNtLoad = 1, kPolNtStore = 2, kPolReverse = 4;

template <bool NT>
__device__ __forceinline__ f4 ld4(const f4* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st4(f4* p, f4 v) {
  if (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}
template <bool USE_ABS>
__device__ __forceinline__ float stat4(f4 v) {
  return fmaxf(fmaxf(stat_of<USE_ABS>(v.x), stat_of<USE_ABS>(v.y)), fmaxf(stat_of<USE_ABS>(v.z), stat_of<USE_ABS>(v.w)));
}

// Each workgroup owns a CONTIGUOUS range of 32 KiB chunks; a chunk never spans two samples.  The lane-local running
// maximum is carried across chunks and only reduced (shuffle tree -> LDS -> one atomic) when the sample changes.
struct ChunkRange {
  int64_t begin, end;   // [begin, end)
};
__device__ __forceinline__ ChunkRange block_range(int64_t total_chunks) {
  const int64_t per = (total_chunks + gridDim.x - 1) / gridDim.x;
  ChunkRange r;
  r.begin = (int64_t)blockIdx.x * per;
  r.end = r.begin + per < total_chunks ? r.begin + per : total_chunks;
  return r;
}

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

// ---------------------------------------------------------------------------------------------------------------
// K2: apply.  ONLINE: threshold = mean of stat_in[0..n);  else threshold = thr[0].
//     STATS (offline only): also produce the per-sample statistic of x into stat_out (fused, same pass).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f4 fq_code4(f4 v, const QParams& q) {
  f4 k;
  k.x = fq_code(v.x, q);
  k.y = fq_code(v.y, q);
  k.z = fq_code(v.z, q);
  k.w = fq_code(v.w, q);
  return k;
}

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

template <int ACT, bool STATS, bool VEC, int U>
__global__ __launch_bounds__(kBlock) void bn_act_stat_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int64_t inner, int hw, int chunks_per_sample,
                                                             int64_t total_chunks, const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             float* __restrict__ stat_out) {
  constexpr int kCh = kBlock * kVec * U;
  __shared__ float red[4];
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
    if (VEC) {
      const f4* p = reinterpret_cast<const f4*>(x + gbase);
      f4* o = reinterpret_cast<f4*>(y + gbase);
      const int nvec = (int)((rem < kCh ? rem : kCh) / kVec);
      f4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * kBlock;
        if (i < nvec) v[u] = p[i];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * kBlock;
        if (i < nvec) {
          const unsigned e0 = (unsigned)(off0 + (int64_t)i * kVec);       // element index inside the sample (< 2^32)
          unsigned ch = e0 / (unsigned)hw;
          unsigned r = e0 - ch * (unsigned)hw;
          float sc = scale[ch], sh = shift[ch];
          f4 q;
          q.x = bn_act1<ACT>(v[u].x, sc, sh);
          if (++r == (unsigned)hw) { r = 0; ++ch; sc = scale[ch]; sh = shift[ch]; }
          q.y = bn_act1<ACT>(v[u].y, sc, sh);
          if (++r == (unsigned)hw) { r = 0; ++ch; sc = scale[ch]; sh = shift[ch]; }
          q.z = bn_act1<ACT>(v[u].z, sc, sh);
          if (++r == (unsigned)hw) { r = 0; ++ch; sc = scale[ch]; sh = shift[ch]; }
          q.w = bn_act1<ACT>(v[u].w, sc, sh);
          if (STATS) m = fmaxf(m, stat4<true>(q));
          o[i] = q;
        }
      }
    } else {
      const int cnt = (int)(rem < kCh ? rem : kCh);
      for (int i = threadIdx.x; i < cnt; i += kBlock) {
        const unsigned e = (unsigned)(off0 + i);
        const unsigned ch = e / (unsigned)hw;
        const float q = bn_act1<ACT>(x[gbase + i], scale[ch], shift[ch]);
        if (STATS) m = fmaxf(m, fabsf(q));
        y[gbase + i] = q;
      }
    }
  }
  if (STATS && cur_s >= 0) {
    m = block_max(m, red);
    if (threadIdx.x == 0) atomic_max_f32(stat_out + cur_s, m);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K2c: depthwise 3x3 (pad 1, stride S) with quantise-on-load and BN / activation / statistic epilogue.
// A workgroup step ("tile") is P whole planes (small planes) or a strip of output rows of one plane (large planes).
// The input rows of a tile are contiguous in memory: they are read once, coalesced (16 B per lane), quantised ONCE per
// element and staged into an LDS tile that carries a zero border, so the 9-tap loop has no bounds checks.  In the
// compute phase consecutive lanes own consecutive output COLUMNS (conflict-free LDS reads, coalesced stores) and slide
// down a segment of rows keeping the 3x3 window in registers: 3 new LDS values per output (6 for stride 2).
// ---------------------------------------------------------------------------------------------------------------
#ifdef FQ_PW_TRACE
// debug build only (tools/pw_trace.py): per-workgroup wall-clock stamps of the fused pointwise kernel's phases
__device__ unsigned long long* g_pw_trace = nullptr;
__device__ int g_pw_dbg = 0;          // experiments: 1 = skip the output stores, 2 = skip the activation loads
#define PW_STAMP(i)                                                                       \
  do {                                                                                    \
    if (threadIdx.x == 0 && g_pw_trace != nullptr) g_pw_trace[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define PW_STAMP(i) do { } while (0)
#endif

struct DwGeom {
  int C, H, W, Ho, Wo;
  int P;        // planes per tile (whole-plane mode) or 1
  int TR;       // output rows per tile
  int strips;   // tiles per plane along rows (1 in whole-plane mode)
  int IR;       // LDS rows per plane (TR*S + 2)
  int WS;       // LDS row stride (>= W + 2)
  int nseg;     // row segments per tile in the compute phase
  int RS;       // output rows per segment
  int vec_in;   // 16-byte loads allowed
};

__device__ __forceinline__ float act_rt(float v, int act) {
  if (act == FQ_ACT_RELU) v = fmaxf(v, 0.0f);
  if (act == FQ_ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
  return v;
}

template <int S, bool QUANT, bool ONLINE>
__global__ __launch_bounds__(kBlock) void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           DwGeom g, int64_t tiles, const float* __restrict__ in_stat,
                                                           int n, const float* __restrict__ in_thr, float levels,
                                                           int lo_neg_max, float eps,
                                                           float* __restrict__ cur_max_out,
                                                           const float* __restrict__ bn_scale,
                                                           const float* __restrict__ bn_shift, int act,
                                                           float* __restrict__ stat_out) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ float red[4];
  QParams q;
  q.lo = q.hi = q.denom = q.scale = 0.0f;
  q.rden = 0.0;
  if (QUANT) {
    const float max_ = ONLINE ? batch_mean_dev(in_stat, n) : in_thr[0];
    q = make_qparams(max_, levels, lo_neg_max != 0, eps);
    if (ONLINE && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
  }
  const int lds_elems = g.P * g.IR * g.WS;
  const int plane_in = g.H * g.W, plane_out = g.Ho * g.Wo;
  const bool has_bn = bn_scale != nullptr;
  const bool has_stat = stat_out != nullptr;

  const ChunkRange rg = block_range(tiles);
  for (int64_t t = rg.begin; t < rg.end; ++t) {
    const int64_t pg = t / g.strips;                   // plane group
    const int strip = (int)(t - pg * g.strips);
    const int64_t plane0 = pg * g.P;                   // first (n*C + c) plane of the tile
    const int ch0 = (int)(plane0 % g.C);               // P divides C: the tile's planes are ch0 .. ch0+P-1 of ONE sample
    const int orow0 = strip * g.TR;                    // first output row
    const int orows = (g.Ho - orow0) < g.TR ? (g.Ho - orow0) : g.TR;
    const int irow_first = orow0 * S - 1;              // input row held by LDS row 0 (may be -1)
    const int r_lo = irow_first < 0 ? 0 : irow_first;
    int r_hi = irow_first + g.IR - 1;
    if (r_hi > g.H - 1) r_hi = g.H - 1;

    __syncthreads();                                   // previous tile fully consumed
    for (int i = threadIdx.x * 4; i < lds_elems; i += kBlock * 4)
      *reinterpret_cast<f4*>(tile + i) = (f4){0.f, 0.f, 0.f, 0.f};     // (allocation is padded to a multiple of 4)
    __syncthreads();
    // ---- load + quantise + stage: rows [r_lo, r_hi] of P consecutive planes -------------------------------------
    // whole-plane mode: r_lo = 0, r_hi = H-1 and the P planes are one contiguous range; strip mode: P = 1.
    {
      const float* src = x + plane0 * (int64_t)plane_in + (int64_t)r_lo * g.W;
      const int rows_per_plane = r_hi - r_lo + 1;
      const int cnt = (g.P > 1) ? g.P * plane_in : rows_per_plane * g.W;
      const unsigned W = (unsigned)g.W, RP = (unsigned)rows_per_plane;
      if (g.vec_in) {
        const f4* p4 = reinterpret_cast<const f4*>(src);
        for (int i = threadIdx.x; i < cnt / 4; i += kBlock) {
          f4 v = p4[i];
          if (QUANT) v = fq_code4(v, q) * q.scale;
          const unsigned e = (unsigned)i * 4u;
          unsigned row = e / W;                                   // row index over the tile's planes
          unsigned col = e - row * W;
          unsigned pl = row / RP;
          unsigned r = row - pl * RP;
          float* d = tile + (pl * g.IR + (r + r_lo - irow_first)) * g.WS + col + 1;
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            d[0] = vv[k];
            ++d;
            if (++col == W) {                                     // next row (possibly next plane)
              col = 0;
              if (++r == RP) { r = 0; ++pl; }
              d = tile + (pl * g.IR + (r + r_lo - irow_first)) * g.WS + 1;
            }
          }
        }
      } else {
        for (int i = threadIdx.x; i < cnt; i += kBlock) {
          float v = src[i];
          if (QUANT) v = fq_code(v, q) * q.scale;
          const unsigned row = (unsigned)i / W;
          const unsigned col = (unsigned)i - row * W;
          const unsigned pl = row / RP;
          const unsigned r = row - pl * RP;
          tile[(pl * g.IR + (r + r_lo - irow_first)) * g.WS + col + 1] = v;
        }
      }
    }
    __syncthreads();
    // ---- sliding 3x3 window down a row segment; lane <-> output column ------------------------------------------
    float m = 0.0f;
    const int items = g.P * g.nseg * g.Wo;
    for (int it = threadIdx.x; it < items; it += kBlock) {
      const unsigned ps = (unsigned)it / (unsigned)g.Wo;
      const unsigned col = (unsigned)it - ps * (unsigned)g.Wo;
      const unsigned pl = ps / (unsigned)g.nseg;
      const unsigned seg = ps - pl * (unsigned)g.nseg;
      const int rr0 = (int)seg * g.RS;
      int rr1 = rr0 + g.RS;
      if (rr1 > orows) rr1 = orows;
      if (rr0 >= rr1) continue;
      const int ch = ch0 + (int)pl;
      const float* wk = wgt + ch * 9;
      const float w00 = wk[0], w01 = wk[1], w02 = wk[2], w10 = wk[3], w11 = wk[4], w12 = wk[5], w20 = wk[6],
                  w21 = wk[7], w22 = wk[8];
      const float bch = bias != nullptr ? bias[ch] : 0.0f;
      const float bsc = has_bn ? bn_scale[ch] : 1.0f, bsh = has_bn ? bn_shift[ch] : 0.0f;
      const float* l = tile + (pl * g.IR + rr0 * S) * g.WS + col * S;       // LDS col 0 == input col -1
      float a0 = l[0], a1 = l[1], a2 = l[2];
      float b0 = 0.f, b1 = 0.f, b2 = 0.f;
      if (S == 1) {
        b0 = l[g.WS];
        b1 = l[g.WS + 1];
        b2 = l[g.WS + 2];
      }
      float* dst = y + (plane0 + pl) * (int64_t)plane_out + (int64_t)(orow0 + rr0) * g.Wo + col;
      for (int r = rr0; r < rr1; ++r) {
        float c0, c1, c2;
        if (S == 1) {
          const float* lc = l + 2 * g.WS;
          c0 = lc[0]; c1 = lc[1]; c2 = lc[2];
        } else {
          const float* lb = l + g.WS;
          b0 = lb[0]; b1 = lb[1]; b2 = lb[2];
          const float* lc = lb + g.WS;
          c0 = lc[0]; c1 = lc[1]; c2 = lc[2];
        }
        float acc = 0.0f;
        acc = fmaf(w00, a0, acc);
        acc = fmaf(w01, a1, acc);
        acc = fmaf(w02, a2, acc);
        acc = fmaf(w10, b0, acc);
        acc = fmaf(w11, b1, acc);
        acc = fmaf(w12, b2, acc);
        acc = fmaf(w20, c0, acc);
        acc = fmaf(w21, c1, acc);
        acc = fmaf(w22, c2, acc);
        if (bias != nullptr) acc = acc + bch;
        if (has_bn) {
          acc = acc * bsc;
          acc = acc + bsh;
        }
        acc = act_rt(acc, act);
        *dst = acc;
        m = fmaxf(m, fabsf(acc));
        dst += g.Wo;
        l += S * g.WS;
        if (S == 1) {
          a0 = b0; a1 = b1; a2 = b2;
          b0 = c0; b1 = c1; b2 = c2;
        } else {
          a0 = c0; a1 = c1; a2 = c2;
        }
      }
    }
    if (has_stat) {
      m = block_max(m, red);
      if (threadIdx.x == 0) atomic_max_f32(stat_out + plane0 / g.C, m);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K2d: depthwise 3x3, register sliding-window form (no LDS, no barriers).  A wavefront holds `segs` independent
// SEGMENTS; a segment is `sw` adjacent output columns of one plane plus halo lanes (stride 1: one on each side, stride
// 2: one on the left).  Every lane streams ITS input column(s) top to bottom — one coalesced 4-byte load per lane per
// input row, quantised once — and gets its horizontal neighbours from the adjacent lanes with wavefront shuffles; the
// three live input rows stay in registers, loads run D rows ahead of their use.
// ---------------------------------------------------------------------------------------------------------------
struct DwColGeom {
  int C, H, W, Ho, Wo;
  int sw;       // output columns per segment
  int nsegx;    // segments per plane row
  int SEG;      // lanes per segment (sw + halo lanes)
  int segs;     // segments per wavefront
};

template <int S, bool QUANT, bool ONLINE>
__global__ __launch_bounds__(kBlock) void dwconv3x3_cols_kernel(
    const float* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
    float* __restrict__ y, DwColGeom g, int64_t total_segs, const float* __restrict__ in_stat, int n,
    const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps, float* __restrict__ cur_max_out,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act, float* __restrict__ stat_out) {
  // output rows of input kept in flight (S=2: 4 loads per row).  (16 / 8 - every row of a 14x14 plane in flight at
  // once - measured SLOWER on the same box: 38 vs 36 us per 512x14x14 layer.)
  constexpr int D = (S == 1) ? 8 : 4;
  constexpr int kStatSlots = 16;
  __shared__ unsigned k_stat[kStatSlots];
  if (threadIdx.x < kStatSlots) k_stat[threadIdx.x] = 0u;
  PW_STAMP(0);
  QParams q;
  q.lo = q.hi = q.denom = q.scale = 0.0f;
  q.rden = 0.0;
  if (QUANT) {
    const float max_ = ONLINE ? batch_mean_dev(in_stat, n) : in_thr[0];
    q = make_qparams(max_, levels, lo_neg_max != 0, eps);
    if (ONLINE && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
  }
  PW_STAMP(1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg_in_wave = lane / g.SEG;
  const int pos = lane - seg_in_wave * g.SEG;
  const bool lane_used = seg_in_wave < g.segs;
  const int64_t segs_per_block = (int64_t)g.segs * (kBlock / 64);
  const int64_t nblk = (total_segs + segs_per_block - 1) / segs_per_block;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const int plane_in = g.H * g.W, plane_out = g.Ho * g.Wo;

  // (Row strips per plane were tried for load balance and measured slower: every strip restarts the prefetch ring.)
  // A workgroup takes a CONTIGUOUS range of blocks (few samples -> a small LDS statistic table, one flush) and its
  // wavefronts run through them without barriers.  Index arithmetic is unsigned 32-bit (host: total_segs < 2^31): the
  // 64-bit divisions it replaces cost each block 2.3 us (tools/dw_trace.py).
  const unsigned nblk_u = (unsigned)nblk, tsegs = (unsigned)total_segs, nsegx = (unsigned)g.nsegx, C_u = (unsigned)g.C;
  const unsigned blk_begin = (unsigned)((uint64_t)nblk_u * blockIdx.x / gridDim.x);
  const unsigned blk_end = (unsigned)((uint64_t)nblk_u * (blockIdx.x + 1) / gridDim.x);
  const unsigned n_samples = tsegs / nsegx / C_u;
  unsigned s_base;
  {
    const unsigned seg0 = blk_begin * (unsigned)segs_per_block;
    s_base = (seg0 < tsegs ? seg0 : tsegs - 1) / nsegx / C_u;
  }
  __syncthreads();                                       // statistic table zeroed
  for (unsigned blk = blk_begin; blk < blk_end; ++blk) {
    const unsigned seg = blk * (unsigned)segs_per_block + (unsigned)wave * (unsigned)g.segs + (unsigned)seg_in_wave;
    const bool seg_ok = lane_used && seg < tsegs;
    const unsigned plane = seg_ok ? seg / nsegx : 0u;
    const int sx = seg_ok ? (int)(seg - plane * nsegx) : 0;
    const unsigned sample_u = plane / C_u;
    const int ch = (int)(plane - sample_u * C_u);
    const int sample = (int)sample_u;
    // column bookkeeping
    int oc, ic0;                                          // output column; first input column this lane loads
    bool is_out;
    if (S == 1) {
      ic0 = sx * g.sw + pos - 1;                          // pos 0 / sw+1 are the halo lanes
      oc = ic0;
      is_out = seg_ok && pos >= 1 && pos <= g.sw && oc < g.Wo;
    } else {
      oc = sx * g.sw + pos - 1;                           // pos 0 is the left-halo lane
      ic0 = 2 * oc;                                       // this lane loads input columns 2*oc and 2*oc + 1
      is_out = seg_ok && pos >= 1 && oc < g.Wo;
    }
    const bool ld0 = seg_ok && (S == 1 ? (ic0 >= 0 && ic0 < g.W) : (pos >= 1 && ic0 < g.W));
    const bool ld1 = seg_ok && S == 2 && (ic0 + 1 >= 0) && (ic0 + 1 < g.W);     // stride 2: second column (halo lane: col 2*sx*sw - 1)
    const float* xp = x + plane * (int64_t)plane_in + ic0;
    float* yp = y + plane * (int64_t)plane_out + oc;
    const float* wk = wgt + ch * 9;
    const float w00 = wk[0], w01 = wk[1], w02 = wk[2], w10 = wk[3], w11 = wk[4], w12 = wk[5], w20 = wk[6],
                w21 = wk[7], w22 = wk[8];
    const float bch = bias != nullptr ? bias[ch] : 0.0f;
    const float bsc = has_bn ? bn_scale[ch] : 1.0f, bsh = has_bn ? bn_shift[ch] : 0.0f;
    float m = 0.0f;
    if (blk == blk_begin) PW_STAMP(2);

    // Loads are UNCONDITIONAL from clamped (always valid) addresses and masked afterwards with a bitwise AND — a
    // `cond ? load : 0` select is turned back into a predicated load by hipcc (CodeGenPrepare sinks the load under a
    // branch), which then waits vmcnt(0) right behind it and the prefetch ring is gone.
    const int last_row = g.H - 1;
    auto keep = [](float v, bool ok) -> float { return __uint_as_float(__float_as_uint(v) & (ok ? 0xFFFFFFFFu : 0u)); };
    if (S == 1) {
      // rows: a = input row r-1, b = row r, c = row r+1 (each as left / centre / right)
      const float* xs = ld0 ? xp : x;                      // lanes with nothing to load read element 0
      auto ldrow = [&](int row) -> float {
        const int rc = row < last_row ? row : last_row;
        return keep(xs[(int64_t)rc * g.W], ld0 && row <= last_row);
      };
      auto emit = [&](int r, float c1, float& a0, float& a1, float& a2, float& b0, float& b1, float& b2) {
        if (QUANT) c1 = fq_code(c1, q) * q.scale;
        const float c0 = __shfl_up(c1, 1, 64), c2 = __shfl_down(c1, 1, 64);
        float acc = 0.0f;
        acc = fmaf(w00, a0, acc);
        acc = fmaf(w01, a1, acc);
        acc = fmaf(w02, a2, acc);
        acc = fmaf(w10, b0, acc);
        acc = fmaf(w11, b1, acc);
        acc = fmaf(w12, b2, acc);
        acc = fmaf(w20, c0, acc);
        acc = fmaf(w21, c1, acc);
        acc = fmaf(w22, c2, acc);
        if (bias != nullptr) acc = acc + bch;
        if (has_bn) {
          acc = acc * bsc;
          acc = acc + bsh;
        }
        acc = act_rt(acc, act);
        m = fmaxf(m, keep(fabsf(acc), is_out));
        if (is_out) yp[(int64_t)r * g.Wo] = acc;
        a0 = b0; a1 = b1; a2 = b2;
        b0 = c0; b1 = c1; b2 = c2;
      };
      float raw[D];
#pragma unroll
      for (int k = 0; k < D; ++k) raw[k] = ldrow(1 + k);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      float b1 = ldrow(0);
      if (QUANT) b1 = fq_code(b1, q) * q.scale;
      float b0 = __shfl_up(b1, 1, 64), b2 = __shfl_down(b1, 1, 64);
      const int rend = g.Ho;                              // same trip count for every lane of the grid
      int r0 = 0;
      for (; r0 + D <= rend; r0 += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const float c1 = raw[k];
          raw[k] = ldrow(r0 + k + 1 + D);
          emit(r0 + k, c1, a0, a1, a2, b0, b1, b2);
        }
      }
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (r0 + k < rend) emit(r0 + k, raw[k], a0, a1, a2, b0, b1, b2);
    } else {
      // stride 2: output row r uses input rows 2r-1 (a), 2r (b), 2r+1 (c); per input row: left = neighbour's 2nd
      // column, centre = own 1st column, right = own 2nd column
      const float* xs0 = ld0 ? xp : x;
      const float* xs1 = ld1 ? xp + 1 : x;
      auto ld_a = [&](int row) -> float {
        const int rc = row < last_row ? row : last_row;
        return keep(xs0[(int64_t)rc * g.W], ld0 && row <= last_row);
      };
      auto ld_b = [&](int row) -> float {
        const int rc = row < last_row ? row : last_row;
        return keep(xs1[(int64_t)rc * g.W], ld1 && row <= last_row);
      };
      auto emit2 = [&](int r, float b1, float b2, float c1, float c2, float& a0, float& a1, float& a2) {
        if (QUANT) {
          b1 = fq_code(b1, q) * q.scale;
          b2 = fq_code(b2, q) * q.scale;
          c1 = fq_code(c1, q) * q.scale;
          c2 = fq_code(c2, q) * q.scale;
        }
        const float b0 = __shfl_up(b2, 1, 64), c0 = __shfl_up(c2, 1, 64);
        float acc = 0.0f;
        acc = fmaf(w00, a0, acc);
        acc = fmaf(w01, a1, acc);
        acc = fmaf(w02, a2, acc);
        acc = fmaf(w10, b0, acc);
        acc = fmaf(w11, b1, acc);
        acc = fmaf(w12, b2, acc);
        acc = fmaf(w20, c0, acc);
        acc = fmaf(w21, c1, acc);
        acc = fmaf(w22, c2, acc);
        if (bias != nullptr) acc = acc + bch;
        if (has_bn) {
          acc = acc * bsc;
          acc = acc + bsh;
        }
        acc = act_rt(acc, act);
        m = fmaxf(m, keep(fabsf(acc), is_out));
        if (is_out) yp[(int64_t)r * g.Wo] = acc;
        a0 = c0; a1 = c1; a2 = c2;
      };
      float rb0[D], rb1[D], rc0[D], rc1[D];
#pragma unroll
      for (int k = 0; k < D; ++k) {
        rb0[k] = ld_a(2 * k);
        rb1[k] = ld_b(2 * k);
        rc0[k] = ld_a(2 * k + 1);
        rc1[k] = ld_b(2 * k + 1);
      }
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      const int rend = g.Ho;
      int r0 = 0;
      for (; r0 + D <= rend; r0 += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const float b1 = rb0[k], b2 = rb1[k], c1 = rc0[k], c2 = rc1[k];
          const int rn = r0 + k + D;
          rb0[k] = ld_a(2 * rn);
          rb1[k] = ld_b(2 * rn);
          rc0[k] = ld_a(2 * rn + 1);
          rc1[k] = ld_b(2 * rn + 1);
          emit2(r0 + k, b1, b2, c1, c2, a0, a1, a2);
        }
      }
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (r0 + k < rend) emit2(r0 + k, rb0[k], rb1[k], rc0[k], rc1[k], a0, a1, a2);
    }
    if (blk == blk_begin) PW_STAMP(3);
    if (has_stat) {
      // per-wave update of the workgroup's LDS table (no barrier inside the block loop); flushed once at the end
      const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane(sample);
      const bool wave_uniform = __all(!is_out || (unsigned)sample == s0);
      if (wave_uniform) {
        const float wm = wave_max(is_out ? m : 0.0f);
        if (lane == 0 && __float_as_uint(wm) != 0u) {
          const unsigned slot = s0 - s_base;
          if (slot < (unsigned)kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
          else atomic_max_f32(stat_out + s0, wm);
        }
      } else if (is_out) {
        const unsigned slot = (unsigned)sample - s_base;
        if (slot < (unsigned)kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
        else atomic_max_f32(stat_out + sample, m);
      }
    }
  }
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kStatSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < (unsigned)n_samples)
      atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
  PW_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------------------
// K2e: the same sliding window with FOUR input columns per lane (16-byte loads, 16-byte stores for stride 1 / 8-byte
// for stride 2) — needs W % 4 == 0.  With 4-byte accesses the kernel above cannot keep enough bytes in flight
// (PMC: 38 % of wave cycles parked on vmcnt at 4.1 TB/s); this form has 4x the bytes per outstanding load.
// Lane p of a segment: p = 0 left-halo lane, 1..L compute lanes (input columns 4(p-1)..4(p-1)+3), L+1 right-halo lane
// (stride 1 only).  Halo lanes load and quantise like the others; their neighbours pick up .w / .x by shuffle.
// ---------------------------------------------------------------------------------------------------------------
template <int S, bool QUANT, bool ONLINE>
__global__ __launch_bounds__(kBlock) void dwconv3x3_cols4_kernel(
    const float* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
    float* __restrict__ y, DwColGeom g, int64_t total_segs, const float* __restrict__ in_stat, int n,
    const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps, float* __restrict__ cur_max_out,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act, float* __restrict__ stat_out) {
  // Input rows are fetched in BURSTS of D rows (double buffered): the D loads of a burst leave the wave back to back
  // and hit the same DRAM pages; one load per step (a ring) spreads them ~700 cycles apart, and with thousands of
  // waves each streaming its own plane every access then opens a new page.
#ifndef FQ_DW4_D1
#define FQ_DW4_D1 4
#define FQ_DW4_D2 2
#endif
  constexpr int D = (S == 1) ? FQ_DW4_D1 : FQ_DW4_D2;
  constexpr int kStatSlots = 16;
  __shared__ unsigned k_stat[kStatSlots];
  if (threadIdx.x < kStatSlots) k_stat[threadIdx.x] = 0u;
  QParams q;
  q.lo = q.hi = q.denom = q.scale = 0.0f;
  q.rden = 0.0;
  if (QUANT) {
    const float max_ = ONLINE ? batch_mean_dev(in_stat, n) : in_thr[0];
    q = make_qparams(max_, levels, lo_neg_max != 0, eps);
    if (ONLINE && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg_in_wave = lane / g.SEG;
  const int pos = lane - seg_in_wave * g.SEG;
  const bool lane_used = seg_in_wave < g.segs;
  const int64_t segs_per_block = (int64_t)g.segs * (kBlock / 64);
  const int64_t nblk = (total_segs + segs_per_block - 1) / segs_per_block;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const int plane_in = g.H * g.W, plane_out = g.Ho * g.Wo;
  const int last_row = g.H - 1;
  auto keep4 = [](f4 v, bool ok) -> f4 {
    const unsigned mk = ok ? 0xFFFFFFFFu : 0u;
    f4 r;
    r.x = __uint_as_float(__float_as_uint(v.x) & mk);
    r.y = __uint_as_float(__float_as_uint(v.y) & mk);
    r.z = __uint_as_float(__float_as_uint(v.z) & mk);
    r.w = __uint_as_float(__float_as_uint(v.w) & mk);
    return r;
  };
  auto keep = [](float v, bool ok) -> float { return __uint_as_float(__float_as_uint(v) & (ok ? 0xFFFFFFFFu : 0u)); };

  // A workgroup takes a CONTIGUOUS range of blocks (few samples -> a small LDS statistic table, one flush) and its
  // wavefronts run through them without barriers.  Index arithmetic is unsigned 32-bit (host: total_segs < 2^31): the
  // 64-bit divisions it replaces cost each block 2.3 us (tools/dw_trace.py).
  const unsigned nblk_u = (unsigned)nblk, tsegs = (unsigned)total_segs, nsegx = (unsigned)g.nsegx, C_u = (unsigned)g.C;
  const unsigned blk_begin = (unsigned)((uint64_t)nblk_u * blockIdx.x / gridDim.x);
  const unsigned blk_end = (unsigned)((uint64_t)nblk_u * (blockIdx.x + 1) / gridDim.x);
  const unsigned n_samples = tsegs / nsegx / C_u;
  unsigned s_base;
  {
    const unsigned seg0 = blk_begin * (unsigned)segs_per_block;
    s_base = (seg0 < tsegs ? seg0 : tsegs - 1) / nsegx / C_u;
  }
  __syncthreads();                                       // statistic table zeroed
  for (unsigned blk = blk_begin; blk < blk_end; ++blk) {
    const unsigned seg = blk * (unsigned)segs_per_block + (unsigned)wave * (unsigned)g.segs + (unsigned)seg_in_wave;
    const bool seg_ok = lane_used && seg < tsegs;
    const unsigned plane = seg_ok ? seg / nsegx : 0u;
    const int sx = seg_ok ? (int)(seg - plane * nsegx) : 0;
    const unsigned sample_u = plane / C_u;
    const int ch = (int)(plane - sample_u * C_u);
    const int sample = (int)sample_u;
    const int ic0 = (sx * g.sw + pos - 1) * 4;            // first of the 4 input columns this lane loads
    const bool ld_ok = seg_ok && ic0 >= 0 && ic0 < g.W;
    const int oc = S == 1 ? ic0 : ic0 / 2;                // first output column (4 outputs for S=1, 2 for S=2)
    const bool is_out = seg_ok && pos >= 1 && pos <= g.sw && oc < g.Wo;
    const f4* xs = reinterpret_cast<const f4*>(ld_ok ? x + plane * (int64_t)plane_in + ic0 : x);
    float* yp = y + plane * (int64_t)plane_out + oc;
    const int rowq = g.W / 4;                             // f4 per input row
    const float* wk = wgt + ch * 9;
    const float w00 = wk[0], w01 = wk[1], w02 = wk[2], w10 = wk[3], w11 = wk[4], w12 = wk[5], w20 = wk[6],
                w21 = wk[7], w22 = wk[8];
    const float bch = bias != nullptr ? bias[ch] : 0.0f;
    const float bsc = has_bn ? bn_scale[ch] : 1.0f, bsh = has_bn ? bn_shift[ch] : 0.0f;
    float m = 0.0f;
    auto ldrow = [&](int row) -> f4 {
      const int rc = row < last_row ? row : last_row;
      return keep4(xs[(int64_t)rc * rowq], ld_ok && row <= last_row);
    };
    auto quant4 = [&](f4 v) -> f4 { return QUANT ? fq_code4(v, q) * q.scale : v; };
    auto finish = [&](float acc) -> float {
      if (bias != nullptr) acc = acc + bch;
      if (has_bn) {
        acc = acc * bsc;
        acc = acc + bsh;
      }
      return act_rt(acc, act);
    };

    if (S == 1) {
      // a, b, c: rows r-1, r, r+1 as (left, v.x, v.y, v.z, v.w, right)
      float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b[6];
      f4 raw[D];
#pragma unroll
      for (int k = 0; k < D; ++k) raw[k] = ldrow(1 + k);
      {
        const f4 v = quant4(ldrow(0));
        b[0] = __shfl_up(v.w, 1, 64);
        b[1] = v.x; b[2] = v.y; b[3] = v.z; b[4] = v.w;
        b[5] = __shfl_down(v.x, 1, 64);
      }
      auto emit = [&](int r, f4 craw) {
        const f4 v = quant4(craw);
        float c[6];
        c[0] = __shfl_up(v.w, 1, 64);
        c[1] = v.x; c[2] = v.y; c[3] = v.z; c[4] = v.w;
        c[5] = __shfl_down(v.x, 1, 64);
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float acc = 0.0f;
          acc = fmaf(w00, a[k], acc);
          acc = fmaf(w01, a[k + 1], acc);
          acc = fmaf(w02, a[k + 2], acc);
          acc = fmaf(w10, b[k], acc);
          acc = fmaf(w11, b[k + 1], acc);
          acc = fmaf(w12, b[k + 2], acc);
          acc = fmaf(w20, c[k], acc);
          acc = fmaf(w21, c[k + 1], acc);
          acc = fmaf(w22, c[k + 2], acc);
          o[k] = finish(acc);
        }
        const float mm = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
        m = fmaxf(m, keep(mm, is_out));
        if (is_out) *reinterpret_cast<f4*>(yp + (int64_t)r * g.Wo) = (f4){o[0], o[1], o[2], o[3]};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          a[k] = b[k];
          b[k] = c[k];
        }
      };
      int r0 = 0;
      for (; r0 + D <= g.Ho; r0 += D) {
        f4 nxt[D];
#pragma unroll
        for (int k = 0; k < D; ++k) nxt[k] = ldrow(r0 + k + 1 + D);
#pragma unroll
        for (int k = 0; k < D; ++k) emit(r0 + k, raw[k]);
#pragma unroll
        for (int k = 0; k < D; ++k) raw[k] = nxt[k];
      }
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (r0 + k < g.Ho) emit(r0 + k, raw[k]);
    } else {
      // stride 2: lane holds input columns 4j..4j+3 -> outputs 2j (cols 4j-1,4j,4j+1) and 2j+1 (cols 4j+1..4j+3)
      float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};              // (left, x, y, z, w) of input row 2r-1
      f4 rb[D], rc[D];
#pragma unroll
      for (int k = 0; k < D; ++k) {
        rb[k] = ldrow(2 * k);
        rc[k] = ldrow(2 * k + 1);
      }
      auto emit2 = [&](int r, f4 braw, f4 craw) {
        const f4 vb = quant4(braw), vc = quant4(craw);
        const float b[5] = {__shfl_up(vb.w, 1, 64), vb.x, vb.y, vb.z, vb.w};
        const float c[5] = {__shfl_up(vc.w, 1, 64), vc.x, vc.y, vc.z, vc.w};
        float o[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          float acc = 0.0f;
          acc = fmaf(w00, a[2 * k], acc);
          acc = fmaf(w01, a[2 * k + 1], acc);
          acc = fmaf(w02, a[2 * k + 2], acc);
          acc = fmaf(w10, b[2 * k], acc);
          acc = fmaf(w11, b[2 * k + 1], acc);
          acc = fmaf(w12, b[2 * k + 2], acc);
          acc = fmaf(w20, c[2 * k], acc);
          acc = fmaf(w21, c[2 * k + 1], acc);
          acc = fmaf(w22, c[2 * k + 2], acc);
          o[k] = finish(acc);
        }
        m = fmaxf(m, keep(fmaxf(fabsf(o[0]), fabsf(o[1])), is_out));
        if (is_out) *reinterpret_cast<float2*>(yp + (int64_t)r * g.Wo) = make_float2(o[0], o[1]);
#pragma unroll
        for (int k = 0; k < 5; ++k) a[k] = c[k];
      };
      int r0 = 0;
      for (; r0 + D <= g.Ho; r0 += D) {
        f4 nb[D], nc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
          nb[k] = ldrow(2 * (r0 + k + D));
          nc[k] = ldrow(2 * (r0 + k + D) + 1);
        }
#pragma unroll
        for (int k = 0; k < D; ++k) emit2(r0 + k, rb[k], rc[k]);
#pragma unroll
        for (int k = 0; k < D; ++k) {
          rb[k] = nb[k];
          rc[k] = nc[k];
        }
      }
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (r0 + k < g.Ho) emit2(r0 + k, rb[k], rc[k]);
    }
    if (has_stat) {
      // per-wave update of the workgroup's LDS table (no barrier inside the block loop); flushed once at the end
      const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane(sample);
      const bool wave_uniform = __all(!is_out || (unsigned)sample == s0);
      if (wave_uniform) {
        const float wm = wave_max(is_out ? m : 0.0f);
        if (lane == 0 && __float_as_uint(wm) != 0u) {
          const unsigned slot = s0 - s_base;
          if (slot < (unsigned)kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
          else atomic_max_f32(stat_out + s0, wm);
        }
      } else if (is_out) {
        const unsigned slot = (unsigned)sample - s_base;
        if (slot < (unsigned)kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
        else atomic_max_f32(stat_out + sample, m);
      }
    }
  }
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kStatSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < (unsigned)n_samples)
      atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K2f: pointwise (1x1) convolution on integer codes with v_mfma_i32_16x16x64_i8.
// Operand layout of the instruction (probed on gfx950, tools/mfma_i8_probe.hip): lane l holds, for A, row l&15 and the
// 16 consecutive k = 16*(l>>4) .. +15 (one 16-byte register quad); the same for B with column l&15; C/D: column l&15,
// rows 4*(l>>4) + r.  NCHW keeps PIXELS contiguous, the MFMA wants K (= input channels) contiguous for both operands:
// weights are stored [co][ci] (fine), the activation tile is transposed on its way into LDS — it is read from HBM
// once, coalesced along pixels, quantised, and written as int8 codes [column][ci] with byte stores.
// A workgroup owns PT_B columns (column = (sample, pixel) flattened) and ALL output channels, so x is read exactly once;
// its 4 waves are arranged wm x wn over (64 output channels) x (64 columns) wave tiles and loop over channel passes.
// ---------------------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct PwGeom {
  int Cin, CinPad, Cout, HW;
  int64_t cols;        // n * HW
  int PT_B;            // columns per workgroup tile (64 * wn)
  int wm, wn;          // wave grid
  int passes;          // ceil(Cout / (64 * wm))
  int stride;          // LDS bytes per column (CinPad + 16)
  int zoff;            // 128 for unsigned codes (stored re-centred), 0 for signed
};

// K2f-A: quantise + transpose.  x (n, Cin, HW) fp32 -> codes [(n*HW + p)][CinPad] int8 (column-major for the GEMM:
// K contiguous).  Workgroup tile = 64 channels x 64 pixels of one sample: every thread loads 4 channel rows x 4 pixels
// (16-byte loads, 256 B contiguous per row across 16 lanes), quantises, transposes its 4x4 block in registers into 4
// dwords (4 channels of one pixel each), and the tile goes through a small LDS stage (odd dword stride: conflict-free)
// so that the stores are 16 bytes per lane, 64 contiguous bytes per pixel.
template <bool ONLINE>
__global__ __launch_bounds__(kBlock) void quant_transpose_i8_kernel(
    const float* __restrict__ x, int8_t* __restrict__ codes, int Cin, int CinPad, int HW, int ptiles, int ctiles,
    int64_t tiles, const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels,
    int lo_neg_max, float eps, int zoff, float* __restrict__ cur_max_out) {
  __shared__ int lds[64 * 17];
  const float max_ = ONLINE ? batch_mean_dev(in_stat, n) : in_thr[0];
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  if (ONLINE && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
  const int c = threadIdx.x & 15;            // pixel quad
  const int rq = threadIdx.x >> 4;           // channel quad (0..15)
  const bool hw_vec = (HW & 3) == 0;
  const ChunkRange rg = block_range(tiles);
  for (int64_t t = rg.begin; t < rg.end; ++t) {
    // tile order: channel tile fastest, then pixel tile, then sample
    const int ct = (int)(t % ctiles);
    const int64_t t2 = t / ctiles;
    const int pt = (int)(t2 % ptiles);
    const int64_t smp = t2 / ptiles;
    const int ci0 = ct * 64 + rq * 4, p0 = pt * 64 + c * 4;
    float v[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ci = ci0 + k;
      const int cic = ci < Cin ? ci : Cin - 1;
      const float* src = x + (smp * Cin + cic) * (int64_t)HW;
      if (hw_vec && p0 + 3 < HW) {
        const f4 r = *reinterpret_cast<const f4*>(src + p0);
        v[k][0] = r.x; v[k][1] = r.y; v[k][2] = r.z; v[k][3] = r.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pe = p0 + e < HW ? p0 + e : HW - 1;
          v[k][e] = src[pe];
        }
      }
    }
    __syncthreads();                                                   // LDS free (previous tile stored)
#pragma unroll
    for (int e = 0; e < 4; ++e) {                                       // pixel p0 + e: channels ci0 .. ci0+3 in one dword
      unsigned packed = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int code = (int)fq_code(v[k][e], q) - zoff;
        if (ci0 + k >= Cin) code = 0;
        packed |= ((unsigned)code & 0xFFu) << (8 * k);
      }
      lds[(c * 4 + e) * 17 + rq] = (int)packed;
    }
    __syncthreads();
    // store: thread (pixel = tid / 4, 16-byte piece = tid % 4)
    const int sp = threadIdx.x >> 2, piece = threadIdx.x & 3;
    const int p = pt * 64 + sp;
    if (p < HW) {
      const int* l = lds + sp * 17 + piece * 4;
      const v4i o = (v4i){l[0], l[1], l[2], l[3]};
      *reinterpret_cast<v4i*>(codes + (smp * HW + p) * (int64_t)CinPad + ct * 64 + piece * 16) = o;
    }
  }
}

// K2f-B: integer GEMM + epilogue.  Both MFMA operands are K-contiguous in global memory (weights [co][CinPad], codes
// [column][CinPad]) and go straight to registers, double buffered over the K loop; no LDS.  A workgroup computes
// (64*wm output channels) x (64*wn columns); waves own 64 x 64 sub-tiles = 16 accumulators of 16x16.
template <int DUMMY>
__global__ __launch_bounds__(kBlock) void pwconv_i8_kernel(
    const int8_t* __restrict__ xc, const int8_t* __restrict__ wc, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwGeom g, int64_t tiles,
    const float* __restrict__ sx_src, float levels, const float* __restrict__ bn_scale,
    const float* __restrict__ bn_shift, int act, float* __restrict__ stat_out) {
  // per-output-channel constants of this workgroup's channel block, staged in LDS once: loading them from global in
  // the epilogue (16 dependent round trips per tile) was 90 % of this kernel's time
  constexpr int kStatSlots = 16;
  __shared__ float k_sxw[256], k_bias[256], k_bsc[256], k_bsh[256];
  __shared__ int k_zs[256];
  __shared__ unsigned k_stat[kStatSlots];
  const float sx = sx_src[0] / levels;                                  // scale = max_/levels, as make_qparams
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wmi = wave % g.wm, wni = wave / g.wm;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const unsigned HW = (unsigned)g.HW;
  const int64_t plane_stride = (int64_t)g.HW;
  const int cblocks = g.passes;                                         // output-channel blocks of 64*wm
  // gridDim.x is a multiple of cblocks (host), so a workgroup keeps ONE channel block over all its tiles
  const int cb = (int)(blockIdx.x % cblocks);
  {
    const int co = cb * g.wm * 64 + threadIdx.x;
    const bool ok = threadIdx.x < g.wm * 64 && co < g.Cout;
    const int coc = ok ? co : 0;
    k_sxw[threadIdx.x] = sx * wscale[coc];
    k_zs[threadIdx.x] = g.zoff * wsum[coc];
    k_bias[threadIdx.x] = bias != nullptr ? bias[coc] : 0.0f;
    k_bsc[threadIdx.x] = has_bn ? bn_scale[coc] : 1.0f;
    k_bsh[threadIdx.x] = has_bn ? bn_shift[coc] : 0.0f;
  }
  __syncthreads();

  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t ctile = t / cblocks;
    const unsigned j0 = (unsigned)(ctile * g.PT_B) + wni * 64;
    const int co0 = (cb * g.wm + wmi) * 64;                             // (weights are zero-padded to 64 rows: no early exit,
    v4i acc[4][4];                                                      //  every wave reaches the barriers below)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (v4i){0, 0, 0, 0};
    // a wave whose 64 channels lie entirely beyond Cout (Cout not a multiple of 64*wm) reads channel block 0 instead:
    // its results are masked in the epilogue, but the weight buffer is only padded to the next multiple of 64 rows
    const int co_ld = co0 < g.Cout ? co0 : 0;
    const int8_t* wrow = wc + (int64_t)(co_ld + (lane & 15)) * g.CinPad + 16 * (lane >> 4);
    const int8_t* xrow = xc + (int64_t)(j0 + (lane & 15)) * g.CinPad + 16 * (lane >> 4);
    const int64_t step16 = (int64_t)16 * g.CinPad;
    v4i a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = *reinterpret_cast<const v4i*>(wrow + i * step16);
      b[i] = *reinterpret_cast<const v4i*>(xrow + i * step16);
    }
    for (int k0 = 0; k0 < g.CinPad; k0 += 64) {
      v4i an[4], bn[4];
      const int kn = k0 + 64 < g.CinPad ? k0 + 64 : k0;                 // last step re-reads its own slab (discarded)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        an[i] = *reinterpret_cast<const v4i*>(wrow + i * step16 + kn);
        bn[i] = *reinterpret_cast<const v4i*>(xrow + i * step16 + kn);
      }
#pragma unroll
      for (int uu = 0; uu < 4; ++uu)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[uu][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b[uu], a[tt], acc[uu][tt], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = an[i];
        b[i] = bn[i];
      }
    }
    // epilogue.  The MFMAs were issued with the ACTIVATION codes as the A operand, so D = (pixels x channels): lane
    // holds, per (uu, tt), the 4 consecutive PIXELS j0 + uu*16 + 4*(lane>>4) + r of channel co0 + tt*16 + (lane&15)
    // -> one 16-byte store per (uu, tt) when the four pixels sit in one sample, and the per-channel constants are
    // per LANE (read once per tt).
    // per-sample maxima of this tile go through an LDS table (a tile spans <= kStatSlots samples), then ONE global
    // atomic per touched sample: per-wave global atomics on the 128 hot addresses serialised in L2 and cost 80 % of the
    // kernel (100 k same-address atomics per launch)
    const unsigned s_base = (unsigned)(ctile * g.PT_B) / HW;
    if (has_stat) {
      __syncthreads();
      if (threadIdx.x < kStatSlots) k_stat[threadIdx.x] = 0u;
      __syncthreads();
    }
    int64_t ybase[4];
    bool cok[4], vec_ok[4];
    unsigned smps[4];
    float m[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) {
      const unsigned j = j0 + uu * 16 + 4 * (lane >> 4);               // first of this lane's 4 pixels
      cok[uu] = j < (unsigned)g.cols;
      const unsigned smp = cok[uu] ? j / HW : 0;
      const unsigned p = j - smp * HW;
      smps[uu] = smp;
      ybase[uu] = ((int64_t)smp * g.Cout) * plane_stride + p;
      vec_ok[uu] = cok[uu] && ((HW & 3u) == 0u) && (p + 3 < HW) && (j + 3 < (unsigned)g.cols);
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int col = wmi * 64 + tt * 16 + (lane & 15);                 // channel index inside the block
      const int co = co0 + tt * 16 + (lane & 15);
      const bool co_ok = co < g.Cout;
      const float sxw = k_sxw[col];
      const int zs = k_zs[col];
      const float bch = k_bias[col], bsc = k_bsc[col], bsh = k_bsh[col];
      const int64_t coff = (int64_t)(co_ok ? co : 0) * plane_stride;
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = (float)(acc[uu][tt][r] + zs) * sxw;
          if (bias != nullptr) v = v + bch;
          if (has_bn) {
            v = v * bsc;
            v = v + bsh;
          }
          o[r] = act_rt(v, act);
        }
        if (co_ok && vec_ok[uu]) {
          *reinterpret_cast<f4*>(y + ybase[uu] + coff) = (f4){o[0], o[1], o[2], o[3]};
          m[uu] = fmaxf(m[uu], fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        } else if (co_ok && cok[uu]) {
          // ragged: the 4 pixels may straddle a sample boundary or the end of the tensor
          const unsigned jb = j0 + uu * 16 + 4 * (lane >> 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned j = jb + r;
            if (j < (unsigned)g.cols) {
              const unsigned smp = j / HW;
              y[((int64_t)smp * g.Cout) * plane_stride + (j - smp * HW) + coff] = o[r];
              if (smp == smps[uu]) m[uu] = fmaxf(m[uu], fabsf(o[r]));
              else if (has_stat) {
                const unsigned slot = smp - s_base;
                if (slot < kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(fabsf(o[r])));
                else atomic_max_f32(stat_out + smp, fabsf(o[r]));
              }
            }
          }
        }
      }
    }
    if (has_stat) {
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        if (cok[uu]) {
          const unsigned slot = smps[uu] - s_base;
          if (slot < kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(m[uu]));
          else atomic_max_f32(stat_out + smps[uu], m[uu]);
        }
      }
      __syncthreads();
      if (threadIdx.x < kStatSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < (unsigned)(g.cols / HW))
        atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K10: integer GEMM on 8-bit codes with exact int32 results - the arithmetic core of the reference's stand-alone
// quantised convolution (nn/quantized_conv.py:134-151: im2col slices x reshaped filters, accumulated as integers).
//   out[n][co][p] = sum_k xc[n*L + p][k] * wc[co][k]  (+ zoff * wsum[co] when the activation codes were stored
//   re-centred by zoff = 128 to fit int8)
// xc: [cols_pad][K] int8, K-contiguous im2col rows (K % 32 == 0, zero padded);  wc: [rows_pad][K] int8.
// v_mfma_i32_32x32x32_i8 with the activation rows as the B operand: lane = pixel, so every store instruction writes two
// full 128-byte lines of the NCHW result.  Lanes past the last column re-read and re-store the last one (benign).
// Correctness first (this block is only exercised by the reference's tests): operands come straight from global / L2.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gemm_i8_codes_kernel(const int8_t* __restrict__ xc,
                                                               const int8_t* __restrict__ wc,
                                                               const int* __restrict__ wsum, int* __restrict__ out,
                                                               int64_t cols, int L, int K, int Cout, int zoff) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, pl = lane & 31;
  const int64_t tiles = (cols + 31) / 32;
  const int KT = K >> 5, CT = (Cout + 31) / 32;
  for (int64_t t = (int64_t)blockIdx.x * (kBlock / 64) + wave; t < tiles; t += (int64_t)gridDim.x * (kBlock / 64)) {
    int64_t col = t * 32 + pl;
    col = col < cols ? col : cols - 1;
    const int64_t smp = col / L;
    const int p = (int)(col - smp * L);
    const int8_t* xrow = xc + col * K + 16 * h;
    int* obase = out + (smp * Cout) * (int64_t)L + p;
    for (int ct = 0; ct < CT; ++ct) {
      v16i acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = ct * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        acc[r] = ch < Cout ? zoff * wsum[ch] : 0;
      }
      const int8_t* wrow = wc + (int64_t)(ct * 32 + pl) * K + 16 * h;     // A fragment: row pl, 16-byte half h
      for (int kt = 0; kt < KT; ++kt) {
        const v4i a = *reinterpret_cast<const v4i*>(wrow + kt * 32);
        const v4i b = *reinterpret_cast<const v4i*>(xrow + kt * 32);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = ct * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        if (ch < Cout) obase[(int64_t)ch * L] = acc[r];
      }
    }
  }
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

// ---------------------------------------------------------------------------------------------------------------
// K12: the evaluation counters of simulate_quantization.py:122-148 (pred = argmax(outputs, axis=1), first index on
// ties as MXNet's argmax; test_num_correct, label_counter[gt], correct_counter[gt]) in ONE launch: a wavefront per
// sample.  The tensor-library formulation is nine launch-bound kernels (~60 us per batch, 4 % of an evaluation step).
// counters = [n_correct, total, correct[classes], label[classes]] as floats: the increments are 1.0, exact below 2^24.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void eval_counters_kernel(const float* __restrict__ logits,
                                                               const long long* __restrict__ labels, int64_t n,
                                                               int classes, float* __restrict__ counters) {
  const int lane = threadIdx.x & 63;
  const int64_t smp = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (smp >= n) return;
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
  for (int i = lane; i < classes; i += 64) take(row[i], i);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bidx, off, 64);
    if (oi != 0x7FFFFFFF) take(ov, oi);
  }
  if (lane == 0) {
    const long long gt = labels[smp];
    atomicAdd(counters + 1, 1.0f);
    if (gt >= 0 && gt < classes) {
      atomicAdd(counters + 2 + classes + gt, 1.0f);
      if ((long long)bidx == gt) {
        atomicAdd(counters, 1.0f);
        atomicAdd(counters + 2 + gt, 1.0f);
      }
    }
  }
}

// Ordering fence for hand-pipelined loops: the asm memory clobber stops IR-level motion of loads, the sched_barrier the
// machine scheduler's (it sinks prefetches next to their use, or hoists every load of an unrolled loop to the top).
#define FQ_PIN()                         \
  do {                                   \
    asm volatile("" ::: "memory");        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

// ---------------------------------------------------------------------------------------------------------------
// K2s: the first ("stem") convolution of the ImageNet nets: dense 3x3, stride 2, pad 1, 3 input channels -> COUT,
// fp32 (the reference excludes the first convolution from quantisation), with BatchNorm/activation folded into the store
// and the per-sample max|y| the next (quantised) layer needs.  A lane owns one output pixel: it gathers its 27 inputs
// (zero padding by clamped address + select), then for every tap multiplies by the COUT weights of that tap, read from
// an LDS copy of the tap-major weights with broadcast ds_read_b128 (4 weights per read).  (Feeding the weights through
// SGPRs looked cheaper but hipcc hoists all 864 scalar loads and spills them to VGPR lanes: a v_readlane per FMA.)
// Stores are contiguous along the lanes for every channel.  MIOpen needed 0.14 ms + a separate 0.08 ms BatchNorm/ReLU/statistic
// pass for this layer at batch 128; the layer moves 77 MB in + 205 MB out.
// ---------------------------------------------------------------------------------------------------------------
template <int CIN, int COUT>
__global__ __launch_bounds__(kBlock) void stem_conv3x3s2_kernel(
    const float* __restrict__ x, const float* __restrict__ wt /*[CIN][3][3][COUT]*/, const float* __restrict__ bias,
    float* __restrict__ y, int H, int W, int Ho, int Wo, int tiles_per_wg, const float* __restrict__ bn_scale,
    const float* __restrict__ bn_shift, int act, float* __restrict__ stat_out) {
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) float wl[CIN * 9 * COUT];
  for (int i = threadIdx.x; i < CIN * 9 * COUT; i += kBlock) wl[i] = wt[i];
  __syncthreads();
  const int smp = blockIdx.y;
  const int HWo = Ho * Wo;
  const float* xs = x + (int64_t)smp * CIN * H * W;
  float* ys = y + (int64_t)smp * COUT * HWo;
  const bool has_bn = bn_scale != nullptr;
  float m = 0.0f;
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int pix = (blockIdx.x * tiles_per_wg + t) * kBlock + threadIdx.x;
    if ((blockIdx.x * tiles_per_wg + t) * kBlock >= HWo) break;          // uniform
    const bool valid = pix < HWo;
    const int pc = valid ? pix : HWo - 1;
    const int oy = pc / Wo, ox = pc - oy * Wo;
    float in[CIN][3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      const bool yin = iy >= 0 && iy < H;
      const int iyc = iy < 0 ? 0 : (iy < H ? iy : H - 1);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        const bool inb = yin && ix >= 0 && ix < W;
        const int ixc = ix < 0 ? 0 : (ix < W ? ix : W - 1);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const float v = xs[((int64_t)ci * H + iyc) * W + ixc];
          in[ci][ky][kx] = inb ? v : 0.0f;
        }
      }
    }
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float v = in[ci][ky][kx];
          const f4* wtap = reinterpret_cast<const f4*>(wl + ((ci * 3 + ky) * 3 + kx) * COUT);
          FQ_PIN();                              // one tap's weights at a time (else all 216 reads are hoisted: spills)
#pragma unroll
          for (int c4 = 0; c4 < COUT / 4; ++c4) {
            const f4 wv = wtap[c4];
            acc[4 * c4 + 0] = __builtin_fmaf(wv.x, v, acc[4 * c4 + 0]);
            acc[4 * c4 + 1] = __builtin_fmaf(wv.y, v, acc[4 * c4 + 1]);
            acc[4 * c4 + 2] = __builtin_fmaf(wv.z, v, acc[4 * c4 + 2]);
            acc[4 * c4 + 3] = __builtin_fmaf(wv.w, v, acc[4 * c4 + 3]);
          }
          // ... and the accumulators pinned per tap: otherwise the optimiser sinks every channel's 27 FMAs down to that
          // channel's store and keeps all 864 weights live instead
#pragma unroll
          for (int c8 = 0; c8 < COUT / 8; ++c8)
            asm volatile("" : "+v"(acc[8 * c8]), "+v"(acc[8 * c8 + 1]), "+v"(acc[8 * c8 + 2]), "+v"(acc[8 * c8 + 3]),
                              "+v"(acc[8 * c8 + 4]), "+v"(acc[8 * c8 + 5]), "+v"(acc[8 * c8 + 6]), "+v"(acc[8 * c8 + 7]));
        }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float v = acc[co];
      if (bias != nullptr) v = v + bias[co];
      if (has_bn) {
        v = v * bn_scale[co];
        v = v + bn_shift[co];
      }
      v = act_rt(v, act);
      if (valid) {
        ys[(int64_t)co * HWo + pix] = v;
        m = fmaxf(m, fabsf(v));
      }
    }
  }
  if (stat_out != nullptr) {
    m = block_max(m, red);
    if (threadIdx.x == 0) atomic_max_f32(stat_out + smp, m);
  }
}

// K2g: the two kernels above in ONE launch, no int8 intermediate in HBM.  A workgroup owns PX = 64*wn columns
// (column = (sample, pixel) flattened): phase 1 reads their fp32 activations once (coalesced along pixels), fake-quantises
// them and leaves the int8 codes, transposed to K-contiguous rows, in an LDS panel [PX][K]; phase 2 runs the integer
// GEMM of that panel against `nblk` blocks of 64*wm output channels: activation fragments come from LDS (ds_read_b128),
// weight fragments straight from global memory (L2 resident), both ping-pong buffered over an explicitly 2x unrolled K
// loop; the epilogue is K2f-B's.  Panel rows are XOR-swizzled in 16-byte chunks so that the 4-byte transposing writes
// and the 16-byte fragment reads are both (nearly) bank-conflict free without padding.
struct PwfGeom {
  int Cin, K, Cout, HW;
  int64_t cols;        // n * HW
  int wm, wn;          // wave grid: wm x wn waves of (64 channels) x (64 columns)
  int PX;              // columns per workgroup = 64 * wn
  int cblocks;         // ceil(Cout / (64 * wm))
  int csplit;          // workgroups sharing one column tile (each takes nblk channel blocks)
  int nblk;            // channel blocks per workgroup = ceil(cblocks / csplit)
  int swz;             // swizzle mask: 7 when K % 128 == 0, else 3
  int zoff;
};

__device__ __forceinline__ int pwf_panel_off(int row, int chunk, int K, int swz) {
  const int sw = swz == 7 ? ((row ^ (row >> 3)) & 7) : ((row >> 1) & 3);
  return row * K + (((chunk & ~swz) | ((chunk ^ sw) & swz)) << 4);
}

template <bool ONLINE, int G>
__global__ __launch_bounds__(kBlock, 3) void pwconv_fused_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wc, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwfGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out) {
  constexpr int kStatSlots = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char pwf_smem[];
  __shared__ unsigned k_stat[kStatSlots];
  int8_t* panel = reinterpret_cast<int8_t*>(pwf_smem);                 // PX * K bytes
  const int nconst = g.nblk * g.wm * 64;                                // per-channel constants of this workgroup
  const int zero_off = g.PX * g.K;                                      // 16 zero bytes: the activation fragment of
  float* k_sxw = reinterpret_cast<float*>(pwf_smem + (size_t)g.PX * g.K + 16);   // the phantom step of an odd K/64
  float* k_bias = k_sxw + nconst;
  float* k_bsc = k_bias + nconst;
  float* k_bsh = k_bsc + nconst;
  int* k_zs = reinterpret_cast<int*>(k_bsh + nconst);

  PW_STAMP(0);
#ifdef FQ_PW_TRACE
  if (threadIdx.x == 0 && g_pw_trace != nullptr)
    g_pw_trace[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) |
                                             (unsigned long long)__builtin_amdgcn_s_getreg(63492);
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wmi = wave % g.wm, wni = wave / g.wm;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const unsigned HW = (unsigned)g.HW;
  const int64_t plane_stride = (int64_t)g.HW;
  // workgroups are dealt to the 8 XCDs round-robin: the csplit workgroups that share a column tile (and re-read the
  // same activations) are placed on the SAME XCD, back to back, so that the repeats hit in that XCD's L2
  const int xcd = (int)(blockIdx.x & 7u);
  const int64_t slot = blockIdx.x >> 3;
  const int cs = (int)(slot % g.csplit);
  const int64_t ctile = (slot / g.csplit) * 8 + xcd;
  if (ctile * g.PX >= g.cols) return;                                  // grid is padded to 8 * csplit
  const int cb_first = cs * g.nblk;
  const unsigned jt0 = (unsigned)(ctile * g.PX);

  // ---- phase 1: quantise + transpose PX columns x K channels into the panel -------------------------------------
  // Units of (64 columns) x (64 channels); a thread owns 4 channels x 4 columns of each (four 16-byte loads).  G units
  // are loaded per group and two groups are in flight (register double buffer): the first group is issued BEFORE the
  // batch statistic is reduced, the next one before the current one is quantised.  Unit indices past the end are
  // clamped (loads are unconditional so that the compiler's waits only cover the older group).
  const int c = threadIdx.x & 15;            // pixel quad inside a 64-column group
  const int rq = threadIdx.x >> 4;           // channel quad inside a 64-channel group (0..15)
  const bool hw_vec = (HW & 3u) == 0u;
  const int ktiles = g.K >> 6;
  const int U = g.wn * ktiles;
  float va[G][4][4], vb[G][4][4];
  auto issue = [&](int u0, float (&v)[G][4][4]) {
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      int u = u0 + gi;
      u = u < U ? u : U - 1;
      const int pb = u / ktiles, ct = u - pb * ktiles;
      const unsigned jb = jt0 + pb * 64 + c * 4;
      const int ci0 = ct * 64 + rq * 4;
      if (ci0 >= g.Cin) continue;                                       // zero padding of K: nothing to read
      if (hw_vec) {
        unsigned j = jb < (unsigned)g.cols ? jb : 0u;                  // cols % 4 == 0 here: the quad is all in or all out
        const unsigned smp = j / HW;
        const float* base = x + (int64_t)smp * g.Cin * plane_stride + (j - smp * HW);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int cic = ci0 + k < g.Cin ? ci0 + k : g.Cin - 1;
          const f4 r = *reinterpret_cast<const f4*>(base + (int64_t)cic * plane_stride);
          v[gi][k][0] = r.x; v[gi][k][1] = r.y; v[gi][k][2] = r.z; v[gi][k][3] = r.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned j = jb + e;
          j = j < (unsigned)g.cols ? j : (unsigned)g.cols - 1;
          const unsigned smp = j / HW;
          const float* base = x + (int64_t)smp * g.Cin * plane_stride + (j - smp * HW);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int cic = ci0 + k < g.Cin ? ci0 + k : g.Cin - 1;
            v[gi][k][e] = base[(int64_t)cic * plane_stride];
          }
        }
      }
    }
  };
#ifdef FQ_PW_TRACE
  const bool skip_loads = (g_pw_dbg & 2) != 0;
#else
  constexpr bool skip_loads = false;
#endif
  if (!skip_loads) issue(0, va);
  __builtin_amdgcn_sched_barrier(0);
  const float max_ = ONLINE ? batch_mean_dev(in_stat, n) : in_thr[0];
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  if (ONLINE && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
  const float sx = q.scale;
  PW_STAMP(1);
  auto process = [&](int u0, float (&v)[G][4][4]) {
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      const int u = u0 + gi;
      if (u < U) {
        const int pb = u / ktiles, ct = u - pb * ktiles;
        const int ci0 = ct * 64 + rq * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned packed = 0;
          if (ci0 < g.Cin) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              int code = fq_code_int(v[gi][k][e], q) - g.zoff;
              if (ci0 + k >= g.Cin) code = 0;
              packed |= ((unsigned)code & 0xFFu) << (8 * k);
            }
          }
          const int row = pb * 64 + c * 4 + e;
          *reinterpret_cast<unsigned*>(panel + pwf_panel_off(row, ct * 4 + (rq >> 2), g.K, g.swz) + (rq & 3) * 4) = packed;
        }
      }
    }
  };
  for (int u0 = 0; u0 < U; u0 += 2 * G) {
    if (!skip_loads) issue(u0 + G, vb);
    __builtin_amdgcn_sched_barrier(0);
    process(u0, va);
    __builtin_amdgcn_sched_barrier(0);
    if (!skip_loads) issue(u0 + 2 * G, va);
    __builtin_amdgcn_sched_barrier(0);
    process(u0 + G, vb);
    __builtin_amdgcn_sched_barrier(0);
  }
  // constants of all channel blocks of this workgroup, and the zero chunk
  if (threadIdx.x < 4) reinterpret_cast<int*>(panel + zero_off)[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < nconst; i += kBlock) {
    const int co = cb_first * g.wm * 64 + i;
    const int coc = co < g.Cout ? co : 0;
    k_sxw[i] = sx * wscale[coc];
    k_zs[i] = g.zoff * wsum[coc];
    k_bias[i] = bias != nullptr ? bias[coc] : 0.0f;
    k_bsc[i] = has_bn ? bn_scale[coc] : 1.0f;
    k_bsh[i] = has_bn ? bn_shift[coc] : 0.0f;
  }
  PW_STAMP(2);
  __syncthreads();
  PW_STAMP(3);

  // ---- phase 2: integer GEMM of the panel against this workgroup's channel blocks ---------------------------------
  const unsigned j0 = jt0 + wni * 64;
  const unsigned s_base = jt0 / HW;
  int64_t ybase[4];
  bool cok[4], vec_ok[4];
  unsigned smps[4];
#pragma unroll
  for (int uu = 0; uu < 4; ++uu) {
    const unsigned j = j0 + uu * 16 + 4 * (lane >> 4);                 // first of this lane's 4 pixels
    cok[uu] = j < (unsigned)g.cols;
    const unsigned smp = cok[uu] ? j / HW : 0;
    const unsigned p = j - smp * HW;
    smps[uu] = smp;
    ybase[uu] = ((int64_t)smp * g.Cout) * plane_stride + p;
    vec_ok[uu] = cok[uu] && ((HW & 3u) == 0u) && (p + 3 < HW) && (j + 3 < (unsigned)g.cols);
  }
  // LDS byte offsets of this lane's four activation fragments (row fixed, chunk advances by 4 per K step)
  int brow[4];
#pragma unroll
  for (int uu = 0; uu < 4; ++uu) brow[uu] = wni * 64 + uu * 16 + (lane & 15);
  const int ksteps = g.K >> 6;
  const int64_t step16 = (int64_t)16 * g.K;

  for (int bi = 0; bi < g.nblk; ++bi) {
    const int cb = cb_first + bi;
    if (cb >= g.cblocks) break;
    const int co0 = (cb * g.wm + wmi) * 64;
    const int co_ld = co0 < g.Cout ? co0 : 0;                           // see K2f-B
    const int8_t* wrow = wc + (int64_t)(co_ld + (lane & 15)) * g.K + 16 * (lane >> 4);
    // accumulators start at the zero-point correction zoff * sum(w codes) of their channel (D: lane -> channel)
    v4i acc[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int zs = k_zs[(bi * g.wm + wmi) * 64 + b * 16 + (lane & 15)];
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a][b] = (v4i){zs, zs, zs, zs};
    }
    v4i a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a0[i] = *reinterpret_cast<const v4i*>(wrow + i * step16);
      b0[i] = *reinterpret_cast<const v4i*>(panel + pwf_panel_off(brow[i], (lane >> 4), g.K, g.swz));
    }
    // Two K steps per iteration on two register sets, NO branch inside: with a conditional load or MFMA group the
    // compiler falls back to vmcnt(0) right after issuing the prefetch.  Indices past the end are clamped (the data
    // is discarded), and for an odd number of steps the second group of the last iteration multiplies by the zero
    // chunk instead of being skipped.
    for (int ks = 0; ks < ksteps; ks += 2) {
      const bool real1 = ks + 1 < ksteps;
      const int k1 = real1 ? ks + 1 : ks;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a1[i] = *reinterpret_cast<const v4i*>(wrow + i * step16 + k1 * 64);
        const int off = pwf_panel_off(brow[i], k1 * 4 + (lane >> 4), g.K, g.swz);
        b1[i] = *reinterpret_cast<const v4i*>(panel + (real1 ? off : zero_off));
      }
      __builtin_amdgcn_sched_barrier(0);        // keep the prefetch ABOVE the MFMA group (the scheduler sinks loads
                                                // next to their use to save registers, which serialises on latency)
#pragma unroll
      for (int uu = 0; uu < 4; ++uu)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[uu][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b0[uu], a0[tt], acc[uu][tt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const int k2 = ks + 2 < ksteps ? ks + 2 : ks;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a0[i] = *reinterpret_cast<const v4i*>(wrow + i * step16 + k2 * 64);
        b0[i] = *reinterpret_cast<const v4i*>(panel + pwf_panel_off(brow[i], k2 * 4 + (lane >> 4), g.K, g.swz));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int uu = 0; uu < 4; ++uu)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[uu][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1[uu], a1[tt], acc[uu][tt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    // epilogue (as K2f-B): D = (pixels x channels); lane holds 4 consecutive pixels of one channel per (uu, tt)
    if (bi == 0) PW_STAMP(4);
    if (has_stat) {
      __syncthreads();
      if (threadIdx.x < kStatSlots) k_stat[threadIdx.x] = 0u;
      __syncthreads();
    }
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    // the per-element arithmetic is specialised at compile time for the combinations the converted nets produce (no
    // bias + folded BN + ReLU / ReLU6 / linear): with run-time flags every output cost 14 VALU instructions (selects
    // after each optional step) instead of 6, and this kernel is instruction-bound
    auto epilogue = [&](auto bias_c, auto bn_c, auto act_c) {
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int col = (bi * g.wm + wmi) * 64 + tt * 16 + (lane & 15);   // index into this workgroup's constants
      const int co = co0 + tt * 16 + (lane & 15);
      const bool co_ok = co < g.Cout;
      const float sxw = k_sxw[col];
      const float bch = k_bias[col], bsc = k_bsc[col], bsh = k_bsh[col];
      const int64_t coff = (int64_t)(co_ok ? co : 0) * plane_stride;
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = (float)acc[uu][tt][r] * sxw;
          if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) v = v + bch;
          if (BN_M == 1 || (BN_M < 0 && has_bn)) {
            v = v * bsc;
            v = v + bsh;
          }
          o[r] = ACT_M < 0 ? act_rt(v, act) : act_rt(v, ACT_M);
        }
        if (co_ok && vec_ok[uu]) {
#ifdef FQ_PW_TRACE
          if (!(g_pw_dbg & 1))
#endif
          *reinterpret_cast<f4*>(y + ybase[uu] + coff) = (f4){o[0], o[1], o[2], o[3]};
          m[uu] = fmaxf(m[uu], fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        } else if (co_ok && cok[uu]) {
          // ragged: the 4 pixels may straddle a sample boundary or the end of the tensor.  The opaque asm keeps the
          // compiler from hoisting 16 copies of this address arithmetic out of the loops (it spilled them)
          unsigned jb = j0 + uu * 16 + 4 * (lane >> 4);
          asm volatile("" : "+v"(jb));
          unsigned sr = jb / HW;
          unsigned pr = jb - sr * HW;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (jb + r < (unsigned)g.cols) {
              while (pr >= HW) {
                pr -= HW;
                ++sr;
              }
              y[((int64_t)sr * g.Cout) * plane_stride + pr + coff] = o[r];
              if (sr == smps[uu]) m[uu] = fmaxf(m[uu], fabsf(o[r]));
              else if (has_stat) {
                const unsigned slot = sr - s_base;
                if (slot < kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(fabsf(o[r])));
                else atomic_max_f32(stat_out + sr, fabsf(o[r]));
              }
            }
            ++pr;
          }
        }
      }
    }
    };
    using std::integral_constant;
#ifdef FQ_PW_GENERIC_EPI
    if (true)
      epilogue(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{});
    else
#endif
    if (bias == nullptr && has_bn && act == FQ_ACT_RELU)
      epilogue(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU>{});
    else if (bias == nullptr && has_bn && act == FQ_ACT_RELU6)
      epilogue(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU6>{});
    else if (bias == nullptr && has_bn && act == FQ_ACT_NONE)
      epilogue(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_NONE>{});
    else
      epilogue(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{});
    if (has_stat) {
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        if (cok[uu]) {
          const unsigned slot = smps[uu] - s_base;
          if (slot < kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(m[uu]));
          else atomic_max_f32(stat_out + smps[uu], m[uu]);
        }
      }
      __syncthreads();
      if (threadIdx.x < kStatSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < (unsigned)(g.cols / HW))
        atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
    }
  }
  PW_STAMP(5);
}

// K2h: streaming form of the pointwise convolution for layers whose whole weight matrix fits in LDS (Cout*K <= 64 KB:
// the large-plane layers, where the bytes are).  v_mfma_i32_32x32x32_i8 with the ACTIVATIONS as the B operand: lane l
// owns pixel l&31 and needs the 16 consecutive channels 16*(l>>5).. of a 32-channel slab in its registers — which is
// exactly what 16 plain dword loads of NCHW give it (for a fixed channel, 32 lanes read 128 contiguous bytes).  So
// there is no transposition at all: load, fake-quantise, pack four codes per dword, multiply.  D comes out with
// lane = pixel, register = channel: every store instruction writes two full 128-byte lines.  No LDS traffic for the
// activations, no barrier inside the loop, ~110 VGPRs: waves are independent and hide each other's latencies, unlike
// the panel kernel above whose workgroups move through load / quantise / multiply / store phases in lock step.
// Weights sit in LDS in fragment order (1 KB per (32 channels x 32 k) fragment, read with one conflict-free
// ds_read_b128 per lane); per-channel constants sit next to them and are read 4 channels at a time (D holds channels
// 8*(r/4) + 4*(l>>5) + r%4 in register r).  A wavefront walks a contiguous range of 32-pixel tiles; the loads of the
// next slab / tile are issued before the current one is quantised (two register buffers).

struct PwsGeom {
  int Cin, K, Cout, CT, HW;   // K: row stride of the weight codes (cin_pad); CT = ceil(Cout / 32)
  int64_t cols, tiles;        // n * HW, ceil(cols / 32)
  int zoff;
};

template <int KT>
__global__ __launch_bounds__(kBlock, 2) void pwconv_stream_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wc, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwsGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out) {
  constexpr int kSlots = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char pws_smem[];
  __shared__ unsigned k_stat[kSlots];
  v4i* ldsA = reinterpret_cast<v4i*>(pws_smem);                        // [CT][KT][64] fragments
  const int nch = g.CT * 32;
  float* c_sxw = reinterpret_cast<float*>(pws_smem + (size_t)g.CT * KT * 1024);
  float* c_bsc = c_sxw + nch;
  float* c_bsh = c_bsc + nch;
  float* c_bias = c_bsh + nch;
  int* c_zs = reinterpret_cast<int*>(c_bias + nch);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, pl = lane & 31;
  const unsigned HW = (unsigned)g.HW, cols = (unsigned)g.cols;
  const int64_t plane = (int64_t)g.HW;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const int64_t nwaves = (int64_t)gridDim.x * 4, wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t t_begin = g.tiles * wid / nwaves, t_end = g.tiles * (wid + 1) / nwaves;
  unsigned s_base;
  {
    const int64_t t0 = g.tiles * ((int64_t)blockIdx.x * 4) / nwaves;
    const unsigned j0 = (unsigned)t0 * 32u;
    s_base = (j0 < cols ? j0 : cols - 1) / HW;
  }

  // per-tile pixel record of this lane
  struct Pix { unsigned smp, p; bool valid; };
  auto pix_of = [&](int64_t t) __attribute__((always_inline)) {
    Pix r;
    unsigned j = (unsigned)t * 32u + (unsigned)pl;
    r.valid = j < cols;
    j = r.valid ? j : cols - 1;
    r.smp = j / HW;
    r.p = j - r.smp * HW;
    return r;
  };
  // addresses: wave-uniform base (SGPR arithmetic) + ONE 32-bit per-lane byte offset per tile (host checks the tensors
  // are < 4 GB).  With 64-bit per-lane pointers the compiler materialised an address pair per load and spilled.
  auto lane_off = [&](const Pix& px, int c0) __attribute__((always_inline)) {
    return (unsigned)((((int64_t)px.smp * g.Cin + c0) * plane + px.p) * 4);
  };
  auto issue = [&](const Pix& px, int kt, float (&v)[16]) __attribute__((always_inline)) {
    const int cg = kt * 32 + 16 * h;                                   // this half-wave's 16 channels of slab kt
    const unsigned off = lane_off(px, cg < g.Cin ? 16 * h : 0);        // padded group: read the valid half, discarded
    const char* ub = reinterpret_cast<const char*>(x) + (int64_t)kt * 32 * plane * 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const float*>(ub + (int64_t)i * plane * 4 + off);
  };

  float bufa[16], bufb[16];
  int64_t t_first = t_begin < g.tiles ? t_begin : g.tiles - 1;
  Pix nxt = pix_of(t_first);
  issue(nxt, 0, bufa);                                                  // in flight during the whole set-up
  FQ_PIN();

  const float max_ = in_stat != nullptr ? batch_mean_dev(in_stat, n) : in_thr[0];
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  if (in_stat != nullptr && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
  const float sx = q.scale;
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;
  // weights -> fragment order: fragment (ct, kt), lane (row % 32) + 32 * (16-byte chunk % 2)
  for (int idx = threadIdx.x; idx < nch * KT * 2; idx += kBlock) {
    const int row = idx / (KT * 2), kc = idx - row * (KT * 2);
    const v4i wv = *reinterpret_cast<const v4i*>(wc + (int64_t)row * g.K + kc * 16);
    ldsA[(((row >> 5) * KT + (kc >> 1)) << 6) + (row & 31) + 32 * (kc & 1)] = wv;
  }
  for (int i = threadIdx.x; i < nch; i += kBlock) {
    const bool ok = i < g.Cout;
    const int ic = ok ? i : 0;
    c_sxw[i] = sx * wscale[ic];
    c_zs[i] = ok ? g.zoff * wsum[ic] : 0;
    c_bias[i] = bias != nullptr ? bias[ic] : 0.0f;
    c_bsc[i] = has_bn ? bn_scale[ic] : 1.0f;
    c_bsh[i] = has_bn ? bn_shift[ic] : 0.0f;
  }
  __syncthreads();

  v4i bfrag[KT];
  auto quant = [&](int kt, const float (&v)[16]) __attribute__((always_inline)) {
    const bool gvalid = kt * 32 + 16 * h < g.Cin;
    v4i f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int packed = pack4_codes(fq_code_int(v[4 * d + 0], q), fq_code_int(v[4 * d + 1], q),
                                     fq_code_int(v[4 * d + 2], q), fq_code_int(v[4 * d + 3], q), 128 - g.zoff);
      f[d] = gvalid ? packed : 0;
    }
    // pin the quantisation HERE: it is pure arithmetic whose results are only needed by the MFMAs, and the optimiser
    // otherwise sinks it below every prefetch, keeping all 16 * KT loaded values live (256 VGPRs + spills)
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    bfrag[kt] = f;
  };
  auto tile_done = [&](const Pix& px, auto bias_c, auto bn_c, auto act_c) __attribute__((always_inline)) {
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
    const unsigned yoff = (unsigned)((((int64_t)px.smp * g.Cout + 4 * h) * plane + px.p) * 4);
    float m = 0.0f;
#pragma unroll 1
    for (int ct = 0; ct < g.CT; ++ct) {
      v16i acc;
      const int cb = ct * 32 + 4 * h;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const v4i z = *reinterpret_cast<const v4i*>(c_zs + cb + 8 * gq);
        acc[4 * gq + 0] = z.x; acc[4 * gq + 1] = z.y; acc[4 * gq + 2] = z.z; acc[4 * gq + 3] = z.w;
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ldsA[((ct * KT + kt) << 6) + lane], bfrag[kt], acc, 0, 0, 0);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int c0 = cb + 8 * gq;                                     // channels c0 .. c0+3 in registers 4gq .. 4gq+3
        const f4 sxw = *reinterpret_cast<const f4*>(c_sxw + c0);
        const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
        const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
        f4 bch = (f4){0.f, 0.f, 0.f, 0.f};
        if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) bch = *reinterpret_cast<const f4*>(c_bias + c0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = (float)acc[4 * gq + r] * sxw[r];
          if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) v = v + bch[r];
          if (BN_M == 1 || (BN_M < 0 && has_bn)) {
            v = v * bsc[r];
            v = v + bsh[r];
          }
          v = ACT_M < 0 ? act_rt(v, act) : act_rt(v, ACT_M);
          // no masks: lanes past the end hold a copy of the last pixel (clamped loads) and re-store its values, and the
          // host guarantees Cout % 32 == 0
          *reinterpret_cast<float*>(reinterpret_cast<char*>(y) + (int64_t)(ct * 32 + 8 * gq + r) * plane * 4 + yoff) = v;
          m = fmaxf(m, fabsf(v));
        }
      }
    }
    if (has_stat) {
      const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)px.smp);
      const bool uniform = __all(!px.valid || px.smp == s0);
      if (uniform) {
        const float wm = wave_max(px.valid ? m : 0.0f);
        if (lane == 0) {
          const unsigned slot = s0 - s_base;
          if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
          else atomic_max_f32(stat_out + s0, wm);
        }
      } else if (px.valid) {
        const unsigned slot = px.smp - s_base;
        if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
        else atomic_max_f32(stat_out + px.smp, m);
      }
    }
  };
  // one tile: its first slab is already in `first`; the prefetch of the following slab / tile alternates buffers
  auto run_tile = [&](int64_t t, float (&first)[16], float (&second)[16], auto bias_c, auto bn_c, auto act_c) __attribute__((always_inline)) {
    const Pix cur = nxt;
    const int64_t tn = t + 1 < g.tiles ? t + 1 : g.tiles - 1;
    nxt = pix_of(tn);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      float (&mine)[16] = (kt & 1) ? second : first;
      float (&other)[16] = (kt & 1) ? first : second;
      if (kt + 1 < KT) issue(cur, kt + 1, other);
      else issue(nxt, 0, other);
      FQ_PIN();
      quant(kt, mine);
      FQ_PIN();
    }
    tile_done(cur, bias_c, bn_c, act_c);
    FQ_PIN();
  };
  auto run_all = [&](auto bias_c, auto bn_c, auto act_c) __attribute__((always_inline)) {
    if (KT & 1) {                                                       // the buffers swap roles from tile to tile
      int64_t t = t_begin;
      for (; t + 1 < t_end; t += 2) {
        run_tile(t, bufa, bufb, bias_c, bn_c, act_c);
        run_tile(t + 1, bufb, bufa, bias_c, bn_c, act_c);
      }
      if (t < t_end) run_tile(t, bufa, bufb, bias_c, bn_c, act_c);
    } else {
      for (int64_t t = t_begin; t < t_end; ++t) run_tile(t, bufa, bufb, bias_c, bn_c, act_c);
    }
  };
  using std::integral_constant;
  if (bias == nullptr && has_bn && act == FQ_ACT_RELU)
    run_all(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU>{});
  else if (bias == nullptr && has_bn && act == FQ_ACT_RELU6)
    run_all(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU6>{});
  else if (bias == nullptr && has_bn && act == FQ_ACT_NONE)
    run_all(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_NONE>{});
  else
    run_all(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{});

  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < cols / HW)
      atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
}

// K2i: the streaming form for layers whose weights do NOT fit in LDS (K = 256 / 512, Cout up to 1024).  Activations
// exactly as K2h (lane = pixel, codes built in registers: K/32 fragments of 4 VGPRs stay resident for the whole tile),
// the weight matrix streams through LDS in chunks of CTC channel tiles (32 KB), double buffered: while the wavefronts
// multiply chunk c (A fragments by ds_read_b128, B fragments from registers) and store its outputs, every thread has
// chunk c+1's 16-byte pieces in flight from L2, and writes them to the other LDS buffer before the (single) barrier of
// the iteration.  The chunk sequence is cyclic, so the pipeline runs across tile batches.  `wsplit` wavefronts share one
// 32-pixel tile and divide a chunk's channel tiles among themselves when there are too few pixels to give every
// wavefront its own tile (7x7 planes); they quantise that tile redundantly.
struct PwcGeom {
  int Cin, K, Cout, CT, HW;   // K: row stride of the weight codes; CT = ceil(Cout / 32)
  int CTC, NC;                // channel tiles per chunk, chunks = ceil(CT / CTC)
  int wsplit;                 // wavefronts per tile: 1, 2 or 4
  int rows;                   // rows of the weight code buffer (Cout rounded up to 64)
  int64_t cols, tiles, batches;   // n * HW, ceil(cols / 32), ceil(tiles / (4 / wsplit))
  int zoff;
};

template <int KT, int PIECES>
__global__ __launch_bounds__(kBlock, 2) void pwconv_chunk_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wc, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwcGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out, float* __restrict__ sink) {
  constexpr int kSlots = 8;
  constexpr int kMaxPieces = PIECES;                                    // 16-byte pieces per thread: CTC * KT / 4
  extern __shared__ __attribute__((aligned(16))) unsigned char pwc_smem[];
  __shared__ unsigned k_stat[kSlots];
  const int chunk_frags = g.CTC * KT;                                   // 1 KB fragments per chunk
  const int nchc = g.CTC * 32;                                          // channels per chunk
  const size_t buf_bytes = (size_t)chunk_frags * 1024 + (size_t)nchc * 5 * sizeof(float);
  auto bufA = [&](int b) __attribute__((always_inline)) { return reinterpret_cast<v4i*>(pwc_smem + (size_t)b * buf_bytes); };
  auto bufC = [&](int b) __attribute__((always_inline)) {
    return reinterpret_cast<float*>(pwc_smem + (size_t)b * buf_bytes + (size_t)chunk_frags * 1024);
  };

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, pl = lane & 31;
  const unsigned HW = (unsigned)g.HW, cols = (unsigned)g.cols;
  const int64_t plane = (int64_t)g.HW;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const int tiles_per_wg = 4 / g.wsplit;
  const int my_tile = wave / g.wsplit, sub = wave - my_tile * g.wsplit;
  const int64_t b_begin = g.batches * blockIdx.x / gridDim.x, b_end = g.batches * (blockIdx.x + 1) / gridDim.x;
  unsigned s_base;
  {
    const unsigned j0 = (unsigned)(b_begin * tiles_per_wg) * 32u;
    s_base = (j0 < cols ? j0 : cols - 1) / HW;
  }

  PW_STAMP(0);
  const float max_ = in_stat != nullptr ? batch_mean_dev(in_stat, n) : in_thr[0];
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  if (in_stat != nullptr && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
  const float sx = q.scale;
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;

  // ---- weight chunk staging (global -> registers -> LDS, fragment order) ---------------------------------------------
  v4i stage[kMaxPieces];
  float cst[5];
  const unsigned w_lane_off = (unsigned)(pl * g.K + h * 16);            // row pl, 16-byte half h of a 32-byte slab
  auto chunk_issue = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i) {
      const int f = wave + 4 * i;                                       // fragment inside the chunk (wave-uniform)
      const int ctl = f / KT, kt = f - ctl * KT;
      int row0 = (c * g.CTC + ctl) * 32;
      row0 = row0 + 32 <= g.rows ? row0 : g.rows - 32;                  // tiles past the padded buffer: discarded channels
      stage[i] = *reinterpret_cast<const v4i*>(wc + ((int64_t)row0 * g.K + kt * 32) + w_lane_off);
    }
    // per-channel constants: RAW loads only here (any arithmetic on them would make the compiler wait for them - and,
    // the counter being in-order, for everything issued before - right at the top of the iteration)
    if ((int)threadIdx.x < nchc) {
      const int ch = c * nchc + threadIdx.x;
      const int cc = ch < g.Cout ? ch : 0;
      cst[0] = wscale[cc];
      cst[1] = has_bn ? bn_scale[cc] : 1.0f;
      cst[2] = has_bn ? bn_shift[cc] : 0.0f;
      cst[3] = bias != nullptr ? bias[cc] : 0.0f;
      cst[4] = __int_as_float(wsum[cc]);
    }
  };
  auto chunk_commit = [&](int b) __attribute__((always_inline)) {
    v4i* A = bufA(b);
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i) A[((wave + 4 * i) << 6) + lane] = stage[i];
    if ((int)threadIdx.x < nchc) {
      float* C = bufC(b);
      C[0 * nchc + threadIdx.x] = sx * cst[0];
      C[1 * nchc + threadIdx.x] = cst[1];
      C[2 * nchc + threadIdx.x] = cst[2];
      C[3 * nchc + threadIdx.x] = cst[3];
      C[4 * nchc + threadIdx.x] = __int_as_float(g.zoff * __float_as_int(cst[4]));
    }
  };

  struct Pix { unsigned smp, p; bool valid; };
  auto pix_of = [&](int64_t t) __attribute__((always_inline)) {
    Pix r;
    unsigned j = (unsigned)t * 32u + (unsigned)pl;
    r.valid = t < g.tiles && j < cols;
    j = r.valid ? j : cols - 1;
    r.smp = j / HW;
    r.p = j - r.smp * HW;
    return r;
  };
  auto issue = [&](const Pix& px, int kt, float (&v)[16]) __attribute__((always_inline)) {
    const int cg = kt * 32 + 16 * h;
    const unsigned off = (unsigned)((((int64_t)px.smp * g.Cin + (cg < g.Cin ? 16 * h : 0)) * plane + px.p) * 4);
    const char* ub = reinterpret_cast<const char*>(x) + (int64_t)kt * 32 * plane * 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const float*>(ub + (int64_t)i * plane * 4 + off);
  };
  v4i bfrag[KT];
  auto quant = [&](int kt, const float (&v)[16]) __attribute__((always_inline)) {
    const bool gvalid = kt * 32 + 16 * h < g.Cin;
    v4i f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int packed = pack4_codes(fq_code_int(v[4 * d + 0], q), fq_code_int(v[4 * d + 1], q),
                                     fq_code_int(v[4 * d + 2], q), fq_code_int(v[4 * d + 3], q), 128 - g.zoff);
      f[d] = gvalid ? packed : 0;
    }
    // pin the quantisation HERE: it is pure arithmetic whose results are only needed by the MFMAs, and the optimiser
    // otherwise sinks it below every prefetch, keeping all 16 * KT loaded values live (256 VGPRs + spills)
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    bfrag[kt] = f;
  };

  PW_STAMP(1);
  chunk_issue(0);
  chunk_commit(0);
  __syncthreads();
  PW_STAMP(2);

  auto run = [&](auto bias_c, auto bn_c, auto act_c) __attribute__((always_inline)) {
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
    float bufa[16], bufb[16], bufc[16];
    for (int64_t b = b_begin; b < b_end; ++b) {
      const Pix px = pix_of(b * tiles_per_wg + my_tile);
      // phase Q: the tile's K/32 slabs -> B fragments.  Three load buffers, two slabs (32 dwords per lane) in flight:
      // with one workgroup per CU a single slab in flight left every quantisation waiting a full HBM latency
      issue(px, 0, bufa);
      issue(px, 1 < KT ? 1 : 0, bufb);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int nk = kt + 2 < KT ? kt + 2 : KT - 1;
        if (kt % 3 == 0) { issue(px, nk, bufc); FQ_PIN(); quant(kt, bufa); }
        else if (kt % 3 == 1) { issue(px, nk, bufa); FQ_PIN(); quant(kt, bufb); }
        else { issue(px, nk, bufb); FQ_PIN(); quant(kt, bufc); }
        FQ_PIN();
      }
      const unsigned yoff = (unsigned)((((int64_t)px.smp * g.Cout + 4 * h) * plane + px.p) * 4);
      float m = 0.0f;
      if (b == b_begin) PW_STAMP(3);
      for (int c = 0; c < g.NC; ++c) {
        const int cur = c & 1;                                          // NC is even or 1 (host): buffers line up across batches
        chunk_issue(c + 1 < g.NC ? c + 1 : 0);
        FQ_PIN();
        const v4i* A = bufA(g.NC == 1 ? 0 : cur);
        const float* C = bufC(g.NC == 1 ? 0 : cur);
        // two channel tiles at a time on independent accumulators: one accumulator chained 16 dependent MFMAs, each
        // waiting out the previous one's full latency
        auto load_zs = [&](int ctl, v16i& acc) __attribute__((always_inline)) {
          const int cb = ctl * 32 + 4 * h;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const v4i z = *reinterpret_cast<const v4i*>(C + 4 * nchc + cb + 8 * gq);
            acc[4 * gq + 0] = z.x; acc[4 * gq + 1] = z.y; acc[4 * gq + 2] = z.z; acc[4 * gq + 3] = z.w;
          }
        };
        // No masks in the epilogue: lanes past the end of the tensor hold a copy of the LAST pixel (clamped loads), so
        // they recompute and re-store that pixel's values (benign duplicates); the host guarantees Cout % 32 == 0 and
        // CT % CTC == 0.  Masked stores cost 2 selects + 64-bit address arithmetic per output - more than the MFMAs.
        auto finish = [&](int ctl, const v16i& acc) __attribute__((always_inline)) {
          const int cb = ctl * 32 + 4 * h;                              // channel index inside the chunk
          char* ybase = reinterpret_cast<char*>(y) + (int64_t)(c * nchc + ctl * 32) * plane * 4;   // wave-uniform
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int c0 = cb + 8 * gq;
            const f4 sxw = *reinterpret_cast<const f4*>(C + c0);
            const f4 bsc = *reinterpret_cast<const f4*>(C + nchc + c0);
            const f4 bsh = *reinterpret_cast<const f4*>(C + 2 * nchc + c0);
            f4 bch = (f4){0.f, 0.f, 0.f, 0.f};
            if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) bch = *reinterpret_cast<const f4*>(C + 3 * nchc + c0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = (float)acc[4 * gq + r] * sxw[r];
              if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) v = v + bch[r];
              if (BN_M == 1 || (BN_M < 0 && has_bn)) {
                v = v * bsc[r];
                v = v + bsh[r];
              }
              v = ACT_M < 0 ? act_rt(v, act) : act_rt(v, ACT_M);
              *reinterpret_cast<float*>(ybase + (int64_t)(8 * gq + r) * plane * 4 + yoff) = v;
              m = fmaxf(m, fabsf(v));
            }
          }
        };
        // A fragments are read two K steps ahead of their MFMAs (LDS latency ~ two 32-cycle MFMAs)
#pragma unroll 1
        for (int ctl = sub; ctl < g.CTC; ctl += 2 * g.wsplit) {
          const int ctl2 = ctl + g.wsplit;
          if (ctl2 < g.CTC) {
            v16i acc0, acc1;
            load_zs(ctl, acc0);
            load_zs(ctl2, acc1);
            const v4i* A0 = A + ((ctl * KT) << 6) + lane;
            const v4i* A1 = A + ((ctl2 * KT) << 6) + lane;
            v4i fa[3], fb[3];
            fa[0] = A0[0]; fb[0] = A1[0];
            fa[1] = A0[(KT > 1 ? 1 : 0) << 6]; fb[1] = A1[(KT > 1 ? 1 : 0) << 6];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
              const int kn = kt + 2 < KT ? kt + 2 : KT - 1;
              fa[(kt + 2) % 3] = A0[kn << 6];
              fb[(kt + 2) % 3] = A1[kn << 6];
              acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[kt % 3], bfrag[kt], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[kt % 3], bfrag[kt], acc1, 0, 0, 0);
            }
            finish(ctl, acc0);
            finish(ctl2, acc1);
          } else {
            v16i acc0;
            load_zs(ctl, acc0);
            const v4i* A0 = A + ((ctl * KT) << 6) + lane;
            v4i fa[3];
            fa[0] = A0[0];
            fa[1] = A0[(KT > 1 ? 1 : 0) << 6];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
              const int kn = kt + 2 < KT ? kt + 2 : KT - 1;
              fa[(kt + 2) % 3] = A0[kn << 6];
              acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[kt % 3], bfrag[kt], acc0, 0, 0, 0);
            }
            finish(ctl, acc0);
          }
        }
        FQ_PIN();
        if (b == b_begin && c == 0) PW_STAMP(6);
        if (g.NC > 1) {
          chunk_commit(cur ^ 1);
          __syncthreads();
        }
        if (b == b_begin && c == 0) PW_STAMP(7);
      }
      if (b == b_begin) PW_STAMP(4);
      if (has_stat) {
        const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)px.smp);
        const bool uniform = __all(!px.valid || px.smp == s0);
        if (uniform) {
          const float wm = wave_max(px.valid ? m : 0.0f);
          if (lane == 0) {
            const unsigned slot = s0 - s_base;
            if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
            else atomic_max_f32(stat_out + s0, wm);
          }
        } else if (px.valid) {
          const unsigned slot = px.smp - s_base;
          if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
          else atomic_max_f32(stat_out + px.smp, m);
        }
      }
    }
  };
  using std::integral_constant;
  if (bias == nullptr && has_bn && act == FQ_ACT_RELU)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU>{});
  else if (bias == nullptr && has_bn && act == FQ_ACT_RELU6)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU6>{});
  else if (bias == nullptr && has_bn && act == FQ_ACT_NONE)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_NONE>{});
  else
    run(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{});

  PW_STAMP(5);
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < cols / HW)
      atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
}

// K2j: tile form for the deep layers (K = 256 / 512 / 1024 on 14x14 and 7x7 planes), where there are few pixels
// (784 or 196 tiles of 32) and a big weight matrix.  ONE 32-pixel tile per workgroup; its four wavefronts
//   1. each quantise a QUARTER of the K/32 channel slabs (lane = pixel, as K2h) and publish the int8 fragments to an LDS
//      panel (1 KB per slab, lane order) - the quantise phase is 4x shorter than one wave per tile and nothing is
//      quantised twice;
//   2. each take a quarter of the 32-channel output tiles and multiply: B fragments from registers (K <= 512) or from
//      the LDS panel (K = 1024), A fragments streamed STRAIGHT from L2 out of the fragment-major copy fq_weight_codes
//      leaves behind the row-major codes (one coalesced 16-byte load per lane per MFMA; no weight staging, no per-chunk
//      barrier), two channel tiles at a time on independent accumulators;
//   3. store with lane = pixel (two full lines per store instruction), per-channel constants from LDS.
// Two barriers per tile.  Against K2i: four times the wavefronts, no serial quantise-then-multiply per wave.
struct PwtGeom {
  int Cin, K, Cout, CT, HW;   // K: padded row length of the weight codes; CT = Cout / 32 (Cout % 256 == 0 here)
  int64_t cols, tiles;
  int zoff;
};

template <int KT, bool BLDS>
__global__ __launch_bounds__(kBlock, 3) void pwconv_tile_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwtGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out) {
  constexpr int kSlots = 8;
  constexpr int SLABS = KT / 4;                                         // slabs each wavefront quantises
  extern __shared__ __attribute__((aligned(16))) unsigned char pwt_smem[];
  __shared__ unsigned k_stat[kSlots];
  v4i* panel = reinterpret_cast<v4i*>(pwt_smem);                       // [KT][64] B fragments of the current tile
  const int nch = g.CT * 32;
  float* c_sxw = reinterpret_cast<float*>(pwt_smem + (size_t)KT * 1024);
  float* c_bsc = c_sxw + nch;
  float* c_bsh = c_bsc + nch;
  float* c_bias = c_bsh + nch;
  int* c_zs = reinterpret_cast<int*>(c_bias + nch);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, pl = lane & 31;
  const unsigned HW = (unsigned)g.HW, cols = (unsigned)g.cols;
  const int64_t plane = (int64_t)g.HW;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const int64_t t_begin = g.tiles * blockIdx.x / gridDim.x, t_end = g.tiles * (blockIdx.x + 1) / gridDim.x;
  unsigned s_base;
  {
    const unsigned j0 = (unsigned)t_begin * 32u;
    s_base = (j0 < cols ? j0 : cols - 1) / HW;
  }
  struct Pix { unsigned smp, p; };
  auto pix_of = [&](int64_t t) __attribute__((always_inline)) {
    Pix r;
    unsigned j = (unsigned)(t < g.tiles ? t : g.tiles - 1) * 32u + (unsigned)pl;
    j = j < cols ? j : cols - 1;                                        // lanes past the end copy the last pixel
    r.smp = j / HW;
    r.p = j - r.smp * HW;
    return r;
  };
  auto issue = [&](const Pix& px, int kt, float (&v)[16]) __attribute__((always_inline)) {
    const int cg = kt * 32 + 16 * h;
    const unsigned off = (unsigned)((((int64_t)px.smp * g.Cin + (cg < g.Cin ? 16 * h : 0)) * plane + px.p) * 4);
    const char* ub = reinterpret_cast<const char*>(x) + (int64_t)kt * 32 * plane * 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const float*>(ub + (int64_t)i * plane * 4 + off);
  };

  float bufa[16], bufb[16];
  Pix px = pix_of(t_begin);
  issue(px, wave, bufa);                                                // in flight during the set-up
  FQ_PIN();
  const float max_ = in_stat != nullptr ? batch_mean_dev(in_stat, n) : in_thr[0];
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  if (in_stat != nullptr && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
  const float sx = q.scale;
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;
  for (int i = threadIdx.x; i < nch; i += kBlock) {
    const bool ok = i < g.Cout;
    const int ic = ok ? i : 0;
    c_sxw[i] = sx * wscale[ic];
    c_zs[i] = ok ? g.zoff * wsum[ic] : 0;
    c_bias[i] = bias != nullptr ? bias[ic] : 0.0f;
    c_bsc[i] = has_bn ? bn_scale[ic] : 1.0f;
    c_bsh[i] = has_bn ? bn_shift[ic] : 0.0f;
  }
  auto quant_to_panel = [&](int kt, const float (&v)[16]) __attribute__((always_inline)) {
    const bool gvalid = kt * 32 + 16 * h < g.Cin;
    v4i f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int packed = pack4_codes(fq_code_int(v[4 * d + 0], q), fq_code_int(v[4 * d + 1], q),
                                     fq_code_int(v[4 * d + 2], q), fq_code_int(v[4 * d + 3], q), 128 - g.zoff);
      f[d] = gvalid ? packed : 0;
    }
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));   // pin the arithmetic here (see K2h)
    panel[(kt << 6) + lane] = f;
  };

  auto run = [&](auto bias_c, auto bn_c, auto act_c) __attribute__((always_inline)) {
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
    for (int64_t t = t_begin; t < t_end; ++t) {
      // ---- 1. my quarter of the slabs -> LDS panel (the first slab of this tile is already in bufa) ----------------
#pragma unroll
      for (int j = 0; j < SLABS; ++j) {
        const int kt = wave + 4 * j;
        float (&mine)[16] = (j & 1) ? bufb : bufa;
        float (&other)[16] = (j & 1) ? bufa : bufb;
        if (j + 1 < SLABS) issue(px, kt + 4, other);
        FQ_PIN();
        quant_to_panel(kt, mine);
        FQ_PIN();
      }
      const Pix cur = px;
      px = pix_of(t + 1);
      static_assert(SLABS % 2 == 0, "the last slab of a tile must leave bufa free");
      issue(px, wave, bufa);                                            // next tile's first slab: in flight during the GEMM
      FQ_PIN();
      __syncthreads();                                                  // panel complete
      v4i bfrag[BLDS ? 1 : KT];
      if (!BLDS) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) bfrag[kt] = panel[(kt << 6) + lane];
      }
      // ---- 2./3. my quarter of the channel tiles, two at a time ------------------------------------------------------
      const unsigned yoff = (unsigned)((((int64_t)cur.smp * g.Cout + 4 * h) * plane + cur.p) * 4);
      float m = 0.0f;
      auto load_zs = [&](int ct, v16i& acc) __attribute__((always_inline)) {
        const int cb = ct * 32 + 4 * h;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const v4i z = *reinterpret_cast<const v4i*>(c_zs + cb + 8 * gq);
          acc[4 * gq + 0] = z.x; acc[4 * gq + 1] = z.y; acc[4 * gq + 2] = z.z; acc[4 * gq + 3] = z.w;
        }
      };
      auto finish = [&](int ct, const v16i& acc) __attribute__((always_inline)) {
        const int cb = ct * 32 + 4 * h;
        char* ybase = reinterpret_cast<char*>(y) + (int64_t)(ct * 32) * plane * 4;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int c0 = cb + 8 * gq;
          const f4 sxw = *reinterpret_cast<const f4*>(c_sxw + c0);
          const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
          const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
          f4 bch = (f4){0.f, 0.f, 0.f, 0.f};
          if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) bch = *reinterpret_cast<const f4*>(c_bias + c0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = (float)acc[4 * gq + r] * sxw[r];
            if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) v = v + bch[r];
            if (BN_M == 1 || (BN_M < 0 && has_bn)) {
              v = v * bsc[r];
              v = v + bsh[r];
            }
            v = ACT_M < 0 ? act_rt(v, act) : act_rt(v, ACT_M);
            *reinterpret_cast<float*>(ybase + (int64_t)(8 * gq + r) * plane * 4 + yoff) = v;
            m = fmaxf(m, fabsf(v));
          }
        }
      };
      const v4i* wf = reinterpret_cast<const v4i*>(wfrag) + lane;
#pragma unroll 1
      for (int ct0 = wave; ct0 < g.CT; ct0 += 8) {
        const int ct1 = ct0 + 4;
        v16i acc0, acc1;
        load_zs(ct0, acc0);
        load_zs(ct1, acc1);
        const v4i* A0 = wf + ((int64_t)(ct0 * KT) << 6);
        const v4i* A1 = wf + ((int64_t)(ct1 * KT) << 6);
        v4i fa[3], fb[3];
        fa[0] = A0[0]; fb[0] = A1[0];
        fa[1] = A0[64]; fb[1] = A1[64];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int kn = kt + 2 < KT ? kt + 2 : KT - 1;
          fa[(kt + 2) % 3] = A0[kn << 6];
          fb[(kt + 2) % 3] = A1[kn << 6];
          const v4i b = BLDS ? panel[(kt << 6) + lane] : bfrag[BLDS ? 0 : kt];
          acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[kt % 3], b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[kt % 3], b, acc1, 0, 0, 0);
          FQ_PIN();                              // only the two-ahead prefetch in flight (else all 2*KT loads are hoisted)
        }
        finish(ct0, acc0);
        finish(ct1, acc1);
      }
      if (has_stat) {
        const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)cur.smp);
        if (__all(cur.smp == s0)) {
          const float wm = wave_max(m);
          if (lane == 0) {
            const unsigned slot = s0 - s_base;
            if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
            else atomic_max_f32(stat_out + s0, wm);
          }
        } else {
          const unsigned slot = cur.smp - s_base;
          if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
          else atomic_max_f32(stat_out + cur.smp, m);
        }
      }
      __syncthreads();                                                  // panel free for the next tile
    }
  };
  __syncthreads();                                                      // constants and the statistic table are staged
  using std::integral_constant;
  if (bias == nullptr && has_bn && act == FQ_ACT_RELU)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU>{});
  else if (bias == nullptr && has_bn && act == FQ_ACT_RELU6)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU6>{});
  else if (bias == nullptr && has_bn && act == FQ_ACT_NONE)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_NONE>{});
  else
    run(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{});
  if (has_stat) {
    if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < cols / HW)
      atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
}

// weight codes: one workgroup per (padded) row: code = roundf(w / (s + eps)), zero padding, row sums
__global__ __launch_bounds__(kBlock) void weight_codes_kernel(const float* __restrict__ w, int rows, int row_len,
                                                              int rows_per_scale, float levels, int row_pad,
                                                              const float* __restrict__ gmax,
                                                              int8_t* __restrict__ codes, float* __restrict__ scales,
                                                              int* __restrict__ rowsum, int8_t* __restrict__ frag) {
  // `frag` (second half of the codes buffer): the same codes in MFMA-fragment order for v_mfma_i32_32x32x32_i8 with the
  // weights as the A operand: fragment (ct = row / 32, kt = k / 32) is 1 KB = 64 lanes x 16 bytes, lane = row % 32 +
  // 32 * ((k % 32) / 16), byte = k % 16 - so a wavefront fetches one fragment with ONE fully coalesced 16-byte load
  __shared__ int red[4];
  const int r = blockIdx.x;
  int8_t* dst = codes + (int64_t)r * row_pad;
  const int kts = row_pad >> 5;
  auto frag_at = [&](int i) -> int8_t* {
    const int kt = i >> 5, hs = (i >> 4) & 1, b = i & 15;
    return frag + ((((int64_t)(r >> 5) * kts + kt) << 6) + (r & 31) + 32 * hs) * 16 + b;
  };
  if (r >= rows) {                                                     // padded row
    for (int i = threadIdx.x; i < row_pad; i += kBlock) {
      dst[i] = 0;
      *frag_at(i) = 0;
    }
    return;
  }
  const float s = gmax[r / rows_per_scale] / levels;
  const float d = s + kEps;
  int acc = 0;
  for (int i = threadIdx.x; i < row_pad; i += kBlock) {
    int c = 0;
    if (i < row_len) c = (int)roundf(w[(int64_t)r * row_len + i] / d);
    dst[i] = (int8_t)c;
    *frag_at(i) = (int8_t)c;
    acc += c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    rowsum[r] = red[0] + red[1] + red[2] + red[3];
    scales[r] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K3: weights, (rows, row_len).  Small rows: a workgroup stages several whole rows in LDS (one HBM read), reduces
// each row with a wavefront, then applies from LDS.  Long rows (layer mode): K1 per row + K3b apply.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kWTile = 8192;   // floats of LDS staging per workgroup (32 KiB)

__global__ __launch_bounds__(kBlock) void weight_rows_lds_kernel(const float* __restrict__ w,
                                                                 float* __restrict__ wq, int64_t rows, int row_len,
                                                                 int rows_per_block, float levels,
                                                                 float* __restrict__ scales_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                       // rows_per_block * row_len
  float* sc = smem + kWTile;                // rows_per_block scales
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int nrows = (int)((rows - r0) < rows_per_block ? (rows - r0) : rows_per_block);
  const int cnt = nrows * row_len;
  const float* src = w + r0 * row_len;
  float* dst = wq + r0 * row_len;
  const bool vec = ((row_len & 3) == 0) && ((((uintptr_t)src) & 15u) == 0) && ((((uintptr_t)dst) & 15u) == 0);
  if (vec) {
    const float4* p = reinterpret_cast<const float4*>(src);
    float4* t4 = reinterpret_cast<float4*>(tile);
    for (int i = threadIdx.x; i < cnt / 4; i += kBlock) t4[i] = p[i];
  } else {
    for (int i = threadIdx.x; i < cnt; i += kBlock) tile[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < nrows; r += kBlock / 64) {
    const float* row = tile + r * row_len;
    float m = 0.0f;
    for (int i = lane; i < row_len; i += 64) m = fmaxf(m, fabsf(row[i]));
    m = wave_max(m);
    if (lane == 0) {
      const float s = m / levels;
      sc[r] = s;
      if (scales_out != nullptr) scales_out[r0 + r] = s;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cnt; i += kBlock) {
    const int r = i / row_len;
    const float s = sc[r];
    dst[i] = roundf(tile[i] / (s + kEps)) * s;
  }
}

// K3b: apply with per-row scale = rowmax[r] / levels (rows long enough that a chunk never spans two rows).
template <bool VEC>
__global__ __launch_bounds__(kBlock) void weight_apply_kernel(const float* __restrict__ w, float* __restrict__ wq,
                                                              int64_t row_len, int chunks_per_row,
                                                              int64_t total_chunks, const float* __restrict__ rowmax,
                                                              float levels, float* __restrict__ scales_out) {
  for (int64_t c = blockIdx.x; c < total_chunks; c += gridDim.x) {
    const int64_t r = c / chunks_per_row;
    const int64_t off0 = (c - r * chunks_per_row) * (int64_t)kChunk;
    const int64_t gbase = r * row_len + off0;
    const int64_t rem = row_len - off0;
    const float s = rowmax[r] / levels;
    const float d = s + kEps;
    if (scales_out != nullptr && off0 == 0 && threadIdx.x == 0) scales_out[r] = s;
    const int cnt = (int)(rem < kChunk ? rem : kChunk);
    if (VEC) {
      const float4* p = reinterpret_cast<const float4*>(w + gbase);
      float4* o = reinterpret_cast<float4*>(wq + gbase);
      for (int i = threadIdx.x; i < cnt / 4; i += kBlock) {
        float4 v = p[i];
        o[i] = make_float4(roundf(v.x / d) * s, roundf(v.y / d) * s, roundf(v.z / d) * s, roundf(v.w / d) * s);
      }
    } else {
      for (int i = threadIdx.x; i < cnt; i += kBlock) wq[gbase + i] = roundf(w[gbase + i] / d) * s;
    }
  }
}

// K3c: generic STE (per-row device scale, optional clip)
__global__ __launch_bounds__(kBlock) void ste_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     int64_t numel, int64_t row_len,
                                                     const float* __restrict__ scales, int has_clip, float lo,
                                                     float hi, float eps) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) {
    const float s = scales[i / row_len];
    float v = x[i];
    if (has_clip) v = fminf(fmaxf(v, lo), hi);
    y[i] = roundf(v / (s + eps)) * s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K4: Winograd-domain per-out-channel weight fake-quant.  One workgroup per output channel; each thread owns
// (ci) filters: U = G g G^T in registers (k-sequential, multiply and add separately rounded — oracle order).
// ---------------------------------------------------------------------------------------------------------------
struct WinoMats {
  float G[8 * 3];     // t x 3
  float GI[3 * 8];    // 3 x t
  float GTI[8 * 3];   // t x 3
};

template <int T>
__device__ __forceinline__ void wino_forward(const float* __restrict__ g9, const WinoMats& M, float (&U)[T][T]) {
  float t1[T][3];
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float acc = M.G[a * 3 + 0] * g9[0 * 3 + j];
      acc = acc + M.G[a * 3 + 1] * g9[1 * 3 + j];
      acc = acc + M.G[a * 3 + 2] * g9[2 * 3 + j];
      t1[a][j] = acc;
    }
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b) {
      float acc = t1[a][0] * M.G[b * 3 + 0];     // G^T[k][b] = G[b][k]
      acc = acc + t1[a][1] * M.G[b * 3 + 1];
      acc = acc + t1[a][2] * M.G[b * 3 + 2];
      U[a][b] = acc;
    }
}

template <int T>
__global__ __launch_bounds__(kBlock) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ wq,
                                                             int cin_g, WinoMats M, float levels,
                                                             float* __restrict__ scales_out) {
  __shared__ float red[4];
  __shared__ float s_scale;
  const int co = blockIdx.x;
  const float* wc = w + (int64_t)co * cin_g * 9;
  float* oc = wq + (int64_t)co * cin_g * 9;
  float m = 0.0f;
  for (int ci = threadIdx.x; ci < cin_g; ci += kBlock) {
    float g9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g9[k] = wc[ci * 9 + k];
    float U[T][T];
    wino_forward<T>(g9, M, U);
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int b = 0; b < T; ++b) m = fmaxf(m, fabsf(U[a][b]));
  }
  m = block_max(m, red);
  if (threadIdx.x == 0) {
    s_scale = m / levels;
    if (scales_out != nullptr) scales_out[co] = s_scale;
  }
  __syncthreads();
  const float s = s_scale;
  const float d = s + kEps;
  for (int ci = threadIdx.x; ci < cin_g; ci += kBlock) {
    float g9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g9[k] = wc[ci * 9 + k];
    float U[T][T];
    wino_forward<T>(g9, M, U);
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int b = 0; b < T; ++b) U[a][b] = roundf(U[a][b] / d) * s;
    // back: t2 = GI (3 x T) . Uq (T x T);  g = t2 (3 x T) . GTI (T x 3)
    float t2[3][T];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int b = 0; b < T; ++b) {
        float acc = M.GI[i * T + 0] * U[0][b];
#pragma unroll
        for (int a = 1; a < T; ++a) acc = acc + M.GI[i * T + a] * U[a][b];
        t2[i][b] = acc;
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float acc = t2[i][0] * M.GTI[0 * 3 + j];
#pragma unroll
        for (int b = 1; b < T; ++b) acc = acc + t2[i][b] * M.GTI[b * 3 + j];
        oc[ci * 9 + i * 3 + j] = acc;
      }
  }
}

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
// K6: global max / min-max (flat)
// ---------------------------------------------------------------------------------------------------------------
template <bool WANT_MIN, bool USE_ABS>
__global__ __launch_bounds__(kBlock) void minmax_kernel(const float* __restrict__ x, int64_t numel, int vec_ok,
                                                        float* __restrict__ out_min, float* __restrict__ out_max) {
  __shared__ float red[4];
  float mx = USE_ABS ? 0.0f : -INFINITY, mn = INFINITY;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  if (vec_ok) {
    const float4* p = reinterpret_cast<const float4*>(x);
    const int64_t nvec = numel / 4;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
      float4 v = p[i];
      mx = fmaxf(mx, fmaxf(fmaxf(stat_of<USE_ABS>(v.x), stat_of<USE_ABS>(v.y)),
                           fmaxf(stat_of<USE_ABS>(v.z), stat_of<USE_ABS>(v.w))));
      if (WANT_MIN) mn = fminf(mn, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
    }
    for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) {
      mx = fmaxf(mx, stat_of<USE_ABS>(x[i]));
      if (WANT_MIN) mn = fminf(mn, x[i]);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) {
      mx = fmaxf(mx, stat_of<USE_ABS>(x[i]));
      if (WANT_MIN) mn = fminf(mn, x[i]);
    }
  }
  mx = block_max(mx, red);
  if (threadIdx.x == 0) atomic_max_f32(out_max, mx);
  if (WANT_MIN) {
    mn = block_min(mn, red);
    if (threadIdx.x == 0) atomic_min_f32(out_min, mn);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K7: 2048-bin histogram, LDS-privatised (one copy per wavefront), zeros skipped, exact uint64 accumulation.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void histogram_kernel(const float* __restrict__ x, int64_t numel, int vec_ok,
                                                           const float* __restrict__ max_dev, int bins,
                                                           unsigned long long* __restrict__ hist,
                                                           unsigned int* __restrict__ neg_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned int lh[];      // 4 * bins
  for (int i = threadIdx.x; i < 4 * bins; i += kBlock) lh[i] = 0u;
  __syncthreads();
  unsigned int* mine = lh + (threadIdx.x >> 6) * bins;
  const float mx = max_dev[0];
  const float scales = (float)bins / (mx + 1e-5f);                       // distribution_calibrate.py:41
  unsigned int neg = 0;
  auto put = [&](float v) {
    neg += (v < 0.0f) ? 1u : 0u;
    const float c = fminf(fmaxf(v, 0.0f), mx);                           // :39
    if (c != 0.0f) {                                                     // :40
      int idx = (int)(c * scales);                                       // :42 (truncation)
      idx = idx < bins ? idx : bins - 1;
      atomicAdd(&mine[idx], 1u);
    }
  };
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  if (vec_ok) {
    const float4* p = reinterpret_cast<const float4*>(x);
    const int64_t nvec = numel / 4;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
      float4 v = p[i];
      put(v.x);
      put(v.y);
      put(v.z);
      put(v.w);
    }
    for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) put(x[i]);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) put(x[i]);
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

// ---------------------------------------------------------------------------------------------------------------
// K9: int-code quantise / dequantise
// ---------------------------------------------------------------------------------------------------------------
__global__ void codes_range_kernel(float* __restrict__ range, int mode, const float* __restrict__ ws) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float mn, mx;
  if (mode == FQ_CODES_INT8) {
    mx = ws[1];
    mn = -mx;
  } else if (mode == FQ_CODES_UINT8) {
    mn = ws[0];
    mx = ws[1];
  } else {
    mn = range[0];
    mx = range[1];
  }
  range[0] = mn;
  range[1] = mx;
  if (mode != FQ_CODES_SCALE) range[2] = (mx == -mn) ? (mx / 127.0f) : ((mx - mn) / 255.0f);
}

__global__ __launch_bounds__(kBlock) void quantize_codes_kernel(const float* __restrict__ x,
                                                                int32_t* __restrict__ codes, int64_t numel,
                                                                const float* __restrict__ range) {
  const float mn = range[0], mx = range[1], sc = range[2];
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) {
    const float c = fminf(fmaxf(x[i], mn), mx);
    codes[i] = (int32_t)roundf(c / sc);
  }
}

__global__ __launch_bounds__(kBlock) void dequantize_kernel(const int32_t* __restrict__ codes, float* __restrict__ y,
                                                            int64_t numel, const float* __restrict__ scale) {
  const float sc = scale[0];
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) y[i] = (float)codes[i] * sc;
}

// ---------------------------------------------------------------------------------------------------------------
// host-side launch helpers
// ---------------------------------------------------------------------------------------------------------------
struct Chunking {
  int chunks_per_sample;
  int64_t total;
};
inline Chunking chunking(int64_t n, int64_t inner, int chunk = kChunk) {
  Chunking c;
  c.chunks_per_sample = (int)((inner + chunk - 1) / chunk);
  c.total = n * c.chunks_per_sample;
  return c;
}
// Small tensors: 32 KiB steps would leave most CUs with < 1 workgroup; use 8 KiB steps (2 accesses in flight per
// lane) once the tensor has fewer 32 KiB chunks than 8 workgroups per CU.
constexpr int kSmallUnroll = 2;
constexpr int kSmallChunk = kBlock * kVec * kSmallUnroll;
inline bool use_small_chunks(int64_t n, int64_t inner) {
  return chunking(n, inner).total < (int64_t)num_cu() * kMaxBlocksPerCU;
}

// Streaming policy, by kernel and tensor size.  Measured on MI355X (profiles/r1_policy_sweep.txt, r1_kbench.txt):
// tensors that (with their output) fit the 256 MiB Infinity Cache want PLAIN loads/stores — the producer just left x
// there and the consumer (the convolution) will find y there; larger tensors want nontemporal loads+stores in the apply
// pass, and the online apply pass walks backwards to start on what the statistic pass read last.
// FQ_POLICY_STAT / FQ_POLICY_ONLINE / FQ_POLICY_OFFLINE (integer OR of the kPol* bits) override for tuning runs;
// FQ_POLICY_BIG_BYTES moves the size threshold.
int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}
int stream_policy(int kernel_id, int64_t numel) {
  static const int ov_stat = env_int("FQ_POLICY_STAT", -1);
  static const int ov_on = env_int("FQ_POLICY_ONLINE", -1);
  static const int ov_off = env_int("FQ_POLICY_OFFLINE", -1);
  static const int64_t big = (int64_t)env_int("FQ_POLICY_BIG_MB", 160) * 1000000;
  const bool is_big = numel * (int64_t)sizeof(float) >= big;
  switch (kernel_id) {
    case FQ_KERNEL_STAT:
      return ov_stat >= 0 ? ov_stat : 0;
    case FQ_KERNEL_APPLY_ONLINE:
      if (ov_on >= 0) return ov_on;
      return is_big ? (kPolReverse | kPolNtLoad | kPolNtStore) : kPolReverse;
    default:
      if (ov_off >= 0) return ov_off;
      return is_big ? (kPolNtLoad | kPolNtStore) : 0;
  }
}

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

inline float act_levels(int width, unsigned flags) {
  return (flags & FQ_ACT_SIGNED) ? (float)((1 << (width - 1)) - 1) : (float)((1 << width) - 1);
}

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

const char* fq_last_error(void) { return g_err; }
int fq_version(void) { return 100; }

int fq_device_info(char* arch, int arch_len, int* compute_units, int* wavefront) {
  int dev = 0;
  FQ_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  FQ_HIP(hipGetDeviceProperties(&prop, dev));
  if (arch != nullptr && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (wavefront) *wavefront = prop.warpSize;
  return FQ_OK;
}

int fq_profile_enable(int on) {
  g_prof_on = on != 0;
  return FQ_OK;
}

int fq_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
  return FQ_OK;
}

int fq_profile_read(int kernel_id, double* total_ms, int64_t* launches, double* total_bytes) {
  FQ_REQUIRE(kernel_id >= 0 && kernel_id < FQ_KERNEL_COUNT, "fq_profile_read: bad kernel id %d", kernel_id);
  FQ_REQUIRE(total_ms && launches && total_bytes, "fq_profile_read: null pointer");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double ms = 0.0, bytes = 0.0;
  int64_t cnt = 0;
  for (auto& r : g_prof) {
    if (r.kid != kernel_id) continue;
    FQ_HIP(hipEventSynchronize(r.b));
    float t = 0.f;
    FQ_HIP(hipEventElapsedTime(&t, r.a, r.b));
    ms += t;
    bytes += r.bytes;
    ++cnt;
  }
  *total_ms = ms;
  *launches = cnt;
  *total_bytes = bytes;
  return FQ_OK;
}

int fq_profile_calibrate(void* scratch, int repeats, double* median_ms, fqStream_t stream) {
  FQ_REQUIRE(scratch && median_ms && repeats > 0 && repeats <= 4096, "fq_profile_calibrate: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  // Enqueue every bracketed launch back to back (a busy queue, like the timed region), synchronise once at the end.
  std::vector<hipEvent_t> ev((size_t)repeats * 2);
  for (auto& e : ev) FQ_HIP(hipEventCreate(&e));
  for (int i = 0; i < repeats; ++i) {
    FQ_HIP(hipEventRecord(ev[2 * i], st));
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, st, (float*)scratch, (int64_t)1, 0.0f);
    FQ_HIP(hipEventRecord(ev[2 * i + 1], st));
  }
  FQ_HIP(hipEventSynchronize(ev.back()));
  std::vector<float> ts;
  for (int i = 0; i < repeats; ++i) {
    float t = 0.f;
    FQ_HIP(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]));
    ts.push_back(t);
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  std::sort(ts.begin(), ts.end());
  *median_ms = ts[ts.size() / 2];
  return FQ_OK;
}

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

int fq_bn_act_stat(const float* x, float* y, int64_t n, int64_t c, int64_t hw, const float* scale,
                   const float* shift, int act, float* stat_out, fqStream_t stream) {
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
  const bool vec = (inner % kVec == 0) && aligned16(x) && aligned16(y);
  const int grid = grid_for(ck.total);
  ProfScope prof(FQ_KERNEL_BN_ACT, 8.0 * (double)n * (double)inner, st);
#define FQ_BN(A, S, V, UU)                                                                                       \
  hipLaunchKernelGGL((bn_act_stat_kernel<A, S, V, UU>), dim3(grid), dim3(kBlock), 0, st, x, y, inner, (int)hw,   \
                     ck.chunks_per_sample, ck.total, scale, shift, stat_out)
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
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_dwconv3x3(const float* x, const float* w, const float* bias, float* y, int64_t n, int64_t c, int64_t h,
                 int64_t wdt, int stride, const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                 float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                 fqStream_t stream) {
  FQ_REQUIRE(x && w && y, "fq_dwconv3x3: null pointer");
  FQ_REQUIRE(n > 0 && c > 0 && h > 0 && wdt > 0 && n * c < (1ll << 31) && h * wdt < (1ll << 28),
             "fq_dwconv3x3: bad shape (n=%lld c=%lld h=%lld w=%lld)", (long long)n, (long long)c, (long long)h,
             (long long)wdt);
  FQ_REQUIRE(stride == 1 || stride == 2, "fq_dwconv3x3: stride must be 1 or 2, got %d", stride);
  FQ_REQUIRE(!(in_stat && in_thr), "fq_dwconv3x3: give in_stat (online) OR in_thr (offline), not both");
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_dwconv3x3: bn_scale and bn_shift go together");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_dwconv3x3: unknown activation %d", act);
  const bool quant = in_stat != nullptr || in_thr != nullptr;
  if (quant) FQ_REQUIRE(in_width >= 2 && in_width <= 16, "fq_dwconv3x3: width %d out of range", in_width);
  hipStream_t st = (hipStream_t)stream;
  static const int form = env_int("FQ_DW_FORM", 0);     // 0 auto, 1 LDS tiles, 2 sliding window 1 col/lane, 3: 4 cols/lane
  const bool can4 = (wdt % 4 == 0) && aligned16(x) && aligned16(y) && ((h * wdt) % 4 == 0) &&
                    (stride == 1 || ((wdt / 2) % 2 == 0));
  if ((form == 3 || form == 0) && can4) {
    DwColGeom cg;
    cg.C = (int)c;
    cg.H = (int)h;
    cg.W = (int)wdt;
    cg.Ho = (int)((h - 1) / stride + 1);
    cg.Wo = (int)((wdt - 1) / stride + 1);
    const int halo = stride == 1 ? 2 : 1;
    const int quads = (int)(wdt / 4);                    // compute lanes per full row
    const int max_sw = 64 - halo;
    cg.nsegx = (quads + max_sw - 1) / max_sw;
    cg.sw = (quads + cg.nsegx - 1) / cg.nsegx;           // compute lanes per segment
    cg.SEG = cg.sw + halo;
    cg.segs = 64 / cg.SEG;
    const int64_t total_segs = n * c * cg.nsegx;
    const int64_t segs_per_block = (int64_t)cg.segs * (kBlock / 64);
    const int64_t nblk = (total_segs + segs_per_block - 1) / segs_per_block;
    FQ_REQUIRE(total_segs < (1ll << 31) - 1024, "fq_dwconv3x3: tensor too large for 32-bit segment indices");
    // every workgroup resident at once (8 per CU), each walking a contiguous range of blocks
    static const int dw_wg_per_cu = env_int("FQ_DW_WG_PER_CU", 8);
    const int grid = (int)(nblk < (int64_t)num_cu() * dw_wg_per_cu ? nblk : (int64_t)num_cu() * dw_wg_per_cu);
    const float levels = act_levels(in_width, in_flags);
    const int lo_neg = (in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
    const float eps = (in_flags & FQ_ACT_NO_EPS) ? 0.0f : kEps;
    if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
    ProfScope prof(FQ_KERNEL_DWCONV, 4.0 * ((double)n * c * h * wdt + (double)n * c * cg.Ho * cg.Wo), st);
#define FQ_DWC4(SS, Q, O)                                                                                         \
  hipLaunchKernelGGL((dwconv3x3_cols4_kernel<SS, Q, O>), dim3(grid), dim3(kBlock), 0, st, x, w, bias, y, cg,       \
                     total_segs, in_stat, (int)n, in_thr, levels, lo_neg, eps, out_current_max, bn_scale,         \
                     bn_shift, act, stat_out)
    if (stride == 1) {
      if (!quant) FQ_DWC4(1, false, false);
      else if (in_stat) FQ_DWC4(1, true, true);
      else FQ_DWC4(1, true, false);
    } else {
      if (!quant) FQ_DWC4(2, false, false);
      else if (in_stat) FQ_DWC4(2, true, true);
      else FQ_DWC4(2, true, false);
    }
#undef FQ_DWC4
    FQ_LAUNCH_CHECK();
    return FQ_OK;
  }
  if (form == 2 || ((form == 0 || form == 3) && wdt >= 14)) {   // narrow planes (7x7): LDS staging coalesces better
    DwColGeom cg;
    cg.C = (int)c;
    cg.H = (int)h;
    cg.W = (int)wdt;
    cg.Ho = (int)((h - 1) / stride + 1);
    cg.Wo = (int)((wdt - 1) / stride + 1);
    const int halo = stride == 1 ? 2 : 1;
    const int max_sw = 64 - halo;
    cg.nsegx = (cg.Wo + max_sw - 1) / max_sw;
    cg.sw = (cg.Wo + cg.nsegx - 1) / cg.nsegx;
    cg.SEG = cg.sw + halo;
    cg.segs = 64 / cg.SEG;
    const int64_t total_segs = n * c * cg.nsegx;
    const int64_t segs_per_block = (int64_t)cg.segs * (kBlock / 64);
    const int64_t nblk = (total_segs + segs_per_block - 1) / segs_per_block;
    FQ_REQUIRE(total_segs < (1ll << 31) - 1024, "fq_dwconv3x3: tensor too large for 32-bit segment indices");
    // every workgroup resident at once (8 per CU), each walking a contiguous range of blocks
    static const int dw_wg_per_cu = env_int("FQ_DW_WG_PER_CU", 8);
    const int grid = (int)(nblk < (int64_t)num_cu() * dw_wg_per_cu ? nblk : (int64_t)num_cu() * dw_wg_per_cu);
    const float levels = act_levels(in_width, in_flags);
    const int lo_neg = (in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
    const float eps = (in_flags & FQ_ACT_NO_EPS) ? 0.0f : kEps;
    if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
    ProfScope prof(FQ_KERNEL_DWCONV, 4.0 * ((double)n * c * h * wdt + (double)n * c * cg.Ho * cg.Wo), st);
#define FQ_DWC(SS, Q, O)                                                                                          \
  hipLaunchKernelGGL((dwconv3x3_cols_kernel<SS, Q, O>), dim3(grid), dim3(kBlock), 0, st, x, w, bias, y, cg,        \
                     total_segs, in_stat, (int)n, in_thr, levels, lo_neg, eps, out_current_max, bn_scale,         \
                     bn_shift, act, stat_out)
    if (stride == 1) {
      if (!quant) FQ_DWC(1, false, false);
      else if (in_stat) FQ_DWC(1, true, true);
      else FQ_DWC(1, true, false);
    } else {
      if (!quant) FQ_DWC(2, false, false);
      else if (in_stat) FQ_DWC(2, true, true);
      else FQ_DWC(2, true, false);
    }
#undef FQ_DWC
    FQ_LAUNCH_CHECK();
    return FQ_OK;
  }
  DwGeom g;
  g.C = (int)c;
  g.H = (int)h;
  g.W = (int)wdt;
  g.Ho = (int)((h - 1) / stride + 1);
  g.Wo = (int)((wdt - 1) / stride + 1);
  g.WS = g.W + 2;
  if ((g.WS & 1) == 0) g.WS += 1;                       // odd dword stride
  const int lds_budget = 8192;                          // floats (32 KiB) -> up to 5 workgroups per CU
  const int plane_in = g.H * g.W;
  if (plane_in <= 4096 && (g.H + 2) * g.WS <= lds_budget) {
    g.TR = g.Ho;
    g.strips = 1;
    g.IR = g.H + 2;
    g.P = 1;
    for (int p = (int)(c < 64 ? c : 64); p >= 1; --p)
      if (c % p == 0 && p * g.IR * g.WS <= lds_budget) {
        g.P = p;
        break;
      }
  } else {
    g.P = 1;
    int tr = (lds_budget / g.WS - 2) / stride;
    FQ_REQUIRE(tr >= 1, "fq_dwconv3x3: rows of %d floats do not fit the LDS tile", g.W);
    g.strips = (g.Ho + tr - 1) / tr;
    g.TR = (g.Ho + g.strips - 1) / g.strips;
    g.IR = g.TR * stride + 2;
  }
  {
    const int per_seg = g.P * g.Wo;
    int nseg = (2 * kBlock + per_seg - 1) / per_seg;    // aim at ~2 work items per lane
    if (nseg < 1) nseg = 1;
    if (nseg > g.TR) nseg = g.TR;
    g.RS = (g.TR + nseg - 1) / nseg;
    g.nseg = (g.TR + g.RS - 1) / g.RS;
  }
  const bool base_ok = aligned16(x);
  if (g.P > 1 || g.strips == 1)
    g.vec_in = base_ok && (((int64_t)g.P * plane_in) % 4 == 0) && (plane_in % 4 == 0 || g.P % 4 == 0);
  else
    g.vec_in = base_ok && (g.W % 4 == 0) && (plane_in % 4 == 0);
  const int64_t tiles = (n * c / g.P) * g.strips;
  size_t lds = (size_t)((g.P * g.IR * g.WS + 3) / 4 * 4 + 16) * sizeof(float);
  const int grid = grid_for(tiles);
  const float levels = act_levels(in_width, in_flags);
  const int lo_neg = (in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
  const float eps = (in_flags & FQ_ACT_NO_EPS) ? 0.0f : kEps;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  ProfScope prof(FQ_KERNEL_DWCONV, 4.0 * ((double)n * c * h * wdt + (double)n * c * g.Ho * g.Wo), st);
#define FQ_DW(SS, Q, O)                                                                                           \
  hipLaunchKernelGGL((dwconv3x3_kernel<SS, Q, O>), dim3(grid), dim3(kBlock), lds, st, x, w, bias, y, g, tiles,    \
                     in_stat, (int)n, in_thr, levels, lo_neg, eps, out_current_max, bn_scale, bn_shift, act,      \
                     stat_out)
  if (stride == 1) {
    if (!quant) FQ_DW(1, false, false);
    else if (in_stat) FQ_DW(1, true, true);
    else FQ_DW(1, true, false);
  } else {
    if (!quant) FQ_DW(2, false, false);
    else if (in_stat) FQ_DW(2, true, true);
    else FQ_DW(2, true, false);
  }
#undef FQ_DW
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_weight_codes(const float* w, int64_t rows, int64_t row_len, int rows_per_scale, int width, int64_t row_pad,
                    int64_t rows_pad, int8_t* codes, float* scales, int32_t* rowsum, void* ws, fqStream_t stream) {
  FQ_REQUIRE(w && codes && scales && rowsum && ws, "fq_weight_codes: null pointer");
  FQ_REQUIRE(rows > 0 && row_len > 0 && rows_per_scale > 0 && rows % rows_per_scale == 0,
             "fq_weight_codes: bad shape (rows=%lld row_len=%lld rows_per_scale=%d)", (long long)rows,
             (long long)row_len, rows_per_scale);
  FQ_REQUIRE(width >= 2 && width <= 8, "fq_weight_codes: width %d does not fit int8 codes", width);
  FQ_REQUIRE(row_pad >= row_len && rows_pad >= rows && rows_pad < (1ll << 31) && row_pad % 32 == 0 && rows_pad % 32 == 0,
             "fq_weight_codes: bad padding (row_pad and rows_pad must be multiples of 32)");
  hipStream_t st = (hipStream_t)stream;
  const int64_t groups = rows / rows_per_scale;
  float* gmax = (float*)ws;
  FQ_HIP(hipMemsetAsync(gmax, 0, groups * sizeof(float), st));
  if (int rc = launch_absmax(w, groups, (int64_t)rows_per_scale * row_len, true, gmax, st)) return rc;
  const float levels = (float)((1 << (width - 1)) - 1);
  hipLaunchKernelGGL(weight_codes_kernel, dim3((unsigned)rows_pad), dim3(kBlock), 0, st, w, (int)rows, (int)row_len,
                     rows_per_scale, levels, (int)row_pad, gmax, codes, scales, (int*)rowsum,
                     codes + rows_pad * row_pad);
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
  hipLaunchKernelGGL(gap_stat_kernel, dim3((unsigned)((planes + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, x, y, planes,
                     (int)c, (int)hw, stat_out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_gemm_i8_codes(const int8_t* xcodes, const int8_t* wcodes, const int32_t* wsum, int32_t* out, int64_t n,
                     int64_t l, int64_t k_pad, int64_t cout, int zoff, fqStream_t stream) {
  FQ_REQUIRE(xcodes && wcodes && wsum && out, "fq_gemm_i8_codes: null pointer");
  FQ_REQUIRE(n > 0 && l > 0 && cout > 0 && k_pad > 0 && k_pad % 32 == 0 && n * l < (1ll << 40) && l < (1ll << 31),
             "fq_gemm_i8_codes: bad shape (k_pad=%lld must be a positive multiple of 32)", (long long)k_pad);
  FQ_REQUIRE(zoff == 0 || zoff == 128, "fq_gemm_i8_codes: zoff must be 0 (signed codes) or 128 (re-centred unsigned)");
  FQ_REQUIRE(aligned16(xcodes) && aligned16(wcodes), "fq_gemm_i8_codes: code buffers must be 16-byte aligned");
  const int64_t tiles = (n * l + 31) / 32;
  int64_t grid = (tiles + (kBlock / 64) - 1) / (kBlock / 64);
  if (grid > (int64_t)num_cu() * 16) grid = (int64_t)num_cu() * 16;
  hipLaunchKernelGGL(gemm_i8_codes_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, xcodes, wcodes,
                     (const int*)wsum, (int*)out, n * l, (int)l, (int)k_pad, (int)cout, zoff);
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

int fq_stem_conv3x3s2(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n, int64_t cin,
                      int64_t cout, int64_t h, int64_t w, const float* bn_scale, const float* bn_shift, int act,
                      float* stat_out, fqStream_t stream) {
  FQ_REQUIRE(x && w_tap_major && y, "fq_stem_conv3x3s2: null pointer");
  FQ_REQUIRE(n > 0 && n < 65536 && h > 0 && w > 0 && h < (1 << 15) && w < (1 << 15), "fq_stem_conv3x3s2: bad shape");
  FQ_REQUIRE(cin == 3 && cout == 32, "fq_stem_conv3x3s2: only 3 -> 32 channels is built (got %lld -> %lld)",
             (long long)cin, (long long)cout);
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_stem_conv3x3s2: bn_scale and bn_shift go together");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_stem_conv3x3s2: unknown activation %d", act);
  hipStream_t st = (hipStream_t)stream;
  const int Ho = (int)((h + 2 - 3) / 2 + 1), Wo = (int)((w + 2 - 3) / 2 + 1);
  const int64_t hwo = (int64_t)Ho * Wo;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  ProfScope prof(FQ_KERNEL_BN_ACT, 4.0 * ((double)n * cin * h * w + (double)n * cout * hwo), st);
  const int tiles = (int)((hwo + kBlock - 1) / kBlock);
  // enough workgroups to fill the chip, as few statistic atomics per sample as that allows
  int tiles_per_wg = 1;
  while (tiles_per_wg < 8 && n * ((tiles + 2 * tiles_per_wg - 1) / (2 * tiles_per_wg)) >= (int64_t)num_cu() * 8) tiles_per_wg *= 2;
  const dim3 grid((unsigned)((tiles + tiles_per_wg - 1) / tiles_per_wg), (unsigned)n);
  hipLaunchKernelGGL((stem_conv3x3s2_kernel<3, 32>), grid, dim3(kBlock), 0, st, x, w_tap_major, bias, y, (int)h, (int)w,
                     Ho, Wo, tiles_per_wg, bn_scale, bn_shift, act, stat_out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

#ifdef FQ_PW_TRACE
int fq_debug_set_pw_trace(unsigned long long* buf) {
  FQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pw_trace), &buf, sizeof(buf)));
  return FQ_OK;
}
int fq_debug_set_pw_dbg(int v) {
  FQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pw_dbg), &v, sizeof(v)));
  return FQ_OK;
}
#endif

size_t fq_pwconv_workspace_bytes(int64_t n, int64_t cin_pad, int64_t hw) {
  const int64_t cols_pad = (n * hw + 255) / 256 * 256;
  return (size_t)cols_pad * (size_t)cin_pad + 64;
}

int fq_pwconv_i8(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                 float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw, const float* in_stat,
                 const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                 const float* bn_scale, const float* bn_shift, int act, float* stat_out, void* ws,
                 fqStream_t stream) {
  FQ_REQUIRE(x && wcodes && wscale && wsum && y && ws, "fq_pwconv_i8: null pointer");
  FQ_REQUIRE(n > 0 && cin > 0 && cout > 0 && hw > 0 && hw < (1ll << 30) && n * hw < (1ll << 31) - 512,
             "fq_pwconv_i8: bad shape");
  FQ_REQUIRE(cin_pad >= cin && cin_pad % 64 == 0 && cin_pad <= 8192, "fq_pwconv_i8: cin_pad=%lld must be a multiple "
             "of 64 covering cin=%lld", (long long)cin_pad, (long long)cin);
  FQ_REQUIRE((in_stat != nullptr) != (in_thr != nullptr), "fq_pwconv_i8: give in_stat (online) OR in_thr (offline): the "
             "integer path needs a quantised input");
  FQ_REQUIRE(in_stat == nullptr || out_current_max != nullptr, "fq_pwconv_i8: online mode needs out_current_max (the "
             "GEMM reads the batch statistic from it)");
  FQ_REQUIRE(in_width >= 2 && in_width <= 8, "fq_pwconv_i8: input width %d does not fit int8 codes", in_width);
  FQ_REQUIRE(!(in_flags & (FQ_ACT_NO_ABS | FQ_ACT_NO_EPS)), "fq_pwconv_i8: unsupported activation flags");
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_pwconv_i8: bn_scale and bn_shift go together");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_pwconv_i8: unknown activation %d", act);
  FQ_REQUIRE(aligned16(wcodes) && aligned16(ws) && aligned16(x), "fq_pwconv_i8: x, wcodes and ws must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const bool signed_codes = (in_flags & FQ_ACT_SIGNED) != 0;
  const int zoff = signed_codes ? 0 : 128;      // unsigned codes are stored re-centred so they fit int8
  const float levels = act_levels(in_width, in_flags);
  const int lo_neg = (in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
  int8_t* codes = (int8_t*)ws;
  ProfScope prof(FQ_KERNEL_PWCONV, 4.0 * ((double)n * cin * hw + (double)n * cout * hw), st);
  static const int pw_form = env_int("FQ_PW_FORM", 0);      // 0 auto, 1 two kernels, 2 panel, 3 stream, 4 chunk, 5 tile
  {
    // streaming form: the whole weight matrix in LDS, activations straight from NCHW into MFMA registers
    const int kt = (int)((cin + 31) / 32);
    const int ct = (int)((cout + 31) / 32);
    const size_t lds = (size_t)ct * kt * 1024 + (size_t)ct * 32 * 5 * sizeof(float);
    const bool kt_ok = kt == 1 || kt == 2 || kt == 3 || kt == 4 || kt == 6 || kt == 8;
    const bool shape_ok = cin % 16 == 0 && cout % 32 == 0 && kt_ok && lds <= 72 * 1024;
    if ((pw_form == 0 || pw_form == 3) && shape_ok) {
      PwsGeom s;
      s.Cin = (int)cin; s.K = (int)cin_pad; s.Cout = (int)cout; s.CT = ct; s.HW = (int)hw;
      s.cols = n * hw; s.tiles = (s.cols + 31) / 32; s.zoff = zoff;
      // persistent workgroups: as many as stay resident (LDS / 2 per SIMD by registers), each wave a contiguous range
      // (measured, tools/pwbench.py: 3 per CU for the 126-VGPR instantiations KT <= 2, 2 above)
      int per_cu = (int)((160 * 1024) / (lds + 1024));
      const int by_regs = kt <= 2 ? 3 : 2;
      per_cu = per_cu > by_regs ? by_regs : per_cu;
      static const int pws_wg = env_int("FQ_PWS_WG_PER_CU", 0);
      if (pws_wg > 0) per_cu = pws_wg;
      int64_t grid = (int64_t)num_cu() * per_cu;
      const int64_t need = (s.tiles + 3) / 4;
      if (grid > need) grid = need;
      if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
#define FQ_PWS_CASE(KT_)                                                                                               \
  case KT_: {                                                                                                          \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_stream_kernel<KT_>),         \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) == hipSuccess; \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the streaming kernel");                   \
    hipLaunchKernelGGL((pwconv_stream_kernel<KT_>), dim3((unsigned)grid), dim3(kBlock), lds, st, x, wcodes, wscale,    \
                       (const int*)wsum, bias, y, s, in_stat, (int)n, in_thr, levels, lo_neg, kEps, out_current_max,   \
                       bn_scale, bn_shift, act, stat_out);                                                             \
  } break;
      switch (kt) {
        FQ_PWS_CASE(1) FQ_PWS_CASE(2) FQ_PWS_CASE(3) FQ_PWS_CASE(4) FQ_PWS_CASE(6) FQ_PWS_CASE(8)
        default: break;
      }
#undef FQ_PWS_CASE
      FQ_LAUNCH_CHECK();
      return FQ_OK;
    }
    FQ_REQUIRE(pw_form != 3, "fq_pwconv_i8: FQ_PW_FORM=3 but the shape does not fit the streaming kernel");
    // tile form (K2j): one 32-pixel tile per workgroup, weights streamed from the fragment-major copy
    // Every workgroup streams the whole weight matrix from L2, so this form only pays when tiles are few (7x7 planes:
    // 196 tiles; measured 41 / 44 us against 55 / 64 us for the chunked / two-kernel forms on 512->1024 and
    // 1024->1024); with 784 tiles (14x14) the 200 MB of weight traffic make it twice as slow as the chunked form.
    const bool few_tiles = (n * hw + 31) / 32 < (int64_t)num_cu() * 2;
    if (((pw_form == 0 && few_tiles) || pw_form == 5) && cin % 16 == 0 && cout % 256 == 0 && cin_pad == cin &&
        (kt == 8 || kt == 16 || kt == 32)) {
      PwtGeom t;
      t.Cin = (int)cin; t.K = (int)cin_pad; t.Cout = (int)cout; t.CT = ct; t.HW = (int)hw;
      t.cols = n * hw; t.tiles = (t.cols + 31) / 32; t.zoff = zoff;
      const size_t ldst = (size_t)kt * 1024 + (size_t)ct * 32 * 5 * sizeof(float);
      const int64_t rows_pad = (cout + 63) / 64 * 64;
      const int8_t* wfrag = wcodes + rows_pad * cin_pad;               // second half of fq_weight_codes' buffer
      static const int pwt_wg = env_int("FQ_PWT_WG_PER_CU", 4);
      int64_t grid = (int64_t)num_cu() * pwt_wg;
      if (grid > t.tiles) grid = t.tiles;
      if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
#define FQ_PWT_CASE(KT_, BL_)                                                                                          \
  if (kt == KT_) {                                                                                                     \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_tile_kernel<KT_, BL_>),      \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess; \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the tile kernel");                        \
    hipLaunchKernelGGL((pwconv_tile_kernel<KT_, BL_>), dim3((unsigned)grid), dim3(kBlock), ldst, st, x, wfrag, wscale, \
                       (const int*)wsum, bias, y, t, in_stat, (int)n, in_thr, levels, lo_neg, kEps, out_current_max,   \
                       bn_scale, bn_shift, act, stat_out);                                                             \
  }
#ifndef FQ_PWT_REGB
#define FQ_PWT_REGB false       // true: B fragments in registers for K <= 512 (spills at the 3-waves-per-SIMD register cap)
#endif
      FQ_PWT_CASE(8, !FQ_PWT_REGB) FQ_PWT_CASE(16, !FQ_PWT_REGB) FQ_PWT_CASE(32, true)
#undef FQ_PWT_CASE
      FQ_LAUNCH_CHECK();
      return FQ_OK;
    }
    FQ_REQUIRE(pw_form != 5, "fq_pwconv_i8: FQ_PW_FORM=5 but the shape does not fit the tile kernel");
    // chunked streaming form: weights through LDS in double-buffered chunks, K = 256 or 512
    static const int pwc_ctc = env_int("FQ_PWC_CTC", 0), pwc_wsplit = env_int("FQ_PWC_WSPLIT", 0);
    if ((pw_form == 0 || pw_form == 4) && cin % 16 == 0 && cout % 32 == 0 && (kt == 8 || kt == 16)) {
      PwcGeom c;
      c.Cin = (int)cin; c.K = (int)cin_pad; c.Cout = (int)cout; c.CT = ct; c.HW = (int)hw;
      c.cols = n * hw; c.tiles = (c.cols + 31) / 32; c.zoff = zoff;
      c.rows = (int)((cout + 63) / 64 * 64);
      // few tiles (7x7 planes): all four wavefronts share one tile and split a 4-tile chunk; otherwise one tile each
      c.wsplit = c.tiles < (int64_t)num_cu() * 2 ? 4 : 1;
      if (pwc_wsplit > 0) c.wsplit = pwc_wsplit;
      c.CTC = c.wsplit == 4 ? 4 : (kt == 16 ? 2 : 4);
      if (pwc_ctc > 0) c.CTC = pwc_ctc;
      c.NC = (ct + c.CTC - 1) / c.CTC;
      const int pieces = c.CTC * kt / 4;
      const bool ok = (c.NC == 1 || c.NC % 2 == 0) && ct % c.CTC == 0 && c.wsplit <= c.CTC && (pieces == 8 || pieces == 16) &&
                      (c.wsplit == 1 || c.wsplit == 2 || c.wsplit == 4);
      if (ok) {
        c.batches = (c.tiles + (4 / c.wsplit) - 1) / (4 / c.wsplit);
        const size_t lds2 = 2 * ((size_t)c.CTC * kt * 1024 + (size_t)c.CTC * 32 * 5 * sizeof(float));
        int per_cu = (int)((160 * 1024) / (lds2 + 1024));
        per_cu = per_cu > 2 ? 2 : (per_cu < 1 ? 1 : per_cu);
        int64_t grid = (int64_t)num_cu() * per_cu;
        if (grid > c.batches) grid = c.batches;
        if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
#define FQ_PWC_CASE(KT_, P_)                                                                                           \
  if (kt == KT_ && pieces == P_) {                                                                                     \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_chunk_kernel<KT_, P_>),      \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess; \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the chunked kernel");                     \
    hipLaunchKernelGGL((pwconv_chunk_kernel<KT_, P_>), dim3((unsigned)grid), dim3(kBlock), lds2, st, x, wcodes, wscale, \
                       (const int*)wsum, bias, y, c, in_stat, (int)n, in_thr, levels, lo_neg, kEps, out_current_max,   \
                       bn_scale, bn_shift, act, stat_out, (float*)ws);                                                 \
  }
        FQ_PWC_CASE(8, 8) FQ_PWC_CASE(8, 16) FQ_PWC_CASE(16, 8) FQ_PWC_CASE(16, 16)
#undef FQ_PWC_CASE
        FQ_LAUNCH_CHECK();
        return FQ_OK;
      }
    }
    FQ_REQUIRE(pw_form != 4, "fq_pwconv_i8: FQ_PW_FORM=4 but the shape does not fit the chunked kernel");
  }
  {
    PwfGeom f;
    f.Cin = (int)cin; f.K = (int)cin_pad; f.Cout = (int)cout; f.HW = (int)hw; f.cols = n * hw;
    if (cout > 128) { f.wm = 4; f.wn = 1; }
    else if (cout > 64) { f.wm = 2; f.wn = 2; }
    else { f.wm = 1; f.wn = 4; }
    f.PX = 64 * f.wn;
    f.cblocks = (int)((cout + 64 * f.wm - 1) / (64 * f.wm));
    f.swz = (cin_pad % 128 == 0) ? 7 : 3;
    f.zoff = zoff;
    const int64_t ctiles = (f.cols + f.PX - 1) / f.PX;
    // split the channel blocks of one column tile over several workgroups while the launch would not fill the chip
    // (the panel is then quantised csplit times from activations that sit in L2 / the Infinity Cache)
    int csplit = 1;
    while (csplit < f.cblocks && ctiles * csplit < (int64_t)num_cu() * 3) csplit *= 2;
    if (csplit > f.cblocks) csplit = f.cblocks;
    f.csplit = csplit;
    f.nblk = (f.cblocks + csplit - 1) / csplit;
    const size_t lds = (size_t)f.PX * f.K + 16 + (size_t)f.nblk * f.wm * 64 * 5 * sizeof(float);
    // csplit >= 4 (few column tiles, e.g. 7x7 planes) re-quantises the panel too often: the two-kernel form with its
    // perfectly parallel quantise+transpose pass is faster there (tools/pwbench.py)
    if ((pw_form == 2 || (pw_form == 0 && csplit <= 2)) && lds <= 80 * 1024 - 256) {   // >= 2 workgroups per CU by LDS
      static const bool attr_ok = [] {
        const int lim = 80 * 1024 - 256;
        const void* fns[4] = {reinterpret_cast<const void*>(&pwconv_fused_kernel<true, 2>),
                              reinterpret_cast<const void*>(&pwconv_fused_kernel<true, 4>),
                              reinterpret_cast<const void*>(&pwconv_fused_kernel<false, 2>),
                              reinterpret_cast<const void*>(&pwconv_fused_kernel<false, 4>)};
        bool ok = true;
        for (const void* fn : fns) ok = ok && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess;
        return ok;
      }();
      FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the fused kernel");
      if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
      const int64_t grid = (ctiles + 7) / 8 * 8 * f.csplit;
      const int units = f.wn * (f.K / 64);
#define FQ_PWF_LAUNCH(ON, GG)                                                                                          \
  hipLaunchKernelGGL((pwconv_fused_kernel<ON, GG>), dim3((unsigned)grid), dim3(kBlock), lds, st, x, wcodes, wscale,    \
                     (const int*)wsum, bias, y, f, in_stat, (int)n, in_thr, levels, lo_neg, kEps, out_current_max,     \
                     bn_scale, bn_shift, act, stat_out)
      if (in_stat) {
        if (units <= 4) FQ_PWF_LAUNCH(true, 2); else FQ_PWF_LAUNCH(true, 4);
      } else {
        if (units <= 4) FQ_PWF_LAUNCH(false, 2); else FQ_PWF_LAUNCH(false, 4);
      }
#undef FQ_PWF_LAUNCH
      FQ_LAUNCH_CHECK();
      return FQ_OK;
    }
    FQ_REQUIRE(pw_form != 2, "fq_pwconv_i8: FQ_PW_FORM=2 but the panel needs %zu bytes of LDS", lds);
  }
  {  // A: quantise + transpose
    const int ptiles = (int)((hw + 63) / 64), ctiles = (int)(cin_pad / 64);
    const int64_t tiles = n * ptiles * ctiles;
    const int grid = grid_for(tiles);
    if (in_stat)
      hipLaunchKernelGGL((quant_transpose_i8_kernel<true>), dim3(grid), dim3(kBlock), 0, st, x, codes, (int)cin,
                         (int)cin_pad, (int)hw, ptiles, ctiles, tiles, in_stat, (int)n, in_thr, levels, lo_neg, kEps,
                         zoff, out_current_max);
    else
      hipLaunchKernelGGL((quant_transpose_i8_kernel<false>), dim3(grid), dim3(kBlock), 0, st, x, codes, (int)cin,
                         (int)cin_pad, (int)hw, ptiles, ctiles, tiles, in_stat, (int)n, in_thr, levels, lo_neg, kEps,
                         zoff, out_current_max);
    FQ_LAUNCH_CHECK();
  }
  PwGeom g;
  g.Cin = (int)cin;
  g.CinPad = (int)cin_pad;
  g.Cout = (int)cout;
  g.HW = (int)hw;
  g.cols = n * hw;
  if (cout > 128) { g.wm = 4; g.wn = 1; }
  else if (cout > 64) { g.wm = 2; g.wn = 2; }
  else { g.wm = 1; g.wn = 4; }
  g.PT_B = 64 * g.wn;
  g.passes = (int)((cout + 64 * g.wm - 1) / (64 * g.wm));
  g.stride = 0;
  g.zoff = zoff;
  const int64_t tiles = ((g.cols + g.PT_B - 1) / g.PT_B) * g.passes;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  int64_t grid64 = tiles < (int64_t)num_cu() * 32 ? tiles : (int64_t)num_cu() * 32;
  grid64 = grid64 / g.passes * g.passes;                  // multiple of the channel blocks (tiles is one already)
  const int grid = (int)(grid64 < g.passes ? g.passes : grid64);
  const float* sx_src = in_stat ? out_current_max : in_thr;
  hipLaunchKernelGGL((pwconv_i8_kernel<0>), dim3(grid), dim3(kBlock), 0, st, (const int8_t*)codes, wcodes, wscale,
                     (const int*)wsum, bias, y, g, tiles, sx_src, levels, bn_scale, bn_shift, act, stat_out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
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

int fq_ste_forward(const float* x, float* y, int64_t rows, int64_t row_len, const float* scales, int has_clip,
                   float clip_lo, float clip_hi, float eps, fqStream_t stream) {
  FQ_REQUIRE(x && y && scales, "fq_ste_forward: null pointer");
  FQ_REQUIRE(rows > 0 && row_len > 0, "fq_ste_forward: empty tensor");
  const int64_t numel = rows * row_len;
  const int grid = grid_for((numel + kBlock * 4 - 1) / (kBlock * 4));
  hipLaunchKernelGGL(ste_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, x, y, numel, row_len, scales,
                     has_clip, clip_lo, clip_hi, eps);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

size_t fq_weight_workspace_bytes(int64_t rows) { return (size_t)(rows < 1 ? 1 : rows) * sizeof(float) + 64; }

int fq_weight_fake_quant(const float* w, float* w_q, int64_t rows, int64_t row_len, int width, float* scales_out,
                         void* ws, fqStream_t stream) {
  FQ_REQUIRE(w && w_q, "fq_weight_fake_quant: null pointer");
  FQ_REQUIRE(rows > 0 && row_len > 0, "fq_weight_fake_quant: empty tensor (rows=%lld row_len=%lld)",
             (long long)rows, (long long)row_len);
  FQ_REQUIRE(width >= 2 && width <= 16, "fq_weight_fake_quant: width %d out of range", width);
  hipStream_t st = (hipStream_t)stream;
  const float levels = (float)((1 << (width - 1)) - 1);
  if (row_len <= kWTile) {
    int rpb = (int)(kWTile / row_len);
    if (rpb > 1024) rpb = 1024;
    // keep >= ~2 workgroups per CU busy when there are many short rows
    const int64_t want_blocks = (int64_t)num_cu() * 2;
    int64_t balanced = (rows + want_blocks - 1) / want_blocks;
    if (balanced < 1) balanced = 1;
    if (rpb > balanced) rpb = (int)balanced;
    const int64_t blocks = (rows + rpb - 1) / rpb;
    const size_t lds = (size_t)(kWTile + 1024) * sizeof(float);
    ProfScope prof(FQ_KERNEL_WEIGHT, 8.0 * (double)rows * (double)row_len, st);
    hipLaunchKernelGGL(weight_rows_lds_kernel, dim3((unsigned)blocks), dim3(kBlock), lds, st, w, w_q, rows,
                       (int)row_len, rpb, levels, scales_out);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
  }
  FQ_REQUIRE(ws, "fq_weight_fake_quant: workspace required for rows longer than %d", kWTile);
  float* rowmax = (float*)ws;
  FQ_HIP(hipMemsetAsync(rowmax, 0, rows * sizeof(float), st));
  if (int rc = launch_absmax(w, rows, row_len, true, rowmax, st)) return rc;
  const Chunking ck = chunking(rows, row_len);
  const bool vec = (row_len % kVec == 0) && aligned16(w) && aligned16(w_q);
  const int grid = grid_for(ck.total);
  if (vec)
    hipLaunchKernelGGL((weight_apply_kernel<true>), dim3(grid), dim3(kBlock), 0, st, w, w_q, row_len,
                       ck.chunks_per_sample, ck.total, rowmax, levels, scales_out);
  else
    hipLaunchKernelGGL((weight_apply_kernel<false>), dim3(grid), dim3(kBlock), 0, st, w, w_q, row_len,
                       ck.chunks_per_sample, ck.total, rowmax, levels, scales_out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_wino_weight_fake_quant(const float* w, float* w_q, int64_t cout, int64_t cin_g, int t, const float* G,
                              const float* GI, const float* GTI, int width, float* scales_out, void* ws,
                              fqStream_t stream) {
  (void)ws;
  FQ_REQUIRE(w && w_q && G && GI && GTI, "fq_wino_weight_fake_quant: null pointer");
  FQ_REQUIRE(t == 4 || t == 6 || t == 8, "fq_wino_weight_fake_quant: t must be 4 (F23), 6 (F43) or 8 (F63), got %d", t);
  FQ_REQUIRE(cout > 0 && cin_g > 0 && cin_g < (1ll << 28), "fq_wino_weight_fake_quant: bad shape");
  FQ_REQUIRE(width >= 2 && width <= 16, "fq_wino_weight_fake_quant: width %d out of range", width);
  WinoMats M;
  memset(&M, 0, sizeof(M));
  memcpy(M.G, G, sizeof(float) * t * 3);
  memcpy(M.GI, GI, sizeof(float) * 3 * t);
  memcpy(M.GTI, GTI, sizeof(float) * t * 3);
  const float levels = (float)((1 << (width - 1)) - 1);
  hipStream_t st = (hipStream_t)stream;
  if (t == 4)
    hipLaunchKernelGGL((wino_weight_kernel<4>), dim3((unsigned)cout), dim3(kBlock), 0, st, w, w_q, (int)cin_g, M,
                       levels, scales_out);
  else if (t == 6)
    hipLaunchKernelGGL((wino_weight_kernel<6>), dim3((unsigned)cout), dim3(kBlock), 0, st, w, w_q, (int)cin_g, M,
                       levels, scales_out);
  else
    hipLaunchKernelGGL((wino_weight_kernel<8>), dim3((unsigned)cout), dim3(kBlock), 0, st, w, w_q, (int)cin_g, M,
                       levels, scales_out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

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
  // every workgroup ends with up to `bins` global 64-bit atomics: keep the grid at ~2 workgroups per CU (enough loads in
  // flight for a read-only stream) so that the flush stays a small fraction of the work
  int64_t hg = (numel + kChunk - 1) / kChunk;
  if (hg > (int64_t)num_cu() * 2) hg = (int64_t)num_cu() * 2;
  const int grid = (int)(hg < 1 ? 1 : hg);
  ProfScope prof(FQ_KERNEL_HISTOGRAM, 4.0 * (double)numel, (hipStream_t)stream);
  hipLaunchKernelGGL(histogram_kernel, dim3(grid), dim3(kBlock), (size_t)4 * bins * sizeof(unsigned int),
                     (hipStream_t)stream, x, numel, aligned16(x) ? 1 : 0, max_dev, bins,
                     (unsigned long long*)hist, (unsigned int*)neg_count);
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

int fq_quantize_codes(const float* x, int32_t* codes, int64_t numel, int mode, float* range_dev, void* ws,
                      fqStream_t stream) {
  FQ_REQUIRE(x && codes && range_dev, "fq_quantize_codes: null pointer");
  FQ_REQUIRE(numel > 0, "fq_quantize_codes: empty tensor");
  FQ_REQUIRE(mode >= FQ_CODES_INT8 && mode <= FQ_CODES_SCALE, "unknown out type: %d", mode);
  hipStream_t st = (hipStream_t)stream;
  float* mm = (float*)ws;   // {min, max}
  const int grid = grid_for((numel + kChunk - 1) / kChunk);
  if (mode == FQ_CODES_INT8 || mode == FQ_CODES_UINT8) {
    FQ_REQUIRE(ws, "fq_quantize_codes: workspace required");
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, st, mm, (int64_t)1, INFINITY);
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, st, mm + 1, (int64_t)1,
                       mode == FQ_CODES_INT8 ? 0.0f : -INFINITY);
    if (mode == FQ_CODES_INT8)
      hipLaunchKernelGGL((minmax_kernel<false, true>), dim3(grid), dim3(kBlock), 0, st, x, numel,
                         aligned16(x) ? 1 : 0, mm, mm + 1);
    else
      hipLaunchKernelGGL((minmax_kernel<true, false>), dim3(grid), dim3(kBlock), 0, st, x, numel,
                         aligned16(x) ? 1 : 0, mm, mm + 1);
    FQ_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(codes_range_kernel, dim3(1), dim3(64), 0, st, range_dev, mode, mm);
  hipLaunchKernelGGL(quantize_codes_kernel, dim3(grid), dim3(kBlock), 0, st, x, codes, numel, range_dev);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_dequantize(const int32_t* codes, float* y, int64_t numel, const float* scale_dev, fqStream_t stream) {
  FQ_REQUIRE(codes && y && scale_dev, "fq_dequantize: null pointer");
  FQ_REQUIRE(numel > 0, "fq_dequantize: empty tensor");
  const int grid = grid_for((numel + kChunk - 1) / kChunk);
  hipLaunchKernelGGL(dequantize_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, codes, y, numel,
                     scale_dev);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // extern "C"
