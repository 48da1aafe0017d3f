// libfakequant — internal header shared by every translation unit of the library (not installed; the public boundary
// is include/fakequant.h).  Host-side state that must exist once (error text, event records, per-device CU counts,
// streaming policy) lives in namespace fqi and is defined in fq_core.hip; everything device-side is header-only and sits
// in an anonymous namespace, so each translation unit carries its own copy and no relocatable device code is needed.
//
// Translation units (csrc/build.py compiles them in parallel and links libfakequant.so):
//   fq_core.hip        errors, event timing (fq_profile_*), device info, streaming policy
//   fq_stream.hip      K1 per-sample statistic, K1b batch means, K2 fake-quant apply, K2b BatchNorm+activation+statistic,
//                      K11 global average pool + statistic, K12 evaluation counters
//   fq_dwconv.hip      K2c/K2d/K2e depthwise 3x3 with quantise-on-load (LDS tiles / 1 column per lane / 4 columns per lane),
//                      K2o whole small planes in registers, K2p 14x14 / 7x7 planes as flat 16-byte ranges through an LDS transpose
//   fq_stem.hip        K2s first convolution 3x3 s2 (3 -> 32) on the vector ALU, K2q 3x3 / 7x7 on the fp32 matrix cores
//   fq_conv3x3.hip     K2n dense 3x3 on int8 codes (implicit GEMM over tap and channel)
//   fq_pw_stream.hip   K2h pointwise on int8 codes, weights resident in LDS (largest planes)
//   fq_pw_sample.hip   K2r pointwise, one block of 96..128 pixels x 256 / 512 channels per workgroup, output-stationary
//   fq_pw_split.hip    K2m pointwise, one (pixel tile, channel group) per workgroup: every other layer from 28x28 planes down
//   fq_pw_generic.hip  K2f pointwise for every other shape (quantise + transpose, then an integer GEMM)
//   fq_pwconv.hip      fq_pwconv_i8: shape-based choice between the pointwise forms; the weight-code kernel (fq_weight_codes)
//   fq_weights.hip     K3 weight fake-quant (layer / group / channel), generic STE, K4 Winograd-domain weights
//   fq_calib.hip       K5 EMA, K6 global max, K7 histogram, K8 KL threshold search
//   fq_codes.hip       K9 int-code quantise / dequantise, K10 exact int8 GEMM (nn.Conv2D(quantized=True))
//
// Design rules applied throughout:
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
//     elementwise / reduction / small-stencil work bounded by HBM or by instruction issue;
//   * hipcc's scheduler is kept honest in the hand-pipelined loops with FQ_PIN (asm memory clobber + sched_barrier) and
//     empty "+v" asm pins (it otherwise sinks arithmetic below prefetches or hoists every load of an unrolled loop).
#ifndef FQ_COMMON_H_
#define FQ_COMMON_H_

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

// ---------------------------------------------------------------------------------------------------------------
// host-side state shared by all translation units (defined in fq_core.hip)
// ---------------------------------------------------------------------------------------------------------------
namespace fqi {

int fail(int code, const char* fmt, ...);           // records the thread-local error text, returns `code`
int num_cu();                                       // compute units of the CURRENT device (cached per device)
int max_lds_bytes();                                // LDS a workgroup may ask for on the CURRENT device (cached per device)
int env_int(const char* name, int dflt);
int stream_policy(int kernel_id, int64_t numel);    // kPol* bits by kernel and tensor size

struct ProfRec {
  int kid;
  double bytes;        // algorithmic bytes (4 B per input and per output element, also for code tensors)
  double moved;        // bytes the launch really moves: 1 B per element on a side that is a C16 code tensor
  hipEvent_t a, b;
};
extern bool g_prof_on;
void prof_push(const ProfRec& r);

// Brackets the launches of one entry point with two events on the launch stream while fq_profile_enable(1) is set.
struct ProfScope {
  bool on;
  ProfRec r;
  hipStream_t st;
  ProfScope(int kid, double bytes, hipStream_t s, double moved = -1.0) : on(g_prof_on), st(s) {
    if (!on) return;
    r.kid = kid;
    r.bytes = bytes;
    r.moved = moved < 0.0 ? bytes : moved;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
      on = false;
      return;
    }
    (void)hipEventRecord(r.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, st);
    prof_push(r);
  }
  void cancel() {                    // nothing was launched after all: no record
    if (!on) return;
    on = false;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
};

// defined in fq_stream.hip; used by the weight paths too
int launch_absmax(const float* x, int64_t n, int64_t inner, bool use_abs, float* out, hipStream_t st);
int init_stat(float* p, int64_t n, bool use_abs, hipStream_t st);

}  // namespace fqi

#define FQ_HIP(expr)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) return fqi::fail(FQ_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));       \
  } while (0)
#define FQ_REQUIRE(cond, ...)                                      \
  do {                                                             \
    if (!(cond)) return fqi::fail(FQ_ERR_INVALID, __VA_ARGS__);    \
  } while (0)
#define FQ_LAUNCH_CHECK() FQ_HIP(hipGetLastError())

// Ordering fence for hand-pipelined loops: the asm memory clobber stops IR-level motion of loads, the sched_barrier the
// machine scheduler's (it sinks prefetches next to their use, or hoists every load of an unrolled loop to the top).
#define FQ_PIN()                         \
  do {                                   \
    asm volatile("" ::: "memory");        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

namespace {

using namespace fqi;

constexpr int kBlock = 256;                       // 4 wavefronts
constexpr int kVec = 4;                           // floats per lane per access (16 B)
constexpr int kUnroll = 8;                        // independent 16 B accesses in flight per lane
constexpr int kChunk = kBlock * kVec * kUnroll;   // 8192 floats = 32 KiB per workgroup step
#ifndef FQ_MAX_WG_PER_CU
#define FQ_MAX_WG_PER_CU 8
#endif
constexpr int kMaxBlocksPerCU = FQ_MAX_WG_PER_CU;   // grid cap of the streaming kernels (tuning: -DFQ_MAX_WG_PER_CU=6)
constexpr float kEps = 1e-10f;                    // ste_func.py:39,41

inline int grid_for(int64_t work_items) {
  int64_t cap = (int64_t)num_cu() * kMaxBlocksPerCU;
  int64_t g = work_items < cap ? work_items : cap;
  return (int)(g < 1 ? 1 : g);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline float act_levels(int width, unsigned flags) {
  return (flags & FQ_ACT_SIGNED) ? (float)((1 << (width - 1)) - 1) : (float)((1 << width) - 1);
}

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

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

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
// (-DFQ_DBG_NOSTAT: an ablation build that drops every statistic atomic.  Its statistics are all zero, hence every threshold,
// hence every activation: kernels then run 2-6 us faster for reasons that have nothing to do with atomics - see
// profiles/r3_atomic_probe.txt before reading anything into its timings.)
#ifdef FQ_DBG_NOSTAT
#define FQ_STAT_FLUSH_MAX(p, v) ((void)(p), (void)(v))
#else
#define FQ_STAT_FLUSH_MAX(p, v) atomicMax((p), (v))
#endif
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
#ifdef FQ_DBG_NOSTAT
  return;
#endif
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

// Buffer addressing: address = (scalar 48-bit base in the resource) + (ONE 32-bit lane offset) + (scalar offset).  The
// fused producers read 16-64 values per lane that differ only by a wave-uniform stride (channel planes): with flat
// addressing hipcc carries a 64-bit VGPR pair and two VALU adds per access, with a buffer resource the stride lives in an
// SGPR and all accesses of a lane share one offset register.  `bytes` (<= 4 GiB - 1) bounds the range: loads past it
// return 0 and stores past it are dropped, so the resource is re-based per workgroup for tensors beyond 4 GiB.
typedef __amdgpu_buffer_rsrc_t fq_rsrc;
__device__ __forceinline__ fq_rsrc make_rsrc(const void* base, int64_t bytes) {
  const unsigned nb = bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)(bytes < 0 ? 0 : bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)nb, 0x00020000);
}
__device__ __forceinline__ float buf_ld_f32(fq_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float buf_ld_f32_nt(fq_rsrc r, unsigned voff, unsigned soff) {             // nontemporal hint
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 2));
}
__device__ __forceinline__ v4i buf_ld_v4i(fq_rsrc r, unsigned voff, unsigned soff) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const v4u t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return __builtin_bit_cast(v4i, t);
}
__device__ __forceinline__ void buf_st_f32(fq_rsrc r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_st_f32_nt(fq_rsrc r, unsigned voff, unsigned soff, float v) {      // nontemporal hint
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 2);
}

// Wavefront-wide reductions of the batch-mean prologue without the LDS pipeline: four DPP steps inside the rows of 16 lanes
// (quad swaps, half-row and row mirrors: after each step every lane of the group holds the group's result), then the four row
// results through scalar registers.  The six __shfl_xor rounds they replace are six ds_bpermute round trips plus lane
// arithmetic in the dependency chain of EVERY consumer kernel's first code.  The sum is only used where every order of the
// additions is exact (batch_mean_dev), so the changed order changes nothing.
#define FQ_DPP_I32(v, ctrl) __builtin_amdgcn_update_dpp((v), (v), (ctrl), 0xF, 0xF, false)
__device__ __forceinline__ double wave_sum_f64(double v) {
#define FQ_WSUM_STEP(ctrl)                                                                   \
  do {                                                                                       \
    const long long b = __builtin_bit_cast(long long, v);                                    \
    const int lo = FQ_DPP_I32((int)(b & 0xFFFFFFFFll), ctrl), hi = FQ_DPP_I32((int)(b >> 32), ctrl); \
    v += __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);                   \
  } while (0)
  FQ_WSUM_STEP(0xB1);       // quad_perm [1,0,3,2]
  FQ_WSUM_STEP(0x4E);       // quad_perm [2,3,0,1]
  FQ_WSUM_STEP(0x141);      // row_half_mirror
  FQ_WSUM_STEP(0x140);      // row_mirror
#undef FQ_WSUM_STEP
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = (int)(b & 0xFFFFFFFFll), hi = (int)(b >> 32);
  double r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    r[k] = __builtin_bit_cast(double, ((long long)__builtin_amdgcn_readlane(hi, 16 * k) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane(lo, 16 * k));
  return (r[0] + r[1]) + (r[2] + r[3]);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  int b = (int)v;                                         // (values are exponent fields: small and non-negative)
#define FQ_WMIN_STEP(ctrl) do { const int o = FQ_DPP_I32(b, ctrl); b = o < b ? o : b; } while (0)
  FQ_WMIN_STEP(0xB1); FQ_WMIN_STEP(0x4E); FQ_WMIN_STEP(0x141); FQ_WMIN_STEP(0x140);
#undef FQ_WMIN_STEP
  const int r0 = __builtin_amdgcn_readlane(b, 0), r1 = __builtin_amdgcn_readlane(b, 16), r2 = __builtin_amdgcn_readlane(b, 32),
            r3 = __builtin_amdgcn_readlane(b, 48);
  const int m01 = r0 < r1 ? r0 : r1, m23 = r2 < r3 ? r2 : r3;
  return (unsigned)(m01 < m23 ? m01 : m23);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  int b = (int)v;
#define FQ_WMAX_STEP(ctrl) do { const int o = FQ_DPP_I32(b, ctrl); b = o > b ? o : b; } while (0)
  FQ_WMAX_STEP(0xB1); FQ_WMAX_STEP(0x4E); FQ_WMAX_STEP(0x141); FQ_WMAX_STEP(0x140);
#undef FQ_WMAX_STEP
  const int r0 = __builtin_amdgcn_readlane(b, 0), r1 = __builtin_amdgcn_readlane(b, 16), r2 = __builtin_amdgcn_readlane(b, 32),
            r3 = __builtin_amdgcn_readlane(b, 48);
  const int m01 = r0 > r1 ? r0 : r1, m23 = r2 > r3 ? r2 : r3;
  return (unsigned)(m01 > m23 ? m01 : m23);
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
  acc = wave_sum_f64(acc);
  emin = wave_min_u32(emin);
  emax = wave_max_u32(emax);
  const int logn = 32 - __clz(n > 1 ? n - 1 : 1);
  const bool exact = emax == 0u || (emax < 255u && (int)(emax - emin) + 24 + logn <= 53);
  if (!exact) return batch_mean_seq(v, n);
  return (float)acc / (float)n;
}

// The threshold a fused consumer quantises with, and the reference's `current_input_max` side output (convert_conv2d.py:56
// computes the batch statistic in EVERY mode).  Online (no in_thr): the mean of the per-sample maxima is the threshold and
// workgroup 0 writes it out.  Offline: the stored threshold quantises; when the per-sample maxima are given as well, ONE
// wavefront of workgroup 0 derives their mean for `cur_max_out` - instead of a one-workgroup kernel launch per layer in
// front of every consumer (53 launches of ~4 us per ResNet-50 / MobileNetV2 forward).  Call with the whole workgroup.
__device__ __forceinline__ float input_threshold(const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr,
                                                 float* __restrict__ cur_max_out, bool first_wg) {
  if (in_thr != nullptr) {
    if (in_stat != nullptr && cur_max_out != nullptr && first_wg && threadIdx.x < 64) {
      const float cm = batch_mean_dev(in_stat, n);
      if (threadIdx.x == 0) cur_max_out[0] = cm;
    }
    return in_thr[0];
  }
  const float m = batch_mean_dev(in_stat, n);
  if (cur_max_out != nullptr && first_wg && threadIdx.x == 0) cur_max_out[0] = m;
  return m;
}

// The same in two steps.  The threshold is the head of every consumer's dependency chain (statistic -> mean -> quantiser
// parameters -> first code): its loads are requested FIRST, ahead of the activation loads of the workgroup's first tile, and
// turned into the threshold while those are in flight.  Queued behind them (64 loads per lane in the pointwise form) the
// statistic arrived 5-6 us into a 13 us workgroup (tools/pw_trace.py).  Buffer loads: lanes past n read 0 (adding 0 is
// exact and zeros do not take part in the exponent range), a null pointer is an empty buffer.
struct ThresholdReq {
  float s[4];        // in_stat[lane + 64 k]
  float thr;         // in_thr[0]
};
__device__ __forceinline__ ThresholdReq threshold_request(const float* __restrict__ in_stat, int n,
                                                          const float* __restrict__ in_thr, bool first_wg) {
  ThresholdReq r;
  const bool want_stat = in_stat != nullptr && (in_thr == nullptr || first_wg);
  const fq_rsrc rs = make_rsrc(in_stat, want_stat ? (int64_t)n * 4 : 0);
  const fq_rsrc rt = make_rsrc(in_thr, in_thr != nullptr ? 4 : 0);
  const unsigned lane = threadIdx.x & 63u;
#pragma unroll
  for (int k = 0; k < 4; ++k) r.s[k] = buf_ld_f32(rs, (lane + 64u * k) * 4u, 0);
  r.thr = buf_ld_f32(rt, 0, 0);
  return r;
}
__device__ __forceinline__ float batch_mean_from(const ThresholdReq& r, const float* __restrict__ v, int n) {
  if (n > 256) return batch_mean_dev(v, n);
  double acc = 0.0;
  unsigned emin = 255u, emax = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float f = r.s[k];
    acc += (double)f;
    unsigned e = (__float_as_uint(f) >> 23) & 0xFFu;
    if ((__float_as_uint(f) & 0x7FFFFFFFu) != 0u) {
      e = e < 1u ? 1u : e;
      emin = e < emin ? e : emin;
      emax = e > emax ? e : emax;
    }
  }
  acc = wave_sum_f64(acc);
  emin = wave_min_u32(emin);
  emax = wave_max_u32(emax);
  const int logn = 32 - __clz(n > 1 ? n - 1 : 1);
  const bool exact = emax == 0u || (emax < 255u && (int)(emax - emin) + 24 + logn <= 53);
  if (!exact) return batch_mean_seq(v, n);
  return (float)acc / (float)n;
}
__device__ __forceinline__ float threshold_finish(const ThresholdReq& r, const float* __restrict__ in_stat, int n,
                                                  const float* __restrict__ in_thr, float* __restrict__ cur_max_out,
                                                  bool first_wg) {
  if (in_thr != nullptr) {
    if (in_stat != nullptr && cur_max_out != nullptr && first_wg && threadIdx.x < 64) {
      const float cm = batch_mean_from(r, in_stat, n);
      if (threadIdx.x == 0) cur_max_out[0] = cm;
    }
    return r.thr;
  }
  const float m = batch_mean_from(r, in_stat, n);
  if (cur_max_out != nullptr && first_wg && threadIdx.x == 0) cur_max_out[0] = m;
  return m;
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

// ---- range mode: the per-tensor quantiser of the reference's stand-alone block, nn.Conv2D(quantized=True) ----------------
// nn/quantized_conv.py:54-72: ONE global range for the whole tensor - int8: [-max|x|, max|x|], scale = max / 127; uint8:
// [min x, max x], scale = (max - min) / 255 and NO zero point, so the codes round(clip(x) / scale) lie in [L, H] with
// L = round(min / scale), wherever that is - no epsilon.  The fused convolution kernels take it as a third value of their
// `lo_neg_max` argument: kRangeMode says that `in_thr` points to a RANGE RECORD (fq_qconv_range writes it) instead of a
// threshold, and - in the kernels on the matrix cores - that `bias` holds the reference's int32 bias codes (:122-127),
// added to the integer sum.  Codes are stored as (code + ubias) ^ 0x80 per byte with ubias = -L (0 <= code - L <= 255),
// i.e. re-centred by zoff = 128 - ubias; the epilogue's zoff * rowsum term puts the integer sum right, in wrapping int32
// arithmetic like the reference's cast.  The two launch-time cases are the record's special cases (unsigned from 0:
// ubias 0; symmetric: ubias 128).
constexpr int kRangeMode = 2;
constexpr int kRecHi = 0, kRecLo = 1, kRecDenom = 2, kRecMul = 3, kRecUbias = 4, kRecFlags = 5, kRecScale = 6, kRecLcode = 7;
constexpr int kRecFloats = 8;        // hi, lo, divisor, multiply-back scale (1 for kernels that keep CODES in fp32), ubias
                                     // (int), flags (int: 1 = codes do not fit a byte / fp32-exact sums not guaranteed: the
                                     // exact direct kernel recomputes), scale, L (int)
// QParams + zoff of a launch: the threshold form (make_qparams) or, in range mode, the record's
__device__ __forceinline__ QParams make_qparams_rt(float max_, float levels, int lo_neg_max, float eps,
                                                   const float* __restrict__ in_thr, int& zoff) {
  if (lo_neg_max == kRangeMode) {
    QParams q;
    q.hi = max_;                                  // == in_thr[kRecHi]
    q.lo = in_thr[kRecLo];
    q.denom = in_thr[kRecDenom];
    q.scale = in_thr[kRecMul];
    q.rden = 1.0 / (double)q.denom;
    zoff = 128 - __float_as_int(in_thr[kRecUbias]);
    return q;
  }
  return make_qparams(max_, levels, lo_neg_max != 0, eps);
}
__device__ __forceinline__ QParams make_qparams_rt(float max_, float levels, int lo_neg_max, float eps,
                                                   const float* __restrict__ in_thr) {
  int unused = 0;
  return make_qparams_rt(max_, levels, lo_neg_max, eps, in_thr, unused);
}
// four codes "0" in the stored representation (zero padding of a 3x3 convolution: clip range always contains 0)
__device__ __forceinline__ int stored_zero4(int ubias) { return (int)(((unsigned)(ubias & 255) * 0x01010101u) ^ 0x80808080u); }

// Correctly rounded fp32 quotient c / d for a divisor that is the same for the whole kernel, in 3 instructions instead
// of the ~11 of the IEEE division expansion:  (float)((double)c * RN_f64(1/d))  ==  RN_f32(c / d)  for ALL fp32 c, d.
// Proof sketch: the double product carries a relative error <= 2^-52, while the exact quotient of two fp32 numbers is
// either an fp32 number or at least 2^-49 (relative) away from every fp32 rounding midpoint (c - M*d is a non-zero
// integer multiple of 2^(g+b) for a 25-bit midpoint M = N*2^g and d = D*2^b), so the product falls on the same side of
// every midpoint as the exact quotient and the final conversion rounds it to the same fp32 number.  0/0 and x/0 behave
// as in IEEE (rden = inf).  Checked exhaustively around every .5 tie in tests/test_gpu_parity.py.
__device__ __forceinline__ float ieee_div_by(float c, double rden) { return (float)((double)c * rden); }

// The same correctly rounded quotient WITHOUT fp64 instructions (round 6; the recompute kernels of fq_pwdw.hip are bound by
// vector-instruction CYCLES, and v_cvt_f64_f32 / v_mul_f64 / v_cvt_f32_f64 cost about twice an fp32 instruction each:
// profiles/r6_pwdw_pmc.txt).  Markstein's correction step: with y = RN_f32(1/d),
//   q = RN(c * y);  r = fma(-q, d, c)  (exact: the residual of a quotient that is within an ulp);  q' = fma(r, y, q)
// is RN_f32(c / d) whenever nothing under- or overflows and the significand of d is not all ones (P. Markstein, "Computation of
// elementary functions on the IBM RISC System/6000", IBM J. Res. Dev. 1990; the FMA-based division chapter of Muller et al.,
// Handbook of Floating-Point Arithmetic).  make_fast_quot() decides once per launch - a wave-uniform branch - whether the divisor
// qualifies (normal, significand not all ones, reciprocal normal) and otherwise the caller stays with ieee_div_by().
// What is NEEDED is less than the theorem gives: only the integer the quotient rounds to, for quotients in [0, levels]; a
// quotient below 0.25 rounds to 0 whatever its last bits (that covers dividends so small that q or r would be denormal).
// tests/test_gpu_pwdw.py drives fq_debug_fast_quotient over every k + 0.5 tie +- 4 ulp for thousands of divisors.
struct FastQuot {
  float d, y;        // divisor, RN_f32(1 / d)
  bool ok;
};
__device__ __forceinline__ FastQuot make_fast_quot(float d) {
  FastQuot f;
  f.d = d;
  // RN_f32(1/d) without double rounding: among the fp32 neighbours of (float)(1.0 / d) take the one whose product with d is
  // closest to 1 (the products of two fp32 numbers are exact in fp64, and so is their distance from 1)
  const double dd = (double)d;
  float y = (float)(1.0 / dd);
  const float ylo = __uint_as_float(__float_as_uint(y) - 1u), yhi = __uint_as_float(__float_as_uint(y) + 1u);
  const double e = fabs(1.0 - (double)y * dd), elo = fabs(1.0 - (double)ylo * dd), ehi = fabs(1.0 - (double)yhi * dd);
  if (elo < e && elo <= ehi) y = ylo;
  else if (ehi < e) y = yhi;
  f.y = y;
  const unsigned db = __float_as_uint(d), yb = __float_as_uint(y);
  const unsigned de = (db >> 23) & 0xFFu, ye = (yb >> 23) & 0xFFu;
  f.ok = (db >> 31) == 0u && de >= 1u && de <= 253u && ye >= 1u && ye <= 253u && (db & 0x7FFFFFu) != 0x7FFFFFu;
  return f;
}
__device__ __forceinline__ float fast_quot(float c, const FastQuot& f) {
  const float q = c * f.y;
  const float r = fmaf(-q, f.d, c);
  return fmaf(r, f.y, q);
}

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

// ---- the integer code in 5 instructions when the quotient cannot be negative (round 3) ------------------------------------
// v_cvt_rpi_i32_f32 is floor(x + 0.5) evaluated EXACTLY (no fp32 rounding of the sum): on gfx950 it equals (int)roundf(x)
// for every one of the 1 200 142 337 non-negative fp32 values up to 70000, pred(0.5) included (tools/cvt_rpi_probe.hip,
// profiles/r3_cvt_rpi_probe.txt).  For x < 0 it rounds halves UP (roundf: away from zero) and NaN gives INT_MAX ((int) of a
// NaN: 0), so it only stands in for `fq_code_int` when the clip range starts at 0 and the divisor is positive - unsigned
// activations (every BASELINE configuration) and the Dense quirk's [0, max] - which is a property of the launch, decided
// once per kernel (`fq_nonneg`).  clip + divide (3) + round-and-convert = 5 instructions instead of 7, and with non-negative
// codes the four bytes of a dword merge with three shift-ors and ONE xor (mask 0x80808080 for unsigned codes stored
// re-centred, 0 for signed ones) instead of four additions, three shift-ors and an xor: 6 instead of 9 per value.
#ifdef FQ_NO_NONNEG                                       // A/B builds: always the general quantiser
__device__ __forceinline__ bool fq_nonneg(const QParams&) { return false; }
#else
__device__ __forceinline__ bool fq_nonneg(const QParams& q) { return q.lo == 0.0f && q.denom > 0.0f; }
#endif
__device__ __forceinline__ unsigned fq_nonneg_xor(int ubias) { return ubias == 0 ? 0x80808080u : 0u; }
__device__ __forceinline__ int fq_code_nonneg(float x, const QParams& q) {
  const float Q = ieee_div_by(fq_clip(x, q), q.rden);
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(Q));
  return r;
}
template <bool NONNEG>
__device__ __forceinline__ int fq_pack4(float a, float b, float c, float d, const QParams& q, int ubias, unsigned nn_xor) {
  if (NONNEG) {
    unsigned u = (unsigned)fq_code_nonneg(a, q);
    u |= (unsigned)fq_code_nonneg(b, q) << 8;
    u |= (unsigned)fq_code_nonneg(c, q) << 16;
    u |= (unsigned)fq_code_nonneg(d, q) << 24;
    return (int)(u ^ nn_xor);
  }
  return pack4_codes(fq_code_int(a, q), fq_code_int(b, q), fq_code_int(c, q), fq_code_int(d, q), ubias);
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

__device__ __forceinline__ f4 buf_ld_v4f(fq_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// the same with the nontemporal hint (cache-policy bit 1 = `nt` on gfx940+): a tensor that is read once
__device__ __forceinline__ f4 buf_ld_v4f_nt(fq_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 2));
}
// 16-byte buffer store.  HAZARD (found in round 3, profiles/r3_dw_flat_race.txt): `buffer_store_dwordx4 v[a:a+3], voff,
// rsrc, sN offen` reads its data registers for several cycles after it issues, and a VALU instruction that follows it
// directly and writes v[a] can overtake that read - lanes 12-15 of every 16 then store the NEW value of v[a] (seen: the next
// store's address offset) in the first dword.  LLVM's hazard recogniser inserts the wait state for stores wider than 8
// bytes only when soffset is NOT a register (GCNHazardRecognizer::createsVALUHazard), so with a scalar offset nothing
// protects the data registers; whether a VALU write lands there is up to the register allocator, and whether it overtakes
// the read is up to how busy the CU's memory pipeline is (never with 4 wavefronts per CU, 0.05 % of the outputs with 8-12).
// Two ways to stay clear of it: `buf_st_v4f` (one store: the inline asm USES the data registers, so they stay allocated
// across it, and holds two wait states - the number LLVM applies on gfx940+ where it does see the hazard), or a burst of
// `buf_st_v4f_unguarded` followed by ONE `hold_store_data(all the data)` (the asm statements of a guarded store are ordered
// against each other, which would serialise the LDS reads feeding a burst).  tools/isa_lint.py checks the built library.
#ifndef FQ_BUFST_NOPS
#define FQ_BUFST_NOPS 1
#endif
__device__ __forceinline__ void buf_st_v4f_unguarded(fq_rsrc r, unsigned voff, unsigned soff, f4 v) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, (int)voff, (int)soff, 0);
}
template <int N>
__device__ __forceinline__ void hold_store_data(const f4 (&v)[N]) {
#if FQ_BUFST_NOPS >= 0
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : : "v"(v[i]));
  asm volatile("s_nop %0" : : "n"(FQ_BUFST_NOPS));
#endif
}
__device__ __forceinline__ void buf_st_v4f(fq_rsrc r, unsigned voff, unsigned soff, f4 v) {
  buf_st_v4f_unguarded(r, voff, soff, v);
#if FQ_BUFST_NOPS >= 0
  asm volatile("s_nop %1" : : "v"(v), "n"(FQ_BUFST_NOPS));
#endif
}

// max over the wavefront of NON-NEGATIVE floats (|x| statistics), the same in every lane: four DPP steps inside the rows
// of 16 lanes, then the four row results through scalar registers (non-negative floats order like their bit patterns).
// wave_max() above goes through six ds_bpermute round trips with lane arithmetic (~1.2 us per call in the depthwise
// kernels, tools/dw_trace.py); this is ~12 instructions without the LDS pipeline.
__device__ __forceinline__ float wave_max_nonneg(float v) {
  int b = __builtin_bit_cast(int, v);
#define FQ_WMAX_STEP(ctrl)                                                      \
  do {                                                                          \
    const int o = __builtin_amdgcn_update_dpp(b, b, ctrl, 0xF, 0xF, false);     \
    b = o > b ? o : b;                                                          \
  } while (0)
  FQ_WMAX_STEP(0xB1);       // quad_perm [1,0,3,2]
  FQ_WMAX_STEP(0x4E);       // quad_perm [2,3,0,1]
  FQ_WMAX_STEP(0x141);      // row_half_mirror
  FQ_WMAX_STEP(0x140);      // row_mirror
#undef FQ_WMAX_STEP
  const int r0 = __builtin_amdgcn_readlane(b, 0), r1 = __builtin_amdgcn_readlane(b, 16),
            r2 = __builtin_amdgcn_readlane(b, 32), r3 = __builtin_amdgcn_readlane(b, 48);
  const int m01 = r0 > r1 ? r0 : r1, m23 = r2 > r3 ? r2 : r3;
  return __builtin_bit_cast(float, m01 > m23 ? m01 : m23);
}

// n / d and n % d for n < 2^31 by one multiplication (d fixed per launch; host: fast_div_for).  k = 31 + ceil(log2 d),
// M = ceil(2^k / d) < 2^32: M * d - 2^k < d and n < 2^31 make floor(n * M / 2^k) == floor(n / d).
struct FastDiv {
  unsigned d, M, sh;     // sh = k - 32
};
__device__ __forceinline__ unsigned fast_div(unsigned n, const FastDiv& f) { return f.d == 1u ? n : __umulhi(n, f.M) >> f.sh; }
__device__ __forceinline__ unsigned fast_mod(unsigned n, const FastDiv& f) { return n - fast_div(n, f) * f.d; }
inline FastDiv fast_div_for(unsigned d) {
  FastDiv f;
  f.d = d;
  f.M = 0;
  f.sh = 0;
  if (d > 1) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    const unsigned k = 31 + l;
    f.M = (unsigned)(((1ull << k) + d - 1) / d);
    f.sh = k - 32;
  }
  return f;
}

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
__device__ __forceinline__ f4 fq_code4(f4 v, const QParams& q) {
  f4 k;
  k.x = fq_code(v.x, q);
  k.y = fq_code(v.y, q);
  k.z = fq_code(v.z, q);
  k.w = fq_code(v.w, q);
  return k;
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

__device__ __forceinline__ float act_rt(float v, int act) {
  if (act == FQ_ACT_RELU) v = fmaxf(v, 0.0f);
  if (act == FQ_ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
  return v;
}

// Lane i <- lane i-1 / lane i+1 of the wavefront in ONE VALU instruction (DPP wave shift, usually folded into the consuming
// FMA's operand); lane 0 / lane 63 receive 0.  __shfl_up / __shfl_down by 1 go through ds_bpermute: lane-id arithmetic,
// a select, and an LDS-pipeline round trip in the middle of the dependency chain (tools/dpp_probe.hip: same data movement).
__device__ __forceinline__ float lane_prev(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_next(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}

// Epilogue of the depthwise kernels.  EPI 0: bias / BN / activation decided at run time (hipcc turns the wave-uniform
// branches into four selects per output); EPI 1 / 2: the fused-inference case - BN, no bias, ReLU / ReLU6 - in 3 / 4
// instructions.  Same operations in the same order either way.
constexpr int kEpiRuntime = 0, kEpiBnRelu = 1, kEpiBnRelu6 = 2;
// bit of the kernels' `act` argument (set by the library's own callers, run-time epilogue only): the per-channel `bias` vector
// MULTIPLIES the sum instead of being added - the dequantisation factor in_scale * w_scale of nn.Conv2D(quantized=True)'s
// depthwise layer, applied to the exact integer sum before a folded BatchNorm (its own multiply and add stay separate)
constexpr int kActBiasMul = 0x10;
// (ACT = false: a compile-time epilogue WITHOUT its activation - for a code output, where ReLU / ReLU6 and the consumer's clip
// are one median: clip(relu6(v), lo <= 0, hi) == med3(v, 0, min(6, hi)))
template <int EPI, bool ACT = true>
__device__ __forceinline__ float dw_finish(float acc, bool has_bias, float bch, bool has_bn, float bsc, float bsh, int act) {
  if (EPI == kEpiRuntime) {
    if (has_bias) acc = (act & kActBiasMul) ? acc * bch : acc + bch;
    if (has_bn) {
      acc = acc * bsc;
      acc = acc + bsh;
    }
    return act_rt(acc, act & 15);
  }
  acc = acc * bsc;
  acc = acc + bsh;
  if (!ACT) return acc;
  acc = fmaxf(acc, 0.0f);
  return EPI == kEpiBnRelu6 ? fminf(acc, 6.0f) : acc;
}

#ifdef FQ_PW_TRACE
// debug build only (tools/pw_trace.py; built as ONE translation unit, csrc/build.py --amalgamate -DFQ_PW_TRACE, so that
// these symbols exist once): per-workgroup wall-clock stamps of the fused pointwise kernel's phases
__device__ unsigned long long* g_pw_trace = nullptr;
__device__ int g_pw_dbg = 0;          // experiments: 1 = skip the output stores, 2 = skip the activation loads
#define PW_STAMP(i)                                                                       \
  do {                                                                                    \
    if (threadIdx.x == 0 && g_pw_trace != nullptr) g_pw_trace[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define PW_STAMP(i) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// K6: global max / min-max (flat)
// ---------------------------------------------------------------------------------------------------------------
template <bool WANT_MIN, bool USE_ABS>
__global__ __launch_bounds__(kBlock) void minmax_kernel(const float* __restrict__ x, int64_t numel, int vec_ok,
                                                        float* __restrict__ out_min, float* __restrict__ out_max) {
  __shared__ float red[4];
  float mx = USE_ABS ? 0.0f : -INFINITY, mn = INFINITY;
  auto take = [&](float v) {
    mx = fmaxf(mx, stat_of<USE_ABS>(v));
    if (WANT_MIN) mn = fminf(mn, v);
  };
  // contiguous ranges of 32 KiB chunks, all 8 loads of a chunk in flight per lane (as K1)
  const int64_t chunks = (numel + kChunk - 1) / kChunk;
  const ChunkRange rg = block_range(chunks);
  for (int64_t c = rg.begin; c < rg.end; ++c) {
    const int64_t base = c * (int64_t)kChunk;
    const int64_t rem = numel - base;
    if (vec_ok && rem >= kChunk) {
      const f4* p = reinterpret_cast<const f4*>(x + base);
      f4 v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) v[u] = p[threadIdx.x + u * kBlock];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        take(v[u].x);
        take(v[u].y);
        take(v[u].z);
        take(v[u].w);
      }
    } else {
      const int cnt = (int)(rem < kChunk ? rem : kChunk);
      for (int i = threadIdx.x; i < cnt; i += kBlock) take(x[base + i]);
    }
  }
  mx = block_max(mx, red);
  if (threadIdx.x == 0) atomic_max_f32(out_max, mx);
  if (WANT_MIN) {
    mn = block_min(mn, red);
    if (threadIdx.x == 0) atomic_min_f32(out_min, mn);
  }
}

}  // namespace

#endif  // FQ_COMMON_H_
