// libfakequant — K2t: the pointwise convolution on 1x1 planes, i.e. the quantised classifier (gluon Dense behind global
// pooling: convert_dense.py:37-70) on the integer codes, optionally with the evaluation counters of its logits
// (simulate_quantization.py:122-148) in the same launch.  (See fq_common.h for the list of translation units.)
//
// The step's tail used to be four launches of 4-12 us each that move less than 2 MB together (apply pass, library GEMM,
// counters, their fills): all of it is latency.  Here a workgroup owns 32 samples x 32 units; its eight wavefronts split K:
// each loads its slabs of the fp32 rows (64 contiguous bytes per lane and slab), quantises them straight into the A operand
// of v_mfma_i32_32x32x32_i8 (activations as rows, so that a lane's accumulators are 32 consecutive UNITS of one sample and
// the stores are contiguous), multiplies with the fragment-major weight codes (one 16-byte load per slab) and the eight
// partial sums meet in LDS.  Same arithmetic as every other form of fq_pwconv_i8: exact int32 sums, then
// fp32(sum + zoff * rowsum) * (sx * sw) [+ bias] [BatchNorm] [activation].
//
// Counters: each workgroup leaves (value, unit) of its best unit per sample as ONE ordered 64-bit key; the LAST workgroup of a
// sample tile (a counter in the caller's zeroed workspace tells which) takes the maximum of the keys and adds the counts.
// argmax with MXNet's rule (first index on ties; NaN first) is a maximum of keys (ordered value, ~index), hence independent
// of the order in which workgroups finish.
//
// The hand-shake of that "last workgroup" pattern, and what it must not be on this chip: the textbook form - store the key,
// __threadfence(), bump the counter; the last one fences again and reads - compiles to `buffer_wbl2 sc1` + `buffer_inv sc1` per
// workgroup, i.e. a write-back and an invalidation of the XCD's whole L2 (eight XCDs, one L2 each: agent scope is beyond an
// L2), and 128 workgroups doing that took this 12 us kernel to 28.5 us.  Instead every access that crosses workgroups is an
// ATOMIC at agent scope - the keys are written with relaxed atomic stores (`global_store ... sc1`: written through to the
// memory side) and read with relaxed atomic loads (`global_load ... sc1`), the counter is an atomic add - and the ordering a
// fence would give is obtained from the in-order memory pipeline of a wavefront: the workgroup-scope release fence is an
// `s_waitcnt vmcnt(0)`, i.e. the wavefront's key stores have been acknowledged by the memory side before the barrier that
// precedes the counter's increment.  (FQ_ROWS_FENCE=0 selects the textbook form for comparison; both pass the same tests.)
#include <type_traits>

#include "fq_pw.h"

namespace {

constexpr int kRowsNW = 8;        // wavefronts per workgroup = ways K is split
constexpr int kRowsRB = 4;        // slabs of 32 k in flight per wavefront

struct PwRowsGeom {
  int n, cin, kts, cout, ut, zoff;
};

__device__ __forceinline__ unsigned long long rows_key(float v, unsigned unit) {
  unsigned ord;
  if (v != v) {
    ord = 0xFFFFFFFFu;
  } else {
    if (v == 0.0f) v = 0.0f;                                   // -0 and +0 tie
    const unsigned b = __float_as_uint(v);
    ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  }
  return ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - unit);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, m, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), m, 64);
  return ((unsigned long long)hi << 32) | lo;
}

template <bool COUNT>
__global__ __launch_bounds__(kRowsNW * 64) void pwconv_rows_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwRowsGeom g,
    const float* __restrict__ in_stat, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out, const long long* __restrict__ labels, float* __restrict__ counters,
    unsigned long long* __restrict__ partial, unsigned* __restrict__ sync, int fence_mode) {
  __shared__ int red[kRowsNW][16][64];
  __shared__ unsigned wg_last, wg_total, wg_correct;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ut = blockIdx.x % g.ut, st = blockIdx.x / g.ut;
  const int s0 = st * 32, u0 = ut * 32;
  const int i = lane & 31, h = lane >> 5;

  const ThresholdReq treq = threshold_request(in_stat, g.n, in_thr, blockIdx.x == 0);   // first in the memory queue
  const fq_rsrc xr = make_rsrc(x, (int64_t)g.n * g.cin * 4);             // rows past n: zeros
  const unsigned xo = ((unsigned)(s0 + i) * (unsigned)g.cin + 16u * h) * 4u;
  const v4i* wf = reinterpret_cast<const v4i*>(wfrag) + (((int64_t)ut * g.kts) << 6) + lane;

  // the epilogue's per-unit constants and the labels are requested now: behind the barrier each would be one more trip to memory
  const int unit = u0 + i;
  const bool uok = unit < g.cout;
  const int uc = uok ? unit : 0;
  const float wsc = wscale[uc];
  const int wsm = wsum[uc];
  const float bch = bias != nullptr ? bias[uc] : 0.0f;
  const bool has_bn = bn_scale != nullptr;
  const float bsc = has_bn ? bn_scale[uc] : 1.0f, bsh = has_bn ? bn_shift[uc] : 0.0f;
  constexpr int TPS = kRowsNW * 64 / 32;                                 // threads per sample in the counting step: 16
  long long my_label = -1;
  if (COUNT && threadIdx.x % TPS == 0 && s0 + (int)threadIdx.x / TPS < g.n) my_label = labels[s0 + threadIdx.x / TPS];

  f4 xv[kRowsRB][4];
  v4i wv[kRowsRB];
  auto issue = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < kRowsRB; ++r) {
      const int kt = kb + kRowsNW * r;
      if (kt < g.kts) {
        // (k past cin inside a row reads the next row - or zeros behind the last one: finite codes that meet zero weight codes)
#pragma unroll
        for (int d = 0; d < 4; ++d) xv[r][d] = buf_ld_v4f(xr, xo + (unsigned)kt * 128u + 16u * d, 0);
        wv[r] = wf[(int64_t)kt << 6];
      }
    }
  };
  int kb = wave;
  issue(kb);
  FQ_PIN();
  const float max_ = threshold_finish(treq, in_stat, g.n, in_thr, cur_max_out, blockIdx.x == 0);
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  const float sx = q.scale;
  const int ubias = 128 - g.zoff;
  const unsigned nn_xor = fq_nonneg_xor(ubias);

  v16i acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0;
  auto multiply = [&](auto nn_c) __attribute__((always_inline)) {
    for (;;) {
#pragma unroll
      for (int r = 0; r < kRowsRB; ++r) {
        if (kb + kRowsNW * r < g.kts) {
          v4i a;
#pragma unroll
          for (int d = 0; d < 4; ++d)
            a[d] = fq_pack4<decltype(nn_c)::value>(xv[r][d][0], xv[r][d][1], xv[r][d][2], xv[r][d][3], q, ubias, nn_xor);
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, wv[r], acc, 0, 0, 0);
        }
      }
      kb += kRowsNW * kRowsRB;
      if (kb >= g.kts) break;
      issue(kb);
    }
  };
  if (fq_nonneg(q))
    multiply(std::true_type{});
  else
    multiply(std::false_type{});

#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  if (COUNT && threadIdx.x == 0) {
    wg_total = 0u;
    wg_correct = 0u;
  }
  __syncthreads();

  // wavefront `wave` finishes accumulators 2 wave, 2 wave + 1: rows (samples) 8 (r / 4) + 4 h + r % 4 of unit u0 + i
  const float sxw = sx * wsc;
  const int zs = g.zoff * wsm;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int r = 2 * wave + rr;
    int sum = 0;
#pragma unroll
    for (int w = 0; w < kRowsNW; ++w) sum += red[w][r][lane];
    const int smp = s0 + 8 * (r >> 2) + 4 * h + (r & 3);
    const bool sok = smp < g.n;
    float v = (float)(sum + zs) * sxw;
    if (bias != nullptr) v = v + bch;
    if (has_bn) {
      v = v * bsc;
      v = v + bsh;
    }
    v = act_rt(v, act);
    if (sok && uok) y[(int64_t)smp * g.cout + unit] = v;
    if (stat_out != nullptr) {
      float m = uok ? fabsf(v) : 0.0f;
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
      if (i == 0 && sok) atomic_max_f32(stat_out + smp, m);
    }
    if (COUNT) {
      unsigned long long key = uok ? rows_key(v, (unsigned)unit) : 0ull;
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        const unsigned long long o = shfl_xor_u64(key, off);
        key = o > key ? o : key;
      }
      if (i == 0 && sok)
        __hip_atomic_store(partial + (int64_t)smp * g.ut + ut, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!COUNT) return;

  // the last workgroup of this sample tile to get here sees every workgroup's keys (release: fence + counter; acquire: the
  // counter + fence, keys read past the vector cache)
  // The short hand-shake is a property of gfx942 / gfx950 (sc1 stores are written through and acknowledged before vmcnt
  // reaches 0, sc1 loads bypass the L2), not of the HIP memory model: any other target gets the textbook fences.
#if defined(__gfx942__) || defined(__gfx950__)
  constexpr bool kShortHandshakeOk = true;
#else
  constexpr bool kShortHandshakeOk = false;
#endif
  if (fence_mode == 0 || !kShortHandshakeOk) {
    __threadfence();
  } else {
    // the wait the comment above relies on, spelled out: outside threadgroup-split mode hipcc lowers a workgroup-scope
    // release fence to NOTHING for global memory (found in round 4 by reading the ISA: key store, then straight to
    // s_barrier), so "this wavefront's key stores have been acknowledged" was never actually waited for.  tools/isa_lint.py
    // now checks that `s_waitcnt vmcnt(0)` sits directly in front of the barrier and that the key accesses carry sc1.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0) wg_last = atomicAdd(sync + st, 1u) == (unsigned)(g.ut - 1) ? 1u : 0u;
  __syncthreads();
  if (wg_last == 0u) return;
  if (fence_mode == 0 || !kShortHandshakeOk)
    __threadfence();
  else
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  {
    const int si = threadIdx.x / TPS, j0 = threadIdx.x % TPS;
    const int smp = s0 + si;
    const bool sok = smp < g.n;
    unsigned long long key = 0ull;
    if (sok)
      for (int j = j0; j < g.ut; j += TPS) {
        const unsigned long long o = __hip_atomic_load(partial + (int64_t)smp * g.ut + j, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        key = o > key ? o : key;
      }
#pragma unroll
    for (int off = TPS / 2; off > 0; off >>= 1) {
      const unsigned long long o = shfl_xor_u64(key, off);
      key = o > key ? o : key;
    }
    if (j0 == 0 && sok) {
      const long long pred = (long long)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
      // (adding 1.0 to an integer below 2^24 is exact: the hardware's fp32 atomic add gives what a CAS loop would, without
      // its two trips to memory)
      const long long gt = my_label;
      atomicAdd(&wg_total, 1u);
      if (gt >= 0 && gt < g.cout) {
        unsafeAtomicAdd(counters + 2 + g.cout + gt, 1.0f);
        if (pred == gt) {
          atomicAdd(&wg_correct, 1u);
          unsafeAtomicAdd(counters + 2 + gt, 1.0f);
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (wg_total) unsafeAtomicAdd(counters + 1, (float)wg_total);        // small integers: exact in fp32
    if (wg_correct) unsafeAtomicAdd(counters, (float)wg_correct);
    atomicExch(sync + st, 0u);                                           // left zero for the next call
  }
}

}  // namespace

namespace fqi {

size_t pw_rows_eval_ws_bytes(int64_t n, int64_t cout) {
  const int64_t st = (n + 31) / 32, ut = (cout + 31) / 32;
  return (size_t)((st * 4 + 255) / 256 * 256 + n * ut * 8);
}

// rows form (K2t): planes of ONE pixel, stride 1, no residual, fp32 in and out; every Cout, Cin a multiple of 4.
int pw_try_rows(const PwCall& a, bool* taken) {
  *taken = false;
  const bool eval = a.eval_labels != nullptr;
  const bool shape_ok = a.hw == 1 && a.stride == 1 && a.residual == nullptr && !a.in_c16 && a.out_thr == nullptr &&
                        a.cin % 4 == 0 && a.n * a.cin * 4 < (1ll << 31) && a.n * a.cout * 4 < (1ll << 31);
  if (!(shape_ok && (a.form == 0 || a.form == 8))) {
    FQ_REQUIRE(a.form != 8 && !eval, "fq_pwconv_i8: the rows form takes planes of one pixel, stride 1, no residual, fp32 "
               "tensors and Cin a multiple of 4 (got hw=%lld cin=%lld)", (long long)a.hw, (long long)a.cin);
    return FQ_OK;
  }
  PwRowsGeom g;
  g.n = (int)a.n; g.cin = (int)a.cin; g.kts = (int)(a.cin_pad / 32); g.cout = (int)a.cout;
  g.ut = (int)((a.cout + 31) / 32); g.zoff = a.zoff;
  const int64_t st = (a.n + 31) / 32;
  const int64_t rows_pad = (a.cout + 63) / 64 * 64;
  const int8_t* wfrag = a.wcodes + rows_pad * a.cin_pad;                 // second half of fq_weight_codes' buffer
  FQ_REQUIRE(st * g.ut < (1ll << 31), "fq_pwconv_i8: too many tiles for the rows form");
  if (int rc = pw_zero_stat(a)) return rc;
  unsigned* sync = reinterpret_cast<unsigned*>(a.eval_ws);
  unsigned long long* partial = eval ? reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.eval_ws) +
                                                                            (st * 4 + 255) / 256 * 256)
                                     : nullptr;
  static const int fence_mode = env_int("FQ_ROWS_FENCE", 1);
  if (eval)
    hipLaunchKernelGGL(pwconv_rows_kernel<true>, dim3((unsigned)(st * g.ut)), dim3(kRowsNW * 64), 0, a.st, a.x, wfrag,
                       a.wscale, (const int*)a.wsum, a.bias, a.y, g, a.in_stat, a.in_thr, a.levels, a.lo_neg, kEps,
                       a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out, a.eval_labels, a.eval_counters,
                       partial, sync, fence_mode);
  else
    hipLaunchKernelGGL(pwconv_rows_kernel<false>, dim3((unsigned)(st * g.ut)), dim3(kRowsNW * 64), 0, a.st, a.x, wfrag,
                       a.wscale, (const int*)a.wsum, a.bias, a.y, g, a.in_stat, a.in_thr, a.levels, a.lo_neg, kEps,
                       a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out, (const long long*)nullptr,
                       (float*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr, 0);
  FQ_LAUNCH_CHECK();
  *taken = true;
  return FQ_OK;
}

}  // namespace fqi
