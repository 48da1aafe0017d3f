// libfakequant — the reference's stand-alone "really quantised" convolution, nn.Conv2D(quantized=True)
// (nn/quantized_conv.py:106-159), as ONE entry point per forward: global range of the input (one 4 B/elem pass) -> range
// record, int32 bias codes, per-layer constants (one small launch) -> the convolution with the quantiser on its loads:
//   1x1 (stride 1)            the pointwise forms on the int8 matrix cores (fq_pw_sample / _split / _stream, range mode)
//   dense 3x3 (s1, p1)        the implicit-GEMM kernel of fq_conv3x3 (Cin 64 ... 512)
//   depthwise 3x3 (s1|2, p1)  the depthwise forms of fq_dwconv with integer CODES in the fp32 fmaf chain (exact: 9 taps)
//   every other geometry      qconv_direct_kernel below: one output per thread, exact, slow
// - no im2col tensor, no int32 code tensor, no casts - followed by the SAME direct kernel as a conditional fix-up that
// returns at once unless the range record says the fast kernel's 8-bit / fp32-exact representation did not hold (codes
// spanning 257 values through a double rounding tie, a clip range that excludes the padding zero, |code| > 14 000).
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_common.h"

namespace fqi {
// fq_pwconv.hip / fq_conv3x3.hip / fq_dwconv.hip
int pw_range_call(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const int32_t* ibias,
                  float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w, int stride,
                  const float* rec, const float* bn_scale, const float* bn_shift, int act, float* stat_out, hipStream_t st,
                  bool* taken);
bool conv3x3_range_shape_ok(int64_t cin, int64_t cout);
int conv3x3_range_call(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const int32_t* ibias,
                       float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, const float* rec,
                       const float* bn_scale, const float* bn_shift, int act, float* stat_out, hipStream_t st);
int dw_range_call(const float* x, const float* wcodes_f32, float* y, int64_t n, int64_t c, int64_t h, int64_t wdt, int stride,
                  const float* rec, const float* svec, const float* bn_scale, const float* bn_shift, int act,
                  float* stat_out, hipStream_t st);
}  // namespace fqi

namespace {

constexpr int kKindDirect = 0, kKindPw = 1, kKindC3 = 2, kKindDw = 3;
constexpr int kFlagFixup = 1;
constexpr int kSlots = 16, kSlotStride = 32;            // range-pass lines: 16 x 128 bytes ({min, max} at the head of each)
constexpr size_t kWsHeader = 2 * kSlots * kSlotStride * 4 + 256;   // two sets of lines | range record | ...

// `_quantize` (nn/quantized_conv.py:54-61): scale of a clip range, and a code
__device__ __forceinline__ float range_scale(float mn, float mx) { return (mx == -mn) ? (mx / 127.0f) : ((mx - mn) / 255.0f); }
__device__ __forceinline__ int range_code(float v, float mn, float mx, float sc) {
  return (int)roundf(fminf(fmaxf(v, mn), mx) / sc);
}

struct QconvShape {
  int n, cin, h, w, cout, kh, kw, sh, sw, ph, pw, groups, ho, wo;
};

// ---- the range record and the per-layer constants ----------------------------------------------------------------------
// mm: kSlots lines of {min, max} as the range pass left them (workgroup b of that pass folds its result into line b % kSlots:
// same-address atomics are served one after the other, ~10 ns each, so 2048 workgroups on ONE pair of words cost the pass
// 40 us - profiles/r3_atomic_probe.txt), combined and re-initialised here for the next forward.  wrec: the weights' record
// (fq_qconv_weights_prepare).  One workgroup.
__global__ __launch_bounds__(kBlock) void qconv_finish_kernel(float* __restrict__ mm, int mode, int padded, float fix_min,
                                                              float fix_max, int kind, const float* __restrict__ wrec,
                                                              const float* __restrict__ bias, int cout,
                                                              float* __restrict__ rec, int* __restrict__ ibias,
                                                              float* __restrict__ svec,
                                                              const float* __restrict__ stat, int nstat,
                                                              const float* __restrict__ x, int64_t numel) {
  __shared__ float sh[4];
  __shared__ float rng[2];
  __shared__ float red3[3][4];
  // (round 4: this one-workgroup kernel sits between every two layers of a fused net - 26 x 5.9 us in the MobileNet step, all of
  // it dependent round trips; everything it reads is now requested up front and the three reductions share one barrier)
  const float w_scale = wrec[kRecScale];
  const float bias0 = (bias != nullptr && ibias != nullptr && (int)threadIdx.x < cout) ? bias[threadIdx.x] : 0.0f;
  // A fused producer's per-sample maxima of a NON-NEGATIVE tensor (BatchNorm + ReLU in its epilogue) stand in for the range
  // pass: max = their maximum, and the minimum is 0 as soon as the tensor holds one zero (half of a ReLU's outputs) - found
  // by the first strides of a scan that only runs to the end, as a plain minimum, when there is none.  Needed for uint8
  // without padding only (with padding the zero is there by construction; int8 ranges are [-max, max]).  A producer whose
  // statistic did not survive (stat[0] < 0: its layer was recomputed by the exact kernel) costs one full scan here.
  if (stat != nullptr) {
    const bool want_min = mode == FQ_CODES_UINT8 && !padded;
    float smax = 0.0f;
    for (int i = threadIdx.x; i < nstat; i += kBlock) smax = fmaxf(smax, stat[i]);
    const float s0 = stat[0];
    float lmin = INFINITY, lmax = 0.0f;
    if (want_min) {                                                     // the scan's first stride, in flight with the statistic
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t i = threadIdx.x + (int64_t)u * kBlock;
        v[u] = i < numel ? x[i] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (threadIdx.x + (int64_t)u * kBlock < numel) {
          lmin = fminf(lmin, v[u]);
          lmax = fmaxf(lmax, v[u]);
        }
      }
    }
    const bool stale = s0 < 0.0f;
    // the scan goes on while the tensor is to be read in full (stale), or no zero has turned up yet
    bool more = stale || (want_min && !__syncthreads_or(lmin <= 0.0f));
    if (more) {
      for (int64_t base = want_min ? (int64_t)kBlock * 8 : 0; base < numel; base += kBlock * 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t i = base + threadIdx.x + (int64_t)u * kBlock;
          if (i < numel) {
            const float v = x[i];
            lmin = fminf(lmin, v);
            lmax = fmaxf(lmax, v);
          }
        }
        if (!stale && __syncthreads_or(lmin <= 0.0f)) break;            // a zero: the minimum of a non-negative tensor
      }
    }
    // one exchange for the three reductions
    {
      float a0 = smax, a1 = lmin, a2 = lmax;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        a0 = fmaxf(a0, __shfl_xor(a0, off, 64));
        a1 = fminf(a1, __shfl_xor(a1, off, 64));
        a2 = fmaxf(a2, __shfl_xor(a2, off, 64));
      }
      if ((threadIdx.x & 63) == 0) {
        red3[0][threadIdx.x >> 6] = a0;
        red3[1][threadIdx.x >> 6] = a1;
        red3[2][threadIdx.x >> 6] = a2;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        const float bmax = fmaxf(fmaxf(red3[0][0], red3[0][1]), fmaxf(red3[0][2], red3[0][3]));
        const float bmin = fminf(fminf(red3[1][0], red3[1][1]), fminf(red3[1][2], red3[1][3]));
        const float bscan = fmaxf(fmaxf(red3[2][0], red3[2][1]), fmaxf(red3[2][2], red3[2][3]));
        rng[0] = (stale || want_min) ? bmin : 0.0f;
        rng[1] = stale ? bscan : bmax;
      }
    }
  }
  if (threadIdx.x == 0) {
    float mn, mx;
    if (stat != nullptr) {
      mx = rng[1];
      if (mode == FQ_CODES_INT8) {
        mx = fmaxf(mx, -rng[0]);                 // (max|x| when the tensor was scanned; rng[0] is 0 otherwise)
        mn = -mx;
      } else {
        mn = padded ? fminf(rng[0], 0.0f) : rng[0];
      }
    } else if (mode == FQ_CODES_INT8) {
      mx = 0.0f;
      for (int k = 0; k < kSlots; ++k) mx = fmaxf(mx, mm[k * kSlotStride + 1]);
      mn = -mx;
    } else if (mode == FQ_CODES_UINT8) {
      mn = INFINITY;
      mx = -INFINITY;
      for (int k = 0; k < kSlots; ++k) {
        mn = fminf(mn, mm[k * kSlotStride]);
        mx = fmaxf(mx, mm[k * kSlotStride + 1]);
      }
      if (padded) {                              // the reference pads BEFORE it takes the range (:108-113)
        mn = fminf(mn, 0.0f);
        mx = fmaxf(mx, 0.0f);
      }
    } else {
      mn = fix_min;
      mx = fix_max;
    }
    for (int k = 0; k < kSlots; ++k) {
      mm[k * kSlotStride] = INFINITY;
      mm[k * kSlotStride + 1] = mode == FQ_CODES_INT8 ? 0.0f : -INFINITY;
    }
    const float sc = range_scale(mn, mx);
    const bool sym = mx == -mn;
    const float ql = roundf(mn / sc), qh = roundf(mx / sc);
    const bool finite = sc > 0.0f && sc < INFINITY && fabsf(ql) < 1e9f && fabsf(qh) < 1e9f;
    const int L = finite ? (int)ql : 0, H = finite ? (int)qh : 0;
    // An ALL-ZERO tensor (a dead ReLU) has scale 0: every code is (int)NaN = 0 in the reference's arithmetic and the layer's
    // output is its bias / BatchNorm shift.  The fast kernels get exactly that from a record that DIVIDES by 1 and
    // multiplies back by 0 (a zero divisor would hand the depthwise kernel's fp32 codes a NaN: round 4's advice) - no fix-up.
    const bool all_zero = sc == 0.0f;
    int flags = 0;
    if (!finite && !all_zero) flags |= kFlagFixup;
    if (!sym && H - L > 255) flags |= kFlagFixup;                       // 257 codes: do not fit a byte
    if (padded && !(mn <= 0.0f && mx >= 0.0f)) flags |= kFlagFixup;     // the padding zero is clipped to a non-zero code
    if (kind == kKindDw && (L < -14000 || H > 14000)) flags |= kFlagFixup;   // 9 * 127 * |code| must stay below 2^24
    rec[kRecHi] = mx;
    rec[kRecLo] = mn;
    rec[kRecDenom] = all_zero ? 1.0f : sc;
    rec[kRecMul] = kind == kKindDw ? 1.0f : sc;
    rec[kRecUbias] = __int_as_float(sym ? 128 : -L);
    rec[kRecFlags] = __int_as_float(flags);
    rec[kRecScale] = sc;
    rec[kRecLcode] = __int_as_float(L);
    sh[0] = sc;
  }
  __syncthreads();
  const float b_scale = sh[0] * w_scale;                                // in_scale * w_scale, fp32 (:123)
  const float b_max = b_scale * 2147483648.0f;                          // (:124)
  for (int c = threadIdx.x; c < cout; c += kBlock) {
    if (ibias != nullptr) {
      int code = 0;
      if (bias != nullptr) {                                            // clip, round, cast (:125-127); 2^31 wraps like the cast
        const float bv = c == (int)threadIdx.x ? bias0 : bias[c];
        const float q = roundf(fminf(fmaxf(bv, -b_max), b_max) / b_scale);
        code = (int)(unsigned)(long long)q;
      }
      ibias[c] = code;
    }
    if (svec != nullptr) svec[c] = b_scale;
  }
}

// two sets of lines: [0, kSlots) for uint8 ranges ({+inf, -inf}), [kSlots, 2 kSlots) for int8 ones ({+inf, 0})
__global__ void qconv_ws_init_kernel(float* mm) {
  const int k = threadIdx.x;
  if (blockIdx.x == 0 && k < 2 * kSlots) {
    mm[k * kSlotStride] = INFINITY;
    mm[k * kSlotStride + 1] = k < kSlots ? -INFINITY : 0.0f;
  }
}

// the range pass: K6's walk (fq_common.h: minmax_kernel) with the workgroup's result folded into line blockIdx % kSlots
template <bool WANT_MIN, bool USE_ABS>
__global__ __launch_bounds__(kBlock) void qconv_minmax_kernel(const float* __restrict__ x, int64_t numel, int vec_ok,
                                                              float* __restrict__ mm) {
  __shared__ float red[4];
  float mx = USE_ABS ? 0.0f : -INFINITY, mn = INFINITY;
  auto take = [&](float v) {
    mx = fmaxf(mx, stat_of<USE_ABS>(v));
    if (WANT_MIN) mn = fminf(mn, v);
  };
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
  float* line = mm + (blockIdx.x % kSlots) * kSlotStride;
  mx = block_max(mx, red);
  if (threadIdx.x == 0) atomic_max_f32(line + 1, mx);
  if (WANT_MIN) {
    mn = block_min(mn, red);
    if (threadIdx.x == 0) atomic_min_f32(line, mn);
  }
}

// ---- weights --------------------------------------------------------------------------------------------------------------
// wrec <- {hi, lo, scale, scale, ubias, flags, scale, L} of the whole weight tensor (`quantize(F, weight, dtype)`, :63-72)
__global__ void qconv_wrec_kernel(const float* __restrict__ mm, int mode, float fix_min, float fix_max,
                                  float* __restrict__ wrec) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float mn, mx;
  if (mode == FQ_CODES_INT8) {
    mx = mm[1];
    mn = -mx;
  } else if (mode == FQ_CODES_UINT8) {
    mn = mm[0];
    mx = mm[1];
  } else {
    mn = fix_min;
    mx = fix_max;
  }
  const float sc = range_scale(mn, mx);
  wrec[kRecHi] = mx;
  wrec[kRecLo] = mn;
  wrec[kRecDenom] = sc;
  wrec[kRecMul] = sc;
  wrec[kRecUbias] = __int_as_float(0);
  wrec[kRecFlags] = __int_as_float(mx == -mn ? 0 : kFlagFixup);          // asymmetric weight codes do not fit int8
  wrec[kRecScale] = sc;
  wrec[kRecLcode] = __int_as_float(0);
}

// One workgroup per padded row of the code matrix (symmetric weights only): code = roundf(clip(w) / scale) - no epsilon
// (:55-61) -, row-major int8 + the MFMA-fragment copy fq_weight_codes also leaves (fq_pwconv.hip), row sums, the scale per
// row.  kind C3: the row is permuted to (tap, ci) order, k = (ky * 3 + kx) * Cin + ci.  kind DW: fp32 copies of the codes in
// the weight's own layout instead (the depthwise kernels multiply in fp32).
__global__ __launch_bounds__(kBlock) void qconv_weight_codes_kernel(const float* __restrict__ w, int rows, int row_len,
                                                                    int row_pad, int kind, int cin,
                                                                    const float* __restrict__ wrec,
                                                                    int8_t* __restrict__ codes, int8_t* __restrict__ frag,
                                                                    float* __restrict__ scales, int* __restrict__ rowsum,
                                                                    float* __restrict__ fcodes) {
  __shared__ int red[4];
  const int r = blockIdx.x;
  const float mx = wrec[kRecHi], mn = wrec[kRecLo], sc = wrec[kRecScale];
  if (kind == kKindDw) {
    if (r < rows)
      for (int i = threadIdx.x; i < row_len; i += kBlock)
        fcodes[(int64_t)r * row_len + i] = (float)range_code(w[(int64_t)r * row_len + i], mn, mx, sc);
    return;
  }
  int8_t* dst = codes + (int64_t)r * row_pad;
  const int kts = row_pad >> 5;
  auto frag_at = [&](int i) -> int8_t* {
    const int kt = i >> 5, hs = (i >> 4) & 1, b = i & 15;
    return frag + ((((int64_t)(r >> 5) * kts + kt) << 6) + (r & 31) + 32 * hs) * 16 + b;
  };
  if (r >= rows) {
    for (int i = threadIdx.x; i < row_pad; i += kBlock) {
      dst[i] = 0;
      *frag_at(i) = 0;
    }
    return;
  }
  int acc = 0;
  for (int i = threadIdx.x; i < row_pad; i += kBlock) {
    int c = 0;
    if (i < row_len) {
      const int src = kind == kKindC3 ? (i % cin) * 9 + i / cin : i;
      c = range_code(w[(int64_t)r * row_len + src], mn, mx, sc);
    }
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
    scales[r] = sc;
  }
}

// ---- the exact direct form ---------------------------------------------------------------------------------------------------
// One output element per thread, every geometry of the block (any kernel size, stride, padding, groups; symmetric or
// asymmetric codes on either side): both operands quantised where they are read, int32 sums in wrapping arithmetic (the
// reference casts to int32, :144), + bias code, activation on the integer, dequantise (:146-158).  `only_if_flagged`: return
// unless the range record asks for the recomputation.
__global__ __launch_bounds__(kBlock) void qconv_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const int* __restrict__ ibias, float* __restrict__ y,
                                                              QconvShape s, const float* __restrict__ rec,
                                                              const float* __restrict__ wrec, int act, int only_if_flagged,
                                                              const float* __restrict__ bn_scale,
                                                              const float* __restrict__ bn_shift,
                                                              float* __restrict__ stat_out) {
  if (only_if_flagged && !((__float_as_int(rec[kRecFlags]) | __float_as_int(wrec[kRecFlags])) & kFlagFixup)) return;
  // this kernel keeps no per-sample statistic: a consumer that was promised one takes its range from the tensor instead
  if (stat_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) stat_out[0] = -1.0f;
  const float xh = rec[kRecHi], xl = rec[kRecLo], xs = rec[kRecScale];
  const float wh = wrec[kRecHi], wl = wrec[kRecLo], wsc = wrec[kRecScale];
  const float deq = xs * wsc;
  const int cin_g = s.cin / s.groups, cout_g = s.cout / s.groups;
  const int64_t total = (int64_t)s.n * s.cout * s.ho * s.wo;
  const int zero_code = range_code(0.0f, xl, xh, xs);                   // the padding is quantised with the tensor (:108-113)
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
    const int ow = (int)(idx % s.wo);
    const int oh = (int)((idx / s.wo) % s.ho);
    const int co = (int)((idx / ((int64_t)s.wo * s.ho)) % s.cout);
    const int n = (int)(idx / ((int64_t)s.wo * s.ho * s.cout));
    const int g = co / cout_g;
    unsigned acc = 0u;
    for (int ci = 0; ci < cin_g; ++ci) {
      const float* xp = x + ((int64_t)n * s.cin + g * cin_g + ci) * s.h * s.w;
      const float* wp = w + ((int64_t)co * cin_g + ci) * s.kh * s.kw;
      for (int ky = 0; ky < s.kh; ++ky) {
        const int ih = oh * s.sh - s.ph + ky;
        for (int kx = 0; kx < s.kw; ++kx) {
          const int iw = ow * s.sw - s.pw + kx;
          const bool in = ih >= 0 && ih < s.h && iw >= 0 && iw < s.w;
          const int cx = in ? range_code(xp[(int64_t)ih * s.w + iw], xl, xh, xs) : zero_code;
          const int cw = range_code(wp[ky * s.kw + kx], wl, wh, wsc);
          acc += (unsigned)cx * (unsigned)cw;
        }
      }
    }
    if (ibias != nullptr) acc += (unsigned)ibias[co];
    int v = (int)acc;
    if (act == FQ_ACT_RELU && bn_scale == nullptr) v = v > 0 ? v : 0;      // on the integers (:154-155)
    float out = (float)v * deq;
    if (bn_scale != nullptr) {                 // a BatchNorm folded behind the block: the activation follows IT
      out = out * bn_scale[co];
      out = out + bn_shift[co];
      out = act_rt(out, act);
    }
    y[idx] = out;
  }
}

// layout of the prepared-weights buffer (bytes): record | scales | row sums | codes (+ fragment copy) or fp32 codes
struct WLayout {
  int kind;
  int64_t rows, row_len, row_pad, rows_pad;
  size_t off_scales, off_rowsum, off_codes, total;
};

inline int qconv_kind(int64_t cin, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw, int groups) {
  if (groups == 1 && kh == 1 && kw == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0 && cin <= 8192) return kKindPw;
  if (groups == 1 && kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && conv3x3_range_shape_ok(cin, cout))
    return kKindC3;
  if (groups == cin && groups == cout && kh == 3 && kw == 3 && sh == sw && (sh == 1 || sh == 2) && ph == 1 && pw == 1)
    return kKindDw;
  return kKindDirect;
}

inline WLayout wlayout(int64_t cin, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw, int groups) {
  WLayout L;
  L.kind = qconv_kind(cin, cout, kh, kw, sh, sw, ph, pw, groups);
  L.rows = cout;
  L.row_len = (cin / groups) * kh * kw;
  L.row_pad = (L.row_len + 63) / 64 * 64;
  L.rows_pad = (cout + 63) / 64 * 64;
  L.off_scales = 64;
  L.off_rowsum = L.off_scales + (size_t)L.rows_pad * 4;
  L.off_codes = (L.off_rowsum + (size_t)L.rows_pad * 4 + 255) / 256 * 256;
  size_t body = 0;
  if (L.kind == kKindPw || L.kind == kKindC3) body = 2 * (size_t)L.rows_pad * (size_t)L.row_pad;
  if (L.kind == kKindDw) body = (size_t)L.rows * (size_t)L.row_len * 4;
  L.total = L.off_codes + body + 256;
  return L;
}

}  // namespace

using namespace fqi;

namespace {
// Which quantiser a prepared-weights buffer was made with (host-side: the record itself lives on the device).  The fast
// kernels multiply int8 codes of ONE symmetric grid: a buffer prepared with uint8 / fixed-range weights (codes that need not
// fit int8: wrec carries kFlagFixup) must take the exact direct kernel whatever `force_direct` says, and a buffer this
// process did not prepare (copied, or prepared through another handle) keeps the conditional fix-up launch behind the fast
// kernel, which reads both records on the device.
std::mutex g_wmode_mu;
std::vector<std::pair<const void*, int>> g_wmode;
void wmode_set(const void* wbuf, int mode) {
  std::lock_guard<std::mutex> lk(g_wmode_mu);
  for (auto& e : g_wmode)
    if (e.first == wbuf) {
      e.second = mode;
      return;
    }
  g_wmode.emplace_back(wbuf, mode);
}
int wmode_get(const void* wbuf) {                                       // -1: unknown
  std::lock_guard<std::mutex> lk(g_wmode_mu);
  for (const auto& e : g_wmode)
    if (e.first == wbuf) return e.second;
  return -1;
}
}  // namespace

extern "C" {

size_t fq_qconv_weights_bytes(int64_t cin, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw, int groups) {
  if (cin <= 0 || cout <= 0 || groups <= 0 || cin % groups || cout % groups || kh <= 0 || kw <= 0) return 0;
  return wlayout(cin, cout, kh, kw, sh, sw, ph, pw, groups).total;
}

int fq_qconv_kind(int64_t cin, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw, int groups) {
  return qconv_kind(cin, cout, kh, kw, sh, sw, ph, pw, groups);
}

int fq_qconv_weights_prepare(const float* w, int64_t cin, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw,
                             int groups, int weight_mode, float w_min, float w_max, void* wbuf, void* ws,
                             fqStream_t stream) {
  FQ_REQUIRE(w && wbuf && ws, "fq_qconv_weights_prepare: null pointer");
  FQ_REQUIRE(cin > 0 && cout > 0 && groups > 0 && cin % groups == 0 && cout % groups == 0 && kh > 0 && kw > 0,
             "fq_qconv_weights_prepare: bad shape");
  FQ_REQUIRE(weight_mode >= FQ_CODES_INT8 && weight_mode <= FQ_CODES_RANGE, "unknown out type: %d", weight_mode);
  FQ_REQUIRE(aligned16(wbuf), "fq_qconv_weights_prepare: wbuf must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const WLayout L = wlayout(cin, cout, kh, kw, sh, sw, ph, pw, groups);
  char* base = (char*)wbuf;
  float* wrec = (float*)base;
  float* mm = (float*)ws;
  const int64_t numel = cout * L.row_len;
  if (weight_mode != FQ_CODES_RANGE) {
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, st, mm, (int64_t)1, INFINITY);
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, st, mm + 1, (int64_t)1,
                       weight_mode == FQ_CODES_INT8 ? 0.0f : -INFINITY);
    const int grid = grid_for((numel + kChunk - 1) / kChunk);
    if (weight_mode == FQ_CODES_INT8)
      hipLaunchKernelGGL((minmax_kernel<false, true>), dim3(grid), dim3(kBlock), 0, st, w, numel, aligned16(w) ? 1 : 0, mm,
                         mm + 1);
    else
      hipLaunchKernelGGL((minmax_kernel<true, false>), dim3(grid), dim3(kBlock), 0, st, w, numel, aligned16(w) ? 1 : 0, mm,
                         mm + 1);
  }
  hipLaunchKernelGGL(qconv_wrec_kernel, dim3(1), dim3(64), 0, st, (const float*)mm, weight_mode, w_min, w_max, wrec);
  if (L.kind != kKindDirect) {
    int8_t* codes = (int8_t*)(base + L.off_codes);
    const unsigned blocks = (unsigned)(L.kind == kKindDw ? L.rows : L.rows_pad);
    hipLaunchKernelGGL(qconv_weight_codes_kernel, dim3(blocks), dim3(kBlock), 0, st, w, (int)L.rows, (int)L.row_len,
                       (int)L.row_pad, L.kind, (int)(cin / groups), (const float*)wrec, codes,
                       codes + L.rows_pad * L.row_pad, (float*)(base + L.off_scales), (int*)(base + L.off_rowsum),
                       (float*)(base + L.off_codes));
  }
  FQ_LAUNCH_CHECK();
  wmode_set(wbuf, weight_mode);
  return FQ_OK;
}

size_t fq_qconv_workspace_bytes(int64_t cout) { return cout > 0 ? kWsHeader + (size_t)((cout + 63) / 64 * 64) * 8 : 0; }

int fq_qconv_workspace_init(void* ws, fqStream_t stream) {
  FQ_REQUIRE(ws, "fq_qconv_workspace_init: null pointer");
  hipLaunchKernelGGL(qconv_ws_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (float*)ws);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_qconv2d_forward(const float* x, const float* w, const void* wbuf, const float* bias, float* y, int64_t n, int64_t cin,
                       int64_t h, int64_t wdt, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw, int groups,
                       int input_mode, float in_min, float in_max, const float* in_stat, int act, const float* bn_scale,
                       const float* bn_shift, float* stat_out, void* ws, int force_direct, fqStream_t stream) {
  FQ_REQUIRE(x && w && wbuf && y && ws, "fq_qconv2d_forward: null pointer");
  FQ_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && wdt > 0 && groups > 0 && cin % groups == 0 && cout % groups == 0 &&
                 kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0,
             "fq_qconv2d_forward: bad shape");
  FQ_REQUIRE(h + 2 * ph >= kh && wdt + 2 * pw >= kw, "fq_qconv2d_forward: kernel larger than the padded input");
  FQ_REQUIRE(input_mode >= FQ_CODES_INT8 && input_mode <= FQ_CODES_RANGE, "unknown out type: %d", input_mode);
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_qconv2d_forward: bn_scale and bn_shift go together");
  FQ_REQUIRE(in_stat == nullptr || input_mode != FQ_CODES_RANGE, "fq_qconv2d_forward: in_stat and a fixed range exclude each "
             "other");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act == FQ_ACT_NONE || act == FQ_ACT_RELU || (act == FQ_ACT_RELU6 && bn_scale != nullptr),
             "fq_qconv2d_forward: activation %d (none or relu: the block applies it to the int32 sums; relu6 only behind a "
             "folded BatchNorm)", act);
  hipStream_t st = (hipStream_t)stream;
  const WLayout L = wlayout(cin, cout, kh, kw, sh, sw, ph, pw, groups);
  const char* wb = (const char*)wbuf;
  const float* wrec = (const float*)wb;
  // workspace: {min, max} | record | int32 bias codes | per-channel dequantisation factor | zeros
  float* mm = (float*)ws;                                  // uint8 lines; the int8 set follows
  float* mm8 = mm + kSlots * kSlotStride;
  float* rec = mm + 2 * kSlots * kSlotStride;
  const int64_t cpad = (cout + 63) / 64 * 64;
  int* ibias = (int*)((char*)ws + kWsHeader);
  float* svec = (float*)(ibias + cpad);
  QconvShape s;
  s.n = (int)n; s.cin = (int)cin; s.h = (int)h; s.w = (int)wdt; s.cout = (int)cout; s.kh = kh; s.kw = kw; s.sh = sh; s.sw = sw;
  s.ph = ph; s.pw = pw; s.groups = groups;
  // (the reference collects range(0, H - kh + 1, s) windows, :42-47)
  s.ho = (int)((h + 2 * ph - kh) / sh + 1);
  s.wo = (int)((wdt + 2 * pw - kw) / sw + 1);
  const int64_t numel = n * cin * h * wdt, out_numel = n * cout * (int64_t)s.ho * s.wo;
  FQ_REQUIRE(numel < (1ll << 40) && out_numel < (1ll << 40), "fq_qconv2d_forward: tensor too large");
  const int wmode = wmode_get(wbuf);
  // (asymmetric / fixed-range weight codes are not on the int8 grid the fast kernels multiply: the exact kernel, whatever
  // the caller asked for)
  int kind = (force_direct || (wmode != -1 && wmode != FQ_CODES_INT8)) ? kKindDirect : L.kind;
  // ---- 1. the input's range (unless given, or known from the producer's per-sample maxima) ----------------------------
  if (input_mode != FQ_CODES_RANGE && in_stat == nullptr) {
    ProfScope prof(FQ_KERNEL_GLOBAL_MAX, 4.0 * (double)numel, st);
    const int grid = grid_for((numel + kChunk - 1) / kChunk);
    if (input_mode == FQ_CODES_INT8)
      hipLaunchKernelGGL((qconv_minmax_kernel<false, true>), dim3(grid), dim3(kBlock), 0, st, x, numel, aligned16(x) ? 1 : 0,
                         mm8);
    else
      hipLaunchKernelGGL((qconv_minmax_kernel<true, false>), dim3(grid), dim3(kBlock), 0, st, x, numel, aligned16(x) ? 1 : 0,
                         mm);
  }
  // ---- 2. record + constants ------------------------------------------------------------------------------------------------
  hipLaunchKernelGGL(qconv_finish_kernel, dim3(1), dim3(kBlock), 0, st, input_mode == FQ_CODES_INT8 ? mm8 : mm, input_mode,
                     (ph > 0 || pw > 0) ? 1 : 0, in_min, in_max, kind, wrec, bias, (int)cout, rec, ibias,
                     kind == kKindDw ? svec : (float*)nullptr, in_stat, (int)n, x, numel);
  FQ_LAUNCH_CHECK();
  // ---- 3. the convolution ---------------------------------------------------------------------------------------------------
  const int* ib = bias != nullptr ? ibias : nullptr;
  const int zflag = prezeroed ? FQ_STAT_PREZEROED : 0;
  bool fast = false;
  if (kind == kKindPw) {
    if (int rc = pw_range_call(x, (const int8_t*)(wb + L.off_codes), (const float*)(wb + L.off_scales),
                               (const int32_t*)(wb + L.off_rowsum), ib, y, n, cin, L.row_pad, cout, h, wdt, 1, rec, bn_scale,
                               bn_shift, act | zflag, stat_out, st, &fast))
      return rc;
  } else if (kind == kKindC3) {
    if (int rc = conv3x3_range_call(x, (const int8_t*)(wb + L.off_codes), (const float*)(wb + L.off_scales),
                                    (const int32_t*)(wb + L.off_rowsum), ib, y, n, cin, cout, h, wdt, rec, bn_scale, bn_shift,
                                    act | zflag, stat_out, st))
      return rc;
    fast = true;
  } else if (kind == kKindDw && bias == nullptr) {
    if (int rc = dw_range_call(x, (const float*)(wb + L.off_codes), y, n, cin, h, wdt, sh, rec, svec, bn_scale, bn_shift,
                               act | zflag, stat_out, st))
      return rc;
    fast = true;
  }
  // ---- 4. the exact direct form: the whole layer, or the conditional fix-up behind a fast kernel ------------------------------
  // (symmetric ranges and [0, max] ranges always fit a byte - L = 0 or codes within +-127 - so no fix-up can be asked for; the
  // per-sample statistic of the output is only offered there, where the fast kernel's result is final)
  // (an all-zero tensor - scale 0 - is no fix-up case: qconv_finish_kernel writes a record the fast kernels are exact with,
  // tests/test_gpu_qconv.py, all-zero inputs; a tensor that holds Inf / NaN has no defined result in the reference either.)
  // The shortcut needs weights KNOWN to be symmetric int8 (a buffer of unknown origin gets the conditional launch, which
  // reads the weight record's flag too).
  const bool representable = input_mode == FQ_CODES_INT8 || (in_stat != nullptr && (ph > 0 || pw > 0));
  if (fast && representable && wmode == FQ_CODES_INT8) return FQ_OK;
  // (as a conditional fix-up - which a record of a [0, max] range with one zero in the tensor never asks for - the launch
  // returns at once: one workgroup per CU keeps that at ~2 us instead of the 4.6 us of 4096 workgroups; the recomputation
  // itself, when it does run, walks its outputs with that smaller grid)
  const int64_t want = (out_numel + kBlock - 1) / kBlock;
  const int64_t cap = (int64_t)num_cu() * (fast ? 1 : 16);
  const int grid = (int)(want < cap ? want : cap);
  hipLaunchKernelGGL(qconv_direct_kernel, dim3(grid), dim3(kBlock), 0, st, x, w, ib, y, s, (const float*)rec, wrec, act,
                     fast ? 1 : 0, bn_scale, bn_shift, stat_out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // extern "C"
