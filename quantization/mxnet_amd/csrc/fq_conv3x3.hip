// libfakequant — K2n dense 3x3 convolution (stride 1, pad 1) on int8 codes: the 3x3 layers of the ResNet bottlenecks
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_conv3x3_kernel.h"

namespace {

// fq_weight_slices: one workgroup per (padded) row.  p = 2^e, the smallest power of two with max|w| <= p * 2^20;
// m = rint(w / p) (exact division; |m| <= 2^20); balanced base-128 digits d3 = ((m + 64) mod 128) - 64, ... so that
// m = d1 2^14 + d2 2^7 + d3 with |d_i| <= 64 (d1: |m| / 2^14 + 1 <= 65).  Slice s: row-major codes, then the fragment-major
// copy (weight_codes_kernel's layout), row sums for the re-centring of unsigned activation codes.
__global__ __launch_bounds__(kBlock) void weight_slices_kernel(const float* __restrict__ w, int rows, int row_len, int row_pad,
                                                               int rows_pad, const float* __restrict__ rmax,
                                                               int8_t* __restrict__ codes, float* __restrict__ pscale,
                                                               int* __restrict__ rowsum) {
  __shared__ int red[3][4];
  const int r = blockIdx.x;
  const int kts = row_pad >> 5;
  const int64_t slice = 2ll * rows_pad * row_pad;
  auto put = [&](int sl, int i, int c) {
    int8_t* base = codes + sl * slice;
    base[(int64_t)r * row_pad + i] = (int8_t)c;
    const int kt = i >> 5, hs = (i >> 4) & 1, b = i & 15;
    base[(int64_t)rows_pad * row_pad + ((((int64_t)(r >> 5) * kts + kt) << 6) + (r & 31) + 32 * hs) * 16 + b] = (int8_t)c;
  };
  if (r >= rows) {
    for (int i = threadIdx.x; i < row_pad; i += kBlock)
      for (int sl = 0; sl < 3; ++sl) put(sl, i, 0);
    return;
  }
  const float mx = rmax[r];
  int e = 0;
  if (mx > 0.0f) {
    (void)frexpf(mx, &e);                                              // mx = f * 2^e, 0.5 <= f < 1  ->  mx < 2^e
    if (ldexpf(1.0f, e - 1) == mx) e -= 1;                              // a power of two itself: mx = 2^(e-1)
  }
  const float p = mx > 0.0f ? ldexpf(1.0f, e - 20) : 1.0f;
  int a0 = 0, a1 = 0, a2 = 0;
  for (int i = threadIdx.x; i < row_pad; i += kBlock) {
    int d1 = 0, d2 = 0, d3 = 0;
    if (i < row_len) {
      const int m = (int)rintf(w[(int64_t)r * row_len + i] / p);
      d3 = ((m + 64) & 127) - 64;
      const int m1 = (m - d3) >> 7;
      d2 = ((m1 + 64) & 127) - 64;
      d1 = (m1 - d2) >> 7;
    }
    put(0, i, d1);
    put(1, i, d2);
    put(2, i, d3);
    a0 += d1; a1 += d2; a2 += d3;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_xor(a0, off, 64);
    a1 += __shfl_xor(a1, off, 64);
    a2 += __shfl_xor(a2, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a0;
    red[1][threadIdx.x >> 6] = a1;
    red[2][threadIdx.x >> 6] = a2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int sl = 0; sl < 3; ++sl) rowsum[sl * rows + r] = red[sl][0] + red[sl][1] + red[sl][2] + red[sl][3];
    pscale[r] = p;
  }
}

}  // namespace

using namespace fqi;

namespace fqi {
int conv3x3_c16_launch(const float* x, const int8_t* wfrag, const float* wscale, const int32_t* wsum, const float* bias,
                       float* y, const void* geom, int kt, int ptw, int wc, int64_t grid, size_t lds, hipStream_t st,
                       const float* in_stat, int n, const float* in_thr, float levels, int lo_neg, float* out_current_max,
                       const float* bn_scale, const float* bn_shift, int act, float* stat_out, bool in16,
                       const float* out_thr, bool* launched);      // fq_conv3x3_16.hip
}

namespace {

int conv3x3_launch(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                   float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, const float* in_stat,
                   const float* in_thr, int in_width, unsigned in_flags, float* out_current_max, const float* bn_scale,
                   const float* bn_shift, int act, float* stat_out, fqStream_t stream, int nsl, bool in_c16 = false,
                   const float* out_thr = nullptr, int out_width = 8, unsigned out_flags = 0, bool range = false) {
  // range: in_thr is a range record (fq_common.h: kRangeMode) and bias holds int32 codes (nn.Conv2D(quantized=True))
  FQ_REQUIRE(x && wcodes && wscale && wsum && y, "fq_conv3x3_i8: null pointer");
  const bool c16 = in_c16 || out_thr != nullptr;
  FQ_REQUIRE(!c16 || nsl == 1, "fq_conv3x3_i8_c16: the three-slice form takes fp32 tensors");
  FQ_REQUIRE(!in_c16 || in_thr != nullptr, "fq_conv3x3_i8_c16: a C16 input was quantised with a stored threshold: give in_thr");
  FQ_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0 && n * h * w < (1ll << 31) - 4096 && w < 4096,
             "fq_conv3x3_i8: bad shape");
  const int kt = (int)(cin / 32);
  FQ_REQUIRE(cin % 32 == 0 && (kt == 2 || kt == 4 || kt == 8 || kt == 16), "fq_conv3x3_i8: Cin must be 64, 128, 256 or "
             "512 (got %lld): slabs of 32 channels must not straddle taps and 9 * Cin codes must sum within int32",
             (long long)cin);
  FQ_REQUIRE(cout >= 32, "fq_conv3x3_i8: Cout must be at least 32, got %lld", (long long)cout);
  FQ_REQUIRE(in_stat != nullptr || in_thr != nullptr, "fq_conv3x3_i8: give in_stat (online), in_thr (offline) or both "
             "(offline, the statistic only feeds out_current_max)");
  FQ_REQUIRE(in_thr != nullptr || out_current_max != nullptr, "fq_conv3x3_i8: online mode needs out_current_max");
  FQ_REQUIRE(in_width >= 2 && in_width <= 8, "fq_conv3x3_i8: input width %d does not fit int8 codes", in_width);
  FQ_REQUIRE(!(in_flags & (FQ_ACT_NO_ABS | FQ_ACT_NO_EPS)), "fq_conv3x3_i8: unsupported activation flags");
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_conv3x3_i8: bn_scale and bn_shift go together");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_conv3x3_i8: unknown activation %d", act);
  FQ_REQUIRE(aligned16(wcodes) && aligned16(x), "fq_conv3x3_i8: x and wcodes must be 16-byte aligned");
  const int64_t hw = h * w, cols = n * hw;
  FQ_REQUIRE((64 * 32 / hw + 3) * cout * hw * 4 < (1ll << 31) && (64 * 32 / hw + 3) * cin * hw * 4 < (1ll << 31),
             "fq_conv3x3_i8: a pixel block must stay within 2 GiB of its first sample");
  hipStream_t st = (hipStream_t)stream;
  const int64_t row_pad = (9 * cin + 63) / 64 * 64, rows_pad = (cout + 63) / 64 * 64;
  FQ_REQUIRE(row_pad == 9 * cin, "fq_conv3x3_i8: 9 * Cin must be a multiple of 64");
  const int8_t* wfrag = wcodes + rows_pad * row_pad;                      // fragment-major copy (fq_weight_codes)
  // wavefront arrangement: four channel tiles per workgroup when the layer has them, else two and two pixel halves
  // eight wavefronts (256 channels per workgroup) for wide layers on small planes: half the channel groups
  static const int nw_tune = env_int("FQ_C3_NW", 0);
  // (measured in the ResNet-50 step: 256 @14x14 32.8 -> 28.6 us; 512 @7x7, where only 196 workgroups would remain, 34.1 -> 35.1)
  const bool nw8_fills = ((cols + 63) / 64) * ((cout + 255) / 256) >= (int64_t)num_cu();
  int nw = (nw_tune == 4 || nw_tune == 8) ? nw_tune : ((cout >= 256 && (kt == 8 || kt == 16) && nw8_fills) ? 8 : 4);
  // (the C16 forms are built for four wavefronts; round 6: the sliced form and the codes-in-AND-codes-out form for eight on the
  // 256-channel 14x14 layers)
  const bool c16_both8 = c16 && in_c16 && out_thr != nullptr && kt == 8 && nsl == 1;
  if (cout < 256 || !(kt == 8 || kt == 16) || (nsl != 1 && kt != 8) || (c16 && !c16_both8)) nw = 4;
  if (nsl != 1) FQ_REQUIRE(cout % 32 == 0, "fq_conv3x3_i8_sliced: Cout must be a multiple of 32, got %lld", (long long)cout);
  const int wc = nw == 8 ? 8 : (cout >= 128 ? 4 : 2);
  const int wp = nw / wc;
  const int64_t cs = (cout + 32 * wc - 1) / (32 * wc);
  // two pixel tiles per wavefront: measured best (or equal) on all four ResNet-50 stages in the model - 63 / 39 / 33 / 34 us
  // at 56x56 / 28x28 / 14x14 / 7x7 against 67 / 45 / 34 / 46 with four and 78 / 44 / 39 / 41 with one
  int ptw = 2;
  while (ptw > 1 && ((cols + 32 * ptw * wp - 1) / (32 * ptw * wp)) * cs < (int64_t)num_cu()) ptw >>= 1;
  const int tune = env_int("FQ_C3_PTW", 0);
  if (tune == 1 || tune == 2 || (tune == 4 && !c16 && nsl == 1)) ptw = tune;
  const int pt = ptw * wp;
  C3Geom g;
  g.Cin = (int)cin; g.Cout = (int)cout; g.H = (int)h; g.W = (int)w; g.HW = (int)hw;
  g.CS = (int)cs; g.CTM = (int)(rows_pad / 32);
  g.RP = 32 * pt + 2 * (int)w + 2; g.RT = (g.RP + 31) / 32; g.ROW = (int)cin + 16;
  g.cols = cols; g.items = ((cols + 32 * pt - 1) / (32 * pt)) * cs; g.zoff = (in_flags & FQ_ACT_SIGNED) ? 0 : 128;
  g.slice_bytes = 2 * rows_pad * row_pad; g.slice_rows = (int)cout;
  FQ_REQUIRE(nsl == 1 || 3 * g.slice_bytes < (1ll << 31), "fq_conv3x3_i8_sliced: the three weight slices must lie within 2 GiB");
  g.CBi = (int)((cin + 15) / 16); g.CBo = (int)((cout + 15) / 16);
  g.out_levels = 0.0f; g.out_lo_neg = 0; g.out_zoff = 0;
  if (out_thr != nullptr) {
    FQ_REQUIRE(out_width >= 2 && out_width <= 8, "fq_conv3x3_i8_c16: output width %d does not fit int8 codes", out_width);
    FQ_REQUIRE(!(out_flags & (FQ_ACT_NO_ABS | FQ_ACT_NO_EPS)), "fq_conv3x3_i8_c16: unsupported output flags");
    g.out_levels = act_levels(out_width, out_flags);
    g.out_lo_neg = (out_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
    g.out_zoff = (out_flags & FQ_ACT_SIGNED) ? 0 : 128;
  }
  const int64_t grid = (g.items + 7) / 8 * 8;
  FQ_REQUIRE(grid < (1ll << 31), "fq_conv3x3_i8: too many pixel blocks");
  const size_t lds = (size_t)g.RT * 32 * g.ROW + (size_t)(32 * wc) * (4 + nsl) * sizeof(float);
  FQ_REQUIRE(lds <= 150 * 1024, "fq_conv3x3_i8: the region of %d pixels x %d channels does not fit LDS", g.RP, (int)cin);
  const float levels = act_levels(in_width, in_flags);
  const int lo_neg = range ? kRangeMode : ((in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0);
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  ProfScope prof(FQ_KERNEL_CONV3X3, 4.0 * ((double)n * cin * hw + (double)n * cout * hw), st,
                 (in_c16 ? (double)n * g.CBi * 16.0 * hw : 4.0 * (double)n * cin * hw) +
                     (out_thr != nullptr ? (double)n * g.CBo * 16.0 * hw : 4.0 * (double)n * cout * hw));
  bool launched = false;
  if (c16) {
    if (int rc = fqi::conv3x3_c16_launch(x, wfrag, wscale, wsum, bias, y, &g, kt, ptw, wc, grid, lds, st, in_stat, (int)n,
                                         in_thr, levels, lo_neg, out_current_max, bn_scale, bn_shift, act, stat_out, in_c16,
                                         out_thr, &launched))
      return rc;
    FQ_REQUIRE(launched, "fq_conv3x3_i8_c16: no instantiation for K/32=%d, %d pixel tiles per wavefront, %d channel tiles "
               "per workgroup", kt, ptw, wc);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
  }
#define FQ_C3_CASE_NS(KT_, PTW_, WC_, D_, LB_, NW_, NSL_)                                                               \
  if (kt == KT_ && ptw == PTW_ && wc == WC_ && nw == NW_ && nsl == NSL_) {                                             \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_i8_kernel<KT_, PTW_, WC_, D_, LB_, NW_, NSL_>),     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;                     \
    FQ_REQUIRE(attr_ok, "fq_conv3x3_i8: cannot raise the dynamic LDS limit");                                          \
    hipLaunchKernelGGL((conv3x3_i8_kernel<KT_, PTW_, WC_, D_, LB_, NW_, NSL_>), dim3((unsigned)grid), dim3(NW_ * 64), lds, st, \
                       x,                                                                                              \
                       wfrag, wscale, (const int*)wsum, bias, y, g, in_stat, (int)n, in_thr, levels, lo_neg, kEps,     \
                       out_current_max, bn_scale, bn_shift, act, stat_out, (const float*)nullptr);                     \
    launched = true;                                                                                                   \
  }
#define FQ_C3_CASE_NW(KT_, PTW_, WC_, D_, LB_, NW_) FQ_C3_CASE_NS(KT_, PTW_, WC_, D_, LB_, NW_, 1)
#define FQ_C3_CASE(KT_, PTW_, WC_, D_, LB_) FQ_C3_CASE_NW(KT_, PTW_, WC_, D_, LB_, 4)
  // three weight slices (Winograd-domain quantised filters): one or two pixel tiles per wavefront, two wavefronts per SIMD
#define FQ_C3_SLICED(KT_)                                                                                              \
  FQ_C3_CASE_NS(KT_, 1, 4, 3, 2, 4, 3) FQ_C3_CASE_NS(KT_, 2, 4, 2, 2, 4, 3)                                            \
  FQ_C3_CASE_NS(KT_, 1, 2, 3, 2, 4, 3) FQ_C3_CASE_NS(KT_, 2, 2, 2, 2, 4, 3)
#define FQ_C3_KT(KT_)                                                                                                  \
  FQ_C3_CASE(KT_, 1, 4, 6, 4) FQ_C3_CASE(KT_, 2, 4, 4, 4) FQ_C3_CASE(KT_, 4, 4, 3, 3)                                  \
  FQ_C3_CASE(KT_, 1, 2, 6, 4) FQ_C3_CASE(KT_, 2, 2, 4, 4) FQ_C3_CASE(KT_, 4, 2, 3, 3)
  FQ_C3_KT(2) FQ_C3_KT(4) FQ_C3_KT(8) FQ_C3_KT(16)
  FQ_C3_CASE_NW(8, 1, 8, 6, 4, 8) FQ_C3_CASE_NW(8, 2, 8, 4, 4, 8) FQ_C3_CASE_NW(16, 1, 8, 6, 4, 8) FQ_C3_CASE_NW(16, 2, 8, 4, 4, 8)
  FQ_C3_SLICED(2) FQ_C3_SLICED(4) FQ_C3_SLICED(8) FQ_C3_SLICED(16)
  FQ_C3_CASE_NS(8, 1, 8, 3, 2, 8, 3) FQ_C3_CASE_NS(8, 2, 8, 2, 2, 8, 3)
#undef FQ_C3_SLICED
#undef FQ_C3_KT
#undef FQ_C3_CASE
#undef FQ_C3_CASE_NW
#undef FQ_C3_CASE_NS
  FQ_REQUIRE(launched, "fq_conv3x3_i8: no instantiation for K/32=%d, %d pixel tiles per wavefront, %d channel tiles per "
             "workgroup", kt, ptw, wc);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // namespace

namespace fqi {
// dense 3x3 (stride 1, padding 1) of nn.Conv2D(quantized=True): quantise on load with the range record, int32 bias codes
bool conv3x3_range_shape_ok(int64_t cin, int64_t cout) {
  return (cin == 64 || cin == 128 || cin == 256 || cin == 512) && cout >= 32;
}
int conv3x3_range_call(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const int32_t* ibias,
                       float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, const float* rec,
                       const float* bn_scale, const float* bn_shift, int act, float* stat_out, hipStream_t st) {
  return conv3x3_launch(x, wcodes, wscale, wsum, reinterpret_cast<const float*>(ibias), y, n, cin, cout, h, w, nullptr, rec,
                        8, 0, nullptr, bn_scale, bn_shift, act, stat_out, (fqStream_t)st, 1, false, nullptr, 8, 0, true);
}
}  // namespace fqi

extern "C" {

int fq_conv3x3_i8(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                  float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, const float* in_stat,
                  const float* in_thr, int in_width, unsigned in_flags, float* out_current_max, const float* bn_scale,
                  const float* bn_shift, int act, float* stat_out, fqStream_t stream) {
  return conv3x3_launch(x, wcodes, wscale, wsum, bias, y, n, cin, cout, h, w, in_stat, in_thr, in_width, in_flags,
                        out_current_max, bn_scale, bn_shift, act, stat_out, stream, 1);
}

int fq_conv3x3_i8_sliced(const float* x, const int8_t* wslices, const float* pscale, const int32_t* wsum, const float* bias,
                         float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, const float* in_stat,
                         const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                         const float* bn_scale, const float* bn_shift, int act, float* stat_out, fqStream_t stream) {
  return conv3x3_launch(x, wslices, pscale, wsum, bias, y, n, cin, cout, h, w, in_stat, in_thr, in_width, in_flags,
                        out_current_max, bn_scale, bn_shift, act, stat_out, stream, 3);
}

int fq_conv3x3_i8_c16(const void* x, int x_is_c16, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                      const float* bias, void* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w,
                      const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                      const float* bn_scale, const float* bn_shift, int act, float* stat_out, const float* out_thr,
                      int out_width, unsigned out_flags, fqStream_t stream) {
  FQ_REQUIRE(x_is_c16 || out_thr != nullptr, "fq_conv3x3_i8_c16: neither side is a C16 tensor (use fq_conv3x3_i8)");
  return conv3x3_launch((const float*)x, wcodes, wscale, wsum, bias, (float*)y, n, cin, cout, h, w, in_stat, in_thr, in_width,
                        in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, stream, 1, x_is_c16 != 0, out_thr,
                        out_width, out_flags);
}

int fq_weight_slices(const float* w, int64_t rows, int64_t row_len, int64_t row_pad, int64_t rows_pad, int8_t* codes,
                     float* pscale, int32_t* rowsum, void* ws, fqStream_t stream) {
  FQ_REQUIRE(w && codes && pscale && rowsum && ws, "fq_weight_slices: null pointer");
  FQ_REQUIRE(rows > 0 && row_len > 0, "fq_weight_slices: bad shape (rows=%lld row_len=%lld)", (long long)rows,
             (long long)row_len);
  FQ_REQUIRE(row_pad >= row_len && rows_pad >= rows && rows_pad < (1ll << 31) && row_pad % 32 == 0 && rows_pad % 32 == 0,
             "fq_weight_slices: bad padding (row_pad and rows_pad must be multiples of 32)");
  hipStream_t st = (hipStream_t)stream;
  float* rmax = (float*)ws;
  FQ_HIP(hipMemsetAsync(rmax, 0, rows * sizeof(float), st));
  if (int rc = launch_absmax(w, rows, row_len, true, rmax, st)) return rc;
  hipLaunchKernelGGL(weight_slices_kernel, dim3((unsigned)rows_pad), dim3(kBlock), 0, st, w, (int)rows, (int)row_len,
                     (int)row_pad, (int)rows_pad, rmax, codes, pscale, (int*)rowsum);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // extern "C"
