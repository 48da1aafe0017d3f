// libfakequant — K2s depthwise 3x3 convolution between two C16 code tensors (fq_dwconv3x3_c16): the depthwise layer of a
// MobileNetV2 unit under OFFLINE input quantisation, whose input the expansion convolution wrote as integer codes and whose
// output the projection convolution reads as integer codes (see fq_common.h for the list of translation units and the design
// rules; the C16 layout: include/fakequant.h at fq_pwconv_i8_c16)
#include "fq_common.h"

// tuning build: -DFQ_DW16_V=<bits>  4: the short quantiser for non-negative output ranges (bits 1 and 2 - unconditional loads
// from clamped addresses + select instead of 62 exec-masked loads, stride 2 keeping the row it shares with the next output
// row - are built in since r4: 526 -> 473 us over MobileNetV2's ten depthwise shapes at batch 128, tools/dw16bench.py).
// -DFQ_DW16_DEPTH=<0|2>: output rows between the fetch of an input row and its use (1: 452 us, 3: four sets, 136 registers, 440 us).
#ifndef FQ_DW16_V
#define FQ_DW16_V 7
#endif
#ifndef FQ_DW16_DEPTH
#define FQ_DW16_DEPTH 2
#endif

namespace {

using namespace fqi;

// The arithmetic is that of fq_dwconv3x3 under an offline threshold, bit for bit: x^ = code * sx (one fp32 rounding, the
// value LinearQuantizeSTE returns), acc = fmaf chain over (ky, kx) in row-major order from 0, + bias, * bn_scale + bn_shift,
// activation, per-sample max of the fp32 result, then the CONSUMER's quantiser.  What differs is the traffic: 1 byte per
// element in and out instead of 4, and 16-byte-per-pixel rows instead of channel planes.
//
// Mapping: a lane owns one output column and a QUARTER of a 16-channel block (4 channels = one dword of a pixel's 16 bytes);
// the 64 lanes of a wavefront are 16 columns x 4 quarters, so a row access of a wavefront is 256 contiguous bytes.  A
// workgroup (4 wavefronts = 64 column slots) walks down the output rows of its (sample, block(s)) with a sliding window of
// three dequantised input rows (3 columns x 4 channels each); planes narrower than 33 columns put several blocks side by
// side in the 64 slots.  Out-of-image taps take the byte pattern of code 0 before they are dequantised.
struct Dw16Geom {
  int C, CB, H, W, Ho, Wo;
  int T;                     // blocks side by side in a workgroup's 64 column slots (W <= 32), else 1
  int col_tiles;             // ceil(Wo / 64) when T == 1
  int groups;                // ceil(CB / T)
  float out_levels;
  int out_lo_neg, out_zoff, in_zoff;
};

template <int S, bool SIGNED_IN, int EPI>
__global__ __launch_bounds__(256, 4) void dwconv3x3_c16_kernel(
    const int8_t* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias, int8_t* __restrict__ y,
    Dw16Geom g, const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max,
    float eps, float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
    int act, float* __restrict__ stat_out, const float* __restrict__ out_thr) {
  __shared__ float red[4];
  const unsigned slot = threadIdx.x >> 2, qd = threadIdx.x & 3u;        // column slot 0..63, quarter of the block
  // workgroup -> (sample, block group, column tile)
  unsigned b = blockIdx.x;
  const unsigned ct = b % (unsigned)g.col_tiles;
  b /= (unsigned)g.col_tiles;
  const unsigned grp = b % (unsigned)g.groups, smp = b / (unsigned)g.groups;
  unsigned blk, xo;                                                     // this lane's block and output column
  bool lane_ok;
  if (g.T > 1) {
    const unsigned bl = slot / (unsigned)g.Wo;
    xo = slot - bl * (unsigned)g.Wo;
    blk = grp * (unsigned)g.T + bl;
    lane_ok = bl < (unsigned)g.T && blk < (unsigned)g.CB;
  } else {
    blk = grp;
    xo = ct * 64u + slot;
    lane_ok = xo < (unsigned)g.Wo;
  }
  // the batch statistic only feeds current_input_max here (the reference computes it in every mode, convert_conv2d.py:56)
  const float max_ = input_threshold(in_stat, n, in_thr, cur_max_out, blockIdx.x == 0);
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  const QParams q2 = make_qparams(out_thr[0], g.out_levels, g.out_lo_neg != 0, eps);
  const float sx = q.scale;
  const int ubias2 = 128 - g.out_zoff;
  const unsigned ch = (lane_ok ? blk : 0u) * 16u + qd * 4u;             // first of this lane's four channels
  float wt[9][4], bch[4], bsc[4], bsh[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bool cok = lane_ok && ch + c < (unsigned)g.C;                 // channels past C: zero weights, code 0 out
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t][c] = cok ? wgt[(ch + c) * 9 + t] : 0.0f;
    bch[c] = cok && bias != nullptr ? bias[ch + c] : 0.0f;
    bsc[c] = cok ? (bn_scale != nullptr ? bn_scale[ch + c] : 1.0f) : 0.0f;
    bsh[c] = cok && bn_shift != nullptr ? bn_shift[ch + c] : 0.0f;
  }
  const unsigned* xin = reinterpret_cast<const unsigned*>(x) + (((size_t)smp * g.CB + (lane_ok ? blk : 0u)) * g.H * g.W) * 4u + qd;
  unsigned* yout = reinterpret_cast<unsigned*>(y) + (((size_t)smp * g.CB + (lane_ok ? blk : 0u)) * g.Ho * g.Wo) * 4u + qd;
  const int xc = (int)xo * S;                                           // centre input column
  // one input row -> 3 columns x 4 dequantised channels, in two steps: `fetch` requests the three dwords (every load
  // unconditional, from an address clamped into the plane: a load under a lane condition is an exec-masked branch of its own
  // with a wait behind it), `cook` selects the padding pattern and dequantises.  The rows are fetched FQ_DW16_DEPTH output rows
  // ahead of their use through three sets of registers in rotation (no copies: a copy needs the loaded value): with the fetch
  // in the iteration that uses it, a lane's walk down the plane was a chain of Ho load latencies - 14 x 14 and 7 x 7 planes,
  // 19 MB per layer, took 18-22 us.
  struct Row { float v[3][4]; };
  struct Raw { unsigned d[3]; };
  auto fetch = [&](int r, Raw& w) __attribute__((always_inline)) {
    const int rr = r < 0 ? 0 : (r < g.H ? r : g.H - 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int col = xc + k - 1;
      const int cc = col < 0 ? 0 : (col < g.W ? col : g.W - 1);
      w.d[k] = xin[((size_t)rr * g.W + cc) * 4u];
    }
  };
  auto cook = [&](int r, const Raw& w, Row& row) __attribute__((always_inline)) {
    const bool rok = lane_ok && r >= 0 && r < g.H;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int col = xc + k - 1;
      // the stored byte of an unsigned code is code ^ 0x80: one xor per dword, then the byte-to-float conversions
      const unsigned d = (rok && col >= 0 && col < g.W) ? (SIGNED_IN ? w.d[k] : w.d[k] ^ 0x80808080u) : 0u;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned byte = (d >> (8 * c)) & 255u;
        const float code = SIGNED_IN ? (float)(int)(int8_t)byte : (float)byte;
        row.v[k][c] = code * sx;
      }
    }
  };
  float m = 0.0f;
  const unsigned nn_xor2 = fq_nonneg_xor(ubias2);
  // the walk down the plane, instantiated with the 5-instruction quantiser of non-negative output ranges and with the generic
  // one (a run-time choice between the two inside the loop computes BOTH for every output and selects)
  auto walk = [&](auto nn_c) __attribute__((always_inline)) {
    constexpr bool NN = decltype(nn_c)::value;
    Row ra, rb, rc;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) ra.v[k][c] = rb.v[k][c] = rc.v[k][c] = 0.0f;      // (row -1: code 0 dequantises to 0)
    // output row r from the window (A, B, C) = input rows above / at / below
    auto emit = [&](int r, const Row& A, const Row& B, const Row& C) __attribute__((always_inline)) {
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc = fmaf(wt[k][c], A.v[k][c], acc);
#pragma unroll
        for (int k = 0; k < 3; ++k) acc = fmaf(wt[3 + k][c], B.v[k][c], acc);
#pragma unroll
        for (int k = 0; k < 3; ++k) acc = fmaf(wt[6 + k][c], C.v[k][c], acc);
        acc = dw_finish<EPI>(acc, bias != nullptr, bch[c], bn_scale != nullptr, bsc[c], bsh[c], act);
        v[c] = acc;
        m = fmaxf(m, fabsf(acc));
      }
      if (lane_ok) {
        const int packed = fq_pack4<NN>(v[0], v[1], v[2], v[3], q2, ubias2, nn_xor2);
        yout[((size_t)r * g.Wo + xo) * 4u] = (unsigned)packed;
      }
    };
    // The walk is unrolled over the period of BOTH rotations - the fetched sets and the window's rows change roles instead of
    // being copied (24 register moves per output row otherwise, of ~140 instructions).
    constexpr bool AHEAD = FQ_DW16_DEPTH != 0;        // (0: fetch and use in the same step - the tuning baseline)
    int r = 0;
    if (S == 1) {
      // output row r needs input rows r - 1, r, r + 1: rows (A, B) carried, row r + 1 cooked into C from the set fetched two
      // output rows ago
      Raw w0, w1, w2;
      fetch(0, w0);
      cook(0, w0, rb);
      if (AHEAD) {
        fetch(1, w0);
        fetch(2, w1);
      }
      auto step = [&](Raw& mine, Raw& far, const Row& A, const Row& B, Row& C) __attribute__((always_inline)) {
        if (AHEAD) fetch(r + 3, far);
        else fetch(r + 1, mine);
        cook(r + 1, mine, C);
        emit(r, A, B, C);
        ++r;
      };
      while (r < g.Ho) {
        step(w0, w2, ra, rb, rc);
        if (r >= g.Ho) break;
        step(w1, w0, rb, rc, ra);
        if (r >= g.Ho) break;
        step(w2, w1, rc, ra, rb);
      }
    } else {
      // output row r needs input rows 2r - 1 (the previous output row's 2r' + 1: its C is this row's A), 2r, 2r + 1
      // (fetched ONE output row = two input rows ahead: two more sets cost the fourth wavefront per SIMD and bought nothing)
      Raw a0, b0, a1, b1;
      if (AHEAD) {
        fetch(0, a0);
        fetch(1, b0);
      }
      auto step = [&](Raw& ma, Raw& mb, Raw& fa, Raw& fb, const Row& A, Row& B, Row& C) __attribute__((always_inline)) {
        if (AHEAD) { fetch(2 * r + 2, fa); fetch(2 * r + 3, fb); }
        else { fetch(2 * r, ma); fetch(2 * r + 1, mb); }
        cook(2 * r, ma, B);
        cook(2 * r + 1, mb, C);
        emit(r, A, B, C);
        ++r;
      };
      while (r < g.Ho) {
        step(a0, b0, a1, b1, ra, rb, rc);
        if (r >= g.Ho) break;
        step(a1, b1, a0, b0, rc, rb, ra);
      }
    }
  };
  if ((FQ_DW16_V & 4) && fq_nonneg(q2)) walk(std::true_type{});
  else walk(std::false_type{});
  if (stat_out != nullptr) {
    const float wm = wave_max_nonneg(lane_ok ? m : 0.0f);
    if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float t = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      if (__float_as_uint(t) != 0u) atomic_max_f32(stat_out + smp, t);
    }
  }
}

}  // namespace

extern "C" {

int fq_dwconv3x3_c16(const void* x, const float* w, const float* bias, void* y, int64_t n, int64_t c, int64_t h, int64_t wdt,
                     int stride, const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                     float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                     const float* out_thr, int out_width, unsigned out_flags, fqStream_t stream) {
  FQ_REQUIRE(x && w && y && in_thr && out_thr, "fq_dwconv3x3_c16: null pointer (x, w, y, in_thr and out_thr are required)");
  FQ_REQUIRE(n > 0 && c > 0 && h > 0 && wdt > 0 && n * ((c + 15) / 16) * h * wdt * 16 < (1ll << 40),
             "fq_dwconv3x3_c16: bad shape (n=%lld c=%lld h=%lld w=%lld)", (long long)n, (long long)c, (long long)h, (long long)wdt);
  FQ_REQUIRE(stride == 1 || stride == 2, "fq_dwconv3x3_c16: stride must be 1 or 2, got %d", stride);
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_dwconv3x3_c16: bn_scale and bn_shift go together");
  FQ_REQUIRE(in_width >= 2 && in_width <= 8 && out_width >= 2 && out_width <= 8, "fq_dwconv3x3_c16: widths must fit int8 codes");
  FQ_REQUIRE(!((in_flags | out_flags) & (FQ_ACT_NO_ABS | FQ_ACT_NO_EPS)), "fq_dwconv3x3_c16: unsupported activation flags");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_dwconv3x3_c16: unknown activation %d", act);
  hipStream_t st = (hipStream_t)stream;
  Dw16Geom g;
  g.C = (int)c; g.CB = (int)((c + 15) / 16); g.H = (int)h; g.W = (int)wdt;
  g.Ho = (int)((h - 1) / stride + 1); g.Wo = (int)((wdt - 1) / stride + 1);
  g.T = g.Wo <= 32 ? 64 / g.Wo : 1;
  g.col_tiles = g.T > 1 ? 1 : (g.Wo + 63) / 64;
  g.groups = (g.CB + g.T - 1) / g.T;
  g.out_levels = act_levels(out_width, out_flags);
  g.out_lo_neg = (out_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
  g.out_zoff = (out_flags & FQ_ACT_SIGNED) ? 0 : 128;
  g.in_zoff = (in_flags & FQ_ACT_SIGNED) ? 0 : 128;
  const int64_t grid = n * g.groups * g.col_tiles;
  FQ_REQUIRE(grid < (1ll << 31), "fq_dwconv3x3_c16: too many workgroups");
  const float levels = act_levels(in_width, in_flags);
  const int lo_neg = (in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  // (SURVEY.md 8d's definition of the algorithmic bytes - 4 B per input and per output element - as for the fp32 form)
  // (moved: both sides are C16 code tensors - 1 B per element, channels padded to blocks of 16)
  ProfScope prof(FQ_KERNEL_DWCONV, 4.0 * ((double)n * c * h * wdt + (double)n * c * g.Ho * g.Wo), st,
                 (double)n * g.CB * 16.0 * ((double)h * wdt + (double)g.Ho * g.Wo));
#define FQ_DW16(S_, SG_, E_)                                                                                            \
  hipLaunchKernelGGL((dwconv3x3_c16_kernel<S_, SG_, E_>), dim3((unsigned)grid), dim3(256), 0, st, (const int8_t*)x, w, bias, \
                     (int8_t*)y, g, in_stat, (int)n, in_thr, levels, lo_neg, kEps, out_current_max, bn_scale, bn_shift,  \
                     act, stat_out, out_thr)
  const bool sg = (in_flags & FQ_ACT_SIGNED) != 0;
  // compile-time epilogues for the fused-inference case (BatchNorm, no bias, ReLU / ReLU6) on unsigned input codes
  const int epi = (!sg && bn_scale != nullptr && bias == nullptr)
                      ? (act == FQ_ACT_RELU ? kEpiBnRelu : act == FQ_ACT_RELU6 ? kEpiBnRelu6 : kEpiRuntime)
                      : kEpiRuntime;
  if (stride == 1) {
    if (sg) FQ_DW16(1, true, kEpiRuntime);
    else if (epi == kEpiBnRelu) FQ_DW16(1, false, kEpiBnRelu);
    else if (epi == kEpiBnRelu6) FQ_DW16(1, false, kEpiBnRelu6);
    else FQ_DW16(1, false, kEpiRuntime);
  } else {
    if (sg) FQ_DW16(2, true, kEpiRuntime);
    else if (epi == kEpiBnRelu) FQ_DW16(2, false, kEpiBnRelu);
    else if (epi == kEpiBnRelu6) FQ_DW16(2, false, kEpiBnRelu6);
    else FQ_DW16(2, false, kEpiRuntime);
  }
#undef FQ_DW16
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // extern "C"
