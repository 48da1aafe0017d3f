// libfakequant — K2s depthwise 3x3 convolution between two C16 code tensors (fq_dwconv3x3_c16): the depthwise layer of a
// MobileNetV2 unit under OFFLINE input quantisation, whose input the expansion convolution wrote as integer codes and whose
// output the projection convolution reads as integer codes (see fq_common.h for the list of translation units and the design
// rules; the C16 layout: include/fakequant.h at fq_pwconv_i8_c16)
#include "fq_common.h"

// tuning builds: -DFQ_DW16_V=3 (the general output quantiser everywhere; bit 4 = the short one for values that cannot be
// negative), -DFQ_DW16_RING=<3|4|5> / -DFQ_DW16_RING2=<2|3>: input rows (stride 2: row pairs) in flight per lane.
#ifndef FQ_DW16_V
#define FQ_DW16_V 7
#endif
// input rows in flight per lane, stride 1 / row PAIRS, stride 2: ordinary buffer loads the compiler waits for.  (Issued and
// awaited by hand - inline assembly and hand-counted s_waitcnt, because hipcc drains the ring where the paths into the
// unrolled walk meet - six rows were another +0.4 % images/s, but the compiler does not know that such registers are pending:
// it spilled and copied them, and full-size repeats beside a competing stream differed - tests/test_gpu_determinism.py.)
#ifndef FQ_DW16_RING
#define FQ_DW16_RING 4
#endif
#ifndef FQ_DW16_RING2                  // (3 pairs: 124-128 registers, -0.2 % images/s)
#define FQ_DW16_RING2 2
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
// workgroup (4 wavefronts = 64 column slots) walks down the output rows of 64 consecutive columns of its sample (columns
// of one block, then of the next): every input row is dequantised once (3 columns x 4 channels) and added to the sums of
// the three output rows it belongs to.  Out-of-image taps take the byte pattern of code 0 before they are dequantised.
struct Dw16Geom {
  int C, CB, H, W, Ho, Wo;
  int wgs_per_sample;        // ceil(CB * Wo / 64): a workgroup's 64 column slots are consecutive columns of the sample
  int span;                  // blocks such a run of 64 columns can touch
  float out_levels;
  int out_lo_neg, out_zoff, in_zoff;
};

// Two fp32 values in a register pair: gfx950 multiplies / adds / fuses both in ONE instruction (v_pk_mul_f32, v_pk_add_f32,
// v_pk_fma_f32: each component an IEEE operation of its own, so the results are those of the scalar form bit for bit).  The
// 36 multiply-adds of a lane's four channels are 18 packed ones, BatchNorm one packed multiply and one packed add per pair
// (~135 -> ~95 instructions per output row and lane).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// What bounds this kernel (round 5; rocprofv3 counters, ablation builds and probes on 144 channels x 56 x 56, batch 128 -
// profiles/r5_dw16_study.txt): the VECTOR UNIT, and among several batches in flight the number of instructions is what
// counts.  At 1 B per element HBM is idle (116 MB in 59 us); a build without the walk's loads and stores takes the same time;
// planes of 28 instead of 56 rows take 0.84 us less per row.  A row-step of a lane (4 outputs) is ~95 instructions: 12
// byte-to-float conversions + 6 packed multiplies (dequantise 3 columns), 18 packed multiply-adds, BatchNorm / ReLU6 /
// statistic (10), the consumer's quantiser (clip, fp64 divide in 3, round: 20), packing (4).  tools/pk_probe.hip prices
// them: v_fma / v_mul / v_add / v_bitop3 issue every ~2.9 cycles per wavefront and SIMD (four wavefronts resident);
// conversions, v_med3 / v_max3, every fp64 and every packed fp32 instruction every ~4.6-5.0 - a packed multiply-add is 0.83
// of two scalar ones, not half.  MobileNetV2 W4 offline, four batches in flight, went 157.5 -> 165.0 k images/s (+4.8 %,
// together with the five-instruction output quantiser of the pointwise kernels) through: buffer addressing (no 64-bit
// address arithmetic in the vector unit), accumulators instead of a window of cooked rows (below) - which freed the
// registers for - a ring of 4 instead of 2 rows in flight per lane, requested before the weights and thresholds (3 rows:
// -1.7 %).  What did NOT help: cutting a plane into chunks of rows for more, shorter workgroups (-3.4 %: every chunk repeats
// the 44 weight loads and two halo rows); weights through bounded buffer loads instead of 44 loads under a lane condition
// (-0.5 %).
template <int S, bool SIGNED_IN, int EPI>
__global__ __launch_bounds__(256, 4) void dwconv3x3_c16_kernel(
    const int8_t* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias, int8_t* __restrict__ y,
    Dw16Geom g, const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max,
    float eps, float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
    int act, float* __restrict__ stat_out, const float* __restrict__ out_thr) {
  __shared__ float red[4];
  const unsigned slot = threadIdx.x >> 2, qd = threadIdx.x & 3u;        // column slot 0..63, quarter of the block
  // workgroup -> (sample, 64 consecutive output columns of the sample's CB x Wo columns): a lane's column may belong to the
  // next block than its neighbour's, so 56- and 28-wide planes fill all 64 slots (one block - or a whole number of them -
  // per workgroup left an eighth of the lanes idle there, and on 112-wide planes whose second column tile is 48 wide)
  const unsigned wgs = (unsigned)g.wgs_per_sample;
  const unsigned grp = blockIdx.x % wgs, smp = blockIdx.x / wgs;
  const unsigned cols_total = (unsigned)(g.CB * g.Wo);
  const unsigned gcol = grp * 64u + slot;
  const bool lane_ok = gcol < cols_total;
  const unsigned blk = (lane_ok ? gcol : cols_total - 1u) / (unsigned)g.Wo;         // this lane's block ...
  const unsigned xo = (lane_ok ? gcol : cols_total - 1u) - blk * (unsigned)g.Wo;    // ... and output column
  const unsigned blk0 = (grp * 64u) / (unsigned)g.Wo;                   // first block of the workgroup (wave-uniform)
  const unsigned bl = blk - blk0;
  const int nrows = g.Ho;
  // Buffer addressing: the resources start at the workgroup's first block of its sample and span the blocks it owns, a lane's
  // three column offsets (clamped into the plane) and its output offset stay in registers for the whole walk and a row is a
  // SCALAR offset - no address arithmetic in the vector unit (flat 64-bit addresses cost ~17 of the ~135 instructions per
  // output row).  A lane without an output gets an offset past the resource: the hardware drops its stores.
  const unsigned span = (unsigned)g.span;                               // blocks 64 consecutive columns can touch
  const unsigned nblk = (unsigned)g.CB - blk0 < span ? (unsigned)g.CB - blk0 : span;
  const unsigned plane_in = (unsigned)(g.H * g.W) * 16u, plane_out = (unsigned)(g.Ho * g.Wo) * 16u;
  const fq_rsrc xr = make_rsrc(x + ((size_t)smp * g.CB + blk0) * plane_in, (int64_t)nblk * plane_in);
  const fq_rsrc yr = make_rsrc(y + ((size_t)smp * g.CB + blk0) * plane_out, (int64_t)nblk * plane_out);
  const int xc = (int)xo * S;                                           // centre input column
  unsigned xoff[3], cmask[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int col = xc + k - 1;
    const int cc = col < 0 ? 0 : (col < g.W ? col : g.W - 1);
    xoff[k] = lane_ok ? bl * plane_in + (unsigned)cc * 16u + qd * 4u : 0u;
    cmask[k] = (lane_ok && col >= 0 && col < g.W) ? 0xFFFFFFFFu : 0u;  // out-of-image taps: the byte pattern of code 0
  }
  const unsigned yoff = lane_ok ? bl * plane_out + xo * 16u + qd * 4u : 0x80000000u;
  const unsigned row_in = (unsigned)g.W * 16u, row_out = (unsigned)g.Wo * 16u;
  // one input row -> 3 columns x 4 dequantised channels, in two steps: `fetch` requests the three dwords (every load
  // unconditional, from a row clamped into the plane: a load under a lane condition is an exec-masked branch of its
  // own with a wait behind it; a repeated row comes from the L1), `cook` dequantises - a row outside the image is all zeros
  // (code 0), chosen by a SCALAR branch.
  struct Row { f2 v[3][2]; };
  struct Raw { unsigned d[3]; };
  auto fetch = [&](int r, Raw& w) __attribute__((always_inline)) {
    const int rr = r < 0 ? 0 : (r < g.H ? r : g.H - 1);
    const unsigned so = (unsigned)rr * row_in;
#pragma unroll
    for (int k = 0; k < 3; ++k) w.d[k] = __float_as_uint(buf_ld_f32(xr, xoff[k], so));
  };
  float sx = 0.0f;                                                      // (set once the threshold has arrived)
  auto cook = [&](int r, const Raw& w, Row& row) __attribute__((always_inline)) {
    if (r >= 0 && r < g.H) {
      const f2 sx2 = (f2){sx, sx};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        // the stored byte of an unsigned code is code ^ 0x80: one xor per dword, then the byte-to-float conversions
        const unsigned d = (SIGNED_IN ? w.d[k] : w.d[k] ^ 0x80808080u) & cmask[k];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f2 code;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const unsigned byte = (d >> (8 * (2 * j + e))) & 255u;
            code[e] = SIGNED_IN ? (float)(int)(int8_t)byte : (float)byte;
          }
          row.v[k][j] = code * sx2;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) row.v[k][0] = row.v[k][1] = (f2){0.0f, 0.0f};
    }
  };
  // ---- the ring: every row it holds is requested NOW, ahead of the weights and the thresholds ---------------------------------
  // S == 1: entry j <-> input row j - 1 (output row t cooks entry t + 2);  S == 2: row -1 (zeros), then pairs: entry t <-> input
  // rows 2t, 2t + 1
  // (the run-time epilogue keeps bias and activation selectors alive: a shorter ring there, or it spills)
  constexpr int NR = S == 1 ? (EPI == kEpiRuntime ? 3 : FQ_DW16_RING) : FQ_DW16_RING2;
  Raw ring[NR], ring_b[S == 1 ? 1 : NR], first;
  if (S == 1) {
#pragma unroll
    for (int j = 0; j < NR; ++j) fetch(j - 1, ring[j]);
  } else {
    fetch(-1, first);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      fetch(2 * j, ring[j]);
      fetch(2 * j + 1, ring_b[j]);
    }
  }
  // (weights and per-channel constants as 44 loads under a lane condition: bounded buffer loads - no branches, ~250
  // instructions fewer per wavefront - were measured and are NOT faster among batches in flight, -0.4 ... -0.6 % with the ring
  // they leave registers for, profiles/r5_dw16_study.txt)
  const unsigned ch = (lane_ok ? blk : 0u) * 16u + qd * 4u;             // first of this lane's four channels
  f2 wt[9][2], bsc[2], bsh[2];
  float bch[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bool cok = lane_ok && ch + c < (unsigned)g.C;                 // channels past C: zero weights, code 0 out
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t][c >> 1][c & 1] = cok ? wgt[(ch + c) * 9 + t] : 0.0f;
    bch[c] = cok && bias != nullptr ? bias[ch + c] : 0.0f;
    bsc[c >> 1][c & 1] = cok ? (bn_scale != nullptr ? bn_scale[ch + c] : 1.0f) : 0.0f;
    bsh[c >> 1][c & 1] = cok && bn_shift != nullptr ? bn_shift[ch + c] : 0.0f;
  }
  // the batch statistic only feeds current_input_max here (the reference computes it in every mode, convert_conv2d.py:56)
  const float max_ = input_threshold(in_stat, n, in_thr, cur_max_out, blockIdx.x == 0);
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  const QParams q2 = make_qparams(out_thr[0], g.out_levels, g.out_lo_neg != 0, eps);
  sx = q.scale;
  const int ubias2 = 128 - g.out_zoff;
  float m = 0.0f;
  const unsigned nn_xor2 = fq_nonneg_xor(ubias2);
  QParams qc = q2;                                                      // the clip range with the activation folded in (see emit)
  if (EPI != kEpiRuntime) {
    qc.lo = 0.0f;
    if (EPI == kEpiBnRelu6) qc.hi = fminf(q2.hi, 6.0f);
  }
  // the walk down the plane, instantiated with the 5-instruction quantiser of non-negative output ranges and with the generic
  // one (a run-time choice between the two inside the loop computes BOTH for every output and selects)
  auto walk = [&](auto nn_c) __attribute__((always_inline)) {
    constexpr bool NN = decltype(nn_c)::value;
    // An input row is cooked ONCE and spent at once: it adds the taps of kernel row `kr` to the accumulators of the (up to
    // three) output rows it belongs to - A-part (kr = 0) starts the row below from zero, B-part continues this row, C-part
    // completes the row above, which is then finished and stored.  Every accumulator still sees its nine taps in row-major
    // order from 0 (the order of the fp32 form), and what stays alive between steps is 4 sums per output row in progress
    // instead of three cooked rows of 12 values (round 5: the registers went to the ring).
    auto taps = [&](int kr, const Row& R, f2 (&a)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) a[j] = fma2(wt[3 * kr + k][j], R.v[k][j], a[j]);
    };
    auto emit = [&](int r, const f2 (&a)[2]) __attribute__((always_inline)) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f2 acc = a[j];
        if (EPI == kEpiRuntime) {
#pragma unroll
          for (int e = 0; e < 2; ++e)
            v[2 * j + e] = dw_finish<EPI>(acc[e], bias != nullptr, bch[2 * j + e], bn_scale != nullptr, bsc[j][e], bsh[j][e], act);
        } else {
          // ReLU / ReLU6 and the consumer's clip are ONE median: clip(relu6(a), lo <= 0, hi) == med3(a, 0, min(6, hi)) for
          // every a (NaN -> 0 on both sides) - `qc` below - and the statistic max_i relu6(a_i) == min(max(0, max_i a_i), 6) is
          // taken from the raw values and clamped once after the walk: a v_med3 less per output
          acc = acc * bsc[j];
          acc = acc + bsh[j];
          v[2 * j] = acc[0];
          v[2 * j + 1] = acc[1];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) m = EPI == kEpiRuntime ? fmaxf(m, fabsf(v[c])) : fmaxf(m, v[c]);
      const int packed = fq_pack4<NN>(v[0], v[1], v[2], v[3], qc, ubias2, nn_xor2);
      buf_st_f32(yr, yoff, (unsigned)r * row_out, __int_as_float(packed));
    };
    const f2 zero2 = (f2){0.0f, 0.0f};
    // The walk is unrolled over the period of BOTH rotations - the ring's entries and the accumulators change roles instead
    // of being copied: every index below is a constant once the inner loop is unrolled.
    Row row;
    if (S == 1) {
      // ring entry j = input row j - 1: A-part of output row j, B-part of j - 1, C-part of j - 2; the sums of output row t live
      // in acc[t % 3]
      constexpr int P = NR % 3 == 0 ? NR : 3 * NR;                      // steps until ring entry and accumulator roles repeat
      f2 acc[3][2];
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i][0] = acc[i][1] = zero2;
      cook(-1, ring[0], row);
      fetch(NR - 1, ring[0]);
      taps(0, row, acc[0]);
      cook(0, ring[1], row);
      fetch(NR, ring[1]);
      taps(1, row, acc[0]);
      if (1 < nrows) taps(0, row, acc[1]);
      for (int t0 = 0; t0 < nrows; t0 += P) {
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const int t = t0 + u;
          if (t >= nrows) break;
          cook(t + 1, ring[(u + 2) % NR], row);
          fetch(t + 1 + NR, ring[(u + 2) % NR]);
          taps(2, row, acc[u % 3]);
          emit(t, acc[u % 3]);
          if (t + 1 < nrows) taps(1, row, acc[(u + 1) % 3]);
          acc[(u + 2) % 3][0] = acc[(u + 2) % 3][1] = zero2;
          if (t + 2 < nrows) taps(0, row, acc[(u + 2) % 3]);
        }
      }
    } else {
      // output row r: input rows 2r - 1 (A; it was the C of the row above), 2r (B), 2r + 1 (C); sums in acc[r % 2]
      constexpr int P = NR % 2 == 0 ? NR : 2 * NR;                      // steps until ring entry and accumulator roles repeat
      f2 acc[2][2];
      acc[0][0] = acc[0][1] = zero2;
      cook(-1, first, row);
      taps(0, row, acc[0]);
      for (int t0 = 0; t0 < nrows; t0 += P) {
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const int r = t0 + u, t = r;
          if (t >= nrows) break;
          cook(2 * r, ring[u % NR], row);
          taps(1, row, acc[u & 1]);
          cook(2 * r + 1, ring_b[u % NR], row);
          fetch(2 * (r + NR), ring[u % NR]);
          fetch(2 * (r + NR) + 1, ring_b[u % NR]);
          taps(2, row, acc[u & 1]);
          emit(r, acc[u & 1]);
          acc[(u + 1) & 1][0] = acc[(u + 1) & 1][1] = zero2;
          if (t + 1 < nrows) taps(0, row, acc[(u + 1) & 1]);
        }
      }
    }
  };
  if ((FQ_DW16_V & 4) && fq_nonneg(qc)) walk(std::true_type{});      // (behind a ReLU every clipped value is >= 0, whatever the consumer's range)
  else walk(std::false_type{});
  if (EPI == kEpiBnRelu6) m = fminf(m, 6.0f);
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
  g.wgs_per_sample = (g.CB * g.Wo + 63) / 64;
  g.span = (64 + g.Wo - 2) / g.Wo + 1;
  g.out_levels = act_levels(out_width, out_flags);
  g.out_lo_neg = (out_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
  g.out_zoff = (out_flags & FQ_ACT_SIGNED) ? 0 : 128;
  g.in_zoff = (in_flags & FQ_ACT_SIGNED) ? 0 : 128;
  const int64_t grid = n * (int64_t)g.wgs_per_sample;
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
