// libfakequant — K2m pointwise (1x1) convolution on int8 codes, split form: the kernel template, shared by the translation
// units that instantiate it (fq_pw_split.hip: fp32 NCHW in and out; fq_pw_split16.hip: C16 code tensors in or out)
#ifndef FQ_PW_SPLIT_KERNEL_H_
#define FQ_PW_SPLIT_KERNEL_H_

#include "fq_pw.h"

namespace {

// K2m: split form.  The deep layers (K = 256 ... 2048 on 14x14 / 7x7 planes) have FEW pixels (784 or 196 tiles of 32)
// and big weight matrices; their tensors sit in the Infinity Cache, so what bounds them is how many independent
// instruction streams the chip has to overlap loads, quantisation, matrix work and stores (profiles/r2_pw_experiments.txt:
// one wavefront per SIMD issues an instruction every ~4 cycles and nothing overlaps).  This form cuts the work into
// tiles x channel groups so that ~800-1600 workgroups (3-6 per CU, 3 wavefronts per SIMD) are resident at once:
//   1. the four wavefronts of a workgroup each quantise a quarter of the tile's K/32 channel slabs (lane = pixel, as
//      K2h; up to 64 dword loads per lane in flight before the first quantisation) into an LDS panel of B fragments;
//      workgroups of different channel groups quantise the same tile redundantly (cheap: VALU is idle here, the tile
//      comes from L2) instead of synchronising;
//   2. every wavefront multiplies CW 32-channel tiles AT ONCE (CW independent accumulators share each B fragment read
//      from the panel), A fragments straight from L2 out of the fragment-major copy of fq_weight_codes through a ring of
//      D K-steps in flight (CW x D 16-byte loads per lane; round 1's tile form kept two, and waited on L2 in every step);
//   3. store with lane = pixel (two full lines per store instruction), per-channel constants from LDS.
// One barrier per workgroup.  Work item -> (tile, group) with the group fastest, so that a tile's workgroups run at the
// same time on different XCDs (item i runs on XCD i % 8) and each XCD's L2 keeps only the groups it serves.
struct PwSplitGeom {
  int Cin, Cout, HW, CS;     // CS: channel groups (workgroups) per tile = ceil(Cout / (128 * CW))
  int CTM;                   // 32-channel tiles present in the weight buffer (rows_pad / 32; rows >= Cout are zero)
  int S, Wo, Win, HWin;      // stride (1 or 2): HW / Wo describe the OUTPUT plane, Win / HWin the input plane
  int64_t cols, tiles, items;   // items = tiles * CS
  int zoff;
  // C16 code tensors (round 3; include/fakequant.h at fq_pwconv_i8_c16): [n][ceil(C / 16)][pixels][16 codes], the bytes as the
  // matrix cores take them ((code + 128 - zoff) ^ 0x80).  IN16: x is such a tensor with CBi blocks per sample; OUT16: y is
  // one with CBo blocks, quantised with the CONSUMER's threshold out_thr / levels / clip range.
  int CBi, CBo;
  float out_levels;
  int out_lo_neg, out_zoff;
  // DUAL: the second output (C16 codes of y under dual_thr; out_levels / out_lo_neg / out_zoff / CBo describe it)
  char* y16;
  const float* dual_thr;
  // SUB (fq_pwconv_i8_sub2, round 6): only the pixels (2 i, 2 j) of every output plane are stored, as a dense
  // (n, Cout, ceil(H / 2), ceil(W / 2)) tensor; SW: the plane's width, SWs / SHWs: width and pixels of a stored plane
  int SW, SWs, SHWs;
};

// LB: wavefronts per SIMD the register allocation aims at (3: <= 168 registers, 4: <= 128) - with one tile per workgroup the
// whole grid should be resident at once (a second round of a few left-over workgroups costs a whole workgroup latency)
// NW: wavefronts per workgroup.  4: 128 * CW output channels per workgroup; 8 (wide layers): 256 * CW - half as many channel
// groups quantise the same tile, i.e. half the redundant loads and quantiser VALU.
// SUB: the output is stored subsampled (see PwSplitGeom); statistic and residual operand cover the whole planes.
template <int KT, int CW, int D, int LB, int NW, bool IN16 = false, bool OUT16 = false, bool DUAL = false, bool SUB = false>
__global__ __launch_bounds__(NW * 64, LB) void pwconv_split_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwSplitGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out, const float* __restrict__ residual, const float* __restrict__ out_thr) {
  static_assert(!SUB || !OUT16, "the subsampled output is built for fp32 output (DUAL: and its code copy)");
  constexpr int kSlots = 8;
  constexpr int SLABS = (KT + NW - 1) / NW;                             // slabs a wavefront quantises (kt = wave + NW j < KT)
  constexpr int RB = SLABS < 4 ? SLABS : 4;                             // slabs (16 loads each) in flight per lane
  constexpr int NCH = NW * CW * 32;                                     // output channels of one workgroup
  constexpr int RS = D + 1;                                             // ring slots
  extern __shared__ __attribute__((aligned(16))) unsigned char pwsp_smem[];
  __shared__ unsigned k_stat[kSlots];
  v4i* panel = reinterpret_cast<v4i*>(pwsp_smem);                       // [KT][64] B fragments of the tile
  float* c_sxw = reinterpret_cast<float*>(pwsp_smem + (size_t)KT * 1024);
  float* c_bsc = c_sxw + NCH;
  float* c_bsh = c_bsc + NCH;
  float* c_bias = c_bsh + NCH;
  int* c_zs = reinterpret_cast<int*>(c_bias + NCH);

  const int lane = threadIdx.x & 63;
  // the wavefront index as a SCALAR, so that everything derived from it lives in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pl = lane & 31;
  const unsigned HW = (unsigned)g.HW, cols = (unsigned)g.cols;
  const unsigned plane4 = HW * 4u;                                      // bytes between two channels of one sample
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  // XCD-aware order: workgroup b runs on XCD b % 8 (each XCD has its own L2), and a 32-pixel tile of a 14x14 / 7x7 plane
  // is 128 bytes that are NOT line-aligned, so neighbouring tiles share their first / last cache line of every channel:
  // give every XCD a CONTIGUOUS range of (tile, group) items so that both halves of such a line meet in one L2
  unsigned item;
  {
    const unsigned per = ((unsigned)g.items + 7u) >> 3;
    item = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || item >= (unsigned)g.items) return;
  }
  const unsigned tile = item / (unsigned)g.CS, cg = item - tile * (unsigned)g.CS;
  const int ch0 = (int)cg * NCH;                                        // first output channel of this workgroup
  const unsigned s_base = (tile * 32u) / HW;                            // first sample the tile touches
  unsigned smp, p;
  {
    unsigned j = tile * 32u + (unsigned)pl;
    j = j < cols ? j : cols - 1;                                        // lanes past the end copy the last pixel
    smp = j / HW;
    p = j - smp * HW;
  }
  // Buffer addressing (fq_common.h): resources based at the tile's first sample; a lane's 16 * SLABS loads share ONE
  // offset register (its pixel, and the half of a slab it owns), the channel stride is a scalar offset.  Channels of a
  // padded slab (Cin % 32 != 0) read the next sample's values, or 0 past the end of the tensor: whatever code they get
  // meets a zero weight code (fq_weight_codes pads K with zeros).
  // A strided (2 x 2) 1x1 convolution reads every second pixel of every second row: only the lane's input offset and the
  // input plane size differ from the stride-1 case.
  const unsigned plane4_in = (unsigned)g.HWin * 4u;
  const int64_t x_samp = (int64_t)g.Cin * g.HWin * 4, y_samp = (int64_t)g.Cout * HW * 4;
  const int64_t n_samp = (int64_t)(cols / HW);
  unsigned p_in = p;
  if (g.S != 1) {
    const unsigned ho = p / (unsigned)g.Wo, wo = p - ho * (unsigned)g.Wo;
    p_in = ho * (unsigned)(g.S * g.Win) + wo * (unsigned)g.S;
  }
  // IN16: the lane's 16 codes of slab kt (channels 32 kt + 16 h ..) are ONE 16-byte vector of block 2 kt + h - already in the
  // representation the panel holds, so a slab costs one load and one LDS store instead of 16 loads and the quantiser
  const int64_t x_samp16 = (int64_t)g.CBi * g.HWin * 16;
  const fq_rsrc xr = IN16 ? make_rsrc(reinterpret_cast<const char*>(x) + s_base * x_samp16, (n_samp - s_base) * x_samp16)
                          : make_rsrc(reinterpret_cast<const char*>(x) + s_base * x_samp, (n_samp - s_base) * x_samp);
  const unsigned xo = IN16 ? ((smp - s_base) * (unsigned)g.CBi + (unsigned)h) * (unsigned)g.HWin * 16u + p_in * 16u
                           : ((smp - s_base) * (unsigned)g.Cin + 16u * h) * plane4_in + p_in * 4u;
  auto issue = [&](int kt, float (&v)[16]) __attribute__((always_inline)) {
    if (IN16) {
      const unsigned off = (2 * kt + h) < g.CBi ? xo : 0x80000000u;      // a block past the tensor's channels: zeros
      const v4i c = buf_ld_v4i(xr, off, (unsigned)(2 * kt) * (unsigned)g.HWin * 16u);
      v[0] = __int_as_float(c[0]); v[1] = __int_as_float(c[1]); v[2] = __int_as_float(c[2]); v[3] = __int_as_float(c[3]);
      return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = buf_ld_f32(xr, xo, (unsigned)(kt * 32 + i) * plane4_in);
  };

  PW_STAMP(0);
#ifdef FQ_PW_TRACE
  if (threadIdx.x == 0 && g_pw_trace != nullptr)
    g_pw_trace[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) |
                                             (unsigned long long)__builtin_amdgcn_s_getreg(63492);
#endif
  // (slabs that are padding altogether - cin_pad is a multiple of 64 - are neither loaded nor written to the panel: whatever
  // bytes the panel holds there meet zero weight codes)
  const int kt_real = (g.Cin + 31) >> 5;
  const ThresholdReq treq = threshold_request(in_stat, n, in_thr, item == 0);   // first in the memory queue
  float buf[RB][16];
#pragma unroll
  for (int i = 0; i < RB; ++i)
    if (wave + NW * i < kt_real) issue(wave + NW * i, buf[i]);          // in flight during the set-up
  FQ_PIN();
  const float max_ = threshold_finish(treq, in_stat, n, in_thr, cur_max_out, item == 0);
  int zoff = g.zoff;
  const QParams q = make_qparams_rt(max_, levels, lo_neg_max, eps, in_thr, zoff);
  const float sx = q.scale;
  // range mode (nn.Conv2D(quantized=True)): `bias` holds int32 codes that join the integer sum
  const int* ibias = lo_neg_max == kRangeMode ? reinterpret_cast<const int*>(bias) : nullptr;
  const float* fbias = lo_neg_max == kRangeMode ? nullptr : bias;
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;
  for (int i = threadIdx.x; i < NCH; i += NW * 64) {
    const bool ok = ch0 + i < g.Cout;                                   // channels past Cout: all-zero constants
    const int ic = ok ? ch0 + i : 0;
    c_sxw[i] = ok ? sx * wscale[ic] : 0.0f;
    c_zs[i] = ok ? zoff * wsum[ic] + (ibias != nullptr ? ibias[ic] : 0) : 0;
    c_bias[i] = ok && fbias != nullptr ? fbias[ic] : 0.0f;
    c_bsc[i] = has_bn && ok ? bn_scale[ic] : (ok ? 1.0f : 0.0f);
    c_bsh[i] = has_bn && ok ? bn_shift[ic] : 0.0f;
  }
  PW_STAMP(1);
  const int ubias = 128 - zoff;
  const unsigned nn_xor = fq_nonneg_xor(ubias);
  auto quant_to_panel = [&](int kt, const float (&v)[16], auto nn_c) __attribute__((always_inline)) {
    v4i f;
    if (IN16) {
      f = (v4i){__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])};
      panel[(kt << 6) + lane] = f;
      return;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d)
      f[d] = fq_pack4<decltype(nn_c)::value>(v[4 * d + 0], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3], q, ubias, nn_xor);
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));   // pin the arithmetic here (see K2h)
    panel[(kt << 6) + lane] = f;
  };
  // ---- 1. my quarter of the slabs -> LDS panel ------------------------------------------------------------------------
  // (non-negative quotients - unsigned activations - take the 5-instruction quantiser of fq_common.h)
  auto fill_panel = [&](auto nn_c) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < SLABS; ++j) {
      if (wave + NW * j < kt_real) quant_to_panel(wave + NW * j, buf[j % RB], nn_c);
      FQ_PIN();
      if (j + RB < SLABS) {
        if (wave + NW * (j + RB) < kt_real) issue(wave + NW * (j + RB), buf[j % RB]);
        FQ_PIN();
      }
    }
  };
  if (fq_nonneg(q)) fill_panel(std::true_type{});
  else fill_panel(std::false_type{});
  // ---- 2. CW channel tiles at once; the first D K-steps of A fragments are requested before the barrier ---------------
  // A fragment (channel tile ct, slab kt) = 1 KB at wfrag + (ct * KT + kt) * 1024
  // Channel tiles past the padded weight buffer read zeros through the bound of the resource; tiles past Cout are not stored.
  const int ctl0 = wave * CW;                                           // first channel tile inside the workgroup
  const int ctg0 = (int)cg * NW * CW + ctl0;                            // ... and in the layer
  const int ct_here = g.CTM - ctg0 < CW ? (g.CTM - ctg0 < 0 ? 0 : g.CTM - ctg0) : CW;
  const fq_rsrc wr = make_rsrc(wfrag + (((int64_t)ctg0 * KT) << 10), (int64_t)ct_here * KT * 1024);
  const unsigned loff = (unsigned)lane * 16u;
  auto a_frag = [&](int c, int kt) __attribute__((always_inline)) {
    return buf_ld_v4i(wr, loff, (unsigned)((c * KT + kt) << 10));
  };
  v4i ring[RS][CW];
#pragma unroll
  for (int d = 0; d < D; ++d) {
#pragma unroll
    for (int c = 0; c < CW; ++c) ring[d][c] = a_frag(c, d < KT ? d : KT - 1);
  }
  FQ_PIN();
  PW_STAMP(2);
  __syncthreads();                                                      // panel, constants and the statistic table
  PW_STAMP(3);
  const int cvalid = g.Cout - (ch0 + ctl0 * 32);                       // valid output channels from this wavefront's first tile on
  QParams q2;
  q2.lo = q2.hi = q2.denom = q2.scale = 0.0f;
  q2.rden = 0.0;
  if (OUT16) q2 = make_qparams(out_thr[0], g.out_levels, g.out_lo_neg != 0, eps);
  // DUAL (fq_pwconv_i8_c16_dual): y is fp32 AND g.y16 receives the codes of the same values under g.dual_thr - the trunk of a
  // ResNet stored a second time, 1 B per element, for the next unit's first 1x1 (the shortcut keeps reading the fp32 tensor)
  if (DUAL) q2 = make_qparams(g.dual_thr[0], g.out_levels, g.out_lo_neg != 0, eps);
  // (nn2_c: the CONSUMER's clip range of a C16 output starts at 0 - the five-instruction quantiser of fq_common.h writes its
  // codes; round 5: until then only the second output's)
  auto run = [&](auto bias_c, auto bn_c, auto act_c, auto nn2_c) __attribute__((always_inline)) {
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
    constexpr bool NN2 = decltype(nn2_c)::value;
    // A code output behind a compile-time ReLU / ReLU6: activation and the consumer's clip are ONE median - clip(relu6(v), lo
    // <= 0, hi) == med3(v, 0, min(6, hi)) for every v, NaN -> 0 on both sides - and the statistic max_i relu6(v_i) ==
    // min(max(0, max_i v_i), 6) is taken from the raw values and clamped once per wavefront (a v_med3 less per output)
    constexpr bool FOLD = OUT16 && !DUAL && (ACT_M == FQ_ACT_RELU || ACT_M == FQ_ACT_RELU6);
    QParams qc = q2;
    if (FOLD) {
      qc.lo = 0.0f;
      if (ACT_M == FQ_ACT_RELU6) qc.hi = fminf(q2.hi, 6.0f);
    }
    // accumulators start at zero (the first MFMA takes the constant): initialising them with the +128 re-centring terms
    // keeps a second set of 16 * CW registers alive next to the destination of the first MFMAs; the terms are added as
    // integers in the epilogue instead (one VALU per output)
    v16i acc[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0;
    constexpr int BA = LB >= 4 ? 1 : 2;                          // B fragments read ahead (LDS latency vs registers)
    v4i bq[BA + 1];
#pragma unroll
    for (int d = 0; d < BA; ++d) bq[d] = panel[(d << 6) + lane];
    FQ_PIN();
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      if (kt + D < KT) {
#pragma unroll
        for (int c = 0; c < CW; ++c) ring[(kt + D) % RS][c] = a_frag(c, kt + D);
      }
      if (kt + BA < KT) bq[(kt + BA) % (BA + 1)] = panel[((kt + BA) << 6) + lane];
#pragma unroll
      for (int c = 0; c < CW; ++c)
        acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ring[kt % RS][c], bq[kt % (BA + 1)], acc[c], 0, 0, 0);
      FQ_PIN();                                  // keep the ring as written (else every load is hoisted to the top)
    }
    PW_STAMP(4);
    // ---- 3. epilogue: lane = pixel, two full lines per store instruction -------------------------------------------------
    // The resource is bounded below 2 GiB, so a lane offset of 0x80000000 is out of range for it: that is how the channels
    // past Cout of a PARTIAL channel tile are masked (the hardware drops the store; no branch, no exec juggling).
    int64_t y_bytes = (n_samp - s_base) * y_samp - (int64_t)(ch0 + ctl0 * 32) * plane4;
    y_bytes = y_bytes < 0x7FFFFFFFll ? y_bytes : 0x7FFFFFFFll;
    // SUB: stores go to the dense tensor of the even pixels of the even rows (plane4s bytes per channel); the residual
    // operand keeps the full planes
    const unsigned plane4s = SUB ? (unsigned)g.SHWs * 4u : plane4;
    const int64_t y_samp_s = SUB ? (int64_t)g.Cout * g.SHWs * 4 : y_samp;
    int64_t ys_bytes = (n_samp - s_base) * y_samp_s - (int64_t)(ch0 + ctl0 * 32) * plane4s;
    ys_bytes = ys_bytes < 0x7FFFFFFFll ? ys_bytes : 0x7FFFFFFFll;
    // OUT16: y is a C16 code tensor; the resource starts at (first sample, this wavefront's first 16-channel block)
    // (SUB + DUAL: the code copy holds the stored pixels only, as y does)
    const unsigned HWo = SUB ? (unsigned)g.SHWs : HW;                      // pixels of a stored plane
    const int64_t y_samp16 = (int64_t)g.CBo * HWo * 16;
    const int cb0 = (ch0 + ctl0 * 32) >> 4;                               // first output block of this wavefront
    int64_t y16_bytes = (n_samp - s_base) * y_samp16 - (int64_t)cb0 * HWo * 16;
    y16_bytes = y16_bytes < 0x7FFFFFFFll ? y16_bytes : 0x7FFFFFFFll;
    const fq_rsrc yr = OUT16 ? make_rsrc(reinterpret_cast<char*>(y) + s_base * y_samp16 + (int64_t)cb0 * HWo * 16, y16_bytes)
                             : make_rsrc(reinterpret_cast<char*>(y) + s_base * y_samp_s + (int64_t)(ch0 + ctl0 * 32) * plane4s, ys_bytes);
    unsigned yo16 = (smp - s_base) * (unsigned)g.CBo * HWo * 16u + p * 16u + 4u * h;
    const fq_rsrc yr16 = make_rsrc(DUAL ? g.y16 + s_base * y_samp16 + (int64_t)cb0 * HWo * 16 : reinterpret_cast<char*>(y),
                                   DUAL ? y16_bytes : 0);
    const int ubias2 = 128 - g.out_zoff;
    // the residual operand (the shortcut of a ResNet / MobileNetV2 unit) has y's shape: same offsets, added after BatchNorm
    const bool has_res = residual != nullptr;
    const fq_rsrc rr = make_rsrc(reinterpret_cast<const char*>(has_res ? residual : y) + s_base * y_samp +
                                 (int64_t)(ch0 + ctl0 * 32) * plane4, has_res ? y_bytes : 0);
    const unsigned yo = ((smp - s_base) * (unsigned)g.Cout + 4u * h) * plane4 + p * 4u;
    unsigned yos = yo;                           // the lane's store offset (SUB: out of range for a pixel that is not stored)
    if (SUB) {
      const unsigned ho = p / (unsigned)g.SW, wo = p - ho * (unsigned)g.SW;
      yos = ((ho | wo) & 1u) ? 0x80000000u
                             : ((smp - s_base) * (unsigned)g.Cout + 4u * h) * plane4s + ((ho >> 1) * (unsigned)g.SWs + (wo >> 1)) * 4u;
      yo16 = ((ho | wo) & 1u) ? 0x80000000u
                              : (smp - s_base) * (unsigned)g.CBo * HWo * 16u + ((ho >> 1) * (unsigned)g.SWs + (wo >> 1)) * 16u + 4u * h;
    }
    float m = 0.0f;
    auto store_tile = [&](int c, int cv, auto masked_c) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked_c)::value;
      const int cb = (ctl0 + c) * 32 + 4 * h;                           // channel inside the workgroup's group
      float res[16];
      if (has_res) {                             // all 16 in flight before the first use
        // channels past Cout of a PARTIAL tile get the same out-of-range offset as their stores and so read 0: the
        // resource runs to the end of the tensor, an unmasked load would fetch the NEXT sample's channels 0.. and, with
        // all-zero constants, carry that foreign value into the statistic `m` below
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const unsigned off = MASKED ? (8 * (i >> 2) + 4 * h + (i & 3) < cv ? yo : 0x80000000u) : yo;
          res[i] = buf_ld_f32(rr, off, (unsigned)(c * 32 + 8 * (i >> 2) + (i & 3)) * plane4);
        }
      }
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int c0 = cb + 8 * gq;
        const v4i zs = *reinterpret_cast<const v4i*>(c_zs + c0);
        const f4 sxw = *reinterpret_cast<const f4*>(c_sxw + c0);
        const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
        const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
        f4 bch = (f4){0.f, 0.f, 0.f, 0.f};
        if (BIAS_M == 1 || (BIAS_M < 0 && fbias != nullptr)) bch = *reinterpret_cast<const f4*>(c_bias + c0);
        float vq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = (float)(acc[c][4 * gq + r] + zs[r]) * sxw[r];
          if (BIAS_M == 1 || (BIAS_M < 0 && fbias != nullptr)) v = v + bch[r];
          if (BN_M == 1 || (BN_M < 0 && has_bn)) {
            v = v * bsc[r];
            v = v + bsh[r];
          }
          if (has_res) v = v + res[4 * gq + r];
          if (!FOLD) v = ACT_M < 0 ? act_rt(v, act) : act_rt(v, ACT_M);
          vq[r] = v;
          if (!OUT16) {
            const unsigned off = MASKED ? (8 * gq + 4 * h + r < cv ? yos : 0x80000000u) : yos;
            buf_st_f32(yr, off, (unsigned)(c * 32 + 8 * gq + r) * plane4s, v);
          }
          m = FOLD ? fmaxf(m, v) : fmaxf(m, fabsf(v));   // channels past Cout have all-zero constants: v == 0
        }
        if (OUT16 || DUAL) {
          // the four channels 8 gq + 4 h .. + 3 of this lane's pixel are bytes 8 (gq & 1) + 4 h .. of block (c * 2 + gq / 2):
          // the CONSUMER's codes of the values just formed (channels past Cout: v == 0 -> code 0), one 4-byte store
          // (DUAL: unsigned codes of a non-negative range only - the five-instruction quantiser of fq_common.h.  One 16-byte
          // store per lane instead of these four 4-byte ones - the halves exchanged with lane ^ 32 - was built and measured:
          // slower in every C16-writing kernel, ResNet-50 offline 41.7 -> 40.1 k images/s)
          const int packed = DUAL ? fq_pack4<true>(vq[0], vq[1], vq[2], vq[3], q2, ubias2, 0x80808080u)
                                  : fq_pack4<NN2>(vq[0], vq[1], vq[2], vq[3], qc, ubias2, fq_nonneg_xor(ubias2));
          const bool blk_ok = !MASKED || 16 * (gq >> 1) < cv;             // a whole block past Cout does not exist
          buf_st_f32(OUT16 ? yr : yr16, blk_ok ? yo16 : 0x80000000u,
                     (unsigned)((c * 2 + (gq >> 1)) * (int)HWo * 16 + 8 * (gq & 1)), __int_as_float(packed));
        }
      }
    };
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      FQ_PIN();                                  // constants of one channel tile at a time (else all are read up front)
      const int cv = cvalid - c * 32;            // valid channels of this tile (wave-uniform)
      if (cv >= 32) store_tile(c, cv, std::false_type{});
      else if (cv > 0) store_tile(c, cv, std::true_type{});
    }
    if (FOLD && ACT_M == FQ_ACT_RELU6) m = fminf(m, 6.0f);
    if (has_stat) {
      const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)smp);
      if (__all(smp == s0)) {
        const float wm = wave_max_nonneg(m);
        if (lane == 0) {
          const unsigned slot = s0 - s_base;
          if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
          else atomic_max_f32(stat_out + s0, wm);
        }
      } else {
        const unsigned slot = smp - s_base;
        if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
        else atomic_max_f32(stat_out + smp, m);
      }
    }
  };
  using std::integral_constant;
  // (only a kernel that writes codes is instantiated twice: with the five-instruction quantiser of fq_common.h where the values it
  // clips cannot be negative - the consumer's range starts at 0, or a ReLU stands in front of it - and with the general one)
  auto go = [&](auto bias_c, auto bn_c, auto act_c, bool nn2) __attribute__((always_inline)) {
    if constexpr (OUT16 && !DUAL) {
      if (nn2) run(bias_c, bn_c, act_c, std::true_type{});
      else run(bias_c, bn_c, act_c, std::false_type{});
    } else {
      run(bias_c, bn_c, act_c, std::false_type{});
    }
  };
  const bool nn2_relu = q2.denom > 0.0f, nn2_any = fq_nonneg(q2);
  if (cvalid <= 0) {
    // nothing to multiply (a channel group wider than the layer): this wavefront only helped to quantise the tile
  } else if (fbias == nullptr && has_bn && act == FQ_ACT_RELU)
    go(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU>{}, nn2_relu);
  else if (fbias == nullptr && has_bn && act == FQ_ACT_RELU6)
    go(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU6>{}, nn2_relu);
  else if (fbias == nullptr && has_bn && act == FQ_ACT_NONE)
    go(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_NONE>{}, nn2_any);
  else
    go(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{}, nn2_any);
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < cols / HW)
      FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
  PW_STAMP(5);
}


}  // namespace

#endif  // FQ_PW_SPLIT_KERNEL_H_
