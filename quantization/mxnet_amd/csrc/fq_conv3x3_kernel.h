// libfakequant — K2n dense 3x3 convolution on int8 codes: the kernel template, shared by the translation units that
// instantiate it (fq_conv3x3.hip: fp32 NCHW in and out, one or three weight slices; fq_conv3x3_16.hip: C16 code tensors)
#ifndef FQ_CONV3X3_KERNEL_H_
#define FQ_CONV3X3_KERNEL_H_

#include "fq_common.h"

namespace {

// K2n.  After fake-quantisation a dense convolution is, like the 1x1 case (K2m), an EXACT integer problem:
//   sum_{ci,ky,kx} w_q * x_q  =  sx * sw[co] * sum cw * cx,   |sum| <= 9 * Cin * 255 * 127 < 2^31 for Cin <= 512.
// It is an implicit GEMM with K = 9 * Cin ordered (tap, ci): the weights arrive permuted to (Cout, 3, 3, Cin), so
// fq_weight_codes' fragment-major copy holds fragment (channel tile, tap * KT + kt) and its row sums cover all 9 * Cin codes.
// One (pixel block, channel group) per workgroup, the structure of K2m:
//   1. a pixel block is 32 * PT CONSECUTIVE pixels of the flattened (n, h, w) order; with its halo - the W + 1 pixels before
//      and after it - that is one contiguous run of RP = 32 * PT + 2 W + 2 pixels, which the four wavefronts load (lane =
//      pixel, 16 channels per lane and slab, buffer addressing as K2m), quantise ONCE and write to an LDS panel laid out
//      [pixel][channel] (row = Cin + 16 bytes: the 16-byte reads below are then bank-conflict free for every Cin here);
//   2. for tap (dy, dx) the B fragment of pixel tile t is the SAME panel read dy * W + dx pixels further on: one
//      ds_read_b128 per lane, no im2col anywhere.  A tap that falls outside the image (or into the neighbouring row /
//      sample of the flattened order) must contribute the code 0: nine validity bits per lane and pixel tile select
//      between the fragment and the byte pattern of code 0 (0x80 re-centred, K2m) - 5 VALU per fragment, hidden under the
//      MFMAs;
//   3. wavefront (wc, wp) multiplies channel tile wc of the group with pixel tiles wp * PTW .. + PTW - 1 (PTW independent
//      accumulators share each A fragment, fetched from L2 through a ring of D K-steps as in K2m);
//   4. epilogue as K2m: lane = pixel, BatchNorm / activation / per-sample statistic on store.
// The halo makes a block's quantisation work (32 PT + 2 W + 2) / (32 PT) of its pixels (1.45 at 56x56 with PT = 8, 1.5 at
// 7x7 with PT = 1) and channel groups repeat it - cheap next to the 9 * Cin * 32 multiply-adds per pixel and group.
//
// NSL = 3 (round 3): weights that are NOT integer multiples of one scale per channel - the Winograd-domain quantisation of
// the reference (convert_conv2d.py:71-83: the int8 grid lives in the Winograd domain, the spatial filter is GI U^ GTI) - as
// THREE int8 slices.  Per output channel p = 2^e with 2^e >= max|w| * 2^-20; m = rint(w / p) (|m| <= 2^20, the division is
// exact) is written in balanced base 128, m = d1 2^14 + d2 2^7 + d3 with digits in [-64, 64]; every slice is an exact int32
// convolution S_i = sum d_i * cx on the matrix cores, and the epilogue combines T = (S1 << 14) + (S2 << 7) + S3 in 64 bits:
// y = fp32(T * (sx * p)).  The activations are quantised once and their fragments are shared by the three slices.  The
// weights are represented to p / 2 <= 2^-20 of the channel maximum and the sum over 9 * Cin products is EXACT: an error of the
// order an fp32 convolution of the same tensors accumulates by rounding every product and partial sum (fq_weight_slices below).
struct C3Geom {
  int Cin, Cout, H, W, HW;
  int CS;                    // channel groups per pixel block = ceil(Cout / (32 * WC))
  int CTM;                   // 32-channel tiles present in the weight buffer
  int RP, RT;                // region pixels, region tiles of 32
  int ROW;                   // bytes per region pixel in the panel
  int64_t cols, items;       // n * HW; pixel blocks * CS
  int zoff;
  int64_t slice_bytes;       // NSL = 3: bytes between the code buffers of two slices; rows between their row sums
  int slice_rows;
  // C16 code tensors on either side (fq_conv3x3_i8_c16; the layout: include/fakequant.h): IN16 - x holds the codes this
  // kernel would make; OUT16 - y receives the CONSUMER's codes of act(BN(conv)) under out_thr
  int CBi, CBo;
  float out_levels;
  int out_lo_neg, out_zoff;
};

// NW wavefronts per workgroup (4, or 8 for wide layers with few pixel blocks: half as many channel groups quantise a region)
template <int KT, int PTW, int WC, int D, int LB, int NW, int NSL, bool IN16 = false, bool OUT16 = false>
__global__ __launch_bounds__(NW * 64, LB) void conv3x3_i8_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, C3Geom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out, const float* __restrict__ out_thr) {
  constexpr int kSlots = 8;
  constexpr int WP = NW / WC;                                           // wavefronts along the pixel direction
  constexpr int PT = PTW * WP;                                          // pixel tiles of a workgroup
  constexpr int NCH = WC * 32;                                          // output channels of a workgroup
  constexpr int RS = D + 1;
  constexpr int NS = 9 * KT;                                            // K-steps
  extern __shared__ __attribute__((aligned(16))) unsigned char c3_smem[];
  __shared__ unsigned k_stat[kSlots];
  unsigned char* panel = c3_smem;                                       // [RT * 32][ROW] codes of the region
  float* c_sxw = reinterpret_cast<float*>(c3_smem + (size_t)g.RT * 32 * g.ROW);
  float* c_bsc = c_sxw + NCH;
  float* c_bsh = c_bsc + NCH;
  float* c_bias = c_bsh + NCH;
  int* c_zs = reinterpret_cast<int*>(c_bias + NCH);                     // [NSL][NCH]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar (see K2m)
  const int h = lane >> 5, pl = lane & 31;
  const unsigned HW = (unsigned)g.HW, W = (unsigned)g.W, cols = (unsigned)g.cols;
  const unsigned plane4 = HW * 4u;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  unsigned item;                                                        // XCD-contiguous work order (K2m)
  {
    const unsigned per = ((unsigned)g.items + 7u) >> 3;
    item = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || item >= (unsigned)g.items) return;
  }
  const unsigned pb = item / (unsigned)g.CS, cg = item - pb * (unsigned)g.CS;
  const int ch0 = (int)cg * NCH;
  const unsigned j0 = pb * (32u * PT);                                  // first pixel of the block
  const int jr0 = (int)j0 - (int)W - 1;                                 // first pixel of the region (may be < 0)
  const unsigned s_base = j0 / HW;                                      // first sample the block's OUTPUT touches
  const unsigned n_base = (unsigned)(jr0 < 0 ? 0 : jr0) / HW;           // first sample the region touches
  const int64_t x_samp = (int64_t)g.Cin * HW * 4, y_samp = (int64_t)g.Cout * HW * 4;
  const int64_t n_samp = (int64_t)(cols / HW);
  const int64_t x_samp16 = (int64_t)g.CBi * HW * 16;
  const fq_rsrc xr = IN16 ? make_rsrc(reinterpret_cast<const char*>(x) + n_base * x_samp16, (n_samp - n_base) * x_samp16)
                          : make_rsrc(reinterpret_cast<const char*>(x) + n_base * x_samp, (n_samp - n_base) * x_samp);

  // ---- 1. region -> LDS panel: units of (region tile of 32 pixels, slab of 32 channels), wave-strided -----------------
  const int NU = g.RT * KT;
  auto unit_off = [&](int u) __attribute__((always_inline)) {           // lane offset of this lane's pixel in unit u
    const int rt = u / KT;
    int jr = jr0 + rt * 32 + pl;
    jr = jr < 0 ? 0 : (jr < (int)cols ? jr : (int)cols - 1);            // outside the tensor: any valid pixel (masked later)
    const unsigned nr = (unsigned)jr / HW;
    if (IN16) return ((nr - n_base) * (unsigned)g.CBi + (unsigned)h) * HW * 16u + ((unsigned)jr - nr * HW) * 16u;
    return ((nr - n_base) * (unsigned)g.Cin + 16u * h) * plane4 + ((unsigned)jr - nr * HW) * 4u;
  };
  auto issue = [&](int u, float (&v)[16]) __attribute__((always_inline)) {
    const unsigned xo = unit_off(u);
    const int kt = u % KT;
    if (IN16) {                                   // the lane's 16 codes of the slab: one 16-byte vector of block 2 kt + h
      const v4i c = buf_ld_v4i(xr, xo, (unsigned)(2 * kt) * HW * 16u);
      v[0] = __int_as_float(c[0]); v[1] = __int_as_float(c[1]); v[2] = __int_as_float(c[2]); v[3] = __int_as_float(c[3]);
      return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = buf_ld_f32(xr, xo, (unsigned)(kt * 32 + i) * plane4);
  };
  float bufa[16], bufb[16];
  PW_STAMP(0);
  const ThresholdReq treq = threshold_request(in_stat, n, in_thr, item == 0);   // first in the memory queue (fq_common.h)
  FQ_PIN();
  if (wave < NU) issue(wave, bufa);                                     // in flight during the set-up
  FQ_PIN();
  const float max_ = threshold_finish(treq, in_stat, n, in_thr, cur_max_out, item == 0);
  int zoff = g.zoff;
  const QParams q = make_qparams_rt(max_, levels, lo_neg_max, eps, in_thr, zoff);
  const float sx = q.scale;
  // range mode (nn.Conv2D(quantized=True)): `bias` holds int32 codes that join the integer sum (one slice only)
  const int* ibias = lo_neg_max == kRangeMode ? reinterpret_cast<const int*>(bias) : nullptr;
  const float* fbias = lo_neg_max == kRangeMode ? nullptr : bias;
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;
  for (int i = threadIdx.x; i < NCH; i += NW * 64) {
    const bool ok = ch0 + i < g.Cout;
    const int ic = ok ? ch0 + i : 0;
    c_sxw[i] = ok ? sx * wscale[ic] : 0.0f;
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
      c_zs[sl * NCH + i] = ok ? zoff * wsum[sl * g.slice_rows + ic] + (sl == 0 && ibias != nullptr ? ibias[ic] : 0) : 0;
    c_bias[i] = ok && fbias != nullptr ? fbias[ic] : 0.0f;
    c_bsc[i] = has_bn && ok ? bn_scale[ic] : (ok ? 1.0f : 0.0f);
    c_bsh[i] = has_bn && ok ? bn_shift[ic] : 0.0f;
  }
  const int ubias = 128 - zoff;
  const unsigned nn_xor = fq_nonneg_xor(ubias);
  auto quant_to_panel = [&](int u, const float (&v)[16], auto nn_c) __attribute__((always_inline)) {
    v4i f;
    if (IN16) {
      f = (v4i){__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])};
    } else {
#pragma unroll
      for (int d = 0; d < 4; ++d)
        f[d] = fq_pack4<decltype(nn_c)::value>(v[4 * d + 0], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3], q, ubias, nn_xor);
    }
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    const int rt = u / KT, kt = u - rt * KT;
    *reinterpret_cast<v4i*>(panel + (size_t)(rt * 32 + pl) * g.ROW + kt * 32 + 16 * h) = f;
  };
  // (non-negative quotients - unsigned activations - take the 5-instruction quantiser of fq_common.h)
  auto fill_panel = [&](auto nn_c) __attribute__((always_inline)) {
    for (int u = wave; u < NU; u += 2 * NW) {
      if (u + NW < NU) issue(u + NW, bufb);
      FQ_PIN();
      quant_to_panel(u, bufa, nn_c);
      FQ_PIN();
      if (u + NW < NU) {
        if (u + 2 * NW < NU) issue(u + 2 * NW, bufa);
        FQ_PIN();
        quant_to_panel(u + NW, bufb, nn_c);
        FQ_PIN();
      }
    }
  };
  PW_STAMP(6);
  if (fq_nonneg(q)) fill_panel(std::true_type{});
  else fill_panel(std::false_type{});
  PW_STAMP(7);

  // ---- 2. this wavefront's channel tile x PTW pixel tiles ---------------------------------------------------------------
  const int wc = wave % WC, wp = wave / WC;
  const int ctg = (int)cg * WC + wc;                                    // channel tile in the layer
  // (NSL slices: one resource over all of them - the host checks that they lie within 2 GiB - and the slice in the scalar offset)
  const fq_rsrc wr = make_rsrc(wfrag + (((int64_t)ctg * NS) << 10),
                               ctg < g.CTM ? (int64_t)(NSL - 1) * g.slice_bytes + (int64_t)NS * 1024 : 0);
  const unsigned loff = (unsigned)lane * 16u;
  auto a_frag = [&](int s, int sl) __attribute__((always_inline)) {
    return buf_ld_v4i(wr, loff, (unsigned)(s << 10) + (unsigned)sl * (unsigned)g.slice_bytes);
  };
  v4i ring[RS][NSL];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) ring[d][sl] = a_frag(d < NS ? d : NS - 1, sl);
  // per pixel tile: the lane's pixel, its nine tap-validity bits, its panel row
  unsigned smp[PTW], pp[PTW], tapmask[PTW], rbase[PTW];
#pragma unroll
  for (int t = 0; t < PTW; ++t) {
    unsigned j = j0 + (unsigned)((wp * PTW + t) * 32 + pl);
    j = j < cols ? j : cols - 1;                                        // lanes past the end copy the last pixel
    smp[t] = j / HW;
    pp[t] = j - smp[t] * HW;
    const int hh = (int)(pp[t] / W), ww = (int)(pp[t] - (unsigned)hh * W);
    unsigned m = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const bool ok = hh + dy >= 0 && hh + dy < g.H && ww + dx >= 0 && ww + dx < g.W;
      m |= ok ? (1u << tap) : 0u;
    }
    tapmask[t] = m;
    // region row of pixel j + shift:  (j - jr0) + shift = (j - j0) + W + 1 + shift; lanes past the end were moved back to
    // the last pixel, so use the real difference
    rbase[t] = (unsigned)((int)j - jr0) * (unsigned)g.ROW + 16u * h;
  }
  FQ_PIN();
  PW_STAMP(1);
  __syncthreads();                                                      // panel, constants and the statistic table
  PW_STAMP(2);
  const int cvalid = g.Cout - (ch0 + wc * 32);                          // valid channels of this wavefront's tile
  const int zb = stored_zero4(ubias);                                   // four codes "0" in the stored representation
  // the consumer's quantiser of a C16 output (nn2_c: its clip range starts at 0 - the five-instruction form of fq_common.h)
  QParams q2;
  q2.lo = q2.hi = q2.denom = q2.scale = 0.0f;
  q2.rden = 0.0;
  if (OUT16) q2 = make_qparams(out_thr[0], g.out_levels, g.out_lo_neg != 0, eps);
  auto run = [&](auto bias_c, auto bn_c, auto act_c, auto nn2_c) __attribute__((always_inline)) {
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
    constexpr bool NN2 = decltype(nn2_c)::value;
    // (a code output behind the compile-time ReLU: activation and the consumer's clip as ONE median, the statistic from the raw
    // values - fq_pw_split_kernel.h)
    constexpr bool FOLD = OUT16 && ACT_M == FQ_ACT_RELU;
    QParams qc = q2;
    if (FOLD) qc.lo = 0.0f;
    v16i acc[NSL][PTW];
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
      for (int t = 0; t < PTW; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[sl][t][i] = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int shift = ((tap / 3 - 1) * (int)W + (tap % 3 - 1)) * g.ROW;   // wave-uniform
      unsigned addr[PTW];
      bool tv[PTW];
#pragma unroll
      for (int t = 0; t < PTW; ++t) {
        addr[t] = (unsigned)((int)rbase[t] + shift);
        tv[t] = (tapmask[t] >> tap) & 1u;
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        constexpr int dummy = 0;
        (void)dummy;
        const int s = tap * KT + kt;
        if (s + D < NS) {
#pragma unroll
          for (int sl = 0; sl < NSL; ++sl) ring[(s + D) % RS][sl] = a_frag(s + D, sl);
        }
        v4i b[PTW];
#pragma unroll
        for (int t = 0; t < PTW; ++t) {
          const v4i raw = *reinterpret_cast<const v4i*>(panel + addr[t] + kt * 32);
          b[t][0] = tv[t] ? raw[0] : zb;
          b[t][1] = tv[t] ? raw[1] : zb;
          b[t][2] = tv[t] ? raw[2] : zb;
          b[t][3] = tv[t] ? raw[3] : zb;
        }
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
          for (int t = 0; t < PTW; ++t)
            acc[sl][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ring[s % RS][sl], b[t], acc[sl][t], 0, 0, 0);
        FQ_PIN();
      }
    }
    PW_STAMP(3);
    // ---- 3. epilogue (K2m): lane = pixel, channels past Cout masked through out-of-range offsets ----------------------
    int64_t y_bytes = (n_samp - s_base) * y_samp - (int64_t)(ch0 + wc * 32) * plane4;
    y_bytes = y_bytes < 0x7FFFFFFFll ? y_bytes : 0x7FFFFFFFll;
    const int64_t y_samp16 = (int64_t)g.CBo * HW * 16;
    const int cb0 = (ch0 + wc * 32) >> 4;                                 // first output block of this wavefront (OUT16)
    int64_t y16_bytes = (n_samp - s_base) * y_samp16 - (int64_t)cb0 * HW * 16;
    y16_bytes = y16_bytes < 0x7FFFFFFFll ? y16_bytes : 0x7FFFFFFFll;
    const fq_rsrc yr = OUT16 ? make_rsrc(reinterpret_cast<char*>(y) + s_base * y_samp16 + (int64_t)cb0 * HW * 16, y16_bytes)
                             : make_rsrc(reinterpret_cast<char*>(y) + s_base * y_samp + (int64_t)(ch0 + wc * 32) * plane4, y_bytes);
    const int ubias2 = 128 - g.out_zoff;
    const int cb = wc * 32 + 4 * h;
    const bool partial = cvalid < 32;
#pragma unroll
    for (int t = 0; t < PTW; ++t) {
      FQ_PIN();
      const unsigned yo = ((smp[t] - s_base) * (unsigned)g.Cout + 4u * h) * plane4 + pp[t] * 4u;
      float m = 0.0f;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int c0 = cb + 8 * gq;
        v4i zs[NSL];
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) zs[sl] = *reinterpret_cast<const v4i*>(c_zs + sl * NCH + c0);
        const f4 sxw = *reinterpret_cast<const f4*>(c_sxw + c0);
        const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
        const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
        f4 bch = (f4){0.f, 0.f, 0.f, 0.f};
        if (BIAS_M == 1 || (BIAS_M < 0 && fbias != nullptr)) bch = *reinterpret_cast<const f4*>(c_bias + c0);
        float vq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v;
          if (NSL == 1) {
            v = (float)(acc[0][t][4 * gq + r] + zs[0][r]) * sxw[r];
          } else {                                 // T = (S1 << 14) + (S2 << 7) + S3 exactly, then ONE rounding chain
            long long T = 0;
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) T = (T << 7) + (long long)(acc[sl][t][4 * gq + r] + zs[sl][r]);
            v = (float)((double)T * (double)sxw[r]);
          }
          if (BIAS_M == 1 || (BIAS_M < 0 && fbias != nullptr)) v = v + bch[r];
          if (BN_M == 1 || (BN_M < 0 && has_bn)) {
            v = v * bsc[r];
            v = v + bsh[r];
          }
          if (!FOLD) v = ACT_M < 0 ? act_rt(v, act) : act_rt(v, ACT_M);
          vq[r] = v;
          if (!OUT16) {
            const unsigned off = partial ? (8 * gq + 4 * h + r < cvalid ? yo : 0x80000000u) : yo;
            buf_st_f32(yr, off, (unsigned)(8 * gq + r) * plane4, v);
          }
          m = FOLD ? fmaxf(m, v) : fmaxf(m, fabsf(v));
        }
        if (OUT16) {                               // (fq_pw_split_kernel.h: the consumer's codes of the four values, 4 bytes)
          const int packed = fq_pack4<NN2>(vq[0], vq[1], vq[2], vq[3], qc, ubias2, fq_nonneg_xor(ubias2));
          const bool blk_ok = !partial || 16 * (gq >> 1) < cvalid;
          const unsigned yo16 = (smp[t] - s_base) * (unsigned)g.CBo * HW * 16u + pp[t] * 16u + 4u * h;
          buf_st_f32(yr, blk_ok ? yo16 : 0x80000000u, (unsigned)((gq >> 1) * (int)HW * 16 + 8 * (gq & 1)), __int_as_float(packed));
        }
      }
      if (has_stat) {
        const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)smp[t]);
        if (__all(smp[t] == s0)) {
          const float wm = wave_max_nonneg(m);
          if (lane == 0) {
            const unsigned slot = s0 - s_base;
            if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
            else atomic_max_f32(stat_out + s0, wm);
          }
        } else {
          const unsigned slot = smp[t] - s_base;
          if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
          else atomic_max_f32(stat_out + smp[t], m);
        }
      }
    }
  };
  using std::integral_constant;
  if (cvalid <= 0) {
    // a channel group wider than the layer: this wavefront only helped to quantise the region
  } else {
    auto go = [&](auto bias_c, auto bn_c, auto act_c, bool nn2) __attribute__((always_inline)) {
      if constexpr (OUT16) {                                            // (only a kernel that writes codes is instantiated twice)
        if (nn2) run(bias_c, bn_c, act_c, std::true_type{});
        else run(bias_c, bn_c, act_c, std::false_type{});
      } else {
        run(bias_c, bn_c, act_c, std::false_type{});
      }
    };
    // (the five-instruction output quantiser where the clipped values cannot be negative: behind the ReLU, or a range from 0)
    if (fbias == nullptr && has_bn && act == FQ_ACT_RELU)
      go(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU>{}, q2.denom > 0.0f);
    else
      go(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{}, fq_nonneg(q2));
  }
  PW_STAMP(4);
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < cols / HW)
      FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
  PW_STAMP(5);
}


}  // namespace

#endif  // FQ_CONV3X3_KERNEL_H_
