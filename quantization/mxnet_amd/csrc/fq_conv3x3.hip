// libfakequant — K2n dense 3x3 convolution (stride 1, pad 1) on int8 codes: the 3x3 layers of the ResNet bottlenecks
// (see fq_common.h for the list of translation units and the design rules)
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
};

// NW wavefronts per workgroup (4, or 8 for wide layers with few pixel blocks: half as many channel groups quantise a region)
template <int KT, int PTW, int WC, int D, int LB, int NW, int NSL>
__global__ __launch_bounds__(NW * 64, LB) void conv3x3_i8_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, C3Geom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out) {
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
  const fq_rsrc xr = make_rsrc(reinterpret_cast<const char*>(x) + n_base * x_samp, (n_samp - n_base) * x_samp);

  // ---- 1. region -> LDS panel: units of (region tile of 32 pixels, slab of 32 channels), wave-strided -----------------
  const int NU = g.RT * KT;
  auto unit_off = [&](int u) __attribute__((always_inline)) {           // lane offset of this lane's pixel in unit u
    const int rt = u / KT;
    int jr = jr0 + rt * 32 + pl;
    jr = jr < 0 ? 0 : (jr < (int)cols ? jr : (int)cols - 1);            // outside the tensor: any valid pixel (masked later)
    const unsigned nr = (unsigned)jr / HW;
    return ((nr - n_base) * (unsigned)g.Cin + 16u * h) * plane4 + ((unsigned)jr - nr * HW) * 4u;
  };
  auto issue = [&](int u, float (&v)[16]) __attribute__((always_inline)) {
    const unsigned xo = unit_off(u);
    const int kt = u % KT;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = buf_ld_f32(xr, xo, (unsigned)(kt * 32 + i) * plane4);
  };
  float bufa[16], bufb[16];
  if (wave < NU) issue(wave, bufa);                                     // in flight during the set-up
  FQ_PIN();
  const float max_ = input_threshold(in_stat, n, in_thr, cur_max_out, item == 0);
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  const float sx = q.scale;
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;
  for (int i = threadIdx.x; i < NCH; i += NW * 64) {
    const bool ok = ch0 + i < g.Cout;
    const int ic = ok ? ch0 + i : 0;
    c_sxw[i] = ok ? sx * wscale[ic] : 0.0f;
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) c_zs[sl * NCH + i] = ok ? g.zoff * wsum[sl * g.slice_rows + ic] : 0;
    c_bias[i] = ok && bias != nullptr ? bias[ic] : 0.0f;
    c_bsc[i] = has_bn && ok ? bn_scale[ic] : (ok ? 1.0f : 0.0f);
    c_bsh[i] = has_bn && ok ? bn_shift[ic] : 0.0f;
  }
  const int ubias = 128 - g.zoff;
  const unsigned nn_xor = fq_nonneg_xor(ubias);
  auto quant_to_panel = [&](int u, const float (&v)[16], auto nn_c) __attribute__((always_inline)) {
    v4i f;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      f[d] = fq_pack4<decltype(nn_c)::value>(v[4 * d + 0], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3], q, ubias, nn_xor);
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
  if (fq_nonneg(q)) fill_panel(std::true_type{});
  else fill_panel(std::false_type{});

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
  __syncthreads();                                                      // panel, constants and the statistic table
  const int cvalid = g.Cout - (ch0 + wc * 32);                          // valid channels of this wavefront's tile
  const int zb = g.zoff ? (int)0x80808080u : 0;                         // four codes "0" in the stored representation
  auto run = [&](auto bias_c, auto bn_c, auto act_c) __attribute__((always_inline)) {
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
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
    // ---- 3. epilogue (K2m): lane = pixel, channels past Cout masked through out-of-range offsets ----------------------
    int64_t y_bytes = (n_samp - s_base) * y_samp - (int64_t)(ch0 + wc * 32) * plane4;
    y_bytes = y_bytes < 0x7FFFFFFFll ? y_bytes : 0x7FFFFFFFll;
    const fq_rsrc yr = make_rsrc(reinterpret_cast<char*>(y) + s_base * y_samp + (int64_t)(ch0 + wc * 32) * plane4, y_bytes);
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
        if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) bch = *reinterpret_cast<const f4*>(c_bias + c0);
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
          if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) v = v + bch[r];
          if (BN_M == 1 || (BN_M < 0 && has_bn)) {
            v = v * bsc[r];
            v = v + bsh[r];
          }
          v = ACT_M < 0 ? act_rt(v, act) : act_rt(v, ACT_M);
          const unsigned off = partial ? (8 * gq + 4 * h + r < cvalid ? yo : 0x80000000u) : yo;
          buf_st_f32(yr, off, (unsigned)(8 * gq + r) * plane4, v);
          m = fmaxf(m, fabsf(v));
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
  } else if (bias == nullptr && has_bn && act == FQ_ACT_RELU)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU>{});
  else
    run(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{});
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < cols / HW)
      atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
}

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

namespace {

int conv3x3_launch(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                   float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, const float* in_stat,
                   const float* in_thr, int in_width, unsigned in_flags, float* out_current_max, const float* bn_scale,
                   const float* bn_shift, int act, float* stat_out, fqStream_t stream, int nsl) {
  FQ_REQUIRE(x && wcodes && wscale && wsum && y, "fq_conv3x3_i8: null pointer");
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
  if (cout < 256 || !(kt == 8 || kt == 16) || nsl != 1) nw = 4;          // (the sliced form is built for four wavefronts)
  if (nsl != 1) FQ_REQUIRE(cout % 32 == 0, "fq_conv3x3_i8_sliced: Cout must be a multiple of 32, got %lld", (long long)cout);
  const int wc = nw == 8 ? 8 : (cout >= 128 ? 4 : 2);
  const int wp = nw / wc;
  const int64_t cs = (cout + 32 * wc - 1) / (32 * wc);
  // two pixel tiles per wavefront: measured best (or equal) on all four ResNet-50 stages in the model - 63 / 39 / 33 / 34 us
  // at 56x56 / 28x28 / 14x14 / 7x7 against 67 / 45 / 34 / 46 with four and 78 / 44 / 39 / 41 with one
  int ptw = 2;
  while (ptw > 1 && ((cols + 32 * ptw * wp - 1) / (32 * ptw * wp)) * cs < (int64_t)num_cu()) ptw >>= 1;
  const int tune = env_int("FQ_C3_PTW", 0);
  if (tune == 1 || tune == 2 || tune == 4) ptw = tune;
  const int pt = ptw * wp;
  C3Geom g;
  g.Cin = (int)cin; g.Cout = (int)cout; g.H = (int)h; g.W = (int)w; g.HW = (int)hw;
  g.CS = (int)cs; g.CTM = (int)(rows_pad / 32);
  g.RP = 32 * pt + 2 * (int)w + 2; g.RT = (g.RP + 31) / 32; g.ROW = (int)cin + 16;
  g.cols = cols; g.items = ((cols + 32 * pt - 1) / (32 * pt)) * cs; g.zoff = (in_flags & FQ_ACT_SIGNED) ? 0 : 128;
  g.slice_bytes = 2 * rows_pad * row_pad; g.slice_rows = (int)cout;
  FQ_REQUIRE(nsl == 1 || 3 * g.slice_bytes < (1ll << 31), "fq_conv3x3_i8_sliced: the three weight slices must lie within 2 GiB");
  const int64_t grid = (g.items + 7) / 8 * 8;
  FQ_REQUIRE(grid < (1ll << 31), "fq_conv3x3_i8: too many pixel blocks");
  const size_t lds = (size_t)g.RT * 32 * g.ROW + (size_t)(32 * wc) * (4 + nsl) * sizeof(float);
  FQ_REQUIRE(lds <= 150 * 1024, "fq_conv3x3_i8: the region of %d pixels x %d channels does not fit LDS", g.RP, (int)cin);
  const float levels = act_levels(in_width, in_flags);
  const int lo_neg = (in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  ProfScope prof(FQ_KERNEL_CONV3X3, 4.0 * ((double)n * cin * hw + (double)n * cout * hw), st);
  bool launched = false;
#define FQ_C3_CASE_NS(KT_, PTW_, WC_, D_, LB_, NW_, NSL_)                                                               \
  if (kt == KT_ && ptw == PTW_ && wc == WC_ && nw == NW_ && nsl == NSL_) {                                             \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_i8_kernel<KT_, PTW_, WC_, D_, LB_, NW_, NSL_>),     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;                     \
    FQ_REQUIRE(attr_ok, "fq_conv3x3_i8: cannot raise the dynamic LDS limit");                                          \
    hipLaunchKernelGGL((conv3x3_i8_kernel<KT_, PTW_, WC_, D_, LB_, NW_, NSL_>), dim3((unsigned)grid), dim3(NW_ * 64), lds, st, \
                       x,                                                                                              \
                       wfrag, wscale, (const int*)wsum, bias, y, g, in_stat, (int)n, in_thr, levels, lo_neg, kEps,     \
                       out_current_max, bn_scale, bn_shift, act, stat_out);                                            \
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
