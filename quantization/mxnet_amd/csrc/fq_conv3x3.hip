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
struct C3Geom {
  int Cin, Cout, H, W, HW;
  int CS;                    // channel groups per pixel block = ceil(Cout / (32 * WC))
  int CTM;                   // 32-channel tiles present in the weight buffer
  int RP, RT;                // region pixels, region tiles of 32
  int ROW;                   // bytes per region pixel in the panel
  int64_t cols, items;       // n * HW; pixel blocks * CS
  int zoff;
};

// NW wavefronts per workgroup (4, or 8 for wide layers with few pixel blocks: half as many channel groups quantise a region)
template <int KT, int PTW, int WC, int D, int LB, int NW>
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
  int* c_zs = reinterpret_cast<int*>(c_bias + NCH);

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
    c_zs[i] = ok ? g.zoff * wsum[ic] : 0;
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
  const fq_rsrc wr = make_rsrc(wfrag + (((int64_t)ctg * NS) << 10), ctg < g.CTM ? (int64_t)NS * 1024 : 0);
  const unsigned loff = (unsigned)lane * 16u;
  auto a_frag = [&](int s) __attribute__((always_inline)) { return buf_ld_v4i(wr, loff, (unsigned)(s << 10)); };
  v4i ring[RS];
#pragma unroll
  for (int d = 0; d < D; ++d) ring[d] = a_frag(d < NS ? d : NS - 1);
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
    v16i acc[PTW];
#pragma unroll
    for (int t = 0; t < PTW; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0;
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
        if (s + D < NS) ring[(s + D) % RS] = a_frag(s + D);
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
        for (int t = 0; t < PTW; ++t)
          acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ring[s % RS], b[t], acc[t], 0, 0, 0);
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
        const v4i zs = *reinterpret_cast<const v4i*>(c_zs + c0);
        const f4 sxw = *reinterpret_cast<const f4*>(c_sxw + c0);
        const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
        const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
        f4 bch = (f4){0.f, 0.f, 0.f, 0.f};
        if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) bch = *reinterpret_cast<const f4*>(c_bias + c0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = (float)(acc[t][4 * gq + r] + zs[r]) * sxw[r];
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

}  // namespace

using namespace fqi;

extern "C" {

int fq_conv3x3_i8(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                  float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, const float* in_stat,
                  const float* in_thr, int in_width, unsigned in_flags, float* out_current_max, const float* bn_scale,
                  const float* bn_shift, int act, float* stat_out, fqStream_t stream) {
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
  if (cout < 256 || !(kt == 8 || kt == 16)) nw = 4;
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
  const int64_t grid = (g.items + 7) / 8 * 8;
  FQ_REQUIRE(grid < (1ll << 31), "fq_conv3x3_i8: too many pixel blocks");
  const size_t lds = (size_t)g.RT * 32 * g.ROW + (size_t)(32 * wc) * 5 * sizeof(float);
  FQ_REQUIRE(lds <= 150 * 1024, "fq_conv3x3_i8: the region of %d pixels x %d channels does not fit LDS", g.RP, (int)cin);
  const float levels = act_levels(in_width, in_flags);
  const int lo_neg = (in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  ProfScope prof(FQ_KERNEL_CONV3X3, 4.0 * ((double)n * cin * hw + (double)n * cout * hw), st);
  bool launched = false;
#define FQ_C3_CASE_NW(KT_, PTW_, WC_, D_, LB_, NW_)                                                                    \
  if (kt == KT_ && ptw == PTW_ && wc == WC_ && nw == NW_) {                                                            \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_i8_kernel<KT_, PTW_, WC_, D_, LB_, NW_>),           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;                     \
    FQ_REQUIRE(attr_ok, "fq_conv3x3_i8: cannot raise the dynamic LDS limit");                                          \
    hipLaunchKernelGGL((conv3x3_i8_kernel<KT_, PTW_, WC_, D_, LB_, NW_>), dim3((unsigned)grid), dim3(NW_ * 64), lds, st, \
                       x,                                                                                              \
                       wfrag, wscale, (const int*)wsum, bias, y, g, in_stat, (int)n, in_thr, levels, lo_neg, kEps,     \
                       out_current_max, bn_scale, bn_shift, act, stat_out);                                            \
    launched = true;                                                                                                   \
  }
#define FQ_C3_CASE(KT_, PTW_, WC_, D_, LB_) FQ_C3_CASE_NW(KT_, PTW_, WC_, D_, LB_, 4)
#define FQ_C3_KT(KT_)                                                                                                  \
  FQ_C3_CASE(KT_, 1, 4, 6, 4) FQ_C3_CASE(KT_, 2, 4, 4, 4) FQ_C3_CASE(KT_, 4, 4, 3, 3)                                  \
  FQ_C3_CASE(KT_, 1, 2, 6, 4) FQ_C3_CASE(KT_, 2, 2, 4, 4) FQ_C3_CASE(KT_, 4, 2, 3, 3)
  FQ_C3_KT(2) FQ_C3_KT(4) FQ_C3_KT(8) FQ_C3_KT(16)
  FQ_C3_CASE_NW(8, 1, 8, 6, 4, 8) FQ_C3_CASE_NW(8, 2, 8, 4, 4, 8) FQ_C3_CASE_NW(16, 1, 8, 6, 4, 8) FQ_C3_CASE_NW(16, 2, 8, 4, 4, 8)
#undef FQ_C3_KT
#undef FQ_C3_CASE
#undef FQ_C3_CASE_NW
  FQ_REQUIRE(launched, "fq_conv3x3_i8: no instantiation for K/32=%d, %d pixel tiles per wavefront, %d channel tiles per "
             "workgroup", kt, ptw, wc);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // extern "C"
