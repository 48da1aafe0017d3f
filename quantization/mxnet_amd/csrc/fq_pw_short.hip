// libfakequant — K2s: the closing 1x1 convolution of a residual unit WITH the unit's shortcut convolution in the same launch
// (fq_pwconv_i8_shortcut; see fq_common.h for the list of translation units and the design rules)
#include "fq_pw.h"

namespace {

// K2s (round 6).  The first unit of a ResNet-v1 stage ends in  act(BN3(conv3(y2)) + BNd(convd(x)))  - gluon's BottleneckV1 with
// a `downsample` branch: convd is a 1x1 convolution of the unit's input x, BNd its BatchNorm, and the sum is the trunk.  Run as
// two launches the shortcut tensor s = BNd(convd(x)) - as large as the trunk: 411 / 205 / 103 / 51 MB at batch 128 - is written
// by one and read back as the other's residual operand.  It has no other reader and nobody needs its statistic (no quantised
// block is fed by it).  Here the closing convolution's workgroup computes BOTH integer sums for its (pixel tile, channel group):
//   1. the four wavefronts quantise the tile's slabs of y2 (threshold of conv3's activation branch) AND of x (threshold of
//      convd's) into two LDS panels of B fragments - the split form's step 1, twice;
//   2. per 32-channel tile: acc_d = W_d . panel_x, s = BNd(fp32(acc_d + zs_d) * sxw_d) kept in 16 registers, then
//      acc = W_3 . panel_y2 and the epilogue of the split form with s in the place of the residual operand's loads.
// The same fp32 operations in the same order as the two launches: bit-equal.  4 (y2 + x) + 4 z bytes instead of
// 4 x + 4 s | 4 y2 + 4 s + 4 z.  Work item -> (tile, group) and the XCD order as in the split form (fq_pw_split_kernel.h).
struct PwShortGeom {
  int Cin, Cin2, Cout, HW, CS;
  int CTM, CTM2;             // 32-channel tiles present in the two weight buffers
  int64_t cols, tiles, items;
  int zoff, zoff2;
  // C16 code tensors (offline thresholds; the layout: include/fakequant.h): A16 - x holds this convolution's codes, CBi blocks of
  // 16 channels per sample; B16 - the shortcut convolution's input likewise (CBi2); DUAL - g.y16 receives the codes of y under
  // dual_thr (CBo blocks; out_levels / out_lo_neg / out_zoff describe that quantiser), as fq_pwconv_i8_c16_dual writes them
  int CBi, CBi2, CBo;
  float out_levels;
  int out_lo_neg, out_zoff;
  char* y16;
  const float* dual_thr;
  int nts;                   // nontemporal stores of y (an output the Infinity Cache cannot hold until its readers start)
};

// the shortcut convolution's operands (the closing convolution's travel as plain kernel arguments, as in the split form)
struct PwShortIn2 {
  const float* x;
  const int8_t* wfrag;
  const float* wscale;
  const int* wsum;
  const float* bn_scale;
  const float* bn_shift;
  const float* in_stat;
  const float* in_thr;
  float* cur_max_out;
  float levels;
  int lo_neg;
};

constexpr int kShCW = 2;                                               // channel tiles per wavefront
constexpr int kShD = 4;                                                // A fragments in flight per wavefront

// NW wavefronts per workgroup: 4 (256 output channels) or 8 (512: half as many workgroups quantise the same tile's slabs)
template <int KT, int KT2, int NW, bool A16 = false, bool B16 = false, bool DUAL = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 3) void pwconv_short_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwShortGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out, PwShortIn2 s2) {
  constexpr int kSlots = 8;
  constexpr int kShNW = NW;
  constexpr int NCH = NW * kShCW * 32;
  constexpr int NS = KT + KT2;                                          // slabs of the two inputs together
  constexpr int SLABS = (NS + kShNW - 1) / kShNW;                       // ... a wavefront quantises (s = wave + 4 j < NS)
  constexpr int RB = SLABS < 4 ? SLABS : 4;                             // slabs (16 loads each) in flight per lane
  extern __shared__ __attribute__((aligned(16))) unsigned char pwsh_smem[];
  __shared__ unsigned k_stat[kSlots];
  v4i* panel = reinterpret_cast<v4i*>(pwsh_smem);                       // [KT][64] B fragments of y2's tile
  v4i* panel2 = panel + KT * 64;                                        // [KT2][64] ... of x's tile
  float* c_sxw = reinterpret_cast<float*>(pwsh_smem + (size_t)NS * 1024);
  float* c_bsc = c_sxw + NCH;
  float* c_bsh = c_bsc + NCH;
  float* c_bias = c_bsh + NCH;
  float* c_sxw2 = c_bias + NCH;
  float* c_bsc2 = c_sxw2 + NCH;
  float* c_bsh2 = c_bsc2 + NCH;
  int* c_zs = reinterpret_cast<int*>(c_bsh2 + NCH);
  int* c_zs2 = c_zs + NCH;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pl = lane & 31;
  const unsigned HW = (unsigned)g.HW, cols = (unsigned)g.cols;
  const unsigned plane4 = HW * 4u;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  unsigned item;
  {
    const unsigned per = ((unsigned)g.items + 7u) >> 3;
    item = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || item >= (unsigned)g.items) return;
  }
  const unsigned tile = item / (unsigned)g.CS, cg = item - tile * (unsigned)g.CS;
  const int ch0 = (int)cg * NCH;
  const unsigned s_base = (tile * 32u) / HW;                            // first sample the tile touches
  unsigned smp, p;
  {
    unsigned j = tile * 32u + (unsigned)pl;
    j = j < cols ? j : cols - 1;                                        // lanes past the end copy the last pixel
    smp = j / HW;
    p = j - smp * HW;
  }
  const int64_t n_samp = (int64_t)(cols / HW);
  const int64_t x_samp = (int64_t)g.Cin * HW * 4, x2_samp = (int64_t)g.Cin2 * HW * 4, y_samp = (int64_t)g.Cout * HW * 4;
  const int64_t x_samp16 = (int64_t)g.CBi * HW * 16, x2_samp16 = (int64_t)g.CBi2 * HW * 16;
  const fq_rsrc xr = A16 ? make_rsrc(reinterpret_cast<const char*>(x) + s_base * x_samp16, (n_samp - s_base) * x_samp16)
                         : make_rsrc(reinterpret_cast<const char*>(x) + s_base * x_samp, (n_samp - s_base) * x_samp);
  const fq_rsrc xr2 = B16 ? make_rsrc(reinterpret_cast<const char*>(s2.x) + s_base * x2_samp16, (n_samp - s_base) * x2_samp16)
                          : make_rsrc(reinterpret_cast<const char*>(s2.x) + s_base * x2_samp, (n_samp - s_base) * x2_samp);
  // (a C16 input: the lane's 16 codes of a slab's half are ONE 16-byte vector of block 2 kt + h - already the B fragment)
  const unsigned xo = A16 ? ((smp - s_base) * (unsigned)g.CBi + (unsigned)h) * HW * 16u + p * 16u
                          : ((smp - s_base) * (unsigned)g.Cin + 16u * h) * plane4 + p * 4u;
  const unsigned xo2 = B16 ? ((smp - s_base) * (unsigned)g.CBi2 + (unsigned)h) * HW * 16u + p * 16u
                           : ((smp - s_base) * (unsigned)g.Cin2 + 16u * h) * plane4 + p * 4u;
  // slab s of the two inputs together: s < KT -> slab s of y2, else slab s - KT of x (wave-uniform)
  auto issue = [&](int s, float (&v)[16]) __attribute__((always_inline)) {
    if (s < KT) {
      if (A16) {
        const v4i c = buf_ld_v4i(xr, xo, (unsigned)(2 * s) * HW * 16u);
        v[0] = __int_as_float(c[0]); v[1] = __int_as_float(c[1]); v[2] = __int_as_float(c[2]); v[3] = __int_as_float(c[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = buf_ld_f32(xr, xo, (unsigned)(s * 32 + i) * plane4);
      }
    } else {
      if (B16) {
        const v4i c = buf_ld_v4i(xr2, xo2, (unsigned)(2 * (s - KT)) * HW * 16u);
        v[0] = __int_as_float(c[0]); v[1] = __int_as_float(c[1]); v[2] = __int_as_float(c[2]); v[3] = __int_as_float(c[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = buf_ld_f32(xr2, xo2, (unsigned)((s - KT) * 32 + i) * plane4);
      }
    }
  };
  const ThresholdReq treq = threshold_request(in_stat, n, in_thr, item == 0);        // first in the memory queue
  const ThresholdReq treq2 = threshold_request(s2.in_stat, n, s2.in_thr, item == 0);
  float buf[RB][16];
#pragma unroll
  for (int i = 0; i < RB; ++i)
    if (wave + kShNW * i < NS) issue(wave + kShNW * i, buf[i]);
  FQ_PIN();
  const float max_ = threshold_finish(treq, in_stat, n, in_thr, cur_max_out, item == 0);
  const float max2_ = threshold_finish(treq2, s2.in_stat, n, s2.in_thr, s2.cur_max_out, item == 0);
  const int zoff = g.zoff, zoff2 = g.zoff2;
  const QParams q = make_qparams_rt(max_, levels, lo_neg_max, eps, in_thr);
  const QParams q2 = make_qparams_rt(max2_, s2.levels, s2.lo_neg, eps, s2.in_thr);
  const float sx = q.scale, sx2 = q2.scale;
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;
  for (int i = threadIdx.x; i < NCH; i += kShNW * 64) {
    const int ic = ch0 + i;                                             // (host: Cout is a multiple of 256)
    c_sxw[i] = sx * wscale[ic];
    c_zs[i] = zoff * wsum[ic];
    c_bias[i] = bias != nullptr ? bias[ic] : 0.0f;
    c_bsc[i] = has_bn ? bn_scale[ic] : 1.0f;
    c_bsh[i] = has_bn ? bn_shift[ic] : 0.0f;
    c_sxw2[i] = sx2 * s2.wscale[ic];
    c_zs2[i] = zoff2 * s2.wsum[ic];
    c_bsc2[i] = s2.bn_scale[ic];
    c_bsh2[i] = s2.bn_shift[ic];
  }
  const int ubias = 128 - zoff, ubias2 = 128 - zoff2;
  const unsigned nn_xor = fq_nonneg_xor(ubias), nn_xor2 = fq_nonneg_xor(ubias2);
  const bool nn1 = fq_nonneg(q), nn2 = fq_nonneg(q2);
  auto quant_to_panel = [&](int s, const float (&v)[16]) __attribute__((always_inline)) {
    v4i f;
    const bool first = s < KT;                                          // (wave-uniform, as are nn1 / nn2: four branches)
    if ((first && A16) || (!first && B16)) {                            // codes: they ARE the fragment
      f = (v4i){__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])};
    } else if (first && nn1) {
#pragma unroll
      for (int d = 0; d < 4; ++d) f[d] = fq_pack4<true>(v[4 * d + 0], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3], q, ubias, nn_xor);
    } else if (first) {
#pragma unroll
      for (int d = 0; d < 4; ++d) f[d] = fq_pack4<false>(v[4 * d + 0], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3], q, ubias, nn_xor);
    } else if (nn2) {
#pragma unroll
      for (int d = 0; d < 4; ++d) f[d] = fq_pack4<true>(v[4 * d + 0], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3], q2, ubias2, nn_xor2);
    } else {
#pragma unroll
      for (int d = 0; d < 4; ++d) f[d] = fq_pack4<false>(v[4 * d + 0], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3], q2, ubias2, nn_xor2);
    }
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));   // pin the arithmetic here (see K2h)
    if (first) panel[(s << 6) + lane] = f;
    else panel2[((s - KT) << 6) + lane] = f;
  };
  // ---- 1. my share of the slabs of both inputs -> the two LDS panels ---------------------------------------------------------
#pragma unroll
  for (int j = 0; j < SLABS; ++j) {
    if (wave + kShNW * j < NS) quant_to_panel(wave + kShNW * j, buf[j % RB]);
    FQ_PIN();
    if (j + RB < SLABS) {
      if (wave + kShNW * (j + RB) < NS) issue(wave + kShNW * (j + RB), buf[j % RB]);
      FQ_PIN();
    }
  }
  // ---- 2. per channel tile: the shortcut's sum, then the closing convolution's; A fragments through a ring of kShD ----------
  const int ctl0 = wave * kShCW;                                        // first channel tile inside the workgroup
  const int ctg0 = (int)cg * kShNW * kShCW + ctl0;                      // ... and in the layer
  const int ct_here = g.CTM - ctg0 < kShCW ? (g.CTM - ctg0 < 0 ? 0 : g.CTM - ctg0) : kShCW;
  const int ct2_here = g.CTM2 - ctg0 < kShCW ? (g.CTM2 - ctg0 < 0 ? 0 : g.CTM2 - ctg0) : kShCW;
  const fq_rsrc wr = make_rsrc(wfrag + (((int64_t)ctg0 * KT) << 10), (int64_t)ct_here * KT * 1024);
  const fq_rsrc wr2 = make_rsrc(s2.wfrag + (((int64_t)ctg0 * KT2) << 10), (int64_t)ct2_here * KT2 * 1024);
  const unsigned loff = (unsigned)lane * 16u;
  // fragment j of channel tile c's sequence: the KT2 fragments of W_d, then the KT fragments of W_3
  auto frag = [&](int c, int j) __attribute__((always_inline)) {
    return j < KT2 ? buf_ld_v4i(wr2, loff, (unsigned)((c * KT2 + j) << 10)) : buf_ld_v4i(wr, loff, (unsigned)((c * KT + (j - KT2)) << 10));
  };
  constexpr int L = KT2 + KT;
  constexpr int DD = kShD < L ? kShD : L;
  static_assert(L % DD == 0, "the ring's slots must line up across channel tiles");
  v4i ring[DD];
#pragma unroll
  for (int d = 0; d < DD; ++d) ring[d] = frag(0, d);
  FQ_PIN();
  __syncthreads();                                                      // panels, constants and the statistic table
  int64_t y_bytes = (n_samp - s_base) * y_samp - (int64_t)(ch0 + ctl0 * 32) * plane4;
  y_bytes = y_bytes < 0x7FFFFFFFll ? y_bytes : 0x7FFFFFFFll;
  const fq_rsrc yr = make_rsrc(reinterpret_cast<char*>(y) + s_base * y_samp + (int64_t)(ch0 + ctl0 * 32) * plane4, y_bytes);
  const unsigned yo = ((smp - s_base) * (unsigned)g.Cout + 4u * h) * plane4 + p * 4u;
  // DUAL: the codes of y under the next unit's first convolution's threshold, beside y (unsigned codes of a [0, thr] range)
  QParams q3;
  q3.lo = q3.hi = q3.denom = q3.scale = 0.0f;
  q3.rden = 0.0;
  if (DUAL) q3 = make_qparams(g.dual_thr[0], g.out_levels, g.out_lo_neg != 0, eps);
  const int ubias3 = 128 - g.out_zoff;
  const int64_t y_samp16 = (int64_t)g.CBo * HW * 16;
  const int cb0 = (ch0 + ctl0 * 32) >> 4;                               // first output block of this wavefront
  int64_t y16_bytes = (n_samp - s_base) * y_samp16 - (int64_t)cb0 * HW * 16;
  y16_bytes = y16_bytes < 0x7FFFFFFFll ? y16_bytes : 0x7FFFFFFFll;
  const fq_rsrc yr16 = make_rsrc(DUAL ? g.y16 + s_base * y_samp16 + (int64_t)cb0 * HW * 16 : reinterpret_cast<char*>(y),
                                 DUAL ? y16_bytes : 0);
  const unsigned yo16 = (smp - s_base) * (unsigned)g.CBo * HW * 16u + p * 16u + 4u * h;
  float m = 0.0f;
#pragma unroll
  for (int c = 0; c < kShCW; ++c) {
    v16i acc2, acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[i] = acc[i] = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const v4i a = ring[j % DD];
      // the next fragment of this tile's sequence - or the first ones of the next tile's
      if (j + DD < L) ring[j % DD] = frag(c, j + DD);
      else if (c + 1 < kShCW) ring[j % DD] = frag(c + 1, j + DD - L);
      if (j < KT2) acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, panel2[(j << 6) + lane], acc2, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, panel[((j - KT2) << 6) + lane], acc, 0, 0, 0);
      FQ_PIN();
    }
    const int cb = (ctl0 + c) * 32 + 4 * h;                             // channel inside the workgroup's group
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int c0 = cb + 8 * gq;
      const v4i zs = *reinterpret_cast<const v4i*>(c_zs + c0);
      const f4 sxw = *reinterpret_cast<const f4*>(c_sxw + c0);
      const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
      const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
      const f4 bch = *reinterpret_cast<const f4*>(c_bias + c0);
      const v4i zs2 = *reinterpret_cast<const v4i*>(c_zs2 + c0);
      const f4 sxw2 = *reinterpret_cast<const f4*>(c_sxw2 + c0);
      const f4 bsc2 = *reinterpret_cast<const f4*>(c_bsc2 + c0);
      const f4 bsh2 = *reinterpret_cast<const f4*>(c_bsh2 + c0);
      float vq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // the shortcut's value, as fq_pwconv_i8_strided computes and stores it: fp32(sum) * (sx * sw), BatchNorm, no activation
        float s = (float)(acc2[4 * gq + r] + zs2[r]) * sxw2[r];
        s = s * bsc2[r];
        s = s + bsh2[r];
        // the closing convolution's, with s where the residual operand's load was
        float v = (float)(acc[4 * gq + r] + zs[r]) * sxw[r];
        if (bias != nullptr) v = v + bch[r];
        if (has_bn) {
          v = v * bsc[r];
          v = v + bsh[r];
        }
        v = v + s;
        v = act_rt(v, act);
        if (g.nts) buf_st_f32_nt(yr, yo, (unsigned)(c * 32 + 8 * gq + r) * plane4, v);
        else buf_st_f32(yr, yo, (unsigned)(c * 32 + 8 * gq + r) * plane4, v);
        m = fmaxf(m, fabsf(v));
        vq[r] = v;
      }
      if (DUAL) {
        // channels 8 gq + 4 h .. + 3 of this lane's pixel are bytes 8 (gq & 1) + 4 h .. of block (c * 2 + gq / 2): one 4-byte store
        const int packed = fq_pack4<true>(vq[0], vq[1], vq[2], vq[3], q3, ubias3, 0x80808080u);
        buf_st_f32(yr16, yo16, (unsigned)((c * 2 + (gq >> 1)) * (int)HW * 16 + 8 * (gq & 1)), __int_as_float(packed));
      }
    }
  }
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
    __syncthreads();
    if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < cols / HW)
      FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
}

}  // namespace

namespace fqi {

// shapes fq_pwconv_i8_shortcut takes: the four stage heads of the v1 bottleneck ResNets (K of the closing convolution / K of the
// shortcut convolution: 64 / 64, 128 / 256, 256 / 512, 512 / 1024), Cout a multiple of 512 (of 256 for 64 / 64).
// (512 / 1024 -> 2048 @7x7 with FOUR wavefronts - eight channel groups quantise the same 48 slabs of a 32-pixel tile - was 89 us
// against 34 + 31 for the two launches; with eight wavefronts it is worth +0.24 % (sd 0.08) of ResNet-50's images/s.)
bool pw_short_shape_ok(int64_t cin, int64_t cin2, int64_t cout) {
  const bool pair = (cin == 64 && cin2 == 64) || (cin == 128 && cin2 == 256) || (cin == 256 && cin2 == 512) || (cin == 512 && cin2 == 1024);
  return pair && cout > 0 && cout % (cin == 64 ? 256 : 512) == 0;
}

int pw_short_launch(const PwCall& a, const PwCall& b) {
  const int kt = (int)(a.cin_pad / 32), kt2 = (int)(b.cin_pad / 32);
  FQ_REQUIRE(pw_short_shape_ok(a.cin, b.cin, a.cout) && a.cin == a.cin_pad && b.cin == b.cin_pad,
             "fq_pwconv_i8_shortcut: shape not taken (cin=%lld, shortcut cin=%lld, cout=%lld): see fq_pwconv_i8_shortcut_supported",
             (long long)a.cin, (long long)b.cin, (long long)a.cout);
  const int64_t tiles = (a.n * a.hw + 31) / 32;
  FQ_REQUIRE((32 / a.hw + 2) * a.cout * a.hw * 4 < (1ll << 31) && (32 / a.hw + 2) * b.cin * a.hw * 4 < (1ll << 31),
             "fq_pwconv_i8_shortcut: a tile must stay within 2 GiB of its first sample");
  static const int nw_tune = env_int("FQ_PWSH_NW", 0);                   // tuning: 4 / 8 wavefronts per workgroup
  const int nw = (nw_tune == 4 || nw_tune == 8) ? (a.cout % 512 == 0 ? nw_tune : 4) : (a.cout % 512 == 0 && kt >= 4 ? 8 : 4);
  PwShortGeom t;
  t.Cin = (int)a.cin; t.Cin2 = (int)b.cin; t.Cout = (int)a.cout; t.HW = (int)a.hw;
  t.CS = (int)(a.cout / (nw * kShCW * 32));
  t.CTM = (int)((a.cout + 63) / 64 * 64 / 32); t.CTM2 = t.CTM;
  t.cols = a.n * a.hw; t.tiles = tiles; t.items = tiles * t.CS;
  t.zoff = a.zoff; t.zoff2 = b.zoff;
  t.CBi = (int)((a.cin + 15) / 16); t.CBi2 = (int)((b.cin + 15) / 16); t.CBo = (int)((a.cout + 15) / 16);
  t.out_levels = a.out_levels; t.out_lo_neg = a.out_lo_neg; t.out_zoff = a.out_zoff;
  t.y16 = (char*)a.y16; t.dual_thr = a.dual_thr;
  static const int nts_mb = env_int("FQ_PWSH_NTS_MB", 150);              // (the streaming form's policy, FQ_PWS_NTS_MB)
  t.nts = 4e-6 * (double)a.n * a.cout * a.hw >= nts_mb ? 1 : 0;
  const bool a16 = a.in_c16, b16 = b.in_c16, dual = a.y16 != nullptr;
  FQ_REQUIRE((!a16 && !b16 && !dual) || (a16 && dual), "fq_pwconv_i8_shortcut_c16: built for fp32 on every side, or for codes in + "
             "the code copy out (the shortcut convolution's input fp32 or codes)");
  FQ_REQUIRE(!a16 || a.in_thr != nullptr, "fq_pwconv_i8_shortcut_c16: a C16 input was quantised with a stored threshold: give in_thr");
  FQ_REQUIRE(!b16 || b.in_thr != nullptr, "fq_pwconv_i8_shortcut_c16: a C16 shortcut input needs in_thr2");
  const int64_t grid = (t.items + 7) / 8 * 8;
  FQ_REQUIRE(grid < (1ll << 31), "fq_pwconv_i8_shortcut: too many tiles");
  const size_t lds = (size_t)(kt + kt2) * 1024 + (size_t)(nw * kShCW * 32) * 9 * sizeof(float);
  const int64_t rows_pad = (a.cout + 63) / 64 * 64;
  const int8_t* wfrag = a.wcodes + rows_pad * a.cin_pad;                 // second halves of fq_weight_codes' buffers
  PwShortIn2 s2;
  s2.x = b.x; s2.wfrag = b.wcodes + rows_pad * b.cin_pad; s2.wscale = b.wscale; s2.wsum = (const int*)b.wsum;
  s2.bn_scale = b.bn_scale; s2.bn_shift = b.bn_shift; s2.in_stat = b.in_stat; s2.in_thr = b.in_thr;
  s2.cur_max_out = b.out_current_max; s2.levels = b.levels; s2.lo_neg = b.lo_neg;
  if (int rc = pw_zero_stat(a)) return rc;
  bool launched = false;
#define FQ_PWSH_CASE(KT_, KT2_, NW_) FQ_PWSH_CASE_C(KT_, KT2_, NW_, false, false, false)
#define FQ_PWSH_CASE_C(KT_, KT2_, NW_, A_, B_, D_)                                                                     \
  if (kt == KT_ && kt2 == KT2_ && nw == NW_ && a16 == A_ && b16 == B_ && dual == D_) {                                 \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_short_kernel<KT_, KT2_, NW_, A_, B_, D_>),           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;                      \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8_shortcut: cannot raise the dynamic LDS limit");                                  \
    hipLaunchKernelGGL((pwconv_short_kernel<KT_, KT2_, NW_, A_, B_, D_>), dim3((unsigned)grid), dim3(NW_ * 64), lds, a.st, a.x, wfrag, \
                       a.wscale, (const int*)a.wsum, a.bias, a.y, t, a.in_stat, (int)a.n, a.in_thr, a.levels, a.lo_neg, \
                       kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out, s2);                        \
    launched = true;                                                                                                   \
  }
  FQ_PWSH_CASE(2, 2, 4) FQ_PWSH_CASE(4, 8, 4) FQ_PWSH_CASE(8, 16, 4) FQ_PWSH_CASE(4, 8, 8) FQ_PWSH_CASE(8, 16, 8)
  FQ_PWSH_CASE(16, 32, 8)
  // stored thresholds: codes in, fp32 + code copy out; the shortcut's input fp32 (stage 1: the pooled first convolution) or codes
  FQ_PWSH_CASE_C(2, 2, 4, true, false, true) FQ_PWSH_CASE_C(4, 8, 8, true, false, true) FQ_PWSH_CASE_C(8, 16, 8, true, false, true)
  FQ_PWSH_CASE_C(4, 8, 8, true, true, true) FQ_PWSH_CASE_C(8, 16, 8, true, true, true)
  FQ_PWSH_CASE_C(16, 32, 8, true, false, true) FQ_PWSH_CASE_C(16, 32, 8, true, true, true)
#undef FQ_PWSH_CASE_C
#undef FQ_PWSH_CASE
  FQ_REQUIRE(launched, "fq_pwconv_i8_shortcut: no instantiation for K/32 = %d and %d", kt, kt2);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // namespace fqi
