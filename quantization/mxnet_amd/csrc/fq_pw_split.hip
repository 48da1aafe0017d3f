// libfakequant — K2m pointwise (1x1) convolution on int8 codes for the deep layers: one (pixel tile, channel group) per
// workgroup (see fq_common.h for the list of translation units and the design rules)
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
};

// LB: wavefronts per SIMD the register allocation aims at (3: <= 168 registers, 4: <= 128) - with one tile per workgroup the
// whole grid should be resident at once (a second round of a few left-over workgroups costs a whole workgroup latency)
// NW: wavefronts per workgroup.  4: 128 * CW output channels per workgroup; 8 (wide layers): 256 * CW - half as many channel
// groups quantise the same tile, i.e. half the redundant loads and quantiser VALU.
template <int KT, int CW, int D, int LB, int NW>
__global__ __launch_bounds__(NW * 64, LB) void pwconv_split_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwSplitGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out, const float* __restrict__ residual) {
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
  const fq_rsrc xr = make_rsrc(reinterpret_cast<const char*>(x) + s_base * x_samp, (n_samp - s_base) * x_samp);
  unsigned p_in = p;
  if (g.S != 1) {
    const unsigned ho = p / (unsigned)g.Wo, wo = p - ho * (unsigned)g.Wo;
    p_in = ho * (unsigned)(g.S * g.Win) + wo * (unsigned)g.S;
  }
  const unsigned xo = ((smp - s_base) * (unsigned)g.Cin + 16u * h) * plane4_in + p_in * 4u;
  auto issue = [&](int kt, float (&v)[16]) __attribute__((always_inline)) {
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
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  const float sx = q.scale;
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;
  for (int i = threadIdx.x; i < NCH; i += NW * 64) {
    const bool ok = ch0 + i < g.Cout;                                   // channels past Cout: all-zero constants
    const int ic = ok ? ch0 + i : 0;
    c_sxw[i] = ok ? sx * wscale[ic] : 0.0f;
    c_zs[i] = ok ? g.zoff * wsum[ic] : 0;
    c_bias[i] = ok && bias != nullptr ? bias[ic] : 0.0f;
    c_bsc[i] = has_bn && ok ? bn_scale[ic] : (ok ? 1.0f : 0.0f);
    c_bsh[i] = has_bn && ok ? bn_shift[ic] : 0.0f;
  }
  PW_STAMP(1);
  const int ubias = 128 - g.zoff;
  const unsigned nn_xor = fq_nonneg_xor(ubias);
  auto quant_to_panel = [&](int kt, const float (&v)[16], auto nn_c) __attribute__((always_inline)) {
    v4i f;
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
  auto run = [&](auto bias_c, auto bn_c, auto act_c) __attribute__((always_inline)) {
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
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
    const fq_rsrc yr = make_rsrc(reinterpret_cast<char*>(y) + s_base * y_samp + (int64_t)(ch0 + ctl0 * 32) * plane4, y_bytes);
    // the residual operand (the shortcut of a ResNet / MobileNetV2 unit) has y's shape: same offsets, added after BatchNorm
    const bool has_res = residual != nullptr;
    const fq_rsrc rr = make_rsrc(reinterpret_cast<const char*>(has_res ? residual : y) + s_base * y_samp +
                                 (int64_t)(ch0 + ctl0 * 32) * plane4, has_res ? y_bytes : 0);
    const unsigned yo = ((smp - s_base) * (unsigned)g.Cout + 4u * h) * plane4 + p * 4u;
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
        if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) bch = *reinterpret_cast<const f4*>(c_bias + c0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = (float)(acc[c][4 * gq + r] + zs[r]) * sxw[r];
          if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) v = v + bch[r];
          if (BN_M == 1 || (BN_M < 0 && has_bn)) {
            v = v * bsc[r];
            v = v + bsh[r];
          }
          if (has_res) v = v + res[4 * gq + r];
          v = ACT_M < 0 ? act_rt(v, act) : act_rt(v, ACT_M);
          const unsigned off = MASKED ? (8 * gq + 4 * h + r < cv ? yo : 0x80000000u) : yo;
          buf_st_f32(yr, off, (unsigned)(c * 32 + 8 * gq + r) * plane4, v);
          m = fmaxf(m, fabsf(v));                  // channels past Cout have all-zero constants: v == 0
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
  if (cvalid <= 0) {
    // nothing to multiply (a channel group wider than the layer): this wavefront only helped to quantise the tile
  } else if (bias == nullptr && has_bn && act == FQ_ACT_RELU)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU>{});
  else if (bias == nullptr && has_bn && act == FQ_ACT_RELU6)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU6>{});
  else if (bias == nullptr && has_bn && act == FQ_ACT_NONE)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_NONE>{});
  else
    run(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{});
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < cols / HW)
      atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
  PW_STAMP(5);
}

}  // namespace

namespace fqi {

// split form (K2m): grid = tiles x channel groups.  Takes every Cout and every Cin whose padded row length (cin_pad, the
// row stride fq_weight_codes was given) is one of the instantiated K/32 below: the MobileNet, MobileNetV2 and ResNet-50
// channel counts.  Measured against the other forms on every such shape (tools/pwforms.py, tools/kprof.sh;
// profiles/r2_pw_split.txt): faster everywhere except on the largest planes (> 4096 tiles of 32 pixels), where the streaming
// form's persistent wavefronts with LDS-resident weights win by a few per cent - so with more tiles than that it only
// takes what the streaming form cannot.
int pw_try_split(const PwCall& a, bool* taken) {
  *taken = false;
  const int kt = (int)(a.cin_pad / 32);
  const int64_t tiles = (a.n * a.hw + 31) / 32;
  const int64_t rows_pad = (a.cout + 63) / 64 * 64;
  static const int kts[] = {2, 4, 6, 8, 10, 12, 16, 18, 30, 32, 64};
  bool shape_ok = false;
  for (int k : kts) shape_ok = shape_ok || k == kt;
  // a tile's stores must stay below 2 GiB from its first sample (out-of-range lane offsets mask partial channel tiles)
  const int64_t hw_in = a.stride == 1 ? a.hw : a.h_in * a.w_in;
  shape_ok = shape_ok && (32 / a.hw + 2) * a.cout * a.hw * 4 < (1ll << 31) && a.cin * hw_in * 4 * (32 / a.hw + 2) < (1ll << 31);
  static const int mode = env_int("FQ_PWS_AUTO", 1);                    // tuning: 0 never, 1 by shape, 2 always
  bool want = a.form == 6 || a.stride != 1;                             // (only this form reads strided inputs)
  if (a.form == 0 && shape_ok && a.stride == 1)
    want = mode == 2 || (mode == 1 && (tiles <= (int64_t)num_cu() * 16 || !pw_stream_shape_ok(a)));
  if (want && shape_ok) {
    // channel tiles per wavefront (cw) and wavefronts per SIMD (lb).  Measured in the model: two tiles per wavefront at
    // four wavefronts per SIMD is the best or within 3 % of it on every shape (the grid then fills the chip in one or two
    // resident rounds); narrow layers take one tile per wavefront so that all four wavefronts have channels.
    int cw = a.cout > 128 ? 2 : 1;
    int lb = 4;
    // eight wavefronts per workgroup for wide layers with FEW tiles (7x7 planes: 196): a tile is then quantised by half as
    // many channel groups (measured in the model: 1024 -> 1024 @7x7 31.2 -> 25.8 us, 512 -> 1024 @7x7 19.8 -> 18.5; the 14x14
    // layers with their 784 tiles get 5 % slower)
    static const int nw_tune = env_int("FQ_PWS_NW", 0);
    const bool nw8_built = kt == 8 || kt == 16 || kt == 32 || kt == 64;
    int nw = (nw_tune == 4 || nw_tune == 8) ? nw_tune : ((a.cout >= 512 && tiles <= (int64_t)num_cu()) ? 8 : 4);
    if (!nw8_built) nw = 4;
    const int tune = env_int("FQ_PWS_CFG", 0);                           // tuning: 10 * lb + cw, read per call
    if (tune > 0) {
      cw = tune % 10;
      lb = tune / 10;
    }
    PwSplitGeom t;
    t.Cin = (int)a.cin; t.Cout = (int)a.cout; t.HW = (int)a.hw;
    if (lb != 4 || cw != 2) nw = 4;                                      // (the tuning configurations are built for 4)
    t.CS = (int)((a.cout + 32 * nw * cw - 1) / (32 * nw * cw));
    t.CTM = (int)(rows_pad / 32);
    t.S = a.stride; t.Wo = (int)a.w_out; t.Win = (int)a.w_in; t.HWin = (int)hw_in;
    t.cols = a.n * a.hw; t.tiles = tiles; t.zoff = a.zoff;
    t.items = tiles * t.CS;
    const int64_t grid = (t.items + 7) / 8 * 8;                           // padded to whole rounds over the 8 XCDs
    FQ_REQUIRE(grid < (1ll << 31), "fq_pwconv_i8: too many tiles for the split form");
    const size_t ldst = (size_t)kt * 1024 + (size_t)(nw * cw * 32) * 5 * sizeof(float);
    const int8_t* wfrag = a.wcodes + rows_pad * a.cin_pad;               // second half of fq_weight_codes' buffer
    if (int rc = pw_zero_stat(a)) return rc;
    bool launched = false;
#define FQ_PWS_CASE_NW(KT_, CW_, D_, LB_, NW_)                                                                         \
  if (kt == KT_ && cw == CW_ && lb == LB_ && nw == NW_) {                                                              \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_split_kernel<KT_, CW_, D_, LB_, NW_>),               \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;                      \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the split kernel");                       \
    hipLaunchKernelGGL((pwconv_split_kernel<KT_, CW_, D_, LB_, NW_>), dim3((unsigned)grid), dim3(NW_ * 64), ldst, a.st, \
                       a.x,                                                                                            \
                       wfrag, a.wscale, (const int*)a.wsum, a.bias, a.y, t, a.in_stat, (int)a.n, a.in_thr, a.levels,   \
                       a.lo_neg, kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out, a.residual);      \
    launched = true;                                                                                                   \
  }
#define FQ_PWS_CASE(KT_, CW_, D_, LB_) FQ_PWS_CASE_NW(KT_, CW_, D_, LB_, 4)
    // every K: the two default configurations; the power-of-two K of the deep MobileNet / ResNet layers also carry the
    // alternatives that tools/pwforms.py and FQ_PWS_CFG compare (three wavefronts per SIMD, four channel tiles)
#define FQ_PWS_KT(KT_) FQ_PWS_CASE(KT_, 1, (KT_ < 7 ? KT_ : 7), 4) FQ_PWS_CASE(KT_, 2, (KT_ < 3 ? KT_ : 3), 4)
#define FQ_PWS_KT_TUNE(KT_) \
  FQ_PWS_CASE(KT_, 1, 7, 3) FQ_PWS_CASE(KT_, 2, 5, 3) FQ_PWS_CASE(KT_, 4, 2, 3)
    FQ_PWS_KT(2) FQ_PWS_KT(4) FQ_PWS_KT(6) FQ_PWS_KT(8) FQ_PWS_KT(10) FQ_PWS_KT(12) FQ_PWS_KT(16) FQ_PWS_KT(18)
    FQ_PWS_KT(30) FQ_PWS_KT(32) FQ_PWS_KT(64)
    FQ_PWS_KT_TUNE(8) FQ_PWS_KT_TUNE(16) FQ_PWS_KT_TUNE(32) FQ_PWS_KT_TUNE(64)
    FQ_PWS_CASE_NW(8, 2, 3, 4, 8) FQ_PWS_CASE_NW(16, 2, 3, 4, 8) FQ_PWS_CASE_NW(32, 2, 3, 4, 8) FQ_PWS_CASE_NW(64, 2, 3, 4, 8)
#undef FQ_PWS_KT
#undef FQ_PWS_KT_TUNE
#undef FQ_PWS_CASE
#undef FQ_PWS_CASE_NW
    FQ_REQUIRE(launched, "fq_pwconv_i8: no instantiation of the split form for K/32=%d, %d tiles per wavefront, %d "
               "wavefronts per SIMD", kt, cw, lb);
    FQ_LAUNCH_CHECK();
    *taken = true;
    return FQ_OK;
  }
  FQ_REQUIRE(a.form != 6 && a.stride == 1, "fq_pwconv_i8: %s but the shape does not fit the split kernel (padded Cin / 32 "
             "= %d is not instantiated)", a.stride == 1 ? "FQ_PW_FORM=6" : "a strided call", kt);
  return FQ_OK;
}

}  // namespace fqi
