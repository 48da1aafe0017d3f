// libfakequant — K2k pointwise (1x1) convolution on int8 codes with the weights STATIONARY IN REGISTERS
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_pw.h"

namespace {

// K2k: the form for the deep layers (K = 256 / 512 / 1024: 14x14 and 7x7 planes of MobileNet, the bottleneck 1x1
// convolutions of ResNet-50), where the weight matrix (128 KB .. 1 MB of int8 codes) is much larger than a pixel tile and
// every earlier form paid for it per tile: the chunked form (K2i) re-stages all weights through LDS for every batch of
// four tiles with one wavefront per SIMD and a serial load -> quantise -> multiply chain per wavefront; the tile form (K2j)
// streams the whole matrix from L2 for every 32-pixel tile, two fragments ahead (SQ_WAIT_ANY 0.72-0.76 of its wave-cycles).
//
// Here a wavefront LOADS ITS WEIGHTS ONCE: CW channel tiles x KT fragments = 128 VGPRs of MFMA A operands (CW * KT = 32)
// that stay put while the workgroup walks over pixel tiles.  A workgroup of NW = 4 or 8 wavefronts therefore owns a SLICE
// of NW * CW * 32 output channels (NW = 8: K = 1024: 256, K = 512: 512, K = 256: 1024) and the grid is S slices x G tile
// groups.
// Per 32-pixel tile a workgroup
//   Q  quantises the tile ONCE across its wavefronts (wavefront w takes slabs w, w+NW, ...: lane = pixel, 16 channel loads
//      per half-slab straight from NCHW as in K2h) and publishes the int8 B fragments to an LDS panel (1 KB per slab);
//      the raw loads of the NEXT tile are already in flight while the current one is multiplied and stored;
//   M  multiplies: B fragment from the panel (one ds_read_b128 feeds CW MFMAs), A fragments from registers,
//      v_mfma_i32_32x32x32_i8, two independent accumulator chains;
//   E  stores with lane = pixel (two full 128-byte lines per store instruction), per-channel constants from LDS, statistic.
// The panel is double buffered, so there is ONE barrier per tile and a wavefront that is done with a tile starts
// quantising the next one while its neighbours still multiply; LDS use is 2 * KT KB + constants (<= 74 KB); registers allow
// two wavefronts per SIMD (one workgroup of 8 or two of 4 per CU).  A tile is quantised S times (once per slice: S = 1 for
// 512 -> 512, 2 for 512 -> 1024, 4 for 1024 -> 1024); the S workgroups of a tile group are placed on one XCD (block b runs on XCD b % 8) so that the tile comes
// out of that XCD's L2 after the first of them touched it.  Arithmetic and results are those of K2h / K2i / K2j: exact
// integer sums, the same quantiser, the same epilogue order (oracle.pwconv_i8).
struct PwrGeom {
  int Cin, Cout, HW;          // K = Cin = KT * 32 (no padding), Cout % (NW * CW * 32) == 0
  int S, G;                   // channel slices, tile groups; gridDim.x = S * G rounded up to a multiple of 8 * S
  int64_t cols, tiles;
  int zoff;
};

template <int KT, int CW, int NW>
__global__ __launch_bounds__(NW * 64, 1) void pwconv_wreg_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwrGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out) {
  static_assert(CW * KT <= 64, "a wavefront holds at most 64 weight fragments (256 registers)");
  constexpr int kSlots = 8;
  static_assert(KT % NW == 0 && KT / NW <= 8, "at most eight slabs (128 registers of raw activations) per wavefront");
  constexpr int kThreads = NW * 64;
  constexpr int SLABS = KT / NW;                                        // slabs each wavefront quantises per tile
  constexpr int NCH = NW * CW * 32;                                     // output channels of a workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char pwr_smem[];
  __shared__ unsigned k_stat[kSlots];
  v4i* panel = reinterpret_cast<v4i*>(pwr_smem);                       // [2][KT][64] B fragments
  float* c_sxw = reinterpret_cast<float*>(pwr_smem + (size_t)2 * KT * 1024);
  float* c_bsc = c_sxw + NCH;
  float* c_bsh = c_bsc + NCH;
  float* c_bias = c_bsh + NCH;
  int* c_zs = reinterpret_cast<int*>(c_bias + NCH);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: addresses built from it stay scalar
  const int h = lane >> 5, pl = lane & 31;
  const unsigned HW = (unsigned)g.HW, cols = (unsigned)g.cols;
  const int64_t plane = (int64_t)g.HW;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  // block -> (slice, group): the S workgroups of one tile group share b % 8, i.e. (as dispatched today) one XCD and its L2
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int slice = idx % g.S;
  const int group = (idx / g.S) * 8 + xcd;
  if (group >= g.G) return;                                             // padding blocks of the last row of 8 (uniform)
  const int64_t t_begin = g.tiles * group / g.G, t_end = g.tiles * (group + 1) / g.G;
  const int ch0 = slice * NCH;                                          // first output channel of this workgroup
  unsigned s_base;
  {
    const unsigned j0 = (unsigned)t_begin * 32u;
    s_base = (j0 < cols ? j0 : cols - 1) / HW;
  }

  struct Pix { unsigned smp, p; };
  auto pix_of = [&](int64_t t) __attribute__((always_inline)) {
    Pix r;
    unsigned j = (unsigned)(t < g.tiles ? t : g.tiles - 1) * 32u + (unsigned)pl;
    j = j < cols ? j : cols - 1;                                        // lanes past the end copy the last pixel
    r.smp = j / HW;
    r.p = j - r.smp * HW;
    return r;
  };
  // one half-slab per lane: 16 consecutive channels of pixel `pl` (for a fixed channel 32 lanes read 128 contiguous bytes)
  auto issue = [&](const Pix& px, int kt, float (&v)[16]) __attribute__((always_inline)) {
    const unsigned off = (unsigned)((((int64_t)px.smp * g.Cin + 16 * h) * plane + px.p) * 4);
    const char* ub = reinterpret_cast<const char*>(x) + (int64_t)kt * 32 * plane * 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const float*>(ub + (int64_t)i * plane * 4 + off);
  };

  // ---- first tile's activations and this wavefront's weights: everything in flight before the first wait ------------
  float raw[SLABS][16];
  Pix px = pix_of(t_begin);
#pragma unroll
  for (int j = 0; j < SLABS; ++j) issue(px, wave + NW * j, raw[j]);
  v4i afrag[CW][KT];
  {
    const v4i* wf = reinterpret_cast<const v4i*>(wfrag) + lane;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int ct = (ch0 >> 5) + wave * CW + c;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) afrag[c][kt] = wf[((int64_t)ct * KT + kt) << 6];
    }
  }
  FQ_PIN();
  const float max_ = in_stat != nullptr ? batch_mean_dev(in_stat, n) : in_thr[0];
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  if (in_stat != nullptr && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
  const float sx = q.scale;
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;
  for (int i = threadIdx.x; i < NCH; i += kThreads) {
    const int ic = ch0 + i;
    c_sxw[i] = sx * wscale[ic];
    c_zs[i] = g.zoff * wsum[ic];
    c_bias[i] = bias != nullptr ? bias[ic] : 0.0f;
    c_bsc[i] = has_bn ? bn_scale[ic] : 1.0f;
    c_bsh[i] = has_bn ? bn_shift[ic] : 0.0f;
  }
  const int ubias = 128 - g.zoff;
  auto quant_to_panel = [&](v4i* pan, int kt, const float (&v)[16]) __attribute__((always_inline)) {
    v4i f;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      f[d] = pack4_codes(fq_code_int(v[4 * d + 0], q), fq_code_int(v[4 * d + 1], q), fq_code_int(v[4 * d + 2], q),
                         fq_code_int(v[4 * d + 3], q), ubias);
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));   // pin the arithmetic here (see K2h)
    pan[(kt << 6) + lane] = f;
  };

  auto run = [&](auto bias_c, auto bn_c, auto act_c) __attribute__((always_inline)) {
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
    int buf = 0;
    for (int64_t t = t_begin; t < t_end; ++t, buf ^= 1) {
      v4i* pan = panel + (size_t)buf * KT * 64;
      // ---- Q: my slabs of this tile -> panel; then the next tile's loads go out ---------------------------------------
#pragma unroll
      for (int j = 0; j < SLABS; ++j) quant_to_panel(pan, wave + NW * j, raw[j]);
      const Pix cur = px;
      px = pix_of(t + 1);
      FQ_PIN();
#pragma unroll
      for (int j = 0; j < SLABS; ++j) issue(px, wave + NW * j, raw[j]);    // in flight during M and E (the last tile
      FQ_PIN();                                                           //  re-reads itself: discarded)
      __syncthreads();                                                     // panel[buf] complete
      // ---- M: CW channel tiles, B from the panel, A from registers, two accumulator chains ------------------------------
      constexpr int NACC = CW == 1 ? 2 : 1;                                // accumulator chains per channel tile
      v16i acc[CW][NACC];
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        const int cb = (wave * CW + c) * 32 + 4 * h;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const v4i z = *reinterpret_cast<const v4i*>(c_zs + cb + 8 * gq);
          acc[c][0][4 * gq + 0] = z.x; acc[c][0][4 * gq + 1] = z.y; acc[c][0][4 * gq + 2] = z.z; acc[c][0][4 * gq + 3] = z.w;
        }
        if (NACC == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[c][NACC - 1][r] = 0;
        }
      }
      v4i bq[3];
      bq[0] = pan[lane];
      bq[1] = pan[(1 << 6) + lane];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        if (kt + 2 < KT) bq[(kt + 2) % 3] = pan[((kt + 2) << 6) + lane];
#pragma unroll
        for (int c = 0; c < CW; ++c)
          acc[c][kt % NACC] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[c][kt], bq[kt % 3], acc[c][kt % NACC], 0, 0, 0);
      }
      // ---- E ---------------------------------------------------------------------------------------------------------------
      const unsigned yoff = (unsigned)((((int64_t)cur.smp * g.Cout + 4 * h) * plane + cur.p) * 4);
      float m = 0.0f;
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        const int cl = (wave * CW + c) * 32;                               // channel offset inside the slice
        const int cb = cl + 4 * h;
        char* ybase = reinterpret_cast<char*>(y) + (int64_t)(ch0 + cl) * plane * 4;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int c0 = cb + 8 * gq;
          const f4 sxw = *reinterpret_cast<const f4*>(c_sxw + c0);
          const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
          const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
          f4 bch = (f4){0.f, 0.f, 0.f, 0.f};
          if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) bch = *reinterpret_cast<const f4*>(c_bias + c0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = (float)(NACC == 2 ? acc[c][0][4 * gq + r] + acc[c][NACC - 1][4 * gq + r] : acc[c][0][4 * gq + r]) * sxw[r];
            if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) v = v + bch[r];
            if (BN_M == 1 || (BN_M < 0 && has_bn)) {
              v = v * bsc[r];
              v = v + bsh[r];
            }
            v = ACT_M < 0 ? act_rt(v, act) : act_rt(v, ACT_M);
            *reinterpret_cast<float*>(ybase + (int64_t)(8 * gq + r) * plane * 4 + yoff) = v;
            m = fmaxf(m, fabsf(v));
          }
        }
      }
      if (has_stat) {
        const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)cur.smp);
        if (__all(cur.smp == s0)) {
          const float wm = wave_max(m);
          if (lane == 0) {
            const unsigned slot = s0 - s_base;
            if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
            else atomic_max_f32(stat_out + s0, wm);
          }
        } else {
          const unsigned slot = cur.smp - s_base;
          if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
          else atomic_max_f32(stat_out + cur.smp, m);
        }
      }
    }
  };
  __syncthreads();                                                      // constants and the statistic table are staged
  using std::integral_constant;
  if (bias == nullptr && has_bn && act == FQ_ACT_RELU)
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
}

}  // namespace

namespace fqi {

// register-stationary form (K2k): K = 256 / 512 / 1024 with Cout a multiple of the slice width
int pw_try_wreg(const PwCall& a, bool* taken) {
  *taken = false;
  const int kt = (int)(a.cin / 32);
  const bool kt_ok = a.cin % 32 == 0 && a.cin_pad == a.cin && (kt == 8 || kt == 16 || kt == 32);
  // (CW, NW) per K: channel tiles per wavefront and wavefronts per workgroup.  FQ_PWR_CW / FQ_PWR_NW override for tuning.
  const int force_cw = env_int("FQ_PWR_CW", 0), force_nw = env_int("FQ_PWR_NW", 0);   // read per call: tests switch them
  int cw = kt == 8 ? 4 : (kt == 16 ? 2 : 1), nw = 4;
  if (force_cw > 0) cw = force_cw;
  if (force_nw > 0) nw = force_nw;
  const int nch = nw * cw * 32;
  const bool built = (kt == 8 && cw == 4 && (nw == 4 || nw == 8)) || (kt == 16 && cw == 2 && (nw == 4 || nw == 8)) ||
                     (kt == 16 && cw == 4 && nw == 4) || (kt == 32 && cw == 1 && (nw == 4 || nw == 8)) ||
                     (kt == 32 && cw == 2 && nw == 4);
  const bool ok = kt_ok && built && a.cout % nch == 0;
  if (!((a.form == 0 || a.form == 6) && ok)) {
    FQ_REQUIRE(a.form != 6, "fq_pwconv_i8: FQ_PW_FORM=6 but the shape does not fit the register-stationary kernel");
    return FQ_OK;
  }
  PwrGeom r;
  r.Cin = (int)a.cin; r.Cout = (int)a.cout; r.HW = (int)a.hw;
  r.cols = a.n * a.hw; r.tiles = (r.cols + 31) / 32; r.zoff = a.zoff;
  r.S = (int)(a.cout / nch);
  const size_t lds = (size_t)2 * kt * 1024 + (size_t)nch * 5 * sizeof(float);
  // resident workgroups: the 4-wavefront forms use the whole register file of a SIMD for one wavefront (one workgroup per
  // CU), the 8-wavefront forms two per SIMD; each group walks a contiguous range of tiles.  G = as many groups as stay
  // resident, but no more than one per `min_tiles` tiles (a workgroup's start-up — its weight fragments, 32-64 KB per
  // wavefront — is only worth it over a few tiles).
  const int wg_per_cu = env_int("FQ_PWR_WG_PER_CU", 1);
  const int min_tiles = env_int("FQ_PWR_MIN_TILES", 2);
  int64_t groups = (int64_t)num_cu() * wg_per_cu / r.S;
  const int64_t by_tiles = (r.tiles + min_tiles - 1) / min_tiles;
  if (groups > by_tiles) groups = by_tiles;
  if (groups < 1) groups = 1;
  r.G = (int)groups;
  const int64_t rows8 = (groups + 7) / 8;                               // rows of 8 groups (one per XCD)
  const int64_t grid = rows8 * r.S * 8;
  const int64_t rows_pad = (a.cout + 63) / 64 * 64;
  const int8_t* wfrag = a.wcodes + rows_pad * a.cin_pad;                // second half of fq_weight_codes' buffer
  if (int rc = pw_zero_stat(a)) return rc;
#define FQ_PWR_CASE(KT_, CW_, NW_)                                                                                     \
  if (kt == KT_ && cw == CW_ && nw == NW_) {                                                                           \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_wreg_kernel<KT_, CW_, NW_>), \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess; \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the register-stationary kernel");         \
    hipLaunchKernelGGL((pwconv_wreg_kernel<KT_, CW_, NW_>), dim3((unsigned)grid), dim3(NW_ * 64), lds, a.st, a.x,      \
                       wfrag, a.wscale, (const int*)a.wsum, a.bias, a.y, r, a.in_stat, (int)a.n, a.in_thr, a.levels,   \
                       a.lo_neg, kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out);                  \
  }
  FQ_PWR_CASE(8, 4, 4) FQ_PWR_CASE(8, 4, 8) FQ_PWR_CASE(16, 2, 4) FQ_PWR_CASE(16, 2, 8) FQ_PWR_CASE(16, 4, 4)
  FQ_PWR_CASE(32, 1, 4) FQ_PWR_CASE(32, 1, 8) FQ_PWR_CASE(32, 2, 4)
#undef FQ_PWR_CASE
  FQ_LAUNCH_CHECK();
  *taken = true;
  return FQ_OK;
}

}  // namespace fqi
