// libfakequant — K2h pointwise (1x1) convolution on int8 codes, whole weight matrix resident in LDS
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_pw.h"

namespace {

// K2h: streaming form of the pointwise convolution for layers whose whole weight matrix fits in LDS (Cout*K <= 64 KB:
// the large-plane layers, where the bytes are).  v_mfma_i32_32x32x32_i8 with the ACTIVATIONS as the B operand: lane l
// owns pixel l&31 and needs the 16 consecutive channels 16*(l>>5).. of a 32-channel slab in its registers — which is
// exactly what 16 plain dword loads of NCHW give it (for a fixed channel, 32 lanes read 128 contiguous bytes).  So
// there is no transposition at all: load, fake-quantise, pack four codes per dword, multiply.  D comes out with
// lane = pixel, register = channel: every store instruction writes two full 128-byte lines.  No LDS traffic for the
// activations, no barrier inside the loop, ~110 VGPRs: waves are independent and hide each other's latencies, unlike
// the panel kernel above whose workgroups move through load / quantise / multiply / store phases in lock step.
// Weights sit in LDS in fragment order (1 KB per (32 channels x 32 k) fragment, read with one conflict-free
// ds_read_b128 per lane); per-channel constants sit next to them and are read 4 channels at a time (D holds channels
// 8*(r/4) + 4*(l>>5) + r%4 in register r).  A wavefront walks a contiguous range of 32-pixel tiles; the loads of the
// next slab / tile are issued before the current one is quantised (two register buffers).

struct PwsGeom {
  int Cin, K, Cout, CT, HW;   // K: row stride of the weight codes (cin_pad); CT = ceil(Cout / 32)
  int64_t cols, tiles;        // n * HW, ceil(cols / 32)
  int zoff;
  // OUT16 (fq_pwconv_i8_c16): y is a C16 code tensor with CBo = Cout / 16 blocks holding the CONSUMER's codes
  // IN16: x is a C16 code tensor with CBi = ceil(Cin / 16) blocks (quantised by its producer: no quantiser here)
  int CBo, CBi;
  float out_levels;
  int out_lo_neg, out_zoff;
  // DUAL (fq_pwconv_i8_c16_dual): y is fp32 AND y16 receives the codes of the same values under dual_thr (CBo / out_* describe it)
  char* y16;
  const float* dual_thr;
};

// RES: a residual operand of y's shape is added after BatchNorm, before the activation (compile-time: the loads of the
// residual would otherwise cost the plain instantiations their occupancy)
// IN16 / PART (round 4: the thin layers of MobileNetV2 on its large planes - 32 -> 16, 96 -> 24, 24 -> 144, 144 -> 24 @112x112 /
// 56x56 - which the split form ran as 12 544 ... 50 176 one-tile workgroups with a prologue each, at 1.0 - 2.0 TB/s):
// IN16: the input is a C16 code tensor - a lane's 16 codes of a half-slab are ONE 16-byte load, no quantiser;
// PART: Cin need not be a multiple of 16 (the loads of a ragged half-slab are clamped to the last channel: whatever code
// they get meets a zero weight code) and Cout need not be a multiple of 32 (channels past Cout get all-zero constants and an
// out-of-range buffer offset: the hardware drops their stores and returns 0 for their residual loads).
// NT (round 5; plain fp32 instantiations only): bit 0 - nontemporal activation loads (the input is dead after this pass), bit 1 -
// nontemporal stores.  The large-plane layers move 300-600 MB per launch, more than the 256 MB Infinity Cache holds: what they
// leave there is what the NEXT launches of the batches in flight find (or do not find).  Chosen per launch by tensor size
// (pw_try_stream; profiles/r5_pws_nt_ab.txt).
template <int KT, bool RES, bool OUT16 = false, bool IN16 = false, bool PART = false, bool DUAL = false, int NT = 0>
__global__ __launch_bounds__(kBlock, 2) void pwconv_stream_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wc, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwsGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out, const float* __restrict__ residual, const float* __restrict__ out_thr) {
  constexpr int kSlots = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char pws_smem[];
  __shared__ unsigned k_stat[kSlots];
  v4i* ldsA = reinterpret_cast<v4i*>(pws_smem);                        // [CT][KT][64] fragments
  const int nch = g.CT * 32;
  float* c_sxw = reinterpret_cast<float*>(pws_smem + (size_t)g.CT * KT * 1024);
  float* c_bsc = c_sxw + nch;
  float* c_bsh = c_bsc + nch;
  float* c_bias = c_bsh + nch;
  int* c_zs = reinterpret_cast<int*>(c_bias + nch);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, pl = lane & 31;
  const unsigned HW = (unsigned)g.HW, cols = (unsigned)g.cols;
  const int64_t plane = (int64_t)g.HW;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const int64_t nwaves = (int64_t)gridDim.x * 4, wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t t_begin = g.tiles * wid / nwaves, t_end = g.tiles * (wid + 1) / nwaves;
  unsigned s_base;
  {
    const int64_t t0 = g.tiles * ((int64_t)blockIdx.x * 4) / nwaves;
    const unsigned j0 = (unsigned)t0 * 32u;
    s_base = (j0 < cols ? j0 : cols - 1) / HW;
  }

  // per-tile pixel record of this lane
  struct Pix { unsigned smp, p; bool valid; };
  auto pix_of = [&](int64_t t) __attribute__((always_inline)) {
    Pix r;
    unsigned j = (unsigned)t * 32u + (unsigned)pl;
    r.valid = j < cols;
    j = r.valid ? j : cols - 1;
    r.smp = j / HW;
    r.p = j - r.smp * HW;
    return r;
  };
  // addresses: wave-uniform base (SGPR arithmetic) + ONE 32-bit per-lane byte offset per tile (host checks the tensors
  // are < 4 GB).  With 64-bit per-lane pointers the compiler materialised an address pair per load and spilled.
  auto lane_off = [&](const Pix& px, int c0) __attribute__((always_inline)) {
    return (unsigned)((((int64_t)px.smp * g.Cin + c0) * plane + px.p) * 4);
  };
  auto issue = [&](const Pix& px, int kt, float (&v)[16]) __attribute__((always_inline)) {
    if (IN16) {
      const int blk = 2 * kt + h;                                      // this half-wave's block of 16 channels
      const bool okb = blk < g.CBi;
      const unsigned off16 = (unsigned)((((int64_t)px.smp * g.CBi + (okb ? blk : 0)) * plane + px.p) * 16);
      v4i c = *reinterpret_cast<const v4i*>(reinterpret_cast<const char*>(x) + off16);
      c = okb ? c : (v4i){0, 0, 0, 0};                                  // (a block past the channels meets zero weight codes)
      v[0] = __int_as_float(c[0]); v[1] = __int_as_float(c[1]); v[2] = __int_as_float(c[2]); v[3] = __int_as_float(c[3]);
      return;
    }
    const int cg = kt * 32 + 16 * h;                                   // this half-wave's 16 channels of slab kt
    const unsigned off = lane_off(px, cg < g.Cin ? 16 * h : 0);        // padded group: read the valid half, discarded
    const char* ub = reinterpret_cast<const char*>(x) + (int64_t)kt * 32 * plane * 4;
    if (PART) {                                                        // a ragged half-slab: stay inside the sample's channels
      const int last = g.Cin - 1 - (cg < g.Cin ? cg : kt * 32);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        v[i] = *reinterpret_cast<const float*>(ub + (int64_t)(i < last ? i : (last < 0 ? 0 : last)) * plane * 4 + off);
      return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float* src = reinterpret_cast<const float*>(ub + (int64_t)i * plane * 4 + off);
      v[i] = (NT & 1) ? __builtin_nontemporal_load(src) : *src;
    }
  };

  float bufa[16], bufb[16];
  int64_t t_first = t_begin < g.tiles ? t_begin : g.tiles - 1;
  Pix nxt = pix_of(t_first);
  issue(nxt, 0, bufa);                                                  // in flight during the whole set-up
  FQ_PIN();

  const float max_ = input_threshold(in_stat, n, in_thr, cur_max_out, blockIdx.x == 0);
  int zoff = g.zoff;
  const QParams q = make_qparams_rt(max_, levels, lo_neg_max, eps, in_thr, zoff);
  const float sx = q.scale;
  // range mode (nn.Conv2D(quantized=True)): `bias` holds int32 codes that join the integer sum
  const int* ibias = lo_neg_max == kRangeMode ? reinterpret_cast<const int*>(bias) : nullptr;
  const float* fbias = lo_neg_max == kRangeMode ? nullptr : bias;
  QParams q2;
  q2.lo = q2.hi = q2.denom = q2.scale = 0.0f;
  q2.rden = 0.0;
  if (OUT16) q2 = make_qparams(out_thr[0], g.out_levels, g.out_lo_neg != 0, eps);
  if (DUAL) q2 = make_qparams(g.dual_thr[0], g.out_levels, g.out_lo_neg != 0, eps);
  const int ubias2 = 128 - g.out_zoff;
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;
  // weights -> fragment order: fragment (ct, kt), lane (row % 32) + 32 * (16-byte chunk % 2)
  for (int idx = threadIdx.x; idx < nch * KT * 2; idx += kBlock) {
    const int row = idx / (KT * 2), kc = idx - row * (KT * 2);
    const v4i wv = *reinterpret_cast<const v4i*>(wc + (int64_t)row * g.K + kc * 16);
    ldsA[(((row >> 5) * KT + (kc >> 1)) << 6) + (row & 31) + 32 * (kc & 1)] = wv;
  }
  for (int i = threadIdx.x; i < nch; i += kBlock) {
    const bool ok = i < g.Cout;
    const int ic = ok ? i : 0;
    c_sxw[i] = (!PART || ok) ? sx * wscale[ic] : 0.0f;
    c_zs[i] = ok ? zoff * wsum[ic] + (ibias != nullptr ? ibias[ic] : 0) : 0;
    c_bias[i] = (fbias != nullptr && (!PART || ok)) ? fbias[ic] : 0.0f;
    c_bsc[i] = (!PART || ok) ? (has_bn ? bn_scale[ic] : 1.0f) : 0.0f;
    c_bsh[i] = (has_bn && (!PART || ok)) ? bn_shift[ic] : 0.0f;
  }
  __syncthreads();

  v4i bfrag[KT];
  const int ubias = 128 - zoff;
  const unsigned nn_xor = fq_nonneg_xor(ubias);
  auto quant = [&](int kt, const float (&v)[16], auto nn_c) __attribute__((always_inline)) {
    if (IN16) {
      bfrag[kt] = (v4i){__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])};
      return;
    }
    const bool gvalid = kt * 32 + 16 * h < g.Cin;
    v4i f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int packed = fq_pack4<decltype(nn_c)::value>(v[4 * d + 0], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3], q, ubias,
                                                         nn_xor);
      f[d] = gvalid ? packed : 0;
    }
    // pin the quantisation HERE: it is pure arithmetic whose results are only needed by the MFMAs, and the optimiser
    // otherwise sinks it below every prefetch, keeping all 16 * KT loaded values live (256 VGPRs + spills)
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    bfrag[kt] = f;
  };
  // (nn2_c: the CONSUMER's clip range of a C16 output starts at 0 - the five-instruction quantiser of fq_common.h writes its
  // codes; round 5: it did so for the second output only, the code output took the nine-instruction one whatever its range)
  auto tile_done = [&](const Pix& px, auto bias_c, auto bn_c, auto act_c, auto nn2_c) __attribute__((always_inline)) {
    constexpr bool NN2 = decltype(nn2_c)::value;
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
    // (a code output behind a compile-time ReLU / ReLU6: activation and the consumer's clip as ONE median, the statistic from
    // the raw values - fq_pw_split_kernel.h)
    constexpr bool FOLD = OUT16 && !DUAL && (ACT_M == FQ_ACT_RELU || ACT_M == FQ_ACT_RELU6);
    QParams qc = q2;
    if (FOLD) {
      qc.lo = 0.0f;
      if (ACT_M == FQ_ACT_RELU6) qc.hi = fminf(q2.hi, 6.0f);
    }
    const unsigned yoff = (unsigned)((((int64_t)px.smp * g.Cout + 4 * h) * plane + px.p) * 4);
    // PART: every store (and residual load) of a channel tile goes through a resource over the whole tensor with an
    // out-of-range offset for channels past Cout
    const int64_t n_smp = (int64_t)(cols / HW);
    const fq_rsrc yrp = make_rsrc(y, PART ? (OUT16 ? n_smp * g.CBo * plane * 16 : n_smp * g.Cout * plane * 4) : 0);
    float m = 0.0f;
#pragma unroll 1
    for (int ct = 0; ct < g.CT; ++ct) {
      v16i acc;
      const int cb = ct * 32 + 4 * h;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const v4i z = *reinterpret_cast<const v4i*>(c_zs + cb + 8 * gq);
        acc[4 * gq + 0] = z.x; acc[4 * gq + 1] = z.y; acc[4 * gq + 2] = z.z; acc[4 * gq + 3] = z.w;
      }
      // residual operand (the shortcut of a ResNet unit; y's shape): requested before the MFMAs, added after BatchNorm
      float res[RES ? 16 : 1];
      if (RES) {
        const fq_rsrc rr = make_rsrc(residual, (int64_t)(cols / HW) * g.Cout * plane * 4);      // (host: y is below 4 GB)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const bool okc = !PART || ct * 32 + 8 * (i >> 2) + 4 * h + (i & 3) < g.Cout;
          res[i] = (NT & 1) ? buf_ld_f32_nt(rr, okc ? yoff : 0x80000000u, (unsigned)((ct * 32 + 8 * (i >> 2) + (i & 3)) * plane * 4))
                            : buf_ld_f32(rr, okc ? yoff : 0x80000000u, (unsigned)((ct * 32 + 8 * (i >> 2) + (i & 3)) * plane * 4));
        }
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ldsA[((ct * KT + kt) << 6) + lane], bfrag[kt], acc, 0, 0, 0);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int c0 = cb + 8 * gq;                                     // channels c0 .. c0+3 in registers 4gq .. 4gq+3
        const f4 sxw = *reinterpret_cast<const f4*>(c_sxw + c0);
        const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
        const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
        f4 bch = (f4){0.f, 0.f, 0.f, 0.f};
        if (BIAS_M == 1 || (BIAS_M < 0 && fbias != nullptr)) bch = *reinterpret_cast<const f4*>(c_bias + c0);
        float vq[4];
        // two channels at a time: scale / bias / BatchNorm as packed fp32 instructions (two IEEE operations each - the same
        // values as the scalar form; the kernel is bound by vector-instruction issue, profiles/r2_pmc_sq.txt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          typedef float f2 __attribute__((ext_vector_type(2)));
          f2 v = (f2){(float)acc[4 * gq + r], (float)acc[4 * gq + r + 1]};
          v = v * (f2){sxw[r], sxw[r + 1]};
          if (BIAS_M == 1 || (BIAS_M < 0 && fbias != nullptr)) v = v + (f2){bch[r], bch[r + 1]};
          if (BN_M == 1 || (BN_M < 0 && has_bn)) {
            v = v * (f2){bsc[r], bsc[r + 1]};
            v = v + (f2){bsh[r], bsh[r + 1]};
          }
          if (RES) v = v + (f2){res[4 * gq + r], res[4 * gq + r + 1]};
          if (!FOLD) {
            v.x = ACT_M < 0 ? act_rt(v.x, act) : act_rt(v.x, ACT_M);
            v.y = ACT_M < 0 ? act_rt(v.y, act) : act_rt(v.y, ACT_M);
          }
          // no masks: lanes past the end hold a copy of the last pixel (clamped loads) and re-store its values, and the
          // host guarantees Cout % 32 == 0
          if (OUT16 || DUAL) {
            vq[r] = v.x;
            vq[r + 1] = v.y;
          }
          if (OUT16) {
          } else if (PART) {
            const int chn = ct * 32 + 8 * gq + 4 * h + r;
            const unsigned so = (unsigned)((ct * 32 + 8 * gq + r) * plane * 4);
            buf_st_f32(yrp, chn < g.Cout ? yoff : 0x80000000u, so, v.x);
            buf_st_f32(yrp, chn + 1 < g.Cout ? yoff : 0x80000000u, so + (unsigned)(plane * 4), v.y);
          } else {
            char* yb = reinterpret_cast<char*>(y) + (int64_t)(ct * 32 + 8 * gq + r) * plane * 4 + yoff;
            if (NT & 2) {
              __builtin_nontemporal_store(v.x, reinterpret_cast<float*>(yb));
              __builtin_nontemporal_store(v.y, reinterpret_cast<float*>(yb + plane * 4));
            } else {
              *reinterpret_cast<float*>(yb) = v.x;
              *reinterpret_cast<float*>(yb + plane * 4) = v.y;
            }
          }
          m = FOLD ? fmaxf(m, fmaxf(v.x, v.y)) : fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y)));
        }
        if (OUT16 || DUAL) {   // channels 8 gq + 4 h .. + 3 of the lane's pixel = bytes 8 (gq & 1) + 4 h .. of block 2 ct + gq / 2
          // (DUAL: the host only asks for unsigned codes of a non-negative range - the five-instruction quantiser, fq_common.h;
          // 167 registers: the third workgroup per CU still fits)
          const int packed = DUAL ? fq_pack4<true>(vq[0], vq[1], vq[2], vq[3], q2, ubias2, 0x80808080u)
                                  : fq_pack4<NN2>(vq[0], vq[1], vq[2], vq[3], qc, ubias2, fq_nonneg_xor(ubias2));
          if (PART && !DUAL) {                         // a whole block past Cout (Cout % 32 == 16) does not exist
            const int blk = 2 * ct + (gq >> 1);
            const unsigned o16 = (unsigned)((((int64_t)px.smp * g.CBo + (blk < g.CBo ? blk : 0)) * plane + px.p) * 16 +
                                            8 * (gq & 1) + 4 * h);
            buf_st_f32(yrp, blk < g.CBo ? o16 : 0x80000000u, 0, __int_as_float(packed));
          } else {                                     // (DUAL: the host asks for Cout % 32 == 0)
            char* yb = (DUAL ? g.y16 : reinterpret_cast<char*>(y)) +
                       (((int64_t)px.smp * g.CBo + 2 * ct + (gq >> 1)) * plane + px.p) * 16 + 8 * (gq & 1) + 4 * h;
            *reinterpret_cast<int*>(yb) = packed;
          }
        }
      }
    }
    if (FOLD && ACT_M == FQ_ACT_RELU6) m = fminf(m, 6.0f);
    if (has_stat) {
      const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)px.smp);
      const bool uniform = __all(!px.valid || px.smp == s0);
      if (uniform) {
        const float wm = wave_max_nonneg(px.valid ? m : 0.0f);
        if (lane == 0) {
          const unsigned slot = s0 - s_base;
          if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
          else atomic_max_f32(stat_out + s0, wm);
        }
      } else if (px.valid) {
        const unsigned slot = px.smp - s_base;
        if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
        else atomic_max_f32(stat_out + px.smp, m);
      }
    }
  };
  // one tile: its first slab is already in `first`; the prefetch of the following slab / tile alternates buffers
  auto run_tile = [&](int64_t t, float (&first)[16], float (&second)[16], auto bias_c, auto bn_c, auto act_c,
                      auto nn_c, auto nn2_c) __attribute__((always_inline)) {
    const Pix cur = nxt;
    const int64_t tn = t + 1 < g.tiles ? t + 1 : g.tiles - 1;
    nxt = pix_of(tn);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      float (&mine)[16] = (kt & 1) ? second : first;
      float (&other)[16] = (kt & 1) ? first : second;
      if (kt + 1 < KT) issue(cur, kt + 1, other);
      else issue(nxt, 0, other);
      FQ_PIN();
      quant(kt, mine, nn_c);
      FQ_PIN();
    }
    tile_done(cur, bias_c, bn_c, act_c, nn2_c);
    FQ_PIN();
  };
  auto run_all = [&](auto bias_c, auto bn_c, auto act_c, auto nn_c, auto nn2_c) __attribute__((always_inline)) {
    if (KT & 1) {                                                       // the buffers swap roles from tile to tile
      int64_t t = t_begin;
      for (; t + 1 < t_end; t += 2) {
        run_tile(t, bufa, bufb, bias_c, bn_c, act_c, nn_c, nn2_c);
        run_tile(t + 1, bufb, bufa, bias_c, bn_c, act_c, nn_c, nn2_c);
      }
      if (t < t_end) run_tile(t, bufa, bufb, bias_c, bn_c, act_c, nn_c, nn2_c);
    } else {
      for (int64_t t = t_begin; t < t_end; ++t) run_tile(t, bufa, bufb, bias_c, bn_c, act_c, nn_c, nn2_c);
    }
  };
  // the compile-time epilogues (BatchNorm, no bias, fixed activation: the fused-inference case) come with the 5-instruction
  // quantiser of non-negative quotients (fq_common.h); signed activations take the generic instantiation
  using std::integral_constant;
  using std::true_type;
  using std::false_type;
  const bool nn = fq_nonneg(q);
  // (only a kernel that writes codes is instantiated twice: with the five-instruction output quantiser where the values it clips
  // cannot be negative - the consumer's range starts at 0, or a ReLU stands in front of it - and with the general one)
  auto go = [&](auto bias_c, auto bn_c, auto act_c, auto nn_c, bool nn2) __attribute__((always_inline)) {
    if constexpr (OUT16 && !DUAL) {
      if (nn2) run_all(bias_c, bn_c, act_c, nn_c, true_type{});
      else run_all(bias_c, bn_c, act_c, nn_c, false_type{});
    } else {
      run_all(bias_c, bn_c, act_c, nn_c, false_type{});
    }
  };
  const bool nn2_relu = q2.denom > 0.0f, nn2_any = fq_nonneg(q2);
  if (nn && fbias == nullptr && has_bn && act == FQ_ACT_RELU)
    go(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU>{}, true_type{}, nn2_relu);
  else if (nn && fbias == nullptr && has_bn && act == FQ_ACT_RELU6)
    go(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU6>{}, true_type{}, nn2_relu);
  else if (nn && fbias == nullptr && has_bn && act == FQ_ACT_NONE)
    go(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_NONE>{}, true_type{}, nn2_any);
  else if (nn)
    go(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{}, true_type{}, nn2_any);
  else
    go(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{}, false_type{}, nn2_any);

  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < cols / HW)
      FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
}


}  // namespace

namespace fqi {

bool pw_stream_shape_ok(const PwCall& c) {
  const int kt = (int)((c.cin + 31) / 32);
  const int ct = (int)((c.cout + 31) / 32);
  const size_t lds = (size_t)ct * kt * 1024 + (size_t)ct * 32 * 5 * sizeof(float);
  const bool kt_ok = kt == 1 || kt == 2 || kt == 3 || kt == 4 || kt == 6 || kt == 8;
  return c.cin % 16 == 0 && c.cout % 32 == 0 && kt_ok && lds <= 72 * 1024;
}

// the thin instantiations (IN16 / PART): any Cin up to 192, any Cout (a multiple of 16 for a C16 output)
static bool pw_stream_thin_ok(const PwCall& c) {
  const int kt = (int)((c.cin + 31) / 32);
  const int ct = (int)((c.cout + 31) / 32);
  const size_t lds = (size_t)ct * kt * 1024 + (size_t)ct * 32 * 5 * sizeof(float);
  static const int thin = env_int("FQ_PWS_THIN", 1);                    // A/B: 0 leaves these shapes to the split form
  // (r4: 3000 instead of 4096 - the 28 x 28 planes of a batch of 128 are 3136 tiles: MobileNetV2 W4 offline 144.8 -> 146.9 k
  // images/s, ResNet-50 offline unchanged; 1500 / 700 add nothing, profiles/r4_thin_tiles_ab.txt)
  static const int thin_min_tiles = env_int("FQ_PWS_THIN_MIN_TILES", 3000);
  // (codes in AND codes out: K = 32 - MobileNetV2's first 1x1 behind a first convolution that hands its codes over - and, round 6,
  // K = 256 / 512: the first 1x1 of ResNet-50's units on the 56x56 and 28x28 planes, which the split form ran as 12 544 / 3 136
  // one-tile workgroups at 2.2 / 2.0 TB/s of their 1-byte-per-element traffic: ResNet-50 offline +2.4 % images/s,
  // profiles/r6_wide_c16_stream_ab.txt)
  static const int both_wide = env_int("FQ_PWS_THIN_WIDE", 1);           // A/B: 0 leaves them to the split form
  const bool both = c.in_c16 && c.out_thr != nullptr;
  const bool wide = both && both_wide && (kt == 8 || kt == 16) && c.cin % 32 == 0 && c.cout % 32 == 0 && c.residual == nullptr;
  return thin && kt >= 1 && (kt <= 6 || wide) && lds <= 72 * 1024 && c.stride == 1 && (c.out_thr == nullptr || c.cout % 16 == 0) &&
         !(both && kt != 1 && !wide) && !(c.out_thr != nullptr && c.residual != nullptr) &&
         (c.n * c.hw + 31) / 32 > thin_min_tiles && c.n * c.cout * c.hw * 4 < (1ll << 32) && c.n * c.cin * c.hw * 4 < (1ll << 32);
}

// streaming form: the whole weight matrix in LDS, activations straight from NCHW into MFMA registers
bool pw_stream_thin_takes(const PwCall& c) {
  const bool ragged = c.cin % 16 != 0 || c.cout % 32 != 0;
  const bool c16 = c.in_c16 || c.out_thr != nullptr;
  // (form 6 is what every C16 call carries; a fp32 call that NAMES the split form gets the split form)
  const bool form_ok = c.form == 0 || c.form == 3 || (c.form == 6 && c16);
  // (a second output - fq_pwconv_i8_c16_dual - where the instantiation exists; the split form has the others)
  const int kt = (int)((c.cin + 31) / 32);
  if (c.y16 != nullptr && !(c.in_c16 && c.residual != nullptr && c.cout % 32 == 0 && (kt == 2 || kt == 4))) return false;
  return (c.in_c16 || ragged) && pw_stream_thin_ok(c) && form_ok;
}

int pw_try_stream(const PwCall& c, bool* taken) {
  *taken = false;
  const int kt = (int)((c.cin + 31) / 32);
  const int ct = (int)((c.cout + 31) / 32);
  const size_t lds = (size_t)ct * kt * 1024 + (size_t)ct * 32 * 5 * sizeof(float);
  const bool shape_ok = pw_stream_shape_ok(c);
  const bool out16 = c.out_thr != nullptr;
  // the thin instantiations: C16 input, ragged Cin / Cout - large planes only (more than 4096 tiles), where the split form's
  // one-tile workgroups are all prologue
  const bool thin = pw_stream_thin_takes(c);
  // (C16 output: large planes only - the split form is faster on few tiles)
  if (!thin) {
    if (out16 && !(shape_ok && !c.in_c16 && c.stride == 1 && c.residual == nullptr && (c.n * c.hw + 31) / 32 > 4096)) return FQ_OK;
    if (!((c.form == 0 || c.form == 3 || out16) && shape_ok) || c.in_c16) {
      FQ_REQUIRE(c.form != 3, "fq_pwconv_i8: FQ_PW_FORM=3 but the shape does not fit the streaming kernel");
      return FQ_OK;
    }
  }
  PwsGeom s;
  s.Cin = (int)c.cin; s.K = (int)c.cin_pad; s.Cout = (int)c.cout; s.CT = ct; s.HW = (int)c.hw;
  s.cols = c.n * c.hw; s.tiles = (s.cols + 31) / 32; s.zoff = c.zoff;
  s.CBo = (int)(c.cout / 16); s.CBi = (int)((c.cin + 15) / 16); s.out_levels = c.out_levels; s.out_lo_neg = c.out_lo_neg; s.out_zoff = c.out_zoff;
  s.y16 = (char*)c.y16; s.dual_thr = c.dual_thr;
  FQ_REQUIRE(c.y16 == nullptr || (thin && c.in_c16 && c.residual != nullptr && c.cout % 32 == 0 && (kt == 2 || kt == 4)),
             "fq_pwconv_i8_c16_dual: the streaming form writes a second output for a C16 input with a residual operand, K = 64 / 128");
  // persistent workgroups: as many as stay resident (LDS / 2 per SIMD by registers), each wave a contiguous range
  // (measured, tools/pwbench.py: 3 per CU for the 126-VGPR instantiations KT <= 2, 2 above)
  int per_cu = (int)((160 * 1024) / (lds + 1024));
  const int by_regs = kt <= 2 ? 3 : 2;
  per_cu = per_cu > by_regs ? by_regs : per_cu;
  static const int pws_wg = env_int("FQ_PWS_WG_PER_CU", 0);
  if (pws_wg > 0) per_cu = pws_wg;
  int64_t grid = (int64_t)num_cu() * per_cu;
  const int64_t need = (s.tiles + 3) / 4;
  if (grid > need) grid = need;
  if (int rc = pw_zero_stat(c)) return rc;
  // nontemporal policy of the plain instantiations, by tensor size (MB; tuning: FQ_PWS_NTL_MB / FQ_PWS_NTS_MB, 0 = always,
  // a huge value = never).  The input of a pointwise layer is dead after this pass: always loaded nontemporally; an output
  // from 150 MB up (the three large-plane layers of MobileNet: 205-411 MB) is stored nontemporally - beside the tensors of two
  // other batches in flight the 256 MB Infinity Cache would not keep it until its consumer starts anyway, and what it does
  // keep (the 50-100 MB tensors of the middle layers) is then still there.  Measured with three batches in flight, 18 runs
  // alternating in three calls: +1.6 ... +3.0 % images/s (profiles/r5_pws_nt_ab.txt, r5_pws_nt_sweep.txt, r5_nt_sweep2.txt; a
  // box wanders by +-3 % between runs, and the kernels timed ALONE do not show it: the gain is what the OTHER launches find).
  static const int ntl_mb = env_int("FQ_PWS_NTL_MB", 0), nts_mb = env_int("FQ_PWS_NTS_MB", 150);
  const double in_mb = 4e-6 * (double)c.n * c.cin * c.hw, out_mb = 4e-6 * (double)c.n * c.cout * c.hw;
  const int nt = (in_mb >= ntl_mb ? 1 : 0) | (out_mb >= nts_mb ? 2 : 0);
#define FQ_PWS_LAUNCH(KT_, RES_, O16_)                                                                                 \
  {                                                                                                                    \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_stream_kernel<KT_, RES_, O16_>), \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) == hipSuccess; \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the streaming kernel");                   \
    hipLaunchKernelGGL((pwconv_stream_kernel<KT_, RES_, O16_>), dim3((unsigned)grid), dim3(kBlock), lds, c.st, c.x,    \
                       c.wcodes, c.wscale, (const int*)c.wsum, c.bias, c.y, s, c.in_stat, (int)c.n, c.in_thr, c.levels, \
                       c.lo_neg, kEps, c.out_current_max, c.bn_scale, c.bn_shift, c.act, c.stat_out, c.residual,       \
                       c.out_thr);                                                                                     \
  }
#define FQ_PWS_LAUNCH_NT(KT_, NT_) FQ_PWS_LAUNCH_NT_R(KT_, false, NT_)
#define FQ_PWS_LAUNCH_NT_R(KT_, RES_, NT_)                                                                             \
  {                                                                                                                    \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_stream_kernel<KT_, RES_, false, false, false, false, NT_>), \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) == hipSuccess;                      \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the streaming kernel");                   \
    hipLaunchKernelGGL((pwconv_stream_kernel<KT_, RES_, false, false, false, false, NT_>), dim3((unsigned)grid),       \
                       dim3(kBlock), lds, c.st, c.x, c.wcodes, c.wscale, (const int*)c.wsum, c.bias, c.y, s, c.in_stat, \
                       (int)c.n, c.in_thr, c.levels, c.lo_neg, kEps, c.out_current_max, c.bn_scale, c.bn_shift, c.act,  \
                       c.stat_out, c.residual, c.out_thr);                                                             \
  }
#define FQ_PWS_CASE(KT_)                                                                                               \
  case KT_:                                                                                                            \
    if (out16) FQ_PWS_LAUNCH(KT_, false, true)                                                                         \
    else if (c.residual != nullptr && nt == 3) FQ_PWS_LAUNCH_NT_R(KT_, true, 3)                                        \
    else if (c.residual != nullptr) FQ_PWS_LAUNCH(KT_, true, false)                                                    \
    else if (nt == 3) FQ_PWS_LAUNCH_NT(KT_, 3) else if (nt == 2) FQ_PWS_LAUNCH_NT(KT_, 2)                              \
    else if (nt == 1) FQ_PWS_LAUNCH_NT(KT_, 1) else FQ_PWS_LAUNCH(KT_, false, false)                                   \
    break;
#define FQ_PWS_THIN(KT_, RES_, O16_, I16_)                                                                             \
  {                                                                                                                    \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_stream_kernel<KT_, RES_, O16_, I16_, true>),         \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) == hipSuccess;                      \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the streaming kernel");                   \
    hipLaunchKernelGGL((pwconv_stream_kernel<KT_, RES_, O16_, I16_, true>), dim3((unsigned)grid), dim3(kBlock), lds,   \
                       c.st, c.x, c.wcodes, c.wscale, (const int*)c.wsum, c.bias, c.y, s, c.in_stat, (int)c.n, c.in_thr, \
                       c.levels, c.lo_neg, kEps, c.out_current_max, c.bn_scale, c.bn_shift, c.act, c.stat_out,         \
                       c.residual, c.out_thr);                                                                         \
  }
#define FQ_PWS_THIN_DUAL(KT_)                                                                                          \
  {                                                                                                                    \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_stream_kernel<KT_, true, false, true, true, true>),  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) == hipSuccess;                      \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the streaming kernel");                   \
    hipLaunchKernelGGL((pwconv_stream_kernel<KT_, true, false, true, true, true>), dim3((unsigned)grid), dim3(kBlock), lds, \
                       c.st, c.x, c.wcodes, c.wscale, (const int*)c.wsum, c.bias, c.y, s, c.in_stat, (int)c.n, c.in_thr, \
                       c.levels, c.lo_neg, kEps, c.out_current_max, c.bn_scale, c.bn_shift, c.act, c.stat_out,         \
                       c.residual, c.out_thr);                                                                         \
  }
#define FQ_PWS_THIN_CASE(KT_)                                                                                          \
  case KT_:                                                                                                            \
    if (out16 && c.in_c16 && KT_ == 1) FQ_PWS_THIN(1, false, true, true)                                               \
    else if (out16) FQ_PWS_THIN(KT_, false, true, false)                                                               \
    else if (c.in_c16 && c.residual != nullptr) FQ_PWS_THIN(KT_, true, false, true)                                    \
    else if (c.in_c16) FQ_PWS_THIN(KT_, false, false, true)                                                            \
    else if (c.residual != nullptr) FQ_PWS_THIN(KT_, true, false, false) else FQ_PWS_THIN(KT_, false, false, false)    \
    break;
  if (thin && c.y16 != nullptr) {
    if (kt == 2) FQ_PWS_THIN_DUAL(2) else FQ_PWS_THIN_DUAL(4)
  } else if (thin && kt == 8) {
    FQ_PWS_THIN(8, false, true, true)
  } else if (thin && kt == 16) {
    FQ_PWS_THIN(16, false, true, true)
  } else if (thin) {
    switch (kt) {
      FQ_PWS_THIN_CASE(1) FQ_PWS_THIN_CASE(2) FQ_PWS_THIN_CASE(3) FQ_PWS_THIN_CASE(4) FQ_PWS_THIN_CASE(5) FQ_PWS_THIN_CASE(6)
      default: break;
    }
  } else {
    switch (kt) {
      FQ_PWS_CASE(1) FQ_PWS_CASE(2) FQ_PWS_CASE(3) FQ_PWS_CASE(4) FQ_PWS_CASE(6) FQ_PWS_CASE(8)
      default: break;
    }
  }
#undef FQ_PWS_THIN_CASE
#undef FQ_PWS_THIN_DUAL
#undef FQ_PWS_THIN
#undef FQ_PWS_CASE
#undef FQ_PWS_LAUNCH_NT_R
#undef FQ_PWS_LAUNCH_NT
#undef FQ_PWS_LAUNCH
  FQ_LAUNCH_CHECK();
  *taken = true;
  return FQ_OK;
}

}  // namespace fqi
