// libfakequant — K2s pointwise (1x1) convolution on int8 codes for 14x14 planes: one SAMPLE per workgroup, output-stationary
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_pw.h"

namespace {

// K2s: sample form.  The split form (K2m) cuts a 14x14 layer into 32-pixel tiles: 128-byte pieces of every channel plane that
// are not line-aligned (a plane is 784 bytes), requested 4 bytes per lane, every tile quantised once per channel group, the
// whole weight matrix streamed through L1 once per tile; a workgroup lives 13 us, 6 of them waiting for its activations,
// and the layer takes two resident rounds of them (profiles/r2_pw_experiments.txt, E).  Here a workgroup of eight wavefronts
// owns one sample and NCH output channels, and runs ONE pipelined loop over the K / 32 channel chunks:
//   load     a chunk is 32 whole planes = 25088 contiguous bytes, requested 16 bytes per lane two chunks ahead (lane = 4
//            consecutive pixels of 4 consecutive channels: the 16 values it needs for one 4-byte panel word per pixel)
//   quantise -> LDS panel of the chunk, [pixel][32 codes] with 32 bytes of padding after every fourth pixel (the panel words of
//            a wavefront then fall on different banks, and a B fragment is still one aligned 16-byte read per lane)
//   multiply every wavefront keeps 7 pixel tiles x CTW channel tiles of int32 accumulators (the whole sample) and feeds them the
//            chunk: 7 * CTW MFMAs per chunk and wavefront, A fragments from the fragment-major weight copy (read once per
//            sample and channel group), B fragments from the panel
// one barrier per chunk (two panels), then the epilogue of the split form.  Every activation is read once per channel group,
// in whole cache lines.
constexpr int kSmpHW = 196, kSmpPT = 7;                 // pixels of a plane, 32-pixel tiles covering it
constexpr int kSmpQuads = 49;                           // 4-pixel groups of a plane
constexpr int kSmpPanelWords = 56 * 40;                 // 56 pixel quads (7 tiles) x (4 pixels x 8 words + 8 words of padding)

struct PwSampleGeom {
  int Cin, Cout, CS;         // CS: channel groups (workgroups) per sample
  int CTM;                   // 32-channel tiles present in the weight buffer
  int n;                     // samples
  int zoff;
};

template <int KT, int CTW>
__global__ __launch_bounds__(512, 1) void pwconv_sample_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwSampleGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out) {
  constexpr int NW = 8;
  constexpr int NCH = NW * CTW * 32;                                    // output channels of one workgroup
#ifndef FQ_PWSMP_PF
#define FQ_PWSMP_PF 4
#endif
  constexpr int PF = FQ_PWSMP_PF < KT ? FQ_PWSMP_PF : KT;               // chunks requested ahead (16 registers each)
  __shared__ __attribute__((aligned(16))) unsigned panel[2][kSmpPanelWords];
  __shared__ __attribute__((aligned(16))) float c_sxw[NCH], c_bsc[NCH], c_bsh[NCH], c_bias[NCH];
  __shared__ __attribute__((aligned(16))) int c_zs[NCH];
  __shared__ float red[NW];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pl = lane & 31;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  // workgroup b runs on XCD b % 8: the channel groups of one sample (they read the same activations) share an XCD's L2
  const unsigned b = blockIdx.x;
  const unsigned xcd = b & 7u, slot = b >> 3;                           // slot: position inside the XCD's share
  const unsigned cg = slot % (unsigned)g.CS;
  const unsigned smp = (slot / (unsigned)g.CS) * 8u + xcd;
  if (smp >= (unsigned)g.n) return;
  const int ch0 = (int)cg * NCH;

  PW_STAMP(0);
  const ThresholdReq treq = threshold_request(in_stat, n, in_thr, b == 0);          // first in the memory queue
  // ---- loads: thread -> (kq: 4 channels of the chunk, pq: 4 pixels); 392 of the 512 threads have an item ---------------
  const unsigned item = threadIdx.x;
  const unsigned pq = item >> 3, kq = item & 7u;                        // consecutive lanes: the 8 channel groups of a pixel quad
  const bool ld_lane = pq < (unsigned)kSmpQuads;
  const fq_rsrc xr = make_rsrc(reinterpret_cast<const char*>(x) + (int64_t)smp * g.Cin * (kSmpHW * 4), (int64_t)g.Cin * (kSmpHW * 4));
  const unsigned xo = ld_lane ? (kq * 4u * kSmpHW + pq * 4u) * 4u : 0x80000000u;
  auto issue = [&](int kt, f4 (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
#ifdef FQ_PWSMP_ABLATE_LOADS
    for (int j = 0; j < 4; ++j) v[j] = buf_ld_v4f(xr, xo, (unsigned)((kt & 1) * 32 + j) * (kSmpHW * 4u));   // (two chunks, cache hits)
#else
    for (int j = 0; j < 4; ++j) v[j] = buf_ld_v4f(xr, xo, (unsigned)(kt * 32 + j) * (kSmpHW * 4u));
#endif
  };
  f4 buf[PF][4];
#pragma unroll
  for (int i = 0; i < PF; ++i) issue(i, buf[i]);
  FQ_PIN();
  // ---- A fragments: wavefront w multiplies channel tiles ct = w * CTW + c of the group ---------------------------------------
  const int ctl0 = wave * CTW;
  const int ctg0 = (int)cg * NW * CTW + ctl0;
  const int ct_here = g.CTM - ctg0 < CTW ? (g.CTM - ctg0 < 0 ? 0 : g.CTM - ctg0) : CTW;
  const fq_rsrc wr = make_rsrc(wfrag + (((int64_t)ctg0 * KT) << 10), (int64_t)ct_here * KT * 1024);
  const unsigned loff = (unsigned)lane * 16u;
  auto a_frag = [&](int c, int kt) __attribute__((always_inline)) {
    return buf_ld_v4i(wr, loff, (unsigned)((c * KT + kt) << 10));
  };
  constexpr int AD = 2;                                                 // A fragments requested ahead (chunks)
  v4i ring[AD + 1][CTW];
#pragma unroll
  for (int d = 0; d < AD; ++d)
#pragma unroll
    for (int c = 0; c < CTW; ++c) ring[d][c] = a_frag(c, d < KT ? d : KT - 1);
  FQ_PIN();
  // ---- threshold, constants ----------------------------------------------------------------------------------------------------
  const float max_ = threshold_finish(treq, in_stat, n, in_thr, cur_max_out, b == 0);
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  const float sx = q.scale;
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
  // panel word of (pixel p, channel group kq): 40 words per pixel quad = 4 pixels x 8 words + 8 words of padding
  // Threads without an item (pixel quads 49..63) loaded zeros; they write the code of 0 into the seven padding quads of the
  // last tile (pixels 196..223, whose products are never stored) - unconditional stores keep the quantiser in the same basic
  // block as the MFMAs, which is what lets the two interleave.
  const unsigned pw_off = (ld_lane ? pq : 49u + (pq - 49u) % 7u) * 40u + kq;      // + 8 * (pixel inside the quad)
  auto quant_to_panel = [&](int kt, const f4 (&v)[4]) __attribute__((always_inline)) {
    // v[j] = channel 4 kq + j, pixels 4 pq .. 4 pq + 3; one word per pixel = the four channels' codes
    unsigned* dst = &panel[kt & 1][pw_off];
    const int ub = 128 - g.zoff;
    const unsigned w0 = (unsigned)pack4_codes(fq_code_int(v[0].x, q), fq_code_int(v[1].x, q), fq_code_int(v[2].x, q), fq_code_int(v[3].x, q), ub);
    const unsigned w1 = (unsigned)pack4_codes(fq_code_int(v[0].y, q), fq_code_int(v[1].y, q), fq_code_int(v[2].y, q), fq_code_int(v[3].y, q), ub);
    const unsigned w2 = (unsigned)pack4_codes(fq_code_int(v[0].z, q), fq_code_int(v[1].z, q), fq_code_int(v[2].z, q), fq_code_int(v[3].z, q), ub);
    const unsigned w3 = (unsigned)pack4_codes(fq_code_int(v[0].w, q), fq_code_int(v[1].w, q), fq_code_int(v[2].w, q), fq_code_int(v[3].w, q), ub);
    dst[0] = w0;
    dst[8] = w1;
    dst[16] = w2;
    dst[24] = w3;
  };
  // B fragment of pixel tile pt for this lane: pixel 32 pt + pl, bytes 16 h .. 16 h + 15 of its 32 codes
  const unsigned bq_off = ((unsigned)pl >> 2) * 40u + ((unsigned)pl & 3u) * 8u + 4u * (unsigned)h;      // + pt * 8 quads * 40
  v16i acc[kSmpPT][CTW];
#pragma unroll
  for (int pt = 0; pt < kSmpPT; ++pt)
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[pt][c][i] = 0;

  // The loop is software-pipelined inside every wavefront: the MFMAs of chunk kt (7 x 64 cycles of the matrix pipe) are
  // issued alternately with slices of the quantisation of chunk kt + 1 (its ~170 vector instructions) - with the two phases
  // one after the other, separated by the barrier, the matrix pipe idles while the wavefronts quantise and the vector ALU
  // while they multiply (measured: 1.1 us per chunk = the SUM of the two).  Writing panel[(kt + 1) & 1] during the
  // multiplication of chunk kt is safe: its last readers (chunk kt - 1) are behind the barrier of this iteration.
  quant_to_panel(0, buf[0]);
  if (PF < KT) issue(PF, buf[0]);
  FQ_PIN();
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    __syncthreads();                                                    // panel[kt & 1] complete (and, first time, the constants)
    if (kt == 0) PW_STAMP(6);
    if (kt == KT / 2) PW_STAMP(7);
    const unsigned* pb = &panel[kt & 1][bq_off];
    v4i bfrag[kSmpPT];
#pragma unroll
    for (int pt = 0; pt < kSmpPT; ++pt) bfrag[pt] = *reinterpret_cast<const v4i*>(pb + pt * 320);
#pragma unroll
    for (int pt = 0; pt < kSmpPT; ++pt)
#pragma unroll
      for (int c = 0; c < CTW; ++c)
        acc[pt][c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ring[kt % (AD + 1)][c], bfrag[pt], acc[pt][c], 0, 0, 0);
    if (kt + 1 < KT) {
      quant_to_panel(kt + 1, buf[(kt + 1) % PF]);
      if (kt + 1 + PF < KT) issue(kt + 1 + PF, buf[(kt + 1) % PF]);
    }
    if (kt + AD < KT) {
#pragma unroll
      for (int c = 0; c < CTW; ++c) ring[(kt + AD) % (AD + 1)][c] = a_frag(c, kt + AD);
    }
    // the schedule: B fragments, then one MFMA followed by a slice of vector work, seven times
    __builtin_amdgcn_sched_group_barrier(0x100, kSmpPT, 0);             // DS reads
#pragma unroll
    for (int pt = 0; pt < kSmpPT * CTW; ++pt) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 26, 0);               // ~1/7 of the quantiser's vector instructions
    }
    FQ_PIN();
  }
  PW_STAMP(2);
  // ---- epilogue: lane = pixel, register = channel 8 gq + 4 h + r of the tile ----------------------------------------------------
  const int cvalid = g.Cout - (ch0 + ctl0 * 32);
  const unsigned plane4 = kSmpHW * 4u;
  float m = 0.0f;
  auto epilogue = [&](auto fast_c) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_c)::value;                      // BatchNorm + ReLU, no bias: fixed at compile time
    const int64_t y_bytes = (int64_t)g.Cout * plane4 - (int64_t)(ch0 + ctl0 * 32) * plane4;
    const fq_rsrc yr = make_rsrc(reinterpret_cast<char*>(y) + ((int64_t)smp * g.Cout + ch0 + ctl0 * 32) * plane4, y_bytes);
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int cv = cvalid - c * 32;                                   // valid channels of this tile (wave-uniform)
      if (cv <= 0) continue;
      const int cb = (ctl0 + c) * 32 + 4 * h;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int c0 = cb + 8 * gq;
        const v4i zs = *reinterpret_cast<const v4i*>(c_zs + c0);
        const f4 sxw = *reinterpret_cast<const f4*>(c_sxw + c0);
        const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
        const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
        const f4 bch = *reinterpret_cast<const f4*>(c_bias + c0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ch_ok = 8 * gq + 4 * h + r < cv;
#pragma unroll
          for (int pt = 0; pt < kSmpPT; ++pt) {
            float v = (float)(acc[pt][c][4 * gq + r] + zs[r]) * sxw[r];
            if (FAST) {
              v = v * bsc[r];
              v = v + bsh[r];
              v = fmaxf(v, 0.0f);
            } else {
              if (bias != nullptr) v = v + bch[r];
              if (has_bn) {
                v = v * bsc[r];
                v = v + bsh[r];
              }
              v = act_rt(v, act);
            }
            const unsigned p = 32u * pt + (unsigned)pl;
            const bool ok = ch_ok && p < (unsigned)kSmpHW;
            buf_st_f32(yr, ok ? (unsigned)(4 * h) * plane4 + p * 4u : 0x80000000u, (unsigned)(c * 32 + 8 * gq + r) * plane4, v);
            m = fmaxf(m, ok ? fabsf(v) : 0.0f);
          }
        }
      }
    }
  };
  if (cvalid > 0) {
    if (bias == nullptr && has_bn && act == FQ_ACT_RELU) epilogue(std::true_type{});
    else epilogue(std::false_type{});
  }
  PW_STAMP(3);
  if (has_stat) {                                                       // the whole workgroup is one sample
    const float wm = wave_max_nonneg(m);
    if (lane == 0) red[wave] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = red[0];
#pragma unroll
      for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
      if (__float_as_uint(t) != 0u) atomic_max_f32(stat_out + smp, t);
    }
  }
  PW_STAMP(4);
  PW_STAMP(5);
}

}  // namespace

namespace fqi {

// sample form (K2s): 14x14 planes, stride 1, no residual operand, Cin a multiple of 32 with K / 32 in {8, 16}, Cout a multiple
// of 256.  grid = samples x channel groups (rounded to whole rounds over the 8 XCDs).
int pw_try_sample(const PwCall& a, bool* taken) {
  *taken = false;
  static const int mode = env_int("FQ_PWSMP", 1);                       // tuning: 0 never, 1 by shape
  const int kt = (int)(a.cin_pad / 32);
  const bool shape_ok = a.hw == kSmpHW && a.stride == 1 && a.residual == nullptr && a.cin == a.cin_pad && (kt == 8 || kt == 16) &&
                        a.cout % 256 == 0 && a.n < (1 << 20) && aligned16(a.x);
  if (!shape_ok || !(a.form == 7 || (a.form == 0 && mode == 1))) return FQ_OK;
  const int64_t rows_pad = (a.cout + 31) / 32 * 32;
  // one channel tile per wavefront = 256 channels per workgroup: 7 x 16 accumulator registers per lane (two tiles would
  // need 224 of the 256 a wavefront has at two per SIMD)
  const int ctw = 1;
  PwSampleGeom t;
  t.Cin = (int)a.cin;
  t.Cout = (int)a.cout;
  t.CS = (int)(a.cout / (256 * ctw));
  t.CTM = (int)(rows_pad / 32);
  t.n = (int)a.n;
  t.zoff = a.zoff;
  const int64_t per_xcd = (a.n + 7) / 8 * t.CS;
  const int64_t grid = per_xcd * 8;
  const int8_t* wfrag = a.wcodes + rows_pad * a.cin_pad;                // second half of fq_weight_codes' buffer
  if (int rc = pw_zero_stat(a)) return rc;
#define FQ_PWSMP_CASE(KT_, CTW_)                                                                                       \
  if (kt == KT_ && ctw == CTW_)                                                                                        \
    hipLaunchKernelGGL((pwconv_sample_kernel<KT_, CTW_>), dim3((unsigned)grid), dim3(512), 0, a.st, a.x, wfrag, a.wscale, \
                       (const int*)a.wsum, a.bias, a.y, t, a.in_stat, (int)a.n, a.in_thr, a.levels, a.lo_neg, kEps,     \
                       a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out);
  FQ_PWSMP_CASE(8, 1) FQ_PWSMP_CASE(16, 1)
#undef FQ_PWSMP_CASE
  FQ_LAUNCH_CHECK();
  *taken = true;
  return FQ_OK;
}

}  // namespace fqi
