// libfakequant — K2r pointwise (1x1) convolution on int8 codes for 14x14 planes: one SAMPLE per workgroup, output-stationary
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_pw.h"

namespace {

// K2r: sample form.  The split form (K2m) cuts a 14x14 layer into 32-pixel tiles: 128-byte pieces of every channel plane that
// are not line-aligned (a plane is 784 bytes), requested 4 bytes per lane, every tile quantised once per channel group, the
// whole weight matrix streamed through L1 once per tile; a workgroup lives 13 us, 6 of them waiting for its activations,
// and the layer takes two resident rounds of them (profiles/r2_pw_experiments.txt, E).  Here a workgroup of eight wavefronts
// owns HALF A SAMPLE (pixels 0..99 or 100..195: whole groups of four pixels) and 512 output channels, keeps all of its
// 4 pixel tiles x 2 channel tiles x 8 wavefronts of int32 accumulators in registers, and runs ONE pipelined loop over the
// K / 32 channel chunks:
//   load     a chunk is 32 planes; a thread owns 4 consecutive pixels of 2 consecutive
//            channels - two 16-byte loads, requested PF chunks ahead
//   quantise -> LDS panel of the chunk, [channel half h][pixel][16 codes] (round 3): a B fragment is one aligned 16-byte read per
//            lane at 2048 h + 16 pixel, and the 16 lanes the LDS serves together for a ds_read_b128 - {0-3, 12-15, 20-27},
//            {4-11, 16-19, 28-31} and the same with h = 1 (MI355X_MICROARCH.md, LDS) - hold 16 different pixels mod 16, i.e. all 64
//            banks once: conflict-free without padding.  (Round 2's [pixel][32 codes] + 32 bytes of padding per pixel quad was
//            laid out for contiguous 16-lane groups and measured 0.59 conflict cycles per active LDS cycle.)
//   multiply 8 MFMAs per chunk and wavefront, A fragments from the fragment-major weight copy, B fragments from the panel;
//            the MFMAs of chunk kt are issued alternately with slices of the quantisation of chunk kt + 1
// one barrier per chunk (two panels), then the epilogue of the split form.  Every activation is read ONCE and quantised ONCE,
// the vector ALU (the quantiser is ~8.5 instructions per value at ~4.3 cycles each, tools/valu_probe.hip) and the matrix
// pipe work at the same time.
constexpr int kSmpPanelWords = 2 * 128 * 4;             // 2 channel halves x 128 pixels x 16 codes
// tools/pw_ablate.py: -DFQ_PWSMP_ABL=<bits> removes one ingredient at a time (results are then WRONG; timing only):
// 1 MFMAs, 2 quantiser arithmetic, 4 barrier per chunk, 8 activation loads, 16 output stores, 32 A-fragment loads, 64 whole loop
#ifndef FQ_PWSMP_ABL
#define FQ_PWSMP_ABL 0
#endif
#ifndef FQ_PWSMP_LB4
#define FQ_PWSMP_LB4 1
#endif

struct PwSampleGeom {
  int Cin, Cout, CS;         // CS: channel groups of 256 * CTW
  int CTM;                   // 32-channel tiles present in the weight buffer
  int n;                     // samples
  int zoff;
  int HW;                    // pixels of a plane (a multiple of 4)
  int nb, qbase, qextra;     // pixel blocks per plane: block k holds qbase + (k < qextra) groups of four pixels (24..32)
};

// (one channel tile per wavefront: 64 accumulator registers - built for two workgroups per CU, whose phases then overlap)
// PT: 32-pixel tiles of a block - 4 (96..128 pixels), or 2 for planes of fewer than 64 pixels taken whole (7x7: the plane is
// not a multiple of four pixels; its last pixel is requested by a 4-byte load of its own).
// RES: a residual operand of y's shape is added after BatchNorm, before the activation (the shortcut of a ResNet unit).
template <int KT, int CTW, int PT, bool RES>
__global__ __launch_bounds__(512, (CTW == 1 && PT == 4 && FQ_PWSMP_LB4) ? 4 : ((CTW == 1 || PT == 2) ? 2 : 1)) void pwconv_sample_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwSampleGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out, const float* __restrict__ residual) {
  // eight wavefronts, two per SIMD (256 registers each): 4 x 2 accumulator tiles = 128 registers.  (Four wavefronts with
  // 4 x 4 tiles and the whole register file each were tried: 12.4 us for the chunk loop instead of 15.9, but a lone wavefront
  // per SIMD exposes every latency of the set-up and the epilogue - 4.7 + 8.4 us instead of 3.7 + 5.1.)
  constexpr int NW = 8;
  constexpr int NCH = NW * CTW * 32;                                    // output channels of one workgroup (256 or 512)
#ifndef FQ_PWSMP_HEAD
#define FQ_PWSMP_HEAD 40
#define FQ_PWSMP_SLICE 5
#endif
#ifndef FQ_PWSMP_HEAD_NN
#define FQ_PWSMP_HEAD_NN 24
#define FQ_PWSMP_SLICE_NN 4
#endif
#ifndef FQ_PWSMP_PF
#define FQ_PWSMP_PF 4
#endif
  // chunks requested ahead (8 registers each); one fewer with one channel tile per wavefront, which then fits 128 registers
  // = two workgroups per CU
  // (128 registers with one channel tile per wavefront = TWO workgroups per CU, whose load and store phases then overlap:
  // 256 -> 256 @28x28 43.9 -> 39.3 us; K / 32 >= 16 only fits them with two chunks ahead)
  constexpr int PF_ = PT == 2 ? 2 : (CTW == 1 ? (KT >= 16 ? FQ_PWSMP_PF - 2 : FQ_PWSMP_PF - 1) : FQ_PWSMP_PF);   // (small planes: a chunk is 6 KB)
  constexpr int PF = PF_ < KT ? PF_ : KT;
  __shared__ __attribute__((aligned(16))) unsigned panel[2][kSmpPanelWords];
  __shared__ __attribute__((aligned(16))) float c_sxw[NCH], c_bsc[NCH], c_bsh[NCH], c_bias[NCH];
  __shared__ __attribute__((aligned(16))) int c_zs[NCH];
  __shared__ float red[NW];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pl = lane & 31;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  // workgroup b runs on XCD b % 8; slot = its position inside the XCD's share: (sample, channel group, pixel half)
  const unsigned b = blockIdx.x;
  const unsigned xcd = b & 7u, slot = b >> 3;
  const unsigned ph = slot % (unsigned)g.nb, cg = (slot / (unsigned)g.nb) % (unsigned)g.CS;
  const unsigned smp = (slot / (unsigned)(g.nb * g.CS)) * 8u + xcd;
  if (smp >= (unsigned)g.n) return;
  const int ch0 = (int)cg * NCH;
  // this block: nquad groups of four pixels from pixel pix0 on (14x14: two blocks of 25 and 24; 28x28: seven of 28)
  const unsigned nquad = (unsigned)g.qbase + (ph < (unsigned)g.qextra ? 1u : 0u);
  const unsigned pix0 = (ph * (unsigned)g.qbase + (ph < (unsigned)g.qextra ? ph : (unsigned)g.qextra)) * 4u;
  const unsigned tail = (unsigned)g.HW & 3u;                            // (only whole planes may be ragged: nb == 1)
  const unsigned npix = tail ? (unsigned)g.HW : nquad * 4u;
  const unsigned plane4 = (unsigned)g.HW * 4u;                          // bytes of a plane

  PW_STAMP(0);
  const ThresholdReq treq = threshold_request(in_stat, n, in_thr, b == 0);          // first in the memory queue
  // ---- loads: thread -> (kq: 2 channels of the chunk, pq: 4 pixels).  A wavefront covers 8 channel pairs x 8 pixel quads:
  // per load instruction 8 channel rows x 128 contiguous bytes ------------------------------------------------------------------
  const unsigned kq = ((unsigned)wave & 1u) * 8u + ((unsigned)lane & 7u);           // channel pair 0..15
  const unsigned pq = ((unsigned)wave >> 1) * 8u + ((unsigned)lane >> 3);           // pixel quad 0..31
  const bool ld_lane = pq < nquad;
  const fq_rsrc xr = make_rsrc(reinterpret_cast<const char*>(x) + (int64_t)smp * g.Cin * plane4, (int64_t)g.Cin * plane4);
  // a ragged plane's last group holds `tail` pixels: one 4-byte load per pixel... only tail == 1 is built (7x7)
  const bool rag_lane = tail != 0u && pq + 1u == nquad;
  const unsigned xoff = kq * 2u * plane4 + (pix0 + pq * 4u) * 4u;
  const unsigned xo = (ld_lane && !rag_lane) ? xoff : 0x80000000u;
  const unsigned xo1 = (ld_lane && rag_lane) ? xoff : 0x80000000u;
  struct Chunk {
    f4 v[2];
    float r[PT == 2 ? 2 : 1];                                           // the ragged form's single pixels (0 for every other lane)
  };
  auto issue = [&](int kt, Chunk& c) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (FQ_PWSMP_ABL & 8) {
        c.v[j] = (f4){(float)kt, (float)lane, 1.0f, 2.0f};
        if (PT == 2) c.r[j] = 0.0f;
        continue;
      }
      c.v[j] = buf_ld_v4f(xr, xo, (unsigned)(kt * 32 + j) * plane4);
      if (PT == 2) c.r[j] = buf_ld_f32(xr, xo1, (unsigned)(kt * 32 + j) * plane4);
    }
  };
  Chunk buf[PF];
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
    if (FQ_PWSMP_ABL & 32) return (v4i){c + kt, lane, 3, 4};
    return buf_ld_v4i(wr, loff, (unsigned)((c * KT + kt) << 10));
  };
  constexpr int AD = 2;                                                 // A fragments requested ahead (chunks)
  v4i ring[AD + 1][CTW];
#pragma unroll
  for (int d = 0; d < AD; ++d)
#pragma unroll
    for (int c = 0; c < CTW; ++c) ring[d][c] = a_frag(c, d < KT ? d : KT - 1);
  FQ_PIN();
  // ---- per-channel constants (one channel per thread: NCH == threads), requested BEFORE the threshold is waited for - the
  // only one that needs it is sx * wscale ------------------------------------------------------------------------------------------
  static_assert(NCH <= NW * 64, "one channel per thread");
  const int ic = ch0 + (int)(threadIdx.x < NCH ? threadIdx.x : 0u);     // < Cout (host: Cout % NCH == 0)
  const float k_ws = wscale[ic];
  const int k_zs = g.zoff * wsum[ic];
  const float k_bias = bias != nullptr ? bias[ic] : 0.0f;
  const float k_bsc = has_bn ? bn_scale[ic] : 1.0f;
  const float k_bsh = has_bn ? bn_shift[ic] : 0.0f;
  FQ_PIN();
  const float max_ = threshold_finish(treq, in_stat, n, in_thr, cur_max_out, b == 0);
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  const float sx = q.scale;
  if (threadIdx.x < NCH) {
    c_sxw[threadIdx.x] = sx * k_ws;
    c_zs[threadIdx.x] = k_zs;
    c_bias[threadIdx.x] = k_bias;
    c_bsc[threadIdx.x] = k_bsc;
    c_bsh[threadIdx.x] = k_bsh;
  }
  PW_STAMP(1);
  // panel: [h][pixel][16 codes]; a thread writes the two codes (16 bits) of its channel pair for each of its four pixels: a
  // wavefront's 2-byte stores fall into 16 words spread over 16 banks, two lanes per word and two words per bank (a 2-way
  // conflict costs a store nothing: its cycles are set by moving address and data to the LDS).  Threads without an item
  // (pixel quads past the half plane) loaded zeros and write the code of 0 for the padding pixels of the last tile (whose
  // products are never stored) - unconditional stores keep the quantiser in the same basic block as the MFMAs, which is
  // what lets the two interleave.
  const unsigned pw_off = (kq >> 3) * 2048u + pq * 64u + (kq & 7u) * 2u;            // bytes; + 16 * (pixel inside the quad)
  const int ub = 128 - g.zoff;
  const unsigned nn_xor16 = fq_nonneg_xor(ub) & 0xFFFFu;
  auto quant_to_panel = [&](int kt, const Chunk& c, auto nn_c) __attribute__((always_inline)) {
    f4 v[2] = {c.v[0], c.v[1]};
    if (PT == 2) {                                                      // (whole groups got 0 in r, the ragged lane 0 in v)
      v[0].x = rag_lane ? c.r[0] : v[0].x;
      v[1].x = rag_lane ? c.r[1] : v[1].x;
    }
    unsigned char* dst = reinterpret_cast<unsigned char*>(panel[kt & 1]) + pw_off;
    auto pair = [&](float a, float b2) -> unsigned short {
      if (FQ_PWSMP_ABL & 2) return (unsigned short)(__float_as_uint(a) ^ (__float_as_uint(b2) >> 7));
      if (decltype(nn_c)::value) {                                      // 5-instruction quantiser of non-negative quotients
        const unsigned u = (unsigned)fq_code_nonneg(a, q) | ((unsigned)fq_code_nonneg(b2, q) << 8);
        return (unsigned short)(u ^ nn_xor16);
      }
      const unsigned u = (unsigned)(fq_code_int(a, q) + ub) | ((unsigned)(fq_code_int(b2, q) + ub) << 8);
      return (unsigned short)(u ^ 0x8080u);
    };
    *reinterpret_cast<unsigned short*>(dst) = pair(v[0].x, v[1].x);
    *reinterpret_cast<unsigned short*>(dst + 16) = pair(v[0].y, v[1].y);
    *reinterpret_cast<unsigned short*>(dst + 32) = pair(v[0].z, v[1].z);
    *reinterpret_cast<unsigned short*>(dst + 48) = pair(v[0].w, v[1].w);
  };
  // B fragment of pixel tile pt for this lane: pixel 32 pt + pl of the block, codes 16 h .. 16 h + 15 of the chunk
  const unsigned bq_off = (unsigned)h * 512u + (unsigned)pl * 4u;                  // words; + pt * 128
  v16i acc[PT][CTW];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[pt][c][i] = 0;

  // The loop is software-pipelined inside every wavefront: the MFMAs of chunk kt are issued alternately with slices of the
  // quantisation of chunk kt + 1 - with the two phases one after the other, separated by the barrier, the
  // matrix pipe idles while the wavefronts quantise and the vector ALU while they multiply (measured on the first version of
  // this form: the chunk time was the SUM of the two).  Writing panel[(kt + 1) & 1] during the multiplication of chunk kt is
  // safe: its last readers (chunk kt - 1) are behind the barrier of this iteration.
  auto chunk_loop = [&](auto nn_c) __attribute__((always_inline)) {
    constexpr bool NN = decltype(nn_c)::value;
    // vector instructions per chunk: ~56 with the short quantiser, ~80 with the general one
    constexpr int HEAD = NN ? FQ_PWSMP_HEAD_NN : FQ_PWSMP_HEAD, SLICE = NN ? FQ_PWSMP_SLICE_NN : FQ_PWSMP_SLICE;
    quant_to_panel(0, buf[0], nn_c);
    if (PF < KT) issue(PF, buf[0]);
    FQ_PIN();
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      if (!(FQ_PWSMP_ABL & 4) || kt == 0)
        __syncthreads();                                                // panel[kt & 1] complete (and, first time, the constants)
      if (kt == 0) PW_STAMP(6);
      if (kt == KT / 2) PW_STAMP(7);
      const unsigned* pb = &panel[kt & 1][bq_off];
      v4i bfrag[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) bfrag[pt] = *reinterpret_cast<const v4i*>(pb + pt * 128);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          if (FQ_PWSMP_ABL & 1) acc[pt][c][kt & 15] += bfrag[pt][kt & 3] ^ ring[kt % (AD + 1)][c][kt & 3];
          else acc[pt][c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ring[kt % (AD + 1)][c], bfrag[pt], acc[pt][c], 0, 0, 0);
        }
      if (kt + 1 < KT) {
        quant_to_panel(kt + 1, buf[(kt + 1) % PF], nn_c);
        if (kt + 1 + PF < KT) issue(kt + 1 + PF, buf[(kt + 1) % PF]);
      }
      if (kt + AD < KT) {
#pragma unroll
        for (int c = 0; c < CTW; ++c) ring[(kt + AD) % (AD + 1)][c] = a_frag(c, kt + AD);
      }
      // the schedule: B fragments, then one MFMA followed by a slice of vector work, eight times
      __builtin_amdgcn_sched_group_barrier(0x100, PT, 0);               // DS reads
      __builtin_amdgcn_sched_group_barrier(0x002, HEAD, 0);             // vector work behind which the B fragments arrive
#pragma unroll
      for (int i = 0; i < PT * CTW; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, SLICE, 0);          // a slice of the quantiser's vector instructions
      }
      FQ_PIN();
    }
  };
  if (FQ_PWSMP_ABL & 64) __syncthreads();
  else if (fq_nonneg(q)) chunk_loop(std::true_type{});
  else chunk_loop(std::false_type{});
  PW_STAMP(2);
  // ---- epilogue: lane = pixel, register = channel 8 gq + 4 h + r of the tile ----------------------------------------------------
  const int cvalid = g.Cout - (ch0 + ctl0 * 32);
  float m = 0.0f;
  auto epilogue = [&](auto fast_c) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_c)::value;                      // BatchNorm + ReLU, no bias: fixed at compile time
    const int64_t y_bytes = (int64_t)g.Cout * plane4 - (int64_t)(ch0 + ctl0 * 32) * plane4;
    const fq_rsrc yr = make_rsrc(reinterpret_cast<char*>(y) + ((int64_t)smp * g.Cout + ch0 + ctl0 * 32) * plane4, y_bytes);
    const fq_rsrc rr = make_rsrc(reinterpret_cast<const char*>(RES ? residual : y) + ((int64_t)smp * g.Cout + ch0 + ctl0 * 32) * plane4,
                                 RES ? y_bytes : 0);
    // channel tiles are whole (host: Cout % 512 == 0); only the LAST pixel tile of a half has pixels past its end (offset out
    // of range: the store is dropped) - the other three take no mask at all
    const bool last_ok = 32u * (PT - 1) + (unsigned)pl < npix;
    const unsigned po0 = (unsigned)(4 * h) * plane4 + (pix0 + (unsigned)pl) * 4u;
    const unsigned po_last = last_ok ? po0 + 32u * (PT - 1) * 4u : 0x80000000u;
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int cb = (ctl0 + c) * 32 + 4 * h;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int c0 = cb + 8 * gq;
        const v4i zs = *reinterpret_cast<const v4i*>(c_zs + c0);
        const f4 sxw = *reinterpret_cast<const f4*>(c_sxw + c0);
        const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
        const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
        const f4 bch = *reinterpret_cast<const f4*>(c_bias + c0);
        float res[RES ? 4 * PT : 1];
        if (RES) {                                 // the 4 x PT values of this group in flight before the first use
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
              res[RES ? r * PT + pt : 0] = buf_ld_f32(rr, pt == PT - 1 ? po_last : po0 + 32u * pt * 4u,
                                                     (unsigned)(c * 32 + 8 * gq + r) * plane4);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          typedef float f2 __attribute__((ext_vector_type(2)));
          // pixel tiles two at a time: they share the channel's constants, so scale / BatchNorm run as packed fp32
          // instructions (two IEEE operations each: the same values as the scalar form)
#pragma unroll
          for (int pp = 0; pp < PT; pp += 2) {
            f2 v = (f2){(float)(acc[pp][c][4 * gq + r] + zs[r]), (float)(acc[pp + 1][c][4 * gq + r] + zs[r])};
            v = v * (f2){sxw[r], sxw[r]};
            if (FAST) {
              v = v * (f2){bsc[r], bsc[r]};
              v = v + (f2){bsh[r], bsh[r]};
              if (RES) v = v + (f2){res[RES ? r * PT + pp : 0], res[RES ? r * PT + pp + 1 : 0]};
              v.x = fmaxf(v.x, 0.0f);
              v.y = fmaxf(v.y, 0.0f);
            } else {
              if (bias != nullptr) v = v + (f2){bch[r], bch[r]};
              if (has_bn) {
                v = v * (f2){bsc[r], bsc[r]};
                v = v + (f2){bsh[r], bsh[r]};
              }
              if (RES) v = v + (f2){res[RES ? r * PT + pp : 0], res[RES ? r * PT + pp + 1 : 0]};
              v.x = act_rt(v.x, act);
              v.y = act_rt(v.y, act);
            }
            const unsigned so = (unsigned)(c * 32 + 8 * gq + r) * plane4;
            if (FQ_PWSMP_ABL & 16) {
              m = fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y));
              continue;
            }
            buf_st_f32(yr, po0 + 32u * pp * 4u, so, v.x);
            if (pp + 1 == PT - 1) {
              buf_st_f32(yr, po_last, so, v.y);
              m = fmaxf(fmaxf(m, fabsf(v.x)), last_ok ? fabsf(v.y) : 0.0f);
            } else {
              buf_st_f32(yr, po0 + 32u * (pp + 1) * 4u, so, v.y);
              m = fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y));
            }
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
    // ONE atomic per workgroup: with one per wavefront (4096 atomics on 128 addresses) the kernel got 3.5-6 us slower
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
  PW_STAMP(5);
}

}  // namespace

namespace fqi {

// sample form (K2r): stride 1, no residual operand, Cout a multiple of 256, Cin a multiple of 32, and either
//   planes of a multiple of 4 pixels that cut into blocks of 96..128 pixels (14x14: two, 28x28: seven), K / 32 in {4, 8, 16}, or
//   whole planes of 45..64 pixels with 0 or 1 pixel past a multiple of four (7x7, 8x8), K / 32 in {16, 32}.
// grid = samples x channel groups x pixel blocks (rounded to whole rounds over the 8 XCDs).
int pw_try_sample(const PwCall& a, bool* taken) {
  *taken = false;
  static const int mode = env_int("FQ_PWSMP", 1);                       // tuning: 0 never, 1 blocked planes up to 1024 pixels, 2 every shape it takes
  const int kt = (int)(a.cin_pad / 32);
  const bool small = a.hw <= 64;                                        // one block of two pixel tiles
  const int64_t quads = (a.hw + 3) / 4;
  const int nb = small ? 1 : (int)((quads + 31) / 32);                  // fewest blocks of at most 32 pixel groups
  const bool plane_ok = small ? (a.hw >= 45 && a.hw % 4 <= 1 && (kt == 16 || kt == 32) && a.residual == nullptr)
                              : (a.hw % 4 == 0 && quads / nb >= 24 && (kt == 4 || kt == 8 || kt == 16 || kt == 32));
  const bool shape_ok = plane_ok && a.stride == 1 && a.cin == a.cin_pad && a.cout % 256 == 0 &&
                        a.n < (1 << 20) && a.cin * a.hw * 4 < (1ll << 31) && (small || aligned16(a.x));
  // by shape: the small and middle planes (measured in the model against the split form: 512 -> 512 @14x14 32.0 -> 25.2 us,
  // 256 -> 512 @14x14 24.0 -> 20.7, 256 -> 256 @28x28 48.0 -> 39.7, 128 -> 256 @28x28 38.8 -> 33.7); the streaming form keeps
  // the large planes
  // (whole small planes are built and tested, but not chosen: 1024 -> 1024 @7x7 27.9 us against the split form's 26.6,
  // 512 -> 1024 @7x7 19.9 against 20.4 - 32 chunks with a barrier each)
  // (nor with a residual operand: ResNet-50 128 -> 512 @28x28 103.9 us against the split form's 99.0, 256 -> 1024 @14x14 55.1
  // against 51.0; without one, 1024 -> 256 @14x14: 30.2 against 34.8)
  const bool by_shape = a.form == 0 && (mode == 2 || (mode == 1 && a.hw <= 1024 && !small && a.residual == nullptr));
  if (!shape_ok || !(a.form == 7 || by_shape)) return FQ_OK;
  const int64_t rows_pad = (a.cout + 31) / 32 * 32;
  static const int ctw_tune = env_int("FQ_PWSMP_CTW", 0);               // tuning: 1 = 256 channels per workgroup everywhere
  const int ctw = (a.cout % 512 == 0 && ctw_tune != 1) ? 2 : 1;
  PwSampleGeom t;
  t.Cin = (int)a.cin;
  t.Cout = (int)a.cout;
  t.CS = (int)(a.cout / (256 * ctw));
  t.CTM = (int)(rows_pad / 32);
  t.n = (int)a.n;
  t.zoff = a.zoff;
  t.HW = (int)a.hw;
  t.nb = nb;
  t.qbase = (int)(quads / nb);
  t.qextra = (int)(quads % nb);
  const int pt = small ? 2 : 4;
  const int64_t grid = (a.n + 7) / 8 * t.CS * nb * 8;
  FQ_REQUIRE(grid < (1ll << 31), "fq_pwconv_i8: too many workgroups for the sample form");
  const int8_t* wfrag = a.wcodes + rows_pad * a.cin_pad;                // second half of fq_weight_codes' buffer
  if (int rc = pw_zero_stat(a)) return rc;
  const bool res = a.residual != nullptr;
  bool launched = false;
#define FQ_PWSMP_CASE_R(KT_, CTW_, PT_, RES_)                                                                          \
  if (kt == KT_ && ctw == CTW_ && pt == PT_ && res == RES_) {                                                          \
    hipLaunchKernelGGL((pwconv_sample_kernel<KT_, CTW_, PT_, RES_>), dim3((unsigned)grid), dim3(512), 0, a.st, a.x,     \
                       wfrag, a.wscale, (const int*)a.wsum, a.bias, a.y, t, a.in_stat, (int)a.n, a.in_thr, a.levels,    \
                       a.lo_neg, kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out, a.residual);      \
    launched = true;                                                                                                   \
  }
#define FQ_PWSMP_CASE(KT_, CTW_, PT_) FQ_PWSMP_CASE_R(KT_, CTW_, PT_, false)
  FQ_PWSMP_CASE(4, 1, 4) FQ_PWSMP_CASE(4, 2, 4) FQ_PWSMP_CASE(8, 1, 4) FQ_PWSMP_CASE(8, 2, 4) FQ_PWSMP_CASE(16, 1, 4)
  FQ_PWSMP_CASE(16, 2, 4) FQ_PWSMP_CASE(32, 1, 4) FQ_PWSMP_CASE(32, 2, 4)
  FQ_PWSMP_CASE(16, 1, 2) FQ_PWSMP_CASE(16, 2, 2) FQ_PWSMP_CASE(32, 1, 2) FQ_PWSMP_CASE(32, 2, 2)
  // with a residual operand: the last 1x1 convolutions of the ResNet bottlenecks (128 -> 512 @28x28, 256 -> 1024 @14x14)
  FQ_PWSMP_CASE_R(4, 2, 4, true) FQ_PWSMP_CASE_R(8, 2, 4, true) FQ_PWSMP_CASE_R(16, 2, 4, true)
#undef FQ_PWSMP_CASE
#undef FQ_PWSMP_CASE_R
  if (!launched) {                                                     // (a combination that is not built: another form takes it)
    FQ_REQUIRE(a.form != 7, "fq_pwconv_i8: this K / channel-group / residual combination of the sample form is not built");
    return FQ_OK;
  }
  FQ_LAUNCH_CHECK();
  *taken = true;
  return FQ_OK;
}

}  // namespace fqi
