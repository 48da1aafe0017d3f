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
#ifndef FQ_PWSMP_NTL
#define FQ_PWSMP_NTL 0       // A/B builds: 1 = activation loads with the nontemporal hint (profiles/r5_nt_sweep3.txt)
#endif
#ifndef FQ_PWSMP_LB4
#define FQ_PWSMP_LB4 1
#endif

#ifndef FQ_PWSMP_RESLDS
#define FQ_PWSMP_RESLDS 1    // A/B builds: 0 = the residual operand through registers on every shape
#endif
// one LDS-DMA wave-instruction (fq_pw_pipe.hip, tools/ldsdma_probe.hip): lane l's 16 bytes at (rsrc base + voff + soff) land at LDS
// byte address lds_base + 16 l; a lane whose offset is out of the resource's range gets zeros
__device__ __forceinline__ void smp_dma16(v4i rsrc, unsigned voff, unsigned soff, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
template <int N>
__device__ __forceinline__ void smp_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct PwSampleGeom {
  int Cin, Cout, CS;         // CS: channel groups of 256 * CTW
  int CTM;                   // 32-channel tiles present in the weight buffer
  int n;                     // samples
  int zoff;
  int HW;                    // pixels of a plane (a multiple of 4)
  int nb, qbase, qextra;     // pixel blocks per plane: block k holds qbase + (k < qextra) groups of four pixels (24..32)
};

// (one channel tile per wavefront: 64 accumulator registers - built for two workgroups per CU, whose phases then overlap)
// PT: 32-pixel tiles of an ITEM - 4 (96..128 pixels) or 2 (up to 64 pixels).  A chunk is always 4096 values, 8 per thread:
//     PT = 4: 32 channels x 128 pixels, one K-step of 32 per chunk;  PT = 2: 64 channels x 64 pixels, KS = 2 K-steps per chunk.
// NI: items per workgroup (round 3).  The form is output-stationary: an item's outputs leave at its end, and with one item
//     per workgroup and one workgroup per CU the whole chip first only loads and then only stores (tools/pw_ablate.py,
//     profiles/r3_pw_ablate.txt: 512 -> 512 @14x14 25.1 us; without the stores 18.1, without the loads 20.6, without both 14.2,
//     without MFMAs AND quantiser 23.0 - loads and stores do not overlap).  With NI = 2 the workgroup's block is cut into two
//     items of <= 64 pixels so that the stores of the first are in flight while the second loads - built, bit-exact, and
//     measured SLOWER on every layer (31.1 against 24.8 us on 512 -> 512 @14x14: twice the A fragments from L2, an epilogue
//     in the middle of the MFMA stream): kept as a tuning build only (-DFQ_PWSMP_BUILD_NI2, FQ_PWSMP_NI=2).
//     Planes of fewer than 64 pixels taken whole (7x7; the plane is not a multiple of four pixels: its last pixel is requested
//     by a 4-byte load of its own) are one item of PT = 2.
// RES: a residual operand of y's shape is added after BatchNorm, before the activation (the shortcut of a ResNet unit).
// GAP (round 6, fq_pwconv_i8_gap; whole planes only, PT = 2): the layer's output is not stored - its only reader is the global average
//     pooling behind it (the last 1x1 of the MobileNets, 26 MB written and read back by a 9 us launch of its own).  The epilogue's
//     values go to an LDS tile [channel][pixel]; thread t then adds up channel t's pixels in the order and the precision
//     fq_global_avg_pool_stat adds them (0 .. HW - 1, fp64) and stores the mean: y is (n, Cout), stat_out the per-sample max|mean|.
template <int KT, int CTW, int PT, bool RES, int NI, bool GAP = false>
__global__ __launch_bounds__(512, (CTW == 1 && FQ_PWSMP_LB4) ? 4 : ((CTW == 1 || PT == 2) ? 2 : 1)) void pwconv_sample_kernel(
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
  // RL: the residual operand arrives by LDS-DMA (two channel tiles x four pixel tiles only: 128 accumulator registers leave room
  // for two groups of residual values, i.e. four exposed memory latencies per item - see the epilogue)
  constexpr bool RL = RES && PT == 4 && CTW == 2 && NI == 1 && FQ_PWSMP_RESLDS;
  static_assert(!GAP || (PT == 2 && NI == 1 && !RL), "the pooling epilogue takes whole planes (one item of two pixel tiles)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smp_dyn[];   // RL: 8 x 16 KB, [wavefront][32 channels][128 pixels]
  constexpr int KS = 4 / PT;                                            // K-steps of 32 channels per chunk
  constexpr int KI = KT / KS;                                           // chunks per item
  constexpr int TI = NI * KI;                                           // chunk iterations of the workgroup
  static_assert(PT == 4 || PT == 2, "pixel tiles per item");
  static_assert(KT % KS == 0, "K / 32 must be a multiple of the K-steps per chunk");
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
#ifndef FQ_PWSMP_RA
#define FQ_PWSMP_RA 2                  // groups of residual values in flight (two channel tiles x four pixel tiles: 2 x 16 registers)
#endif
  // chunks requested ahead (8 registers each); fewer with one channel tile per wavefront, which then fits 128 registers
  // = TWO workgroups per CU, whose load and store phases overlap (256 -> 256 @28x28 43.9 -> 39.3 us; K / 32 >= 16 only fits
  // them with two chunks ahead)
  constexpr int PF_ = CTW == 1 ? ((KT >= 16 || NI == 2) ? FQ_PWSMP_PF - 2 : FQ_PWSMP_PF - 1) : FQ_PWSMP_PF;
  constexpr int PF = PF_ < TI ? PF_ : TI;
  __shared__ __attribute__((aligned(16))) unsigned panel[2][kSmpPanelWords];
  __shared__ __attribute__((aligned(16))) float c_sxw[NCH], c_bsc[NCH], c_bsh[NCH], c_bias[NCH];
  __shared__ __attribute__((aligned(16))) int c_zs[NCH];
  __shared__ float red[NW];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pl = lane & 31;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  // workgroup b runs on XCD b % 8; slot = its position inside the XCD's share: (sample, channel group, pixel block)
  const unsigned b = blockIdx.x;
  const unsigned xcd = b & 7u, slot = b >> 3;
  const unsigned ph = slot % (unsigned)g.nb, cg = (slot / (unsigned)g.nb) % (unsigned)g.CS;
  const unsigned smp = (slot / (unsigned)(g.nb * g.CS)) * 8u + xcd;
  if (smp >= (unsigned)g.n) return;
  const int ch0 = (int)cg * NCH;
  // this block: nquad groups of four pixels from pixel pix0 on (14x14: two blocks of 25 and 24; 28x28: seven of 28);
  // item i of the block: quads [iq0[i], iq0[i] + inq[i])
  const unsigned nquad = (unsigned)g.qbase + (ph < (unsigned)g.qextra ? 1u : 0u);
  const unsigned pix0 = (ph * (unsigned)g.qbase + (ph < (unsigned)g.qextra ? ph : (unsigned)g.qextra)) * 4u;
  const unsigned tail = (unsigned)g.HW & 3u;                            // (only whole planes may be ragged: nb == 1, NI == 1)
  const unsigned plane4 = (unsigned)g.HW * 4u;                          // bytes of a plane
  unsigned inq[NI], ipix0[NI], inpix[NI];
  {
    unsigned q0 = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      inq[i] = (nquad - q0 + (unsigned)(NI - i) - 1u) / (unsigned)(NI - i);    // the remaining quads split evenly
      ipix0[i] = pix0 + q0 * 4u;
      inpix[i] = (tail && NI == 1) ? (unsigned)g.HW : inq[i] * 4u;
      q0 += inq[i];
    }
  }

  PW_STAMP(0);
  const ThresholdReq treq = threshold_request(in_stat, n, in_thr, b == 0);          // first in the memory queue
  // ---- loads: thread -> (kq: 2 channels of the chunk, pq: 4 pixels).  A wavefront covers 8 channel pairs x 8 pixel quads:
  // per load instruction 8 channel rows x 128 contiguous bytes ------------------------------------------------------------------
  const unsigned kq = ((unsigned)wave % (2u * KS)) * 8u + ((unsigned)lane & 7u);    // channel pair 0 .. 16 KS - 1
  const unsigned pq = ((unsigned)wave / (2u * KS)) * 8u + ((unsigned)lane >> 3);    // pixel quad 0 .. 8 PT - 1
  const fq_rsrc xr = make_rsrc(reinterpret_cast<const char*>(x) + (int64_t)smp * g.Cin * plane4, (int64_t)g.Cin * plane4);
  // a ragged plane's last group holds `tail` pixels: one 4-byte load per pixel... only tail == 1 is built (7x7)
  const bool rag_lane = NI == 1 && tail != 0u && pq + 1u == nquad;
  // (the item's first pixel goes into the SCALAR offset of its loads: with it in the lane offset the scalar offsets of
  // item 1 equal those of item 0 and hipcc keeps all of them alive across the loop - 106 SGPRs and spills)
  unsigned xo[NI], xo1 = 0x80000000u;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const unsigned off = kq * 2u * plane4 + (pix0 + pq * 4u) * 4u;
    xo[i] = (pq < inq[i] && !rag_lane) ? off : 0x80000000u;
    if (i == 0 && pq < inq[i] && rag_lane) xo1 = off;
  }
  // RL: channel tile 0 of this wavefront's residual values - 32 channels x 128 pixels, 16 KB - is requested NOW, into LDS, and is
  // there long before the epilogue; instruction i brings channels 2 i and 2 i + 1 (lane: channel 2 i + (lane >> 5), four pixels from
  // 4 (lane & 31) on; pixel groups past the block's end are out of range: zeros, never stored)
  const int64_t res_bytes = (int64_t)g.Cout * plane4 - (int64_t)(ch0 + (int)wave * CTW * 32) * plane4;
  const fq_rsrc rr = make_rsrc(reinterpret_cast<const char*>(RES ? residual : x) +
                                   (RES ? ((int64_t)smp * g.Cout + ch0 + (int)wave * CTW * 32) * plane4 : 0), RES ? res_bytes : 0);
  v4i rrd;                                                              // (the same descriptor as four scalars, for the asm statement)
  {
    const unsigned long long rb = (unsigned long long)(size_t)(RES ? residual : x) +
                                  (unsigned long long)(RES ? ((int64_t)smp * g.Cout + ch0 + (int)wave * CTW * 32) * plane4 : 0);
    rrd[0] = __builtin_amdgcn_readfirstlane((int)(rb & 0xFFFFFFFFull));
    rrd[1] = __builtin_amdgcn_readfirstlane((int)(rb >> 32));
    rrd[2] = __builtin_amdgcn_readfirstlane((int)(res_bytes > 0x7FFFFFFFll ? 0x7FFFFFFF : (RES ? res_bytes : 0)));
    rrd[3] = 0x00020000;
  }
  const unsigned rl_base = __builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)smp_dyn + 16384u * (unsigned)wave));
  const unsigned rl_voff = (((unsigned)lane & 31u) < inq[0]) ? ((unsigned)lane >> 5) * plane4 + (ipix0[0] + 4u * ((unsigned)lane & 31u)) * 4u
                                                           : 0x80000000u;
  if (RL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) smp_dma16(rrd, rl_voff, (unsigned)(2 * i) * plane4, rl_base + 1024u * i);
  }
  struct Chunk {
    f4 v[2];
    float r[PT == 2 && NI == 1 ? 2 : 1];                                // the ragged form's single pixels (0 for every other lane)
  };
  constexpr bool RAGGED = PT == 2 && NI == 1;
  auto issue = [&](int it, Chunk& c) __attribute__((always_inline)) {  // `it`: chunk iteration = item * KI + chunk of the item
    const int item = it / KI, ki = it % KI;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (FQ_PWSMP_ABL & 8) {
        c.v[j] = (f4){(float)it, (float)lane, 1.0f, 2.0f};
        if (RAGGED) c.r[j] = 0.0f;
        continue;
      }
      c.v[j] = FQ_PWSMP_NTL ? buf_ld_v4f_nt(xr, xo[item], (unsigned)(ki * KS * 32 + j) * plane4 + (ipix0[item] - pix0) * 4u)
                            : buf_ld_v4f(xr, xo[item], (unsigned)(ki * KS * 32 + j) * plane4 + (ipix0[item] - pix0) * 4u);
      if (RAGGED) c.r[j] = buf_ld_f32(xr, xo1, (unsigned)(ki * KS * 32 + j) * plane4);
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
  v4i ring[AD + 1][KS][CTW];
  auto a_chunk = [&](int it, v4i (&dst)[KS][CTW]) __attribute__((always_inline)) {
    const int ki = (it < TI ? it : TI - 1) % KI;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int c = 0; c < CTW; ++c) dst[ks][c] = a_frag(c, ki * KS + ks);
  };
#pragma unroll
  for (int d = 0; d < AD; ++d) a_chunk(d, ring[d]);
  FQ_PIN();
  // ---- per-channel constants (one channel per thread: NCH == threads), requested BEFORE the threshold is waited for - the
  // only one that needs it is sx * wscale ------------------------------------------------------------------------------------------
  static_assert(NCH <= NW * 64, "one channel per thread");
  const int ic = ch0 + (int)(threadIdx.x < NCH ? threadIdx.x : 0u);     // < Cout (host: Cout % NCH == 0)
  // range mode (nn.Conv2D(quantized=True)): `bias` holds int32 codes that join the integer sum
  const int* ibias = lo_neg_max == kRangeMode ? reinterpret_cast<const int*>(bias) : nullptr;
  const float* fbias = lo_neg_max == kRangeMode ? nullptr : bias;
  const float k_ws = wscale[ic];
  const int k_wsum = wsum[ic];
  const int k_ib = ibias != nullptr ? ibias[ic] : 0;
  const float k_bias = fbias != nullptr ? fbias[ic] : 0.0f;
  const float k_bsc = has_bn ? bn_scale[ic] : 1.0f;
  const float k_bsh = has_bn ? bn_shift[ic] : 0.0f;
  FQ_PIN();
  const float max_ = threshold_finish(treq, in_stat, n, in_thr, cur_max_out, b == 0);
  int zoff = g.zoff;
  const QParams q = make_qparams_rt(max_, levels, lo_neg_max, eps, in_thr, zoff);
  const float sx = q.scale;
  if (threadIdx.x < NCH) {
    c_sxw[threadIdx.x] = sx * k_ws;
    c_zs[threadIdx.x] = zoff * k_wsum + k_ib;
    c_bias[threadIdx.x] = k_bias;
    c_bsc[threadIdx.x] = k_bsc;
    c_bsh[threadIdx.x] = k_bsh;
  }
  PW_STAMP(1);
  // panel: [K-step of the chunk][h][pixel of the item][16 codes]; a thread writes the two codes (16 bits) of its channel pair
  // for each of its four pixels: a wavefront's 2-byte stores fall into 16 words spread over 16 banks, two lanes per word and
  // two words per bank (a 2-way conflict costs a store nothing: its cycles are set by moving address and data to the LDS).
  // Threads without an item (pixel quads past the item) loaded zeros and write the code of 0 for the padding pixels of the
  // last tile (whose products are never stored) - unconditional stores keep the quantiser in the same basic block as the
  // MFMAs, which is what lets the two interleave.
  const unsigned pw_off = ((kq >> 4) * (PT * 64u) + ((kq >> 3) & 1u) * (PT * 32u) + pq * 4u) * 16u + (kq & 7u) * 2u;   // bytes; + 16 * (pixel inside the quad)
  const int ub = 128 - zoff;
  const unsigned nn_xor16 = fq_nonneg_xor(ub) & 0xFFFFu;
  auto quant_to_panel = [&](int it, const Chunk& c, auto nn_c) __attribute__((always_inline)) {
    f4 v[2] = {c.v[0], c.v[1]};
    if (RAGGED) {                                                       // (whole groups got 0 in r, the ragged lane 0 in v)
      v[0].x = rag_lane ? c.r[0] : v[0].x;
      v[1].x = rag_lane ? c.r[1] : v[1].x;
    }
    unsigned char* dst = reinterpret_cast<unsigned char*>(panel[it & 1]) + pw_off;
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
  // B fragment (K-step ks, pixel tile pt) of this lane: pixel 32 pt + pl of the item, codes 16 h .. 16 h + 15 of the step
  const unsigned bq_off = (unsigned)h * (PT * 128u) + (unsigned)pl * 4u;            // words; + ks * PT * 256 + pt * 128
  v16i acc[PT][CTW];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[pt][c][i] = 0;
  };
  zero_acc();

  // ---- epilogue of one item: lane = pixel, register = channel 8 gq + 4 h + r of the tile ---------------------------------------
  const int cvalid = g.Cout - (ch0 + ctl0 * 32);
  float m = 0.0f;
  auto epilogue = [&](int item, auto fast_c) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_c)::value;                      // BatchNorm + ReLU, no bias: fixed at compile time
    const int64_t y_bytes = (int64_t)g.Cout * plane4 - (int64_t)(ch0 + ctl0 * 32) * plane4;
    const fq_rsrc yr = make_rsrc(reinterpret_cast<char*>(y) + ((int64_t)smp * g.Cout + ch0 + ctl0 * 32) * plane4, y_bytes);
    // channel tiles are whole (host: Cout % NCH == 0); only the LAST pixel tile of an item has pixels past its end (offset out
    // of range: the store is dropped) - the others take no mask at all
    const bool last_ok = 32u * (PT - 1) + (unsigned)pl < inpix[item];
    const unsigned po0 = (unsigned)(4 * h) * plane4 + (ipix0[item] + (unsigned)pl) * 4u;
    const unsigned po_last = last_ok ? po0 + 32u * (PT - 1) * 4u : 0x80000000u;
    // the residual operand: the 4 x PT values of RA groups of four channels are in flight ahead of the group that is being
    // stored.  (Through round 5 each group asked for its values right before using them - and, a buffer store being something the
    // compiler may not move a buffer load across, behind the previous group's stores: 4 CTW exposed memory latencies per item with
    // one or two wavefronts per SIMD to hide them.  RA = 2: what fits beside the 128 accumulators without spilling - ResNet-50
    // online +1.3 % images/s; 3: +1.2 % with 20 bytes of scratch; 4 and 6 spill 156 / 308 bytes: -4.8 / -5.0 %,
    // profiles/r6_sample_residual_ab.txt.  Whole planes of 49 pixels - two pixel tiles - hold all eight groups.)
    // RL (two channel tiles x four pixel tiles): no registers at all.  Channel tile 0 has been in LDS since the prologue; as soon as
    // a group of it (8 channels, 4 KB) has been read, the same group of channel tile 1 is requested into its place - four
    // requests that are all in flight while tile 0 is stored.  vmcnt counts loads and stores in issue order: before group gq of
    // tile 1 is read, what may still be outstanding is everything issued after its request - counted below.
    constexpr int NG = CTW * 4;                                         // groups of four channels per wavefront
    constexpr int RA = (!RES || RL) ? 1 : (16 * PT * NG <= 64 ? NG : (PT == 4 ? FQ_PWSMP_RA : NG));
    const float* const rl_rd = reinterpret_cast<const float*>(smp_dyn + 16384u * (unsigned)wave) + (4 * h) * 128 + pl;
    float resv[RES ? RA * 4 * PT : 1];
    auto res_issue = [&](int g_) __attribute__((always_inline)) {
      const int c = g_ / 4, gq = g_ % 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
          resv[RES ? ((g_ % RA) * 4 + r) * PT + pt : 0] =
              buf_ld_f32(rr, pt == PT - 1 ? po_last : po0 + 32u * pt * 4u, (unsigned)(c * 32 + 8 * gq + r) * plane4);
    };
    if (RES && !RL) {
#pragma unroll
      for (int g_ = 0; g_ < RA && g_ < NG; ++g_) res_issue(g_);
      FQ_PIN();
    }
    if (RL) smp_wait_vm<0>();                                           // (the prologue's requests: long done)
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
        if (RL) {
          // what is issued after the request of (tile 1, group gq) - it goes out right behind the LDS reads of (tile 0, group gq),
          // ahead of that group's stores: 16 stores + [4 requests + 16 stores] x (3 - gq) of tile 0, then 16 gq stores of tile 1 =
          // 76 - 4 gq >= 64 vector memory instructions.  The counter holds 63 at most: `vmcnt(63)` is the weakest wait that
          // proves the request complete (it waits for at most 13 stores more than necessary)
          if (c == 1) smp_wait_vm<63>();
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) res[r * PT + pt] = rl_rd[(8 * gq + r) * 128 + 32 * pt];
          if (c == 0) {                            // this group's values are in registers: tile 1's group into its place
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j)
              smp_dma16(rrd, rl_voff, (unsigned)(32 + 8 * gq + 2 * j) * plane4, rl_base + (unsigned)(8 * gq + 2 * j) * 512u);
          }
        } else if (RES) {
#pragma unroll
          for (int i = 0; i < 4 * PT; ++i) res[i] = resv[((c * 4 + gq) % RA) * 4 * PT + i];
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
              if (fbias != nullptr) v = v + (f2){bch[r], bch[r]};
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
            if (GAP) {                               // (PT = 2: pixel tiles 0 and 1; the tile keeps [channel][pixel])
              float* const tp = reinterpret_cast<float*>(smp_dyn) + ((ctl0 + c) * 32 + 8 * gq + 4 * h + r) * g.HW + pl;
              tp[0] = v.x;
              if (last_ok) tp[32] = v.y;
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
        if (RES && !RL && c * 4 + gq + RA < NG) {
          FQ_PIN();
          res_issue(c * 4 + gq + RA);
          FQ_PIN();
        }
      }
    }
  };
  const bool fast_epi = fbias == nullptr && has_bn && act == FQ_ACT_RELU;
  auto finish_item = [&](int item) __attribute__((always_inline)) {
    if (cvalid > 0) {
      if (fast_epi) epilogue(item, std::true_type{});
      else epilogue(item, std::false_type{});
    }
  };

  // The loop is software-pipelined inside every wavefront: the MFMAs of chunk `it` are issued alternately with slices of the
  // quantisation of chunk it + 1 - with the two phases one after the other, separated by the barrier, the
  // matrix pipe idles while the wavefronts quantise and the vector ALU while they multiply (measured on the first version of
  // this form: the chunk time was the SUM of the two).  Writing panel[(it + 1) & 1] during the multiplication of chunk `it` is
  // safe: its last readers (chunk it - 1) are behind the barrier of this iteration.  The chunk sequence runs THROUGH the item
  // boundary: the first chunk of item i + 1 is quantised during the last multiplication of item i, its loads were requested
  // PF chunks earlier, and item i's epilogue sits between the two iterations.
  auto chunk_loop = [&](auto nn_c) __attribute__((always_inline)) {
    constexpr bool NN = decltype(nn_c)::value;
    // vector instructions per chunk: ~56 with the short quantiser, ~80 with the general one
    constexpr int HEAD = NN ? FQ_PWSMP_HEAD_NN : FQ_PWSMP_HEAD, SLICE = NN ? FQ_PWSMP_SLICE_NN : FQ_PWSMP_SLICE;
    quant_to_panel(0, buf[0], nn_c);
    if (PF < TI) issue(PF, buf[0]);
    FQ_PIN();
    // (two nested loops, both unrolled: one loop with the epilogue behind a test of `it` is too large for hipcc's
    // unroller, and the register arrays indexed by `it` then live in scratch memory)
#pragma unroll
    for (int item = 0; item < NI; ++item) {
    if (item > 0) {                                                     // item boundary: the finished item leaves
      finish_item(item - 1);
      zero_acc();
      FQ_PIN();
    }
#pragma unroll
    for (int ki = 0; ki < KI; ++ki) {
      const int it = item * KI + ki;
      if (!(FQ_PWSMP_ABL & 4) || it == 0)
        __syncthreads();                                                // panel[it & 1] complete (and, first time, the constants)
      if (it == 0) PW_STAMP(6);
      if (it == TI / 2) PW_STAMP(7);
      const unsigned* pb = &panel[it & 1][bq_off];
      v4i bfrag[KS][PT];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) bfrag[ks][pt] = *reinterpret_cast<const v4i*>(pb + ks * (PT * 256) + pt * 128);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
          for (int c = 0; c < CTW; ++c) {
            if (FQ_PWSMP_ABL & 1) acc[pt][c][it & 15] += bfrag[ks][pt][it & 3] ^ ring[it % (AD + 1)][ks][c][it & 3];
            else acc[pt][c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ring[it % (AD + 1)][ks][c], bfrag[ks][pt], acc[pt][c], 0, 0, 0);
          }
      if (it + 1 < TI) {
        quant_to_panel(it + 1, buf[(it + 1) % PF], nn_c);
        if (it + 1 + PF < TI) issue(it + 1 + PF, buf[(it + 1) % PF]);
      }
      if (it + AD < TI) a_chunk(it + AD, ring[(it + AD) % (AD + 1)]);
      // the schedule: B fragments, then one MFMA followed by a slice of vector work, eight times
      __builtin_amdgcn_sched_group_barrier(0x100, KS * PT, 0);          // DS reads
      __builtin_amdgcn_sched_group_barrier(0x002, HEAD, 0);             // vector work behind which the B fragments arrive
#pragma unroll
      for (int i = 0; i < KS * PT * CTW; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, SLICE, 0);          // a slice of the quantiser's vector instructions
      }
      FQ_PIN();
    }
    }
  };
  if (FQ_PWSMP_ABL & 64) __syncthreads();
  else if (fq_nonneg(q)) chunk_loop(std::true_type{});
  else chunk_loop(std::false_type{});
  PW_STAMP(2);
  finish_item(NI - 1);
  PW_STAMP(3);
  if (GAP) {
    __syncthreads();                                                    // the tile is complete
    if (threadIdx.x < NCH) {
      const float* const p = reinterpret_cast<const float*>(smp_dyn) + threadIdx.x * g.HW;
      double acc = 0.0;
      for (int i = 0; i < g.HW; ++i) acc += (double)p[i];
      const float v = (float)acc / (float)g.HW;
      y[(int64_t)smp * g.Cout + ch0 + (int)threadIdx.x] = v;
      m = fabsf(v);
    } else {
      m = 0.0f;
    }
  }
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

// sample form (K2r): stride 1, Cout a multiple of 256, Cin a multiple of 32, and either
//   planes of a multiple of 4 pixels that cut into blocks of 96..128 pixels (14x14: two, 28x28: seven), K / 32 in {4, 8, 16}, or
//   whole planes of 45..64 pixels with 0 or 1 pixel past a multiple of four (7x7, 8x8), K / 32 in {16, 32, 64} (a residual operand: 16).
// grid = samples x channel groups x pixel blocks (rounded to whole rounds over the 8 XCDs).
int pw_try_sample(const PwCall& a, bool* taken) {
  *taken = false;
  // 0 never, 1 blocked planes up to 1024 pixels without a residual operand (rounds 2-4), 2 every shape it takes (round 5)
  static const int mode = env_int("FQ_PWSMP", 2);
  const int kt = (int)(a.cin_pad / 32);
  const bool small = a.hw <= 64;                                        // one block of two pixel tiles
  const int64_t quads = (a.hw + 3) / 4;
  const int nb = small ? 1 : (int)((quads + 31) / 32);                  // fewest blocks of at most 32 pixel groups
  // (K = 320 - MobileNetV2's last 1x1 - only with the pooling epilogue, fq_pwconv_i8_gap)
  const bool plane_ok = small ? (a.hw >= 45 && a.hw % 4 <= 1 && (kt == 16 || ((kt == 32 || kt == 64) && a.residual == nullptr) ||
                                                                 (kt == 10 && a.gap && a.residual == nullptr)))
                              : (a.hw % 4 == 0 && quads / nb >= 24 && (kt == 4 || kt == 8 || kt == 16 || kt == 32));
  const bool shape_ok = plane_ok && a.stride == 1 && a.cin == a.cin_pad && a.cout % 256 == 0 &&
                        a.n < (1 << 20) && a.cin * a.hw * 4 < (1ll << 31) && (small || aligned16(a.x)) &&
                        (small || a.residual == nullptr || aligned16(a.residual));    // (16-byte LDS-DMA of the residual operand)
  // by shape: the small and middle planes (measured in the model against the split form: 512 -> 512 @14x14 32.0 -> 25.2 us,
  // 256 -> 512 @14x14 24.0 -> 20.7, 256 -> 256 @28x28 48.0 -> 39.7, 128 -> 256 @28x28 38.8 -> 33.7); the streaming form keeps
  // the large planes
  // Whole small planes and layers with a residual operand were left to the split form through round 4, on times measured with
  // each kernel ALONE (1024 -> 1024 @7x7 27.9 us against the split form's 26.6, 512 -> 1024 @7x7 19.9 against 20.4; ResNet-50
  // 128 -> 512 @28x28 + residual 103.9 against 99.0, 256 -> 1024 @14x14 + residual 55.1 against 51.0).  Among three batches
  // in flight the sample form wins both - every value is quantised once per 256 or 512 output channels instead of once per
  // 256: ResNet-50 online +1.8 % (sd 0.1; the residual layers), MobileNet default workload +0.4 ... +0.5 % (the 7x7 layers);
  // planes of more than 1024 pixels: no difference (profiles/r5_pwsmp_modes_ab.txt)
  // (tuning: 3 = as 1 + whole small planes, 4 = as 1 + layers with a residual operand, 5 = as 1 + planes of more than 1024 pixels)
  const bool base = a.hw <= 1024 && !small && a.residual == nullptr;
  const bool by_shape = a.form == 0 && (mode == 2 || (mode == 1 && base) ||
                                        (mode == 3 && (base || (small && a.residual == nullptr))) ||
                                        (mode == 4 && a.hw <= 1024 && !small) || (mode == 5 && !small && a.residual == nullptr));
  if (a.gap && !(shape_ok && small)) return FQ_OK;                        // (fq_pwconv_i8_gap: whole small planes only; the caller refuses)
  if (!shape_ok || !(a.form == 7 || by_shape || a.gap)) return FQ_OK;
  // ... and only where its workgroups (one per sample, block and 256 channels) reach a quarter of the CUs: below that the
  // split form's finer items win (MobileNet images/s +3.5 % at batch 8, equal at 16: profiles/r6_small_batch_forms_ab.txt)
  if (a.form != 7 && !a.gap && a.n * (a.cout / 256) * nb < num_cu() / 4) return FQ_OK;
  const int64_t rows_pad = (a.cout + 31) / 32 * 32;
  static const int ctw_tune = env_int("FQ_PWSMP_CTW", 0);               // tuning: 1 = 256 channels per workgroup everywhere
  // two channel tiles per wavefront (512 channels per workgroup, every value quantised once) from K = 512 up; below that
  // two workgroups of 256 channels per CU are faster (256 -> 512 @14x14: 20.1 -> 18.4 us; 512 -> 512: 25.2 against 25.7)
  // ... unless the 512-channel workgroups would be fewer than 3/8 of the CUs (batch 32 and below on the 14x14 and 7x7
  // layers): MobileNet images/s with 256 channels everywhere +2.1 % at batch 16, +0.9 % at 32, -0.35 % at 48, -0.7 % at 64
  const bool few = ctw_tune == 0 && a.residual == nullptr && a.n * (a.cout / 512) * nb < num_cu() * 3 / 8;
  const int ctw = (a.cout % 512 == 0 && ctw_tune != 1 && !few && (kt >= 16 || ctw_tune == 2 || a.residual != nullptr)) ? 2 : 1;
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
  const bool res = a.residual != nullptr;
  // items per workgroup: blocked planes are cut into two items of two pixel tiles each (NI = 2, PT = 2: the first item's
  // stores overlap the second item's loads) or taken as one item of four (NI = 1, PT = 4; round 2)
  // MEASURED (profiles/r3_pw_experiments.txt): two items are SLOWER everywhere - 512 -> 512 @14x14 31.1 us against 24.8,
  // 256 -> 256 @28x28 46.8 against 40.5, 128 -> 256 @28x28 36.7 against 34.3, 256 -> 512 @14x14 21.0 against 18.6 - so one item
  // is what the shape-based choice takes; the two-item instantiations are only built with -DFQ_PWSMP_BUILD_NI2 (tuning).
  static const int ni_tune = env_int("FQ_PWSMP_NI", 0);                 // tuning: 2 = two items (needs FQ_PWSMP_BUILD_NI2)
  const int ni = (!small && ni_tune == 2 && !res) ? 2 : 1;
  const int pt = (small || ni == 2) ? 2 : 4;
  const int64_t grid = (a.n + 7) / 8 * t.CS * nb * 8;
  FQ_REQUIRE(grid < (1ll << 31), "fq_pwconv_i8: too many workgroups for the sample form");
  const int8_t* wfrag = a.wcodes + rows_pad * a.cin_pad;                // second half of fq_weight_codes' buffer
  if (int rc = pw_zero_stat(a)) return rc;
  bool launched = false;
#define FQ_PWSMP_CASE_R(KT_, CTW_, PT_, RES_, NI_)                                                                     \
  if (!a.gap && kt == KT_ && ctw == CTW_ && pt == PT_ && res == RES_ && ni == NI_) {                                   \
    /* (the residual operand staged in LDS: 8 wavefronts x 16 KB beside the 18 KB of panels and constants) */          \
    constexpr size_t dyn_ = (RES_ && PT_ == 4 && CTW_ == 2 && NI_ == 1 && FQ_PWSMP_RESLDS) ? 8 * 16384 : 0;             \
    if (dyn_ != 0) {                                                                                                   \
      static const bool attr_ok =                                                                                      \
          hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_sample_kernel<KT_, CTW_, PT_, RES_, NI_>),         \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_) == hipSuccess;                    \
      FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the sample kernel");                    \
    }                                                                                                                  \
    hipLaunchKernelGGL((pwconv_sample_kernel<KT_, CTW_, PT_, RES_, NI_>), dim3((unsigned)grid), dim3(512), dyn_, a.st,  \
                       a.x, wfrag, a.wscale, (const int*)a.wsum, a.bias, a.y, t, a.in_stat, (int)a.n, a.in_thr,         \
                       a.levels, a.lo_neg, kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out,          \
                       a.residual);                                                                                    \
    launched = true;                                                                                                   \
  }
#define FQ_PWSMP_CASE(KT_, CTW_, PT_, NI_) FQ_PWSMP_CASE_R(KT_, CTW_, PT_, false, NI_)
  // one item of four pixel tiles (round 2)
  FQ_PWSMP_CASE(4, 1, 4, 1) FQ_PWSMP_CASE(4, 2, 4, 1) FQ_PWSMP_CASE(8, 1, 4, 1) FQ_PWSMP_CASE(8, 2, 4, 1)
  FQ_PWSMP_CASE(16, 1, 4, 1) FQ_PWSMP_CASE(16, 2, 4, 1) FQ_PWSMP_CASE(32, 1, 4, 1) FQ_PWSMP_CASE(32, 2, 4, 1)
#ifdef FQ_PWSMP_BUILD_NI2
  // two items of two pixel tiles
  FQ_PWSMP_CASE(4, 1, 2, 2) FQ_PWSMP_CASE(4, 2, 2, 2) FQ_PWSMP_CASE(8, 1, 2, 2) FQ_PWSMP_CASE(8, 2, 2, 2)
  FQ_PWSMP_CASE(16, 1, 2, 2) FQ_PWSMP_CASE(16, 2, 2, 2) FQ_PWSMP_CASE(32, 1, 2, 2) FQ_PWSMP_CASE(32, 2, 2, 2)
#endif
  // whole small planes
  FQ_PWSMP_CASE(16, 1, 2, 1) FQ_PWSMP_CASE(16, 2, 2, 1) FQ_PWSMP_CASE(32, 1, 2, 1) FQ_PWSMP_CASE(32, 2, 2, 1)
  FQ_PWSMP_CASE(64, 1, 2, 1) FQ_PWSMP_CASE(64, 2, 2, 1)                 // (2048 -> 512 @7x7, ResNet-50's last stage)
  // with a residual operand: the last 1x1 convolutions of the ResNet bottlenecks (128 -> 512 @28x28, 256 -> 1024 @14x14)
  FQ_PWSMP_CASE_R(4, 2, 4, true, 1) FQ_PWSMP_CASE_R(8, 2, 4, true, 1) FQ_PWSMP_CASE_R(16, 2, 4, true, 1)
  // ... and 512 -> 2048 @7x7 (a whole small plane)
  FQ_PWSMP_CASE_R(16, 2, 2, true, 1)
#undef FQ_PWSMP_CASE
#undef FQ_PWSMP_CASE_R
  // ... with the global average pooling behind them in the same launch (fq_pwconv_i8_gap): an LDS tile of NCH x HW floats
#define FQ_PWSMP_GAP(KT_, CTW_, RES_)                                                                                  \
  if (a.gap && kt == KT_ && ctw == CTW_ && pt == 2 && res == RES_ && ni == 1) {                                        \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_sample_kernel<KT_, CTW_, 2, RES_, 1, true>),         \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;                     \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8_gap: cannot raise the dynamic LDS limit of the sample kernel");                  \
    hipLaunchKernelGGL((pwconv_sample_kernel<KT_, CTW_, 2, RES_, 1, true>), dim3((unsigned)grid), dim3(512),           \
                       (size_t)(256 * CTW_) * (size_t)a.hw * sizeof(float), a.st, a.x, wfrag, a.wscale, (const int*)a.wsum, \
                       a.bias, a.y, t, a.in_stat, (int)a.n, a.in_thr, a.levels, a.lo_neg, kEps, a.out_current_max,     \
                       a.bn_scale, a.bn_shift, a.act, a.stat_out, a.residual);                                         \
    launched = true;                                                                                                   \
  }
  FQ_PWSMP_GAP(16, 1, false) FQ_PWSMP_GAP(16, 2, false) FQ_PWSMP_GAP(32, 1, false) FQ_PWSMP_GAP(32, 2, false)
  FQ_PWSMP_GAP(64, 1, false) FQ_PWSMP_GAP(64, 2, false) FQ_PWSMP_GAP(16, 2, true)
  FQ_PWSMP_GAP(10, 1, false) FQ_PWSMP_GAP(10, 2, false)                 // (320 -> 1280 @7x7: MobileNetV2)
#undef FQ_PWSMP_GAP
  if (!launched) {                                                     // (a combination that is not built: another form takes it)
    FQ_REQUIRE(a.form != 7, "fq_pwconv_i8: this K / channel-group / residual combination of the sample form is not built");
    return FQ_OK;
  }
  FQ_LAUNCH_CHECK();
  *taken = true;
  return FQ_OK;
}

// shapes fq_pwconv_i8_gap takes: what pw_try_sample's whole-small-plane instantiations take
bool pw_sample_gap_shape_ok(int64_t n, int64_t cin, int64_t cout, int64_t hw, bool residual) {
  const int64_t kt = cin / 32;
  return n > 0 && n < (1 << 20) && cin % 64 == 0 && hw >= 45 && hw <= 64 && hw % 4 <= 1 && cout % 256 == 0 &&
         (kt == 16 || ((kt == 10 || kt == 32 || kt == 64) && !residual)) && (!residual || cout % 512 == 0) &&
         cin * hw * 4 < (1ll << 31);
}

}  // namespace fqi
