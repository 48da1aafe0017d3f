// libfakequant — K2w pointwise (1x1) convolution on int8 codes for 14x14-class planes: weight-stationary, pixel-outer pipeline
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_pw.h"

namespace {

// K2w: pipe form (round 5).  The sample form (K2r) is output-stationary: a workgroup of 512 threads owns ~100 pixels x 512
// output channels, walks the K / 32 channel chunks once and stores everything at the end - so on a chip where every CU holds
// exactly one such workgroup (512 -> 512 @14x14 at batch 128: 256 workgroups) all of them first load, then multiply, then
// store, and the kernel is the SUM of the three (tools/pw_ablate.py, profiles/r3_pw_ablate.txt: 25.1 us = 14.2 compute-only
// + 3.9 for the loads + 6.4 for the stores).  An output can only leave once ALL K channels of its pixel have arrived, so
// loads and stores of one workgroup overlap only when the PIXELS are the outer loop - and that order re-streams the whole
// weight matrix (256 KB) per pixel tile unless the weights stay on the CU.  Here they do, in REGISTERS:
//   * a wavefront keeps the A fragments of its 64 output channels for ALL K-steps (2 x K/32 x 4 registers: 128 at K = 512 -
//     half of the 256 each of the 8 wavefronts of a CU-filling workgroup may hold), loaded once from the fragment-major copy;
//   * the workgroup's block of 96..128 pixels is cut into SUB-TILES of <= SQ groups of four pixels (<= 32 pixels = one MFMA
//     pixel tile); sub-tile s + 1 and s + 2 are on their way while s is multiplied and s - 1's stores drain;
//   * activations arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`: 16 bytes per lane straight into the staging buffer, no
//     registers in flight - the register file is full of weights), two staging buffers of K x SQ x 16 bytes;
//   * all 512 threads quantise the landed sub-tile into the B-fragment panel [K-step][h][pixel][16 codes] (a thread: four
//     channels x four pixels = four 16-byte reads, four dword writes), two panels;
//   * every wavefront reads the K/32 pixel fragments and runs 2 x K/32 MFMAs (v_mfma_i32_32x32x32_i8) with the PIXELS as the
//     matrix rows and its resident weights as the columns: the result comes out with lane = CHANNEL, register = pixel
//     8 (r / 4) + 4 h + r % 4.  A lane then needs the constants of ONE channel per tile - ten registers for the whole kernel
//     instead of a 16-byte LDS read per four values (the first version of this form, lane = pixel like the other forms, spent
//     9.5 of its 36.8 us reading them: profiles/r5_pipe_ablate_v1.txt).  Four consecutive registers are four consecutive pixels
//     of the lane's channel - but stored from there, a store instruction writes 32 channels x 32 bytes, and those scattered
//     pieces cost 12 of 37.9 us (profiles/r5_pipe_ablate_v2.txt: the 4-byte stores of whole 112-byte rows had cost 3.3).  So the
//     finished values take one more trip through LDS: each wavefront writes its 64 channels x <= 32 pixels into its 7 KB of
//     the staging buffer the quantiser has just emptied, reads them back as 16-byte pieces in row order and stores 9 whole
//     channel rows per instruction;
//   * the DMA of sub-tile s + 1 is issued at the top of sub-tile s, behind the barrier that says every wavefront is done with
//     the buffer it lands in (as a staging buffer AND as the previous sub-tile's transposition space).
// Two raw barriers per sub-tile (staged data landed / panel complete); the DMA and the stores are counted by hand
// (`s_waitcnt vmcnt`): hipcc neither sees the DMA instructions (inline asm) nor waits for stores, and `__syncthreads()` would
// drain both (cdna_hip_programming.md, 'Pipelining across barriers').  vmcnt counts loads and stores of a wavefront in issue
// order on gfx9, which is what the counts below rely on.
struct PwPipeGeom {
  int Cin, Cout, CS;         // CS: channel groups of 512
  int n;                     // samples
  int zoff;
  int HW;                    // pixels of a plane (a multiple of 4)
  int nb, qbase, qextra;     // pixel blocks per plane: block k holds qbase + (k < qextra) groups of four pixels (<= 32)
};

#ifndef FQ_PWPIPE_ABL
// tools/pipe_ablate.py (timing only, results are then wrong): 1 MFMAs, 2 quantiser arithmetic, 4 barriers, 8 activation DMA,
// 16 output stores, 32 A-fragment loads, 64 whole quantiser, 128 whole multiplication (B-fragment reads too), 256 whole epilogue
#define FQ_PWPIPE_ABL 0
#endif
#ifndef FQ_PWPIPE_UNROLL
#define FQ_PWPIPE_UNROLL 0
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// all LDS traffic of this wavefront done, then the workgroup barrier - ONE statement, so that nothing is scheduled between
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one LDS-DMA wave-instruction: lane l's 16 bytes at (rsrc base + voff + soff) land at LDS byte address lds_base + 16 l
// (voff out of the resource's range: the lane fetches nothing)
__device__ __forceinline__ void dma16(v4i rsrc, unsigned voff, unsigned soff, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}

// KT = K / 32; SQ: pixel groups per sub-tile (K * SQ * 16 bytes per staging buffer)
template <int KT, int SQ>
__global__ __launch_bounds__(512, 2) void pwconv_pipe_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwPipeGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out) {
  constexpr int NW = 8, CTW = 2;
  constexpr int NCH = NW * CTW * 32;                                    // 512 output channels per workgroup
  constexpr int K = KT * 32;
  constexpr int kStage = K * SQ * 16;                                   // bytes of one staging buffer: [channel][SQ][4 floats]
  constexpr int kPanel = KT * 1024;                                     // bytes of one panel: [K-step][h][32 pixels][16 codes]
  constexpr int ND = (K * SQ + 511) / 512;                              // DMA wave-instructions per wavefront and sub-tile
  constexpr int NU = (K / 4 * SQ + 511) / 512;                          // quantiser units (4 channels x 4 pixels) per thread
  constexpr int NST = SQ;                                               // store instructions per wavefront and sub-tile
  constexpr int kTrans = NW * 64 * SQ * 16;                             // transposition space: per wavefront [64 channels][SQ][4 floats]
  constexpr bool kTransInStage = kTrans <= kStage;                      // K = 512: exactly the staging buffer; K = 256: its own 64 KB
  extern __shared__ __attribute__((aligned(16))) unsigned char pwp_smem[];
  unsigned char* const stage = pwp_smem;                                // 2 x kStage
  unsigned char* const panel = pwp_smem + 2 * kStage;                   // kPanel
  unsigned char* const trans = panel + kPanel;                          // kTrans unless kTransInStage
  // per-channel constants of the workgroup's 512 channels, [sxw | bsc | bsh | bias | zs][512]: a lane reads the five of ITS
  // channel per tile and sub-tile (ten 4-byte reads; kept in registers they pushed four weight fragments into scratch)
  float* const c_tab = reinterpret_cast<float*>(trans + (kTransInStage ? 0 : kTrans));
  float* const red = c_tab + 5 * NCH;

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
  const unsigned nquad = (unsigned)g.qbase + (ph < (unsigned)g.qextra ? 1u : 0u);
  const unsigned pix0 = (ph * (unsigned)g.qbase + (ph < (unsigned)g.qextra ? ph : (unsigned)g.qextra)) * 4u;
  const unsigned plane4 = (unsigned)g.HW * 4u;                          // bytes of a plane
  // sub-tiles: the block's pixel groups split evenly over the fewest sub-tiles of at most SQ
  const unsigned NS = (nquad + (unsigned)SQ - 1u) / (unsigned)SQ;
  const unsigned sbase = nquad / NS, sextra = nquad - sbase * NS;
  auto sub_q0 = [&](unsigned s) { return s * sbase + (s < sextra ? s : sextra); };
  auto sub_nq = [&](unsigned s) { return sbase + (s < sextra ? 1u : 0u); };

  const ThresholdReq treq = threshold_request(in_stat, n, in_thr, b == 0);          // first in the memory queue
  // ---- per-channel constants (one channel per thread: NCH == threads), requested before the first DMA -------------------------
  const int* ibias = lo_neg_max == kRangeMode ? reinterpret_cast<const int*>(bias) : nullptr;
  const float* fbias = lo_neg_max == kRangeMode ? nullptr : bias;
  const int ic = ch0 + (int)threadIdx.x;                                // < Cout (host: Cout % NCH == 0)
  const float k_ws = wscale[ic];
  const int k_wsum = wsum[ic];
  const int k_ib = ibias != nullptr ? ibias[ic] : 0;
  const float k_bias = fbias != nullptr ? fbias[ic] : 0.0f;
  const float k_bsc = has_bn ? bn_scale[ic] : 1.0f;
  const float k_bsh = has_bn ? bn_shift[ic] : 0.0f;
  // ---- activation DMA: slot = (i * 8 + wave) * 64 + lane of the sub-tile's K x SQ pixel groups -> (channel, group) ------------
  v4i xr;
  {
    const unsigned long long xb = (unsigned long long)(reinterpret_cast<const char*>(x) + (int64_t)smp * g.Cin * plane4);
    xr[0] = __builtin_amdgcn_readfirstlane((int)(xb & 0xFFFFFFFFull));
    xr[1] = __builtin_amdgcn_readfirstlane((int)(xb >> 32));
    xr[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)g.Cin * plane4));
    xr[3] = 0x00020000;
  }
  const unsigned lds0 = (unsigned)(size_t)stage;                        // LDS byte address of staging buffer 0
  // (the (channel, group) of a slot is recomputed per sub-tile from an opaque copy of the lane index - ~4 instructions per DMA
  // instruction; kept in 2 x ND registers across the loop they were spilled, and scratch reloads count on vmcnt)
  auto dma_sub = [&](unsigned s) __attribute__((always_inline)) {
    const unsigned nq = sub_nq(s);
    const unsigned soff = __builtin_amdgcn_readfirstlane((pix0 + sub_q0(s) * 4u) * 4u);
    const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (s & 1u) * (unsigned)kStage + (unsigned)wave * 1024u);
    unsigned ln = (unsigned)lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      if (FQ_PWPIPE_ABL & 8) continue;
      const unsigned sl = (unsigned)(i * 8 + wave) * 64u + ln;
      const unsigned c = sl / (unsigned)SQ, j = sl - c * (unsigned)SQ;
      const bool want = j < nq && c < (unsigned)K;                      // (slots past K x SQ: never requested)
      dma16(xr, want ? c * plane4 + j * 16u : 0x80000000u, soff, base + (unsigned)i * 8192u);
    }
  };
  dma_sub(0);
  // ---- weight fragments: wavefront w multiplies channel tiles ct = 2 w, 2 w + 1 of the group, resident for the whole kernel.
  // Requested in the order the first sub-tile's multiplication consumes them (hipcc waits for each where it is first used, in
  // issue order), between the DMA of the first and of the second sub-tile ------------------------------------------------------
  const int ctg0 = (int)cg * NW * CTW + wave * CTW;
  const fq_rsrc wr = make_rsrc(wfrag + (((int64_t)ctg0 * KT) << 10), (int64_t)CTW * KT * 1024);
  v4i W[CTW][KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int c = 0; c < CTW; ++c)
      W[c][kt] = (FQ_PWPIPE_ABL & 32) ? (v4i){c + kt, lane, 3, 4} : buf_ld_v4i(wr, (unsigned)lane * 16u, (unsigned)((c * KT + kt) << 10));
  FQ_PIN();
  if (NS > 1u) dma_sub(1);
  // ---- threshold -> quantiser parameters -> this lane's channel constants -------------------------------------------------------
  const float max_ = threshold_finish(treq, in_stat, n, in_thr, cur_max_out, b == 0);
  int zoff = g.zoff;
  const QParams q = make_qparams_rt(max_, levels, lo_neg_max, eps, in_thr, zoff);
  const float sx = q.scale;
  c_tab[threadIdx.x] = sx * k_ws;
  c_tab[NCH + threadIdx.x] = k_bsc;
  c_tab[2 * NCH + threadIdx.x] = k_bsh;
  c_tab[3 * NCH + threadIdx.x] = k_bias;
  reinterpret_cast<int*>(c_tab)[4 * NCH + threadIdx.x] = zoff * k_wsum + k_ib;
  const int ub = 128 - zoff;
  const unsigned nn_xor = fq_nonneg_xor(ub);
  const bool nonneg = fq_nonneg(q);
  // quantiser units: u = t + 512 k -> (channel quad cq, pixel group j); reads 4 x 16 bytes, writes 4 dwords
  auto quantise = [&](unsigned s, auto nn_c) __attribute__((always_inline)) {
    constexpr bool NN = decltype(nn_c)::value;
    const unsigned char* src = stage + (s & 1u) * (unsigned)kStage;
    unsigned char* dst = panel;
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                                       // (addresses recomputed per sub-tile, not kept)
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      const unsigned u = tid + 512u * (unsigned)k;
      const unsigned cq = u / (unsigned)SQ, j = u - cq * (unsigned)SQ;
      if (cq >= (unsigned)(K / 4)) continue;
      const unsigned u_rd = (cq * 4u * (unsigned)SQ + j) * 16u;         // channel 4 cq, group j
      // panel: channel c = 4 cq -> K-step c / 32, half (c / 16) % 2, byte c % 16; pixel 4 j + e
      const unsigned u_wr = (((cq >> 3) * 2u + ((cq >> 2) & 1u)) * 32u + 4u * j) * 16u + (cq & 3u) * 4u;
      f4 v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = *reinterpret_cast<const f4*>(src + u_rd + (unsigned)r * (SQ * 16u));
      int w[4];
      if (FQ_PWPIPE_ABL & 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = __float_as_int(v[0][e]) ^ (__float_as_int(v[1][e]) >> 8) ^ __float_as_int(v[2][e] + v[3][e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = fq_pack4<NN>(v[0][e], v[1][e], v[2][e], v[3][e], q, ub, nn_xor);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) *reinterpret_cast<int*>(dst + u_wr + (unsigned)e * 16u) = w[e];
      FQ_PIN();                                                         // one unit's 16 values at a time (the registers are the weights')
    }
  };

  // ---- epilogue of one sub-tile: lane = channel 32 c + (l & 31) of the wavefront's 64, registers 4 gq .. 4 gq + 3 = the four
  // pixels of group 2 gq + h of the sub-tile -> this wavefront's transposition rows -> whole rows to memory -------------------
  const int64_t y_bytes = (int64_t)(g.Cout - (ch0 + wave * 64)) * plane4;
  const fq_rsrc yr = make_rsrc(reinterpret_cast<char*>(y) + ((int64_t)smp * g.Cout + ch0 + wave * 64) * plane4, y_bytes);
  float m = 0.0f;
  const bool fast_epi = fbias == nullptr && has_bn && act == FQ_ACT_RELU;
  auto epilogue = [&](unsigned s, const v16i (&acc)[CTW], auto fast_c) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_c)::value;                      // BatchNorm + ReLU, no bias: fixed at compile time
    typedef float f2 __attribute__((ext_vector_type(2)));
    const unsigned nq = sub_nq(s);
    const unsigned pix_s = __builtin_amdgcn_readfirstlane((pix0 + sub_q0(s) * 4u) * 4u);
    unsigned char* const tw = (kTransInStage ? stage + (s & 1u) * (unsigned)kStage : trans) + (unsigned)wave * (64u * SQ * 16u);
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int cl = wave * 64 + c * 32 + pl;                           // this lane's channel of the workgroup's 512
      const float k_sxw = c_tab[cl], k_bsc = c_tab[NCH + cl], k_bsh = c_tab[2 * NCH + cl], k_bias = c_tab[3 * NCH + cl];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const unsigned qd = (unsigned)(2 * gq + h);
        const bool ok = qd < nq;
        f4 out;
        // pairs of pixels share the channel's constants: scale / BatchNorm run as packed fp32 instructions (two IEEE operations
        // each: the same values as the scalar form); the integer sum already holds zoff * rowsum + bias code (the accumulators start
        // from it)
#pragma unroll
        for (int pp = 0; pp < 4; pp += 2) {
          f2 v = (f2){(float)acc[c][4 * gq + pp], (float)acc[c][4 * gq + pp + 1]};
          v = v * (f2){k_sxw, k_sxw};
          if (FAST) {
            v = v * (f2){k_bsc, k_bsc};
            v = v + (f2){k_bsh, k_bsh};
            v.x = fmaxf(v.x, 0.0f);
            v.y = fmaxf(v.y, 0.0f);
          } else {
            if (fbias != nullptr) v = v + (f2){k_bias, k_bias};
            if (has_bn) {
              v = v * (f2){k_bsc, k_bsc};
              v = v + (f2){k_bsh, k_bsh};
            }
            v.x = act_rt(v.x, act);
            v.y = act_rt(v.y, act);
          }
          out[pp] = v.x;
          out[pp + 1] = v.y;
        }
        const float gm = fmaxf(fmaxf(fabsf(out.x), fabsf(out.y)), fmaxf(fabsf(out.z), fabsf(out.w)));
        m = fmaxf(m, ok ? gm : 0.0f);
        // (group SQ of a 7-group row does not exist; groups past the sub-tile's own are written and never stored)
        if (qd < (unsigned)SQ) *reinterpret_cast<f4*>(tw + ((unsigned)(c * 32 + pl) * SQ + qd) * 16u) = out;
      }
    }
    if (FQ_PWPIPE_ABL & 16) return;
    // rows back out, 16 bytes per lane in row order: piece t = 64 i + lane -> channel t / SQ, group t % SQ
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // (this wavefront's own rows: no barrier)
    unsigned ln = (unsigned)lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int i = 0; i < SQ; ++i) {
      const unsigned t = (unsigned)i * 64u + ln;
      const unsigned row = t / (unsigned)SQ, j = t - row * (unsigned)SQ;
      const f4 v = *reinterpret_cast<const f4*>(tw + t * 16u);
      // (the guarded form: fq_common.h, the 16-byte buffer store's data hazard)
      buf_st_v4f(yr, j < nq ? row * plane4 + j * 16u : 0x80000000u, pix_s, v);
      if ((i & 1) == 1) FQ_PIN();                                       // two pieces in flight
    }
  };

  // one sub-tile.  FIRST (s = 0) is a copy of its own in front of the loop: only there are loads that hipcc knows about still
  // on their way (the weight fragments), and its waits for them - the last one a vmcnt(0) - inside a rolled loop would be
  // executed by EVERY iteration and drain the DMA and the stores each time.
  auto sub_tile = [&](unsigned s, auto nn_c, auto first_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value;
    constexpr int NWF = CTW * KT;
    // (1) this wavefront's DMA of sub-tile s has landed.  Issued after it: s = 0 - the weight fragments and the DMA of sub-tile
    // 1 (when there is one); s >= 1 - the stores of s - 1
    if (FIRST) {
      if (NS > 1u) wait_vm<(NWF + ND < 63 ? NWF + ND : 63)>();
      else wait_vm<(NWF < 63 ? NWF : 63)>();
    } else {
      wait_vm<NST>();
    }
    // ... and everybody else's; every wavefront is also done with sub-tile s - 1: its panel reads and its transposition rows
    if (!(FQ_PWPIPE_ABL & 4) || FIRST) lds_barrier();
    // (1b) the DMA of s + 1 into the other staging buffer (the second sub-tile's was issued in the set-up)
    if (!FIRST && s + 1u < NS) dma_sub(s + 1u);
    // (2) quantise into the panel
    if (!(FQ_PWPIPE_ABL & 64)) quantise(s, nn_c);
    if (!(FQ_PWPIPE_ABL & 4)) lds_barrier();                            // (3) panel complete, staging buffer s & 1 read
    // (5) multiply: rows = pixels (the panel's fragments), columns = this wavefront's channels; the sums start from
    // zoff * rowsum + bias code of the lane's channel
    v16i acc[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int zs = reinterpret_cast<const int*>(c_tab)[4 * NCH + wave * 64 + c * 32 + pl];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = zs;
    }
    const unsigned char* pb = panel + ((unsigned)h * 32u + (unsigned)pl) * 16u;
    if (!(FQ_PWPIPE_ABL & 128)) {
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) {
        const v4i pf = *reinterpret_cast<const v4i*>(pb + ks * 1024);
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          if (FQ_PWPIPE_ABL & 1) acc[c][ks & 15] += pf[ks & 3] ^ W[c][ks][ks & 3];
          else acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(pf, W[c][ks], acc[c], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) acc[c][ks & 15] += W[c][ks][ks & 3];
    }
    if (FIRST) {
      // the fragments stay where they are for the rest of the kernel: opaque from here on (hipcc would otherwise be free to
      // re-load them - they are loop-invariant loads of read-only memory - instead of keeping 128 registers)
#pragma unroll
      for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) asm volatile("" : "+v"(W[c][kt]));
    }
    // (6) scale, BatchNorm, activation, statistic, stores
    if (FQ_PWPIPE_ABL & 256) {
#pragma unroll
      for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) m = fmaxf(m, __int_as_float(acc[c][i] & 0x3FFFFFFF));
    } else if (fast_epi) epilogue(s, acc, std::true_type{});
    else epilogue(s, acc, std::false_type{});
  };
  auto run = [&](auto nn_c) __attribute__((always_inline)) {
    sub_tile(0u, nn_c, std::true_type{});
    // a ROLLED loop over the other sub-tiles (one copy of the ~1000 instructions, re-run from the instruction cache by every
    // sub-tile and wavefront; -DFQ_PWPIPE_UNROLL=1: MAXS - 1 copies)
#if FQ_PWPIPE_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
    for (unsigned s = 1; s < NS; ++s) sub_tile(s, nn_c, std::false_type{});
  };
  if (nonneg) run(std::true_type{});
  else run(std::false_type{});

  if (has_stat) {                                                       // the whole workgroup is one sample
    const float wm = wave_max_nonneg(m);
    if (lane == 0) red[wave] = wm;
    lds_barrier();
    if (threadIdx.x == 0) {
      float t = red[0];
#pragma unroll
      for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
      if (__float_as_uint(t) != 0u) atomic_max_f32(stat_out + smp, t);
    }
  }
}

}  // namespace

namespace fqi {

// pipe form (K2w): stride 1, no residual operand, fp32 in and out, Cout a multiple of 512, Cin = 256 or 512 (unpadded), planes
// of a multiple of 4 pixels (up to 1024) cut into the fewest blocks of at most 32 pixel groups.
// grid = samples x channel groups x pixel blocks (rounded to whole rounds over the 8 XCDs), one workgroup per CU.
int pw_try_pipe(const PwCall& a, bool* taken) {
  *taken = false;
  static const int mode = env_int("FQ_PWPIPE", 0);                      // tuning: 0 only when named (form 9), 1 by shape
  const int kt = (int)(a.cin_pad / 32);
  const int64_t quads = a.hw / 4;
  const bool shape_ok = a.stride == 1 && a.residual == nullptr && !a.in_c16 && a.out_thr == nullptr && a.y16 == nullptr &&
                        a.eval_labels == nullptr && a.cin == a.cin_pad && (kt == 8 || kt == 16) && a.cout % 512 == 0 &&
                        a.hw % 4 == 0 && a.hw >= 16 && a.hw <= 1024 && a.n < (1 << 20) && a.cin * a.hw * 4 < (1ll << 31) &&
                        aligned16(a.x);
  // by shape: the 14x14 planes of the deep MobileNet layers (see the header comment; the 28x28 planes stay with the sample form)
  const bool by_shape = a.form == 0 && mode == 1 && a.hw <= 256;
  if (!shape_ok || !(a.form == 9 || by_shape)) return FQ_OK;
  const int nb = (int)((quads + 31) / 32);
  PwPipeGeom t;
  t.Cin = (int)a.cin;
  t.Cout = (int)a.cout;
  t.CS = (int)(a.cout / 512);
  t.n = (int)a.n;
  t.zoff = a.zoff;
  t.HW = (int)a.hw;
  t.nb = nb;
  t.qbase = (int)(quads / nb);
  t.qextra = (int)(quads % nb);
  const int64_t grid = (a.n + 7) / 8 * t.CS * nb * 8;
  FQ_REQUIRE(grid < (1ll << 31), "fq_pwconv_i8: too many workgroups for the pipe form");
  const int8_t* wfrag = a.wcodes + ((a.cout + 31) / 32 * 32) * a.cin_pad;   // second half of fq_weight_codes' buffer
  if (int rc = pw_zero_stat(a)) return rc;
  bool launched = false;
#define FQ_PWPIPE_CASE(KT_, SQ_)                                                                                        \
  if (kt == KT_) {                                                                                                      \
    constexpr size_t stage_b = (size_t)(KT_ * 32 * SQ_ * 16), trans_b = (size_t)(8 * 64 * SQ_ * 16);                   \
    constexpr size_t lds = 2 * stage_b + (size_t)(KT_ * 1024) + (trans_b <= stage_b ? 0 : trans_b) + 512 * 5 * 4 + 64;             \
    static const bool attr_ok =                                                                                         \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_pipe_kernel<KT_, SQ_>),                               \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;                        \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the pipe kernel");                         \
    hipLaunchKernelGGL((pwconv_pipe_kernel<KT_, SQ_>), dim3((unsigned)grid), dim3(512), lds, a.st, a.x, wfrag,          \
                       a.wscale, (const int*)a.wsum, a.bias, a.y, t, a.in_stat, (int)a.n, a.in_thr, a.levels, a.lo_neg, \
                       kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out);                             \
    launched = true;                                                                                                    \
  }
  FQ_PWPIPE_CASE(16, 7) FQ_PWPIPE_CASE(8, 8)
#undef FQ_PWPIPE_CASE
  FQ_REQUIRE(launched, "fq_pwconv_i8: no instantiation of the pipe form for K/32=%d", kt);
  FQ_LAUNCH_CHECK();
  *taken = true;
  return FQ_OK;
}

}  // namespace fqi
