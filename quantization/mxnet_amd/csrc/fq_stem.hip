// libfakequant — K2s first convolution 3x3 stride 2 (3 -> 32)
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// K2s: the first ("stem") convolution of the ImageNet nets: dense 3x3, stride 2, pad 1, 3 input channels -> COUT,
// fp32 (the reference excludes the first convolution from quantisation), with BatchNorm/activation folded into the store
// and the per-sample max|y| the next (quantised) layer needs.  A lane owns one output pixel: it gathers its 27 inputs
// (zero padding by clamped address + select), then for every tap multiplies by the COUT weights of that tap, read from
// an LDS copy of the tap-major weights with broadcast ds_read_b128 (4 weights per read).  (Feeding the weights through
// SGPRs looked cheaper but hipcc hoists all 864 scalar loads and spills them to VGPR lanes: a v_readlane per FMA.)
// Stores are contiguous along the lanes for every channel.  MIOpen needed 0.14 ms + a separate 0.08 ms BatchNorm/ReLU/statistic
// pass for this layer at batch 128; the layer moves 77 MB in + 205 MB out.
// ---------------------------------------------------------------------------------------------------------------
template <int CIN, int COUT>
__global__ __launch_bounds__(kBlock) void stem_conv3x3s2_kernel(
    const float* __restrict__ x, const float* __restrict__ wt /*[CIN][3][3][COUT]*/, const float* __restrict__ bias,
    float* __restrict__ y, int H, int W, int Ho, int Wo, int tiles_per_wg, const float* __restrict__ bn_scale,
    const float* __restrict__ bn_shift, int act, float* __restrict__ stat_out) {
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) float wl[CIN * 9 * COUT];
  for (int i = threadIdx.x; i < CIN * 9 * COUT; i += kBlock) wl[i] = wt[i];
  __syncthreads();
  const int smp = blockIdx.y;
  const int HWo = Ho * Wo;
  const float* xs = x + (int64_t)smp * CIN * H * W;
  float* ys = y + (int64_t)smp * COUT * HWo;
  const bool has_bn = bn_scale != nullptr;
  float m = 0.0f;
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int pix = (blockIdx.x * tiles_per_wg + t) * kBlock + threadIdx.x;
    if ((blockIdx.x * tiles_per_wg + t) * kBlock >= HWo) break;          // uniform
    const bool valid = pix < HWo;
    const int pc = valid ? pix : HWo - 1;
    const int oy = pc / Wo, ox = pc - oy * Wo;
    float in[CIN][3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      const bool yin = iy >= 0 && iy < H;
      const int iyc = iy < 0 ? 0 : (iy < H ? iy : H - 1);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        const bool inb = yin && ix >= 0 && ix < W;
        const int ixc = ix < 0 ? 0 : (ix < W ? ix : W - 1);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const float v = xs[((int64_t)ci * H + iyc) * W + ixc];
          in[ci][ky][kx] = inb ? v : 0.0f;
        }
      }
    }
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float v = in[ci][ky][kx];
          const f4* wtap = reinterpret_cast<const f4*>(wl + ((ci * 3 + ky) * 3 + kx) * COUT);
          FQ_PIN();                              // one tap's weights at a time (else all 216 reads are hoisted: spills)
#pragma unroll
          for (int c4 = 0; c4 < COUT / 4; ++c4) {
            const f4 wv = wtap[c4];
            acc[4 * c4 + 0] = __builtin_fmaf(wv.x, v, acc[4 * c4 + 0]);
            acc[4 * c4 + 1] = __builtin_fmaf(wv.y, v, acc[4 * c4 + 1]);
            acc[4 * c4 + 2] = __builtin_fmaf(wv.z, v, acc[4 * c4 + 2]);
            acc[4 * c4 + 3] = __builtin_fmaf(wv.w, v, acc[4 * c4 + 3]);
          }
          // ... and the accumulators pinned per tap: otherwise the optimiser sinks every channel's 27 FMAs down to that
          // channel's store and keeps all 864 weights live instead
#pragma unroll
          for (int c8 = 0; c8 < COUT / 8; ++c8)
            asm volatile("" : "+v"(acc[8 * c8]), "+v"(acc[8 * c8 + 1]), "+v"(acc[8 * c8 + 2]), "+v"(acc[8 * c8 + 3]),
                              "+v"(acc[8 * c8 + 4]), "+v"(acc[8 * c8 + 5]), "+v"(acc[8 * c8 + 6]), "+v"(acc[8 * c8 + 7]));
        }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float v = acc[co];
      if (bias != nullptr) v = v + bias[co];
      if (has_bn) {
        v = v * bn_scale[co];
        v = v + bn_shift[co];
      }
      v = act_rt(v, act);
      if (valid) {
        ys[(int64_t)co * HWo + pix] = v;
        m = fmaxf(m, fabsf(v));
      }
    }
  }
  if (stat_out != nullptr) {
    m = block_max(m, red);
    if (threadIdx.x == 0) atomic_max_f32(stat_out + smp, m);
  }
}


// ---------------------------------------------------------------------------------------------------------------
// K2q: the same first convolution (K x K, stride 2, padding K/2, 3 input channels) on the fp32 MATRIX cores, for 3x3 -> 32
// (MobileNets) and 7x7 -> 64 (ResNets).  v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate, so this buys no FLOPs - it buys
// instruction slots: one MFMA replaces 64 v_fma per lane-pair and the VALU is left with addresses and the epilogue (the VALU
// form above issues 864 FMAs + 216 LDS reads per pixel-wave and reaches ~35 % of the vector peak).  It is BIT-IDENTICAL to the
// VALU form and its oracle: the instruction accumulates as an fmaf chain in ascending k (tools/mfma_f32_probe.hip: 1024 of
// 1024 outputs bit-equal), and k runs over (ci, ky, kx) exactly as the chain above; padded taps multiply a zero.
// GEMM view: D[co][pixel] += W[co][k] * X[k][pixel].  A tile is 32 consecutive output pixels; lane l supplies, per step s,
// W[co = l % 32][k = 2 s + l / 32] (3x3: 14 registers per wavefront for good; 7x7: LDS) and X[k][pixel l % 32]: ONE 4-byte
// buffer load per lane and step, gathered straight from NCHW (each input pixel is used by ~K^2 / 4 outputs: L1 / L2 serve the
// repeats), taps outside the image masked by an out-of-range offset (the load returns 0).  D comes out lane = pixel, register =
// channel - the layout of the pointwise kernels: BatchNorm / activation / statistic on store, 128-byte lines per channel.
// ---------------------------------------------------------------------------------------------------------------
typedef float v16f __attribute__((ext_vector_type(16)));
#ifndef FQ_STEM_CH
#define FQ_STEM_CH 16
#endif
#ifndef FQ_STEM_NTS
// nontemporal stores of the first convolution's fp32 output (205 MB at batch 128, read once by the first depthwise layer): +1.5 %
// images/s with three batches in flight in two alternating A/Bs of 5-6 rounds (profiles/r5_nt_sweep4.txt, r5_nt_sweep5.txt; 0 = off)
#define FQ_STEM_NTS 1
#endif
#ifndef FQ_STEM_NTL
#define FQ_STEM_NTL 0        // A/B builds: 1 = the gather of the input image with the nontemporal hint
#endif
#ifndef FQ_STEM_INTERLEAVE
// The four wavefronts of a workgroup take the workgroup's tiles in turn (tile T0 + wave, + 4, ...) instead of a quarter of the
// range each: they then walk the same output rows at the same time and their input rows meet in the L1 / L2 - a wavefront on
// its own 14 output rows keeps 78 KB of input alive, the 16-32 wavefronts of a CU together far more than the caches hold
// (PMC: 106.6 MB fetched for the 77 MB input).  0: the round-4 assignment (A/B builds).
#define FQ_STEM_INTERLEAVE 1
#endif
#ifndef FQ_STEM_NOSTORE      // tuning only (tools/stembench.py): the statistic without the stores - what a recomputation would cost
#define FQ_STEM_NOSTORE 0
#endif

// Tile bookkeeping is 32-bit and incremental (host: fewer than 2^31 tiles; the wavefront's range, the divisions by the
// tiles per image and by the output width arrive as per / rem and multiplicative inverses).  With 64-bit tile indices hipcc
// expanded two 64-bit divisions per tile on the SCALAR unit: ~640 of the ~1000 instructions of a tile.
struct StemGeom {
  int H, W, Ho, Wo;
  unsigned tiles_per_img, total_tiles, n_samples;
  unsigned per, rem;        // wavefront w works on tiles [w * per + min(w, rem), ...) - per + (w < rem) of them
  FastDiv by_tpi, by_wo;
};

// OUT16 (round 4; fq_stem_conv3x3s2_c16): y is a C16 code tensor holding the CONSUMER's codes of the values just formed under
// its stored threshold (offline input quantisation): MobileNetV2's first 1x1 then reads 1 byte per element instead of 4 and
// this kernel writes 1 instead of 4 (the layer moves 77 + 205 MB otherwise).  The statistic is that of the fp32 values.
struct StemCodes {
  const float* thr;
  float levels;
  int lo_neg, zoff, CBo;
};

template <int KS, int COUT, int EPI, bool OUT16 = false>
__global__ __launch_bounds__(kBlock) void stem_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ wt /*[3][KS][KS][COUT]*/, const float* __restrict__ bias,
    float* __restrict__ y, StemGeom g, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out, StemCodes oc) {
  const int H = g.H, W = g.W, Ho = g.Ho, Wo = g.Wo;
  constexpr int K = 3 * KS * KS, NS = (K + 1) / 2, CT = COUT / 32, PAD = KS / 2;
  constexpr bool WREG = NS * CT <= 16;                                  // weights in registers (3x3 -> 32), else LDS
  constexpr int CH = NS < 16 ? NS : FQ_STEM_CH;                         // steps whose loads are in flight together
  constexpr int kSlots = 8;
  __shared__ __attribute__((aligned(16))) float wl[NS * 2 * COUT];      // [k][co], zero row for the padded k
  __shared__ __attribute__((aligned(16))) float c_bias[COUT], c_bsc[COUT], c_bsh[COUT];   // per-channel epilogue constants
  __shared__ unsigned k_stat[kSlots];
  for (int i = threadIdx.x; i < NS * 2 * COUT; i += kBlock) wl[i] = i < K * COUT ? wt[i] : 0.0f;
  for (int i = threadIdx.x; i < COUT; i += kBlock) {
    c_bias[i] = bias != nullptr ? bias[i] : 0.0f;
    c_bsc[i] = bn_scale != nullptr ? bn_scale[i] : 1.0f;
    c_bsh[i] = bn_scale != nullptr ? bn_shift[i] : 0.0f;
  }
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;
  PW_STAMP(0);
  __syncthreads();
  PW_STAMP(1);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pl = lane & 31;
  const int HWo = Ho * Wo;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const unsigned wid = blockIdx.x * 4u + (unsigned)wave, wid0 = blockIdx.x * 4u;
  // the workgroup's tiles [wg_begin, wg_end) = those of its four wavefronts; a wavefront takes every TSTEP-th of them
  constexpr unsigned TSTEP = FQ_STEM_INTERLEAVE ? 4u : 1u;
  const unsigned wg_begin = wid0 * g.per + (wid0 < g.rem ? wid0 : g.rem);
  const unsigned wid1 = wid0 + 4u, wg_end = wid1 * g.per + (wid1 < g.rem ? wid1 : g.rem);
  const unsigned t_begin = FQ_STEM_INTERLEAVE ? wg_begin + (unsigned)wave : wid * g.per + (wid < g.rem ? wid : g.rem);
  const unsigned t_end = FQ_STEM_INTERLEAVE ? wg_end : t_begin + g.per + (wid < g.rem ? 1u : 0u);
  const unsigned s_base = fast_div(wid0 * g.per + (wid0 < g.rem ? wid0 : g.rem), g.by_tpi);
  float areg[WREG ? NS * CT : 1];
  if (WREG) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int s = 0; s < NS; ++s) areg[c * NS + s] = wl[(2 * s + h) * COUT + c * 32 + pl];
  }
  const unsigned x_img = (unsigned)(3 * H * W) * 4u, W4 = (unsigned)W * 4u, HW4 = (unsigned)(H * W) * 4u;

  struct Pix { unsigned smp, tin, jp; int pixoff; unsigned ym, xm; };
  // (smp, tin) = image and tile inside the image; past the last tile the last one is repeated (its loads are never used)
  auto pix_at = [&](unsigned smp, unsigned tin) __attribute__((always_inline)) {
    Pix r;
    r.smp = smp;
    r.tin = tin;
    unsigned jp = tin * 32u + (unsigned)pl;
    jp = jp < (unsigned)HWo ? jp : (unsigned)HWo - 1;                   // lanes past the end copy the last pixel
    r.jp = jp;
    const int oy = (int)fast_div(jp, g.by_wo), ox = (int)(jp - (unsigned)oy * (unsigned)Wo);
    const int iy0 = 2 * oy - PAD, ix0 = 2 * ox - PAD;
    r.pixoff = (iy0 * W + ix0) * 4;
    unsigned ym = 0, xm = 0;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      ym |= (iy0 + k >= 0 && iy0 + k < H) ? (1u << k) : 0u;
      xm |= (ix0 + k >= 0 && ix0 + k < W) ? (1u << k) : 0u;
    }
    r.ym = ym;
    r.xm = xm;
    return r;
  };
  auto pix_next = [&](const Pix& p, bool more) __attribute__((always_inline)) {
    unsigned smp = p.smp, tin = p.tin;
    if (more) {
      tin += TSTEP;
      while (tin >= g.tiles_per_img) {
        tin -= g.tiles_per_img;
        ++smp;
      }
    }
    return pix_at(smp, tin);
  };
  // the lane's input value of step s: tap k = 2 s + h, i.e. (ci, ky, kx); an invalid tap (outside the image, or the padded
  // k) gets an offset the resource bounds out
  auto issue = [&](const fq_rsrc& xr, const Pix& px, int s) __attribute__((always_inline)) {
    const int k0 = 2 * s, k1 = 2 * s + 1;
    const int ky0 = (k0 / KS) % KS, kx0 = k0 % KS, ci0 = k0 / (KS * KS);
    const int ky1 = k1 < K ? (k1 / KS) % KS : 0, kx1 = k1 < K ? k1 % KS : 0, ci1 = k1 < K ? k1 / (KS * KS) : 0;
    const unsigned t0 = (unsigned)ci0 * HW4 + (unsigned)ky0 * W4 + (unsigned)kx0 * 4u;
    const unsigned t1 = (unsigned)ci1 * HW4 + (unsigned)ky1 * W4 + (unsigned)kx1 * 4u;
    const bool v0 = ((px.ym >> ky0) & (px.xm >> kx0) & 1u) != 0u;
    const bool v1 = k1 < K && ((px.ym >> ky1) & (px.xm >> kx1) & 1u) != 0u;
    const bool v = h ? v1 : v0;
    const unsigned off = (unsigned)px.pixoff + (h ? t1 : t0);
    return FQ_STEM_NTL ? buf_ld_f32_nt(xr, v ? off : 0x80000000u, 0u) : buf_ld_f32(xr, v ? off : 0x80000000u, 0u);
  };
  auto rsrc_of = [&](const Pix& px) __attribute__((always_inline)) {
    return make_rsrc(reinterpret_cast<const char*>(x) + (int64_t)px.smp * x_img, x_img);
  };

  float bbuf[2][CH];
  Pix cur;
  {
    const unsigned tc = t_begin < g.total_tiles ? t_begin : g.total_tiles - 1;
    const unsigned smp0 = fast_div(tc, g.by_tpi);
    cur = pix_at(smp0, tc - smp0 * g.tiles_per_img);
  }
  if (t_begin < t_end) {
    const fq_rsrc xr = rsrc_of(cur);
#pragma unroll
    for (int i = 0; i < CH; ++i) bbuf[0][i] = issue(xr, cur, i);
  }
  for (unsigned t = t_begin; t < t_end; t += TSTEP) {
    if (t == t_begin + TSTEP) PW_STAMP(2);
    const Pix nxt = pix_next(cur, t + TSTEP < g.total_tiles);
    const fq_rsrc xr = rsrc_of(cur), xn = rsrc_of(nxt);
    v16f acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.0f;
    // chunks of CH steps: the loads of chunk c + 1 (or of the next tile's first chunk) are issued before chunk c's MFMAs
    constexpr int NCHUNK = (NS + CH - 1) / CH;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      float (&mine)[CH] = bbuf[c & 1];
      float (&other)[CH] = bbuf[(c + 1) & 1];
      if (c + 1 < NCHUNK) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
          if ((c + 1) * CH + i < NS) other[i] = issue(xr, cur, (c + 1) * CH + i);
      } else {
        static_assert(NCHUNK % 2 == 1 || true, "");
#pragma unroll
        for (int i = 0; i < CH; ++i) other[i] = issue(xn, nxt, i);
      }
      FQ_PIN();
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int s = c * CH + i;
        if (s < NS) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const float a = WREG ? areg[ct * NS + s] : wl[(2 * s + h) * COUT + ct * 32 + pl];
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, mine[i], acc[ct], 0, 0, 0);
          }
          if (!WREG && (i & 3) == 3) FQ_PIN();   // LDS weights: four steps' reads at a time (else all are hoisted: registers)
        }
      }
      FQ_PIN();
    }
    // the next tile's first chunk must sit in bbuf[0]: NCHUNK odd leaves it in bbuf[1]
    if (NCHUNK % 2 == 1) {
#pragma unroll
      for (int i = 0; i < CH; ++i) bbuf[0][i] = bbuf[1][i];
    }
    // ---- epilogue: lane = pixel, register = channel 8 gq + 4 h + r; constants four at a time from LDS, buffer stores --
    float m = 0.0f;
    const unsigned HWo4 = (unsigned)HWo * 4u;
    const fq_rsrc yr = OUT16 ? make_rsrc(reinterpret_cast<char*>(y) + (int64_t)cur.smp * oc.CBo * HWo * 16, (int64_t)oc.CBo * HWo * 16)
                             : make_rsrc(reinterpret_cast<char*>(y) + (int64_t)cur.smp * COUT * HWo4, (int64_t)COUT * HWo4);
    const unsigned yo = (unsigned)(4 * h) * HWo4 + cur.jp * 4u;
    QParams q2;
    q2.lo = q2.hi = q2.denom = q2.scale = 0.0f;
    q2.rden = 0.0;
    if (OUT16) q2 = make_qparams(oc.thr[0], oc.levels, oc.lo_neg != 0, kEps);
    const int ubias2 = 128 - oc.zoff;
    // (a code output behind a compile-time ReLU / ReLU6: activation and the consumer's clip as ONE median, the statistic from
    // the raw values - fq_pw_split_kernel.h)
    constexpr bool FOLD = OUT16 && EPI != kEpiRuntime;
    QParams qc = q2;
    if (FOLD) {
      qc.lo = 0.0f;
      if (EPI == kEpiBnRelu6) qc.hi = fminf(q2.hi, 6.0f);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int c0 = ct * 32 + 8 * gq + 4 * h;
        const f4 bch = *reinterpret_cast<const f4*>(c_bias + c0);
        const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
        const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
        float vq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = dw_finish<EPI, !FOLD>(acc[ct][4 * gq + r], bias != nullptr, bch[r], has_bn, bsc[r], bsh[r], act);
          if (!OUT16 && !FQ_STEM_NOSTORE) {
            if (FQ_STEM_NTS) buf_st_f32_nt(yr, yo, (unsigned)(ct * 32 + 8 * gq + r) * HWo4, v);
            else buf_st_f32(yr, yo, (unsigned)(ct * 32 + 8 * gq + r) * HWo4, v);
          }
          vq[r] = v;
          m = FOLD ? fmaxf(m, v) : fmaxf(m, fabsf(v));
        }
        if (OUT16) {     // channels 8 gq + 4 h .. + 3 of the lane's pixel = bytes 8 (gq & 1) + 4 h .. of block 2 ct + gq / 2
          // (a clip range that starts at 0 - unsigned activations - takes the five-instruction quantiser of fq_common.h: a
          // scalar branch; the empty asm statements keep it one - without them both forms are computed and selected)
          int packed;
          if (fq_nonneg(qc)) {
            asm volatile("");
            packed = fq_pack4<true>(vq[0], vq[1], vq[2], vq[3], qc, ubias2, fq_nonneg_xor(ubias2));
          } else {
            asm volatile("");
            packed = fq_pack4<false>(vq[0], vq[1], vq[2], vq[3], qc, ubias2, 0u);
          }
          buf_st_f32(yr, cur.jp * 16u + (unsigned)(8 * (gq & 1) + 4 * h), (unsigned)((2 * ct + (gq >> 1)) * HWo * 16),
                     __int_as_float(packed));
        }
      }
    if (FOLD && EPI == kEpiBnRelu6) m = fminf(m, 6.0f);
    if (has_stat) {                               // a tile lies within one sample
      const float wm = wave_max_nonneg(m);
      if (lane == 0) {
        const unsigned slot = cur.smp - s_base;
        if (slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
        else atomic_max_f32(stat_out + cur.smp, wm);
      }
    }
    cur = nxt;
  }
  PW_STAMP(4);
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < g.n_samples)
      FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
  PW_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------------------
// K2r (round 6, last hours): the 3x3 -> 32 first convolution with its INPUT staged in LDS.  K2q gathers its B operand from global
// memory - 14 four-byte loads per lane and tile with an 8-byte lane stride - and without its stores it still takes 53 us for an
// 18 us matrix chain (tools/stembench.py, -DFQ_STEM_NOSTORE): the texture path bounds it, as it bounded the 7x7 head until
// its input rows were staged (fq_stem_pool.hip, profiles/r6_stem_pool_lds_ab.txt).  Here a workgroup of eight wavefronts walks
// down a band of output rows of ONE image, four rows (Wo / 8 tiles of 32 consecutive pixels: whole 128-byte lines per channel,
// as before) per step; the nine input rows 8 q - 1 .. 8 q + 7 of step q live in LDS ([slot = (iy + 1) mod 17][ci][4 zeros | W |
// 4 zeros]: the padding is data), the eight new rows of step q + 1 - requested a step earlier, two steps of rows are in flight: with
// one the step waited for memory, the first convolution of a step 70 us against 60 - are written into the other eight slots behind
// the tiles of step q - ONE barrier per step.  Same k order (ci, ky, kx), same fmaf chain: bit-identical to K2q.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kR3Slots = 17, kR3Rows = 4, kR3ST = 3;                    // 16-byte loads per thread for eight rows: 6 W <= 512 kR3ST

struct Stem3Geom {
  int H, W, Ho, Wo;
  int nbands, steps_per_band, total_steps;
  FastDiv by_wo;
};

template <int EPI>
__global__ __launch_bounds__(512, EPI == kEpiRuntime ? 2 : 4) void stem3_rows_kernel(const float* __restrict__ x, const float* __restrict__ wt /*[3][3][3][32]*/,
                                                            const float* __restrict__ bias, float* __restrict__ y, Stem3Geom g,
                                                            const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                            int act, float* __restrict__ stat_out) {
  constexpr int K = 27, NS = 14, COUT = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char s3_smem[];
  float* const c_bias = reinterpret_cast<float*>(s3_smem);
  float* const c_bsc = c_bias + COUT;
  float* const c_bsh = c_bsc + COUT;
  float* const red = c_bsh + COUT;                                      // 8 (+ 24 unused: keeps xin 16-byte aligned)
  float* const xin = red + 32;                                          // [17][3][XW]
  const int H = g.H, W = g.W, Ho = g.Ho, Wo = g.Wo;
  const int XW = W + 8;
  for (int i = threadIdx.x; i < COUT; i += 512) {
    c_bias[i] = bias != nullptr ? bias[i] : 0.0f;
    c_bsc[i] = bn_scale != nullptr ? bn_scale[i] : 1.0f;
    c_bsh[i] = bn_scale != nullptr ? bn_shift[i] : 0.0f;
  }
  for (int i = threadIdx.x; i < kR3Slots * 3 * XW; i += 512) xin[i] = 0.0f;   // (the borders stay zero: rows are written from column 4 on)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pl = lane & 31;
  const bool has_bn = bn_scale != nullptr;
  const int smp = (int)blockIdx.x / g.nbands, band = (int)blockIdx.x - smp * g.nbands;
  const int q_begin = band * g.steps_per_band;
  const int q_end = q_begin + g.steps_per_band < g.total_steps ? q_begin + g.steps_per_band : g.total_steps;
  const float* const xs = x + (int64_t)smp * 3 * H * W;
  const int HWo = Ho * Wo;
  // the lane's weights: W[co = pl][k = 2 s + h], the padded k a zero
  float areg[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) areg[s] = 2 * s + h < K ? wt[(2 * s + h) * COUT + pl] : 0.0f;
  __syncthreads();

  // ---- staging: eight input rows (x three channels) per call, thread -> (row, channel, 16-byte column) ---------------------
  const int W4 = W >> 2;
  int st_r[kR3ST], st_ci[kR3ST], st_c4[kR3ST];
#pragma unroll
  for (int k = 0; k < kR3ST; ++k) {
    const int idx = (int)threadIdx.x + 512 * k;
    const int line = idx / W4;
    st_c4[k] = idx - line * W4;
    st_r[k] = line < 24 ? line / 3 : -1;
    st_ci[k] = line % 3;
  }
  f4 sreg[kR3ST], sreg2[kR3ST];                                         // two steps of rows in flight
  auto stage_load = [&](f4 (&sreg)[kR3ST], int row_lo, int nrows) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < kR3ST; ++k) {
      const int iy = row_lo + st_r[k];
      const bool ok = st_r[k] >= 0 && st_r[k] < nrows && iy >= 0 && iy < H;
      sreg[k] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
      if (ok) sreg[k] = *reinterpret_cast<const f4*>(xs + ((int64_t)st_ci[k] * H + iy) * W + 4 * st_c4[k]);
    }
  };
  auto stage_store = [&](f4 (&sreg)[kR3ST], int row_lo, int nrows) __attribute__((always_inline)) {
    const int s0 = (row_lo + 1) % kR3Slots;                             // (uniform; row_lo >= -1)
#pragma unroll
    for (int k = 0; k < kR3ST; ++k) {
      if (st_r[k] < 0 || st_r[k] >= nrows) continue;
      int sl = s0 + st_r[k];
      sl -= sl >= kR3Slots ? kR3Slots : 0;
      *reinterpret_cast<f4*>(xin + (sl * 3 + st_ci[k]) * XW + 4 + 4 * st_c4[k]) = sreg[k];
    }
  };
  stage_load(sreg, 8 * q_begin - 1, 8);
  stage_load(sreg2, 8 * q_begin + 7, 1);
  stage_store(sreg, 8 * q_begin - 1, 8);
  stage_store(sreg2, 8 * q_begin + 7, 1);
  if (q_begin + 1 < q_end) stage_load(sreg2, 8 * q_begin + 8, 8);      // the second step's rows: stored at the end of the first
  __syncthreads();

  const unsigned HWo4 = (unsigned)HWo * 4u;
  const fq_rsrc yr = make_rsrc(reinterpret_cast<char*>(y) + (int64_t)smp * COUT * HWo4, (int64_t)COUT * HWo4);
  float m = 0.0f;
  // step q: the rows of step q + 2 are requested (into `ld`), the tiles of step q computed, the rows of step q + 1 (requested a step
  // ago, in `st`) written into the eight slots this step does not read - a load has a whole step and more to arrive
  auto step = [&](int q, f4 (&ld)[kR3ST], f4 (&st)[kR3ST]) __attribute__((always_inline)) {
    if (q + 2 < q_end) stage_load(ld, 8 * q + 16, 8);
    const int rows = Ho - kR3Rows * q < kR3Rows ? Ho - kR3Rows * q : kR3Rows;
    const int p0 = kR3Rows * q * Wo, nt = (rows * Wo + 31) >> 5;
    const int sbase = (8 * q) % kR3Slots;                               // slot of input row 8 q - 1
    for (int t = wave; t < nt; t += 8) {
      int jp = p0 + 32 * t + pl;
      jp = jp < HWo ? jp : HWo - 1;                                     // lanes past the image's end copy its last pixel
      const int oy = (int)fast_div((unsigned)jp, g.by_wo), ox = jp - oy * Wo;
      const int rel = 2 * (oy - kR3Rows * q);                           // input row 2 oy - 1 + ky = (8 q - 1) + rel + ky
      int rb[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        int sl = sbase + rel + ky;
        sl -= sl >= kR3Slots ? kR3Slots : 0;
        rb[ky] = sl * 3 * XW + 2 * ox + 3;                              // + column of tap kx = 0: ix + 4 = 2 ox - 1 + 4
      }
      float bv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int k0 = 2 * s, k1 = 2 * s + 1;
        const int ky0 = (k0 / 3) % 3, kx0 = k0 % 3, ci0 = k0 / 9;
        const int ky1 = k1 < K ? (k1 / 3) % 3 : 0, kx1 = k1 < K ? k1 % 3 : 0, ci1 = k1 < K ? k1 / 9 : 0;
        const int o0 = rb[ky0] + ci0 * XW + kx0;
        const int o1 = k1 < K ? rb[ky1] + ci1 * XW + kx1 : 0;           // the padded k: a border zero (its weight is zero too)
        bv[s] = xin[h ? o1 : o0];
      }
      v16f acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
      for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], bv[s], acc, 0, 0, 0);
      const unsigned yo = (unsigned)(4 * h) * HWo4 + (unsigned)jp * 4u;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int c0 = 8 * gq + 4 * h;
        const f4 bch = *reinterpret_cast<const f4*>(c_bias + c0);
        const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
        const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = dw_finish<EPI>(acc[4 * gq + r], bias != nullptr, bch[r], has_bn, bsc[r], bsh[r], act);
          if (FQ_STEM_NTS) buf_st_f32_nt(yr, yo, (unsigned)(8 * gq + r) * HWo4, v);
          else buf_st_f32(yr, yo, (unsigned)(8 * gq + r) * HWo4, v);
          m = fmaxf(m, fabsf(v));
        }
      }
    }
    if (q + 1 < q_end) stage_store(st, 8 * q + 8, 8);                   // (slots no tile of this step reads)
    __syncthreads();
  };
  for (int q = q_begin; q < q_end; q += 2) {
    step(q, sreg, sreg2);
    if (q + 1 < q_end) step(q + 1, sreg2, sreg);                        // (uniform)
  }
  if (stat_out != nullptr) {
    const float wm = wave_max_nonneg(m);
    if (lane == 0) red[wave] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = red[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) t = fmaxf(t, red[i]);
      if (__float_as_uint(t) != 0u) atomic_max_f32(stat_out + smp, t);
    }
  }
}

}  // namespace

extern "C" {

// shared by both entry points: ksize 3 (3 -> 32) or 7 (3 -> 64)
static int stem_launch(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n, int64_t cin,
                       int64_t cout, int64_t h, int64_t w, int ksize, const float* bn_scale, const float* bn_shift, int act,
                       float* stat_out, fqStream_t stream, const char* who, const float* out_thr = nullptr, int out_width = 8,
                       unsigned out_flags = 0) {
  FQ_REQUIRE(x && w_tap_major && y, "%s: null pointer", who);
  FQ_REQUIRE(n > 0 && n < 65536 && h > 0 && w > 0 && h < (1 << 15) && w < (1 << 15) && 3 * h * w * 4 < (1ll << 31),
             "%s: bad shape", who);
  FQ_REQUIRE(cin == 3 && ((ksize == 3 && cout == 32) || (ksize == 7 && cout == 64)), "%s: only 3 -> 32 channels with a "
             "3x3 kernel and 3 -> 64 with a 7x7 kernel are built (got %lld -> %lld, %dx%d)", who, (long long)cin,
             (long long)cout, ksize, ksize);
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "%s: bn_scale and bn_shift go together", who);
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "%s: unknown activation %d", who, act);
  hipStream_t st = (hipStream_t)stream;
  const int pad = ksize / 2;
  const int Ho = (int)((h + 2 * pad - ksize) / 2 + 1), Wo = (int)((w + 2 * pad - ksize) / 2 + 1);
  const int64_t hwo = (int64_t)Ho * Wo;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  ProfScope prof(FQ_KERNEL_STEM, 4.0 * ((double)n * cin * h * w + (double)n * cout * hwo), st,
                 4.0 * (double)n * cin * h * w + (out_thr != nullptr ? 1.0 : 4.0) * (double)n * cout * hwo);
  static const int form = env_int("FQ_STEM_FORM", 0);                   // tuning: 0 auto, 1 VALU form (3x3 only), 2 MFMA form
  StemCodes oc;
  oc.thr = out_thr; oc.levels = 0.0f; oc.lo_neg = 0; oc.zoff = 0; oc.CBo = (int)((cout + 15) / 16);
  if (out_thr != nullptr) {
    FQ_REQUIRE(ksize == 3, "%s: the code output is built for the 3x3 form", who);
    FQ_REQUIRE(out_width >= 2 && out_width <= 8 && !(out_flags & (FQ_ACT_NO_ABS | FQ_ACT_NO_EPS)), "%s: bad output quantiser", who);
    FQ_REQUIRE((int64_t)oc.CBo * hwo * 16 < (1ll << 31), "%s: output plane too large", who);
    oc.levels = act_levels(out_width, out_flags);
    oc.lo_neg = (out_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
    oc.zoff = (out_flags & FQ_ACT_SIGNED) ? 0 : 128;
  }
  if (ksize == 3 && form == 1 && out_thr == nullptr) {
    const int tiles = (int)((hwo + kBlock - 1) / kBlock);
    // enough workgroups to fill the chip, as few statistic atomics per sample as that allows
    int tiles_per_wg = 1;
    while (tiles_per_wg < 8 && n * ((tiles + 2 * tiles_per_wg - 1) / (2 * tiles_per_wg)) >= (int64_t)num_cu() * 8) tiles_per_wg *= 2;
    const dim3 grid((unsigned)((tiles + tiles_per_wg - 1) / tiles_per_wg), (unsigned)n);
    hipLaunchKernelGGL((stem_conv3x3s2_kernel<3, 32>), grid, dim3(kBlock), 0, st, x, w_tap_major, bias, y, (int)h, (int)w,
                       Ho, Wo, tiles_per_wg, bn_scale, bn_shift, act, stat_out);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
  }
  // K2r: the 3x3 form with its input rows staged in LDS (fp32 output; rows of a multiple of eight output columns so that a
  // step of four rows is whole tiles; FQ_STEM_ROWS=0: never)
  static const int use_rows = env_int("FQ_STEM_ROWS", 1);
  if (ksize == 3 && out_thr == nullptr && form == 0 && use_rows != 0 && (w & 3) == 0 && (Wo & 7) == 0 && 6 * w <= 512 * kR3ST &&
      ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(y)) & 15) == 0 && (int64_t)cout * hwo * 4 < (1ll << 31)) {
    Stem3Geom g3;
    g3.H = (int)h; g3.W = (int)w; g3.Ho = Ho; g3.Wo = Wo;
    g3.total_steps = (Ho + kR3Rows - 1) / kR3Rows;
    static const int rows_wg = env_int("FQ_STEM_ROWS_WG", 2);          // workgroups per CU the grid is cut for
    int nb = (int)(((rows_wg > 0 ? rows_wg : 2) * (int64_t)num_cu() + n - 1) / n);   // as few bands as that allows
    nb = nb < 1 ? 1 : (nb > g3.total_steps ? g3.total_steps : nb);
    g3.steps_per_band = (g3.total_steps + nb - 1) / nb;
    g3.nbands = (g3.total_steps + g3.steps_per_band - 1) / g3.steps_per_band;
    g3.by_wo = fast_div_for((unsigned)Wo);
    const size_t lds3 = (size_t)(3 * 32 + 32 + kR3Slots * 3 * (w + 8)) * sizeof(float);
    const int epi3 = (bn_scale != nullptr && bias == nullptr)
                         ? (act == FQ_ACT_RELU ? kEpiBnRelu : act == FQ_ACT_RELU6 ? kEpiBnRelu6 : kEpiRuntime)
                         : kEpiRuntime;
    const dim3 grid3((unsigned)(n * g3.nbands));
#define FQ_STEM3_LAUNCH(E_)                                                                                            \
  {                                                                                                                    \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem3_rows_kernel<E_>),             \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess; \
    FQ_REQUIRE(attr_ok, "%s: cannot raise the dynamic LDS limit", who);                                                \
    hipLaunchKernelGGL((stem3_rows_kernel<E_>), grid3, dim3(512), lds3, st, x, w_tap_major, bias, y, g3, bn_scale,     \
                       bn_shift, act, stat_out);                                                                       \
  }
    if (lds3 <= 64 * 1024) {
      if (epi3 == kEpiBnRelu) FQ_STEM3_LAUNCH(kEpiBnRelu)
      else if (epi3 == kEpiBnRelu6) FQ_STEM3_LAUNCH(kEpiBnRelu6)
      else FQ_STEM3_LAUNCH(kEpiRuntime)
      FQ_LAUNCH_CHECK();
      return FQ_OK;
    }
#undef FQ_STEM3_LAUNCH
  }
  const int tiles_per_img = (int)((hwo + 31) / 32);
  const int64_t total = (int64_t)tiles_per_img * n;
  // persistent workgroups, all resident: four per CU for the 3x3 form (86 -> 77 us against the VALU form's 86 in the
  // MobileNet step; six: 89), two for the 7x7 form (208 registers + 38 KB of weights in LDS)
  static const int wg_tune = env_int("FQ_STEM_WG_PER_CU", 0);
  const int wg_per_cu = wg_tune > 0 ? wg_tune : (ksize == 3 ? 4 : 2);
  int64_t grid = (int64_t)num_cu() * wg_per_cu;
  if (grid > (total + 3) / 4) grid = (total + 3) / 4;
  FQ_REQUIRE(total < (1ll << 31), "%s: too many tiles", who);
  StemGeom g;
  g.H = (int)h; g.W = (int)w; g.Ho = Ho; g.Wo = Wo;
  g.tiles_per_img = (unsigned)tiles_per_img; g.total_tiles = (unsigned)total; g.n_samples = (unsigned)n;
  g.per = (unsigned)(total / (grid * 4)); g.rem = (unsigned)(total % (grid * 4));
  g.by_tpi = fast_div_for((unsigned)tiles_per_img);
  g.by_wo = fast_div_for((unsigned)Wo);
  const int epi = (bn_scale != nullptr && bias == nullptr)
                      ? (act == FQ_ACT_RELU ? kEpiBnRelu : act == FQ_ACT_RELU6 ? kEpiBnRelu6 : kEpiRuntime)
                      : kEpiRuntime;
#define FQ_STEM_LAUNCH(KS_, CO_, E_)                                                                                \
  hipLaunchKernelGGL((stem_mfma_kernel<KS_, CO_, E_>), dim3((unsigned)grid), dim3(kBlock), 0, st, x, w_tap_major, bias, \
                     y, g, bn_scale, bn_shift, act, stat_out, oc)
#define FQ_STEM_LAUNCH16(E_)                                                                                        \
  hipLaunchKernelGGL((stem_mfma_kernel<3, 32, E_, true>), dim3((unsigned)grid), dim3(kBlock), 0, st, x, w_tap_major, bias, \
                     y, g, bn_scale, bn_shift, act, stat_out, oc)
#define FQ_STEM_EPI(KS_, CO_)                                                                                       \
  do {                                                                                                              \
    if (epi == kEpiBnRelu) FQ_STEM_LAUNCH(KS_, CO_, kEpiBnRelu);                                                    \
    else if (epi == kEpiBnRelu6) FQ_STEM_LAUNCH(KS_, CO_, kEpiBnRelu6);                                             \
    else FQ_STEM_LAUNCH(KS_, CO_, kEpiRuntime);                                                                     \
  } while (0)
  if (out_thr != nullptr) {
    if (epi == kEpiBnRelu) FQ_STEM_LAUNCH16(kEpiBnRelu);
    else if (epi == kEpiBnRelu6) FQ_STEM_LAUNCH16(kEpiBnRelu6);
    else FQ_STEM_LAUNCH16(kEpiRuntime);
  } else if (ksize == 3) FQ_STEM_EPI(3, 32);
  else FQ_STEM_EPI(7, 64);
#undef FQ_STEM_EPI
#undef FQ_STEM_LAUNCH16
#undef FQ_STEM_LAUNCH
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_stem_conv3x3s2(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n, int64_t cin,
                      int64_t cout, int64_t h, int64_t w, const float* bn_scale, const float* bn_shift, int act,
                      float* stat_out, fqStream_t stream) {
  return stem_launch(x, w_tap_major, bias, y, n, cin, cout, h, w, 3, bn_scale, bn_shift, act, stat_out, stream,
                     "fq_stem_conv3x3s2");
}

int fq_stem_conv3x3s2_c16(const float* x, const float* w_tap_major, const float* bias, void* y16, int64_t n, int64_t cin,
                          int64_t cout, int64_t h, int64_t w, const float* bn_scale, const float* bn_shift, int act,
                          float* stat_out, const float* out_thr, int out_width, unsigned out_flags, fqStream_t stream) {
  FQ_REQUIRE(out_thr != nullptr, "fq_stem_conv3x3s2_c16: out_thr is the consumer's stored threshold");
  return stem_launch(x, w_tap_major, bias, (float*)y16, n, cin, cout, h, w, 3, bn_scale, bn_shift, act, stat_out, stream,
                     "fq_stem_conv3x3s2_c16", out_thr, out_width, out_flags);
}

int fq_stem_conv7x7s2(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n, int64_t cin,
                      int64_t cout, int64_t h, int64_t w, const float* bn_scale, const float* bn_shift, int act,
                      float* stat_out, fqStream_t stream) {
  return stem_launch(x, w_tap_major, bias, y, n, cin, cout, h, w, 7, bn_scale, bn_shift, act, stat_out, stream,
                     "fq_stem_conv7x7s2");
}

}  // extern "C"
