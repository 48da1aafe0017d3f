// libfakequant — K2y: a pointwise (1x1) convolution on int8 codes AND the depthwise 3x3 behind it in one launch
// (see fq_common.h for the list of translation units and the design rules)
//
// Why (round 6): under ONLINE thresholds every fused producer writes its output in fp32 only for the next launch to read it
// back - the batch statistic that scales the consumer's quantiser (convert_conv2d.py:56-58: mean over the batch of the
// per-sample maxima of the WHOLE tensor) is not known before the producer's last workgroup has finished, so codes cannot be
// handed over.  But nothing forbids computing the producer twice: a statistic-only pass (fq_pwconv_i8_stat: the pointwise
// kernel without its stores) leaves the per-sample maxima, and this kernel then recomputes the pointwise outputs from the
// SAME input - bit for bit the same fp32 values: exact int32 sums, the same epilogue - quantises them with the now known
// threshold and feeds the depthwise stencil without the tensor ever reaching HBM.  A MobileNet pair (x -> 1x1 -> y -> 3x3 -> z,
// y twice as many channels as x) moves 4x + 4x + 4z bytes instead of 4x + 4y + 4y + 4z: 0.31 instead of 1.13 GB on the first
// pair of MobileNet1.0 at batch 128.  (The other direction - depthwise into pointwise - saves one tensor of x's size per pair;
// this one saves the EXPANDED tensor twice.)
//
// Form.  v_mfma_i32_32x32x32_i8 with the ACTIVATIONS as the A operand and the weight fragment as B, i.e. D = pixels x
// channels: lane l owns output channel (l & 31) and its 16 registers are 16 PIXELS.  The 32 MFMA rows are given to the
// pixels of a 32-column strip of one image row in an order that makes those 16 pixels consecutive columns (row i = 8g + 4h +
// i' holds pixel 16h + 4g + i'; the loads of a channel still cover one 128-byte line), so a lane holds columns 16h .. 16h+15 of
// ITS channel: the depthwise stencil's horizontal neighbours are neighbouring registers (one v_permlane32_swap per row hands
// column 15 / 16 across the two lane halves), its weights and both BatchNorms are 15 per-lane registers instead of LDS reads,
// and the vertical neighbours are the same registers one input row later: a wavefront walks its strip down a band of rows and
// keeps, per pixel, the partial sums of the output rows still open (stride 1: two, stride 2: one), adding each input row's
// three products in the order of the row-major 3x3 chain (fq_dwconv3x3's arithmetic: fmaf by fmaf from +0).
// A workgroup = all strips of a band x the channel groups; finished output rows go through an LDS row buffer (two rows) and
// leave as whole rows, 16 bytes per lane.
#include "fq_pw.h"

#include <climits>

namespace {

struct PwDwGeom {
  int Cin, KTS, Cout, H, W, Ho, Wo;   // KTS: 32-channel slabs per row of the fragment-major weight copy (cin_pad / 32)
  int strips, ctg, bands, RB;         // column strips per row, channel groups (Cout / 32 / CW), row bands, output rows per band
  int zoff, pitch;                    // pitch: floats per channel in the LDS row buffer (odd: conflict-free lane = channel writes)
  int vw;                             // floats per lane of the row store: 4, 2 or 1 (Wo % vw == 0)
  FastDiv qv;                         // Wo / vw
};

constexpr int kRowSlack = 8;          // floats in front of each row buffer (a strip's invalid first column lands there)

// slabs in flight per lane: a ring of NB buffers of 16 values, the loads of slab q + NB - 1 issued before slab q is quantised
// (stride 1 keeps 32 more partial sums per channel tile: two buffers there)
#ifdef FQ_PWDW_OCC3      // tuning build: three wavefronts per SIMD (at most 168 registers), two buffers
#define FQ_PWDW_OCC __attribute__((amdgpu_waves_per_eu(3, 3)))
template <int KT, int S> struct PwDwRing { static constexpr int NB = 2; };
#else
#define FQ_PWDW_OCC
template <int KT, int S> struct PwDwRing { static constexpr int NB = S == 1 ? 2 : (KT <= 2 ? 3 : 4); };
#endif

// FAST 0: every epilogue decided at run time, general quantisers; 1: the fused-inference case - no bias, BatchNorm and ReLU
// (ReLU6 when `relu6`) behind both convolutions, unsigned activations (both clip ranges start at 0)
template <int KT, int CW, int S, int LZ, int FAST>
__global__ __launch_bounds__(512) FQ_PWDW_OCC void pwdw_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias1, PwDwGeom g, const float* __restrict__ in_stat, int n,
    const float* __restrict__ in_thr, float levels1, int lo_neg1, float eps, const float* __restrict__ bn1_scale,
    const float* __restrict__ bn1_shift, int act1, const float* __restrict__ mid_stat, const float* __restrict__ mid_thr,
    float levels2, int lo_neg2, float* __restrict__ mid_cur_out, const float* __restrict__ dww,
    const float* __restrict__ bias2, const float* __restrict__ bn2_scale, const float* __restrict__ bn2_shift, int act2,
    float* __restrict__ y, float* __restrict__ stat_out) {
  constexpr int NB = PwDwRing<KT, S>::NB;
  extern __shared__ __attribute__((aligned(16))) unsigned char pwdw_smem[];
  __shared__ float red[8];
  const int CT = g.ctg * CW;
  v4i* ldsW = reinterpret_cast<v4i*>(pwdw_smem);                                    // [CT][KT][64] fragments
  const int rb_stride = g.Cout * g.pitch + kRowSlack;
  float* rowbuf = reinterpret_cast<float*>(pwdw_smem + (size_t)CT * KT * 1024) + kRowSlack;     // [2][slack + Cout * pitch]
  // stride 1 keeps two open sums per pixel: the depthwise layer's per-channel constants stay in LDS there (13 floats per channel:
  // an odd stride, every lane its own bank) and are read where they are used
  constexpr bool kDwLds = S == 1;
  float* ldsK = rowbuf + 2 * rb_stride;                                                          // [Cout][13]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, i32 = lane & 31;
  const int s = wave % g.strips, cgi = wave / g.strips;
  const int band = (int)(blockIdx.x % (unsigned)g.bands);
  const int smp = n - 1 - (int)(blockIdx.x / (unsigned)g.bands);                    // last samples first: what the statistic pass
                                                                                    // read last is what the caches still hold
  const int HW = g.H * g.W;

  // ---- geometry of this wave's strip ----------------------------------------------------------------------------------------
  // local pixel p = 0 .. 31 of the strip is input column c0 + p - 1.  Stride 1: 30 outputs per strip (p = 1 .. 30); the LAST
  // strip is anchored at the right edge (c0 = W - 30: the columns it shares with its neighbour come out the same from both), so
  // the zero column right of the image is always p = 31; a plane narrower than 30 is one strip whose zero column LEFT of the
  // image is p = LZ = 30 - W.  Stride 2 (even W): 15 outputs per strip, centre p = 2 oc + 1; only the left zero column exists.
  int c0;
  if (S == 1) c0 = (s == g.strips - 1) ? g.W - 30 : 30 * s;      // (W < 30: one strip, c0 = -LZ)
  else c0 = 30 * s;
  // A-operand pixel of this lane (MFMA row i32): p = 16 * ((i32 >> 2) & 1) + 4 * (i32 >> 3) + (i32 & 3)
  const int pl = 16 * ((i32 >> 2) & 1) + 4 * (i32 >> 3) + (i32 & 3);
  int col_ld = c0 + pl - 1;
  col_ld = col_ld < 0 ? 0 : (col_ld >= g.W ? g.W - 1 : col_ld);
  const fq_rsrc rs = make_rsrc(x + (int64_t)smp * g.Cin * HW, (int64_t)g.Cin * HW * 4);
  const unsigned voff = (unsigned)((16 * h * HW + col_ld) * 4);
  // LDS float offset of this lane's output 0 inside a channel row: stride 1 - own pixel k is output column c0 + 16 h + k - 1
  // (k = 0 of the lower half and k = 15 of the upper half are no outputs); stride 2 - output oc = 8 h + c is column 15 s + oc
  // (c = 7 of the upper half is no output).  Columns past the plane fall into the row's padding.
  const int obase = S == 1 ? c0 + 16 * h - 1 : 15 * s + 8 * h;

  const int ro0 = band * g.RB;
  const int ro1 = ro0 + g.RB < g.Ho ? ro0 + g.RB : g.Ho;
  const int t_first = S == 1 ? ro0 - 1 : 2 * ro0 - 1;
  const int t_last = S == 1 ? ro1 : 2 * (ro1 - 1) + 1;            // inclusive (may be H: a zero row)
  float* yb = y + (int64_t)smp * g.Cout * g.Ho * g.Wo;

  // slab q of the band = (row t_first + q / KT, 32-channel slab q % KT); buffer q % NB
  float raw[NB][16];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) raw[b][i] = 0.0f;
  auto issue = [&](int t, int kt, float (&v)[16]) __attribute__((always_inline)) {
    if (t < 0 || t >= g.H || t > t_last) return;
    const unsigned so = (unsigned)((kt * 32 * HW + t * g.W) * 4);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = buf_ld_f32(rs, voff, so + (unsigned)(i * HW * 4));
  };
#pragma unroll
  for (int q = 0; q < NB - 1; ++q) issue(t_first + q / KT, q % KT, raw[q]);
  FQ_PIN();

  // ---- thresholds: the pointwise input's (statistic of x) and the depthwise input's (statistic of the pointwise output) ----
  const float max1 = input_threshold(in_stat, n, in_thr, nullptr, false);
  const float max2 = input_threshold(mid_stat, n, mid_thr, mid_cur_out, blockIdx.x == 0);
  const QParams q1 = make_qparams(max1, levels1, lo_neg1 != 0, eps);
  const QParams q2 = make_qparams(max2, levels2, lo_neg2 != 0, eps);
  const int ubias1 = 128 - g.zoff;
  const unsigned nn_xor1 = fq_nonneg_xor(ubias1);
  const bool relu6 = act1 == FQ_ACT_RELU6;                        // (FAST: both activations are the same)
  const float top = relu6 ? 6.0f : INFINITY;

  // ---- weights -> LDS (already in fragment order in HBM) ----------------------------------------------------------------
  for (int idx = threadIdx.x; idx < CT * KT * 64; idx += blockDim.x) {
    const int f = idx >> 6, ct = f / KT, kt = f - ct * KT;
    ldsW[idx] = *reinterpret_cast<const v4i*>(wfrag + (((int64_t)ct * g.KTS + kt) << 10) + ((idx & 63) << 4));
  }

  // ---- per-lane constants: this lane's channel of each of the wave's CW channel tiles -------------------------------------
  float k_sxw[CW], k_b1[CW], k_bsc1[CW], k_bsh1[CW], k_w[CW][9], k_b2[CW], k_bsc2[CW], k_bsh2[CW];
  int k_zs[CW];
  const bool has_b1 = bias1 != nullptr, has_bn1 = bn1_scale != nullptr, has_b2 = bias2 != nullptr, has_bn2 = bn2_scale != nullptr;
#pragma unroll
  for (int j = 0; j < CW; ++j) {
    const int c = (cgi * CW + j) * 32 + i32;
    k_sxw[j] = q1.scale * wscale[c];
    k_zs[j] = g.zoff * wsum[c];
    k_b1[j] = has_b1 ? bias1[c] : 0.0f;
    k_bsc1[j] = has_bn1 ? bn1_scale[c] : 1.0f;
    k_bsh1[j] = has_bn1 ? bn1_shift[c] : 0.0f;
#pragma unroll
    for (int t = 0; t < 9; ++t) k_w[j][t] = kDwLds ? 0.0f : dww[c * 9 + t];
    k_b2[j] = (has_b2 && !kDwLds) ? bias2[c] : 0.0f;
    k_bsc2[j] = (has_bn2 && !kDwLds) ? bn2_scale[c] : 1.0f;
    k_bsh2[j] = (has_bn2 && !kDwLds) ? bn2_shift[c] : 0.0f;
  }
  if (kDwLds)
    for (int c = threadIdx.x; c < g.Cout; c += blockDim.x) {
#pragma unroll
      for (int t = 0; t < 9; ++t) ldsK[c * 13 + t] = dww[c * 9 + t];
      ldsK[c * 13 + 9] = has_b2 ? bias2[c] : 0.0f;
      ldsK[c * 13 + 10] = has_bn2 ? bn2_scale[c] : 1.0f;
      ldsK[c * 13 + 11] = has_bn2 ? bn2_shift[c] : 0.0f;
    }
  const bool zero_left = (S == 1 && LZ != 0) ? (h == (LZ >> 4)) : (s == 0 && h == 0);       // lanes holding the column left of the image
  const bool zero_right = S == 1 && s == g.strips - 1 && h == 1;                             // ... right of it (p = 31)
  __syncthreads();                                                // weights in LDS
  FQ_PIN();

  float m = 0.0f;
  const FastQuot fq1 = make_fast_quot(q1.denom), fq2 = make_fast_quot(q2.denom);
  // NN: both clip ranges start at 0 AND both divisors take the fp32 quotient (fq_common.h: fast_quot); else the general forms
  auto body = [&](auto nn_c) __attribute__((always_inline)) {
    constexpr bool NN = decltype(nn_c)::value;
    auto pack_x = [&](float a, float b, float c, float d) __attribute__((always_inline)) {
      if constexpr (NN) {
        auto code = [&](float v) __attribute__((always_inline)) {
          int r;
          const float Q = fast_quot(fq_clip(v, q1), fq1);
          asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(Q));
          return (unsigned)r;
        };
        unsigned u = code(a);
        u |= code(b) << 8;
        u |= code(c) << 16;
        u |= code(d) << 24;
        return (int)(u ^ nn_xor1);
      } else {
        return fq_pack4<false>(a, b, c, d, q1, ubias1, nn_xor1);
      }
    };
    // open partial sums: stride 1 - A (output row t-1: has its first two tap rows) and B (output row t: its first);
    // stride 2 - A only (8 outputs per lane)
    constexpr int NO = S == 1 ? 16 : 8;
    float sumA[CW][NO], sumB[S == 1 ? CW : 1][NO];
#pragma unroll
    for (int j = 0; j < CW; ++j)
#pragma unroll
      for (int k = 0; k < NO; ++k) {
        sumA[j][k] = 0.0f;
        if (S == 1) sumB[j][k] = 0.0f;
      }
    QParams qc = q2;                               // FAST: ReLU / ReLU6 and the clip as one median (clip range starts at 0)
    if (FAST) qc.hi = fminf(q2.hi, top);

    // one input row; PH: its first slab is slab (PH * KT) mod NB of the ring
    auto row = [&](int t, auto ph_c) __attribute__((always_inline)) {
      constexpr int PH = decltype(ph_c)::value;
      const bool row_ok = t >= 0 && t < g.H;
      v4i afrag[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        constexpr int dummy = 0;
        (void)dummy;
        const int qn = kt + NB - 1;                                  // the slab NB - 1 ahead goes into the buffer freed last
        issue(t + qn / KT, qn % KT, raw[(PH * KT + kt + NB - 1) % NB]);
        FQ_PIN();
        float (&mine)[16] = raw[(PH * KT + kt) % NB];
        v4i f;
#pragma unroll
        for (int d = 0; d < 4; ++d)
          f[d] = pack_x(mine[4 * d + 0], mine[4 * d + 1], mine[4 * d + 2], mine[4 * d + 3]);
        asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
        afrag[kt] = f;
        FQ_PIN();
      }
      const int r_emit = S == 1 ? t - 1 : (t - 1) >> 1;          // output row an odd (S = 2) / any (S = 1) input row closes
      const bool emits = (S == 1 || (t & 1)) && r_emit >= ro0 && r_emit < ro1;
      float* rbuf = rowbuf + (r_emit & 1) * rb_stride;
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const int ct = cgi * CW + j;
        const float* kc = ldsK + (ct * 32 + i32) * 13;
        // a finished row of this lane's channel -> LDS row buffer (the statistic is taken when the row leaves it)
        auto finish_store = [&](const float (&sum)[NO]) __attribute__((always_inline)) {
          if (!emits) return;
          float* dst = rbuf + (ct * 32 + i32) * g.pitch + obase;
          const float b2 = kDwLds ? kc[9] : k_b2[j], bsc2 = kDwLds ? kc[10] : k_bsc2[j], bsh2 = kDwLds ? kc[11] : k_bsh2[j];
#pragma unroll
          for (int k = 0; k < NO; ++k) {
            float o;
            if (FAST == 0) {
              o = dw_finish<kEpiRuntime>(sum[k], has_b2, b2, has_bn2, bsc2, bsh2, act2);
            } else {
              o = sum[k] * bsc2;
              o = o + bsh2;
              o = __builtin_amdgcn_fmed3f(o, 0.0f, top);      // = min(max(o, 0), top), a NaN gives 0 either way
            }
            if (S == 1 && k == 0) {
              if (h == 1) dst[k] = o;
            } else if (k == NO - 1) {
              if (h == 0) dst[k] = o;
            } else {
              dst[k] = o;
            }
          }
        };
        if (!row_ok) {
          // a row of zero padding adds nothing to any sum (every product is +-0 and the sums are never -0): the row it
          // closes leaves, the open sums move up
          if (S == 1) {
            finish_store(sumA[j]);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
              sumA[j][k] = sumB[j][k];
              sumB[j][k] = 0.0f;
            }
          } else if (t & 1) {
            finish_store(sumA[j]);
#pragma unroll
            for (int c = 0; c < 8; ++c) sumA[j][c] = 0.0f;
          }
          FQ_PIN();
          continue;
        }
        float e[18];                                // dequantised pointwise outputs: e[k + 1] = own pixel k, e[0] / e[17] the neighbours'
        {
          // (the re-centring term zoff * rowsum joins AFTER the multiplication: as the accumulator's initial value it is a
          // loop-invariant 16-register splat per channel tile, which the compiler keeps alive across the whole row loop)
          v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          int wl = lane;                             // (opaque per row: the weight fragments are re-read from LDS, not kept in
          asm volatile("" : "+v"(wl));               //  CW * KT * 4 registers across the row loop)
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[kt], ldsW[((ct * KT + kt) << 6) + wl], acc, 0, 0, 0);
#ifdef FQ_PWDW_PK        // measured and left out: 64 packed instead of 128 plain instructions per channel tile and row, 2 % SLOWER
                         // (pair 1: 98.0 against 95.9 us, the step -0.36 %: profiles/r6_pwdw_packed_ab.txt)
          if constexpr (FAST != 0 && NN) {
            // the fused-inference chain two pixels at a time as packed fp32 instructions (two IEEE operations each: the same
            // values)
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
              f2 v = (f2){(float)(acc[k] + k_zs[j]), (float)(acc[k + 1] + k_zs[j])};
              v = v * (f2){k_sxw[j], k_sxw[j]};
              v = v * (f2){k_bsc1[j], k_bsc1[j]};
              v = v + (f2){k_bsh1[j], k_bsh1[j]};
              f2 c = (f2){fq_clip(v.x, qc), fq_clip(v.y, qc)};
              f2 q = c * (f2){fq2.y, fq2.y};
              const f2 nd = (f2){-fq2.d, -fq2.d};
              const f2 r = __builtin_elementwise_fma(q, nd, c);          // fma(-q, d, c) = fma(q, -d, c)
              q = __builtin_elementwise_fma(r, (f2){fq2.y, fq2.y}, q);
              q = q + (f2){0.49999997f, 0.49999997f};
              q = (f2){truncf(q.x), truncf(q.y)};
              q = q * (f2){q2.scale, q2.scale};
              e[k + 1] = q.x;
              e[k + 2] = q.y;
            }
          } else
#endif
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            float v = (float)(acc[k] + k_zs[j]) * k_sxw[j];
            if (FAST == 0) {
              if (has_b1) v = v + k_b1[j];
              if (has_bn1) {
                v = v * k_bsc1[j];
                v = v + k_bsh1[j];
              }
              v = act_rt(v, act1);
              e[k + 1] = NN ? truncf(fast_quot(fq_clip(v, q2), fq2) + 0.49999997f) * q2.scale : fq_code(v, q2) * q2.scale;
            } else {
              v = v * k_bsc1[j];
              v = v + k_bsh1[j];
              // clip(relu(v), 0, hi) = med3(v, 0, hi); the quotient is non-negative: roundf = trunc(Q + pred(0.5))
              if (NN) e[k + 1] = truncf(fast_quot(fq_clip(v, qc), fq2) + 0.49999997f) * q2.scale;
              else e[k + 1] = fq_code(act_rt(v, relu6 ? FQ_ACT_RELU6 : FQ_ACT_RELU), q2) * q2.scale;
            }
          }
          // the zero padding of the depthwise layer: columns left / right of the image
          e[1 + (LZ & 15)] = zero_left ? 0.0f : e[1 + (LZ & 15)];
          if (S == 1) e[16] = zero_right ? 0.0f : e[16];
          // column 16 h - 1 and 16 h + 16: the other half's pixel 15 / pixel 0
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(e[1]), __float_as_uint(e[16]), false, false);
          e[0] = __uint_as_float(sw[0]);             // upper half: the lower half's pixel 15 (lower half: its own pixel 0, unused)
          e[17] = __uint_as_float(sw[1]);            // lower half: the upper half's pixel 0 (upper half: its own pixel 15, unused)
        }
        float w[9];
        if (!kDwLds) {
#pragma unroll
          for (int t = 0; t < 9; ++t) w[t] = k_w[j][t];
        }
        if (S == 1) {
          // A: += tap row 2 -> output row t - 1 complete; B: += tap row 1; C (new): tap row 0 from +0
          w[6] = kc[6]; w[7] = kc[7]; w[8] = kc[8];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            float a = sumA[j][k];
            a = fmaf(w[6], e[k], a);
            a = fmaf(w[7], e[k + 1], a);
            a = fmaf(w[8], e[k + 2], a);
            sumA[j][k] = a;
          }
          finish_store(sumA[j]);
          w[0] = kc[0]; w[1] = kc[1]; w[2] = kc[2]; w[3] = kc[3]; w[4] = kc[4]; w[5] = kc[5];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            float b = sumB[j][k];
            b = fmaf(w[3], e[k], b);
            b = fmaf(w[4], e[k + 1], b);
            b = fmaf(w[5], e[k + 2], b);
            float c = 0.0f;
            c = fmaf(w[0], e[k], c);
            c = fmaf(w[1], e[k + 1], c);
            c = fmaf(w[2], e[k + 2], c);
            sumA[j][k] = b;
            sumB[j][k] = c;
          }
        } else {
          // output oc = 8 h + c: columns p = 2c, 2c + 1, 2c + 2 of the own 16 (+ the neighbour's first) = e[2c + 1 .. 2c + 3]
          if (t & 1) {                              // t = 2r + 1: tap row 2 of output r, then tap row 0 of output r + 1
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              float a = sumA[j][c];
              a = fmaf(w[6], e[2 * c + 1], a);
              a = fmaf(w[7], e[2 * c + 2], a);
              a = fmaf(w[8], e[2 * c + 3], a);
              sumA[j][c] = a;
            }
            finish_store(sumA[j]);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              float a = 0.0f;
              a = fmaf(w[0], e[2 * c + 1], a);
              a = fmaf(w[1], e[2 * c + 2], a);
              a = fmaf(w[2], e[2 * c + 3], a);
              sumA[j][c] = a;
            }
          } else {                                  // t = 2r: tap row 1 of output r
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              float a = sumA[j][c];
              a = fmaf(w[3], e[2 * c + 1], a);
              a = fmaf(w[4], e[2 * c + 2], a);
              a = fmaf(w[5], e[2 * c + 3], a);
              sumA[j][c] = a;
            }
          }
        }
        FQ_PIN();                                   // (one channel tile after the other: their registers need not coexist)
      }
      // ---- a finished output row leaves: LDS row buffer -> whole rows, 16 / 8 / 4 bytes per lane; its maximum on the way ----
      if (emits) {
        __syncthreads();
        const unsigned per_row = g.qv.d;                      // vectors per channel row
        const int vecs = g.Cout * (int)per_row;
        float* yr = yb + (int64_t)r_emit * g.Wo;
        for (int idx = threadIdx.x; idx < vecs; idx += blockDim.x) {
          const unsigned c = fast_div((unsigned)idx, g.qv), qd = (unsigned)idx - c * per_row;
          if (g.vw == 4) {
            const float* src = rbuf + c * g.pitch + 4 * qd;
            const f4 v = (f4){src[0], src[1], src[2], src[3]};
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            *reinterpret_cast<f4*>(yr + (int64_t)c * g.Ho * g.Wo + 4 * qd) = v;
          } else if (g.vw == 2) {
            const float* src = rbuf + c * g.pitch + 2 * qd;
            const float2 v = make_float2(src[0], src[1]);
            m = fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y)));
            *reinterpret_cast<float2*>(yr + (int64_t)c * g.Ho * g.Wo + 2 * qd) = v;
          } else {
            const float v = rbuf[c * g.pitch + qd];
            m = fmaxf(m, fabsf(v));
            yr[(int64_t)c * g.Ho * g.Wo + qd] = v;
          }
        }
      }
    };
    // rows in groups whose ring phases are static: U rows per turn, U * KT a multiple of NB
    constexpr int U = (KT % NB == 0) ? 1 : ((2 * KT) % NB == 0 ? 2 : NB);
    using std::integral_constant;
    for (int t = t_first; t <= t_last; t += U) {
      row(t, integral_constant<int, 0>{});
      if (U > 1 && t + 1 <= t_last) row(t + 1, integral_constant<int, 1 % U>{});
      if (U > 2 && t + 2 <= t_last) row(t + 2, integral_constant<int, 2 % U>{});
    }
  };
  using std::false_type;
  using std::true_type;
  if (fq_nonneg(q1) && fq_nonneg(q2) && fq1.ok && fq2.ok) body(true_type{});
  else body(false_type{});

  if (stat_out != nullptr) {
    m = wave_max_nonneg(m);
    __syncthreads();
    if (lane == 0) red[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      float mm = 0.0f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) mm = fmaxf(mm, red[w]);
      if (__float_as_uint(mm) != 0u) FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + smp, __float_as_uint(mm));
    }
  }
}

// K2z: the statistic-only pass of a pointwise convolution (fq_pwconv_i8_stat): stat_out[n] <- max |act(BN(conv))| over the
// sample, nothing else leaves the chip.  Same operand roles as above (lane = output channel, registers = 16 pixels of a
// 32-pixel tile), and ONE more observation: every step of the epilogue - int32 -> fp32, * sx*sw, + bias, * bn_scale, + bn_shift,
// the activation - is a monotone function of the integer sum (rounding is monotone, a negative factor only turns the order
// round), so the maximum of |act(BN(.))| over a set of pixels is attained at the largest or the smallest INTEGER sum of the
// channel.  The kernel therefore keeps, per channel and sample, max and min of the int32 sums (v_max3 / v_min3: one
// instruction per output value instead of the storing kernel's six) and runs the fp32 epilogue on those two only, once per
// sample: bit for bit the statistic of fq_pwconv_i8.  That turns the pass from instruction-bound (3.5 TB/s as the storing
// kernel without its stores) into a read of x.
struct PwStatGeom {
  int Cin, KTS, Cout, CT, HW;
  int64_t cols, tiles;
  int zoff;
  int table;                 // 1: per-sample maxima through the workgroup's LDS table (few tiles per wavefront)
  FastDiv hw;
};

template <int KT>
__global__ __launch_bounds__(kBlock, 3) void pw_stat_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, PwStatGeom g, const float* __restrict__ in_stat, int n,
    const float* __restrict__ in_thr, float levels, int lo_neg, float eps, float* __restrict__ cur_max_out,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act, float* __restrict__ stat_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pst_smem[];
  v4i* ldsW = reinterpret_cast<v4i*>(pst_smem);                                           // [CT][KT][64]
  int* c_zs = reinterpret_cast<int*>(pst_smem + (size_t)g.CT * KT * 1024);                 // [CT * 32]
  float* c_k = reinterpret_cast<float*>(c_zs + g.CT * 32);                                 // [4][CT * 32]: w scale, bias, BN scale / shift
  int* accs = reinterpret_cast<int*>(c_k + 4 * g.CT * 32);                                 // [4 waves][CT][2][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, i32 = lane & 31;
  const unsigned HW = (unsigned)g.HW, cols = (unsigned)g.cols;
  const int64_t nwaves = (int64_t)gridDim.x * 4, wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t t_begin = g.tiles * wid / nwaves, t_end = g.tiles * (wid + 1) / nwaves;
  const int pl = 16 * ((i32 >> 2) & 1) + 4 * (i32 >> 3) + (i32 & 3);      // tile pixel of this lane's MFMA row
  int* my = accs + (size_t)wave * g.CT * 128;
  constexpr int kSlots = 8;
  __shared__ unsigned k_stat[kSlots];
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;
  unsigned s_base;                               // first sample of the workgroup's tile range
  {
    const int64_t t0 = g.tiles * ((int64_t)blockIdx.x * 4) / nwaves;
    const unsigned j0 = (unsigned)(t0 < g.tiles ? t0 : g.tiles - 1) * 32u;
    s_base = fast_div(j0 < cols ? j0 : cols - 1, g.hw);
  }

  struct Pix { unsigned smp, p; };
  auto pix_of = [&](int64_t t) __attribute__((always_inline)) {
    unsigned j = (unsigned)t * 32u + (unsigned)pl;
    j = j < cols ? j : cols - 1;                                          // (a copy of the last pixel: changes no maximum)
    Pix r;
    r.smp = fast_div(j, g.hw);
    r.p = j - r.smp * HW;
    return r;
  };
  // slab q of the wave's range = (tile t_begin + q / KT, 32-channel slab q % KT), kept in buffer q % NB; the loads of slab
  // q + NB - 1 leave before slab q is quantised: NB - 1 slabs (4 KB each) in flight per wavefront, three wavefronts per SIMD.  One tile ahead - the first
  // version - left every wavefront waiting a memory latency per tile (3.9 TB/s on the 205 MB tensor of the first pair).
  constexpr int NB = 3;
  const fq_rsrc rs = make_rsrc(x, (int64_t)(cols / HW) * g.Cin * HW * 4);
  auto issue = [&](int64_t t, int kt, float (&v)[16]) __attribute__((always_inline)) {
    if (t >= t_end) return;
    const Pix px = pix_of(t);
    // (a ragged half-slab: channels past Cin are out of the buffer's range only for the LAST sample - clamp and zero instead)
    // (a half-slab past Cin: an offset out of the buffer's range - the loads return 0, whose code meets zero weight codes)
    const bool ok = kt * 32 + 16 * h < g.Cin;
    const unsigned vo = ok ? (unsigned)(((px.smp * (unsigned)g.Cin + 16u * h) * HW + px.p) * 4u) : 0x80000000u;
    const unsigned so = (unsigned)kt * 32u * HW * 4u;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = buf_ld_f32(rs, vo, so + (unsigned)i * HW * 4u);
  };
  float raw[NB][16];
#pragma unroll
  for (int b2 = 0; b2 < NB; ++b2)
#pragma unroll
    for (int i = 0; i < 16; ++i) raw[b2][i] = 0.0f;
#pragma unroll
  for (int qq = 0; qq < NB - 1; ++qq) issue(t_begin + qq / KT, qq % KT, raw[qq]);
  FQ_PIN();
  const float max_ = input_threshold(in_stat, n, in_thr, cur_max_out, blockIdx.x == 0);
  const QParams q = make_qparams(max_, levels, lo_neg != 0, eps);
  const int ubias = 128 - g.zoff;
  const unsigned nn_xor = fq_nonneg_xor(ubias);
  for (int idx = threadIdx.x; idx < g.CT * KT * 64; idx += kBlock) {
    const int f = idx >> 6, ct = f / KT, kt = f - ct * KT;
    ldsW[idx] = *reinterpret_cast<const v4i*>(wfrag + (((int64_t)ct * g.KTS + kt) << 10) + ((idx & 63) << 4));
  }
  // (the epilogue's per-channel constants too: read from HBM inside flush() they cost every wavefront a memory latency per
  // channel tile and sample - a fixed 6-25 us that made the pass as slow at batch 32 as at batch 128)
  const int nchs = g.CT * 32;
  for (int i = threadIdx.x; i < nchs; i += kBlock) {
    const bool okc = i < g.Cout;
    c_zs[i] = okc ? g.zoff * wsum[i] : 0;
    c_k[i] = okc ? wscale[i] : 0.0f;
    c_k[nchs + i] = (okc && bias != nullptr) ? bias[i] : 0.0f;
    c_k[2 * nchs + i] = (okc && bn_scale != nullptr) ? bn_scale[i] : 1.0f;
    c_k[3 * nchs + i] = (okc && bn_scale != nullptr) ? bn_shift[i] : 0.0f;
  }
  auto reset = [&]() __attribute__((always_inline)) {
    for (int ct = 0; ct < g.CT; ++ct) {
      my[ct * 128 + lane] = INT_MIN;
      my[ct * 128 + 64 + lane] = INT_MAX;
    }
  };
  reset();
  __syncthreads();
  const bool has_bn = bn_scale != nullptr;
  // the fp32 epilogue of fq_pwconv_i8 on the two extreme sums of every channel, then the maximum over the wave's channels
  auto flush = [&](unsigned smp) __attribute__((always_inline)) {
    float m = 0.0f;
    for (int ct = 0; ct < g.CT; ++ct) {
      const int c = ct * 32 + i32;
      int hi = my[ct * 128 + lane], lo = my[ct * 128 + 64 + lane];
      // (both halves hold the same channel: pixels 0..15 and 16..31)
      const int ohi = __shfl_xor(hi, 32, 64), olo = __shfl_xor(lo, 32, 64);
      hi = hi > ohi ? hi : ohi;
      lo = lo < olo ? lo : olo;
      if (c < g.Cout && hi >= lo) {
        hi += c_zs[c];
        lo += c_zs[c];
        const float sxw = q.scale * c_k[c];
        const float bch = c_k[nchs + c];
        const float bsc = c_k[2 * nchs + c], bsh = c_k[3 * nchs + c];
        auto f = [&](int a) {
          float v = (float)a * sxw;
          if (bias != nullptr) v = v + bch;
          if (has_bn) {
            v = v * bsc;
            v = v + bsh;
          }
          return fabsf(act_rt(v, act));
        };
        m = fmaxf(m, fmaxf(f(hi), f(lo)));
      }
    }
    m = wave_max_nonneg(m);
    // into the workgroup's table; ONE global atomic per (workgroup, sample) at the end.  (A first version issued one per
    // wavefront: 3072 atomics on the n floats of stat_out - a single cache line at batch 32 - serialise in L2 at ~10 ns each,
    // a fixed 30 us that made the pass as slow at batch 32 as at batch 128: profiles/r6_pwstat_atomics.txt)
    // Where the maximum goes.  Few tiles per wavefront (small batches): into the workgroup's LDS table, ONE global atomic per
    // (workgroup, sample) at the end - one atomic per wavefront means thousands of same-cache-line atomics (stat_out is n
    // floats: one line at batch 32) that serialise in L2 at ~10 ns each, a fixed 30 us.  Many tiles per wavefront: straight to
    // memory - the atomics of a long kernel are spread over its run time and overlap with the other wavefronts' work, while
    // the table's flush would put them all at the end (40 against 47 us on the first pair at batch 128).
    // profiles/r6_pwstat_atomics.txt
    if (lane == 0 && __float_as_uint(m) != 0u) {
      const unsigned slot = smp - s_base;
      if (g.table && slot < (unsigned)kSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
      else FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + smp, __float_as_uint(m));
    }
    reset();
  };
  v4i afrag[KT];
  const FastQuot fqx = make_fast_quot(q.denom);
  auto quant = [&](int kt, const float (&v)[16], auto nn_c) __attribute__((always_inline)) {
    v4i f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if constexpr (decltype(nn_c)::value) {     // clip range from 0 and a divisor that takes the fp32 quotient (fq_common.h)
        unsigned u = 0u;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          int r;
          const float Q = fast_quot(fq_clip(v[4 * d + b], q), fqx);
          asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(Q));
          u |= (unsigned)r << (8 * b);
        }
        f[d] = (int)(u ^ nn_xor);
      } else {
        f[d] = fq_pack4<false>(v[4 * d + 0], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3], q, ubias, nn_xor);
      }
    }
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    afrag[kt] = f;
  };
  // pixels [plo, phi) of the tile (in tile order: this lane's registers are pixels 16 h + r) join the open sample
  auto reduce_tile = [&](int plo, int phi) __attribute__((always_inline)) {
    const bool whole = plo <= 0 && phi >= 32;
#pragma unroll 1
    for (int ct = 0; ct < g.CT; ++ct) {
      v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // (zoff * rowsum joins at the flush: max / min commute with it)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[kt], ldsW[((ct * KT + kt) << 6) + lane], acc, 0, 0, 0);
      int hi = INT_MIN, lo = INT_MAX;
      if (whole) {
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          hi = max(max(acc[k], acc[k + 1]), hi);
          lo = min(min(acc[k], acc[k + 1]), lo);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int p = 16 * h + k;
          const bool in = p >= plo && p < phi;
          hi = in && acc[k] > hi ? acc[k] : hi;
          lo = in && acc[k] < lo ? acc[k] : lo;
        }
      }
      atomicMax(&my[ct * 128 + lane], hi);
      atomicMin(&my[ct * 128 + 64 + lane], lo);
    }
  };
  unsigned cur = fast_div((unsigned)((t_begin < g.tiles ? t_begin : g.tiles - 1) * 32), g.hw);
  auto run_tile = [&](int64_t t, auto ph_c, auto nn_c) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value;                 // the tile's first slab sits in buffer (PH * KT) % NB
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int qn = kt + NB - 1;
      issue(t + qn / KT, qn % KT, raw[(PH * KT + kt + NB - 1) % NB]);
      FQ_PIN();
#if defined(FQ_PST_ABLATE) && FQ_PST_ABLATE == 1       // ablation build: the loads alone (values folded so that they stay alive)
      {
        float (&mine)[16] = raw[(PH * KT + kt) % NB];
        float mm = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) mm = fmaxf(mm, mine[i]);
        afrag[kt] = (v4i){__float_as_int(mm), 0, 0, 0};
      }
#else
      quant(kt, raw[(PH * KT + kt) % NB], nn_c);
#endif
      FQ_PIN();
    }
#if defined(FQ_PST_ABLATE)                             // ... 2: loads + quantiser, no matrix cores, no reduction
    {
      int keep = 0;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) keep |= afrag[kt][0] | afrag[kt][1] | afrag[kt][2] | afrag[kt][3];
      if (keep == 0x7fffffff) my[lane] = keep;
      return;
    }
#endif
    const unsigned j0 = (unsigned)t * 32u, j1 = j0 + 31u < cols ? j0 + 31u : cols - 1;
    const unsigned s0 = fast_div(j0, g.hw), s1 = fast_div(j1, g.hw);
    if (s0 != cur) {
      flush(cur);
      cur = s0;
    }
    if (s0 == s1) {
      reduce_tile(0, 32);
    } else {                                   // a tile across two samples (planes of at least 32 pixels: never three)
      const int pb = (int)(s1 * HW - j0);
      reduce_tile(0, pb);
      flush(cur);
      cur = s1;
      reduce_tile(pb, 32);
    }
    FQ_PIN();
  };
  auto run_all = [&](auto nn_c) __attribute__((always_inline)) {
    constexpr int U = (KT % NB == 0) ? 1 : NB;                  // tiles per turn: U * KT is a multiple of NB (NB prime)
    using std::integral_constant;
    for (int64_t t = t_begin; t < t_end; t += U) {
      run_tile(t, integral_constant<int, 0>{}, nn_c);
      if (U > 1 && t + 1 < t_end) run_tile(t + 1, integral_constant<int, 1 % U>{}, nn_c);
      if (U > 2 && t + 2 < t_end) run_tile(t + 2, integral_constant<int, 2 % U>{}, nn_c);
      if (U > 3 && t + 3 < t_end) run_tile(t + 3, integral_constant<int, 3 % U>{}, nn_c);
    }
  };
  if (fq_nonneg(q) && fqx.ok) run_all(std::true_type{});
  else run_all(std::false_type{});
  if (t_begin < t_end) flush(cur);
  __syncthreads();
  if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < cols / HW)
    FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
}

// test hook (fq_debug_fast_quotient): the quotient the recompute kernels divide with, element by element
__global__ void fast_quotient_kernel(const float* __restrict__ c, int64_t n, const float* __restrict__ d, float* __restrict__ out,
                                     int* __restrict__ ok_out) {
  const FastQuot f = make_fast_quot(d[0]);
  const double rden = 1.0 / (double)d[0];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) ok_out[0] = f.ok ? 1 : 0;
  if (i < n) out[i] = f.ok ? fast_quot(c[i], f) : ieee_div_by(c[i], rden);
}

struct PwDwPlan {
  int kt, cw, lz, strips, ctg, bands, rb, pitch, threads;
  size_t lds;
  bool ok;
};

PwDwPlan pwdw_plan(int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, int stride) {
  PwDwPlan p;
  memset(&p, 0, sizeof(p));
  if (n <= 0 || cin <= 0 || cout <= 0 || h < 3 || w < 8 || (stride != 1 && stride != 2)) return p;
  if (cout % 32 != 0 || (stride == 2 && (w % 2 != 0 || h % 2 != 0))) return p;
  const int kt = (int)((cin + 31) / 32), ct = (int)(cout / 32);
  if (!(kt == 1 || kt == 2 || kt == 4 || kt == 8)) return p;
  const int wo = (int)((w - 1) / stride + 1), ho = (int)((h - 1) / stride + 1);
  p.kt = kt;
  p.cw = ct % 2 == 0 ? 2 : 1;
  p.strips = (int)((w + 29) / 30);
  p.lz = (stride == 1 && p.strips == 1) ? (int)(30 - w) : 0;
  if (!(p.lz == 0 || p.lz == 2)) return p;                  // (instantiated: planes from 30 columns up, and 28)
  p.ctg = ct / p.cw;
  p.threads = 64 * p.strips * p.ctg;
  if (p.threads > 512) return p;
  // row-buffer pitch: the widest column any lane writes (invalid outputs past the plane included) + the columns left of it
  const int reach = stride == 1 ? (int)w + 2 + p.lz : 15 * p.strips + 1;
  p.pitch = (reach > wo ? reach : wo) + p.lz;
  p.pitch |= 1;
  p.lds = (size_t)ct * kt * 1024 + 2 * ((size_t)cout * p.pitch + kRowSlack) * 4 + kRowSlack * 4 + (stride == 1 ? (size_t)cout * 13 * 4 : 0);
  if (p.lds + 10 * 1024 > (size_t)max_lds_bytes()) return p;
  // rows per band: the bands of all samples should fill the chip's workgroup slots (two wavefronts per SIMD by registers, the
  // row buffers by LDS) in as few rounds as possible, a band re-computing the one or two pointwise rows above / below it.
  // (MobileNet1.0 at batch 128: 14 rows per band on the 56-row outputs - 512 workgroups - and 7 on the 28-row ones; measured
  // with 14 / 10 / 7: profiles/r6_pwdw_bands.txt)
  static const int rb_env = env_int("FQ_PWDW_RB", 0);
  {
    const int waves = p.threads / 64;
    int occ = (int)((160 * 1024) / (p.lds + 512));
    const int by_regs = 8 / waves > 0 ? 8 / waves : 1;
    occ = occ < by_regs ? occ : by_regs;
    occ = occ < 1 ? 1 : occ;
    const double slots = (double)num_cu() * occ;
    double best = 1e30;
    int best_rb = ho;
    for (int b = 1; b <= ho; ++b) {
      const int rb = (ho + b - 1) / b, bands = (ho + rb - 1) / rb;
      const double wgs = (double)n * bands, rounds = ceil(wgs / slots), fill = wgs < slots ? wgs / slots : 1.0;
      const double cost = rounds * (stride * rb + 2) / fill;
      if (cost < best - 1e-9) {
        best = cost;
        best_rb = rb;
      }
    }
    p.rb = rb_env > 0 ? (rb_env < ho ? rb_env : ho) : best_rb;
  }
  p.bands = (ho + p.rb - 1) / p.rb;
  if (n * cin * h * w * 4 >= (1ll << 32) || n * cout * ho * wo >= (1ll << 31)) return p;
  p.ok = true;
  return p;
}

}  // namespace

namespace fqi {

bool pw_stat_shape_ok(int64_t n, int64_t cin, int64_t cout, int64_t hw) {
  if (n <= 0 || cin <= 0 || cout <= 0 || hw < 32 || hw >= (1ll << 30) || n * hw >= (1ll << 31) - 512) return false;
  const int kt = (int)((cin + 31) / 32), ct = (int)((cout + 31) / 32);
  if (!(kt == 1 || kt == 2 || kt == 4 || kt == 8) || cin % 16 != 0) return false;
  const size_t lds = (size_t)ct * kt * 1024 + 5 * (size_t)ct * 32 * 4 + 4 * (size_t)ct * 128 * 4;
  return lds <= 120 * 1024 && n * cin * hw * 4 < (1ll << 32);
}

int pw_stat_launch(const PwCall& c) {
  const int kt = (int)((c.cin + 31) / 32), ct = (int)((c.cout + 31) / 32);
  const size_t lds = (size_t)ct * kt * 1024 + 5 * (size_t)ct * 32 * 4 + 4 * (size_t)ct * 128 * 4;
  PwStatGeom g;
  g.Cin = (int)c.cin; g.KTS = (int)(c.cin_pad / 32); g.Cout = (int)c.cout; g.CT = ct; g.HW = (int)c.hw;
  g.cols = c.n * c.hw; g.tiles = (g.cols + 31) / 32; g.zoff = c.zoff;
  g.hw = fast_div_for((unsigned)c.hw);
  if (int rc = pw_zero_stat(c)) return rc;
  int per_cu = (int)((160 * 1024) / (lds + 1024));
  per_cu = per_cu > 3 ? 3 : per_cu;
  static const int wg_env = env_int("FQ_PWSTAT_WG_PER_CU", 0);
  if (wg_env > 0) per_cu = wg_env;
  int64_t grid = (int64_t)num_cu() * per_cu;
  const int64_t need = (g.tiles + 3) / 4;
  if (grid > need) grid = need;
  static const int table_env = env_int("FQ_PWSTAT_TABLE", -1);
  g.table = table_env >= 0 ? table_env : (g.tiles < grid * 4 * 10 ? 1 : 0);
#define FQ_PST_GO(KT_)                                                                                                     \
  case KT_: {                                                                                                              \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_stat_kernel<KT_>),                   \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) == hipSuccess; \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8_stat: cannot raise the dynamic LDS limit");                                          \
    hipLaunchKernelGGL((pw_stat_kernel<KT_>), dim3((unsigned)grid), dim3(kBlock), lds, c.st, c.x, c.wcodes, c.wscale,       \
                       (const int*)c.wsum, c.bias, g, c.in_stat, (int)c.n, c.in_thr, c.levels, c.lo_neg, kEps,              \
                       c.out_current_max, c.bn_scale, c.bn_shift, c.act, c.stat_out);                                      \
  } break;
  switch (kt) {
    FQ_PST_GO(1) FQ_PST_GO(2) FQ_PST_GO(4) FQ_PST_GO(8)
    default: return fail(FQ_ERR_INVALID, "fq_pwconv_i8_stat: K / 32 = %d is not instantiated", kt);
  }
#undef FQ_PST_GO
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // namespace fqi

using namespace fqi;

extern "C" {

int fq_debug_fast_quotient(const float* c, int64_t n, const float* d, float* out, int* took_fast_path, fqStream_t stream) {
  FQ_REQUIRE(c && d && out && took_fast_path && n > 0, "fq_debug_fast_quotient: null pointer");
  hipLaunchKernelGGL(fast_quotient_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c, n, d, out,
                     took_fast_path);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_pwdw_fused_supported(int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, int stride) {
  return pwdw_plan(n, cin, cout, h, w, stride).ok ? 1 : 0;
}

int fq_pwdw_fused(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* pw_bias,
                  int64_t n, int64_t cin, int64_t cin_pad, int64_t cout_pad, int64_t cout, int64_t h, int64_t w,
                  const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, const float* pw_bn_scale,
                  const float* pw_bn_shift, int pw_act, const float* mid_stat, const float* mid_thr, int mid_width,
                  unsigned mid_flags, float* mid_current_max, const float* dw_w, const float* dw_bias, int dw_stride,
                  const float* dw_bn_scale, const float* dw_bn_shift, int dw_act, float* y, float* stat_out,
                  fqStream_t stream) {
  FQ_REQUIRE(x && wcodes && wscale && wsum && dw_w && y, "fq_pwdw_fused: null pointer");
  FQ_REQUIRE(in_stat != nullptr || in_thr != nullptr, "fq_pwdw_fused: give in_stat (online) or in_thr (offline)");
  FQ_REQUIRE(mid_stat != nullptr || mid_thr != nullptr, "fq_pwdw_fused: give mid_stat (the per-sample maxima of the pointwise "
             "output, from fq_pwconv_i8_stat) or mid_thr (a stored threshold)");
  FQ_REQUIRE(in_width >= 2 && in_width <= 8, "fq_pwdw_fused: input width %d does not fit int8 codes", in_width);
  FQ_REQUIRE(mid_width >= 2 && mid_width <= 16, "fq_pwdw_fused: bad width %d of the depthwise input", mid_width);
  FQ_REQUIRE(!((in_flags | mid_flags) & (FQ_ACT_NO_ABS | FQ_ACT_NO_EPS)), "fq_pwdw_fused: unsupported activation flags");
  FQ_REQUIRE((pw_bn_scale == nullptr) == (pw_bn_shift == nullptr) && (dw_bn_scale == nullptr) == (dw_bn_shift == nullptr),
             "fq_pwdw_fused: a BatchNorm's scale and shift go together");
  FQ_REQUIRE(cin_pad >= cin && cin_pad % 64 == 0 && cout_pad >= cout && cout_pad % 32 == 0,
             "fq_pwdw_fused: cin_pad / cout_pad must be those of fq_weight_codes");
  const bool prezeroed = (dw_act & FQ_STAT_PREZEROED) != 0;
  dw_act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(pw_act >= FQ_ACT_NONE && pw_act <= FQ_ACT_RELU6 && dw_act >= FQ_ACT_NONE && dw_act <= FQ_ACT_RELU6,
             "fq_pwdw_fused: unknown activation");
  FQ_REQUIRE(aligned16(wcodes) && aligned16(x) && aligned16(y), "fq_pwdw_fused: x, wcodes and y must be 16-byte aligned");
  const PwDwPlan p = pwdw_plan(n, cin, cout, h, w, dw_stride);
  FQ_REQUIRE(p.ok, "fq_pwdw_fused: shape not taken (n=%lld cin=%lld cout=%lld %lldx%lld stride %d): see fq_pwdw_fused_supported",
             (long long)n, (long long)cin, (long long)cout, (long long)h, (long long)w, dw_stride);
  hipStream_t st = (hipStream_t)stream;
  const int64_t ho = (h - 1) / dw_stride + 1, wo = (w - 1) / dw_stride + 1;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  PwDwGeom g;
  g.Cin = (int)cin; g.KTS = (int)(cin_pad / 32); g.Cout = (int)cout; g.H = (int)h; g.W = (int)w; g.Ho = (int)ho; g.Wo = (int)wo;
  g.strips = p.strips; g.ctg = p.ctg; g.bands = p.bands; g.RB = p.rb;
  g.zoff = (in_flags & FQ_ACT_SIGNED) ? 0 : 128;
  g.pitch = p.pitch;
  g.vw = wo % 4 == 0 ? 4 : (wo % 2 == 0 ? 2 : 1);
  g.qv = fast_div_for((unsigned)(wo / g.vw));
  const int8_t* wfrag = wcodes + cout_pad * cin_pad;
  const float levels1 = act_levels(in_width, in_flags), levels2 = act_levels(mid_width, mid_flags);
  const int lo1 = (in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0, lo2 = (mid_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
  // the fused-inference case as a compile-time epilogue
  int fast = 0;
  if (!pw_bias && !dw_bias && pw_bn_scale && dw_bn_scale && !lo1 && !lo2 && pw_act == dw_act &&
      (pw_act == FQ_ACT_RELU || pw_act == FQ_ACT_RELU6))
    fast = 1;
  static const int no_fast = env_int("FQ_PWDW_NO_FAST", 0);
  if (no_fast) fast = 0;
  // algorithmic bytes: those of the depthwise layer this launch stands for (its input is never written: SURVEY.md 8d counts
  // what the layer's arithmetic needs); moved: what the launch really reads and writes
  const double in_elems = (double)n * cin * h * w, mid_elems = (double)n * cout * h * w, out_elems = (double)n * cout * ho * wo;
  ProfScope prof(FQ_KERNEL_DWCONV, 4.0 * (mid_elems + out_elems), st, 4.0 * (in_elems + out_elems));
  const unsigned grid = (unsigned)(n * p.bands);
// (-DFQ_PWDW_DEV: a tuning build with two instantiations - the first two pairs of MobileNet1.0 - instead of 48)
#ifdef FQ_PWDW_DEV
#define FQ_PWDW_INST(KT_, CW_, S_, LZ_, F_) ((F_) == 1 && (LZ_) == 0 && (CW_) == 2 && (((KT_) == 1 && (S_) == 2) || ((KT_) == 2 && (S_) == 1)))
#else
#define FQ_PWDW_INST(KT_, CW_, S_, LZ_, F_) true
#endif
#define FQ_PWDW_GO(KT_, CW_, S_, LZ_, F_)                                                                                   \
  if constexpr (!FQ_PWDW_INST(KT_, CW_, S_, LZ_, F_)) {                                                                    \
    return fail(FQ_ERR_INVALID, "fq_pwdw_fused: not instantiated in this (FQ_PWDW_DEV) build");                            \
  } else {                                                                                                                 \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&pwdw_kernel<KT_, CW_, S_, LZ_, F_>),    \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess; \
    FQ_REQUIRE(attr_ok, "fq_pwdw_fused: cannot raise the dynamic LDS limit");                                              \
    hipLaunchKernelGGL((pwdw_kernel<KT_, CW_, S_, LZ_, F_>), dim3(grid), dim3((unsigned)p.threads), p.lds, st, x, wfrag,    \
                       wscale, (const int*)wsum, pw_bias, g, in_stat, (int)n, in_thr, levels1, lo1, kEps, pw_bn_scale,      \
                       pw_bn_shift, pw_act, mid_stat, mid_thr, levels2, lo2, mid_current_max, dw_w, dw_bias, dw_bn_scale,   \
                       dw_bn_shift, dw_act, y, stat_out);                                                                  \
  }
#define FQ_PWDW_F(KT_, CW_, S_, LZ_)                                                                                        \
  {                                                                                                                        \
    if (fast == 1) FQ_PWDW_GO(KT_, CW_, S_, LZ_, 1) else FQ_PWDW_GO(KT_, CW_, S_, LZ_, 0)                                   \
  }
#define FQ_PWDW_S(KT_, CW_)                                                                                                 \
  {                                                                                                                        \
    if (dw_stride == 2) FQ_PWDW_F(KT_, CW_, 2, 0)                                                                          \
    else if (p.lz == 0) FQ_PWDW_F(KT_, CW_, 1, 0) else FQ_PWDW_F(KT_, CW_, 1, 2)                                            \
  }
#define FQ_PWDW_K(KT_)                                                                                                      \
  case KT_:                                                                                                                \
    if (p.cw == 2) FQ_PWDW_S(KT_, 2) else FQ_PWDW_S(KT_, 1)                                                                \
    break;
  switch (p.kt) {
    FQ_PWDW_K(1) FQ_PWDW_K(2) FQ_PWDW_K(4) FQ_PWDW_K(8)
    default: break;
  }
#undef FQ_PWDW_K
#undef FQ_PWDW_S
#undef FQ_PWDW_F
#undef FQ_PWDW_GO
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // extern "C"
