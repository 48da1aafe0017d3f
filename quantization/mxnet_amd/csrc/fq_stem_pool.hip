// libfakequant — K2u first convolution 7x7 stride 2 (3 -> 64) + BatchNorm + activation + MaxPool 3x3 stride 2 in ONE kernel
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_common.h"

namespace {

// K2u (round 5): the head of the ImageNet ResNets - Conv2D(3 -> 64, 7x7, stride 2, pad 3) -> BatchNorm -> ReLU ->
// MaxPool2D(3, 2, 1) - used to be two launches: `stem_mfma_kernel<7, 64>` (fq_stem.hip) wrote the 411 MB convolution output
// (batch 128), `bn_act_maxpool_stat_kernel` read it back and wrote the 103 MB pooled tensor (296 + 114 us; the stem at 0.21 of
// the HBM roofline because its fp32 MFMA chain, 193 us, is what bounds it - profiles/r4_kprof_resnet50_offline.txt).  Here the
// convolution output never leaves the CU: a workgroup of EIGHT wavefronts (one per CU) walks down a band of pooled rows of
// one image; per pooled row p
//   conv   wavefront w computes quarter (w & 3) of convolution row 2 p + (w >> 2): a tile of <= 32 pixels x 64 channels by the
//          same fp32 MFMA chain as K2q (v_mfma_f32_32x32x2_f32, k ascending over (ci, ky, kx): bit-identical to the fmaf
//          chain of the oracle), inputs gathered straight from NCHW with one 4-byte buffer load per lane and step, BatchNorm /
//          activation on the accumulators, values written into a ring of THREE convolution rows in LDS ([row][channel][col]);
//   pool   after a barrier all 512 threads take the 64 x Wp outputs of pooled row p: max over the 3 x 3 window read from
//          the ring (rows 2 p - 1 .. 2 p + 1: row 2 p - 1 is still there from the previous step; a band's first row is
//          computed once more as a halo), store, per-sample max|y| of the POOLED tensor;
// two barriers per pooled row.  Eight tiles per step on eight wavefronts keep the four SIMDs evenly loaded (the 112-pixel rows
// are 3.5 tiles of 32: quarter rows of 28 pixels waste the same 1/8 of the matrix pipe and need no tile to straddle rows).
// The gather of the NEXT step's first chunk is requested before the pooling phase, as K2q does between tiles.
// Traffic: 77 MB in + 103 MB out instead of 77 + 411 + 411 + 103.
constexpr int kKS = 7, kCout = 64, kCT = 2, kPad = 3;
constexpr int kK = 3 * kKS * kKS, kNS = (kK + 1) / 2;                   // 147 taps, 74 MFMA steps of k = 2
constexpr int kCH = 16;                                                 // steps whose loads are in flight together
constexpr int kNW = 8;

typedef float v16f __attribute__((ext_vector_type(16)));

struct StemPoolGeom {
  int H, W, Ho, Wo, Hp, Wp;  // input, convolution output, pooled output
  int QW;                    // columns of a quarter row (ceil(Wo / 4) <= 32)
  int nbands, rows_per_band; // bands of pooled rows per image
  int n;
};

template <int EPI>
__global__ __launch_bounds__(512, 2) void stem7_pool_kernel(const float* __restrict__ x, const float* __restrict__ wt /*[3][7][7][64]*/,
                                                            const float* __restrict__ bias, float* __restrict__ y, StemPoolGeom g,
                                                            const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                            int act, float* __restrict__ stat_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
  float* const wl = reinterpret_cast<float*>(sp_smem);                  // [k][co], zero row for the padded k
  float* const c_bias = wl + kNS * 2 * kCout;
  float* const c_bsc = c_bias + kCout;
  float* const c_bsh = c_bsc + kCout;
  float* const red = c_bsh + kCout;                                     // 8 floats
  float* const ring = red + 16;                                         // [3][64][Wo]
  const int H = g.H, W = g.W, Ho = g.Ho, Wo = g.Wo, Wp = g.Wp;
  for (int i = threadIdx.x; i < kNS * 2 * kCout; i += 512) wl[i] = i < kK * kCout ? wt[i] : 0.0f;
  for (int i = threadIdx.x; i < kCout; i += 512) {
    c_bias[i] = bias != nullptr ? bias[i] : 0.0f;
    c_bsc[i] = bn_scale != nullptr ? bn_scale[i] : 1.0f;
    c_bsh[i] = bn_scale != nullptr ? bn_shift[i] : 0.0f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pl = lane & 31;
  const bool has_bn = bn_scale != nullptr;
  const int smp = (int)blockIdx.x / g.nbands, band = (int)blockIdx.x - smp * g.nbands;
  const int p_begin = band * g.rows_per_band;
  const int p_end = p_begin + g.rows_per_band < g.Hp ? p_begin + g.rows_per_band : g.Hp;
  const unsigned x_img = (unsigned)(3 * H * W) * 4u, W4 = (unsigned)W * 4u, HW4 = (unsigned)(H * W) * 4u;
  const fq_rsrc xr = make_rsrc(reinterpret_cast<const char*>(x) + (int64_t)smp * x_img, x_img);
  const int ring_row = kCout * Wo;                                      // floats of one ring row

  // this wavefront's tile of a step: convolution row `oy`, columns qw0 .. qw0 + QW - 1 (lane pl: column qw0 + pl)
  const int quarter = wave & 3, rsel = wave >> 2;
  const int ox = quarter * g.QW + pl;
  const bool col_ok = pl < g.QW && ox < Wo;
  unsigned xm = 0;
  const int ix0 = 2 * ox - kPad;
#pragma unroll
  for (int k = 0; k < kKS; ++k) xm |= (col_ok && ix0 + k >= 0 && ix0 + k < W) ? (1u << k) : 0u;
  struct Row { int pixoff; unsigned ym; };
  auto row_at = [&](int oy) __attribute__((always_inline)) {           // oy out of range: nothing is loaded
    Row r;
    const int iy0 = 2 * oy - kPad;
    r.pixoff = (iy0 * W + ix0) * 4;
    unsigned ym = 0;
#pragma unroll
    for (int k = 0; k < kKS; ++k) ym |= (oy >= 0 && oy < Ho && iy0 + k >= 0 && iy0 + k < H) ? (1u << k) : 0u;
    r.ym = ym;
    return r;
  };
  // the lane's input value of step s: tap k = 2 s + h, i.e. (ci, ky, kx); an invalid tap (outside the image, or the padded
  // k) gets an offset the resource bounds out
  auto issue = [&](const Row& rw, int s) __attribute__((always_inline)) {
    const int k0 = 2 * s, k1 = 2 * s + 1;
    const int ky0 = (k0 / kKS) % kKS, kx0 = k0 % kKS, ci0 = k0 / (kKS * kKS);
    const int ky1 = k1 < kK ? (k1 / kKS) % kKS : 0, kx1 = k1 < kK ? k1 % kKS : 0, ci1 = k1 < kK ? k1 / (kKS * kKS) : 0;
    const unsigned t0 = (unsigned)ci0 * HW4 + (unsigned)ky0 * W4 + (unsigned)kx0 * 4u;
    const unsigned t1 = (unsigned)ci1 * HW4 + (unsigned)ky1 * W4 + (unsigned)kx1 * 4u;
    const bool v0 = ((rw.ym >> ky0) & (xm >> kx0) & 1u) != 0u;
    const bool v1 = k1 < kK && ((rw.ym >> ky1) & (xm >> kx1) & 1u) != 0u;
    const bool v = h ? v1 : v0;
    const unsigned off = (unsigned)rw.pixoff + (h ? t1 : t0);
    return buf_ld_f32(xr, v ? off : 0x80000000u, 0u);
  };
  float bbuf[2][kCH];
  // one tile: convolution row oy (its first chunk of loads already in bbuf[0]) -> ring slot oy % 3; `next`: the row whose first
  // chunk is requested behind this tile's last one
  auto conv_tile = [&](int oy, const Row& cur, const Row& next) __attribute__((always_inline)) {
    v16f acc[kCT];
#pragma unroll
    for (int c = 0; c < kCT; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.0f;
    constexpr int NCHUNK = (kNS + kCH - 1) / kCH;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      float (&mine)[kCH] = bbuf[c & 1];
      float (&other)[kCH] = bbuf[(c + 1) & 1];
      if (c + 1 < NCHUNK) {
#pragma unroll
        for (int i = 0; i < kCH; ++i)
          if ((c + 1) * kCH + i < kNS) other[i] = issue(cur, (c + 1) * kCH + i);
      } else {
#pragma unroll
        for (int i = 0; i < kCH; ++i) other[i] = issue(next, i);
      }
      FQ_PIN();
#pragma unroll
      for (int i = 0; i < kCH; ++i) {
        const int s = c * kCH + i;
        if (s < kNS) {
#pragma unroll
          for (int ct = 0; ct < kCT; ++ct)
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(2 * s + h) * kCout + ct * 32 + pl], mine[i], acc[ct], 0, 0, 0);
          if ((i & 3) == 3) FQ_PIN();            // four steps' weight reads at a time (else all are hoisted: registers)
        }
      }
      FQ_PIN();
    }
    if (NCHUNK % 2 == 1) {                       // the next tile's first chunk must sit in bbuf[0]
#pragma unroll
      for (int i = 0; i < kCH; ++i) bbuf[0][i] = bbuf[1][i];
    }
    // epilogue: lane = column, register = channel 8 gq + 4 h + r -> ring[oy % 3][channel][column]
    if (oy >= 0 && oy < Ho) {
      float* const dst = ring + (oy % 3) * ring_row + ox;
#pragma unroll
      for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int c0 = ct * 32 + 8 * gq + 4 * h;
          const f4 bch = *reinterpret_cast<const f4*>(c_bias + c0);
          const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
          const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = dw_finish<EPI>(acc[ct][4 * gq + r], bias != nullptr, bch[r], has_bn, bsc[r], bsh[r], act);
            if (col_ok) dst[(c0 + r) * Wo] = v;
          }
        }
    }
  };

  // ---- the band's halo row 2 p_begin - 1 (wavefronts 0..3; none for the image's first band: the pool's padding) -------------
  float m = 0.0f;
  {
    const int oy_h = 2 * p_begin - 1;
    const Row rh = row_at(rsel == 0 ? oy_h : -1);                       // (wavefronts 4..7 and oy = -1: no loads, no stores)
    const Row r0 = row_at(2 * p_begin + rsel);
#pragma unroll
    for (int i = 0; i < kCH; ++i) bbuf[0][i] = issue(p_begin > 0 ? rh : r0, i);
    if (p_begin > 0) {                                                  // (uniform)
      conv_tile(rsel == 0 ? oy_h : -1, rh, r0);
    }
  }
  // this thread's pooled outputs (channel, column), the same in every step: idx = t + 512 k < 64 * Wp
  constexpr int kPoolPer = (kCout * 64 + 511) / 512;                    // Wp <= 64
  const int total = kCout * Wp;
  int po_off[kPoolPer];                                                 // ch * Wo + 2 px | (left tap present) << 20 | (right tap present) << 21; < 0: none
  int po_out[kPoolPer];                                                 // ch * Hp * Wp + px
#pragma unroll
  for (int k = 0; k < kPoolPer; ++k) {
    const int idx = (int)threadIdx.x + 512 * k;
    const int ch = idx / Wp, px = idx - ch * Wp;
    const bool ok = idx < total;
    po_off[k] = ok ? ((ch * Wo + 2 * px) | ((px > 0 ? 1 : 0) << 20) | ((2 * px + 1 < Wo ? 1 : 0) << 21)) : -1;
    po_out[k] = ch * g.Hp * Wp + px;
  }
  float* const yo = y + ((int64_t)smp * kCout * g.Hp) * Wp;
  // rows of a multiple of eight convolution columns (224 x 224 images: 112): FOUR pooled outputs per task - the nine columns
  // 2 px - 1 .. 2 px + 7 of a ring row are one 4-byte and two 16-byte LDS reads instead of twelve 4-byte ones, the outputs one
  // 16-byte store (round 6: the pooling phase, in which the matrix pipe idles, was 63 LDS reads per thread and pooled row)
  const bool pool4 = (Wo & 7) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0;
  const int wq = Wp >> 2, total4 = kCout * wq;                          // tasks (channel, group of four pooled columns)
  constexpr int kPool4Per = (kCout * 16 + 511) / 512;                   // Wp <= 64: <= 16 groups per channel
  int p4_off[kPool4Per], p4_out[kPool4Per];                             // ch * Wo + 8 g4 | (left tap present) << 20; < 0: none
#pragma unroll
  for (int k = 0; k < kPool4Per; ++k) {
    const int idx = (int)threadIdx.x + 512 * k;
    const int ch = idx / (wq > 0 ? wq : 1), g4 = idx - ch * wq;
    p4_off[k] = (pool4 && idx < total4) ? ((ch * Wo + 8 * g4) | ((g4 > 0 ? 1 : 0) << 20)) : -1;
    p4_out[k] = ch * g.Hp * Wp + 4 * g4;
  }
  for (int p = p_begin; p < p_end; ++p) {
    const int oy = 2 * p + rsel;
    const Row cur = row_at(oy), nxt = row_at(oy + 2);
    conv_tile(oy, cur, p + 1 < p_end ? nxt : row_at(-1));
    __syncthreads();                                                    // rows 2 p, 2 p + 1 are in the ring (2 p - 1 since the last step)
    // ---- pool: outputs (channel, column) of pooled row p: all nine taps requested at once (clamped addresses; a tap outside
    // the plane repeats one inside the window - or, for the rows, is masked: row 2 p - 1 of the first pooled row does not exist)
    const int r_hi = 2 * p + 1 < Ho ? 2 * p + 1 : Ho - 1;
    const bool top = p > 0;
    const float* const row0 = ring + ((2 * p + 2) % 3) * ring_row;      // row 2 p - 1 (slot (2 p - 1) mod 3)
    const float* const row1 = ring + ((2 * p) % 3) * ring_row;
    const float* const row2 = ring + (r_hi % 3) * ring_row;
    if (pool4) {                                                        // (uniform)
#pragma unroll
      for (int k = 0; k < kPool4Per; ++k) {
        if (p4_off[k] < 0) continue;
        const int base = p4_off[k] & 0xFFFFF, c_lo = (p4_off[k] >> 20) & 1;
        f4 hm[3];                                                       // per ring row: the four horizontal maxima
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float* src = (r == 0 ? row0 : r == 1 ? row1 : row2) + base;
          const float lo = src[-c_lo];
          const f4 a = *reinterpret_cast<const f4*>(src), b = *reinterpret_cast<const f4*>(src + 4);
          // (the same association as the scalar form: max(max(left, middle), right))
          hm[r] = (f4){fmaxf(fmaxf(lo, a.x), a.y), fmaxf(fmaxf(a.y, a.z), a.w), fmaxf(fmaxf(a.w, b.x), b.y),
                       fmaxf(fmaxf(b.y, b.z), b.w)};
        }
        f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float m0 = fmaxf(hm[1][j], hm[2][j]);
          m0 = top ? fmaxf(m0, hm[0][j]) : m0;
          o[j] = m0;
          m = fmaxf(m, fabsf(m0));
        }
        *reinterpret_cast<f4*>(yo + p4_out[k] + (int64_t)p * Wp) = o;
      }
    } else {
#pragma unroll
    for (int k = 0; k < kPoolPer; ++k) {
      if (po_off[k] < 0) continue;
      const int base = po_off[k] & 0xFFFFF, c_lo = (po_off[k] >> 20) & 1, c_hi = (po_off[k] >> 21) & 1;   // ch * Wo + 2 px, taps present
      float v[9];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float* src = (r == 0 ? row0 : r == 1 ? row1 : row2) + base;
        v[3 * r + 0] = src[-c_lo];
        v[3 * r + 1] = src[0];
        v[3 * r + 2] = src[c_hi];
      }
      float m0 = fmaxf(fmaxf(v[3], v[4]), v[5]);
      const float m2 = fmaxf(fmaxf(v[6], v[7]), v[8]);
      const float mt = fmaxf(fmaxf(v[0], v[1]), v[2]);
      m0 = fmaxf(m0, m2);
      m0 = top ? fmaxf(m0, mt) : m0;
      yo[po_out[k] + (int64_t)p * Wp] = m0;
      m = fmaxf(m, fabsf(m0));
    }
    }
    __syncthreads();                                                    // the next step overwrites rows 2 p - 1 and 2 p
  }
  if (stat_out != nullptr) {
    const float wm = wave_max_nonneg(m);
    if (lane == 0) red[wave] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = red[0];
#pragma unroll
      for (int i = 1; i < kNW; ++i) t = fmaxf(t, red[i]);
      if (__float_as_uint(t) != 0u) atomic_max_f32(stat_out + smp, t);
    }
  }
}

// K2u' (round 6, last hours): the same kernel with the INPUT staged in LDS.  The form above gathers its B operand straight from
// global memory - 74 four-byte loads per lane and tile with an 8-byte lane stride - and the convolution alone takes ~324 us for a
// ~220 us matrix chain (profiles/r6_stem_pool_interleave_ab.txt): the texture path is as busy as the matrix pipe.  Here the nine
// input rows 4 p - 3 .. 4 p + 5 a step's two convolution rows read live in LDS ([slot = (iy + 8) mod 9][ci][4 zeros | W | 4 zeros]:
// the padding is data, no masks), filled by whole-row 16-byte loads: the four rows of the NEXT step are requested before the
// chain, written behind the step's first barrier (into the slots of the four rows no later step reads) and visible behind its
// second.  The chain reads one value per lane and step from LDS (lane stride 8 bytes: the 32 lanes of a half-wave hit 32 different
// banks), operands of the next four steps requested before the MFMAs of these four.  Same k order, same values.
constexpr int kXR = 9;                                                  // input rows resident
constexpr int kSTK = 2;                                                 // 16-byte loads per thread for four rows: 3 W <= 512 kSTK

template <int EPI>
__global__ __launch_bounds__(512, 2) void stem7_pool_lds_kernel(const float* __restrict__ x, const float* __restrict__ wt /*[3][7][7][64]*/,
                                                                const float* __restrict__ bias, float* __restrict__ y, StemPoolGeom g,
                                                                const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                                int act, float* __restrict__ stat_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
  float* const wl = reinterpret_cast<float*>(sp_smem);                  // [k][co], zero row for the padded k
  float* const c_bias = wl + kNS * 2 * kCout;
  float* const c_bsc = c_bias + kCout;
  float* const c_bsh = c_bsc + kCout;
  float* const red = c_bsh + kCout;                                     // 8 floats
  float* const ring = red + 16;                                         // [3][64][Wo]
  const int H = g.H, W = g.W, Ho = g.Ho, Wo = g.Wo, Wp = g.Wp;
  const int XW = W + 8;
  float* const xin = ring + 3 * kCout * Wo;                             // [9][3][XW]
  for (int i = threadIdx.x; i < kNS * 2 * kCout; i += 512) wl[i] = i < kK * kCout ? wt[i] : 0.0f;
  for (int i = threadIdx.x; i < kCout; i += 512) {
    c_bias[i] = bias != nullptr ? bias[i] : 0.0f;
    c_bsc[i] = bn_scale != nullptr ? bn_scale[i] : 1.0f;
    c_bsh[i] = bn_scale != nullptr ? bn_shift[i] : 0.0f;
  }
  for (int i = threadIdx.x; i < kXR * 3 * XW; i += 512) xin[i] = 0.0f;  // (the borders stay zero: rows are written from column 4 on)
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pl = lane & 31;
  const bool has_bn = bn_scale != nullptr;
  const int smp = (int)blockIdx.x / g.nbands, band = (int)blockIdx.x - smp * g.nbands;
  const int p_begin = band * g.rows_per_band;
  const int p_end = p_begin + g.rows_per_band < g.Hp ? p_begin + g.rows_per_band : g.Hp;
  const float* const xs = x + (int64_t)smp * 3 * H * W;
  const int ring_row = kCout * Wo;

  const int quarter = wave & 3, rsel = wave >> 2;
  const int ox = quarter * g.QW + pl;
  const bool col_ok = pl < g.QW && ox < Wo;
  const int lane_col = 2 * (ox < Wo ? ox : Wo - 1) + 1;                 // LDS column of tap kx = 0: ix + 4 = 2 ox - 3 + 4

  // ---- staging: four input rows (x three channels) per call, thread -> (row, channel, 16-byte column) ----------------------
  const int W4 = W >> 2;
  int st_r[kSTK], st_ci[kSTK], st_c4[kSTK];
#pragma unroll
  for (int k = 0; k < kSTK; ++k) {
    const int idx = (int)threadIdx.x + 512 * k;
    const int line = idx / W4;
    st_c4[k] = idx - line * W4;
    st_r[k] = line < 12 ? line / 3 : -1;
    st_ci[k] = line % 3;
  }
  f4 sreg[kSTK];
  auto stage_load = [&](int row_lo, int nrows) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < kSTK; ++k) {
      const int iy = row_lo + st_r[k];
      const bool ok = st_r[k] >= 0 && st_r[k] < nrows && iy >= 0 && iy < H;
      sreg[k] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
      if (ok) sreg[k] = *reinterpret_cast<const f4*>(xs + ((int64_t)st_ci[k] * H + iy) * W + 4 * st_c4[k]);
    }
  };
  auto stage_store = [&](int row_lo, int nrows) __attribute__((always_inline)) {
    const int s0 = (row_lo + 8) % kXR;                                  // (uniform; row_lo >= -5)
#pragma unroll
    for (int k = 0; k < kSTK; ++k) {
      if (st_r[k] < 0 || st_r[k] >= nrows) continue;
      int sl = s0 + st_r[k];
      sl -= sl >= kXR ? kXR : 0;
      *reinterpret_cast<f4*>(xin + (sl * 3 + st_ci[k]) * XW + 4 + 4 * st_c4[k]) = sreg[k];
    }
  };

  // ---- one tile: convolution row oy (wave-uniform) -> ring slot oy % 3 --------------------------------------------------------
  auto conv_tile = [&](int oy) __attribute__((always_inline)) {
    v16f acc[kCT];
#pragma unroll
    for (int c = 0; c < kCT; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.0f;
    int ro[kKS];                                                        // float offset of input row 2 oy - 3 + ky (uniform)
    {
      const int s0 = (2 * oy + 5) % kXR;                                // oy >= -1
#pragma unroll
      for (int ky = 0; ky < kKS; ++ky) {
        int sl = s0 + ky;
        sl -= sl >= kXR ? kXR : 0;
        ro[ky] = sl * 3 * XW;
      }
    }
    constexpr int G = 4, NG = (kNS + G - 1) / G;
    float wa[2][G][kCT], xb[2][G];
    auto load_group = [&](int gi, int set) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int s = gi * G + i;
        if (s >= kNS) continue;
        const int k0 = 2 * s, k1 = 2 * s + 1;
        const int ky0 = (k0 / kKS) % kKS, kx0 = k0 % kKS, ci0 = k0 / (kKS * kKS);
        const int ky1 = k1 < kK ? (k1 / kKS) % kKS : 0, kx1 = k1 < kK ? k1 % kKS : 0, ci1 = k1 < kK ? k1 / (kKS * kKS) : 0;
        const int o0 = ro[ky0] + ci0 * XW + kx0 + lane_col;
        const int o1 = k1 < kK ? ro[ky1] + ci1 * XW + kx1 + lane_col : 0;      // the padded k: a border zero (its weight row is zero)
        xb[set][i] = xin[h ? o1 : o0];
#pragma unroll
        for (int ct = 0; ct < kCT; ++ct) wa[set][i][ct] = wl[(2 * s + h) * kCout + ct * 32 + pl];
      }
    };
    load_group(0, 0);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG) load_group(gi + 1, (gi + 1) & 1);
      FQ_PIN();
#pragma unroll
      for (int i = 0; i < G; ++i) {
        if (gi * G + i >= kNS) continue;
#pragma unroll
        for (int ct = 0; ct < kCT; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[gi & 1][i][ct], xb[gi & 1][i], acc[ct], 0, 0, 0);
      }
      FQ_PIN();
    }
    if (oy >= 0 && oy < Ho) {
      float* const dst = ring + (oy % 3) * ring_row + ox;
#pragma unroll
      for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int c0 = ct * 32 + 8 * gq + 4 * h;
          const f4 bch = *reinterpret_cast<const f4*>(c_bias + c0);
          const f4 bsc = *reinterpret_cast<const f4*>(c_bsc + c0);
          const f4 bsh = *reinterpret_cast<const f4*>(c_bsh + c0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = dw_finish<EPI>(acc[ct][4 * gq + r], bias != nullptr, bch[r], has_bn, bsc[r], bsh[r], act);
            if (col_ok) dst[(c0 + r) * Wo] = v;
          }
        }
    }
  };

  // ---- prologue: the nine rows of the first step (and, below the image's first band, the halo row 2 p_begin - 1 before them) ----
  {
    const int lo = p_begin > 0 ? 4 * p_begin - 5 : -3;
    for (int r = 0; r < kXR; r += 4) {
      stage_load(lo + r, kXR - r < 4 ? kXR - r : 4);
      stage_store(lo + r, kXR - r < 4 ? kXR - r : 4);
    }
    __syncthreads();
    if (p_begin > 0) {                                                  // (uniform)
      conv_tile(rsel == 0 ? 2 * p_begin - 1 : -1);
      __syncthreads();                                                  // rows 4 p_begin - 5, - 4 are read no more
      stage_load(4 * p_begin + 4, 2);
      stage_store(4 * p_begin + 4, 2);
      __syncthreads();
    }
  }
  // (the pooling phase is the one of the kernel above: four pooled outputs per task)
  const bool pool4 = (Wo & 7) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0;
  const int wq = Wp >> 2, total4 = kCout * wq;
  constexpr int kPool4Per = (kCout * 16 + 511) / 512;
  int p4_off[kPool4Per], p4_out[kPool4Per];
#pragma unroll
  for (int k = 0; k < kPool4Per; ++k) {
    const int idx = (int)threadIdx.x + 512 * k;
    const int ch = idx / (wq > 0 ? wq : 1), g4 = idx - ch * wq;
    p4_off[k] = (pool4 && idx < total4) ? ((ch * Wo + 8 * g4) | ((g4 > 0 ? 1 : 0) << 20)) : -1;
    p4_out[k] = ch * g.Hp * Wp + 4 * g4;
  }
  float* const yo = y + ((int64_t)smp * kCout * g.Hp) * Wp;
  float m = 0.0f;
  for (int p = p_begin; p < p_end; ++p) {
    const bool more = p + 1 < p_end;
    if (more) stage_load(4 * p + 6, 4);                                 // the next step's four new rows: in flight under the chain
    conv_tile(2 * p + rsel);
    __syncthreads();                                                    // rows 2 p, 2 p + 1 are in the ring; input rows 4 p - 3 .. 4 p are read no more
    if (more) stage_store(4 * p + 6, 4);
    const int r_hi = 2 * p + 1 < Ho ? 2 * p + 1 : Ho - 1;
    const bool top = p > 0;
    const float* const row0 = ring + ((2 * p + 2) % 3) * ring_row;      // row 2 p - 1
    const float* const row1 = ring + ((2 * p) % 3) * ring_row;
    const float* const row2 = ring + (r_hi % 3) * ring_row;
#pragma unroll
    for (int k = 0; k < kPool4Per; ++k) {
      if (p4_off[k] < 0) continue;
      const int base = p4_off[k] & 0xFFFFF, c_lo = (p4_off[k] >> 20) & 1;
      f4 hm[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float* src = (r == 0 ? row0 : r == 1 ? row1 : row2) + base;
        const float lo = src[-c_lo];
        const f4 a = *reinterpret_cast<const f4*>(src), b = *reinterpret_cast<const f4*>(src + 4);
        hm[r] = (f4){fmaxf(fmaxf(lo, a.x), a.y), fmaxf(fmaxf(a.y, a.z), a.w), fmaxf(fmaxf(a.w, b.x), b.y),
                     fmaxf(fmaxf(b.y, b.z), b.w)};
      }
      f4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float m0 = fmaxf(hm[1][j], hm[2][j]);
        m0 = top ? fmaxf(m0, hm[0][j]) : m0;
        o[j] = m0;
        m = fmaxf(m, fabsf(m0));
      }
      *reinterpret_cast<f4*>(yo + p4_out[k] + (int64_t)p * Wp) = o;
    }
    __syncthreads();                                                    // the next step overwrites ring rows 2 p - 1 and 2 p; its input rows are in place
  }
  if (stat_out != nullptr) {
    const float wm = wave_max_nonneg(m);
    if (lane == 0) red[wave] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = red[0];
#pragma unroll
      for (int i = 1; i < kNW; ++i) t = fmaxf(t, red[i]);
      if (__float_as_uint(t) != 0u) atomic_max_f32(stat_out + smp, t);
    }
  }
}

}  // namespace

using namespace fqi;

extern "C" {

int fq_stem_conv7x7s2_pool_supported(int64_t h, int64_t w) {
  const int64_t Ho = (h + 6 - 7) / 2 + 1, Wo = (w + 6 - 7) / 2 + 1;
  const size_t lds = (size_t)(kNS * 2 * kCout + 3 * kCout + 16 + 3 * kCout * Wo) * sizeof(float);
  return (h > 0 && w > 0 && Wo >= 4 && (Wo + 3) / 4 <= 32 && Ho >= 1 && lds + 1024 <= (size_t)max_lds_bytes() &&
          3 * h * w * 4 < (1ll << 31)) ? 1 : 0;
}

// Conv2D(3 -> 64, 7x7, stride 2, pad 3) [+ bias] [-> BatchNorm] [-> activation] -> MaxPool2D(3, stride 2, pad 1), fp32
// (reference: the first blocks of gluoncv's resnet*_v1 behind examples/simulate_quantization.py's --exclude-first-conv: MXNet's
// Convolution, BatchNorm, Activation, Pooling operators one after the other).  y: (n, 64, Hp, Wp) with Hp = (Ho - 1) / 2 + 1;
// stat_out (optional, n floats, zeroed here unless FQ_STAT_PREZEROED): per-sample max|y| of the pooled tensor.
int fq_stem_conv7x7s2_pool(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n, int64_t cin,
                           int64_t cout, int64_t h, int64_t w, const float* bn_scale, const float* bn_shift, int act,
                           float* stat_out, fqStream_t stream) {
  FQ_REQUIRE(x && w_tap_major && y, "fq_stem_conv7x7s2_pool: null pointer");
  FQ_REQUIRE(cin == 3 && cout == 64, "fq_stem_conv7x7s2_pool: built for 3 -> 64 channels (got %lld -> %lld)", (long long)cin,
             (long long)cout);
  FQ_REQUIRE(n > 0 && n < 65536 && fq_stem_conv7x7s2_pool_supported(h, w), "fq_stem_conv7x7s2_pool: bad shape (rows of the "
             "convolution output must cut into four tiles of at most 32 columns and three of them must fit LDS)");
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_stem_conv7x7s2_pool: bn_scale and bn_shift go together");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_stem_conv7x7s2_pool: unknown activation %d", act);
  hipStream_t st = (hipStream_t)stream;
  StemPoolGeom g;
  g.H = (int)h; g.W = (int)w;
  g.Ho = (int)((h + 6 - 7) / 2 + 1); g.Wo = (int)((w + 6 - 7) / 2 + 1);
  g.Hp = (g.Ho - 1) / 2 + 1; g.Wp = (g.Wo - 1) / 2 + 1;
  g.QW = (g.Wo + 3) / 4;
  g.n = (int)n;
  // bands: enough workgroups for every CU, as few halo rows as that allows
  int nb = (int)((num_cu() + n - 1) / n);
  if (nb < 1) nb = 1;
  if (nb > g.Hp) nb = g.Hp;
  g.rows_per_band = (g.Hp + nb - 1) / nb;
  g.nbands = (g.Hp + g.rows_per_band - 1) / g.rows_per_band;
  const size_t lds = (size_t)(kNS * 2 * kCout + 3 * kCout + 16 + 3 * kCout * g.Wo) * sizeof(float);
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  const double in_b = 4.0 * (double)n * 3 * h * w, out_b = 4.0 * (double)n * 64 * g.Hp * g.Wp;
  ProfScope prof(FQ_KERNEL_STEM, in_b + out_b, st);
  const int epi = (bn_scale != nullptr && bias == nullptr)
                      ? (act == FQ_ACT_RELU ? kEpiBnRelu : act == FQ_ACT_RELU6 ? kEpiBnRelu6 : kEpiRuntime)
                      : kEpiRuntime;
  const dim3 grid((unsigned)(n * g.nbands));
#define FQ_SP_LAUNCH(E_)                                                                                               \
  {                                                                                                                    \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem7_pool_kernel<E_>),             \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) == hipSuccess; \
    FQ_REQUIRE(attr_ok, "fq_stem_conv7x7s2_pool: cannot raise the dynamic LDS limit");                                 \
    hipLaunchKernelGGL((stem7_pool_kernel<E_>), grid, dim3(512), lds, st, x, w_tap_major, bias, y, g, bn_scale, bn_shift, \
                       act, stat_out);                                                                                 \
  }
  // the input staged in LDS (stem7_pool_lds_kernel) where it fits beside the ring: whole rows of 16-byte loads, rows of a
  // multiple of eight convolution columns (its pooling phase is the four-outputs-per-task one), FQ_STEM_POOL_LDS=0: never
  static const int use_lds = env_int("FQ_STEM_POOL_LDS", 1);
  const size_t lds_in = lds + (size_t)kXR * 3 * (w + 8) * sizeof(float);
  const bool staged = use_lds != 0 && (w & 3) == 0 && (g.Wo & 7) == 0 && 3 * w <= 512 * kSTK &&
                      ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(y)) & 15) == 0 &&
                      lds_in + 1024 <= (size_t)max_lds_bytes();
#define FQ_SPL_LAUNCH(E_)                                                                                              \
  {                                                                                                                    \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem7_pool_lds_kernel<E_>),         \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) == hipSuccess; \
    FQ_REQUIRE(attr_ok, "fq_stem_conv7x7s2_pool: cannot raise the dynamic LDS limit");                                 \
    hipLaunchKernelGGL((stem7_pool_lds_kernel<E_>), grid, dim3(512), lds_in, st, x, w_tap_major, bias, y, g, bn_scale,  \
                       bn_shift, act, stat_out);                                                                       \
  }
  if (staged) {
    if (epi == kEpiBnRelu) FQ_SPL_LAUNCH(kEpiBnRelu)
    else if (epi == kEpiBnRelu6) FQ_SPL_LAUNCH(kEpiBnRelu6)
    else FQ_SPL_LAUNCH(kEpiRuntime)
  } else if (epi == kEpiBnRelu) FQ_SP_LAUNCH(kEpiBnRelu)
  else if (epi == kEpiBnRelu6) FQ_SP_LAUNCH(kEpiBnRelu6)
  else FQ_SP_LAUNCH(kEpiRuntime)
#undef FQ_SP_LAUNCH
#undef FQ_SPL_LAUNCH
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // extern "C"
