// libfakequant — K2j pointwise (1x1) convolution on int8 codes for few-tile layers; weight codes (fq_weight_codes)
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_pw.h"

namespace {

// K2j: tile form for the deep layers (K = 256 / 512 / 1024 on 14x14 and 7x7 planes), where there are few pixels
// (784 or 196 tiles of 32) and a big weight matrix.  ONE 32-pixel tile per workgroup; its four wavefronts
//   1. each quantise a QUARTER of the K/32 channel slabs (lane = pixel, as K2h) and publish the int8 fragments to an LDS
//      panel (1 KB per slab, lane order) - the quantise phase is 4x shorter than one wave per tile and nothing is
//      quantised twice;
//   2. each take a quarter of the 32-channel output tiles and multiply: B fragments from registers (K <= 512) or from
//      the LDS panel (K = 1024), A fragments streamed STRAIGHT from L2 out of the fragment-major copy fq_weight_codes
//      leaves behind the row-major codes (one coalesced 16-byte load per lane per MFMA; no weight staging, no per-chunk
//      barrier), two channel tiles at a time on independent accumulators;
//   3. store with lane = pixel (two full lines per store instruction), per-channel constants from LDS.
// Two barriers per tile.  Against K2i: four times the wavefronts, no serial quantise-then-multiply per wave.
struct PwtGeom {
  int Cin, K, Cout, CT, HW;   // K: padded row length of the weight codes; CT = Cout / 32 (Cout % 256 == 0 here)
  int64_t cols, tiles;
  int zoff;
};

#ifndef FQ_PWT_LB
#define FQ_PWT_LB 2                 // wavefronts per SIMD the register allocation aims at (3 spills ~400 registers)
#endif
template <int KT, bool BLDS>
__global__ __launch_bounds__(kBlock, FQ_PWT_LB) void pwconv_tile_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wfrag, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwtGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out) {
  constexpr int kSlots = 8;
  constexpr int SLABS = KT / 4;                                         // slabs each wavefront quantises
  extern __shared__ __attribute__((aligned(16))) unsigned char pwt_smem[];
  __shared__ unsigned k_stat[kSlots];
  v4i* panel = reinterpret_cast<v4i*>(pwt_smem);                       // [KT][64] B fragments of the current tile
  const int nch = g.CT * 32;
  float* c_sxw = reinterpret_cast<float*>(pwt_smem + (size_t)KT * 1024);
  float* c_bsc = c_sxw + nch;
  float* c_bsh = c_bsc + nch;
  float* c_bias = c_bsh + nch;
  int* c_zs = reinterpret_cast<int*>(c_bias + nch);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, pl = lane & 31;
  const unsigned HW = (unsigned)g.HW, cols = (unsigned)g.cols;
  const int64_t plane = (int64_t)g.HW;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const int64_t t_begin = g.tiles * blockIdx.x / gridDim.x, t_end = g.tiles * (blockIdx.x + 1) / gridDim.x;
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
  auto issue = [&](const Pix& px, int kt, float (&v)[16]) __attribute__((always_inline)) {
    const int cg = kt * 32 + 16 * h;
    const unsigned off = (unsigned)((((int64_t)px.smp * g.Cin + (cg < g.Cin ? 16 * h : 0)) * plane + px.p) * 4);
    const char* ub = reinterpret_cast<const char*>(x) + (int64_t)kt * 32 * plane * 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const float*>(ub + (int64_t)i * plane * 4 + off);
  };

  float bufa[16], bufb[16];
  Pix px = pix_of(t_begin);
  issue(px, wave, bufa);                                                // in flight during the set-up
  FQ_PIN();
  const float max_ = in_stat != nullptr ? batch_mean_dev(in_stat, n) : in_thr[0];
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  if (in_stat != nullptr && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
  const float sx = q.scale;
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;
  for (int i = threadIdx.x; i < nch; i += kBlock) {
    const bool ok = i < g.Cout;
    const int ic = ok ? i : 0;
    c_sxw[i] = sx * wscale[ic];
    c_zs[i] = ok ? g.zoff * wsum[ic] : 0;
    c_bias[i] = bias != nullptr ? bias[ic] : 0.0f;
    c_bsc[i] = has_bn ? bn_scale[ic] : 1.0f;
    c_bsh[i] = has_bn ? bn_shift[ic] : 0.0f;
  }
  auto quant_to_panel = [&](int kt, const float (&v)[16]) __attribute__((always_inline)) {
    const bool gvalid = kt * 32 + 16 * h < g.Cin;
    v4i f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int packed = pack4_codes(fq_code_int(v[4 * d + 0], q), fq_code_int(v[4 * d + 1], q),
                                     fq_code_int(v[4 * d + 2], q), fq_code_int(v[4 * d + 3], q), 128 - g.zoff);
      f[d] = gvalid ? packed : 0;
    }
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));   // pin the arithmetic here (see K2h)
    panel[(kt << 6) + lane] = f;
  };

  auto run = [&](auto bias_c, auto bn_c, auto act_c) __attribute__((always_inline)) {
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
    for (int64_t t = t_begin; t < t_end; ++t) {
      // ---- 1. my quarter of the slabs -> LDS panel (the first slab of this tile is already in bufa) ----------------
#pragma unroll
      for (int j = 0; j < SLABS; ++j) {
        const int kt = wave + 4 * j;
        float (&mine)[16] = (j & 1) ? bufb : bufa;
        float (&other)[16] = (j & 1) ? bufa : bufb;
        if (j + 1 < SLABS) issue(px, kt + 4, other);
        FQ_PIN();
        quant_to_panel(kt, mine);
        FQ_PIN();
      }
      const Pix cur = px;
      px = pix_of(t + 1);
      static_assert(SLABS % 2 == 0, "the last slab of a tile must leave bufa free");
      issue(px, wave, bufa);                                            // next tile's first slab: in flight during the GEMM
      FQ_PIN();
      __syncthreads();                                                  // panel complete
      v4i bfrag[BLDS ? 1 : KT];
      if (!BLDS) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) bfrag[kt] = panel[(kt << 6) + lane];
      }
      // ---- 2./3. my quarter of the channel tiles, two at a time ------------------------------------------------------
      const unsigned yoff = (unsigned)((((int64_t)cur.smp * g.Cout + 4 * h) * plane + cur.p) * 4);
      float m = 0.0f;
      auto load_zs = [&](int ct, v16i& acc) __attribute__((always_inline)) {
        const int cb = ct * 32 + 4 * h;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const v4i z = *reinterpret_cast<const v4i*>(c_zs + cb + 8 * gq);
          acc[4 * gq + 0] = z.x; acc[4 * gq + 1] = z.y; acc[4 * gq + 2] = z.z; acc[4 * gq + 3] = z.w;
        }
      };
      auto finish = [&](int ct, const v16i& acc) __attribute__((always_inline)) {
        const int cb = ct * 32 + 4 * h;
        char* ybase = reinterpret_cast<char*>(y) + (int64_t)(ct * 32) * plane * 4;
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
            float v = (float)acc[4 * gq + r] * sxw[r];
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
      };
      const v4i* wf = reinterpret_cast<const v4i*>(wfrag) + lane;
#pragma unroll 1
      for (int ct0 = wave; ct0 < g.CT; ct0 += 8) {
        const int ct1 = ct0 + 4;
        v16i acc0, acc1;
        load_zs(ct0, acc0);
        load_zs(ct1, acc1);
        const v4i* A0 = wf + ((int64_t)(ct0 * KT) << 6);
        const v4i* A1 = wf + ((int64_t)(ct1 * KT) << 6);
        v4i fa[3], fb[3];
        fa[0] = A0[0]; fb[0] = A1[0];
        fa[1] = A0[64]; fb[1] = A1[64];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int kn = kt + 2 < KT ? kt + 2 : KT - 1;
          fa[(kt + 2) % 3] = A0[kn << 6];
          fb[(kt + 2) % 3] = A1[kn << 6];
          const v4i b = BLDS ? panel[(kt << 6) + lane] : bfrag[BLDS ? 0 : kt];
          acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[kt % 3], b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[kt % 3], b, acc1, 0, 0, 0);
          FQ_PIN();                              // only the two-ahead prefetch in flight (else all 2*KT loads are hoisted)
        }
        finish(ct0, acc0);
        finish(ct1, acc1);
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
      __syncthreads();                                                  // panel free for the next tile
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
    if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < cols / HW)
      atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
}

// weight codes: one workgroup per (padded) row: code = roundf(w / (s + eps)), zero padding, row sums
__global__ __launch_bounds__(kBlock) void weight_codes_kernel(const float* __restrict__ w, int rows, int row_len,
                                                              int rows_per_scale, float levels, int row_pad,
                                                              const float* __restrict__ gmax,
                                                              int8_t* __restrict__ codes, float* __restrict__ scales,
                                                              int* __restrict__ rowsum, int8_t* __restrict__ frag) {
  // `frag` (second half of the codes buffer): the same codes in MFMA-fragment order for v_mfma_i32_32x32x32_i8 with the
  // weights as the A operand: fragment (ct = row / 32, kt = k / 32) is 1 KB = 64 lanes x 16 bytes, lane = row % 32 +
  // 32 * ((k % 32) / 16), byte = k % 16 - so a wavefront fetches one fragment with ONE fully coalesced 16-byte load
  __shared__ int red[4];
  const int r = blockIdx.x;
  int8_t* dst = codes + (int64_t)r * row_pad;
  const int kts = row_pad >> 5;
  auto frag_at = [&](int i) -> int8_t* {
    const int kt = i >> 5, hs = (i >> 4) & 1, b = i & 15;
    return frag + ((((int64_t)(r >> 5) * kts + kt) << 6) + (r & 31) + 32 * hs) * 16 + b;
  };
  if (r >= rows) {                                                     // padded row
    for (int i = threadIdx.x; i < row_pad; i += kBlock) {
      dst[i] = 0;
      *frag_at(i) = 0;
    }
    return;
  }
  const float s = gmax[r / rows_per_scale] / levels;
  const float d = s + kEps;
  int acc = 0;
  for (int i = threadIdx.x; i < row_pad; i += kBlock) {
    int c = 0;
    if (i < row_len) c = (int)roundf(w[(int64_t)r * row_len + i] / d);
    dst[i] = (int8_t)c;
    *frag_at(i) = (int8_t)c;
    acc += c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    rowsum[r] = red[0] + red[1] + red[2] + red[3];
    scales[r] = s;
  }
}


}  // namespace

namespace fqi {

// tile form (K2j): one 32-pixel tile per workgroup, weights streamed from the fragment-major copy.
// Every workgroup streams the whole weight matrix from L2, so this form only pays when tiles are few (7x7 planes:
// 196 tiles; measured 41 / 44 us against 55 / 64 us for the chunked / two-kernel forms on 512->1024 and
// 1024->1024); with 784 tiles (14x14) the 200 MB of weight traffic make it twice as slow as the chunked form.
int pw_try_tile(const PwCall& a, bool* taken) {
  *taken = false;
  const int kt = (int)((a.cin + 31) / 32);
  const int ct = (int)((a.cout + 31) / 32);
  const bool few_tiles = (a.n * a.hw + 31) / 32 < (int64_t)num_cu() * 2;
  if (((a.form == 0 && few_tiles) || a.form == 5) && a.cin % 16 == 0 && a.cout % 256 == 0 && a.cin_pad == a.cin &&
      (kt == 8 || kt == 16 || kt == 32)) {
    PwtGeom t;
    t.Cin = (int)a.cin; t.K = (int)a.cin_pad; t.Cout = (int)a.cout; t.CT = ct; t.HW = (int)a.hw;
    t.cols = a.n * a.hw; t.tiles = (t.cols + 31) / 32; t.zoff = a.zoff;
    const size_t ldst = (size_t)kt * 1024 + (size_t)ct * 32 * 5 * sizeof(float);
    const int64_t rows_pad = (a.cout + 63) / 64 * 64;
    const int8_t* wfrag = a.wcodes + rows_pad * a.cin_pad;               // second half of fq_weight_codes' buffer
    static const int pwt_wg = env_int("FQ_PWT_WG_PER_CU", 4);
    int64_t grid = (int64_t)num_cu() * pwt_wg;
    if (grid > t.tiles) grid = t.tiles;
    if (int rc = pw_zero_stat(a)) return rc;
#define FQ_PWT_CASE(KT_, BL_)                                                                                          \
  if (kt == KT_) {                                                                                                     \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_tile_kernel<KT_, BL_>),      \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess; \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the tile kernel");                        \
    hipLaunchKernelGGL((pwconv_tile_kernel<KT_, BL_>), dim3((unsigned)grid), dim3(kBlock), ldst, a.st, a.x, wfrag,     \
                       a.wscale, (const int*)a.wsum, a.bias, a.y, t, a.in_stat, (int)a.n, a.in_thr, a.levels, a.lo_neg, \
                       kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out);                            \
  }
#ifndef FQ_PWT_REGB
#define FQ_PWT_REGB false       // true: B fragments in registers for K <= 512 (spills at the 3-waves-per-SIMD register cap)
#endif
    FQ_PWT_CASE(8, !FQ_PWT_REGB) FQ_PWT_CASE(16, !FQ_PWT_REGB) FQ_PWT_CASE(32, true)
#undef FQ_PWT_CASE
    FQ_LAUNCH_CHECK();
    *taken = true;
    return FQ_OK;
  }
  FQ_REQUIRE(a.form != 5, "fq_pwconv_i8: FQ_PW_FORM=5 but the shape does not fit the tile kernel");
  return FQ_OK;
}

}  // namespace fqi

extern "C" {

int fq_weight_codes(const float* w, int64_t rows, int64_t row_len, int rows_per_scale, int width, int64_t row_pad,
                    int64_t rows_pad, int8_t* codes, float* scales, int32_t* rowsum, void* ws, fqStream_t stream) {
  FQ_REQUIRE(w && codes && scales && rowsum && ws, "fq_weight_codes: null pointer");
  FQ_REQUIRE(rows > 0 && row_len > 0 && rows_per_scale > 0 && rows % rows_per_scale == 0,
             "fq_weight_codes: bad shape (rows=%lld row_len=%lld rows_per_scale=%d)", (long long)rows,
             (long long)row_len, rows_per_scale);
  FQ_REQUIRE(width >= 2 && width <= 8, "fq_weight_codes: width %d does not fit int8 codes", width);
  FQ_REQUIRE(row_pad >= row_len && rows_pad >= rows && rows_pad < (1ll << 31) && row_pad % 32 == 0 && rows_pad % 32 == 0,
             "fq_weight_codes: bad padding (row_pad and rows_pad must be multiples of 32)");
  hipStream_t st = (hipStream_t)stream;
  const int64_t groups = rows / rows_per_scale;
  float* gmax = (float*)ws;
  FQ_HIP(hipMemsetAsync(gmax, 0, groups * sizeof(float), st));
  if (int rc = launch_absmax(w, groups, (int64_t)rows_per_scale * row_len, true, gmax, st)) return rc;
  const float levels = (float)((1 << (width - 1)) - 1);
  hipLaunchKernelGGL(weight_codes_kernel, dim3((unsigned)rows_pad), dim3(kBlock), 0, st, w, (int)rows, (int)row_len,
                     rows_per_scale, levels, (int)row_pad, gmax, codes, scales, (int*)rowsum,
                     codes + rows_pad * row_pad);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // extern "C"
