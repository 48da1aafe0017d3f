// libfakequant — K2i pointwise (1x1) convolution on int8 codes, weights streamed through LDS in chunks
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_pw.h"

namespace {

// K2i: the streaming form for layers whose weights do NOT fit in LDS (K = 256 / 512, Cout up to 1024).  Activations
// exactly as K2h (lane = pixel, codes built in registers: K/32 fragments of 4 VGPRs stay resident for the whole tile),
// the weight matrix streams through LDS in chunks of CTC channel tiles (32 KB), double buffered: while the wavefronts
// multiply chunk c (A fragments by ds_read_b128, B fragments from registers) and store its outputs, every thread has
// chunk c+1's 16-byte pieces in flight from L2, and writes them to the other LDS buffer before the (single) barrier of
// the iteration.  The chunk sequence is cyclic, so the pipeline runs across tile batches.  `wsplit` wavefronts share one
// 32-pixel tile and divide a chunk's channel tiles among themselves when there are too few pixels to give every
// wavefront its own tile (7x7 planes); they quantise that tile redundantly.
struct PwcGeom {
  int Cin, K, Cout, CT, HW;   // K: row stride of the weight codes; CT = ceil(Cout / 32)
  int CTC, NC;                // channel tiles per chunk, chunks = ceil(CT / CTC)
  int wsplit;                 // wavefronts per tile: 1, 2 or 4
  int rows;                   // rows of the weight code buffer (Cout rounded up to 64)
  int64_t cols, tiles, batches;   // n * HW, ceil(cols / 32), ceil(tiles / (4 / wsplit))
  int zoff;
};

template <int KT, int PIECES>
__global__ __launch_bounds__(kBlock, 2) void pwconv_chunk_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wc, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwcGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out, float* __restrict__ sink) {
  constexpr int kSlots = 8;
  constexpr int kMaxPieces = PIECES;                                    // 16-byte pieces per thread: CTC * KT / 4
  extern __shared__ __attribute__((aligned(16))) unsigned char pwc_smem[];
  __shared__ unsigned k_stat[kSlots];
  const int chunk_frags = g.CTC * KT;                                   // 1 KB fragments per chunk
  const int nchc = g.CTC * 32;                                          // channels per chunk
  const size_t buf_bytes = (size_t)chunk_frags * 1024 + (size_t)nchc * 5 * sizeof(float);
  auto bufA = [&](int b) __attribute__((always_inline)) { return reinterpret_cast<v4i*>(pwc_smem + (size_t)b * buf_bytes); };
  auto bufC = [&](int b) __attribute__((always_inline)) {
    return reinterpret_cast<float*>(pwc_smem + (size_t)b * buf_bytes + (size_t)chunk_frags * 1024);
  };

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, pl = lane & 31;
  const unsigned HW = (unsigned)g.HW, cols = (unsigned)g.cols;
  const int64_t plane = (int64_t)g.HW;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const int tiles_per_wg = 4 / g.wsplit;
  const int my_tile = wave / g.wsplit, sub = wave - my_tile * g.wsplit;
  const int64_t b_begin = g.batches * blockIdx.x / gridDim.x, b_end = g.batches * (blockIdx.x + 1) / gridDim.x;
  unsigned s_base;
  {
    const unsigned j0 = (unsigned)(b_begin * tiles_per_wg) * 32u;
    s_base = (j0 < cols ? j0 : cols - 1) / HW;
  }

  PW_STAMP(0);
  const float max_ = in_stat != nullptr ? batch_mean_dev(in_stat, n) : in_thr[0];
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  if (in_stat != nullptr && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
  const float sx = q.scale;
  if (threadIdx.x < kSlots) k_stat[threadIdx.x] = 0u;

  // ---- weight chunk staging (global -> registers -> LDS, fragment order) ---------------------------------------------
  v4i stage[kMaxPieces];
  float cst[5];
  const unsigned w_lane_off = (unsigned)(pl * g.K + h * 16);            // row pl, 16-byte half h of a 32-byte slab
  auto chunk_issue = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i) {
      const int f = wave + 4 * i;                                       // fragment inside the chunk (wave-uniform)
      const int ctl = f / KT, kt = f - ctl * KT;
      int row0 = (c * g.CTC + ctl) * 32;
      row0 = row0 + 32 <= g.rows ? row0 : g.rows - 32;                  // tiles past the padded buffer: discarded channels
      stage[i] = *reinterpret_cast<const v4i*>(wc + ((int64_t)row0 * g.K + kt * 32) + w_lane_off);
    }
    // per-channel constants: RAW loads only here (any arithmetic on them would make the compiler wait for them - and,
    // the counter being in-order, for everything issued before - right at the top of the iteration)
    if ((int)threadIdx.x < nchc) {
      const int ch = c * nchc + threadIdx.x;
      const int cc = ch < g.Cout ? ch : 0;
      cst[0] = wscale[cc];
      cst[1] = has_bn ? bn_scale[cc] : 1.0f;
      cst[2] = has_bn ? bn_shift[cc] : 0.0f;
      cst[3] = bias != nullptr ? bias[cc] : 0.0f;
      cst[4] = __int_as_float(wsum[cc]);
    }
  };
  auto chunk_commit = [&](int b) __attribute__((always_inline)) {
    v4i* A = bufA(b);
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i) A[((wave + 4 * i) << 6) + lane] = stage[i];
    if ((int)threadIdx.x < nchc) {
      float* C = bufC(b);
      C[0 * nchc + threadIdx.x] = sx * cst[0];
      C[1 * nchc + threadIdx.x] = cst[1];
      C[2 * nchc + threadIdx.x] = cst[2];
      C[3 * nchc + threadIdx.x] = cst[3];
      C[4 * nchc + threadIdx.x] = __int_as_float(g.zoff * __float_as_int(cst[4]));
    }
  };

  struct Pix { unsigned smp, p; bool valid; };
  auto pix_of = [&](int64_t t) __attribute__((always_inline)) {
    Pix r;
    unsigned j = (unsigned)t * 32u + (unsigned)pl;
    r.valid = t < g.tiles && j < cols;
    j = r.valid ? j : cols - 1;
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
  v4i bfrag[KT];
  auto quant = [&](int kt, const float (&v)[16]) __attribute__((always_inline)) {
    const bool gvalid = kt * 32 + 16 * h < g.Cin;
    v4i f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int packed = pack4_codes(fq_code_int(v[4 * d + 0], q), fq_code_int(v[4 * d + 1], q),
                                     fq_code_int(v[4 * d + 2], q), fq_code_int(v[4 * d + 3], q), 128 - g.zoff);
      f[d] = gvalid ? packed : 0;
    }
    // pin the quantisation HERE: it is pure arithmetic whose results are only needed by the MFMAs, and the optimiser
    // otherwise sinks it below every prefetch, keeping all 16 * KT loaded values live (256 VGPRs + spills)
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    bfrag[kt] = f;
  };

  PW_STAMP(1);
  chunk_issue(0);
  chunk_commit(0);
  __syncthreads();
  PW_STAMP(2);

  auto run = [&](auto bias_c, auto bn_c, auto act_c) __attribute__((always_inline)) {
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
    float bufa[16], bufb[16], bufc[16];
    for (int64_t b = b_begin; b < b_end; ++b) {
      const Pix px = pix_of(b * tiles_per_wg + my_tile);
      // phase Q: the tile's K/32 slabs -> B fragments.  Three load buffers, two slabs (32 dwords per lane) in flight:
      // with one workgroup per CU a single slab in flight left every quantisation waiting a full HBM latency
      issue(px, 0, bufa);
      issue(px, 1 < KT ? 1 : 0, bufb);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int nk = kt + 2 < KT ? kt + 2 : KT - 1;
        if (kt % 3 == 0) { issue(px, nk, bufc); FQ_PIN(); quant(kt, bufa); }
        else if (kt % 3 == 1) { issue(px, nk, bufa); FQ_PIN(); quant(kt, bufb); }
        else { issue(px, nk, bufb); FQ_PIN(); quant(kt, bufc); }
        FQ_PIN();
      }
      const unsigned yoff = (unsigned)((((int64_t)px.smp * g.Cout + 4 * h) * plane + px.p) * 4);
      float m = 0.0f;
      if (b == b_begin) PW_STAMP(3);
      for (int c = 0; c < g.NC; ++c) {
        const int cur = c & 1;                                          // NC is even or 1 (host): buffers line up across batches
        chunk_issue(c + 1 < g.NC ? c + 1 : 0);
        FQ_PIN();
        const v4i* A = bufA(g.NC == 1 ? 0 : cur);
        const float* C = bufC(g.NC == 1 ? 0 : cur);
        // two channel tiles at a time on independent accumulators: one accumulator chained 16 dependent MFMAs, each
        // waiting out the previous one's full latency
        auto load_zs = [&](int ctl, v16i& acc) __attribute__((always_inline)) {
          const int cb = ctl * 32 + 4 * h;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const v4i z = *reinterpret_cast<const v4i*>(C + 4 * nchc + cb + 8 * gq);
            acc[4 * gq + 0] = z.x; acc[4 * gq + 1] = z.y; acc[4 * gq + 2] = z.z; acc[4 * gq + 3] = z.w;
          }
        };
        // No masks in the epilogue: lanes past the end of the tensor hold a copy of the LAST pixel (clamped loads), so
        // they recompute and re-store that pixel's values (benign duplicates); the host guarantees Cout % 32 == 0 and
        // CT % CTC == 0.  Masked stores cost 2 selects + 64-bit address arithmetic per output - more than the MFMAs.
        auto finish = [&](int ctl, const v16i& acc) __attribute__((always_inline)) {
          const int cb = ctl * 32 + 4 * h;                              // channel index inside the chunk
          char* ybase = reinterpret_cast<char*>(y) + (int64_t)(c * nchc + ctl * 32) * plane * 4;   // wave-uniform
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int c0 = cb + 8 * gq;
            const f4 sxw = *reinterpret_cast<const f4*>(C + c0);
            const f4 bsc = *reinterpret_cast<const f4*>(C + nchc + c0);
            const f4 bsh = *reinterpret_cast<const f4*>(C + 2 * nchc + c0);
            f4 bch = (f4){0.f, 0.f, 0.f, 0.f};
            if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) bch = *reinterpret_cast<const f4*>(C + 3 * nchc + c0);
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
        // A fragments are read two K steps ahead of their MFMAs (LDS latency ~ two 32-cycle MFMAs)
#pragma unroll 1
        for (int ctl = sub; ctl < g.CTC; ctl += 2 * g.wsplit) {
          const int ctl2 = ctl + g.wsplit;
          if (ctl2 < g.CTC) {
            v16i acc0, acc1;
            load_zs(ctl, acc0);
            load_zs(ctl2, acc1);
            const v4i* A0 = A + ((ctl * KT) << 6) + lane;
            const v4i* A1 = A + ((ctl2 * KT) << 6) + lane;
            v4i fa[3], fb[3];
            fa[0] = A0[0]; fb[0] = A1[0];
            fa[1] = A0[(KT > 1 ? 1 : 0) << 6]; fb[1] = A1[(KT > 1 ? 1 : 0) << 6];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
              const int kn = kt + 2 < KT ? kt + 2 : KT - 1;
              fa[(kt + 2) % 3] = A0[kn << 6];
              fb[(kt + 2) % 3] = A1[kn << 6];
              acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[kt % 3], bfrag[kt], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[kt % 3], bfrag[kt], acc1, 0, 0, 0);
            }
            finish(ctl, acc0);
            finish(ctl2, acc1);
          } else {
            v16i acc0;
            load_zs(ctl, acc0);
            const v4i* A0 = A + ((ctl * KT) << 6) + lane;
            v4i fa[3];
            fa[0] = A0[0];
            fa[1] = A0[(KT > 1 ? 1 : 0) << 6];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
              const int kn = kt + 2 < KT ? kt + 2 : KT - 1;
              fa[(kt + 2) % 3] = A0[kn << 6];
              acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[kt % 3], bfrag[kt], acc0, 0, 0, 0);
            }
            finish(ctl, acc0);
          }
        }
        FQ_PIN();
        if (b == b_begin && c == 0) PW_STAMP(6);
        if (g.NC > 1) {
          chunk_commit(cur ^ 1);
          __syncthreads();
        }
        if (b == b_begin && c == 0) PW_STAMP(7);
      }
      if (b == b_begin) PW_STAMP(4);
      if (has_stat) {
        const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)px.smp);
        const bool uniform = __all(!px.valid || px.smp == s0);
        if (uniform) {
          const float wm = wave_max(px.valid ? m : 0.0f);
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
    }
  };
  using std::integral_constant;
  if (bias == nullptr && has_bn && act == FQ_ACT_RELU)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU>{});
  else if (bias == nullptr && has_bn && act == FQ_ACT_RELU6)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU6>{});
  else if (bias == nullptr && has_bn && act == FQ_ACT_NONE)
    run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_NONE>{});
  else
    run(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{});

  PW_STAMP(5);
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < cols / HW)
      atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
}


}  // namespace

namespace fqi {

// chunked streaming form: weights through LDS in double-buffered chunks, K = 256 or 512
int pw_try_chunk(const PwCall& a, bool* taken) {
  *taken = false;
  const int kt = (int)((a.cin + 31) / 32);
  const int ct = (int)((a.cout + 31) / 32);
  static const int pwc_ctc = env_int("FQ_PWC_CTC", 0), pwc_wsplit = env_int("FQ_PWC_WSPLIT", 0);
  if ((a.form == 0 || a.form == 4) && a.cin % 16 == 0 && a.cout % 32 == 0 && (kt == 8 || kt == 16)) {
    PwcGeom c;
    c.Cin = (int)a.cin; c.K = (int)a.cin_pad; c.Cout = (int)a.cout; c.CT = ct; c.HW = (int)a.hw;
    c.cols = a.n * a.hw; c.tiles = (c.cols + 31) / 32; c.zoff = a.zoff;
    c.rows = (int)((a.cout + 63) / 64 * 64);
    // few tiles (7x7 planes): all four wavefronts share one tile and split a 4-tile chunk; otherwise one tile each
    c.wsplit = c.tiles < (int64_t)num_cu() * 2 ? 4 : 1;
    if (pwc_wsplit > 0) c.wsplit = pwc_wsplit;
    c.CTC = c.wsplit == 4 ? 4 : (kt == 16 ? 2 : 4);
    if (pwc_ctc > 0) c.CTC = pwc_ctc;
    c.NC = (ct + c.CTC - 1) / c.CTC;
    const int pieces = c.CTC * kt / 4;
    const bool ok = (c.NC == 1 || c.NC % 2 == 0) && ct % c.CTC == 0 && c.wsplit <= c.CTC && (pieces == 8 || pieces == 16) &&
                    (c.wsplit == 1 || c.wsplit == 2 || c.wsplit == 4);
    if (ok) {
      c.batches = (c.tiles + (4 / c.wsplit) - 1) / (4 / c.wsplit);
      const size_t lds2 = 2 * ((size_t)c.CTC * kt * 1024 + (size_t)c.CTC * 32 * 5 * sizeof(float));
      int per_cu = (int)((160 * 1024) / (lds2 + 1024));
      per_cu = per_cu > 2 ? 2 : (per_cu < 1 ? 1 : per_cu);
      int64_t grid = (int64_t)num_cu() * per_cu;
      if (grid > c.batches) grid = c.batches;
      if (int rc = pw_zero_stat(a)) return rc;
#define FQ_PWC_CASE(KT_, P_)                                                                                           \
  if (kt == KT_ && pieces == P_) {                                                                                     \
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_chunk_kernel<KT_, P_>),      \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess; \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the chunked kernel");                     \
    hipLaunchKernelGGL((pwconv_chunk_kernel<KT_, P_>), dim3((unsigned)grid), dim3(kBlock), lds2, a.st, a.x, a.wcodes,  \
                       a.wscale, (const int*)a.wsum, a.bias, a.y, c, a.in_stat, (int)a.n, a.in_thr, a.levels, a.lo_neg, \
                       kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out, (float*)a.ws);              \
  }
      FQ_PWC_CASE(8, 8) FQ_PWC_CASE(8, 16) FQ_PWC_CASE(16, 8) FQ_PWC_CASE(16, 16)
#undef FQ_PWC_CASE
      FQ_LAUNCH_CHECK();
      *taken = true;
      return FQ_OK;
    }
  }
  FQ_REQUIRE(a.form != 4, "fq_pwconv_i8: FQ_PW_FORM=4 but the shape does not fit the chunked kernel");
  return FQ_OK;
}

}  // namespace fqi
