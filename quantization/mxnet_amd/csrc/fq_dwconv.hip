// libfakequant — K2c/K2d/K2e depthwise 3x3 with quantise-on-load
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_common.h"

#ifndef FQ_DWFLAT_NTL
// the flat form's activation loads carry the nontemporal hint: its input (a 26-103 MB tensor the producer left in the Infinity
// Cache) is read once - +0.9 % images/s with three batches in flight, five alternating runs (profiles/r5_nt_sweep3.txt; 0 = off)
#define FQ_DWFLAT_NTL 1
#endif

namespace {

// Experiment hook: cap the scalar registers of the depthwise kernels (-DFQ_DW_SGPR=80).  A 256-thread workgroup is admitted
// 8 per CU only with <= 80 SGPRs (MI355X_MICROARCH.md, residency); hipcc uses up to 106 when not told otherwise.
#ifdef FQ_DW_SGPR
#define FQ_DW_ATTR __attribute__((amdgpu_num_sgpr(FQ_DW_SGPR)))
#else
#define FQ_DW_ATTR
#endif

// ---------------------------------------------------------------------------------------------------------------
// K2c: depthwise 3x3 (pad 1, stride S) with quantise-on-load and BN / activation / statistic epilogue.
// A workgroup step ("tile") is P whole planes (small planes) or a strip of output rows of one plane (large planes).
// The input rows of a tile are contiguous in memory: they are read once, coalesced (16 B per lane), quantised ONCE per
// element and staged into an LDS tile that carries a zero border, so the 9-tap loop has no bounds checks.  In the
// compute phase consecutive lanes own consecutive output COLUMNS (conflict-free LDS reads, coalesced stores) and slide
// down a segment of rows keeping the 3x3 window in registers: 3 new LDS values per output (6 for stride 2).
// ---------------------------------------------------------------------------------------------------------------
struct DwGeom {
  int C, H, W, Ho, Wo;
  int P;        // planes per tile (whole-plane mode) or 1
  int TR;       // output rows per tile
  int strips;   // tiles per plane along rows (1 in whole-plane mode)
  int IR;       // LDS rows per plane (TR*S + 2)
  int WS;       // LDS row stride (>= W + 2)
  int nseg;     // row segments per tile in the compute phase
  int RS;       // output rows per segment
  int vec_in;   // 16-byte loads allowed
};

template <int S, bool QUANT, bool ONLINE>
__global__ __launch_bounds__(kBlock) FQ_DW_ATTR void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           DwGeom g, int64_t tiles, const float* __restrict__ in_stat,
                                                           int n, const float* __restrict__ in_thr, float levels,
                                                           int lo_neg_max, float eps,
                                                           float* __restrict__ cur_max_out,
                                                           const float* __restrict__ bn_scale,
                                                           const float* __restrict__ bn_shift, int act,
                                                           float* __restrict__ stat_out) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ float red[4];
  QParams q;
  q.lo = q.hi = q.denom = q.scale = 0.0f;
  q.rden = 0.0;
  if (QUANT) {
    const float max_ = input_threshold(in_stat, n, ONLINE ? nullptr : in_thr, cur_max_out, blockIdx.x == 0);
    q = make_qparams_rt(max_, levels, lo_neg_max, eps, in_thr);
  }
  const int lds_elems = g.P * g.IR * g.WS;
  const int plane_in = g.H * g.W, plane_out = g.Ho * g.Wo;
  const bool has_bn = bn_scale != nullptr;
  const bool has_stat = stat_out != nullptr;

  const ChunkRange rg = block_range(tiles);
  for (int64_t t = rg.begin; t < rg.end; ++t) {
    const int64_t pg = t / g.strips;                   // plane group
    const int strip = (int)(t - pg * g.strips);
    const int64_t plane0 = pg * g.P;                   // first (n*C + c) plane of the tile
    const int ch0 = (int)(plane0 % g.C);               // P divides C: the tile's planes are ch0 .. ch0+P-1 of ONE sample
    const int orow0 = strip * g.TR;                    // first output row
    const int orows = (g.Ho - orow0) < g.TR ? (g.Ho - orow0) : g.TR;
    const int irow_first = orow0 * S - 1;              // input row held by LDS row 0 (may be -1)
    const int r_lo = irow_first < 0 ? 0 : irow_first;
    int r_hi = irow_first + g.IR - 1;
    if (r_hi > g.H - 1) r_hi = g.H - 1;

    __syncthreads();                                   // previous tile fully consumed
    for (int i = threadIdx.x * 4; i < lds_elems; i += kBlock * 4)
      *reinterpret_cast<f4*>(tile + i) = (f4){0.f, 0.f, 0.f, 0.f};     // (allocation is padded to a multiple of 4)
    __syncthreads();
    // ---- load + quantise + stage: rows [r_lo, r_hi] of P consecutive planes -------------------------------------
    // whole-plane mode: r_lo = 0, r_hi = H-1 and the P planes are one contiguous range; strip mode: P = 1.
    {
      const float* src = x + plane0 * (int64_t)plane_in + (int64_t)r_lo * g.W;
      const int rows_per_plane = r_hi - r_lo + 1;
      const int cnt = (g.P > 1) ? g.P * plane_in : rows_per_plane * g.W;
      const unsigned W = (unsigned)g.W, RP = (unsigned)rows_per_plane;
      if (g.vec_in) {
        const f4* p4 = reinterpret_cast<const f4*>(src);
        for (int i = threadIdx.x; i < cnt / 4; i += kBlock) {
          f4 v = p4[i];
          if (QUANT) v = fq_code4(v, q) * q.scale;
          const unsigned e = (unsigned)i * 4u;
          unsigned row = e / W;                                   // row index over the tile's planes
          unsigned col = e - row * W;
          unsigned pl = row / RP;
          unsigned r = row - pl * RP;
          float* d = tile + (pl * g.IR + (r + r_lo - irow_first)) * g.WS + col + 1;
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            d[0] = vv[k];
            ++d;
            if (++col == W) {                                     // next row (possibly next plane)
              col = 0;
              if (++r == RP) { r = 0; ++pl; }
              d = tile + (pl * g.IR + (r + r_lo - irow_first)) * g.WS + 1;
            }
          }
        }
      } else {
        for (int i = threadIdx.x; i < cnt; i += kBlock) {
          float v = src[i];
          if (QUANT) v = fq_code(v, q) * q.scale;
          const unsigned row = (unsigned)i / W;
          const unsigned col = (unsigned)i - row * W;
          const unsigned pl = row / RP;
          const unsigned r = row - pl * RP;
          tile[(pl * g.IR + (r + r_lo - irow_first)) * g.WS + col + 1] = v;
        }
      }
    }
    __syncthreads();
    // ---- sliding 3x3 window down a row segment; lane <-> output column ------------------------------------------
    float m = 0.0f;
    const int items = g.P * g.nseg * g.Wo;
    for (int it = threadIdx.x; it < items; it += kBlock) {
      const unsigned ps = (unsigned)it / (unsigned)g.Wo;
      const unsigned col = (unsigned)it - ps * (unsigned)g.Wo;
      const unsigned pl = ps / (unsigned)g.nseg;
      const unsigned seg = ps - pl * (unsigned)g.nseg;
      const int rr0 = (int)seg * g.RS;
      int rr1 = rr0 + g.RS;
      if (rr1 > orows) rr1 = orows;
      if (rr0 >= rr1) continue;
      const int ch = ch0 + (int)pl;
      const float* wk = wgt + ch * 9;
      const float w00 = wk[0], w01 = wk[1], w02 = wk[2], w10 = wk[3], w11 = wk[4], w12 = wk[5], w20 = wk[6],
                  w21 = wk[7], w22 = wk[8];
      const float bch = bias != nullptr ? bias[ch] : 0.0f;
      const float bsc = has_bn ? bn_scale[ch] : 1.0f, bsh = has_bn ? bn_shift[ch] : 0.0f;
      const float* l = tile + (pl * g.IR + rr0 * S) * g.WS + col * S;       // LDS col 0 == input col -1
      float a0 = l[0], a1 = l[1], a2 = l[2];
      float b0 = 0.f, b1 = 0.f, b2 = 0.f;
      if (S == 1) {
        b0 = l[g.WS];
        b1 = l[g.WS + 1];
        b2 = l[g.WS + 2];
      }
      float* dst = y + (plane0 + pl) * (int64_t)plane_out + (int64_t)(orow0 + rr0) * g.Wo + col;
      for (int r = rr0; r < rr1; ++r) {
        float c0, c1, c2;
        if (S == 1) {
          const float* lc = l + 2 * g.WS;
          c0 = lc[0]; c1 = lc[1]; c2 = lc[2];
        } else {
          const float* lb = l + g.WS;
          b0 = lb[0]; b1 = lb[1]; b2 = lb[2];
          const float* lc = lb + g.WS;
          c0 = lc[0]; c1 = lc[1]; c2 = lc[2];
        }
        float acc = 0.0f;
        acc = fmaf(w00, a0, acc);
        acc = fmaf(w01, a1, acc);
        acc = fmaf(w02, a2, acc);
        acc = fmaf(w10, b0, acc);
        acc = fmaf(w11, b1, acc);
        acc = fmaf(w12, b2, acc);
        acc = fmaf(w20, c0, acc);
        acc = fmaf(w21, c1, acc);
        acc = fmaf(w22, c2, acc);
        if (bias != nullptr) acc = (act & kActBiasMul) ? acc * bch : acc + bch;
        if (has_bn) {
          acc = acc * bsc;
          acc = acc + bsh;
        }
        acc = act_rt(acc, act & 15);
        *dst = acc;
        m = fmaxf(m, fabsf(acc));
        dst += g.Wo;
        l += S * g.WS;
        if (S == 1) {
          a0 = b0; a1 = b1; a2 = b2;
          b0 = c0; b1 = c1; b2 = c2;
        } else {
          a0 = c0; a1 = c1; a2 = c2;
        }
      }
    }
    if (has_stat) {
      m = block_max(m, red);
      if (threadIdx.x == 0) atomic_max_f32(stat_out + plane0 / g.C, m);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K2d: depthwise 3x3, register sliding-window form (no LDS, no barriers).  A wavefront holds `segs` independent
// SEGMENTS; a segment is `sw` adjacent output columns of one plane plus halo lanes (stride 1: one on each side, stride
// 2: one on the left).  Every lane streams ITS input column(s) top to bottom — one coalesced 4-byte load per lane per
// input row, quantised once — and gets its horizontal neighbours from the adjacent lanes with wavefront shuffles; the
// three live input rows stay in registers, loads run D rows ahead of their use.
// ---------------------------------------------------------------------------------------------------------------
struct DwColGeom {
  int C, H, W, Ho, Wo;
  int sw;       // output columns per segment
  int nsegx;    // segments per plane row
  int SEG;      // lanes per segment (sw + halo lanes)
  int segs;     // segments per wavefront
  int nts;      // cols4 form: nontemporal stores (an output the Infinity Cache cannot hold beside the other batches' tensors)
};

template <int S, bool QUANT, bool ONLINE, int EPI>
__global__ __launch_bounds__(kBlock) FQ_DW_ATTR void dwconv3x3_cols_kernel(
    const float* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
    float* __restrict__ y, DwColGeom g, int64_t total_segs, const float* __restrict__ in_stat, int n,
    const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps, float* __restrict__ cur_max_out,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act, float* __restrict__ stat_out) {
  // output rows of input kept in flight (S=2: 4 loads per row).  (16 / 8 - every row of a 14x14 plane in flight at
  // once - measured SLOWER on the same box: 38 vs 36 us per 512x14x14 layer.)
  constexpr int D = (S == 1) ? 8 : 4;
  constexpr int kStatSlots = 16;
  __shared__ unsigned k_stat[kStatSlots];
  if (threadIdx.x < kStatSlots) k_stat[threadIdx.x] = 0u;
  PW_STAMP(0);
  // The quantisation parameters are derived AFTER the first block's loads have been issued (ensure_q below): the batch
  // statistic is a dependent chain of two cold loads + an fp64 tree (~2.8 us per workgroup, tools/dw_trace.py) that
  // otherwise sits in front of the first useful load of every workgroup.
  QParams q;
  q.lo = q.hi = q.denom = q.scale = 0.0f;
  q.rden = 0.0;
  bool q_ready = !QUANT;
  auto ensure_q = [&]() __attribute__((always_inline)) {
    if (QUANT && !q_ready) {
      const float max_ = input_threshold(in_stat, n, ONLINE ? nullptr : in_thr, cur_max_out, blockIdx.x == 0);
      q = make_qparams_rt(max_, levels, lo_neg_max, eps, in_thr);
      q_ready = true;
    }
  };
  PW_STAMP(1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg_in_wave = lane / g.SEG;
  const int pos = lane - seg_in_wave * g.SEG;
  const bool lane_used = seg_in_wave < g.segs;
  const int64_t segs_per_block = (int64_t)g.segs * (kBlock / 64);
  const int64_t nblk = (total_segs + segs_per_block - 1) / segs_per_block;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const int plane_in = g.H * g.W, plane_out = g.Ho * g.Wo;

  // (Row strips per plane were tried for load balance and measured slower: every strip restarts the prefetch ring.)
  // A workgroup takes a CONTIGUOUS range of blocks (few samples -> a small LDS statistic table, one flush) and its
  // wavefronts run through them without barriers.  Index arithmetic is unsigned 32-bit (host: total_segs < 2^31): the
  // 64-bit divisions it replaces cost each block 2.3 us (tools/dw_trace.py).
  const unsigned nblk_u = (unsigned)nblk, tsegs = (unsigned)total_segs, nsegx = (unsigned)g.nsegx, C_u = (unsigned)g.C;
  const unsigned blk_begin = (unsigned)((uint64_t)nblk_u * blockIdx.x / gridDim.x);
  const unsigned blk_end = (unsigned)((uint64_t)nblk_u * (blockIdx.x + 1) / gridDim.x);
  const unsigned n_samples = tsegs / nsegx / C_u;
  unsigned s_base;
  {
    const unsigned seg0 = blk_begin * (unsigned)segs_per_block;
    s_base = (seg0 < tsegs ? seg0 : tsegs - 1) / nsegx / C_u;
  }
  __syncthreads();                                       // statistic table zeroed
  for (unsigned blk = blk_begin; blk < blk_end; ++blk) {
    const unsigned seg = blk * (unsigned)segs_per_block + (unsigned)wave * (unsigned)g.segs + (unsigned)seg_in_wave;
    const bool seg_ok = lane_used && seg < tsegs;
    const unsigned plane = seg_ok ? seg / nsegx : 0u;
    const int sx = seg_ok ? (int)(seg - plane * nsegx) : 0;
    const unsigned sample_u = plane / C_u;
    const int ch = (int)(plane - sample_u * C_u);
    const int sample = (int)sample_u;
    // column bookkeeping
    int oc, ic0;                                          // output column; first input column this lane loads
    bool is_out;
    if (S == 1) {
      ic0 = sx * g.sw + pos - 1;                          // pos 0 / sw+1 are the halo lanes
      oc = ic0;
      is_out = seg_ok && pos >= 1 && pos <= g.sw && oc < g.Wo;
    } else {
      oc = sx * g.sw + pos - 1;                           // pos 0 is the left-halo lane
      ic0 = 2 * oc;                                       // this lane loads input columns 2*oc and 2*oc + 1
      is_out = seg_ok && pos >= 1 && oc < g.Wo;
    }
    const bool ld0 = seg_ok && (S == 1 ? (ic0 >= 0 && ic0 < g.W) : (pos >= 1 && ic0 < g.W));
    const bool ld1 = seg_ok && S == 2 && (ic0 + 1 >= 0) && (ic0 + 1 < g.W);     // stride 2: second column (halo lane: col 2*sx*sw - 1)
    const float* xp = x + plane * (int64_t)plane_in + ic0;
    float* yp = y + plane * (int64_t)plane_out + oc;
    const float* wk = wgt + ch * 9;
    const float w00 = wk[0], w01 = wk[1], w02 = wk[2], w10 = wk[3], w11 = wk[4], w12 = wk[5], w20 = wk[6],
                w21 = wk[7], w22 = wk[8];
    const float bch = bias != nullptr ? bias[ch] : 0.0f;
    const float bsc = has_bn ? bn_scale[ch] : 1.0f, bsh = has_bn ? bn_shift[ch] : 0.0f;
    float m = 0.0f;
    if (blk == blk_begin) PW_STAMP(2);

    // Loads are UNCONDITIONAL from clamped (always valid) addresses and masked afterwards with a bitwise AND — a
    // `cond ? load : 0` select is turned back into a predicated load by hipcc (CodeGenPrepare sinks the load under a
    // branch), which then waits vmcnt(0) right behind it and the prefetch ring is gone.
    const int last_row = g.H - 1;
    auto keep = [](float v, bool ok) -> float { return __uint_as_float(__float_as_uint(v) & (ok ? 0xFFFFFFFFu : 0u)); };
    if (S == 1) {
      // rows: a = input row r-1, b = row r, c = row r+1 (each as left / centre / right)
      const float* xs = ld0 ? xp : x;                      // lanes with nothing to load read element 0
      auto ldrow = [&](int row) -> float {
        const int rc = row < last_row ? row : last_row;
        return keep(xs[(int64_t)rc * g.W], ld0 && row <= last_row);
      };
      auto emit = [&](int r, float c1, float& a0, float& a1, float& a2, float& b0, float& b1, float& b2) {
        if (QUANT) c1 = fq_code(c1, q) * q.scale;
        const float c0 = lane_prev(c1), c2 = lane_next(c1);
        float acc = 0.0f;
        acc = fmaf(w00, a0, acc);
        acc = fmaf(w01, a1, acc);
        acc = fmaf(w02, a2, acc);
        acc = fmaf(w10, b0, acc);
        acc = fmaf(w11, b1, acc);
        acc = fmaf(w12, b2, acc);
        acc = fmaf(w20, c0, acc);
        acc = fmaf(w21, c1, acc);
        acc = fmaf(w22, c2, acc);
        acc = dw_finish<EPI>(acc, bias != nullptr, bch, has_bn, bsc, bsh, act);
        m = fmaxf(m, keep(fabsf(acc), is_out));
        if (is_out) yp[(int64_t)r * g.Wo] = acc;
        a0 = b0; a1 = b1; a2 = b2;
        b0 = c0; b1 = c1; b2 = c2;
      };
      float raw[D];
#pragma unroll
      for (int k = 0; k < D; ++k) raw[k] = ldrow(1 + k);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      float b1 = ldrow(0);
      FQ_PIN();
      ensure_q();
      if (QUANT) b1 = fq_code(b1, q) * q.scale;
      float b0 = lane_prev(b1), b2 = lane_next(b1);
      const int rend = g.Ho;                              // same trip count for every lane of the grid
      int r0 = 0;
      for (; r0 + D <= rend; r0 += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const float c1 = raw[k];
          raw[k] = ldrow(r0 + k + 1 + D);
          emit(r0 + k, c1, a0, a1, a2, b0, b1, b2);
        }
      }
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (r0 + k < rend) emit(r0 + k, raw[k], a0, a1, a2, b0, b1, b2);
    } else {
      // stride 2: output row r uses input rows 2r-1 (a), 2r (b), 2r+1 (c); per input row: left = neighbour's 2nd
      // column, centre = own 1st column, right = own 2nd column
      const float* xs0 = ld0 ? xp : x;
      const float* xs1 = ld1 ? xp + 1 : x;
      auto ld_a = [&](int row) -> float {
        const int rc = row < last_row ? row : last_row;
        return keep(xs0[(int64_t)rc * g.W], ld0 && row <= last_row);
      };
      auto ld_b = [&](int row) -> float {
        const int rc = row < last_row ? row : last_row;
        return keep(xs1[(int64_t)rc * g.W], ld1 && row <= last_row);
      };
      auto emit2 = [&](int r, float b1, float b2, float c1, float c2, float& a0, float& a1, float& a2) {
        if (QUANT) {
          b1 = fq_code(b1, q) * q.scale;
          b2 = fq_code(b2, q) * q.scale;
          c1 = fq_code(c1, q) * q.scale;
          c2 = fq_code(c2, q) * q.scale;
        }
        const float b0 = lane_prev(b2), c0 = lane_prev(c2);
        float acc = 0.0f;
        acc = fmaf(w00, a0, acc);
        acc = fmaf(w01, a1, acc);
        acc = fmaf(w02, a2, acc);
        acc = fmaf(w10, b0, acc);
        acc = fmaf(w11, b1, acc);
        acc = fmaf(w12, b2, acc);
        acc = fmaf(w20, c0, acc);
        acc = fmaf(w21, c1, acc);
        acc = fmaf(w22, c2, acc);
        acc = dw_finish<EPI>(acc, bias != nullptr, bch, has_bn, bsc, bsh, act);
        m = fmaxf(m, keep(fabsf(acc), is_out));
        if (is_out) yp[(int64_t)r * g.Wo] = acc;
        a0 = c0; a1 = c1; a2 = c2;
      };
      float rb0[D], rb1[D], rc0[D], rc1[D];
#pragma unroll
      for (int k = 0; k < D; ++k) {
        rb0[k] = ld_a(2 * k);
        rb1[k] = ld_b(2 * k);
        rc0[k] = ld_a(2 * k + 1);
        rc1[k] = ld_b(2 * k + 1);
      }
      FQ_PIN();
      ensure_q();
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      const int rend = g.Ho;
      int r0 = 0;
      for (; r0 + D <= rend; r0 += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const float b1 = rb0[k], b2 = rb1[k], c1 = rc0[k], c2 = rc1[k];
          const int rn = r0 + k + D;
          rb0[k] = ld_a(2 * rn);
          rb1[k] = ld_b(2 * rn);
          rc0[k] = ld_a(2 * rn + 1);
          rc1[k] = ld_b(2 * rn + 1);
          emit2(r0 + k, b1, b2, c1, c2, a0, a1, a2);
        }
      }
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (r0 + k < rend) emit2(r0 + k, rb0[k], rb1[k], rc0[k], rc1[k], a0, a1, a2);
    }
    if (blk == blk_begin) PW_STAMP(3);
    if (has_stat) {
      // per-wave update of the workgroup's LDS table (no barrier inside the block loop); flushed once at the end
      const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane(sample);
      const bool wave_uniform = __all(!is_out || (unsigned)sample == s0);
      if (wave_uniform) {
        const float wm = wave_max_nonneg(is_out ? m : 0.0f);
        if (lane == 0 && __float_as_uint(wm) != 0u) {
          const unsigned slot = s0 - s_base;
          if (slot < (unsigned)kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
          else atomic_max_f32(stat_out + s0, wm);
        }
      } else if (is_out) {
        const unsigned slot = (unsigned)sample - s_base;
        if (slot < (unsigned)kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
        else atomic_max_f32(stat_out + sample, m);
      }
    }
  }
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kStatSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < (unsigned)n_samples)
      FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
  PW_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------------------
// K2o: small planes (H rows known at compile time, a plane row fits one wavefront): a wavefront holds EVERY input row of
// its planes in registers and requests the rows, weights and constants of its NEXT block before it works on the current
// one (K2d restarts its prefetch ring in every block: two exposed memory latencies + the weight gather per 14 row steps).
// These layers are bound by instruction issue, not by memory (ablation without loads and stores: 21 of 25 us on
// 512x14x14, profiles/r2_dw_planes.txt), so the form is built around the instruction count per output:
//   * CPL = 2 columns per lane where W is even: 8-byte loads and stores, and the 3x3 sums of the two outputs run as ONE
//     chain of packed fp32 FMAs (v_pk_fma_f32: two IEEE FMAs per lane per instruction, same order as the scalar chain);
//   * no halo lanes: a plane takes W / CPL lanes, its edge lanes replace the neighbour's value by 0 (one select);
//   * horizontal neighbours by DPP wave shifts (lane_prev / lane_next), epilogue fixed at compile time (EPI).
// Buffer addressing: idle lanes and the tail read 0 and store nothing through an out-of-range offset, row strides sit in
// scalar registers.
// ---------------------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 buf_ld_2f32(fq_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_st_2f32(fq_rsrc r, unsigned voff, unsigned soff, f2 v) {
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ f2 splat2(float v) { return (f2){v, v}; }

template <int EPI>
__device__ __forceinline__ f2 dw_finish2(f2 acc, bool has_bias, float bch, bool has_bn, float bsc, float bsh, int act) {
  if (EPI == kEpiRuntime) {
    acc.x = dw_finish<EPI>(acc.x, has_bias, bch, has_bn, bsc, bsh, act);
    acc.y = dw_finish<EPI>(acc.y, has_bias, bch, has_bn, bsc, bsh, act);
    return acc;
  }
  acc = acc * splat2(bsc);
  acc = acc + splat2(bsh);
  acc.x = fmaxf(acc.x, 0.0f);
  acc.y = fmaxf(acc.y, 0.0f);
  if (EPI == kEpiBnRelu6) {
    acc.x = fminf(acc.x, 6.0f);
    acc.y = fminf(acc.y, 6.0f);
  }
  return acc;
}

template <int S, int H, int CPL>
struct DwPlanesBlk {
  static constexpr int kIn = (S == 2 || CPL == 2) ? 2 : 1;     // input columns per lane
  float raw[kIn * H];      // row-major: (column 0[, column 1]) of rows 0..H-1
  float w[9];
  float bch, bsc, bsh;
};

template <int S, bool QUANT, bool ONLINE, int H, int CPL, int EPI>
__global__ __launch_bounds__(kBlock) FQ_DW_ATTR void dwconv3x3_planes_kernel(
    const float* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
    float* __restrict__ y, DwColGeom g, int64_t total_segs, const float* __restrict__ in_stat, int n,
    const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps, float* __restrict__ cur_max_out,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act, float* __restrict__ stat_out) {
  static_assert(S == 1 || CPL == 2, "stride 2 reads two input columns per lane");
  typedef DwPlanesBlk<S, H, CPL> Blk;
  constexpr int KIN = Blk::kIn;
  constexpr int HO = (H - 1) / S + 1;
  constexpr int kStatSlots = 16;
  constexpr unsigned kOob = 0x80000000u;                 // beyond every resource of this kernel (host: tensors < 2 GiB)
  __shared__ unsigned k_stat[kStatSlots];
  if (threadIdx.x < kStatSlots) k_stat[threadIdx.x] = 0u;
  PW_STAMP(0);
  const int lane = threadIdx.x & 63;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int seg_in_wave = lane / g.SEG;                   // g.SEG lanes per plane: W / CPL (stride 1), Wo (stride 2)
  const int pos = lane - seg_in_wave * g.SEG;
  const bool lane_used = seg_in_wave < g.segs;
  const bool first = pos == 0, last = pos == g.SEG - 1;   // edge lanes of a plane: no left / right neighbour
  const unsigned segs_per_block = (unsigned)g.segs * (kBlock / 64);
  const unsigned tsegs = (unsigned)total_segs, C_u = (unsigned)g.C;           // one segment per plane: tsegs planes
  const unsigned nblk = (tsegs + segs_per_block - 1) / segs_per_block;
  const unsigned blk_begin = (unsigned)((uint64_t)nblk * blockIdx.x / gridDim.x);
  const unsigned blk_end = (unsigned)((uint64_t)nblk * (blockIdx.x + 1) / gridDim.x);
  const unsigned n_samples = tsegs / C_u;
  unsigned s_base;
  {
    const unsigned seg0 = blk_begin * segs_per_block;
    s_base = (seg0 < tsegs ? seg0 : tsegs - 1) / C_u;
  }
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr, has_bias = bias != nullptr;
  const unsigned plane_in = (unsigned)(H * g.W), plane_out = (unsigned)(HO * g.Wo);
  const unsigned row_in = (unsigned)g.W * 4u, row_out = (unsigned)g.Wo * 4u;
  const fq_rsrc rx = make_rsrc(x, (int64_t)tsegs * plane_in * 4), ry = make_rsrc(y, (int64_t)tsegs * plane_out * 4);
  const fq_rsrc rw = make_rsrc(wgt, (int64_t)g.C * 36);
  const unsigned ic0 = (unsigned)pos * KIN;               // first input column of this lane
  const unsigned oc0 = S == 1 ? ic0 : (unsigned)pos;      // first output column of this lane

  auto issue = [&](unsigned blk, Blk& b) __attribute__((always_inline)) {
    const unsigned seg = blk * segs_per_block + wave * (unsigned)g.segs + (unsigned)seg_in_wave;
    const bool seg_ok = lane_used && seg < tsegs;
    const unsigned xoff = seg_ok ? (seg * plane_in + ic0) * 4u : kOob;
#pragma unroll
    for (int r = 0; r < H; ++r) {
      if (KIN == 1) {
        b.raw[r] = buf_ld_f32(rx, xoff, (unsigned)r * row_in);
      } else {
        const f2 v = buf_ld_2f32(rx, xoff, (unsigned)r * row_in);
        b.raw[2 * r] = v.x;
        b.raw[2 * r + 1] = v.y;
      }
    }
    const unsigned ch = seg_ok ? seg % C_u : 0u;
#pragma unroll
    for (int k = 0; k < 9; ++k) b.w[k] = buf_ld_f32(rw, ch * 36u, (unsigned)k * 4u);
    b.bch = has_bias ? bias[ch] : 0.0f;
    b.bsc = has_bn ? bn_scale[ch] : 1.0f;
    b.bsh = has_bn ? bn_shift[ch] : 0.0f;
  };

  Blk nxt;
  ThresholdReq treq;
  if (QUANT) treq = threshold_request(in_stat, n, ONLINE ? nullptr : in_thr, blockIdx.x == 0);   // first in the memory queue
  FQ_PIN();
  if (blk_begin < blk_end) issue(blk_begin, nxt);
  FQ_PIN();
  QParams q;
  q.lo = q.hi = q.denom = q.scale = 0.0f;
  q.rden = 0.0;
  if (QUANT) {                                            // while the first block is on its way
    const float max_ = threshold_finish(treq, in_stat, n, ONLINE ? nullptr : in_thr, cur_max_out, blockIdx.x == 0);
    q = make_qparams_rt(max_, levels, lo_neg_max, eps, in_thr);
  }
  PW_STAMP(1);
  __syncthreads();                                        // statistic table zeroed
  for (unsigned blk = blk_begin; blk < blk_end; ++blk) {
    if (blk == blk_begin + 1) PW_STAMP(2);
#ifdef FQ_PW_TRACE
    if (blk == blk_begin) {                               // trace build: separate "first block's data arrives" from "first pass through the code"
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PW_STAMP(4);
    }
#endif
    Blk cur = nxt;
    FQ_PIN();
    if (blk + 1 < blk_end) issue(blk + 1, nxt);
    FQ_PIN();
    const unsigned seg = blk * segs_per_block + wave * (unsigned)g.segs + (unsigned)seg_in_wave;
    const bool is_out = lane_used && seg < tsegs;
    const unsigned yoff = is_out ? (seg * plane_out + oc0) * 4u : kOob;
    const unsigned sample = (seg < tsegs ? seg : tsegs - 1) / C_u;
    float m = 0.0f;
    auto fq = [&](float v) -> float { return QUANT ? fq_code(v, q) * q.scale : v; };
    // the value the lane on the left / right holds.  The shift is taken by EVERY lane and masked afterwards: under a
    // branch the edge lanes would be switched off, and a DPP read from a switched-off lane returns 0
    auto left_of = [&](float v) -> float {
      const float t = lane_prev(v);
      return first ? 0.0f : t;
    };
    auto right_of = [&](float v) -> float {
      const float t = lane_next(v);
      return last ? 0.0f : t;
    };
    if (S == 1 && CPL == 2) {
      // two outputs per lane: (L, q0, q1) and (q0, q1, R) per input row, summed as one packed chain
      f2 aL = {0.f, 0.f}, aC = {0.f, 0.f}, aR = {0.f, 0.f};     // row r-1: (L, q0), (q0, q1), (q1, R)
      f2 bL, bC, bR;
      {
        const float q0 = fq(cur.raw[0]), q1 = fq(cur.raw[1]);
        bL = (f2){left_of(q1), q0};
        bC = (f2){q0, q1};
        bR = (f2){q1, right_of(q0)};
      }
#pragma unroll
      for (int r = 0; r < H; ++r) {
        f2 cL = {0.f, 0.f}, cC = {0.f, 0.f}, cR = {0.f, 0.f};
        if (r + 1 < H) {
          const float q0 = fq(cur.raw[2 * r + 2]), q1 = fq(cur.raw[2 * r + 3]);
          cL = (f2){left_of(q1), q0};
          cC = (f2){q0, q1};
          cR = (f2){q1, right_of(q0)};
        }
        f2 acc = {0.f, 0.f};
        acc = __builtin_elementwise_fma(splat2(cur.w[0]), aL, acc);
        acc = __builtin_elementwise_fma(splat2(cur.w[1]), aC, acc);
        acc = __builtin_elementwise_fma(splat2(cur.w[2]), aR, acc);
        acc = __builtin_elementwise_fma(splat2(cur.w[3]), bL, acc);
        acc = __builtin_elementwise_fma(splat2(cur.w[4]), bC, acc);
        acc = __builtin_elementwise_fma(splat2(cur.w[5]), bR, acc);
        acc = __builtin_elementwise_fma(splat2(cur.w[6]), cL, acc);
        acc = __builtin_elementwise_fma(splat2(cur.w[7]), cC, acc);
        acc = __builtin_elementwise_fma(splat2(cur.w[8]), cR, acc);
        acc = dw_finish2<EPI>(acc, has_bias, cur.bch, has_bn, cur.bsc, cur.bsh, act);
        m = fmaxf(m, fmaxf(fabsf(acc.x), fabsf(acc.y)));
        buf_st_2f32(ry, yoff, (unsigned)r * row_out, acc);
        aL = bL; aC = bC; aR = bR;
        bL = cL; bC = cC; bR = cR;
      }
    } else {
      auto finish = [&](int r, float acc) __attribute__((always_inline)) {
        acc = dw_finish<EPI>(acc, has_bias, cur.bch, has_bn, cur.bsc, cur.bsh, act);
        m = fmaxf(m, fabsf(acc));
        buf_st_f32(ry, yoff, (unsigned)r * row_out, acc);
      };
      auto window = [&](float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) -> float {
        float acc = 0.0f;
        acc = fmaf(cur.w[0], a0, acc);
        acc = fmaf(cur.w[1], a1, acc);
        acc = fmaf(cur.w[2], a2, acc);
        acc = fmaf(cur.w[3], b0, acc);
        acc = fmaf(cur.w[4], b1, acc);
        acc = fmaf(cur.w[5], b2, acc);
        acc = fmaf(cur.w[6], c0, acc);
        acc = fmaf(cur.w[7], c1, acc);
        acc = fmaf(cur.w[8], c2, acc);
        return acc;
      };
      if (S == 1) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        float b1 = fq(cur.raw[0]);
        float b0 = left_of(b1), b2 = right_of(b1);
#pragma unroll
        for (int r = 0; r < H; ++r) {
          float c0 = 0.f, c1 = 0.f, c2 = 0.f;
          if (r + 1 < H) {
            c1 = fq(cur.raw[r + 1]);
            c0 = left_of(c1);
            c2 = right_of(c1);
          }
          finish(r, window(a0, a1, a2, b0, b1, b2, c0, c1, c2));
          a0 = b0; a1 = b1; a2 = b2;
          b0 = c0; b1 = c1; b2 = c2;
        }
      } else {
        // output row r: input rows 2r-1 (a), 2r (b), 2r+1 (c); per row: left = the left lane's odd column, centre / right own
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int r = 0; r < HO; ++r) {
          const float b1 = fq(cur.raw[4 * r]), b2 = fq(cur.raw[4 * r + 1]);
          const float b0 = left_of(b2);
          float c0 = 0.f, c1 = 0.f, c2 = 0.f;
          if (2 * r + 1 < H) {
            c1 = fq(cur.raw[4 * r + 2]);
            c2 = fq(cur.raw[4 * r + 3]);
            c0 = left_of(c2);
          }
          finish(r, window(a0, a1, a2, b0, b1, b2, c0, c1, c2));
          a0 = c0; a1 = c1; a2 = c2;
        }
      }
    }
    if (has_stat) {
      m = is_out ? m : 0.0f;
      const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)sample);
      const bool wave_uniform = __all(!is_out || sample == s0);
      if (wave_uniform) {
        const float wm = wave_max_nonneg(m);
        if (lane == 0 && __float_as_uint(wm) != 0u) {
          const unsigned slot = s0 - s_base;
          if (slot < (unsigned)kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
          else atomic_max_f32(stat_out + s0, wm);
        }
      } else if (is_out) {
        const unsigned slot = sample - s_base;
        if (slot < (unsigned)kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
        else atomic_max_f32(stat_out + sample, m);
      }
    }
  }
  PW_STAMP(3);
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kStatSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < n_samples)
      FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
  PW_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------------------
// K2p: small planes through an LDS transpose.  The column-walking forms above read and write a plane row by row: every
// memory instruction of a wavefront touches 56-byte pieces of 4-9 different planes (9-18 cache lines, each of them again
// by the next two rows), and the kernel is bound by the rate at which the CU's memory pipeline takes such instructions
// (tools/dw_trace.py: 26 loads per wavefront take 2.5 us to ISSUE when 12 wavefronts per CU do it together; no loads /
// no stores ablation: 21 of 25 us remain).  Here a wavefront moves its P planes (one contiguous P * H * W * 4 byte range)
// with 16 bytes per lane on consecutive addresses - every cache line is touched once, by one instruction:
//   A  flat range -> registers (requested ONE BLOCK AHEAD) -> quantise (no neighbours needed) -> wavefront-private LDS tile
//   B  lane = (plane, column pair): walk down the rows reading the tile (8 bytes per lane and row), 3x3 sums as packed
//      fp32 FMA chains, epilogue, result written over the tile row just consumed
//   C  tile -> registers -> flat 16-byte stores
// No workgroup barrier: a wavefront only ever touches its own tile (LDS operations of a wavefront complete in order).
// ---------------------------------------------------------------------------------------------------------------
// Phase boundary of the flat form.  Default: compiler-level ordering only (LDS operations of one wavefront execute in
// order).  tools/dw_race_repro.py builds the alternatives to bisect the 28x28 stride-2 irreproducibility:
//   -DFQ_DWF_SYNC   a workgroup barrier;  -DFQ_DWF_DRAIN  s_waitcnt vmcnt(0) lgkmcnt(0) before going on
#if defined(FQ_DWF_SYNC)
#define FQ_DWF_PHASE() __syncthreads()
#elif defined(FQ_DWF_DRAIN)
#define FQ_DWF_PHASE()                                             \
  do {                                                             \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_wave_barrier();                               \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");         \
  } while (0)
#else
#define FQ_DWF_PHASE()                                             \
  do {                                                             \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         \
    __builtin_amdgcn_wave_barrier();                               \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");         \
  } while (0)
#endif

struct DwFlatGeom {
  int C;
  unsigned planes;          // n * c
  unsigned per, rem;        // workgroup b works on blocks [b * per + min(b, rem), ...) - per + (b < rem) of them
  FastDiv by_c;             // plane % C, plane / C
};

template <int S, bool QUANT, bool ONLINE, int H, int W, int EPI>
__global__ __launch_bounds__(kBlock) void dwconv3x3_flat_kernel(
    const float* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
    float* __restrict__ y, DwFlatGeom g, const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr,
    float levels, int lo_neg_max, float eps, float* __restrict__ cur_max_out, const float* __restrict__ bn_scale,
    const float* __restrict__ bn_shift, int act, float* __restrict__ stat_out) {
  // phase B shapes: PAIR - stride 1, even W: two columns per lane, packed FMA chains, in place;
  //                 ONE  - stride 1, odd W: one column per lane, in place;  DOWN - stride 2 (even W): one OUTPUT column per lane
  constexpr bool PAIR = S == 1 && W % 2 == 0, DOWN = S == 2;
  static_assert(S == 1 || W % 2 == 0, "stride 2 reads column pairs");
  constexpr int HO = (H - 1) / S + 1, WO = (W - 1) / S + 1;
  constexpr int IN = H * W, OUT = HO * WO;           // floats per plane
  constexpr int LPP = DOWN ? WO : (PAIR ? W / 2 : W);      // lanes per plane in phase B
  constexpr int PMAX = 64 / LPP;
  // planes per wavefront and block: the flat ranges must be whole 16-byte groups (planes of 49 floats: multiples of 4 planes)
  constexpr int P = (IN % 4 == 0 && OUT % 4 == 0) ? PMAX : PMAX / 4 * 4;
  static_assert(P >= 1 && (P * IN) % 4 == 0 && (P * OUT) % 4 == 0, "no 16-byte flat range for this plane size");
  constexpr bool TAIL_OK = IN % 4 == 0 && OUT % 4 == 0;  // else the host only takes tensors of whole blocks (planes % P == 0)
  constexpr int NFI = P * IN / 4, NFO = P * OUT / 4;   // 16-byte groups per wavefront and block, in / out
  constexpr int NLI = (NFI + 63) / 64, NLO = (NFO + 63) / 64;
  constexpr int kStatSlots = 16;
  constexpr unsigned kOob = 0x80000000u;   // beyond every resource of this kernel (host: tensors < 2 GiB)
  __shared__ f4 tile[kBlock / 64][NLI * 64];
  __shared__ f4 otile[kBlock / 64][DOWN ? NLO * 64 : 1];  // stride 1 writes its results over the tile rows already consumed
  __shared__ unsigned k_stat[kStatSlots];
  if (threadIdx.x < kStatSlots) k_stat[threadIdx.x] = 0u;
#ifdef FQ_DWF_PADLDS                                      // bisect: more static LDS = fewer workgroups per CU
  __shared__ unsigned lds_pad[FQ_DWF_PADLDS / 4];
  if (threadIdx.x == 0 && n < 0) lds_pad[n & 15] = 1u;
#endif
  PW_STAMP(0);
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned b = blockIdx.x;
  const unsigned blk_begin = b * g.per + (b < g.rem ? b : g.rem);
  const unsigned blk_end = blk_begin + g.per + (b < g.rem ? 1u : 0u);
  const unsigned planes = g.planes;
  const unsigned n_samples = (unsigned)n;
  unsigned s_base;
  {
    const unsigned p0 = blk_begin * (P * (kBlock / 64));
    s_base = fast_div(p0 < planes ? p0 : planes - 1, g.by_c);
  }
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr, has_bias = bias != nullptr;
  const fq_rsrc rx = make_rsrc(x, (int64_t)planes * IN * 4), ry = make_rsrc(y, (int64_t)planes * OUT * 4);
  const fq_rsrc rw = make_rsrc(wgt, (int64_t)g.C * 36);
  // phase B coordinates of this lane
  const unsigned j = lane / LPP, pos = lane - j * LPP;
  const bool b_lane = lane < P * LPP;
  const bool first = pos == 0, last = pos == LPP - 1;
  float* const my_tile = reinterpret_cast<float*>(tile[wave]);
  float* const my_out = DOWN ? reinterpret_cast<float*>(otile[wave]) : my_tile;

  struct Blk {
    f4 raw[NLI];
    f4 w03, w47;
    float w8, bch, bsc, bsh;
  };
  auto issue = [&](unsigned blk, Blk& k) __attribute__((always_inline)) {
    const unsigned base = (blk * (kBlock / 64) + wave) * P;                  // first plane of this wavefront (uniform)
    const unsigned left = base < planes ? planes - base : 0u;
    const unsigned np = left < (unsigned)P ? left : (unsigned)P;
    const unsigned lim = TAIL_OK ? np * (IN / 4) : (np == (unsigned)P ? (unsigned)NFI : 0u);   // 16-byte groups that exist
#pragma unroll
    for (int i = 0; i < NLI; ++i)
      k.raw[i] = FQ_DWFLAT_NTL ? buf_ld_v4f_nt(rx, (lane + 64u * i) < lim ? lane * 16u : kOob, base * (IN * 4u) + 1024u * i)
                               : buf_ld_v4f(rx, (lane + 64u * i) < lim ? lane * 16u : kOob, base * (IN * 4u) + 1024u * i);
    const unsigned ch = (b_lane && j < left) ? fast_mod(base + j, g.by_c) : 0u;
    k.w03 = buf_ld_v4f(rw, ch * 36u, 0);
    k.w47 = buf_ld_v4f(rw, ch * 36u, 16);
    k.w8 = buf_ld_f32(rw, ch * 36u, 32);
    k.bch = has_bias ? bias[ch] : 0.0f;
    k.bsc = has_bn ? bn_scale[ch] : 1.0f;
    k.bsh = has_bn ? bn_shift[ch] : 0.0f;
  };

  Blk nxt;
  ThresholdReq treq;
  if (QUANT) treq = threshold_request(in_stat, n, ONLINE ? nullptr : in_thr, blockIdx.x == 0);   // first in the memory queue
  FQ_PIN();
  PW_STAMP(6);
  if (blk_begin < blk_end) issue(blk_begin, nxt);
  FQ_PIN();
  PW_STAMP(7);
  QParams q;
  q.lo = q.hi = q.denom = q.scale = 0.0f;
  q.rden = 0.0;
  if (QUANT) {                                            // while the first block is on its way
    const float max_ = threshold_finish(treq, in_stat, n, ONLINE ? nullptr : in_thr, cur_max_out, blockIdx.x == 0);
    q = make_qparams_rt(max_, levels, lo_neg_max, eps, in_thr);
  }
  PW_STAMP(1);
  __syncthreads();                                        // statistic table zeroed
  for (unsigned blk = blk_begin; blk < blk_end; ++blk) {
    if (blk == blk_begin + 1) PW_STAMP(2);
#ifdef FQ_DWF_NOPREFETCH                                  // bisect: no block requested ahead
    if (blk != blk_begin) issue(blk, nxt);
    FQ_PIN();
    Blk cur = nxt;
#else
    Blk cur = nxt;
    FQ_PIN();
    if (blk + 1 < blk_end) issue(blk + 1, nxt);
#endif
    FQ_PIN();
    const unsigned base = (blk * (kBlock / 64) + wave) * P;
    const unsigned left = base < planes ? planes - base : 0u;
    const unsigned np = left < (unsigned)P ? left : (unsigned)P;
    const unsigned lim_out = TAIL_OK ? np * (OUT / 4) : (np == (unsigned)P ? (unsigned)NFO : 0u);
    // ---- A: quantise in flat order, stage ---------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NLI; ++i) {
      f4 v = cur.raw[i];
      if (QUANT) v = fq_code4(v, q) * q.scale;
      tile[wave][lane + 64 * i] = v;
    }
    FQ_DWF_PHASE();
    // ---- B: 3x3 on the tile ----------------------------------------------------------------------------------------------
    float m = 0.0f;
    const bool is_out = b_lane && j < left;
    if (b_lane) {
#ifdef FQ_DWF_NODPP                                       // bisect: ds_bpermute shuffles instead of DPP wave shifts
      auto left_of = [&](float v) -> float {
        const float t = __shfl_up(v, 1);
        return first ? 0.0f : t;
      };
      auto right_of = [&](float v) -> float {
        const float t = __shfl_down(v, 1);
        return last ? 0.0f : t;
      };
#else
      auto left_of = [&](float v) -> float {              // (every lane takes the shift; the edge lanes drop it afterwards)
        const float t = lane_prev(v);
        return first ? 0.0f : t;
      };
      auto right_of = [&](float v) -> float {
        const float t = lane_next(v);
        return last ? 0.0f : t;
      };
#endif
      if (PAIR) {
        float* lp = my_tile + j * IN + pos * 2;
        const f2 w0 = splat2(cur.w03.x), w1 = splat2(cur.w03.y), w2 = splat2(cur.w03.z), w3 = splat2(cur.w03.w),
                 w4 = splat2(cur.w47.x), w5 = splat2(cur.w47.y), w6 = splat2(cur.w47.z), w7 = splat2(cur.w47.w),
                 w8 = splat2(cur.w8);
        f2 aL = {0.f, 0.f}, aC = {0.f, 0.f}, aR = {0.f, 0.f};     // row r-1: (L, q0), (q0, q1), (q1, R)
        f2 bL, bC, bR;
        {
          const f2 v = *reinterpret_cast<const f2*>(lp);
          bL = (f2){left_of(v.y), v.x};
          bC = v;
          bR = (f2){v.y, right_of(v.x)};
        }
#pragma unroll
        for (int r = 0; r < H; ++r) {
          f2 cL = {0.f, 0.f}, cC = {0.f, 0.f}, cR = {0.f, 0.f};
          if (r + 1 < H) {
            const f2 v = *reinterpret_cast<const f2*>(lp + (r + 1) * W);
            cL = (f2){left_of(v.y), v.x};
            cC = v;
            cR = (f2){v.y, right_of(v.x)};
          }
          f2 acc = {0.f, 0.f};
          acc = __builtin_elementwise_fma(w0, aL, acc);
          acc = __builtin_elementwise_fma(w1, aC, acc);
          acc = __builtin_elementwise_fma(w2, aR, acc);
          acc = __builtin_elementwise_fma(w3, bL, acc);
          acc = __builtin_elementwise_fma(w4, bC, acc);
          acc = __builtin_elementwise_fma(w5, bR, acc);
          acc = __builtin_elementwise_fma(w6, cL, acc);
          acc = __builtin_elementwise_fma(w7, cC, acc);
          acc = __builtin_elementwise_fma(w8, cR, acc);
          acc = dw_finish2<EPI>(acc, has_bias, cur.bch, has_bn, cur.bsc, cur.bsh, act);
          m = fmaxf(m, fmaxf(fabsf(acc.x), fabsf(acc.y)));
          *reinterpret_cast<f2*>(lp + r * W) = acc;          // row r of the tile was read in the previous step
          aL = bL; aC = bC; aR = bR;
          bL = cL; bC = cC; bR = cR;
        }
      } else {
        auto window = [&](float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) -> float {
          float acc = 0.0f;
          acc = fmaf(cur.w03.x, a0, acc);
          acc = fmaf(cur.w03.y, a1, acc);
          acc = fmaf(cur.w03.z, a2, acc);
          acc = fmaf(cur.w03.w, b0, acc);
          acc = fmaf(cur.w47.x, b1, acc);
          acc = fmaf(cur.w47.y, b2, acc);
          acc = fmaf(cur.w47.z, c0, acc);
          acc = fmaf(cur.w47.w, c1, acc);
          acc = fmaf(cur.w8, c2, acc);
          acc = dw_finish<EPI>(acc, has_bias, cur.bch, has_bn, cur.bsc, cur.bsh, act);
          m = fmaxf(m, fabsf(acc));
          return acc;
        };
        if (!DOWN) {
          float* lp = my_tile + j * IN + pos;
          float a0 = 0.f, a1 = 0.f, a2 = 0.f;
          float b1 = lp[0];
          float b0 = left_of(b1), b2 = right_of(b1);
#pragma unroll
          for (int r = 0; r < H; ++r) {
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if (r + 1 < H) {
              c1 = lp[(r + 1) * W];
              c0 = left_of(c1);
              c2 = right_of(c1);
            }
            lp[r * W] = window(a0, a1, a2, b0, b1, b2, c0, c1, c2);
            a0 = b0; a1 = b1; a2 = b2;
            b0 = c0; b1 = c1; b2 = c2;
          }
        } else {
          // output row r: input rows 2r-1 (a), 2r (b), 2r+1 (c); per row: left = the left lane's odd column, centre / right own
          const float* lp = my_tile + j * IN + pos * 2;
          float* op = my_out + j * OUT + pos;
          float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
          for (int r = 0; r < HO; ++r) {
            const f2 vb = *reinterpret_cast<const f2*>(lp + (2 * r) * W);
            const float b0 = left_of(vb.y);
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if (2 * r + 1 < H) {
              const f2 vc = *reinterpret_cast<const f2*>(lp + (2 * r + 1) * W);
              c1 = vc.x;
              c2 = vc.y;
              c0 = left_of(vc.y);
            }
            op[r * WO] = window(a0, a1, a2, b0, vb.x, vb.y, c0, c1, c2);
            a0 = c0; a1 = c1; a2 = c2;
          }
        }
      }
    }
    FQ_DWF_PHASE();
    // ---- C: flat stores ---------------------------------------------------------------------------------------------------
    {
      f4 ov[NLO];
#pragma unroll
      for (int i = 0; i < NLO; ++i) ov[i] = DOWN ? otile[wave][lane + 64 * i] : tile[wave][lane + 64 * i];
#pragma unroll
      for (int i = 0; i < NLO; ++i)
        buf_st_v4f_unguarded(ry, (lane + 64u * i) < lim_out ? lane * 16u : kOob, base * (OUT * 4u) + 1024u * i, ov[i]);
      hold_store_data(ov);              // fq_common.h: nothing may write a store's data registers right behind it
    }
    if (has_stat) {
      m = is_out ? m : 0.0f;
      const unsigned sample = fast_div(base + (is_out ? j : 0u), g.by_c);
      const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)sample);
      const bool wave_uniform = __all(!is_out || sample == s0);
      if (wave_uniform) {
        const float wm = wave_max_nonneg(m);
        if (lane == 0 && __float_as_uint(wm) != 0u) {
          const unsigned slot = s0 - s_base;
          if (slot < (unsigned)kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
          else atomic_max_f32(stat_out + s0, wm);
        }
      } else if (is_out) {
        const unsigned slot = sample - s_base;
        if (slot < (unsigned)kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
        else atomic_max_f32(stat_out + sample, m);
      }
    }
#if defined(FQ_DWF_SYNC) || defined(FQ_DWF_DRAIN)
    FQ_DWF_PHASE();
#else
    __builtin_amdgcn_wave_barrier();                      // (the next block's phase A overwrites the tile)
#endif
  }
  PW_STAMP(3);
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kStatSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < n_samples)
      FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
  PW_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------------------
// K2e: the same sliding window with FOUR input columns per lane (16-byte loads, 16-byte stores for stride 1 / 8-byte
// for stride 2) — needs W % 4 == 0.  With 4-byte accesses the kernel above cannot keep enough bytes in flight
// (PMC: 38 % of wave cycles parked on vmcnt at 4.1 TB/s); this form has 4x the bytes per outstanding load.
// Lane p of a segment: p = 0 left-halo lane, 1..L compute lanes (input columns 4(p-1)..4(p-1)+3), L+1 right-halo lane
// (stride 1 only).  Halo lanes load and quantise like the others; their neighbours pick up .w / .x by shuffle.
// ---------------------------------------------------------------------------------------------------------------
// NT: nontemporal LOADS - an input beyond the 256 MB Infinity Cache is dead after this pass and should not displace the
// output, which the consumer does find there (measured in the model: 64 @112x112 stride 2, 411 MB in / 103 MB out, 110 -> 99 us;
// with nontemporal stores too, or on the 205 MB inputs of the stride-1 layers, the step gets slower)
template <int S, bool QUANT, bool ONLINE, bool NT, int EPI>
__global__ __launch_bounds__(kBlock) FQ_DW_ATTR void dwconv3x3_cols4_kernel(
    const float* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
    float* __restrict__ y, DwColGeom g, int64_t total_segs, const float* __restrict__ in_stat, int n,
    const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps, float* __restrict__ cur_max_out,
    const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act, float* __restrict__ stat_out) {
  // Input rows are fetched in BURSTS of D rows (double buffered): the D loads of a burst leave the wave back to back
  // and hit the same DRAM pages; one load per step (a ring) spreads them ~700 cycles apart, and with thousands of
  // waves each streaming its own plane every access then opens a new page.
#ifndef FQ_DW4_D1
#define FQ_DW4_D1 4
#define FQ_DW4_D2 2
#endif
  constexpr int D = (S == 1) ? FQ_DW4_D1 : FQ_DW4_D2;
  constexpr int kStatSlots = 16;
  __shared__ unsigned k_stat[kStatSlots];
  if (threadIdx.x < kStatSlots) k_stat[threadIdx.x] = 0u;
  // (deriving q after the first block's loads, as K2d does, was measured 3-5 % SLOWER here: the row bursts of these large
  // planes already hide the prologue, and the extra live state costs registers)
  QParams q;
  q.lo = q.hi = q.denom = q.scale = 0.0f;
  q.rden = 0.0;
  if (QUANT) {
    const float max_ = input_threshold(in_stat, n, ONLINE ? nullptr : in_thr, cur_max_out, blockIdx.x == 0);
    q = make_qparams_rt(max_, levels, lo_neg_max, eps, in_thr);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg_in_wave = lane / g.SEG;
  const int pos = lane - seg_in_wave * g.SEG;
  const bool lane_used = seg_in_wave < g.segs;
  const int64_t segs_per_block = (int64_t)g.segs * (kBlock / 64);
  const int64_t nblk = (total_segs + segs_per_block - 1) / segs_per_block;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const int plane_in = g.H * g.W, plane_out = g.Ho * g.Wo;
  const int last_row = g.H - 1;
  auto keep4 = [](f4 v, bool ok) -> f4 {
    const unsigned mk = ok ? 0xFFFFFFFFu : 0u;
    f4 r;
    r.x = __uint_as_float(__float_as_uint(v.x) & mk);
    r.y = __uint_as_float(__float_as_uint(v.y) & mk);
    r.z = __uint_as_float(__float_as_uint(v.z) & mk);
    r.w = __uint_as_float(__float_as_uint(v.w) & mk);
    return r;
  };
  auto keep = [](float v, bool ok) -> float { return __uint_as_float(__float_as_uint(v) & (ok ? 0xFFFFFFFFu : 0u)); };

  // A workgroup takes a CONTIGUOUS range of blocks (few samples -> a small LDS statistic table, one flush) and its
  // wavefronts run through them without barriers.  Index arithmetic is unsigned 32-bit (host: total_segs < 2^31): the
  // 64-bit divisions it replaces cost each block 2.3 us (tools/dw_trace.py).
  const unsigned nblk_u = (unsigned)nblk, tsegs = (unsigned)total_segs, nsegx = (unsigned)g.nsegx, C_u = (unsigned)g.C;
  const unsigned blk_begin = (unsigned)((uint64_t)nblk_u * blockIdx.x / gridDim.x);
  const unsigned blk_end = (unsigned)((uint64_t)nblk_u * (blockIdx.x + 1) / gridDim.x);
  const unsigned n_samples = tsegs / nsegx / C_u;
  unsigned s_base;
  {
    const unsigned seg0 = blk_begin * (unsigned)segs_per_block;
    s_base = (seg0 < tsegs ? seg0 : tsegs - 1) / nsegx / C_u;
  }
  __syncthreads();                                       // statistic table zeroed
  for (unsigned blk = blk_begin; blk < blk_end; ++blk) {
    const unsigned seg = blk * (unsigned)segs_per_block + (unsigned)wave * (unsigned)g.segs + (unsigned)seg_in_wave;
    const bool seg_ok = lane_used && seg < tsegs;
    const unsigned plane = seg_ok ? seg / nsegx : 0u;
    const int sx = seg_ok ? (int)(seg - plane * nsegx) : 0;
    const unsigned sample_u = plane / C_u;
    const int ch = (int)(plane - sample_u * C_u);
    const int sample = (int)sample_u;
    const int ic0 = (sx * g.sw + pos - 1) * 4;            // first of the 4 input columns this lane loads
    const bool ld_ok = seg_ok && ic0 >= 0 && ic0 < g.W;
    const int oc = S == 1 ? ic0 : ic0 / 2;                // first output column (4 outputs for S=1, 2 for S=2)
    const bool is_out = seg_ok && pos >= 1 && pos <= g.sw && oc < g.Wo;
    const f4* xs = reinterpret_cast<const f4*>(ld_ok ? x + plane * (int64_t)plane_in + ic0 : x);
    float* yp = y + plane * (int64_t)plane_out + oc;
    const int rowq = g.W / 4;                             // f4 per input row
    const float* wk = wgt + ch * 9;
    const float w00 = wk[0], w01 = wk[1], w02 = wk[2], w10 = wk[3], w11 = wk[4], w12 = wk[5], w20 = wk[6],
                w21 = wk[7], w22 = wk[8];
    const float bch = bias != nullptr ? bias[ch] : 0.0f;
    const float bsc = has_bn ? bn_scale[ch] : 1.0f, bsh = has_bn ? bn_shift[ch] : 0.0f;
    float m = 0.0f;
    auto ldrow = [&](int row) -> f4 {
      const int rc = row < last_row ? row : last_row;
      return keep4(ld4<NT>(xs + (int64_t)rc * rowq), ld_ok && row <= last_row);
    };
    auto quant4 = [&](f4 v) -> f4 { return QUANT ? fq_code4(v, q) * q.scale : v; };
    auto finish = [&](float acc) -> float { return dw_finish<EPI>(acc, bias != nullptr, bch, has_bn, bsc, bsh, act); };

    if (S == 1) {
      // a, b, c: rows r-1, r, r+1 as (left, v.x, v.y, v.z, v.w, right)
      float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b[6];
      f4 raw[D];
#pragma unroll
      for (int k = 0; k < D; ++k) raw[k] = ldrow(1 + k);
      {
        const f4 v = quant4(ldrow(0));
        b[0] = lane_prev(v.w);
        b[1] = v.x; b[2] = v.y; b[3] = v.z; b[4] = v.w;
        b[5] = lane_next(v.x);
      }
      auto emit = [&](int r, f4 craw) {
        const f4 v = quant4(craw);
        float c[6];
        c[0] = lane_prev(v.w);
        c[1] = v.x; c[2] = v.y; c[3] = v.z; c[4] = v.w;
        c[5] = lane_next(v.x);
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float acc = 0.0f;
          acc = fmaf(w00, a[k], acc);
          acc = fmaf(w01, a[k + 1], acc);
          acc = fmaf(w02, a[k + 2], acc);
          acc = fmaf(w10, b[k], acc);
          acc = fmaf(w11, b[k + 1], acc);
          acc = fmaf(w12, b[k + 2], acc);
          acc = fmaf(w20, c[k], acc);
          acc = fmaf(w21, c[k + 1], acc);
          acc = fmaf(w22, c[k + 2], acc);
          o[k] = finish(acc);
        }
        const float mm = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
        m = fmaxf(m, keep(mm, is_out));
        if (is_out) {
          if (g.nts) __builtin_nontemporal_store((f4){o[0], o[1], o[2], o[3]}, reinterpret_cast<f4*>(yp + (int64_t)r * g.Wo));
          else *reinterpret_cast<f4*>(yp + (int64_t)r * g.Wo) = (f4){o[0], o[1], o[2], o[3]};
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          a[k] = b[k];
          b[k] = c[k];
        }
      };
      int r0 = 0;
      for (; r0 + D <= g.Ho; r0 += D) {
        f4 nxt[D];
#pragma unroll
        for (int k = 0; k < D; ++k) nxt[k] = ldrow(r0 + k + 1 + D);
#pragma unroll
        for (int k = 0; k < D; ++k) emit(r0 + k, raw[k]);
#pragma unroll
        for (int k = 0; k < D; ++k) raw[k] = nxt[k];
      }
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (r0 + k < g.Ho) emit(r0 + k, raw[k]);
    } else {
      // stride 2: lane holds input columns 4j..4j+3 -> outputs 2j (cols 4j-1,4j,4j+1) and 2j+1 (cols 4j+1..4j+3)
      float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};              // (left, x, y, z, w) of input row 2r-1
      f4 rb[D], rc[D];
#pragma unroll
      for (int k = 0; k < D; ++k) {
        rb[k] = ldrow(2 * k);
        rc[k] = ldrow(2 * k + 1);
      }
      auto emit2 = [&](int r, f4 braw, f4 craw) {
        const f4 vb = quant4(braw), vc = quant4(craw);
        const float b[5] = {lane_prev(vb.w), vb.x, vb.y, vb.z, vb.w};
        const float c[5] = {lane_prev(vc.w), vc.x, vc.y, vc.z, vc.w};
        float o[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          float acc = 0.0f;
          acc = fmaf(w00, a[2 * k], acc);
          acc = fmaf(w01, a[2 * k + 1], acc);
          acc = fmaf(w02, a[2 * k + 2], acc);
          acc = fmaf(w10, b[2 * k], acc);
          acc = fmaf(w11, b[2 * k + 1], acc);
          acc = fmaf(w12, b[2 * k + 2], acc);
          acc = fmaf(w20, c[2 * k], acc);
          acc = fmaf(w21, c[2 * k + 1], acc);
          acc = fmaf(w22, c[2 * k + 2], acc);
          o[k] = finish(acc);
        }
        m = fmaxf(m, keep(fmaxf(fabsf(o[0]), fabsf(o[1])), is_out));
        if (is_out) {
          typedef float f2v __attribute__((ext_vector_type(2)));
          if (g.nts) __builtin_nontemporal_store((f2v){o[0], o[1]}, reinterpret_cast<f2v*>(yp + (int64_t)r * g.Wo));
          else *reinterpret_cast<float2*>(yp + (int64_t)r * g.Wo) = make_float2(o[0], o[1]);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) a[k] = c[k];
      };
      int r0 = 0;
      for (; r0 + D <= g.Ho; r0 += D) {
        f4 nb[D], nc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
          nb[k] = ldrow(2 * (r0 + k + D));
          nc[k] = ldrow(2 * (r0 + k + D) + 1);
        }
#pragma unroll
        for (int k = 0; k < D; ++k) emit2(r0 + k, rb[k], rc[k]);
#pragma unroll
        for (int k = 0; k < D; ++k) {
          rb[k] = nb[k];
          rc[k] = nc[k];
        }
      }
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (r0 + k < g.Ho) emit2(r0 + k, rb[k], rc[k]);
    }
    if (has_stat) {
      // per-wave update of the workgroup's LDS table (no barrier inside the block loop); flushed once at the end
      const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane(sample);
      const bool wave_uniform = __all(!is_out || (unsigned)sample == s0);
      if (wave_uniform) {
        const float wm = wave_max_nonneg(is_out ? m : 0.0f);
        if (lane == 0 && __float_as_uint(wm) != 0u) {
          const unsigned slot = s0 - s_base;
          if (slot < (unsigned)kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(wm));
          else atomic_max_f32(stat_out + s0, wm);
        }
      } else if (is_out) {
        const unsigned slot = (unsigned)sample - s_base;
        if (slot < (unsigned)kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(m));
        else atomic_max_f32(stat_out + sample, m);
      }
    }
  }
  if (has_stat) {
    __syncthreads();
    if (threadIdx.x < kStatSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < (unsigned)n_samples)
      FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
  }
}


}  // namespace

// in_flags bit of the library's own callers: `in_thr` is a range record (fq_common.h: kRangeMode), the weights are integer
// codes held in fp32 and the quantiser hands on CODES (multiply-back scale 1): the depthwise layer of nn.Conv2D(quantized=True)
constexpr unsigned kFlagRangeRecord = 0x100u;
constexpr unsigned kFlagBiasMultiplies = 0x200u;      // `bias` is the per-channel dequantisation factor (kActBiasMul)

static int dwconv3x3_impl(const float* x, const float* w, const float* bias, float* y, int64_t n, int64_t c, int64_t h,
                          int64_t wdt, int stride, const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                          float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                          fqStream_t stream) {
  FQ_REQUIRE(x && w && y, "fq_dwconv3x3: null pointer");
  FQ_REQUIRE(n > 0 && c > 0 && h > 0 && wdt > 0 && n * c < (1ll << 31) && h * wdt < (1ll << 28),
             "fq_dwconv3x3: bad shape (n=%lld c=%lld h=%lld w=%lld)", (long long)n, (long long)c, (long long)h,
             (long long)wdt);
  FQ_REQUIRE(stride == 1 || stride == 2, "fq_dwconv3x3: stride must be 1 or 2, got %d", stride);
  // in_stat alone: online; in_thr alone: offline; both: offline, the statistic only feeds out_current_max
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_dwconv3x3: bn_scale and bn_shift go together");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_dwconv3x3: unknown activation %d", act);
  const bool quant = in_stat != nullptr || in_thr != nullptr;
  if (quant) FQ_REQUIRE(in_width >= 2 && in_width <= 16, "fq_dwconv3x3: width %d out of range", in_width);
  if (in_flags & kFlagBiasMultiplies) act |= kActBiasMul;   // (after the range check; bias != NULL: run-time epilogue)
  static const int epi_on = env_int("FQ_DW_EPI", 1);    // 0: always the run-time epilogue (A/B)
  const int epi = (epi_on && bn_scale != nullptr && bias == nullptr)
                      ? (act == FQ_ACT_RELU ? kEpiBnRelu : act == FQ_ACT_RELU6 ? kEpiBnRelu6 : kEpiRuntime)
                      : kEpiRuntime;
  hipStream_t st = (hipStream_t)stream;
  static const int form = env_int("FQ_DW_FORM", 0);     // 0 auto, 1 LDS tiles, 2 sliding window 1 col/lane, 3: 4 cols/lane
  const bool can4 = (wdt % 4 == 0) && aligned16(x) && aligned16(y) && ((h * wdt) % 4 == 0) &&
                    (stride == 1 || ((wdt / 2) % 2 == 0));
  // 14x14 (stride 1 and 2), 7x7 and 28x28 (stride 1 and 2) planes: flat 16-byte accesses through an LDS transpose (K2p)
  // tuning: bit 0 14x14 s1, bit 1 14x14 s2, bit 2 7x7, bit 3 28x28 s1, bit 4 28x28 s2
  // (28x28: 40.1 -> 34.6 us stride 1.  Stride 2 on 28x28 was dropped in round 2 because 0.05 % of its outputs changed from run
  // to run at full size; round 3 found the cause - NOT a data race: a VALU write of a 16-byte buffer store's data register
  // right behind the store, a hazard hipcc does not guard when soffset is a register (fq_common.h at buf_st_v4f,
  // profiles/r3_dw_flat_race.txt, tools/dw_race_repro.py, tools/isa_lint.py) - every instantiation of this kernel had the
  // pattern one instruction further away.  With the stores guarded the form is exact at every occupancy.  Through round 4 it
  // was built and tested but not chosen by shape - one batch at a time it was no faster than the four-columns-per-lane form
  // (28.3 us against ~25; whole step 1.1967 against 1.1919 ms, profiles/r3_dw_flat_race.txt); with three batches in flight and
  // its nontemporal loads it is: default workload +1.77 % images/s (sd 0.03, profiles/r5_heuristics_ab.txt).)
  static const int flat_on = env_int("FQ_DW_FLAT", 31) & 31;
  {
    const int kind = (h == 14 && wdt == 14) ? (stride == 1 ? 0 : 1)
                     : (h == 7 && wdt == 7 && stride == 1) ? 2
                     : (h == 28 && wdt == 28) ? (stride == 1 ? 3 : 4) : -1;
    const int kP = kind == 0 ? 9 : kind >= 3 ? 4 : 8;      // planes per wavefront and block (dwconv3x3_flat_kernel: P)
    const bool whole = kind == 0 || kind >= 3 || (n * c) % kP == 0;     // planes of 49 floats: no 16-byte tail
    if (kind >= 0 && (form == 5 || (form == 0 && ((flat_on >> kind) & 1))) && whole && aligned16(x) && aligned16(y) &&
        n * c * h * wdt * 4 < (1ll << 31)) {
      DwFlatGeom fg;
      fg.C = (int)c;
      fg.planes = (unsigned)(n * c);
      fg.by_c = fast_div_for((unsigned)c);
      const int64_t nblk = (n * c + kP * (kBlock / 64) - 1) / (kP * (kBlock / 64));
      static const int fl_tune = env_int("FQ_DW_FLAT_WG_PER_CU", 0);
      // three workgroups per CU, each with one block in work and one requested: measured best for all three shapes
      // (512 x 14x14: 2 / 3 / 4 / 6 per CU = 22.2 / 19.6 / 20.7 / 21.9 us; 1024 x 7x7: 16.1 / 14.6 / 17.3 (8) / 16.4 us)
      const int fl_wg_per_cu = fl_tune > 0 ? fl_tune : 3;
      const int grid = (int)(nblk < (int64_t)num_cu() * fl_wg_per_cu ? nblk : (int64_t)num_cu() * fl_wg_per_cu);
      fg.per = (unsigned)(nblk / grid);
      fg.rem = (unsigned)(nblk % grid);
      const float levels = act_levels(in_width, in_flags);
      const int lo_neg = (in_flags & kFlagRangeRecord) ? kRangeMode : ((in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0);
      const float eps = (in_flags & FQ_ACT_NO_EPS) ? 0.0f : kEps;
      const int64_t ho = (h - 1) / stride + 1, wo = (wdt - 1) / stride + 1;
      if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
      ProfScope prof(FQ_KERNEL_DWCONV, 4.0 * ((double)n * c * h * wdt + (double)n * c * ho * wo), st);
#define FQ_DWF_E(SS, HH, Q, O, E)                                                                                 \
  hipLaunchKernelGGL((dwconv3x3_flat_kernel<SS, Q, O, HH, HH, E>), dim3(grid), dim3(kBlock), 0, st, x, w, bias, y, \
                     fg, in_stat, (int)n, in_thr, levels, lo_neg, eps, out_current_max, bn_scale, bn_shift, act,  \
                     stat_out)
#define FQ_DWF(SS, HH, Q, O)                                                                                      \
  do {                                                                                                            \
    if (epi == kEpiBnRelu) FQ_DWF_E(SS, HH, Q, O, kEpiBnRelu);                                                    \
    else if (epi == kEpiBnRelu6) FQ_DWF_E(SS, HH, Q, O, kEpiBnRelu6);                                             \
    else FQ_DWF_E(SS, HH, Q, O, kEpiRuntime);                                                                     \
  } while (0)
#define FQ_DWF_Q(SS, HH)                                                                                          \
  do {                                                                                                            \
    if (!quant) FQ_DWF(SS, HH, false, false);                                                                     \
    else if (!in_thr) FQ_DWF(SS, HH, true, true);                                                                 \
    else FQ_DWF(SS, HH, true, false);                                                                             \
  } while (0)
      if (kind == 0) FQ_DWF_Q(1, 14);
      else if (kind == 1) FQ_DWF_Q(2, 14);
      else if (kind == 2) FQ_DWF_Q(1, 7);
      else if (kind == 4) FQ_DWF_Q(2, 28);
      else FQ_DWF_Q(1, 28);
#undef FQ_DWF_Q
#undef FQ_DWF
#undef FQ_DWF_E
      FQ_LAUNCH_CHECK();
      return FQ_OK;
    }
  }
  // small planes: whole planes in registers, pipelined across blocks (K2o)
  static const int planes_on = env_int("FQ_DW_PLANES", 1);
  static const int planes_cpl = env_int("FQ_DW_PLANES_CPL", 2);          // 1: one column per lane even where W is even (A/B)
  const int64_t ho_ = (h - 1) / stride + 1, wo_ = (wdt - 1) / stride + 1;
  const bool al8 = ((((uintptr_t)x) | ((uintptr_t)y)) & 7) == 0;
  const bool two_cols = wdt % 2 == 0 && al8 && (stride == 2 || planes_cpl == 2);
  const bool can_planes = (h == 14 || h == 7) && wdt <= 64 && (stride == 1 || (two_cols && h == 14)) &&
                          n * c * h * wdt * 4 < (1ll << 31);
  if ((form == 4 || (form == 0 && planes_on)) && can_planes) {
    DwColGeom cg = {};
    cg.C = (int)c;
    cg.H = (int)h;
    cg.W = (int)wdt;
    cg.Ho = (int)ho_;
    cg.Wo = (int)wo_;
    cg.nsegx = 1;
    cg.sw = cg.Wo;
    cg.SEG = stride == 2 ? cg.Wo : (two_cols ? cg.W / 2 : cg.W);       // lanes per plane (no halo lanes)
    cg.segs = 64 / cg.SEG;
    const int64_t total_segs = n * c;
    const int64_t segs_per_block = (int64_t)cg.segs * (kBlock / 64);
    const int64_t nblk = (total_segs + segs_per_block - 1) / segs_per_block;
    static const int pl_wg_per_cu = env_int("FQ_DW_PLANES_WG_PER_CU", 5);
    const int grid = (int)(nblk < (int64_t)num_cu() * pl_wg_per_cu ? nblk : (int64_t)num_cu() * pl_wg_per_cu);
    const float levels = act_levels(in_width, in_flags);
    const int lo_neg = (in_flags & kFlagRangeRecord) ? kRangeMode : ((in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0);
    const float eps = (in_flags & FQ_ACT_NO_EPS) ? 0.0f : kEps;
    if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
    ProfScope prof(FQ_KERNEL_DWCONV, 4.0 * ((double)n * c * h * wdt + (double)n * c * cg.Ho * cg.Wo), st);
#define FQ_DWP_E(SS, Q, O, HH, CP, E)                                                                             \
  hipLaunchKernelGGL((dwconv3x3_planes_kernel<SS, Q, O, HH, CP, E>), dim3(grid), dim3(kBlock), 0, st, x, w, bias,  \
                     y, cg, total_segs, in_stat, (int)n, in_thr, levels, lo_neg, eps, out_current_max, bn_scale,  \
                     bn_shift, act, stat_out)
#define FQ_DWP(SS, Q, O, HH, CP)                                                                                  \
  do {                                                                                                            \
    if (epi == kEpiBnRelu) FQ_DWP_E(SS, Q, O, HH, CP, kEpiBnRelu);                                                \
    else if (epi == kEpiBnRelu6) FQ_DWP_E(SS, Q, O, HH, CP, kEpiBnRelu6);                                         \
    else FQ_DWP_E(SS, Q, O, HH, CP, kEpiRuntime);                                                                 \
  } while (0)
#define FQ_DWP_Q(SS, HH, CP)                                                                                      \
  do {                                                                                                            \
    if (!quant) FQ_DWP(SS, false, false, HH, CP);                                                                 \
    else if (!in_thr) FQ_DWP(SS, true, true, HH, CP);                                                             \
    else FQ_DWP(SS, true, false, HH, CP);                                                                         \
  } while (0)
    if (stride == 2) FQ_DWP_Q(2, 14, 2);
    else if (h == 14 && two_cols) FQ_DWP_Q(1, 14, 2);
    else if (h == 14) FQ_DWP_Q(1, 14, 1);
    else if (two_cols) FQ_DWP_Q(1, 7, 2);
    else FQ_DWP_Q(1, 7, 1);
#undef FQ_DWP_Q
#undef FQ_DWP
#undef FQ_DWP_E
    FQ_LAUNCH_CHECK();
    return FQ_OK;
  }
  if ((form == 3 || form == 0) && can4) {
    DwColGeom cg = {};
    cg.C = (int)c;
    cg.H = (int)h;
    cg.W = (int)wdt;
    cg.Ho = (int)((h - 1) / stride + 1);
    cg.Wo = (int)((wdt - 1) / stride + 1);
    const int halo = stride == 1 ? 2 : 1;
    const int quads = (int)(wdt / 4);                    // compute lanes per full row
    const int max_sw = 64 - halo;
    cg.nsegx = (quads + max_sw - 1) / max_sw;
    cg.sw = (quads + cg.nsegx - 1) / cg.nsegx;           // compute lanes per segment
    cg.SEG = cg.sw + halo;
    cg.segs = 64 / cg.SEG;
    const int64_t total_segs = n * c * cg.nsegx;
    const int64_t segs_per_block = (int64_t)cg.segs * (kBlock / 64);
    const int64_t nblk = (total_segs + segs_per_block - 1) / segs_per_block;
    FQ_REQUIRE(total_segs < (1ll << 31) - 1024, "fq_dwconv3x3: tensor too large for 32-bit segment indices");
    // every workgroup resident at once, each walking a contiguous range of blocks.  6 per CU: the kernels use ~100 scalar
    // registers, and a CU admits 256-thread workgroups 8 at a time only up to 80 (MI355X_MICROARCH.md, residency) - with
    // 8 per CU asked for, a quarter of the grid ran as a second, thin round (14x14 layers: 31.7 -> 28.0 us; capping the
    // scalar registers with -DFQ_DW_SGPR=80 instead makes all 8 resident and is no faster)
    static const int dw_wg_per_cu = env_int("FQ_DW_WG_PER_CU", 6);
    const int grid = (int)(nblk < (int64_t)num_cu() * dw_wg_per_cu ? nblk : (int64_t)num_cu() * dw_wg_per_cu);
    const float levels = act_levels(in_width, in_flags);
    const int lo_neg = (in_flags & kFlagRangeRecord) ? kRangeMode : ((in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0);
    const float eps = (in_flags & FQ_ACT_NO_EPS) ? 0.0f : kEps;
    if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
    ProfScope prof(FQ_KERNEL_DWCONV, 4.0 * ((double)n * c * h * wdt + (double)n * c * cg.Ho * cg.Wo), st);
    // nontemporal loads above this many MB of input (300 through round 4 = the 411 MB tensor only; 200 takes in the three
    // 205 MB ones: +0.4 ... +0.7 % images/s in two alternating A/Bs, 100 +0.3 %, 500 -0.5 %: profiles/r5_heuristics_ab.txt)
    static const int dw_nt_mb = env_int("FQ_DW_NT_MB", 200);
    const bool nt = 4.0 * (double)n * c * h * wdt > 1e6 * dw_nt_mb;
    static const int dw_nts_mb = env_int("FQ_DW_NTS_MB", 1 << 30);        // nontemporal stores from this many MB of output on
    cg.nts = 4.0 * (double)n * c * cg.Ho * cg.Wo >= 1e6 * dw_nts_mb ? 1 : 0;
#define FQ_DWC4_E(SS, Q, O, NT_, E)                                                                               \
  hipLaunchKernelGGL((dwconv3x3_cols4_kernel<SS, Q, O, NT_, E>), dim3(grid), dim3(kBlock), 0, st, x, w, bias, y,  \
                     cg, total_segs, in_stat, (int)n, in_thr, levels, lo_neg, eps, out_current_max, bn_scale,     \
                     bn_shift, act, stat_out)
#define FQ_DWC4_N(SS, Q, O, NT_)                                                                                  \
  do {                                                                                                            \
    if (epi == kEpiBnRelu) FQ_DWC4_E(SS, Q, O, NT_, kEpiBnRelu);                                                  \
    else if (epi == kEpiBnRelu6) FQ_DWC4_E(SS, Q, O, NT_, kEpiBnRelu6);                                           \
    else FQ_DWC4_E(SS, Q, O, NT_, kEpiRuntime);                                                                   \
  } while (0)
#define FQ_DWC4(SS, Q, O)                                                                                         \
  do {                                                                                                            \
    if (nt) FQ_DWC4_N(SS, Q, O, true);                                                                            \
    else FQ_DWC4_N(SS, Q, O, false);                                                                              \
  } while (0)
    if (stride == 1) {
      if (!quant) FQ_DWC4(1, false, false);
      else if (!in_thr) FQ_DWC4(1, true, true);
      else FQ_DWC4(1, true, false);
    } else {
      if (!quant) FQ_DWC4(2, false, false);
      else if (!in_thr) FQ_DWC4(2, true, true);
      else FQ_DWC4(2, true, false);
    }
#undef FQ_DWC4
#undef FQ_DWC4_N
#undef FQ_DWC4_E
    FQ_LAUNCH_CHECK();
    return FQ_OK;
  }
  if (form == 2 || ((form == 0 || form == 3) && wdt >= 14)) {   // narrow planes (7x7): LDS staging coalesces better
    DwColGeom cg = {};
    cg.C = (int)c;
    cg.H = (int)h;
    cg.W = (int)wdt;
    cg.Ho = (int)((h - 1) / stride + 1);
    cg.Wo = (int)((wdt - 1) / stride + 1);
    const int halo = stride == 1 ? 2 : 1;
    const int max_sw = 64 - halo;
    cg.nts = 0;
    cg.nsegx = (cg.Wo + max_sw - 1) / max_sw;
    cg.sw = (cg.Wo + cg.nsegx - 1) / cg.nsegx;
    cg.SEG = cg.sw + halo;
    cg.segs = 64 / cg.SEG;
    const int64_t total_segs = n * c * cg.nsegx;
    const int64_t segs_per_block = (int64_t)cg.segs * (kBlock / 64);
    const int64_t nblk = (total_segs + segs_per_block - 1) / segs_per_block;
    FQ_REQUIRE(total_segs < (1ll << 31) - 1024, "fq_dwconv3x3: tensor too large for 32-bit segment indices");
    // every workgroup resident at once, each walking a contiguous range of blocks.  6 per CU: the kernels use ~100 scalar
    // registers, and a CU admits 256-thread workgroups 8 at a time only up to 80 (MI355X_MICROARCH.md, residency) - with
    // 8 per CU asked for, a quarter of the grid ran as a second, thin round (14x14 layers: 31.7 -> 28.0 us; capping the
    // scalar registers with -DFQ_DW_SGPR=80 instead makes all 8 resident and is no faster)
    static const int dw_wg_per_cu = env_int("FQ_DW_WG_PER_CU", 6);
    const int grid = (int)(nblk < (int64_t)num_cu() * dw_wg_per_cu ? nblk : (int64_t)num_cu() * dw_wg_per_cu);
    const float levels = act_levels(in_width, in_flags);
    const int lo_neg = (in_flags & kFlagRangeRecord) ? kRangeMode : ((in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0);
    const float eps = (in_flags & FQ_ACT_NO_EPS) ? 0.0f : kEps;
    if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
    ProfScope prof(FQ_KERNEL_DWCONV, 4.0 * ((double)n * c * h * wdt + (double)n * c * cg.Ho * cg.Wo), st);
#define FQ_DWC_E(SS, Q, O, E)                                                                                     \
  hipLaunchKernelGGL((dwconv3x3_cols_kernel<SS, Q, O, E>), dim3(grid), dim3(kBlock), 0, st, x, w, bias, y, cg,     \
                     total_segs, in_stat, (int)n, in_thr, levels, lo_neg, eps, out_current_max, bn_scale,         \
                     bn_shift, act, stat_out)
#define FQ_DWC(SS, Q, O)                                                                                          \
  do {                                                                                                            \
    if (epi == kEpiBnRelu) FQ_DWC_E(SS, Q, O, kEpiBnRelu);                                                        \
    else if (epi == kEpiBnRelu6) FQ_DWC_E(SS, Q, O, kEpiBnRelu6);                                                 \
    else FQ_DWC_E(SS, Q, O, kEpiRuntime);                                                                         \
  } while (0)
    if (stride == 1) {
      if (!quant) FQ_DWC(1, false, false);
      else if (!in_thr) FQ_DWC(1, true, true);
      else FQ_DWC(1, true, false);
    } else {
      if (!quant) FQ_DWC(2, false, false);
      else if (!in_thr) FQ_DWC(2, true, true);
      else FQ_DWC(2, true, false);
    }
#undef FQ_DWC
#undef FQ_DWC_E
    FQ_LAUNCH_CHECK();
    return FQ_OK;
  }
  DwGeom g;
  g.C = (int)c;
  g.H = (int)h;
  g.W = (int)wdt;
  g.Ho = (int)((h - 1) / stride + 1);
  g.Wo = (int)((wdt - 1) / stride + 1);
  g.WS = g.W + 2;
  if ((g.WS & 1) == 0) g.WS += 1;                       // odd dword stride
  const int lds_budget = 8192;                          // floats (32 KiB) -> up to 5 workgroups per CU
  const int plane_in = g.H * g.W;
  if (plane_in <= 4096 && (g.H + 2) * g.WS <= lds_budget) {
    g.TR = g.Ho;
    g.strips = 1;
    g.IR = g.H + 2;
    g.P = 1;
    for (int p = (int)(c < 64 ? c : 64); p >= 1; --p)
      if (c % p == 0 && p * g.IR * g.WS <= lds_budget) {
        g.P = p;
        break;
      }
  } else {
    g.P = 1;
    int tr = (lds_budget / g.WS - 2) / stride;
    FQ_REQUIRE(tr >= 1, "fq_dwconv3x3: rows of %d floats do not fit the LDS tile", g.W);
    g.strips = (g.Ho + tr - 1) / tr;
    g.TR = (g.Ho + g.strips - 1) / g.strips;
    g.IR = g.TR * stride + 2;
  }
  {
    const int per_seg = g.P * g.Wo;
    int nseg = (2 * kBlock + per_seg - 1) / per_seg;    // aim at ~2 work items per lane
    if (nseg < 1) nseg = 1;
    if (nseg > g.TR) nseg = g.TR;
    g.RS = (g.TR + nseg - 1) / nseg;
    g.nseg = (g.TR + g.RS - 1) / g.RS;
  }
  const bool base_ok = aligned16(x);
  if (g.P > 1 || g.strips == 1)
    g.vec_in = base_ok && (((int64_t)g.P * plane_in) % 4 == 0) && (plane_in % 4 == 0 || g.P % 4 == 0);
  else
    g.vec_in = base_ok && (g.W % 4 == 0) && (plane_in % 4 == 0);
  const int64_t tiles = (n * c / g.P) * g.strips;
  size_t lds = (size_t)((g.P * g.IR * g.WS + 3) / 4 * 4 + 16) * sizeof(float);
  const int grid = grid_for(tiles);
  const float levels = act_levels(in_width, in_flags);
  const int lo_neg = (in_flags & kFlagRangeRecord) ? kRangeMode : ((in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0);
  const float eps = (in_flags & FQ_ACT_NO_EPS) ? 0.0f : kEps;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  ProfScope prof(FQ_KERNEL_DWCONV, 4.0 * ((double)n * c * h * wdt + (double)n * c * g.Ho * g.Wo), st);
#define FQ_DW(SS, Q, O)                                                                                           \
  hipLaunchKernelGGL((dwconv3x3_kernel<SS, Q, O>), dim3(grid), dim3(kBlock), lds, st, x, w, bias, y, g, tiles,    \
                     in_stat, (int)n, in_thr, levels, lo_neg, eps, out_current_max, bn_scale, bn_shift, act,      \
                     stat_out)
  if (stride == 1) {
    if (!quant) FQ_DW(1, false, false);
    else if (!in_thr) FQ_DW(1, true, true);
    else FQ_DW(1, true, false);
  } else {
    if (!quant) FQ_DW(2, false, false);
    else if (!in_thr) FQ_DW(2, true, true);
    else FQ_DW(2, true, false);
  }
#undef FQ_DW
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

namespace fqi {
// depthwise 3x3 of nn.Conv2D(quantized=True): x quantised on load with the range record `rec` (codes, not values, enter the
// sums), wcodes_f32 = the int8 weight codes as fp32, y = act(sum * svec[c]) with svec = in_scale * w_scale, or - a BatchNorm
// folded behind the block - y = act((sum * svec[c]) * bn_scale[c] + bn_shift[c]), every step separately rounded
int dw_range_call(const float* x, const float* wcodes_f32, float* y, int64_t n, int64_t c, int64_t h, int64_t wdt, int stride,
                  const float* rec, const float* svec, const float* bn_scale, const float* bn_shift, int act,
                  float* stat_out, hipStream_t st) {
  return dwconv3x3_impl(x, wcodes_f32, svec, y, n, c, h, wdt, stride, nullptr, rec, 8, kFlagRangeRecord | kFlagBiasMultiplies,
                        nullptr, bn_scale, bn_shift, act, stat_out, (fqStream_t)st);
}
}  // namespace fqi

extern "C" {

int fq_dwconv3x3(const float* x, const float* w, const float* bias, float* y, int64_t n, int64_t c, int64_t h,
                 int64_t wdt, int stride, const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                 float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                 fqStream_t stream) {
  FQ_REQUIRE(!(in_flags & ~(FQ_ACT_SIGNED | FQ_ACT_LO_NEG_MAX | FQ_ACT_NO_ABS | FQ_ACT_NO_EPS)), "fq_dwconv3x3: unknown flags");
  return dwconv3x3_impl(x, w, bias, y, n, c, h, wdt, stride, in_stat, in_thr, in_width, in_flags, out_current_max, bn_scale,
                        bn_shift, act, stat_out, stream);
}

}  // extern "C"
