// libfakequant — K2f pointwise (1x1) convolution on int8 codes for shapes the specialised forms do not take
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_pw.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// K2f: pointwise (1x1) convolution on integer codes with v_mfma_i32_16x16x64_i8.
// Operand layout of the instruction (probed on gfx950, tools/mfma_i8_probe.hip): lane l holds, for A, row l&15 and the
// 16 consecutive k = 16*(l>>4) .. +15 (one 16-byte register quad); the same for B with column l&15; C/D: column l&15,
// rows 4*(l>>4) + r.  NCHW keeps PIXELS contiguous, the MFMA wants K (= input channels) contiguous for both operands:
// weights are stored [co][ci] (fine), the activation tile is transposed on its way into LDS — it is read from HBM
// once, coalesced along pixels, quantised, and written as int8 codes [column][ci] with byte stores.
// A workgroup owns PT_B columns (column = (sample, pixel) flattened) and ALL output channels, so x is read exactly once;
// its 4 waves are arranged wm x wn over (64 output channels) x (64 columns) wave tiles and loop over channel passes.
// ---------------------------------------------------------------------------------------------------------------
struct PwGeom {
  int Cin, CinPad, Cout, HW;
  int64_t cols;        // n * HW
  int PT_B;            // columns per workgroup tile (64 * wn)
  int wm, wn;          // wave grid
  int passes;          // ceil(Cout / (64 * wm))
  int stride;          // LDS bytes per column (CinPad + 16)
  int zoff;            // 128 for unsigned codes (stored re-centred), 0 for signed
};

// K2f-A: quantise + transpose.  x (n, Cin, HW) fp32 -> codes [(n*HW + p)][CinPad] int8 (column-major for the GEMM:
// K contiguous).  Workgroup tile = 64 channels x 64 pixels of one sample: every thread loads 4 channel rows x 4 pixels
// (16-byte loads, 256 B contiguous per row across 16 lanes), quantises, transposes its 4x4 block in registers into 4
// dwords (4 channels of one pixel each), and the tile goes through a small LDS stage (odd dword stride: conflict-free)
// so that the stores are 16 bytes per lane, 64 contiguous bytes per pixel.
template <bool ONLINE>
__global__ __launch_bounds__(kBlock) void quant_transpose_i8_kernel(
    const float* __restrict__ x, int8_t* __restrict__ codes, int Cin, int CinPad, int HW, int ptiles, int ctiles,
    int64_t tiles, const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels,
    int lo_neg_max, float eps, int zoff, float* __restrict__ cur_max_out) {
  __shared__ int lds[64 * 17];
  const float max_ = input_threshold(in_stat, n, ONLINE ? nullptr : in_thr, cur_max_out, blockIdx.x == 0);
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  const int c = threadIdx.x & 15;            // pixel quad
  const int rq = threadIdx.x >> 4;           // channel quad (0..15)
  const bool hw_vec = (HW & 3) == 0;
  const ChunkRange rg = block_range(tiles);
  for (int64_t t = rg.begin; t < rg.end; ++t) {
    // tile order: channel tile fastest, then pixel tile, then sample
    const int ct = (int)(t % ctiles);
    const int64_t t2 = t / ctiles;
    const int pt = (int)(t2 % ptiles);
    const int64_t smp = t2 / ptiles;
    const int ci0 = ct * 64 + rq * 4, p0 = pt * 64 + c * 4;
    float v[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ci = ci0 + k;
      const int cic = ci < Cin ? ci : Cin - 1;
      const float* src = x + (smp * Cin + cic) * (int64_t)HW;
      if (hw_vec && p0 + 3 < HW) {
        const f4 r = *reinterpret_cast<const f4*>(src + p0);
        v[k][0] = r.x; v[k][1] = r.y; v[k][2] = r.z; v[k][3] = r.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pe = p0 + e < HW ? p0 + e : HW - 1;
          v[k][e] = src[pe];
        }
      }
    }
    __syncthreads();                                                   // LDS free (previous tile stored)
#pragma unroll
    for (int e = 0; e < 4; ++e) {                                       // pixel p0 + e: channels ci0 .. ci0+3 in one dword
      unsigned packed = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int code = (int)fq_code(v[k][e], q) - zoff;
        if (ci0 + k >= Cin) code = 0;
        packed |= ((unsigned)code & 0xFFu) << (8 * k);
      }
      lds[(c * 4 + e) * 17 + rq] = (int)packed;
    }
    __syncthreads();
    // store: thread (pixel = tid / 4, 16-byte piece = tid % 4)
    const int sp = threadIdx.x >> 2, piece = threadIdx.x & 3;
    const int p = pt * 64 + sp;
    if (p < HW) {
      const int* l = lds + sp * 17 + piece * 4;
      const v4i o = (v4i){l[0], l[1], l[2], l[3]};
      *reinterpret_cast<v4i*>(codes + (smp * HW + p) * (int64_t)CinPad + ct * 64 + piece * 16) = o;
    }
  }
}

// K2f-B: integer GEMM + epilogue.  Both MFMA operands are K-contiguous in global memory (weights [co][CinPad], codes
// [column][CinPad]) and go straight to registers, double buffered over the K loop; no LDS.  A workgroup computes
// (64*wm output channels) x (64*wn columns); waves own 64 x 64 sub-tiles = 16 accumulators of 16x16.
template <int DUMMY>
__global__ __launch_bounds__(kBlock) void pwconv_i8_kernel(
    const int8_t* __restrict__ xc, const int8_t* __restrict__ wc, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwGeom g, int64_t tiles,
    const float* __restrict__ sx_src, float levels, const float* __restrict__ bn_scale,
    const float* __restrict__ bn_shift, int act, float* __restrict__ stat_out) {
  // per-output-channel constants of this workgroup's channel block, staged in LDS once: loading them from global in
  // the epilogue (16 dependent round trips per tile) was 90 % of this kernel's time
  constexpr int kStatSlots = 16;
  __shared__ float k_sxw[256], k_bias[256], k_bsc[256], k_bsh[256];
  __shared__ int k_zs[256];
  __shared__ unsigned k_stat[kStatSlots];
  const float sx = sx_src[0] / levels;                                  // scale = max_/levels, as make_qparams
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wmi = wave % g.wm, wni = wave / g.wm;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const unsigned HW = (unsigned)g.HW;
  const int64_t plane_stride = (int64_t)g.HW;
  const int cblocks = g.passes;                                         // output-channel blocks of 64*wm
  // gridDim.x is a multiple of cblocks (host), so a workgroup keeps ONE channel block over all its tiles
  const int cb = (int)(blockIdx.x % cblocks);
  {
    const int co = cb * g.wm * 64 + threadIdx.x;
    const bool ok = threadIdx.x < g.wm * 64 && co < g.Cout;
    const int coc = ok ? co : 0;
    k_sxw[threadIdx.x] = sx * wscale[coc];
    k_zs[threadIdx.x] = g.zoff * wsum[coc];
    k_bias[threadIdx.x] = bias != nullptr ? bias[coc] : 0.0f;
    k_bsc[threadIdx.x] = has_bn ? bn_scale[coc] : 1.0f;
    k_bsh[threadIdx.x] = has_bn ? bn_shift[coc] : 0.0f;
  }
  __syncthreads();

  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t ctile = t / cblocks;
    const unsigned j0 = (unsigned)(ctile * g.PT_B) + wni * 64;
    const int co0 = (cb * g.wm + wmi) * 64;                             // (weights are zero-padded to 64 rows: no early exit,
    v4i acc[4][4];                                                      //  every wave reaches the barriers below)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (v4i){0, 0, 0, 0};
    // a wave whose 64 channels lie entirely beyond Cout (Cout not a multiple of 64*wm) reads channel block 0 instead:
    // its results are masked in the epilogue, but the weight buffer is only padded to the next multiple of 64 rows
    const int co_ld = co0 < g.Cout ? co0 : 0;
    const int8_t* wrow = wc + (int64_t)(co_ld + (lane & 15)) * g.CinPad + 16 * (lane >> 4);
    const int8_t* xrow = xc + (int64_t)(j0 + (lane & 15)) * g.CinPad + 16 * (lane >> 4);
    const int64_t step16 = (int64_t)16 * g.CinPad;
    v4i a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = *reinterpret_cast<const v4i*>(wrow + i * step16);
      b[i] = *reinterpret_cast<const v4i*>(xrow + i * step16);
    }
    for (int k0 = 0; k0 < g.CinPad; k0 += 64) {
      v4i an[4], bn[4];
      const int kn = k0 + 64 < g.CinPad ? k0 + 64 : k0;                 // last step re-reads its own slab (discarded)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        an[i] = *reinterpret_cast<const v4i*>(wrow + i * step16 + kn);
        bn[i] = *reinterpret_cast<const v4i*>(xrow + i * step16 + kn);
      }
#pragma unroll
      for (int uu = 0; uu < 4; ++uu)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[uu][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b[uu], a[tt], acc[uu][tt], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = an[i];
        b[i] = bn[i];
      }
    }
    // epilogue.  The MFMAs were issued with the ACTIVATION codes as the A operand, so D = (pixels x channels): lane
    // holds, per (uu, tt), the 4 consecutive PIXELS j0 + uu*16 + 4*(lane>>4) + r of channel co0 + tt*16 + (lane&15)
    // -> one 16-byte store per (uu, tt) when the four pixels sit in one sample, and the per-channel constants are
    // per LANE (read once per tt).
    // per-sample maxima of this tile go through an LDS table (a tile spans <= kStatSlots samples), then ONE global
    // atomic per touched sample: per-wave global atomics on the 128 hot addresses serialised in L2 and cost 80 % of the
    // kernel (100 k same-address atomics per launch)
    const unsigned s_base = (unsigned)(ctile * g.PT_B) / HW;
    if (has_stat) {
      __syncthreads();
      if (threadIdx.x < kStatSlots) k_stat[threadIdx.x] = 0u;
      __syncthreads();
    }
    int64_t ybase[4];
    bool cok[4], vec_ok[4];
    unsigned smps[4];
    float m[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) {
      const unsigned j = j0 + uu * 16 + 4 * (lane >> 4);               // first of this lane's 4 pixels
      cok[uu] = j < (unsigned)g.cols;
      const unsigned smp = cok[uu] ? j / HW : 0;
      const unsigned p = j - smp * HW;
      smps[uu] = smp;
      ybase[uu] = ((int64_t)smp * g.Cout) * plane_stride + p;
      vec_ok[uu] = cok[uu] && ((HW & 3u) == 0u) && (p + 3 < HW) && (j + 3 < (unsigned)g.cols);
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int col = wmi * 64 + tt * 16 + (lane & 15);                 // channel index inside the block
      const int co = co0 + tt * 16 + (lane & 15);
      const bool co_ok = co < g.Cout;
      const float sxw = k_sxw[col];
      const int zs = k_zs[col];
      const float bch = k_bias[col], bsc = k_bsc[col], bsh = k_bsh[col];
      const int64_t coff = (int64_t)(co_ok ? co : 0) * plane_stride;
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = (float)(acc[uu][tt][r] + zs) * sxw;
          if (bias != nullptr) v = v + bch;
          if (has_bn) {
            v = v * bsc;
            v = v + bsh;
          }
          o[r] = act_rt(v, act);
        }
        if (co_ok && vec_ok[uu]) {
          *reinterpret_cast<f4*>(y + ybase[uu] + coff) = (f4){o[0], o[1], o[2], o[3]};
          m[uu] = fmaxf(m[uu], fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        } else if (co_ok && cok[uu]) {
          // ragged: the 4 pixels may straddle a sample boundary or the end of the tensor
          const unsigned jb = j0 + uu * 16 + 4 * (lane >> 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned j = jb + r;
            if (j < (unsigned)g.cols) {
              const unsigned smp = j / HW;
              y[((int64_t)smp * g.Cout) * plane_stride + (j - smp * HW) + coff] = o[r];
              if (smp == smps[uu]) m[uu] = fmaxf(m[uu], fabsf(o[r]));
              else if (has_stat) {
                const unsigned slot = smp - s_base;
                if (slot < kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(fabsf(o[r])));
                else atomic_max_f32(stat_out + smp, fabsf(o[r]));
              }
            }
          }
        }
      }
    }
    if (has_stat) {
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        if (cok[uu]) {
          const unsigned slot = smps[uu] - s_base;
          if (slot < kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(m[uu]));
          else atomic_max_f32(stat_out + smps[uu], m[uu]);
        }
      }
      __syncthreads();
      if (threadIdx.x < kStatSlots && k_stat[threadIdx.x] != 0u && s_base + threadIdx.x < (unsigned)(g.cols / HW))
        FQ_STAT_FLUSH_MAX(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
    }
  }
}

}  // namespace

namespace fqi {

// two kernels (K2f): quantise + transpose to K-contiguous int8 in the workspace, then the 16x16x64 GEMM
int pw_two_kernels(const PwCall& a) {
  int8_t* codes = (int8_t*)a.ws;
  {
    const int ptiles = (int)((a.hw + 63) / 64), ctiles = (int)(a.cin_pad / 64);
    const int64_t tiles = a.n * ptiles * ctiles;
    const int grid = grid_for(tiles);
    if (a.in_thr == nullptr)
      hipLaunchKernelGGL((quant_transpose_i8_kernel<true>), dim3(grid), dim3(kBlock), 0, a.st, a.x, codes, (int)a.cin,
                         (int)a.cin_pad, (int)a.hw, ptiles, ctiles, tiles, a.in_stat, (int)a.n, a.in_thr, a.levels,
                         a.lo_neg, kEps, a.zoff, a.out_current_max);
    else
      hipLaunchKernelGGL((quant_transpose_i8_kernel<false>), dim3(grid), dim3(kBlock), 0, a.st, a.x, codes, (int)a.cin,
                         (int)a.cin_pad, (int)a.hw, ptiles, ctiles, tiles, a.in_stat, (int)a.n, a.in_thr, a.levels,
                         a.lo_neg, kEps, a.zoff, a.out_current_max);
    FQ_LAUNCH_CHECK();
  }
  PwGeom g;
  g.Cin = (int)a.cin;
  g.CinPad = (int)a.cin_pad;
  g.Cout = (int)a.cout;
  g.HW = (int)a.hw;
  g.cols = a.n * a.hw;
  if (a.cout > 128) { g.wm = 4; g.wn = 1; }
  else if (a.cout > 64) { g.wm = 2; g.wn = 2; }
  else { g.wm = 1; g.wn = 4; }
  g.PT_B = 64 * g.wn;
  g.passes = (int)((a.cout + 64 * g.wm - 1) / (64 * g.wm));
  g.stride = 0;
  g.zoff = a.zoff;
  const int64_t tiles = ((g.cols + g.PT_B - 1) / g.PT_B) * g.passes;
  if (int rc = pw_zero_stat(a)) return rc;
  int64_t grid64 = tiles < (int64_t)num_cu() * 32 ? tiles : (int64_t)num_cu() * 32;
  grid64 = grid64 / g.passes * g.passes;                  // multiple of the channel blocks (tiles is one already)
  const int grid = (int)(grid64 < g.passes ? g.passes : grid64);
  const float* sx_src = a.in_thr ? a.in_thr : a.out_current_max;      // offline: the stored threshold; online: kernel A wrote the batch mean
  hipLaunchKernelGGL((pwconv_i8_kernel<0>), dim3(grid), dim3(kBlock), 0, a.st, (const int8_t*)codes, a.wcodes, a.wscale,
                     (const int*)a.wsum, a.bias, a.y, g, tiles, sx_src, a.levels, a.bn_scale, a.bn_shift, a.act,
                     a.stat_out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // namespace fqi
