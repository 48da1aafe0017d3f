// libfakequant — K2f / K2g pointwise (1x1) convolution on int8 codes for shapes the specialised forms do not take
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
  const float max_ = ONLINE ? batch_mean_dev(in_stat, n) : in_thr[0];
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  if (ONLINE && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
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
        atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
    }
  }
}


// K2g: the two kernels above in ONE launch, no int8 intermediate in HBM.  A workgroup owns PX = 64*wn columns
// (column = (sample, pixel) flattened): phase 1 reads their fp32 activations once (coalesced along pixels), fake-quantises
// them and leaves the int8 codes, transposed to K-contiguous rows, in an LDS panel [PX][K]; phase 2 runs the integer
// GEMM of that panel against `nblk` blocks of 64*wm output channels: activation fragments come from LDS (ds_read_b128),
// weight fragments straight from global memory (L2 resident), both ping-pong buffered over an explicitly 2x unrolled K
// loop; the epilogue is K2f-B's.  Panel rows are XOR-swizzled in 16-byte chunks so that the 4-byte transposing writes
// and the 16-byte fragment reads are both (nearly) bank-conflict free without padding.
struct PwfGeom {
  int Cin, K, Cout, HW;
  int64_t cols;        // n * HW
  int wm, wn;          // wave grid: wm x wn waves of (64 channels) x (64 columns)
  int PX;              // columns per workgroup = 64 * wn
  int cblocks;         // ceil(Cout / (64 * wm))
  int csplit;          // workgroups sharing one column tile (each takes nblk channel blocks)
  int nblk;            // channel blocks per workgroup = ceil(cblocks / csplit)
  int swz;             // swizzle mask: 7 when K % 128 == 0, else 3
  int zoff;
};

__device__ __forceinline__ int pwf_panel_off(int row, int chunk, int K, int swz) {
  const int sw = swz == 7 ? ((row ^ (row >> 3)) & 7) : ((row >> 1) & 3);
  return row * K + (((chunk & ~swz) | ((chunk ^ sw) & swz)) << 4);
}

template <bool ONLINE, int G>
__global__ __launch_bounds__(kBlock, 3) void pwconv_fused_kernel(
    const float* __restrict__ x, const int8_t* __restrict__ wc, const float* __restrict__ wscale,
    const int* __restrict__ wsum, const float* __restrict__ bias, float* __restrict__ y, PwfGeom g,
    const float* __restrict__ in_stat, int n, const float* __restrict__ in_thr, float levels, int lo_neg_max, float eps,
    float* __restrict__ cur_max_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    float* __restrict__ stat_out) {
  constexpr int kStatSlots = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char pwf_smem[];
  __shared__ unsigned k_stat[kStatSlots];
  int8_t* panel = reinterpret_cast<int8_t*>(pwf_smem);                 // PX * K bytes
  const int nconst = g.nblk * g.wm * 64;                                // per-channel constants of this workgroup
  const int zero_off = g.PX * g.K;                                      // 16 zero bytes: the activation fragment of
  float* k_sxw = reinterpret_cast<float*>(pwf_smem + (size_t)g.PX * g.K + 16);   // the phantom step of an odd K/64
  float* k_bias = k_sxw + nconst;
  float* k_bsc = k_bias + nconst;
  float* k_bsh = k_bsc + nconst;
  int* k_zs = reinterpret_cast<int*>(k_bsh + nconst);

  PW_STAMP(0);
#ifdef FQ_PW_TRACE
  if (threadIdx.x == 0 && g_pw_trace != nullptr)
    g_pw_trace[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) |
                                             (unsigned long long)__builtin_amdgcn_s_getreg(63492);
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wmi = wave % g.wm, wni = wave / g.wm;
  const bool has_bn = bn_scale != nullptr, has_stat = stat_out != nullptr;
  const unsigned HW = (unsigned)g.HW;
  const int64_t plane_stride = (int64_t)g.HW;
  // workgroups are dealt to the 8 XCDs round-robin: the csplit workgroups that share a column tile (and re-read the
  // same activations) are placed on the SAME XCD, back to back, so that the repeats hit in that XCD's L2
  const int xcd = (int)(blockIdx.x & 7u);
  const int64_t slot = blockIdx.x >> 3;
  const int cs = (int)(slot % g.csplit);
  const int64_t ctile = (slot / g.csplit) * 8 + xcd;
  if (ctile * g.PX >= g.cols) return;                                  // grid is padded to 8 * csplit
  const int cb_first = cs * g.nblk;
  const unsigned jt0 = (unsigned)(ctile * g.PX);

  // ---- phase 1: quantise + transpose PX columns x K channels into the panel -------------------------------------
  // Units of (64 columns) x (64 channels); a thread owns 4 channels x 4 columns of each (four 16-byte loads).  G units
  // are loaded per group and two groups are in flight (register double buffer): the first group is issued BEFORE the
  // batch statistic is reduced, the next one before the current one is quantised.  Unit indices past the end are
  // clamped (loads are unconditional so that the compiler's waits only cover the older group).
  const int c = threadIdx.x & 15;            // pixel quad inside a 64-column group
  const int rq = threadIdx.x >> 4;           // channel quad inside a 64-channel group (0..15)
  const bool hw_vec = (HW & 3u) == 0u;
  const int ktiles = g.K >> 6;
  const int U = g.wn * ktiles;
  float va[G][4][4], vb[G][4][4];
  auto issue = [&](int u0, float (&v)[G][4][4]) {
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      int u = u0 + gi;
      u = u < U ? u : U - 1;
      const int pb = u / ktiles, ct = u - pb * ktiles;
      const unsigned jb = jt0 + pb * 64 + c * 4;
      const int ci0 = ct * 64 + rq * 4;
      if (ci0 >= g.Cin) continue;                                       // zero padding of K: nothing to read
      if (hw_vec) {
        unsigned j = jb < (unsigned)g.cols ? jb : 0u;                  // cols % 4 == 0 here: the quad is all in or all out
        const unsigned smp = j / HW;
        const float* base = x + (int64_t)smp * g.Cin * plane_stride + (j - smp * HW);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int cic = ci0 + k < g.Cin ? ci0 + k : g.Cin - 1;
          const f4 r = *reinterpret_cast<const f4*>(base + (int64_t)cic * plane_stride);
          v[gi][k][0] = r.x; v[gi][k][1] = r.y; v[gi][k][2] = r.z; v[gi][k][3] = r.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned j = jb + e;
          j = j < (unsigned)g.cols ? j : (unsigned)g.cols - 1;
          const unsigned smp = j / HW;
          const float* base = x + (int64_t)smp * g.Cin * plane_stride + (j - smp * HW);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int cic = ci0 + k < g.Cin ? ci0 + k : g.Cin - 1;
            v[gi][k][e] = base[(int64_t)cic * plane_stride];
          }
        }
      }
    }
  };
#ifdef FQ_PW_TRACE
  const bool skip_loads = (g_pw_dbg & 2) != 0;
#else
  constexpr bool skip_loads = false;
#endif
  if (!skip_loads) issue(0, va);
  __builtin_amdgcn_sched_barrier(0);
  const float max_ = ONLINE ? batch_mean_dev(in_stat, n) : in_thr[0];
  const QParams q = make_qparams(max_, levels, lo_neg_max != 0, eps);
  if (ONLINE && cur_max_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) cur_max_out[0] = max_;
  const float sx = q.scale;
  PW_STAMP(1);
  auto process = [&](int u0, float (&v)[G][4][4]) {
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      const int u = u0 + gi;
      if (u < U) {
        const int pb = u / ktiles, ct = u - pb * ktiles;
        const int ci0 = ct * 64 + rq * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned packed = 0;
          if (ci0 < g.Cin) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              int code = fq_code_int(v[gi][k][e], q) - g.zoff;
              if (ci0 + k >= g.Cin) code = 0;
              packed |= ((unsigned)code & 0xFFu) << (8 * k);
            }
          }
          const int row = pb * 64 + c * 4 + e;
          *reinterpret_cast<unsigned*>(panel + pwf_panel_off(row, ct * 4 + (rq >> 2), g.K, g.swz) + (rq & 3) * 4) = packed;
        }
      }
    }
  };
  for (int u0 = 0; u0 < U; u0 += 2 * G) {
    if (!skip_loads) issue(u0 + G, vb);
    __builtin_amdgcn_sched_barrier(0);
    process(u0, va);
    __builtin_amdgcn_sched_barrier(0);
    if (!skip_loads) issue(u0 + 2 * G, va);
    __builtin_amdgcn_sched_barrier(0);
    process(u0 + G, vb);
    __builtin_amdgcn_sched_barrier(0);
  }
  // constants of all channel blocks of this workgroup, and the zero chunk
  if (threadIdx.x < 4) reinterpret_cast<int*>(panel + zero_off)[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < nconst; i += kBlock) {
    const int co = cb_first * g.wm * 64 + i;
    const int coc = co < g.Cout ? co : 0;
    k_sxw[i] = sx * wscale[coc];
    k_zs[i] = g.zoff * wsum[coc];
    k_bias[i] = bias != nullptr ? bias[coc] : 0.0f;
    k_bsc[i] = has_bn ? bn_scale[coc] : 1.0f;
    k_bsh[i] = has_bn ? bn_shift[coc] : 0.0f;
  }
  PW_STAMP(2);
  __syncthreads();
  PW_STAMP(3);

  // ---- phase 2: integer GEMM of the panel against this workgroup's channel blocks ---------------------------------
  const unsigned j0 = jt0 + wni * 64;
  const unsigned s_base = jt0 / HW;
  int64_t ybase[4];
  bool cok[4], vec_ok[4];
  unsigned smps[4];
#pragma unroll
  for (int uu = 0; uu < 4; ++uu) {
    const unsigned j = j0 + uu * 16 + 4 * (lane >> 4);                 // first of this lane's 4 pixels
    cok[uu] = j < (unsigned)g.cols;
    const unsigned smp = cok[uu] ? j / HW : 0;
    const unsigned p = j - smp * HW;
    smps[uu] = smp;
    ybase[uu] = ((int64_t)smp * g.Cout) * plane_stride + p;
    vec_ok[uu] = cok[uu] && ((HW & 3u) == 0u) && (p + 3 < HW) && (j + 3 < (unsigned)g.cols);
  }
  // LDS byte offsets of this lane's four activation fragments (row fixed, chunk advances by 4 per K step)
  int brow[4];
#pragma unroll
  for (int uu = 0; uu < 4; ++uu) brow[uu] = wni * 64 + uu * 16 + (lane & 15);
  const int ksteps = g.K >> 6;
  const int64_t step16 = (int64_t)16 * g.K;

  for (int bi = 0; bi < g.nblk; ++bi) {
    const int cb = cb_first + bi;
    if (cb >= g.cblocks) break;
    const int co0 = (cb * g.wm + wmi) * 64;
    const int co_ld = co0 < g.Cout ? co0 : 0;                           // see K2f-B
    const int8_t* wrow = wc + (int64_t)(co_ld + (lane & 15)) * g.K + 16 * (lane >> 4);
    // accumulators start at the zero-point correction zoff * sum(w codes) of their channel (D: lane -> channel)
    v4i acc[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int zs = k_zs[(bi * g.wm + wmi) * 64 + b * 16 + (lane & 15)];
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a][b] = (v4i){zs, zs, zs, zs};
    }
    v4i a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a0[i] = *reinterpret_cast<const v4i*>(wrow + i * step16);
      b0[i] = *reinterpret_cast<const v4i*>(panel + pwf_panel_off(brow[i], (lane >> 4), g.K, g.swz));
    }
    // Two K steps per iteration on two register sets, NO branch inside: with a conditional load or MFMA group the
    // compiler falls back to vmcnt(0) right after issuing the prefetch.  Indices past the end are clamped (the data
    // is discarded), and for an odd number of steps the second group of the last iteration multiplies by the zero
    // chunk instead of being skipped.
    for (int ks = 0; ks < ksteps; ks += 2) {
      const bool real1 = ks + 1 < ksteps;
      const int k1 = real1 ? ks + 1 : ks;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a1[i] = *reinterpret_cast<const v4i*>(wrow + i * step16 + k1 * 64);
        const int off = pwf_panel_off(brow[i], k1 * 4 + (lane >> 4), g.K, g.swz);
        b1[i] = *reinterpret_cast<const v4i*>(panel + (real1 ? off : zero_off));
      }
      __builtin_amdgcn_sched_barrier(0);        // keep the prefetch ABOVE the MFMA group (the scheduler sinks loads
                                                // next to their use to save registers, which serialises on latency)
#pragma unroll
      for (int uu = 0; uu < 4; ++uu)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[uu][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b0[uu], a0[tt], acc[uu][tt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const int k2 = ks + 2 < ksteps ? ks + 2 : ks;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a0[i] = *reinterpret_cast<const v4i*>(wrow + i * step16 + k2 * 64);
        b0[i] = *reinterpret_cast<const v4i*>(panel + pwf_panel_off(brow[i], k2 * 4 + (lane >> 4), g.K, g.swz));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int uu = 0; uu < 4; ++uu)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[uu][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1[uu], a1[tt], acc[uu][tt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    // epilogue (as K2f-B): D = (pixels x channels); lane holds 4 consecutive pixels of one channel per (uu, tt)
    if (bi == 0) PW_STAMP(4);
    if (has_stat) {
      __syncthreads();
      if (threadIdx.x < kStatSlots) k_stat[threadIdx.x] = 0u;
      __syncthreads();
    }
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    // the per-element arithmetic is specialised at compile time for the combinations the converted nets produce (no
    // bias + folded BN + ReLU / ReLU6 / linear): with run-time flags every output cost 14 VALU instructions (selects
    // after each optional step) instead of 6, and this kernel is instruction-bound
    auto epilogue = [&](auto bias_c, auto bn_c, auto act_c) {
    constexpr int BIAS_M = decltype(bias_c)::value, BN_M = decltype(bn_c)::value, ACT_M = decltype(act_c)::value;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int col = (bi * g.wm + wmi) * 64 + tt * 16 + (lane & 15);   // index into this workgroup's constants
      const int co = co0 + tt * 16 + (lane & 15);
      const bool co_ok = co < g.Cout;
      const float sxw = k_sxw[col];
      const float bch = k_bias[col], bsc = k_bsc[col], bsh = k_bsh[col];
      const int64_t coff = (int64_t)(co_ok ? co : 0) * plane_stride;
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = (float)acc[uu][tt][r] * sxw;
          if (BIAS_M == 1 || (BIAS_M < 0 && bias != nullptr)) v = v + bch;
          if (BN_M == 1 || (BN_M < 0 && has_bn)) {
            v = v * bsc;
            v = v + bsh;
          }
          o[r] = ACT_M < 0 ? act_rt(v, act) : act_rt(v, ACT_M);
        }
        if (co_ok && vec_ok[uu]) {
#ifdef FQ_PW_TRACE
          if (!(g_pw_dbg & 1))
#endif
          *reinterpret_cast<f4*>(y + ybase[uu] + coff) = (f4){o[0], o[1], o[2], o[3]};
          m[uu] = fmaxf(m[uu], fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        } else if (co_ok && cok[uu]) {
          // ragged: the 4 pixels may straddle a sample boundary or the end of the tensor.  The opaque asm keeps the
          // compiler from hoisting 16 copies of this address arithmetic out of the loops (it spilled them)
          unsigned jb = j0 + uu * 16 + 4 * (lane >> 4);
          asm volatile("" : "+v"(jb));
          unsigned sr = jb / HW;
          unsigned pr = jb - sr * HW;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (jb + r < (unsigned)g.cols) {
              while (pr >= HW) {
                pr -= HW;
                ++sr;
              }
              y[((int64_t)sr * g.Cout) * plane_stride + pr + coff] = o[r];
              if (sr == smps[uu]) m[uu] = fmaxf(m[uu], fabsf(o[r]));
              else if (has_stat) {
                const unsigned slot = sr - s_base;
                if (slot < kStatSlots) atomicMax(&k_stat[slot], __float_as_uint(fabsf(o[r])));
                else atomic_max_f32(stat_out + sr, fabsf(o[r]));
              }
            }
            ++pr;
          }
        }
      }
    }
    };
    using std::integral_constant;
#ifdef FQ_PW_GENERIC_EPI
    if (true)
      epilogue(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{});
    else
#endif
    if (bias == nullptr && has_bn && act == FQ_ACT_RELU)
      epilogue(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU>{});
    else if (bias == nullptr && has_bn && act == FQ_ACT_RELU6)
      epilogue(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_RELU6>{});
    else if (bias == nullptr && has_bn && act == FQ_ACT_NONE)
      epilogue(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, FQ_ACT_NONE>{});
    else
      epilogue(integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{});
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
        atomicMax(reinterpret_cast<unsigned*>(stat_out) + s_base + threadIdx.x, k_stat[threadIdx.x]);
    }
  }
  PW_STAMP(5);
}


}  // namespace

namespace fqi {

// panel form (K2g): one launch, codes transposed into an XOR-swizzled LDS panel
int pw_try_panel(const PwCall& a, bool* taken) {
  *taken = false;
  PwfGeom f;
  f.Cin = (int)a.cin; f.K = (int)a.cin_pad; f.Cout = (int)a.cout; f.HW = (int)a.hw; f.cols = a.n * a.hw;
  if (a.cout > 128) { f.wm = 4; f.wn = 1; }
  else if (a.cout > 64) { f.wm = 2; f.wn = 2; }
  else { f.wm = 1; f.wn = 4; }
  f.PX = 64 * f.wn;
  f.cblocks = (int)((a.cout + 64 * f.wm - 1) / (64 * f.wm));
  f.swz = (a.cin_pad % 128 == 0) ? 7 : 3;
  f.zoff = a.zoff;
  const int64_t ctiles = (f.cols + f.PX - 1) / f.PX;
  // split the channel blocks of one column tile over several workgroups while the launch would not fill the chip
  // (the panel is then quantised csplit times from activations that sit in L2 / the Infinity Cache)
  int csplit = 1;
  while (csplit < f.cblocks && ctiles * csplit < (int64_t)num_cu() * 3) csplit *= 2;
  if (csplit > f.cblocks) csplit = f.cblocks;
  f.csplit = csplit;
  f.nblk = (f.cblocks + csplit - 1) / csplit;
  const size_t lds = (size_t)f.PX * f.K + 16 + (size_t)f.nblk * f.wm * 64 * 5 * sizeof(float);
  // csplit >= 4 (few column tiles, e.g. 7x7 planes) re-quantises the panel too often: the two-kernel form with its
  // perfectly parallel quantise+transpose pass is faster there (tools/pwbench.py)
  if ((a.form == 2 || (a.form == 0 && csplit <= 2)) && lds <= 80 * 1024 - 256) {   // >= 2 workgroups per CU by LDS
    static const bool attr_ok = [] {
      const int lim = 80 * 1024 - 256;
      const void* fns[4] = {reinterpret_cast<const void*>(&pwconv_fused_kernel<true, 2>),
                            reinterpret_cast<const void*>(&pwconv_fused_kernel<true, 4>),
                            reinterpret_cast<const void*>(&pwconv_fused_kernel<false, 2>),
                            reinterpret_cast<const void*>(&pwconv_fused_kernel<false, 4>)};
      bool ok = true;
      for (const void* fn : fns) ok = ok && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess;
      return ok;
    }();
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the fused kernel");
    if (int rc = pw_zero_stat(a)) return rc;
    const int64_t grid = (ctiles + 7) / 8 * 8 * f.csplit;
    const int units = f.wn * (f.K / 64);
#define FQ_PWF_LAUNCH(ON, GG)                                                                                          \
  hipLaunchKernelGGL((pwconv_fused_kernel<ON, GG>), dim3((unsigned)grid), dim3(kBlock), lds, a.st, a.x, a.wcodes,      \
                     a.wscale, (const int*)a.wsum, a.bias, a.y, f, a.in_stat, (int)a.n, a.in_thr, a.levels, a.lo_neg,  \
                     kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out)
    if (a.in_stat) {
      if (units <= 4) FQ_PWF_LAUNCH(true, 2); else FQ_PWF_LAUNCH(true, 4);
    } else {
      if (units <= 4) FQ_PWF_LAUNCH(false, 2); else FQ_PWF_LAUNCH(false, 4);
    }
#undef FQ_PWF_LAUNCH
    FQ_LAUNCH_CHECK();
    *taken = true;
    return FQ_OK;
  }
  FQ_REQUIRE(a.form != 2, "fq_pwconv_i8: FQ_PW_FORM=2 but the panel needs %zu bytes of LDS", lds);
  return FQ_OK;
}

// two kernels (K2f): quantise + transpose to K-contiguous int8 in the workspace, then the 16x16x64 GEMM
int pw_two_kernels(const PwCall& a) {
  int8_t* codes = (int8_t*)a.ws;
  {
    const int ptiles = (int)((a.hw + 63) / 64), ctiles = (int)(a.cin_pad / 64);
    const int64_t tiles = a.n * ptiles * ctiles;
    const int grid = grid_for(tiles);
    if (a.in_stat)
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
  const float* sx_src = a.in_stat ? a.out_current_max : a.in_thr;
  hipLaunchKernelGGL((pwconv_i8_kernel<0>), dim3(grid), dim3(kBlock), 0, a.st, (const int8_t*)codes, a.wcodes, a.wscale,
                     (const int*)a.wsum, a.bias, a.y, g, tiles, sx_src, a.levels, a.bn_scale, a.bn_shift, a.act,
                     a.stat_out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // namespace fqi
