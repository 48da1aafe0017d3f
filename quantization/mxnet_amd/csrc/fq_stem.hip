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


}  // namespace

extern "C" {

int fq_stem_conv3x3s2(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n, int64_t cin,
                      int64_t cout, int64_t h, int64_t w, const float* bn_scale, const float* bn_shift, int act,
                      float* stat_out, fqStream_t stream) {
  FQ_REQUIRE(x && w_tap_major && y, "fq_stem_conv3x3s2: null pointer");
  FQ_REQUIRE(n > 0 && n < 65536 && h > 0 && w > 0 && h < (1 << 15) && w < (1 << 15), "fq_stem_conv3x3s2: bad shape");
  FQ_REQUIRE(cin == 3 && cout == 32, "fq_stem_conv3x3s2: only 3 -> 32 channels is built (got %lld -> %lld)",
             (long long)cin, (long long)cout);
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_stem_conv3x3s2: bn_scale and bn_shift go together");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_stem_conv3x3s2: unknown activation %d", act);
  hipStream_t st = (hipStream_t)stream;
  const int Ho = (int)((h + 2 - 3) / 2 + 1), Wo = (int)((w + 2 - 3) / 2 + 1);
  const int64_t hwo = (int64_t)Ho * Wo;
  if (stat_out && !prezeroed) FQ_HIP(hipMemsetAsync(stat_out, 0, n * sizeof(float), st));
  ProfScope prof(FQ_KERNEL_STEM, 4.0 * ((double)n * cin * h * w + (double)n * cout * hwo), st);
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

}  // extern "C"
