// libfakequant — K9 int-code quantise / dequantise, K10 exact int8 GEMM
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// K9: int-code quantise / dequantise
// ---------------------------------------------------------------------------------------------------------------
__global__ void codes_range_kernel(float* __restrict__ range, int mode, const float* __restrict__ ws) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float mn, mx;
  if (mode == FQ_CODES_INT8) {
    mx = ws[1];
    mn = -mx;
  } else if (mode == FQ_CODES_UINT8) {
    mn = ws[0];
    mx = ws[1];
  } else {
    mn = range[0];
    mx = range[1];
  }
  range[0] = mn;
  range[1] = mx;
  if (mode != FQ_CODES_SCALE) range[2] = (mx == -mn) ? (mx / 127.0f) : ((mx - mn) / 255.0f);
}

__global__ __launch_bounds__(kBlock) void quantize_codes_kernel(const float* __restrict__ x,
                                                                int32_t* __restrict__ codes, int64_t numel,
                                                                const float* __restrict__ range) {
  const float mn = range[0], mx = range[1], sc = range[2];
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) {
    const float c = fminf(fmaxf(x[i], mn), mx);
    codes[i] = (int32_t)roundf(c / sc);
  }
}

__global__ __launch_bounds__(kBlock) void dequantize_kernel(const int32_t* __restrict__ codes, float* __restrict__ y,
                                                            int64_t numel, const float* __restrict__ scale) {
  const float sc = scale[0];
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) y[i] = (float)codes[i] * sc;
}


// ---------------------------------------------------------------------------------------------------------------
// K10: integer GEMM on 8-bit codes with exact int32 results - the arithmetic core of the reference's stand-alone
// quantised convolution (nn/quantized_conv.py:134-151: im2col slices x reshaped filters, accumulated as integers).
//   out[n][co][p] = sum_k xc[n*L + p][k] * wc[co][k]  (+ zoff * wsum[co] when the activation codes were stored
//   re-centred by zoff = 128 to fit int8)
// xc: [cols_pad][K] int8, K-contiguous im2col rows (K % 32 == 0, zero padded);  wc: [rows_pad][K] int8.
// v_mfma_i32_32x32x32_i8 with the activation rows as the B operand: lane = pixel, so every store instruction writes two
// full 128-byte lines of the NCHW result.  Lanes past the last column re-read and re-store the last one (benign).
// Correctness first (this block is only exercised by the reference's tests): operands come straight from global / L2.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gemm_i8_codes_kernel(const int8_t* __restrict__ xc,
                                                               const int8_t* __restrict__ wc,
                                                               const int* __restrict__ wsum, int* __restrict__ out,
                                                               int64_t cols, int L, int K, int Cout, int zoff) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, pl = lane & 31;
  const int64_t tiles = (cols + 31) / 32;
  const int KT = K >> 5, CT = (Cout + 31) / 32;
  for (int64_t t = (int64_t)blockIdx.x * (kBlock / 64) + wave; t < tiles; t += (int64_t)gridDim.x * (kBlock / 64)) {
    int64_t col = t * 32 + pl;
    col = col < cols ? col : cols - 1;
    const int64_t smp = col / L;
    const int p = (int)(col - smp * L);
    const int8_t* xrow = xc + col * K + 16 * h;
    int* obase = out + (smp * Cout) * (int64_t)L + p;
    for (int ct = 0; ct < CT; ++ct) {
      v16i acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = ct * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        acc[r] = ch < Cout ? zoff * wsum[ch] : 0;
      }
      const int8_t* wrow = wc + (int64_t)(ct * 32 + pl) * K + 16 * h;     // A fragment: row pl, 16-byte half h
      for (int kt = 0; kt < KT; ++kt) {
        const v4i a = *reinterpret_cast<const v4i*>(wrow + kt * 32);
        const v4i b = *reinterpret_cast<const v4i*>(xrow + kt * 32);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = ct * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        if (ch < Cout) obase[(int64_t)ch * L] = acc[r];
      }
    }
  }
}


}  // namespace

extern "C" {

int fq_quantize_codes(const float* x, int32_t* codes, int64_t numel, int mode, float* range_dev, void* ws,
                      fqStream_t stream) {
  FQ_REQUIRE(x && codes && range_dev, "fq_quantize_codes: null pointer");
  FQ_REQUIRE(numel > 0, "fq_quantize_codes: empty tensor");
  FQ_REQUIRE(mode >= FQ_CODES_INT8 && mode <= FQ_CODES_SCALE, "unknown out type: %d", mode);
  hipStream_t st = (hipStream_t)stream;
  float* mm = (float*)ws;   // {min, max}
  const int grid = grid_for((numel + kChunk - 1) / kChunk);
  if (mode == FQ_CODES_INT8 || mode == FQ_CODES_UINT8) {
    FQ_REQUIRE(ws, "fq_quantize_codes: workspace required");
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, st, mm, (int64_t)1, INFINITY);
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, st, mm + 1, (int64_t)1,
                       mode == FQ_CODES_INT8 ? 0.0f : -INFINITY);
    if (mode == FQ_CODES_INT8)
      hipLaunchKernelGGL((minmax_kernel<false, true>), dim3(grid), dim3(kBlock), 0, st, x, numel,
                         aligned16(x) ? 1 : 0, mm, mm + 1);
    else
      hipLaunchKernelGGL((minmax_kernel<true, false>), dim3(grid), dim3(kBlock), 0, st, x, numel,
                         aligned16(x) ? 1 : 0, mm, mm + 1);
    FQ_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(codes_range_kernel, dim3(1), dim3(64), 0, st, range_dev, mode, mm);
  hipLaunchKernelGGL(quantize_codes_kernel, dim3(grid), dim3(kBlock), 0, st, x, codes, numel, range_dev);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_dequantize(const int32_t* codes, float* y, int64_t numel, const float* scale_dev, fqStream_t stream) {
  FQ_REQUIRE(codes && y && scale_dev, "fq_dequantize: null pointer");
  FQ_REQUIRE(numel > 0, "fq_dequantize: empty tensor");
  const int grid = grid_for((numel + kChunk - 1) / kChunk);
  hipLaunchKernelGGL(dequantize_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, codes, y, numel,
                     scale_dev);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_gemm_i8_codes(const int8_t* xcodes, const int8_t* wcodes, const int32_t* wsum, int32_t* out, int64_t n,
                     int64_t l, int64_t k_pad, int64_t cout, int zoff, fqStream_t stream) {
  FQ_REQUIRE(xcodes && wcodes && wsum && out, "fq_gemm_i8_codes: null pointer");
  FQ_REQUIRE(n > 0 && l > 0 && cout > 0 && k_pad > 0 && k_pad % 32 == 0 && n * l < (1ll << 40) && l < (1ll << 31),
             "fq_gemm_i8_codes: bad shape (k_pad=%lld must be a positive multiple of 32)", (long long)k_pad);
  FQ_REQUIRE(zoff == 0 || zoff == 128, "fq_gemm_i8_codes: zoff must be 0 (signed codes) or 128 (re-centred unsigned)");
  FQ_REQUIRE(aligned16(xcodes) && aligned16(wcodes), "fq_gemm_i8_codes: code buffers must be 16-byte aligned");
  const int64_t tiles = (n * l + 31) / 32;
  int64_t grid = (tiles + (kBlock / 64) - 1) / (kBlock / 64);
  if (grid > (int64_t)num_cu() * 16) grid = (int64_t)num_cu() * 16;
  hipLaunchKernelGGL(gemm_i8_codes_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, xcodes, wcodes,
                     (const int*)wsum, (int*)out, n * l, (int)l, (int)k_pad, (int)cout, zoff);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // extern "C"
