// libfakequant — K3 weight fake-quant, generic STE, K4 Winograd-domain weights
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// K3: weights, (rows, row_len).  Small rows: a workgroup stages several whole rows in LDS (one HBM read), reduces
// each row with a wavefront, then applies from LDS.  Long rows (layer mode): K1 per row + K3b apply.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kWTile = 8192;   // floats of LDS staging per workgroup (32 KiB)

__global__ __launch_bounds__(kBlock) void weight_rows_lds_kernel(const float* __restrict__ w,
                                                                 float* __restrict__ wq, int64_t rows, int row_len,
                                                                 int rows_per_block, float levels,
                                                                 float* __restrict__ scales_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                       // rows_per_block * row_len
  float* sc = smem + kWTile;                // rows_per_block scales
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int nrows = (int)((rows - r0) < rows_per_block ? (rows - r0) : rows_per_block);
  const int cnt = nrows * row_len;
  const float* src = w + r0 * row_len;
  float* dst = wq + r0 * row_len;
  const bool vec = ((row_len & 3) == 0) && ((((uintptr_t)src) & 15u) == 0) && ((((uintptr_t)dst) & 15u) == 0);
  if (vec) {
    const float4* p = reinterpret_cast<const float4*>(src);
    float4* t4 = reinterpret_cast<float4*>(tile);
    for (int i = threadIdx.x; i < cnt / 4; i += kBlock) t4[i] = p[i];
  } else {
    for (int i = threadIdx.x; i < cnt; i += kBlock) tile[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < nrows; r += kBlock / 64) {
    const float* row = tile + r * row_len;
    float m = 0.0f;
    for (int i = lane; i < row_len; i += 64) m = fmaxf(m, fabsf(row[i]));
    m = wave_max_nonneg(m);
    if (lane == 0) {
      const float s = m / levels;
      sc[r] = s;
      if (scales_out != nullptr) scales_out[r0 + r] = s;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cnt; i += kBlock) {
    const int r = i / row_len;
    const float s = sc[r];
    dst[i] = roundf(tile[i] / (s + kEps)) * s;
  }
}

// K3b: apply with per-row scale = rowmax[r] / levels (rows long enough that a chunk never spans two rows).
template <bool VEC>
__global__ __launch_bounds__(kBlock) void weight_apply_kernel(const float* __restrict__ w, float* __restrict__ wq,
                                                              int64_t row_len, int chunks_per_row,
                                                              int64_t total_chunks, const float* __restrict__ rowmax,
                                                              float levels, float* __restrict__ scales_out) {
  for (int64_t c = blockIdx.x; c < total_chunks; c += gridDim.x) {
    const int64_t r = c / chunks_per_row;
    const int64_t off0 = (c - r * chunks_per_row) * (int64_t)kChunk;
    const int64_t gbase = r * row_len + off0;
    const int64_t rem = row_len - off0;
    const float s = rowmax[r] / levels;
    const float d = s + kEps;
    if (scales_out != nullptr && off0 == 0 && threadIdx.x == 0) scales_out[r] = s;
    const int cnt = (int)(rem < kChunk ? rem : kChunk);
    if (VEC) {
      const float4* p = reinterpret_cast<const float4*>(w + gbase);
      float4* o = reinterpret_cast<float4*>(wq + gbase);
      for (int i = threadIdx.x; i < cnt / 4; i += kBlock) {
        float4 v = p[i];
        o[i] = make_float4(roundf(v.x / d) * s, roundf(v.y / d) * s, roundf(v.z / d) * s, roundf(v.w / d) * s);
      }
    } else {
      for (int i = threadIdx.x; i < cnt; i += kBlock) wq[gbase + i] = roundf(w[gbase + i] / d) * s;
    }
  }
}

// K3c: generic STE (per-row device scale, optional clip)
__global__ __launch_bounds__(kBlock) void ste_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     int64_t numel, int64_t row_len,
                                                     const float* __restrict__ scales, int has_clip, float lo,
                                                     float hi, float eps) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) {
    const float s = scales[i / row_len];
    float v = x[i];
    if (has_clip) v = fminf(fmaxf(v, lo), hi);
    y[i] = roundf(v / (s + eps)) * s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K4: Winograd-domain per-out-channel weight fake-quant.  One workgroup per output channel; each thread owns
// (ci) filters: U = G g G^T in registers (k-sequential, multiply and add separately rounded — oracle order).
// ---------------------------------------------------------------------------------------------------------------
struct WinoMats {
  float G[8 * 3];     // t x 3
  float GI[3 * 8];    // 3 x t
  float GTI[8 * 3];   // t x 3
};

template <int T>
__device__ __forceinline__ void wino_forward(const float* __restrict__ g9, const WinoMats& M, float (&U)[T][T]) {
  float t1[T][3];
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float acc = M.G[a * 3 + 0] * g9[0 * 3 + j];
      acc = acc + M.G[a * 3 + 1] * g9[1 * 3 + j];
      acc = acc + M.G[a * 3 + 2] * g9[2 * 3 + j];
      t1[a][j] = acc;
    }
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b) {
      float acc = t1[a][0] * M.G[b * 3 + 0];     // G^T[k][b] = G[b][k]
      acc = acc + t1[a][1] * M.G[b * 3 + 1];
      acc = acc + t1[a][2] * M.G[b * 3 + 2];
      U[a][b] = acc;
    }
}

template <int T>
__global__ __launch_bounds__(kBlock) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ wq,
                                                             int cin_g, WinoMats M, float levels,
                                                             float* __restrict__ scales_out) {
  __shared__ float red[4];
  __shared__ float s_scale;
  const int co = blockIdx.x;
  const float* wc = w + (int64_t)co * cin_g * 9;
  float* oc = wq + (int64_t)co * cin_g * 9;
  float m = 0.0f;
  for (int ci = threadIdx.x; ci < cin_g; ci += kBlock) {
    float g9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g9[k] = wc[ci * 9 + k];
    float U[T][T];
    wino_forward<T>(g9, M, U);
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int b = 0; b < T; ++b) m = fmaxf(m, fabsf(U[a][b]));
  }
  m = block_max(m, red);
  if (threadIdx.x == 0) {
    s_scale = m / levels;
    if (scales_out != nullptr) scales_out[co] = s_scale;
  }
  __syncthreads();
  const float s = s_scale;
  const float d = s + kEps;
  for (int ci = threadIdx.x; ci < cin_g; ci += kBlock) {
    float g9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g9[k] = wc[ci * 9 + k];
    float U[T][T];
    wino_forward<T>(g9, M, U);
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int b = 0; b < T; ++b) U[a][b] = roundf(U[a][b] / d) * s;
    // back: t2 = GI (3 x T) . Uq (T x T);  g = t2 (3 x T) . GTI (T x 3)
    float t2[3][T];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int b = 0; b < T; ++b) {
        float acc = M.GI[i * T + 0] * U[0][b];
#pragma unroll
        for (int a = 1; a < T; ++a) acc = acc + M.GI[i * T + a] * U[a][b];
        t2[i][b] = acc;
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float acc = t2[i][0] * M.GTI[0 * 3 + j];
#pragma unroll
        for (int b = 1; b < T; ++b) acc = acc + t2[i][b] * M.GTI[b * 3 + j];
        oc[ci * 9 + i * 3 + j] = acc;
      }
  }
}


}  // namespace

extern "C" {

int fq_ste_forward(const float* x, float* y, int64_t rows, int64_t row_len, const float* scales, int has_clip,
                   float clip_lo, float clip_hi, float eps, fqStream_t stream) {
  FQ_REQUIRE(x && y && scales, "fq_ste_forward: null pointer");
  FQ_REQUIRE(rows > 0 && row_len > 0, "fq_ste_forward: empty tensor");
  const int64_t numel = rows * row_len;
  const int grid = grid_for((numel + kBlock * 4 - 1) / (kBlock * 4));
  hipLaunchKernelGGL(ste_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, x, y, numel, row_len, scales,
                     has_clip, clip_lo, clip_hi, eps);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

size_t fq_weight_workspace_bytes(int64_t rows) { return (size_t)(rows < 1 ? 1 : rows) * sizeof(float) + 64; }

int fq_weight_fake_quant(const float* w, float* w_q, int64_t rows, int64_t row_len, int width, float* scales_out,
                         void* ws, fqStream_t stream) {
  FQ_REQUIRE(w && w_q, "fq_weight_fake_quant: null pointer");
  FQ_REQUIRE(rows > 0 && row_len > 0, "fq_weight_fake_quant: empty tensor (rows=%lld row_len=%lld)",
             (long long)rows, (long long)row_len);
  FQ_REQUIRE(width >= 2 && width <= 16, "fq_weight_fake_quant: width %d out of range", width);
  hipStream_t st = (hipStream_t)stream;
  const float levels = (float)((1 << (width - 1)) - 1);
  if (row_len <= kWTile) {
    int rpb = (int)(kWTile / row_len);
    if (rpb > 1024) rpb = 1024;
    // keep >= ~2 workgroups per CU busy when there are many short rows
    const int64_t want_blocks = (int64_t)num_cu() * 2;
    int64_t balanced = (rows + want_blocks - 1) / want_blocks;
    if (balanced < 1) balanced = 1;
    if (rpb > balanced) rpb = (int)balanced;
    const int64_t blocks = (rows + rpb - 1) / rpb;
    const size_t lds = (size_t)(kWTile + 1024) * sizeof(float);
    ProfScope prof(FQ_KERNEL_WEIGHT, 8.0 * (double)rows * (double)row_len, st);
    hipLaunchKernelGGL(weight_rows_lds_kernel, dim3((unsigned)blocks), dim3(kBlock), lds, st, w, w_q, rows,
                       (int)row_len, rpb, levels, scales_out);
    FQ_LAUNCH_CHECK();
    return FQ_OK;
  }
  FQ_REQUIRE(ws, "fq_weight_fake_quant: workspace required for rows longer than %d", kWTile);
  float* rowmax = (float*)ws;
  FQ_HIP(hipMemsetAsync(rowmax, 0, rows * sizeof(float), st));
  if (int rc = launch_absmax(w, rows, row_len, true, rowmax, st)) return rc;
  const Chunking ck = chunking(rows, row_len);
  const bool vec = (row_len % kVec == 0) && aligned16(w) && aligned16(w_q);
  const int grid = grid_for(ck.total);
  if (vec)
    hipLaunchKernelGGL((weight_apply_kernel<true>), dim3(grid), dim3(kBlock), 0, st, w, w_q, row_len,
                       ck.chunks_per_sample, ck.total, rowmax, levels, scales_out);
  else
    hipLaunchKernelGGL((weight_apply_kernel<false>), dim3(grid), dim3(kBlock), 0, st, w, w_q, row_len,
                       ck.chunks_per_sample, ck.total, rowmax, levels, scales_out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

int fq_wino_weight_fake_quant(const float* w, float* w_q, int64_t cout, int64_t cin_g, int t, const float* G,
                              const float* GI, const float* GTI, int width, float* scales_out, void* ws,
                              fqStream_t stream) {
  (void)ws;
  FQ_REQUIRE(w && w_q && G && GI && GTI, "fq_wino_weight_fake_quant: null pointer");
  FQ_REQUIRE(t == 4 || t == 6 || t == 8, "fq_wino_weight_fake_quant: t must be 4 (F23), 6 (F43) or 8 (F63), got %d", t);
  FQ_REQUIRE(cout > 0 && cin_g > 0 && cin_g < (1ll << 28), "fq_wino_weight_fake_quant: bad shape");
  FQ_REQUIRE(width >= 2 && width <= 16, "fq_wino_weight_fake_quant: width %d out of range", width);
  WinoMats M;
  memset(&M, 0, sizeof(M));
  memcpy(M.G, G, sizeof(float) * t * 3);
  memcpy(M.GI, GI, sizeof(float) * 3 * t);
  memcpy(M.GTI, GTI, sizeof(float) * t * 3);
  const float levels = (float)((1 << (width - 1)) - 1);
  hipStream_t st = (hipStream_t)stream;
  if (t == 4)
    hipLaunchKernelGGL((wino_weight_kernel<4>), dim3((unsigned)cout), dim3(kBlock), 0, st, w, w_q, (int)cin_g, M,
                       levels, scales_out);
  else if (t == 6)
    hipLaunchKernelGGL((wino_weight_kernel<6>), dim3((unsigned)cout), dim3(kBlock), 0, st, w, w_q, (int)cin_g, M,
                       levels, scales_out);
  else
    hipLaunchKernelGGL((wino_weight_kernel<8>), dim3((unsigned)cout), dim3(kBlock), 0, st, w, w_q, (int)cin_g, M,
                       levels, scales_out);
  FQ_LAUNCH_CHECK();
  return FQ_OK;
}

}  // extern "C"
