// CPU ORACLE (C++ / OpenMP) — test infrastructure, NOT product code.
//
// A restatement of the reference's fake-quantisation / calibration algorithm for the host cores, exported with the
// signatures of include/fakequant.h plus a `_host` suffix (include/fakequant_host.h).  Written from the reference's
// Python, function by function (file:line cited at each entry point; /root/reference = hey-yahei/Quantization.MXNet) and
// from oracle/fq_oracle.py, against which it is pinned bit-for-bit in tests/test_host_oracle.py (which in turn is pinned
// against the golden vectors made from the reference's own code, tests/golden/).
//
// Who may load oracle/libfq_host.so: tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg — as the checker
// and as the timed CPU baseline, never as a fallback of the product (quantization/mxnet_amd refuses host tensors).
//
// Primitive semantics: fp32 everywhere unless stated; round = C roundf (half AWAY from zero); tensor /
// scalar = IEEE fp32 divide; clip(a, lo, hi) = min(max(a, lo), hi); cast to int32 truncates; batch mean =
// fp32(sequential fp64 sum) / fp32(N).  Compiled with -ffp-contract=off -fno-fast-math: no fusion, no re-association.
// OpenMP only ever splits work whose result does not depend on the order (max, integer counts, independent elements /
// rows / candidates); every order-dependent sum runs sequentially inside one thread, in the reference's order.
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "fakequant_host.h"

namespace {

thread_local char g_err[512] = "";
int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return FQ_ERR_INVALID;
}
#define REQUIRE(cond, ...)            \
  do {                                \
    if (!(cond)) return fail(__VA_ARGS__); \
  } while (0)

constexpr float kEps = 1e-10f;   // ste_func.py:39,41

inline float clipf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// Threads for a loop over `work` elements: small tensors are not worth waking 256 threads for (a fork/join of a full team
// costs tens of microseconds; the tail layers hold a few hundred thousand elements).
inline int team(int64_t work) {
  const int64_t want = work / 32768;
  const int mx = omp_get_max_threads();
  return (int)(want < 1 ? 1 : (want > mx ? mx : want));
}

// C roundf (half away from zero) in a form the compiler vectorises: x - trunc(x) is exact, so comparing it with 0.5 decides
// the tie exactly as roundf does; NaN and infinities pass through (the comparison is false).
inline float round_half_away(float x) {
  const float t = __builtin_truncf(x);
  return (fabsf(x - t) >= 0.5f) ? t + __builtin_copysignf(1.0f, x) : t;
}

inline float act_levels(int width, unsigned flags) {
  return (flags & FQ_ACT_SIGNED) ? (float)((1 << (width - 1)) - 1) : (float)((1 << width) - 1);
}

struct QP {
  float lo, hi, denom, scale;
};
// convert_conv2d.py:59-64 + ste_func.py:41
inline QP make_qp(float max_, float levels, bool lo_neg_max, float eps) {
  QP q;
  q.hi = max_;
  q.lo = lo_neg_max ? -max_ : 0.0f;
  q.scale = max_ / levels;
  q.denom = q.scale + eps;
  return q;
}
inline float code_of(float x, const QP& q) { return round_half_away(clipf(x, q.lo, q.hi) / q.denom); }

inline float act_of(float v, int act) {
  if (act == FQ_ACT_RELU) v = fmaxf(v, 0.0f);
  if (act == FQ_ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
  return v;
}

// `.mean()` of the per-sample maxima (convert_conv2d.py:56): fp32(sum in fp64, sample order) / fp32(n)
inline float batch_mean(const float* v, int64_t n) {
  double acc = 0.0;
  for (int64_t i = 0; i < n; ++i) acc += (double)v[i];
  return (float)acc / (float)n;
}

void per_sample_stat(const float* x, int64_t n, int64_t inner, bool use_abs, float* out) {
  // F.max(F.abs(x), axis=(1,2,3)) (convert_conv2d.py:56); without abs for convert_act.py:50
  const int64_t kPiece = 1 << 16;
  const int64_t pieces = (inner + kPiece - 1) / kPiece;
  const float init = use_abs ? 0.0f : -INFINITY;
  std::vector<float> part((size_t)(n * pieces));
#pragma omp parallel for collapse(2) schedule(static) num_threads(team(n * inner))
  for (int64_t s = 0; s < n; ++s)
    for (int64_t p = 0; p < pieces; ++p) {
      const float* b = x + s * inner + p * kPiece;
      const int64_t cnt = std::min(kPiece, inner - p * kPiece);
      float m = init;
      if (use_abs)
        for (int64_t i = 0; i < cnt; ++i) m = fmaxf(m, fabsf(b[i]));
      else
        for (int64_t i = 0; i < cnt; ++i) m = fmaxf(m, b[i]);
      part[(size_t)(s * pieces + p)] = m;
    }
  for (int64_t s = 0; s < n; ++s) {
    float m = init;
    for (int64_t p = 0; p < pieces; ++p) m = fmaxf(m, part[(size_t)(s * pieces + p)]);
    out[s] = m;
  }
}

void apply_quant(const float* x, float* y, int32_t* codes, int64_t numel, const QP& q) {
#pragma omp parallel for schedule(static) num_threads(team(numel))
  for (int64_t i = 0; i < numel; ++i) {
    const float k = code_of(x[i], q);
    if (codes) codes[i] = (int32_t)k;
    y[i] = k * q.scale;
  }
}

void stat_of_output(const float* y, int64_t n, int64_t inner, float* stat_out) {
  if (stat_out == nullptr) return;
  std::vector<float> m((size_t)n);
  per_sample_stat(y, n, inner, true, m.data());
  for (int64_t s = 0; s < n; ++s) stat_out[s] = fmaxf(stat_out[s], m[(size_t)s]);
}

inline void zero_stat(float* stat_out, int64_t n, bool prezeroed) {
  if (stat_out && !prezeroed)
    for (int64_t s = 0; s < n; ++s) stat_out[s] = 0.0f;
}

}  // namespace

extern "C" {

const char* fq_last_error_host(void) { return g_err; }
int fq_version_host(void) { return 100; }
int fq_threads_host(void) { return omp_get_max_threads(); }
int fq_set_threads_host(int k) {
  omp_set_num_threads(k > 0 ? k : omp_get_num_procs());
  return FQ_OK;
}

// ---- activations ------------------------------------------------------------------------------------------------
int fq_absmax_per_sample_host(const float* x, int64_t n, int64_t inner, unsigned flags, float* out_max, fqStream_t) {
  REQUIRE(x && out_max, "fq_absmax_per_sample_host: null pointer");
  REQUIRE(n > 0 && inner > 0, "fq_absmax_per_sample_host: empty tensor");
  per_sample_stat(x, n, inner, !(flags & FQ_ACT_NO_ABS), out_max);
  return FQ_OK;
}

int fq_batch_mean_host(const float* v, int64_t n, float* out, fqStream_t) {
  REQUIRE(v && out && n > 0, "fq_batch_mean_host: bad arguments");
  out[0] = batch_mean(v, n);
  return FQ_OK;
}

int fq_batch_mean_rows_host(const float* v, int64_t rows, int64_t n, int64_t row_stride, float* out, fqStream_t) {
  REQUIRE(v && out && rows > 0 && n > 0 && row_stride >= n, "fq_batch_mean_rows_host: bad arguments");
  for (int64_t r = 0; r < rows; ++r) out[r] = batch_mean(v + r * row_stride, n);
  return FQ_OK;
}

int fq_stat_rows_sum_host(const float* v, int64_t rows, int64_t n, int64_t row_stride, double* out, fqStream_t) {
  REQUIRE(v && out && rows > 0 && n >= 0 && row_stride >= n, "fq_stat_rows_sum_host: bad arguments");
  for (int64_t r = 0; r < rows; ++r) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) acc += (double)v[r * row_stride + i];
    out[r] = acc;
  }
  out[rows] = (double)n;
  return FQ_OK;
}

int fq_mean_from_sums_host(const double* sums, int64_t rows, float* out, fqStream_t) {
  REQUIRE(sums && out && rows > 0, "fq_mean_from_sums_host: bad arguments");
  for (int64_t r = 0; r < rows; ++r) out[r] = (float)sums[r] / (float)sums[rows];
  return FQ_OK;
}

int fq_batch_mean_gathered_host(const float* packs, int world, int64_t stride, float* out, fqStream_t) {
  REQUIRE(packs && out && world > 0 && stride > 1, "fq_batch_mean_gathered_host: bad arguments");
  double acc = 0.0;
  long long total = 0;
  for (int w = 0; w < world; ++w) {
    const float* rec = packs + (int64_t)w * stride;
    const int c = (int)rec[0];
    for (int i = 0; i < c; ++i) acc += (double)rec[1 + i];
    total += c;
  }
  out[0] = (float)acc / (float)total;
  return FQ_OK;
}

// convert_conv2d.py:53-66 (conv), convert_dense.py:39-49 (Dense: FQ_ACT_LO_NEG_MAX never set), convert_act.py:49-54
// (FQ_ACT_NO_ABS | FQ_ACT_NO_EPS) + ste_func.py:41
int fq_fake_quant_online_host(const float* x, float* y, int64_t n, int64_t inner, int width, unsigned flags,
                              float* out_current_max, int32_t* codes, void*, fqStream_t) {
  REQUIRE(x && y, "fq_fake_quant_online_host: null pointer");
  REQUIRE(n > 0 && inner > 0 && width >= 2 && width <= 16, "fq_fake_quant_online_host: bad arguments");
  std::vector<float> stat((size_t)n);
  per_sample_stat(x, n, inner, !(flags & FQ_ACT_NO_ABS), stat.data());
  const float max_ = batch_mean(stat.data(), n);
  if (out_current_max) out_current_max[0] = max_;
  const QP q = make_qp(max_, act_levels(width, flags), (flags & FQ_ACT_LO_NEG_MAX) != 0,
                       (flags & FQ_ACT_NO_EPS) ? 0.0f : kEps);
  apply_quant(x, y, codes, n * inner, q);
  return FQ_OK;
}

int fq_fake_quant_online_prestat_host(const float* x, float* y, int64_t n, int64_t inner, const float* stat, int width,
                                      unsigned flags, float* out_current_max, int32_t* codes, fqStream_t) {
  REQUIRE(x && y && stat, "fq_fake_quant_online_prestat_host: null pointer");
  REQUIRE(n > 0 && inner > 0 && width >= 2 && width <= 16, "fq_fake_quant_online_prestat_host: bad arguments");
  const float max_ = batch_mean(stat, n);
  if (out_current_max) out_current_max[0] = max_;
  const QP q = make_qp(max_, act_levels(width, flags), (flags & FQ_ACT_LO_NEG_MAX) != 0,
                       (flags & FQ_ACT_NO_EPS) ? 0.0f : kEps);
  apply_quant(x, y, codes, n * inner, q);
  return FQ_OK;
}

int fq_fake_quant_offline_host(const float* x, float* y, int64_t n, int64_t inner, const float* threshold, int width,
                               unsigned flags, float* out_current_max, int32_t* codes, void*, fqStream_t) {
  REQUIRE(x && y && threshold, "fq_fake_quant_offline_host: null pointer");
  REQUIRE(n > 0 && inner > 0 && width >= 2 && width <= 16, "fq_fake_quant_offline_host: bad arguments");
  if (out_current_max) {   // the reference computes the batch statistic in every mode (convert_conv2d.py:56)
    std::vector<float> stat((size_t)n);
    per_sample_stat(x, n, inner, !(flags & FQ_ACT_NO_ABS), stat.data());
    out_current_max[0] = batch_mean(stat.data(), n);
  }
  const QP q = make_qp(threshold[0], act_levels(width, flags), (flags & FQ_ACT_LO_NEG_MAX) != 0,
                       (flags & FQ_ACT_NO_EPS) ? 0.0f : kEps);
  apply_quant(x, y, codes, n * inner, q);
  return FQ_OK;
}

// The reference's op chain as separate passes with temporaries (the baseline workload, see the header)
int fq_unfused_chain_host(const float* x, float* y, int64_t n, int64_t inner, int width, unsigned flags,
                          float* out_current_max, float* tmp, fqStream_t) {
  REQUIRE(x && y && n > 0 && inner > 0, "fq_unfused_chain_host: bad arguments");
  const int64_t numel = n * inner;
  std::vector<float> own;
  if (tmp == nullptr) {
    own.resize((size_t)numel * 2);
    tmp = own.data();
  }
  float* a = tmp;
  float* b = tmp + numel;
#pragma omp parallel for schedule(static) num_threads(team(numel))
  for (int64_t i = 0; i < numel; ++i) a[i] = fabsf(x[i]);                     // F.abs(x)
  std::vector<float> stat((size_t)n);
  per_sample_stat(a, n, inner, false, stat.data());                            // F.max(..., axis=(1,2,3))
  const float max_ = batch_mean(stat.data(), n);                               // .mean().asscalar()
  if (out_current_max) out_current_max[0] = max_;
  const QP q = make_qp(max_, act_levels(width, flags), (flags & FQ_ACT_LO_NEG_MAX) != 0, kEps);
#pragma omp parallel for schedule(static) num_threads(team(numel))
  for (int64_t i = 0; i < numel; ++i) a[i] = clipf(x[i], q.lo, q.hi);         // x.clip(min_, max_)
#pragma omp parallel for schedule(static) num_threads(team(numel))
  for (int64_t i = 0; i < numel; ++i) b[i] = a[i] / q.denom;                  // / (scale + 1e-10)
#pragma omp parallel for schedule(static) num_threads(team(numel))
  for (int64_t i = 0; i < numel; ++i) a[i] = round_half_away(b[i]);                    // .round()
#pragma omp parallel for schedule(static) num_threads(team(numel))
  for (int64_t i = 0; i < numel; ++i) y[i] = a[i] * q.scale;                  // * scale
  return FQ_OK;
}

// ---- fused producers (project additions; arithmetic of oracle.bn_act / global_avg_pool / stem / dwconv / pwconv) ----
int fq_bn_act_stat_host(const float* x, float* y, int64_t n, int64_t c, int64_t hw, const float* scale,
                        const float* shift, int act, float* stat_out, fqStream_t) {
  REQUIRE(x && y && scale && shift && n > 0 && c > 0 && hw > 0, "fq_bn_act_stat_host: bad arguments");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  zero_stat(stat_out, n, prezeroed);
#pragma omp parallel for schedule(static)
  for (int64_t pl = 0; pl < n * c; ++pl) {
    const float sc = scale[pl % c], sh = shift[pl % c];
    for (int64_t i = 0; i < hw; ++i) {
      float r = x[pl * hw + i] * sc;
      r = r + sh;
      y[pl * hw + i] = act_of(r, act);
    }
  }
  stat_of_output(y, n, c * hw, stat_out);
  return FQ_OK;
}

// BatchNorm + activation, then MaxPool2D(3, stride 2, padding 1): padding never wins
int fq_bn_act_maxpool_stat_host(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w,
                                const float* scale, const float* shift, int act, float* stat_out, fqStream_t) {
  REQUIRE(x && y && scale && shift && n > 0 && c > 0 && h > 0 && w > 0, "fq_bn_act_maxpool_stat_host: bad arguments");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  zero_stat(stat_out, n, prezeroed);
  const int64_t ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
#pragma omp parallel for schedule(static)
  for (int64_t pl = 0; pl < n * c; ++pl) {
    const float sc = scale[pl % c], sh = shift[pl % c];
    for (int64_t r = 0; r < ho; ++r)
      for (int64_t q = 0; q < wo; ++q) {
        float m = -INFINITY;
        for (int64_t ky = 0; ky < 3; ++ky)
          for (int64_t kx = 0; kx < 3; ++kx) {
            const int64_t iy = 2 * r - 1 + ky, ix = 2 * q - 1 + kx;
            if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
            float t = x[(pl * h + iy) * w + ix] * sc;
            t = t + sh;
            m = std::max(m, act_of(t, act));
          }
        y[(pl * ho + r) * wo + q] = m;
      }
  }
  stat_of_output(y, n, c * ho * wo, stat_out);
  return FQ_OK;
}

int fq_add_act_stat_host(const float* a, const float* b, float* y, int64_t n, int64_t inner, int act, float* stat_out,
                         fqStream_t) {
  REQUIRE(a && b && y && n > 0 && inner > 0, "fq_add_act_stat_host: bad arguments");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  zero_stat(stat_out, n, prezeroed);
#pragma omp parallel for schedule(static) num_threads(team(n * inner))
  for (int64_t i = 0; i < n * inner; ++i) y[i] = act_of(a[i] + b[i], act);
  stat_of_output(y, n, inner, stat_out);
  return FQ_OK;
}

int fq_global_avg_pool_stat_host(const float* x, float* y, int64_t n, int64_t c, int64_t hw, int flags,
                                 float* stat_out, fqStream_t) {
  REQUIRE(x && y && n > 0 && c > 0 && hw > 0, "fq_global_avg_pool_stat_host: bad arguments");
  zero_stat(stat_out, n, (flags & FQ_STAT_PREZEROED) != 0);
#pragma omp parallel for schedule(static)
  for (int64_t pl = 0; pl < n * c; ++pl) {
    double acc = 0.0;
    for (int64_t i = 0; i < hw; ++i) acc += (double)x[pl * hw + i];
    y[pl] = (float)acc / (float)hw;
  }
  stat_of_output(y, n, c, stat_out);
  return FQ_OK;
}

int fq_gemm_i8_codes_host(const int8_t* xcodes, const int8_t* wcodes, const int32_t* wsum, int32_t* out, int64_t n,
                          int64_t l, int64_t k_pad, int64_t cout, int zoff, fqStream_t) {
  REQUIRE(xcodes && wcodes && wsum && out && n > 0 && l > 0 && cout > 0 && k_pad > 0, "fq_gemm_i8_codes_host: bad "
          "arguments");
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t s = 0; s < n; ++s)
    for (int64_t co = 0; co < cout; ++co) {
      const int8_t* wr = wcodes + co * k_pad;
      for (int64_t p = 0; p < l; ++p) {
        const int8_t* xr = xcodes + (s * l + p) * k_pad;
        int32_t acc = zoff * wsum[co];
        for (int64_t k = 0; k < k_pad; ++k) acc += (int32_t)xr[k] * (int32_t)wr[k];
        out[(s * cout + co) * l + p] = acc;
      }
    }
  return FQ_OK;
}

// examples/simulate_quantization.py:122-148
int fq_eval_counters_host(const float* logits, const int64_t* labels, int64_t n, int64_t classes, float* counters,
                          fqStream_t) {
  REQUIRE(logits && labels && counters && n > 0 && classes > 0, "fq_eval_counters_host: bad arguments");
  for (int64_t s = 0; s < n; ++s) {
    const float* row = logits + s * classes;
    int64_t best = 0;
    for (int64_t i = 1; i < classes; ++i) {       // argmax: first index among equal maxima, NaN is the maximum
      const bool bn = row[best] != row[best], vn = row[i] != row[i];
      if ((vn && !bn) || (!bn && !vn && row[i] > row[best])) best = i;
    }
    const int64_t gt = labels[s];
    counters[1] += 1.0f;
    if (gt >= 0 && gt < classes) {
      counters[2 + classes + gt] += 1.0f;
      if (best == gt) {
        counters[0] += 1.0f;
        counters[2 + gt] += 1.0f;
      }
    }
  }
  return FQ_OK;
}

// first convolution K x K, stride 2, padding K / 2: an fmaf chain over (ci, ky, kx), then bias, folded BN, activation
static int stem_conv_impl(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n, int64_t cin,
                          int64_t cout, int64_t h, int64_t w, int ks, const float* bn_scale, const float* bn_shift, int act,
                          float* stat_out) {
  REQUIRE(x && w_tap_major && y && n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "fq_stem_conv_host: bad arguments");
  REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_stem_conv_host: bn_scale and bn_shift go together");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  zero_stat(stat_out, n, prezeroed);
  const int64_t pad = ks / 2;
  const int64_t ho = (h + 2 * pad - ks) / 2 + 1, wo = (w + 2 * pad - ks) / 2 + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t s = 0; s < n; ++s)
    for (int64_t co = 0; co < cout; ++co)
      for (int64_t oy = 0; oy < ho; ++oy)
        for (int64_t ox = 0; ox < wo; ++ox) {
          float acc = 0.0f;
          for (int64_t ci = 0; ci < cin; ++ci)
            for (int ky = 0; ky < ks; ++ky)
              for (int kx = 0; kx < ks; ++kx) {
                const int64_t iy = oy * 2 - pad + ky, ix = ox * 2 - pad + kx;
                const float v = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? x[((s * cin + ci) * h + iy) * w + ix] : 0.0f;
                acc = __builtin_fmaf(w_tap_major[((ci * ks + ky) * ks + kx) * cout + co], v, acc);
              }
          if (bias) acc = acc + bias[co];
          if (bn_scale) {
            acc = acc * bn_scale[co];
            acc = acc + bn_shift[co];
          }
          y[((s * cout + co) * ho + oy) * wo + ox] = act_of(acc, act);
        }
  stat_of_output(y, n, cout * ho * wo, stat_out);
  return FQ_OK;
}

int fq_stem_conv3x3s2_host(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n,
                           int64_t cin, int64_t cout, int64_t h, int64_t w, const float* bn_scale,
                           const float* bn_shift, int act, float* stat_out, fqStream_t) {
  return stem_conv_impl(x, w_tap_major, bias, y, n, cin, cout, h, w, 3, bn_scale, bn_shift, act, stat_out);
}

int fq_stem_conv7x7s2_host(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n,
                           int64_t cin, int64_t cout, int64_t h, int64_t w, const float* bn_scale,
                           const float* bn_shift, int act, float* stat_out, fqStream_t) {
  return stem_conv_impl(x, w_tap_major, bias, y, n, cin, cout, h, w, 7, bn_scale, bn_shift, act, stat_out);
}

// fq_stem_conv7x7s2_pool: the 7x7 first convolution (BatchNorm / activation folded) followed by MaxPool2D(3, 2, 1) - the two
// host entry points above, one after the other (reference: gluoncv resnet*_v1 features[0..3] = Convolution, BatchNorm,
// Activation, Pooling operators in sequence)
int fq_stem_conv7x7s2_pool_host(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n,
                                int64_t cin, int64_t cout, int64_t h, int64_t w, const float* bn_scale,
                                const float* bn_shift, int act, float* stat_out, fqStream_t stream) {
  REQUIRE(x && w_tap_major && y && n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "fq_stem_conv7x7s2_pool_host: bad arguments");
  const int64_t ho = (h + 6 - 7) / 2 + 1, wo = (w + 6 - 7) / 2 + 1;
  std::vector<float> conv((size_t)(n * cout * ho * wo)), one((size_t)cout, 1.0f), zero((size_t)cout, 0.0f);
  if (int rc = stem_conv_impl(x, w_tap_major, bias, conv.data(), n, cin, cout, h, w, 7, bn_scale, bn_shift,
                              (act & ~FQ_STAT_PREZEROED) | FQ_STAT_PREZEROED, nullptr))
    return rc;
  return fq_bn_act_maxpool_stat_host(conv.data(), y, n, cout, ho, wo, one.data(), zero.data(),
                                     FQ_ACT_NONE | (act & FQ_STAT_PREZEROED), stat_out, stream);
}

int fq_dwconv3x3_host(const float* x, const float* w, const float* bias, float* y, int64_t n, int64_t c, int64_t h,
                      int64_t wdt, int stride, const float* in_stat, const float* in_thr, int in_width,
                      unsigned in_flags, float* out_current_max, const float* bn_scale, const float* bn_shift, int act,
                      float* stat_out, fqStream_t) {
  REQUIRE(x && w && y && n > 0 && c > 0 && h > 0 && wdt > 0, "fq_dwconv3x3_host: bad arguments");
  REQUIRE(stride == 1 || stride == 2, "fq_dwconv3x3_host: stride must be 1 or 2");
  // in_stat alone: online; in_thr alone: offline; both: offline, the statistic only feeds out_current_max
  REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_dwconv3x3_host: bn_scale and bn_shift go together");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  zero_stat(stat_out, n, prezeroed);
  const bool quant = in_stat || in_thr;
  QP q = {0, 0, 1, 1};
  if (quant) {
    const float max_ = in_thr ? in_thr[0] : batch_mean(in_stat, n);
    if (in_stat && out_current_max) out_current_max[0] = in_thr ? batch_mean(in_stat, n) : max_;
    q = make_qp(max_, act_levels(in_width, in_flags), (in_flags & FQ_ACT_LO_NEG_MAX) != 0,
                (in_flags & FQ_ACT_NO_EPS) ? 0.0f : kEps);
  }
  const int64_t ho = (h - 1) / stride + 1, wo = (wdt - 1) / stride + 1;
#pragma omp parallel for schedule(static)
  for (int64_t pl = 0; pl < n * c; ++pl) {
    const int64_t ch = pl % c;
    const float* xp = x + pl * h * wdt;
    const float* wp = w + ch * 9;
    for (int64_t oy = 0; oy < ho; ++oy)
      for (int64_t ox = 0; ox < wo; ++ox) {
        float acc = 0.0f;
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) {
            const int64_t iy = oy * stride - 1 + ky, ix = ox * stride - 1 + kx;
            float v = 0.0f;
            if (iy >= 0 && iy < h && ix >= 0 && ix < wdt) {
              v = xp[iy * wdt + ix];
              if (quant) v = code_of(v, q) * q.scale;
            }
            acc = __builtin_fmaf(wp[ky * 3 + kx], v, acc);
          }
        if (bias) acc = acc + bias[ch];
        if (bn_scale) {
          acc = acc * bn_scale[ch];
          acc = acc + bn_shift[ch];
        }
        y[(pl * ho + oy) * wo + ox] = act_of(acc, act);
      }
  }
  stat_of_output(y, n, c * ho * wo, stat_out);
  return FQ_OK;
}

// integer codes of the weight fake-quant (convert_conv2d.py:70-95): code = round_half_away(w / (s + 1e-10))
int fq_weight_codes_host(const float* w, int64_t rows, int64_t row_len, int rows_per_scale, int width, int64_t row_pad,
                         int64_t rows_pad, int8_t* codes, float* scales, int32_t* rowsum, void*, fqStream_t) {
  REQUIRE(w && codes && scales && rowsum, "fq_weight_codes_host: null pointer");
  REQUIRE(rows > 0 && row_len > 0 && rows_per_scale > 0 && rows % rows_per_scale == 0 && width >= 2 && width <= 8 &&
          row_pad >= row_len && rows_pad >= rows, "fq_weight_codes_host: bad arguments");
  const float levels = (float)((1 << (width - 1)) - 1);
  const int64_t groups = rows / rows_per_scale;
  std::vector<float> gmax((size_t)groups);
  per_sample_stat(w, groups, (int64_t)rows_per_scale * row_len, true, gmax.data());
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows_pad; ++r) {
    int8_t* dst = codes + r * row_pad;
    if (r >= rows) {
      memset(dst, 0, (size_t)row_pad);
      continue;
    }
    const float s = gmax[(size_t)(r / rows_per_scale)] / levels;
    const float d = s + kEps;
    int acc = 0;
    for (int64_t i = 0; i < row_pad; ++i) {
      int cde = 0;
      if (i < row_len) cde = (int)round_half_away(w[r * row_len + i] / d);
      dst[i] = (int8_t)cde;
      acc += cde;
    }
    rowsum[r] = acc;
    scales[r] = s;
  }
  return FQ_OK;
}

static int pwconv_i8_impl(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                          const float* bias, float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw,
                          const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                          float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                          const float* residual, const int32_t* xcodes = nullptr) {
  REQUIRE((x || xcodes) && wcodes && wscale && wsum && y, "fq_pwconv_i8_host: null pointer");
  REQUIRE(n > 0 && cin > 0 && cout > 0 && hw > 0 && cin_pad >= cin, "fq_pwconv_i8_host: bad shape");
  REQUIRE(in_stat != nullptr || in_thr != nullptr, "fq_pwconv_i8_host: give in_stat, in_thr or both");
  REQUIRE(in_width >= 2 && in_width <= 8, "fq_pwconv_i8_host: input width does not fit int8 codes");
  REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_pwconv_i8_host: bn_scale and bn_shift go together");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  zero_stat(stat_out, n, prezeroed);
  const float max_ = in_thr ? in_thr[0] : batch_mean(in_stat, n);
  if (in_stat && out_current_max) out_current_max[0] = in_thr ? batch_mean(in_stat, n) : max_;
  const QP q = make_qp(max_, act_levels(in_width, in_flags), (in_flags & FQ_ACT_LO_NEG_MAX) != 0, kEps);
  const float sx = q.scale;
  (void)wsum;   // the device kernel stores unsigned codes re-centred by 128 and adds 128 * wsum back: same integer sum
#pragma omp parallel
  {
    std::vector<int32_t> cx((size_t)cin * 64);
#pragma omp for collapse(2) schedule(static)
    for (int64_t s = 0; s < n; ++s)
      for (int64_t p0 = 0; p0 < hw; p0 += 64) {
        const int64_t pc = std::min<int64_t>(64, hw - p0);
        for (int64_t ci = 0; ci < cin; ++ci)
          for (int64_t p = 0; p < pc; ++p)
            cx[(size_t)(ci * 64 + p)] = xcodes ? xcodes[(s * cin + ci) * hw + p0 + p]
                                               : (int32_t)code_of(x[(s * cin + ci) * hw + p0 + p], q);
        for (int64_t co = 0; co < cout; ++co) {
          const int8_t* wr = wcodes + co * cin_pad;
          int32_t acc[64];
          for (int64_t p = 0; p < pc; ++p) acc[p] = 0;
          for (int64_t ci = 0; ci < cin; ++ci) {
            const int32_t wv = wr[ci];
            const int32_t* cr = &cx[(size_t)(ci * 64)];
            for (int64_t p = 0; p < pc; ++p) acc[p] += wv * cr[p];
          }
          const float sxw = sx * wscale[co];
          for (int64_t p = 0; p < pc; ++p) {
            float v = (float)acc[p] * sxw;
            if (bias) v = v + bias[co];
            if (bn_scale) {
              v = v * bn_scale[co];
              v = v + bn_shift[co];
            }
            if (residual) v = v + residual[(s * cout + co) * hw + p0 + p];
            y[(s * cout + co) * hw + p0 + p] = act_of(v, act);
          }
        }
      }
  }
  stat_of_output(y, n, cout * hw, stat_out);
  return FQ_OK;
}

int fq_pwconv_i8_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                      const float* bias, float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw,
                      const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                      float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                      void*, fqStream_t) {
  return pwconv_i8_impl(x, wcodes, wscale, wsum, bias, y, n, cin, cin_pad, cout, hw, in_stat, in_thr, in_width, in_flags,
                        out_current_max, bn_scale, bn_shift, act, stat_out, nullptr);
}

// The recompute pair (include/fakequant.h at fq_pwconv_i8_stat / fq_pwdw_fused): on the host the tensor between the two
// layers simply exists - the twins are the two storing twins back to back, which is the definition the device kernels are
// held to (the reference's chain: Conv2D 1x1 -> BatchNorm -> activation -> the depthwise block's activation branch
// convert_conv2d.py:53-66 -> F.Convolution :108 -> BatchNorm -> activation).
int fq_pwconv_i8_stat_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                           int64_t n, int64_t cin, int64_t cin_pad, int64_t, int64_t cout, int64_t hw, const float* in_stat,
                           const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                           const float* bn_scale, const float* bn_shift, int act, float* stat_out, fqStream_t) {
  REQUIRE(stat_out && n > 0 && cout > 0 && hw > 0, "fq_pwconv_i8_stat_host: bad arguments");
  std::vector<float> y((size_t)n * cout * hw);
  return pwconv_i8_impl(x, wcodes, wscale, wsum, bias, y.data(), n, cin, cin_pad, cout, hw, in_stat, in_thr, in_width, in_flags,
                        out_current_max, bn_scale, bn_shift, act & ~FQ_STAT_PREZEROED, stat_out, nullptr);
}

int fq_pwdw_fused_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* pw_bias,
                       int64_t n, int64_t cin, int64_t cin_pad, int64_t, int64_t cout, int64_t h, int64_t w,
                       const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, const float* pw_bn_scale,
                       const float* pw_bn_shift, int pw_act, const float* mid_stat, const float* mid_thr, int mid_width,
                       unsigned mid_flags, float* mid_current_max, const float* dw_w, const float* dw_bias, int dw_stride,
                       const float* dw_bn_scale, const float* dw_bn_shift, int dw_act, float* y, float* stat_out, fqStream_t) {
  REQUIRE(x && y && dw_w && n > 0 && cout > 0 && h > 0 && w > 0, "fq_pwdw_fused_host: bad arguments");
  REQUIRE(mid_stat || mid_thr, "fq_pwdw_fused_host: give mid_stat or mid_thr");
  std::vector<float> mid((size_t)n * cout * h * w);
  float cur1 = 0.0f;
  if (int rc = pwconv_i8_impl(x, wcodes, wscale, wsum, pw_bias, mid.data(), n, cin, cin_pad, cout, h * w, in_stat, in_thr,
                              in_width, in_flags, &cur1, pw_bn_scale, pw_bn_shift, pw_act, nullptr, nullptr))
    return rc;
  float cur2 = 0.0f;
  return fq_dwconv3x3_host(mid.data(), dw_w, dw_bias, y, n, cout, h, w, dw_stride, mid_stat, mid_thr, mid_width, mid_flags,
                           mid_current_max ? mid_current_max : &cur2, dw_bn_scale, dw_bn_shift, dw_act & ~FQ_STAT_PREZEROED,
                           stat_out, nullptr);
}

// The classifier on the codes + the evaluation counters of its logits (include/fakequant.h at fq_dense_i8_eval):
// fq_pwconv_i8 with planes of one pixel, then fq_eval_counters on what it wrote.
int fq_dense_i8_eval_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                          const float* bias, float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout,
                          const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                          float* out_current_max, const int64_t* labels, float* counters, void*, void*, fqStream_t) {
  REQUIRE(labels && counters, "fq_dense_i8_eval_host: null pointer");
  if (int rc = pwconv_i8_impl(x, wcodes, wscale, wsum, bias, y, n, cin, cin_pad, cout, 1, in_stat, in_thr, in_width,
                              in_flags, out_current_max, nullptr, nullptr, FQ_ACT_NONE, nullptr, nullptr))
    return rc;
  return fq_eval_counters_host(y, labels, n, cout, counters, nullptr);
}

// Strided 1x1 convolution: the stride-1 arithmetic on the subsampled input x[:, :, ::s, ::s]; optional residual operand of
// y's shape, added after BatchNorm and before the activation.
int fq_pwconv_i8_strided_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                              const float* bias, float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout,
                              int64_t h, int64_t w, int stride, const float* in_stat, const float* in_thr, int in_width,
                              unsigned in_flags, float* out_current_max, const float* bn_scale, const float* bn_shift,
                              int act, float* stat_out, const float* residual, void*, fqStream_t) {
  REQUIRE(x && h > 0 && w > 0 && (stride == 1 || stride == 2), "fq_pwconv_i8_strided_host: bad arguments");
  const int64_t ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
  if (stride == 1)
    return pwconv_i8_impl(x, wcodes, wscale, wsum, bias, y, n, cin, cin_pad, cout, h * w, in_stat, in_thr, in_width,
                          in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, residual);
  std::vector<float> sub((size_t)(n * cin * ho * wo));
#pragma omp parallel for schedule(static)
  for (int64_t pc = 0; pc < n * cin; ++pc)
    for (int64_t r = 0; r < ho; ++r)
      for (int64_t c = 0; c < wo; ++c) sub[(size_t)((pc * ho + r) * wo + c)] = x[(pc * h + r * stride) * w + c * stride];
  return pwconv_i8_impl(sub.data(), wcodes, wscale, wsum, bias, y, n, cin, cin_pad, cout, ho * wo, in_stat, in_thr,
                        in_width, in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, residual);
}

// fq_pwconv_i8_shortcut: the two storing twins back to back (the shortcut tensor exists on the host).
int fq_pwconv_i8_shortcut_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                               float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw, const float* in_stat,
                               const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                               const float* bn_scale, const float* bn_shift, int act, float* stat_out, const float* x2,
                               const int8_t* wcodes2, const float* wscale2, const int32_t* wsum2, int64_t cin2, int64_t cin2_pad,
                               const float* in_stat2, const float* in_thr2, int in_width2, unsigned in_flags2,
                               float* out_current_max2, const float* bn_scale2, const float* bn_shift2, fqStream_t) {
  REQUIRE(x && x2 && y && hw > 0, "fq_pwconv_i8_shortcut_host: bad arguments");
  std::vector<float> sc((size_t)(n * cout * hw));
  if (int rc = pwconv_i8_impl(x2, wcodes2, wscale2, wsum2, nullptr, sc.data(), n, cin2, cin2_pad, cout, hw, in_stat2, in_thr2,
                              in_width2, in_flags2, out_current_max2, bn_scale2, bn_shift2, FQ_ACT_NONE, nullptr, nullptr))
    return rc;
  return pwconv_i8_impl(x, wcodes, wscale, wsum, bias, y, n, cin, cin_pad, cout, hw, in_stat, in_thr, in_width, in_flags,
                        out_current_max, bn_scale, bn_shift, act, stat_out, sc.data());
}

// fq_pwconv_i8_gap: the two storing twins back to back (the convolution's output exists on the host).
int fq_pwconv_i8_gap_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                          float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw, const float* in_stat,
                          const float* in_thr, int in_width, unsigned in_flags, float* out_current_max, const float* bn_scale,
                          const float* bn_shift, int act, float* stat_out, const float* residual, void*, fqStream_t st) {
  REQUIRE(x && y && hw > 0, "fq_pwconv_i8_gap_host: bad arguments");
  std::vector<float> full((size_t)(n * cout * hw));
  const int prez = act & FQ_STAT_PREZEROED;
  if (int rc = pwconv_i8_impl(x, wcodes, wscale, wsum, bias, full.data(), n, cin, cin_pad, cout, hw, in_stat, in_thr, in_width,
                              in_flags, out_current_max, bn_scale, bn_shift, act & ~FQ_STAT_PREZEROED, nullptr, residual))
    return rc;
  return fq_global_avg_pool_stat_host(full.data(), y, n, cout, hw, prez, stat_out, st);
}

// fq_pwconv_i8_sub2: the whole stride-1 output on the host, of which the even pixels of the even rows are handed back.
int fq_pwconv_i8_sub2_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                           float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                           const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                           float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                           const float* residual, void*, fqStream_t) {
  REQUIRE(x && y && h > 0 && w > 0, "fq_pwconv_i8_sub2_host: bad arguments");
  std::vector<float> full((size_t)(n * cout * h * w));
  if (int rc = pwconv_i8_impl(x, wcodes, wscale, wsum, bias, full.data(), n, cin, cin_pad, cout, h * w, in_stat, in_thr,
                              in_width, in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, residual))
    return rc;
  const int64_t hs = (h + 1) / 2, ws_ = (w + 1) / 2;
  for (int64_t pc = 0; pc < n * cout; ++pc)
    for (int64_t r = 0; r < hs; ++r)
      for (int64_t c = 0; c < ws_; ++c) y[(pc * hs + r) * ws_ + c] = full[(size_t)((pc * h + 2 * r) * w + 2 * c)];
  return 0;
}

// Dense 3x3 convolution (stride 1, pad 1) on the integer codes: exact integer sums over (ky, kx, ci), zero padding = code 0.
// wcodes rows are ordered (tap, ci) - the weights were permuted to (cout, 3, 3, cin) before fq_weight_codes_host.
static int conv3x3_i8_impl(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                           const float* bias, float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w,
                           const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                           float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                           const int32_t* xcodes) {
  REQUIRE((x || xcodes) && wcodes && wscale && wsum && y, "fq_conv3x3_i8_host: null pointer");
  REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "fq_conv3x3_i8_host: bad shape");
  REQUIRE(in_stat != nullptr || in_thr != nullptr, "fq_conv3x3_i8_host: give in_stat, in_thr or both");
  REQUIRE(in_width >= 2 && in_width <= 8, "fq_conv3x3_i8_host: input width does not fit int8 codes");
  REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_conv3x3_i8_host: bn_scale and bn_shift go together");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  zero_stat(stat_out, n, prezeroed);
  const float max_ = in_thr ? in_thr[0] : batch_mean(in_stat, n);
  if (in_stat && out_current_max) out_current_max[0] = in_thr ? batch_mean(in_stat, n) : max_;
  const QP q = make_qp(max_, act_levels(in_width, in_flags), (in_flags & FQ_ACT_LO_NEG_MAX) != 0, kEps);
  const float sx = q.scale;
  (void)wsum;
  const int64_t hw = h * w, k9 = 9 * cin;
#pragma omp parallel
  {
    std::vector<int32_t> cx((size_t)(cin * (h + 2) * (w + 2)));        // zero-padded codes of one sample
    std::vector<int32_t> acc((size_t)hw);
#pragma omp for schedule(static)
    for (int64_t s = 0; s < n; ++s) {
      std::fill(cx.begin(), cx.end(), 0);
      for (int64_t ci = 0; ci < cin; ++ci)
        for (int64_t r = 0; r < h; ++r)
          for (int64_t c = 0; c < w; ++c)
            cx[(size_t)((ci * (h + 2) + r + 1) * (w + 2) + c + 1)] =
                xcodes ? xcodes[((s * cin + ci) * h + r) * w + c] : (int32_t)code_of(x[((s * cin + ci) * h + r) * w + c], q);
      for (int64_t co = 0; co < cout; ++co) {
        std::fill(acc.begin(), acc.end(), 0);
        const int8_t* wr = wcodes + co * k9;
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap % 3;
          for (int64_t ci = 0; ci < cin; ++ci) {
            const int32_t wv = wr[tap * cin + ci];
            if (wv == 0) continue;
            const int32_t* src = &cx[(size_t)((ci * (h + 2) + ky) * (w + 2) + kx)];
            for (int64_t r = 0; r < h; ++r)
              for (int64_t c = 0; c < w; ++c) acc[(size_t)(r * w + c)] += wv * src[r * (w + 2) + c];
          }
        }
        const float sxw = sx * wscale[co];
        for (int64_t p = 0; p < hw; ++p) {
          float v = (float)acc[(size_t)p] * sxw;
          if (bias) v = v + bias[co];
          if (bn_scale) {
            v = v * bn_scale[co];
            v = v + bn_shift[co];
          }
          y[(s * cout + co) * hw + p] = act_of(v, act);
        }
      }
    }
  }
  stat_of_output(y, n, cout * hw, stat_out);
  return FQ_OK;
}

int fq_conv3x3_i8_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                       const float* bias, float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w,
                       const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                       float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                       fqStream_t) {
  return conv3x3_i8_impl(x, wcodes, wscale, wsum, bias, y, n, cin, cout, h, w, in_stat, in_thr, in_width, in_flags,
                         out_current_max, bn_scale, bn_shift, act, stat_out, nullptr);
}

// ---- C16 code tensors (include/fakequant.h at fq_pwconv_i8_c16): [n][ceil(C/16)][pixels][16], byte = (code + 128 - zoff) ^ 0x80
static void c16_decode(const int8_t* t, int64_t n, int64_t c, int64_t hw, int zoff, int32_t* codes) {
  const int64_t cb = (c + 15) / 16;
#pragma omp parallel for schedule(static)
  for (int64_t s = 0; s < n; ++s)
    for (int64_t ch = 0; ch < c; ++ch)
      for (int64_t p = 0; p < hw; ++p) {
        const int b = ((uint8_t)t[((s * cb + ch / 16) * hw + p) * 16 + ch % 16]) ^ 0x80;     // code + 128 - zoff (mod 256)
        codes[(s * c + ch) * hw + p] = zoff == 128 ? b : (int)(int8_t)(uint8_t)((b - 128) & 255);
      }
}

static void c16_encode(const float* y, int64_t n, int64_t c, int64_t hw, const float* thr, int width, unsigned flags,
                       int8_t* t) {
  const QP q2 = make_qp(thr[0], act_levels(width, flags), (flags & FQ_ACT_LO_NEG_MAX) != 0, kEps);
  const int zoff = (flags & FQ_ACT_SIGNED) ? 0 : 128;
  const int64_t cb = (c + 15) / 16;
#pragma omp parallel for schedule(static)
  for (int64_t s = 0; s < n; ++s)
    for (int64_t ch = 0; ch < cb * 16; ++ch)
      for (int64_t p = 0; p < hw; ++p) {
        const int code = ch < c ? (int)code_of(y[(s * c + ch) * hw + p], q2) : 0;
        t[((s * cb + ch / 16) * hw + p) * 16 + ch % 16] = (int8_t)(uint8_t)(((code + 128 - zoff) & 255) ^ 0x80);
      }
}

int fq_pwconv_i8_c16_host(const void* x, int x_is_c16, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                          const float* bias, void* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h,
                          int64_t w, int stride, const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                          float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                          const float* residual, const float* out_thr, int out_width, unsigned out_flags, void*, fqStream_t) {
  REQUIRE(x && y && h > 0 && w > 0 && (stride == 1 || stride == 2), "fq_pwconv_i8_c16_host: bad arguments");
  REQUIRE(x_is_c16 || out_thr, "fq_pwconv_i8_c16_host: neither side is a C16 tensor");
  REQUIRE(!x_is_c16 || in_thr, "fq_pwconv_i8_c16_host: a C16 input needs in_thr");
  const int64_t ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
  std::vector<int32_t> codes, sub_c;
  std::vector<float> sub_x, tmp;
  const float* xf = (const float*)x;
  const int32_t* xc = nullptr;
  if (x_is_c16) {
    codes.resize((size_t)(n * cin * h * w));
    c16_decode((const int8_t*)x, n, cin, h * w, (in_flags & FQ_ACT_SIGNED) ? 0 : 128, codes.data());
    xc = codes.data();
    xf = nullptr;
  }
  if (stride != 1) {
    if (xc) {
      sub_c.resize((size_t)(n * cin * ho * wo));
      for (int64_t pc = 0; pc < n * cin; ++pc)
        for (int64_t r = 0; r < ho; ++r)
          for (int64_t c = 0; c < wo; ++c) sub_c[(size_t)((pc * ho + r) * wo + c)] = xc[(pc * h + r * stride) * w + c * stride];
      xc = sub_c.data();
    } else {
      sub_x.resize((size_t)(n * cin * ho * wo));
      for (int64_t pc = 0; pc < n * cin; ++pc)
        for (int64_t r = 0; r < ho; ++r)
          for (int64_t c = 0; c < wo; ++c) sub_x[(size_t)((pc * ho + r) * wo + c)] = xf[(pc * h + r * stride) * w + c * stride];
      xf = sub_x.data();
    }
  }
  float* yf = (float*)y;
  if (out_thr) {
    REQUIRE(residual == nullptr, "fq_pwconv_i8_c16_host: a residual operand goes with fp32 output");
    tmp.resize((size_t)(n * cout * ho * wo));
    yf = tmp.data();
  }
  if (int rc = pwconv_i8_impl(xf, wcodes, wscale, wsum, bias, yf, n, cin, cin_pad, cout, ho * wo, in_stat, in_thr, in_width,
                              in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, residual, xc))
    return rc;
  if (out_thr) c16_encode(yf, n, cout, ho * wo, out_thr, out_width, out_flags, (int8_t*)y);
  return FQ_OK;
}

// the 3x3 first convolution handing its consumer's codes over: the fp32 layer, then the codes of its output
int fq_stem_conv3x3s2_c16_host(const float* x, const float* w_tap_major, const float* bias, void* y16, int64_t n, int64_t cin,
                               int64_t cout, int64_t h, int64_t w, const float* bn_scale, const float* bn_shift, int act,
                               float* stat_out, const float* out_thr, int out_width, unsigned out_flags, fqStream_t stream) {
  REQUIRE(y16 && out_thr, "fq_stem_conv3x3s2_c16_host: null pointer");
  const int64_t ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  std::vector<float> tmp((size_t)(n * cout * ho * wo));
  if (int rc = fq_stem_conv3x3s2_host(x, w_tap_major, bias, tmp.data(), n, cin, cout, h, w, bn_scale, bn_shift, act, stat_out,
                                      stream))
    return rc;
  c16_encode(tmp.data(), n, cout, ho * wo, out_thr, out_width, out_flags, (int8_t*)y16);
  return FQ_OK;
}

// the closing 1x1 of a residual unit with two outputs: the fp32 result, and its codes under the next consumer's threshold
int fq_pwconv_i8_c16_dual_host(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                               float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                               const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                               float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                               const float* residual, const float* out_thr, int out_width, unsigned out_flags, void* ws,
                               fqStream_t stream) {
  REQUIRE(y16 && out_thr && residual && in_thr, "fq_pwconv_i8_c16_dual_host: null pointer");
  if (int rc = fq_pwconv_i8_c16_host(x, 1, wcodes, wscale, wsum, bias, y, n, cin, cin_pad, cout, h, w, 1, in_stat, in_thr,
                                     in_width, in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, residual, nullptr,
                                     8, 0, ws, stream))
    return rc;
  c16_encode(y, n, cout, h * w, out_thr, out_width, out_flags, (int8_t*)y16);
  return FQ_OK;
}

// fq_pwconv_i8_shortcut_c16: the shortcut convolution's twin (input fp32 or codes), then the dual form's with it as residual
int fq_pwconv_i8_shortcut_c16_host(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                                   float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw,
                                   const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                                   float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                                   const void* x2, int x2_is_c16, const int8_t* wcodes2, const float* wscale2, const int32_t* wsum2,
                                   int64_t cin2, int64_t cin2_pad, const float* in_stat2, const float* in_thr2, int in_width2,
                                   unsigned in_flags2, float* out_current_max2, const float* bn_scale2, const float* bn_shift2,
                                   const float* out_thr, int out_width, unsigned out_flags, fqStream_t st) {
  REQUIRE(x && x2 && y && y16 && hw > 0, "fq_pwconv_i8_shortcut_c16_host: bad arguments");
  std::vector<float> sc((size_t)(n * cout * hw));
  if (x2_is_c16) {
    if (int rc = fq_pwconv_i8_c16_host(x2, 1, wcodes2, wscale2, wsum2, nullptr, sc.data(), n, cin2, cin2_pad, cout, hw, 1, 1,
                                       in_stat2, in_thr2, in_width2, in_flags2, out_current_max2, bn_scale2, bn_shift2,
                                       FQ_ACT_NONE, nullptr, nullptr, nullptr, 8, 0, nullptr, st))
      return rc;
  } else if (int rc = pwconv_i8_impl((const float*)x2, wcodes2, wscale2, wsum2, nullptr, sc.data(), n, cin2, cin2_pad, cout, hw,
                                     in_stat2, in_thr2, in_width2, in_flags2, out_current_max2, bn_scale2, bn_shift2,
                                     FQ_ACT_NONE, nullptr, nullptr)) {
    return rc;
  }
  return fq_pwconv_i8_c16_dual_host(x, wcodes, wscale, wsum, bias, y, y16, n, cin, cin_pad, cout, hw, 1, in_stat, in_thr, in_width,
                                    in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, sc.data(), out_thr, out_width,
                                    out_flags, nullptr, st);
}

// ... both outputs subsampled: the whole fp32 output on the host, its even pixels of the even rows, and their codes
int fq_pwconv_i8_c16_dual_sub2_host(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                                    const float* bias, float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout,
                                    int64_t h, int64_t w, const float* in_stat, const float* in_thr, int in_width,
                                    unsigned in_flags, float* out_current_max, const float* bn_scale, const float* bn_shift,
                                    int act, float* stat_out, const float* residual, const float* out_thr, int out_width,
                                    unsigned out_flags, void* ws, fqStream_t stream) {
  REQUIRE(y && y16 && out_thr && residual && in_thr, "fq_pwconv_i8_c16_dual_sub2_host: null pointer");
  std::vector<float> full((size_t)(n * cout * h * w));
  if (int rc = fq_pwconv_i8_c16_host(x, 1, wcodes, wscale, wsum, bias, full.data(), n, cin, cin_pad, cout, h, w, 1, in_stat,
                                     in_thr, in_width, in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, residual,
                                     nullptr, 8, 0, ws, stream))
    return rc;
  const int64_t hs = (h + 1) / 2, ws_ = (w + 1) / 2;
  for (int64_t pc = 0; pc < n * cout; ++pc)
    for (int64_t r = 0; r < hs; ++r)
      for (int64_t c = 0; c < ws_; ++c) y[(pc * hs + r) * ws_ + c] = full[(size_t)((pc * h + 2 * r) * w + 2 * c)];
  c16_encode(y, n, cout, hs * ws_, out_thr, out_width, out_flags, (int8_t*)y16);
  return FQ_OK;
}

int fq_conv3x3_i8_c16_host(const void* x, int x_is_c16, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                           const float* bias, void* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w,
                           const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                           float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                           const float* out_thr, int out_width, unsigned out_flags, fqStream_t) {
  REQUIRE(x && y, "fq_conv3x3_i8_c16_host: null pointer");
  REQUIRE(x_is_c16 || out_thr, "fq_conv3x3_i8_c16_host: neither side is a C16 tensor");
  REQUIRE(!x_is_c16 || in_thr, "fq_conv3x3_i8_c16_host: a C16 input needs in_thr");
  std::vector<int32_t> codes;
  std::vector<float> tmp;
  const int32_t* xc = nullptr;
  if (x_is_c16) {
    codes.resize((size_t)(n * cin * h * w));
    c16_decode((const int8_t*)x, n, cin, h * w, (in_flags & FQ_ACT_SIGNED) ? 0 : 128, codes.data());
    xc = codes.data();
  }
  float* yf = (float*)y;
  if (out_thr) {
    tmp.resize((size_t)(n * cout * h * w));
    yf = tmp.data();
  }
  if (int rc = conv3x3_i8_impl(x_is_c16 ? nullptr : (const float*)x, wcodes, wscale, wsum, bias, yf, n, cin, cout, h, w,
                               in_stat, in_thr, in_width, in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, xc))
    return rc;
  if (out_thr) c16_encode(yf, n, cout, h * w, out_thr, out_width, out_flags, (int8_t*)y);
  return FQ_OK;
}

// Depthwise 3x3 between two C16 code tensors: x^ = code * sx (what the fake-quant of the fp32 tensor gives), the arithmetic
// of fq_dwconv3x3_host on it, then the consumer's codes.
int fq_dwconv3x3_c16_host(const void* x, const float* w, const float* bias, void* y, int64_t n, int64_t c, int64_t h,
                          int64_t wdt, int stride, const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                          float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                          const float* out_thr, int out_width, unsigned out_flags, fqStream_t) {
  REQUIRE(x && w && y && in_thr && out_thr && n > 0 && c > 0 && h > 0 && wdt > 0, "fq_dwconv3x3_c16_host: bad arguments");
  REQUIRE(stride == 1 || stride == 2, "fq_dwconv3x3_c16_host: stride must be 1 or 2");
  const QP q = make_qp(in_thr[0], act_levels(in_width, in_flags), (in_flags & FQ_ACT_LO_NEG_MAX) != 0, kEps);
  if (in_stat && out_current_max) out_current_max[0] = batch_mean(in_stat, n);
  std::vector<int32_t> codes((size_t)(n * c * h * wdt));
  c16_decode((const int8_t*)x, n, c, h * wdt, (in_flags & FQ_ACT_SIGNED) ? 0 : 128, codes.data());
  std::vector<float> xq(codes.size());
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)codes.size(); ++i) xq[(size_t)i] = (float)codes[(size_t)i] * q.scale;
  const int64_t ho = (h - 1) / stride + 1, wo = (wdt - 1) / stride + 1;
  std::vector<float> yf((size_t)(n * c * ho * wo));
  if (int rc = fq_dwconv3x3_host(xq.data(), w, bias, yf.data(), n, c, h, wdt, stride, nullptr, nullptr, in_width, in_flags,
                                 nullptr, bn_scale, bn_shift, act, stat_out, nullptr))
    return rc;
  c16_encode(yf.data(), n, c, ho * wo, out_thr, out_width, out_flags, (int8_t*)y);
  return FQ_OK;
}

// fq_weight_slices: per row p = 2^e (smallest power of two with max|w| <= p * 2^20), m = rint(w / p), balanced base-128
// digits; codes: 3 buffers of 2 * rows_pad * row_pad bytes - the row-major digits in the first half (the fragment-major
// second half is a device layout and stays zero here).
int fq_weight_slices_host(const float* w, int64_t rows, int64_t row_len, int64_t row_pad, int64_t rows_pad, int8_t* codes,
                          float* pscale, int32_t* rowsum, void*, fqStream_t) {
  REQUIRE(w && codes && pscale && rowsum && rows > 0 && row_len > 0 && row_pad >= row_len && rows_pad >= rows,
          "fq_weight_slices_host: bad arguments");
  const int64_t slice = 2 * rows_pad * row_pad;
  memset(codes, 0, (size_t)(3 * slice));
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r) {
    float mx = 0.0f;
    for (int64_t i = 0; i < row_len; ++i) mx = std::max(mx, std::fabs(w[r * row_len + i]));
    int e = 0;
    if (mx > 0.0f) {
      (void)std::frexp(mx, &e);
      if (std::ldexp(1.0f, e - 1) == mx) e -= 1;
    }
    const float p = mx > 0.0f ? std::ldexp(1.0f, e - 20) : 1.0f;
    int32_t a[3] = {0, 0, 0};
    for (int64_t i = 0; i < row_len; ++i) {
      const int m = (int)std::rint(w[r * row_len + i] / p);
      const int d3 = ((m + 64) & 127) - 64;
      const int m1 = (m - d3) >> 7;
      const int d2 = ((m1 + 64) & 127) - 64;
      const int d1 = (m1 - d2) >> 7;
      codes[0 * slice + r * row_pad + i] = (int8_t)d1;
      codes[1 * slice + r * row_pad + i] = (int8_t)d2;
      codes[2 * slice + r * row_pad + i] = (int8_t)d3;
      a[0] += d1; a[1] += d2; a[2] += d3;
    }
    for (int sl = 0; sl < 3; ++sl) rowsum[sl * rows + r] = a[sl];
    pscale[r] = p;
  }
  return FQ_OK;
}

// The sliced dense 3x3 convolution: T = sum (d1 2^14 + d2 2^7 + d3) * cx exactly (64 bits), y = fp32(fp64(T) * fp64(sx * p)).
int fq_conv3x3_i8_sliced_host(const float* x, const int8_t* wslices, const float* pscale, const int32_t* wsum,
                              const float* bias, float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w,
                              const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                              float* out_current_max, const float* bn_scale, const float* bn_shift, int act,
                              float* stat_out, fqStream_t) {
  REQUIRE(x && wslices && pscale && wsum && y, "fq_conv3x3_i8_sliced_host: null pointer");
  REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "fq_conv3x3_i8_sliced_host: bad shape");
  REQUIRE(in_stat != nullptr || in_thr != nullptr, "fq_conv3x3_i8_sliced_host: give in_stat, in_thr or both");
  REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_conv3x3_i8_sliced_host: bn_scale and bn_shift go together");
  const bool prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  zero_stat(stat_out, n, prezeroed);
  const float max_ = in_thr ? in_thr[0] : batch_mean(in_stat, n);
  if (in_stat && out_current_max) out_current_max[0] = in_thr ? batch_mean(in_stat, n) : max_;
  const QP q = make_qp(max_, act_levels(in_width, in_flags), (in_flags & FQ_ACT_LO_NEG_MAX) != 0, kEps);
  const float sx = q.scale;
  const int64_t hw = h * w, k9 = 9 * cin;
  const int64_t rows_pad = (cout + 63) / 64 * 64, slice = 2 * rows_pad * k9;
#pragma omp parallel
  {
    std::vector<int32_t> cx((size_t)(cin * (h + 2) * (w + 2)));
    std::vector<int64_t> acc((size_t)hw);
#pragma omp for schedule(static)
    for (int64_t s = 0; s < n; ++s) {
      std::fill(cx.begin(), cx.end(), 0);
      for (int64_t ci = 0; ci < cin; ++ci)
        for (int64_t r = 0; r < h; ++r)
          for (int64_t c = 0; c < w; ++c)
            cx[(size_t)((ci * (h + 2) + r + 1) * (w + 2) + c + 1)] = (int32_t)code_of(x[((s * cin + ci) * h + r) * w + c], q);
      for (int64_t co = 0; co < cout; ++co) {
        std::fill(acc.begin(), acc.end(), 0);
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap % 3;
          for (int64_t ci = 0; ci < cin; ++ci) {
            const int64_t at = co * k9 + tap * cin + ci;
            const int64_t m = ((int64_t)wslices[at] << 14) + ((int64_t)wslices[slice + at] << 7) + (int64_t)wslices[2 * slice + at];
            if (m == 0) continue;
            const int32_t* src = &cx[(size_t)((ci * (h + 2) + ky) * (w + 2) + kx)];
            for (int64_t r = 0; r < h; ++r)
              for (int64_t c = 0; c < w; ++c) acc[(size_t)(r * w + c)] += m * src[r * (w + 2) + c];
          }
        }
        const float sxp = sx * pscale[co];
        for (int64_t p = 0; p < hw; ++p) {
          float v = (float)((double)acc[(size_t)p] * (double)sxp);
          if (bias) v = v + bias[co];
          if (bn_scale) {
            v = v * bn_scale[co];
            v = v + bn_shift[co];
          }
          y[(s * cout + co) * hw + p] = act_of(v, act);
        }
      }
    }
  }
  stat_of_output(y, n, cout * hw, stat_out);
  return FQ_OK;
}

// ---- weights ------------------------------------------------------------------------------------------------------
// LinearQuantizeSTE.forward, ste_func.py:37-41
int fq_ste_forward_host(const float* x, float* y, int64_t rows, int64_t row_len, const float* scales, int has_clip,
                        float clip_lo, float clip_hi, float eps, fqStream_t) {
  REQUIRE(x && y && scales && rows > 0 && row_len > 0, "fq_ste_forward_host: bad arguments");
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < rows * row_len; ++i) {
    const float s = scales[i / row_len];
    float v = x[i];
    if (has_clip) v = clipf(v, clip_lo, clip_hi);
    y[i] = round_half_away(v / (s + eps)) * s;
  }
  return FQ_OK;
}

// convert_conv2d.py:70-95 / convert_dense.py:52-63
int fq_weight_fake_quant_host(const float* w, float* w_q, int64_t rows, int64_t row_len, int width, float* scales_out,
                              void*, fqStream_t) {
  REQUIRE(w && w_q && rows > 0 && row_len > 0 && width >= 2 && width <= 16, "fq_weight_fake_quant_host: bad arguments");
  const float levels = (float)((1 << (width - 1)) - 1);
  std::vector<float> rmax((size_t)rows);
  per_sample_stat(w, rows, row_len, true, rmax.data());
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r) {
    const float s = rmax[(size_t)r] / levels;
    const float d = s + kEps;
    if (scales_out) scales_out[r] = s;
    for (int64_t i = 0; i < row_len; ++i) w_q[r * row_len + i] = round_half_away(w[r * row_len + i] / d) * s;
  }
  return FQ_OK;
}

// convert_conv2d.py:71-83 + wino_matrix.py: U = G g G^T (k-sequential, separately rounded multiply and add), per-channel
// abs-max in the Winograd domain, STE, back through the pseudo-inverses
int fq_wino_weight_fake_quant_host(const float* w, float* w_q, int64_t cout, int64_t cin_g, int t, const float* G,
                                   const float* GI, const float* GTI, int width, float* scales_out, void*,
                                   fqStream_t) {
  REQUIRE(w && w_q && G && GI && GTI, "fq_wino_weight_fake_quant_host: null pointer");
  REQUIRE(t == 4 || t == 6 || t == 8, "fq_wino_weight_fake_quant_host: t must be 4, 6 or 8");
  REQUIRE(cout > 0 && cin_g > 0 && width >= 2 && width <= 16, "fq_wino_weight_fake_quant_host: bad arguments");
  const float levels = (float)((1 << (width - 1)) - 1);
  auto forward = [&](const float* g9, float* U) {
    float t1[8][3];
    for (int a = 0; a < t; ++a)
      for (int j = 0; j < 3; ++j) {
        float acc = G[a * 3 + 0] * g9[0 * 3 + j];
        acc = acc + G[a * 3 + 1] * g9[1 * 3 + j];
        acc = acc + G[a * 3 + 2] * g9[2 * 3 + j];
        t1[a][j] = acc;
      }
    for (int a = 0; a < t; ++a)
      for (int b = 0; b < t; ++b) {
        float acc = t1[a][0] * G[b * 3 + 0];
        acc = acc + t1[a][1] * G[b * 3 + 1];
        acc = acc + t1[a][2] * G[b * 3 + 2];
        U[a * t + b] = acc;
      }
  };
#pragma omp parallel for schedule(static)
  for (int64_t co = 0; co < cout; ++co) {
    const float* wc = w + co * cin_g * 9;
    float* oc = w_q + co * cin_g * 9;
    float U[64];
    float m = 0.0f;
    for (int64_t ci = 0; ci < cin_g; ++ci) {
      forward(wc + ci * 9, U);
      for (int k = 0; k < t * t; ++k) m = fmaxf(m, fabsf(U[k]));
    }
    const float s = m / levels, d = s + kEps;
    if (scales_out) scales_out[co] = s;
    for (int64_t ci = 0; ci < cin_g; ++ci) {
      forward(wc + ci * 9, U);
      for (int k = 0; k < t * t; ++k) U[k] = round_half_away(U[k] / d) * s;
      float t2[3][8];
      for (int i = 0; i < 3; ++i)
        for (int b = 0; b < t; ++b) {
          float acc = GI[i * t + 0] * U[0 * t + b];
          for (int a = 1; a < t; ++a) acc = acc + GI[i * t + a] * U[a * t + b];
          t2[i][b] = acc;
        }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          float acc = t2[i][0] * GTI[0 * 3 + j];
          for (int b = 1; b < t; ++b) acc = acc + t2[i][b] * GTI[b * 3 + j];
          oc[ci * 9 + i * 3 + j] = acc;
        }
    }
  }
  return FQ_OK;
}

// ---- calibration ----------------------------------------------------------------------------------------------------
// `_update_ema`, convert.py:66-79
int fq_ema_update_host(float* state, const float* current, int64_t count, double momentum, fqStream_t) {
  REQUIRE(state && current && count > 0, "fq_ema_update_host: bad arguments");
  const float omm = (float)(1.0 - momentum), m = (float)momentum;
  for (int64_t i = 0; i < count; ++i) {
    const float a = omm * current[i];
    const float b = state[i] * m;
    state[i] = a + b;
  }
  return FQ_OK;
}

// distribution_calibrate.py:33-34
int fq_global_max_host(const float* x, int64_t numel, float* out, fqStream_t) {
  REQUIRE(x && out && numel > 0, "fq_global_max_host: bad arguments");
  float m = -INFINITY;
#pragma omp parallel for reduction(max : m) schedule(static)
  for (int64_t i = 0; i < numel; ++i) m = fmaxf(m, x[i]);
  out[0] = m;
  return FQ_OK;
}

// `_discrete_histogram` + accumulation, distribution_calibrate.py:31-47,103-104 (index == bins clamped)
int fq_histogram_accumulate_host(const float* x, int64_t numel, const float* max_dev, int bins, uint64_t* hist,
                                 uint32_t* neg_count, fqStream_t) {
  REQUIRE(x && max_dev && hist && numel > 0 && bins > 0, "fq_histogram_accumulate_host: bad arguments");
  const float mx = max_dev[0];
  const float scales = (float)bins / (mx + 1e-5f);
  uint64_t negs = 0;
#pragma omp parallel
  {
    std::vector<uint64_t> mine((size_t)bins, 0);
    uint64_t neg = 0;
#pragma omp for schedule(static) nowait
    for (int64_t i = 0; i < numel; ++i) {
      const float v = x[i];
      neg += (v < 0.0f) ? 1u : 0u;
      const float c = clipf(v, 0.0f, mx);
      if (c != 0.0f) {
        int idx = (int)(c * scales);
        idx = idx < bins ? idx : bins - 1;
        mine[(size_t)idx] += 1;
      }
    }
#pragma omp critical(fq_hist)
    {
      for (int b = 0; b < bins; ++b) hist[b] += mine[(size_t)b];
      negs += neg;
    }
  }
  if (neg_count) *neg_count += (uint32_t)negs;
  return FQ_OK;
}

// the producers that bin what they store (fq_bn_act_stat_hist / fq_add_act_stat_hist): the plain pass, then the histogram of
// its result - which is what the device forms are defined to equal
int fq_bn_act_stat_hist_host(const float* x, float* y, int64_t n, int64_t c, int64_t hw, const float* scale,
                             const float* shift, int act, float* stat_out, const float* hist_max, int bins, uint64_t* hist,
                             uint32_t* neg_count, fqStream_t stream) {
  REQUIRE(stat_out && hist_max && hist && bins > 0 && bins <= 4096, "fq_bn_act_stat_hist_host: bad arguments");
  if (int rc = fq_bn_act_stat_host(x, y, n, c, hw, scale, shift, act, stat_out, stream)) return rc;
  return fq_histogram_accumulate_host(y, n * c * hw, hist_max, bins, hist, neg_count, stream);
}

int fq_add_act_stat_hist_host(const float* a, const float* b, float* y, int64_t n, int64_t inner, int act, float* stat_out,
                              const float* hist_max, int bins, uint64_t* hist, uint32_t* neg_count, fqStream_t stream) {
  REQUIRE(stat_out && hist_max && hist && bins > 0 && bins <= 4096, "fq_add_act_stat_hist_host: bad arguments");
  if (int rc = fq_add_act_stat_host(a, b, y, n, inner, act, stat_out, stream)) return rc;
  return fq_histogram_accumulate_host(y, n * inner, hist_max, bins, hist, neg_count, stream);
}

// BatchNorm + residual + activation in one pass: on the host the two passes it is defined to equal
int fq_bn_add_act_stat_host(const float* x, const float* residual, float* y, int64_t n, int64_t c, int64_t hw,
                            const float* scale, const float* shift, int act, float* stat_out, fqStream_t stream) {
  REQUIRE(x && residual && y && stat_out && n > 0 && c > 0 && hw > 0, "fq_bn_add_act_stat_host: bad arguments");
  std::vector<float> t((size_t)n * c * hw);
  if (int rc = fq_bn_act_stat_host(x, t.data(), n, c, hw, scale, shift, FQ_ACT_NONE, nullptr, stream)) return rc;
  return fq_add_act_stat_host(t.data(), residual, y, n, c * hw, act, stat_out, stream);
}

int fq_bn_add_act_stat_hist_host(const float* x, const float* residual, float* y, int64_t n, int64_t c, int64_t hw,
                                 const float* scale, const float* shift, int act, float* stat_out, const float* hist_max,
                                 int bins, uint64_t* hist, uint32_t* neg_count, fqStream_t stream) {
  REQUIRE(stat_out && hist_max && hist && bins > 0 && bins <= 4096, "fq_bn_add_act_stat_hist_host: bad arguments");
  if (int rc = fq_bn_add_act_stat_host(x, residual, y, n, c, hw, scale, shift, act, stat_out, stream)) return rc;
  return fq_histogram_accumulate_host(y, n * c * hw, hist_max, bins, hist, neg_count, stream);
}

int fq_hist_to_float_host(const uint64_t* hist, float* out, int64_t count, fqStream_t) {
  REQUIRE(hist && out && count > 0, "fq_hist_to_float_host: bad arguments");
  for (int64_t i = 0; i < count; ++i) out[i] = (float)hist[i];
  return FQ_OK;
}

// `kl_calibrate`, distribution_calibrate.py:117-171: every sum in the reference's order and precision
int fq_kl_search_host(const float* hist, int64_t L, int bins, int levels, int min_bins, int32_t* out_best, void*,
                      fqStream_t) {
  REQUIRE(hist && out_best && L > 0, "fq_kl_search_host: bad arguments");
  REQUIRE(min_bins >= levels, "min_bins should be greater than levels (%d vs. %d)", min_bins, levels);
  REQUIRE(levels >= 2 && bins > min_bins, "fq_kl_search_host: bad bins / levels");
  std::vector<double> div((size_t)L * bins, std::numeric_limits<double>::infinity());
#pragma omp parallel
  {
    std::vector<double> q((size_t)levels);
#pragma omp for collapse(2) schedule(dynamic, 8)
    for (int64_t layer = 0; layer < L; ++layer)
      for (int i = min_bins; i < bins; ++i) {
        const float* d = hist + layer * bins;
        float tail = 0.0f;                                                  // :143-144 python sum() over fp32
        for (int j = i; j < bins; ++j) tail = tail + d[j];
        const float plast = d[i - 1] + tail;
        float s = 0.0f;                                                     // :145
        for (int j = 0; j < i - 1; ++j) s = s + d[j];
        s = s + plast;
        for (int l = 0; l < levels; ++l) q[(size_t)l] = 0.0;                // :149-152
        const double di = (double)i;
        for (int j = 0; j < i; ++j) {
          const int fl = (int)((double)((long long)j * levels) / di);
          q[(size_t)fl] += (double)d[j];
        }
        double qs = 0.0;                                                    // :154-161
        for (int j = 0; j < i; ++j) {
          const double b = (double)((long long)j * levels) / di;
          const int fl = (int)b;
          int ce = (int)ceil(b);
          ce = ce > levels - 1 ? levels - 1 : ce;
          const double qf = q[(size_t)fl];
          double qe = (q[(size_t)ce] - qf) * (b - (double)fl) + qf;
          const float pj = ((j == i - 1) ? plast : d[j]) / s;
          qe = qe * ((pj != 0.0f) ? 1.0 : 0.0);
          qs = qs + qe;
        }
        double dv = 0.0;                                                    // :164-166
        for (int j = 0; j < i; ++j) {
          const double b = (double)((long long)j * levels) / di;
          const int fl = (int)b;
          int ce = (int)ceil(b);
          ce = ce > levels - 1 ? levels - 1 : ce;
          const double qf = q[(size_t)fl];
          double qe = (q[(size_t)ce] - qf) * (b - (double)fl) + qf;
          const float pj = ((j == i - 1) ? plast : d[j]) / s;
          qe = qe * ((pj != 0.0f) ? 1.0 : 0.0);
          qe = qe / qs;
          if (qe != 0.0) dv = dv + (double)pj * log((double)pj / qe);
        }
        div[(size_t)(layer * bins + i)] = dv;
      }
  }
  for (int64_t layer = 0; layer < L; ++layer) {                             // :167-169 strict <, first minimum wins
    double m = std::numeric_limits<double>::infinity();
    int b = min_bins;
    for (int i = min_bins; i < bins; ++i)
      if (div[(size_t)(layer * bins + i)] < m) {
        m = div[(size_t)(layer * bins + i)];
        b = i;
      }
    out_best[layer] = b;
  }
  return FQ_OK;
}

// ---- int-code path, nn/quantized_conv.py:54-76 ---------------------------------------------------------------------
int fq_quantize_codes_host(const float* x, int32_t* codes, int64_t numel, int mode, float* range_dev, void*,
                           fqStream_t) {
  REQUIRE(x && codes && range_dev && numel > 0, "fq_quantize_codes_host: bad arguments");
  REQUIRE(mode >= FQ_CODES_INT8 && mode <= FQ_CODES_SCALE, "unknown out type: %d", mode);
  float mn, mx;
  if (mode == FQ_CODES_INT8) {
    float m = 0.0f;
#pragma omp parallel for reduction(max : m) schedule(static)
    for (int64_t i = 0; i < numel; ++i) m = fmaxf(m, fabsf(x[i]));
    mx = m;
    mn = -m;
  } else if (mode == FQ_CODES_UINT8) {
    float hi = -INFINITY, lo = INFINITY;
#pragma omp parallel for reduction(max : hi) reduction(min : lo) schedule(static)
    for (int64_t i = 0; i < numel; ++i) {
      hi = fmaxf(hi, x[i]);
      lo = fminf(lo, x[i]);
    }
    mx = hi;
    mn = lo;
  } else {
    mn = range_dev[0];
    mx = range_dev[1];
  }
  range_dev[0] = mn;
  range_dev[1] = mx;
  if (mode != FQ_CODES_SCALE) range_dev[2] = (mx == -mn) ? (mx / 127.0f) : ((mx - mn) / 255.0f);
  const float sc = range_dev[2];
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < numel; ++i) codes[i] = (int32_t)round_half_away(clipf(x[i], mn, mx) / sc);
  return FQ_OK;
}

int fq_dequantize_host(const int32_t* codes, float* y, int64_t numel, const float* scale_dev, fqStream_t) {
  REQUIRE(codes && y && scale_dev && numel > 0, "fq_dequantize_host: bad arguments");
  const float sc = scale_dev[0];
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < numel; ++i) y[i] = (float)codes[i] * sc;
  return FQ_OK;
}

// nn/quantized_conv.py:106-159 (`Conv2D.hybrid_forward`, quantized=True) - the twin of fq_qconv2d_forward: pad, ONE global
// range per tensor (:63-72; int8 [-max|x|, max|x|], uint8 [min, max] of the PADDED tensor, fixed `_input_range`), codes
// round(clip(x) / scale) with scale = max/127 if symmetric else (max - min)/255 - no zero point, no epsilon (:54-61) -, int32
// bias codes (:122-127), integer correlation per group (the im2col + dot of :129-151 as exact integer sums in wrapping int32,
// the reference's cast, :144), activation on the integers, dequantise by in_scale * w_scale (:157-158).
// wbuf: 8 floats written by fq_qconv_weights_prepare_host = {w_max, w_min, w_scale, ...}.
int fq_qconv_weights_prepare_host(const float* w, int64_t cin, int64_t cout, int kh, int kw, int, int, int, int, int groups,
                                  int weight_mode, float w_min, float w_max, void* wbuf, void*, fqStream_t) {
  REQUIRE(w && wbuf && cin > 0 && cout > 0 && groups > 0 && cin % groups == 0 && kh > 0 && kw > 0,
          "fq_qconv_weights_prepare_host: bad arguments");
  REQUIRE(weight_mode >= FQ_CODES_INT8 && weight_mode <= FQ_CODES_RANGE, "unknown out type: %d", weight_mode);
  const int64_t numel = cout * (cin / groups) * kh * kw;
  float mn = w_min, mx = w_max;
  if (weight_mode == FQ_CODES_INT8) {
    float m = 0.0f;
    for (int64_t i = 0; i < numel; ++i) m = fmaxf(m, fabsf(w[i]));
    mx = m;
    mn = -m;
  } else if (weight_mode == FQ_CODES_UINT8) {
    mx = -INFINITY;
    mn = INFINITY;
    for (int64_t i = 0; i < numel; ++i) {
      mx = fmaxf(mx, w[i]);
      mn = fminf(mn, w[i]);
    }
  }
  float* rec = (float*)wbuf;
  rec[0] = mx;
  rec[1] = mn;
  rec[2] = (mx == -mn) ? (mx / 127.0f) : ((mx - mn) / 255.0f);
  for (int i = 3; i < 8; ++i) rec[i] = 0.0f;
  return FQ_OK;
}

int fq_qconv2d_forward_host(const float* x, const float* w, const void* wbuf, const float* bias, float* y, int64_t n,
                            int64_t cin, int64_t h, int64_t wdt, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw,
                            int groups, int input_mode, float in_min, float in_max, const float* in_stat, int act,
                            const float* bn_scale, const float* bn_shift, float* stat_out, void*, int, fqStream_t) {
  REQUIRE(x && w && wbuf && y && n > 0 && cin > 0 && cout > 0 && h > 0 && wdt > 0 && groups > 0 && cin % groups == 0 &&
              cout % groups == 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0,
          "fq_qconv2d_forward_host: bad arguments");
  REQUIRE(input_mode >= FQ_CODES_INT8 && input_mode <= FQ_CODES_RANGE, "unknown out type: %d", input_mode);
  act &= ~FQ_STAT_PREZEROED;
  const int64_t numel = n * cin * h * wdt;
  float mn = in_min, mx = in_max;
  (void)in_stat;      // a producer's per-sample maxima spare the device library its range pass and change no result
  if (input_mode == FQ_CODES_INT8) {
    float m = 0.0f;
#pragma omp parallel for reduction(max : m) schedule(static)
    for (int64_t i = 0; i < numel; ++i) m = fmaxf(m, fabsf(x[i]));
    mx = m;
    mn = -m;
  } else if (input_mode == FQ_CODES_UINT8) {
    float hi = -INFINITY, lo = INFINITY;
#pragma omp parallel for reduction(max : hi) reduction(min : lo) schedule(static)
    for (int64_t i = 0; i < numel; ++i) {
      hi = fmaxf(hi, x[i]);
      lo = fminf(lo, x[i]);
    }
    if (ph > 0 || pw > 0) {                    // the zeros of the padding belong to the tensor whose range is taken (:108-113)
      hi = fmaxf(hi, 0.0f);
      lo = fminf(lo, 0.0f);
    }
    mx = hi;
    mn = lo;
  }
  const float xs = (mx == -mn) ? (mx / 127.0f) : ((mx - mn) / 255.0f);
  const float* wrec = (const float*)wbuf;
  const float wh = wrec[0], wl = wrec[1], ws_ = wrec[2];
  const float deq = xs * ws_;
  const float b_max = deq * 2147483648.0f;
  const int64_t cin_g = cin / groups, cout_g = cout / groups;
  const int64_t ho = (h + 2 * ph - kh) / sh + 1, wo = (wdt + 2 * pw - kw) / sw + 1;
  const int zero_code = (int)round_half_away(clipf(0.0f, mn, mx) / xs);
  // the codes of x once (the direct kernel of the device library quantises where it reads: same values)
  std::vector<int32_t> cx((size_t)numel);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < numel; ++i) cx[(size_t)i] = (int32_t)round_half_away(clipf(x[i], mn, mx) / xs);
  const int64_t wnumel = cout * cin_g * kh * kw;
  std::vector<int32_t> cw((size_t)wnumel);
  for (int64_t i = 0; i < wnumel; ++i) cw[(size_t)i] = (int32_t)round_half_away(clipf(w[i], wl, wh) / ws_);
  if (stat_out != nullptr)
    for (int64_t s = 0; s < n; ++s) stat_out[s] = 0.0f;
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t s = 0; s < n; ++s)
    for (int64_t co = 0; co < cout; ++co) {
      const int64_t g = co / cout_g;
      uint32_t bcode = 0;
      if (bias != nullptr) bcode = (uint32_t)(int64_t)round_half_away(clipf(bias[co], -b_max, b_max) / deq);
      float m = 0.0f;
      for (int64_t oh = 0; oh < ho; ++oh)
        for (int64_t ow = 0; ow < wo; ++ow) {
          uint32_t acc = 0;
          for (int64_t ci = 0; ci < cin_g; ++ci) {
            const int32_t* xp = cx.data() + ((s * cin + g * cin_g + ci) * h) * wdt;
            const int32_t* wp = cw.data() + (co * cin_g + ci) * kh * kw;
            for (int ky = 0; ky < kh; ++ky) {
              const int64_t ih = oh * sh - ph + ky;
              for (int kx = 0; kx < kw; ++kx) {
                const int64_t iw = ow * sw - pw + kx;
                const bool in = ih >= 0 && ih < h && iw >= 0 && iw < wdt;
                const int32_t c = in ? xp[ih * wdt + iw] : zero_code;
                acc += (uint32_t)c * (uint32_t)wp[ky * kw + kx];
              }
            }
          }
          acc += bcode;
          int32_t v = (int32_t)acc;
          if (act == FQ_ACT_RELU && bn_scale == nullptr) v = v > 0 ? v : 0;       // on the integers (:154-155)
          float out = (float)v * deq;
          if (bn_scale != nullptr) {           // a BatchNorm folded behind the block: the activation follows IT
            out = out * bn_scale[co];
            out = out + bn_shift[co];
            if (act == FQ_ACT_RELU) out = fmaxf(out, 0.0f);
            if (act == FQ_ACT_RELU6) out = fminf(fmaxf(out, 0.0f), 6.0f);
          }
          y[((s * cout + co) * ho + oh) * wo + ow] = out;
          m = fmaxf(m, fabsf(out));
        }
      if (stat_out != nullptr) {
#pragma omp critical
        stat_out[s] = fmaxf(stat_out[s], m);
      }
    }
  return FQ_OK;
}

}  // extern "C"
