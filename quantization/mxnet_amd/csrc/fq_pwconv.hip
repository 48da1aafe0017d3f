// libfakequant — fq_pwconv_i8: argument checks and the shape-based choice between the pointwise forms; fq_weight_codes
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_pw.h"

namespace {

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

using namespace fqi;

extern "C" {

#ifdef FQ_PW_TRACE
int fq_debug_set_pw_trace(unsigned long long* buf) {
  FQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pw_trace), &buf, sizeof(buf)));
  return FQ_OK;
}
int fq_debug_set_pw_dbg(int v) {
  FQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pw_dbg), &v, sizeof(v)));
  return FQ_OK;
}
#endif

size_t fq_pwconv_workspace_bytes(int64_t n, int64_t cin_pad, int64_t hw) {
  const int64_t cols_pad = (n * hw + 255) / 256 * 256;
  return (size_t)cols_pad * (size_t)cin_pad + 64;
}

static int pwconv_dispatch(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                           const float* bias, float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw,
                           int stride, int64_t h_in, int64_t w_in, int64_t w_out, const float* in_stat,
                           const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                           const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                           const float* residual, void* ws, fqStream_t stream, bool in_c16 = false,
                           const float* out_thr = nullptr, int out_width = 8, unsigned out_flags = 0,
                           const long long* eval_labels = nullptr, float* eval_counters = nullptr,
                           void* eval_ws = nullptr, bool* range_taken = nullptr, void* y16 = nullptr, bool sub = false,
                           bool gap = false) {
  // range_taken != nullptr: range mode (fq_common.h: kRangeMode) - in_thr is a range record, bias holds int32 codes; only
  // the one-launch forms serve it and *range_taken says whether one took the shape
  const bool range = range_taken != nullptr;
  if (range && x && (!aligned16(x) || !(hw < (1ll << 30) && n * hw < (1ll << 31) - 512))) {
    // a legal input of the stand-alone block that the one-launch forms do not take (a batch slice of an odd-sized tensor, an
    // oversized plane): not an error there - the exact direct kernel serves the layer
    *range_taken = false;
    return FQ_OK;
  }
  FQ_REQUIRE(x && wcodes && wscale && wsum && y && (ws || range), "fq_pwconv_i8: null pointer");
  FQ_REQUIRE(n > 0 && cin > 0 && cout > 0 && hw > 0 && hw < (1ll << 30) && n * hw < (1ll << 31) - 512,
             "fq_pwconv_i8: bad shape");
  FQ_REQUIRE(cin_pad >= cin && cin_pad % 64 == 0 && cin_pad <= 8192, "fq_pwconv_i8: cin_pad=%lld must be a multiple "
             "of 64 covering cin=%lld", (long long)cin_pad, (long long)cin);
  FQ_REQUIRE(in_stat != nullptr || in_thr != nullptr, "fq_pwconv_i8: give in_stat (online), in_thr (offline) or both "
             "(offline, the statistic only feeds out_current_max): the integer path needs a quantised input");
  FQ_REQUIRE(in_thr != nullptr || out_current_max != nullptr, "fq_pwconv_i8: online mode needs out_current_max (the "
             "GEMM reads the batch statistic from it)");
  FQ_REQUIRE(in_width >= 2 && in_width <= 8, "fq_pwconv_i8: input width %d does not fit int8 codes", in_width);
  FQ_REQUIRE(!(in_flags & (FQ_ACT_NO_ABS | FQ_ACT_NO_EPS)), "fq_pwconv_i8: unsupported activation flags");
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_pwconv_i8: bn_scale and bn_shift go together");
  PwCall c;
  c.prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  const int forced_form = (act >> 12) & 15;             // FQ_PW_FORM(f): tests and tuning runs name the form per call
  act &= ~(FQ_STAT_PREZEROED | FQ_PW_FORM(15));
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_pwconv_i8: unknown activation %d", act);
  FQ_REQUIRE(aligned16(wcodes) && aligned16(ws) && aligned16(x), "fq_pwconv_i8: x, wcodes and ws must be 16-byte aligned");
  if (range) *range_taken = true;
  c.x = x; c.wcodes = wcodes; c.wscale = wscale; c.wsum = wsum; c.bias = bias; c.y = y;
  c.n = n; c.cin = cin; c.cin_pad = cin_pad; c.cout = cout; c.hw = hw;
  c.stride = stride; c.h_in = h_in; c.w_in = w_in; c.w_out = w_out;
  c.in_stat = in_stat; c.in_thr = in_thr;
  c.levels = act_levels(in_width, in_flags);
  c.lo_neg = range ? kRangeMode : ((in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0);
  c.zoff = (in_flags & FQ_ACT_SIGNED) ? 0 : 128;      // unsigned codes are stored re-centred so they fit int8
  c.out_current_max = out_current_max; c.bn_scale = bn_scale; c.bn_shift = bn_shift; c.act = act;
  c.stat_out = stat_out; c.residual = residual; c.ws = ws; c.st = (hipStream_t)stream;
  c.in_c16 = in_c16; c.out_thr = y16 != nullptr ? nullptr : out_thr;
  if (y16 != nullptr) {                                 // dual output: y fp32 + y16 codes under out_thr
    c.y16 = y16;
    c.dual_thr = out_thr;
  }
  if (out_thr != nullptr) {
    FQ_REQUIRE(out_width >= 2 && out_width <= 8, "fq_pwconv_i8_c16: output width %d does not fit int8 codes", out_width);
    FQ_REQUIRE(!(out_flags & (FQ_ACT_NO_ABS | FQ_ACT_NO_EPS)), "fq_pwconv_i8_c16: unsupported output flags");
    c.out_levels = act_levels(out_width, out_flags);
    c.out_lo_neg = (out_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
    c.out_zoff = (out_flags & FQ_ACT_SIGNED) ? 0 : 128;
  }
  c.eval_labels = eval_labels; c.eval_counters = eval_counters; c.eval_ws = eval_ws;
  c.sub = sub;
  c.gap = gap;
  static const int pw_form = env_int("FQ_PW_FORM", 0);      // 0 auto, 1 two kernels, 3 stream, 6 split, 7 sample, 8 rows, 9 pipe
  c.form = (in_c16 || c.out_thr) ? 6 : (eval_labels ? 8 : (forced_form ? forced_form : pw_form));
  // tuning: FQ_PW_FORM_AT="<pixels per plane>:<form>[,...]" names a form for the layers of one plane size (A/Bs of a form choice
  // inside a model, where a kernel's time alone is not what decides - DESIGN.md 3.6); parsed once
  {
    struct FormAt { int hw[8], form[8], n; };
    static const FormAt at = [] {
      FormAt t;
      t.n = 0;
      if (const char* e = getenv("FQ_PW_FORM_AT")) {
        while (*e && t.n < 8) {
          char* end = nullptr;
          const long a = strtol(e, &end, 10);
          if (end == e || *end != ':') break;
          const long f = strtol(end + 1, &end, 10);
          t.hw[t.n] = (int)a;
          t.form[t.n++] = (int)f;
          if (*end != ',') break;
          e = end + 1;
        }
      }
      return t;
    }();
    if (c.form == 0 && !range)
      for (int k = 0; k < at.n; ++k)
        if (at.hw[k] == (int)hw) c.form = at.form[k];
  }
  FQ_REQUIRE(c.form == 0 || c.form == 1 || c.form == 3 || c.form == 6 || c.form == 7 || c.form == 8 || c.form == 9,
             "fq_pwconv_i8: unknown "
             "form %d (1 two kernels, 3 stream, 6 split, 7 sample, 8 rows, 9 pipe; the panel / chunk / tile forms 2, 4, 5 were retired "
             "in favour of the split form)", c.form);
  FQ_REQUIRE(stride == 1 || c.form == 0 || c.form == 6, "fq_pwconv_i8_strided: only the split form reads strided inputs");
  FQ_REQUIRE(residual == nullptr || c.form != 1, "fq_pwconv_i8_strided: the two-kernel form takes no residual operand");
  // algorithmic bytes: the input pixels the outputs need, the outputs, and the residual operand when there is one
  // (SURVEY.md 8d's definition - 4 B per input and per output element - also when a side is a C16 code tensor: the line of
  // such a run says so and `frac` then measures what the hand-over saves)
  // (the classifier - planes of one pixel - is accounted on its own: 1 MB of latency-bound work is no pointwise layer)
  // (second figure: the bytes really moved - 1 B per element of a side that is a C16 code tensor, 16-channel blocks padded)
  const double in_elems = (double)n * cin * hw, out_elems = (double)n * cout * hw;
  // (a subsampled output: the layer's algorithmic bytes stay those of the whole tensor; moved: the quarter that is stored)
  // (the pooling behind it in the same launch: the layer's and the pooling pass's algorithmic bytes; moved: n * cout means)
  const double stored = gap ? (double)n * cout : (sub ? (double)n * cout * ((h_in + 1) / 2) * ((w_in + 1) / 2) : out_elems);
  const double moved = (in_c16 ? (double)n * ((cin + 15) / 16 * 16) * hw : 4.0 * in_elems) +
                       (c.out_thr != nullptr ? (double)n * ((cout + 15) / 16 * 16) * hw : 4.0 * stored) +
                       (y16 != nullptr ? (double)n * ((cout + 15) / 16 * 16) * (sub ? stored / ((double)n * cout) : (double)hw) : 0.0) +
                       (residual ? 4.0 * out_elems : 0.0);
  ProfScope prof(hw == 1 ? FQ_KERNEL_DENSE : FQ_KERNEL_PWCONV,
                 4.0 * (in_elems + (residual ? 2.0 : 1.0) * out_elems + (gap ? out_elems + (double)n * cout : 0.0)), c.st, moved);
  bool taken = false;
  out_thr = c.out_thr;                                  // (from here on: "y is a C16 tensor")
  if (gap) {                                            // (the sample form's whole-plane instantiations alone pool)
    FQ_REQUIRE(c.form == 0 || c.form == 7, "fq_pwconv_i8_gap: only the sample form pools");
    if (int rc = pw_try_sample(c, &taken)) return rc;
    FQ_REQUIRE(taken, "fq_pwconv_i8_gap: shape not taken (see fq_pwconv_i8_gap_supported)");
    return FQ_OK;
  }
  if (sub) {                                            // (the split form alone stores subsampled)
    FQ_REQUIRE(c.form == 0 || c.form == 6, "fq_pwconv_i8_sub2: only the split form stores a subsampled output");
    if (int rc = pw_try_split(c, &taken)) return rc;
    FQ_REQUIRE(taken, "fq_pwconv_i8_sub2: shape not taken (see fq_pwconv_i8_sub2_supported)");
    return FQ_OK;
  }
  if (hw == 1 && !(in_c16 || out_thr) && !range) {      // (the rows form is not built for range records)
    if (int rc = pw_try_rows(c, &taken)) return rc;
    if (taken) return FQ_OK;
  }
  FQ_REQUIRE(c.form != 8, "fq_pwconv_i8: the rows form takes planes of one pixel");
  if (!(in_c16 || out_thr)) {
#ifdef FQ_DEV_FORMS
    if (int rc = pw_try_pipe(c, &taken)) return rc;
    if (taken) return FQ_OK;
    FQ_REQUIRE(c.form != 9, "fq_pwconv_i8: the pipe form takes stride 1, no residual, fp32 in and out, Cin = 256 or 512, Cout a "
               "multiple of 512 and planes of a multiple of four pixels (16..1024)");
#else
    FQ_REQUIRE(c.form != 9, "fq_pwconv_i8: the pipe form (9) was measured and shelved (DESIGN.md 3.3): this library was built "
               "without it - `python -m quantization.mxnet_amd.csrc.build --dev` builds libfakequant_dev.so with it");
#endif
    if (int rc = pw_try_sample(c, &taken)) return rc;
    if (taken) return FQ_OK;
  } else if (out_thr && !in_c16) {                      // C16 output on the largest planes: the streaming form writes it too
    if (int rc = pw_try_stream(c, &taken)) return rc;
    if (taken) return FQ_OK;
  }
  if (!taken && pw_stream_thin_takes(c)) {              // thin layers on large planes (C16 input, ragged Cin / Cout): round 4
    if (int rc = pw_try_stream(c, &taken)) return rc;
    if (taken) return FQ_OK;
  }
  FQ_REQUIRE(c.form != 7, "fq_pwconv_i8: the sample form takes stride 1, no residual, Cout a multiple of 256 and planes that "
             "cut into blocks of 96..128 pixels (Cin 128 ... 1024; a residual operand with 512 channels per workgroup) or whole planes "
             "of 45..64 pixels (Cin 512 / 1024, no residual)");
  if (int rc = pw_try_split(c, &taken)) return rc;
  if (taken) return FQ_OK;
  if (int rc = pw_try_stream(c, &taken)) return rc;
  if (taken) return FQ_OK;
  FQ_REQUIRE(residual == nullptr, "fq_pwconv_i8_strided: a residual operand needs a shape the split or the streaming "
             "form takes");
  if (range) {                                          // (the two-kernel form is not built for range records)
    *range_taken = false;
    prof.cancel();
    return FQ_OK;
  }
  return pw_two_kernels(c);
}

int fq_pwconv_i8(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                 float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw, const float* in_stat,
                 const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                 const float* bn_scale, const float* bn_shift, int act, float* stat_out, void* ws,
                 fqStream_t stream) {
  return pwconv_dispatch(x, wcodes, wscale, wsum, bias, y, n, cin, cin_pad, cout, hw, 1, 0, 0, 0, in_stat, in_thr,
                         in_width, in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, nullptr, ws, stream);
}

// The statistic-only pass in front of fq_pwdw_fused (round 6): the per-sample maxima fq_pwconv_i8 would leave in stat_out
// (and its out_current_max) without computing, let alone storing, more of the output than its extreme sums (fq_pwdw.hip, K2z).
int fq_pwconv_i8_stat_supported(int64_t n, int64_t cin, int64_t cout, int64_t hw) {
  return pw_stat_shape_ok(n, cin, cout, hw) ? 1 : 0;
}

int fq_pwconv_i8_stat(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                      int64_t n, int64_t cin, int64_t cin_pad, int64_t cout_pad, int64_t cout, int64_t hw,
                      const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                      const float* bn_scale, const float* bn_shift, int act, float* stat_out, fqStream_t stream) {
  FQ_REQUIRE(x && wcodes && wscale && wsum && stat_out, "fq_pwconv_i8_stat: null pointer");
  FQ_REQUIRE(pw_stat_shape_ok(n, cin, cout, hw), "fq_pwconv_i8_stat: shape not taken (n=%lld cin=%lld cout=%lld hw=%lld): see "
             "fq_pwconv_i8_stat_supported", (long long)n, (long long)cin, (long long)cout, (long long)hw);
  FQ_REQUIRE(cin_pad >= cin && cin_pad % 64 == 0 && cin_pad <= 8192 && cout_pad >= cout && cout_pad % 32 == 0,
             "fq_pwconv_i8_stat: cin_pad / cout_pad must be those of fq_weight_codes");
  FQ_REQUIRE(in_stat != nullptr || in_thr != nullptr, "fq_pwconv_i8_stat: give in_stat (online), in_thr (offline) or both");
  FQ_REQUIRE(in_width >= 2 && in_width <= 8, "fq_pwconv_i8_stat: input width %d does not fit int8 codes", in_width);
  FQ_REQUIRE(!(in_flags & (FQ_ACT_NO_ABS | FQ_ACT_NO_EPS)), "fq_pwconv_i8_stat: unsupported activation flags");
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_pwconv_i8_stat: bn_scale and bn_shift go together");
  PwCall c;
  c.prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_pwconv_i8_stat: unknown activation %d", act);
  FQ_REQUIRE(aligned16(wcodes) && aligned16(x), "fq_pwconv_i8_stat: x and wcodes must be 16-byte aligned");
  c.x = x; c.wcodes = wcodes + cout_pad * cin_pad; c.wscale = wscale; c.wsum = wsum; c.bias = bias; c.y = nullptr;
  c.n = n; c.cin = cin; c.cin_pad = cin_pad; c.cout = cout; c.hw = hw; c.stride = 1; c.h_in = c.w_in = c.w_out = 0;
  c.in_stat = in_stat; c.in_thr = in_thr;
  c.levels = act_levels(in_width, in_flags);
  c.lo_neg = (in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
  c.zoff = (in_flags & FQ_ACT_SIGNED) ? 0 : 128;
  c.out_current_max = out_current_max; c.bn_scale = bn_scale; c.bn_shift = bn_shift; c.act = act;
  c.stat_out = stat_out; c.residual = nullptr; c.ws = nullptr; c.st = (hipStream_t)stream; c.form = 0;
  // algorithmic bytes: the pointwise layer's (this launch stands for it); moved: its input only
  const double in_elems = (double)n * cin * hw, out_elems = (double)n * cout * hw;
  ProfScope prof(FQ_KERNEL_PWCONV, 4.0 * (in_elems + out_elems), c.st, 4.0 * in_elems);
  return pw_stat_launch(c);
}

int fq_pwconv_i8_strided(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                         const float* bias, float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h,
                         int64_t w, int stride, const float* in_stat, const float* in_thr, int in_width,
                         unsigned in_flags, float* out_current_max, const float* bn_scale, const float* bn_shift,
                         int act, float* stat_out, const float* residual, void* ws, fqStream_t stream) {
  FQ_REQUIRE(h > 0 && w > 0 && (stride == 1 || stride == 2), "fq_pwconv_i8_strided: bad plane %lld x %lld or stride %d",
             (long long)h, (long long)w, stride);
  const int64_t ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
  return pwconv_dispatch(x, wcodes, wscale, wsum, bias, y, n, cin, cin_pad, cout, ho * wo, stride, h, w, wo, in_stat,
                         in_thr, in_width, in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, residual, ws,
                         stream);
}

// The closing 1x1 of a ResNet-v1 stage when both readers of its output are stride-2 1x1 convolutions (round 6): the values of
// fq_pwconv_i8_strided with stride 1, of which only y[:, :, ::2, ::2] is stored - densely, (n, cout, ceil(h/2), ceil(w/2)) -
// while stat_out (and the residual operand) cover all of y.  The readers then run with stride 1 on a quarter of the bytes.
int fq_pwconv_i8_sub2_supported(int64_t cin, int64_t cout) {
  return pw_split_sub_shape_ok((cin + 63) / 64 * 64, cout) ? 1 : 0;
}

int fq_pwconv_i8_sub2(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                      float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                      const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                      const float* bn_scale, const float* bn_shift, int act, float* stat_out, const float* residual, void* ws,
                      fqStream_t stream) {
  FQ_REQUIRE(h > 0 && w > 0, "fq_pwconv_i8_sub2: bad plane %lld x %lld", (long long)h, (long long)w);
  FQ_REQUIRE(pw_split_sub_shape_ok(cin_pad, cout), "fq_pwconv_i8_sub2: built for 64 / 128 / 256 / 512 (padded) input channels "
             "and more than 128 output channels (got cin_pad=%lld cout=%lld)", (long long)cin_pad, (long long)cout);
  return pwconv_dispatch(x, wcodes, wscale, wsum, bias, y, n, cin, cin_pad, cout, h * w, 1, h, w, w, in_stat, in_thr,
                         in_width, in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, residual, ws, stream, false,
                         nullptr, 8, 0, nullptr, nullptr, nullptr, nullptr, nullptr, true);
}

// A 1x1 convolution and the global average pooling behind it in ONE launch (round 6): the last 1x1 of the MobileNets
// (features: ... Conv2D 1x1, BatchNorm, ReLU, GlobalAvgPool2D, Flatten), whose output has no other reader.  y: (n, cout) = what
// fq_global_avg_pool_stat gives for the output of fq_pwconv_i8_strided(stride 1, residual): the same values added up in the same
// order (pixels 0 .. hw - 1, fp64) - bit for bit; stat_out[n] = max|y[n]|.  Whole planes of 45 .. 64 pixels (7x7, 8x8).
int fq_pwconv_i8_gap_supported(int64_t n, int64_t cin, int64_t cout, int64_t hw, int has_residual) {
  return pw_sample_gap_shape_ok(n, cin, cout, hw, has_residual != 0) ? 1 : 0;
}

int fq_pwconv_i8_gap(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                     float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw, const float* in_stat,
                     const float* in_thr, int in_width, unsigned in_flags, float* out_current_max, const float* bn_scale,
                     const float* bn_shift, int act, float* stat_out, const float* residual, void* ws, fqStream_t stream) {
  FQ_REQUIRE(cin == cin_pad && pw_sample_gap_shape_ok(n, cin, cout, hw, residual != nullptr),
             "fq_pwconv_i8_gap: shape not taken (n=%lld cin=%lld cout=%lld hw=%lld%s): see fq_pwconv_i8_gap_supported",
             (long long)n, (long long)cin, (long long)cout, (long long)hw, residual ? ", residual" : "");
  return pwconv_dispatch(x, wcodes, wscale, wsum, bias, y, n, cin, cin_pad, cout, hw, 1, 0, 0, 0, in_stat, in_thr, in_width,
                         in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, residual, ws, stream, false, nullptr, 8, 0,
                         nullptr, nullptr, nullptr, nullptr, nullptr, false, true);
}

// The closing 1x1 convolution of a residual unit and the unit's shortcut convolution in ONE launch (round 6; K2s, fq_pw_short.hip):
//   y = act(BN3(conv3(x)) + BNd(convd(x2)))
// - the values fq_pwconv_i8_strided(x2 ..., stride 1, no activation) followed by fq_pwconv_i8_strided(x ..., residual = that)
// give, bit for bit, without the shortcut tensor.  Both inputs fp32 (n, cin / cin2, hw); both convolutions quantise their inputs
// on load with their own statistic or threshold and leave their own current_input_max.
int fq_pwconv_i8_shortcut_supported(int64_t cin, int64_t cin2, int64_t cout) {
  return pw_short_shape_ok(cin, cin2, cout) ? 1 : 0;
}

static int pwconv_shortcut(const void* xv, bool x_c16, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                           const float* bias, float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw,
                           const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                           const float* bn_scale, const float* bn_shift, int act, float* stat_out, const void* x2v, bool x2_c16,
                           const int8_t* wcodes2, const float* wscale2, const int32_t* wsum2, int64_t cin2, int64_t cin2_pad,
                           const float* in_stat2, const float* in_thr2, int in_width2, unsigned in_flags2,
                           float* out_current_max2, const float* bn_scale2, const float* bn_shift2, const float* out_thr,
                           int out_width, unsigned out_flags, fqStream_t stream) {
  const float* x = (const float*)xv;
  const float* x2 = (const float*)x2v;
  FQ_REQUIRE(x && wcodes && wscale && wsum && y && x2 && wcodes2 && wscale2 && wsum2 && bn_scale2 && bn_shift2,
             "fq_pwconv_i8_shortcut: null pointer (the shortcut convolution needs its BatchNorm constants)");
  FQ_REQUIRE(n > 0 && hw > 0 && hw < (1ll << 30) && n * hw < (1ll << 31) - 512, "fq_pwconv_i8_shortcut: bad shape");
  FQ_REQUIRE((in_stat || in_thr) && (in_stat2 || in_thr2), "fq_pwconv_i8_shortcut: each convolution needs in_stat (online) or in_thr");
  FQ_REQUIRE((in_thr || out_current_max) && (in_thr2 || out_current_max2), "fq_pwconv_i8_shortcut: online mode needs out_current_max");
  FQ_REQUIRE(in_width >= 2 && in_width <= 8 && in_width2 >= 2 && in_width2 <= 8, "fq_pwconv_i8_shortcut: input widths must fit int8 codes");
  FQ_REQUIRE(!((in_flags | in_flags2) & (FQ_ACT_NO_ABS | FQ_ACT_NO_EPS)), "fq_pwconv_i8_shortcut: unsupported activation flags");
  FQ_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "fq_pwconv_i8_shortcut: bn_scale and bn_shift go together");
  PwCall a, b;
  a.prezeroed = (act & FQ_STAT_PREZEROED) != 0;
  act &= ~FQ_STAT_PREZEROED;
  FQ_REQUIRE(act >= FQ_ACT_NONE && act <= FQ_ACT_RELU6, "fq_pwconv_i8_shortcut: unknown activation %d", act);
  FQ_REQUIRE(aligned16(wcodes) && aligned16(wcodes2), "fq_pwconv_i8_shortcut: the weight codes must be 16-byte aligned");
  a.x = x; a.wcodes = wcodes; a.wscale = wscale; a.wsum = wsum; a.bias = bias; a.y = y;
  a.n = n; a.cin = cin; a.cin_pad = cin_pad; a.cout = cout; a.hw = hw; a.stride = 1; a.h_in = a.w_in = a.w_out = 0;
  a.in_stat = in_stat; a.in_thr = in_thr; a.levels = act_levels(in_width, in_flags);
  a.lo_neg = (in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0; a.zoff = (in_flags & FQ_ACT_SIGNED) ? 0 : 128;
  a.out_current_max = out_current_max; a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.act = act; a.stat_out = stat_out;
  a.residual = nullptr; a.ws = nullptr; a.st = (hipStream_t)stream; a.form = 0;
  b = a;
  b.x = x2; b.wcodes = wcodes2; b.wscale = wscale2; b.wsum = wsum2; b.bias = nullptr; b.y = nullptr;
  b.cin = cin2; b.cin_pad = cin2_pad; b.in_stat = in_stat2; b.in_thr = in_thr2; b.levels = act_levels(in_width2, in_flags2);
  b.lo_neg = (in_flags2 & FQ_ACT_LO_NEG_MAX) ? 1 : 0; b.zoff = (in_flags2 & FQ_ACT_SIGNED) ? 0 : 128;
  b.out_current_max = out_current_max2; b.bn_scale = bn_scale2; b.bn_shift = bn_shift2; b.act = FQ_ACT_NONE; b.stat_out = nullptr;
  a.in_c16 = x_c16;
  b.in_c16 = x2_c16;
  if (y16 != nullptr) {                                 // the code copy of y under the next unit's first convolution's threshold
    FQ_REQUIRE(out_thr != nullptr && out_width >= 2 && out_width <= 8 && !(out_flags & (FQ_ACT_SIGNED | FQ_ACT_LO_NEG_MAX | FQ_ACT_NO_ABS | FQ_ACT_NO_EPS)),
               "fq_pwconv_i8_shortcut_c16: the code copy holds unsigned codes of a [0, thr] range and needs out_thr");
    a.y16 = y16; a.dual_thr = out_thr; a.out_levels = act_levels(out_width, out_flags); a.out_lo_neg = 0; a.out_zoff = 128;
    FQ_REQUIRE((32 / hw + 2) * ((cout + 15) / 16) * hw * 16 < (1ll << 31), "fq_pwconv_i8_shortcut_c16: plane too large");
  }
  // algorithmic bytes: those of the two layers it stands for (shortcut: in + out; closing: in + residual + out); moved: both
  // inputs and the output
  const double in1 = (double)n * cin * hw, in2 = (double)n * cin2 * hw, out = (double)n * cout * hw;
  ProfScope prof(FQ_KERNEL_PWCONV, 4.0 * (in2 + out + in1 + 2.0 * out), a.st,
                 (x_c16 ? 1.0 : 4.0) * in1 + (x2_c16 ? 1.0 : 4.0) * in2 + (y16 != nullptr ? 5.0 : 4.0) * out);
  return pw_short_launch(a, b);
}

int fq_pwconv_i8_shortcut(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                          float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw, const float* in_stat,
                          const float* in_thr, int in_width, unsigned in_flags, float* out_current_max, const float* bn_scale,
                          const float* bn_shift, int act, float* stat_out, const float* x2, const int8_t* wcodes2,
                          const float* wscale2, const int32_t* wsum2, int64_t cin2, int64_t cin2_pad, const float* in_stat2,
                          const float* in_thr2, int in_width2, unsigned in_flags2, float* out_current_max2,
                          const float* bn_scale2, const float* bn_shift2, fqStream_t stream) {
  return pwconv_shortcut(x, false, wcodes, wscale, wsum, bias, y, nullptr, n, cin, cin_pad, cout, hw, in_stat, in_thr, in_width,
                         in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, x2, false, wcodes2, wscale2, wsum2, cin2,
                         cin2_pad, in_stat2, in_thr2, in_width2, in_flags2, out_current_max2, bn_scale2, bn_shift2, nullptr, 8, 0,
                         stream);
}

// ... under stored thresholds: x a C16 code tensor (the unit's 3x3 handed its codes over), y fp32 AND y16 its code copy for the next
// unit's first 1x1 (fq_pwconv_i8_c16_dual's pair of outputs), the shortcut convolution's input fp32 or a C16 tensor.
int fq_pwconv_i8_shortcut_c16(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                              float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw,
                              const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                              const float* bn_scale, const float* bn_shift, int act, float* stat_out, const void* x2, int x2_is_c16,
                              const int8_t* wcodes2, const float* wscale2, const int32_t* wsum2, int64_t cin2, int64_t cin2_pad,
                              const float* in_stat2, const float* in_thr2, int in_width2, unsigned in_flags2,
                              float* out_current_max2, const float* bn_scale2, const float* bn_shift2, const float* out_thr,
                              int out_width, unsigned out_flags, fqStream_t stream) {
  FQ_REQUIRE(y16 != nullptr && in_thr != nullptr, "fq_pwconv_i8_shortcut_c16: null pointer (y16, in_thr: codes in, code copy out)");
  return pwconv_shortcut(x, true, wcodes, wscale, wsum, bias, y, y16, n, cin, cin_pad, cout, hw, in_stat, in_thr, in_width,
                         in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, x2, x2_is_c16 != 0, wcodes2, wscale2, wsum2,
                         cin2, cin2_pad, in_stat2, in_thr2, in_width2, in_flags2, out_current_max2, bn_scale2, bn_shift2, out_thr,
                         out_width, out_flags, stream);
}

int fq_pwconv_i8_c16(const void* x, int x_is_c16, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                     const float* bias, void* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                     int stride, const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                     float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                     const float* residual, const float* out_thr, int out_width, unsigned out_flags, void* ws,
                     fqStream_t stream) {
  FQ_REQUIRE(h > 0 && w > 0 && (stride == 1 || stride == 2), "fq_pwconv_i8_c16: bad plane %lld x %lld or stride %d",
             (long long)h, (long long)w, stride);
  FQ_REQUIRE(x_is_c16 || out_thr != nullptr, "fq_pwconv_i8_c16: neither side is a C16 tensor (use fq_pwconv_i8)");
  const int64_t ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
  const int64_t cbi = (cin + 15) / 16, cbo = (cout + 15) / 16;
  FQ_REQUIRE(!x_is_c16 || (32 / (ho * wo) + 2) * cbi * h * w * 16 < (1ll << 31), "fq_pwconv_i8_c16: input plane too large");
  FQ_REQUIRE(out_thr == nullptr || (32 / (ho * wo) + 2) * cbo * ho * wo * 16 < (1ll << 31),
             "fq_pwconv_i8_c16: output plane too large");
  return pwconv_dispatch((const float*)x, wcodes, wscale, wsum, bias, (float*)y, n, cin, cin_pad, cout, ho * wo, stride, h,
                         w, wo, in_stat, in_thr, in_width, in_flags, out_current_max, bn_scale, bn_shift, act, stat_out,
                         residual, ws, stream, x_is_c16 != 0, out_thr, out_width, out_flags);
}

static int pwconv_dual(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                       float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                       const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                       const float* bn_scale, const float* bn_shift, int act, float* stat_out, const float* residual,
                       const float* out_thr, int out_width, unsigned out_flags, void* ws, fqStream_t stream, bool sub) {
  FQ_REQUIRE(h > 0 && w > 0, "fq_pwconv_i8_c16_dual: bad plane %lld x %lld", (long long)h, (long long)w);
  FQ_REQUIRE(y16 != nullptr && out_thr != nullptr && residual != nullptr && in_thr != nullptr,
             "fq_pwconv_i8_c16_dual: null pointer (y16, out_thr, residual, in_thr: the closing 1x1 of a residual unit, C16 in)");
  FQ_REQUIRE(!(out_flags & (FQ_ACT_SIGNED | FQ_ACT_LO_NEG_MAX)), "fq_pwconv_i8_c16_dual: the side tensor holds UNSIGNED codes of a "
             "[0, thr] range (the non-negative quantiser)");
  FQ_REQUIRE(cout % 32 == 0 && (cin_pad == 64 || cin_pad == 128 || cin_pad == 256 || cin_pad == 512),
             "fq_pwconv_i8_c16_dual: built for 64 / 128 / 256 / 512 input channels and Cout a multiple of 32");
  const int64_t cbi = (cin + 15) / 16, cbo = (cout + 15) / 16;
  FQ_REQUIRE((32 / (h * w) + 2) * cbi * h * w * 16 < (1ll << 31) && (32 / (h * w) + 2) * cbo * h * w * 16 < (1ll << 31),
             "fq_pwconv_i8_c16_dual: plane too large");
  FQ_REQUIRE(!sub || pw_split_sub_shape_ok(cin_pad, cout), "fq_pwconv_i8_c16_dual_sub2: more than 128 output channels");
  return pwconv_dispatch((const float*)x, wcodes, wscale, wsum, bias, y, n, cin, cin_pad, cout, h * w, 1, h, w, w, in_stat,
                         in_thr, in_width, in_flags, out_current_max, bn_scale, bn_shift, act, stat_out, residual, ws, stream,
                         true, out_thr, out_width, out_flags, nullptr, nullptr, nullptr, nullptr, y16, sub);
}

int fq_pwconv_i8_c16_dual(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                          float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                          const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                          const float* bn_scale, const float* bn_shift, int act, float* stat_out, const float* residual,
                          const float* out_thr, int out_width, unsigned out_flags, void* ws, fqStream_t stream) {
  return pwconv_dual(x, wcodes, wscale, wsum, bias, y, y16, n, cin, cin_pad, cout, h, w, in_stat, in_thr, in_width, in_flags,
                     out_current_max, bn_scale, bn_shift, act, stat_out, residual, out_thr, out_width, out_flags, ws, stream, false);
}

// fq_pwconv_i8_c16_dual with BOTH outputs subsampled as fq_pwconv_i8_sub2 stores y: y is (n, cout, ceil(h/2), ceil(w/2)) fp32 and
// y16 the C16 code tensor of that shape; statistic and residual operand over the whole planes.
int fq_pwconv_i8_c16_dual_sub2(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                               float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                               const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                               float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                               const float* residual, const float* out_thr, int out_width, unsigned out_flags, void* ws,
                               fqStream_t stream) {
  return pwconv_dual(x, wcodes, wscale, wsum, bias, y, y16, n, cin, cin_pad, cout, h, w, in_stat, in_thr, in_width, in_flags,
                     out_current_max, bn_scale, bn_shift, act, stat_out, residual, out_thr, out_width, out_flags, ws, stream, true);
}

}  // extern "C"

namespace fqi {
// 1x1 convolution of nn.Conv2D(quantized=True) (fq_qconv.hip): x quantised on load with the range record `rec`, int32 bias
// codes in the integer sum, y = act(float(sum) * (rec scale * wscale[co])) [BatchNorm folded when given]
int pw_range_call(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const int32_t* ibias,
                  float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w, int stride,
                  const float* rec, const float* bn_scale, const float* bn_shift, int act, float* stat_out, hipStream_t st,
                  bool* taken) {
  const int64_t ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
  return pwconv_dispatch(x, wcodes, wscale, wsum, reinterpret_cast<const float*>(ibias), y, n, cin, cin_pad, cout, ho * wo,
                         stride, h, w, wo, nullptr, rec, 8, 0, nullptr, bn_scale, bn_shift, act, stat_out, nullptr, nullptr,
                         (fqStream_t)st, false, nullptr, 8, 0, nullptr, nullptr, nullptr, taken);
}
}  // namespace fqi

extern "C" {

size_t fq_dense_i8_eval_workspace_bytes(int64_t n, int64_t cout) {
  return n > 0 && cout > 0 ? pw_rows_eval_ws_bytes(n, cout) : 0;
}

int fq_dense_i8_eval(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                     float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, const float* in_stat,
                     const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                     const int64_t* labels, float* counters, void* eval_ws, void* ws, fqStream_t stream) {
  FQ_REQUIRE(labels && counters && eval_ws, "fq_dense_i8_eval: null pointer (labels, counters, eval_ws)");
  FQ_REQUIRE((reinterpret_cast<uintptr_t>(eval_ws) & 7u) == 0, "fq_dense_i8_eval: eval_ws must be 8-byte aligned");
  return pwconv_dispatch(x, wcodes, wscale, wsum, bias, y, n, cin, cin_pad, cout, 1, 1, 0, 0, 0, in_stat, in_thr, in_width,
                         in_flags, out_current_max, nullptr, nullptr, FQ_ACT_NONE, nullptr, nullptr, ws, stream, false,
                         nullptr, 8, 0, (const long long*)labels, counters, eval_ws);
}

}  // extern "C"

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
