// libfakequant — fq_pwconv_i8: argument checks and the shape-based choice between the pointwise forms
// (see fq_common.h for the list of translation units and the design rules)
#include "fq_pw.h"

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

int fq_pwconv_i8(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                 float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw, const float* in_stat,
                 const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                 const float* bn_scale, const float* bn_shift, int act, float* stat_out, void* ws,
                 fqStream_t stream) {
  FQ_REQUIRE(x && wcodes && wscale && wsum && y && ws, "fq_pwconv_i8: null pointer");
  FQ_REQUIRE(n > 0 && cin > 0 && cout > 0 && hw > 0 && hw < (1ll << 30) && n * hw < (1ll << 31) - 512,
             "fq_pwconv_i8: bad shape");
  FQ_REQUIRE(cin_pad >= cin && cin_pad % 64 == 0 && cin_pad <= 8192, "fq_pwconv_i8: cin_pad=%lld must be a multiple "
             "of 64 covering cin=%lld", (long long)cin_pad, (long long)cin);
  FQ_REQUIRE((in_stat != nullptr) != (in_thr != nullptr), "fq_pwconv_i8: give in_stat (online) OR in_thr (offline): the "
             "integer path needs a quantised input");
  FQ_REQUIRE(in_stat == nullptr || out_current_max != nullptr, "fq_pwconv_i8: online mode needs out_current_max (the "
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
  c.x = x; c.wcodes = wcodes; c.wscale = wscale; c.wsum = wsum; c.bias = bias; c.y = y;
  c.n = n; c.cin = cin; c.cin_pad = cin_pad; c.cout = cout; c.hw = hw;
  c.in_stat = in_stat; c.in_thr = in_thr;
  c.levels = act_levels(in_width, in_flags);
  c.lo_neg = (in_flags & FQ_ACT_LO_NEG_MAX) ? 1 : 0;
  c.zoff = (in_flags & FQ_ACT_SIGNED) ? 0 : 128;      // unsigned codes are stored re-centred so they fit int8
  c.out_current_max = out_current_max; c.bn_scale = bn_scale; c.bn_shift = bn_shift; c.act = act;
  c.stat_out = stat_out; c.ws = ws; c.st = (hipStream_t)stream;
  static const int pw_form = env_int("FQ_PW_FORM", 0);      // 0 auto, 1 two kernels, 2 panel, 3 stream, 4 chunk, 5 tile, 6 split
  c.form = forced_form ? forced_form : pw_form;
  ProfScope prof(FQ_KERNEL_PWCONV, 4.0 * ((double)n * cin * hw + (double)n * cout * hw), c.st);
  bool taken = false;
  if (int rc = pw_try_split(c, &taken)) return rc;
  if (taken) return FQ_OK;
  if (int rc = pw_try_stream(c, &taken)) return rc;
  if (taken) return FQ_OK;
  if (int rc = pw_try_tile(c, &taken)) return rc;
  if (taken) return FQ_OK;
  if (int rc = pw_try_chunk(c, &taken)) return rc;
  if (taken) return FQ_OK;
  if (int rc = pw_try_panel(c, &taken)) return rc;
  if (taken) return FQ_OK;
  return pw_two_kernels(c);
}

}  // extern "C"
