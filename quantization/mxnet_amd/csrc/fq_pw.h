// libfakequant — internal interface between fq_pwconv_i8 (fq_pwconv.hip) and the pointwise forms, one translation unit
// each.  A form inspects the call, launches when the shape is one it takes and reports that through *taken.
#ifndef FQ_PW_H_
#define FQ_PW_H_

#include "fq_common.h"

namespace fqi {

struct PwCall {
  const float* x;
  const int8_t* wcodes;          // [rows_pad][cin_pad] int8 (+ the fragment-major copy behind it, fq_weight_codes)
  const float* wscale;
  const int32_t* wsum;
  const float* bias;
  float* y;
  int64_t n, cin, cin_pad, cout, hw;   // hw: pixels of an OUTPUT plane
  int stride;                    // 1, or 2 (split form only): the input plane is h_in x w_in, the output w_out wide
  int64_t h_in, w_in, w_out;
  const float* in_stat;          // online: per-sample statistic of x
  const float* in_thr;           // offline: threshold
  float levels;
  int lo_neg, zoff;
  float* out_current_max;
  const float* bn_scale;
  const float* bn_shift;
  int act;
  float* stat_out;
  const float* residual;         // (n, cout, hw) tensor added after BatchNorm, before the activation (split / stream forms)
  bool prezeroed;
  void* ws;
  hipStream_t st;
  int form;                      // FQ_PW_FORM: 0 auto, 1 two kernels, 3 stream, 6 split, 7 sample, 8 rows, 9 pipe
  // C16 code tensors (fq_pwconv_i8_c16): x / y are [n][ceil(C/16)][pixels][16 codes] instead of fp32 NCHW
  bool in_c16 = false;
  const float* out_thr = nullptr;   // non-null: y is a C16 tensor holding the CONSUMER's codes for this threshold
  float out_levels = 0.0f;
  int out_lo_neg = 0, out_zoff = 0;
  // fq_pwconv_i8_c16_dual: y stays fp32 and y16 receives the codes of the same values under dual_thr (out_levels / out_lo_neg /
  // out_zoff describe that quantiser); C16 input + residual operand only (the closing 1x1 of a ResNet unit)
  void* y16 = nullptr;
  const float* dual_thr = nullptr;
  // fq_pwconv_i8_sub2: stride 1, y is the dense (n, cout, ceil(h_in / 2), ceil(w_in / 2)) tensor of the output's even pixels of
  // its even rows; statistic and residual operand cover the whole h_in x w_in planes (split form only)
  bool sub = false;
  // fq_pwconv_i8_gap: y is (n, cout) - the mean of every output plane (global average pooling) - and stat_out the per-sample
  // maximum of |mean|; the convolution's own output is not stored (sample form, whole small planes only)
  bool gap = false;
  // fq_dense_i8_eval (rows form): the evaluation counters of the logits in the same launch
  const long long* eval_labels = nullptr;
  float* eval_counters = nullptr;
  void* eval_ws = nullptr;
};

// K2z  fq_pwdw.hip: the statistic-only pass (fq_pwconv_i8_stat): stat_out and out_current_max only, c.y is not touched
bool pw_stat_shape_ok(int64_t n, int64_t cin, int64_t cout, int64_t hw);
int pw_stat_launch(const PwCall& c);
// K2s  fq_pw_short.hip: the closing 1x1 of a residual unit (a) with the unit's shortcut convolution (b) in the same launch
bool pw_short_shape_ok(int64_t cin, int64_t cin2, int64_t cout);
int pw_short_launch(const PwCall& a, const PwCall& b);
int pw_try_stream(const PwCall& c, bool* taken);    // K2h  fq_pw_stream.hip
bool pw_stream_shape_ok(const PwCall& c);           //      shapes the streaming form takes
bool pw_stream_thin_takes(const PwCall& c);         //      ... and the thin instantiations (C16 input, ragged Cin / Cout) on large planes
int pw_try_split(const PwCall& c, bool* taken);     // K2m  fq_pw_split.hip
int pw_split16_launch(const PwCall& a, const void* geom, int kt, int cw, int64_t grid, size_t lds, const int8_t* wfrag,
                      bool* launched, int nw = 4);
bool pw_split_sub_shape_ok(int64_t cin_pad, int64_t cout);        // fq_pw_split_sub.hip: the subsampled-output instantiations
int pw_split_sub_launch(const PwCall& a, const void* geom, int kt, int64_t grid, size_t lds, const int8_t* wfrag, bool* launched);
int pw_try_sample(const PwCall& c, bool* taken);    // K2r  fq_pw_sample.hip (14x14 planes)
bool pw_sample_gap_shape_ok(int64_t n, int64_t cin, int64_t cout, int64_t hw, bool residual);   //      shapes of fq_pwconv_i8_gap
int pw_try_pipe(const PwCall& c, bool* taken);      // K2w  fq_pw_pipe.hip (14x14 planes, K = 256 / 512: weights resident in registers)
int pw_try_rows(const PwCall& c, bool* taken);      // K2t  fq_pw_rows.hip (planes of one pixel: the classifier)
size_t pw_rows_eval_ws_bytes(int64_t n, int64_t cout);
int pw_two_kernels(const PwCall& c);                // K2f  fq_pw_generic.hip (takes every shape)

inline int pw_zero_stat(const PwCall& c) {
  if (c.stat_out && !c.prezeroed) {
    hipError_t e = hipMemsetAsync(c.stat_out, 0, c.n * sizeof(float), c.st);
    if (e != hipSuccess) return fail(FQ_ERR_HIP, "hipMemsetAsync(stat_out): %s", hipGetErrorString(e));
  }
  return FQ_OK;
}

}  // namespace fqi

#endif  // FQ_PW_H_
