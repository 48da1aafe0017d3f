/*
 * fakequant_host.h — the SAME entry points as fakequant.h with a `_host` suffix, taking HOST pointers and running on
 * the CPU cores (C++ / OpenMP).  SURVEY.md 8(b)/8(d): "identical-signature *_host CPU entry points for the baseline
 * timing".
 *
 * These are NOT part of the product: they are implemented by oracle/libfq_host.so (built from oracle/fq_host.cpp by
 * oracle/Makefile), a C restatement of the reference's algorithm that is test infrastructure.  Only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg load it; libfakequant.so has no CPU path and
 * quantization/mxnet_amd refuses host tensors.  Two uses:
 *   - the CPU baseline of bench.py: the reference's arithmetic on all host cores (`cpu_baseline.kind = "port"`);
 *   - a fast oracle for parity checks at BASELINE's full sizes (the numpy oracle needs minutes there); it is itself
 *     pinned bit-for-bit against oracle/fq_oracle.py and the golden vectors in tests/test_host_oracle.py.
 *
 * Conventions as in fakequant.h: int status (0 = ok), `fq_last_error_host()`; `stream` is accepted and ignored; `ws`
 * may be NULL (scratch is allocated internally).  Results are bit-identical to libfakequant.so on the same inputs for
 * every function below; the arithmetic each one restates is documented at its twin in fakequant.h.
 */
#ifndef FAKEQUANT_HOST_H_
#define FAKEQUANT_HOST_H_

#include "fakequant.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* fq_last_error_host(void);
int fq_version_host(void);
/* threads OpenMP will use (after fq_set_threads_host(k), k > 0; k <= 0 restores the default = all cores) */
int fq_threads_host(void);
int fq_set_threads_host(int k);

int fq_absmax_per_sample_host(const float* x, int64_t n, int64_t inner, unsigned flags, float* out_max,
                              fqStream_t stream);
int fq_batch_mean_host(const float* v, int64_t n, float* out, fqStream_t stream);
int fq_batch_mean_rows_host(const float* v, int64_t rows, int64_t n, int64_t row_stride, float* out,
                            fqStream_t stream);
int fq_stat_rows_sum_host(const float* v, int64_t rows, int64_t n, int64_t row_stride, double* out, fqStream_t stream);
int fq_mean_from_sums_host(const double* sums, int64_t rows, float* out, fqStream_t stream);
int fq_batch_mean_gathered_host(const float* packs, int world, int64_t stride, float* out, fqStream_t stream);
int fq_fake_quant_online_host(const float* x, float* y, int64_t n, int64_t inner, int width, unsigned flags,
                              float* out_current_max, int32_t* codes, void* ws, fqStream_t stream);
int fq_fake_quant_offline_host(const float* x, float* y, int64_t n, int64_t inner, const float* threshold, int width,
                               unsigned flags, float* out_current_max, int32_t* codes, void* ws, fqStream_t stream);
int fq_fake_quant_online_prestat_host(const float* x, float* y, int64_t n, int64_t inner, const float* stat,
                                      int width, unsigned flags, float* out_current_max, int32_t* codes,
                                      fqStream_t stream);
int fq_bn_act_stat_host(const float* x, float* y, int64_t n, int64_t c, int64_t hw, const float* scale,
                        const float* shift, int act, float* stat_out, fqStream_t stream);
int fq_bn_act_maxpool_stat_host(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w,
                                const float* scale, const float* shift, int act, float* stat_out, fqStream_t stream);
int fq_add_act_stat_host(const float* a, const float* b, float* y, int64_t n, int64_t inner, int act, float* stat_out,
                         fqStream_t stream);
/* (the producers that bin what they store: by definition the pass above followed by fq_histogram_accumulate_host over y) */
int fq_bn_act_stat_hist_host(const float* x, float* y, int64_t n, int64_t c, int64_t hw, const float* scale,
                             const float* shift, int act, float* stat_out, const float* hist_max, int bins, uint64_t* hist,
                             uint32_t* neg_count, fqStream_t stream);
int fq_bn_add_act_stat_host(const float* x, const float* residual, float* y, int64_t n, int64_t c, int64_t hw,
                            const float* scale, const float* shift, int act, float* stat_out, fqStream_t stream);
int fq_bn_add_act_stat_hist_host(const float* x, const float* residual, float* y, int64_t n, int64_t c, int64_t hw,
                                 const float* scale, const float* shift, int act, float* stat_out, const float* hist_max,
                                 int bins, uint64_t* hist, uint32_t* neg_count, fqStream_t stream);
int fq_add_act_stat_hist_host(const float* a, const float* b, float* y, int64_t n, int64_t inner, int act, float* stat_out,
                              const float* hist_max, int bins, uint64_t* hist, uint32_t* neg_count, fqStream_t stream);
int fq_global_avg_pool_stat_host(const float* x, float* y, int64_t n, int64_t c, int64_t hw, int flags,
                                 float* stat_out, fqStream_t stream);
int fq_gemm_i8_codes_host(const int8_t* xcodes, const int8_t* wcodes, const int32_t* wsum, int32_t* out, int64_t n,
                          int64_t l, int64_t k_pad, int64_t cout, int zoff, fqStream_t stream);
int fq_eval_counters_host(const float* logits, const int64_t* labels, int64_t n, int64_t classes, float* counters,
                          fqStream_t stream);
int fq_stem_conv3x3s2_host(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n,
                           int64_t cin, int64_t cout, int64_t h, int64_t w, const float* bn_scale,
                           const float* bn_shift, int act, float* stat_out, fqStream_t stream);
int fq_stem_conv3x3s2_c16_host(const float* x, const float* w_tap_major, const float* bias, void* y16, int64_t n, int64_t cin,
                               int64_t cout, int64_t h, int64_t w, const float* bn_scale, const float* bn_shift, int act,
                               float* stat_out, const float* out_thr, int out_width, unsigned out_flags, fqStream_t stream);
int fq_stem_conv7x7s2_host(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n,
                           int64_t cin, int64_t cout, int64_t h, int64_t w, const float* bn_scale,
                           const float* bn_shift, int act, float* stat_out, fqStream_t stream);
int fq_stem_conv7x7s2_pool_host(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n,
                                int64_t cin, int64_t cout, int64_t h, int64_t w, const float* bn_scale,
                                const float* bn_shift, int act, float* stat_out, fqStream_t stream);
int fq_dwconv3x3_host(const float* x, const float* w, const float* bias, float* y, int64_t n, int64_t c, int64_t h,
                      int64_t wdt, int stride, const float* in_stat, const float* in_thr, int in_width,
                      unsigned in_flags, float* out_current_max, const float* bn_scale, const float* bn_shift, int act,
                      float* stat_out, fqStream_t stream);
/* codes: only the row-major half [rows_pad][row_pad] is written / read on the host (no MFMA fragment copy). */
int fq_weight_codes_host(const float* w, int64_t rows, int64_t row_len, int rows_per_scale, int width,
                         int64_t row_pad, int64_t rows_pad, int8_t* codes, float* scales, int32_t* rowsum, void* ws,
                         fqStream_t stream);
int fq_pwconv_i8_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                      const float* bias, float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw,
                      const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                      float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                      void* ws, fqStream_t stream);
int fq_pwconv_i8_strided_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                              const float* bias, float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout,
                              int64_t h, int64_t w, int stride, const float* in_stat, const float* in_thr, int in_width,
                              unsigned in_flags, float* out_current_max, const float* bn_scale, const float* bn_shift,
                              int act, float* stat_out, const float* residual, void* ws, fqStream_t stream);
int fq_pwconv_i8_shortcut_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                               float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw, const float* in_stat,
                               const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                               const float* bn_scale, const float* bn_shift, int act, float* stat_out, const float* x2,
                               const int8_t* wcodes2, const float* wscale2, const int32_t* wsum2, int64_t cin2, int64_t cin2_pad,
                               const float* in_stat2, const float* in_thr2, int in_width2, unsigned in_flags2,
                               float* out_current_max2, const float* bn_scale2, const float* bn_shift2, fqStream_t stream);
int fq_pwconv_i8_shortcut_c16_host(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                                   float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw,
                                   const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                                   float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                                   const void* x2, int x2_is_c16, const int8_t* wcodes2, const float* wscale2, const int32_t* wsum2,
                                   int64_t cin2, int64_t cin2_pad, const float* in_stat2, const float* in_thr2, int in_width2,
                                   unsigned in_flags2, float* out_current_max2, const float* bn_scale2, const float* bn_shift2,
                                   const float* out_thr, int out_width, unsigned out_flags, fqStream_t stream);
int fq_pwconv_i8_gap_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                          float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw, const float* in_stat,
                          const float* in_thr, int in_width, unsigned in_flags, float* out_current_max, const float* bn_scale,
                          const float* bn_shift, int act, float* stat_out, const float* residual, void* ws, fqStream_t stream);
int fq_pwconv_i8_sub2_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                           float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                           const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                           float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                           const float* residual, void* ws, fqStream_t stream);
/* The recompute pair: the two storing twins back to back (the tensor between the layers exists on the host). */
int fq_pwconv_i8_stat_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                           int64_t n, int64_t cin, int64_t cin_pad, int64_t cout_pad, int64_t cout, int64_t hw,
                           const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                           float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                           fqStream_t stream);
int fq_pwdw_fused_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* pw_bias,
                       int64_t n, int64_t cin, int64_t cin_pad, int64_t cout_pad, int64_t cout, int64_t h, int64_t w,
                       const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, const float* pw_bn_scale,
                       const float* pw_bn_shift, int pw_act, const float* mid_stat, const float* mid_thr, int mid_width,
                       unsigned mid_flags, float* mid_current_max, const float* dw_w, const float* dw_bias, int dw_stride,
                       const float* dw_bn_scale, const float* dw_bn_shift, int dw_act, float* y, float* stat_out,
                       fqStream_t stream);
/* fq_pwconv_i8_host on planes of one pixel, then fq_eval_counters_host on the logits it wrote (eval_ws, ws: unused). */
int fq_dense_i8_eval_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                          const float* bias, float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout,
                          const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                          float* out_current_max, const int64_t* labels, float* counters, void* eval_ws, void* ws,
                          fqStream_t stream);
/* wcodes: row-major [rows_pad][9 * cin] codes of the weights permuted to (cout, 3, 3, cin) (fq_weight_codes_host). */
int fq_conv3x3_i8_host(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                       const float* bias, float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w,
                       const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                       float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                       fqStream_t stream);
int fq_pwconv_i8_c16_host(const void* x, int x_is_c16, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                          const float* bias, void* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h,
                          int64_t w, int stride, const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                          float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                          const float* residual, const float* out_thr, int out_width, unsigned out_flags, void* ws,
                          fqStream_t stream);
int fq_pwconv_i8_c16_dual_host(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                               float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                               const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                               float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                               const float* residual, const float* out_thr, int out_width, unsigned out_flags, void* ws,
                               fqStream_t stream);
int fq_pwconv_i8_c16_dual_sub2_host(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                                    const float* bias, float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout,
                                    int64_t h, int64_t w, const float* in_stat, const float* in_thr, int in_width,
                                    unsigned in_flags, float* out_current_max, const float* bn_scale, const float* bn_shift,
                                    int act, float* stat_out, const float* residual, const float* out_thr, int out_width,
                                    unsigned out_flags, void* ws, fqStream_t stream);
int fq_conv3x3_i8_c16_host(const void* x, int x_is_c16, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                           const float* bias, void* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w,
                           const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                           float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                           const float* out_thr, int out_width, unsigned out_flags, fqStream_t stream);
int fq_dwconv3x3_c16_host(const void* x, const float* w, const float* bias, void* y, int64_t n, int64_t c, int64_t h,
                          int64_t wdt, int stride, const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                          float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                          const float* out_thr, int out_width, unsigned out_flags, fqStream_t stream);
int fq_weight_slices_host(const float* w, int64_t rows, int64_t row_len, int64_t row_pad, int64_t rows_pad, int8_t* codes,
                          float* pscale, int32_t* rowsum, void* ws, fqStream_t stream);
int fq_conv3x3_i8_sliced_host(const float* x, const int8_t* wslices, const float* pscale, const int32_t* wsum,
                              const float* bias, float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w,
                              const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                              float* out_current_max, const float* bn_scale, const float* bn_shift, int act,
                              float* stat_out, fqStream_t stream);
int fq_ste_forward_host(const float* x, float* y, int64_t rows, int64_t row_len, const float* scales, int has_clip,
                        float clip_lo, float clip_hi, float eps, fqStream_t stream);
int fq_weight_fake_quant_host(const float* w, float* w_q, int64_t rows, int64_t row_len, int width,
                              float* scales_out, void* ws, fqStream_t stream);
int fq_wino_weight_fake_quant_host(const float* w, float* w_q, int64_t cout, int64_t cin_g, int t, const float* G,
                                   const float* GI, const float* GTI, int width, float* scales_out, void* ws,
                                   fqStream_t stream);
int fq_ema_update_host(float* state, const float* current, int64_t count, double momentum, fqStream_t stream);
int fq_global_max_host(const float* x, int64_t numel, float* out, fqStream_t stream);
int fq_histogram_accumulate_host(const float* x, int64_t numel, const float* max_dev, int bins, uint64_t* hist,
                                 uint32_t* neg_count, fqStream_t stream);
int fq_hist_to_float_host(const uint64_t* hist, float* out, int64_t count, fqStream_t stream);
int fq_kl_search_host(const float* hist, int64_t L, int bins, int levels, int min_bins, int32_t* out_best, void* ws,
                      fqStream_t stream);
int fq_quantize_codes_host(const float* x, int32_t* codes, int64_t numel, int mode, float* range_dev, void* ws,
                           fqStream_t stream);
int fq_dequantize_host(const int32_t* codes, float* y, int64_t numel, const float* scale_dev, fqStream_t stream);

int fq_qconv_weights_prepare_host(const float* w, int64_t cin, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw,
                                  int groups, int weight_mode, float w_min, float w_max, void* wbuf, void* ws,
                                  fqStream_t stream);
int fq_qconv2d_forward_host(const float* x, const float* w, const void* wbuf, const float* bias, float* y, int64_t n,
                            int64_t cin, int64_t h, int64_t wdt, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw,
                            int groups, int input_mode, float in_min, float in_max, const float* in_stat, int act,
                            const float* bn_scale, const float* bn_shift, float* stat_out, void* ws, int force_direct,
                            fqStream_t stream);

/* The reference's UNFUSED op chain for one activation tensor, pass by pass, as MXNet's CPU NDArray ops would run it
 * (convert_conv2d.py:56-66, ste_func.py:41): abs (temp) -> per-sample max -> mean -> clip (temp) -> divide (temp) ->
 * round (temp) -> multiply; each pass an OpenMP loop over the whole tensor.  Same results as
 * fq_fake_quant_online_host; 44 B/elem of memory traffic instead of 12.  tmp: 2 * n * inner floats (or NULL).      */
int fq_unfused_chain_host(const float* x, float* y, int64_t n, int64_t inner, int width, unsigned flags,
                          float* out_current_max, float* tmp, fqStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FAKEQUANT_HOST_H_ */
