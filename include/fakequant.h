/*
 * fakequant.h — C ABI of libfakequant.so: the MI355X (gfx950) simulated-quantisation hot path of
 * hey-yahei/Quantization.MXNet, as a drop-in for the bodies of the reference's patched `hybrid_forward`s.
 *
 * The reference has no native boundary on this path: every op is a Python call into Apache MXNet NDArray ops
 * (SURVEY.md F1/F2).  The only FFI it shows is the libmxnet convention at quantize/freeze/freeze.py:67-76,125
 * (`check_call(_LIB.MXQuantizeSymbol(...))`: int status, out-params, error text fetched separately).  This header
 * keeps that convention, and each entry point names the reference lines whose op chain it replaces.  The ctypes
 * binding a reference maintainer would add is shown in INTEGRATION.md and lives in quantization/mxnet_amd/_lib.py.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; `fq_last_error()` returns the thread-local message;
 *   - all tensor pointers are DEVICE pointers owned by the caller (its tensor library); fp32, C-contiguous (NCHW);
 *   - the library never allocates, frees or synchronises: scratch comes from the caller (`*_workspace_bytes`),
 *     every launch is asynchronous on the `hipStream_t` passed as `stream` (void* here so C callers need no HIP
 *     headers); results the reference pulled to the host with `.asscalar()` stay in device scalars;
 *   - no CPU fallback exists in this library.
 */
#ifndef FAKEQUANT_H_
#define FAKEQUANT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQ_OK 0
#define FQ_ERR_INVALID 1
#define FQ_ERR_HIP 2

typedef void* fqStream_t; /* hipStream_t */

/* flags for the activation entry points */
#define FQ_ACT_SIGNED 1u       /* levels = 2^(w-1)-1 instead of 2^w-1        (convert_conv2d.py:59-64)            */
#define FQ_ACT_LO_NEG_MAX 2u   /* clip to [-max, max] instead of [0, max]    (conv signed :60; Dense never :49)   */
#define FQ_ACT_NO_ABS 4u       /* statistic = max(x) not max|x|              (convert_act.py:50)                  */
#define FQ_ACT_NO_EPS 8u       /* divide by scale, not scale + 1e-10         (convert_act.py:54)                  */

/* quantize_codes modes (nn/quantized_conv.py:63-72,112-127) */
#define FQ_CODES_INT8 0   /* range = [-max|x|, max|x|]                      */
#define FQ_CODES_UINT8 1  /* range = [min x, max x]                         */
#define FQ_CODES_RANGE 2  /* range given in range_dev[0..1]                 */
#define FQ_CODES_SCALE 3  /* range in range_dev[0..1], scale in range_dev[2] (the int32 bias, :122-127) */

const char* fq_last_error(void);
int fq_version(void);
/* sha1 (40 hex digits) of the sources and flags this library was built from: csrc/build.py compares it with the tree to
 * decide whether a built file is current (content, not modification times). */
const char* fq_build_id(void);
/* 1 when the library carries the named optional part, else 0.  "pipe": the pipe form of fq_pwconv_i8 (FQ_PW_FORM(9)), built,
 * measured, slower than the sample form and therefore left out of the shipped library (csrc/build.py --dev builds it). */
int fq_build_has(const char* feature);
/* Name (e.g. "gfx950"), compute units and wavefront size of the current HIP device. */
int fq_device_info(char* arch, int arch_len, int* compute_units, int* wavefront);

/* ---- optional per-kernel timing (bench.py's roofline leg) ------------------------------------------------------
 * When enabled, every launch of the streaming kernels is bracketed by a pair of HIP events recorded on the launch
 * stream.  fq_profile_read() waits for the recorded events (the ONLY call in this library that synchronises),
 * adds up durations / launches / algorithmic bytes per kernel id since the last reset, and releases the events.
 * Do not enable during hipGraph capture.                                                                          */
#define FQ_KERNEL_STAT 0          /* absmax_per_sample_kernel      : 4 B/elem (read x)                              */
#define FQ_KERNEL_APPLY_ONLINE 1  /* act_apply_kernel<ONLINE>      : 8 B/elem (read x, write y)                     */
#define FQ_KERNEL_APPLY_OFFLINE 2 /* act_apply_kernel<!ONLINE>     : 8 B/elem                                       */
#define FQ_KERNEL_WEIGHT 3        /* weight kernels                : 8 B/elem                                       */
#define FQ_KERNEL_HISTOGRAM 4     /* histogram_kernel              : 4 B/elem                                       */
#define FQ_KERNEL_BN_ACT 5        /* bn_act_stat_kernel            : 8 B/elem                                       */
#define FQ_KERNEL_DWCONV 6        /* dwconv3x3_*_kernel            : 4 B/in elem + 4 B/out elem                     */
#define FQ_KERNEL_PWCONV 7        /* pwconv_{stream,sample,split}_kernel (32x32x32 int8 MFMA; quant_transpose_i8 + pwconv_i8
                                     for the remaining shapes)     : 4 B/in elem + 4 B/out elem                     */
#define FQ_KERNEL_STEM 8          /* stem_mfma_kernel              : 4 B/in elem + 4 B/out elem                     */
#define FQ_KERNEL_POOL 9          /* gap_stat_kernel               : 4 B/in elem + 4 B/out elem                     */
#define FQ_KERNEL_GLOBAL_MAX 10   /* minmax_kernel (calibration)   : 4 B/elem                                       */
#define FQ_KERNEL_CONV3X3 11      /* conv3x3_i8_kernel             : 4 B/in elem + 4 B/out elem                     */
#define FQ_KERNEL_DENSE 12        /* pwconv_rows_kernel (the classifier on the codes: planes of one pixel; latency-bound)  */
#define FQ_KERNEL_COUNT 13
int fq_profile_enable(int on);
int fq_profile_reset(void);
int fq_profile_read(int kernel_id, double* total_ms, int64_t* launches, double* total_bytes);
/* The bytes the recorded launches of `kernel_id` really moved: equal to fq_profile_read's algorithmic bytes (4 B per input and
 * per output element) except where a side of a launch was a C16 code tensor - 1 B per element there (fq_pwconv_i8_c16,
 * fq_conv3x3_i8_c16, fq_dwconv3x3_c16, fq_stem_conv3x3s2_c16).  A roofline fraction of a code-hand-over run is
 * moved / time / peak; algorithmic / time / peak can exceed 1 there (it measures what the hand-over SAVES). */
int fq_profile_read_moved(int kernel_id, double* total_moved_bytes);
/* The fixed cost an event pair adds to a bracketed launch, measured on `stream`:
 *   pair_ms        <- median elapsed time of an event pair around ONE one-element fill kernel (`repeats` samples);
 *   null_kernel_ms <- elapsed time of `repeats` back-to-back launches of that kernel inside ONE event pair / repeats, i.e.
 *                     what the kernel itself (dispatch included) costs when nothing brackets it.
 * pair_ms - null_kernel_ms is what bench.py removes from every bracketed launch.  `scratch`: >= 4 device bytes.          */
int fq_profile_calibrate(void* scratch, int repeats, double* pair_ms, double* null_kernel_ms, fqStream_t stream);
/* What bracketing a launch with an event pair adds to what the launch costs inside a stream of back-to-back launches, measured
 * on a kernel long enough (a one-wavefront spin of `spin_us` on the wall clock) that dispatch cannot hide behind it as it does
 * behind fq_profile_calibrate's one-element kernel:  overhead_ms <- median pair time of `repeats` bracketed launches - (the
 * same launches inside ONE pair) / repeats;  spin_ms <- the kernel's own median clock distance.  bench.py removes overhead_ms
 * per launch from the raw event time; tools/check_events_vs_rocprof.py holds the result against rocprofv3's kernel table.
 * Synchronises.  scratch: repeats * 16 device bytes.                                                                       */
int fq_profile_launch_overhead(void* scratch, int repeats, double spin_us, double* overhead_ms, double* spin_ms,
                               fqStream_t stream);

/* ---- activations ---------------------------------------------------------------------------------------------
 * x is (n, inner) = (N, C*H*W).  `ws` is a caller workspace of fq_act_workspace_bytes(n) bytes.                  */
size_t fq_act_workspace_bytes(int64_t n);

/* Replaces `F.max(F.abs(x), axis=(1,2,3))` (convert_conv2d.py:56; convert_dense.py:41; convert_act.py:50 with
 * FQ_ACT_NO_ABS).  out_max[n] <- per-sample statistic.                                                          */
int fq_absmax_per_sample(const float* x, int64_t n, int64_t inner, unsigned flags, float* out_max,
                         fqStream_t stream);

/* Replaces `.mean()` of the per-sample maxima (convert_conv2d.py:56): out[0] <- fp32(sum_fp64(v[0..n)))/n.      */
int fq_batch_mean(const float* v, int64_t n, float* out, fqStream_t stream);
/* Row-wise form for L layers at once (the multi-GPU calibration step): out[r] <- batch mean of v[r*row_stride ..+n). */
int fq_batch_mean_rows(const float* v, int64_t rows, int64_t n, int64_t row_stride, float* out, fqStream_t stream);

/* The calibration-step collective of a sharded batch (dist.py; reference arithmetic: the `.mean()` of convert_conv2d.py:56
 * feeding `_update_ema`, convert.py:66-70, taken over the GLOBAL batch):
 *   fq_stat_rows_sum : out[r] <- sum_fp64(v[r*row_stride .. +n)) in sample order for r < rows, out[rows] <- n (as a double):
 *                      one record of rows+1 doubles per rank; the ranks' records are then summed by ONE all-reduce;
 *   fq_mean_from_sums: out[r] <- fp32(sums[r]) / fp32(sums[rows]) — the batch mean of the global batch for every layer.
 * n may be 0 (a rank without a batch in this step contributes an empty record).                                        */
int fq_stat_rows_sum(const float* v, int64_t rows, int64_t n, int64_t row_stride, double* out, fqStream_t stream);
int fq_mean_from_sums(const double* sums, int64_t rows, float* out, fqStream_t stream);

/* Sharded-batch form (dist.py): `packs` holds `world` records of `stride` floats, record w = {n_w, v_w[0..n_w)} as
 * all-gathered from the ranks; out[0] <- the batch mean over the concatenation v_0 | v_1 | ... (global sample order),
 * same ordered fp64 accumulation.  No host round trip for the (possibly ragged) counts.                            */
int fq_batch_mean_gathered(const float* packs, int world, int64_t stride, float* out, fqStream_t stream);

/* Replaces convert_conv2d.py:56-66 + ste_func.py:41 in ONLINE mode (threshold = this batch's statistic):
 *   cur = mean_n max|x[n]|;  scale = cur/levels;  y = roundf(clip(x, lo, cur) / (scale + eps)) * scale.
 * out_current_max (device scalar, may be NULL) <- cur.  codes (int32, may be NULL) <- the integer stage.
 * Two launches (statistic pass, apply pass) + one memset on `stream`; algorithmic traffic 12 B/elem.            */
int fq_fake_quant_online(const float* x, float* y, int64_t n, int64_t inner, int width, unsigned flags,
                         float* out_current_max, int32_t* codes, void* ws, fqStream_t stream);

/* Same arithmetic in OFFLINE mode: threshold read from the device scalar `threshold` (`input_max`,
 * convert_conv2d.py:58).  If out_current_max != NULL the batch statistic is ALSO produced (the reference computes
 * it in every mode, :56) — fused into the same pass over x, so traffic stays 8 B/elem.                          */
int fq_fake_quant_offline(const float* x, float* y, int64_t n, int64_t inner, const float* threshold, int width,
                          unsigned flags, float* out_current_max, int32_t* codes, void* ws, fqStream_t stream);

/* ONLINE mode when the per-sample statistic of x is ALREADY known (its producer computed it, see fq_bn_act_stat):
 * only the apply pass runs (8 B/elem instead of 12).  stat[n] = max|x[n]| exactly as fq_absmax_per_sample gives.   */
int fq_fake_quant_online_prestat(const float* x, float* y, int64_t n, int64_t inner, const float* stat, int width,
                                 unsigned flags, float* out_current_max, int32_t* codes, fqStream_t stream);

/* ---- producer fusion (inference) -----------------------------------------------------------------------------------
 * What sits between two quantised convolutions in the reference's nets is BatchNorm (inference) + ReLU as separate
 * Gluon blocks, i.e. two more full passes over the tensor before the next layer's statistic pass.  This entry point
 * does all three in one pass: x is (n, c, hw);  y = act(x * scale[c] + shift[c])  with separately rounded multiply and
 * add, act: 0 none, 1 relu, 2 relu6 (clip to [0,6]);  stat_out[n] (may be NULL) <- max|y[n]| for the consumer's
 * fq_fake_quant_online_prestat.  scale = gamma/sqrt(var+eps), shift = beta - mean*scale are prepared by the caller.  */
#define FQ_ACT_NONE 0
#define FQ_ACT_RELU 1
#define FQ_ACT_RELU6 2
/* OR into `act`: stat_out is already zero (the caller zeroes ONE arena per forward instead of one memset per layer) */
#define FQ_STAT_PREZEROED 0x100
int fq_bn_act_stat(const float* x, float* y, int64_t n, int64_t c, int64_t hw, const float* scale,
                   const float* shift, int act, float* stat_out, fqStream_t stream);
/* The same followed by a 3x3 / stride 2 / padding 1 max pooling, in one pass (the head of the ImageNet ResNets behind
 * their first convolution: BatchNorm -> ReLU -> MaxPool2D(3, 2, 1)):  y[n, c, ho, wo] = max over the window of
 * act(x * scale[c] + shift[c]), padding never wins;  y: (n, c, (h-1)/2+1, w/2);  stat_out[n] <- max|y[n]|.  w % 4 == 0.
 * 4 B/in elem + 4 B/out elem instead of 8 + 5 (+ the consumer's statistic pass).                                    */
int fq_bn_act_maxpool_stat(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w, const float* scale,
                           const float* shift, int act, float* stat_out, fqStream_t stream);

/* The residual tail of a ResNet unit — `(x + residual).relu()` in the gluon model zoo's BasicBlockV1 / BottleneckV1, two
 * more elementwise passes followed by one statistic pass per quantised consumer — in one pass: y = act(a + b) over
 * (n, inner) with stat_out[n] (may be NULL) <- max|y[n]|.  act / FQ_STAT_PREZEROED as in fq_bn_act_stat.  12 B/elem.     */
int fq_add_act_stat(const float* a, const float* b, float* y, int64_t n, int64_t inner, int act, float* stat_out,
                    fqStream_t stream);

/* The two producers above while the KL calibration collects feature maps (/root/reference/quantize/distribution_calibrate.py:
 * 91-106: every quantised block's input is histogrammed once per batch, its range fixed by the first batch, :97-101): the
 * pass that stores y also adds y's histogram to `hist` - what fq_histogram_accumulate(y, n * inner, hist_max, bins, hist,
 * neg_count) would add in a pass of its own (4 B/elem read back), same binning, same counts.  stat_out is required here;
 * bins <= 4096.  The first batch of a collection still takes the separate passes (fq_global_max must see the whole tensor
 * before anything can be binned).                                                                                        */
int fq_bn_act_stat_hist(const float* x, float* y, int64_t n, int64_t c, int64_t hw, const float* scale,
                        const float* shift, int act, float* stat_out, const float* hist_max, int bins, uint64_t* hist,
                        uint32_t* neg_count, fqStream_t stream);
int fq_add_act_stat_hist(const float* a, const float* b, float* y, int64_t n, int64_t inner, int act, float* stat_out,
                         const float* hist_max, int bins, uint64_t* hist, uint32_t* neg_count, fqStream_t stream);

/* BatchNorm + residual add + activation in ONE pass (round 6): y = act(fl(fl(x * scale[c]) + shift[c]) + residual), i.e. the
 * values of fq_bn_act_stat (FQ_ACT_NONE) followed by fq_add_act_stat, 12 B/elem instead of 8 + 12 - the closing BatchNorm of a
 * ResNet unit, `(body(x) + shortcut).relu()`, while its convolution does not run on the integer codes (quantisation switched
 * off: the collection forward of the KL calibration, /root/reference/quantize/distribution_calibrate.py:78-106 via
 * examples/simulate_quantization.py:298, or an fp32 evaluation).  stat_out is required; the _hist form bins what it stores. */
int fq_bn_add_act_stat(const float* x, const float* residual, float* y, int64_t n, int64_t c, int64_t hw, const float* scale,
                       const float* shift, int act, float* stat_out, fqStream_t stream);
int fq_bn_add_act_stat_hist(const float* x, const float* residual, float* y, int64_t n, int64_t c, int64_t hw,
                            const float* scale, const float* shift, int act, float* stat_out, const float* hist_max, int bins,
                            uint64_t* hist, uint32_t* neg_count, fqStream_t stream);

/* Global average pooling (gluon GlobalAvgPool2D -> F.Pooling(global_pool=True, pool_type='avg'), the block in front of
 * the classifier of every model of the zoo) with the per-sample statistic the following quantised Dense needs
 * (convert_dense.py:40-41): x (n, c, hw) -> y (n, c) = fp32(sum over hw accumulated in fp64, in order) / fp32(hw);
 * stat_out[n] (may be NULL) <- max_c |y[n][c]|.  flags: 0 or FQ_STAT_PREZEROED.                                       */
int fq_global_avg_pool_stat(const float* x, float* y, int64_t n, int64_t c, int64_t hw, int flags, float* stat_out,
                            fqStream_t stream);

/* The integer core of the reference's stand-alone quantised convolution (nn/quantized_conv.py:134-151: every im2col slice
 * times the reshaped filters, accumulated as integers) on the int8 matrix cores, exact in int32:
 *   out[n][co][p] = sum_k xcodes[n*l + p][k] * wcodes[co][k] + zoff * wsum[co]
 * xcodes: (n*l rows rounded up to 32) x k_pad int8, the im2col rows, k contiguous, zero padded to k_pad % 32 == 0;
 *         unsigned codes are passed re-centred (code - 128) with zoff = 128, signed codes as they are with zoff = 0;
 * wcodes: (cout rounded up to 32) x k_pad int8, zero padded;  wsum[co] = sum_k wcodes[co][k];
 * out   : (n, cout, l) int32.  Bias, activation and the dequantisation by in_scale*w_scale stay with the caller
 *         (fq_dequantize), as in the reference (:146-158).                                                          */
int fq_gemm_i8_codes(const int8_t* xcodes, const int8_t* wcodes, const int32_t* wsum, int32_t* out, int64_t n,
                     int64_t l, int64_t k_pad, int64_t cout, int zoff, fqStream_t stream);

/* The evaluation counters of the reference CLI (examples/simulate_quantization.py:122-148: pred = argmax(outputs, axis=1),
 * test_num_correct += (pred == y), label_counter[gt] += 1, correct_counter[gt] += (pred == gt)) accumulated on the device
 * in one launch.  logits: (n, classes) fp32;  labels: (n) int64 (labels outside [0, classes) only count in `total`);
 * counters (2 + 2*classes floats, accumulated into, the caller zeroes them once per evaluation):
 *   [0] n_correct, [1] total, [2 .. 2+classes) correct_counter, [2+classes .. 2+2*classes) label_counter.
 * argmax takes the FIRST index among equal maxima and treats NaN as the maximum.  Exact while every count < 2^24.     */
int fq_eval_counters(const float* logits, const int64_t* labels, int64_t n, int64_t classes, float* counters,
                     fqStream_t stream);

/* The first ("stem") convolution of the ImageNet nets, which the reference leaves un-quantised (simulate_quantization.py
 * excludes the first convolution; it runs as F.Convolution, mxnet gluon/nn/conv_layers.py, followed by separate
 * BatchNorm / Activation blocks and then the next layer's statistic pass): dense 3x3, stride 2, pad 1, fp32, with
 * BatchNorm / activation folded into the store and the per-sample statistic of the result.
 *   acc = sum over ci, ky, kx (in that order) of fmaf(w[co][ci][ky][kx], x[..], acc), zero padding, + bias[co] if given;
 *   y   = act(acc * bn_scale[co] + bn_shift[co]) if bn_scale given else act(acc);   stat_out[n] (may be NULL) <- max|y[n]|
 * x: (n, cin, h, w);  y: (n, cout, ho, wo) with ho = (h - 1) / 2 + 1.  The weights are passed TAP-MAJOR,
 * w_tap_major[ci][ky][kx][co] (the caller permutes the (cout, cin, 3, 3) parameter once): the cout weights of a tap are
 * then one contiguous, wave-uniform block.  Built for cin = 3, cout = 32 (MobileNet v1 / v2); other shapes: FQ_INVALID. */
int fq_stem_conv3x3s2(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n, int64_t cin,
                      int64_t cout, int64_t h, int64_t w, const float* bn_scale, const float* bn_shift, int act,
                      float* stat_out, fqStream_t stream);
/* The same for the first convolution of the ImageNet ResNets: 7x7, stride 2, padding 3, 3 -> 64 channels
 * (w_tap_major: [3][7][7][64]).  Both run on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 accumulates as an fmaf chain in
 * ascending k, so the results are those of the fmaf chain over (ci, ky, kx) bit for bit).                              */
/* The 3x3 first convolution with a C16 code tensor as output (round 4): y16 (n, cout, ho, wo) holds the codes of
 * act(BN(conv)) under out_thr / out_width / out_flags - the stored threshold of the layer's single consumer (MobileNetV2's
 * first 1x1 under offline input quantisation), exactly what quantising fq_stem_conv3x3s2's fp32 output gives; stat_out is the
 * statistic of the fp32 values.  1 byte per element written instead of 4, and read by the consumer instead of 4.          */
int fq_stem_conv3x3s2_c16(const float* x, const float* w_tap_major, const float* bias, void* y16, int64_t n, int64_t cin,
                          int64_t cout, int64_t h, int64_t w, const float* bn_scale, const float* bn_shift, int act,
                          float* stat_out, const float* out_thr, int out_width, unsigned out_flags, fqStream_t stream);
int fq_stem_conv7x7s2(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n, int64_t cin,
                      int64_t cout, int64_t h, int64_t w, const float* bn_scale, const float* bn_shift, int act,
                      float* stat_out, fqStream_t stream);
/* The head of the ImageNet ResNets in ONE launch (round 5): fq_stem_conv7x7s2 followed by MaxPool2D(3, stride 2, pad 1) -
 * gluoncv's resnet*_v1 `features[0..3]` (Conv2D, BatchNorm, Activation, MaxPool2D), which examples/simulate_quantization.py
 * leaves un-quantised (--exclude-first-conv).  y: (n, 64, hp, wp) with hp = (ho - 1) / 2 + 1; y = max over the 3x3 window
 * (padding never wins) of act(BN(conv)): the values fq_stem_conv7x7s2 + fq_bn_act_maxpool_stat (identity BatchNorm) give,
 * bit for bit; stat_out[n] (may be NULL) <- max|y[n]| of the POOLED tensor.  The convolution output stays in LDS: 77 + 103 MB
 * of traffic at batch 128 instead of 77 + 411 + 411 + 103.  fq_stem_conv7x7s2_pool_supported(h, w): 1 when a row of the
 * convolution output cuts into four tiles of at most 32 columns and three rows of it fit LDS (224 x 224: yes). */
int fq_stem_conv7x7s2_pool_supported(int64_t h, int64_t w);
int fq_stem_conv7x7s2_pool(const float* x, const float* w_tap_major, const float* bias, float* y, int64_t n, int64_t cin,
                           int64_t cout, int64_t h, int64_t w, const float* bn_scale, const float* bn_shift, int act,
                           float* stat_out, fqStream_t stream);

/* Depthwise 3x3 convolution (pad 1, dilation 1, stride 1|2, one filter per channel) with the fake-quant of its INPUT
 * folded into the load and BatchNorm/activation/statistic folded into the store — per depthwise layer x is read once
 * and y written once (the reference: fake-quant 4 passes, F.Convolution, BatchNorm, ReLU, next layer's statistic).
 *   xq   = in_thr/in_stat given ? roundf(clip(x, lo, max_)/(max_/levels + eps)) * (max_/levels) : x
 *          with max_ = in_thr[0] (offline / pre-reduced) when in_thr is given, else the batch mean of in_stat[0..n)
 *          (online) — bit-identical to fq_fake_quant_offline / _online_prestat.  BOTH may be given: in_thr quantises and
 *          the batch mean of in_stat is only written to out_current_max (the reference computes `current_input_max` in
 *          every mode, convert_conv2d.py:56) — the same holds for fq_pwconv_i8(_strided) and fq_conv3x3_i8;
 *   acc  = sum over ky,kx (row-major) of fmaf(w[c][ky][kx], xq[..], acc), zero padding, + bias[c] if given;
 *   y    = act(acc * bn_scale[c] + bn_shift[c]) if bn_scale given else act(acc);
 *   stat_out[n] (may be NULL) <- max|y[n]|;  out_current_max (may be NULL) <- batch mean of in_stat when in_stat is given.
 * x: (n, c, h, w);  w: (c, 1, 3, 3);  y: (n, c, ho, wo), ho = (h - 1) / stride + 1.                                 */
int fq_dwconv3x3(const float* x, const float* w, const float* bias, float* y, int64_t n, int64_t c, int64_t h,
                 int64_t wdt, int stride, const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                 float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                 fqStream_t stream);

/* ---- pointwise (1x1) convolution on the integer codes -----------------------------------------------------------------
 * After fake-quantisation both operands of a convolution are integers times a scale: x_q = cx * sx (cx in [0, 2^w-1]
 * or [-(2^(w-1)-1), 2^(w-1)-1]) and w_q[co,:] = cw * sw[co].  For a 1x1 convolution the reference's fp32
 * F.Convolution computes  sum_ci w_q * x_q  with fp32 rounding at every step; the SAME sum is  sx*sw[co] * sum_ci cw*cx,
 * and the integer sum is exact on the int8 matrix cores (int32 accumulate; v_mfma_i32_32x32x32_i8 in the stream / sample /
 * split forms that take the MobileNet / MobileNetV2 / ResNet shapes, v_mfma_i32_16x16x64_i8 in the generic form).  This entry point
 * does that, with the fake-quant of x folded into the load and BatchNorm / activation / per-sample statistic into the
 * store (same contract as fq_dwconv3x3):
 *   cx   = roundf(clip(x, lo, max_) / (max_/levels + eps))          bit-identical to the fake-quant kernels
 *   y    = act( (float(sum_ci cw[co,ci]*cx[ci]) * (sx * wscale[co]) + bias[co]) * bn_scale[co] + bn_shift[co] )
 * x: (n, cin, hw) fp32;  y: (n, cout, hw) fp32;  wcodes: the buffer written by fq_weight_codes: int8 [cout_pad][cin_pad]
 * row-major (zero padded; cin_pad % 64 == 0, cout_pad % 64 == 0) FOLLOWED by the same codes in MFMA-fragment order
 * (another cout_pad*cin_pad bytes: fragment (co/32, ci/32) = 64 lanes x 16 bytes, lane = co%32 + 32*((ci%32)/16), byte =
 * ci%16), i.e. `codes` must hold 2*rows_pad*row_pad bytes;  wscale[cout], wsum[cout] (= sum_ci cw, used for the +128 re-centring of
 * unsigned activations).  Needs in_width <= 8.                                                                       */
int fq_weight_codes(const float* w, int64_t rows, int64_t row_len, int rows_per_scale, int width, int64_t row_pad,
                    int64_t rows_pad, int8_t* codes, float* scales, int32_t* rowsum, void* ws, fqStream_t stream);
/* One launch for the shapes of the stream, sample and split forms (csrc/fq_pw_stream.hip, fq_pw_sample.hip,
 * fq_pw_split.hip; chosen by shape, FQ_PW_FORM forces one for tuning); every other shape takes two: (A) quantise + transpose x into int8 codes [(n*hw)][cin_pad] in
 * `ws` (fq_pwconv_workspace_bytes), (B) the integer GEMM with both operands K-contiguous + epilogue.  Online mode
 * requires out_current_max.                                                                                          */
/* Planes of ONE pixel (hw == 1: a Dense layer behind global pooling, convert_dense.py:37-70) take the rows form
 * (csrc/fq_pw_rows.hip): x is (n, cin) row-major, y (n, cout); needs cin % 4 == 0.
 * OR into `act`: take this form instead of the shape-based choice (1 two kernels, 3 stream, 6 split, 7 sample, 8 rows);
 * FQ_INVALID when the shape does not fit it.  For parity tests and tuning runs.                                  */
#define FQ_PW_FORM(f) ((f) << 12)
size_t fq_pwconv_workspace_bytes(int64_t n, int64_t cin_pad, int64_t hw);
int fq_pwconv_i8(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                 float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw, const float* in_stat,
                 const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                 const float* bn_scale, const float* bn_shift, int act, float* stat_out, void* ws,
                 fqStream_t stream);

/* The quantised classifier AND the evaluation counters of its logits in one launch - the tail of an evaluation step of the
 * reference CLI: `outputs = net(X)` ends in the converted Dense (convert_dense.py:37-70: input fake-quant with the [0, max]
 * clip, weight fake-quant, FullyConnected), then simulate_quantization.py:122-148 counts.  Same values as fq_pwconv_i8 with
 * hw = 1 (y: (n, cout) logits, written as well) followed by fq_eval_counters(y, labels, n, cout, counters) - the argmax rule,
 * the label range rule and the counter layout are fq_eval_counters'.  `eval_ws`: fq_dense_i8_eval_workspace_bytes(n, cout)
 * bytes, 8-byte aligned, ZEROED ONCE by the caller before the first call; every call leaves it zeroed where it must be.  Not
 * to be shared by calls that may run concurrently on different streams.  Needs cin % 4 == 0.                          */
size_t fq_dense_i8_eval_workspace_bytes(int64_t n, int64_t cout);
int fq_dense_i8_eval(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                     float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, const float* in_stat,
                     const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                     const int64_t* labels, float* counters, void* eval_ws, void* ws, fqStream_t stream);

/* The same for a STRIDED 1x1 convolution (stride 2 in both directions, no padding: the shortcut and first convolutions
 * of the ResNet stages): x is (n, cin, h, w), y is (n, cout, ceil(h/2), ceil(w/2)) - the kernel reads every second pixel of
 * every second row, nothing is subsampled beforehand.  stride = 1 is fq_pwconv_i8 with hw = h * w.  Strided calls need a
 * shape the split form takes (cin_pad / 32 in {2, 4, 6, 8, 10, 12, 16, 18, 30, 32, 64}), FQ_ERR_INVALID otherwise.
 * `residual` (may be NULL; y's shape): the shortcut of a residual unit, added after BatchNorm and before the activation,
 *   y = act((acc * sx * sw + bias) * bn_scale + bn_shift + residual)
 * - the `(body(x) + shortcut).relu()` tail of the ResNet units and the `out + x` of MobileNetV2's linear bottlenecks in the
 * epilogue of the unit's last 1x1 convolution (its output is then never written and read back: 8 instead of 16 bytes per
 * element).  Needs a shape the split or the streaming form takes.                                                   */
int fq_pwconv_i8_strided(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                         const float* bias, float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h,
                         int64_t w, int stride, const float* in_stat, const float* in_thr, int in_width,
                         unsigned in_flags, float* out_current_max, const float* bn_scale, const float* bn_shift,
                         int act, float* stat_out, const float* residual, void* ws, fqStream_t stream);

/* The closing 1x1 convolution of the last unit of a ResNet-v1 stage (round 6).  Its output - `(body(x) + shortcut).relu()`, the
 * trunk - has exactly two readers, the first 1x1 and the shortcut 1x1 of the next stage's first unit, both with stride 2 and no
 * padding (gluon model_zoo BottleneckV1; the reference wraps each in convert_conv2d.py:53-66,108): three quarters of the tensor
 * are written and never read, and the quarter that is read is fetched with half of every cache line wasted.  This entry point
 * computes the values of fq_pwconv_i8_strided(stride = 1, residual) and stores y[:, :, ::2, ::2] only, as a dense
 * (n, cout, ceil(h/2), ceil(w/2)) tensor; stat_out[n] is max|y[n]| over ALL of y (the readers' online thresholds are those of
 * the whole tensor, convert_conv2d.py:56-58), `residual` has the whole (n, cout, h, w) shape.  The readers then call
 * fq_pwconv_i8 (stride 1) on that tensor with this statistic.  Shapes: fq_pwconv_i8_sub2_supported(cin, cout).          */
int fq_pwconv_i8_sub2_supported(int64_t cin, int64_t cout);
int fq_pwconv_i8_sub2(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                      float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                      const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                      const float* bn_scale, const float* bn_shift, int act, float* stat_out, const float* residual, void* ws,
                      fqStream_t stream);

/* The closing 1x1 convolution of a residual unit and the unit's shortcut convolution in ONE launch (round 6).  gluon's BottleneckV1
 * with a `downsample` branch - the first unit of every stage of the v1 ResNets - computes
 *   act(BatchNorm3(conv3(x)) + BatchNormD(convD(x2)))            (conv3: the body's last 1x1; convD: `downsample[0]`, a 1x1 of the
 * unit's input; each wrapped by convert_conv2d.py:53-66,108): the shortcut tensor is as large as the result, has no other reader and
 * feeds no quantised block.  This entry point computes the sum without it - both integer convolutions per (pixel tile, channel
 * group), the same fp32 operations in the same order as fq_pwconv_i8_strided(x2 ..., no activation) followed by
 * fq_pwconv_i8_strided(x ..., residual = that): bit-equal.  x: (n, cin, hw), x2: (n, cin2, hw) fp32 (a strided shortcut reads the
 * subsampled trunk of fq_pwconv_i8_sub2); `..2` arguments are the shortcut convolution's (no bias; its BatchNorm is mandatory);
 * stat_out[n] = max|y[n]|.  Shapes: fq_pwconv_i8_shortcut_supported(cin, cin2, cout).                                            */
int fq_pwconv_i8_shortcut_supported(int64_t cin, int64_t cin2, int64_t cout);
int fq_pwconv_i8_shortcut(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                          float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw, const float* in_stat,
                          const float* in_thr, int in_width, unsigned in_flags, float* out_current_max, const float* bn_scale,
                          const float* bn_shift, int act, float* stat_out, const float* x2, const int8_t* wcodes2,
                          const float* wscale2, const int32_t* wsum2, int64_t cin2, int64_t cin2_pad, const float* in_stat2,
                          const float* in_thr2, int in_width2, unsigned in_flags2, float* out_current_max2,
                          const float* bn_scale2, const float* bn_shift2, fqStream_t stream);

/* ... under stored thresholds (the unit's 3x3 handed its codes over): x is a C16 code tensor made with in_thr, the result leaves as
 * fp32 y AND as y16, its C16 code copy under out_thr / out_width / out_flags for the next unit's first 1x1 - the two outputs of
 * fq_pwconv_i8_c16_dual - and the shortcut convolution's input x2 is fp32 (x2_is_c16 = 0) or a C16 tensor made with in_thr2.      */
int fq_pwconv_i8_shortcut_c16(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                              float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw,
                              const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                              const float* bn_scale, const float* bn_shift, int act, float* stat_out, const void* x2, int x2_is_c16,
                              const int8_t* wcodes2, const float* wscale2, const int32_t* wsum2, int64_t cin2, int64_t cin2_pad,
                              const float* in_stat2, const float* in_thr2, int in_width2, unsigned in_flags2,
                              float* out_current_max2, const float* bn_scale2, const float* bn_shift2, const float* out_thr,
                              int out_width, unsigned out_flags, fqStream_t stream);

/* A 1x1 convolution and the global average pooling behind it in ONE launch (round 6).  The last 1x1 convolution of the MobileNets
 * (gluon model_zoo: ... Conv2D(1x1), BatchNorm, Activation, GlobalAvgPool2D, Flatten, Dense; the reference wraps the convolution in
 * convert_conv2d.py:53-66,108 and leaves the pooling to MXNet's Pooling operator) writes a tensor whose only reader is that pooling.
 * This entry point computes the values of fq_pwconv_i8_strided(stride = 1, residual) and, without storing them, their plane means
 * as fq_global_avg_pool_stat computes them - pixels added in the order 0 .. hw - 1 in fp64, divided by fp32(hw): bit for bit.
 * y: (n, cout); stat_out[n] = max|y[n]| (the statistic the classifier's activation branch needs, convert_dense.py:41).
 * Shapes: fq_pwconv_i8_gap_supported(n, cin, cout, hw, has_residual) - whole planes of 45 .. 64 pixels.                        */
int fq_pwconv_i8_gap_supported(int64_t n, int64_t cin, int64_t cout, int64_t hw, int has_residual);
int fq_pwconv_i8_gap(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                     float* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t hw, const float* in_stat,
                     const float* in_thr, int in_width, unsigned in_flags, float* out_current_max, const float* bn_scale,
                     const float* bn_shift, int act, float* stat_out, const float* residual, void* ws, fqStream_t stream);

/* ---- a pointwise convolution and the depthwise 3x3 behind it WITHOUT the tensor between them (round 6) ---------------------
 * Under ONLINE input quantisation (convert_conv2d.py:56-58: the threshold of a block is the batch mean of the per-sample maxima
 * of the tensor it is handed in this very forward) a producer cannot hand codes to its consumer - the threshold exists only
 * after the producer's last value.  It CAN be computed twice: for the pair  x -> [1x1 + BN + act] -> y -> [3x3 depthwise + BN +
 * act] -> z  of the MobileNets (y has up to twice x's channels)
 *   (A) fq_pwconv_i8_stat: fq_pwconv_i8 without its stores - stat_out[n] <- max|y[n]|, out_current_max <- the 1x1 block's
 *       `current_input_max`; reads x, writes n floats;
 *   (B) fq_pwdw_fused: recomputes y from x tile by tile (the same exact int32 sums and the same epilogue: bit for bit the
 *       values (A) took the maxima of), fake-quantises it with the threshold mid_stat now determines (LinearQuantizeSTE,
 *       ste_func.py:41, as the depthwise block's activation branch convert_conv2d.py:53-66 would) and runs the depthwise
 *       chain of fq_dwconv3x3 on it: z, stat_out[n] <- max|z[n]|, mid_current_max <- the depthwise block's
 *       `current_input_max`.  y never exists in memory: 4x + 4x + 4z bytes instead of 4x + 4y + 4y + 4z.
 * Values: exactly those of fq_pwconv_i8 followed by fq_dwconv3x3 (tests/test_gpu_pwdw.py: bit-equal z, statistics and
 * thresholds, and against the host twins).  Offline thresholds work too (in_thr / mid_thr: then (A) is not needed at all).
 * Shapes: fq_pwdw_fused_supported - cin in (0, 256] with ceil(cin / 32) in {1, 2, 4, 8}, cout % 32 == 0 (at most 8 wavefronts:
 * ceil(w / 30) * cout / 64 <= 8), w % 4 == 0, w >= 30 or w == 28; stride 2 needs even h and w % 8 == 0.  x: (n, cin, h, w);
 * z: (n, cout, ho, wo).  wcodes: fq_weight_codes' buffer (both copies), cout_pad its rows_pad.  dw_act may carry
 * FQ_STAT_PREZEROED.                                                                                                      */
int fq_pwconv_i8_stat_supported(int64_t n, int64_t cin, int64_t cout, int64_t hw);
int fq_pwconv_i8_stat(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                      int64_t n, int64_t cin, int64_t cin_pad, int64_t cout_pad, int64_t cout, int64_t hw,
                      const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                      const float* bn_scale, const float* bn_shift, int act, float* stat_out, fqStream_t stream);
int fq_pwdw_fused_supported(int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, int stride);
/* Test hook: out[i] <- the fp32 quotient c[i] / d[0] as the two kernels above compute it (the fp32 correction step of
 * csrc/fq_common.h: fast_quot, or the fp64-reciprocal form when d[0] does not qualify; took_fast_path[0] says which) - must
 * equal the IEEE fp32 division bit for bit wherever a code depends on it (tests/test_gpu_pwdw.py).                         */
int fq_debug_fast_quotient(const float* c, int64_t n, const float* d, float* out, int* took_fast_path, fqStream_t stream);
int fq_pwdw_fused(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* pw_bias,
                  int64_t n, int64_t cin, int64_t cin_pad, int64_t cout_pad, int64_t cout, int64_t h, int64_t w,
                  const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, const float* pw_bn_scale,
                  const float* pw_bn_shift, int pw_act, const float* mid_stat, const float* mid_thr, int mid_width,
                  unsigned mid_flags, float* mid_current_max, const float* dw_w, const float* dw_bias, int dw_stride,
                  const float* dw_bn_scale, const float* dw_bn_shift, int dw_act, float* y, float* stat_out,
                  fqStream_t stream);

/* ---- int8 hand-over between fused convolutions under OFFLINE input quantisation (round 3) ------------------------------
 * When the consumer quantises its input with a STORED threshold (`--quantize-input-offline`: convert_conv2d.py:58 takes
 * `input_max`, known before the producer runs), the producer's epilogue can apply the consumer's LinearQuantizeSTE
 * (ste_func.py:41) itself and hand over the integer codes - 1 byte per element instead of 4 written and 4 read - in the
 * layout the matrix cores consume:
 *   C16 code tensor of an (n, C, h, w) activation: int8 [n][ceil(C/16)][h*w][16], byte = (code + 128 - zoff) ^ 0x80 with
 *   zoff = 128 for unsigned codes, 0 for signed ones (what fq_pwconv_i8 builds in LDS from fp32 input); channels past C
 *   hold the code 0.  A pixel's 16 channels are ONE 16-byte vector: a consumer lane loads a B fragment with one instruction
 *   (NCHW fp32: 16 loads and 16 quantisations), a producer lane stores a channel tile with 4 instructions instead of 16.
 * fq_pwconv_i8_c16 is fq_pwconv_i8_strided with either side as a C16 tensor (both at once from 256 input channels up):
 *   x_is_c16 != 0: x is a C16 tensor quantised with in_thr / in_width / in_flags (in_thr required; in_stat, when given, only
 *                  feeds out_current_max - the reference computes `current_input_max` in every mode);
 *   out_thr != NULL: y is a C16 tensor holding the codes of act(BN(conv)) under the CONSUMER's out_thr[0] / out_width /
 *                  out_flags; stat_out still receives max|act(BN(conv))| of the fp32 values (no residual operand then).
 * The values are those of the fp32 hand-over bit for bit: codes are a function of the same fp32 numbers and the same
 * threshold (tests/test_gpu_c16.py: a net with hand-overs == the same net without, logits bit-equal).                     */
int fq_pwconv_i8_c16(const void* x, int x_is_c16, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                     const float* bias, void* y, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                     int stride, const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                     float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                     const float* residual, const float* out_thr, int out_width, unsigned out_flags, void* ws,
                     fqStream_t stream);

/* The closing 1x1 of a residual unit with TWO outputs (round 4): x is a C16 tensor (the unit's 3x3 handed its codes over),
 * y (n, cout, h, w) fp32 = act(BN(conv) + residual) as fq_pwconv_i8_c16 computes it - the shortcut of the NEXT unit needs these
 * values - and y16, a C16 tensor of the same shape, = the codes of y under out_thr / out_width / out_flags, the stored threshold of
 * the next unit's first 1x1, which then reads 1 byte per element instead of 4 (it is itself a codes-in / codes-out call of
 * fq_pwconv_i8_c16 when it hands over to its 3x3).  Same fp32 values, same codes as quantising y afterwards.  stride 1;
 * cin_pad in {64, 128, 256, 512}, cout % 32 == 0.                                                                          */
int fq_pwconv_i8_c16_dual(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                          float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                          const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                          const float* bn_scale, const float* bn_shift, int act, float* stat_out, const float* residual,
                          const float* out_thr, int out_width, unsigned out_flags, void* ws, fqStream_t stream);
/* ... of the LAST unit of a ResNet-v1 stage, whose two readers have stride 2 (see fq_pwconv_i8_sub2): both outputs hold the even
 * pixels of the even rows only - y (n, cout, ceil(h/2), ceil(w/2)) fp32, y16 the C16 tensor of that shape; stat_out and `residual`
 * cover the whole (n, cout, h, w) tensor.  cin_pad as above, cout > 128.                                                       */
int fq_pwconv_i8_c16_dual_sub2(const void* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                               float* y, void* y16, int64_t n, int64_t cin, int64_t cin_pad, int64_t cout, int64_t h, int64_t w,
                               const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                               float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                               const float* residual, const float* out_thr, int out_width, unsigned out_flags, void* ws,
                               fqStream_t stream);

/* Dense 3x3 convolution (stride 1, padding 1, no groups / dilation) on the integer codes: the same identity as
 * fq_pwconv_i8 with K = 9 * Cin, i.e. what the reference's fp32 F.Convolution of the two fake-quantised tensors computes
 * (nn/quantized_conv.py:134-151 spells the integer form out), here exact in int32.  Same contract as fq_pwconv_i8
 * (quantise-on-load, bias / folded BatchNorm / activation / per-sample statistic on store, FQ_STAT_PREZEROED).
 * x: (n, cin, h, w) fp32;  y: (n, cout, h, w) fp32;  cin in {64, 128, 256, 512};  cout >= 32.
 * wcodes: the buffer fq_weight_codes writes for the weights PERMUTED to (cout, 3, 3, cin), i.e. rows = cout, row_len =
 * row_pad = 9 * cin (K ordered tap-major so that a 32-channel slab never straddles two taps), rows_pad = cout rounded up
 * to 64; wscale / wsum as there.                                                                                      */
int fq_conv3x3_i8(const float* x, const int8_t* wcodes, const float* wscale, const int32_t* wsum, const float* bias,
                  float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, const float* in_stat,
                  const float* in_thr, int in_width, unsigned in_flags, float* out_current_max, const float* bn_scale,
                  const float* bn_shift, int act, float* stat_out, fqStream_t stream);

/* fq_conv3x3_i8 with C16 code tensors on either side or both (see fq_pwconv_i8_c16): the 3x3 convolution in the middle
 * of a ResNet bottleneck takes the codes its 1x1 predecessor wrote and writes the codes its 1x1 successor reads.            */
int fq_conv3x3_i8_c16(const void* x, int x_is_c16, const int8_t* wcodes, const float* wscale, const int32_t* wsum,
                      const float* bias, void* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w,
                      const float* in_stat, const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                      const float* bn_scale, const float* bn_shift, int act, float* stat_out, const float* out_thr,
                      int out_width, unsigned out_flags, fqStream_t stream);

/* fq_dwconv3x3 between TWO C16 code tensors: the depthwise layer of a MobileNetV2 unit, whose input the expansion
 * convolution wrote as codes (under this layer's in_thr / in_width / in_flags) and whose output the projection convolution
 * reads as codes (under out_thr / out_width / out_flags).  x^ = code * (in_thr / levels), then fq_dwconv3x3's arithmetic
 * (fmaf chain over (ky, kx), bias, BatchNorm, activation, stat_out <- max|y[n]| of the fp32 values), then the consumer's
 * quantiser.  in_stat, when given, only feeds out_current_max.                                                            */
int fq_dwconv3x3_c16(const void* x, const float* w, const float* bias, void* y, int64_t n, int64_t c, int64_t h, int64_t wdt,
                     int stride, const float* in_stat, const float* in_thr, int in_width, unsigned in_flags,
                     float* out_current_max, const float* bn_scale, const float* bn_shift, int act, float* stat_out,
                     const float* out_thr, int out_width, unsigned out_flags, fqStream_t stream);

/* The same convolution for weights that are NOT integer multiples of one scale per output channel: the filters the
 * reference obtains under Winograd-domain quantisation (convert_conv2d.py:71-83: U^ = STE(s)(G g G^T) lives on the int8
 * grid, the spatial filter g^ = GI U^ GTI that F.Convolution then multiplies does not) - BASELINE config 5.  The filter is
 * split into THREE int8 slices by fq_weight_slices: per output channel p = 2^e, the smallest power of two with
 * max|g^| <= p * 2^20;  m = rint(g^ / p)  (|m| <= 2^20, exact division);  m = d1 2^14 + d2 2^7 + d3, digits in [-64, 64].
 * Every slice is an exact int32 convolution on the matrix cores over the SAME quantised activations; the epilogue forms
 * T = (S1 << 14) + (S2 << 7) + S3 in 64 bits and y = fp32(fp64(T) * fp64(sx * p)), then bias / BatchNorm / activation /
 * statistic as fq_conv3x3_i8.  The sum over the 9 * Cin products is exact; the only error is the filter's representation,
 * p / 2 <= 2^-20 of its channel maximum per weight - the order of the rounding an fp32 convolution of the same two tensors
 * accumulates over 576 ... 4608 products (each product and partial sum rounds at 2^-24 there; parity test: within
 * 3e-6 of the largest output of the fp64 convolution).
 * fq_weight_slices: w (rows, row_len) fp32, here the filter permuted to (cout, 3, 3, cin);  codes: 3 buffers of
 * 2 * rows_pad * row_pad bytes each (row-major codes, then fq_weight_codes' fragment-major copy), slice 1 = d1 first;
 * pscale[rows] = p;  rowsum[3][rows];  ws: rows floats.  cout % 32 == 0.                                                  */
int fq_weight_slices(const float* w, int64_t rows, int64_t row_len, int64_t row_pad, int64_t rows_pad, int8_t* codes,
                     float* pscale, int32_t* rowsum, void* ws, fqStream_t stream);
int fq_conv3x3_i8_sliced(const float* x, const int8_t* wslices, const float* pscale, const int32_t* wsum, const float* bias,
                         float* y, int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, const float* in_stat,
                         const float* in_thr, int in_width, unsigned in_flags, float* out_current_max,
                         const float* bn_scale, const float* bn_shift, int act, float* stat_out, fqStream_t stream);

/* Generic LinearQuantizeSTE.forward (ste_func.py:37-41) for API completeness: x viewed as (rows, row_len) with one
 * scale per row read from the DEVICE array `scales` (rows = 1: scalar scale; rows = Cout: (Cout,1,1,1) broadcast):
 *   y = roundf( (has_clip ? clip(x, clip_lo, clip_hi) : x) / (scales[r] + eps) ) * scales[r].                    */
int fq_ste_forward(const float* x, float* y, int64_t rows, int64_t row_len, const float* scales, int has_clip,
                   float clip_lo, float clip_hi, float eps, fqStream_t stream);

/* ---- weights ---------------------------------------------------------------------------------------------------
 * Replaces convert_conv2d.py:70-95 / convert_dense.py:52-63: w viewed as (rows, row_len); per-row
 *   s = max|w[r,:]| / (2^(width-1)-1);   w_q = roundf(w / (s + 1e-10)) * s     (no clipping).
 * rows = 1 (layer), = Cout (channel / depthwise group), = G with G in {1, Cout} (group).
 * scales_out (device, rows floats, may be NULL) <- s.  ws: fq_weight_workspace_bytes(rows) bytes.               */
size_t fq_weight_workspace_bytes(int64_t rows);
int fq_weight_fake_quant(const float* w, float* w_q, int64_t rows, int64_t row_len, int width, float* scales_out,
                         void* ws, fqStream_t stream);

/* Replaces convert_conv2d.py:71-83 (+ wino_matrix.py): per-out-channel fake-quant in the Winograd domain.
 * w is (cout, cin_g, 3, 3); t = 4/6/8 for F23/F43/F63; G is t x 3, GI = pinv(G) is 3 x t, GTI = pinv(G^T) is
 * t x 3 — HOST pointers, row-major fp32 (the reference computes the pseudo-inverses on the host with numpy).     */
int fq_wino_weight_fake_quant(const float* w, float* w_q, int64_t cout, int64_t cin_g, int t, const float* G,
                              const float* GI, const float* GTI, int width, float* scales_out, void* ws,
                              fqStream_t stream);

/* ---- multi-GPU calibration collectives over RCCL ------------------------------------------------
 * The reference is single-device; sharding the calibration / evaluation batch over the GPUs of a node needs three tiny
 * exchanges (all <= 434 KB, latency-bound, issued on the compute stream): the all-reduce(sum) of L + 1 doubles per
 * naive-EMA calibration step (fq_stat_rows_sum -> HERE -> fq_mean_from_sums, then fq_ema_update: every rank applies
 * `_update_ema`, convert.py:66-79, with the batch mean of the GLOBAL batch), the all-reduce of the exact int64 KL
 * histograms (distribution_calibrate.py:103-104) and of the evaluation counters (simulate_quantization.py:123-147).
 * One process per GPU, one communicator per process, on the HIP device current at fq_comm_init.  librccl.so is bound at
 * run time (no link-time dependency).  Rank 0 calls fq_comm_unique_id and ships the 128 bytes to the other ranks by any
 * channel it has (kvstore, MPI, a file); every rank then calls fq_comm_init(rank, world, id).                          */
#define FQ_COMM_UNIQUE_ID_BYTES 128
#define FQ_COMM_SUM 0
#define FQ_COMM_MAX 1
int fq_comm_unique_id(void* id128);
int fq_comm_init(int rank, int world, const void* id128);
int fq_comm_world(void);   /* ranks of the live communicator, 0 when there is none */
int fq_allreduce_f32(float* buf, int64_t count, int op, fqStream_t stream);   /* in place, device pointer */
int fq_allreduce_f64(double* buf, int64_t count, int op, fqStream_t stream);
int fq_allreduce_i64(int64_t* buf, int64_t count, int op, fqStream_t stream);
int fq_comm_destroy(void);

/* ---- calibration -------------------------------------------------------------------------------------------------
 * Replaces `_update_ema` (convert.py:66-79) for L scalars at once: state <- (1-m)*current + m*state.            */
int fq_ema_update(float* state, const float* current, int64_t count, double momentum, fqStream_t stream);

/* out[0] <- max(x) over the whole tensor (distribution_calibrate.py:33-34: first batch fixes the range).        */
int fq_global_max(const float* x, int64_t numel, float* out, fqStream_t stream);

/* Replaces `_discrete_histogram` + the accumulation at distribution_calibrate.py:39-45,103-104:
 * hist[bins] (uint64, device) += histogram of clip(x, 0, *max_dev) without exact zeros, index
 * = int(v * (bins / (max + 1e-5))), clamped to bins-1.  neg_count (may be NULL) += #elements < 0 (the reference
 * asserts there are none, :35).                                                                                  */
int fq_histogram_accumulate(const float* x, int64_t numel, const float* max_dev, int bins, uint64_t* hist,
                            uint32_t* neg_count, fqStream_t stream);
int fq_hist_to_float(const uint64_t* hist, float* out, int64_t count, fqStream_t stream);

/* Replaces `kl_calibrate` (distribution_calibrate.py:117-171) for L histograms at once.
 * hist: (L, bins) fp32; out_best: (L) int32.  ws: fq_kl_workspace_bytes(L, bins) bytes.                         */
size_t fq_kl_workspace_bytes(int64_t L, int bins);
int fq_kl_search(const float* hist, int64_t L, int bins, int levels, int min_bins, int32_t* out_best, void* ws,
                 fqStream_t stream);

/* ---- int-code path (nn/quantized_conv.py:54-76) --------------------------------------------------------------------
 * codes <- int32(roundf(clip(x, min, max) / scale)); scale = max/127 if max == -min else (max-min)/255.
 * range_dev: device float[3] = {min, max, scale}; written by modes INT8/UINT8 (and scale by RANGE), read by SCALE.
 * ws: fq_act_workspace_bytes(1).                                                                                 */
int fq_quantize_codes(const float* x, int32_t* codes, int64_t numel, int mode, float* range_dev, void* ws,
                      fqStream_t stream);
/* y <- float(codes) * scale_dev[0]  (dequantize, :74-76) */
int fq_dequantize(const int32_t* codes, float* y, int64_t numel, const float* scale_dev, fqStream_t stream);

/* ---- nn.Conv2D(quantized=True): the reference's stand-alone quantised convolution (nn/quantized_conv.py:106-159) ---------
 * `Conv2D.hybrid_forward` with quantized=True, as one call per forward:  pad -> `quantize(inputs, input_dtype)` (ONE global
 * range: int8 [-max|x|, max|x|], scale max/127; uint8 [min, max] of the padded tensor, scale (max - min)/255, no zero point,
 * no epsilon, :54-72) -> `quantize(weight, weight_dtype)` -> int32 bias codes clip(b, +-s 2^31)/s with s = in_scale * w_scale
 * (:122-127) -> integer correlation per group (the im2col + dot of :129-151, here exact in wrapping int32 for any accumulator
 * size) -> activation on the integers -> `dequantize` by s (:157-158).  y = float(relu?(sum + bias_code)) * s, bit for bit
 * what oracle.qconv2d_forward computes.  No im2col tensor, no int32 code tensor: the quantiser sits on the convolution's loads.
 *   1x1 stride 1 no padding, groups 1          -> the pointwise forms on the int8 matrix cores (fq_pwconv_i8's kernels)
 *   3x3 stride 1 padding 1, groups 1, Cin 64.. -> the implicit-GEMM kernel of fq_conv3x3_i8
 *   3x3 stride 1|2 padding 1, depthwise        -> the depthwise forms of fq_dwconv3x3 on integer codes (no bias)
 *   anything else, uint8 / fixed-range weights -> an exact one-output-per-thread kernel (force_direct != 0 asks for it)
 * followed by a conditional exact recomputation that returns at once unless the range record says the fast kernel's 8-bit
 * representation did not hold for this input (a code span of 257 values, a clip range without the padding zero).
 *
 * fq_qconv_weights_prepare: once per weight version.  wbuf: fq_qconv_weights_bytes(...) bytes, 16-byte aligned; ws: 16 bytes.
 *   weight_mode FQ_CODES_INT8 / FQ_CODES_UINT8 / FQ_CODES_RANGE (`_weight_range` = [w_min, w_max]); only int8 weights take the
 *   fast kernels (the caller passes force_direct for the others).
 * fq_qconv2d_forward: x (n, cin, h, w) fp32 UNPADDED; w (cout, cin/groups, kh, kw) fp32 (read by the direct form); bias fp32
 *   (cout) or NULL; y (n, cout, ho, wo) fp32 with ho = (h + 2 ph - kh)/sh + 1.  input_mode FQ_CODES_INT8 / UINT8, or
 *   FQ_CODES_RANGE with `_input_range` = [in_min, in_max].  in_stat (may be NULL; n floats): per-sample maxima of a NON-NEGATIVE
 *   x as a fused producer left them in its epilogue (BatchNorm + ReLU): the range pass over x is skipped - max = their maximum;
 *   the minimum is 0 by construction with padding, and without padding (uint8) it is found by a scan that stops at the first
 *   zero (stat[0] < 0 = "the producer's layer was recomputed": one full scan instead).  Results are those of the range pass.
 *   act: FQ_ACT_NONE / FQ_ACT_RELU [| FQ_STAT_PREZEROED]; FQ_ACT_RELU6 only with a folded BatchNorm.  bn_scale / bn_shift (may
 *   be NULL): an inference BatchNorm behind the block folded into the store, y' = y * bn_scale[c] + bn_shift[c] (separately
 *   rounded, as fq_bn_act_stat), the activation then applies to y' (as the separate Activation block would).  stat_out (may
 *   be NULL): per-sample max|y'|, or stat_out[0] = -1 when the layer went through the exact direct kernel.  ws:
 *   fq_qconv_workspace_bytes(cout) bytes, 128-byte aligned, initialised ONCE by fq_qconv_workspace_init (every forward leaves
 *   it initialised); not to be shared by forwards that may run concurrently.                                              */
size_t fq_qconv_weights_bytes(int64_t cin, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw, int groups);
/* which kernel family a geometry takes: 0 direct, 1 pointwise (matrix cores), 2 dense 3x3 (matrix cores), 3 depthwise 3x3 */
int fq_qconv_kind(int64_t cin, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw, int groups);
int fq_qconv_weights_prepare(const float* w, int64_t cin, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw,
                             int groups, int weight_mode, float w_min, float w_max, void* wbuf, void* ws,
                             fqStream_t stream);
size_t fq_qconv_workspace_bytes(int64_t cout);
int fq_qconv_workspace_init(void* ws, fqStream_t stream);
int fq_qconv2d_forward(const float* x, const float* w, const void* wbuf, const float* bias, float* y, int64_t n, int64_t cin,
                       int64_t h, int64_t wdt, int64_t cout, int kh, int kw, int sh, int sw, int ph, int pw, int groups,
                       int input_mode, float in_min, float in_max, const float* in_stat, int act, const float* bn_scale,
                       const float* bn_shift, float* stat_out, void* ws, int force_direct, fqStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FAKEQUANT_H_ */
