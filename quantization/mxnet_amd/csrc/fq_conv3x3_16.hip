// libfakequant — K2n dense 3x3 convolution with C16 code tensors on either side (fq_conv3x3_i8_c16): the instantiations
// (see fq_common.h for the list of translation units and the design rules; the kernel: fq_conv3x3_kernel.h)
#include "fq_conv3x3_kernel.h"

namespace fqi {

int conv3x3_c16_launch(const float* x, const int8_t* wfrag, const float* wscale, const int32_t* wsum, const float* bias,
                       float* y, const void* geom, int kt, int ptw, int wc, int64_t grid, size_t lds, hipStream_t st,
                       const float* in_stat, int n, const float* in_thr, float levels, int lo_neg, float* out_current_max,
                       const float* bn_scale, const float* bn_shift, int act, float* stat_out, bool in16,
                       const float* out_thr, bool* launched) {
  const C3Geom& g = *static_cast<const C3Geom*>(geom);
  const bool out16 = out_thr != nullptr;
#define FQ_C316_CASE(KT_, PTW_, WC_, D_, IN_, OUT_)                                                                    \
  if (kt == KT_ && ptw == PTW_ && wc == WC_ && in16 == IN_ && out16 == OUT_) {                                         \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_i8_kernel<KT_, PTW_, WC_, D_, 4, 4, 1, IN_, OUT_>), \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;                     \
    FQ_REQUIRE(attr_ok, "fq_conv3x3_i8_c16: cannot raise the dynamic LDS limit");                                      \
    hipLaunchKernelGGL((conv3x3_i8_kernel<KT_, PTW_, WC_, D_, 4, 4, 1, IN_, OUT_>), dim3((unsigned)grid), dim3(256), lds, st, \
                       x, wfrag, wscale, (const int*)wsum, bias, y, g, in_stat, n, in_thr, levels, lo_neg, kEps,       \
                       out_current_max, bn_scale, bn_shift, act, stat_out, out_thr);                                   \
    *launched = true;                                                                                                  \
  }
#define FQ_C316_IO(KT_, PTW_, WC_, D_)                                                                                 \
  FQ_C316_CASE(KT_, PTW_, WC_, D_, true, true) FQ_C316_CASE(KT_, PTW_, WC_, D_, true, false)                           \
  FQ_C316_CASE(KT_, PTW_, WC_, D_, false, true)
#define FQ_C316_KT(KT_) FQ_C316_IO(KT_, 1, 4, 6) FQ_C316_IO(KT_, 2, 4, 4) FQ_C316_IO(KT_, 1, 2, 6) FQ_C316_IO(KT_, 2, 2, 4)
  FQ_C316_KT(2) FQ_C316_KT(4) FQ_C316_KT(8) FQ_C316_KT(16)
#undef FQ_C316_KT
#undef FQ_C316_IO
#undef FQ_C316_CASE
  // eight wavefronts (256 channels per workgroup): codes in and out on the 256-channel layers (round 6)
#define FQ_C316_CASE8(KT_, PTW_, D_)                                                                                   \
  if (kt == KT_ && ptw == PTW_ && wc == 8 && in16 && out16) {                                                          \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_i8_kernel<KT_, PTW_, 8, D_, 4, 8, 1, true, true>),  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;                     \
    FQ_REQUIRE(attr_ok, "fq_conv3x3_i8_c16: cannot raise the dynamic LDS limit");                                      \
    hipLaunchKernelGGL((conv3x3_i8_kernel<KT_, PTW_, 8, D_, 4, 8, 1, true, true>), dim3((unsigned)grid), dim3(512), lds, st, \
                       x, wfrag, wscale, (const int*)wsum, bias, y, g, in_stat, n, in_thr, levels, lo_neg, kEps,       \
                       out_current_max, bn_scale, bn_shift, act, stat_out, out_thr);                                   \
    *launched = true;                                                                                                  \
  }
  FQ_C316_CASE8(8, 1, 6) FQ_C316_CASE8(8, 2, 4)
#undef FQ_C316_CASE8
  return FQ_OK;
}

}  // namespace fqi
