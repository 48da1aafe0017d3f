// libfakequant — K2m split form with C16 code tensors on either side (fq_pwconv_i8_c16): the instantiations
// (see fq_common.h for the list of translation units and the design rules; the kernel: fq_pw_split_kernel.h)
#include "fq_pw_split_kernel.h"

namespace fqi {

// IN16 / OUT16 variants of the two default configurations of every K (four wavefronts per SIMD, one or two channel tiles per
// wavefront).  Both sides at once (round 4): the first 1x1 of a ResNet unit whose producer stored the trunk a second time as
// this convolution's codes (fq_pwconv_i8_c16_dual) reads codes and hands codes to the unit's 3x3 - K = 256 ... 2048 only.
int pw_split16_launch(const PwCall& a, const void* geom, int kt, int cw, int64_t grid, size_t lds, const int8_t* wfrag,
                      bool* launched, int nw) {
  const PwSplitGeom& t = *static_cast<const PwSplitGeom*>(geom);
  const bool in16 = a.in_c16, out16 = a.out_thr != nullptr, dual = a.y16 != nullptr, sub = a.sub;
  FQ_REQUIRE(!(in16 && out16) || kt >= 8, "fq_pwconv_i8_c16: codes in AND codes out is built for 256 input channels and more");
  FQ_REQUIRE(!in16 || a.in_thr != nullptr, "fq_pwconv_i8_c16: a C16 input was quantised with a stored threshold: give in_thr");
#define FQ_PWS16_CASE(KT_, CW_, D_, IN_, OUT_) FQ_PWS16_CASE_D(KT_, CW_, D_, 4, IN_, OUT_, false)
#define FQ_PWS16_CASE_D(KT_, CW_, D_, LB_, IN_, OUT_, DUAL_) FQ_PWS16_CASE_S(KT_, CW_, D_, LB_, IN_, OUT_, DUAL_, false)
#define FQ_PWS16_CASE_S(KT_, CW_, D_, LB_, IN_, OUT_, DUAL_, SUB_)                                                     \
  if (nw == 4 && kt == KT_ && cw == CW_ && in16 == IN_ && out16 == OUT_ && dual == DUAL_ && sub == SUB_) {             \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_split_kernel<KT_, CW_, D_, LB_, 4, IN_, OUT_, DUAL_, SUB_>), \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;                      \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8_c16: cannot raise the dynamic LDS limit of the split kernel");                   \
    hipLaunchKernelGGL((pwconv_split_kernel<KT_, CW_, D_, LB_, 4, IN_, OUT_, DUAL_, SUB_>), dim3((unsigned)grid), dim3(256), lds, a.st, \
                       a.x, wfrag, a.wscale, (const int*)a.wsum, a.bias, a.y, t, a.in_stat, (int)a.n, a.in_thr, a.levels, \
                       a.lo_neg, kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out, a.residual,       \
                       a.out_thr);                                                                                     \
    *launched = true;                                                                                                  \
  }
#define FQ_PWS16_KT(KT_)                                                                                               \
  FQ_PWS16_CASE(KT_, 1, (KT_ < 7 ? KT_ : 7), true, false) FQ_PWS16_CASE(KT_, 2, (KT_ < 3 ? KT_ : 3), true, false)       \
  FQ_PWS16_CASE(KT_, 1, (KT_ < 7 ? KT_ : 7), false, true) FQ_PWS16_CASE(KT_, 2, (KT_ < 3 ? KT_ : 3), false, true)
  FQ_PWS16_KT(2) FQ_PWS16_KT(4) FQ_PWS16_KT(6) FQ_PWS16_KT(8) FQ_PWS16_KT(10) FQ_PWS16_KT(12) FQ_PWS16_KT(16)
  FQ_PWS16_KT(18) FQ_PWS16_KT(30) FQ_PWS16_KT(32) FQ_PWS16_KT(64)
#define FQ_PWS16_BOTH(KT_) FQ_PWS16_CASE(KT_, 1, 7, true, true) FQ_PWS16_CASE(KT_, 2, 3, true, true)
  FQ_PWS16_BOTH(8) FQ_PWS16_BOTH(16) FQ_PWS16_BOTH(32) FQ_PWS16_BOTH(64)
#undef FQ_PWS16_BOTH
  // codes in, fp32 out AND a code copy of it (the closing 1x1 of a ResNet unit: K = 64 ... 512, Cout = 256 ... 2048)
#ifndef FQ_PWS16_DUAL_D                 // tuning: ring depth and wavefronts per SIMD of the two-tile dual-output instantiations
#define FQ_PWS16_DUAL_D 2
#endif
#ifndef FQ_PWS16_DUAL_LB
#define FQ_PWS16_DUAL_LB 4
#endif
#define FQ_PWS16_DUAL(KT_) FQ_PWS16_CASE_D(KT_, 1, (KT_ < 7 ? KT_ : 7), 4, true, false, true) \
  FQ_PWS16_CASE_D(KT_, 2, (KT_ < FQ_PWS16_DUAL_D ? KT_ : FQ_PWS16_DUAL_D), FQ_PWS16_DUAL_LB, true, false, true)
  FQ_PWS16_DUAL(2) FQ_PWS16_DUAL(4) FQ_PWS16_DUAL(8) FQ_PWS16_DUAL(16)
#undef FQ_PWS16_DUAL
  // ... both outputs subsampled (fq_pwconv_i8_c16_dual_sub2: the last unit of a ResNet-v1 stage, two channel tiles per wavefront)
#define FQ_PWS16_DUAL_SUB(KT_) \
  FQ_PWS16_CASE_S(KT_, 2, (KT_ < FQ_PWS16_DUAL_D ? KT_ : FQ_PWS16_DUAL_D), FQ_PWS16_DUAL_LB, true, false, true, true)
  FQ_PWS16_DUAL_SUB(2) FQ_PWS16_DUAL_SUB(4) FQ_PWS16_DUAL_SUB(8) FQ_PWS16_DUAL_SUB(16)
#undef FQ_PWS16_DUAL_SUB
#undef FQ_PWS16_CASE_S
  // the dual form with EIGHT wavefronts (512 channels per workgroup): 256 -> 1024 @14x14, 512 -> 2048 @7x7
#define FQ_PWS16_DUAL8(KT_)                                                                                            \
  if (nw == 8 && kt == KT_ && cw == 2 && in16 && !out16 && dual && !sub) {                                             \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_split_kernel<KT_, 2, FQ_PWS16_DUAL_D, FQ_PWS16_DUAL_LB, 8, true, false, true>), \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;                      \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8_c16_dual: cannot raise the dynamic LDS limit of the split kernel");              \
    hipLaunchKernelGGL((pwconv_split_kernel<KT_, 2, FQ_PWS16_DUAL_D, FQ_PWS16_DUAL_LB, 8, true, false, true>), dim3((unsigned)grid), \
                       dim3(512), lds, a.st, a.x, wfrag, a.wscale, (const int*)a.wsum, a.bias, a.y, t, a.in_stat, (int)a.n, \
                       a.in_thr, a.levels, a.lo_neg, kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out, \
                       a.residual, a.out_thr);                                                                         \
    *launched = true;                                                                                                  \
  }
  FQ_PWS16_DUAL8(8) FQ_PWS16_DUAL8(16)
#undef FQ_PWS16_DUAL8
#undef FQ_PWS16_CASE_D
#undef FQ_PWS16_KT
#undef FQ_PWS16_CASE
  return FQ_OK;
}

}  // namespace fqi
