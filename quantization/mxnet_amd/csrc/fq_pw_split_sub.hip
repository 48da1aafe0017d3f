// libfakequant — K2m split form storing only the even pixels of the even rows of its output (fq_pwconv_i8_sub2): the instantiations
// (see fq_common.h for the list of translation units and the design rules; the kernel: fq_pw_split_kernel.h)
#include "fq_pw_split_kernel.h"

namespace fqi {

// The closing 1x1 convolution of the LAST unit of a ResNet-v1 stage (64 -> 256 @56x56, 128 -> 512 @28x28, 256 -> 1024 @14x14; the
// deeper nets of the family have the same three): both readers of its output - the first 1x1 and the shortcut 1x1 of the next
// stage's first unit - are stride-2 convolutions without padding, i.e. three quarters of the tensor are never read.  Two channel
// tiles per wavefront, four wavefronts per workgroup and per SIMD: the default configuration of the fp32 instantiations.
bool pw_split_sub_shape_ok(int64_t cin_pad, int64_t cout) {
  const int64_t kt = cin_pad / 32;
  return (kt == 2 || kt == 4 || kt == 8 || kt == 16) && cout > 128;
}

int pw_split_sub_launch(const PwCall& a, const void* geom, int kt, int64_t grid, size_t lds, const int8_t* wfrag, bool* launched) {
  const PwSplitGeom& t = *static_cast<const PwSplitGeom*>(geom);
#define FQ_PWSUB_CASE(KT_, D_)                                                                                         \
  if (kt == KT_) {                                                                                                     \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_split_kernel<KT_, 2, D_, 4, 4, false, false, false, true>), \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;                      \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8_sub2: cannot raise the dynamic LDS limit of the split kernel");                  \
    hipLaunchKernelGGL((pwconv_split_kernel<KT_, 2, D_, 4, 4, false, false, false, true>), dim3((unsigned)grid), dim3(256), lds, \
                       a.st, a.x, wfrag, a.wscale, (const int*)a.wsum, a.bias, a.y, t, a.in_stat, (int)a.n, a.in_thr,   \
                       a.levels, a.lo_neg, kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out,         \
                       a.residual, (const float*)nullptr);                                                             \
    *launched = true;                                                                                                  \
  }
  FQ_PWSUB_CASE(2, 2) FQ_PWSUB_CASE(4, 3) FQ_PWSUB_CASE(8, 3) FQ_PWSUB_CASE(16, 3)
#undef FQ_PWSUB_CASE
  return FQ_OK;
}

}  // namespace fqi
