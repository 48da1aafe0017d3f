// libfakequant — K2m pointwise (1x1) convolution on int8 codes for the deep layers: one (pixel tile, channel group) per
// workgroup (see fq_common.h for the list of translation units and the design rules; the kernel: fq_pw_split_kernel.h)
#include "fq_pw_split_kernel.h"

namespace fqi {

// split form (K2m): grid = tiles x channel groups.  Takes every Cout and every Cin whose padded row length (cin_pad, the
// row stride fq_weight_codes was given) is one of the instantiated K/32 below: the MobileNet, MobileNetV2 and ResNet-50
// channel counts.  Measured against the other forms on every such shape (tools/pwforms.py, tools/kprof.sh;
// profiles/r2_pw_split.txt): faster everywhere except on the largest planes (> 4096 tiles of 32 pixels), where the streaming
// form's persistent wavefronts with LDS-resident weights win by a few per cent - so with more tiles than that it only
// takes what the streaming form cannot.
int pw_try_split(const PwCall& a, bool* taken) {
  *taken = false;
  const int kt = (int)(a.cin_pad / 32);
  const int64_t tiles = (a.n * a.hw + 31) / 32;
  const int64_t rows_pad = (a.cout + 63) / 64 * 64;
  static const int kts[] = {2, 4, 6, 8, 10, 12, 16, 18, 30, 32, 64};
  bool shape_ok = false;
  for (int k : kts) shape_ok = shape_ok || k == kt;
  // a tile's stores must stay below 2 GiB from its first sample (out-of-range lane offsets mask partial channel tiles)
  const int64_t hw_in = a.stride == 1 ? a.hw : a.h_in * a.w_in;
  shape_ok = shape_ok && (32 / a.hw + 2) * a.cout * a.hw * 4 < (1ll << 31) && a.cin * hw_in * 4 * (32 / a.hw + 2) < (1ll << 31);
  static const int mode = env_int("FQ_PWS_AUTO", 1);                    // tuning: 0 never, 1 by shape, 2 always
  const bool c16 = a.in_c16 || a.out_thr != nullptr;                    // (only this form reads / writes C16 code tensors)
  FQ_REQUIRE(a.y16 == nullptr || (a.in_c16 && a.out_thr == nullptr && a.stride == 1), "fq_pwconv_i8_c16_dual: a second output "
             "goes with a C16 input, fp32 y and stride 1");
  // (r4, tools/pwforms.py --resnet: with a residual operand the streaming form loses its lead on the largest planes -
  // 64 -> 256 @56x56 + residual, ResNet-50's largest kernel: 207.5 us against 183.7 here)
  static const int res_split = env_int("FQ_PWS_RES_SPLIT", 1);
  bool want = a.form == 6 || a.stride != 1 || c16 || a.sub;             // (only this form reads strided inputs)
  // (a subsampled output: fp32 in and out, or - the dual form - codes in, fp32 out and its code copy, both subsampled)
  if (a.sub) shape_ok = shape_ok && pw_split_sub_shape_ok(a.cin_pad, a.cout) && a.stride == 1 && (!c16 || a.y16 != nullptr);
  if (a.form == 0 && shape_ok && a.stride == 1 && !a.sub)
    want = mode == 2 || (mode == 1 && (tiles <= (int64_t)num_cu() * 16 || !pw_stream_shape_ok(a) ||
                                       (res_split && a.residual != nullptr && a.cin <= 64)));
  if (want && shape_ok) {
    // channel tiles per wavefront (cw) and wavefronts per SIMD (lb).  Measured in the model: two tiles per wavefront at
    // four wavefronts per SIMD is the best or within 3 % of it on every shape (the grid then fills the chip in one or two
    // resident rounds); narrow layers take one tile per wavefront so that all four wavefronts have channels.
    int cw = a.cout > 128 ? 2 : 1;
    int lb = 4;
    // eight wavefronts per workgroup for wide layers: a tile is then quantised by half as many channel groups (measured in
    // the model: 1024 -> 1024 @7x7 31.2 -> 25.8 us, 512 -> 1024 @7x7 19.8 -> 18.5).  Alone the 14x14 layers with their 784
    // tiles get 5 % slower that way (round 3: hence "few tiles only"); among three batches in flight what counts is the work
    // saved: ResNet-50 online +0.75 % with the limit at 1024 tiles, +1.2 % at 4096 = 16 tiles per CU (28x28 planes included),
    // +1.1 % without limit (round 5, alternating runs: profiles/r5_r50_nw8_ab.txt)
    static const int nw_tune = env_int("FQ_PWS_NW", 0);
    const bool nw8_built = kt == 8 || kt == 16 || kt == 32 || kt == 64;
    static const int nw8_tiles = env_int("FQ_PWS_NW8_TILES", 0);        // tuning: most tiles a layer may have to take nw = 8
    const int64_t nw8_max = nw8_tiles > 0 ? nw8_tiles : 16 * (int64_t)num_cu();
    int nw = (nw_tune == 4 || nw_tune == 8) ? nw_tune : ((a.cout >= 512 && tiles <= nw8_max) ? 8 : 4);
    // (C16 code tensors: four wavefronts - except the dual form of the 14x14 / 7x7 stages, K = 256 / 512, round 6: eight, i.e. 512
    // channels per workgroup and half as many workgroups copying the same tile's codes - FQ_PWS16_DUAL_NW8=0 for the A/B)
    static const int dual_nw8 = env_int("FQ_PWS16_DUAL_NW8", 1);
    const bool dual8 = c16 && a.y16 != nullptr && !a.sub && dual_nw8 && (kt == 8 || kt == 16) && a.cout >= 512 && tiles <= nw8_max;
    if (!nw8_built || (c16 && !dual8) || a.sub) nw = 4;
    if (dual8) nw = 8;
    const int tune = (c16 || a.sub) ? 0 : env_int("FQ_PWS_CFG", 0);      // tuning: 10 * lb + cw, read per call
    if (tune > 0) {
      cw = tune % 10;
      lb = tune / 10;
    }
    PwSplitGeom t;
    t.Cin = (int)a.cin; t.Cout = (int)a.cout; t.HW = (int)a.hw;
    if (lb != 4 || cw != 2) nw = 4;                                      // (the tuning configurations are built for 4)
    t.CS = (int)((a.cout + 32 * nw * cw - 1) / (32 * nw * cw));
    t.CTM = (int)(rows_pad / 32);
    t.S = a.stride; t.Wo = (int)a.w_out; t.Win = (int)a.w_in; t.HWin = (int)hw_in;
    t.cols = a.n * a.hw; t.tiles = tiles; t.zoff = a.zoff;
    t.items = tiles * t.CS;
    t.CBi = (int)((a.cin + 15) / 16); t.CBo = (int)((a.cout + 15) / 16);
    t.out_levels = a.out_levels; t.out_lo_neg = a.out_lo_neg; t.out_zoff = a.out_zoff;
    t.y16 = (char*)a.y16; t.dual_thr = a.dual_thr;
    t.SW = a.sub ? (int)a.w_in : 0;
    t.SWs = a.sub ? (int)((a.w_in + 1) / 2) : 0;
    t.SHWs = a.sub ? (int)((a.h_in + 1) / 2) * t.SWs : 0;
    const int64_t grid = (t.items + 7) / 8 * 8;                           // padded to whole rounds over the 8 XCDs
    FQ_REQUIRE(grid < (1ll << 31), "fq_pwconv_i8: too many tiles for the split form");
    const size_t ldst = (size_t)kt * 1024 + (size_t)(nw * cw * 32) * 5 * sizeof(float);
    const int8_t* wfrag = a.wcodes + rows_pad * a.cin_pad;               // second half of fq_weight_codes' buffer
    if (int rc = pw_zero_stat(a)) return rc;
    bool launched = false;
    if (a.sub && !c16) {
      if (int rc = pw_split_sub_launch(a, &t, kt, grid, ldst, wfrag, &launched)) return rc;
      FQ_REQUIRE(launched, "fq_pwconv_i8_sub2: no instantiation for K/32=%d", kt);
      FQ_LAUNCH_CHECK();
      *taken = true;
      return FQ_OK;
    }
    if (c16) {
      FQ_REQUIRE(a.residual == nullptr || a.out_thr == nullptr, "fq_pwconv_i8_c16: a residual operand goes with fp32 output");
      if (int rc = pw_split16_launch(a, &t, kt, cw, grid, ldst, wfrag, &launched, nw)) return rc;
      FQ_REQUIRE(launched, "fq_pwconv_i8_c16: no instantiation for K/32=%d with %d channel tiles per wavefront", kt, cw);
      FQ_LAUNCH_CHECK();
      *taken = true;
      return FQ_OK;
    }
#define FQ_PWS_CASE_NW(KT_, CW_, D_, LB_, NW_)                                                                         \
  if (kt == KT_ && cw == CW_ && lb == LB_ && nw == NW_) {                                                              \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_split_kernel<KT_, CW_, D_, LB_, NW_>),               \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;                      \
    FQ_REQUIRE(attr_ok, "fq_pwconv_i8: cannot raise the dynamic LDS limit of the split kernel");                       \
    hipLaunchKernelGGL((pwconv_split_kernel<KT_, CW_, D_, LB_, NW_>), dim3((unsigned)grid), dim3(NW_ * 64), ldst, a.st, \
                       a.x,                                                                                            \
                       wfrag, a.wscale, (const int*)a.wsum, a.bias, a.y, t, a.in_stat, (int)a.n, a.in_thr, a.levels,   \
                       a.lo_neg, kEps, a.out_current_max, a.bn_scale, a.bn_shift, a.act, a.stat_out, a.residual,       \
                       (const float*)nullptr);                                                                         \
    launched = true;                                                                                                   \
  }
#define FQ_PWS_CASE(KT_, CW_, D_, LB_) FQ_PWS_CASE_NW(KT_, CW_, D_, LB_, 4)
    // every K: the two default configurations; the power-of-two K of the deep MobileNet / ResNet layers also carry the
    // alternatives that tools/pwforms.py and FQ_PWS_CFG compare (three wavefronts per SIMD, four channel tiles)
#define FQ_PWS_KT(KT_) FQ_PWS_CASE(KT_, 1, (KT_ < 7 ? KT_ : 7), 4) FQ_PWS_CASE(KT_, 2, (KT_ < 3 ? KT_ : 3), 4)
#define FQ_PWS_KT_TUNE(KT_) \
  FQ_PWS_CASE(KT_, 1, 7, 3) FQ_PWS_CASE(KT_, 2, 5, 3) FQ_PWS_CASE(KT_, 4, 2, 3)
    FQ_PWS_KT(2) FQ_PWS_KT(4) FQ_PWS_KT(6) FQ_PWS_KT(8) FQ_PWS_KT(10) FQ_PWS_KT(12) FQ_PWS_KT(16) FQ_PWS_KT(18)
    FQ_PWS_KT(30) FQ_PWS_KT(32) FQ_PWS_KT(64)
    FQ_PWS_KT_TUNE(8) FQ_PWS_KT_TUNE(16) FQ_PWS_KT_TUNE(32) FQ_PWS_KT_TUNE(64)
    FQ_PWS_CASE_NW(8, 2, 3, 4, 8) FQ_PWS_CASE_NW(16, 2, 3, 4, 8) FQ_PWS_CASE_NW(32, 2, 3, 4, 8) FQ_PWS_CASE_NW(64, 2, 3, 4, 8)
#undef FQ_PWS_KT
#undef FQ_PWS_KT_TUNE
#undef FQ_PWS_CASE
#undef FQ_PWS_CASE_NW
    FQ_REQUIRE(launched, "fq_pwconv_i8: no instantiation of the split form for K/32=%d, %d tiles per wavefront, %d "
               "wavefronts per SIMD", kt, cw, lb);
    FQ_LAUNCH_CHECK();
    *taken = true;
    return FQ_OK;
  }
  FQ_REQUIRE(!a.sub, "fq_pwconv_i8_sub2: shape not taken (see fq_pwconv_i8_sub2_supported)");
  FQ_REQUIRE(!c16, "fq_pwconv_i8_c16: the shape does not fit the split kernel (padded Cin / 32 = %d is not instantiated)", kt);
  FQ_REQUIRE(a.form != 6 && a.stride == 1, "fq_pwconv_i8: %s but the shape does not fit the split kernel (padded Cin / 32 "
             "= %d is not instantiated)", a.stride == 1 ? "FQ_PW_FORM=6" : "a strided call", kt);
  return FQ_OK;
}

}  // namespace fqi
