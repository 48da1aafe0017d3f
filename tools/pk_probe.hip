// Probe (round 5): what a packed fp32 multiply-add costs on gfx950 when its three operands are DIFFERENT register pairs.
// tools/valu_probe.hip (round 2) measured v_pk_fma_f32 at 5.1 cycles per wavefront instruction - with the same pair as second and
// third operand.  Here: hard-coded registers, so that the banks (register number mod 4) of the operands are known.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_probe.hip -o build_tools/pk_probe && build_tools/pk_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43"
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float a, int reps) {
  asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, %0\n\tv_mov_b32 v24, %0\n\tv_mov_b32 v25, %0\n\t"
               "v_mov_b32 v26, %0\n\tv_mov_b32 v27, %0\n\tv_mov_b32 v28, %0\n\tv_mov_b32 v29, %0\n\tv_mov_b32 v30, %0\n\tv_mov_b32 v31, %0\n\t"
               "v_mov_b32 v32, %0\n\tv_mov_b32 v33, %0\n\tv_mov_b32 v34, %0\n\tv_mov_b32 v35, %0\n\tv_mov_b32 v36, %0\n\tv_mov_b32 v37, %0\n\t"
               "v_mov_b32 v38, %0\n\tv_mov_b32 v39, %0\n\tv_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v42, %0\n\tv_mov_b32 v43, %0" ::"v"(a) : CLOB);
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      // four independent accumulators v[20:21] v[22:23] v[24:25] v[26:27]; sources v[28..43]
      if (OP == 0)   // scalar fma, three distinct registers
        asm volatile("v_fma_f32 v20, v28, v33, v20\n\tv_fma_f32 v21, v29, v34, v21\n\tv_fma_f32 v22, v30, v35, v22\n\tv_fma_f32 v23, v31, v32, v23" ::: CLOB);
      if (OP == 1)   // packed, sources the SAME pair (the round-2 probe)
        asm volatile("v_pk_fma_f32 v[20:21], v[28:29], v[28:29], v[20:21]\n\tv_pk_fma_f32 v[22:23], v[30:31], v[30:31], v[22:23]\n\t"
                     "v_pk_fma_f32 v[24:25], v[32:33], v[32:33], v[24:25]\n\tv_pk_fma_f32 v[26:27], v[34:35], v[34:35], v[26:27]" ::: CLOB);
      if (OP == 2)   // packed, three different pairs, all starting in bank 0
        asm volatile("v_pk_fma_f32 v[20:21], v[28:29], v[32:33], v[20:21]\n\tv_pk_fma_f32 v[24:25], v[36:37], v[40:41], v[24:25]\n\t"
                     "v_pk_fma_f32 v[20:21], v[28:29], v[32:33], v[20:21]\n\tv_pk_fma_f32 v[24:25], v[36:37], v[40:41], v[24:25]" ::: CLOB);
      if (OP == 3)   // packed, three different pairs: accumulator in banks 0-1, sources in banks 2-3 and 0-1
        asm volatile("v_pk_fma_f32 v[20:21], v[30:31], v[32:33], v[20:21]\n\tv_pk_fma_f32 v[24:25], v[34:35], v[36:37], v[24:25]\n\t"
                     "v_pk_fma_f32 v[20:21], v[38:39], v[40:41], v[20:21]\n\tv_pk_fma_f32 v[24:25], v[42:43], v[28:29], v[24:25]" ::: CLOB);
      if (OP == 4)   // packed multiply, two different pairs
        asm volatile("v_pk_mul_f32 v[20:21], v[30:31], v[32:33]\n\tv_pk_mul_f32 v[24:25], v[34:35], v[36:37]\n\t"
                     "v_pk_mul_f32 v[22:23], v[38:39], v[40:41]\n\tv_pk_mul_f32 v[26:27], v[42:43], v[28:29]" ::: CLOB);
      if (OP == 5)   // packed multiply-add with a scalar pair as one source
        asm volatile("v_pk_fma_f32 v[20:21], v[30:31], s[2:3], v[20:21]\n\tv_pk_fma_f32 v[24:25], v[34:35], s[2:3], v[24:25]\n\t"
                     "v_pk_fma_f32 v[22:23], v[38:39], s[2:3], v[22:23]\n\tv_pk_fma_f32 v[26:27], v[42:43], s[2:3], v[26:27]" ::: CLOB);
      if (OP == 6)   // packed, accumulate in place with op_sel broadcasting ONE register of a pair (second source low half twice)
        asm volatile("v_pk_fma_f32 v[20:21], v[30:31], v[32:33], v[20:21] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[24:25], v[34:35], v[36:37], v[24:25] op_sel_hi:[1,0,1]\n\t"
                     "v_pk_fma_f32 v[22:23], v[38:39], v[40:41], v[22:23] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[26:27], v[42:43], v[28:29], v[26:27] op_sel_hi:[1,0,1]" ::: CLOB);
      // ---- the other instructions of the depthwise-on-codes kernel's inner loop (four independent destinations each) ----
#define FOUR(OPC, D0, D1, D2, D3, SRC) asm volatile(OPC " " D0 ", " SRC "\n\t" OPC " " D1 ", " SRC "\n\t" OPC " " D2 ", " SRC "\n\t" OPC " " D3 ", " SRC ::: CLOB)
      if (OP == 10) FOUR("v_cvt_f32_ubyte1", "v20", "v21", "v22", "v23", "v30");
      if (OP == 11) FOUR("v_cvt_f64_f32", "v[20:21]", "v[22:23]", "v[24:25]", "v[26:27]", "v30");
      if (OP == 12) FOUR("v_mul_f64", "v[20:21]", "v[22:23]", "v[24:25]", "v[26:27]", "v[30:31], v[32:33]");
      if (OP == 13) FOUR("v_cvt_f32_f64", "v20", "v21", "v22", "v23", "v[30:31]");
      if (OP == 14) FOUR("v_cvt_rpi_i32_f32", "v20", "v21", "v22", "v23", "v30");
      if (OP == 15) FOUR("v_med3_f32", "v20", "v21", "v22", "v23", "v30, v31, v32");
      if (OP == 16) FOUR("v_max3_f32", "v20", "v21", "v22", "v23", "v30, v31, v32");
      if (OP == 17) FOUR("v_mul_f32", "v20", "v21", "v22", "v23", "v30, v31");
      if (OP == 18) FOUR("v_lshl_or_b32", "v20", "v21", "v22", "v23", "v30, 8, v32");
      if (OP == 19) FOUR("v_bitop3_b32", "v20", "v21", "v22", "v23", "v30, v31, v32 bitop3:0x36");
      if (OP == 20) FOUR("v_cvt_i32_f32", "v20", "v21", "v22", "v23", "v30");
      if (OP == 21) FOUR("v_cvt_f32_i32", "v20", "v21", "v22", "v23", "v30");
      if (OP == 22) FOUR("v_add_u32", "v20", "v21", "v22", "v23", "v30, v31");
      if (OP == 23) FOUR("v_cvt_pk_u8_f32", "v20", "v21", "v22", "v23", "v30, 1, v32");
      if (OP == 24) FOUR("v_rcp_f32", "v20", "v21", "v22", "v23", "v30");
      if (OP == 25) FOUR("v_fma_f64", "v[20:21]", "v[22:23]", "v[24:25]", "v[26:27]", "v[30:31], v[32:33], v[34:35]");
      if (OP == 26) FOUR("v_perm_b32", "v20", "v21", "v22", "v23", "v30, v31, v32");
      if (OP == 27) FOUR("v_mov_b32_dpp", "v20", "v21", "v22", "v23", "v30 row_shr:1 row_mask:0xf bank_mask:0xf");
      if (OP == 28) FOUR("v_fmac_f32", "v20", "v21", "v22", "v23", "v30, v31");
      if (OP == 29) FOUR("v_rndne_f32", "v20", "v21", "v22", "v23", "v30");
    }
  }
  float v;
  asm volatile("v_add_f32 %0, v20, v24" : "=v"(v)::CLOB);
  if (v == 12345.678f) out[0] = v;
}
template <int OP>
float run(float* d, int grid, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> t;
  for (int it = 0; it < 20; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 1.0001f, reps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    t.push_back(ms * 1000.f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}
template <int OP>
void report(float* d, const char* name) {
  for (int wps : {4}) {                                    // wavefronts per SIMD
    const int grid = 256 * wps;
    const float t1 = run<OP>(d, grid, 64), t2 = run<OP>(d, grid, 192);      // 4096 and 12288 instructions per wavefront
    const double per_simd = (double)wps * (12288 - 4096);
    printf("  %-58s %d wavefront(s) per SIMD: %5.2f cycles per wavefront instruction at 2.4 GHz\n", name, wps,
           (t2 - t1) * 1e3 / per_simd * 2.4);
  }
}
int main() {
  float* d;
  hipMalloc(&d, 4096);
  report<0>(d, "v_fma_f32, three registers");
  report<1>(d, "v_pk_fma_f32, both sources the same pair");
  report<2>(d, "v_pk_fma_f32, three pairs, all in banks 0-1");
  report<3>(d, "v_pk_fma_f32, three pairs, sources in banks 2-3 / 0-1");
  report<4>(d, "v_pk_mul_f32, two pairs");
  report<5>(d, "v_pk_fma_f32, one source a scalar pair");
  report<6>(d, "v_pk_fma_f32, second source broadcast (op_sel_hi)");
  report<10>(d, "v_cvt_f32_ubyte1");
  report<11>(d, "v_cvt_f64_f32");
  report<12>(d, "v_mul_f64");
  report<13>(d, "v_cvt_f32_f64");
  report<14>(d, "v_cvt_rpi_i32_f32");
  report<15>(d, "v_med3_f32");
  report<16>(d, "v_max3_f32");
  report<17>(d, "v_mul_f32");
  report<18>(d, "v_lshl_or_b32");
  report<19>(d, "v_bitop3_b32");
  report<20>(d, "v_cvt_i32_f32");
  report<21>(d, "v_cvt_f32_i32");
  report<22>(d, "v_add_u32");
  report<23>(d, "v_cvt_pk_u8_f32");
  report<24>(d, "v_rcp_f32");
  report<25>(d, "v_fma_f64");
  report<26>(d, "v_perm_b32");
  report<27>(d, "v_mov_b32_dpp row_shr:1");
  report<28>(d, "v_fmac_f32");
  report<29>(d, "v_rndne_f32");
  return 0;
}
