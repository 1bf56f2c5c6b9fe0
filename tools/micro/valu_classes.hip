// VALU issue cost by instruction class on gfx950 (development tool): cycles one wave64 instruction occupies a SIMD,
// measured with 8 independent register chains per wave at 1, 2, 4 and 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/valu_classes.hip -o /tmp/valu_classes && /tmp/valu_classes
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(OP)                                                                                          \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                               \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
               : "v"(c0), "v"(c1))

#define I_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define I_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define I_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_LSHR(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define I_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define I_CMP(i) "v_cmp_gt_f32 vcc, %" #i ", %8\n"
#define I_CVT(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define I_CVTSDWA(i) "v_cvt_f32_u32_sdwa %" #i ", %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define I_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define I_LOG(i) "v_log_f32 %" #i ", %" #i "\n"
#define I_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define I_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define I_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define I_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_DPPW(i) "v_mov_b32_dpp %" #i ", %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADDDPP(i) "v_add_f32_dpp %" #i ", %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define I_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define I_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 8\n"
#define I_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 8\n"
#define I_CVTU8(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define I_MIX(i) "v_fma_mix_f32 %" #i ", %" #i ", %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define I_CVTH(i) "v_cvt_f32_f16 %" #i ", %" #i "\n"
#define I_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %10, %10\n"

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  float a0 = threadIdx.x * 0.001f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float c0 = 1.0001f, c1 = 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) REP8(I_ADD); else if (KIND == 1) REP8(I_MUL); else if (KIND == 2) REP8(I_FMA); else if (KIND == 3) REP8(I_MAX);
      else if (KIND == 4) REP8(I_ADDU); else if (KIND == 5) REP8(I_AND); else if (KIND == 6) REP8(I_LSHR); else if (KIND == 7) REP8(I_MOV);
      else if (KIND == 8) REP8(I_CND); else if (KIND == 9) REP8(I_CMP); else if (KIND == 10) REP8(I_CVT); else if (KIND == 11) REP8(I_CVTSDWA);
      else if (KIND == 12) REP8(I_EXP); else if (KIND == 13) REP8(I_LOG); else if (KIND == 14) REP8(I_RCP); else if (KIND == 15) REP8(I_RSQ);
      else if (KIND == 16) REP8(I_SQRT); else if (KIND == 17) REP8(I_DPP); else if (KIND == 18) REP8(I_DPPW); else if (KIND == 19) REP8(I_ADDDPP);
      else if (KIND == 20) REP8(I_PERM); else if (KIND == 21) REP8(I_MAX3); else if (KIND == 22) REP8(I_MAD24); else if (KIND == 23) REP8(I_MULLO);
      else if (KIND == 24) REP8(I_BFE); else if (KIND == 25) REP8(I_ALIGN); else if (KIND == 26) REP8(I_CVTU8);
      else if (KIND == 27) REP8(I_MIX); else REP8(I_CVTH);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char *name, float *out) {
  const int iters = 1000;
  printf("%-22s", name);
  for (int bpc = 1; bpc <= 8; bpc *= 2) {
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    const int grid = 256 * bpc;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(s);
      hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, iters);
      hipEventRecord(e);
      hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      if (ms < best) best = ms;
    }
    const double instr_per_simd = (double)iters * 8 * 8 * bpc;   // one wave per SIMD per block
    printf("  %d w/SIMD: %5.2f ns", bpc, best * 1e6 / instr_per_simd);
  }
  printf("\n");
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  printf("ns per wave64 instruction per SIMD (multiply by the shader clock in GHz for cycles)\n");
  run<0>("v_add_f32", out); run<1>("v_mul_f32", out); run<2>("v_fma_f32", out); run<3>("v_max_f32", out);
  run<21>("v_max3_f32", out); run<4>("v_add_u32", out); run<5>("v_and_b32", out); run<6>("v_lshrrev_b32", out);
  run<24>("v_bfe_u32", out); run<25>("v_alignbit_b32", out); run<20>("v_perm_b32", out); run<22>("v_mad_u32_u24", out);
  run<23>("v_mul_lo_u32", out); run<7>("v_mov_b32", out); run<8>("v_cndmask_b32", out); run<9>("v_cmp_gt_f32", out);
  run<10>("v_cvt_f32_u32", out); run<11>("v_cvt_f32_u32 sdwa", out); run<26>("v_cvt_f32_ubyte1", out);
  run<12>("v_exp_f32", out); run<13>("v_log_f32", out); run<14>("v_rcp_f32", out); run<15>("v_rsq_f32", out); run<16>("v_sqrt_f32", out);
  run<27>("v_fma_mix_f32 (f16 src)", out); run<28>("v_cvt_f32_f16", out);
  run<17>("v_mov_dpp quad_perm", out); run<18>("v_mov_dpp wave_shr", out); run<19>("v_add_f32 dpp", out);
  return 0;
}
