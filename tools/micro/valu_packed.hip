// Packed-fp32 VALU issue cost on gfx950 (development tool): ns one wave64 v_pk_*_f32 instruction occupies a SIMD, next
// to the scalar forms, 8 independent chains per wave at 1, 2, 4 and 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/valu_packed.hip -o tools/micro/bin/valu_packed && tools/micro/bin/valu_packed
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(OP)                                                                                          \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                               \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
               : "v"(c0), "v"(c1))

#define P_ADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define P_MUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define P_FMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define P_ADDSEL(i) "v_pk_add_f32 %" #i ", %" #i ", %8 op_sel:[1,0] op_sel_hi:[0,1]\n"
#define P_MOV(i) "v_pk_mov_b32 %" #i ", %" #i ", %8 op_sel:[1,0]\n"

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  const float f = threadIdx.x * 0.001f + 1.0f;
  v2f a0 = {f, f + 1}, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const v2f c0 = {1.0001f, 0.9999f}, c1 = {0.5f, 0.25f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) REP8(P_ADD); else if (KIND == 1) REP8(P_MUL); else if (KIND == 2) REP8(P_FMA);
      else if (KIND == 3) REP8(P_ADDSEL); else REP8(P_MOV);
    }
  }
  const v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

template <int KIND>
void run(const char *name, float *out) {
  const int iters = 1000;
  printf("%-26s", name);
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
  printf("ns per wave64 packed instruction (two fp32 results per lane) per SIMD\n");
  run<0>("v_pk_add_f32", out); run<1>("v_pk_mul_f32", out); run<2>("v_pk_fma_f32", out);
  run<3>("v_pk_add_f32 op_sel", out); run<4>("v_pk_mov_b32", out);
  return 0;
}
