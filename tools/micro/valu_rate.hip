// VALU issue-rate microbenchmark (development tool): how many cycles does one wave64
// v_add_f32 / v_pk_add_f32 / v_fma_f32 occupy a gfx950 SIMD at 1..8 waves per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pc = {1.0f, 2.0f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) {
        asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                     "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1.0f));
      } else if (KIND == 1) {
        asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                     "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));
      } else {
        asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                     "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1.0001f));
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
  const int iters = 2000;
  const char *names[3] = {"v_add_f32", "v_pk_add_f32", "v_fma_f32"};
  for (int kind = 0; kind < 3; ++kind)
    for (int blocks_per_cu = 1; blocks_per_cu <= 8; blocks_per_cu *= 2) {
      hipEvent_t s, e;
      hipEventCreate(&s); hipEventCreate(&e);
      const int grid = 256 * blocks_per_cu;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(s);
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters);
        if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters);
        if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e);
        hipEventSynchronize(e);
      }
      float ms; hipEventElapsedTime(&ms, s, e);
      const double instr_per_simd = (double)iters * 16 * 8 * blocks_per_cu;  // one wave per SIMD per block
      printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", names[kind],
             blocks_per_cu, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
  return 0;
}
