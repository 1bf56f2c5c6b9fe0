// Probe of v_permlane32_swap / v_permlane16_swap lane semantics on gfx950 (development tool).
//   hipcc --offload-arch=gfx950 -O2 -o permlane_probe tools/micro/permlane_probe.hip && ./permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
  const unsigned lane = threadIdx.x;
  unsigned a = lane, b = 100 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[lane] = r[0];
  o[64 + lane] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[128 + lane] = q[0];
  o[192 + lane] = q[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
  for (int v = 0; v < 4; ++v) {
    printf("%s:", names[v]);
    for (int i = 0; i < 64; i += 8) printf(" [%u..%u]", h[v * 64 + i], h[v * 64 + i + 7]);
    printf("\n");
  }
  return 0;
}
