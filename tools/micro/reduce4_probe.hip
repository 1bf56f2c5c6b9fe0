// Check wave_max4 / wave_sum4 (csrc/common.h) against a host reduction (development tool).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -o reduce4_probe tools/micro/reduce4_probe.hip
#include "../../onnx_image_processing_amd/csrc/common.h"
#include <cstdio>
__global__ void k(const float *in, float *o) {
  const int lane = threadIdx.x;
  float v[4], w[4];
  for (int r = 0; r < 4; ++r) { v[r] = in[r * 64 + lane]; w[r] = v[r]; }
  wave_max4(v);
  wave_sum4(w);
  if (lane == 37) for (int r = 0; r < 4; ++r) { o[r] = v[r]; o[4 + r] = w[r]; }
}
int main() {
  float h[256], *d, *o, ho[8];
  for (int i = 0; i < 256; ++i) h[i] = (float)((i * 37 + 11) % 101) - 50.0f + (i / 64) * 1000.0f;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) {
    float mx = -1e30f, s = 0.f;
    for (int i = 0; i < 64; ++i) { mx = h[r * 64 + i] > mx ? h[r * 64 + i] : mx; s += h[r * 64 + i]; }
    printf("row %d: max %g (want %g)  sum %g (want %g)\n", r, ho[r], mx, ho[4 + r], s);
  }
  return 0;
}
