// LDS bank probe (development tool): time ds_read_b32 with a lane stride of 1..128 dwords, and ds_read_b128 /
// ds_read2_b32 at a 16-byte lane stride.  Build: hipcc --offload-arch=gfx950 -O3 lds_banks_probe.hip -o lds_banks_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64) void probe_b32(int stride, int iters, unsigned long long *out, int *sink) {
  __shared__ int lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = i;
  __syncthreads();
  const int idx = (threadIdx.x * stride) & 16383;
  int acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += *(volatile int *)&lds[(idx + k * 64 * 0 + i) & 16383];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = acc;
}

__global__ __launch_bounds__(64) void probe_b128(int mode, int iters, unsigned long long *out, int *sink) {
  __shared__ int4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = make_int4(i, i, i, i);
  __syncthreads();
  int acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int a = (threadIdx.x + i + k) & 4095;
      if (mode == 0) {
        typedef int i4v __attribute__((ext_vector_type(4)));
        i4v v = *reinterpret_cast<const i4v *>(&lds[a]);
        asm volatile("" : "+v"(v));
        acc += v.x + v.w;
      } else {
        const volatile int *p = (const volatile int *)&lds[a];
        acc += p[1] + p[2];          // two dwords of the 16-byte chunk: ds_read2_b32 at a 16-byte lane stride
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = acc;
}

int main() {
  unsigned long long *d, h;
  int *sink;
  hipMalloc(&d, 8);
  hipMalloc(&sink, 256);
  const int iters = 2000;
  for (int stride : {1, 2, 4, 8, 16, 32, 64, 128, 33}) {
    probe_b32<<<1, 64>>>(stride, iters, d, sink);
    probe_b32<<<1, 64>>>(stride, iters, d, sink);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("ds_read_b32 lane stride %3d dwords: %.2f cycles per read\n", stride, (double)h / (iters * 16.0));
  }
  for (int mode : {0, 1}) {
    probe_b128<<<1, 64>>>(mode, iters, d, sink);
    probe_b128<<<1, 64>>>(mode, iters, d, sink);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%s at a 16-byte lane stride: %.2f cycles per read\n", mode == 0 ? "ds_read_b128" : "2 x ds_read_b32 (middle dwords)",
           (double)h / (iters * 16.0));
  }
  return 0;
}
