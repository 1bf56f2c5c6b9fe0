// Shared host/device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mi355x_match.h"

#define MI_WAVE 64

static inline int mi_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MI_OK : (int)e;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share
// an XCD and its private 4 MiB L2).  This bijection hands each XCD one CONTIGUOUS range of logical
// tile ids, so neighbouring tiles -- which share halo rows/columns -- are fetched through the same
// L2 instead of once per XCD.  Placement only changes speed, never results.
__device__ __forceinline__ unsigned xcd_contiguous_id(unsigned bid, unsigned nblocks) {
  const unsigned q = nblocks >> 3, r = nblocks & 7u;
  const unsigned xcd = bid & 7u, k = bid >> 3;
  const unsigned base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return base + k;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
