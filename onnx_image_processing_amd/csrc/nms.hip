// K2: square-window NMS (max-pool semantics) + candidate compaction.
// Semantics: reference pytorch_model/utils/keypoint_utils.py:12-44 (mask) and :71-92 (masking,
// border, threshold).  HBM traffic: 4 B/pixel read; candidates are ~1 % of pixels.
//
// One 256-thread workgroup owns a 128x32 tile: the score tile (+r halo, -inf outside the
// image) is staged in LDS, a separable max (row pass into a second LDS plane, column pass in
// registers) gives the (2r+1)^2 window maximum.  Survivors are packed into 64-bit keys
// (score bits high, inverted linear index low) and appended to the per-image candidate list
// with one atomic per wave (ballot + prefix popcount), so the later top-k sort has a total
// order that does not depend on the append order.
#include "common.h"

#include <math.h>

namespace {

constexpr int NT_W = 128, NT_H = 32;

__device__ __forceinline__ void emit_candidate(bool keep, float m, uint32_t lin, uint64_t *cand,
                                               uint32_t *count, uint32_t capacity) {
  const unsigned long long ballot = __ballot(keep);
  if (ballot == 0ull) return;
  const int lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(count, (uint32_t)__popcll(ballot));
  base = __shfl(base, 0, 64);
  if (keep) {
    const uint32_t slot = base + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
    if (slot < capacity)
      cand[slot] = ((uint64_t)__float_as_uint(m) << 32) | (uint64_t)(0xFFFFFFFFu - lin);
  }
}

// MODE 0: write the float mask.  MODE 1: border + threshold + compaction (mask not stored).
template <int MODE>
__global__ __launch_bounds__(256) void nms_kernel(const float *__restrict__ score, int h, int w, int r,
                                                  int tiles_x, int tiles_y, float *__restrict__ mask,
                                                  float thr_eff, int margin, uint64_t *__restrict__ cand,
                                                  uint32_t *__restrict__ count, uint32_t capacity) {
  extern __shared__ float lds[];
  const int sw = NT_W + 2 * r;       // staged width
  const int sh = NT_H + 2 * r;       // staged height
  float *tile = lds;                 // [sh][sw]
  float *rmax = lds + sh * sw;       // [sh][NT_W] horizontal window maxima

  const int t = threadIdx.x;
  int bid = blockIdx.x;
  const int tx_tile = bid % tiles_x;
  bid /= tiles_x;
  const int ty_tile = bid % tiles_y;
  const int img = bid / tiles_y;
  const int x0 = tx_tile * NT_W, y0 = ty_tile * NT_H;
  const float *sc = score + (size_t)img * h * w;

  for (int i = t; i < sh * sw; i += 256) {
    const int rr = i / sw, cc = i - rr * sw;
    const int gy = y0 - r + rr, gx = x0 - r + cc;
    tile[i] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? sc[(size_t)gy * w + gx] : -INFINITY;
  }
  __syncthreads();
  for (int i = t; i < sh * NT_W; i += 256) {
    const int rr = i / NT_W, cc = i - rr * NT_W;
    const float *p = tile + rr * sw + cc;
    float m = p[0];
    for (int d = 1; d <= 2 * r; ++d) m = fmaxf(m, p[d]);
    rmax[i] = m;
  }
  __syncthreads();

  const int cx = t & (NT_W - 1);     // column inside the tile
  const int half = t >> 7;           // rows [half*16, half*16+16)
  const int gx = x0 + cx;
  for (int k = 0; k < NT_H / 2; ++k) {
    const int ly = half * (NT_H / 2) + k;
    const int gy = y0 + ly;
    const bool inside = (gx < w) && (gy < h);
    float m = -INFINITY, s = 0.f;
    if (inside) {
      const float *p = rmax + ly * NT_W + cx;
      m = p[0];
      for (int d = 1; d <= 2 * r; ++d) m = fmaxf(m, p[d * NT_W]);
      s = tile[(ly + r) * sw + cx + r];
    }
    const bool is_max = inside && (s >= (m - 1e-7f));          // keypoint_utils.py:43
    if (MODE == 0) {
      if (inside) mask[((size_t)img * h + gy) * w + gx] = is_max ? 1.0f : 0.0f;
    } else {
      const bool in_border = (margin <= 0) || (gy >= margin && gy < h - margin && gx >= margin && gx < w - margin);
      const bool keep = is_max && in_border && (s > thr_eff);
      emit_candidate(keep, s, (uint32_t)(gy * w + gx), cand + (size_t)img * capacity, count + img, capacity);
    }
  }
}

// Explicit-mask form (the reference's separate select_topk_keypoints call): elementwise.
__global__ __launch_bounds__(256) void select_kernel(const float *__restrict__ score,
                                                     const float *__restrict__ mask, int h, int w,
                                                     float thr, float thr_eff, int margin,
                                                     uint64_t *__restrict__ cand, uint32_t *__restrict__ count,
                                                     uint32_t capacity) {
  const int img = blockIdx.y;
  const int hw = h * w;
  for (int base = blockIdx.x * 256; base < hw; base += gridDim.x * 256) {
    const int i = base + threadIdx.x;
    bool keep = false;
    float m = 0.f;
    if (i < hw) {
      const int gy = i / w, gx = i - gy * w;
      const float border =
          (margin <= 0 || (gy >= margin && gy < h - margin && gx >= margin && gx < w - margin)) ? 1.0f : 0.0f;
      m = score[(size_t)img * hw + i] * mask[(size_t)img * hw + i];
      if (margin > 0) m = m * border;
      keep = (m > thr) && (m > thr_eff);                        // keypoint_utils.py:88-92 then "> 0" at :108
    }
    emit_candidate(keep, m, (uint32_t)i, cand + (size_t)img * capacity, count + img, capacity);
  }
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return MI_OK;
  if (bytes > 160 * 1024) return MI_E_PARAM;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? MI_OK : (int)e;
}

int check_common(const void *a, const void *b, int n, int h, int w) {
  if (!a || !b) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0 || (long long)h * w > 0x7fffffffLL) return MI_E_SHAPE;
  return MI_OK;
}

}  // namespace

extern "C" int mi_nms_mask(const float *score, int n, int h, int w, int radius, float *mask,
                           mi_stream_t stream) {
  int e = check_common(score, mask, n, h, w);
  if (e) return e;
  if (radius < 0 || radius > 24) return MI_E_PARAM;
  const int tiles_x = ceil_div(w, NT_W), tiles_y = ceil_div(h, NT_H);
  const size_t lds = ((size_t)(NT_H + 2 * radius) * (NT_W + 2 * radius) + (size_t)(NT_H + 2 * radius) * NT_W) * 4;
  if ((e = allow_lds(nms_kernel<0>, lds)) != MI_OK) return e;
  hipLaunchKernelGGL(nms_kernel<0>, dim3((unsigned)(n * tiles_x * tiles_y)), dim3(256), lds,
                     (hipStream_t)stream, score, h, w, radius, tiles_x, tiles_y, mask, 0.f, 0, nullptr,
                     nullptr, 0u);
  return mi_launch_status();
}

extern "C" int mi_nms_candidates(const float *score, int n, int h, int w, int radius, float score_threshold,
                                 int border_margin, uint64_t *cand, uint32_t *count, uint32_t capacity,
                                 mi_stream_t stream) {
  int e = check_common(score, cand, n, h, w);
  if (e) return e;
  if (!count) return MI_E_NULL;
  if (radius < 0 || radius > 24) return MI_E_PARAM;
  if (capacity == 0) return MI_E_CAPACITY;
  const int tiles_x = ceil_div(w, NT_W), tiles_y = ceil_div(h, NT_H);
  const size_t lds = ((size_t)(NT_H + 2 * radius) * (NT_W + 2 * radius) + (size_t)(NT_H + 2 * radius) * NT_W) * 4;
  const float thr_eff = score_threshold > 0.f ? score_threshold : 0.f;
  if ((e = allow_lds(nms_kernel<1>, lds)) != MI_OK) return e;
  hipLaunchKernelGGL(nms_kernel<1>, dim3((unsigned)(n * tiles_x * tiles_y)), dim3(256), lds,
                     (hipStream_t)stream, score, h, w, radius, tiles_x, tiles_y, nullptr, thr_eff,
                     border_margin, cand, count, capacity);
  return mi_launch_status();
}

extern "C" int mi_select_candidates(const float *score, const float *mask, int n, int h, int w,
                                    float score_threshold, int border_margin, uint64_t *cand,
                                    uint32_t *count, uint32_t capacity, mi_stream_t stream) {
  int e = check_common(score, mask, n, h, w);
  if (e) return e;
  if (!cand || !count) return MI_E_NULL;
  if (capacity == 0) return MI_E_CAPACITY;
  const int gx = ceil_div(h * w, 256) < 1024 ? ceil_div(h * w, 256) : 1024;
  hipLaunchKernelGGL(select_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, score, mask, h, w,
                     score_threshold, 0.f, border_margin, cand, count, capacity);
  return mi_launch_status();
}
