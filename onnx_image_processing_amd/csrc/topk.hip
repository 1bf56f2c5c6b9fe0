// K3: per-image top-k of the candidate keys -> keypoints.
// Semantics: reference pytorch_model/utils/keypoint_utils.py:94-115 (torch.topk sorted=True,
// index -> (y, x), invalid -> (-1, -1) / score 0) with the build's tie policy: keys are
// (score bits << 32 | inverted linear index), all distinct, so "descending key" means
// score descending, then linear index ascending -- independent of how K2 ordered them.
//
// Input: the segmented candidate buffer of K2 (one segment per 128x32 tile, count per
// segment).  One 1024-thread workgroup per image; each of its 16 waves walks whole segments.
// Up to 4096 candidates are gathered into LDS and bitonic-sorted directly.  Longer lists
// (plateau images can make every pixel a candidate) first run an exact 8-pass MSB radix select
// for the k-th largest key straight from global memory, then sort only the k survivors.
#include "common.h"

namespace {

constexpr int TK_THREADS = 1024;
constexpr int TK_WAVES = TK_THREADS / 64;
constexpr int TK_MAX = 4096;

__device__ __forceinline__ void bitonic_sort_desc(uint64_t *keys, int npad, int t) {
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < (npad >> 1); i += TK_THREADS) {
        const int pos = 2 * i - (i & (stride - 1));
        const int par = pos + stride;
        const bool desc = ((pos & size) == 0);
        const uint64_t a = keys[pos], b = keys[par];
        if ((a < b) == desc) {
          keys[pos] = b;
          keys[par] = a;
        }
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(TK_THREADS) void topk_kernel(const uint64_t *__restrict__ cand,
                                                          const uint32_t *__restrict__ count, int segments,
                                                          uint32_t seg_cap, int w, int k,
                                                          float *__restrict__ kpts,
                                                          float *__restrict__ kscores) {
  __shared__ uint64_t keys[TK_MAX];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t wsum[TK_WAVES];
  __shared__ uint64_t s_prefix;
  __shared__ uint32_t s_krem, s_fill;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int img = blockIdx.x;
  const uint64_t *list = cand + (size_t)img * segments * seg_cap;
  const uint32_t *cnt = count + (size_t)img * segments;

  // total number of candidates of this image
  uint32_t part = 0;
  for (int s = t; s < segments; s += TK_THREADS) part += min(cnt[s], seg_cap);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) wsum[wave] = part;
  if (t == 0) s_fill = 0u;
  __syncthreads();
  uint32_t n = 0;
#pragma unroll
  for (int q = 0; q < TK_WAVES; ++q) n += wsum[q];
  int nsel;

  if (n <= (uint32_t)TK_MAX) {
    // gather every segment into LDS (order irrelevant: sorted next); one LDS atomic per segment
    for (int s = wave; s < segments; s += TK_WAVES) {
      const uint32_t c = min(cnt[s], seg_cap);
      if (c == 0u) continue;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&s_fill, c);
      base = __shfl(base, 0, 64);
      const uint64_t *seg = list + (size_t)s * seg_cap;
      for (uint32_t i = lane; i < c; i += 64) keys[base + i] = seg[i];
    }
    int npad = 2;
    while (npad < (int)n) npad <<= 1;
    __syncthreads();
    for (int i = (int)n + t; i < npad; i += TK_THREADS) keys[i] = 0ull;
    __syncthreads();
    bitonic_sort_desc(keys, npad, t);
    nsel = (int)n < k ? (int)n : k;
  } else {
    // exact k-th largest key by MSB radix select (8 digits of 8 bits); n > TK_MAX >= k here
    if (t == 0) { s_prefix = 0ull; s_krem = (uint32_t)k; }
    uint64_t mask = 0ull;
    for (int shift = 56; shift >= 0; shift -= 8) {
      if (t < 256) hist[t] = 0u;
      __syncthreads();
      const uint64_t prefix = s_prefix;
      for (int s = wave; s < segments; s += TK_WAVES) {
        const uint32_t c = min(cnt[s], seg_cap);
        const uint64_t *seg = list + (size_t)s * seg_cap;
        for (uint32_t base = 0; base < c; base += 64) {
          const uint32_t i = base + lane;
          bool act = false;
          uint32_t digit = 0;
          if (i < c) {
            const uint64_t key = seg[i];
            act = ((key & mask) == prefix);
            digit = (uint32_t)(key >> shift) & 255u;
          }
          // wave-level aggregation: the leading digits are shared by almost every key
          const unsigned long long am = __ballot(act);
          if (am) {
            const int leader = __ffsll((long long)am) - 1;
            const uint32_t d0 = __shfl(digit, leader, 64);
            const unsigned long long same = __ballot(act && digit == d0);
            if (lane == leader) atomicAdd(&hist[d0], (uint32_t)__popcll(same));
            if (act && digit != d0) atomicAdd(&hist[digit], 1u);
          }
        }
      }
      __syncthreads();
      if (t < 64) {
        // lane l owns bins 255-4l .. 252-4l (descending); find the bin holding the krem-th key
        const uint32_t krem = s_krem;
        uint32_t c[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c[q] = hist[255 - 4 * t - q]; tot += c[q]; }
        uint32_t incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t v = __shfl_up(incl, o, 64);
          if (t >= o) incl += v;
        }
        uint32_t above = incl - tot;
        if (above < krem && krem <= incl) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (krem <= above + c[q]) {
              s_prefix = prefix | ((uint64_t)(255 - 4 * t - q) << shift);
              s_krem = krem - above;
              break;
            }
            above += c[q];
          }
        }
      }
      mask |= (0xFFull << shift);
      __syncthreads();
    }
    const uint64_t kth = s_prefix;   // keys are distinct: exactly k keys are >= kth
    for (int s = wave; s < segments; s += TK_WAVES) {
      const uint32_t c = min(cnt[s], seg_cap);
      const uint64_t *seg = list + (size_t)s * seg_cap;
      for (uint32_t i = lane; i < c; i += 64) {
        const uint64_t key = seg[i];
        if (key >= kth) {
          const uint32_t slot = atomicAdd(&s_fill, 1u);
          if (slot < (uint32_t)TK_MAX) keys[slot] = key;
        }
      }
    }
    __syncthreads();
    nsel = (int)(s_fill < (uint32_t)k ? s_fill : (uint32_t)k);
    int npad = 2;
    while (npad < k) npad <<= 1;
    for (int i = nsel + t; i < npad; i += TK_THREADS) keys[i] = 0ull;
    __syncthreads();
    bitonic_sort_desc(keys, npad, t);
  }

  for (int j = t; j < k; j += TK_THREADS) {
    float y = -1.0f, x = -1.0f, s = 0.0f;
    if (j < nsel) {
      const uint64_t key = keys[j];
      const uint32_t lin = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
      y = (float)(lin / (uint32_t)w);
      x = (float)(lin % (uint32_t)w);
      s = __uint_as_float((uint32_t)(key >> 32));
    }
    kpts[((size_t)img * k + j) * 2 + 0] = y;
    kpts[((size_t)img * k + j) * 2 + 1] = x;
    kscores[(size_t)img * k + j] = s;
  }
}

}  // namespace

extern "C" int mi_topk_keypoints(const uint64_t *cand, const uint32_t *count, int segments,
                                 int segment_capacity, int n, int w, int k, float *keypoints, float *kscores,
                                 mi_stream_t stream) {
  if (!cand || !count || !keypoints || !kscores) return MI_E_NULL;
  if (n <= 0 || w <= 0 || segments <= 0) return MI_E_SHAPE;
  if (k <= 0 || k > TK_MAX) return MI_E_PARAM;
  if (segment_capacity <= 0) return MI_E_CAPACITY;
  hipLaunchKernelGGL(topk_kernel, dim3(n), dim3(TK_THREADS), 0, (hipStream_t)stream, cand, count, segments,
                     (uint32_t)segment_capacity, w, k, keypoints, kscores);
  return mi_launch_status();
}
