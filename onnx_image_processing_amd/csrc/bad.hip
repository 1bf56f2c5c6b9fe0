// K4: sparse BAD (Box Average Difference) descriptors at keypoints.
// Semantics: reference pytorch_model/descriptor/bad.py:436-576, non-oriented branch,
// sampling_mode="nearest".  The reference builds a dense 8-channel 15x15 box-mean bank over
// the whole image and gathers 2*8*K*P samples from it; here nothing dense is built.
//
// One wave (64 lanes) per keypoint.  Every box of every pair lies inside a 34x34 window of
// the replicate-extended image around the keypoint (see window proof below), so the wave
// stages that window, turns it into an fp64 summed-area table in LDS (exact for integer-valued
// images, ~1e-13 otherwise) and evaluates each box with four 8-byte LDS reads.  Lane l owns
// pairs l, l+64, ...; a wave ballot turns 64 sign tests into one 64-bit word of the packed
// descriptor, and its popcount gives the L2 norm of the bit vector exactly.
//
// Window proof (non-oriented): table offsets satisfy -16 <= o - r, o + r <= 15 (boxes stay
// inside the 32x32 patch).  With f = floor(ky), the unclamped centre c = nearbyint(ky + o)
// lies in [f + o, f + o + 1]; clamping c into the image and then taking rows c-r..c+r of the
// replicate-extended image never leaves [f - 16, f + 17].
#include "common.h"

#include <math.h>

namespace {

constexpr int WIN = 34;          // window edge
constexpr int WOFF = 16;         // window origin = floor(k) - WOFF
constexpr int SP = WIN + 1;      // SAT edge (leading zero row/column)

// ATen grid_sampler semantics used by bad.py:518-556 (align_corners=True, padding "border",
// mode "nearest"): normalise with fp32(2/(size-1+1e-8)), un-normalise, clip, round half even.
__device__ __forceinline__ int nearest_centre(float pos, float scale, int size) {
  const float g = pos * scale - 1.0f;
  float x = ((g + 1.0f) / 2.0f) * (float)(size - 1);
  x = fminf(fmaxf(x, 0.0f), (float)(size - 1));
  return (int)nearbyintf(x);
}

__global__ __launch_bounds__(64) void sparse_bad_kernel(const float *__restrict__ image, int h, int w,
                                                        const float *__restrict__ kpts, int k,
                                                        const uint32_t *__restrict__ geom,
                                                        const float *__restrict__ thr, int num_pairs,
                                                        int mode, float temperature, int normalize,
                                                        float scale_y, float scale_x,
                                                        float *__restrict__ desc,
                                                        uint32_t *__restrict__ bits) {
  __shared__ double sat[SP * SP];
  __shared__ float vals[1024];               // un-normalised descriptor row (num_pairs <= 1024)
  const int lane = threadIdx.x;
  const int kp = blockIdx.x;                 // keypoint index within the image
  const int img = blockIdx.y;
  const float *im = image + (size_t)img * h * w;
  const float ky_raw = kpts[((size_t)img * k + kp) * 2 + 0];
  const float kx_raw = kpts[((size_t)img * k + kp) * 2 + 1];
  const bool valid = ky_raw >= 0.0f;                                   // bad.py:461
  const float ky = fminf(fmaxf(ky_raw, 0.0f), (float)(h - 1));         // bad.py:464-465
  const float kx = fminf(fmaxf(kx_raw, 0.0f), (float)(w - 1));
  const int oy = (int)floorf(ky) - WOFF, ox = (int)floorf(kx) - WOFF;

  // ---- summed-area table of the replicate-extended window, sat[r+1][c+1] = sum of rows<=r, cols<=c
  // Lane c < 34 owns window column c: its 34 row loads are all issued before any use (34 loads in
  // flight per lane, each wave-instruction one 136-byte row segment), the column prefix runs in
  // registers, then lane r < 34 takes row r for the horizontal prefix (stride 35 doubles between
  // lanes: conflict-free for ds_read_b64).
  const int groups = num_pairs / 64;
  uint32_t qg[8];
  float tg[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    qg[g] = (g < groups) ? geom[g * 64 + lane] : 0u;
    tg[g] = (g < groups) ? thr[g * 64 + lane] : 0.0f;
  }
  for (int i = lane; i < SP; i += 64) { sat[i] = 0.0; sat[i * SP] = 0.0; }
  if (lane < WIN) {
    const int gx = clampi(ox + lane, 0, w - 1);
    float px[WIN];
#pragma unroll
    for (int r = 0; r < WIN; ++r) px[r] = im[(size_t)clampi(oy + r, 0, h - 1) * w + gx];
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < WIN; ++r) {
      acc += (double)px[r];
      sat[(r + 1) * SP + (lane + 1)] = acc;
    }
  }
  __syncthreads();
  if (lane < WIN) {
    double *row = sat + (lane + 1) * SP + 1;
    double v[WIN];
#pragma unroll
    for (int c = 0; c < WIN; ++c) v[c] = row[c];
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < WIN; ++c) { acc += v[c]; row[c] = acc; }
  }
  __syncthreads();

  const size_t drow = ((size_t)img * k + kp) * (size_t)num_pairs;
  const int words = num_pairs / 32;
  uint32_t *brow = bits ? bits + ((size_t)img * k + kp) * words : nullptr;
  int pop = 0;
  float sumsq = 0.0f;

  // one group = 64 pairs (one per lane).  Geometry/threshold words of the first 8 groups were
  // fetched before the SAT was built (their latency hides behind it).
  auto eval_group = [&](int g, uint32_t q, float thr_p) {
    const int p = g * 64 + lane;
    const int x1 = (int)(q & 31u) - 16, x2 = (int)((q >> 5) & 31u) - 16;
    const int y1 = (int)((q >> 10) & 31u) - 16, y2 = (int)((q >> 15) & 31u) - 16;
    const int r = (int)((q >> 20) & 15u);
    const int c1y = nearest_centre(ky + (float)y1, scale_y, h) - oy;
    const int c1x = nearest_centre(kx + (float)x1, scale_x, w) - ox;
    const int c2y = nearest_centre(ky + (float)y2, scale_y, h) - oy;
    const int c2x = nearest_centre(kx + (float)x2, scale_x, w) - ox;
    // box rows [c-r, c+r] of the window -> SAT rows c-r and c+r+1 (clamped: cannot trigger for
    // table geometry; keeps arbitrary user tables memory-safe)
    const int a1 = clampi(c1y - r, 0, WIN), b1 = clampi(c1y + r + 1, 0, WIN);
    const int l1 = clampi(c1x - r, 0, WIN), r1 = clampi(c1x + r + 1, 0, WIN);
    const int a2 = clampi(c2y - r, 0, WIN), b2 = clampi(c2y + r + 1, 0, WIN);
    const int l2 = clampi(c2x - r, 0, WIN), r2 = clampi(c2x + r + 1, 0, WIN);
    const double s1 = (sat[b1 * SP + r1] - sat[a1 * SP + r1]) - (sat[b1 * SP + l1] - sat[a1 * SP + l1]);
    const double s2 = (sat[b2 * SP + r2] - sat[a2 * SP + r2]) - (sat[b2 * SP + l2] - sat[a2 * SP + l2]);
    const double area = (double)((2 * r + 1) * (2 * r + 1));
    const double t = (double)thr_p;
    if (mode == MI_BAD_HARD) {
      // bit = (mean1 - mean2 - t <= 0)  <=>  s1 - s2 <= t * area   (t*area exact in fp64)
      const bool bit = valid && ((s1 - s2) <= t * area);                // bad.py:567,570
      const unsigned long long word = __ballot(bit);
      pop += (int)__popcll(word);
      if (brow && lane == 0) {
        brow[2 * g] = (uint32_t)word;
        brow[2 * g + 1] = (uint32_t)(word >> 32);
      }
      if (desc) vals[p] = bit ? 1.0f : 0.0f;
    } else {
      const float c = (float)((s1 - s2) / area - t);                    // bad.py:559
      float v = c;
      if (mode == MI_BAD_SOFT) v = 1.0f / (1.0f + expf(c * temperature));  // sigmoid(-c*T), bad.py:565
      v = valid ? v : 0.0f;
      sumsq += v * v;
      vals[p] = v;
    }
  };
#pragma unroll
  for (int g = 0; g < 8; ++g)
    if (g < groups) eval_group(g, qg[g], tg[g]);
#pragma unroll 1
  for (int g = 8; g < groups; ++g) eval_group(g, geom[g * 64 + lane], thr[g * 64 + lane]);

  if (!desc) return;
  float inv = 1.0f;
  if (normalize) {                                                      // F.normalize(p=2, eps=1e-12), bad.py:573
    const float ss = (mode == MI_BAD_HARD) ? (float)pop : wave_sum(sumsq);
    inv = fmaxf(sqrtf(ss), 1e-12f);
  }
  // each lane re-reads only what it wrote itself (p = g*64 + lane): no barrier needed
  for (int g = 0; g < groups; ++g) {
    const float v = vals[g * 64 + lane];
    desc[drow + g * 64 + lane] = normalize ? v / inv : v;
  }
}

}  // namespace

extern "C" int mi_sparse_bad(const float *image, int n, int h, int w, const float *keypoints, int k,
                             const uint32_t *pair_geom, const float *pair_thr, int num_pairs, int mode,
                             float temperature, int normalize, float *desc, uint32_t *bits,
                             mi_stream_t stream) {
  if (!image || !keypoints || !pair_geom || !pair_thr) return MI_E_NULL;
  if (!desc && !bits) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0 || k <= 0) return MI_E_SHAPE;
  if (num_pairs <= 0 || num_pairs % 64 != 0 || num_pairs > 1024) return MI_E_PARAM;
  if (mode != MI_BAD_RAW && mode != MI_BAD_SOFT && mode != MI_BAD_HARD) return MI_E_PARAM;
  if (bits && mode != MI_BAD_HARD) return MI_E_PARAM;
  if (n > 65535) return MI_E_SHAPE;
  // bad.py:469-470: python double 2/(size-1+1e-8), multiplied into an fp32 tensor
  const float scale_y = (float)(2.0 / ((double)(h - 1) + 1e-8));
  const float scale_x = (float)(2.0 / ((double)(w - 1) + 1e-8));
  hipLaunchKernelGGL(sparse_bad_kernel, dim3(k, n), dim3(64), 0, (hipStream_t)stream, image, h, w, keypoints,
                     k, pair_geom, pair_thr, num_pairs, mode, temperature, normalize, scale_y, scale_x, desc,
                     bits);
  return mi_launch_status();
}
