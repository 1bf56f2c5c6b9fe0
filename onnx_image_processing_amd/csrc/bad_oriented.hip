// K4 (oriented): sparse BAD descriptors with the pair offsets rotated by the keypoint's angle.
// Semantics: reference pytorch_model/descriptor/bad.py:487-517 (oriented branch) + :518-574:
//   theta  = grid_sample(orientation, keypoint, nearest)            (or a per-keypoint angle)
//   dy = ox*sin + oy*cos ; dx = ox*cos - oy*sin   (ox, oy = table offset - 16, fp32 op by op)
//   pos = keypoint + (dy, dx) ; centre = grid_sample "nearest" arithmetic ; box mean over the
//   replicate-extended image ; response = mean1 - mean2 - thr ; raw / sigmoid / hard bit.
// Rotated offsets reach sqrt(15^2+15^2) = 21.2 px; with the box radius (<= 7) and the rounding
// of the centre every box lies in the 60x60 window [f-29, f+30] around f = floor(keypoint)
// (same clamping argument as the non-oriented kernel); the extra neighbour of the bilinear mode
// (floor(pos) + 1) still lies inside it.  One wave per keypoint, fp64 summed-area table in LDS
// (exact for integer images): lane c owns window column c, then row c.  sampling_mode="bilinear"
// interpolates the box means of the four neighbouring centres with ATen's grid_sampler weights; the
// non-oriented bilinear case is this kernel with angle 0 (cos = 1, sin = 0: the offsets unchanged).
#include "common.h"

#include <math.h>

namespace {

// Window edge OW (template argument): 60 covers any table of the 32x32 patch frame (rotated offsets reach
// sqrt(15^2+15^2) = 21.2 px, radius <= 7, bilinear neighbour + 1); 48 covers tables whose largest |offset| + radius
// is <= 22.5 px in the nearest mode -- the caller states that bound (max_reach) -- [f-23, f+24] then holds every box:
// frac(k) + reach + 0.5 (rounding of the centre) < 24.  Both reference tables reach 22.22 px.  A 48-pixel window is
// 9.6 instead of 14.9 KB of LDS per keypoint and 48 instead of 60 loads per lane: more windows in flight per CU, which
// is what bounds this kernel.

__device__ __forceinline__ int nearest_centre_o(float pos, float scale, int size) {
  const float g = pos * scale - 1.0f;
  float x = ((g + 1.0f) / 2.0f) * (float)(size - 1);
  x = fminf(fmaxf(x, 0.0f), (float)(size - 1));
  return (int)nearbyintf(x);
}

// SAT = int: exact for uint8-valued windows (checked per keypoint; others are flagged status = 0 and left
// to the SAT = double instance), half the LDS, so twice the resident keypoints per CU.
// DESC = false (packed bits only, what the matchers ask for): no float staging array -- 14.9 instead of 19 KB of LDS per
// keypoint, i.e. 10 instead of 8 resident keypoints per CU; the kernel is bound by how many windows are in flight.
template <typename SAT, bool DESC, int OW>
__global__ __launch_bounds__(64) void bad_oriented_kernel(const float *__restrict__ image, int h, int w,
                                                          const float *__restrict__ kpts, int k,
                                                          const float *__restrict__ theta_map,
                                                          const float *__restrict__ theta_kp,
                                                          const uint32_t *__restrict__ geom,
                                                          const float *__restrict__ thr, int num_pairs, int mode,
                                                          float temperature, int normalize, float scale_y,
                                                          float scale_x, int bilinear,
                                                          float *__restrict__ desc,
                                                          uint32_t *__restrict__ bits,
                                                          uint8_t *__restrict__ status) {
  constexpr bool INT = sizeof(SAT) == 4;
  constexpr int OOFF = OW / 2 - 1;  // window origin = floor(k) - OOFF
  constexpr int OSP = OW + 1;       // SAT edge
  constexpr int RB = OW / 4;        // rows / columns per batch of the in-LDS prefix pass
  __shared__ SAT sat[OSP * OSP];
  __shared__ float vals[DESC ? 1024 : 1];
  const int lane = threadIdx.x;
  const int flat = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  if (!INT && status && status[flat]) return;                          // the integer instance did this keypoint
  const int img = flat / k;
  const float *im = image + (size_t)img * h * w;
  const float ky_raw = kpts[(size_t)flat * 2 + 0];
  const float kx_raw = kpts[(size_t)flat * 2 + 1];
  const bool valid = ky_raw >= 0.0f;                                   // bad.py:461
  const float ky = fminf(fmaxf(ky_raw, 0.0f), (float)(h - 1));         // bad.py:464-465
  const float kx = fminf(fmaxf(kx_raw, 0.0f), (float)(w - 1));
  float theta;
  if (theta_map) {                                                     // bad.py:490-500
    const int cy = nearest_centre_o(ky, scale_y, h), cx = nearest_centre_o(kx, scale_x, w);
    theta = theta_map[((size_t)img * h + cy) * w + cx];
  } else {
    theta = theta_kp[flat];
  }
  const float cos_t = cosf(theta), sin_t = sinf(theta);                // bad.py:502-503
  const int oy = (int)floorf(ky) - OOFF, ox = (int)floorf(kx) - OOFF;
  const int groups = num_pairs / 64;
  const int words = num_pairs / 32;

  for (int i = lane; i < OSP; i += 64) { sat[i] = (SAT)0; sat[i * OSP] = (SAT)0; }
  bool integral = true;
  if (lane < OW) {
    const int gx = clampi(ox + lane, 0, w - 1);
    SAT acc = (SAT)0;
    // ALL of the column's 60 loads are issued before the first is used (round 3): in four batches of 15 every window
    // cost four dependent round trips to memory, and at 8-10 waves per CU (LDS) nothing hides them
    float px[OW];
#pragma unroll
    for (int r = 0; r < OW; ++r) px[r] = im[(size_t)clampi(oy + r, 0, h - 1) * w + gx];
#pragma unroll
    for (int r = 0; r < OW; ++r) {
      if (INT) {
        const int v = (int)px[r];
        integral = integral && ((float)v == px[r]) && (v >= 0) && (v <= 255);
      }
      acc += (SAT)px[r];
      sat[(r + 1) * OSP + (lane + 1)] = acc;
    }
  }
  if (INT) {
    const bool ok = __all(integral);
    if (lane == 0) status[flat] = ok ? 1 : 0;
    if (!ok) return;                                                   // wave-uniform
  }
  __syncthreads();
  if (lane < OW) {
    SAT *row = sat + (lane + 1) * OSP + 1;
    SAT acc = (SAT)0;
#pragma unroll 4
    for (int c0 = 0; c0 < OW; c0 += RB) {
      SAT v[RB];
#pragma unroll
      for (int c = 0; c < RB; ++c) v[c] = row[c0 + c];
#pragma unroll
      for (int c = 0; c < RB; ++c) { acc += v[c]; row[c0 + c] = acc; }
    }
  }
  __syncthreads();

  uint32_t *brow = bits ? bits + (size_t)flat * words : nullptr;
  int pop = 0;
  float sumsq = 0.0f;
  for (int g = 0; g < groups; ++g) {
    const int p = g * 64 + lane;
    const uint32_t q = geom[p];
    const float ox1 = (float)((int)(q & 31u) - 16), ox2 = (float)((int)((q >> 5) & 31u) - 16);
    const float oy1 = (float)((int)((q >> 10) & 31u) - 16), oy2 = (float)((int)((q >> 15) & 31u) - 16);
    const int r = (int)((q >> 20) & 15u);
    // bad.py:505-517: rot_dy = ox*sin + oy*cos ; rot_dx = ox*cos - oy*sin ; pos = kp + rot
    const float p1y = ky + (ox1 * sin_t + oy1 * cos_t), p1x = kx + (ox1 * cos_t - oy1 * sin_t);
    const float p2y = ky + (ox2 * sin_t + oy2 * cos_t), p2x = kx + (ox2 * cos_t - oy2 * sin_t);
    // sum of the (2r+1)^2 box centred on image pixel (cy, cx), replicate-extended, from the window's table
    auto box_sum = [&](int cy, int cx) {
      const int wy = cy - oy, wx = cx - ox;
      const int a = clampi(wy - r, 0, OW), b = clampi(wy + r + 1, 0, OW);
      const int l = clampi(wx - r, 0, OW), rr = clampi(wx + r + 1, 0, OW);
      return (double)((sat[b * OSP + rr] - sat[a * OSP + rr]) - (sat[b * OSP + l] - sat[a * OSP + l]));
    };
    const double area = (double)((2 * r + 1) * (2 * r + 1));
    if (bilinear) {
      // sampling_mode="bilinear" (bad.py:535-549): ATen grid_sampler_2d, align_corners, border padding, on the
      // box-mean maps: un-normalise, clip, four neighbours weighted nw/ne/sw/se, out-of-range corners skipped
      auto sample = [&](float py, float px) {
        float iy = (((py * scale_y - 1.0f) + 1.0f) / 2.0f) * (float)(h - 1);
        float ix = (((px * scale_x - 1.0f) + 1.0f) / 2.0f) * (float)(w - 1);
        iy = fminf(fmaxf(iy, 0.0f), (float)(h - 1));
        ix = fminf(fmaxf(ix, 0.0f), (float)(w - 1));
        const float y0f = floorf(iy), x0f = floorf(ix);
        const int y0 = (int)y0f, x0 = (int)x0f;
        const float wy1 = iy - y0f, wx1 = ix - x0f, wy0 = (y0f + 1.0f) - iy, wx0 = (x0f + 1.0f) - ix;
        const bool y1ok = y0 + 1 <= h - 1, x1ok = x0 + 1 <= w - 1;
        float acc = (float)(box_sum(y0, x0) / area) * (wx0 * wy0);
        if (x1ok) acc += (float)(box_sum(y0, x0 + 1) / area) * (wx1 * wy0);
        if (y1ok) acc += (float)(box_sum(y0 + 1, x0) / area) * (wx0 * wy1);
        if (y1ok && x1ok) acc += (float)(box_sum(y0 + 1, x0 + 1) / area) * (wx1 * wy1);
        return acc;
      };
      const float c = (sample(p1y, p1x) - sample(p2y, p2x)) - thr[p];   // bad.py:556-559
      float v = c;
      if (mode == MI_BAD_SOFT) v = 1.0f / (1.0f + expf(c * temperature));
      else if (mode == MI_BAD_HARD) v = (c <= 0.0f) ? 1.0f : 0.0f;
      v = valid ? v : 0.0f;
      if (mode == MI_BAD_HARD) {
        const unsigned long long word = __ballot(v != 0.0f);
        pop += (int)__popcll(word);
        if (brow && lane == 0) {
          brow[2 * g] = (uint32_t)word;
          brow[2 * g + 1] = (uint32_t)(word >> 32);
        }
      }
      sumsq += v * v;
      if (DESC && desc) vals[p] = v;
      continue;
    }
    const double s1 = box_sum(nearest_centre_o(p1y, scale_y, h), nearest_centre_o(p1x, scale_x, w));
    const double s2 = box_sum(nearest_centre_o(p2y, scale_y, h), nearest_centre_o(p2x, scale_x, w));
    const double t = (double)thr[p];
    if (mode == MI_BAD_HARD) {
      const bool bitv = valid && ((s1 - s2) <= t * area);               // bad.py:567,570
      const unsigned long long word = __ballot(bitv);
      pop += (int)__popcll(word);
      if (brow && lane == 0) {
        brow[2 * g] = (uint32_t)word;
        brow[2 * g + 1] = (uint32_t)(word >> 32);
      }
      if (DESC && desc) vals[p] = bitv ? 1.0f : 0.0f;
    } else {
      const float c = (float)((s1 - s2) / area - t);                    // bad.py:559
      float v = c;
      if (mode == MI_BAD_SOFT) v = 1.0f / (1.0f + expf(c * temperature));   // bad.py:565
      v = valid ? v : 0.0f;
      sumsq += v * v;
      if (DESC) vals[p] = v;
    }
  }
  if (!DESC || !desc) return;
  float inv = 1.0f;
  if (normalize) {                                                      // bad.py:573
    const float ss = (mode == MI_BAD_HARD) ? (float)pop : wave_sum(sumsq);
    inv = fmaxf(sqrtf(ss), 1e-12f);
  }
  for (int g = 0; g < groups; ++g) {
    const float v = vals[g * 64 + lane];
    desc[(size_t)flat * num_pairs + g * 64 + lane] = normalize ? v / inv : v;
  }
}

}  // namespace

extern "C" int mi_sparse_bad_oriented(const float *image, int n, int h, int w, const float *keypoints, int k,
                                      const float *orientation_map, const float *keypoint_angles,
                                      const uint32_t *pair_geom, const float *pair_thr, int num_pairs, int mode,
                                      float temperature, int normalize, int bilinear, float max_reach, float *desc,
                                      uint32_t *bits, uint8_t *status, mi_stream_t stream) {
  MI_ENTER();
  if (!image || !keypoints || !pair_geom || !pair_thr) return MI_E_NULL;
  if (!orientation_map == !keypoint_angles) return MI_E_NULL;          // exactly one angle source
  if (!desc && !bits) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0 || k <= 0 || (long long)n * k > 0x7fffffffLL) return MI_E_SHAPE;
  if (num_pairs <= 0 || num_pairs % 64 != 0 || num_pairs > 1024) return MI_E_PARAM;
  if (mode != MI_BAD_RAW && mode != MI_BAD_SOFT && mode != MI_BAD_HARD) return MI_E_PARAM;
  if (bits && mode != MI_BAD_HARD) return MI_E_PARAM;
  const float scale_y = (float)(2.0 / ((double)(h - 1) + 1e-8));
  const float scale_x = (float)(2.0 / ((double)(w - 1) + 1e-8));
  if (!(max_reach >= 0.0f)) return MI_E_PARAM;
  const bool small = !bilinear && max_reach > 0.0f && max_reach <= 22.5f;   // see the window note at the top
#define BO_LAUNCH(SAT, DESC, OWIN) hipLaunchKernelGGL((bad_oriented_kernel<SAT, DESC, OWIN>), dim3((unsigned)(n * k)), dim3(64), 0, (hipStream_t)stream, image, h, w, keypoints, k, orientation_map, keypoint_angles, pair_geom, pair_thr, num_pairs, mode, temperature, normalize, scale_y, scale_x, bilinear ? 1 : 0, desc, bits, status)
#define BO_PICK(SAT) do { if (desc) { if (small) BO_LAUNCH(SAT, true, 48); else BO_LAUNCH(SAT, true, 60); } else { if (small) BO_LAUNCH(SAT, false, 48); else BO_LAUNCH(SAT, false, 60); } } while (0)
  if (status) BO_PICK(int);                                             // integer tables first, the rest in fp64
  BO_PICK(double);
#undef BO_PICK
#undef BO_LAUNCH
  return mi_launch_status();
}
