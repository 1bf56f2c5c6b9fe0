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
#include "hooks.h"

#include <math.h>
#include <stdint.h>

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
struct OrientedArgs {
  MiSets images;
  int h, w;
  const float *kpts;
  int k;
  const float *theta_map, *theta_kp;
  const uint32_t *geom;
  const float *thr;
  int num_pairs, mode;
  float temperature;
  int normalize;
  float scale_y, scale_x;
  int bilinear;
  float *desc;
  uint32_t *bits;
  uint8_t *status;
};

// One keypoint (`flat`) by one wave; `sat` = (OW + 1)^2 table entries, `vals` = 1024 floats when DESC.
template <typename SAT, bool DESC, int OW>
__device__ __forceinline__ void bad_oriented_body(const OrientedArgs &A, int flat, SAT *sat, float *vals) {
  constexpr bool INT = sizeof(SAT) == 4;
  constexpr int OOFF = OW / 2 - 1;  // window origin = floor(k) - OOFF
  constexpr int OSP = OW + 1;       // SAT edge
  constexpr int RB = OW / 4;        // rows / columns per batch of the in-LDS prefix pass
  const float *kpts = A.kpts, *theta_map = A.theta_map, *theta_kp = A.theta_kp, *thr = A.thr;
  const uint32_t *geom = A.geom;
  const int h = A.h, w = A.w, k = A.k, num_pairs = A.num_pairs, mode = A.mode, normalize = A.normalize;
  const int bilinear = A.bilinear;
  const float temperature = A.temperature, scale_y = A.scale_y, scale_x = A.scale_x;
  float *desc = A.desc;
  uint32_t *bits = A.bits;
  uint8_t *status = A.status;
  const int lane = threadIdx.x;
  const int img = flat / k;
  const float *im = mi_set_item<float>(A.images, img, (size_t)h * w);
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

template <typename SAT, bool DESC, int OW>
__global__ __launch_bounds__(64) void bad_oriented_kernel(OrientedArgs A) {
  __shared__ SAT sat[(OW + 1) * (OW + 1)];
  __shared__ float vals[DESC ? 1024 : 1];
  const int flat = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  if (sizeof(SAT) != 4 && A.status && A.status[flat]) return;          // the integer instance did this keypoint
  bad_oriented_body<SAT, DESC, OW>(A, flat, sat, vals);
}

// What the integer instances flagged status = 0 (windows that are not uint8-valued): one wave reads 64 status bytes at
// once and runs the fp64 body for the flagged ones -- n * k / 64 short waves instead of n * k waves that each load one
// byte and leave (15.8 us per 65 k keypoints, twice per step).
template <bool DESC, int OW>
__global__ __launch_bounds__(64) void bad_oriented_rest_kernel(OrientedArgs A, int total) {
  __shared__ double sat[(OW + 1) * (OW + 1)];
  __shared__ float vals[DESC ? 1024 : 1];
  const int base = (int)blockIdx.x * 64;
  const int idx = base + (int)threadIdx.x;
  unsigned long long todo = __ballot(idx < total && A.status[idx] == 0);
  while (todo) {                                                        // wave-uniform
    const int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    bad_oriented_body<double, DESC, OW>(A, base + b, sat, vals);
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The matchers' instance (round 4): nearest sampling, uint8-valued window, 64 * GROUPS pairs; packed hard bits or the
// float descriptor.  Same values as bad_oriented_kernel<int, ., OW>, rebuilt around what bounded that kernel (178 us per
// 65 k keypoints; SQ counters: vector pipe ~100 % busy, every instruction four cycles, two dependent global loads --
// geometry word, threshold -- inside its rolled pair loop):
//   * every load of the wave -- the 48 (60) window rows, the GROUPS geometry words and thresholds -- is issued before
//     the first is used; the angle's sine / cosine are computed under them;
//   * window rows are BUFFER loads whose row offset is a scalar operand (one s_add per row for a window inside the image);
//   * the table's rows are OW + 2 = even words long, so the row pass reads and writes 8 bytes per instruction
//     (conflict-free: 50-word stride = every even bank once per 32 lanes) -- half the LDS instructions of the first pass;
//   * "is this window uint8-valued" costs one convert and one v_bitop3 per pixel;
//   * the pair phase is straight-line code for all GROUPS groups (its own comment below).
// A keypoint whose window is not uint8-valued is flagged status = 0 and left to bad_oriented_rest_kernel.
// 132 us per 65 k keypoints at 512 bits (123.7 in the VO step), 1110 vector + 176 scalar instructions per keypoint.
// clamp(x, 0, hi) for hi >= 0 as ONE v_med3_i32 (the compiler cannot assume hi >= 0 and emits min + compare + select)
__device__ __forceinline__ int med3_0(int x, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(hi));
  return r;
}

__device__ __forceinline__ void rotated_positions(uint32_t q, float ky, float kx, float sin_t, float cos_t, float &p1y,
                                                  float &p1x, float &p2y, float &p2x, int &r) {
  const float ox1 = (float)((int)(q & 31u) - 16), ox2 = (float)((int)((q >> 5) & 31u) - 16);
  const float oy1 = (float)((int)((q >> 10) & 31u) - 16), oy2 = (float)((int)((q >> 15) & 31u) - 16);
  r = (int)((q >> 20) & 15u);
  // bad.py:505-517: rot_dy = ox*sin + oy*cos ; rot_dx = ox*cos - oy*sin ; pos = kp + rot
  p1y = ky + (ox1 * sin_t + oy1 * cos_t);
  p1x = kx + (ox1 * cos_t - oy1 * sin_t);
  p2y = ky + (ox2 * sin_t + oy2 * cos_t);
  p2x = kx + (ox2 * cos_t - oy2 * sin_t);
}

// DESC = false: packed bits of the hard mode (`bits`).  DESC = true: the float descriptor (`desc`) in any mode -- raw /
// sigmoid / 0-1, normalised or not --, the pair value formed by the generic kernel's own expressions (fp64 mean minus
// threshold) and kept in registers until the norm is known (the generic kernel stages it in LDS).
template <int OW, int GROUPS, bool DESC>
__global__ __launch_bounds__(64) void bad_oriented_bits_kernel(MiSets images, int h, int w,
                                                               const float *__restrict__ kpts, int k,
                                                               const float *__restrict__ theta_map,
                                                               const float *__restrict__ theta_kp,
                                                               const uint32_t *__restrict__ geom,
                                                               const float *__restrict__ thr, float scale_y,
                                                               float scale_x, uint32_t *__restrict__ bits,
                                                               float *__restrict__ desc, int mode, float temperature,
                                                               int normalize, uint8_t *__restrict__ status) {
  constexpr int OOFF = OW / 2 - 1;  // window origin = floor(k) - OOFF
  constexpr int OSP = OW + 2;       // words per table row: [0] = the zero column, [1 .. OW] the sums, [OW + 1] padding
  __shared__ __attribute__((aligned(16))) int sat[(OW + 1) * OSP];
  const int lane = threadIdx.x;
  const int flat = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int img = flat / k;
  const float *im = mi_set_item<float>(images, img, (size_t)h * w);
  const float ky_raw = kpts[(size_t)flat * 2 + 0];
  const float kx_raw = kpts[(size_t)flat * 2 + 1];
  uint4 *brow = DESC ? nullptr : reinterpret_cast<uint4 *>(bits + (size_t)flat * (2 * GROUPS));
  float *drow = DESC ? desc + (size_t)flat * (64 * GROUPS) : nullptr;
  if (!(ky_raw >= 0.0f)) {                                              // bad.py:461: an invalid keypoint's descriptor is 0
    if (DESC) {
#pragma unroll
      for (int g = 0; g < GROUPS; ++g) drow[g * 64 + lane] = 0.0f;
    } else if (lane < GROUPS / 2) {
      brow[lane] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (lane == 0) status[flat] = 1;
    return;
  }
  const float ky = fminf(fmaxf(ky_raw, 0.0f), (float)(h - 1));          // bad.py:464-465
  const float kx = fminf(fmaxf(kx_raw, 0.0f), (float)(w - 1));
  const int oy = (int)floorf(ky) - OOFF, ox = (int)floorf(kx) - OOFF;

  // the window: lane c owns column c; all rows in flight at once
  // (buffer loads: the row's byte offset is a SCALAR operand -- a handful of scalar instructions per row --
  // where flat addressing spent seven vector instructions per row on clamp, 64-bit multiply and add)
  float px[OW];
  {
    const int oys = __builtin_amdgcn_readfirstlane(oy), oxs = __builtin_amdgcn_readfirstlane(ox);
    const __amdgpu_buffer_rsrc_t plane = __builtin_amdgcn_make_buffer_rsrc((void *)im, 0, h * w * 4, 0x00020000);
    const int voff = clampi(oxs + (lane < OW ? lane : OW - 1), 0, w - 1) * 4;
    const int rowbytes = w * 4;
    if (oys >= 0 && oys + OW <= h) {                                    // (wave-uniform) all rows inside: one s_add per row
      int soff = oys * rowbytes;
#pragma unroll
      for (int r = 0; r < OW; ++r) {
        px[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(plane, voff, soff, 0));
        soff += rowbytes;
      }
    } else {
#pragma unroll
      for (int r = 0; r < OW; ++r)
        px[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(plane, voff, clampi(oys + r, 0, h - 1) * rowbytes, 0));
    }
  }
  uint32_t q[GROUPS];
  float t[GROUPS];
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) { q[g] = geom[g * 64 + lane]; t[g] = thr[g * 64 + lane]; }
  float theta;
  if (theta_map) {                                                      // bad.py:490-500
    const int cy = nearest_centre_o(ky, scale_y, h), cx = nearest_centre_o(kx, scale_x, w);
    theta = theta_map[((size_t)img * h + cy) * w + cx];
  } else {
    theta = theta_kp[flat];
  }
  for (int i = lane; i < OSP; i += 64) sat[i] = 0;                      // row 0
  if (lane <= OW) sat[lane * OSP] = 0;                                  // column 0
  const float cos_t = cosf(theta), sin_t = sinf(theta);                 // bad.py:502-503

  // uint8-valued window: every pixel has the bits of (float)(uint8)pixel (-0.0 counts as differing: fp64 path).
  // differs |= back ^ px as ONE v_bitop3 behind an asm: written in C the optimiser turns the or-chain into 48 compares
  // that it collects behind the last load, with 48 more live registers (129: one more than four waves per SIMD have)
  uint32_t differs = 0u;
  if (lane < OW) {
    int acc = 0;
#pragma unroll
    for (int r = 0; r < OW; ++r) {
      // one s_waitcnt per eight rows instead of one per row (loads return in order: row r + 7 here means rows <= r + 7)
      if (r % 8 == 0) asm volatile("" ::"v"(px[r + 7 < OW ? r + 7 : OW - 1]));
      const int v = (int)px[r];
      const float back = (float)(v & 0xff);
      asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(differs) : "v"(back), "v"(px[r]));   // a | (b ^ c)
      acc += v;
      sat[(r + 1) * OSP + (lane + 1)] = acc;
    }
  }
  bool tame = true;                                                     // thresholds finite and |t| <= 1e30 (else: fp64 path)
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) tame = tame && (DESC || fabsf(t[g]) <= 1e30f);   // (only the fp32 bit test needs it)
  const bool ok = __all(differs == 0u && tame);
  if (lane == 0) status[flat] = ok ? 1 : 0;
  if (!ok) return;                                                      // wave-uniform
  __syncthreads();
  if (lane < OW) {
    int2 *row = reinterpret_cast<int2 *>(sat + (lane + 1) * OSP);
    int acc = 0;
    int2 v[OSP / 2];
#pragma unroll
    for (int c = 0; c < OSP / 2; ++c) v[c] = row[c];
#pragma unroll
    for (int c = 0; c < OSP / 2; ++c) {
      acc += v[c].x;
      v[c].x = acc;
      acc += v[c].y;
      v[c].y = acc;
      row[c] = v[c];
    }
  }
  __syncthreads();

  // The pair phase, straight-line for all groups.  Same values as the generic kernel's, fewer instructions:
  //   * centre = clamp(rint(x)) in integers (= rint(clamp(x)): the bounds are integers; NaN -> 0 both ways), with the
  //     0.5 folded into the size factor (a power of two: the same product);
  //   * the box is placed by its CENTRE, clamped so that it lies in the window -- a no-op under the window guarantee
  //     (header comment), like the generic kernel's four edge clamps -- and its four corners are one address + three
  //     per-pair constants;
  //   * bit = (d <= t * area) for the integer d = s1 - s2, decided in fp32 with the product's exact residual:
  //     p = fl(t * a), e = fma(t, a, -p) = t * a - p exactly; d - p is exact when d and p lie within a factor two of each
  //     other, and otherwise its rounding cannot change its sign or bring it under |e| <= ulp(p) / 2: (d - p <= e) is
  //     (d <= t * a).  (|t| <= 1e30, checked above, keeps p finite.)  Five fp64 instructions per 64 pairs less.
  constexpr int ROWB = OSP * 4;
  const float half_h = 0.5f * (float)(h - 1), half_w = 0.5f * (float)(w - 1);
  uint32_t words[DESC ? 1 : 2 * GROUPS];
  float vals[DESC ? GROUPS : 1];
  int pop = 0;
  float sumsq = 0.0f;
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {
    float p1y, p1x, p2y, p2x;
    int r;
    rotated_positions(q[g], ky, kx, sin_t, cos_t, p1y, p1x, p2y, p2x, r);
    auto centre = [](float pos, float scale, float half, int size) {
      const float x = ((pos * scale - 1.0f) + 1.0f) * half;            // = ((g + 1) / 2) * (size - 1) of nearest_centre_o
      return med3_0(__float2int_rn(x), size - 1);
    };
    const int oyr = oy + r, oxr = ox + r, amax = OW - 1 - 2 * r;        // top-left corner of the box in window coordinates
    const int r4 = (r << 3) | 4;                                        // (2r + 1) * 4 bytes
    const int ro = __umul24(r4, OSP), ror = ro + r4;
    auto box_sum = [&](int cy, int cx) {
      const int a = med3_0(cy - oyr, amax), l = med3_0(cx - oxr, amax);
      const char *base = reinterpret_cast<const char *>(sat) + (__umul24(a, ROWB) + (l << 2));
      const int s_al = *reinterpret_cast<const int *>(base), s_ar = *reinterpret_cast<const int *>(base + r4);
      const int s_bl = *reinterpret_cast<const int *>(base + ro), s_br = *reinterpret_cast<const int *>(base + ror);
      return (s_br - s_ar) - (s_bl - s_al);
    };
    const int s1 = box_sum(centre(p1y, scale_y, half_h, h), centre(p1x, scale_x, half_w, w));
    const int s2 = box_sum(centre(p2y, scale_y, half_h, h), centre(p2x, scale_x, half_w, w));
    if constexpr (DESC) {
      // the generic kernel's expressions (bad.py:559,565,567): fp64 mean difference minus threshold
      const double area = (double)((2 * r + 1) * (2 * r + 1));
      const double td = (double)t[g];
      float v;
      if (mode == MI_BAD_HARD) {
        const bool bitv = (double)(s1 - s2) <= td * area;
        pop += (int)__popcll(__ballot(bitv));
        v = bitv ? 1.0f : 0.0f;
      } else {
        const float c = (float)((double)(s1 - s2) / area - td);
        v = c;
        if (mode == MI_BAD_SOFT) v = 1.0f / (1.0f + expf(c * temperature));
        sumsq += v * v;
      }
      vals[g] = v;
    } else {
      // bad.py:567,570: bit = (mean1 - mean2 - thr <= 0) = (s1 - s2 <= t * area)
      const float area = (float)((2 * r + 1) * (2 * r + 1));
      const float p = t[g] * area, e = __builtin_fmaf(t[g], area, -p);
      const bool bitv = ((float)(s1 - s2) - p) <= e;
      const unsigned long long word = __ballot(bitv);
      words[2 * g] = (uint32_t)word;
      words[2 * g + 1] = (uint32_t)(word >> 32);
    }
  }
  if constexpr (DESC) {
    float inv = 1.0f;
    if (normalize) {                                                    // bad.py:573
      const float ss = (mode == MI_BAD_HARD) ? (float)pop : wave_sum(sumsq);
      inv = fmaxf(sqrtf(ss), 1e-12f);
    }
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) drow[g * 64 + lane] = normalize ? vals[g] / inv : vals[g];
  } else if (lane == 0) {
#pragma unroll
    for (int i = 0; i < GROUPS / 2; ++i)
      brow[i] = make_uint4(words[4 * i], words[4 * i + 1], words[4 * i + 2], words[4 * i + 3]);
  }
}
}  // namespace

static int sparse_bad_oriented_launch(MiSets images, int n, int h, int w, const float *keypoints, int k,
                                      const float *orientation_map, const float *keypoint_angles,
                                      const uint32_t *pair_geom, const float *pair_thr, int num_pairs, int mode,
                                      float temperature, int normalize, int bilinear, float max_reach, float *desc,
                                      uint32_t *bits, uint8_t *status, mi_stream_t stream) {
  const float *image = static_cast<const float *>(images.a);
  if (!image || (images.per_set < n && !images.b) || !keypoints || !pair_geom || !pair_thr) return MI_E_NULL;
  if (!orientation_map == !keypoint_angles) return MI_E_NULL;          // exactly one angle source
  if (orientation_map && images.per_set < n) return MI_E_PARAM;         // a dense angle map belongs to ONE image batch
  if (!desc && !bits) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0 || k <= 0 || (long long)n * k > 0x7fffffffLL) return MI_E_SHAPE;
  if (num_pairs <= 0 || num_pairs % 64 != 0 || num_pairs > 1024) return MI_E_PARAM;
  if (mode != MI_BAD_RAW && mode != MI_BAD_SOFT && mode != MI_BAD_HARD) return MI_E_PARAM;
  if (bits && mode != MI_BAD_HARD) return MI_E_PARAM;
  const float scale_y = (float)(2.0 / ((double)(h - 1) + 1e-8));
  const float scale_x = (float)(2.0 / ((double)(w - 1) + 1e-8));
  if (!(max_reach >= 0.0f)) return MI_E_PARAM;
  const bool small = !bilinear && max_reach > 0.0f && max_reach <= 22.5f;   // see the window note at the top
  OrientedArgs A{images, h, w, keypoints, k, orientation_map, keypoint_angles, pair_geom, pair_thr, num_pairs, mode,
                 temperature, normalize, scale_y, scale_x, bilinear ? 1 : 0, desc, bits, status};
  const unsigned total = (unsigned)(n * k);
  hipStream_t s = (hipStream_t)stream;
#define BO_LAUNCH(SAT, DESC, OWIN) hipLaunchKernelGGL((bad_oriented_kernel<SAT, DESC, OWIN>), dim3(total), dim3(64), 0, s, A)
#define BO_PICK(SAT) do { if (desc) { if (small) BO_LAUNCH(SAT, true, 48); else BO_LAUNCH(SAT, true, 60); } else { if (small) BO_LAUNCH(SAT, false, 48); else BO_LAUNCH(SAT, false, 60); } } while (0)
#define BO_REST(DESC, OWIN) hipLaunchKernelGGL((bad_oriented_rest_kernel<DESC, OWIN>), dim3((total + 63u) / 64u), dim3(64), 0, s, A, (int)total)
#define BO_FAST(OWIN, G, D) hipLaunchKernelGGL((bad_oriented_bits_kernel<OWIN, G, D>), dim3(total), dim3(64), 0, s, images, h, w, keypoints, k, orientation_map, keypoint_angles, pair_geom, pair_thr, scale_y, scale_x, bits, desc, mode, temperature, normalize, status)
  if (!status) {                                                        // no scratch for the two-pass form: fp64 for all
    BO_PICK(double);
    return mi_launch_status();
  }
  // integer tables first (uint8-valued windows), then the flagged rest in fp64
  const bool fast_form = MI_HOOK(bad_oriented_impl, 0) == 0 && !bilinear && (num_pairs == 512 || num_pairs == 256) &&
                         (long long)h * w * 4 < 0x7fffffffLL;
  const bool bits_form = fast_form && mode == MI_BAD_HARD && bits && !desc && (reinterpret_cast<uintptr_t>(bits) & 15u) == 0;
  const bool desc_form = fast_form && desc && !bits;
  if (bits_form) {
    if (num_pairs == 512) { if (small) BO_FAST(48, 8, false); else BO_FAST(60, 8, false); }
    else { if (small) BO_FAST(48, 4, false); else BO_FAST(60, 4, false); }
  } else if (desc_form) {
    if (num_pairs == 512) { if (small) BO_FAST(48, 8, true); else BO_FAST(60, 8, true); }
    else { if (small) BO_FAST(48, 4, true); else BO_FAST(60, 4, true); }
  } else {
    BO_PICK(int);
  }
  if (desc) { if (small) BO_REST(true, 48); else BO_REST(true, 60); }
  else { if (small) BO_REST(false, 48); else BO_REST(false, 60); }
#undef BO_FAST
#undef BO_REST
#undef BO_PICK
#undef BO_LAUNCH
  return mi_launch_status();
}

extern "C" int mi_sparse_bad_oriented(const float *image, int n, int h, int w, const float *keypoints, int k,
                                      const float *orientation_map, const float *keypoint_angles,
                                      const uint32_t *pair_geom, const float *pair_thr, int num_pairs, int mode,
                                      float temperature, int normalize, int bilinear, float max_reach, float *desc,
                                      uint32_t *bits, uint8_t *status, mi_stream_t stream) {
  MI_ENTER();
  return sparse_bad_oriented_launch(mi_one_set(image, n), n, h, w, keypoints, k, orientation_map, keypoint_angles, pair_geom,
                                    pair_thr, num_pairs, mode, temperature, normalize, bilinear, max_reach, desc, bits,
                                    status, stream);
}

extern "C" int mi_sparse_bad_oriented_pair(const float *image_a, const float *image_b, int per_set, int h, int w,
                                           const float *keypoints, int k, const float *keypoint_angles,
                                           const uint32_t *pair_geom, const float *pair_thr, int num_pairs, int mode,
                                           float temperature, int normalize, int bilinear, float max_reach, float *desc,
                                           uint32_t *bits, uint8_t *status, mi_stream_t stream) {
  MI_ENTER();
  if (!image_b) return MI_E_NULL;
  if (per_set <= 0 || per_set > 0x3fffffff) return MI_E_SHAPE;
  return sparse_bad_oriented_launch(MiSets{image_a, image_b, per_set}, 2 * per_set, h, w, keypoints, k, nullptr,
                                    keypoint_angles, pair_geom, pair_thr, num_pairs, mode, temperature, normalize, bilinear,
                                    max_reach, desc, bits, status, stream);
}
