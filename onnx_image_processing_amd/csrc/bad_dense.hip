// K4 (dense): BAD response for EVERY pixel -- the reference's BADDescriptor.forward.
// Semantics: reference pytorch_model/descriptor/bad.py:62-110 (_compute_diff_map) and :189-218:
// replicate-pad the image by the max radius, box centre = clamp(pixel + offset) into the image,
// box mean over [c-r, c+r] of the padded image, response = mean1 - mean2 - thr, then
// raw / sigmoid(-c*T) / (c <= 0).  Output (n, P, h, w), no normalisation.
// The reference evaluates the box sums from an fp32 integral image (inexact above 2^24: up to
// ~1 intensity unit off); here the summed-area table is fp64 over a 48x48 window
// per 16x16 pixel tile, i.e. exact for integer-valued images -- tolerance parity by design.
// Plus the two gather helpers of bad.py:221-333.
#include "common.h"

#include <math.h>

namespace {

constexpr int DT = 16;            // tile edge
constexpr int DW = DT + 32;       // window edge: offsets -16..+15 around the tile
constexpr int DSP = DW + 1;

__global__ __launch_bounds__(256) void bad_dense_kernel(const float *__restrict__ image, int h, int w,
                                                        const uint32_t *__restrict__ geom,
                                                        const float *__restrict__ thr, int num_pairs, int mode,
                                                        float temperature, float *__restrict__ out, int tiles_x,
                                                        int tiles_y) {
  __shared__ double sat[DSP * DSP];
  const int t = threadIdx.x;
  int bid = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int tx_tile = bid % tiles_x;
  bid /= tiles_x;
  const int ty_tile = bid % tiles_y;
  const int img = bid / tiles_y;
  const int x0 = tx_tile * DT, y0 = ty_tile * DT;
  const int wy0 = y0 - 16, wx0 = x0 - 16;           // window origin (replicate-extended coordinates)
  const float *im = image + (size_t)img * h * w;
  for (int i = t; i < DSP; i += 256) { sat[i] = 0.0; sat[i * DSP] = 0.0; }
  for (int i = t; i < DW * DW; i += 256) {
    const int r = i / DW, c = i - r * DW;
    sat[(r + 1) * DSP + c + 1] = (double)im[(size_t)clampi(wy0 + r, 0, h - 1) * w + clampi(wx0 + c, 0, w - 1)];
  }
  __syncthreads();
  if (t < DW) {                                      // prefix along rows: thread = row
    double *row = sat + (t + 1) * DSP + 1;
    double acc = 0.0;
    for (int c = 0; c < DW; ++c) { acc += row[c]; row[c] = acc; }
  }
  __syncthreads();
  if (t < DW) {                                      // prefix along columns: thread = column
    double *col = sat + DSP + t + 1;
    double acc = 0.0;
    for (int r = 0; r < DW; ++r) { acc += col[r * DSP]; col[r * DSP] = acc; }
  }
  __syncthreads();
  const int lx = t & 15, ly = t >> 4;
  const int y = y0 + ly, x = x0 + lx;
  if (y >= h || x >= w) return;
  float *dst = out + (size_t)img * num_pairs * h * w + (size_t)y * w + x;
  for (int p = 0; p < num_pairs; ++p) {
    const uint32_t q = geom[p];                      // wave-uniform
    const int ox1 = (int)(q & 31u) - 16, ox2 = (int)((q >> 5) & 31u) - 16;
    const int oy1 = (int)((q >> 10) & 31u) - 16, oy2 = (int)((q >> 15) & 31u) - 16;
    const int r = (int)((q >> 20) & 15u);
    const int c1y = clampi(y + oy1, 0, h - 1) - wy0, c1x = clampi(x + ox1, 0, w - 1) - wx0;   // bad.py:81-82
    const int c2y = clampi(y + oy2, 0, h - 1) - wy0, c2x = clampi(x + ox2, 0, w - 1) - wx0;
    const int a1 = clampi(c1y - r, 0, DW), b1 = clampi(c1y + r + 1, 0, DW);
    const int l1 = clampi(c1x - r, 0, DW), r1 = clampi(c1x + r + 1, 0, DW);
    const int a2 = clampi(c2y - r, 0, DW), b2 = clampi(c2y + r + 1, 0, DW);
    const int l2 = clampi(c2x - r, 0, DW), r2 = clampi(c2x + r + 1, 0, DW);
    const double s1 = (sat[b1 * DSP + r1] - sat[a1 * DSP + r1]) - (sat[b1 * DSP + l1] - sat[a1 * DSP + l1]);
    const double s2 = (sat[b2 * DSP + r2] - sat[a2 * DSP + r2]) - (sat[b2 * DSP + l2] - sat[a2 * DSP + l2]);
    const double area = (double)((2 * r + 1) * (2 * r + 1));
    const float c = (float)((s1 - s2) / area - (double)thr[p]);
    float v = c;                                                          // bad.py:212
    if (mode == MI_BAD_SOFT) v = 1.0f / (1.0f + expf(c * temperature));   // bad.py:216
    else if (mode == MI_BAD_HARD) v = ((s1 - s2) <= (double)thr[p] * area) ? 1.0f : 0.0f;   // bad.py:218
    dst[(size_t)p * h * w] = v;
  }
}

// ---- dense, per-pixel oriented (bad.py:112-187): the pair offsets of every pixel are rotated by the
// orientation map's angle there and the box-mean maps are sampled bilinearly (grid_sample, align_corners,
// border padding).  Rotated offsets reach 21.2 px, +1 for the bilinear neighbour, +7 box radius: every box
// of a 16x16 tile lies in the 74x74 replicate-clamped window starting 29 px up/left of the tile.
constexpr int OT = 16;            // tile edge
constexpr int OWN = OT + 58;      // window edge
constexpr int OSPN = OWN + 1;

__global__ __launch_bounds__(256) void bad_dense_oriented_kernel(const float *__restrict__ image,
                                                                 const float *__restrict__ orientation, int h, int w,
                                                                 const uint32_t *__restrict__ geom,
                                                                 const float *__restrict__ thr, int num_pairs,
                                                                 int mode, float temperature, float scale_y,
                                                                 float scale_x, float *__restrict__ out, int tiles_x,
                                                                 int tiles_y) {
  __shared__ double sat[OSPN * OSPN];
  const int t = threadIdx.x;
  int bid = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int tx_tile = bid % tiles_x;
  bid /= tiles_x;
  const int ty_tile = bid % tiles_y;
  const int img = bid / tiles_y;
  const int x0 = tx_tile * OT, y0 = ty_tile * OT;
  const int wy0 = y0 - 29, wx0 = x0 - 29;
  const float *im = image + (size_t)img * h * w;
  for (int i = t; i < OSPN; i += 256) { sat[i] = 0.0; sat[i * OSPN] = 0.0; }
  for (int i = t; i < OWN * OWN; i += 256) {
    const int r = i / OWN, c = i - r * OWN;
    sat[(r + 1) * OSPN + c + 1] = (double)im[(size_t)clampi(wy0 + r, 0, h - 1) * w + clampi(wx0 + c, 0, w - 1)];
  }
  __syncthreads();
  if (t < OWN) {
    double *row = sat + (t + 1) * OSPN + 1;
    double acc = 0.0;
    for (int c = 0; c < OWN; ++c) { acc += row[c]; row[c] = acc; }
  }
  __syncthreads();
  if (t < OWN) {
    double *col = sat + OSPN + t + 1;
    double acc = 0.0;
    for (int r = 0; r < OWN; ++r) { acc += col[r * OSPN]; col[r * OSPN] = acc; }
  }
  __syncthreads();
  const int lx = t & 15, ly = t >> 4;
  const int y = y0 + ly, x = x0 + lx;
  if (y >= h || x >= w) return;
  const float theta = orientation[((size_t)img * h + y) * w + x];
  const float cos_t = cosf(theta), sin_t = sinf(theta);                  // bad.py:146-147
  float *dst = out + (size_t)img * num_pairs * h * w + (size_t)y * w + x;
  for (int p = 0; p < num_pairs; ++p) {
    const uint32_t q = geom[p];                                           // wave-uniform
    const float ox1 = (float)((int)(q & 31u) - 16), ox2 = (float)((int)((q >> 5) & 31u) - 16);
    const float oy1 = (float)((int)((q >> 10) & 31u) - 16), oy2 = (float)((int)((q >> 15) & 31u) - 16);
    const int r = (int)((q >> 20) & 15u);
    const double area = (double)((2 * r + 1) * (2 * r + 1));
    auto box_mean = [&](int cy, int cx) {
      const int wy = cy - wy0, wx = cx - wx0;
      const int a = clampi(wy - r, 0, OWN), b = clampi(wy + r + 1, 0, OWN);
      const int l = clampi(wx - r, 0, OWN), rr = clampi(wx + r + 1, 0, OWN);
      return (float)(((sat[b * OSPN + rr] - sat[a * OSPN + rr]) - (sat[b * OSPN + l] - sat[a * OSPN + l])) / area);
    };
    auto sample = [&](float dy, float dx) {                               // bad.py:160-181
      float iy = (((((float)y + dy) * scale_y - 1.0f) + 1.0f) / 2.0f) * (float)(h - 1);
      float ix = (((((float)x + dx) * scale_x - 1.0f) + 1.0f) / 2.0f) * (float)(w - 1);
      iy = fminf(fmaxf(iy, 0.0f), (float)(h - 1));
      ix = fminf(fmaxf(ix, 0.0f), (float)(w - 1));
      const float y0f = floorf(iy), x0f = floorf(ix);
      const int yi = (int)y0f, xi = (int)x0f;
      const float wy1 = iy - y0f, wx1 = ix - x0f, wy0f = (y0f + 1.0f) - iy, wx0f = (x0f + 1.0f) - ix;
      const bool y1ok = yi + 1 <= h - 1, x1ok = xi + 1 <= w - 1;
      float acc = box_mean(yi, xi) * (wx0f * wy0f);
      if (x1ok) acc += box_mean(yi, xi + 1) * (wx1 * wy0f);
      if (y1ok) acc += box_mean(yi + 1, xi) * (wx0f * wy1);
      if (y1ok && x1ok) acc += box_mean(yi + 1, xi + 1) * (wx1 * wy1);
      return acc;
    };
    // rot_dy = ox*sin + oy*cos ; rot_dx = ox*cos - oy*sin (bad.py:154-157)
    const float s1 = sample(ox1 * sin_t + oy1 * cos_t, ox1 * cos_t - oy1 * sin_t);
    const float s2 = sample(ox2 * sin_t + oy2 * cos_t, ox2 * cos_t - oy2 * sin_t);
    const float c = (s1 - s2) - thr[p];                                   // bad.py:212
    float v = c;
    if (mode == MI_BAD_SOFT) v = 1.0f / (1.0f + expf(c * temperature));   // bad.py:216
    else if (mode == MI_BAD_HARD) v = (c <= 0.0f) ? 1.0f : 0.0f;          // bad.py:218
    dst[(size_t)p * h * w] = v;
  }
}

// bad.py:221-274 (nearest: integer truncation of the coordinates) and :277-333 (bilinear
// grid_sample, align_corners=True, padding "border", with the reference's own normalisation
// k / (size-1+1e-8) * 2 - 1).  (B,D,H,W), (B,N,2) -> (B,N,D)
__global__ __launch_bounds__(256) void gather_desc_kernel(const float *__restrict__ map, int d, int h, int w,
                                                          const float *__restrict__ kpts, int nk, int bilinear,
                                                          float *__restrict__ out) {
  const int b = blockIdx.y;
  const int kp = blockIdx.x;
  const float ky = kpts[((size_t)b * nk + kp) * 2 + 0], kx = kpts[((size_t)b * nk + kp) * 2 + 1];
  const float *mp = map + (size_t)b * d * h * w;
  float *op = out + ((size_t)b * nk + kp) * d;
  if (!bilinear) {
    const long long yi = (long long)ky, xi = (long long)kx;             // .long(): truncation
    const long long flat = yi * w + xi;
    for (int c = threadIdx.x; c < d; c += 256) op[c] = mp[(size_t)c * h * w + flat];
    return;
  }
  const float gy = ky / (float)((double)(h - 1) + 1e-8) * 2.0f - 1.0f;
  const float gx = kx / (float)((double)(w - 1) + 1e-8) * 2.0f - 1.0f;
  float fy = ((gy + 1.0f) / 2.0f) * (float)(h - 1), fx = ((gx + 1.0f) / 2.0f) * (float)(w - 1);
  fy = fminf(fmaxf(fy, 0.0f), (float)(h - 1));
  fx = fminf(fmaxf(fx, 0.0f), (float)(w - 1));
  const float y0f = floorf(fy), x0f = floorf(fx);
  const int iy0 = (int)y0f, ix0 = (int)x0f;
  const int iy1 = iy0 + 1, ix1 = ix0 + 1;
  const float wy1 = fy - y0f, wx1 = fx - x0f, wy0 = 1.0f - wy1, wx0 = 1.0f - wx1;
  // ATen grid_sampler_2d bilinear: nw*v00 + ne*v01 + sw*v10 + se*v11, out-of-range corners contribute 0
  const float nw = wy0 * wx0, ne = wy0 * wx1, sw = wy1 * wx0, se = wy1 * wx1;
  const bool y1ok = iy1 <= h - 1, x1ok = ix1 <= w - 1;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float *pl = mp + (size_t)c * h * w;
    float acc = pl[(size_t)iy0 * w + ix0] * nw;
    if (x1ok) acc += pl[(size_t)iy0 * w + ix1] * ne;
    if (y1ok) acc += pl[(size_t)iy1 * w + ix0] * sw;
    if (y1ok && x1ok) acc += pl[(size_t)iy1 * w + ix1] * se;
    op[c] = acc;
  }
}

}  // namespace

extern "C" int mi_bad_dense(const float *image, int n, int h, int w, const uint32_t *pair_geom,
                            const float *pair_thr, int num_pairs, int mode, float temperature, float *out,
                            mi_stream_t stream) {
  MI_ENTER();
  if (!image || !pair_geom || !pair_thr || !out) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  if (num_pairs <= 0 || num_pairs > 1024) return MI_E_PARAM;
  if (mode != MI_BAD_RAW && mode != MI_BAD_SOFT && mode != MI_BAD_HARD) return MI_E_PARAM;
  const int tiles_x = ceil_div(w, DT), tiles_y = ceil_div(h, DT);
  const long long blocks = (long long)n * tiles_x * tiles_y;
  if (blocks > 0x7fffffffLL) return MI_E_SHAPE;
  hipLaunchKernelGGL(bad_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, image, h, w,
                     pair_geom, pair_thr, num_pairs, mode, temperature, out, tiles_x, tiles_y);
  return mi_launch_status();
}

extern "C" int mi_bad_dense_oriented(const float *image, const float *orientation, int n, int h, int w,
                                     const uint32_t *pair_geom, const float *pair_thr, int num_pairs, int mode,
                                     float temperature, float *out, mi_stream_t stream) {
  MI_ENTER();
  if (!image || !orientation || !pair_geom || !pair_thr || !out) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  if (num_pairs <= 0 || num_pairs > 1024) return MI_E_PARAM;
  if (mode != MI_BAD_RAW && mode != MI_BAD_SOFT && mode != MI_BAD_HARD) return MI_E_PARAM;
  const int tiles_x = ceil_div(w, OT), tiles_y = ceil_div(h, OT);
  const long long blocks = (long long)n * tiles_x * tiles_y;
  if (blocks > 0x7fffffffLL) return MI_E_SHAPE;
  const float scale_y = (float)(2.0 / ((double)(h - 1) + 1e-8));       // bad.py:162-163
  const float scale_x = (float)(2.0 / ((double)(w - 1) + 1e-8));
  hipLaunchKernelGGL(bad_dense_oriented_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, image,
                     orientation, h, w, pair_geom, pair_thr, num_pairs, mode, temperature, scale_y, scale_x, out,
                     tiles_x, tiles_y);
  return mi_launch_status();
}

extern "C" int mi_gather_descriptors(const float *descriptor_map, int batch, int d, int h, int w,
                                     const float *keypoints, int nk, int bilinear, float *out,
                                     mi_stream_t stream) {
  MI_ENTER();
  if (!descriptor_map || !keypoints || !out) return MI_E_NULL;
  if (batch <= 0 || d <= 0 || h <= 0 || w <= 0 || nk <= 0 || batch > 65535) return MI_E_SHAPE;
  hipLaunchKernelGGL(gather_desc_kernel, dim3(nk, batch), dim3(256), 0, (hipStream_t)stream, descriptor_map, d, h, w,
                     keypoints, nk, bilinear, out);
  return mi_launch_status();
}
