// K9: AKAZE detector pieces (BASELINE config 4).
// Semantics: reference pytorch_model/detector/akaze.py
//   :98-131  NonLinearDiffusion.forward  -- one explicit step: g = conv(L, sobel/8, zero pad);
//            mag = sqrt(gx^2+gy^2+1e-8); c = 1/(1+(mag/kappa)^2); flux = c*g;
//            div = conv(flux_x, sobel_x/8) + conv(flux_y, sobel_y/8) (zero pad); L += 0.25*div
//   :146-254 HessianDetector             -- Lxx, Lyy (/16), Lxy (/4) 3x3 convs (zero pad);
//            response = Lxx*Lyy - Lxy^2; keep where response == max over nms_size^2 (pool padding
//            = -inf) and response > threshold; clamp >= 0
//   :384-453 AKAZE.forward scale selection -- score = max over scales; orientation = mean of the
//            orientations of the scales that attain it
// All fp32, no FMA contraction.  The 3x3 stencils add their taps in row-major order, which is the
// order the reference's CPU conv produced when the golden vectors were recorded (the oracle, with
// the same order, reproduces tests/golden/akaze_pipeline.npz bit for bit); tap weights are powers
// of two, so scaling before or after the sum rounds identically.
// 64x16 tiles through LDS, four pixels per thread; the NMS window maximum is separable (rows, then columns).
#include "common.h"

#include <math.h>

namespace {

constexpr int AK_W = 64, AK_H = 16;      // tile; 256 threads, 4 rows of one column per thread
constexpr int AK_RPT = AK_H / 4;

__device__ __forceinline__ void tile_coords(int tiles_x, int tiles_y, int &img, int &x0, int &y0) {
  int bid = (int)blockIdx.x;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  img = bid / tiles_y;
  x0 = tx * AK_W;
  y0 = ty * AK_H;
}

// stage a (AK_H + 2*halo) x (AK_W + 2*halo) tile, `fill` outside the image
template <int HALO>
__device__ __forceinline__ void stage(const float *__restrict__ src, int h, int w, int x0, int y0, float fill,
                                      float (*tile)[AK_W + 2 * HALO]) {
  constexpr int SW = AK_W + 2 * HALO, SH = AK_H + 2 * HALO;
  for (int i = threadIdx.x; i < SW * SH; i += 256) {
    const int r = i / SW, c = i - r * SW;
    const int gy = y0 - HALO + r, gx = x0 - HALO + c;
    tile[r][c] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? src[(size_t)gy * w + gx] : fill;
  }
}

__global__ __launch_bounds__(256) void diffuse_kernel(const float *__restrict__ lin, int h, int w, float kappa,
                                                      float dt, float *__restrict__ lout, int tiles_x,
                                                      int tiles_y) {
  __shared__ float L[AK_H + 4][AK_W + 4];
  __shared__ float FX[AK_H + 2][AK_W + 2], FY[AK_H + 2][AK_W + 2];
  int img, x0, y0;
  tile_coords(tiles_x, tiles_y, img, x0, y0);
  const float *src = lin + (size_t)img * h * w;
  stage<2>(src, h, w, x0, y0, 0.0f, L);
  __syncthreads();
  // flux on the tile + 1 halo; zero outside the image (the second conv zero-pads the flux)
  for (int i = threadIdx.x; i < (AK_H + 2) * (AK_W + 2); i += 256) {
    const int r = i / (AK_W + 2), c = i - r * (AK_W + 2);
    const int gy = y0 - 1 + r, gx = x0 - 1 + c;
    float fx = 0.0f, fy = 0.0f;
    if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
      // 3x3 block at L[r..r+2][c..c+2]
      const float a = L[r][c], b = L[r][c + 1], cc = L[r][c + 2], d = L[r + 1][c], f = L[r + 1][c + 2],
                  g = L[r + 2][c], hh = L[r + 2][c + 1], k = L[r + 2][c + 2];
      // taps accumulated in row-major order (see the header note on summation order)
      const float gxv = (((((cc - a) - 2.0f * d) + 2.0f * f) - g) + k) * 0.125f;          // akaze.py:50-63,82
      const float gyv = ((((((-a) - 2.0f * b) - cc) + g) + 2.0f * hh) + k) * 0.125f;
      const float mag = sqrtf(gxv * gxv + gyv * gyv + 1e-8f);                             // :116
      const float q = mag / kappa;
      const float cond = 1.0f / (1.0f + q * q);                                           // :96
      fx = cond * gxv;
      fy = cond * gyv;
    }
    FX[r][c] = fx;
    FY[r][c] = fy;
  }
  __syncthreads();
  const int lx = threadIdx.x & (AK_W - 1), lyb = (threadIdx.x >> 6) * AK_RPT;
  const int gx = x0 + lx;
#pragma unroll
  for (int k = 0; k < AK_RPT; ++k) {
    const int ly = lyb + k, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    // divergence: sobel_x/8 on flux_x + sobel_y/8 on flux_y (:125-126)
    const float dx = (((((FX[ly][lx + 2] - FX[ly][lx]) - 2.0f * FX[ly + 1][lx]) + 2.0f * FX[ly + 1][lx + 2]) -
                       FX[ly + 2][lx]) + FX[ly + 2][lx + 2]) * 0.125f;
    const float dy = ((((((-FY[ly][lx]) - 2.0f * FY[ly][lx + 1]) - FY[ly][lx + 2]) + FY[ly + 2][lx]) +
                       2.0f * FY[ly + 2][lx + 1]) + FY[ly + 2][lx + 2]) * 0.125f;
    lout[((size_t)img * h + gy) * w + gx] = L[ly + 2][lx + 2] + dt * (dx + dy);          // :129
  }
}

__global__ __launch_bounds__(256) void hessian_kernel(const float *__restrict__ lin, int h, int w, float threshold,
                                                      int nms_half, float *__restrict__ scores, int tiles_x,
                                                      int tiles_y) {
  extern __shared__ float lds[];
  const int rh = AK_H + 2 * nms_half, rw = AK_W + 2 * nms_half;       // response tile with the NMS halo
  const int lh = rh + 2, lw = rw + 2;                                  // image tile with one more ring
  float *L = lds;                                                      // [lh][lw], zero outside (conv zero pad)
  float *R = lds + lh * lw;                                            // [rh][rw], -inf outside (pool padding)
  int img, x0, y0;
  tile_coords(tiles_x, tiles_y, img, x0, y0);
  const float *src = lin + (size_t)img * h * w;
  for (int i = threadIdx.x; i < lh * lw; i += 256) {
    const int r = i / lw, c = i - r * lw;
    const int gy = y0 - nms_half - 1 + r, gx = x0 - nms_half - 1 + c;
    L[i] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? src[(size_t)gy * w + gx] : 0.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < rh * rw; i += 256) {
    const int r = i / rw, c = i - r * rw;
    const int gy = y0 - nms_half + r, gx = x0 - nms_half + c;
    float resp = -INFINITY;
    if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
      const float *p = L + r * lw + c;                                 // 3x3 block at rows r..r+2, cols c..c+2
      const float a = p[0], b = p[1], cc = p[2], d = p[lw], e = p[lw + 1], f = p[lw + 2], g = p[2 * lw],
                  hh = p[2 * lw + 1], k = p[2 * lw + 2];
      const float lxx = ((((((((a - 2.0f * b) + cc) + 2.0f * d) - 4.0f * e) + 2.0f * f) + g) - 2.0f * hh) + k) *
                        0.0625f;                                                                             // :153-157
      const float lyy = ((((((((a + 2.0f * b) + cc) - 2.0f * d) - 4.0f * e) - 2.0f * f) + g) + 2.0f * hh) + k) *
                        0.0625f;                                                                             // :159-163
      const float lxy = (((a - cc) - g) + k) * 0.25f;                                                        // :165-169
      resp = lxx * lyy - lxy * lxy;                                                                          // :196
    }
    R[i] = resp;
  }
  __syncthreads();
  // window maximum, separable: rows of R into H2 [rh][AK_W], then down the columns
  float *H2 = R + rh * rw;
  for (int i = threadIdx.x; i < rh * AK_W; i += 256) {
    const int r = i / AK_W, c = i - r * AK_W;
    const float *p = R + r * rw + c;
    float mx = p[0];
    for (int d = 1; d <= 2 * nms_half; ++d) mx = fmaxf(mx, p[d]);
    H2[i] = mx;
  }
  __syncthreads();
  const int lx = threadIdx.x & (AK_W - 1), lyb = (threadIdx.x >> 6) * AK_RPT;
  const int gx = x0 + lx;
#pragma unroll
  for (int k = 0; k < AK_RPT; ++k) {
    const int ly = lyb + k, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const float resp = R[(ly + nms_half) * rw + lx + nms_half];
    float mx = -INFINITY;
    for (int d = 0; d <= 2 * nms_half; ++d) mx = fmaxf(mx, H2[(ly + d) * AK_W + lx]);
    const float keep = (resp == mx && resp > threshold) ? 1.0f : 0.0f;                                       // :223,:245
    scores[((size_t)img * h + gy) * w + gx] = fmaxf(resp * keep, 0.0f);                                      // :249-252
  }
}

// score = max over scales; orientation = mean orientation of the scales attaining it (:442-451)
__global__ __launch_bounds__(256) void combine_kernel(const float *__restrict__ scores_s,
                                                      const float *__restrict__ oris_s, int nscales,
                                                      size_t plane, float *__restrict__ scores,
                                                      float *__restrict__ oris) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= plane) return;
  float mx = -INFINITY;
  for (int s = 0; s < nscales; ++s) mx = fmaxf(mx, scores_s[(size_t)s * plane + i]);
  float cnt = 0.0f;
  for (int s = 0; s < nscales; ++s) cnt += (scores_s[(size_t)s * plane + i] == mx) ? 1.0f : 0.0f;
  cnt = fmaxf(cnt, 1.0f);
  scores[i] = mx;
  if (!oris) return;                                   // scores only: oris_s may be NULL too
  float acc = 0.0f;
  for (int s = 0; s < nscales; ++s) {
    const float m = ((scores_s[(size_t)s * plane + i] == mx) ? 1.0f : 0.0f) / cnt;
    acc += oris_s[(size_t)s * plane + i] * m;
  }
  oris[i] = acc;
}

// the same selection, only at the keypoints: scores_s (S,n,h,w), theta_s (S,n,k) -> theta (n,k)
__global__ __launch_bounds__(256) void combine_kp_kernel(const float *__restrict__ scores_s,
                                                         const float *__restrict__ theta_s, int nscales, int n,
                                                         int h, int w, const float *__restrict__ kpts, int k,
                                                         float *__restrict__ theta) {
  const int flat = blockIdx.x * 256 + threadIdx.x;
  if (flat >= n * k) return;
  const int img = flat / k;
  const float ky = fminf(fmaxf(kpts[(size_t)flat * 2 + 0], 0.0f), (float)(h - 1));
  const float kx = fminf(fmaxf(kpts[(size_t)flat * 2 + 1], 0.0f), (float)(w - 1));
  const float sy = (float)(2.0 / ((double)(h - 1) + 1e-8)), sx = (float)(2.0 / ((double)(w - 1) + 1e-8));
  const float ny = ((ky * sy - 1.0f + 1.0f) / 2.0f) * (float)(h - 1);
  const float nx = ((kx * sx - 1.0f + 1.0f) / 2.0f) * (float)(w - 1);
  const int cy = (int)nearbyintf(fminf(fmaxf(ny, 0.0f), (float)(h - 1)));
  const int cx = (int)nearbyintf(fminf(fmaxf(nx, 0.0f), (float)(w - 1)));
  const size_t plane = (size_t)n * h * w;
  const size_t pix = ((size_t)img * h + cy) * w + cx;
  float mx = -INFINITY;
  for (int s = 0; s < nscales; ++s) mx = fmaxf(mx, scores_s[(size_t)s * plane + pix]);
  float cnt = 0.0f;
  for (int s = 0; s < nscales; ++s) cnt += (scores_s[(size_t)s * plane + pix] == mx) ? 1.0f : 0.0f;
  cnt = fmaxf(cnt, 1.0f);
  float acc = 0.0f;
  for (int s = 0; s < nscales; ++s) {
    const float m = ((scores_s[(size_t)s * plane + pix] == mx) ? 1.0f : 0.0f) / cnt;
    acc += theta_s[(size_t)s * n * k + flat] * m;
  }
  theta[flat] = acc;
}

int grid_for(int n, int h, int w, int &tiles_x, int &tiles_y) {
  tiles_x = ceil_div(w, AK_W);
  tiles_y = ceil_div(h, AK_H);
  const long long blocks = (long long)n * tiles_x * tiles_y;
  return blocks > 0x7fffffffLL ? -1 : (int)blocks;
}

}  // namespace

extern "C" int mi_akaze_diffuse(const float *l_in, int n, int h, int w, float kappa, float dt, float *l_out,
                                mi_stream_t stream) {
  MI_ENTER();
  if (!l_in || !l_out || l_in == l_out) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  if (!(kappa > 0.0f)) return MI_E_PARAM;
  int tx, ty;
  const int blocks = grid_for(n, h, w, tx, ty);
  if (blocks < 0) return MI_E_SHAPE;
  hipLaunchKernelGGL(diffuse_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, l_in, h, w, kappa, dt, l_out, tx,
                     ty);
  return mi_launch_status();
}

extern "C" int mi_akaze_hessian_scores(const float *l, int n, int h, int w, float threshold, int nms_size,
                                       float *scores, mi_stream_t stream) {
  MI_ENTER();
  if (!l || !scores) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  if (nms_size <= 0 || (nms_size & 1) == 0 || nms_size > 15) return MI_E_PARAM;
  int tx, ty;
  const int blocks = grid_for(n, h, w, tx, ty);
  if (blocks < 0) return MI_E_SHAPE;
  const int nh = nms_size / 2;
  const size_t lds = ((size_t)(AK_H + 2 * nh + 2) * (AK_W + 2 * nh + 2) + (size_t)(AK_H + 2 * nh) * (AK_W + 2 * nh) +
                      (size_t)(AK_H + 2 * nh) * AK_W) * 4;
  hipLaunchKernelGGL(hessian_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, l, h, w, threshold, nh, scores,
                     tx, ty);
  return mi_launch_status();
}

extern "C" int mi_akaze_combine(const float *scale_scores, const float *scale_orientations, int num_scales, int n,
                                int h, int w, float *scores, float *orientations, mi_stream_t stream) {
  MI_ENTER();
  if (!scale_scores || !scores) return MI_E_NULL;
  if (orientations && !scale_orientations) return MI_E_NULL;
  if (num_scales <= 0 || n <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  const size_t plane = (size_t)n * h * w;
  const size_t blocks = (plane + 255) / 256;
  if (blocks > 0x7fffffffULL) return MI_E_SHAPE;
  hipLaunchKernelGGL(combine_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, scale_scores,
                     scale_orientations, num_scales, plane, scores, orientations);
  return mi_launch_status();
}

extern "C" int mi_akaze_orientation_at_keypoints(const float *scale_scores, const float *scale_theta,
                                                 int num_scales, int n, int h, int w, const float *keypoints,
                                                 int k, float *theta, mi_stream_t stream) {
  MI_ENTER();
  if (!scale_scores || !scale_theta || !keypoints || !theta) return MI_E_NULL;
  if (num_scales <= 0 || n <= 0 || h <= 0 || w <= 0 || k <= 0 || (long long)n * k > 0x7fffffffLL) return MI_E_SHAPE;
  hipLaunchKernelGGL(combine_kp_kernel, dim3(ceil_div(n * k, 256)), dim3(256), 0, (hipStream_t)stream, scale_scores,
                     scale_theta, num_scales, n, h, w, keypoints, k, theta);
  return mi_launch_status();
}
