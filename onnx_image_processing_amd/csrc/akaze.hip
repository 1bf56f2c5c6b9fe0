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
#include "akaze_math.h"
#include "hooks.h"

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

// scores only (what the matchers need: the orientation is evaluated at the keypoints), four pixels per thread with
// 16-byte loads: (S + 1) * 4 bytes per pixel, HBM-bound
__global__ __launch_bounds__(256) void combine_scores4_kernel(const float4 *__restrict__ scores_s, int nscales,
                                                              size_t plane4, float4 *__restrict__ scores) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= plane4) return;
  float4 mx = scores_s[i];
  for (int s = 1; s < nscales; ++s) {
    const float4 v = scores_s[(size_t)s * plane4 + i];
    mx.x = fmaxf(mx.x, v.x); mx.y = fmaxf(mx.y, v.y); mx.z = fmaxf(mx.z, v.z); mx.w = fmaxf(mx.w, v.w);
  }
  scores[i] = mx;
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

// ---- one launch per scale: ITERS diffusion steps, the Hessian response and its NMS on one LDS-resident tile -----
// The per-step form above moves 8 B/px through HBM per diffusion step and 8 more for the Hessian pass (32 B/px per
// scale, 4 launches); here an output tile is staged once with a halo of 2*ITERS + 1 + NH pixels, the steps run
// in place in LDS (each step's valid region shrinks by 2: gradients need one ring, the divergence of the flux
// another), the diffused tile is written once and the Hessian determinant + window-equality NMS are evaluated on the
// tile still in LDS: 12 B/px per scale (read L, write L, write scores), one launch.  The halo is recomputed per tile
// (about 1.5x the arithmetic), every expression is the per-step kernels' (same operation order), so the maps equal
// theirs bit for bit.
// (build-time overridable for sweeps on the GPU box: MI_BUILD_DEFINES, build.py.  Measured, 128 pairs per step:
//  8 waves x 48 rows 3.11 ms; 8 x 32 3.37; 8 x 40 3.37; 8 x 56 3.86; 8 x 64 3.78; 4 x 32 3.68; 4 x 48 3.92;
//  6 x 48 4.62; 10 x 48 3.96; 12 x 48 3.62; 16 x 32 4.19; 16 x 48 3.49)
#ifndef MI_AS_H
#define MI_AS_H 48
#endif
#ifndef MI_AS_WAVES
#define MI_AS_WAVES 8
#endif
constexpr int AS_H = MI_AS_H;            // output tile height; its width is 64 - 2 * halo (see the kernel)
constexpr int AS_WAVES = MI_AS_WAVES;    // waves per workgroup: each phase splits its rows into this many blocks

// (the exact-rounding helpers ak_sqrt / ak_div / ak_div_by / ak_rcp live in akaze_math.h)

template <int ITERS, int NH>
__global__ __launch_bounds__(64 * AS_WAVES) void akaze_scale_kernel(const float *__restrict__ lin, int h, int w, float kappa,
                                                          float dt, float threshold, float *__restrict__ lout,
                                                          float *__restrict__ scores, int tiles_x, int tiles_y) {
  constexpr int HALO = 2 * ITERS + 1 + NH;
  // the staged tile is exactly one wave wide (64 columns: every row operation uses all 64 lanes, one row per wave
  // instruction, no index arithmetic), so the OUTPUT tile is 64 - 2 * HALO columns wide
  constexpr int TW = 64 - 2 * HALO;
  constexpr int LW = 64, LH = AS_H + 2 * HALO;
  __shared__ float L[LH][LW];              // the image tile, zero outside the image (both convolutions zero-pad)
  __shared__ float FX[LH][LW], FY[LH][LW]; // the flux (same coordinates); later the response and its row maxima
  int bid = (int)blockIdx.x;
  const int txi = bid % tiles_x;
  bid /= tiles_x;
  const int tyi = bid % tiles_y;
  const int img = bid / tiles_y;
  const int x0 = txi * TW - HALO, y0 = tyi * AS_H - HALO;            // global coordinates of L[0][0]
  const float *src = lin + (size_t)img * h * w;
  // thread = (column tx, row group ty): a wave handles one staged row per instruction, rows ty, ty + 4, ...
  // (ty is wave-uniform: kept in an SGPR so that every row index, row bound and row predicate below is scalar work)
  const int tx = threadIdx.x & 63, ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the image rows / columns the tile covers, in tile coordinates: [ry0, ry1) x [cx0, cx1) lies inside the image
  const int ry0 = max(0, -y0), ry1 = min(LH, h - y0), cx0 = max(0, -x0), cx1 = min(LW, w - x0);
  // staging: ALL of a lane's rows are requested before the first one is used -- branch-free loads from clamped (always
  // valid) addresses, the zero padding applied afterwards.  With the load inside the `inside the image` branch every row
  // was its own round trip to memory: 9 dependent round trips = 10.4 k of a tile's 43.8 k clocks (in-kernel stamps).
  {
    constexpr int NST = (LH + AS_WAVES - 1) / AS_WAVES;
    const float *col = src + clampi(x0 + tx, 0, w - 1);
    const bool cin = tx >= cx0 && tx < cx1;
    float v[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) v[q] = col[(size_t)clampi(y0 + ty + q * AS_WAVES, 0, h - 1) * w];
#pragma unroll
    for (int q = 0; q < NST; ++q) {
      const int r = ty + q * AS_WAVES;
      if (r < LH) L[r][tx] = (cin && r >= ry0 && r < ry1) ? v[q] : 0.0f;
    }
  }
  __syncthreads();
  // Row blocks: every phase hands each of the 4 waves a contiguous block of rows and walks down it with the rows
  // above and below in registers (3 new LDS reads per flux value instead of 8, 5 + 1 per update instead of 12 + 1),
  // the next row's loads issued before the current row's arithmetic.
  const int c = tx;
  const float rkappa = 1.0f / kappa;                  // IEEE division: the correctly rounded reciprocal (ak_div_by)
#pragma unroll
  for (int s = 0; s < ITERS; ++s) {
    // flux on rows / columns [2s+1, L? - 2s - 1): needs L one ring further out
    {
      const int f0 = 2 * s + 1, nrows = LH - 2 * f0, per = (nrows + AS_WAVES - 1) / AS_WAVES;
      const int rbeg = f0 + ty * per, rend = min(rbeg + per, LH - f0);
      if (c >= f0 && c < LW - f0 && rbeg < rend) {
        const bool cin = c >= cx0 && c < cx1;
        float t0 = L[rbeg - 1][c - 1], t1 = L[rbeg - 1][c], t2 = L[rbeg - 1][c + 1];
        float m0 = L[rbeg][c - 1], m1 = L[rbeg][c], m2 = L[rbeg][c + 1];
        float b0 = L[rbeg + 1][c - 1], b1 = L[rbeg + 1][c], b2 = L[rbeg + 1][c + 1];
        // straight-line body (no branch per row: the scheduler can overlap one row's square root / reciprocal chain with
        // the next row's stencil): the row below the next one is always fetched (clamped to the tile), the flux is
        // computed for every lane and row and zeroed by a select where the pixel lies outside the image
        for (int r = rbeg; r < rend; ++r) {
          const int rn = min(r + 2, LH - 1);
          const float n0 = L[rn][c - 1], n1 = L[rn][c], n2 = L[rn][c + 1];                          // next row, early
          const bool inside = cin && r >= ry0 && r < ry1;                   // the flux is zero-padded outside the image
          const float gxv = (((((t2 - t0) - 2.0f * m0) + 2.0f * m2) - b0) + b2) * 0.125f;          // akaze.py:50-63,82
          const float gyv = ((((((-t0) - 2.0f * t1) - t2) + b0) + 2.0f * b1) + b2) * 0.125f;
          const float mag = ak_sqrt_fp<1>(gxv * gxv + gyv * gyv + 1e-8f);                       // :116
          const float q = ak_div_by(mag, kappa, rkappa);
          const float cond = ak_rcp(1.0f + q * q);                                              // :96
          const float fx = inside ? cond * gxv : 0.0f;
          const float fy = inside ? cond * gyv : 0.0f;
          FX[r][c] = fx;
          FY[r][c] = fy;
          t0 = m0; t1 = m1; t2 = m2;
          m0 = b0; m1 = b1; m2 = b2;
          b0 = n0; b1 = n1; b2 = n2;
        }
      }
    }
    __syncthreads();
    // L += dt * div(flux) on rows / columns [2s+2, L? - 2s - 2), in place (only pixels of the image evolve)
    {
      const int u0 = 2 * s + 2, nrows = LH - 2 * u0, per = (nrows + AS_WAVES - 1) / AS_WAVES;
      const int rbeg = max(u0 + ty * per, ry0), rend = min(min(u0 + (ty + 1) * per, LH - u0), ry1);
      if (c >= max(u0, cx0) && c < min(LW - u0, cx1) && rbeg < rend) {
        float xt0 = FX[rbeg - 1][c - 1], xt2 = FX[rbeg - 1][c + 1];
        float yt0 = FY[rbeg - 1][c - 1], yt1 = FY[rbeg - 1][c], yt2 = FY[rbeg - 1][c + 1];
        float xm0 = FX[rbeg][c - 1], xm2 = FX[rbeg][c + 1];
        float ym0 = FY[rbeg][c - 1], ym1 = FY[rbeg][c], ym2 = FY[rbeg][c + 1];
        for (int r = rbeg; r < rend; ++r) {
          const float xb0 = FX[r + 1][c - 1], xb2 = FX[r + 1][c + 1];
          const float yb0 = FY[r + 1][c - 1], yb1 = FY[r + 1][c], yb2 = FY[r + 1][c + 1];
          const float lc = L[r][c];
          const float dx = (((((xt2 - xt0) - 2.0f * xm0) + 2.0f * xm2) - xb0) + xb2) * 0.125f;       // :125-126
          const float dy = ((((((-yt0) - 2.0f * yt1) - yt2) + yb0) + 2.0f * yb1) + yb2) * 0.125f;
          L[r][c] = lc + dt * (dx + dy);                                                        // :129
          xt0 = xm0; xt2 = xm2; yt0 = ym0; yt1 = ym1; yt2 = ym2;
          xm0 = xb0; xm2 = xb2; ym0 = yb0; ym1 = yb1; ym2 = yb2;
        }
      }
    }
    __syncthreads();
  }
  // the diffused tile (AS_W = 64 columns: one column chunk) and the Hessian determinant on the tile + NH (into FX;
  // -inf outside the image: the pool's padding)
  for (int r = HALO + ty; r < min(HALO + AS_H, ry1); r += AS_WAVES) {
    const int c = tx;
    if (c >= HALO && c < HALO + TW && c < cx1) lout[((size_t)img * h + (y0 + r)) * w + (x0 + c)] = L[r][c];
  }
  constexpr int R0 = 2 * ITERS + 1;
  // Hessian determinant and its row-window maximum in ONE phase: a wave walks a contiguous block of rows with the 3 x 3
  // window of L sliding through registers (3 new LDS reads per row instead of 9), writes the row of responses to FX and
  // reads the row's neighbours straight back -- the row was written by this very wave, whose LDS operations execute in
  // order, so no workgroup barrier is needed; the empty asm keeps the COMPILER from moving the reads above the write
  // (per thread the addresses differ, so it otherwise may).
  {
    constexpr int nrows = LH - 2 * R0, per = (nrows + AS_WAVES - 1) / AS_WAVES;
    const int rbeg = R0 + ty * per, rend = min(rbeg + per, LH - R0);
    if (c >= R0 && c < LW - R0 && rbeg < rend) {
      const bool cin = c >= cx0 && c < cx1;
      const bool ctile = c >= HALO && c < HALO + TW;
      float t0 = L[rbeg - 1][c - 1], t1 = L[rbeg - 1][c], t2 = L[rbeg - 1][c + 1];
      float m0 = L[rbeg][c - 1], m1 = L[rbeg][c], m2 = L[rbeg][c + 1];
      for (int r = rbeg; r < rend; ++r) {
        const float b0 = L[r + 1][c - 1], b1 = L[r + 1][c], b2 = L[r + 1][c + 1];
        const float lxx = ((((((((t0 - 2.0f * t1) + t2) + 2.0f * m0) - 4.0f * m1) + 2.0f * m2) + b0) - 2.0f * b1) + b2) * 0.0625f;
        const float lyy = ((((((((t0 + 2.0f * t1) + t2) - 2.0f * m0) - 4.0f * m1) - 2.0f * m2) + b0) + 2.0f * b1) + b2) * 0.0625f;
        const float lxy = (((t0 - t2) - b0) + b2) * 0.25f;
        const float det = lxx * lyy - lxy * lxy;                                                  // :196
        const float resp = (cin && r >= ry0 && r < ry1) ? det : -INFINITY;   // -inf outside the image: the pool's padding
        FX[r][c] = resp;
        asm volatile("" ::: "memory");
        if (ctile) {
          float mx = FX[r][c - NH];
#pragma unroll
          for (int d = 1; d <= 2 * NH; ++d) mx = fmaxf(mx, FX[r][c - NH + d]);
          FY[r][c] = mx;
        }
        asm volatile("" ::: "memory");
        t0 = m0; t1 = m1; t2 = m2;
        m0 = b0; m1 = b1; m2 = b2;
      }
    }
  }
  __syncthreads();
  for (int r = HALO + ty; r < min(HALO + AS_H, ry1); r += AS_WAVES) {
    const int c = tx;
    if (c < HALO || c >= HALO + TW || c >= cx1) continue;
    const float resp = FX[r][c];
    float mx = -INFINITY;
#pragma unroll
    for (int d = -NH; d <= NH; ++d) mx = fmaxf(mx, FY[r + d][c]);
    const float keep = (resp == mx && resp > threshold) ? 1.0f : 0.0f;                       // :223,:245
    scores[((size_t)img * h + (y0 + r)) * w + (x0 + c)] = fmaxf(resp * keep, 0.0f);          // :249-252
  }
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
  if (!orientations && plane % 4 == 0 && ((uintptr_t)scale_scores | (uintptr_t)scores) % 16 == 0) {
    const size_t plane4 = plane / 4, blocks4 = (plane4 + 255) / 256;
    if (blocks4 > 0x7fffffffULL) return MI_E_SHAPE;
    hipLaunchKernelGGL(combine_scores4_kernel, dim3((unsigned)blocks4), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(scale_scores), num_scales, plane4,
                       reinterpret_cast<float4 *>(scores));
    return mi_launch_status();
  }
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

// ---- AKAZE.forward's per-scale body in one launch (akaze.py:430-440): l_out = NonLinearDiffusion(l_in) with
// `iterations` steps, scores = HessianDetector(l_out).  Fused for iterations 1..3 and nms_size 3 / 5 / 7 (the
// reference's defaults are 3 and 5); other values run the per-step kernels through a scratch image `tmp` (n*h*w
// floats, only needed then; may be NULL when the fused form applies -- mi_akaze_scale_fused says which).
extern "C" int mi_akaze_scale_fused(int iterations, int nms_size) {
  return iterations >= 1 && iterations <= 3 && (nms_size == 3 || nms_size == 5 || nms_size == 7);
}

extern "C" int mi_akaze_scale(const float *l_in, int n, int h, int w, int iterations, float kappa, float dt,
                              float threshold, int nms_size, float *l_out, float *scores, float *tmp,
                              mi_stream_t stream) {
  MI_ENTER();
  if (!l_in || !l_out || !scores || l_in == l_out) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  if (iterations <= 0 || !mi_akaze_kappa_ok(kappa) || nms_size <= 0 || (nms_size & 1) == 0 || nms_size > 15) return MI_E_PARAM;
  hipStream_t s = (hipStream_t)stream;
  // the streaming rolling-window form (akaze_stream.hip): even widths, 8-byte aligned maps, nms_size 3 / 5
  if (MI_HOOK(akaze_impl, 0) == 0 && mi_akaze_stream_supported(h, w, iterations, nms_size, l_in, l_out, scores))
    return mi_akaze_scale_stream(l_in, nullptr, n, n, h, w, iterations, kappa, dt, threshold, nms_size, l_out, scores, 0,
                                 nullptr, 0, nullptr, stream);
  if (mi_akaze_scale_fused(iterations, nms_size)) {
    const int halo = 2 * iterations + 1 + nms_size / 2;
    const int tx = ceil_div(w, 64 - 2 * halo), ty = ceil_div(h, AS_H);
    const long long blocks = (long long)n * tx * ty;
    if (blocks > 0x7fffffffLL) return MI_E_SHAPE;
#define AKS(I, NHALF) hipLaunchKernelGGL((akaze_scale_kernel<I, NHALF>), dim3((unsigned)blocks), dim3(64 * AS_WAVES), 0, s, l_in, h, w, kappa, dt, threshold, l_out, scores, tx, ty)
    const int nh = nms_size / 2;
    if (iterations == 1) { if (nh == 1) AKS(1, 1); else if (nh == 2) AKS(1, 2); else AKS(1, 3); }
    else if (iterations == 2) { if (nh == 1) AKS(2, 1); else if (nh == 2) AKS(2, 2); else AKS(2, 3); }
    else { if (nh == 1) AKS(3, 1); else if (nh == 2) AKS(3, 2); else AKS(3, 3); }
#undef AKS
    return mi_launch_status();
  }
  // general parameters: the per-step kernels, ping-ponging l_out and tmp so that the last step lands in l_out
  if (iterations > 1 && !tmp) return MI_E_NULL;
  const float *cur = l_in;
  for (int i = 0; i < iterations; ++i) {
    float *dst = ((iterations - 1 - i) % 2 == 0) ? l_out : tmp;
    const int e = mi_akaze_diffuse(cur, n, h, w, kappa, dt, dst, stream);
    if (e != MI_OK) return e;
    cur = dst;
  }
  return mi_akaze_hessian_scores(l_out, n, h, w, threshold, nms_size, scores, stream);
}

// ---- the FIRST scale of two equally shaped batches behind one launch (image1 / image2 of a matcher): l_out and scores
// hold the 2 * per_set images stacked, so every later scale (and NMS / top-k) runs on one batch of twice the size --
// the streaming kernel then cuts an image into half as many row chunks (half the warm-up rows per output row)
extern "C" int mi_akaze_scale_sets(const float *l_in_a, const float *l_in_b, int per_set, int h, int w, int iterations,
                                   float kappa, float dt, float threshold, int nms_size, float *l_out, float *scores,
                                   float *tmp, mi_stream_t stream) {
  MI_ENTER();
  if (!l_in_a || !l_in_b || !l_out || !scores) return MI_E_NULL;
  if (per_set <= 0 || h <= 0 || w <= 0 || per_set > 0x3fffffff) return MI_E_SHAPE;
  if (iterations <= 0 || !mi_akaze_kappa_ok(kappa) || nms_size <= 0 || (nms_size & 1) == 0 || nms_size > 15) return MI_E_PARAM;
  const size_t half = (size_t)per_set * h * w;
  if (l_in_a == l_out || l_in_b == l_out + half) return MI_E_NULL;
  if (MI_HOOK(akaze_impl, 0) == 0 && mi_akaze_stream_supported(h, w, iterations, nms_size, l_in_a, l_out, scores) &&
      ((uintptr_t)l_in_b & 7u) == 0)
    return mi_akaze_scale_stream(l_in_a, l_in_b, per_set, 2 * per_set, h, w, iterations, kappa, dt, threshold, nms_size, l_out,
                                 scores, 0, nullptr, 0, nullptr, stream);
  const int e = mi_akaze_scale(l_in_a, per_set, h, w, iterations, kappa, dt, threshold, nms_size, l_out, scores, tmp, stream);
  if (e != MI_OK) return e;
  return mi_akaze_scale(l_in_b, per_set, h, w, iterations, kappa, dt, threshold, nms_size, l_out + half, scores + half, tmp,
                        stream);
}

// ---- the LAST scale with AKAZE.forward's selection across scales folded in (akaze.py:436-451) ---------------------
namespace {
// best = max(prev maps, cur); attain bit s = prev map s reaches it, bit num_prev = cur does.  cur may alias best.
__global__ __launch_bounds__(256) void select_kernel(const float *__restrict__ prev, int num_prev, size_t plane,
                                                     const float *cur, float *best, uint8_t *__restrict__ attain) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= plane) return;
  const float c = cur[i];
  float mx = c;
  for (int s = 0; s < num_prev; ++s) mx = fmaxf(mx, prev[(size_t)s * plane + i]);
  unsigned a = (c == mx) ? (1u << num_prev) : 0u;
  for (int s = 0; s < num_prev; ++s) a |= (prev[(size_t)s * plane + i] == mx) ? (1u << s) : 0u;
  best[i] = mx;
  attain[i] = (uint8_t)a;
}

// orientation = mean of the per-scale keypoint orientations over the scales that attain the maximum (:442-451)
__global__ __launch_bounds__(256) void attain_kp_kernel(const uint8_t *__restrict__ attain,
                                                        const float *__restrict__ theta_s, int nscales, int n, int h,
                                                        int w, const float *__restrict__ kpts, int k,
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
  const unsigned a = attain[((size_t)img * h + cy) * w + cx];
  const float cnt = fmaxf((float)__popc(a & ((1u << nscales) - 1u)), 1.0f);
  float acc = 0.0f;
  for (int s = 0; s < nscales; ++s) {
    const float m = (((a >> s) & 1u) ? 1.0f : 0.0f) / cnt;
    acc += theta_s[(size_t)s * n * k + flat] * m;
  }
  theta[flat] = acc;
}
}  // namespace

extern "C" int mi_akaze_scale_select(const float *l_in, int n, int h, int w, int iterations, float kappa, float dt,
                                     float threshold, int nms_size, float *l_out, const float *prev_scores,
                                     int num_prev, float *best, uint8_t *attain, float *tmp, mi_stream_t stream) {
  MI_ENTER();
  if (!l_in || !l_out || !best || !attain || l_in == l_out || (num_prev > 0 && !prev_scores)) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  if (num_prev < 0 || num_prev > 7) return MI_E_PARAM;
  if (iterations <= 0 || !mi_akaze_kappa_ok(kappa) || nms_size <= 0 || (nms_size & 1) == 0 || nms_size > 15) return MI_E_PARAM;
  if (MI_HOOK(akaze_impl, 0) == 0 && mi_akaze_stream_supported(h, w, iterations, nms_size, l_in, l_out, best) &&
      num_prev <= MI_AKAZE_STREAM_MAX_PREV && ((uintptr_t)prev_scores & 7u) == 0 && ((uintptr_t)attain & 1u) == 0)
    return mi_akaze_scale_stream(l_in, nullptr, n, n, h, w, iterations, kappa, dt, threshold, nms_size, l_out, best, 1,
                                 prev_scores, num_prev, attain, stream);
  // general parameters: this scale's map into `best`, then the selection in place
  const int e = mi_akaze_scale(l_in, n, h, w, iterations, kappa, dt, threshold, nms_size, l_out, best, tmp, stream);
  if (e != MI_OK) return e;
  const size_t plane = (size_t)n * h * w, blocks = (plane + 255) / 256;
  if (blocks > 0x7fffffffULL) return MI_E_SHAPE;
  hipLaunchKernelGGL(select_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, prev_scores, num_prev, plane,
                     best, best, attain);
  return mi_launch_status();
}

extern "C" int mi_akaze_orientation_from_attain(const uint8_t *attain, const float *scale_theta, int num_scales, int n,
                                                int h, int w, const float *keypoints, int k, float *theta,
                                                mi_stream_t stream) {
  MI_ENTER();
  if (!attain || !scale_theta || !keypoints || !theta) return MI_E_NULL;
  if (num_scales <= 0 || num_scales > 8 || n <= 0 || h <= 0 || w <= 0 || k <= 0 || (long long)n * k > 0x7fffffffLL) return MI_E_SHAPE;
  hipLaunchKernelGGL(attain_kp_kernel, dim3(ceil_div(n * k, 256)), dim3(256), 0, (hipStream_t)stream, attain, scale_theta,
                     num_scales, n, h, w, keypoints, k, theta);
  return mi_launch_status();
}
