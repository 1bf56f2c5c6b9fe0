// K8: orientation from Gaussian-weighted first moments.
// Semantics: reference pytorch_model/orientation/angle_estimation.py:86-172 (AngleEstimator):
//   moments = conv2d(image, moment_kernels (2,1,ps,ps), padding = ps//2)   [ZERO padding]
//   angle   = atan2(m01, m10),  m10 = sum x*G*I,  m01 = sum y*G*I
// The weights are the module's own `moment_kernels` buffer (built on the host exactly as the
// reference builds it), so both sides multiply by bit-identical fp32 weights; only the fp32
// summation order differs (the reference's is oneDNN's), hence tolerance parity.
//
// Two entry points:
//  * mi_angle_map: the dense (n,1,h,w) map AngleEstimator.forward returns.  Tile 64x16 in LDS
//    (+halo), each thread 4 adjacent pixels, the weight planes in LDS.
//  * mi_angle_at_keypoints: only what the matching pipeline consumes -- the angle at the K
//    keypoints (the reference samples its dense map there with grid_sample(nearest),
//    descriptor/bad.py:487-500).  One wave per keypoint, 4 pixels of the patch per lane.
#include "common.h"

#include <math.h>

namespace {

constexpr int AT_W = 64, AT_H = 16;   // dense tile
constexpr int MAX_PS = 31;

__global__ __launch_bounds__(256) void angle_map_kernel(const float *__restrict__ image, int h, int w, int ps,
                                                        const float *__restrict__ weights,
                                                        float *__restrict__ angle, int tiles_x, int tiles_y) {
  extern __shared__ float lds[];
  const int half = ps / 2;
  const int sw = AT_W + 2 * half, sh = AT_H + 2 * half;
  float *tile = lds;                       // [sh][sw], zero outside the image
  float *wx = lds + sh * sw;               // [ps*ps] x-moment weights
  float *wy = wx + ps * ps;                // [ps*ps] y-moment weights
  const int t = threadIdx.x;
  int bid = (int)blockIdx.x;
  const int tx_tile = bid % tiles_x;
  bid /= tiles_x;
  const int ty_tile = bid % tiles_y;
  const int img = bid / tiles_y;
  const int x0 = tx_tile * AT_W, y0 = ty_tile * AT_H;
  const float *im = image + (size_t)img * h * w;
  for (int i = t; i < sh * sw; i += 256) {
    const int r = i / sw, c = i - r * sw;
    const int gy = y0 - half + r, gx = x0 - half + c;
    tile[i] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? im[(size_t)gy * w + gx] : 0.0f;
  }
  for (int i = t; i < ps * ps; i += 256) { wx[i] = weights[i]; wy[i] = weights[ps * ps + i]; }
  __syncthreads();
  const int lx = (t & 15) * 4, ly = t >> 4;          // 16 x 16 threads, 4 pixels each
  float m10[4] = {0.f, 0.f, 0.f, 0.f}, m01[4] = {0.f, 0.f, 0.f, 0.f};
  for (int dy = 0; dy < ps; ++dy) {
    const float *row = tile + (ly + dy) * sw + lx;
    const float *rx = wx + dy * ps, *ry = wy + dy * ps;
    for (int dx = 0; dx < ps; ++dx) {
      const float a = rx[dx], b = ry[dx];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = row[dx + q];
        m10[q] += a * v;
        m01[q] += b * v;
      }
    }
  }
  const int gy = y0 + ly;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int gx = x0 + lx + q;
    if (gy < h && gx < w) angle[((size_t)img * h + gy) * w + gx] = atan2f(m01[q], m10[q]);   // :170
  }
}

// the pixel the reference's grid_sample(nearest) reads for a keypoint: same normalise / un-normalise / round-half-even
// arithmetic as the descriptor's centres (descriptor/bad.py:464-465,487-500)
__device__ __forceinline__ void nearest_pixel(const float *__restrict__ kpts, int flat, int h, int w, int &cy, int &cx) {
  const float ky = fminf(fmaxf(kpts[(size_t)flat * 2 + 0], 0.0f), (float)(h - 1));
  const float kx = fminf(fmaxf(kpts[(size_t)flat * 2 + 1], 0.0f), (float)(w - 1));
  const float sy = (float)(2.0 / ((double)(h - 1) + 1e-8)), sx = (float)(2.0 / ((double)(w - 1) + 1e-8));
  const float ny = ((ky * sy - 1.0f + 1.0f) / 2.0f) * (float)(h - 1);
  const float nx = ((kx * sx - 1.0f + 1.0f) / 2.0f) * (float)(w - 1);
  cy = (int)nearbyintf(fminf(fmaxf(ny, 0.0f), (float)(h - 1)));
  cx = (int)nearbyintf(fminf(fmaxf(nx, 0.0f), (float)(w - 1)));
}

// atan2 of the Gaussian-weighted first moments of the ps x ps patch centred on (cy, cx), one wave per patch, the lane's
// patch elements lane, lane + 64, ... (zero padding).  PS > 0: compile-time patch size -- the element coordinates are
// constants per lane and every load of the lane is issued before the first multiply (branch-free: clamped address, the
// value dropped afterwards); PS = 0: any size.  Summation order: per lane ascending, then wave_sum -- both instances.
template <int PS>
__device__ __forceinline__ float patch_angle(const float *__restrict__ im, int h, int w, int cy, int cx, int ps_rt,
                                             const float *__restrict__ weights, int lane) {
  const int ps = PS > 0 ? PS : ps_rt;
  const int half = ps / 2, area = ps * ps;
  float m10 = 0.0f, m01 = 0.0f;
  if constexpr (PS > 0) {
    constexpr int NQ = (PS * PS + 63) / 64;
    float v[NQ], wx[NQ], wy[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = lane + 64 * q, ic = min(i, PS * PS - 1);
      const int dy = ic / PS, dx = ic - dy * PS;
      const int gy = cy + dy - PS / 2, gx = cx + dx - PS / 2;
      const bool in = i < PS * PS && gy >= 0 && gy < h && gx >= 0 && gx < w;
      const float raw = im[(size_t)clampi(gy, 0, h - 1) * w + clampi(gx, 0, w - 1)];
      wx[q] = weights[ic];
      wy[q] = weights[PS * PS + ic];
      v[q] = in ? raw : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      m10 += wx[q] * v[q];
      m01 += wy[q] * v[q];
    }
  } else {
    for (int i = lane; i < area; i += 64) {
      const int dy = i / ps, dx = i - dy * ps;
      const int gy = cy + dy - half, gx = cx + dx - half;
      const float v = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? im[(size_t)gy * w + gx] : 0.0f;   // zero padding
      m10 += weights[i] * v;
      m01 += weights[area + i] * v;
    }
  }
  m10 = wave_sum(m10);
  m01 = wave_sum(m01);
  return atan2f(m01, m10);                                                                       // :170
}

template <int PS>
__global__ __launch_bounds__(64) void angle_kp_kernel(MiSets images, int h, int w,
                                                      const float *__restrict__ kpts, int k, int ps,
                                                      const float *__restrict__ weights,
                                                      float *__restrict__ theta) {
  const int lane = threadIdx.x;
  const int flat = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int img = flat / k;
  int cy, cx;
  nearest_pixel(kpts, flat, h, w, cy, cx);
  const float a = patch_angle<PS>(mi_set_item<float>(images, img, (size_t)h * w), h, w, cy, cx, ps, weights, lane);
  if (lane == 0) theta[flat] = a;
}

// AKAZE.forward's orientation (akaze.py:436-451) at keypoints, in one launch: the mean of the orientations of the
// scales that reach the maximum score at the keypoint's pixel -- `attain` (mi_akaze_scale_select) says which -- computed
// only for those scales (typically one of three).  scale_images: num_scales maps (n,h,w), `scale_stride` floats apart.
template <int PS>
__global__ __launch_bounds__(64) void akaze_angle_kp_kernel(const float *__restrict__ scale_images, size_t scale_stride,
                                                            int nscales, const uint8_t *__restrict__ attain, int h,
                                                            int w, const float *__restrict__ kpts, int k, int ps,
                                                            const float *__restrict__ weights,
                                                            float *__restrict__ theta) {
  const int lane = threadIdx.x;
  const int flat = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int img = flat / k;
  int cy, cx;
  nearest_pixel(kpts, flat, h, w, cy, cx);
  const size_t plane = (size_t)h * w;
  unsigned a = attain[(size_t)img * plane + (size_t)cy * w + cx] & ((1u << nscales) - 1u);
  a = __builtin_amdgcn_readfirstlane(a);                       // wave-uniform: the scale loop is scalar control flow
  const float cnt = fmaxf((float)__popc(a), 1.0f);
  float acc = 0.0f;
  for (int s = 0; s < nscales; ++s) {
    if (!((a >> s) & 1u)) continue;                            // theta_s * 0 / cnt adds nothing
    const float t = patch_angle<PS>(scale_images + (size_t)s * scale_stride + (size_t)img * plane, h, w, cy, cx, ps,
                                    weights, lane);
    acc += t * (1.0f / cnt);
  }
  if (lane == 0) theta[flat] = acc;
}

}  // namespace

extern "C" int mi_angle_map(const float *image, int n, int h, int w, int patch_size, const float *moment_kernels,
                            float *angle, mi_stream_t stream) {
  MI_ENTER();
  if (!image || !moment_kernels || !angle) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  if (patch_size <= 0 || (patch_size & 1) == 0 || patch_size > MAX_PS) return MI_E_PARAM;
  const int tiles_x = ceil_div(w, AT_W), tiles_y = ceil_div(h, AT_H);
  const long long blocks = (long long)n * tiles_x * tiles_y;
  if (blocks > 0x7fffffffLL) return MI_E_SHAPE;
  const int half = patch_size / 2;
  const size_t lds = ((size_t)(AT_W + 2 * half) * (AT_H + 2 * half) + 2 * (size_t)patch_size * patch_size) * 4;
  hipLaunchKernelGGL(angle_map_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, image, h, w,
                     patch_size, moment_kernels, angle, tiles_x, tiles_y);
  return mi_launch_status();
}

static int angle_at_keypoints_launch(MiSets images, int n, int h, int w, const float *keypoints, int k, int patch_size,
                                     const float *moment_kernels, float *theta, mi_stream_t stream) {
  if (!images.a || (images.per_set < n && !images.b) || !keypoints || !moment_kernels || !theta) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0 || k <= 0 || (long long)n * k > 0x7fffffffLL) return MI_E_SHAPE;
  if (patch_size <= 0 || (patch_size & 1) == 0 || patch_size > MAX_PS) return MI_E_PARAM;
  if (patch_size == 15)
    hipLaunchKernelGGL(angle_kp_kernel<15>, dim3((unsigned)(n * k)), dim3(64), 0, (hipStream_t)stream, images, h, w,
                       keypoints, k, patch_size, moment_kernels, theta);
  else
    hipLaunchKernelGGL(angle_kp_kernel<0>, dim3((unsigned)(n * k)), dim3(64), 0, (hipStream_t)stream, images, h, w,
                       keypoints, k, patch_size, moment_kernels, theta);
  return mi_launch_status();
}

extern "C" int mi_angle_at_keypoints(const float *image, int n, int h, int w, const float *keypoints, int k,
                                     int patch_size, const float *moment_kernels, float *theta,
                                     mi_stream_t stream) {
  MI_ENTER();
  return angle_at_keypoints_launch(mi_one_set(image, n), n, h, w, keypoints, k, patch_size, moment_kernels, theta, stream);
}

extern "C" int mi_angle_at_keypoints_pair(const float *image_a, const float *image_b, int per_set, int h, int w,
                                          const float *keypoints, int k, int patch_size, const float *moment_kernels,
                                          float *theta, mi_stream_t stream) {
  MI_ENTER();
  if (!image_b || per_set <= 0 || per_set > 0x3fffffff) return image_b ? MI_E_SHAPE : MI_E_NULL;
  return angle_at_keypoints_launch(MiSets{image_a, image_b, per_set}, 2 * per_set, h, w, keypoints, k, patch_size,
                                   moment_kernels, theta, stream);
}

extern "C" int mi_akaze_orientation_select(const float *scale_images, size_t scale_stride, int num_scales,
                                           const uint8_t *attain, int n, int h, int w, const float *keypoints, int k,
                                           int patch_size, const float *moment_kernels, float *theta,
                                           mi_stream_t stream) {
  MI_ENTER();
  if (!scale_images || !attain || !keypoints || !moment_kernels || !theta) return MI_E_NULL;
  if (num_scales <= 0 || num_scales > 8 || n <= 0 || h <= 0 || w <= 0 || k <= 0 || (long long)n * k > 0x7fffffffLL)
    return MI_E_SHAPE;
  if (num_scales > 1 && scale_stride < (size_t)n * h * w) return MI_E_SHAPE;
  if (patch_size <= 0 || (patch_size & 1) == 0 || patch_size > MAX_PS) return MI_E_PARAM;
  if (patch_size == 15)
    hipLaunchKernelGGL(akaze_angle_kp_kernel<15>, dim3((unsigned)(n * k)), dim3(64), 0, (hipStream_t)stream, scale_images,
                       scale_stride, num_scales, attain, h, w, keypoints, k, patch_size, moment_kernels, theta);
  else
    hipLaunchKernelGGL(akaze_angle_kp_kernel<0>, dim3((unsigned)(n * k)), dim3(64), 0, (hipStream_t)stream, scale_images,
                       scale_stride, num_scales, attain, h, w, keypoints, k, patch_size, moment_kernels, theta);
  return mi_launch_status();
}
