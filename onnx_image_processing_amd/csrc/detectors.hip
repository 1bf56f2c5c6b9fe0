// K11: the remaining detectors of pytorch_model/detector -- FAST and Difference-of-Gaussians.
//
// FAST, reference pytorch_model/detector/fast.py:198-239: the 16 pixels of the radius-3 Bresenham
// circle (replicate padding, :72) against the centre: dark bit i = (I_i - I_c >= t), bright bit
// i = (I_i - I_c <= -t) (:128-129); score 1.0 where either 16-bit ring holds 9 consecutive set bits
// (circularly, :139-196), else 0.0.  The reference finds the run with a 24-bit buffer and 16 modulo
// tests; here the same predicate is x & x>>1 & ... on the 24-bit buffer (run length >= 9 <=> a bit
// survives shifts by 1, 2, 4, 1).  Comparisons of fp32 differences exactly as the reference forms
// them, so the output is bit-identical for any input.
//
// DoG, reference pytorch_model/detector/dog.py:100-142: replicate-pad by ks/2, blur with num_scales
// normalised ks x ks Gaussians, difference of consecutive scales -> (N, S-1, H, W).  The normalised
// 2-D Gaussian is exactly the outer product of its own row sums, so the blur runs as two 1-D passes
// (2*ks instead of ks^2 multiply-adds per pixel per scale); fp32, tolerance parity (the reference's
// own conv order is the backend's).
#include "common.h"

#include <math.h>

namespace {

constexpr int FAST_DY[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};     // fast.py:47-52
constexpr int FAST_DX[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};

__device__ __forceinline__ bool run9(uint32_t bits16) {
  uint32_t x = bits16 | ((bits16 & 0xFFu) << 16);      // 24-bit circular buffer (:160-165)
  x &= x >> 1;
  x &= x >> 2;
  x &= x >> 4;
  x &= x >> 1;                                         // bit s set <=> bits s..s+8 were all set
  return (x & 0xFFFFu) != 0u;
}

constexpr int FT_W = 64, FT_H = 16;                    // tile; 3-pixel halo

__global__ __launch_bounds__(256) void fast_kernel(const float *__restrict__ image, int h, int w, float thr,
                                                   float *__restrict__ score, int tiles_x, int tiles_y) {
  __shared__ float tile[FT_H + 6][FT_W + 6 + 2];
  int bid = (int)blockIdx.x;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int img = bid / tiles_y;
  const int x0 = tx * FT_W, y0 = ty * FT_H;
  const float *im = image + (size_t)img * h * w;
  for (int i = threadIdx.x; i < (FT_H + 6) * (FT_W + 6); i += 256) {
    const int r = i / (FT_W + 6), c = i - r * (FT_W + 6);
    tile[r][c] = im[(size_t)clampi(y0 - 3 + r, 0, h - 1) * w + clampi(x0 - 3 + c, 0, w - 1)];
  }
  __syncthreads();
  const int lx = threadIdx.x & 63, ly0 = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < FT_H / 4; ++k) {
    const int ly = ly0 * (FT_H / 4) + k;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) continue;
    const float centre = tile[ly + 3][lx + 3];
    uint32_t dark = 0u, bright = 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float diff = tile[ly + 3 + FAST_DY[i]][lx + 3 + FAST_DX[i]] - centre;
      dark |= (diff >= thr ? 1u : 0u) << i;
      bright |= (diff <= -thr ? 1u : 0u) << i;
    }
    score[((size_t)img * h + gy) * w + gx] = (run9(dark) || run9(bright)) ? 1.0f : 0.0f;
  }
}

// ---- DoG -------------------------------------------------------------------------------------------
constexpr int DG_T = 32;          // output tile edge
constexpr int DG_MAXH = 24;       // half kernel <= 24 (ks <= 49)
constexpr int DG_MAXS = 8;        // scales <= 8

__global__ __launch_bounds__(256) void dog_kernel(const float *__restrict__ image, int h, int w,
                                                  const float *__restrict__ w1d, int num_scales, int ks,
                                                  float *__restrict__ out, float *__restrict__ score,
                                                  int tiles_x, int tiles_y) {
  extern __shared__ float lds[];
  const int half = ks / 2, ext = DG_T + 2 * half;
  float *tile = lds;                       // [ext][ext + 1]   replicate-padded input
  float *hrow = lds + ext * (ext + 1);     // [ext][DG_T + 1]  horizontally blurred rows of one scale
  float *wt = hrow + ext * (DG_T + 1);     // [ks]             this scale's 1-D weights
  int bid = (int)blockIdx.x;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int img = bid / tiles_y;
  const int x0 = tx * DG_T, y0 = ty * DG_T;
  const float *im = image + (size_t)img * h * w;
  for (int i = threadIdx.x; i < ext * ext; i += 256) {
    const int r = i / ext, c = i - r * ext;
    tile[r * (ext + 1) + c] = im[(size_t)clampi(y0 - half + r, 0, h - 1) * w + clampi(x0 - half + c, 0, w - 1)];
  }
  const int lx = threadIdx.x & 31, lyb = threadIdx.x >> 5;      // thread = column lx, rows lyb, lyb+8, lyb+16, lyb+24
  float prev[4] = {0.f, 0.f, 0.f, 0.f};
  float best[4] = {0.f, 0.f, 0.f, 0.f};                          // max over scales of |DoG| (dog.py:197-203)
  for (int s = 0; s < num_scales; ++s) {
    __syncthreads();                                             // tile staged / previous scale's hrow consumed
    for (int i = threadIdx.x; i < ks; i += 256) wt[i] = w1d[s * ks + i];
    __syncthreads();
    for (int i = threadIdx.x; i < ext * DG_T; i += 256) {        // horizontal pass
      const int r = i / DG_T, c = i - r * DG_T;
      const float *src = tile + r * (ext + 1) + c;
      float acc = 0.0f;
      for (int k = 0; k < ks; ++k) acc += wt[k] * src[k];
      hrow[r * (DG_T + 1) + c] = acc;
    }
    __syncthreads();
    float cur[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                // vertical pass
      const int ly = lyb + 8 * q;
      float acc = 0.0f;
      for (int k = 0; k < ks; ++k) acc += wt[k] * hrow[(ly + k) * (DG_T + 1) + lx];
      cur[q] = acc;
    }
    if (s > 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int gx = x0 + lx, gy = y0 + lyb + 8 * q;
        const float d = cur[q] - prev[q];                                                                               // dog.py:140
        best[q] = fmaxf(best[q], fabsf(d));
        if (out && gx < w && gy < h) out[(((size_t)img * (num_scales - 1) + (s - 1)) * h + gy) * w + gx] = d;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) prev[q] = cur[q];
  }
  if (score) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int gx = x0 + lx, gy = y0 + lyb + 8 * q;
      if (gx < w && gy < h) score[((size_t)img * h + gy) * w + gx] = best[q];
    }
  }
}

// The same blurs for one compile-time kernel size (KS = 39: the constructor default, sigmas 1.6 ... 6.4), register-blocked
// (round 4): the kernel above reads LDS twice per multiply-add (weight and sample) -- 6.6 ms per 256 images of 640x480,
// 0.1 TB/s.  Here a 64 x 32 output tile per workgroup; in the horizontal pass a thread forms 8 consecutive outputs of a row
// from the 8 + KS - 1 samples it loads ONCE into registers, in the vertical pass 8 consecutive rows of a column likewise;
// the weight of a tap is one broadcast LDS read per 8 multiply-adds, which are fused (the results stay within the
// tolerance the reference's own 2-D convolution is held to; both outputs come from the same sums).
template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void dog_blocked_kernel(const float *__restrict__ image, int h, int w,
                                                          const float *__restrict__ w1d, int num_scales,
                                                          float *__restrict__ out, float *__restrict__ score,
                                                          int tiles_x, int tiles_y) {
  constexpr int HALF = KS / 2, TW = 64, TH = 32, EX = TW + 2 * HALF, EY = TH + 2 * HALF;
  constexpr int TP = EX + 1, HP = TW + 1, NIN = 8 + KS - 1;
  __shared__ float tile[EY * TP];          // replicate-padded input
  __shared__ float hrow[EY * HP];          // horizontally blurred rows of one scale
  __shared__ float wt[KS];
  int bid = (int)blockIdx.x;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int img = bid / tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const float *im = image + (size_t)img * h * w;
  for (int i = threadIdx.x; i < EY * EX; i += 256) {
    const int r = i / EX, c = i - r * EX;
    tile[r * TP + c] = im[(size_t)clampi(y0 - HALF + r, 0, h - 1) * w + clampi(x0 - HALF + c, 0, w - 1)];
  }
  const int vc = threadIdx.x & 63, vr = (threadIdx.x >> 6) * 8;   // vertical pass: column vc, rows vr .. vr + 7
  float prev[8], best[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { prev[q] = 0.0f; best[q] = 0.0f; }
  for (int s = 0; s < num_scales; ++s) {
    __syncthreads();                                             // tile staged / previous scale's hrow consumed
    if (threadIdx.x < KS) wt[threadIdx.x] = w1d[s * KS + threadIdx.x];
    __syncthreads();
    for (int item = threadIdx.x; item < EY * (TW / 8); item += 256) {   // horizontal pass: 8 outputs of row r
      const int r = item >> 3, c0 = (item & 7) * 8;
      const float *src = tile + r * TP + c0;
      float x[NIN], acc[8];
#pragma unroll
      for (int t = 0; t < NIN; ++t) x[t] = src[t];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const float wk = wt[k];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(wk, x[k + j], acc[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) hrow[r * HP + c0 + j] = acc[j];
    }
    __syncthreads();
    float cur[8];
    {
      const float *src = hrow + vr * HP + vc;                    // vertical pass
      float x[NIN];
#pragma unroll
      for (int t = 0; t < NIN; ++t) x[t] = src[t * HP];
#pragma unroll
      for (int j = 0; j < 8; ++j) cur[j] = 0.0f;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const float wk = wt[k];
#pragma unroll
        for (int j = 0; j < 8; ++j) cur[j] = __builtin_fmaf(wk, x[k + j], cur[j]);
      }
    }
    if (s > 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int gx = x0 + vc, gy = y0 + vr + q;
        const float d = cur[q] - prev[q];                                                                               // dog.py:140
        best[q] = fmaxf(best[q], fabsf(d));
        if (out && gx < w && gy < h) out[(((size_t)img * (num_scales - 1) + (s - 1)) * h + gy) * w + gx] = d;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) prev[q] = cur[q];
  }
  if (score) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int gx = x0 + vc, gy = y0 + vr + q;
      if (gx < w && gy < h) score[((size_t)img * h + gy) * w + gx] = best[q];                                         // dog.py:197-203
    }
  }
}

}  // namespace

extern "C" int mi_fast_score(const float *image, int n, int h, int w, float threshold, float *score,
                             mi_stream_t stream) {
  MI_ENTER();
  if (!image || !score) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  const int tiles_x = ceil_div(w, FT_W), tiles_y = ceil_div(h, FT_H);
  const long long blocks = (long long)n * tiles_x * tiles_y;
  if (blocks > 0x7fffffffLL) return MI_E_SHAPE;
  hipLaunchKernelGGL(fast_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, image, h, w, threshold,
                     score, tiles_x, tiles_y);
  return mi_launch_status();
}

extern "C" int mi_dog_responses(const float *image, int n, int h, int w, const float *weights_1d, int num_scales,
                                int kernel_size, float *out, float *score, mi_stream_t stream) {
  MI_ENTER();
  if (!image || !weights_1d || (!out && !score)) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  if (num_scales < 2 || num_scales > DG_MAXS || kernel_size <= 0 || (kernel_size & 1) == 0 ||
      kernel_size / 2 > DG_MAXH)
    return MI_E_PARAM;
  if (kernel_size == 39) {                         // the constructor default: register-blocked kernel
    const int bx = ceil_div(w, 64), by = ceil_div(h, 32);
    const long long nblocks = (long long)n * bx * by;
    if (nblocks > 0x7fffffffLL) return MI_E_SHAPE;
    hipLaunchKernelGGL(dog_blocked_kernel<39>, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, image, h, w,
                       weights_1d, num_scales, out, score, bx, by);
    return mi_launch_status();
  }
  const int tiles_x = ceil_div(w, DG_T), tiles_y = ceil_div(h, DG_T);
  const long long blocks = (long long)n * tiles_x * tiles_y;
  if (blocks > 0x7fffffffLL) return MI_E_SHAPE;
  const int ext = DG_T + 2 * (kernel_size / 2);
  const size_t lds = ((size_t)ext * (ext + 1) + (size_t)ext * (DG_T + 1) + (size_t)kernel_size) * sizeof(float);
  hipLaunchKernelGGL(dog_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, image, h, w, weights_1d,
                     num_scales, kernel_size, out, score, tiles_x, tiles_y);
  return mi_launch_status();
}

// ---- u8 ingest for the entry points without a uint8 form: uint8 -> float32, the conversion the reference's
// hosts do on the CPU before calling the model (sample/visual_odometry.py:65-92, sample/image_matching.py:42-46)
namespace {
__global__ __launch_bounds__(256) void convert_u8_f32_kernel(const uint8_t *__restrict__ src, float *__restrict__ dst,
                                                             size_t count) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 4 <= count && ((uintptr_t)src % 4) == 0) {
    const uint32_t v = *reinterpret_cast<const uint32_t *>(src + i);
    dst[i] = (float)(v & 0xFFu); dst[i + 1] = (float)((v >> 8) & 0xFFu);
    dst[i + 2] = (float)((v >> 16) & 0xFFu); dst[i + 3] = (float)(v >> 24);
  } else {
    for (size_t j = i; j < count && j < i + 4; ++j) dst[j] = (float)src[j];
  }
}
}  // namespace

extern "C" int mi_convert_u8_f32(const uint8_t *src, long long count, float *dst, mi_stream_t stream) {
  MI_ENTER();
  if (!src || !dst) return MI_E_NULL;
  if (count <= 0) return MI_E_SHAPE;
  const size_t blocks = ((size_t)count + 1023) / 1024;
  if (blocks > 0x7fffffffULL) return MI_E_SHAPE;
  hipLaunchKernelGGL(convert_u8_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst,
                     (size_t)count);
  return mi_launch_status();
}
