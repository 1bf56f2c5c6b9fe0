// K1 corner_response: Shi-Tomasi minimum-eigenvalue score map.
// Semantics: reference pytorch_model/detector/shi_tomasi.py:66-112 (see include/mi355x_match.h).
//
// HBM-bound stencil: 4 B read + 4 B written per pixel.  One 256-thread workgroup owns a
// 128 x (8*R) output tile.  The replicate-clamped image tile (+halo) is staged once in LDS
// with 16-byte coalesced row loads; every thread then slides down R rows for 4 adjacent
// columns keeping the 3-row Sobel window and the block_size rows of horizontally summed
// gradient products in registers, so each image value is read from LDS ~1.5x and each
// product is formed once.  All arithmetic is plain fp32 with contraction off: for
// uint8-valued input every sum is an exact integer < 2^24, hence order-independent, and the
// eigenvalue tail reproduces the reference op for op (IEEE sqrt).
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int TW = 128;  // tile width: 32 threads x 4 pixels
constexpr int LPAD = 4;  // LDS padding each side, one float4 (>= 1 + block/2 for block <= 7)
constexpr int LW4 = (TW + 2 * LPAD) / 4;

__device__ __forceinline__ float lambda_min(float a, float c, float b) {
  // shi_tomasi.py:102-110, one rounding per op
  float half_trace = (a + c) * 0.5f;
  float half_diff = (a - c) * 0.5f;
  float disc = half_diff * half_diff + b * b;
  float root = sqrtf(disc + 1e-10f);
  return fmaxf(half_trace - root, 0.0f);
}

template <int BS, int R>
__global__ __launch_bounds__(256) void corner_tile_kernel(const float *__restrict__ image,
                                                          float *__restrict__ score, int h, int w,
                                                          int tiles_x, int tiles_y) {
  constexpr int HP = BS / 2;       // halo of the product maps
  constexpr int HL = HP + 1;       // halo of the image
  constexpr int TH = 8 * R;        // tile height
  constexpr int LH = TH + 2 * HL;  // staged rows
  constexpr int NG = 4 + 2 * HP;   // gradient columns per thread
  constexpr int NP = R + 2 * HP;   // product rows per thread
  __shared__ float4 tile[LH][LW4];

  const int t = threadIdx.x;
  int bid = blockIdx.x;
  const int tx_tile = bid % tiles_x;
  bid /= tiles_x;
  const int ty_tile = bid % tiles_y;
  const int img = bid / tiles_y;
  const int x0 = tx_tile * TW, y0 = ty_tile * TH;
  const float *im = image + (size_t)img * h * w;

  // ---- stage the clamped tile: every float4 chunk is wholly inside or wholly outside (w % 4 == 0)
  // All of a thread's 16-byte loads are issued before the first LDS store, so ~10 loads per lane
  // are in flight instead of one dependent HBM round trip per chunk.
  {
    constexpr int NCH = (LH * LW4 + 255) / 256;
    float4 v[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int i = t + q * 256;
      const int r = i / LW4, c = i - r * LW4;
      const int gy = clampi(y0 - HL + r, 0, h - 1);
      const int gx = x0 - LPAD + 4 * c;
      const float *row = im + (size_t)gy * w;
      v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < LH * LW4) {
        if (gx < 0) {
          const float e = row[0];
          v[q] = make_float4(e, e, e, e);
        } else if (gx >= w) {
          const float e = row[w - 1];
          v[q] = make_float4(e, e, e, e);
        } else {
          v[q] = *reinterpret_cast<const float4 *>(row + gx);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int i = t + q * 256;
      if (i < LH * LW4) (&tile[0][0])[i] = v[q];
    }
  }
  __syncthreads();

  const int tx = t & 31, ty = t >> 5;
  const int x = x0 + 4 * tx;
  const int ybase = y0 + ty * R;
  if (x >= w || ybase >= h) return;
  const bool left_edge = (x == 0);
  const bool right_edge = (x + 4 >= w);
  const bool top_edge = (ybase == 0);

  float win[3][12];       // rolling image rows, columns x-4 .. x+7
  float hs[NP][3][4];     // horizontally summed products per product row (xx, yy, xy)

#pragma unroll
  for (int ir = 0; ir < R + 2 * HL; ++ir) {
    // image row (ybase - HL + ir) lives in LDS row ty*R + ir
    {
      const float4 *src = &tile[ty * R + ir][tx];
      const float4 a = src[0], b = src[1], c = src[2];
      float *d = win[ir % 3];
      d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
      d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
      d[8] = c.x; d[9] = c.y; d[10] = c.z; d[11] = c.w;
    }
    if (ir < 2) continue;
    const int pr = ir - 2;                 // product row index, global row ybase - HP + pr
    const float *top = win[(ir - 2) % 3], *mid = win[(ir - 1) % 3], *bot = win[ir % 3];
    const int gy = ybase - HP + pr;

    // separable Sobel: vertical 1-2-1 / difference per column, then the horizontal taps
    float sm[NG + 2], df[NG + 2];
#pragma unroll
    for (int k = 0; k < NG + 2; ++k) {
      const int q = 3 - HP + k;            // window column of gradient column k-1
      sm[k] = (top[q] + bot[q]) + 2.0f * mid[q];
      df[k] = bot[q] - top[q];
    }
    float gx_[NG], gy_[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      gx_[j] = sm[j + 2] - sm[j];
      gy_[j] = (df[j] + df[j + 2]) + 2.0f * df[j + 1];
    }
    // replicate padding of the PRODUCT maps == gradients taken at the clamped column
    if (left_edge) {
#pragma unroll
      for (int j = 0; j < HP; ++j) { gx_[j] = gx_[HP]; gy_[j] = gy_[HP]; }
    }
    if (right_edge) {
#pragma unroll
      for (int j = 0; j < HP; ++j) { gx_[4 + HP + j] = gx_[3 + HP]; gy_[4 + HP + j] = gy_[3 + HP]; }
    }
    float pxx[NG], pyy[NG], pxy[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      pxx[j] = gx_[j] * gx_[j];
      pyy[j] = gy_[j] * gy_[j];
      pxy[j] = gx_[j] * gy_[j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = pxx[i], c = pyy[i], b = pxy[i];
#pragma unroll
      for (int dx = 1; dx < BS; ++dx) { a += pxx[i + dx]; c += pyy[i + dx]; b += pxy[i + dx]; }
      hs[pr][0][i] = a; hs[pr][1][i] = c; hs[pr][2][i] = b;
    }
    // rows below the image repeat the last in-image product row
    if (pr > 0 && gy > h - 1) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) hs[pr][q][i] = hs[pr - 1][q][i];
    }
    if (pr < 2 * HP) continue;
    const int orow = pr - 2 * HP;          // output row ybase + orow
    if (ybase + orow >= h) continue;
    float out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float acc[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        float s = 0.0f;
#pragma unroll
        for (int dy = 0; dy < BS; ++dy) {
          const int r = orow + dy;         // product row; rows above the image repeat row `HP`
          const float v = (top_edge && r < HP) ? hs[HP][q][i] : hs[r][q][i];
          s = (dy == 0) ? v : s + v;
        }
        acc[q] = s;
      }
      out[i] = lambda_min(acc[0], acc[1], acc[2]);
    }
    float *dst = score + ((size_t)img * h + (ybase + orow)) * w + x;
    *reinterpret_cast<float4 *>(dst) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

// Generic path: any width, any odd block size.  One thread per pixel, straight from global
// memory (L2 absorbs the re-reads).  Same arithmetic, same clamping rules.
__global__ __launch_bounds__(256) void corner_generic_kernel(const float *__restrict__ image,
                                                             float *__restrict__ score, int n, int h,
                                                             int w, int bs) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)n * h * w;
  if (idx >= total) return;
  const int x = (int)(idx % w);
  const int y = (int)((idx / w) % h);
  const float *im = image + (idx / ((size_t)h * w)) * (size_t)h * w;
  const int hp = bs / 2;
  float a = 0.f, c = 0.f, b = 0.f;
  for (int dy = -hp; dy <= hp; ++dy) {
    const int py = clampi(y + dy, 0, h - 1);
    const float *r0 = im + (size_t)clampi(py - 1, 0, h - 1) * w;
    const float *r1 = im + (size_t)py * w;
    const float *r2 = im + (size_t)clampi(py + 1, 0, h - 1) * w;
    for (int dx = -hp; dx <= hp; ++dx) {
      const int px = clampi(x + dx, 0, w - 1);
      const int xl = clampi(px - 1, 0, w - 1), xr = clampi(px + 1, 0, w - 1);
      const float sl = (r0[xl] + r2[xl]) + 2.0f * r1[xl];
      const float sr = (r0[xr] + r2[xr]) + 2.0f * r1[xr];
      const float dl = r2[xl] - r0[xl], dc = r2[px] - r0[px], dr = r2[xr] - r0[xr];
      const float gx = sr - sl;
      const float gy = (dl + dr) + 2.0f * dc;
      a += gx * gx;
      c += gy * gy;
      b += gx * gy;
    }
  }
  score[idx] = lambda_min(a, c, b);
}

template <int BS, int R>
int launch_tile(const float *image, int n, int h, int w, float *score, hipStream_t s) {
  const int tiles_x = ceil_div(w, TW), tiles_y = ceil_div(h, 8 * R);
  const long long blocks = (long long)n * tiles_x * tiles_y;
  if (blocks > 0x7fffffffLL) return MI_E_SHAPE;
  hipLaunchKernelGGL((corner_tile_kernel<BS, R>), dim3((unsigned)blocks), dim3(256), 0, s, image, score,
                     h, w, tiles_x, tiles_y);
  return mi_launch_status();
}

}  // namespace

extern "C" int mi_corner_response(const float *image, int n, int h, int w, int block_size, float *score,
                                  mi_stream_t stream) {
  if (!image || !score) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  if (block_size <= 0 || (block_size & 1) == 0) return MI_E_PARAM;
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = (w % 4 == 0) && (((uintptr_t)image | (uintptr_t)score) % 16 == 0);
  if (aligned && h >= 4 && w >= 8) {
    if (block_size == 3) return launch_tile<3, 8>(image, n, h, w, score, s);
    if (block_size == 5) return launch_tile<5, 8>(image, n, h, w, score, s);
    if (block_size == 7) return launch_tile<7, 8>(image, n, h, w, score, s);
  }
  const size_t total = (size_t)n * h * w;
  const size_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffULL) return MI_E_SHAPE;
  hipLaunchKernelGGL(corner_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, s, image, score, n, h,
                     w, block_size);
  return mi_launch_status();
}
