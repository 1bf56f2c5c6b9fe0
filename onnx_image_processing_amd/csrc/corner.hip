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

#include <type_traits>

#include "hooks.h"

namespace {

constexpr int TW = 128;  // tile width: 32 threads x 4 pixels
#ifndef MI_K1_POOLS
#define MI_K1_POOLS 64
#endif
constexpr int K1_POOLS = MI_K1_POOLS;          // ticket counters of the dynamic tile schedule (corner_stream_kernel)
constexpr int K1_POOL_STRIDE = 64;             // words between them: 256 B, one memory channel each
static_assert((K1_POOLS * K1_POOL_STRIDE + 1) * 4 <= MI_TILE_COUNTER_BYTES, "include/mi355x_match.h: MI_TILE_COUNTER_BYTES");
constexpr int LPAD = 4;  // LDS padding each side, one float4 (>= 1 + block/2 for block <= 7)
constexpr int LW4 = (TW + 2 * LPAD) / 4;

// Correctly rounded fp32 sqrt for normal, finite x (here x >= 1e-10) on the fp32 pipe only: y = v_rsq_f32(x), g = x y,
// h = y / 2, one correction g + (x - g g) h with the residual exact in one fma.  Compared with sqrtf() over EVERY float
// in [1e-10, 2^126] by the exhaustive GPU test (tests/test_gpu_parity.py::test_akaze_fast_division_is_exact; the helper
// is csrc/akaze_math.h's ak_sqrt_fp<1>).  The earlier form -- v_sqrt_f32, then {s - 1ulp, s, s + 1ulp} picked by two
// exact-sign fma residuals: the compiler's IEEE expansion without its denormal scaling -- spends two compares and two
// selects on the pick, which issue at half the rate of fp32 multiplies and adds on gfx950 (tools/micro/valu_classes.hip).
__device__ __forceinline__ float sqrt_rn(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float g = x * y;
  const float h = 0.5f * y;
  return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}

__device__ __forceinline__ float lambda_min(float a, float c, float b) {
  // shi_tomasi.py:102-110, one rounding per op
  float half_trace = (a + c) * 0.5f;
  float half_diff = (a - c) * 0.5f;
  float disc = half_diff * half_diff + b * b;
  float root = sqrt_rn(disc + 1e-10f);
  return fmaxf(half_trace - root, 0.0f);
}

// uint8 tiles (the u8 ingest path, mi_corner_response_u8): one dword = 4 pixels, LWD dwords per staged row
constexpr int LWD = (TW + 2 * LPAD) / 4;

// The 12 window columns x-4 .. x+7 of staged row `row` for thread column tx.  fp32 tile: three float4.  uint8
// tile: three dwords, the eight columns the stencil uses (window indices 2..9) converted with
// v_cvt_f32_ubyteN (exact; everything downstream is the fp32 arithmetic of the fp32 path, so both produce the
// same bits); indices 0, 1, 10, 11 only exist so that the edge splats below stay uniform.
__device__ __forceinline__ void load_window(const float4 *__restrict__ tile, int row, int tx, float *d) {
  const float4 *src = &tile[row * LW4 + tx];
  const float4 a = src[0], b = src[1], c = src[2];
  d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
  d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
  d[8] = c.x; d[9] = c.y; d[10] = c.z; d[11] = c.w;
}
__device__ __forceinline__ void load_window(const uint32_t *__restrict__ tile, int row, int tx, float *d) {
  const uint32_t *src = &tile[row * LWD + tx];
  const uint32_t a = src[0], b = src[1], c = src[2];
  d[0] = 0.0f; d[1] = 0.0f;
  d[2] = (float)((a >> 16) & 0xFFu); d[3] = (float)(a >> 24);
  d[4] = (float)(b & 0xFFu); d[5] = (float)((b >> 8) & 0xFFu); d[6] = (float)((b >> 16) & 0xFFu); d[7] = (float)(b >> 24);
  d[8] = (float)(c & 0xFFu); d[9] = (float)((c >> 8) & 0xFFu);
  d[10] = 0.0f; d[11] = 0.0f;
}

// Per-thread stencil on a staged tile (shared by the tile and the streaming kernels).
// `tile` is the row-major [LH][LW4] float4 (or [LH][LWD] dword, 4 uint8 pixels each) image of the clamped image
// region whose first row is y0 - HL and first column x0 - LPAD.
template <int BS, int R, typename TILE>
__device__ __forceinline__ void corner_compute(const TILE *__restrict__ tile, float *__restrict__ score,
                                               int img, int h, int w, int x0, int y0, int t) {
  constexpr int HP = BS / 2;       // halo of the product maps
  constexpr int HL = HP + 1;       // halo of the image
  constexpr int TH = 8 * R;        // tile height
  constexpr int NG = 4 + 2 * HP;   // gradient columns per thread
  constexpr int NP = R + 2 * HP;   // product rows per thread
  const int tx = t & 31, ty = t >> 5;
  const int x = x0 + 4 * tx;
  const int ybase = y0 + ty * R;
  if (x >= w || ybase >= h) return;
  // image-border handling only exists on border tiles (workgroup-uniform branches; the empty asm keeps the
  // compiler from flattening them into per-row selects that every tile would execute)
#define MI_KEEP_BRANCH() asm volatile("")
  const bool tile_left = (x0 == 0), tile_right = (x0 + TW >= w);
  const bool tile_top = (y0 == 0), tile_bottom = (y0 + TH + HP > h);
  const bool left_edge = (x == 0);
  const bool right_edge = (x + 4 >= w);
  const bool top_edge = (ybase == 0);

  constexpr int NC = NG + 2;   // image columns feeding the NG gradient columns
  constexpr int C0 = 3 - HP;   // window index of the first of them
  float win[3][12];            // rolling image rows, columns x-4 .. x+7
  float cprev[NC];             // row(ir-2) + row(ir-1): half of the vertical 1-2-1
  float hs[NP][3][4];          // horizontally summed products per product row (xx, yy, xy)
  float vq[3][4];              // hs[r+1] + hs[r+2], shared by output rows r and r+1 (BS == 3)

#pragma unroll
  for (int ir = 0; ir < R + 2 * HL; ++ir) {
    // image row (ybase - HL + ir) lives in LDS row ty*R + ir
    {
      float *d = win[ir % 3];
      load_window(tile, ty * R + ir, tx, d);
      // replicate padding of the IMAGE in x: chunks left/right of the image take the edge pixel
      // (the register-staged kernel already stored them that way; LDS-DMA cannot splat)
      if (tile_left) {
        MI_KEEP_BRANCH();
        if (left_edge) { d[0] = d[4]; d[1] = d[4]; d[2] = d[4]; d[3] = d[4]; }
      }
      if (tile_right) {
        MI_KEEP_BRANCH();
        if (right_edge) { d[8] = d[7]; d[9] = d[7]; d[10] = d[7]; d[11] = d[7]; }
      }
    }
    if (ir < 1) continue;
    const float *mid = win[(ir - 1) % 3], *bot = win[ir % 3];
    // vertical 1-2-1 as (top+mid) + (mid+bot): each pair sum is formed once and used twice
    float cnew[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) cnew[k] = mid[C0 + k] + bot[C0 + k];
    if (ir < 2) {
#pragma unroll
      for (int k = 0; k < NC; ++k) cprev[k] = cnew[k];
      continue;
    }
    const int pr = ir - 2;                 // product row index, global row ybase - HP + pr
    const float *top = win[(ir - 2) % 3];
    const int gy = ybase - HP + pr;

    float sm[NC], df[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      sm[k] = cprev[k] + cnew[k];
      df[k] = bot[C0 + k] - top[C0 + k];
      cprev[k] = cnew[k];
    }
    // horizontal taps: gx = sm[j+2]-sm[j];  gy = df[j] + 2 df[j+1] + df[j+2] = e[j] + e[j+1]
    float e[NC - 1];
#pragma unroll
    for (int k = 0; k < NC - 1; ++k) e[k] = df[k] + df[k + 1];
    float gx_[NG], gy_[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      gx_[j] = sm[j + 2] - sm[j];
      gy_[j] = e[j] + e[j + 1];
    }
    // replicate padding of the PRODUCT maps == gradients taken at the clamped column
    if (tile_left) {
      MI_KEEP_BRANCH();
      if (left_edge) {
#pragma unroll
        for (int j = 0; j < HP; ++j) { gx_[j] = gx_[HP]; gy_[j] = gy_[HP]; }
      }
    }
    if (tile_right) {
      MI_KEEP_BRANCH();
      if (right_edge) {
#pragma unroll
        for (int j = 0; j < HP; ++j) { gx_[4 + HP + j] = gx_[3 + HP]; gy_[4 + HP + j] = gy_[3 + HP]; }
      }
    }
    float pxx[NG], pyy[NG], pxy[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      pxx[j] = gx_[j] * gx_[j];
      pyy[j] = gy_[j] * gy_[j];
      pxy[j] = gx_[j] * gy_[j];
    }
    if constexpr (BS == 3) {
      // 4 sliding 3-sums out of 6 values with 7 adds: pair sums s01 s23 s45
      const float *pp[3] = {pxx, pyy, pxy};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float *p = pp[q];
        const float s01 = p[0] + p[1], s23 = p[2] + p[3], s45 = p[4] + p[5];
        hs[pr][q][0] = s01 + p[2];
        hs[pr][q][1] = p[1] + s23;
        hs[pr][q][2] = s23 + p[4];
        hs[pr][q][3] = p[3] + s45;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a = pxx[i], c = pyy[i], b = pxy[i];
#pragma unroll
        for (int dx = 1; dx < BS; ++dx) { a += pxx[i + dx]; c += pyy[i + dx]; b += pxy[i + dx]; }
        hs[pr][0][i] = a; hs[pr][1][i] = c; hs[pr][2][i] = b;
      }
    }
    // rows below the image repeat the last in-image product row
    if (tile_bottom) {
      if (pr > 0 && gy > h - 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) hs[pr][q][i] = hs[pr - 1][q][i];
      }
    }
    // rows above the image repeat product row 0 of the image (= local row HP)
    if (pr == HP) {
      if (tile_top) {
        if (top_edge) {
#pragma unroll
          for (int r = 0; r < HP; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
              for (int i = 0; i < 4; ++i) hs[r][q][i] = hs[HP][q][i];
        }
      }
    }
    if (pr < 2 * HP) continue;
    const int orow = pr - 2 * HP;          // output row ybase + orow
    float acc[3][4];
    if constexpr (BS == 3) {
      // out(r) = hs[r] + (hs[r+1] + hs[r+2]);  out(r+1) = (hs[r+1] + hs[r+2]) + hs[r+3]
      if ((orow & 1) == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            vq[q][i] = hs[orow + 1][q][i] + hs[orow + 2][q][i];
            acc[q][i] = hs[orow][q][i] + vq[q][i];
          }
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[q][i] = vq[q][i] + hs[orow + 2][q][i];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float sacc = hs[orow][q][i];
#pragma unroll
          for (int dy = 1; dy < BS; ++dy) sacc += hs[orow + dy][q][i];
          acc[q][i] = sacc;
        }
    }
    if (ybase + orow >= h) continue;
    float out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = lambda_min(acc[0][i], acc[1][i], acc[2][i]);
    // scalar base (the image plane) + 32-bit byte offset: no 64-bit address arithmetic per row
    char *plane = reinterpret_cast<char *>(score + (size_t)img * h * w);
    *reinterpret_cast<float4 *>(plane + (uint32_t)((ybase + orow) * w + x) * 4u) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

// ---- u8 ingest, block 3: the stencil in packed integer arithmetic -------------------------------------------
// For uint8 pixels every intermediate of the score map up to the box sums is an integer: Sobel sums |.| <= 1020 fit
// int16 (two per register: v_pk_add/sub_i16, v_pk_mad_i16, neighbours by v_alignbit), products and their 3x3 box sums
// (<= 9.4e6 < 2^24) fit int32 (v_dot2_i32_i16 forms a product pair's sum in one instruction) and convert to fp32
// exactly -- so the eigenvalue tail sees the very values the fp32 path computes and the scores are bit-identical,
// at roughly half the vector instructions in front of the tail (the fp32 kernel is bound by VALU issue, not by HBM,
// once its input shrinks to one byte per pixel).
typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s as_v2s(uint32_t x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ uint32_t as_u32(v2s x) { return __builtin_bit_cast(uint32_t, x); }

// a.lo * b.lo and a.hi * b.hi of two int16 pairs as int32 (v_mad_i32_i16 with op_sel; the compiler's own lowering of
// the C expression sign-extends both halves first)
__device__ __forceinline__ int mul_lo16(v2s a, v2s b) {
  int d;
  asm("v_mad_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ int mul_hi16(v2s a, v2s b) {
  int d;
  asm("v_mad_i32_i16 %0, %1, %2, 0 op_sel:[1,1,0,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

template <int R>
__device__ __forceinline__ void corner_compute_u8(const uint32_t *__restrict__ tile, float *__restrict__ score, int img,
                                                  int h, int w, int x0, int y0, int t) {
  constexpr int TH = 8 * R, NP = R + 2;
  const int tx = t & 31, ty = t >> 5;
  const int x = x0 + 4 * tx;
  const int ybase = y0 + ty * R;
  if (x >= w || ybase >= h) return;
  const bool tile_left = (x0 == 0), tile_right = (x0 + TW >= w);
  const bool tile_top = (y0 == 0), tile_bottom = (y0 + TH + 1 > h);
  const bool left_edge = (x == 0), right_edge = (x + 4 >= w), top_edge = (ybase == 0);

  v2s win[3][4];           // rolling rows: image columns x-2 .. x+5 as four (lo, hi) pairs of int16
  v2s cprev[4];            // row(ir-2) + row(ir-1)
  int hs[NP][3][4];        // horizontally summed products per product row (xx, yy, xy)
  int vq[3][4];

#pragma unroll
  for (int ir = 0; ir < R + 4; ++ir) {
    {
      const uint32_t *src = &tile[(ty * R + ir) * LWD + tx];
      const uint32_t a = src[0], b = src[1], c = src[2];
      v2s *d = win[ir % 3];
      d[0] = as_v2s(__builtin_amdgcn_perm(0u, a, 0x0c030c02u));        // (x-2, x-1)
      d[1] = as_v2s(__builtin_amdgcn_perm(0u, b, 0x0c010c00u));        // (x,   x+1)
      d[2] = as_v2s(__builtin_amdgcn_perm(0u, b, 0x0c030c02u));        // (x+2, x+3)
      d[3] = as_v2s(__builtin_amdgcn_perm(0u, c, 0x0c010c00u));        // (x+4, x+5)
      // replicate padding of the IMAGE in x (the chunks outside the image were fetched from a clamped address)
      if (tile_left) {
        MI_KEEP_BRANCH();
        if (left_edge) d[0] = as_v2s(__builtin_amdgcn_perm(0u, b, 0x0c000c00u));
      }
      if (tile_right) {
        MI_KEEP_BRANCH();
        if (right_edge) d[3] = as_v2s(__builtin_amdgcn_perm(0u, b, 0x0c030c03u));
      }
    }
    if (ir < 1) continue;
    const v2s *mid = win[(ir - 1) % 3], *bot = win[ir % 3];
    v2s cnew[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cnew[k] = mid[k] + bot[k];
    if (ir < 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) cprev[k] = cnew[k];
      continue;
    }
    const int pr = ir - 2;                  // product row index, global row ybase - 1 + pr
    const v2s *top = win[(ir - 2) % 3];
    const int gy = ybase - 1 + pr;
    v2s sm[4], df[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sm[k] = cprev[k] + cnew[k];           // top + 2 mid + bot
      df[k] = bot[k] - top[k];
      cprev[k] = cnew[k];
    }
    // gradient columns j = 0..5 (image columns x-1 .. x+4) as pairs (0,1) (2,3) (4,5)
    v2s gx[3], gyv[3];
    const v2s two = {2, 2};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      gx[j] = sm[j + 1] - sm[j];                                                   // sm[c+2] - sm[c]
      const v2s sh = as_v2s(__builtin_amdgcn_alignbit(as_u32(df[j + 1]), as_u32(df[j]), 16));   // (df[2j+1], df[2j+2])
      gyv[j] = (sh * two + df[j]) + df[j + 1];                                      // df[c] + 2 df[c+1] + df[c+2]
    }
    // replicate padding of the PRODUCT maps == gradients taken at the clamped column
    if (tile_left) {
      MI_KEEP_BRANCH();
      if (left_edge) {
        gx[0] = as_v2s(__builtin_amdgcn_perm(0u, as_u32(gx[0]), 0x03020302u));
        gyv[0] = as_v2s(__builtin_amdgcn_perm(0u, as_u32(gyv[0]), 0x03020302u));
      }
    }
    if (tile_right) {
      MI_KEEP_BRANCH();
      if (right_edge) {
        gx[2] = as_v2s(__builtin_amdgcn_perm(0u, as_u32(gx[2]), 0x01000100u));
        gyv[2] = as_v2s(__builtin_amdgcn_perm(0u, as_u32(gyv[2]), 0x01000100u));
      }
    }
    // products and their sliding 3-sums p0+p1+p2, p1+p2+p3, p2+p3+p4, p3+p4+p5: the odd product out is one
    // v_mad_i32_i16 (op_sel picks the half, no sign extension needed), the pair's two products join it through one
    // accumulating v_dot2c_i32_i16 -- two instructions per 3-sum
    {
      const v2s *aa[3] = {gx, gyv, gx};
      const v2s *bb[3] = {gx, gyv, gyv};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const v2s *A = aa[q], *B = bb[q];
        hs[pr][q][0] = __builtin_amdgcn_sdot2(A[0], B[0], mul_lo16(A[1], B[1]), false);
        hs[pr][q][1] = __builtin_amdgcn_sdot2(A[1], B[1], mul_hi16(A[0], B[0]), false);
        hs[pr][q][2] = __builtin_amdgcn_sdot2(A[1], B[1], mul_lo16(A[2], B[2]), false);
        hs[pr][q][3] = __builtin_amdgcn_sdot2(A[2], B[2], mul_hi16(A[1], B[1]), false);
      }
    }
    // rows below the image repeat the last in-image product row
    if (tile_bottom) {
      if (pr > 0 && gy > h - 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) hs[pr][q][i] = hs[pr - 1][q][i];
      }
    }
    // rows above the image repeat product row 0 of the image (= local row 1)
    if (pr == 1) {
      if (tile_top) {
        if (top_edge) {
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) hs[0][q][i] = hs[1][q][i];
        }
      }
    }
    if (pr < 2) continue;
    const int orow = pr - 2;               // output row ybase + orow
    int acc[3][4];
    if ((orow & 1) == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          vq[q][i] = hs[orow + 1][q][i] + hs[orow + 2][q][i];
          acc[q][i] = hs[orow][q][i] + vq[q][i];
        }
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[q][i] = vq[q][i] + hs[orow + 2][q][i];
    }
    if (ybase + orow >= h) continue;
    float out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = lambda_min((float)acc[0][i], (float)acc[1][i], (float)acc[2][i]);
    char *plane = reinterpret_cast<char *>(score + (size_t)img * h * w);
    *reinterpret_cast<float4 *>(plane + (uint32_t)((ybase + orow) * w + x) * 4u) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

template <int BS, int R>
__global__ __launch_bounds__(256) void corner_tile_kernel(MiSets images,
                                                          float *__restrict__ score, int h, int w,
                                                          int tiles_x, int tiles_y) {
  constexpr int HP = BS / 2;       // halo of the product maps
  constexpr int HL = HP + 1;       // halo of the image
  constexpr int TH = 8 * R;        // tile height
  constexpr int LH = TH + 2 * HL;  // staged rows
  __shared__ float4 tile[LH][LW4];

  const int t = threadIdx.x;
  int bid = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);   // neighbouring tiles share an XCD's L2
  const int tx_tile = bid % tiles_x;
  bid /= tiles_x;
  const int ty_tile = bid % tiles_y;
  const int img = bid / tiles_y;
  const int x0 = tx_tile * TW, y0 = ty_tile * TH;
  const float *im = mi_set_item<float>(images, img, (size_t)h * w);

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
  corner_compute<BS, R>(&tile[0][0], score, img, h, w, x0, y0, t);
}

// Streaming form: persistent workgroups, two LDS buffers, the next tile arrives by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no wave stalled on the load) while the current
// tile is being computed.  Each wave issues NCH 1-KiB DMA pieces per tile; the LDS image is the
// same linear [LH][LW4] float4 array (chunk i of the tile at byte 16*i), so lane l of piece q
// lands exactly where the register-staged kernel would have stored chunk q*256 + t.
//   loop:  s_waitcnt vmcnt(R): THIS tile's DMA has landed (the previous tile's R stores, issued
//          later, may still be in flight); barrier; issue DMA(next tile -> other buffer); compute
// Row clamping is done by the DMA source address itself; chunks left/right of the image are
// fetched from a clamped (valid) address and replaced by the edge pixel in registers
// (corner_compute), so no LDS access ever has to wait for an in-flight DMA.
// PIX = float: 16-byte pieces (global_load_lds_dwordx4), 8 B/px of HBM traffic per pixel (4 read + 4 written).
// PIX = uint8_t (the u8 ingest path): the same chunk grid with 4-byte pieces (global_load_lds_dword, 4 pixels
// each), 5 B/px; the tile is a quarter of the LDS, the stencil converts on the LDS read (load_window).
// Tile schedule.  Static (tile_ctr == NULL): workgroup b takes tiles b, b + G, b + 2G, ...  Measured on 448 images
// (mi_debug_clock_probe): every workgroup starts within 0.7 us, yet they finish between 132 and 226 us (uint8: 63 ...
// 232 us) although each owns the same number of tiles, and the lifetime grows with blockIdx -- the SIMDs issue
// oldest-first, so of the 4-6 workgroups sharing a CU the first dispatched runs at nearly full speed and the last gets
// what is left; the kernel lasts as long as the youngest workgroup, the CUs thinning out on the way.
// Dynamic (tile_ctr != NULL): the first two tiles are static, every further one comes from a ticket counter, so all
// workgroups finish together.  K1_POOLS counters, 256 B apart: workgroup b draws from pool b % K1_POOLS (its members sit
// on one XCD and come from every dispatch-age class), ticket k of pool p is tile 2G + k * K1_POOLS + p.  ONE counter
// does not do: same-address atomics retire at about one per 12 ns, 32,000 tickets then take 390 us (measured) where
// the whole kernel should take 200.  Lane 0 draws the ticket of tile i + 2 at the start of tile i, BEFORE the DMA
// pieces of tile i + 1, as inline asm returning into an AGPR: a compiler-visible returning atomic makes hipcc wait
// vmcnt(0) where its value joins the control flow (which drains the DMA just issued), whereas here the value is waited
// for with a COUNTED wait at the end of the tile (vmcnt(NCH + R): in-order retirement, the DMA pieces and the tile's
// stores were issued after it) and the AGPR keeps the register allocator from copying it while it is in flight.
template <int BS, int R, bool U8>
__global__ __launch_bounds__(256) void corner_stream_kernel(MiSets images,
                                                            float *__restrict__ score, int h, int w,
                                                            int tiles_x, int tiles_y, int total_tiles,
                                                            unsigned *tile_ctr, unsigned long long *clk) {
  // development aid (mi_debug_clock_probe): every workgroup records the shader clock (s_memtime) and the constant
  // 100 MHz clock at entry and exit -> the SCLK the kernel ran at and the workgroups' lifetimes
  if (clk && threadIdx.x == 0) { clk[4 * blockIdx.x + 0] = clock64(); clk[4 * blockIdx.x + 1] = wall_clock64(); }
  using CH = typename std::conditional<U8, uint32_t, float4>::type;   // 4 pixels
  constexpr int HP = BS / 2, HL = HP + 1, TH = 8 * R, LH = TH + 2 * HL;
  constexpr int NCH = (LH * LW4 + 255) / 256;   // DMA pieces per wave per tile
  constexpr int BUF = NCH * 256;                // chunk slots per buffer (tail slots are scratch)
  // ONE LDS object on purpose, and not a byte more than the two tile buffers (fp32: 2 x 20 KiB, exactly four workgroups
  // per CU): the ticket travels through the last scratch slot of the buffer that has just been computed -- no DMA is in
  // flight for it, the next one is issued only after the slot has been read.  With a second __shared__ variable the LDS
  // lowering attaches alias scopes, the waitcnt pass then knows that the tile reads may
  // alias the LDS-DMA writes and puts `s_waitcnt vmcnt(0)` in front of the first tile read of every iteration -- which
  // drains the DMA just issued for the NEXT tile (checked in the .s: the only vmcnt waits are the two hand-placed ones)
  __shared__ CH lds[2 * BUF];
  static_assert(LH * LW4 < BUF, "the ticket slot needs one scratch chunk slot");

  const int t = threadIdx.x;
  const int wave_base = t & ~63;

  // tile id -> (image, x0, y0).  Blockidx-derived, hence scalar (SGPR) arithmetic.
  auto decode = [&](int v, int &img, int &x0, int &y0) {
    int id = (int)xcd_contiguous_id((unsigned)v, (unsigned)total_tiles);
    const int tx_tile = id % tiles_x;
    id /= tiles_x;
    const int ty_tile = id % tiles_y;
    img = id / tiles_y;
    x0 = tx_tile * TW;
    y0 = ty_tile * TH;
  };
  // per-thread chunk coordinates of its NCH pieces never change: (row, 4*col) inside the tile
  int prow[NCH], pcol[NCH];
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    int i = t + q * 256;
    if (i >= LH * LW4) i = 0;                                     // scratch slots: any valid source
    prow[q] = i / LW4 - HL;
    pcol[q] = 4 * (i % LW4) - LPAD;
  }
  auto issue = [&](int img, int x0, int y0, int buf) {
    // The source pointers have concrete types on purpose: with a template-dependent pointer type hipcc 7.2 checks
    // the builtin's size immediate at instantiation time in the HOST pass, rejects 16 there without a diagnostic
    // and drops the instance's host stub (undefined symbol at load time).
    const float *imf = mi_set_item<float>(images, img, (size_t)h * w);
    const uint8_t *imb = mi_set_item<uint8_t>(images, img, (size_t)h * w);
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int gy = clampi(y0 + prow[q], 0, h - 1);
      const int gx = clampi(x0 + pcol[q], 0, w - 4);
      auto *dst = (__attribute__((address_space(3))) void *)&lds[buf * BUF + q * 256 + wave_base];
      // the last scratch slot of a buffer carries the ticket (below): its lane sits the last piece out, or a wave that
      // has read the ticket and issued this DMA could overwrite it before a slower wave has read it
      if (q == NCH - 1 && t == 255) continue;
      if constexpr (U8) __builtin_amdgcn_global_load_lds(imb + (gy * w + gx), dst, 4, 0, 0);
      else __builtin_amdgcn_global_load_lds(imf + (gy * w + gx), dst, 16, 0, 0);
    }
  };

  const int G = (int)gridDim.x;
  unsigned *my_ctr = tile_ctr + (blockIdx.x % K1_POOLS) * K1_POOL_STRIDE;
  const int pool_base = 2 * G + (int)(blockIdx.x % K1_POOLS);
  int v = blockIdx.x;
  if (v >= total_tiles) return;
  int img, x0, y0;
  decode(v, img, x0, y0);
  issue(img, x0, y0, 0);
  bool prev_full = false;                         // previous tile issued exactly R stores per lane
  for (int it = 0; (unsigned)v < (unsigned)total_tiles; ++it) {
    const int cur = it & 1;
    // This tile's DMA pieces were issued BEFORE the previous tile's R stores, and vmcnt retires
    // in issue order: leaving R operations outstanding waits for the DMA but not for the stores.
    if (prev_full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tile_ctr) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // lane 0's ticket slot write
    // One barrier per tile: every wave's pieces have landed, and every wave has finished
    // computing the previous tile, whose buffer the next DMA is about to overwrite.
    __builtin_amdgcn_s_barrier();
    int nxt = v + G;                               // static schedule; also the dynamic one's second tile
    if (tile_ctr && it > 0) nxt = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(&lds[(cur ^ 1) * BUF + BUF - 1]));
    int nimg = 0, nx0 = 0, ny0 = 0;
    const bool more = (unsigned)nxt < (unsigned)total_tiles;   // workgroup-uniform (unsigned: a corrupt counter block cannot send a tile index below zero)
    unsigned ticket = 0u;
    if (more) {
      if (tile_ctr && t == 0) {
        const unsigned zero = 0u, one = 1u;
        asm volatile("global_atomic_add %0, %1, %2, %3 sc0 ; MI_TICKET_DRAW" : "=a"(ticket) : "v"(zero), "a"(one), "s"(my_ctr) : "memory");
      }
      decode(nxt, nimg, nx0, ny0);
      issue(nimg, nx0, ny0, cur ^ 1);
    }
    prev_full = (x0 + TW <= w) && (y0 + TH <= h);  // workgroup-uniform
    const CH *tile = &lds[cur * BUF];
    if constexpr (U8 && BS == 3) corner_compute_u8<R>(tile, score, img, h, w, x0, y0, t);
    else corner_compute<BS, R>(tile, score, img, h, w, x0, y0, t);
    if (tile_ctr && more) {
      // the ticket (tile it + 2): NCH DMA pieces and, on a full tile, R stores were issued after the atomic
      // (one tied statement only: with two, the register copy of the other path lands before its wait)
      if (!prev_full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt vmcnt(%1) ; MI_TICKET_WAIT" : "+a"(ticket) : "n"(NCH + R) : "memory");
      if (t == 0) *reinterpret_cast<int *>(&lds[cur * BUF + BUF - 1]) = pool_base + (int)min(ticket, 0x1ffffffu) * K1_POOLS;
    }
    v = nxt; img = nimg; x0 = nx0; y0 = ny0;
  }
  if (tile_ctr && t < 64) {
    // leave the counters as they were found (zero): the workgroup that reports done last resets them (every ticket of
    // a workgroup has been waited for before it reports, so no late increment can follow the reset)
    unsigned done = 0u;
    if (t == 0) done = atomicAdd(tile_ctr + K1_POOLS * K1_POOL_STRIDE, 1u);
    done = (unsigned)__builtin_amdgcn_readfirstlane((int)done);
    if (done == (unsigned)G - 1u) {
      for (int p = t; p < K1_POOLS; p += 64) __hip_atomic_store(tile_ctr + p * K1_POOL_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == 0) __hip_atomic_store(tile_ctr + K1_POOLS * K1_POOL_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (clk && threadIdx.x == 0) { clk[4 * blockIdx.x + 2] = clock64(); clk[4 * blockIdx.x + 3] = wall_clock64(); }
}

// Generic path: any width, any odd block size.  One thread per pixel, straight from global
// memory (L2 absorbs the re-reads).  Same arithmetic, same clamping rules.
template <typename PIX>
__global__ __launch_bounds__(256) void corner_generic_kernel(MiSets images,
                                                             float *__restrict__ score, int n, int h,
                                                             int w, int bs) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)n * h * w;
  if (idx >= total) return;
  const int x = (int)(idx % w);
  const int y = (int)((idx / w) % h);
  const PIX *im = mi_set_item<PIX>(images, (int)(idx / ((size_t)h * w)), (size_t)h * w);
  const int hp = bs / 2;
  float a = 0.f, c = 0.f, b = 0.f;
  for (int dy = -hp; dy <= hp; ++dy) {
    const int py = clampi(y + dy, 0, h - 1);
    const PIX *r0 = im + (size_t)clampi(py - 1, 0, h - 1) * w;
    const PIX *r1 = im + (size_t)py * w;
    const PIX *r2 = im + (size_t)clampi(py + 1, 0, h - 1) * w;
    for (int dx = -hp; dx <= hp; ++dx) {
      const int px = clampi(x + dx, 0, w - 1);
      const int xl = clampi(px - 1, 0, w - 1), xr = clampi(px + 1, 0, w - 1);
      const float sl = ((float)r0[xl] + (float)r2[xl]) + 2.0f * (float)r1[xl];
      const float sr = ((float)r0[xr] + (float)r2[xr]) + 2.0f * (float)r1[xr];
      const float dl = (float)r2[xl] - (float)r0[xl], dc = (float)r2[px] - (float)r0[px], dr = (float)r2[xr] - (float)r0[xr];
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
int launch_tile(MiSets image, int n, int h, int w, float *score, hipStream_t s) {
  const int tiles_x = ceil_div(w, TW), tiles_y = ceil_div(h, 8 * R);
  const long long blocks = (long long)n * tiles_x * tiles_y;
  if (blocks > 0x7fffffffLL) return MI_E_SHAPE;
  hipLaunchKernelGGL((corner_tile_kernel<BS, R>), dim3((unsigned)blocks), dim3(256), 0, s, image, score,
                     h, w, tiles_x, tiles_y);
  return mi_launch_status();
}

// Persistent grid: 2 workgroups per CU (2 x 80 KiB of LDS) on the 256 CUs of an MI355X.
template <int BS, int R, typename PIX>
int launch_stream(MiSets image, int n, int h, int w, float *score, unsigned *tile_ctr, hipStream_t s) {
  const int tiles_x = ceil_div(w, TW), tiles_y = ceil_div(h, 8 * R);
  const long long total = (long long)n * tiles_x * tiles_y;
  if (total > 0x7fffffffLL) return MI_E_SHAPE;
  // persistent grid: as many workgroups per CU as two LDS buffers allow, on 256 CUs
  constexpr int LH = 8 * R + 2 * (BS / 2 + 1);
  constexpr int PIECE = std::is_same<PIX, float>::value ? 16 : 4;
  constexpr int LDS_BYTES = 2 * ((LH * LW4 + 255) / 256) * 256 * PIECE;
  constexpr int PER_CU_LDS = (160 * 1024) / LDS_BYTES;
  // fp32: LDS allows 4 workgroups per CU.  uint8: LDS would allow 16; the integer stencil needs 82-86 VGPRs at 4 or 5
  // rows per thread (5 waves per SIMD) and 102 at 8 rows (4).
#ifdef MI_K1_U8_PER_CU
  constexpr int PER_CU = std::is_same<PIX, float>::value ? (PER_CU_LDS > 8 ? 8 : PER_CU_LDS) : MI_K1_U8_PER_CU;
#else
  constexpr int PER_CU = std::is_same<PIX, float>::value ? (PER_CU_LDS > 8 ? 8 : PER_CU_LDS) : (R <= 5 ? 5 : 4);
#endif
  const int resident = 256 * PER_CU;
  const int grid = total < resident ? (int)total : resident;
  // dynamic schedule only when a workgroup owns more than two tiles.  The counter block is cleared HERE, on the
  // stream, ahead of every launch that draws from it: a block left dirty by a launch that died between its first
  // draw and its last workgroup's reset would otherwise start this launch's tickets mid-range (tiles skipped, score
  // rows never written, rc 0).  The kernel still leaves the block zero; nothing relies on that any more.
  unsigned *ctr = total > 2LL * grid ? tile_ctr : nullptr;
  if (ctr) {                                       // (a kernel, not hipMemsetAsync: common.h, mi_zero_async)
    const int e = mi_zero_async(ctr, MI_TILE_COUNTER_BYTES, s);
    if (e != MI_OK) return e;
  }
  hipLaunchKernelGGL((corner_stream_kernel<BS, R, std::is_same<PIX, uint8_t>::value>), dim3(grid), dim3(256), 0,
                     s, image, score, h, w, tiles_x, tiles_y, (int)total, ctr, MI_HOOK(corner_clk, (unsigned long long *)nullptr));
  return mi_launch_status();
}

}  // namespace

// The shared launcher: `images` names one batch or two (MiSets), n = the total number of images.
int mi_corner_response_sets(MiSets images, int pix_u8, int n, int h, int w, int block_size, float *score,
                            unsigned *tile_ctr, mi_stream_t stream) {
  if (!images.a || (images.per_set < n && !images.b) || !score) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  if (block_size <= 0 || (block_size & 1) == 0) return MI_E_PARAM;
  hipStream_t s = (hipStream_t)stream;
  const uintptr_t bases = (uintptr_t)images.a | (uintptr_t)images.b;
  const int rows = MI_HOOK(corner_rows, 4);            // rows per thread of the streaming kernel (tile height = 8 * rows)
  const size_t total = (size_t)n * h * w;
  const size_t blocks = (total + 255) / 256;
  if (pix_u8) {
    // 4-pixel DMA pieces: rows must start on a dword (w % 4 == 0, bases 4-byte aligned); score rows are written as float4
    const bool aligned = (w % 4 == 0) && (bases % 4 == 0) && ((uintptr_t)score % 16 == 0);
    if (aligned && h >= 4 && w >= 8 && block_size == 3) {
      // 5 rows per thread unless the test hook asks otherwise (measured per 448 images: 217 us at 5 rows, 221 at 8,
      // 259 at 4 -- the fp32 kernel's best -- with 6 workgroups per CU)
      if (rows == 8) return launch_stream<3, 8, uint8_t>(images, n, h, w, score, tile_ctr, s);
      if (rows == 4 && MI_HOOK(corner_rows_u8_default, 1) == 0) return launch_stream<3, 4, uint8_t>(images, n, h, w, score, tile_ctr, s);
      return launch_stream<3, 5, uint8_t>(images, n, h, w, score, tile_ctr, s);
    }
    if (blocks > 0x7fffffffULL) return MI_E_SHAPE;
    hipLaunchKernelGGL(corner_generic_kernel<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, s, images, score, n, h, w,
                       block_size);
    return mi_launch_status();
  }
  const bool aligned = (w % 4 == 0) && ((bases | (uintptr_t)score) % 16 == 0);
  if (aligned && h >= 4 && w >= 8) {
    if (block_size == 3 && MI_HOOK(corner_impl, 0) == 0) {   // 0 = streaming (LDS-DMA) kernel, 1 = register-staged tile kernel
      if (rows == 4) return launch_stream<3, 4, float>(images, n, h, w, score, tile_ctr, s);
      if (rows == 5) return launch_stream<3, 5, float>(images, n, h, w, score, tile_ctr, s);
      return launch_stream<3, 8, float>(images, n, h, w, score, tile_ctr, s);
    }
    if (block_size == 3) return launch_tile<3, 8>(images, n, h, w, score, s);
    // blocks 5 and 7 (round 4): the same streaming kernel -- the stencil is shared with the tile kernel, the scores are
    // the same bits; per 256 images at block 5: 0.189 ms at 4 rows per thread, 0.201 at 5, 0.196 at 8, tile kernel 0.226
    if (block_size == 5 && MI_HOOK(corner_impl, 0) == 0) {
      if (rows == 4) return launch_stream<5, 4, float>(images, n, h, w, score, tile_ctr, s);
      if (rows == 5) return launch_stream<5, 5, float>(images, n, h, w, score, tile_ctr, s);
      return launch_stream<5, 8, float>(images, n, h, w, score, tile_ctr, s);
    }
    // (block 7, per 256 images: 0.22 ms at 8 rows per thread, 0.253 at 4, tile kernel 0.279)
    if (block_size == 7 && MI_HOOK(corner_impl, 0) == 0) return launch_stream<7, 8, float>(images, n, h, w, score, tile_ctr, s);
    if (block_size == 5) return launch_tile<5, 8>(images, n, h, w, score, s);
    if (block_size == 7) return launch_tile<7, 8>(images, n, h, w, score, s);
  }
  if (blocks > 0x7fffffffULL) return MI_E_SHAPE;
  hipLaunchKernelGGL(corner_generic_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, images, score, n, h, w,
                     block_size);
  return mi_launch_status();
}

extern "C" int mi_corner_response(const float *image, int n, int h, int w, int block_size, float *score,
                                  mi_stream_t stream) {
  MI_ENTER();
  return mi_corner_response_sets(mi_one_set(image, n), 0, n, h, w, block_size, score, nullptr, stream);
}

// u8 ingest (the camera-frame path of sample/visual_odometry.py:65-92 without the host-side
// uint8 -> float32 conversion): the same score map from uint8 pixels, 1 + 4 instead of 4 + 4 bytes per pixel.
extern "C" int mi_corner_response_u8(const uint8_t *image, int n, int h, int w, int block_size, float *score,
                                     mi_stream_t stream) {
  MI_ENTER();
  return mi_corner_response_sets(mi_one_set(image, n), 1, n, h, w, block_size, score, nullptr, stream);
}

// The same two calls with a tile counter: MI_TILE_COUNTER_BYTES of device memory (4-byte aligned, any content: cleared
// by a small zeroing KERNEL ahead of the stencil -- common.h: mi_zero_async; never hipMemsetAsync, whose captured node
// zeroes on the first graph replay only: tools/graph_memset_probe.py), owned by one stream at a time.  With it, large batches hand the tiles of the
// streaming kernel out dynamically (see the schedule note at corner_stream_kernel): same scores, the launch ends
// when the work does instead of when the youngest workgroup's fixed share does.
extern "C" int mi_corner_response_balanced(const void *image, int pixels_are_u8, int n, int h, int w, int block_size,
                                           float *score, uint32_t *tile_counter, mi_stream_t stream) {
  MI_ENTER();
  if (tile_counter && ((uintptr_t)tile_counter % 4) != 0) return MI_E_ALIGN;
  return mi_corner_response_sets(mi_one_set(image, n), pixels_are_u8 ? 1 : 0, n, h, w, block_size, score, tile_counter,
                                 stream);
}

// image1 / image2 of a matcher behind one launch: score is (2 * per_set, h, w), batch a first
extern "C" int mi_corner_response_pair(const void *image_a, const void *image_b, int pixels_are_u8, int per_set, int h,
                                       int w, int block_size, float *score, uint32_t *tile_counter, mi_stream_t stream) {
  MI_ENTER();
  if (!image_b) return MI_E_NULL;
  if (per_set <= 0 || per_set > 0x3fffffff) return MI_E_SHAPE;
  if (tile_counter && ((uintptr_t)tile_counter % 4) != 0) return MI_E_ALIGN;
  return mi_corner_response_sets(MiSets{image_a, image_b, per_set}, pixels_are_u8 ? 1 : 0, 2 * per_set, h, w, block_size,
                                 score, tile_counter, stream);
}
