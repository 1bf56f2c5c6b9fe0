// K5: descriptor cost matrix -> augmented Sinkhorn log-score matrix Z.
// Semantics: reference pytorch_model/matching/sinkhorn.py:95-108 (cost) and :178
// (Z = -cost/epsilon; the constant dustbin padding of :187 is applied inside K6).
//
// bits path  : hard-binarised descriptors stay packed (64 B per 512-bit descriptor); their bits are
//              expanded in registers -- to FP4 nibbles for v_mfma_f32_32x32x64_f8f6f4 (256 / 512 bits),
//              to 0/1 bytes for v_mfma_i32_32x32x32_i8 (other lengths) -- so dot(a,b) = popcount(a & b)
//              is exact.  The epilogue
//              applies cost = |a|^2 + |b|^2 - 2 a.b with a = bit/sqrt(pop) (normalised) or
//              a = bit (Hamming) in fp32 and writes the n x m core of Z.
// f32 path   : arbitrary float descriptors through v_mfma_f32_32x32x2_f32 (exact fp32 fma
//              chain, no reduced precision: epsilon = 0.05 multiplies any cost error by 20).
// l1 path    : |a-b| sums on the VALU from the same LDS tiles.
#include "common.h"
#include "hooks.h"

#include <type_traits>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

// 4 bits -> 4 bytes of 0/1 (bit i -> byte i)
__device__ __forceinline__ int spread4(uint32_t nib) { return (int)(((nib & 15u) * 0x00204081u) & 0x01010101u); }

__device__ __forceinline__ v4i spread16(uint32_t half) {
  v4i r;
  r[0] = spread4(half);
  r[1] = spread4(half >> 4);
  r[2] = spread4(half >> 8);
  r[3] = spread4(half >> 12);
  return r;
}

constexpr int CB_T = 128;  // block tile edge (bits path)

// DOTS = false: write z = -cost/eps as fp32 (pitch in floats).
// DOTS = true : write the exact integer dot products as uint16 (pitch in uint16 elements) plus the
//               per-descriptor (scale, squared norm) pairs; K6 rebuilds z from them on the fly,
//               which halves the bytes every Sinkhorn iteration has to stream.
// WORDS > 0: the descriptor length in 32-bit words at compile time (16 / 8: the two learned tables) -- the k loop is then
// straight-line code: the LDS reads and the bit expansion of later steps run under the MFMAs of earlier ones (rolled, every
// step waited for its own four LDS reads); WORDS = 0: any length.
// FP4 = true (gfx950, WORDS even): the dot products through v_mfma_f32_32x32x64_f8f6f4 on FP4 (E2M1) operands -- a bit
// becomes the nibble 0x2 = 1.0 or 0x0 = 0.0, the products are 0 / 1 and their fp32 sum (<= 4096) is exact, so the result is
// the same popcount.  Twice the K per instruction at the cycles of the int8 form, and the expansion is two instructions
// per operand dword instead of three for half as many dwords: which K index a bit lands on is free as long as both
// operands use the same order, so dword d of a lane's 32 bits simply takes the bits d, d + 4, d + 8, ... where they
// already sit -- ((w << 1) >> d) & 0x22222222 -- instead of being spread out bit by bit.
typedef int v8i __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v8i nibbles_fp4(uint32_t w) {
  v8i r;
  r[0] = (int)((w << 1) & 0x22222222u);
  r[1] = (int)(w & 0x22222222u);
  r[2] = (int)((w >> 1) & 0x22222222u);
  r[3] = (int)((w >> 2) & 0x22222222u);
  r[4] = r[5] = r[6] = r[7] = 0;
  return r;
}

template <bool DOTS, int WORDS, bool FP4 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void cost_bits_kernel(const uint32_t *__restrict__ bits1,
                                                        const uint32_t *__restrict__ bits2, int n, int m,
                                                        int words_rt, int normalized, float eps,
                                                        float *__restrict__ z, int pitch,
                                                        uint16_t *__restrict__ dots,
                                                        float2 *__restrict__ row_info,
                                                        float2 *__restrict__ col_info, uint4 *__restrict__ zero16,
                                                        size_t zero_count) {
  extern __shared__ uint32_t lds_u[];
  const int words = WORDS > 0 ? WORDS : words_rt;
  // side job for mi_match_pairs: clear the next stage's hand-off area (the single-launch Sinkhorn's granule tags),
  // which saves that stage its own zeroing kernel (common.h: mi_zero_async) on the one-pair-per-call path; zero_count 16-byte words, all
  // workgroups share them
  if (zero_count) {
    const size_t nthreads = (size_t)gridDim.x * gridDim.y * gridDim.z * 256;
    const size_t me = (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    for (size_t i = me; i < zero_count; i += nthreads) zero16[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  const int wp = words + 1;                          // +1 word: conflict-free column reads
  uint32_t *sa = lds_u;                              // [128][wp]
  uint32_t *sb = sa + CB_T * wp;                     // [128][wp]
  float *inv_a = reinterpret_cast<float *>(sb + CB_T * wp);  // [128] scale of row descriptors
  float *nrm_a = inv_a + CB_T;                       // [128] squared norms
  float *inv_b = nrm_a + CB_T;
  float *nrm_b = inv_b + CB_T;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int j0 = blockIdx.x * CB_T, i0 = blockIdx.y * CB_T, b = blockIdx.z;
  const uint32_t *a_g = bits1 + (size_t)b * n * words;
  const uint32_t *b_g = bits2 + (size_t)b * m * words;

  if ((words & 3) == 0 && words <= 32 && (((uintptr_t)bits1 | (uintptr_t)bits2) & 15) == 0) {
    // a tile's 128 descriptors are one contiguous block of global memory: 16-byte chunks, all of a thread's
    // loads issued before the first LDS store (one round trip; no per-word division)
    const int cpr = words >> 2;                      // chunks per descriptor
    const int chunks = CB_T * cpr;
    uint4 va[4], vb[4];                              // words <= 32: at most 4 chunks per thread and tile
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      const int r = ch / cpr;
      va[q] = make_uint4(0u, 0u, 0u, 0u);
      vb[q] = va[q];
      if (ch < chunks && i0 + r < n) va[q] = reinterpret_cast<const uint4 *>(a_g + (size_t)i0 * words)[ch];
      if (ch < chunks && j0 + r < m) vb[q] = reinterpret_cast<const uint4 *>(b_g + (size_t)j0 * words)[ch];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = t + 256 * q;
      if (ch < chunks) {
        const int r = ch / cpr, c = (ch - r * cpr) * 4;
        uint32_t *da = sa + r * wp + c, *db = sb + r * wp + c;
        da[0] = va[q].x; da[1] = va[q].y; da[2] = va[q].z; da[3] = va[q].w;
        db[0] = vb[q].x; db[1] = vb[q].y; db[2] = vb[q].z; db[3] = vb[q].w;
      }
    }
  } else {
    for (int i = t; i < CB_T * words; i += 256) {
      const int r = i / words, c = i - r * words;
      sa[r * wp + c] = (i0 + r < n) ? a_g[(size_t)(i0 + r) * words + c] : 0u;
      sb[r * wp + c] = (j0 + r < m) ? b_g[(size_t)(j0 + r) * words + c] : 0u;
    }
  }
  __syncthreads();
  {
    const int r = t & 127;
    const uint32_t *src = (t < 128 ? sa : sb) + r * wp;
    int pop = 0;
    for (int c0 = 0; c0 < words; c0 += 8) {          // eight LDS reads in flight, not one round trip per word
      uint32_t wv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) wv[q] = (c0 + q < words) ? src[c0 + q] : 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q) pop += __popc(wv[q]);
    }
    float inv = 1.0f, nrm = (float)pop;
    if (normalized) {
      // F.normalize: bit / max(sqrt(pop), 1e-12); squared norm re-formed in fp32 like sinkhorn.py:98
      inv = pop > 0 ? 1.0f / sqrtf((float)pop) : 0.0f;
      nrm = (float)pop * (inv * inv);
    }
    (t < 128 ? inv_a : inv_b)[r] = inv;
    (t < 128 ? nrm_a : nrm_b)[r] = nrm;
  }
  __syncthreads();

  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;   // wave's 64x64 sub-tile
  const int lr = lane & 31, lh = lane >> 5;
  typedef typename std::conditional<FP4, v16f, v16i>::type acc_t;
  acc_t acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0;

  static_assert(!FP4 || (WORDS > 0 && WORDS % 2 == 0), "the FP4 form takes two words per step");
  auto kstep = [&](int ks) {
    if constexpr (FP4) {
      v8i fa[2], fb[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        fa[q] = nibbles_fp4(sa[(wm + q * 32 + lr) * wp + 2 * ks + lh]);
        fb[q] = nibbles_fp4(sb[(wn + q * 32 + lr) * wp + 2 * ks + lh]);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[mi], fb[ni], acc[mi][ni], 4, 4, 0, 0, 0, 0);
      return;
    } else {
    v4i fa[2], fb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      fa[q] = spread16(sa[(wm + q * 32 + lr) * wp + ks] >> (16 * lh));
      fb[q] = spread16(sb[(wn + q * 32 + lr) * wp + ks] >> (16 * lh));
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
    }
  };
  if constexpr (FP4) {
#pragma unroll
    for (int ks = 0; ks < WORDS / 2; ++ks) kstep(ks);
  } else if constexpr (WORDS > 0) {
#pragma unroll
    for (int ks = 0; ks < WORDS; ++ks) kstep(ks);
  } else {
#pragma unroll 4
    for (int ks = 0; ks < words; ++ks) kstep(ks);
  }

  if constexpr (DOTS) {
    // per-descriptor (scale, squared norm): written once, by the first tile column / row
    if (blockIdx.x == 0 && t < CB_T && i0 + t < n) row_info[(size_t)b * n + i0 + t] = make_float2(inv_a[t], nrm_a[t]);
    if (blockIdx.y == 0 && t < CB_T && j0 + t < m) col_info[(size_t)b * m + j0 + t] = make_float2(inv_b[t], nrm_b[t]);
    __syncthreads();                                    // bit tiles are dead: reuse LDS as the u16 staging tile
    constexpr int SP16 = CB_T + 8;                      // staging pitch (uint16), rows stay 16-byte aligned
    uint16_t *stage = reinterpret_cast<uint16_t *>(lds_u);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rl = wm + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          const int cl = wn + ni * 32 + lr;
          stage[rl * SP16 + cl] = (uint16_t)(int)acc[mi][ni][e];
        }
    __syncthreads();
    uint16_t *db = dots + (size_t)b * (size_t)n * pitch;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int id = t + 256 * k;
      const int row = id >> 4, c8 = (id & 15) * 8;
      if (i0 + row < n && j0 + c8 < m)                  // pitch >= round_up(m, 8): the 16-byte store stays in the row
        *reinterpret_cast<uint4 *>(db + (size_t)(i0 + row) * pitch + j0 + c8) =
            *reinterpret_cast<const uint4 *>(stage + row * SP16 + c8);
    }
  } else {
    float *zb = z + (size_t)b * (size_t)n * pitch;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rl = wm + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          const int cl = wn + ni * 32 + lr;
          const int gi = i0 + rl, gj = j0 + cl;
          if (gi < n && gj < m) {
            const float dot = (float)acc[mi][ni][e];
            const float cross = dot * (inv_a[rl] * inv_b[cl]);
            const float cost = fmaxf((nrm_a[rl] + nrm_b[cl]) - 2.0f * cross, 0.0f);   // sinkhorn.py:101-103
            zb[(size_t)gi * pitch + gj] = -cost / eps;                                 // sinkhorn.py:178
          }
        }
  }
}

// ------------------------------------------------------------------------------------------
constexpr int CF_T = 64;   // block tile edge (float paths)
constexpr int CF_K = 32;   // k chunk
constexpr int CF_P = CF_K + 1;

template <int DIST>
__global__ __launch_bounds__(256) void cost_f32_kernel(const float *__restrict__ d1,
                                                       const float *__restrict__ d2, int n, int m, int d,
                                                       float eps, float *__restrict__ z, int pitch) {
  __shared__ float sa[CF_T * CF_P];
  __shared__ float sb[CF_T * CF_P];
  __shared__ float nrm_a[CF_T], nrm_b[CF_T];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int j0 = blockIdx.x * CF_T, i0 = blockIdx.y * CF_T, b = blockIdx.z;
  const float *a_g = d1 + (size_t)b * n * d;
  const float *b_g = d2 + (size_t)b * m * d;

  const int srow = t >> 2, sk = (t & 3) * 8;          // staging: 8 consecutive k of one row
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int lr = lane & 31, lh = lane >> 5;

  v16f acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  float l1acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) l1acc[p][q] = 0.0f;
  float ssa = 0.0f, ssb = 0.0f;

  for (int k0 = 0; k0 < d; k0 += CF_K) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int kk = k0 + sk + q;
      const float va = (i0 + srow < n && kk < d) ? a_g[(size_t)(i0 + srow) * d + kk] : 0.0f;
      const float vb = (j0 + srow < m && kk < d) ? b_g[(size_t)(j0 + srow) * d + kk] : 0.0f;
      sa[srow * CF_P + sk + q] = va;
      sb[srow * CF_P + sk + q] = vb;
      ssa += va * va;
      ssb += vb * vb;
    }
    __syncthreads();
    if (DIST == MI_DIST_L2) {
#pragma unroll
      for (int kk = 0; kk < CF_K; kk += 2) {
        const float fa = sa[(wm + lr) * CF_P + kk + lh];
        const float fb = sb[(wn + lr) * CF_P + kk + lh];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
      }
    } else {
      const int ti = (t >> 4) * 4, tj = (t & 15) * 4;  // 4x4 outputs per thread
      for (int kk = 0; kk < CF_K; ++kk) {
        float av[4], bv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { av[p] = sa[(ti + p) * CF_P + kk]; bv[p] = sb[(tj + p) * CF_P + kk]; }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) l1acc[p][q] += fabsf(av[p] - bv[q]);
      }
    }
    __syncthreads();
  }

  float *zb = z + (size_t)b * (size_t)n * pitch;
  if (DIST == MI_DIST_L2) {
    // squared norms: 4 staging threads per row hold partial sums
    ssa += __shfl_xor(ssa, 1, 64); ssa += __shfl_xor(ssa, 2, 64);
    ssb += __shfl_xor(ssb, 1, 64); ssb += __shfl_xor(ssb, 2, 64);
    if ((t & 3) == 0) { nrm_a[srow] = ssa; nrm_b[srow] = ssb; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = wm + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const int cl = wn + lr;
      const int gi = i0 + rl, gj = j0 + cl;
      if (gi < n && gj < m) {
        const float cost = fmaxf((nrm_a[rl] + nrm_b[cl]) - 2.0f * acc[e], 0.0f);
        zb[(size_t)gi * pitch + gj] = -cost / eps;
      }
    }
  } else {
    const int ti = (t >> 4) * 4, tj = (t & 15) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int gi = i0 + ti + p, gj = j0 + tj + q;
        if (gi < n && gj < m) zb[(size_t)gi * pitch + gj] = -l1acc[p][q] / eps;
      }
  }
}

// L2, d % 4 == 0: 128x128 tile, each of the 4 waves owns 64x64 of it as 2x2 MFMA tiles (one LDS fragment
// read per MFMA instead of two), 16-byte staging loads with the next k-slab's loads in flight during the
// MFMAs of the current one.  Same arithmetic as cost_f32_kernel (fp32 fma chain over k in the MFMA).
constexpr int CG_T = 128, CG_K = 32, CG_P = CG_K + 1;

__global__ __launch_bounds__(256) void cost_f32_big_kernel(const float *__restrict__ d1, const float *__restrict__ d2,
                                                           int n, int m, int d, float eps, float *__restrict__ z,
                                                           int pitch) {
  __shared__ float sa[CG_T * CG_P];
  __shared__ float sb[CG_T * CG_P];
  __shared__ float nrm_a[CG_T], nrm_b[CG_T];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int j0 = blockIdx.x * CG_T, i0 = blockIdx.y * CG_T, b = blockIdx.z;
  const float *a_g = d1 + (size_t)b * n * d;
  const float *b_g = d2 + (size_t)b * m * d;
  const int srow = t >> 3, sk = (t & 7) * 4;            // staging: one float4 of rows srow, +32, +64, +96
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int lr = lane & 31, lh = lane >> 5;

  v16f acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
  float ssa[4] = {0.f, 0.f, 0.f, 0.f}, ssb[4] = {0.f, 0.f, 0.f, 0.f};

  auto fetch = [&](int k0, float4 (&va)[4], float4 (&vb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = srow + 32 * q, kk = k0 + sk;
      va[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      vb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i0 + r < n && kk < d) va[q] = *reinterpret_cast<const float4 *>(a_g + (size_t)(i0 + r) * d + kk);
      if (j0 + r < m && kk < d) vb[q] = *reinterpret_cast<const float4 *>(b_g + (size_t)(j0 + r) * d + kk);
    }
  };
  float4 va[4], vb[4], na[4], nb[4];
  fetch(0, va, vb);
  for (int k0 = 0; k0 < d; k0 += CG_K) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float *pa = sa + (srow + 32 * q) * CG_P + sk, *pb = sb + (srow + 32 * q) * CG_P + sk;
      pa[0] = va[q].x; pa[1] = va[q].y; pa[2] = va[q].z; pa[3] = va[q].w;
      pb[0] = vb[q].x; pb[1] = vb[q].y; pb[2] = vb[q].z; pb[3] = vb[q].w;
      ssa[q] += (va[q].x * va[q].x + va[q].y * va[q].y) + (va[q].z * va[q].z + va[q].w * va[q].w);
      ssb[q] += (vb[q].x * vb[q].x + vb[q].y * vb[q].y) + (vb[q].z * vb[q].z + vb[q].w * vb[q].w);
    }
    __syncthreads();
    if (k0 + CG_K < d) fetch(k0 + CG_K, na, nb);        // in flight during the MFMAs below
#pragma unroll
    for (int kk = 0; kk < CG_K; kk += 2) {
      float fa[2], fb[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        fa[q] = sa[(wm + q * 32 + lr) * CG_P + kk + lh];
        fb[q] = sb[(wn + q * 32 + lr) * CG_P + kk + lh];
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) { va[q] = na[q]; vb[q] = nb[q]; }
  }
  // squared norms: the 8 staging threads of a row hold partial sums
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float x = ssa[q], y = ssb[q];
    x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64);
    y += __shfl_xor(y, 1, 64); y += __shfl_xor(y, 2, 64); y += __shfl_xor(y, 4, 64);
    if ((t & 7) == 0) { nrm_a[srow + 32 * q] = x; nrm_b[srow + 32 * q] = y; }
  }
  __syncthreads();
  float *zb = z + (size_t)b * (size_t)n * pitch;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = wm + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int cl = wn + ni * 32 + lr;
        const int gi = i0 + rl, gj = j0 + cl;
        if (gi < n && gj < m) {
          const float cost = fmaxf((nrm_a[rl] + nrm_b[cl]) - 2.0f * acc[mi][ni][e], 0.0f);     // sinkhorn.py:101-103
          zb[(size_t)gi * pitch + gj] = -cost / eps;                                           // :178
        }
      }
}

int check_z(const void *a, const void *b, const void *z, int batch, int n, int m, int pitch, double eps) {
  if (!a || !b || !z) return MI_E_NULL;
  if (batch <= 0 || n <= 0 || m <= 0 || batch > 65535) return MI_E_SHAPE;
  if (pitch < m || pitch % 4 != 0 || ((uintptr_t)z % 16) != 0) return MI_E_ALIGN;
  if (!(eps > 0.0)) return MI_E_PARAM;
  return MI_OK;
}

}  // namespace

extern "C" int mi_cost_logscores_bits(const uint32_t *bits1, const uint32_t *bits2, int batch, int n, int m,
                                      int num_bits, int normalized, double epsilon, float *z, int pitch,
                                      mi_stream_t stream) {
  MI_ENTER();
  int e = check_z(bits1, bits2, z, batch, n, m, pitch, epsilon);
  if (e) return e;
  if (num_bits <= 0 || num_bits % 32 != 0 || num_bits > 4096) return MI_E_PARAM;
  const int words = num_bits / 32;
  const size_t lds = (size_t)2 * CB_T * (words + 1) * 4 + 4 * CB_T * 4;
  dim3 grid(ceil_div(m, CB_T), ceil_div(n, CB_T), batch);
  uint4 *const no_zero = nullptr;
  auto kern = words == 16 ? cost_bits_kernel<false, 16, true> : words == 8 ? cost_bits_kernel<false, 8, true> : cost_bits_kernel<false, 0>;
  if (MI_HOOK(cost_impl, 0) == 1) kern = words == 16 ? cost_bits_kernel<false, 16> : words == 8 ? cost_bits_kernel<false, 8> : cost_bits_kernel<false, 0>;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, bits1, bits2, n, m,
                     words, normalized, (float)epsilon, z, pitch, nullptr, nullptr, nullptr, no_zero, (size_t)0);
  return mi_launch_status();
}

// shared launcher; zero16 / zero_bytes: see cost_bits_kernel's side job (16-byte aligned, a multiple of 16 bytes)
int mi_cost_dots_bits_zeroing(const uint32_t *bits1, const uint32_t *bits2, int batch, int n, int m, int num_bits,
                              int normalized, uint16_t *dots, int pitch, float *row_info, float *col_info,
                              void *zero_ptr, size_t zero_bytes, mi_stream_t stream) {
  if (!bits1 || !bits2 || !dots || !row_info || !col_info) return MI_E_NULL;
  if (batch <= 0 || n <= 0 || m <= 0 || batch > 65535) return MI_E_SHAPE;
  if (pitch < m || pitch % 8 != 0 || ((uintptr_t)dots % 16) != 0 || ((uintptr_t)row_info % 8) != 0 ||
      ((uintptr_t)col_info % 8) != 0 || ((uintptr_t)zero_ptr % 16) != 0 || zero_bytes % 16 != 0)
    return MI_E_ALIGN;
  if (num_bits <= 0 || num_bits % 32 != 0 || num_bits > 4096) return MI_E_PARAM;   // dot <= 4096 fits uint16
  const int words = num_bits / 32;
  size_t lds = (size_t)2 * CB_T * (words + 1) * 4 + 4 * CB_T * 4;
  const size_t stage = (size_t)CB_T * (CB_T + 8) * 2;
  if (lds < stage) lds = stage;
  dim3 grid(ceil_div(m, CB_T), ceil_div(n, CB_T), batch);
  auto kern = words == 16 ? cost_bits_kernel<true, 16, true> : words == 8 ? cost_bits_kernel<true, 8, true> : cost_bits_kernel<true, 0>;
  if (MI_HOOK(cost_impl, 0) == 1) kern = words == 16 ? cost_bits_kernel<true, 16> : words == 8 ? cost_bits_kernel<true, 8> : cost_bits_kernel<true, 0>;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, bits1, bits2, n, m, words,
                     normalized, 1.0f, nullptr, pitch, dots, reinterpret_cast<float2 *>(row_info),
                     reinterpret_cast<float2 *>(col_info), reinterpret_cast<uint4 *>(zero_ptr),
                     zero_ptr ? zero_bytes / 16 : (size_t)0);
  return mi_launch_status();
}

extern "C" int mi_cost_dots_bits(const uint32_t *bits1, const uint32_t *bits2, int batch, int n, int m,
                                 int num_bits, int normalized, uint16_t *dots, int pitch, float *row_info,
                                 float *col_info, mi_stream_t stream) {
  MI_ENTER();
  return mi_cost_dots_bits_zeroing(bits1, bits2, batch, n, m, num_bits, normalized, dots, pitch, row_info, col_info,
                                   nullptr, 0, stream);
}

extern "C" int mi_cost_logscores_f32(const float *desc1, const float *desc2, int batch, int n, int m, int d,
                                     int distance, double epsilon, float *z, int pitch, mi_stream_t stream) {
  MI_ENTER();
  int e = check_z(desc1, desc2, z, batch, n, m, pitch, epsilon);
  if (e) return e;
  if (d <= 0) return MI_E_SHAPE;
  dim3 grid(ceil_div(m, CF_T), ceil_div(n, CF_T), batch);
  const float eps = (float)epsilon;
  if (distance == MI_DIST_L2 && d % 4 == 0 && (((uintptr_t)desc1 | (uintptr_t)desc2) % 16) == 0 && n >= 64 && m >= 64) {
    dim3 big(ceil_div(m, CG_T), ceil_div(n, CG_T), batch);
    hipLaunchKernelGGL(cost_f32_big_kernel, big, dim3(256), 0, (hipStream_t)stream, desc1, desc2, n, m, d, eps, z, pitch);
    return mi_launch_status();
  }
  if (distance == MI_DIST_L2)
    hipLaunchKernelGGL(cost_f32_kernel<MI_DIST_L2>, grid, dim3(256), 0, (hipStream_t)stream, desc1, desc2, n,
                       m, d, eps, z, pitch);
  else if (distance == MI_DIST_L1)
    hipLaunchKernelGGL(cost_f32_kernel<MI_DIST_L1>, grid, dim3(256), 0, (hipStream_t)stream, desc1, desc2, n,
                       m, d, eps, z, pitch);
  else
    return MI_E_PARAM;
  return mi_launch_status();
}
