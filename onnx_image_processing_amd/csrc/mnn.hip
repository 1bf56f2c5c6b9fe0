// K7: mutual-nearest-neighbour match extraction from the Sinkhorn assignment matrix.
// Semantics: reference pytorch_model/matching/match_extraction.py:72-181: argmax over rows and
// columns of P[:n,:m] (first index on ties), mutual check, score >= threshold, the
// max_matches best by score (build tie policy: lower row index first), gather keypoints,
// valid = score > 0; non-matches carry score -1 exactly as the reference's top-k over
// `where(valid, score, -1)` does, and slots beyond n (n < max_matches) are zero-padded.
#include "common.h"
#include "hooks.h"

#include <math.h>

namespace {

constexpr int MX_THREADS = 1024;
constexpr int MX_MAX = 4096;

__device__ __forceinline__ uint64_t best_key(float p, uint32_t idx) {
  return ((uint64_t)__float_as_uint(p) << 32) | (uint64_t)(0xFFFFFFFFu - idx);   // p >= 0
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t k) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint64_t other = __shfl_xor(k, o, 64);
    k = other > k ? other : k;
  }
  return k;
}

// one wave per row i < n: best column
__global__ __launch_bounds__(256) void mnn_row_kernel(const float *__restrict__ p, int n, int m,
                                                      uint64_t *__restrict__ row_best) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, i = blockIdx.x * 4 + wave;
  if (i >= n) return;
  const float *pr = p + ((size_t)b * (n + 1) + i) * (size_t)(m + 1);
  uint64_t k = 0ull;
  for (int j = lane; j < m; j += 64) {
    const uint64_t c = best_key(pr[j], (uint32_t)j);
    k = c > k ? c : k;
  }
  k = wave_max_u64(k);
  if (lane == 0) row_best[(size_t)b * n + i] = k;
}

// 64 columns per workgroup, rows split over 4 waves: best row per column j < m
__global__ __launch_bounds__(256) void mnn_col_kernel(const float *__restrict__ p, int n, int m,
                                                      uint64_t *__restrict__ col_best) {
  __shared__ uint64_t red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, j = blockIdx.x * 64 + lane;
  const float *pb = p + (size_t)b * (n + 1) * (size_t)(m + 1);
  uint64_t k = 0ull;
  if (j < m) {
    for (int i = wave; i < n; i += 4) {
      const uint64_t c = best_key(pb[(size_t)i * (m + 1) + j], (uint32_t)i);
      k = c > k ? c : k;
    }
  }
  red[wave][lane] = k;
  __syncthreads();
  if (wave == 0 && j < m) {
    for (int w = 1; w < 4; ++w) k = red[w][lane] > k ? red[w][lane] : k;
    col_best[(size_t)b * m + j] = k;
  }
}

// Rows AND columns in one pass over P (round 4; m <= 1024): the two kernels above each walk P in loops of dependent loads
// (174 + 150 us per 448 pairs of 513 x 513: 2.7-3.2 TB/s, P read twice).  Here a workgroup takes 32 rows (8 waves x 4), a
// lane the columns lane, lane + 64, ... of its wave's rows with every load issued up front; the rows' winners leave by a
// wave maximum, the columns' winners are merged over the waves in LDS and then over the workgroups of the pair by a 64-bit
// atomic maximum on col_best (zeroed ahead of the launch): maxima of (score, inverted index) keys are exact whatever the
// order, so row_best / col_best are what the two kernels above produce.
template <int Q>
__global__ __launch_bounds__(512) void mnn_p_kernel(const float *__restrict__ p, int n, int m, uint64_t *__restrict__ row_best,
                                                    unsigned long long *__restrict__ col_best) {
  constexpr int NW = 8, RW = 4;
  __shared__ uint64_t red[NW][Q * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y, row0 = ((int)blockIdx.x * NW + wave) * RW;
  float x[RW][Q];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const float *pr = p + ((size_t)b * (n + 1) + min(row0 + r, n - 1)) * (size_t)(m + 1);
#pragma unroll
    for (int q = 0; q < Q; ++q) x[r][q] = pr[min(q * 64 + lane, m - 1)];
  }
  float cbest[Q];
  int cidx[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) { cbest[q] = -1.0f; cidx[q] = 0; }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int i = row0 + r;
    const bool live = i < n;
    float rbest = -1.0f;
    int rj = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = q * 64 + lane;
      const float v = (live && j < m) ? x[r][q] : -1.0f;
      if (v > rbest) { rbest = v; rj = j; }               // columns ascend within a lane: strict > keeps the first
      if (v > cbest[q]) { cbest[q] = v; cidx[q] = i; }    // rows ascend: likewise
    }
    uint64_t key = rbest >= 0.0f ? best_key(rbest, (uint32_t)rj) : 0ull;
    key = wave_max_u64(key);
    if (lane == 0 && live) row_best[(size_t)b * n + i] = key;
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) red[wave][q * 64 + lane] = cbest[q] >= 0.0f ? best_key(cbest[q], (uint32_t)cidx[q]) : 0ull;
  __syncthreads();
  for (int c = threadIdx.x; c < Q * 64 && c < m; c += 64 * NW) {
    uint64_t k = red[0][c];
#pragma unroll
    for (int w = 1; w < NW; ++w) k = red[w][c] > k ? red[w][c] : k;
    if (k != 0ull) atomicMax(col_best + (size_t)b * m + c, (unsigned long long)k);
  }
}

// col_part != NULL (m <= 1024): the per-band column winners of mnn_band_kernel are merged here, in the prologue, instead
// of by a separate mnn_colmerge_kernel launch (one dependent launch less; what one pair per call is made of)
__global__ __launch_bounds__(MX_THREADS) void mnn_select_kernel(
    int n, int m, const uint64_t *__restrict__ row_best, const uint64_t *__restrict__ col_best,
    const uint64_t *__restrict__ col_part, int nb,
    const float *__restrict__ kpts1, const float *__restrict__ kpts2, int max_matches, float threshold,
    const uint32_t *__restrict__ solver_status,
    float *__restrict__ mk1, float *__restrict__ mk2, float *__restrict__ scores,
    uint8_t *__restrict__ valid, int32_t *__restrict__ match_ij) {
  __shared__ uint64_t keys[MX_MAX];
  // the Sinkhorn call that produced the duals timed out (mi_sinkhorn_dots_status_word): no match of this call is valid
  const bool dead = solver_status && *solver_status != 0u;
  __shared__ uint64_t cmerged[1024];
  const int t = threadIdx.x, b = blockIdx.x;
  const uint64_t *rb = row_best + (size_t)b * n;
  const uint64_t *cb = col_best + (size_t)b * m;
  if (col_part) {
    for (int j = t; j < m; j += MX_THREADS) {
      uint64_t k = 0ull;
      int band = 0;
      for (; band + 8 <= nb; band += 8) {            // eight loads in flight
        uint64_t c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = col_part[((size_t)b * nb + band + q) * m + j];
#pragma unroll
        for (int q = 0; q < 8; ++q) k = c[q] > k ? c[q] : k;
      }
      for (; band < nb; ++band) {
        const uint64_t c = col_part[((size_t)b * nb + band) * m + j];
        k = c > k ? c : k;
      }
      cmerged[j] = k;
    }
    __syncthreads();
    cb = cmerged;
  }
  int npad = 2;
  while (npad < n) npad <<= 1;
  for (int i = t; i < npad; i += MX_THREADS) {
    uint64_t key = 0ull;
    if (i < n) {
      // r == 0: the row has no winner (every probability NaN -- poisoned duals, or NaN in a caller's P): no match,
      // and no column to look up
      const uint64_t r = rb[i];
      const uint32_t jw = 0xFFFFFFFFu - (uint32_t)(r & 0xFFFFFFFFull);
      const bool has = r != 0ull && jw < (uint32_t)m;
      const uint32_t j = has ? jw : 0u;
      const float val = __uint_as_float((uint32_t)(r >> 32));
      const uint32_t ib = 0xFFFFFFFFu - (uint32_t)(cb[j] & 0xFFFFFFFFull);
      const bool ok = has && !dead && (ib == (uint32_t)i) && (val >= threshold);
      // high word: 0 for non-matches, bits(score)+1 for matches (keeps score 0.0 above them);
      // low word: inverted row index -> lower row first among equal scores.  +1 so that a real
      // row never collides with the all-zero padding key.
      key = ((uint64_t)(ok ? (uint32_t)(r >> 32) + 1u : 0u) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
    }
    keys[i] = key;
  }
  __syncthreads();
  const uint64_t *sorted = keys;
  if (npad >= 128 && npad <= MX_THREADS) {
    // Counting-rank order of the rows' keys (round 3; the bitonic network below needs 45 barrier-separated stages for
    // 512 keys): real keys are distinct (their low word is the inverted row index), so a key's place in the descending
    // order is the number of keys above it.  1024 / npad threads share a key, each counting over its slice; a wave's
    // lanes hold consecutive keys and the same slice, so every read of the scan is a broadcast.  The upper part of the
    // key array is free for npad <= 1024: [1024, 2048) takes the ordered keys, [2048, ...) the per-key counters.
    uint64_t *ordered = keys + MX_THREADS;
    uint32_t *above_of = reinterpret_cast<uint32_t *>(keys + 2 * MX_THREADS);
    const int tpk = MX_THREADS / npad, slice = npad / tpk;
    const int ki = t & (npad - 1), part = t / npad;
    if (t < npad) { above_of[t] = 0u; ordered[t] = 0ull; }
    const uint64_t mine = keys[ki];
    uint32_t above = 0;
    if (mine != 0ull) {
      const ulonglong2 *lst = reinterpret_cast<const ulonglong2 *>(keys + part * slice);
#pragma unroll 8
      for (int j = 0; j < slice / 2; ++j) {
        const ulonglong2 two = lst[j];
        above += (two.x > mine ? 1u : 0u) + (two.y > mine ? 1u : 0u);
      }
    }
    __syncthreads();
    if (mine != 0ull && above) atomicAdd(&above_of[ki], above);
    __syncthreads();
    if (part == 0 && mine != 0ull) ordered[above_of[ki]] = mine;
    __syncthreads();
    sorted = ordered;
  } else {
    for (int size = 2; size <= npad; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = t; i < (npad >> 1); i += MX_THREADS) {
          const int pos = 2 * i - (i & (stride - 1));
          const int par = pos + stride;
          const bool desc = ((pos & size) == 0);
          const uint64_t a = keys[pos], c = keys[par];
          if ((a < c) == desc) { keys[pos] = c; keys[par] = a; }
        }
        __syncthreads();
      }
    }
  }
  const int cnt = max_matches < n ? max_matches : n;
  for (int s = t; s < max_matches; s += MX_THREADS) {
    float sc = 0.0f;
    uint32_t i = 0;
    bool matched = false;
    if (s < cnt) {
      const uint64_t key = sorted[s];
      i = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
      const uint32_t hi = (uint32_t)(key >> 32);
      matched = hi != 0u;
      sc = matched ? __uint_as_float(hi - 1u) : -1.0f;
    }
    const uint32_t jw = 0xFFFFFFFFu - (uint32_t)(rb[i] & 0xFFFFFFFFull);
    const uint32_t j = jw < (uint32_t)m ? jw : 0u;
    const size_t o = (size_t)b * max_matches + s;
    mk1[o * 2 + 0] = kpts1[((size_t)b * n + i) * 2 + 0];
    mk1[o * 2 + 1] = kpts1[((size_t)b * n + i) * 2 + 1];
    mk2[o * 2 + 0] = kpts2[((size_t)b * m + j) * 2 + 0];
    mk2[o * 2 + 1] = kpts2[((size_t)b * m + j) * 2 + 1];
    scores[o] = sc;
    const bool ok = sc > 0.0f;
    valid[o] = ok ? 1 : 0;
    if (match_ij) {
      match_ij[o * 2 + 0] = ok ? (int32_t)i : -1;
      match_ij[o * 2 + 1] = ok ? (int32_t)j : -1;
    }
  }
}

// ---- mutual-NN straight from the Sinkhorn duals: P is never written ------------------------------
// P_ij = mi_prob_exp((z_ij + u_i) + v_j) is evaluated in registers with the very expression K6's final pass
// uses, so row/column winners (and their scores) are bit-identical to running mi_sinkhorn with a P
// buffer followed by mi_mnn_extract.  One pass over the log-scores instead of a write and two reads
// of P.  A workgroup owns a band of NW*RW rows (a wave holds RW whole rows, 8 consecutive columns
// per lane per 512-column chunk): row winners by a wave reduction, per-band column winners merged
// over the waves in LDS; a second tiny kernel merges the bands.
struct ZSourceF32 {
  const float *z;       // (batch, n, pitch)
  int pitch;
  struct Raw { float4 a, c; };
  struct Row {};
  __device__ __forceinline__ Raw load_raw(int b, int n, int i, int j, int m) const {
    const float *src = z + ((size_t)b * n + i) * pitch + j;
    Raw r;
    r.a = make_float4(0.f, 0.f, 0.f, 0.f);
    r.c = r.a;
    if (j < m) r.a = *reinterpret_cast<const float4 *>(src);           // pitch % 4 == 0
    if (j + 4 < m) r.c = *reinterpret_cast<const float4 *>(src + 4);
    return r;
  }
  __device__ __forceinline__ void decode(const Raw &r, float (&out)[8]) const {
    out[0] = r.a.x; out[1] = r.a.y; out[2] = r.a.z; out[3] = r.a.w;
    out[4] = r.c.x; out[5] = r.c.y; out[6] = r.c.z; out[7] = r.c.w;
  }
  __device__ __forceinline__ Row load_row(int, int, int) const { return Row(); }
  __device__ __forceinline__ void begin_row(const Row &) {}
  __device__ __forceinline__ float col(int, int) const { return 0.0f; }
  __device__ __forceinline__ void finish(float (&)[8], const float2 (&)[8]) const {}
};

struct ZSourceDots {
  const uint16_t *dots;  // (batch, n, pitch) uint16 dot products
  int pitch;
  const float2 *row_info, *col_info;
  float neg_inv_eps;
  float2 ri;
  typedef uint4 Raw;
  typedef float2 Row;
  __device__ __forceinline__ Raw load_raw(int b, int n, int i, int j, int m) const {
    // no branch around the load (it would end the run of loads in flight); lanes past the matrix read a valid
    // chunk of the row
    // (their values are never used: the kernel replaces p by -1 for j >= m)
    return *reinterpret_cast<const uint4 *>(dots + ((size_t)b * n + i) * pitch + min(j, pitch - 8));   // pitch % 8 == 0
  }
  __device__ __forceinline__ void decode(const Raw &r, float (&out)[8]) const {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) out[q] = (float)((q & 1) ? (w[q >> 1] >> 16) : (w[q >> 1] & 0xFFFFu));
  }
  __device__ __forceinline__ Row load_row(int b, int n, int i) const { return row_info[(size_t)b * n + i]; }
  __device__ __forceinline__ void begin_row(const Row &r) { ri = r; }
  __device__ __forceinline__ void finish(float (&x)[8], const float2 (&ci)[8]) const {
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = mi_z_from_dot(x[q], ri, ci[q], neg_inv_eps);
  }
};

// FULL: m == 512 * E8 and n a multiple of the band height (the export configuration): no row or column of the band lies
// past the matrix, the per-element "inside?" select is dropped
// SPLIT (round 4; 512 < m <= 1024, NW = 16): the two 512-column chunks of a row go to two waves (wave 2g: columns 0..511,
// wave 2g + 1: 512..1023 of the same RW rows) running the one-chunk code -- half the registers per lane, twice the
// waves per SIMD -- and the two half-row winners meet in LDS.  Winners are maxima of (score, index) keys: exact,
// whatever the grouping, so the matches are those of the two-chunk kernel.  waves_per_eu(8, 8) for this form only: 64
// VGPRs with 11 dwords of scratch, two 16-wave workgroups per CU (78 KB of LDS each) -- 139.9 us per 128 pairs of
// 1024 x 1024 against 166.6 at 75 VGPRs (one workgroup per CU) and 158.5 for the two-chunk kernel (137 VGPRs).  The
// one-chunk kernel itself loses by the same squeeze (77 -> 110 us per 448 pairs): its three 8-wave workgroups per CU
// already overlap each other's phases.
template <typename SRC, int E8, int RW, int NW, bool FULL = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(SPLIT ? 8 : 1, 8))) void mnn_band_kernel(SRC src, int n, int m, const float *__restrict__ u,
                                                           const float *__restrict__ v,
                                                           const float2 *__restrict__ col_info,
                                                           uint64_t *__restrict__ row_best,
                                                           uint64_t *__restrict__ col_part) {
  static_assert(!SPLIT || (E8 == 1 && NW % 2 == 0), "the split form runs the one-chunk code on wave pairs");
  constexpr int RG = SPLIT ? NW / 2 : NW;        // row groups of the workgroup
  constexpr int BAND = RG * RW;
  constexpr int NC = 512 * E8 * (SPLIT ? 2 : 1);   // columns covered by the workgroup
  __shared__ uint64_t red[RG][NC];
  __shared__ uint64_t halfkey[SPLIT ? NW : 1][RW];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: rows in SGPRs
  const int b = blockIdx.y, band = blockIdx.x, nb = gridDim.x;
  const int rg = SPLIT ? wave >> 1 : wave, half = SPLIT ? wave & 1 : 0, cbase = half * 512;
  const int row0 = band * BAND + rg * RW;

  // all rows of the wave are requested before anything else (a load issued where it is used costs the wave one
  // memory round trip per row), ahead of the column data's trip through LDS
  typename SRC::Raw raw[RW][E8];
  typename SRC::Row rowd[RW];
  float uis[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int ic = min(row0 + r, n - 1);
    uis[r] = u[(size_t)b * (n + 1) + ic];
    rowd[r] = src.load_row(b, n, ic);
#pragma unroll
    for (int e = 0; e < E8; ++e) raw[r][e] = src.load_raw(b, n, ic, cbase + e * 512 + lane * 8, m);
  }
  // per-column data: the workgroup fetches v and col_info once with coalesced loads and every lane picks its
  // eight consecutive columns out of LDS (sixteen strided 4- and 8-byte loads per lane otherwise: the address
  // unit, not the data, was what this prologue cost)
  __shared__ float s_v[NC];
  __shared__ float2 s_ci[NC];
  {
    constexpr int PER = NC / (64 * NW);            // columns per thread (1 or 2): both loads issued, then both stores
    float tv[PER];
    float2 tc[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int c = threadIdx.x + q * 64 * NW;
      tv[q] = c < m ? v[(size_t)b * (m + 1) + c] : 0.0f;
      tc[q] = (col_info && c < m) ? col_info[(size_t)b * m + c] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int c = threadIdx.x + q * 64 * NW;
      s_v[c] = tv[q];
      s_ci[c] = tc[q];
    }
  }
  __syncthreads();
  float vv[E8][8];
  float2 ci[E8][8];
#pragma unroll
  for (int e = 0; e < E8; ++e)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      vv[e][q] = s_v[cbase + e * 512 + lane * 8 + q];
      ci[e][q] = s_ci[cbase + e * 512 + lane * 8 + q];
    }

  float cbest[E8][8];      // per-lane column winners over this wave's rows: rows ascend, strict > keeps the first
  int cidx[E8][8];
#pragma unroll
  for (int e = 0; e < E8; ++e)
#pragma unroll
    for (int q = 0; q < 8; ++q) { cbest[e][q] = -1.0f; cidx[e][q] = 0; }

#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int i = row0 + r;
    const bool live = i < n;
    const float ui = uis[r];
    src.begin_row(rowd[r]);
    float rbest = -1.0f;   // per-lane row winner: columns ascend within a lane
    int rj = 0;
#pragma unroll
    for (int e = 0; e < E8; ++e) {
      float x[8];
      src.decode(raw[r][e], x);
      src.finish(x, ci[e]);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = cbase + e * 512 + lane * 8 + q;
        float p = mi_prob_exp((x[q] + ui) + vv[e][q]);                         // sinkhorn.py:145,206
        if constexpr (!FULL) p = (live && j < m) ? p : -1.0f;
        if (p > rbest) { rbest = p; rj = j; }
        if (p > cbest[e][q]) { cbest[e][q] = p; cidx[e][q] = i; }
      }
    }
    uint64_t key = rbest >= 0.0f ? best_key(rbest, (uint32_t)rj) : 0ull;
    key = wave_max_u64(key);
    if constexpr (SPLIT) {
      if (lane == 0) halfkey[wave][r] = key;
    } else {
      if (lane == 0 && live) row_best[(size_t)b * n + i] = key;
    }
  }
#pragma unroll
  for (int e = 0; e < E8; ++e)
#pragma unroll
    for (int q = 0; q < 8; ++q)
      red[rg][cbase + e * 512 + lane * 8 + q] = cbest[e][q] >= 0.0f ? best_key(cbest[e][q], (uint32_t)cidx[e][q]) : 0ull;
  __syncthreads();
  if constexpr (SPLIT) {                         // the rows' winners: the better of the two halves
    if (half == 0 && lane < RW && row0 + lane < n) {
      const uint64_t a = halfkey[wave][lane], c = halfkey[wave + 1][lane];
      row_best[(size_t)b * n + row0 + lane] = c > a ? c : a;
    }
  }
  for (int c = threadIdx.x; c < NC && c < m; c += 64 * NW) {
    uint64_t k = red[0][c];
#pragma unroll
    for (int w = 1; w < RG; ++w) k = red[w][c] > k ? red[w][c] : k;
    col_part[((size_t)b * nb + band) * m + c] = k;
  }
}

__global__ __launch_bounds__(256) void mnn_colmerge_kernel(const uint64_t *__restrict__ col_part, int nb, int m,
                                                           uint64_t *__restrict__ col_best) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  uint64_t k = 0ull;
  int band = 0;
  for (; band + 8 <= nb; band += 8) {              // eight loads in flight (one round trip, not eight)
    uint64_t c[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] = col_part[((size_t)b * nb + band + q) * m + j];
#pragma unroll
    for (int q = 0; q < 8; ++q) k = c[q] > k ? c[q] : k;
  }
  for (; band < nb; ++band) {
    const uint64_t c = col_part[((size_t)b * nb + band) * m + j];
    k = c > k ? c : k;
  }
  col_best[(size_t)b * m + j] = k;
}

constexpr int DUALS_BAND = 32;     // rows per workgroup of mnn_band_kernel (8 waves x 4, or 16 x 2)

struct DualsWork {
  uint64_t *row_best, *col_best, *col_part;
};
size_t duals_bytes(int batch, int n, int m) {
  return ((size_t)batch * n + (size_t)batch * m + (size_t)batch * ceil_div(n, DUALS_BAND) * m) * sizeof(uint64_t);
}
DualsWork duals_carve(void *workspace, int batch, int n, int m) {
  DualsWork w;
  w.row_best = reinterpret_cast<uint64_t *>(workspace);
  w.col_best = w.row_best + (size_t)batch * n;
  w.col_part = w.col_best + (size_t)batch * m;
  return w;
}

template <typename SRC>
int mnn_from_source(SRC src, const float2 *col_info, int batch, int n, int m, const float *u, const float *v,
                    const float *kpts1, const float *kpts2, int max_matches, float threshold, void *workspace,
                    size_t workspace_bytes, const uint32_t *solver_status, float *mk1, float *mk2, float *scores,
                    uint8_t *valid, int32_t *match_ij, hipStream_t s) {
  if (!u || !v || !kpts1 || !kpts2 || !workspace || !mk1 || !mk2 || !scores || !valid) return MI_E_NULL;
  if (batch <= 0 || n <= 0 || m <= 0 || batch > 65535) return MI_E_SHAPE;
  if (n > MX_MAX || m > 1024 || max_matches <= 0) return MI_E_PARAM;
  if (((uintptr_t)workspace % 8) != 0) return MI_E_ALIGN;
  if (workspace_bytes < duals_bytes(batch, n, m)) return MI_E_CAPACITY;
  const DualsWork w = duals_carve(workspace, batch, n, m);
  const int nb = ceil_div(n, DUALS_BAND);
  const bool full = n % DUALS_BAND == 0;
  if (m == 512 && full) {
    hipLaunchKernelGGL((mnn_band_kernel<SRC, 1, 4, 8, true>), dim3(nb, batch), dim3(512), 0, s, src, n, m, u, v, col_info,
                       w.row_best, w.col_part);
  } else if (m <= 512) {
    hipLaunchKernelGGL((mnn_band_kernel<SRC, 1, 4, 8>), dim3(nb, batch), dim3(512), 0, s, src, n, m, u, v, col_info,
                       w.row_best, w.col_part);
  } else if (MI_HOOK(mnn_pair_waves, 1) == 0) {
    hipLaunchKernelGGL((mnn_band_kernel<SRC, 2, 4, 8>), dim3(nb, batch), dim3(512), 0, s, src, n, m, u, v, col_info,
                       w.row_best, w.col_part);
  } else if (m == 1024 && full) {                  // 512 < m <= 1024: two waves per row group, one chunk each (SPLIT)
    hipLaunchKernelGGL((mnn_band_kernel<SRC, 1, 4, 16, true, true>), dim3(nb, batch), dim3(1024), 0, s, src, n, m, u, v,
                       col_info, w.row_best, w.col_part);
  } else {
    hipLaunchKernelGGL((mnn_band_kernel<SRC, 1, 4, 16, false, true>), dim3(nb, batch), dim3(1024), 0, s, src, n, m, u, v,
                       col_info, w.row_best, w.col_part);
  }
  if (batch <= 32) {
    // few pairs: the select kernel merges the bands' column winners itself (one launch less on the latency path)
    hipLaunchKernelGGL(mnn_select_kernel, dim3(batch), dim3(MX_THREADS), 0, s, n, m, w.row_best, w.col_best,
                       (const uint64_t *)w.col_part, nb, kpts1, kpts2, max_matches, threshold, solver_status, mk1, mk2,
                       scores, valid, match_ij);
    return mi_launch_status();
  }
  hipLaunchKernelGGL(mnn_colmerge_kernel, dim3(ceil_div(m, 256), batch), dim3(256), 0, s, w.col_part, nb, m,
                     w.col_best);
  hipLaunchKernelGGL(mnn_select_kernel, dim3(batch), dim3(MX_THREADS), 0, s, n, m, w.row_best, w.col_best,
                     (const uint64_t *)nullptr, 0, kpts1, kpts2, max_matches, threshold, solver_status, mk1, mk2, scores,
                     valid, match_ij);
  return mi_launch_status();
}

}  // namespace

extern "C" size_t mi_mnn_duals_workspace_bytes(int batch, int n, int m) {
  if (batch <= 0 || n <= 0 || m <= 0 || m > 1024 || n > MX_MAX) return 0;
  return duals_bytes(batch, n, m);
}

extern "C" int mi_mnn_from_duals(const float *z, int batch, int n, int m, int pitch, const float *u, const float *v,
                                 const float *kpts1, const float *kpts2, int max_matches, float threshold,
                                 void *workspace, size_t workspace_bytes, float *mk1, float *mk2, float *scores,
                                 uint8_t *valid, int32_t *match_ij, mi_stream_t stream) {
  MI_ENTER();
  if (!z) return MI_E_NULL;
  if (pitch < m || pitch % 4 != 0 || ((uintptr_t)z % 16) != 0) return MI_E_ALIGN;
  ZSourceF32 src;
  src.z = z;
  src.pitch = pitch;
  return mnn_from_source(src, nullptr, batch, n, m, u, v, kpts1, kpts2, max_matches, threshold, workspace,
                         workspace_bytes, nullptr, mk1, mk2, scores, valid, match_ij, (hipStream_t)stream);
}

extern "C" int mi_mnn_from_duals_dots(const uint16_t *dots, const float *row_info, const float *col_info, int batch,
                                      int n, int m, int pitch, double epsilon, const float *u, const float *v,
                                      const float *kpts1, const float *kpts2, int max_matches, float threshold,
                                      void *workspace, size_t workspace_bytes, const uint32_t *solver_status,
                                      float *mk1, float *mk2, float *scores, uint8_t *valid, int32_t *match_ij,
                                      mi_stream_t stream) {
  MI_ENTER();
  if (!dots || !row_info || !col_info) return MI_E_NULL;
  if (pitch < m || pitch % 8 != 0 || ((uintptr_t)dots % 16) != 0) return MI_E_ALIGN;
  if (!(epsilon > 0.0)) return MI_E_PARAM;
  ZSourceDots src;
  src.dots = dots;
  src.pitch = pitch;
  src.row_info = reinterpret_cast<const float2 *>(row_info);
  src.col_info = reinterpret_cast<const float2 *>(col_info);
  src.neg_inv_eps = (float)(-1.0 / epsilon);
  src.ri = make_float2(0.f, 0.f);
  if (solver_status && ((uintptr_t)solver_status % 4) != 0) return MI_E_ALIGN;
  return mnn_from_source(src, src.col_info, batch, n, m, u, v, kpts1, kpts2, max_matches, threshold, workspace,
                         workspace_bytes, solver_status, mk1, mk2, scores, valid, match_ij, (hipStream_t)stream);
}

extern "C" int mi_mnn_extract(const float *p, int batch, int n, int m, const float *kpts1, const float *kpts2,
                              int max_matches, float threshold, uint64_t *row_best, uint64_t *col_best,
                              float *mk1, float *mk2, float *scores, uint8_t *valid, int32_t *match_ij,
                              mi_stream_t stream) {
  MI_ENTER();
  if (!p || !kpts1 || !kpts2 || !row_best || !col_best || !mk1 || !mk2 || !scores || !valid) return MI_E_NULL;
  if (batch <= 0 || n <= 0 || m <= 0 || batch > 65535) return MI_E_SHAPE;
  if (n > MX_MAX || max_matches <= 0) return MI_E_PARAM;
  if (((uintptr_t)row_best % 8) != 0 || ((uintptr_t)col_best % 8) != 0) return MI_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (m <= 1024 && MI_HOOK(mnn_one_pass, 1) != 0) {
    const int e = mi_zero_async(col_best, (size_t)batch * m * sizeof(uint64_t), s);   // (a kernel, not hipMemsetAsync: common.h)
    if (e != MI_OK) return e;
    if (m <= 512)
      hipLaunchKernelGGL(mnn_p_kernel<8>, dim3(ceil_div(n, 32), batch), dim3(512), 0, s, p, n, m, row_best,
                         reinterpret_cast<unsigned long long *>(col_best));
    else
      hipLaunchKernelGGL(mnn_p_kernel<16>, dim3(ceil_div(n, 32), batch), dim3(512), 0, s, p, n, m, row_best,
                         reinterpret_cast<unsigned long long *>(col_best));
  } else {
    hipLaunchKernelGGL(mnn_row_kernel, dim3(ceil_div(n, 4), batch), dim3(256), 0, s, p, n, m, row_best);
    hipLaunchKernelGGL(mnn_col_kernel, dim3(ceil_div(m, 64), batch), dim3(256), 0, s, p, n, m, col_best);
  }
  hipLaunchKernelGGL(mnn_select_kernel, dim3(batch), dim3(MX_THREADS), 0, s, n, m, row_best, col_best,
                     (const uint64_t *)nullptr, 0, kpts1, kpts2, max_matches, threshold, (const uint32_t *)nullptr, mk1,
                     mk2, scores, valid, match_ij);
  return mi_launch_status();
}

// ---- outlier filters on P: reference matching/sinkhorn.py:317-465 (SinkhornMatcherWithFilters) --
// Per row i < n of P: best and second-best core probability (top-2 with multiplicity, as
// torch.topk), dustbin entry d = P[i, m].  valid = (ratio_threshold <= 0 or best/(second+1e-8) >=
// ratio_threshold) and (dustbin_margin < 0 or best - d >= dustbin_margin).  Rows that fail are
// rewritten in place as the reference does (:441-463): core entries *0, dustbin entry 1.
namespace {
// p: rows of `row_stride` floats, planes of `plane_stride`; the row's dustbin entry sits at column m when
// HAS_DUST.  REWRITE: failing rows are rewritten in place (SinkhornMatcherWithFilters); otherwise p is only read
// (the mask-only form of matching/outlier_filters.py).
template <bool REWRITE, bool HAS_DUST>
__global__ __launch_bounds__(256) void match_filters_kernel(float *__restrict__ p, int n, int m, size_t row_stride,
                                                            size_t plane_stride, float ratio_threshold,
                                                            float dustbin_margin, uint8_t *__restrict__ valid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, i = blockIdx.x * 4 + wave;
  if (i >= n) return;
  float *pr = p + (size_t)b * plane_stride + (size_t)i * row_stride;
  float a1 = -INFINITY, a2 = -INFINITY;             // lane-local two largest
  for (int j0 = 0; j0 < m; j0 += 512) {             // eight loads in flight, then the same updates in the same order
    float x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = j0 + q * 64 + lane;
      x[q] = j < m ? pr[j] : -INFINITY;             // (-inf changes neither maximum: x > a1 / x > a2 are false)
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (x[q] > a1) { a2 = a1; a1 = x[q]; } else if (x[q] > a2) { a2 = x[q]; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float b1 = __shfl_xor(a1, o, 64), b2 = __shfl_xor(a2, o, 64);
    const float hi = fmaxf(a1, b1), lo = fminf(a1, b1);
    a2 = fmaxf(lo, fmaxf(a2, b2));
    a1 = hi;
  }
  const float best = a1;
  const float second = (m >= 2) ? a2 : 0.0f;        // sinkhorn.py:346-348
  const float dust = HAS_DUST ? pr[m] : 0.0f;
  bool ok = true;
  if (ratio_threshold > 0.0f) ok = ok && (best / (second + 1e-8f) >= ratio_threshold);   // :350-351
  if (HAS_DUST && dustbin_margin >= 0.0f) ok = ok && ((best - dust) >= dustbin_margin);  // :384-386
  if (lane == 0) valid[(size_t)b * n + i] = ok ? 1 : 0;
  if (REWRITE && !ok) {
    for (int j = lane; j < m; j += 64) pr[j] = pr[j] * 0.0f;
    if (lane == 0) pr[m] = 1.0f + 0.0f * dust;
  }
}
}  // namespace

extern "C" int mi_match_filters(float *p, int batch, int n, int m, float ratio_threshold, float dustbin_margin,
                                uint8_t *valid, mi_stream_t stream) {
  MI_ENTER();
  if (!p || !valid) return MI_E_NULL;
  if (batch <= 0 || n <= 0 || m <= 0 || batch > 65535) return MI_E_SHAPE;
  hipLaunchKernelGGL((match_filters_kernel<true, true>), dim3(ceil_div(n, 4), batch), dim3(256), 0, (hipStream_t)stream, p,
                     n, m, (size_t)(m + 1), (size_t)(n + 1) * (size_t)(m + 1), ratio_threshold, dustbin_margin, valid);
  return mi_launch_status();
}

extern "C" int mi_match_filter_masks(const float *p, int batch, int n, int m, int has_dustbin, float ratio_threshold,
                                     float dustbin_margin, uint8_t *valid, mi_stream_t stream) {
  MI_ENTER();
  if (!p || !valid) return MI_E_NULL;
  if (batch <= 0 || n <= 0 || m <= 0 || batch > 65535) return MI_E_SHAPE;
  if (!has_dustbin && dustbin_margin >= 0.0f) return MI_E_PARAM;       // a margin test needs the dustbin column
  float *q = const_cast<float *>(p);                                    // REWRITE = false: only read
  const dim3 grid(ceil_div(n, 4), batch);
  if (has_dustbin)
    hipLaunchKernelGGL((match_filters_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, q, n, m,
                       (size_t)(m + 1), (size_t)(n + 1) * (size_t)(m + 1), ratio_threshold, dustbin_margin, valid);
  else
    hipLaunchKernelGGL((match_filters_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, q, n, m, (size_t)m,
                       (size_t)n * (size_t)m, ratio_threshold, dustbin_margin, valid);
  return mi_launch_status();
}

// ---- SinkhornMatcherWithScores (matching/sinkhorn.py:228-259): row / column maxima of the core of P
namespace {
__global__ __launch_bounds__(256) void core_rowmax_kernel(const float *__restrict__ p, int n, int m,
                                                          float *__restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, i = blockIdx.x * 4 + wave;
  if (i >= n) return;
  const float *pr = p + ((size_t)b * (n + 1) + i) * (size_t)(m + 1);
  float mx = -INFINITY;
  for (int j = lane; j < m; j += 64) mx = fmaxf(mx, pr[j]);
  mx = wave_max_dpp(mx);
  if (lane == 0) out[(size_t)b * n + i] = mx;
}
__global__ __launch_bounds__(256) void core_colmax_kernel(const float *__restrict__ p, int n, int m,
                                                          float *__restrict__ out) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const float *pb = p + (size_t)b * (n + 1) * (size_t)(m + 1) + j;
  float mx = -INFINITY;
  for (int i = 0; i < n; ++i) mx = fmaxf(mx, pb[(size_t)i * (m + 1)]);
  out[(size_t)b * m + j] = mx;
}
}  // namespace

extern "C" int mi_core_maxima(const float *p, int batch, int n, int m, float *row_max, float *col_max,
                              mi_stream_t stream) {
  MI_ENTER();
  if (!p || !row_max || !col_max) return MI_E_NULL;
  if (batch <= 0 || n <= 0 || m <= 0 || batch > 65535) return MI_E_SHAPE;
  hipLaunchKernelGGL(core_rowmax_kernel, dim3(ceil_div(n, 4), batch), dim3(256), 0, (hipStream_t)stream, p, n, m, row_max);
  hipLaunchKernelGGL(core_colmax_kernel, dim3(ceil_div(m, 256), batch), dim3(256), 0, (hipStream_t)stream, p, n, m,
                     col_max);
  return mi_launch_status();
}

extern "C" int mi_abi_version(void) { return 3; }

extern "C" const char *mi_error_string(int code) {
  switch (code) {
    case MI_OK: return "ok";
    case MI_E_NULL: return "required pointer is NULL";
    case MI_E_SHAPE: return "non-positive or inconsistent extent";
    case MI_E_PARAM: return "parameter outside the supported set";
    case MI_E_CAPACITY: return "workspace or capacity too small";
    case MI_E_ALIGN: return "pointer or pitch not aligned as documented";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}
