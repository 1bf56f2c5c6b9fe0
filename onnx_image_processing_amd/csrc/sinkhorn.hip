// K6: log-space Sinkhorn with dustbins.
// Semantics: reference pytorch_model/matching/sinkhorn.py:112-147 (iterations) and :178-206
// (augmented matrix, marginals, exp).  u, v start at 0; every iteration is
//     u_i = log mu_i - LSE_j(Z_ij + v_j)   then   v_j = log nu_j - LSE_i(Z_ij + u_i)
// over the (n+1) x (m+1) augmented matrix whose last row/column/corner are the constant
// dustbin score.  Those constants are never stored: Z holds only the n x m core (rows are
// 16-byte aligned, pitch % 4 == 0) and the dustbin terms are added analytically.
//
// Both passes are pure streaming reductions over Z (bandwidth-bound: 4 B per element per
// pass).  Row pass: one wave per row, 16-byte loads, the row lives in registers, max and sum
// by wave shuffles.  Column pass: one workgroup per strip of columns, rows split over the 4
// waves, per-lane online (max, sum) carried down the column in chunks of 8 rows so 8 loads
// are in flight per lane; partials merged through LDS.
#include "common.h"

#include <math.h>

#include "hooks.h"

// test hook (mi_debug_set key 4, include/mi355x_match_debug.h): 0 = probability-form band kernel, lean instruction stream (default);
// 2 = the first probability-form kernel; 1 = log-domain (max, sum) band partials, two exps per element

namespace {

constexpr int ROWS_PER_WAVE = 4;

// exp(d) for d <= 0 in the O(n*m) inner sums: one multiply + v_exp_f32 (2^x, <= 1 ulp) instead
// of the ~15-instruction libm expf.  The argument's rounding error (|d| * log2e * 2^-24, at most
// ~4e-6 for |d| <= 60) bounds the term's relative error; terms that far below the row maximum
// do not move the sum.  Merges of partials, log() and the final P = exp(...) use libm.
__device__ __forceinline__ float sk_exp(float d) { return __builtin_amdgcn_exp2f(d * 1.4426950408889634f); }

// ---- row pass -----------------------------------------------------------------------------
// E4 = float4 loads per lane: covers m <= 256 * E4 columns.
template <int E4>
__global__ __launch_bounds__(256) void sk_row_kernel(const float *__restrict__ z, int n, int m, int pitch,
                                                     float dust, const float *__restrict__ v,
                                                     float *__restrict__ u, float log_m, int v_is_zero) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const float *vb = v + (size_t)b * (m + 1);
  float vv[E4][4];
#pragma unroll
  for (int e = 0; e < E4; ++e)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = e * 256 + lane * 4 + q;
      vv[e][q] = (j < m) ? (v_is_zero ? 0.0f : vb[j]) : -INFINITY;
    }
  const float vd = v_is_zero ? 0.0f : vb[m];
  const float xd = dust + vd;                       // dustbin column term of every real row

  const int row0 = (blockIdx.x * 4 + wave) * ROWS_PER_WAVE;
#pragma unroll 1
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
    const int i = row0 + rr;
    if (i > n) break;
    float x[E4][4];
    if (i < n) {
      const float *zr = z + ((size_t)b * n + i) * pitch;
#pragma unroll
      for (int e = 0; e < E4; ++e) {
        const int j = e * 256 + lane * 4;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < m) q = *reinterpret_cast<const float4 *>(zr + j);   // pitch >= round_up(m,4): in bounds
        // columns >= m (row padding) are masked explicitly: the padding may hold anything
        x[e][0] = (j + 0 < m) ? q.x + vv[e][0] : -INFINITY;
        x[e][1] = (j + 1 < m) ? q.y + vv[e][1] : -INFINITY;
        x[e][2] = (j + 2 < m) ? q.z + vv[e][2] : -INFINITY;
        x[e][3] = (j + 3 < m) ? q.w + vv[e][3] : -INFINITY;
      }
    } else {                                         // dustbin row: Z = dust everywhere
#pragma unroll
      for (int e = 0; e < E4; ++e)
#pragma unroll
        for (int q = 0; q < 4; ++q) x[e][q] = dust + vv[e][q];
    }
    float mx = xd;
#pragma unroll
    for (int e = 0; e < E4; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) mx = fmaxf(mx, x[e][q]);
    mx = wave_max(mx);
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < E4; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) s += sk_exp(x[e][q] - mx);        // exp(-inf) = 0 for padding lanes
    s = wave_sum(s) + expf(xd - mx);
    if (lane == 0) u[(size_t)b * (n + 1) + i] = ((i == n) ? log_m : 0.0f) - (logf(s) + mx);
  }
}

// Any m: strided scalar loads, two sweeps (max, then sum).
__global__ __launch_bounds__(256) void sk_row_generic_kernel(const float *__restrict__ z, int n, int m,
                                                             int pitch, float dust,
                                                             const float *__restrict__ v,
                                                             float *__restrict__ u, float log_m,
                                                             int v_is_zero) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const float *vb = v + (size_t)b * (m + 1);
  const float xd = dust + (v_is_zero ? 0.0f : vb[m]);
  const int row0 = (blockIdx.x * 4 + wave) * ROWS_PER_WAVE;
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
    const int i = row0 + rr;
    if (i > n) break;
    const float *zr = z + ((size_t)b * n + (i < n ? i : 0)) * pitch;
    float mx = xd;
    for (int j = lane; j < m; j += 64) mx = fmaxf(mx, (i < n ? zr[j] : dust) + (v_is_zero ? 0.0f : vb[j]));
    mx = wave_max(mx);
    float s = 0.0f;
    for (int j = lane; j < m; j += 64) s += expf(((i < n ? zr[j] : dust) + (v_is_zero ? 0.0f : vb[j])) - mx);
    s = wave_sum(s) + expf(xd - mx);
    if (lane == 0) u[(size_t)b * (n + 1) + i] = ((i == n) ? log_m : 0.0f) - (logf(s) + mx);
  }
}

// ---- column pass --------------------------------------------------------------------------
// Strip of 64*VEC columns per workgroup; wave w takes rows w, w+4, ... in chunks of CH.
template <int VEC>
__global__ __launch_bounds__(256) void sk_col_kernel(const float *__restrict__ z, int n, int m, int pitch,
                                                     float dust, const float *__restrict__ u,
                                                     float *__restrict__ v, float log_n) {
  constexpr int CH = 8;
  __shared__ float red_m[4][64 * VEC];
  __shared__ float red_s[4][64 * VEC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int j = blockIdx.x * 64 * VEC + lane * VEC;          // first column of this lane
  const float *ub = u + (size_t)b * (n + 1);
  const float *zb = z + (size_t)b * n * pitch;
  // the last strip index (gridDim.x - 1) is the dustbin column: handled below
  const bool dust_col = (blockIdx.x == gridDim.x - 1);

  float mx[VEC], s[VEC];
#pragma unroll
  for (int c = 0; c < VEC; ++c) { mx[c] = -INFINITY; s[c] = 0.0f; }

  if (!dust_col) {
    const bool act = j < m;                                   // m % VEC == 0 guaranteed by the host
    for (int i0 = wave * CH; i0 < n; i0 += 4 * CH) {
      float x[CH][VEC];
#pragma unroll
      for (int r = 0; r < CH; ++r) {
        const int i = i0 + r;
        if (i < n && act) {
          const float ui = ub[i];
          const float *p = zb + (size_t)i * pitch + j;
          if (VEC == 4) {
            const float4 q = *reinterpret_cast<const float4 *>(p);
            x[r][0] = q.x + ui; x[r][1 % VEC] = q.y + ui; x[r][2 % VEC] = q.z + ui; x[r][3 % VEC] = q.w + ui;
          } else if (VEC == 2) {
            const float2 q = *reinterpret_cast<const float2 *>(p);
            x[r][0] = q.x + ui; x[r][1 % VEC] = q.y + ui;
          } else {
            x[r][0] = p[0] + ui;
          }
        } else {
#pragma unroll
          for (int c = 0; c < VEC; ++c) x[r][c] = -INFINITY;
        }
      }
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        float cm = x[0][c];
#pragma unroll
        for (int r = 1; r < CH; ++r) cm = fmaxf(cm, x[r][c]);
        if (cm > mx[c]) { s[c] *= expf(mx[c] - cm); mx[c] = cm; }   // exp(-inf - cm) = 0 on first use
        if (mx[c] > -INFINITY) {
#pragma unroll
          for (int r = 0; r < CH; ++r) s[c] += sk_exp(x[r][c] - mx[c]);
        }
      }
    }
  } else {
    // dustbin column: x_i = dust + u_i for i <= n ; lanes stride over rows, reduced below
    for (int i = threadIdx.x; i <= n; i += 256) {
      const float x = dust + ub[i];
      if (x > mx[0]) { s[0] *= expf(mx[0] - x); mx[0] = x; }
      s[0] += expf(x - mx[0]);
    }
  }

  if (dust_col) {
    // merge all 256 partials: wave shuffle, then LDS across waves
    float gm = wave_max(mx[0]);
    float gs = wave_sum(mx[0] > -INFINITY ? s[0] * expf(mx[0] - gm) : 0.0f);
    if (lane == 0) { red_m[wave][0] = gm; red_s[wave][0] = gs; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float fm = red_m[0][0];
      for (int w = 1; w < 4; ++w) fm = fmaxf(fm, red_m[w][0]);
      float fs = 0.0f;
      for (int w = 0; w < 4; ++w) fs += red_s[w][0] * expf(red_m[w][0] - fm);
      v[(size_t)b * (m + 1) + m] = log_n - (logf(fs) + fm);
    }
    return;
  }

#pragma unroll
  for (int c = 0; c < VEC; ++c) { red_m[wave][lane * VEC + c] = mx[c]; red_s[wave][lane * VEC + c] = s[c]; }
  __syncthreads();
  if (wave == 0 && j < m) {
    const float xd = dust + ub[n];                           // dustbin row term of every real column
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      float fm = xd;
      for (int w = 0; w < 4; ++w) fm = fmaxf(fm, red_m[w][lane * VEC + c]);
      float fs = expf(xd - fm);
      for (int w = 0; w < 4; ++w) {
        const float pm = red_m[w][lane * VEC + c];
        if (pm > -INFINITY) fs += red_s[w][lane * VEC + c] * expf(pm - fm);
      }
      v[(size_t)b * (m + 1) + j + c] = 0.0f - (logf(fs) + fm);
    }
  }
}

// ---- P = exp(Z + u + v) over the augmented matrix -------------------------------------------
__global__ __launch_bounds__(256) void sk_exp_kernel(const float *__restrict__ z, int n, int m, int pitch,
                                                     float dust, const float *__restrict__ u,
                                                     const float *__restrict__ v, float *__restrict__ p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int i = blockIdx.x * 4 + wave;
  if (i > n) return;
  const float ui = u[(size_t)b * (n + 1) + i];
  const float *vb = v + (size_t)b * (m + 1);
  const float *zr = z + ((size_t)b * n + (i < n ? i : 0)) * pitch;
  float *pr = p + ((size_t)b * (n + 1) + i) * (size_t)(m + 1);
  for (int j = lane; j <= m; j += 64) {
    const float zz = (i < n && j < m) ? zr[j] : dust;
    pr[j] = mi_prob_exp((zz + ui) + vb[j]);                         // sinkhorn.py:145,206
  }
}

// The same P, four rows per wave with every load issued before the first use (round 4; m <= 1024): the loop above is nine
// dependent round trips per row.  A lane owns columns lane, lane + 64, ...: loads and stores of 256 contiguous bytes per
// instruction, v loaded once per wave for its rows.  The same expressions: the same P bit for bit.
template <int Q>
__global__ __launch_bounds__(256) void sk_exp_rows_kernel(const float *__restrict__ z, int n, int m, int pitch, float dust,
                                                          const float *__restrict__ u, const float *__restrict__ v,
                                                          float *__restrict__ p) {
  constexpr int RW = 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y, nbands = (int)gridDim.x - 1;
  const float *vb = v + (size_t)b * (m + 1);
  if ((int)blockIdx.x == nbands) {                   // the dustbin row
    const float un = u[(size_t)b * (n + 1) + n];
    float *pr = p + ((size_t)b * (n + 1) + n) * (size_t)(m + 1);
    for (int j = threadIdx.x; j <= m; j += 256) pr[j] = mi_prob_exp((dust + un) + vb[j]);
    return;
  }
  const int row0 = ((int)blockIdx.x * 4 + wave) * RW;
  if (row0 >= n) return;
  float zz[RW][Q], ui[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int ic = min(row0 + r, n - 1);
    ui[r] = u[(size_t)b * (n + 1) + ic];
    const float *zr = z + ((size_t)b * n + ic) * pitch;
#pragma unroll
    for (int q = 0; q < Q; ++q) zz[r][q] = zr[min(q * 64 + lane, m - 1)];
  }
  float vv[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) vv[q] = vb[min(q * 64 + lane, m - 1)];
  const float vd = vb[m];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    if (row0 + r >= n) break;                        // wave-uniform
    float *pr = p + ((size_t)b * (n + 1) + row0 + r) * (size_t)(m + 1);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = q * 64 + lane;
      const float val = mi_prob_exp((zz[r][q] + ui[r]) + vv[q]);     // sinkhorn.py:145,206
      if (j < m) pr[j] = val;
    }
    if (lane == 0) pr[m] = mi_prob_exp((dust + ui[r]) + vd);         // the dustbin column
  }
}

void launch_exp(const float *z, int n, int m, int pitch, float dust, const float *u, const float *v, float *p, int batch,
                hipStream_t s) {
  if (m > 1024 || MI_HOOK(sinkhorn_exp_rows, 1) == 0)
    hipLaunchKernelGGL(sk_exp_kernel, dim3(ceil_div(n + 1, 4), batch), dim3(256), 0, s, z, n, m, pitch, dust, u, v, p);
  else if (m <= 512)
    hipLaunchKernelGGL(sk_exp_rows_kernel<8>, dim3(ceil_div(n, 16) + 1, batch), dim3(256), 0, s, z, n, m, pitch, dust, u, v, p);
  else
    hipLaunchKernelGGL(sk_exp_rows_kernel<16>, dim3(ceil_div(n, 16) + 1, batch), dim3(256), 0, s, z, n, m, pitch, dust, u, v, p);
}

// ---- fused iteration, probability form: ONE exp per matrix element per iteration ----------------
// After the row pass the row-normalised entries are already known:
//     P_ij = exp(Z_ij + u_i + v_j) = e_ij / s_i,  e_ij = exp(Z_ij + v_j - max_i),  s_i = sum_j e_ij
// (all in [0,1]), and the column update of sinkhorn.py:141 is
//     v_j <- log nu_j - LSE_i(Z_ij + u_i) = v_j + log nu_j - log(sum_i P_ij).
// So the column pass needs no second exponential: a band just adds up its P_ij per column
// (plain fp32 sums of numbers in [0,1]) and the combine kernel takes one log per column.  The
// dustbin ROW's term is kept in the log domain (B_j = dust + u_n + v_j) and merged with a
// log-sum-exp, so a column whose regular entries all underflow still gets the exact dustbin
// answer.  Per element: add, max, sub, mul, v_exp, add (row) + mul, add (column) instead of
// two full exp chains -- the loop is VALU-bound on gfx950, so this is what sets its speed.
// Partials: one float per column per band (band nb = the dustbin row's B_j).
template <int E4, int RW, int NW>
__global__ __launch_bounds__(64 * NW) void sk_band_p_kernel(const float *__restrict__ z, int n, int m, int pitch,
                                                            float dust, const float *__restrict__ v,
                                                            float *__restrict__ u, float *__restrict__ part,
                                                            float log_m, int v_is_zero) {
  constexpr int BAND = NW * RW;
  constexpr int NT = 64 * NW;
  constexpr int NC = 256 * E4;
  __shared__ float red[NW][NC + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, band = blockIdx.x, nb = gridDim.x - 1;
  const float *vb = v + (size_t)b * (m + 1);
  float *pb = part + ((size_t)b * (nb + 1) + band) * (size_t)(m + 1);
  const float vd = v_is_zero ? 0.0f : vb[m];

  if (band == nb) {
    // dustbin row: u_n = log m - LSE_j(dust + v_j); its log-probabilities B_j = dust + u_n + v_j
    float mx = dust + vd;
    for (int j = threadIdx.x; j < m; j += NT) mx = fmaxf(mx, dust + (v_is_zero ? 0.0f : vb[j]));
    mx = wave_max_dpp(mx);
    if (lane == 0) red[wave][0] = mx;
    __syncthreads();
    mx = red[0][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w][0]);
    float s = 0.0f;
    for (int j = threadIdx.x; j < m; j += NT) s += expf((dust + (v_is_zero ? 0.0f : vb[j])) - mx);
    s = wave_sum_dpp(s);
    if (lane == 0) red[wave][1] = s;
    __syncthreads();
    s = red[0][1];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += red[w][1];
    s += expf((dust + vd) - mx);
    const float un = log_m - (logf(s) + mx);
    if (threadIdx.x == 0) u[(size_t)b * (n + 1) + n] = un;
    for (int j = threadIdx.x; j <= m; j += NT) pb[j] = (dust + un) + (v_is_zero ? 0.0f : vb[j]);
    return;
  }

  float vv[E4][4];
#pragma unroll
  for (int e = 0; e < E4; ++e)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = e * 256 + lane * 4 + q;
      vv[e][q] = (j < m && !v_is_zero) ? vb[j] : 0.0f;
    }
  const float xd = dust + vd;
  const int row0 = band * BAND + wave * RW;

  float x[RW][E4][4];            // Z_ij + v_j, then e_ij in place
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int i = row0 + r;
    const float *src = z + ((size_t)b * n + (i < n ? i : 0)) * pitch;
#pragma unroll
    for (int e = 0; e < E4; ++e) {
      const int j = e * 256 + lane * 4;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < n && j < m) q = *reinterpret_cast<const float4 *>(src + j);
      // outside the matrix (row padding, rows past n): -inf, i.e. no contribution anywhere
      x[r][e][0] = (i < n && j + 0 < m) ? q.x + vv[e][0] : -INFINITY;
      x[r][e][1] = (i < n && j + 1 < m) ? q.y + vv[e][1] : -INFINITY;
      x[r][e][2] = (i < n && j + 2 < m) ? q.z + vv[e][2] : -INFINITY;
      x[r][e][3] = (i < n && j + 3 < m) ? q.w + vv[e][3] : -INFINITY;
    }
  }

  float colsum[E4][4];
#pragma unroll
  for (int e = 0; e < E4; ++e)
#pragma unroll
    for (int q = 0; q < 4; ++q) colsum[e][q] = 0.0f;
  float dustcol = 0.0f;          // sum of P_i,dustbin over this wave's rows (wave-uniform)
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    float mx = xd;
#pragma unroll
    for (int e = 0; e < E4; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) mx = fmaxf(mx, x[r][e][q]);
    mx = wave_max_dpp(mx);
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < E4; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        x[r][e][q] = sk_exp(x[r][e][q] - mx);        // e_ij; exp(-inf) = 0 outside the matrix
        s += x[r][e][q];
      }
    const float ed = expf(xd - mx);                  // dustbin column entry of this row
    s = wave_sum_dpp(s) + ed;
    const float inv_s = 1.0f / s;
    const bool live = row0 + r < n;
    if (lane == 0 && live) u[(size_t)b * (n + 1) + row0 + r] = 0.0f - (logf(s) + mx);   // sinkhorn.py:139
    const float wgt = live ? inv_s : 0.0f;
#pragma unroll
    for (int e = 0; e < E4; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) colsum[e][q] += x[r][e][q] * wgt;                     // P_ij
    dustcol += ed * wgt;
  }
#pragma unroll
  for (int e = 0; e < E4; ++e)
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][e * 256 + lane * 4 + q] = colsum[e][q];
  if (lane == 0) red[wave][NC] = dustcol;
  __syncthreads();
  for (int c = threadIdx.x; c <= NC; c += NT) {
    const int j = (c == NC) ? m : c;
    if (c < NC && j >= m) continue;
    float t = red[0][c];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += red[w][c];
    pb[j] = t;
  }
}

// v_j <- v_j + log nu_j - log(sum of the bands' P_ij + exp(B_j))   (B_j: dustbin row, log domain)
__global__ __launch_bounds__(256) void sk_vcombine_p_kernel(const float *__restrict__ part, int m, int nparts,
                                                            float *__restrict__ v, float log_n, int v_is_zero) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j > m) return;
  const float *p = part + (size_t)b * nparts * (size_t)(m + 1) + j;
  float s = 0.0f;
  for (int k = 0; k < nparts - 1; ++k) s += p[(size_t)k * (m + 1)];
  const float bj = p[(size_t)(nparts - 1) * (m + 1)];
  // log(s + exp(bj)) as a two-term log-sum-exp; s == 0 (everything underflowed) leaves bj
  const float a = s > 0.0f ? logf(s) : -INFINITY;
  const float hi = fmaxf(a, bj), lo = fminf(a, bj);
  const float lse = hi + log1pf(expf(lo - hi));
  const float vold = v_is_zero ? 0.0f : v[(size_t)b * (m + 1) + j];
  v[(size_t)b * (m + 1) + j] = (vold + ((j == m) ? log_n : 0.0f)) - lse;
}

// ---- probability-form band kernel, lean instruction stream (the default) -------------------------
// Same arithmetic plan as sk_band_p_kernel.  The loop is bound by vector-instruction ISSUE on
// gfx950 (4 cycles per wave instruction, 8 for v_exp_f32), so what matters is the instruction
// count per matrix element:
//   * v lives in an aligned, padded copy (vp: -inf in the padding columns, the dustbin column at
//     index NC) so the per-lane column data are 16-byte loads and the matrix edge needs no selects
//     (x = z + -inf = -inf); rows past n run on row n-1 and get weight 0;
//   * the shift and the 2^x scaling are one fma: e = exp2(fma(x, log2e, -mx*log2e)); u is taken
//     from the same shift, so the row is exactly normalised by what was summed;
//   * 1/s and log s by v_rcp_f32 / v_log_f32 (1 ulp) instead of the IEEE expansions.
// Per element: add, max, fma, v_exp, add, fma.
constexpr float SK_L2E = 1.4426950408889634f, SK_LN2 = 0.6931471805599453f;

template <int E4, int RW, int NW>
__global__ __launch_bounds__(64 * NW) void sk_band_p2_kernel(const float *__restrict__ z, int n, int m, int pitch,
                                                             float dust, const float *__restrict__ vp, int vpitch,
                                                             float *__restrict__ u, float *__restrict__ part,
                                                             float log_m, int v_is_zero) {
  constexpr int BAND = NW * RW;
  constexpr int NT = 64 * NW;
  constexpr int NC = 256 * E4;
  __shared__ float red[NW][NC + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, band = blockIdx.x, nb = gridDim.x - 1;
  const float *vb = vp + (size_t)b * vpitch;
  float *pb = part + ((size_t)b * (nb + 1) + band) * (size_t)(m + 1);
  const float vd = v_is_zero ? 0.0f : vb[NC];

  if (band == nb) {
    // dustbin row: u_n = log m - LSE_j(dust + v_j); its log-probabilities B_j = dust + u_n + v_j
    float mx = dust + vd;
    for (int j = threadIdx.x; j < m; j += NT) mx = fmaxf(mx, dust + (v_is_zero ? 0.0f : vb[j]));
    mx = wave_max_dpp(mx);
    if (lane == 0) red[wave][0] = mx;
    __syncthreads();
    mx = red[0][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w][0]);
    float s = 0.0f;
    for (int j = threadIdx.x; j < m; j += NT) s += expf((dust + (v_is_zero ? 0.0f : vb[j])) - mx);
    s = wave_sum_dpp(s);
    if (lane == 0) red[wave][1] = s;
    __syncthreads();
    s = red[0][1];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += red[w][1];
    s += expf((dust + vd) - mx);
    const float un = log_m - (logf(s) + mx);
    if (threadIdx.x == 0) u[(size_t)b * (n + 1) + n] = un;
    for (int j = threadIdx.x; j < m; j += NT) pb[j] = (dust + un) + (v_is_zero ? 0.0f : vb[j]);
    if (threadIdx.x == 0) pb[m] = (dust + un) + vd;
    return;
  }

  float4 vv[E4];
#pragma unroll
  for (int e = 0; e < E4; ++e) {
    const int j = e * 256 + lane * 4;
    if (v_is_zero) {
      vv[e] = make_float4(j + 0 < m ? 0.0f : -INFINITY, j + 1 < m ? 0.0f : -INFINITY, j + 2 < m ? 0.0f : -INFINITY,
                          j + 3 < m ? 0.0f : -INFINITY);
    } else {
      vv[e] = *reinterpret_cast<const float4 *>(vb + j);
    }
  }
  const float xd = dust + vd;
  const int row0 = band * BAND + wave * RW;

  float4 x[RW][E4];              // Z_ij, then Z_ij + v_j, then e_ij in place
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int i = min(row0 + r, n - 1);
    const float *src = z + ((size_t)b * n + i) * pitch;
#pragma unroll
    for (int e = 0; e < E4; ++e) {
      const int j = e * 256 + lane * 4;
      x[r][e] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < m) x[r][e] = *reinterpret_cast<const float4 *>(src + j);      // j + 3 < pitch (pitch % 4 == 0)
    }
  }
  if (m & 3) {
    // the float4 that straddles column m holds row padding (arbitrary bits): clear it
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int e = 0; e < E4; ++e) {
        const int j = e * 256 + lane * 4;
        if (j + 1 >= m) x[r][e].y = 0.0f;
        if (j + 2 >= m) x[r][e].z = 0.0f;
        if (j + 3 >= m) x[r][e].w = 0.0f;
      }
  }

  float4 colsum[E4];
#pragma unroll
  for (int e = 0; e < E4; ++e) colsum[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  float dustcol = 0.0f;          // sum of P_i,dustbin over this wave's rows (wave-uniform)
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    float mx = xd;
#pragma unroll
    for (int e = 0; e < E4; ++e) {
      x[r][e].x += vv[e].x;
      x[r][e].y += vv[e].y;
      x[r][e].z += vv[e].z;
      x[r][e].w += vv[e].w;
      mx = fmaxf(fmaxf(mx, x[r][e].x), fmaxf(x[r][e].y, fmaxf(x[r][e].z, x[r][e].w)));
    }
    mx = wave_max_dpp(mx);
    const float nm = -(mx * SK_L2E);
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < E4; ++e) {
      x[r][e].x = __builtin_amdgcn_exp2f(__builtin_fmaf(x[r][e].x, SK_L2E, nm));   // exp2(-inf) = 0 outside
      x[r][e].y = __builtin_amdgcn_exp2f(__builtin_fmaf(x[r][e].y, SK_L2E, nm));
      x[r][e].z = __builtin_amdgcn_exp2f(__builtin_fmaf(x[r][e].z, SK_L2E, nm));
      x[r][e].w = __builtin_amdgcn_exp2f(__builtin_fmaf(x[r][e].w, SK_L2E, nm));
      s += (x[r][e].x + x[r][e].y) + (x[r][e].z + x[r][e].w);
    }
    const float ed = __builtin_amdgcn_exp2f(__builtin_fmaf(xd, SK_L2E, nm));       // dustbin column entry
    s = wave_sum_dpp(s) + ed;
    const bool live = row0 + r < n;
    if (lane == 0 && live)                                                          // sinkhorn.py:139
      u[(size_t)b * (n + 1) + row0 + r] = (nm - __builtin_amdgcn_logf(s)) * SK_LN2;
    const float wgt = live ? __builtin_amdgcn_rcpf(s) : 0.0f;
#pragma unroll
    for (int e = 0; e < E4; ++e) {
      colsum[e].x = __builtin_fmaf(x[r][e].x, wgt, colsum[e].x);                    // += P_ij
      colsum[e].y = __builtin_fmaf(x[r][e].y, wgt, colsum[e].y);
      colsum[e].z = __builtin_fmaf(x[r][e].z, wgt, colsum[e].z);
      colsum[e].w = __builtin_fmaf(x[r][e].w, wgt, colsum[e].w);
    }
    dustcol = __builtin_fmaf(ed, wgt, dustcol);
  }
#pragma unroll
  for (int e = 0; e < E4; ++e) *reinterpret_cast<float4 *>(&red[wave][e * 256 + lane * 4]) = colsum[e];
  if (lane == 0) red[wave][NC] = dustcol;
  __syncthreads();
  for (int c = threadIdx.x; c <= NC; c += NT) {
    const int j = (c == NC) ? m : c;
    if (c < NC && j >= m) continue;
    float t = red[0][c];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += red[w][c];
    pb[j] = t;
  }
}

// v_j <- v_j + log nu_j - log(sum of the bands' P_ij + exp(B_j)); writes the caller's v (pitch m+1)
// and the padded copy the band kernel reads (columns m..NC-1 = -inf, dustbin column at NC)
__global__ __launch_bounds__(256) void sk_vcombine_p2_kernel(const float *__restrict__ part, int m, int nparts,
                                                             float *__restrict__ v, float *__restrict__ vp,
                                                             int vpitch, int nc, float log_n, int v_is_zero) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c > nc) return;
  float *vpb = vp + (size_t)b * vpitch;
  if (c >= m && c < nc) {
    vpb[c] = -INFINITY;
    return;
  }
  const int j = (c == nc) ? m : c;
  const float *p = part + (size_t)b * nparts * (size_t)(m + 1) + j;
  // band order, loads issued eight at a time (one L2 round trip instead of nparts)
  float s = 0.0f;
  int k = 0;
  for (; k + 8 <= nparts - 1; k += 8) {
    float t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = p[(size_t)(k + q) * (m + 1)];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += t[q];
  }
  for (; k < nparts - 1; ++k) s += p[(size_t)k * (m + 1)];
  const float bj = p[(size_t)(nparts - 1) * (m + 1)];
  // log(s + exp(bj)) as a two-term log-sum-exp; s == 0 (everything underflowed) leaves bj
  const float a = s > 0.0f ? logf(s) : -INFINITY;
  const float hi = fmaxf(a, bj), lo = fminf(a, bj);
  const float lse = hi + log1pf(expf(lo - hi));
  const float vold = v_is_zero ? 0.0f : vpb[c];
  const float vnew = (vold + ((j == m) ? log_n : 0.0f)) - lse;
  vpb[c] = vnew;
  v[(size_t)b * (m + 1) + j] = vnew;
}

// ---- fused iteration: Z is read ONCE per iteration ----------------------------------------------
// A workgroup owns a band of 4*RW rows.  Each wave keeps RW whole rows of Z in registers
// (RW * E4 float4 per lane, all loads issued up front), computes u for them (row pass), and
// -- with u fresh in registers -- the band's contribution to every column's log-sum-exp:
// a (max, sum) pair per column, merged over the 4 waves in LDS and written as one float2 per
// column.  A tiny second kernel merges the bands' partials into v.  Per iteration Z moves
// 4 B/element once instead of twice; partials are 8 B per column per band (~6 % extra).
// Band index nb = number of row bands carries the dustbin ROW (x = dust + u_n, computed
// without Z); column index m of every partial carries the dustbin COLUMN (x = dust + u_i).
template <int E4, int RW, int NW>
__global__ __launch_bounds__(64 * NW) void sk_band_kernel(const float *__restrict__ z, int n, int m, int pitch,
                                                      float dust, const float *__restrict__ v,
                                                      float *__restrict__ u, float2 *__restrict__ part,
                                                      float log_m, int v_is_zero) {
  constexpr int BAND = NW * RW;   // NW waves x RW rows each
  constexpr int NT = 64 * NW;
  __shared__ float red_m[NW][256 * E4 + 1];
  __shared__ float red_s[NW][256 * E4 + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, band = blockIdx.x, nb = gridDim.x - 1;
  const float *vb = v + (size_t)b * (m + 1);
  float2 *pb = part + ((size_t)b * (nb + 1) + band) * (size_t)(m + 1);
  const float vd = v_is_zero ? 0.0f : vb[m];

  if (band == nb) {
    // dustbin row: u_n = log m - LSE_j(dust + v_j), then its term for every column
    float mx = dust + vd;
    for (int j = threadIdx.x; j < m; j += NT) mx = fmaxf(mx, dust + (v_is_zero ? 0.0f : vb[j]));
    mx = wave_max_dpp(mx);
    if (lane == 0) red_m[wave][0] = mx;
    __syncthreads();
    mx = red_m[0][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red_m[w][0]);
    float s = 0.0f;
    for (int j = threadIdx.x; j < m; j += NT) s += expf((dust + (v_is_zero ? 0.0f : vb[j])) - mx);
    s = wave_sum_dpp(s);
    if (lane == 0) red_s[wave][0] = s;
    __syncthreads();
    s = red_s[0][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += red_s[w][0];
    s += expf((dust + vd) - mx);
    const float un = log_m - (logf(s) + mx);
    if (threadIdx.x == 0) u[(size_t)b * (n + 1) + n] = un;
    for (int j = threadIdx.x; j <= m; j += NT) pb[j] = make_float2(dust + un, 1.0f);
    return;
  }

  float vv[E4][4];
#pragma unroll
  for (int e = 0; e < E4; ++e)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = e * 256 + lane * 4 + q;
      vv[e][q] = (j < m && !v_is_zero) ? vb[j] : 0.0f;
    }
  const float xd = dust + vd;
  const int row0 = band * BAND + wave * RW;

  float zr[RW][E4][4];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int i = row0 + r;
    const float *src = z + ((size_t)b * n + (i < n ? i : 0)) * pitch;
#pragma unroll
    for (int e = 0; e < E4; ++e) {
      const int j = e * 256 + lane * 4;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < n && j < m) q = *reinterpret_cast<const float4 *>(src + j);
      // outside the matrix (row padding, rows past n): -inf, i.e. no contribution anywhere
      zr[r][e][0] = (i < n && j + 0 < m) ? q.x : -INFINITY;
      zr[r][e][1] = (i < n && j + 1 < m) ? q.y : -INFINITY;
      zr[r][e][2] = (i < n && j + 2 < m) ? q.z : -INFINITY;
      zr[r][e][3] = (i < n && j + 3 < m) ? q.w : -INFINITY;
    }
  }

  // row pass (sinkhorn.py:139): u_i = log mu_i - LSE_j(Z_ij + v_j), dustbin column included
  float ur[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    float mx = xd;
#pragma unroll
    for (int e = 0; e < E4; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) mx = fmaxf(mx, zr[r][e][q] + vv[e][q]);
    mx = wave_max_dpp(mx);
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < E4; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) s += sk_exp((zr[r][e][q] + vv[e][q]) - mx);
    s = wave_sum_dpp(s) + expf(xd - mx);
    ur[r] = 0.0f - (logf(s) + mx);
    if (lane == 0 && row0 + r < n) u[(size_t)b * (n + 1) + row0 + r] = ur[r];
  }

  // column partials over this wave's RW rows (sinkhorn.py:141: Z_ij + u_i)
#pragma unroll
  for (int e = 0; e < E4; ++e)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float cm = -INFINITY;
#pragma unroll
      for (int r = 0; r < RW; ++r) cm = fmaxf(cm, zr[r][e][q] + ur[r]);
      float cs = 0.0f;
      if (cm > -INFINITY) {
#pragma unroll
        for (int r = 0; r < RW; ++r) cs += sk_exp((zr[r][e][q] + ur[r]) - cm);
      }
      red_m[wave][e * 256 + lane * 4 + q] = cm;
      red_s[wave][e * 256 + lane * 4 + q] = cs;
    }
  {
    // dustbin column: x_i = dust + u_i over this wave's valid rows (uniform across lanes)
    float cm = -INFINITY;
#pragma unroll
    for (int r = 0; r < RW; ++r)
      if (row0 + r < n) cm = fmaxf(cm, dust + ur[r]);
    float cs = 0.0f;
#pragma unroll
    for (int r = 0; r < RW; ++r)
      if (row0 + r < n) cs += expf((dust + ur[r]) - cm);
    if (lane == 0) { red_m[wave][256 * E4] = cm; red_s[wave][256 * E4] = cs; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c <= 256 * E4; c += NT) {
    const int j = (c == 256 * E4) ? m : c;
    if (c < 256 * E4 && j >= m) continue;
    float fm = red_m[0][c];
#pragma unroll
    for (int w = 1; w < NW; ++w) fm = fmaxf(fm, red_m[w][c]);
    float fs = 0.0f;
    if (fm > -INFINITY) {
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if (red_m[w][c] > -INFINITY) fs += red_s[w][c] * expf(red_m[w][c] - fm);
    }
    pb[j] = make_float2(fm, fs);
  }
}

// v_j = log nu_j - LSE over all bands' partials (including the dustbin-row band)
__global__ __launch_bounds__(256) void sk_vcombine_kernel(const float2 *__restrict__ part, int m, int nparts,
                                                          float *__restrict__ v, float log_n) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j > m) return;
  const float2 *p = part + (size_t)b * nparts * (size_t)(m + 1) + j;
  float mx = -INFINITY;
  for (int k = 0; k < nparts; ++k) mx = fmaxf(mx, p[(size_t)k * (m + 1)].x);
  float s = 0.0f;
  for (int k = 0; k < nparts; ++k) {
    const float2 q = p[(size_t)k * (m + 1)];
    if (q.x > -INFINITY) s += q.y * expf(q.x - mx);
  }
  v[(size_t)b * (m + 1) + j] = ((j == m) ? log_n : 0.0f) - (logf(s) + mx);
}

// >= 64 pairs (round 4): two half batches on two streams, scheduled by mi_sinkhorn_dots' per-stream tuner under a shape
// key of its own -- an iteration is a big row kernel and a tiny column kernel that depend on each other, and one half's
// row kernel fills the other half's gap (sinkhorn_dots.hip, launch_dots).  The halves are independent problems: same
// duals.  A caller opts out per stream with mi_sinkhorn_dots_set_schedule(stream, MI_SCHEDULE_UNSPLIT).
constexpr int SK_F32_SHAPE_KEY = 1 << 20;      // added to `iterations` in the tuner's shape: never a dots shape
template <int E4, int RW, int NW>
int launch_fused(const float *z, int batch, int n, int m, int pitch, float dust, int iterations, float *u,
                 float *v, float2 *part, float *vp, float log_m, float log_n, hipStream_t s) {
  const int nb = ceil_div(n, NW * RW);
  constexpr int NC = 256 * E4;
  const int band_form = MI_HOOK(sinkhorn_log_partials, 0);
  MiFork fk;
  const int fe = mi_fork_begin(s, batch, n, m, iterations + SK_F32_SHAPE_KEY, &fk);
  if (fe != MI_OK) return fe;
  for (int it = 0; it < iterations; ++it) {     // iteration by iteration, so that no stream runs far ahead of the other
    const int vz = it == 0 ? 1 : 0;
    for (int q = 0; q < fk.parts; ++q) {
      const int b0 = (int)((long long)batch * q / fk.parts), nbatch = (int)((long long)batch * (q + 1) / fk.parts) - b0;
      hipStream_t st = fk.stream[q];
      const float *z0 = z + (size_t)b0 * n * pitch;
      float *u0 = u + (size_t)b0 * (n + 1), *v0 = v + (size_t)b0 * (m + 1), *vp0 = vp + (size_t)b0 * (NC + 4);
      float2 *part0 = part + (size_t)b0 * (nb + 1) * (size_t)(m + 1);
      float *pf = reinterpret_cast<float *>(part) + (size_t)b0 * (nb + 1) * (size_t)(m + 1);
      if (band_form == 0) {
        hipLaunchKernelGGL((sk_band_p2_kernel<E4, RW, NW>), dim3(nb + 1, nbatch), dim3(64 * NW), 0, st, z0, n, m, pitch,
                           dust, vp0, NC + 4, u0, pf, log_m, vz);
        hipLaunchKernelGGL(sk_vcombine_p2_kernel, dim3(ceil_div(NC + 1, 256), nbatch), dim3(256), 0, st, pf, m, nb + 1, v0,
                           vp0, NC + 4, NC, log_n, vz);
      } else if (band_form == 1) {
        hipLaunchKernelGGL((sk_band_kernel<E4, RW, NW>), dim3(nb + 1, nbatch), dim3(64 * NW), 0, st, z0, n, m, pitch, dust,
                           v0, u0, part0, log_m, vz);
        hipLaunchKernelGGL(sk_vcombine_kernel, dim3(ceil_div(m + 1, 256), nbatch), dim3(256), 0, st, part0, m, nb + 1,
                           v0, log_n);
      } else {
        hipLaunchKernelGGL((sk_band_p_kernel<E4, RW, NW>), dim3(nb + 1, nbatch), dim3(64 * NW), 0, st, z0, n, m, pitch,
                           dust, v0, u0, pf, log_m, vz);
        hipLaunchKernelGGL(sk_vcombine_p_kernel, dim3(ceil_div(m + 1, 256), nbatch), dim3(256), 0, st, pf, m, nb + 1, v0,
                           log_n, vz);
      }
    }
  }
  return mi_fork_end(s, &fk);
}

int fused_rows_per_band(int m) {
  const int e4 = ceil_div(m, 256);
  if (e4 <= 2) return 32;
  if (e4 <= 4) return 16;
  return 0;  // not supported by the fused kernels
}

// band partials (float2 per column per band), then the padded copy of v (256*E4 + 4 floats per matrix)
size_t partials_bytes(int batch, int n, int m, int band) {
  const size_t b = (size_t)batch * (size_t)(ceil_div(n, band) + 1) * (size_t)(m + 1) * sizeof(float2);
  return (b + 15) & ~(size_t)15;
}

}  // namespace

extern "C" size_t mi_sinkhorn_workspace_bytes(int batch, int n, int m) {
  const int band = fused_rows_per_band(m);
  if (batch <= 0 || n <= 0 || m <= 0 || band == 0) return 0;
  return partials_bytes(batch, n, m, band) + (size_t)batch * (size_t)(256 * ceil_div(m, 256) + 4) * sizeof(float);
}

extern "C" int mi_sinkhorn(const float *z, int batch, int n, int m, int pitch, float dustbin_logscore,
                           int iterations, float *u, float *v, float *p, void *workspace,
                           size_t workspace_bytes, mi_stream_t stream) {
  MI_ENTER();
  if (!z || !u || !v) return MI_E_NULL;
  if (batch <= 0 || n <= 0 || m <= 0 || batch > 65535) return MI_E_SHAPE;
  if (pitch < m || pitch % 4 != 0 || ((uintptr_t)z % 16) != 0) return MI_E_ALIGN;
  if (iterations <= 0) return MI_E_PARAM;
  hipStream_t s = (hipStream_t)stream;
  const float log_m = logf((float)m), log_n = logf((float)n);      // sinkhorn.py:197-198
  const size_t need = mi_sinkhorn_workspace_bytes(batch, n, m);
  if (need > 0 && workspace && ((uintptr_t)workspace % 16) == 0) {
    if (workspace_bytes < need) return MI_E_CAPACITY;
    float2 *part = reinterpret_cast<float2 *>(workspace);
    float *vp = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) +
                                          partials_bytes(batch, n, m, fused_rows_per_band(m)));
    const int e4 = ceil_div(m, 256);
    int fr;
    if (e4 == 1) fr = launch_fused<1, 4, 8>(z, batch, n, m, pitch, dustbin_logscore, iterations, u, v, part, vp, log_m, log_n, s);
    else if (e4 == 2) fr = launch_fused<2, 4, 8>(z, batch, n, m, pitch, dustbin_logscore, iterations, u, v, part, vp, log_m, log_n, s);
    else fr = launch_fused<4, 2, 8>(z, batch, n, m, pitch, dustbin_logscore, iterations, u, v, part, vp, log_m, log_n, s);
    if (fr != MI_OK) return fr;
    if (p) {
      launch_exp(z, n, m, pitch, dustbin_logscore, u, v, p, batch, s);
    }
    return mi_launch_status();
  }
  // two-pass form (no workspace, or m > 1024)
  const dim3 rgrid(ceil_div(n + 1, 4 * ROWS_PER_WAVE), batch);
  const int e4 = ceil_div(m, 256);
  // widest column vector that divides m (so a lane never straddles the matrix edge)
  const int vec = (m % 4 == 0) ? 4 : ((m % 2 == 0) ? 2 : 1);
  const dim3 cgrid(ceil_div(m, 64 * vec) + 1, batch);
  for (int it = 0; it < iterations; ++it) {
    const int vz = (it == 0) ? 1 : 0;
    switch (e4) {
      case 1: hipLaunchKernelGGL(sk_row_kernel<1>, rgrid, dim3(256), 0, s, z, n, m, pitch, dustbin_logscore, v, u, log_m, vz); break;
      case 2: hipLaunchKernelGGL(sk_row_kernel<2>, rgrid, dim3(256), 0, s, z, n, m, pitch, dustbin_logscore, v, u, log_m, vz); break;
      case 3: case 4: hipLaunchKernelGGL(sk_row_kernel<4>, rgrid, dim3(256), 0, s, z, n, m, pitch, dustbin_logscore, v, u, log_m, vz); break;
      default: hipLaunchKernelGGL(sk_row_generic_kernel, rgrid, dim3(256), 0, s, z, n, m, pitch, dustbin_logscore, v, u, log_m, vz); break;
    }
    if (vec == 4) hipLaunchKernelGGL(sk_col_kernel<4>, cgrid, dim3(256), 0, s, z, n, m, pitch, dustbin_logscore, u, v, log_n);
    else if (vec == 2) hipLaunchKernelGGL(sk_col_kernel<2>, cgrid, dim3(256), 0, s, z, n, m, pitch, dustbin_logscore, u, v, log_n);
    else hipLaunchKernelGGL(sk_col_kernel<1>, cgrid, dim3(256), 0, s, z, n, m, pitch, dustbin_logscore, u, v, log_n);
  }
  if (p) {
    launch_exp(z, n, m, pitch, dustbin_logscore, u, v, p, batch, s);
  }
  return mi_launch_status();
}
