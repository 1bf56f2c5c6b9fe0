// K10: essential-matrix head on the Sinkhorn assignment matrix (weighted 8-point algorithm).
// Semantics: reference pytorch_model/geometry/essential_matrix_estimator.py:302-431 (forward),
// :150-173 (_min_eigvec9), :175-248 (_project_onto_E_manifold), :250-300 (_hartley_normalization) and
// the composites' _estimate_essential_matrix with validity masking
// (feature_detection/shi_tomasi_angle_sparse_bad_sinkhorn_essential_matrix.py:184-271):
//   core = P[:n,:m] * valid1 * valid2;  weight_ij = core_ij where it is among the top_k of its row AND
//   of its column (k-th largest with multiplicity, >=) AND > 0.01, else 0;  Hartley-normalise both
//   point sets with the row / column sums of the weights;  M = sum_ij w_ij (f1_i (x) f2_j)(f1_i (x) f2_j)^T
//   through the Kronecker factorisation F1^T (W F2);  minimum eigenvector by 30 steps of shifted power
//   iteration from the all-ones vector;  denormalise;  project onto singular values (s, s, 0).
// One 1024-thread workgroup per pair: the matrix (1 MB at K = 512) is streamed four times from L2/HBM
// (row thresholds, column thresholds, weight sums, normal equations), everything else lives in LDS;
// all reductions have a fixed order, so the result is deterministic.  fp32 throughout like the
// reference; its GEMM / sum orders are BLAS's, so parity is by tolerance (|dE| ~ 1e-5 measured).
#include "common.h"

#include <math.h>

namespace {

constexpr int EM_T = 1024;       // threads per workgroup
constexpr int EM_W = EM_T / 64;  // waves
constexpr int EM_MAXK = 8;       // top_k <= 8
constexpr int EM_MAXN = 1024;    // n, m <= 1024 (LDS-resident per-row / per-column state)

// keep the EM_MAXK largest values seen, sorted descending (multiplicity kept)
__device__ __forceinline__ void topk_insert(float (&top)[EM_MAXK], float x) {
#pragma unroll
  for (int q = 0; q < EM_MAXK; ++q) {
    const float hi = fmaxf(top[q], x), lo = fminf(top[q], x);
    top[q] = hi;
    x = lo;
  }
}

__device__ __forceinline__ float norm3(const float *v) { return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
__device__ __forceinline__ float det3(const float (*m)[3]) {
  return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
         m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}
__device__ __forceinline__ float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
__device__ __forceinline__ void matvec3(const float (*a)[3], const float *v, float *out) {
#pragma unroll
  for (int r = 0; r < 3; ++r) out[r] = (a[r][0] * v[0] + a[r][1] * v[1]) + a[r][2] * v[2];
}
__device__ __forceinline__ void unit3(float *v) {
  const float nn = norm3(v) + 1e-8f;
#pragma unroll
  for (int r = 0; r < 3; ++r) v[r] = v[r] / nn;
}
__device__ __forceinline__ void cross3(const float *a, const float *b, float *o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// weighted centroid and scale of one point set by one wave (essential_matrix_estimator.py:270-283)
__device__ __forceinline__ void hartley_wave(const float *__restrict__ pts, const float *wts, int cnt, int lane,
                                             float *out /* cx, cy, s */) {
  float sw = 0.0f, sx = 0.0f, sy = 0.0f;
  for (int i = lane; i < cnt; i += 64) {
    const float w = wts[i];
    sw += w;
    sx += w * pts[2 * i + 0];
    sy += w * pts[2 * i + 1];
  }
  const float w_sum = wave_sum_dpp(sw) + 1e-8f;
  const float cx = wave_sum_dpp(sx) / w_sum, cy = wave_sum_dpp(sy) / w_sum;
  float sd = 0.0f;
  for (int i = lane; i < cnt; i += 64) {
    const float dx = pts[2 * i + 0] - cx, dy = pts[2 * i + 1] - cy;
    sd += wts[i] * (dx * dx + dy * dy);
  }
  const float mean_dist = sqrtf(wave_sum_dpp(sd) / w_sum + 1e-8f);
  if (lane == 0) {
    out[0] = cx;
    out[1] = cy;
    out[2] = sqrtf(2.0f) / (mean_dist + 1e-8f);
  }
}

// Where the assignment matrix comes from.  EmSrcP: a materialised P (b, n+1, m+1).  EmSrcDots (round 4): P is never
// written -- every entry is rebuilt from the packed descriptors' uint16 dot products and the Sinkhorn duals with the very
// expression the solver's final pass uses (sinkhorn_dots.hip: sk_exp_dots_kernel), so both sources give the head the
// same bits; the band kernel then reads 2 instead of 4 bytes per entry and the 4-byte matrix is neither written nor
// re-read (138 + 137 MB per 128 pairs at K = 512).
struct EmSrcP {
  const float *p;
  struct Row {};
  struct Col {};
  __device__ __forceinline__ Row row(int, int, int) const { return Row(); }
  __device__ __forceinline__ Col col(int, int, int) const { return Col(); }
  __device__ __forceinline__ float at(int b, int n, int m, int i, int j, const Row &, const Col &) const {
    return p[((size_t)b * (n + 1) + i) * (size_t)(m + 1) + j];
  }
  // the same entry in two steps: the load, and what turns the loaded word into the entry (em_band_kernel issues the
  // loads of all its rows before it uses the first)
  // em_band_kernel's view: which column lane `lane` handles as its q-th (0 <= q < Q), a row's loads (issued for all
  // the wave's rows before the first is used), and what turns a loaded word into the entry
  // (the same column order as EmSrcDots: the candidates of a row are then listed, and their weights added, in the same
  // order from both sources -- E from P and E from dots + duals stay identical bit for bit)
  static __device__ __forceinline__ int column(int lane, int q) { return 2 * lane + 128 * (q >> 1) + (q & 1); }
  template <int Q> struct RawRow { float w[Q]; };
  template <int Q> __device__ __forceinline__ RawRow<Q> load_row(int b, int n, int m, int i, int lane) const {
    RawRow<Q> r;
#pragma unroll
    for (int q = 0; q < Q; ++q) r.w[q] = p[((size_t)b * (n + 1) + i) * (size_t)(m + 1) + min(column(lane, q), m - 1)];
    return r;
  }
  // the row's / column's part of the core entry (P * valid1 * valid2, :334-343) as the band kernel keeps it
  struct BandRow { float v1; };
  struct BandCol { float v2; };
  __device__ __forceinline__ BandRow band_row(int, int, int, bool valid) const { return BandRow{valid ? 1.0f : 0.0f}; }
  __device__ __forceinline__ BandCol band_col(int, int, int, bool valid) const { return BandCol{valid ? 1.0f : 0.0f}; }
  template <int Q> __device__ __forceinline__ float value(const RawRow<Q> &r, int q, const BandRow &br, const BandCol &bc) const {
    return (r.w[q] * br.v1) * bc.v2;
  }
};
struct EmSrcDots {
  const uint16_t *dots;          // (batch, n, pitch)
  int pitch;
  const float2 *row_info, *col_info;
  const float *u, *v;            // (batch, n + 1), (batch, m + 1)
  float neg_inv_eps;
  struct Row { float2 ri; float u; };
  struct Col { float2 ci; float v; };
  __device__ __forceinline__ Row row(int b, int n, int i) const { return Row{row_info[(size_t)b * n + i], u[(size_t)b * (n + 1) + i]}; }
  __device__ __forceinline__ Col col(int b, int m, int j) const { return Col{col_info[(size_t)b * m + j], v[(size_t)b * (m + 1) + j]}; }
  __device__ __forceinline__ float at(int b, int n, int, int i, int j, const Row &r, const Col &c) const {
    const float dot = (float)dots[((size_t)b * n + i) * pitch + j];
    return mi_prob_exp((mi_z_from_dot(dot, r.ri, c.ci, neg_inv_eps) + r.u) + c.v);     // sinkhorn.py:145,206
  }
  // a lane handles PAIRS of adjacent columns: one 4-byte load brings two dots (half the load instructions and half the
  // registers of the prefetched rows); pitch % 8 == 0 and a 4-byte aligned base are checked by the entry point
  static __device__ __forceinline__ int column(int lane, int q) { return 2 * lane + 128 * (q >> 1) + (q & 1); }
  template <int Q> struct RawRow { uint32_t w[Q / 2]; };
  template <int Q> __device__ __forceinline__ RawRow<Q> load_row(int b, int n, int, int i, int lane) const {
    RawRow<Q> r;
    const uint16_t *row = dots + ((size_t)b * n + i) * pitch;
#pragma unroll
    for (int h = 0; h < Q / 2; ++h) r.w[h] = *reinterpret_cast<const uint32_t *>(row + min(2 * lane + 128 * h, pitch - 2));
    return r;
  }
  // an invalid row / column has its dual at -inf: the entry is exp(-inf) = 0 = P * 0, a valid one's is P * 1 = P -- the
  // core entry (:334-343) without a validity factor per row and column in registers
  typedef Row BandRow;
  typedef Col BandCol;
  __device__ __forceinline__ BandRow band_row(int b, int n, int i, bool valid) const {
    Row r = row(b, n, i);
    r.u = valid ? r.u : -INFINITY;
    return r;
  }
  __device__ __forceinline__ BandCol band_col(int b, int m, int j, bool valid) const {
    Col c = col(b, m, j);
    c.v = valid ? c.v : -INFINITY;
    return c;
  }
  template <int Q> __device__ __forceinline__ float value(const RawRow<Q> &rr, int q, const Row &r, const Col &c) const {
    const uint32_t w = rr.w[q >> 1];
    const float dot = (float)((q & 1) ? (w >> 16) : (w & 0xffffu));
    return mi_prob_exp((mi_z_from_dot(dot, r.ri, c.ci, neg_inv_eps) + r.u) + c.v);
  }
};

// LDS state shared by the dense front, the sparse front and the solve
struct EmShared {
  float thr_row[EM_MAXN], thr_col[EM_MAXN], w1[EM_MAXN], w2[EM_MAXN], v1s[EM_MAXN], v2s[EM_MAXN];
  float f2x[EM_MAXN], f2y[EM_MAXN];
  float mpart[EM_W][81];
  float mflat[81];
  float hart[6];        // c1x, c1y, s1, c2x, c2y, s2
};

// Dense front: thresholds, weights, Hartley parameters and the 81 normal-equation sums of pair b, by one 1024-thread
// workgroup streaming P four times.  Leaves S.mflat / S.hart ready (a barrier has been passed).
template <typename SRC>
__device__ void em_dense_front(EmShared &S, const SRC src, int b, int n, int m,
                               const float *__restrict__ pts1, const float *__restrict__ pts2,
                               const uint8_t *__restrict__ valid1, const uint8_t *__restrict__ valid2, int top_k) {
  float *thr_row = S.thr_row, *thr_col = S.thr_col, *w1 = S.w1, *w2 = S.w2, *v1s = S.v1s, *v2s = S.v2s, *f2x = S.f2x, *f2y = S.f2y;
  float (*mpart)[81] = S.mpart;
  float *mflat = S.mflat, *hart = S.hart;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float *q1 = pts1 + (size_t)b * n * 2, *q2 = pts2 + (size_t)b * m * 2;

  for (int i = t; i < n; i += EM_T) v1s[i] = valid1 ? (valid1[(size_t)b * n + i] ? 1.0f : 0.0f) : 1.0f;
  for (int j = t; j < m; j += EM_T) v2s[j] = valid2 ? (valid2[(size_t)b * m + j] ? 1.0f : 0.0f) : 1.0f;
  __syncthreads();
  auto core = [&](int i, int j) { return (src.at(b, n, m, i, j, src.row(b, n, i), src.col(b, m, j)) * v1s[i]) * v2s[j]; };

  // ---- k-th largest per row (one wave per row) and per column (one thread per column)
  for (int i = wave; i < n; i += EM_W) {
    float top[EM_MAXK];
#pragma unroll
    for (int q = 0; q < EM_MAXK; ++q) top[q] = -INFINITY;
    for (int j = lane; j < m; j += 64) topk_insert(top, core(i, j));
    float kth = -INFINITY;
    for (int s = 0; s < top_k; ++s) {            // pop the wave-wide maximum top_k times
      kth = wave_max_dpp(top[0]);
      const unsigned long long owners = __ballot(top[0] == kth);
      if (lane == __ffsll((long long)owners) - 1) {
#pragma unroll
        for (int q = 0; q + 1 < EM_MAXK; ++q) top[q] = top[q + 1];
        top[EM_MAXK - 1] = -INFINITY;
      }
    }
    if (lane == 0) thr_row[i] = kth;
  }
  for (int j = t; j < m; j += EM_T) {
    float top[EM_MAXK];
#pragma unroll
    for (int q = 0; q < EM_MAXK; ++q) top[q] = -INFINITY;
    for (int i = 0; i < n; ++i) topk_insert(top, core(i, j));
    float kth = top[0];
#pragma unroll
    for (int q = 1; q < EM_MAXK; ++q) kth = (q < top_k) ? top[q] : kth;
    thr_col[j] = kth;
  }
  __syncthreads();
  auto weight = [&](int i, int j) {
    const float x = core(i, j);
    return (x >= thr_row[i] && x >= thr_col[j] && x > 0.01f) ? x : 0.0f;      // :345-358
  };

  // ---- row and column sums of the weights
  for (int i = wave; i < n; i += EM_W) {
    float s = 0.0f;
    for (int j = lane; j < m; j += 64) s += weight(i, j);
    s = wave_sum_dpp(s);
    if (lane == 0) w1[i] = s;
  }
  for (int j = t; j < m; j += EM_T) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += weight(i, j);
    w2[j] = s;
  }
  __syncthreads();
  if (wave == 0) hartley_wave(q1, w1, n, lane, hart);
  if (wave == 1) hartley_wave(q2, w2, m, lane, hart + 3);
  __syncthreads();
  for (int j = t; j < m; j += EM_T) {
    f2x[j] = (q2[2 * j + 0] - hart[3]) * hart[5];
    f2y[j] = (q2[2 * j + 1] - hart[4]) * hart[5];
  }
  __syncthreads();

  // ---- normal equations: M_flat[pr][qs] = sum_i F1[i][pr] * (sum_j w_ij F2[j][qs])   (:401-414)
  float acc0 = 0.0f, acc1 = 0.0f;                 // lane holds entries `lane` and `64 + lane` (< 81)
  for (int i = wave; i < n; i += EM_W) {
    float sxx = 0.0f, sxy = 0.0f, sx = 0.0f, syy = 0.0f, sy = 0.0f, s1 = 0.0f;
    for (int j = lane; j < m; j += 64) {
      const float w = weight(i, j);
      const float x = f2x[j], y = f2y[j];
      sxx += w * (x * x);
      sxy += w * (x * y);
      sx += w * x;
      syy += w * (y * y);
      sy += w * y;
      s1 += w;
    }
    sxx = wave_sum_dpp(sxx); sxy = wave_sum_dpp(sxy); sx = wave_sum_dpp(sx);
    syy = wave_sum_dpp(syy); sy = wave_sum_dpp(sy); s1 = wave_sum_dpp(s1);
    const float wf2[9] = {sxx, sxy, sx, sxy, syy, sy, sx, sy, s1};
    const float f1[3] = {(q1[2 * i + 0] - hart[0]) * hart[2], (q1[2 * i + 1] - hart[1]) * hart[2], 1.0f};
    {
      const int e = lane, pr = e / 9, qs = e - pr * 9;
      float wsel = 0.0f;
#pragma unroll
      for (int c = 0; c < 9; ++c) wsel = (qs == c) ? wf2[c] : wsel;
      acc0 += (f1[pr / 3] * f1[pr % 3]) * wsel;
    }
    if (lane < 81 - 64) {
      const int e = 64 + lane, pr = e / 9, qs = e - pr * 9;
      float wsel = 0.0f;
#pragma unroll
      for (int c = 0; c < 9; ++c) wsel = (qs == c) ? wf2[c] : wsel;
      acc1 += (f1[pr / 3] * f1[pr % 3]) * wsel;
    }
  }
  mpart[wave][lane] = acc0;
  if (lane < 81 - 64) mpart[wave][64 + lane] = acc1;
  __syncthreads();
  if (t < 81) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < EM_W; ++w) s += mpart[w][t];
    mflat[t] = s;
  }
  __syncthreads();
}

// The tail, by ONE WAVE (round 4; it was one thread: 30 power iterations of a 9x9 matrix-vector product, a norm and nine
// divisions in one lane's dependent chain -- about half of em_sparse_kernel's 51 us).  Lane a < 9 holds row a of the
// shifted matrix and forms element a of the product; the nine elements go to every lane through v_readlane, every lane
// adds their squares in the serial order, lane a divides its element, and the new vector is read back the same way:
// per value the same operations in the same order as the one-thread form, a quarter of the instructions per iteration.
// The two 3x3 power iterations of the manifold projection run side by side in lanes 0 and 1.  Everything else is
// computed redundantly by all lanes; lane 0 writes E.
__device__ __forceinline__ float lane_value(float x, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src_lane));
}
__device__ void em_solve(const EmShared &S, int b, int n_iter, int n_iter_manifold, float *__restrict__ e_out) {
  const float *mflat = S.mflat, *hart = S.hart;
  const int lane = threadIdx.x & 63;

  // ---- 9x9 minimum eigenvector by shifted power iteration (:150-173)
  const int a = lane < 9 ? lane : 8;
  float row[9];
  float lam = 0.0f;
#pragma unroll
  for (int d = 0; d < 9; ++d) {                                    // trace: M_mat[d][d] = M_flat[3p+p][3q+q]
    const int pp = d / 3, qq = d % 3;
    lam += mflat[(3 * pp + pp) * 9 + (3 * qq + qq)];
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    const int pp = a / 3, qq = a % 3, rr = c / 3, ss = c % 3;     // M_mat[3p+q][3r+s] = M_flat[3p+r][3q+s]  (:414)
    row[c] = (a == c ? lam : 0.0f) - mflat[(3 * pp + rr) * 9 + (3 * qq + ss)];
  }
  float v[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) v[c] = 1.0f / 3.0f;
  for (int it = 0; it < n_iter; ++it) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 9; ++c) s += row[c] * v[c];
    float ss = 0.0f;
#pragma unroll
    for (int x = 0; x < 9; ++x) { const float sx = lane_value(s, x); ss += sx * sx; }
    const float nn = sqrtf(ss) + 1e-8f;
    const float mine = s / nn;
#pragma unroll
    for (int x = 0; x < 9; ++x) v[x] = lane_value(mine, x);
  }
  // ---- denormalise: E = T2^T E_raw T1 (:424)
  const float t1[3][3] = {{hart[2], 0.0f, -hart[2] * hart[0]}, {0.0f, hart[2], -hart[2] * hart[1]}, {0.0f, 0.0f, 1.0f}};
  const float t2[3][3] = {{hart[5], 0.0f, -hart[5] * hart[3]}, {0.0f, hart[5], -hart[5] * hart[4]}, {0.0f, 0.0f, 1.0f}};
  float tmp[3][3], e[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) tmp[r][c] = (t2[0][r] * v[0 * 3 + c] + t2[1][r] * v[1 * 3 + c]) + t2[2][r] * v[2 * 3 + c];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) e[r][c] = (tmp[r][0] * t1[0][c] + tmp[r][1] * t1[1][c]) + tmp[r][2] * t1[2][c];
  // ---- manifold projection (:175-248)
  float bm[3][3], bs[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) bm[r][c] = (e[0][r] * e[0][c] + e[1][r] * e[1][c]) + e[2][r] * e[2][c];
  const float lam3 = (bm[0][0] + bm[1][1]) + bm[2][2];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) bs[r][c] = (r == c ? lam3 : 0.0f) - bm[r][c];
  const float inv_sqrt3 = 1.0f / sqrtf(3.0f);
  float va[3] = {inv_sqrt3, inv_sqrt3, inv_sqrt3}, vc[3] = {inv_sqrt3, inv_sqrt3, inv_sqrt3}, vb[3], w3[3];
  {
    float bx[3][3], vx[3] = {inv_sqrt3, inv_sqrt3, inv_sqrt3};      // lane 0: bm -> va; every other lane: bs -> vc
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) bx[r][c] = lane == 0 ? bm[r][c] : bs[r][c];
    for (int it = 0; it < n_iter_manifold; ++it) { matvec3(bx, vx, w3); vx[0] = w3[0]; vx[1] = w3[1]; vx[2] = w3[2]; unit3(vx); }
    for (int r = 0; r < 3; ++r) { va[r] = lane_value(vx[r], 0); vc[r] = lane_value(vx[r], 1); }
  }
  cross3(vc, va, vb);
  unit3(vb);
  float vm[3][3] = {{va[0], vb[0], vc[0]}, {va[1], vb[1], vc[1]}, {va[2], vb[2], vc[2]}};   // columns v1 v2 v3
  const float sgn_v = signf(det3(vm));
  for (int r = 0; r < 3; ++r) vm[r][2] *= sgn_v;
  const float c0[3] = {vm[0][0], vm[1][0], vm[2][0]}, c1[3] = {vm[0][1], vm[1][1], vm[2][1]};
  float ev0[3], ev1[3], u3[3];
  matvec3(e, c0, ev0);
  matvec3(e, c1, ev1);
  const float sg1 = norm3(ev0), sg2 = norm3(ev1);
  const float s_avg = (sg1 + sg2) / 2.0f;
  float u1[3], u2[3];
  for (int r = 0; r < 3; ++r) { u1[r] = ev0[r] / (sg1 + 1e-8f); u2[r] = ev1[r] / (sg2 + 1e-8f); }
  cross3(u1, u2, u3);
  float um[3][3] = {{u1[0], u2[0], u3[0]}, {u1[1], u2[1], u3[1]}, {u1[2], u2[2], u3[2]}};
  const float sgn_u = signf(det3(um));
  for (int r = 0; r < 3; ++r) um[r][2] *= sgn_u;
  // E = U diag(s, s, 0) V^T
  float *out = e_out + (size_t)b * 9;
  if (lane == 0)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) out[r * 3 + c] = (um[r][0] * s_avg) * vm[c][0] + (um[r][1] * s_avg) * vm[c][1];
}

template <typename SRC>
__global__ __launch_bounds__(EM_T) void em_estimate_kernel(const SRC src, int n, int m,
                                                           const float *__restrict__ pts1,
                                                           const float *__restrict__ pts2,
                                                           const uint8_t *__restrict__ valid1,
                                                           const uint8_t *__restrict__ valid2, int top_k, int n_iter,
                                                           int n_iter_manifold, float *__restrict__ e_out) {
  __shared__ EmShared S;
  em_dense_front(S, src, (int)blockIdx.x, n, m, pts1, pts2, valid1, valid2, top_k);
  if (threadIdx.x < 64) em_solve(S, (int)blockIdx.x, n_iter, n_iter_manifold, e_out);   // wave 0
}

// ---- round 3: the same head spread over the chip --------------------------------------------------------------------
// The dense kernel above gives one workgroup per pair four passes over a 1 MB matrix: ~0.6 ms per launch whatever the
// batch (128 pairs: 128 workgroups on 256 CUs, each bound by the latency of its own dependent loads) -- three times
// the rest of the VO model for one pair per call.  But the weights are SPARSE: an entry counts only if it is among the
// top_k of its row (and of its column, and > 0.01), i.e. at most top_k + ties entries per row.  So:
//   em_band_kernel    grid (bands of 32 rows, pairs): ONE pass over the matrix.  Per row: the k-th largest value (wave
//                     reduction over per-lane sorted insertion, as above) and the row's candidates -- entries >= it and
//                     > 0.01, at most EM_CAND, else the row is flagged; per column: the band's top_k values (lanes own
//                     columns, merged over the band's waves in LDS).
//   em_sparse_kernel  one workgroup per pair: column thresholds from the bands' lists, weights of the candidates, row
//                     sums, column sums (every column's few contributions ordered by row before they are added:
//                     deterministic), Hartley parameters, normal equations from per-row moment sums (same wave / row
//                     partition as the dense kernel), then the same serial solve.  A flagged row or a column with more
//                     than EM_CAND contributions (massive exact ties) sends the pair through em_dense_front instead.
// Same definition of every quantity; sums over a row's / column's handful of weights are formed in a different order than
// the dense kernel's lane-strided sums, so the two agree to fp32 rounding (the tests' tolerance), not bit for bit.
constexpr int EM_CAND = 8;          // candidates kept per row / contributions kept per column
constexpr int EB_WAVES = 8, EB_RPW = 4, EB_ROWS = EB_WAVES * EB_RPW;   // band: 8 waves x 4 rows

template <typename SRC, int K, int Q>   // K = top_k (1..4), Q = 64-column groups per row (8: m <= 512, 16: m <= 1024)
__global__ __launch_bounds__(64 * EB_WAVES) __attribute__((amdgpu_waves_per_eu(6, 6))) void em_band_kernel(const SRC src, int n, int m,
                                                                const uint8_t *__restrict__ valid1,
                                                                const uint8_t *__restrict__ valid2,
                                                                float *__restrict__ thr_row, uint8_t *__restrict__ cand_cnt,
                                                                int *__restrict__ cand_j, float *__restrict__ cand_x,
                                                                float *__restrict__ col_part) {
  __shared__ float ctop_s[EB_WAVES][64 * Q][K];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y, band = blockIdx.x, nb = gridDim.x;
  typename SRC::BandCol cols[Q];     // the columns' part of an entry; a column past the matrix counts as invalid (entry 0)
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int j = SRC::column(lane, q);
    const bool ok = j < m && (valid2 ? valid2[(size_t)b * m + min(j, m - 1)] != 0 : true);
    cols[q] = src.band_col(b, m, min(j, m - 1), ok);
  }
  float ctop[Q][K];                  // this wave's top K of every column it has seen (descending)
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int k = 0; k < K; ++k) ctop[q][k] = -INFINITY;
  // every load of the wave's rows is issued here, before the first row is worked on: row by row the kernel was a chain
  // of dependent round trips (the candidate stores of one row keep the compiler from hoisting the next row's loads)
  typename SRC::template RawRow<Q> raws[EB_RPW];
  typename SRC::BandRow rowcs[EB_RPW];
#pragma unroll
  for (int r = 0; r < EB_RPW; ++r) {
    const int i = min(band * EB_ROWS + wave * EB_RPW + r, n - 1);
    rowcs[r] = src.band_row(b, n, i, valid1 ? valid1[(size_t)b * n + i] != 0 : true);
    raws[r] = src.template load_row<Q>(b, n, m, i, lane);
  }
#pragma unroll
  for (int r = 0; r < EB_RPW; ++r) {
    const int i = band * EB_ROWS + wave * EB_RPW + r;
    if (i >= n) break;                                            // wave-uniform
    float x[Q];                                                   // core entries (:334-343); 0 past the matrix
#pragma unroll
    for (int q = 0; q < Q; ++q) x[q] = src.template value<Q>(raws[r], q, rowcs[r], cols[q]);
    // Only entries > 0.01 can carry weight (:345-358), and after 20 Sinkhorn iterations at eps = 0.05 a row has a
    // handful of them, so the selection works on B = {x > 0.01} alone (round 4; the kernel spent two thirds of its
    // instructions inserting zeros into sorted lists):
    //   * the row's candidates {x >= k-th largest of the row, x > 0.01} are {x in B, x >= k-th largest of B}, and all
    //     of B when |B| <= K -- the k-th largest (a wave-wide selection) is only formed when |B| > K;
    //   * a column's threshold matters for entries of B only; the column's k-th largest is the k-th largest of its B
    //     entries when it has K of them and passes every candidate otherwise, as -inf does: the column lists take B only;
    //   * a 64-column group without an entry of B is skipped whole (wave-uniform).
    // thr_row / col_part hold -inf where the full lists held values <= 0.01: the same candidates, weights and E.
    unsigned long long big[Q];
    uint32_t nbig = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      big[q] = __ballot(x[q] > 0.01f);
      nbig += (uint32_t)__popcll(big[q]);
    }
    const bool select = nbig > (uint32_t)K;                        // wave-uniform
    float top[K];
#pragma unroll
    for (int k = 0; k < K; ++k) top[k] = -INFINITY;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (big[q] == 0ull) continue;
      const float xb = x[q] > 0.01f ? x[q] : -INFINITY;
      if (select) {
        float y = xb;
#pragma unroll
        for (int k = 0; k < K; ++k) { const float hi = fmaxf(top[k], y), lo = fminf(top[k], y); top[k] = hi; y = lo; }
      }
      float z = xb;
#pragma unroll
      for (int k = 0; k < K; ++k) { const float hi = fmaxf(ctop[q][k], z), lo = fminf(ctop[q][k], z); ctop[q][k] = hi; z = lo; }
    }
    float kth = -INFINITY;
    if (select) {
#pragma unroll
      for (int s = 0; s < K; ++s) {                               // pop the wave-wide maximum K times (multiplicity kept)
        kth = wave_max_dpp(top[0]);
        const unsigned long long owners = __ballot(top[0] == kth);
        if (lane == __ffsll((long long)owners) - 1) {
#pragma unroll
          for (int k = 0; k + 1 < K; ++k) top[k] = top[k + 1];
          top[K - 1] = -INFINITY;
        }
      }
    }
    // candidates of the row: entries >= the k-th largest that can carry weight at all (> 0.01, :345-358)
    uint32_t cnt = 0;
    const size_t cbase = ((size_t)b * n + i) * EM_CAND;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (big[q] == 0ull) continue;
      const bool c = x[q] >= kth && x[q] > 0.01f;
      const unsigned long long mk = __ballot(c);
      if (c) {
        const uint32_t slot = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        if (slot < (uint32_t)EM_CAND) { cand_j[cbase + slot] = SRC::column(lane, q); cand_x[cbase + slot] = x[q]; }
      }
      cnt += (uint32_t)__popcll(mk);
    }
    if (lane == 0) {
      thr_row[(size_t)b * n + i] = kth;
      cand_cnt[(size_t)b * n + i] = cnt > (uint32_t)EM_CAND ? (uint8_t)255 : (uint8_t)cnt;
    }
  }
  // the band's top K per column: merge the waves' lists
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int k = 0; k < K; ++k) ctop_s[wave][SRC::column(lane, q)][k] = ctop[q][k];
  __syncthreads();
  for (int j = threadIdx.x; j < m; j += 64 * EB_WAVES) {
    float top[K];
#pragma unroll
    for (int k = 0; k < K; ++k) top[k] = ctop_s[0][j][k];
    for (int w = 1; w < EB_WAVES; ++w)
#pragma unroll
      for (int k2 = 0; k2 < K; ++k2) {
        float y = ctop_s[w][j][k2];
        if (y == -INFINITY) break;                                // descending lists, mostly empty
#pragma unroll
        for (int k = 0; k < K; ++k) { const float hi = fmaxf(top[k], y), lo = fminf(top[k], y); top[k] = hi; y = lo; }
      }
#pragma unroll
    for (int k = 0; k < K; ++k) col_part[(((size_t)b * nb + band) * m + j) * K + k] = top[k];
  }
}

template <typename SRC, int K>
__global__ __launch_bounds__(EM_T) void em_sparse_kernel(const SRC src, int n, int m,
                                                         const float *__restrict__ pts1, const float *__restrict__ pts2,
                                                         const uint8_t *__restrict__ valid1,
                                                         const uint8_t *__restrict__ valid2, int n_iter,
                                                         int n_iter_manifold, const float *__restrict__ thr_row_g,
                                                         const uint8_t *__restrict__ cand_cnt,
                                                         const int *__restrict__ cand_j, const float *__restrict__ cand_x,
                                                         const float *__restrict__ col_part, int nb,
                                                         float *__restrict__ e_out) {
  __shared__ EmShared S;
  __shared__ float cw[EM_MAXN][EM_CAND];        // a column's contributions: weights ...
  __shared__ short ci[EM_MAXN][EM_CAND];        // ... and the rows they come from
  __shared__ int ccount[EM_MAXN];
  __shared__ float rmom[EM_MAXN][6];            // per row: sum_j w f2x^2, w f2x f2y, w f2x, w f2y^2, w f2y, w
  __shared__ float2 p1s[EM_MAXN], p2s[EM_MAXN]; // the pair's points (read four times below)
  __shared__ int s_dense;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = blockIdx.x;
  // One workgroup per pair: the kernel's time is its chain of dependent memory round trips, not its arithmetic (128
  // workgroups on 256 CUs).  So everything a thread will need from global memory is requested HERE, at once (round 4):
  // its row's candidates (thread t = row t: n <= EM_T), the pair's points, and -- below, eight bands at a time -- the
  // bands' column lists.
  const bool has_row = t < n;
  const size_t rbase = (size_t)b * n + (has_row ? t : 0);
  const int cnt_raw = has_row ? (int)cand_cnt[rbase] : 0;
  int cj[EM_CAND];
  float cx[EM_CAND];
  {
    const int4 *pj = reinterpret_cast<const int4 *>(cand_j + rbase * EM_CAND);
    const float4 *px = reinterpret_cast<const float4 *>(cand_x + rbase * EM_CAND);
#pragma unroll
    for (int c = 0; c < EM_CAND / 4; ++c) {
      const int4 vj = pj[c];
      const float4 vx = px[c];
      cj[4 * c] = vj.x; cj[4 * c + 1] = vj.y; cj[4 * c + 2] = vj.z; cj[4 * c + 3] = vj.w;
      cx[4 * c] = vx.x; cx[4 * c + 1] = vx.y; cx[4 * c + 2] = vx.z; cx[4 * c + 3] = vx.w;
    }
  }
  if (t < n) p1s[t] = make_float2(pts1[((size_t)b * n + t) * 2], pts1[((size_t)b * n + t) * 2 + 1]);
  if (t < m) p2s[t] = make_float2(pts2[((size_t)b * m + t) * 2], pts2[((size_t)b * m + t) * 2 + 1]);
  const float *q1 = reinterpret_cast<const float *>(p1s), *q2 = reinterpret_cast<const float *>(p2s);
  if (t == 0) s_dense = 0;
  for (int j = t; j < m; j += EM_T) ccount[j] = 0;
  // column thresholds: the k-th largest of the bands' top-K lists (their union holds the column's top K with
  // multiplicity; an empty slot is -inf)
  for (int j = t; j < m; j += EM_T) {
    float top[K];
#pragma unroll
    for (int k = 0; k < K; ++k) top[k] = -INFINITY;
    for (int band0 = 0; band0 < nb; band0 += 8) {
      float y[8][K];
#pragma unroll
      for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int k2 = 0; k2 < K; ++k2)
          y[d][k2] = band0 + d < nb ? col_part[(((size_t)b * nb + band0 + d) * m + j) * K + k2] : -INFINITY;
#pragma unroll
      for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int k2 = 0; k2 < K; ++k2) {
          float z = y[d][k2];
          if (z == -INFINITY) continue;
#pragma unroll
          for (int k = 0; k < K; ++k) { const float hi = fmaxf(top[k], z), lo = fminf(top[k], z); top[k] = hi; z = lo; }
        }
    }
    S.thr_col[j] = top[K - 1];
  }
  __syncthreads();
  if (cnt_raw == 255) s_dense = 1;                                // more tied candidates than kept: dense front
  const int cnt = cnt_raw == 255 ? 0 : cnt_raw;
  // weights of the candidates; a row's sum; every weighted entry filed under its column
  if (has_row) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < EM_CAND; ++c) {
      if (c >= cnt) break;
      const int j = cj[c];
      const float w = cx[c] >= S.thr_col[j] ? cx[c] : 0.0f;      // (>= thr_row and > 0.01 hold for every candidate)
      cx[c] = w;                                                  // from here on: the weight
      s += w;
      if (w != 0.0f) {
        const int slot = atomicAdd(&ccount[j], 1);
        if (slot < EM_CAND) { cw[j][slot] = w; ci[j][slot] = (short)t; } else s_dense = 1;
      }
    }
    S.w1[t] = s;
  }
  __syncthreads();
  if (s_dense) {                                                  // workgroup-uniform
    __syncthreads();
    em_dense_front(S, src, b, n, m, pts1, pts2, valid1, valid2, K);
  } else {
    // column sums: a column's contributions arrive in any order; they are ADDED in ascending row order
    for (int j = t; j < m; j += EM_T) {
      const int ccnt = ccount[j];
      float wv[EM_CAND];
      short iv[EM_CAND];
#pragma unroll
      for (int c = 0; c < EM_CAND; ++c) { wv[c] = c < ccnt ? cw[j][c] : 0.0f; iv[c] = c < ccnt ? ci[j][c] : (short)0x7fff; }
#pragma unroll
      for (int a2 = 1; a2 < EM_CAND; ++a2)                        // insertion sort by row (<= 8 entries, registers)
#pragma unroll
        for (int c = a2; c > 0; --c)
          if (iv[c] < iv[c - 1]) { const short ti = iv[c]; iv[c] = iv[c - 1]; iv[c - 1] = ti; const float tw = wv[c]; wv[c] = wv[c - 1]; wv[c - 1] = tw; }
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < EM_CAND; ++c) s += wv[c];
      S.w2[j] = s;
    }
    __syncthreads();
    if (wave == 0) hartley_wave(q1, S.w1, n, lane, S.hart);
    if (wave == 1) hartley_wave(q2, S.w2, m, lane, S.hart + 3);
    __syncthreads();
    for (int j = t; j < m; j += EM_T) {
      S.f2x[j] = (q2[2 * j + 0] - S.hart[3]) * S.hart[5];
      S.f2y[j] = (q2[2 * j + 1] - S.hart[4]) * S.hart[5];
    }
    __syncthreads();
    // per-row moment sums over the row's weighted candidates (:401-414, inner factor W F2)
    if (has_row) {
      float sxx = 0.0f, sxy = 0.0f, sx = 0.0f, syy = 0.0f, sy = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int c = 0; c < EM_CAND; ++c) {
        if (c >= cnt) break;
        const int j = cj[c];
        const float w = cx[c];
        const float x = S.f2x[j], y = S.f2y[j];
        sxx += w * (x * x); sxy += w * (x * y); sx += w * x; syy += w * (y * y); sy += w * y; s1 += w;
      }
      rmom[t][0] = sxx; rmom[t][1] = sxy; rmom[t][2] = sx; rmom[t][3] = syy; rmom[t][4] = sy; rmom[t][5] = s1;
    }
    __syncthreads();
    // outer factor F1^T (.): the dense kernel's wave / row partition and accumulation order
    float acc0 = 0.0f, acc1 = 0.0f;
    for (int i = wave; i < n; i += EM_W) {
      const float wf2[9] = {rmom[i][0], rmom[i][1], rmom[i][2], rmom[i][1], rmom[i][3], rmom[i][4], rmom[i][2], rmom[i][4], rmom[i][5]};
      const float f1[3] = {(q1[2 * i + 0] - S.hart[0]) * S.hart[2], (q1[2 * i + 1] - S.hart[1]) * S.hart[2], 1.0f};
      {
        const int e = lane, pr = e / 9, qs = e - pr * 9;
        float wsel = 0.0f;
#pragma unroll
        for (int c = 0; c < 9; ++c) wsel = (qs == c) ? wf2[c] : wsel;
        acc0 += (f1[pr / 3] * f1[pr % 3]) * wsel;
      }
      if (lane < 81 - 64) {
        const int e = 64 + lane, pr = e / 9, qs = e - pr * 9;
        float wsel = 0.0f;
#pragma unroll
        for (int c = 0; c < 9; ++c) wsel = (qs == c) ? wf2[c] : wsel;
        acc1 += (f1[pr / 3] * f1[pr % 3]) * wsel;
      }
    }
    S.mpart[wave][lane] = acc0;
    if (lane < 81 - 64) S.mpart[wave][64 + lane] = acc1;
    __syncthreads();
    if (t < 81) {
      float s = 0.0f;
#pragma unroll
      for (int w = 0; w < EM_W; ++w) s += S.mpart[w][t];
      S.mflat[t] = s;
    }
    __syncthreads();
  }
  if (wave == 0) em_solve(S, b, n_iter, n_iter_manifold, e_out);
}

// workspace of the banded form, per pair: thr_row (n floats), candidate counts (n bytes, padded), candidates
// (n x EM_CAND x (int + float)), the bands' column lists (bands x m x K floats)
struct EmWork {
  float *thr_row;
  uint8_t *cand_cnt;
  int *cand_j;
  float *cand_x;
  float *col_part;
  size_t total;
};
EmWork em_carve(void *ws, int batch, int n, int m, int k) {
  const int nb = ceil_div(n, EB_ROWS);
  char *base = reinterpret_cast<char *>(ws);
  size_t off = 0;
  auto take = [&](size_t bytes) { char *q = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return q; };
  EmWork w;
  w.thr_row = reinterpret_cast<float *>(take((size_t)batch * n * 4));
  w.cand_cnt = reinterpret_cast<uint8_t *>(take((size_t)batch * n));
  w.cand_j = reinterpret_cast<int *>(take((size_t)batch * n * EM_CAND * 4));
  w.cand_x = reinterpret_cast<float *>(take((size_t)batch * n * EM_CAND * 4));
  w.col_part = reinterpret_cast<float *>(take((size_t)batch * nb * m * k * 4));
  w.total = off;
  return w;
}

}  // namespace

namespace {
// (y, x) pixel keypoints -> normalised image coordinates (x, y) = first two rows of K^-1 [x, y, 1]^T
// (feature_detection/..._essential_matrix.py:334-360)
__global__ __launch_bounds__(256) void em_normalise_kernel(const float *__restrict__ kpts, long long total,
                                                           const float *__restrict__ k_inv,
                                                           float *__restrict__ pts) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float y = kpts[2 * i + 0], x = kpts[2 * i + 1];
  pts[2 * i + 0] = (x * k_inv[0] + y * k_inv[1]) + k_inv[2];
  pts[2 * i + 1] = (x * k_inv[3] + y * k_inv[4]) + k_inv[5];
}
}  // namespace

extern "C" int mi_normalise_keypoints(const float *keypoints, long long count, const float *k_inv, float *points,
                                      mi_stream_t stream) {
  MI_ENTER();
  if (!keypoints || !k_inv || !points) return MI_E_NULL;
  if (count <= 0 || count > 0x7fffffffLL * 256LL) return MI_E_SHAPE;
  hipLaunchKernelGGL(em_normalise_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     keypoints, count, k_inv, points);
  return mi_launch_status();
}

extern "C" size_t mi_essential_matrix_workspace_bytes(int batch, int n, int m, int top_k) {
  if (batch <= 0 || n <= 0 || m <= 0 || n > EM_MAXN || m > EM_MAXN || top_k < 1 || top_k > 4 || batch > 65535) return 0;
  return em_carve(nullptr, batch, n, m, top_k).total;
}

template <typename SRC>
static int em_launch(const SRC src, int batch, int n, int m, const float *pts1, const float *pts2, const uint8_t *valid1,
                     const uint8_t *valid2, int top_k, int n_iter, int n_iter_manifold, float *e, void *workspace,
                     size_t workspace_bytes, hipStream_t s) {
  if (!pts1 || !pts2 || !e) return MI_E_NULL;
  if ((valid1 == nullptr) != (valid2 == nullptr)) return MI_E_NULL;
  if (batch <= 0 || n <= 0 || m <= 0) return MI_E_SHAPE;
  if (n > EM_MAXN || m > EM_MAXN) return MI_E_PARAM;
  if (top_k <= 0 || top_k > EM_MAXK || top_k > n || top_k > m || n_iter < 0 || n_iter_manifold < 0) return MI_E_PARAM;
  const size_t need = mi_essential_matrix_workspace_bytes(batch, n, m, top_k);
  if (workspace && need > 0) {
    // banded form: one pass over the matrix spread over the chip, then one workgroup per pair on the sparse weights
    if (((uintptr_t)workspace % 16) != 0) return MI_E_ALIGN;
    if (workspace_bytes < need) return MI_E_CAPACITY;
    const EmWork w = em_carve(workspace, batch, n, m, top_k);
    const int nb = ceil_div(n, EB_ROWS);
    const dim3 grid(nb, batch);
#define EM_BAND(K, Q) hipLaunchKernelGGL((em_band_kernel<SRC, K, Q>), grid, dim3(64 * EB_WAVES), 0, s, src, n, m, valid1, valid2, w.thr_row, w.cand_cnt, w.cand_j, w.cand_x, w.col_part)
#define EM_SPARSE(K) hipLaunchKernelGGL((em_sparse_kernel<SRC, K>), dim3(batch), dim3(EM_T), 0, s, src, n, m, pts1, pts2, valid1, valid2, n_iter, n_iter_manifold, w.thr_row, w.cand_cnt, w.cand_j, w.cand_x, w.col_part, nb, e)
    const bool wide = m > 512;
    switch (top_k) {
      case 1: if (wide) EM_BAND(1, 16); else EM_BAND(1, 8); MI_CHECK_LAUNCH(); EM_SPARSE(1); break;
      case 2: if (wide) EM_BAND(2, 16); else EM_BAND(2, 8); MI_CHECK_LAUNCH(); EM_SPARSE(2); break;
      case 3: if (wide) EM_BAND(3, 16); else EM_BAND(3, 8); MI_CHECK_LAUNCH(); EM_SPARSE(3); break;
      default: if (wide) EM_BAND(4, 16); else EM_BAND(4, 8); MI_CHECK_LAUNCH(); EM_SPARSE(4); break;
    }
#undef EM_BAND
#undef EM_SPARSE
    return mi_launch_status();
  }
  hipLaunchKernelGGL(em_estimate_kernel<SRC>, dim3(batch), dim3(EM_T), 0, s, src, n, m, pts1, pts2, valid1, valid2, top_k,
                     n_iter, n_iter_manifold, e);
  return mi_launch_status();
}

extern "C" int mi_essential_matrix(const float *p, int batch, int n, int m, const float *pts1, const float *pts2,
                                   const uint8_t *valid1, const uint8_t *valid2, int top_k, int n_iter,
                                   int n_iter_manifold, float *e, void *workspace, size_t workspace_bytes,
                                   mi_stream_t stream) {
  MI_ENTER();
  if (!p) return MI_E_NULL;
  return em_launch(EmSrcP{p}, batch, n, m, pts1, pts2, valid1, valid2, top_k, n_iter, n_iter_manifold, e, workspace,
                   workspace_bytes, (hipStream_t)stream);
}

// The head straight from the packed-descriptor Sinkhorn solution (mi_cost_dots_bits + mi_sinkhorn_dots with p = NULL):
// see include/mi355x_match.h
extern "C" int mi_essential_matrix_dots(const uint16_t *dots, const float *row_info, const float *col_info, int pitch,
                                        double epsilon, const float *u, const float *v, int batch, int n, int m,
                                        const float *pts1, const float *pts2, const uint8_t *valid1,
                                        const uint8_t *valid2, int top_k, int n_iter, int n_iter_manifold, float *e,
                                        void *workspace, size_t workspace_bytes, mi_stream_t stream) {
  MI_ENTER();
  if (!dots || !row_info || !col_info || !u || !v) return MI_E_NULL;
  if (pitch < m || pitch % 8 != 0 || (reinterpret_cast<uintptr_t>(dots) & 3u) != 0) return MI_E_ALIGN;
  if (!(epsilon >= MI_DOTS_MIN_EPSILON)) return MI_E_PARAM;
  EmSrcDots src;
  src.dots = dots;
  src.pitch = pitch;
  src.row_info = reinterpret_cast<const float2 *>(row_info);
  src.col_info = reinterpret_cast<const float2 *>(col_info);
  src.u = u;
  src.v = v;
  src.neg_inv_eps = (float)(-1.0 / epsilon);
  return em_launch(src, batch, n, m, pts1, pts2, valid1, valid2, top_k, n_iter, n_iter_manifold, e, workspace,
                   workspace_bytes, (hipStream_t)stream);
}
