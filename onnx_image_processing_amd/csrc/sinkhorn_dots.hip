// K6 (packed-descriptor form): log-space Sinkhorn straight from the integer dot products.
// Semantics: reference pytorch_model/matching/sinkhorn.py:95-103 (cost), :112-147 (iterations),
// :178-206 (dustbins, marginals, exp) for hard-binarised descriptors.
//
// K5 stores dot(a_i, b_j) = popcount(a_i & b_j) as uint16 (exact) plus one (scale, squared norm)
// pair per descriptor.  Every pass rebuilds
//     z_ij = -max((|a_i|^2 + |b_j|^2) - 2 * dot_ij * (s_i * s_j), 0) * (1/epsilon)
// in registers, so an iteration streams 2 bytes per matrix element instead of 4.  z is evaluated in
// the factored form c_i + d_j + dot*t_j*g_i (no clamp), a few ulp of |z| from the fp32 path.
// The iteration is the probability-form band kernel of sinkhorn.hip (one exp per element: the
// row pass leaves P_ij = e_ij / s_i in registers, the column update is v_j += log nu_j -
// log(sum_i P_ij), the dustbin row is merged in the log domain).
#include "common.h"

#include <mutex>
#include "sk_tuner.h"
#include "stream_registry.h"

#include <math.h>

#include <atomic>

#include "hooks.h"
// test hooks (include/mi355x_match_debug.h, debug library only; csrc/hooks.h):
//   key 7: 1 = batches of <= SKP_MAX_BATCH pairs (n, m <= 512) run the single-launch form (default), 0 = always the
//          multi-launch form.  Same duals bit for bit.   key 8: 1 = the single-launch kernel records phase time stamps.
//   key 6: number of batch parts run on separate streams (1 = one stream).  Whole bench step: 1.88 / 1.82 / 1.92 ms
//          with 1 / 2 / 3 parts of 256 pairs.

namespace {

__device__ __forceinline__ int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ float sk_exp(float d) { return __builtin_amdgcn_exp2f(d * 1.4426950408889634f); }

constexpr float SKD_L2E = 1.4426950408889634f, SKD_LN2 = 0.6931471805599453f;

struct ZParams {
  float neg_inv_eps;   // -1/epsilon
  float dust;          // -unused_score/epsilon
  float g_bound;       // G (bounded-shift path only)
  float d_bound;       // D
};

__device__ __forceinline__ float z_of(float dot, float2 row, float2 col, float neg_inv_eps) {
  return mi_z_from_dot(dot, row, col, neg_inv_eps);
}

// Bounded-shift row pass.  When the caller can bound the squared norms y of the scaled descriptors
// (sqnorm_bound; 1 for unit descriptors), every dot_ij * t_j * g_i is <= G = 2/eps * sqnorm_bound (for bit
// vectors dot <= min(pop_i, pop_j) <= sqrt(pop_i pop_j)) and every -c_i is <= D = sqnorm_bound/eps, so
//     S = max(max_j wp_j + G, dust + v_m + D)
// is an upper bound of every exponent of the pair: used as the shift of EVERY row, nothing overflows, and a
// row's largest term is >= 2^(-G log2 e), so with G log2 e < SKD_FAST_LIMIT all sums stay far inside the
// normal range (unit descriptors: down to eps ~ 0.03).  The per-row maximum -- a max per element, a wave
// reduction per row and the shift fma ahead of every v_exp_f32 -- disappears.  max_j wp_j comes from the
// column kernels: one maximum per wave of theirs (SKD_AUX slots per pair in the workspace; no barrier in those
// latency-bound kernels).  Otherwise the per-row-maximum kernel runs.
constexpr int SKD_AUX = 20;      // up to 5 blocks (m <= 1024) x 4 waves
constexpr float SKD_FAST_LIMIT = 90.0f;

// wave w of block k of a column kernel publishes the maximum of its 64 columns in slot 4k + w
__device__ __forceinline__ void publish_wave_max(float x, float *__restrict__ aux_pair) {
  x = wave_max_dpp(x);
  if ((threadIdx.x & 63) == 0) aux_pair[blockIdx.x * 4 + (threadIdx.x >> 6)] = x;
}

// Maximum of the low 32 lanes, wave-uniform.  v_max_f32 with the DPP operand inside the instruction: five instructions;
// the compiler's form of the same butterfly (fmaxf on update_dpp) is four per step -- a register copy, the DPP move and
// a canonicalising v_max of each operand in front of the maximum proper.  (The s_nop cover the two wait states a DPP
// read of a freshly written VGPR needs: the assembler pads nothing inside an asm statement.)
__device__ __forceinline__ float max32_lane31(float v) {
  asm volatile(
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
}

// v_j <- v_j + log nu_j - log(sum of the bands' P_ij + exp(B_j))   (B_j: dustbin row, log domain)
__device__ __forceinline__ float combine_column(const float *__restrict__ part_b, int nparts, int m, int j,
                                                float vold, float log_n) {
  const float *p = part_b + j;
  const size_t stride = (size_t)(m + 1);
  // the bands' sums, added in band order.  This kernel is a dependency between two row kernels, so its own
  // latency is on the iteration's critical path: the dustbin row and sixteen bands at a time are requested
  // before anything is added (one L2 round trip for n = 512; eight at a time cost two and a third).
  const float bj = p[(size_t)(nparts - 1) * stride];
  float s = 0.0f;
  int k = 0;
  for (; k + 16 <= nparts - 1; k += 16) {
    float t[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) t[q] = p[(size_t)(k + q) * stride];
#pragma unroll
    for (int q = 0; q < 16; ++q) s += t[q];
  }
  for (; k + 4 <= nparts - 1; k += 4) {
    float t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = p[(size_t)(k + q) * stride];
#pragma unroll
    for (int q = 0; q < 4; ++q) s += t[q];
  }
  for (; k < nparts - 1; ++k) s += p[(size_t)k * stride];
  // log(s + exp(bj)) as a two-term log-sum-exp; s == 0 (everything underflowed) leaves bj.
  // v_log_f32 / v_exp_f32 (1 ulp): this runs at the head of every workgroup of the iteration kernel.
  const float a = s > 0.0f ? __builtin_amdgcn_logf(s) * SKD_LN2 : -INFINITY;
  const float hi = fmaxf(a, bj), lo = fminf(a, bj);
  const float lse = hi + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f((lo - hi) * SKD_L2E)) * SKD_LN2;
  return (vold + ((j == m) ? log_n : 0.0f)) - lse;
}

// Row half of one iteration.  A workgroup owns a band of NW*RW rows (band index nb = the dustbin
// row): u for its rows, and the band's partial column sums of P for the column half
// (sk_vcombine_dots_kernel).  E8 = 16-byte (8 x uint16) loads per lane per row: m <= 512 * E8.
// (Folding the column half into the head of this kernel -- every workgroup recombining all columns --
// was measured slower than the separate 4 us launch: it sits on every workgroup's critical path.)
//
// z_ij = -cost_ij/eps splits into a row constant, a column constant and one product:
//     z_ij = c_i + nie*nb_j + dot_ij * t_j * g_i,   c_i = nie*na_i,  g_i = -2*nie*sa_i,  t_j = sb_j
// (nie = -1/eps; the reference's clamp of the cost at 0 only acts on rounding noise of identical
// descriptors, |z| <= 1e-6/eps there).  The row constant shifts a row's log-sum-exp without changing
// its probabilities, so the element loop is convert, multiply, fma, max, fma, v_exp, add, fma; c_i
// re-enters in u_i and in the dustbin-column entry.  Per-column data come from aligned, padded
// arrays (16-byte loads): tp = t_j (0 in the padding), wp = nie*nb_j + v_j (-inf in the padding:
// such a column contributes nowhere), the latter rebuilt by the combine kernel every iteration.
// fp32(half(bits of w's low / high 16 bits) * t): one v_fma_mix_f32 (addend +0: the product is never negative here)
__device__ __forceinline__ float mix_mul_lo(uint32_t w, float t) {
  float p;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(p) : "v"(w), "v"(t));
  return p;
}
__device__ __forceinline__ float mix_mul_hi(uint32_t w, float t) {
  float p;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(p) : "v"(w), "v"(t));
  return p;
}

// MIX (round 4; the caller vouches for dots < 1024, MI_SOLVER_DOTS_BELOW_1024): a uint16 below 1024 read as an fp16 is the
// denormal dot * 2^-24, which v_fma_mix_f32 widens on the fly -- dot * t becomes ONE instruction (fma(half, t, +0): the
// same single rounding as the multiply, scaled by 2^-24) instead of a convert and a multiply, and the row factor carries
// the 2^24 back (an exact scaling, and the fma forms the exact product before it rounds): the same x bit for bit.
// SPLIT (round 4; 512 < m <= 1024; launched with NW = 16: 32-row bands, half the partial sums): instead of one wave holding
// two 512-column chunks of two rows (E8 = 2, RW = 2), the waves of a workgroup pair up -- wave 2g holds columns 0..511 and wave 2g + 1 columns 512..1023 of the SAME four rows --
// and run the one-chunk code (four rows reduced together, the per-row scalars lane-parallel, the packed multiply-adds);
// the two half sums of a row meet through LDS (total = left + right in both waves).  The sums associate differently from
// the two-chunk kernel's (rounding only: no other kernel restates m > 512; the column kernel takes the band count as an
// argument).  C3 (128 pairs, K = 1024): the call 1.237 -> 1.127 ms, the column kernel 13.9 -> 8.1 us.
template <int E8, int RW, int NW, bool FAST, bool MIX = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * NW) void sk_band_dots_kernel(const uint16_t *__restrict__ dots, int n, int m,
                                                               int pitch, const float2 *__restrict__ row_info,
                                                               ZParams zp, const float *__restrict__ v,
                                                               float *__restrict__ u, float *__restrict__ part,
                                                               float log_m, int v_is_zero,
                                                               const float *__restrict__ wp,
                                                               const float *__restrict__ tp, int cpitch,
                                                               const float *__restrict__ aux) {
  static_assert(!SPLIT || (E8 == 1 && RW == 4 && NW % 2 == 0 && FAST), "the split form is the one-chunk bounded-shift kernel");
  constexpr int RG = SPLIT ? NW / 2 : NW;   // row groups of the workgroup
  constexpr int BAND = RG * RW;   // RG groups x RW rows each
  constexpr int NT = 64 * NW;
  constexpr int NC = 512 * E8;    // columns covered by one wave
  constexpr int NCW = SPLIT ? 2 * NC : NC;   // columns covered by the workgroup
  __shared__ float red[RG][NCW + 1];
  __shared__ float halfsum[SPLIT ? NW : 1][4];   // SPLIT: a wave's four half-row sums, for its partner; and the dustbin band's scratch
  // the wave index is uniform: readfirstlane lets the compiler keep row numbers, row_info and the live flags in
  // SGPRs (scalar loads) instead of per-lane copies
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y, band = blockIdx.x, nb = gridDim.x - 1;
  const float *vb = v + (size_t)b * (m + 1);
  float *pb = part + ((size_t)b * (nb + 1) + band) * (size_t)(m + 1);
  const float vd = v_is_zero ? 0.0f : vb[m];
  const float dust = zp.dust;
  const int rg = SPLIT ? wave >> 1 : wave, half = SPLIT ? wave & 1 : 0, cbase = half * NC;
  const int row0 = band * BAND + rg * RW;

  if (band == nb) {
    auto scr = [&](int w, int i) -> float & { if constexpr (SPLIT) return halfsum[w][i]; else return red[w][i]; };
    // dustbin row: u_n = log m - LSE_j(dust + v_j); its log-probabilities B_j = dust + u_n + v_j
    float mx = dust + vd;
    for (int j = threadIdx.x; j < m; j += NT) mx = fmaxf(mx, dust + (v_is_zero ? 0.0f : vb[j]));
    mx = wave_max_dpp(mx);
    if (lane == 0) scr(wave, 0) = mx;
    __syncthreads();
    mx = scr(0, 0);
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, scr(w, 0));
    float s = 0.0f;
    for (int j = threadIdx.x; j < m; j += NT) s += expf((dust + (v_is_zero ? 0.0f : vb[j])) - mx);
    s = wave_sum_dpp(s);
    if (lane == 0) scr(wave, 1) = s;
    __syncthreads();
    s = scr(0, 1);
#pragma unroll
    for (int w = 1; w < NW; ++w) s += scr(w, 1);
    s += expf((dust + vd) - mx);
    const float un = log_m - (logf(s) + mx);
    if (threadIdx.x == 0) u[(size_t)b * (n + 1) + n] = un;
    for (int j = threadIdx.x; j <= m; j += NT) pb[j] = (dust + un) + (v_is_zero ? 0.0f : vb[j]);
    return;
  }

  // column data (aligned, padded arrays) and the wave's rows: every load is issued before any of them is used
  float4 tl[E8][2], wl[E8][2];
#pragma unroll
  for (int e = 0; e < E8; ++e) {
    const int j = cbase + e * 512 + lane * 8;
    tl[e][0] = *reinterpret_cast<const float4 *>(tp + (size_t)b * cpitch + j);
    tl[e][1] = *reinterpret_cast<const float4 *>(tp + (size_t)b * cpitch + j + 4);
    wl[e][0] = *reinterpret_cast<const float4 *>(wp + (size_t)b * cpitch + j);
    wl[e][1] = *reinterpret_cast<const float4 *>(wp + (size_t)b * cpitch + j + 4);
  }
  uint4 raw[RW][E8];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int i = min(row0 + r, n - 1);          // rows past n run on row n-1 and are given weight 0
    const uint16_t *src = dots + ((size_t)b * n + i) * pitch;
#pragma unroll
    for (int e = 0; e < E8; ++e) {
      // no branch around the load (a branch ends the run of loads in flight: the rows were fetched two at a
      // time, one memory round trip per pair); lanes past the matrix read a valid chunk of the row
      const int j = cbase + e * 512 + lane * 8;
      raw[r][e] = *reinterpret_cast<const uint4 *>(src + min(j, pitch - 8));   // pitch >= round_up(m,8)
    }
  }

  float tq[E8][8], wq[E8][8];
#pragma unroll
  for (int e = 0; e < E8; ++e) {
    const float4 t0 = tl[e][0], t1 = tl[e][1], w0 = wl[e][0], w1 = wl[e][1];
    tq[e][0] = t0.x; tq[e][1] = t0.y; tq[e][2] = t0.z; tq[e][3] = t0.w;
    tq[e][4] = t1.x; tq[e][5] = t1.y; tq[e][6] = t1.z; tq[e][7] = t1.w;
    wq[e][0] = w0.x; wq[e][1] = w0.y; wq[e][2] = w0.z; wq[e][3] = w0.w;
    wq[e][4] = w1.x; wq[e][5] = w1.y; wq[e][6] = w1.z; wq[e][7] = w1.w;
  }
  // (lanes past the matrix hold some other chunk of the row: harmless, their tq is 0 and their wq -inf, so every
  // x of theirs is -inf and every e 0 whatever the finite dot value)
  const float xd0 = dust + vd;

  // Bounded-shift path: one shift S for every row of the pair, known before the row is read; wq becomes
  // wq*log2(e) - S*log2(e) once per wave and x lands in the exponent's units.
  float nm_pair = 0.0f;          // -S * log2(e)
  if constexpr (FAST) {
    // one slot per 64 columns; the slots past the matrix hold -inf (sk_dots_init_kernel, the column kernel's idle
    // waves), so all of them are read: one load per lane and a wave maximum instead of a chain of dependent scalar
    // loads at the head of every workgroup
    static_assert(SKD_AUX <= 64, "one aux slot per lane");
    static_assert(SKD_AUX <= 32, "max32_to_lane31 folds the low 32 lanes");
    const float wmax = max32_lane31(lane < SKD_AUX ? aux[(size_t)b * SKD_AUX + lane] : -INFINITY);
    const float S = fmaxf(wmax + zp.g_bound, xd0 + zp.d_bound);
    nm_pair = -(S * SKD_L2E);
#pragma unroll
    for (int e = 0; e < E8; ++e)
#pragma unroll
      for (int q = 0; q < 8; ++q) wq[e][q] = __builtin_fmaf(wq[e][q], SKD_L2E, nm_pair);
  }

  float colsum[E8][8];
#pragma unroll
  for (int e = 0; e < E8; ++e)
#pragma unroll
    for (int q = 0; q < 8; ++q) colsum[e][q] = 0.0f;
  float dustcol = 0.0f;          // sum of P_i,dustbin over this wave's rows (wave-uniform)
  // The RW rows of the wave go through the three phases together, so that their wave-wide
  // reductions are done four at a time (wave_max4 / wave_sum4).
  float x[RW][E8][8];            // (z_ij - c_i) + v_j (in the exponent's units, shifted, on the fast path), then e_ij in place
  float mx[RW], xd[RW], ci[RW];
  bool live[RW];
  // ROWS (RW == 4): what is computed once per row -- its constants, its dustbin-column entry, log and reciprocal of its
  // sum, its u -- is done in the lanes, the 16 lanes of group g working for row MI_ROW_OF_GROUP(g) (the layout
  // wave_sum4_rows leaves the sums in): 1 instruction instead of 4 wave-uniform ones, 3 transcendentals instead of 12,
  // and the few values the element loops need as scalars are read out with v_readlane.  Same operations on the same
  // values: nothing changes in the results.
  constexpr bool ROWS = RW == 4;
  const int myrow = MI_ROW_OF_GROUP(lane >> 4);
  float civ = 0.0f, giv = 0.0f, xdv = 0.0f;
  if constexpr (ROWS) {
    const float2 riv = row_info[(size_t)b * n + min(row0 + myrow, n - 1)];
    const float g0 = -2.0f * zp.neg_inv_eps * riv.x;
    giv = FAST ? g0 * SKD_L2E : g0;
    if constexpr (MIX) giv *= 16777216.0f;       // 2^24, exact (|g| < 2^11): undoes the 2^-24 of the half-read dots
    civ = riv.y * zp.neg_inv_eps;
    xdv = xd0 - civ;
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int i = row0 + r;
    live[r] = i < n;
    float gi;
    if constexpr (ROWS) {
      gi = mi_readlane_f(giv, 16 * MI_ROW_OF_GROUP(r));
      if constexpr (!FAST) { ci[r] = mi_readlane_f(civ, 16 * MI_ROW_OF_GROUP(r)); xd[r] = xd0 - ci[r]; }
    } else {
      const float2 ri = row_info[(size_t)b * n + min(i, n - 1)];                     // wave-uniform
      const float g0 = -2.0f * zp.neg_inv_eps * ri.x;
      gi = FAST ? g0 * SKD_L2E : g0;
      if constexpr (MIX) gi *= 16777216.0f;
      ci[r] = ri.y * zp.neg_inv_eps;
      xd[r] = xd0 - ci[r];
    }
#pragma unroll
    for (int e = 0; e < E8; ++e) {
      const uint32_t w4[4] = {raw[r][e].x, raw[r][e].y, raw[r][e].z, raw[r][e].w};
      if constexpr (E8 == 1) {
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          // two columns per instruction: v_pk_mul_f32 / v_pk_fma_f32 (the same IEEE operations; a packed fma costs 1.9 ns
          // against 2 x 1.5)
          typedef float v2f __attribute__((ext_vector_type(2)));
          const v2f tt = {tq[e][q], tq[e][q + 1]}, ww = {wq[e][q], wq[e][q + 1]}, gg = {gi, gi};
          v2f dt;
          if constexpr (MIX) {
            dt.x = mix_mul_lo(w4[q >> 1], tt.x);
            dt.y = mix_mul_hi(w4[q >> 1], tt.y);
          } else {
            const v2f dot = {(float)(w4[q >> 1] & 0xFFFFu), (float)(w4[q >> 1] >> 16)};
            dt = dot * tt;
          }
          const v2f xx = __builtin_elementwise_fma(dt, gg, ww);
          x[r][e][q] = xx.x;
          x[r][e][q + 1] = xx.y;
        }
      } else {                 // the two-chunk instance: the pairs' register tuples cost it a wave of occupancy (65 VGPRs)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float dt;
          if constexpr (MIX) dt = (q & 1) ? mix_mul_hi(w4[q >> 1], tq[e][q]) : mix_mul_lo(w4[q >> 1], tq[e][q]);
          else dt = (float)((q & 1) ? (w4[q >> 1] >> 16) : (w4[q >> 1] & 0xFFFFu)) * tq[e][q];
          x[r][e][q] = __builtin_fmaf(dt, gi, wq[e][q]);
        }
      }
      if constexpr (!FAST)
        mx[r] = fmaxf(e == 0 ? xd[r] : mx[r], fmaxf(fmaxf(fmaxf(x[r][e][0], x[r][e][1]), fmaxf(x[r][e][2], x[r][e][3])),
                                                    fmaxf(fmaxf(x[r][e][4], x[r][e][5]), fmaxf(x[r][e][6], x[r][e][7]))));
    }
  }
  // shift and 2^x scaling in one fma; u is taken from the same shift, so the row is normalised by
  // exactly what was summed (same scheme as sk_band_p2_kernel)
  float nm[RW], s[RW], ed[RW];
  if constexpr (FAST) {
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      nm[r] = nm_pair;
      s[r] = 0.0f;
#pragma unroll
      for (int e = 0; e < E8; ++e) {
#pragma unroll
        for (int q = 0; q < 8; ++q) x[r][e][q] = __builtin_amdgcn_exp2f(x[r][e][q]);           // 0 outside the matrix
        const float tree = ((x[r][e][0] + x[r][e][1]) + (x[r][e][2] + x[r][e][3])) + ((x[r][e][4] + x[r][e][5]) + (x[r][e][6] + x[r][e][7]));
        s[r] = e == 0 ? tree : s[r] + tree;      // (0 + tree == tree: a sum of exponentials is never -0)
      }
      if constexpr (!ROWS) ed[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(xd[r], SKD_L2E, nm[r]));   // dustbin column entry
    }
  } else {
    if constexpr (RW % 4 == 0) {
#pragma unroll
      for (int r = 0; r < RW; r += 4) wave_max4(mx + r);
    } else {
#pragma unroll
      for (int r = 0; r < RW; ++r) mx[r] = wave_max_dpp(mx[r]);
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      nm[r] = -(mx[r] * SKD_L2E);
      s[r] = 0.0f;
#pragma unroll
      for (int e = 0; e < E8; ++e) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          x[r][e][q] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[r][e][q], SKD_L2E, nm[r]));   // 0 outside the matrix
        const float tree = ((x[r][e][0] + x[r][e][1]) + (x[r][e][2] + x[r][e][3])) + ((x[r][e][4] + x[r][e][5]) + (x[r][e][6] + x[r][e][7]));
        s[r] = e == 0 ? tree : s[r] + tree;      // (0 + tree == tree: a sum of exponentials is never -0)
      }
      if constexpr (!ROWS) ed[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(xd[r], SKD_L2E, nm[r]));   // dustbin column entry
    }
  }
  float wgtv = 0.0f, edv = 0.0f;
  if constexpr (ROWS) {
    // per row, in the lanes (see ROWS above): dustbin-column entry, the row's total, u_i, 1 / total
    float nmv = nm[0];
    if constexpr (!FAST) nmv = myrow == 1 ? nm[1] : myrow == 2 ? nm[2] : myrow == 3 ? nm[3] : nm[0];
    edv = __builtin_amdgcn_exp2f(__builtin_fmaf(xdv, SKD_L2E, nmv));
    float rowsum = wave_sum4_rows(s);
    if constexpr (SPLIT) {                         // left half + right half, the same expression in both waves
      if ((lane & 15) == 0) halfsum[wave][lane >> 4] = rowsum;
      __syncthreads();
      const float other = halfsum[wave ^ 1][lane >> 4];
      rowsum = half == 0 ? rowsum + other : other + rowsum;
    }
    const float stv = rowsum + edv;
    const bool livev = row0 + myrow < n;
    if ((lane & 15) == 0 && livev && half == 0)                                     // sinkhorn.py:139
      u[(size_t)b * (n + 1) + row0 + myrow] = (nmv - __builtin_amdgcn_logf(stv)) * SKD_LN2 - civ;
    wgtv = livev ? __builtin_amdgcn_rcpf(stv) : 0.0f;
  } else {
#pragma unroll
    for (int r = 0; r < RW; ++r) s[r] = wave_sum_dpp(s[r]);
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    float wgt;
    if constexpr (ROWS) {
      wgt = mi_readlane_f(wgtv, 16 * MI_ROW_OF_GROUP(r));
      ed[r] = mi_readlane_f(edv, 16 * MI_ROW_OF_GROUP(r));
    } else {
      const float st = s[r] + ed[r];
      if (lane == 0 && live[r])                                                     // sinkhorn.py:139
        u[(size_t)b * (n + 1) + row0 + r] = (nm[r] - __builtin_amdgcn_logf(st)) * SKD_LN2 - ci[r];
      wgt = live[r] ? __builtin_amdgcn_rcpf(st) : 0.0f;
    }
#pragma unroll
    for (int e = 0; e < E8; ++e)
#pragma unroll
      for (int q = 0; q < 8; ++q) colsum[e][q] = __builtin_fmaf(x[r][e][q], wgt, colsum[e][q]);   // += P_ij
    dustcol = __builtin_fmaf(ed[r], wgt, dustcol);
  }
#pragma unroll
  for (int e = 0; e < E8; ++e)
#pragma unroll
    for (int q = 0; q < 8; ++q) red[rg][cbase + e * 512 + lane * 8 + q] = colsum[e][q];
  if (lane == 0 && half == 0) red[rg][NCW] = dustcol;
  __syncthreads();
  for (int c = threadIdx.x; c <= NCW; c += NT) {
    const int j = (c == NCW) ? m : c;
    if (c < NCW && j >= m) continue;
    float t = red[0][c];
#pragma unroll
    for (int w = 1; w < RG; ++w) t += red[w][c];
    pb[j] = t;
  }
}

// column half: v_j <- v_j + log nu_j - log(...), and wp_j = nie*nb_j + v_j for the next row half
__global__ __launch_bounds__(256) void sk_vcombine_dots_kernel(const float *__restrict__ part, int m, int nparts,
                                                               float *__restrict__ v, float log_n,
                                                               int v_is_zero, const float2 *__restrict__ col_info,
                                                               float neg_inv_eps, float *__restrict__ wp,
                                                               int cpitch, float *__restrict__ aux) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  float w = -INFINITY;
  if (j <= m) {
    const float vold = v_is_zero ? 0.0f : v[(size_t)b * (m + 1) + j];
    const float vnew = combine_column(part + (size_t)b * nparts * (size_t)(m + 1), nparts, m, j, vold, log_n);
    v[(size_t)b * (m + 1) + j] = vnew;
    if (j < m) {
      w = col_info[(size_t)b * m + j].y * neg_inv_eps + vnew;
      wp[(size_t)b * cpitch + j] = w;
    }
  }
  publish_wave_max(w, aux + (size_t)b * SKD_AUX);   // this wave's share of max_j wp_j for the next row pass
}

// first-iteration column data: tp = column scale, wp = nie * squared norm (v = 0); padding 0 / -inf
__global__ __launch_bounds__(256) void sk_dots_init_kernel(const float2 *__restrict__ col_info, int m, int cpitch,
                                                           float neg_inv_eps, float *__restrict__ wp,
                                                           float *__restrict__ tp, float *__restrict__ aux,
                                                           unsigned *__restrict__ status) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;      // the grid covers cpitch exactly (a multiple of 256)
  if (b == 0 && j == 0) *status = 0u;                // the multi-launch form cannot time out: the call's status word is 0
  const float2 c = (j < m) ? col_info[(size_t)b * m + j] : make_float2(0.0f, 0.0f);
  const float w = (j < m) ? c.y * neg_inv_eps : -INFINITY;
  tp[(size_t)b * cpitch + j] = c.x;
  wp[(size_t)b * cpitch + j] = w;
  publish_wave_max(w, aux + (size_t)b * SKD_AUX);
  // slots no block of this kernel owns: the column kernel has one block more when m is a multiple of 256 (column m)
  if (blockIdx.x == 0 && threadIdx.x >= 4 * gridDim.x && threadIdx.x < SKD_AUX) aux[(size_t)b * SKD_AUX + threadIdx.x] = -INFINITY;
}

// P = exp(Z + u + v) over the augmented matrix; one wave per row
__global__ __launch_bounds__(256) void sk_exp_dots_kernel(const uint16_t *__restrict__ dots, int n, int m, int pitch,
                                                          const float2 *__restrict__ row_info,
                                                          const float2 *__restrict__ col_info, ZParams zp,
                                                          const float *__restrict__ u, const float *__restrict__ v,
                                                          float *__restrict__ p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int i = blockIdx.x * 4 + wave;
  if (i > n) return;
  const float ui = u[(size_t)b * (n + 1) + i];
  const float *vb = v + (size_t)b * (m + 1);
  const uint16_t *dr = dots + ((size_t)b * n + (i < n ? i : 0)) * pitch;
  const float2 ri = row_info[(size_t)b * n + (i < n ? i : 0)];
  float *pr = p + ((size_t)b * (n + 1) + i) * (size_t)(m + 1);
  for (int j = lane; j <= m; j += 64) {
    float zz = zp.dust;
    if (i < n && j < m) zz = z_of((float)dr[j], ri, col_info[(size_t)b * m + j], zp.neg_inv_eps);
    pr[j] = mi_prob_exp((zz + ui) + vb[j]);                         // sinkhorn.py:145,206
  }
}

// The same P, four rows per wave with every load issued before the first use (round 4).  The kernel above walks a row
// in a loop of nine dependent round trips (three loads, exp, store, next 64 columns): 308 us per 448 pairs of 512 x 512 =
// 2.3 TB/s for 235 MB of dots read and 472 MB of P written.  Here a lane owns columns lane, lane + 64, ... (its stores are
// 256 contiguous bytes per instruction), the per-column data (col_info, v) are loaded once per wave for its RW rows, the
// RW x 8 E8 dot products of the wave's rows by as many independent 2-byte loads.  The expressions are the kernel's above,
// operand for operand: the same P bit for bit (and so the same as mnn_band_kernel's registers).
template <int E8>
__global__ __launch_bounds__(256) void sk_exp_rows_kernel(const uint16_t *__restrict__ dots, int n, int m, int pitch,
                                                          const float2 *__restrict__ row_info,
                                                          const float2 *__restrict__ col_info, ZParams zp,
                                                          const float *__restrict__ u, const float *__restrict__ v,
                                                          float *__restrict__ p) {
  constexpr int RW = 4, Q = 8 * E8;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y, nbands = (int)gridDim.x - 1;
  const float *vb = v + (size_t)b * (m + 1);
  if ((int)blockIdx.x == nbands) {                   // the dustbin row
    const float un = u[(size_t)b * (n + 1) + n];
    float *pr = p + ((size_t)b * (n + 1) + n) * (size_t)(m + 1);
    for (int j = threadIdx.x; j <= m; j += 256) pr[j] = mi_prob_exp((zp.dust + un) + vb[j]);
    return;
  }
  const int row0 = ((int)blockIdx.x * 4 + wave) * RW;
  if (row0 >= n) return;
  uint16_t raw[RW][Q];
  float ui[RW];
  float2 ri[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int ic = min(row0 + r, n - 1);
    ui[r] = u[(size_t)b * (n + 1) + ic];
    ri[r] = row_info[(size_t)b * n + ic];
    const uint16_t *dr = dots + ((size_t)b * n + ic) * pitch;
#pragma unroll
    for (int q = 0; q < Q; ++q) raw[r][q] = dr[min(q * 64 + lane, pitch - 1)];
  }
  float vv[Q];
  float2 ci[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int j = min(q * 64 + lane, m - 1);
    vv[q] = vb[j];
    ci[q] = col_info[(size_t)b * m + j];
  }
  const float vd = vb[m];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    if (row0 + r >= n) break;                        // wave-uniform
    float *pr = p + ((size_t)b * (n + 1) + row0 + r) * (size_t)(m + 1);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = q * 64 + lane;
      const float zz = z_of((float)raw[r][q], ri[r], ci[q], zp.neg_inv_eps);
      const float val = mi_prob_exp((zz + ui[r]) + vv[q]);           // sinkhorn.py:145,206
      if (j < m) pr[j] = val;
    }
    if (lane == 0) pr[m] = mi_prob_exp((zp.dust + ui[r]) + vd);      // the dustbin column
  }
}

// ---- single-launch form for a few pairs (the one-pair-per-call latency path) ---------------------------------
// With one pair per call the 41 dependent launches above are all latency: each kernel boundary costs more than
// the work between two of them (MI355X_MICROARCH.md, "boundary" row).  Here ONE launch runs every iteration: the
// nb <= 16 band workgroups of a pair keep their 32 rows of dot products in registers for the whole solve, and the
// bands' column sums cross workgroups as 8-byte {iteration tag, value} granules written write-through (sc1) and
// polled with sc1 loads -- the data is the flag, no fence, no separate barrier (cdna_hip_programming.md section 6,
// Guideline 16, form R2).  Every workgroup gathers all bands' granules of a column and recomputes the column
// update v_j (and the dustbin row, a 513-term reduction) for itself: one hand-off per iteration.  Two granule
// buffers alternate by iteration parity: a workgroup can publish iteration k+2 only after consuming every band's
// iteration k+1, which every band publishes only after consuming iteration k -- so nobody still reads what is
// overwritten.  The arithmetic (operation order included) is that of sk_band_dots_kernel / sk_vcombine_dots_kernel,
// so the duals equal the multi-launch form's bit for bit (asserted in tests/test_gpu_parity.py).
// Requirements: n, m <= 512 and all batch * nb workgroups resident at once (batch <= SKP_MAX_BATCH: 128 workgroups
// of 512 threads; the host checks that the grid fits the device, persist_capacity); granule tags are zeroed by a small
// kernel (common.h: mi_zero_async -- not hipMemsetAsync, see there) ahead of the launch; every spin is bounded (SKP_SPIN_LIMIT polls of >= ~2 us each: about a second) and a
// time-out is LOUD: the call's status word is set and the workgroup writes NaN into its pair's duals before it leaves
// (P, scores and `valid` downstream are then visibly dead; mi_mnn_from_duals_dots also reads the status word).
constexpr int SKP_MAX_BATCH = 8;
constexpr int SKP_COLS = 520;                    // granules per band row: columns 0..m (<= 513), padded
constexpr unsigned SKP_SPIN_LIMIT = 1u << 19;
constexpr size_t SKP_PROF_BYTES = 4096;          // phase time stamps of the development aid (key 8), after the fail word

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS operations, NOT for its outstanding global
// memory operations -- __syncthreads() would drain vmcnt first, i.e. stall every wave until the write-through granule
// stores have been acknowledged and the granule loads in flight have returned (about 1 us each on this path).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <bool FAST, int RW>
__global__ __launch_bounds__(512) void sk_persist_kernel(const uint16_t *__restrict__ dots, int n, int m, int pitch,
                                                         const float2 *__restrict__ row_info,
                                                         const float2 *__restrict__ col_info, ZParams zp, int iterations,
                                                         float *__restrict__ u, float *__restrict__ v,
                                                         unsigned long long *gran, unsigned *fail, float log_m,
                                                         float log_n, unsigned long long *prof, int batch) {
  constexpr int NW = 8, BAND = NW * RW, NT = 64 * NW, NC = 512;
  constexpr int MAXB = 512 / BAND;                 // bands of a pair: 16 (RW = 4) or 32 (RW = 2)
  constexpr int RP = 4;                            // the four-at-a-time wave reductions run on RP rows (padding past RW)
  // development aid (mi_debug_set key 8): band 0 of pair 0 stamps the phases of every iteration (100 MHz clock)
#define SKP_STAMP(slot) do { if (prof && blockIdx.x == 0 && threadIdx.x == 0) prof[it * 8 + (slot)] = wall_clock64(); } while (0)
  __shared__ float red[NW][NC + 1];
  __shared__ float s_w[NC];                      // nie * |b_j|^2 + v_j (-inf past m): the row pass's per-column term
  __shared__ float s_part[3][NW][2];             // maxima / sums of compute_un, maxima of reduce_wmax
  __shared__ float s_vd;                         // v_m, the dustbin column's dual
  __shared__ int s_fail;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md, workgroup dispatch): linear block L sits
  // on XCD L % 8.  The bands of one pair take the blocks L = xcd + 8 * slot of ONE XCD, so that their granules meet
  // in that XCD's L2 (placement is a speed matter only: stores are write-through and loads bypass L1 either way).
  const int nb = ceil_div_dev(n, BAND);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int b = xcd + 8 * (slot / nb), band = slot % nb;
  if (b >= batch) return;
  const int row0 = band * BAND + wave * RW;
  const float dust = zp.dust;
  const int t = threadIdx.x;
  if (t == 0) s_fail = 0;

  // ---- loaded once: this lane's 8 columns' scales, the wave's 4 rows of dot products, the rows' constants
  float tq[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int j = lane * 8 + q;
    tq[q] = (j < m) ? col_info[(size_t)b * m + min(j, m - 1)].x : 0.0f;
  }
  // dt = dot_ij * t_j, the first factor of every iteration's x_ij = fma(dot * t_j, g_i, w_j): converted and multiplied
  // ONCE (the same two roundings as in the multi-launch row kernel, which redoes them per iteration from the uint16 it
  // has to re-read anyway; here they were two of the ~8 instructions per element and iteration)
  float dt[RW][8];
  float gi[RW], ci[RW];
  bool live[RW];
  {
    uint4 raw[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int i = min(row0 + r, n - 1);
      live[r] = row0 + r < n;
      raw[r] = *reinterpret_cast<const uint4 *>(dots + ((size_t)b * n + i) * pitch + min(lane * 8, pitch - 8));
      const float2 ri = row_info[(size_t)b * n + i];
      const float g0 = -2.0f * zp.neg_inv_eps * ri.x;
      gi[r] = FAST ? g0 * SKD_L2E : g0;
      ci[r] = ri.y * zp.neg_inv_eps;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const uint32_t w4[4] = {raw[r].x, raw[r].y, raw[r].z, raw[r].w};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float dot = (float)((q & 1) ? (w4[q >> 1] >> 16) : (w4[q >> 1] & 0xFFFFu));
        dt[r][q] = dot * tq[q];
      }
    }
  }
  // ROWS (RW == 4): the per-row scalars of an iteration computed in the lanes, as in sk_band_dots_kernel
  constexpr bool ROWS = RW == 4;
  const int myrow = MI_ROW_OF_GROUP(lane >> 4);
  const bool livev = row0 + myrow < n;
  float civ = 0.0f;
  if constexpr (ROWS) civ = row_info[(size_t)b * n + min(row0 + myrow, n - 1)].y * zp.neg_inv_eps;
  // ---- the column half's state lives in registers: thread t owns column t (its v_j and nie * |b_j|^2); the dustbin
  // column m is owned by thread m, or by thread 0 as a second column when m == NT
  const bool own0 = t <= m;                                  // column t exists (t == m: the dustbin column)
  const bool core0 = t < m;
  const bool own1 = (t == 0) && (m == NT);                   // column m = 512
  const float cy0 = core0 ? col_info[(size_t)b * m + t].y * zp.neg_inv_eps : -INFINITY;
  float v0 = 0.0f, v1 = 0.0f;                                // v = 0 (sinkhorn.py:134)
  s_w[t] = cy0;                                              // NT == NC: every column of the row pass's array
  if (t == 0) s_vd = 0.0f;

  float un = 0.0f, wmax = 0.0f, vd = 0.0f;
  // State derived from v, by every workgroup for itself, operation for operation what the dustbin band of
  // sk_band_dots_kernel and the column kernel's aux maxima compute:
  //   wmax = max_j (nie |b_j|^2 + v_j): needed by the NEXT row pass (bounded shift) -- one barrier, which also
  //          publishes the s_w / s_vd / s_fail written just before it;
  //   u_n  = log m - LSE_j(dust + v_j), the dustbin row's dual: needed only by the column update, so it is computed
  //          after the granules of the row pass have been published, in the shadow of their flight (two barriers).
  auto reduce_wmax = [&]() {
    float wm = core0 ? cy0 + v0 : -INFINITY;
    wm = wave_max_dpp(wm);
    if (lane == 0) s_part[2][wave][0] = wm;
    lds_barrier();
    vd = s_vd;
    wm = s_part[2][0][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) wm = fmaxf(wm, s_part[2][w][0]);
    wmax = wm;
  };
  auto compute_un = [&]() {
    float mx = core0 ? dust + v0 : -INFINITY;
    mx = wave_max_dpp(mx);
    if (lane == 0) s_part[0][wave][0] = mx;
    lds_barrier();
    mx = fmaxf(dust + vd, s_part[0][0][0]);
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, s_part[0][w][0]);
    float sum = core0 ? expf((dust + v0) - mx) : 0.0f;
    sum = wave_sum_dpp(sum);
    if (lane == 0) s_part[1][wave][0] = sum;
    lds_barrier();
    sum = s_part[1][0][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) sum += s_part[1][w][0];
    sum += expf((dust + vd) - mx);
    un = log_m - (logf(sum) + mx);
  };
  // time-out exit: NaN into every dual of this pair this workgroup can name (its rows' u, every v, the dustbin u) -- all
  // bands of the pair end up here once one of them has stopped publishing, and NaN from any of them is final
  auto poison = [&]() {
    const float qnan = __builtin_nanf("");
#pragma unroll
    for (int r = 0; r < RW; ++r)
      if (lane == 0 && live[r]) u[(size_t)b * (n + 1) + row0 + r] = qnan;
    if (own0) v[(size_t)b * (m + 1) + t] = qnan;
    if (own1) v[(size_t)b * (m + 1) + m] = qnan;
    if (t == 0) u[(size_t)b * (n + 1) + n] = qnan;
  };
  reduce_wmax();

  for (int it = 0; it < iterations; ++it) {
    SKP_STAMP(0);
    const float xd0 = dust + vd;
    float wq[8];
    {
      const float4 w0 = *reinterpret_cast<const float4 *>(&s_w[lane * 8]), w1 = *reinterpret_cast<const float4 *>(&s_w[lane * 8 + 4]);
      wq[0] = w0.x; wq[1] = w0.y; wq[2] = w0.z; wq[3] = w0.w; wq[4] = w1.x; wq[5] = w1.y; wq[6] = w1.z; wq[7] = w1.w;
    }
    float nm_pair = 0.0f;
    if constexpr (FAST) {
      const float S = fmaxf(wmax + zp.g_bound, xd0 + zp.d_bound);
      nm_pair = -(S * SKD_L2E);
#pragma unroll
      for (int q = 0; q < 8; ++q) wq[q] = __builtin_fmaf(wq[q], SKD_L2E, nm_pair);
    }
    // ---- row half (sk_band_dots_kernel's arithmetic)
    float x[RW][8], mx[RP], xd[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      xd[r] = xd0 - ci[r];
#pragma unroll
      for (int q = 0; q < 8; ++q) x[r][q] = __builtin_fmaf(dt[r][q], gi[r], wq[q]);
      if constexpr (!FAST)
        mx[r] = fmaxf(xd[r], fmaxf(fmaxf(fmaxf(x[r][0], x[r][1]), fmaxf(x[r][2], x[r][3])),
                                   fmaxf(fmaxf(x[r][4], x[r][5]), fmaxf(x[r][6], x[r][7]))));
    }
    float nm[RW], sr[RP], ed[RW];
#pragma unroll
    for (int r = RW; r < RP; ++r) { sr[r] = 0.0f; mx[r] = -INFINITY; }
    if constexpr (FAST) {
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        nm[r] = nm_pair;
#pragma unroll
        for (int q = 0; q < 8; ++q) x[r][q] = __builtin_amdgcn_exp2f(x[r][q]);
        sr[r] = ((x[r][0] + x[r][1]) + (x[r][2] + x[r][3])) + ((x[r][4] + x[r][5]) + (x[r][6] + x[r][7]));   // (never -0)
        if constexpr (!ROWS) ed[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(xd[r], SKD_L2E, nm[r]));
      }
    } else {
      wave_max4(mx);
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        nm[r] = -(mx[r] * SKD_L2E);
#pragma unroll
        for (int q = 0; q < 8; ++q) x[r][q] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[r][q], SKD_L2E, nm[r]));
        sr[r] = ((x[r][0] + x[r][1]) + (x[r][2] + x[r][3])) + ((x[r][4] + x[r][5]) + (x[r][6] + x[r][7]));   // (never -0)
        if constexpr (!ROWS) ed[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(xd[r], SKD_L2E, nm[r]));
      }
    }
    float wgtv = 0.0f, edv = 0.0f;
    if constexpr (ROWS) {
      float nmv = nm[0];
      if constexpr (!FAST) nmv = myrow == 1 ? nm[1] : myrow == 2 ? nm[2] : myrow == 3 ? nm[3] : nm[0];
      edv = __builtin_amdgcn_exp2f(__builtin_fmaf(xd0 - civ, SKD_L2E, nmv));
      const float stv = wave_sum4_rows(sr) + edv;
      if (it == iterations - 1 && (lane & 15) == 0 && livev)                             // sinkhorn.py:139
        u[(size_t)b * (n + 1) + row0 + myrow] = (nmv - __builtin_amdgcn_logf(stv)) * SKD_LN2 - civ;
      wgtv = livev ? __builtin_amdgcn_rcpf(stv) : 0.0f;
    } else {
      wave_sum4(sr);
    }
    float colsum[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) colsum[q] = 0.0f;
    float dustcol = 0.0f;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      float wgt;
      if constexpr (ROWS) {
        wgt = mi_readlane_f(wgtv, 16 * MI_ROW_OF_GROUP(r));
        ed[r] = mi_readlane_f(edv, 16 * MI_ROW_OF_GROUP(r));
      } else {
        const float st = sr[r] + ed[r];
        if (it == iterations - 1 && lane == 0 && live[r])                                // sinkhorn.py:139
          u[(size_t)b * (n + 1) + row0 + r] = (nm[r] - __builtin_amdgcn_logf(st)) * SKD_LN2 - ci[r];
        wgt = live[r] ? __builtin_amdgcn_rcpf(st) : 0.0f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) colsum[q] = __builtin_fmaf(x[r][q], wgt, colsum[q]);
      dustcol = __builtin_fmaf(ed[r], wgt, dustcol);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) red[wave][lane * 8 + q] = colsum[q];
    if (lane == 0) red[wave][NC] = dustcol;
    lds_barrier();
    SKP_STAMP(1);
    // ---- publish this band's column sums: one 8-byte {tag, value} granule per column, write-through
    const unsigned tag = (unsigned)it + 1u;
    unsigned long long *gbuf = gran + ((size_t)(it & 1) * batch + b) * (size_t)nb * SKP_COLS;
    auto publish = [&](int c, int j) {                        // c: index into red, j: the column it belongs to
      float s8 = red[0][c];
#pragma unroll
      for (int w = 1; w < NW; ++w) s8 += red[w][c];
      __hip_atomic_store(gbuf + (size_t)band * SKP_COLS + j, ((unsigned long long)tag << 32) | __float_as_uint(s8),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (core0) publish(t, t);
    if (t == 0) publish(NC, m);                               // the dustbin column's sum sits at red[.][NC]
    SKP_STAMP(4);
    // ---- column half, by every workgroup for all columns (sk_vcombine_dots_kernel's arithmetic).  Thread t gathers
    // the bands' granules of column t (thread 0 also column m when m == NT), all loads of a sweep in flight
    // together; late bands are polled again.
    float ssum0 = 0.0f, ssum1 = 0.0f;
    unsigned long long g[MAXB], g1[MAXB];
    auto sweep = [&]() {
      const unsigned long long *gcol = gbuf + t;
#pragma unroll
      for (int k = 0; k < MAXB; ++k)
        g[k] = (k < nb && own0) ? __hip_atomic_load(gcol + (size_t)k * SKP_COLS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : ((unsigned long long)tag << 32);
      if (own1) {
#pragma unroll
        for (int k = 0; k < MAXB; ++k)
          g1[k] = (k < nb) ? __hip_atomic_load(gbuf + m + (size_t)k * SKP_COLS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                           : ((unsigned long long)tag << 32);
      }
    };
    // u_n first (LDS barriers only: about the time the write-through stores need to become visible), then the sweep:
    // a sweep issued too early finds stale tags and costs a whole extra round trip
    compute_un();
    sweep();
    for (unsigned spins = 0;; ++spins) {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < MAXB; ++k) ok = ok && (unsigned)(g[k] >> 32) == tag;
      if (own1) {
#pragma unroll
        for (int k = 0; k < MAXB; ++k) ok = ok && (unsigned)(g1[k] >> 32) == tag;
      }
      if (ok) break;
      if (spins > SKP_SPIN_LIMIT) {                           // a band never arrived: give up, flag it, leave
        s_fail = 1;
        __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      sweep();
    }
#pragma unroll
    for (int k = 0; k < MAXB; ++k)
      if (k < nb) ssum0 += __uint_as_float((unsigned)g[k]);    // band order, as sk_vcombine_dots_kernel adds them
    if (own1) {
#pragma unroll
      for (int k = 0; k < MAXB; ++k)
        if (k < nb) ssum1 += __uint_as_float((unsigned)g1[k]);
    }
    auto column_update = [&](float ssum, float vold, bool dustbin_col) -> float {
      const float bj = (dust + un) + vold;                     // the dustbin row's log-probability
      const float a = ssum > 0.0f ? __builtin_amdgcn_logf(ssum) * SKD_LN2 : -INFINITY;
      const float hi = fmaxf(a, bj), lo = fminf(a, bj);
      const float lse = hi + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f((lo - hi) * SKD_L2E)) * SKD_LN2;
      return (vold + (dustbin_col ? log_n : 0.0f)) - lse;      // v_j (sinkhorn.py:142)
    };
    if (own0) {
      if (t == 0) SKP_STAMP(5);
      v0 = column_update(ssum0, v0, t == m);
      if (core0) s_w[t] = cy0 + v0; else s_vd = v0;
    }
    if (own1) {
      v1 = column_update(ssum1, v1, true);
      s_vd = v1;
    }
    SKP_STAMP(2);
    if (it == iterations - 1) {
      lds_barrier();
      if (s_fail) { poison(); return; }
      if (band == 0) {
        if (own0) v[(size_t)b * (m + 1) + t] = v0;
        if (own1) v[(size_t)b * (m + 1) + m] = v1;
        if (t == 0) u[(size_t)b * (n + 1) + n] = un;
      }
      return;
    }
    reduce_wmax();                                             // (its barrier publishes s_w / s_vd / s_fail)
    if (s_fail) { poison(); return; }                          // uniform: every thread reads it after a barrier
    SKP_STAMP(3);
  }
#undef SKP_STAMP
}

// Helper streams for the split schedule below.  Fork/join by events, so the caller's stream semantics are
// unchanged: everything is ordered after earlier work on `s` and before later work on it.  The streams and
// events belong to ONE caller (device, stream) -- stream_registry.h says why -- and are created on the first
// call with >= 64 pairs on that stream; at most SK_MAX_CALLERS callers get them, later ones run unforked.
//
// Which streams the two half-batches go to is SELF-TUNED per caller and per shape (round 3; hardened in round 4: the
// decision logic and its rules live in sk_tuner.h, HIP-free and unit-tested with injected timings).  Whether two
// streams' kernels overlap depends on how the runtime mapped them onto the device's few hardware queues, i.e. on
// everything the process created before them: with the halves on {caller's stream, helper 0} a 448-pair call takes
// 0.92 ms -- unless an RCCL communicator was created first, then 1.15 ms, while {helper 0, helper 1} takes 0.92 there and
// 1.02 without the communicator; the unsplit call takes 1.00 either way.  So the first calls of a shape on a caller
// stream try the three schedules in turn, each bracketed by two timing events on the caller's stream; later calls
// collect the elapsed times WITHOUT waiting (hipEventQuery) and from then on the shape uses the fastest.  What this
// adds to the caller's stream: two hipEventRecord per trial call (SK_TRIALS calls per shape and window).  Inside a
// stream capture nothing is tried, queried or recorded: the capture gets the decision in force, or the UNSPLIT schedule
// when there is none (no cross-stream fork inside a captured graph unless it was measured -- or pinned -- to pay).  The
// timing calls are made in relaxed capture mode (hipThreadExchangeStreamCaptureMode), so a capture in global mode on
// ANOTHER thread of the process does not turn them into capture errors.  MI_SOLVER_NO_FORK rules the fork out
// altogether; mi_sinkhorn_dots_set_schedule pins a schedule, mi_sinkhorn_dots_schedule reports the one in force.
constexpr int SK_MAX_PARTS = 4;
constexpr size_t SK_MAX_CALLERS = 64;
using mi::SK_SCHEDULES;
using mi::SK_SHAPES;
using mi::SK_TRIALS;
// hipEventRecord / hipEventQuery / stream and event creation under another thread's global-mode capture
struct RelaxedCaptureMode {
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  bool swapped;
  RelaxedCaptureMode() {
    swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
    if (!swapped) (void)hipGetLastError();
  }
  ~RelaxedCaptureMode() {
    if (swapped && hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) (void)hipGetLastError();
  }
  RelaxedCaptureMode(const RelaxedCaptureMode &) = delete;
  RelaxedCaptureMode &operator=(const RelaxedCaptureMode &) = delete;
};
struct ForkJoin {
  hipStream_t side[SK_MAX_PARTS - 1] = {};
  hipEvent_t fork = nullptr, join[SK_MAX_PARTS - 1] = {};
  bool ok = false;
  // self-tuning state (guarded by mu; calls on one stream are normally serial anyway)
  std::mutex mu;
  mi::TunerLogic logic;
  hipEvent_t t0[SK_SHAPES][SK_TRIALS] = {}, t1[SK_SHAPES][SK_TRIALS] = {};   // created when an entry's first trial begins
  bool closed[SK_SHAPES][SK_TRIALS] = {};                                     // t1 recorded: waiting to be harvested
  ForkJoin() {
    RelaxedCaptureMode relaxed;
    ok = hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < SK_MAX_PARTS - 1; ++i)
      ok = ok && hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&join[i], hipEventDisableTiming) == hipSuccess;
  }
  ~ForkJoin() {
    for (int e = 0; e < SK_SHAPES; ++e)
      for (int i = 0; i < SK_TRIALS; ++i) {
        if (t0[e][i]) (void)hipEventDestroy(t0[e][i]);
        if (t1[e][i]) (void)hipEventDestroy(t1[e][i]);
      }
    for (int i = 0; i < SK_MAX_PARTS - 1; ++i) {
      if (join[i]) (void)hipEventDestroy(join[i]);
      if (side[i]) (void)hipStreamDestroy(side[i]);
    }
    if (fork) (void)hipEventDestroy(fork);
  }
  ForkJoin(const ForkJoin &) = delete;
  ForkJoin &operator=(const ForkJoin &) = delete;
  // finished trials -> the decision logic; never waits (mu held, relaxed capture mode)
  void harvest() {
    for (int e = 0; e < SK_SHAPES; ++e)
      for (int i = 0; i < SK_TRIALS; ++i) {
        if (!closed[e][i]) continue;
        if (hipEventQuery(t1[e][i]) != hipSuccess) { (void)hipGetLastError(); continue; }
        closed[e][i] = false;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, t0[e][i], t1[e][i]) == hipSuccess && ms > 0.0f) {
          logic.finish(e, i, (double)ms);
        } else {
          (void)hipGetLastError();
          logic.abandon(e, i);
        }
      }
  }
  bool events_for(int e) {
    for (int i = 0; i < SK_TRIALS; ++i) {
      if (!t0[e][i] && hipEventCreate(&t0[e][i]) != hipSuccess) { t0[e][i] = nullptr; (void)hipGetLastError(); return false; }
      if (!t1[e][i] && hipEventCreate(&t1[e][i]) != hipSuccess) { t1[e][i] = nullptr; (void)hipGetLastError(); return false; }
    }
    return true;
  }
  // The schedule of this call and, while its shape is being tuned, the trial (*entry, *slot) whose events bracket it
  // (-1: none).  Never blocks.  Every trial handed out must be closed (close_trial) or abandoned (abandon_trial).
  int pick(hipStream_t s, const mi::TunerShape &shape, int *entry, int *slot) {
    *entry = *slot = -1;
    std::lock_guard<std::mutex> lock(mu);
    // inside a stream capture nothing is queried or recorded: the decision in force, else the unsplit schedule
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return logic.for_capture(shape); }
    if (cap != hipStreamCaptureStatusNone) return logic.for_capture(shape);
    RelaxedCaptureMode relaxed;
    harvest();
    const int sched = logic.begin(shape, entry, slot);
    if (*slot >= 0) {
      closed[*entry][*slot] = false;
      if (!events_for(*entry) || hipEventRecord(t0[*entry][*slot], s) != hipSuccess) {
        (void)hipGetLastError();
        logic.abandon(*entry, *slot);                          // nothing learnt from this slot; the call still runs `sched`
        *entry = *slot = -1;
      }
    }
    return sched;
  }
  void close_trial(int entry, int slot, hipStream_t s) {
    std::lock_guard<std::mutex> lock(mu);
    RelaxedCaptureMode relaxed;
    if (hipEventRecord(t1[entry][slot], s) == hipSuccess) {
      closed[entry][slot] = true;
    } else {                                                   // the slot ends without a sample
      (void)hipGetLastError();
      logic.abandon(entry, slot);
    }
  }
  void abandon_trial(int entry, int slot) {                    // an error between the trial's two events
    std::lock_guard<std::mutex> lock(mu);
    logic.abandon(entry, slot);
  }
  int current(const mi::TunerShape &shape) {
    std::lock_guard<std::mutex> lock(mu);
    RelaxedCaptureMode relaxed;
    harvest();
    return logic.current(shape);
  }
  bool set(int schedule) {
    std::lock_guard<std::mutex> lock(mu);
    return logic.set(schedule);
  }
};
using ForkJoinKey = std::pair<int, hipStream_t>;   // (device, caller stream)
mi::KeyedRegistry<ForkJoinKey, ForkJoin> &fork_join_registry() {
  // leaked on purpose: destroying streams from a static destructor races with the HIP runtime's own teardown
  static auto *reg = new mi::KeyedRegistry<ForkJoinKey, ForkJoin>(SK_MAX_CALLERS);
  return *reg;
}
ForkJoin *fork_join_for(hipStream_t s) {
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess) return nullptr;     // launches on `s` need its device current anyway
  return fork_join_registry().get(ForkJoinKey(device, s), []() {
    std::unique_ptr<ForkJoin> fj(new ForkJoin());
    if (!fj->ok) fj.reset();
    return fj;
  });
}

template <int E8, int RW, int NW, bool FAST, bool MIX = false, bool SPLIT = false>
int launch_dots(const uint16_t *dots, const float2 *ri, const float2 *ci, int batch, int n, int m, int pitch,
                 ZParams zp, int iterations, float *u, float *v, float *part, float *wp, float *tp, float *aux,
                 float log_m, float log_n, unsigned *statusw, bool no_fork, hipStream_t s) {
  const int nb = ceil_div(n, (SPLIT ? NW / 2 : NW) * RW);
  constexpr int CP = 512 * E8 * (SPLIT ? 2 : 1);    // padded column count of wp / tp
  hipLaunchKernelGGL(sk_dots_init_kernel, dim3(CP / 256, batch), dim3(256), 0, s, ci, m, CP, zp.neg_inv_eps, wp, tp,
                     aux, statusw);
  // An iteration is a big row kernel and a tiny column kernel that depend on each other, so between them
  // the GPU drains and refills (about 5 us per iteration).  With enough pairs the batch is cut into parts
  // on separate streams: while one part is in its column kernel / launch gap another part's row kernel
  // keeps the CUs busy.  The parts are independent problems, so results do not change.
  MI_CHECK_LAUNCH();                                           // (before any trial is handed out)
  int parts = MI_HOOK(sinkhorn_split, 2);
  if (parts > SK_MAX_PARTS) parts = SK_MAX_PARTS;
  if (parts < 1 || batch < 32 * parts || no_fork) parts = 1;   // MI_SOLVER_NO_FORK: everything on the caller's stream
  ForkJoin *fj = parts > 1 ? fork_join_for(s) : nullptr;
  if (!fj) parts = 1;
  // first_side: the first part that runs on a helper stream (1: part 0 stays on the caller's stream; 0: every part on a
  // helper).  With the default two parts the caller's schedule is self-tuned (ForkJoin above).
  int first_side = 1, trial_entry = -1, trial = -1;
  if (parts == 2) {
    const int fixed = MI_HOOK(sinkhorn_schedule, -1);
    mi::TunerShape shape;
    shape.batch = batch; shape.n = n; shape.m = m; shape.iterations = iterations;
    const int sched = fixed >= 0 ? fixed : fj->pick(s, shape, &trial_entry, &trial);
    if (sched == 1) first_side = 0;
    if (sched == 2) parts = 1;
  }
  if (parts > 1) {
    // any failure here leaves the side streams unused: run unforked (a side stream that already waits is harmless)
    bool forked = hipEventRecord(fj->fork, s) == hipSuccess;
    for (int q = first_side; forked && q < parts; ++q)
      forked = hipStreamWaitEvent(fj->side[q - first_side], fj->fork, 0) == hipSuccess;
    if (!forked) {
      (void)hipGetLastError();
      parts = 1;
    }
  }
  // enqueue iteration by iteration so that no stream runs far ahead of the others
  for (int it = 0; it < iterations; ++it) {
    const int vz = it == 0 ? 1 : 0;
    for (int q = 0; q < parts; ++q) {
      const int b0 = (int)((long long)batch * q / parts), nbatch = (int)((long long)batch * (q + 1) / parts) - b0;
      hipStream_t st = (parts > 1 && q >= first_side) ? fj->side[q - first_side] : s;
      const uint16_t *d0 = dots + (size_t)b0 * n * pitch;
      const float2 *ri0 = ri + (size_t)b0 * n, *ci0 = ci + (size_t)b0 * m;
      float *u0 = u + (size_t)b0 * (n + 1), *v0 = v + (size_t)b0 * (m + 1);
      float *part0 = part + (size_t)b0 * (nb + 1) * (size_t)(m + 1);
      float *wp0 = wp + (size_t)b0 * CP, *tp0 = tp + (size_t)b0 * CP, *aux0 = aux + (size_t)b0 * SKD_AUX;
      hipLaunchKernelGGL((sk_band_dots_kernel<E8, RW, NW, FAST, MIX, SPLIT>), dim3(nb + 1, nbatch), dim3(64 * NW), 0, st, d0, n, m, pitch,
                         ri0, zp, v0, u0, part0, log_m, vz, wp0, tp0, CP, aux0);
      hipLaunchKernelGGL(sk_vcombine_dots_kernel, dim3(ceil_div(m + 1, 256), nbatch), dim3(256), 0, st, part0, m, nb + 1,
                         v0, log_n, vz, ci0, zp.neg_inv_eps, wp0, CP, aux0);
    }
  }
  // join: without it later work on `s` would not be ordered after the side streams, so a failure is an error
  for (int q = first_side; q < parts; ++q) {
    hipError_t e = hipEventRecord(fj->join[q - first_side], fj->side[q - first_side]);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, fj->join[q - first_side], 0);
    if (e != hipSuccess) {
      if (trial >= 0) fj->abandon_trial(trial_entry, trial);   // the slot must end, or its window never closes
      return (int)e;
    }
  }
  const int launched = mi_launch_status();
  if (trial >= 0) {
    if (launched == MI_OK) fj->close_trial(trial_entry, trial, s);
    else fj->abandon_trial(trial_entry, trial);
  }
  return launched;
}

}  // namespace

// The same two-half-batches-on-two-streams form for mi_sinkhorn (fp32 log-scores; sinkhorn.hip): begin decides the
// schedule with THIS file's per-stream tuner (its own shapes: `key` tells the solvers apart) and records the fork, end
// joins and closes the trial.  parts == 1: everything on the caller's stream.
int mi_fork_begin(hipStream_t s, int batch, int n, int m, int key, MiFork *f) {
  f->parts = 1;
  f->stream[0] = f->stream[1] = s;
  f->handle = nullptr;
  f->trial_entry = f->trial = -1;
  f->first_side = 1;
  if (batch < 64) return MI_OK;
  ForkJoin *fj = fork_join_for(s);
  if (!fj) return MI_OK;
  const int fixed = MI_HOOK(sinkhorn_schedule, -1);
  mi::TunerShape shape;
  shape.batch = batch; shape.n = n; shape.m = m; shape.iterations = key;
  const int sched = fixed >= 0 ? fixed : fj->pick(s, shape, &f->trial_entry, &f->trial);
  f->handle = fj;
  if (sched == 2) return MI_OK;                                // unsplit (a trial of it is still timed)
  f->first_side = sched == 1 ? 0 : 1;
  bool forked = hipEventRecord(fj->fork, s) == hipSuccess;
  for (int q = f->first_side; forked && q < 2; ++q)
    forked = hipStreamWaitEvent(fj->side[q - f->first_side], fj->fork, 0) == hipSuccess;
  if (!forked) {                                               // run unforked (a side stream that already waits is harmless)
    (void)hipGetLastError();
    return MI_OK;
  }
  f->parts = 2;
  for (int q = 0; q < 2; ++q) f->stream[q] = q >= f->first_side ? fj->side[q - f->first_side] : s;
  return MI_OK;
}

int mi_fork_end(hipStream_t s, MiFork *f) {
  ForkJoin *fj = static_cast<ForkJoin *>(f->handle);
  if (f->parts > 1) {
    for (int q = f->first_side; q < 2; ++q) {                  // a failed join is an error: later work on `s` would not be ordered
      hipError_t e = hipEventRecord(fj->join[q - f->first_side], fj->side[q - f->first_side]);
      if (e == hipSuccess) e = hipStreamWaitEvent(s, fj->join[q - f->first_side], 0);
      if (e != hipSuccess) {
        if (f->trial >= 0) fj->abandon_trial(f->trial_entry, f->trial);
        return (int)e;
      }
    }
  }
  const int launched = mi_launch_status();
  if (fj && f->trial >= 0) {
    if (launched == MI_OK) fj->close_trial(f->trial_entry, f->trial, s);
    else fj->abandon_trial(f->trial_entry, f->trial);
  }
  return launched;
}

namespace {

int dots_rows_per_band(int m) { return m <= 512 ? 32 : (m <= 1024 ? 16 : 0); }   // 8 rows per wave measured slower

// workspace: band partials (one float per column per band), then the padded per-column arrays wp, tp
size_t dots_partials_bytes(int batch, int n, int m, int band) {
  const size_t b = (size_t)batch * (size_t)(ceil_div(n, band) + 1) * (size_t)(m + 1) * sizeof(float);
  return (b + 15) & ~(size_t)15;
}
int dots_cpitch(int m) { return m <= 512 ? 512 : 1024; }

// single-launch form: two granule buffers of nb band rows per pair (then the status word, which every shape has)
bool persist_shape(int batch, int n, int m) { return batch <= SKP_MAX_BATCH && n <= 512 && m <= 512; }
int persist_grid(int batch, int n) { return 8 * ceil_div(n, 32) * ceil_div(batch, 8); }
// P from the duals (key 18: 1 = four rows per wave, every load up front; 0 = the one-row-per-wave loop; the same P)
void launch_exp(const uint16_t *dots, int n, int m, int pitch, const float2 *ri, const float2 *ci, ZParams zp, const float *u,
                const float *v, float *p, int batch, hipStream_t s) {
  if (MI_HOOK(sinkhorn_exp_rows, 1) == 0)
    hipLaunchKernelGGL(sk_exp_dots_kernel, dim3(ceil_div(n + 1, 4), batch), dim3(256), 0, s, dots, n, m, pitch, ri, ci, zp, u, v, p);
  else if (m <= 512)
    hipLaunchKernelGGL(sk_exp_rows_kernel<1>, dim3(ceil_div(n, 16) + 1, batch), dim3(256), 0, s, dots, n, m, pitch, ri, ci, zp, u, v, p);
  else
    hipLaunchKernelGGL(sk_exp_rows_kernel<2>, dim3(ceil_div(n, 16) + 1, batch), dim3(256), 0, s, dots, n, m, pitch, ri, ci, zp, u, v, p);
}

// Workgroups of the single-launch kernel the current device can hold at once: occupancy per compute unit x compute
// units, the smaller of the two row-pass variants, asked once per device (0 when the query fails: multi-launch form).
// A CU-masked or partitioned device reports what it really has, which is the point.
int persist_capacity() {
  static std::atomic<int> cache[64];
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) { (void)hipGetLastError(); return 0; }
  int cap = cache[device].load(std::memory_order_relaxed);
  if (cap != 0) return cap > 0 ? cap : 0;
  int cus = 0, a = 0, c = 0;
  const bool ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
                  hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, sk_persist_kernel<true, 4>, 512, 0) == hipSuccess &&
                  hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, sk_persist_kernel<false, 4>, 512, 0) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  cap = ok ? (a < c ? a : c) * cus : 0;
  cache[device].store(cap > 0 ? cap : -1, std::memory_order_relaxed);
  return cap;
}
// Rows per wave of the single-launch kernel.  4 = the multi-launch kernel's banding, which its column sums must share
// to come out bit-identical (a band's column sum is an fma chain over its rows: 16-row bands -- twice the workgroups,
// half the row pass -- were measured 0.5 us per iteration faster but round differently).
int persist_rows_per_wave(int) { return 4; }
size_t persist_granule_bytes(int batch, int n) {
  return 2 * (size_t)batch * (size_t)ceil_div(n, 32) * SKP_COLS * sizeof(unsigned long long);
}
size_t dots_base_bytes(int batch, int n, int m) {
  const size_t b = dots_partials_bytes(batch, n, m, dots_rows_per_band(m)) +
                   (2 * (size_t)batch * dots_cpitch(m) + (size_t)batch * SKD_AUX) * sizeof(float);
  return (b + 15) & ~(size_t)15;
}
// workspace: [multi-launch arrays | granules (single-launch shapes only) | status word, 16 bytes | stamps (ditto)]
size_t dots_status_offset(int batch, int n, int m) {
  return dots_base_bytes(batch, n, m) + (persist_shape(batch, n, m) ? persist_granule_bytes(batch, n) : 0);
}

}  // namespace

// The form mi_sinkhorn_dots takes: a pure function of the extents, the caller's flags and what the device can hold
// (mi_debug_sinkhorn_dots_form of the debug library exposes it to the CPU tests).
int mi_sinkhorn_dots_form_host(int batch, int n, int m, int flags, int blocks_per_cu, int cus) {
  if (!persist_shape(batch, n, m) || (flags & MI_SOLVER_MULTI_LAUNCH) != 0) return 0;
  if (MI_HOOK(sinkhorn_persist, 1) == 0) return 0;
  return (long long)blocks_per_cu * cus >= persist_grid(batch, n) ? 1 : 0;
}
static bool use_single_launch(int batch, int n, int m, int flags) {
  if (mi_sinkhorn_dots_form_host(batch, n, m, flags, 1, 1 << 20) == 0) return false;      // shape / flags / hook say no
  return mi_sinkhorn_dots_form_host(batch, n, m, flags, persist_capacity(), 1) != 0;
}

extern "C" size_t mi_sinkhorn_dots_workspace_bytes(int batch, int n, int m) {
  const int band = dots_rows_per_band(m);
  if (batch <= 0 || n <= 0 || m <= 0 || band == 0) return 0;
  return dots_status_offset(batch, n, m) + 16 + (persist_shape(batch, n, m) ? SKP_PROF_BYTES : 0);
}

extern "C" const uint32_t *mi_sinkhorn_dots_status_word(const void *workspace, int batch, int n, int m) {
  if (!workspace || mi_sinkhorn_dots_workspace_bytes(batch, n, m) == 0) return nullptr;
  return reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(workspace) + dots_status_offset(batch, n, m));
}

// see include/mi355x_match.h: a caller that destroys a stream it passed to mi_sinkhorn_dots / mi_match_pairs
// releases the helper streams and events held for it (they would otherwise live until process exit)
extern "C" int mi_release_stream_resources(mi_stream_t stream) {
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess) return MI_E_PARAM;
  fork_join_registry().release(ForkJoinKey(device, (hipStream_t)stream));
  return MI_OK;
}

// The stream schedule of mi_sinkhorn_dots for >= 64 pairs, see include/mi355x_match.h
extern "C" int mi_sinkhorn_dots_schedule(mi_stream_t stream, int batch, int n, int m, int iterations) {
  const int fixed = MI_HOOK(sinkhorn_schedule, -1);
  if (fixed >= 0) return fixed;
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); return MI_SCHEDULE_UNDECIDED; }
  ForkJoin *fj = fork_join_registry().find(ForkJoinKey(device, (hipStream_t)stream));
  if (!fj) return MI_SCHEDULE_UNDECIDED;
  mi::TunerShape shape;
  shape.batch = batch; shape.n = n; shape.m = m; shape.iterations = iterations;
  return fj->current(shape);
}

extern "C" int mi_sinkhorn_dots_set_schedule(mi_stream_t stream, int schedule) {
  if (schedule < -1 || schedule >= SK_SCHEDULES) return MI_E_PARAM;
  ForkJoin *fj = fork_join_for((hipStream_t)stream);          // creates the caller's helper resources if need be
  if (!fj) return MI_E_CAPACITY;
  return fj->set(schedule) ? MI_OK : MI_E_PARAM;
}

size_t mi_sinkhorn_dots_handoff_region(void *workspace, int batch, int n, int m, int flags, void **region) {
  if (!use_single_launch(batch, n, m, flags)) return 0;
  *region = reinterpret_cast<char *>(workspace) + dots_base_bytes(batch, n, m);
  return persist_granule_bytes(batch, n) + 16;
}

extern "C" int mi_sinkhorn_dots(const uint16_t *dots, const float *row_info, const float *col_info, int batch,
                                int n, int m, int pitch, double epsilon, double unused_score, double sqnorm_bound,
                                int iterations, float *u, float *v, float *p, void *workspace, size_t workspace_bytes,
                                int flags, mi_stream_t stream) {
  MI_ENTER();
  return mi_sinkhorn_dots_impl(dots, row_info, col_info, batch, n, m, pitch, epsilon, unused_score, sqnorm_bound,
                               iterations, u, v, p, workspace, workspace_bytes, flags, 0, stream);
}

int mi_sinkhorn_dots_impl(const uint16_t *dots, const float *row_info, const float *col_info, int batch, int n, int m,
                          int pitch, double epsilon, double unused_score, double sqnorm_bound, int iterations, float *u,
                          float *v, float *p, void *workspace, size_t workspace_bytes, int flags, int prezeroed,
                          mi_stream_t stream) {
  if (!dots || !row_info || !col_info || !u || !v || !workspace) return MI_E_NULL;
  if (batch <= 0 || n <= 0 || m <= 0 || batch > 65535) return MI_E_SHAPE;
  if (pitch < m || pitch % 8 != 0 || ((uintptr_t)dots % 16) != 0 || ((uintptr_t)workspace % 16) != 0) return MI_E_ALIGN;
  if (iterations <= 0 || !(epsilon >= MI_DOTS_MIN_EPSILON)) return MI_E_PARAM;   // below it: the fp32-Z form (clamped cost)
  if ((flags & ~(MI_SOLVER_MULTI_LAUNCH | MI_SOLVER_NO_FORK | MI_SOLVER_DOTS_BELOW_1024)) != 0) return MI_E_PARAM;
  const size_t need = mi_sinkhorn_dots_workspace_bytes(batch, n, m);
  if (need == 0) return MI_E_PARAM;                 // m > 1024: use the fp32 form
  if (workspace_bytes < need) return MI_E_CAPACITY;
  hipStream_t s = (hipStream_t)stream;
  const float log_m = logf((float)m), log_n = logf((float)n);      // sinkhorn.py:197-198
  ZParams zp;
  zp.neg_inv_eps = (float)(-1.0 / epsilon);
  zp.dust = (float)(-unused_score / epsilon);
  const float2 *ri = reinterpret_cast<const float2 *>(row_info);
  const float2 *ci = reinterpret_cast<const float2 *>(col_info);
  float *part = reinterpret_cast<float *>(workspace);
  float *wp = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) +
                                        dots_partials_bytes(batch, n, m, dots_rows_per_band(m)));
  float *tp = wp + (size_t)batch * dots_cpitch(m);
  float *aux = tp + (size_t)batch * dots_cpitch(m);
  // bounded-shift row pass when the caller's norm bound makes it safe (see SKD_AUX)
  zp.g_bound = (float)(2.0 / epsilon * sqnorm_bound);
  zp.d_bound = (float)(sqnorm_bound / epsilon);
  const bool fast = sqnorm_bound > 0.0 && (double)zp.g_bound * 1.4426950408889634 < (double)SKD_FAST_LIMIT;
  unsigned *statusw = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(workspace) + dots_status_offset(batch, n, m));
  if (use_single_launch(batch, n, m, flags)) {
    // single-launch form: zero the granule tags and the status word (one small kernel: common.h, mi_zero_async), then
    // one kernel
    char *gbase = reinterpret_cast<char *>(workspace) + dots_base_bytes(batch, n, m);
    const size_t gbytes = persist_granule_bytes(batch, n);
    if (!prezeroed) {
      const int me = mi_zero_async(gbase, gbytes + 16, s);
      if (me != MI_OK) return me;
    }
    unsigned long long *gran = reinterpret_cast<unsigned long long *>(gbase);
    unsigned *failw = statusw;                       // == gbase + gbytes
    unsigned long long *prof = MI_HOOK(sinkhorn_stamps, 0) && iterations * 64 <= (int)SKP_PROF_BYTES
                                   ? reinterpret_cast<unsigned long long *>(gbase + gbytes + 16) : nullptr;
    const int rw = persist_rows_per_wave(batch);
    const dim3 grid(persist_grid(batch, n));
#define SKP_LAUNCH(FAST, RW) hipLaunchKernelGGL((sk_persist_kernel<FAST, RW>), grid, dim3(512), 0, s, dots, n, m, pitch, ri, ci, zp, iterations, u, v, gran, failw, log_m, log_n, prof, batch)
    if (fast) { if (rw == 2) SKP_LAUNCH(true, 2); else SKP_LAUNCH(true, 4); }
    else { if (rw == 2) SKP_LAUNCH(false, 2); else SKP_LAUNCH(false, 4); }
#undef SKP_LAUNCH
    MI_CHECK_LAUNCH();
    if (p)
      launch_exp(dots, n, m, pitch, ri, ci, zp, u, v, p, batch, s);
    return mi_launch_status();
  }
  // dots vouched to be < 1024 (MI_SOLVER_DOTS_BELOW_1024): the row kernel reads them as fp16 denormals (MIX above)
  const bool mix = (flags & MI_SOLVER_DOTS_BELOW_1024) != 0 && MI_HOOK(sinkhorn_mix, 1) != 0;
#define SKD_LAUNCH_NW(E8, RW, NW, FAST, MIX, SPLIT) launch_dots<E8, RW, NW, FAST, MIX, SPLIT>(dots, ri, ci, batch, n, m, pitch, zp, iterations, u, v, part, wp, tp, aux, log_m, log_n, statusw, (flags & MI_SOLVER_NO_FORK) != 0, s)
#define SKD_LAUNCH(E8, RW, FAST, MIX, SPLIT) SKD_LAUNCH_NW(E8, RW, 8, FAST, MIX, SPLIT)
  int e;
  if (m <= 512) {
    if (mix) e = fast ? SKD_LAUNCH(1, 4, true, true, false) : SKD_LAUNCH(1, 4, false, true, false);
    else e = fast ? SKD_LAUNCH(1, 4, true, false, false) : SKD_LAUNCH(1, 4, false, false, false);
  } else if (fast && MI_HOOK(sinkhorn_pair_waves, 1) != 0) {   // two waves per row group, one 512-column chunk each (SPLIT above)
    e = mix ? SKD_LAUNCH_NW(1, 4, 16, true, true, true) : SKD_LAUNCH_NW(1, 4, 16, true, false, true);
  } else {
    if (mix) e = fast ? SKD_LAUNCH(2, 2, true, true, false) : SKD_LAUNCH(2, 2, false, true, false);
    else e = fast ? SKD_LAUNCH(2, 2, true, false, false) : SKD_LAUNCH(2, 2, false, false, false);
  }
#undef SKD_LAUNCH
#undef SKD_LAUNCH_NW
  if (e != MI_OK) return e;
  if (p) {
    launch_exp(dots, n, m, pitch, ri, ci, zp, u, v, p, batch, s);
  }
  return mi_launch_status();
}
