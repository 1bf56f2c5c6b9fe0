// Shared host/device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mi355x_match.h"

#define MI_WAVE 64

// Every launching entry point starts with MI_ENTER(): hipGetLastError() also reports (and clears) an error left
// behind by an unrelated earlier HIP call of this thread, which would otherwise be blamed on this entry point.
#define MI_ENTER() (void)hipGetLastError()
static inline int mi_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MI_OK : (int)e;
}
// inside multi-launch sequences: stop at the first launch that failed
#define MI_CHECK_LAUNCH()                          \
  do {                                             \
    const int _mi_e = mi_launch_status();          \
    if (_mi_e != MI_OK) return _mi_e;              \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Clearing device memory on a stream WITHOUT hipMemsetAsync.  A memset call captured into a hipGraph becomes a memset
// node, and on this stack (ROCm 7.2, torch 2.10 graphs) such a node zeroes correctly on the first replay only: later
// replays were observed filling the range with a 16-byte pattern taken from a recycled argument block (the arguments of
// whatever kernel the host launched since; reproduced with a bare hipMemsetAsync in a captured stream once the host
// synchronises the device between replays: tools/graph_memset_probe.py).  A one-pair-per-call host synchronises
// after every replay, so the status word of mi_sinkhorn_dots came back non-zero ("solver timed out": every match
// invalid) and K1's ticket counters would start mid-range.  A kernel node keeps its arguments.  16-byte stores when
// the block allows, 4-byte stores otherwise.
template <typename W>
static __global__ __launch_bounds__(256) void mi_zero_kernel(W *__restrict__ p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = W{};
}
static inline int mi_zero_async(void *p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return MI_OK;
  if (((uintptr_t)p % 4) != 0 || bytes % 4 != 0) return MI_E_ALIGN;
  if (((uintptr_t)p % 16) == 0 && bytes % 16 == 0) {
    const size_t n = bytes / 16;
    hipLaunchKernelGGL(mi_zero_kernel<uint4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<uint4 *>(p), n);
  } else {
    const size_t n = bytes / 4;
    hipLaunchKernelGGL(mi_zero_kernel<uint32_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<uint32_t *>(p), n);
  }
  return mi_launch_status();
}

// Two equally shaped batches behind ONE launch (image1 / image2 of mi_match_pairs, and the keypoint arrays that belong
// to them): item i of n = 2 * per_set lives in `a` for i < per_set and in `b` otherwise.  A single batch is
// {ptr, nullptr, n}.  Which batch an item came from never changes what is computed for it.
struct MiSets {
  const void *a;
  const void *b;
  int per_set;
};
static inline MiSets mi_one_set(const void *p, int n) { return MiSets{p, nullptr, n}; }
template <typename T>
__device__ __forceinline__ const T *mi_set_item(const MiSets &s, int i, size_t item_elems) {
  const bool second = i >= s.per_set;
  return static_cast<const T *>(second ? s.b : s.a) + (size_t)(second ? i - s.per_set : i) * item_elems;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share
// an XCD and its private 4 MiB L2).  This bijection hands each XCD one CONTIGUOUS range of logical
// tile ids, so neighbouring tiles -- which share halo rows/columns -- are fetched through the same
// L2 instead of once per XCD.  Placement only changes speed, never results.
__device__ __forceinline__ unsigned xcd_contiguous_id(unsigned bid, unsigned nblocks) {
  const unsigned q = nblocks >> 3, r = nblocks & 7u;
  const unsigned xcd = bid & 7u, k = bid >> 3;
  const unsigned base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return base + k;
}

// Wave-wide reductions on the DPP path (VALU with lane-select modifiers: no LDS crossbar round
// trips as with __shfl/ds_bpermute).  quad_perm -> row_half_mirror -> row_mirror leave every
// lane of a 16-lane row with its row's result; row_bcast15 / row_bcast31 fold the four rows into
// lane 63, which v_readlane broadcasts as a scalar.  Returns the same value in every lane.
#define MI_DPP(old, src, ctrl, rmask) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, MI_DPP(v, v, 0xB1, 0xf));    // quad_perm [1,0,3,2]
  v = fmaxf(v, MI_DPP(v, v, 0x4E, 0xf));    // quad_perm [2,3,0,1]
  v = fmaxf(v, MI_DPP(v, v, 0x141, 0xf));   // row_half_mirror
  v = fmaxf(v, MI_DPP(v, v, 0x140, 0xf));   // row_mirror
  v = fmaxf(v, MI_DPP(v, v, 0x142, 0xa));   // row_bcast15 into rows 1 and 3
  v = fmaxf(v, MI_DPP(v, v, 0x143, 0xc));   // row_bcast31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += MI_DPP(0.0f, v, 0xB1, 0xf);
  v += MI_DPP(0.0f, v, 0x4E, 0xf);
  v += MI_DPP(0.0f, v, 0x141, 0xf);
  v += MI_DPP(0.0f, v, 0x140, 0xf);
  v += MI_DPP(0.0f, v, 0x142, 0xa);
  v += MI_DPP(0.0f, v, 0x143, 0xc);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Four wave-wide reductions at once (gfx950): v_permlane32_swap / v_permlane16_swap fold the four
// 64-lane vectors into ONE register whose 16-lane rows hold 16 partials of vectors 0, 2, 1, 3; four
// DPP steps finish inside the rows and four v_readlane broadcast the results.  ~18 instructions
// instead of 4 x 13 for four separate butterflies, and far fewer DPP hazard stalls.
#define MI_SWAP32(a, b, ra, rb)                                                                              \
  {                                                                                                           \
    auto _r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), \
                                               false, false);                                                \
    const unsigned _r0 = _r[0], _r1 = _r[1]; /* scalars first: bit_cast of a vector ELEMENT reads element 0 */ \
    ra = __builtin_bit_cast(float, _r0);                                                                      \
    rb = __builtin_bit_cast(float, _r1);                                                                      \
  }
#define MI_SWAP16(a, b, ra, rb)                                                                              \
  {                                                                                                           \
    auto _r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), \
                                               false, false);                                                \
    const unsigned _r0 = _r[0], _r1 = _r[1];                                                                  \
    ra = __builtin_bit_cast(float, _r0);                                                                      \
    rb = __builtin_bit_cast(float, _r1);                                                                      \
  }
__device__ __forceinline__ float mi_readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ void wave_max4(float *v) {     // v[0..3], registers
  float a0, a1, b0, b1, c0, c1;
  MI_SWAP32(v[0], v[1], a0, a1);                 // [v0.lo | v1.lo], [v0.hi | v1.hi]
  MI_SWAP32(v[2], v[3], b0, b1);
  const float m01 = fmaxf(a0, a1), m23 = fmaxf(b0, b1);   // halves: vector 0 | vector 1 ; vector 2 | vector 3
  MI_SWAP16(m01, m23, c0, c1);                   // rows: [0, 2, 1, 3] of each
  float m = fmaxf(c0, c1);
  m = fmaxf(m, MI_DPP(m, m, 0xB1, 0xf));         // quad_perm [1,0,3,2]
  m = fmaxf(m, MI_DPP(m, m, 0x4E, 0xf));         // quad_perm [2,3,0,1]
  m = fmaxf(m, MI_DPP(m, m, 0x141, 0xf));        // row_half_mirror
  m = fmaxf(m, MI_DPP(m, m, 0x140, 0xf));        // row_mirror
  v[0] = mi_readlane_f(m, 0);
  v[2] = mi_readlane_f(m, 16);
  v[1] = mi_readlane_f(m, 32);
  v[3] = mi_readlane_f(m, 48);
}
// The same reduction left in the lanes: every lane of 16-lane row g holds the total of vector MI_ROW_OF_GROUP(g)
// (no broadcast).  For per-row work that is itself done once per row: one lane-parallel instruction instead of four
// wave-uniform ones.
#define MI_ROW_OF_GROUP(g) ((g) == 1 ? 2 : (g) == 2 ? 1 : (g))     // rows [0, 2, 1, 3]; its own inverse
__device__ __forceinline__ float wave_sum4_rows(const float *v) {
  float a0, a1, b0, b1, c0, c1;
  MI_SWAP32(v[0], v[1], a0, a1);
  MI_SWAP32(v[2], v[3], b0, b1);
  const float m01 = a0 + a1, m23 = b0 + b1;
  MI_SWAP16(m01, m23, c0, c1);
  float m = c0 + c1;
  m += MI_DPP(0.0f, m, 0xB1, 0xf);
  m += MI_DPP(0.0f, m, 0x4E, 0xf);
  m += MI_DPP(0.0f, m, 0x141, 0xf);
  m += MI_DPP(0.0f, m, 0x140, 0xf);
  return m;
}
__device__ __forceinline__ void wave_sum4(float *v) {
  float a0, a1, b0, b1, c0, c1;
  MI_SWAP32(v[0], v[1], a0, a1);
  MI_SWAP32(v[2], v[3], b0, b1);
  const float m01 = a0 + a1, m23 = b0 + b1;
  MI_SWAP16(m01, m23, c0, c1);
  float m = c0 + c1;
  m += MI_DPP(0.0f, m, 0xB1, 0xf);
  m += MI_DPP(0.0f, m, 0x4E, 0xf);
  m += MI_DPP(0.0f, m, 0x141, 0xf);
  m += MI_DPP(0.0f, m, 0x140, 0xf);
  v[0] = mi_readlane_f(m, 0);
  v[2] = mi_readlane_f(m, 16);
  v[1] = mi_readlane_f(m, 32);
  v[3] = mi_readlane_f(m, 48);
}

// P = exp(log P) of the final assignment matrix (reference matching/sinkhorn.py:145,206), ONE definition
// shared by K6's final pass and K7's matches-from-duals so that both produce the same bits:
// v_exp_f32(t * log2 e).  Relative error <= 1 ulp + |t| * 2^-24 (the rounding of the scaled argument),
// i.e. < 1e-6 for every |t| that yields a representable P -- two orders inside the 1e-4 parity bound --
// at 3 instructions instead of libm's ~20.
__device__ __forceinline__ float mi_prob_exp(float t) { return __builtin_amdgcn_exp2f(t * 1.4426950408889634f); }

// log-score of the packed-descriptor (uint16 dot product) form, shared by K6 and K7 so that both
// rebuild the same bits: z = nie*(na + nb) + dot*sb*(-2*nie*sa), nie = -1/epsilon, (scale, squared
// norm) pairs per descriptor (reference matching/sinkhorn.py:101-103,178).  The cost's clamp at 0 only acts on
// the rounding noise of identical descriptors and is dropped: that noise (<= ~3e-7 for unit descriptors, exactly
// 0 for unnormalised bit vectors) enters z divided by epsilon, which is why the entry points of this form refuse
// epsilon < MI_DOTS_MIN_EPSILON (the fp32-Z form clamps).
__device__ __forceinline__ float mi_z_from_dot(float dot, float2 row, float2 col, float neg_inv_eps) {
  return __builtin_fmaf(dot * col.x, -2.0f * neg_inv_eps * row.x, col.y * neg_inv_eps) + row.y * neg_inv_eps;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// The fused AKAZE scale kernels divide through akaze_math.h's exact-rounding helpers, verified exhaustively for
// kappa in [1e-3, 3] and image gradients of the uint8 range (1 + (|g| / kappa)^2 < 2^40); outside
// [MI_AKAZE_KAPPA_MIN, MI_AKAZE_KAPPA_MAX] 1 / kappa or |g| / kappa can overflow or leave the verified range, so
// mi_akaze_scale / mi_akaze_scale_select refuse (MI_E_PARAM); mi_akaze_diffuse (IEEE operators) takes any kappa > 0.
static inline bool mi_akaze_kappa_ok(float kappa) { return kappa >= MI_AKAZE_KAPPA_MIN && kappa <= MI_AKAZE_KAPPA_MAX; }
// akaze_stream.hip: the rolling-window form of one AKAZE scale (mode 0: this scale's score map; mode 1: the selection
// across scales); mi_akaze_stream_supported says whether it applies (even width, 8-byte aligned maps, nms_size 3 / 5,
// iterations 1..3)
#define MI_AKAZE_STREAM_MAX_PREV 3
int mi_akaze_stream_supported(int h, int w, int iterations, int nms_size, const void *l_in, const void *l_out,
                              const void *scores);
int mi_akaze_scale_stream(const float *l_in, const float *l_in_b, int per_set, int n, int h, int w, int iterations, float kappa, float dt, float threshold,
                          int nms_size, float *l_out, float *scores, int mode, const float *prev_scores, int num_prev,
                          uint8_t *attain, mi_stream_t stream);

// ---- internal launchers shared between translation units (not part of the C ABI): the entry points of
// include/mi355x_match.h with `MiSets` in place of a single batch pointer.  mi_match_pairs uses them to put both images
// of every pair behind one launch per stage when the batch is small (the one-pair-per-call latency path).
int mi_corner_response_sets(MiSets images, int pix_u8, int n, int h, int w, int block_size, float *score,
                            unsigned *tile_ctr, mi_stream_t stream);
int mi_topk_keypoints_sets(const uint64_t *cand, const uint32_t *count, int segments, int segment_capacity, int n, int w,
                           int k, MiSets keypoints, float *kscores, mi_stream_t stream);
int mi_sparse_bad_sets(MiSets images, int pix_u8, int n, int h, int w, MiSets keypoints, int k, const uint32_t *pair_geom,
                       const float *pair_thr, int num_pairs, int mode, float temperature, int normalize, float *desc,
                       uint32_t *bits, const void *plan, uint8_t *status, mi_stream_t stream);
int mi_cost_dots_bits_zeroing(const uint32_t *bits1, const uint32_t *bits2, int batch, int n, int m, int num_bits,
                              int normalized, uint16_t *dots, int pitch, float *row_info, float *col_info,
                              void *zero_ptr, size_t zero_bytes, mi_stream_t stream);
// mi_sinkhorn_dots' single-launch form polls tagged granules that must start out zero: by default it clears them with
// a zeroing kernel of its own (mi_zero_async above; never hipMemsetAsync); a caller that has them cleared by an earlier kernel of the same call (the region this
// returns; 0 bytes when the multi-launch form will run) passes prezeroed = 1.
size_t mi_sinkhorn_dots_handoff_region(void *workspace, int batch, int n, int m, int flags, void **region);
int mi_sinkhorn_dots_impl(const uint16_t *dots, const float *row_info, const float *col_info, int batch, int n, int m,
                          int pitch, double epsilon, double unused_score, double sqnorm_bound, int iterations, float *u,
                          float *v, float *p, void *workspace, size_t workspace_bytes, int flags, int prezeroed,
                          mi_stream_t stream);

// A >= 64-pair solve as two half batches on two streams (sinkhorn_dots.hip: the per-stream helper streams, events and
// schedule tuner), for solvers outside that file: begin -> enqueue part q on f.stream[q] -> end.
struct MiFork {
  int parts;                 // 1: everything on the caller's stream; 2: two half batches
  hipStream_t stream[2];
  void *handle;
  int trial_entry, trial, first_side;
};
int mi_fork_begin(hipStream_t s, int batch, int n, int m, int key, MiFork *f);
int mi_fork_end(hipStream_t s, MiFork *f);

