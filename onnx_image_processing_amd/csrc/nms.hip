// K2: square-window NMS (max-pool semantics) + candidate compaction.
// Semantics: reference pytorch_model/utils/keypoint_utils.py:12-44 (mask) and :71-92 (masking,
// border, threshold).  HBM traffic: 4 B/pixel read; candidates are ~1 % of pixels.
//
// One workgroup (512 threads on the fast path, 256 on the generic one) owns a 128x32 tile.  The score tile (+r halo, -inf outside the
// image) is staged in LDS, a separable max (row pass in place on the fast path, into a second LDS plane on
// the generic one; column pass in registers) gives the (2r+1)^2 window maximum.  Survivors are packed into 64-bit keys
// (score bits high, inverted linear index low) and written to the tile's OWN segment of the
// candidate buffer (4096 slots = every pixel of the tile, so it cannot overflow) with the
// count in count[img][tile]: no global atomics, no pre-zeroed counters, and the later top-k
// sort has a total order, so the result does not depend on the order inside a segment.
#include "common.h"

#include <math.h>

namespace {

constexpr int NT_W = 128, NT_H = 32;
constexpr int SEG_CAP = NT_W * NT_H;  // slots per tile segment
constexpr int NMS_THREADS = 512;      // fast kernel: 8 waves per 128x32 tile, 24 KB of LDS (4 workgroups = 32 waves per CU)

__device__ __forceinline__ uint64_t make_key(float m, uint32_t lin) {
  return ((uint64_t)__float_as_uint(m) << 32) | (uint64_t)(0xFFFFFFFFu - lin);
}

// v_max_f32 / v_max3_f32 exactly as written.  fmaxf() makes the compiler first quiet a possible signalling
// NaN in every value that comes from memory (one `v_max x, x` each: +45 % instructions in the two maximum
// passes, which are bound by VALU issue); the instructions already return the other operand for a NaN.
__device__ __forceinline__ float vmax2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// max(acc, e[0..N)) with two new operands per instruction
template <int N>
__device__ __forceinline__ float vmax_fold(float acc, const float *e) {
#pragma unroll
  for (int i = 0; i + 1 < N; i += 2) acc = vmax3(acc, e[i], e[i + 1]);
  if constexpr (N & 1) acc = vmax2(acc, e[N - 1]);
  return acc;
}

// Workgroup-wide compaction of up to NQ survivors per thread into one segment.
// kidx[q] == 0xFFFFFFFF marks "not a survivor".  Must be called by all NWV * 64 threads.
template <int NQ, int NWV>
__device__ __forceinline__ void compact_tile(uint32_t nkeep, const float (&kval)[NQ], const uint32_t (&kidx)[NQ],
                                             uint64_t *__restrict__ seg, uint32_t *__restrict__ seg_count) {
  __shared__ uint32_t wave_total[NWV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  (void)nkeep;
  // slot of survivor q of this lane inside the wave = survivors of rounds < q + survivors of round q in
  // lower lanes: one ballot and one mbcnt per round, counts carried in SGPRs (a 6-step __shfl_up scan
  // would be six dependent trips through the LDS crossbar at the very end of the kernel)
  uint32_t wave_cnt = 0;
  uint32_t slot_in_wave[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const unsigned long long m = __ballot(kidx[q] != 0xFFFFFFFFu);
    slot_in_wave[q] = wave_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    wave_cnt += (uint32_t)__popcll(m);
  }
  if (lane == 0) wave_total[wave] = wave_cnt;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NWV; ++w) {
    const uint32_t c = wave_total[w];
    base += (w < wave) ? c : 0u;
    total += c;
  }
  if (threadIdx.x == 0) *seg_count = total;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (kidx[q] != 0xFFFFFFFFu) seg[base + slot_in_wave[q]] = make_key(kval[q], kidx[q]);
  }
}

// Same, from lane predicates: the predicates live in SGPR pairs (their ANDs run on the scalar unit and a
// ballot is free), and everything a survivor needs -- slot, linear index, key -- is computed inside the
// branch only survivors take.  lin_of(q) gives survivor q's linear pixel index.
template <int NQ, int NWV, class LinOf>
__device__ __forceinline__ void compact_tile_pred(const bool (&keep)[NQ], const float (&kval)[NQ], LinOf lin_of,
                                                  uint64_t *__restrict__ seg, uint32_t *__restrict__ seg_count) {
  __shared__ uint32_t wave_total[NWV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long m[NQ];
  uint32_t before[NQ];                         // survivors of earlier rounds in this wave (wave-uniform)
  uint32_t wave_cnt = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    m[q] = __ballot(keep[q]);
    before[q] = wave_cnt;
    wave_cnt += (uint32_t)__popcll(m[q]);
  }
  if (lane == 0) wave_total[wave] = wave_cnt;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NWV; ++w) {
    const uint32_t c = wave_total[w];
    base += (w < wave) ? c : 0u;
    total += c;
  }
  if (threadIdx.x == 0) *seg_count = total;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (keep[q]) {
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[q] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[q], 0u));
      seg[base + before[q] + below] = make_key(kval[q], lin_of(q));
    }
  }
}

__device__ __forceinline__ bool in_border(int gy, int gx, int h, int w, int margin) {
  return (margin <= 0) || (gy >= margin && gy < h - margin && gx >= margin && gx < w - margin);
}

// ---- generic path: any width, radius up to 24 (dynamic LDS) ---------------------------------
// MODE 0: write the float mask.  MODE 1: border + threshold + compaction (mask not stored).
template <int MODE>
__global__ __launch_bounds__(256) void nms_kernel(const float *__restrict__ score, int h, int w, int r,
                                                  int tiles_x, int tiles_y, float *__restrict__ mask,
                                                  float thr_eff, int margin, uint64_t *__restrict__ cand,
                                                  uint32_t *__restrict__ count) {
  extern __shared__ float lds[];
  const int sw = NT_W + 2 * r;       // staged width
  const int sh = NT_H + 2 * r;       // staged height
  float *tile = lds;                 // [sh][sw]
  float *rmax = lds + sh * sw;       // [sh][NT_W] horizontal window maxima

  const int t = threadIdx.x;
  const int seg_id = blockIdx.x;
  int bid = blockIdx.x;
  const int tx_tile = bid % tiles_x;
  bid /= tiles_x;
  const int ty_tile = bid % tiles_y;
  const int img = bid / tiles_y;
  const int x0 = tx_tile * NT_W, y0 = ty_tile * NT_H;
  const float *sc = score + (size_t)img * h * w;

  for (int i = t; i < sh * sw; i += 256) {
    const int rr = i / sw, cc = i - rr * sw;
    const int gy = y0 - r + rr, gx = x0 - r + cc;
    tile[i] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? sc[(size_t)gy * w + gx] : -INFINITY;
  }
  __syncthreads();
  for (int i = t; i < sh * NT_W; i += 256) {
    const int rr = i / NT_W, cc = i - rr * NT_W;
    const float *p = tile + rr * sw + cc;
    float m = p[0];
    for (int d = 1; d <= 2 * r; ++d) m = fmaxf(m, p[d]);
    rmax[i] = m;
  }
  __syncthreads();

  const int cx = t & (NT_W - 1);     // column inside the tile
  const int half = t >> 7;           // rows [half*16, half*16+16)
  const int gx = x0 + cx;
  uint32_t nkeep = 0;
  float kval[16];
  uint32_t kidx[16];
#pragma unroll
  for (int k = 0; k < NT_H / 2; ++k) {
    const int ly = half * (NT_H / 2) + k;
    const int gy = y0 + ly;
    const bool inside = (gx < w) && (gy < h);
    float m = -INFINITY, s = 0.f;
    if (inside) {
      const float *p = rmax + ly * NT_W + cx;
      m = p[0];
      for (int d = 1; d <= 2 * r; ++d) m = fmaxf(m, p[d * NT_W]);
      s = tile[(ly + r) * sw + cx + r];
    }
    const bool is_max = inside && (s >= (m - 1e-7f));          // keypoint_utils.py:43
    if (MODE == 0) {
      if (inside) mask[((size_t)img * h + gy) * w + gx] = is_max ? 1.0f : 0.0f;
    } else {
      const bool keep = is_max && in_border(gy, gx, h, w, margin) && (s > thr_eff);
      kval[k] = s;
      kidx[k] = keep ? (uint32_t)(gy * w + gx) : 0xFFFFFFFFu;
      nkeep += keep ? 1u : 0u;
    }
  }
  if (MODE == 1) compact_tile<16, 4>(nkeep, kval, kidx, cand + (size_t)seg_id * SEG_CAP, count + seg_id);
}

// Explicit-mask form (the reference's separate select_topk_keypoints call), same tiling.
__global__ __launch_bounds__(256) void select_kernel(const float *__restrict__ score,
                                                     const float *__restrict__ mask, int h, int w,
                                                     int tiles_x, int tiles_y, float thr, int margin,
                                                     uint64_t *__restrict__ cand, uint32_t *__restrict__ count) {
  const int t = threadIdx.x;
  const int seg_id = blockIdx.x;
  int bid = blockIdx.x;
  const int tx_tile = bid % tiles_x;
  bid /= tiles_x;
  const int ty_tile = bid % tiles_y;
  const int img = bid / tiles_y;
  const int gx = tx_tile * NT_W + (t & (NT_W - 1));
  const int half = t >> 7;
  uint32_t nkeep = 0;
  float kval[16];
  uint32_t kidx[16];
#pragma unroll
  for (int k = 0; k < NT_H / 2; ++k) {
    const int gy = ty_tile * NT_H + half * (NT_H / 2) + k;
    bool keep = false;
    float m = 0.f;
    if (gx < w && gy < h) {
      const size_t i = ((size_t)img * h + gy) * w + gx;
      m = score[i] * mask[i];
      if (margin > 0) m = m * (in_border(gy, gx, h, w, margin) ? 1.0f : 0.0f);
      keep = (m > thr) && (m > 0.0f);                          // keypoint_utils.py:88-92, then "> 0" at :108
    }
    kval[k] = m;
    kidx[k] = keep ? (uint32_t)(gy * w + gx) : 0xFFFFFFFFu;
    nkeep += keep ? 1u : 0u;
  }
  compact_tile<16, 4>(nkeep, kval, kidx, cand + (size_t)seg_id * SEG_CAP, count + seg_id);
}

// ---- fast path: w % 4 == 0, radius 1..8 -----------------------------------------------------
// Same tile, but every stage moves float4: staging issues all of a thread's 16-byte global
// loads before the first LDS store (loads in flight, not one dependent round trip per element),
// the row pass forms 4 adjacent window maxima from 12/20 registers sharing the common core of the
// four windows, the column pass does the same down 4+2R rows.
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

template <int MODE, int R, int NTH>
__global__ __launch_bounds__(NTH) void nms_fast_kernel(const float *__restrict__ score, int h, int w,
                                                       int tiles_x, int tiles_y, float *__restrict__ mask,
                                                       float thr_eff, int margin,
                                                       uint64_t *__restrict__ cand, uint32_t *__restrict__ count) {
  constexpr int PC = (R + 3) / 4;              // padding chunks each side
  constexpr int AW4 = NT_W / 4 + 2 * PC;       // staged row, in float4
  constexpr int LH = NT_H + 2 * R;             // staged rows
  constexpr int NCH = (LH * AW4 + NTH - 1) / NTH;            // staging chunks per thread
  constexpr int NRP = (LH * (NT_W / 4) + NTH - 1) / NTH;     // row-pass items per thread
  constexpr int NV = 4 * (1 + 2 * PC);         // floats a thread reads per row in the row pass
  constexpr int B0 = 4 * PC;                   // index of output column 0 inside those floats
  constexpr int RPT = NT_H / (NTH / 32);       // rows per thread in the column pass (4 or 2)
  static_assert(2 * R >= RPT - 1, "the RPT windows of a thread must share rows");
  // One LDS plane: scores (+halo, -inf outside the image); the row pass overwrites its interior columns with
  // the horizontal window maxima IN PLACE.  That is safe because a wave's 64 row-pass items are two whole rows
  // (32 column groups each): every read of a row is issued by the wave that later writes it, and a wave's LDS
  // operations execute in order.  The centre pixels the column pass compares against are fetched from global
  // memory (L2 hits) up front instead.  24 KB per workgroup instead of 46: 4 workgroups (32 waves) per CU.
  __shared__ float4 pa[LH][AW4];
  static_assert(NT_W / 4 == 32, "a wave's row-pass items must be whole rows");

  const int t = threadIdx.x;
  const int seg_id = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);   // neighbouring tiles share an XCD's L2
  int bid = seg_id;
  const int tx_tile = bid % tiles_x;
  bid /= tiles_x;
  const int ty_tile = bid % tiles_y;
  const int img = bid / tiles_y;
  const int x0 = tx_tile * NT_W, y0 = ty_tile * NT_H;
  const float *sc = score + (size_t)img * h * w;
  const float ninf = -INFINITY;
  const int tx = t & 31, ty = t >> 5;          // column-pass coordinates: 4 columns x RPT rows
  const int gx = x0 + 4 * tx;
  float4 centre[RPT];

  {
    float4 v[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int i = t + q * NTH;
      const int rr = i / AW4, cc = i - rr * AW4;
      const int gy = y0 - R + rr, gx = x0 - 4 * PC + 4 * cc;
      v[q] = make_float4(ninf, ninf, ninf, ninf);
      if (i < LH * AW4 && gy >= 0 && gy < h && gx >= 0 && gx < w)
        v[q] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc) + (uint32_t)(gy * w + gx) * 4u);   // scalar base + 32-bit offset
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int gy = y0 + ty * RPT + k;
      centre[k] = make_float4(ninf, ninf, ninf, ninf);
      if (gx < w && gy < h)
        centre[k] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc) + (uint32_t)(gy * w + gx) * 4u);
    }
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int i = t + q * NTH;
      if (i < LH * AW4) (&pa[0][0])[i] = v[q];
    }
  }
  __syncthreads();

  // row pass: LH rows x 32 column groups
#pragma unroll
  for (int it = 0; it < NRP; ++it) {
    const int i = t + it * NTH;
    if (i >= LH * (NT_W / 4)) break;
    const int rr = i >> 5, cg = i & 31;
    float v[NV];
    {
      // keep the 16-byte reads whole: left alone the compiler fetches only the 4+2r floats it needs with
      // ds_read2_b32, whose 16-byte lane stride is a 4-way bank conflict; ds_read_b128 has none.  The empty asm that
      // pins a chunk follows ALL of the item's reads (round 3): pinned one by one, every read was waited for before
      // the next was issued -- NV / 4 dependent LDS round trips per item (s_waitcnt lgkmcnt(0) after each in the .s).
      typedef float f4v __attribute__((ext_vector_type(4)));
      f4v q[NV / 4];
#pragma unroll
      for (int c = 0; c < NV / 4; ++c) q[c] = *reinterpret_cast<const f4v *>(&pa[rr][cg + c]);
#pragma unroll
      for (int c = 0; c < NV / 4; ++c) asm volatile("" : "+v"(q[c]));
#pragma unroll
      for (int c = 0; c < NV / 4; ++c) { v[4 * c] = q[c].x; v[4 * c + 1] = q[c].y; v[4 * c + 2] = q[c].z; v[4 * c + 3] = q[c].w; }
    }
    // windows [B0+o-R, B0+o+R], o = 0..3; for R >= 2 they share the core [B0+3-R, B0+R]
    float o[4];
    if constexpr (R >= 2) {
      const float core = vmax_fold<2 * R - 3>(v[B0 + 3 - R], &v[B0 + 4 - R]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float ext[3];                          // 3 - k values left of the core, k right of it
#pragma unroll
        for (int e = 0; e < 3; ++e) ext[e] = (e < 3 - k) ? v[B0 + k - R + e] : v[B0 + R + 1 + (e - (3 - k))];
        o[k] = vmax_fold<3>(core, ext);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = vmax3(v[B0 + k - 1], v[B0 + k], v[B0 + k + 1]);
    }
    pa[rr][cg + PC] = make_float4(o[0], o[1], o[2], o[3]);      // in place (see above)
  }
  __syncthreads();

  // column pass: thread = 4 columns x RPT rows; window k covers rows [k, k + 2R], all sharing [RPT-1, 2R]
  float rc[4][RPT + 2 * R];                    // [column][row]
#pragma unroll
  for (int q = 0; q < RPT + 2 * R; ++q) {
    const float4 rq = pa[ty * RPT + q][tx + PC];
    rc[0][q] = rq.x; rc[1][q] = rq.y; rc[2][q] = rq.z; rc[3][q] = rq.w;
  }
  float core[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) core[c] = vmax_fold<2 * R - RPT + 1>(rc[c][RPT - 1], &rc[c][RPT]);
  float kval[RPT * 4];
  bool keep[RPT * 4];
  // image / border tests per column and per row of the thread, not per pixel (w % 4 == 0: a float4 chunk is
  // wholly inside or outside the image)
  bool col_ok[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) col_ok[c] = (gx < w) && (margin <= 0 || (gx + c >= margin && gx + c < w - margin));
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    float mv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float ext[RPT > 1 ? RPT - 1 : 1];        // RPT-1-k rows above the core, k below it
#pragma unroll
      for (int e = 0; e < RPT - 1; ++e) ext[e] = (e < RPT - 1 - k) ? rc[c][k + e] : rc[c][2 * R + 1 + (e - (RPT - 1 - k))];
      mv[c] = vmax_fold<RPT - 1>(core[c], ext);
    }
    const int ly = ty * RPT + k, gy = y0 + ly;
    const float4 s = centre[k];
    const bool in_img = (gx < w) && (gy < h);
    const bool row_ok = (gy < h) && (margin <= 0 || (gy >= margin && gy < h - margin));
    const float sv[4] = {s.x, s.y, s.z, s.w};
    float outv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool is_max = sv[c] >= (mv[c] - 1e-7f);                      // keypoint_utils.py:43
      outv[c] = is_max ? 1.0f : 0.0f;
      if (MODE == 1) {
        keep[k * 4 + c] = is_max && (sv[c] > thr_eff) && col_ok[c] && row_ok;
        kval[k * 4 + c] = sv[c];
      }
    }
    if (MODE == 0 && in_img)
      *reinterpret_cast<float4 *>(mask + ((size_t)img * h + gy) * w + gx) = make_float4(outv[0], outv[1], outv[2], outv[3]);
  }
  if (MODE == 1) {
    const uint32_t lin0 = (uint32_t)((y0 + ty * RPT) * w + gx);
    compact_tile_pred<RPT * 4, NTH / 64>(keep, kval, [&](int q) { return lin0 + (uint32_t)((q >> 2) * w + (q & 3)); },
                                         cand + (size_t)seg_id * SEG_CAP, count + seg_id);
  }
}

template <int MODE>
bool launch_fast(const float *score, int n, int h, int w, int radius, float *mask, float thr_eff, int margin,
                 uint64_t *cand, uint32_t *count, hipStream_t s) {
  if (w % 4 != 0 || radius < 1 || radius > 8 || ((uintptr_t)score % 16) != 0) return false;
  if ((long long)h * w >= (1LL << 30)) return false;        // the fast kernel addresses a plane with 32-bit byte offsets
  if (MODE == 0 && ((uintptr_t)mask % 16) != 0) return false;
  const int tiles_x = ceil_div(w, NT_W), tiles_y = ceil_div(h, NT_H);
  const dim3 grid((unsigned)(n * tiles_x * tiles_y));
#define MI_NMS_CASE(RR)                                                                                      \
  case RR:                                                                                                   \
    hipLaunchKernelGGL((nms_fast_kernel<MODE, RR, NMS_THREADS>), grid, dim3(NMS_THREADS), 0, s, score, h, w,   \
                       tiles_x, tiles_y, mask, thr_eff, margin, cand, count);                                     \
    break;
  switch (radius) {
    MI_NMS_CASE(1) MI_NMS_CASE(2) MI_NMS_CASE(3) MI_NMS_CASE(4)
    MI_NMS_CASE(5) MI_NMS_CASE(6) MI_NMS_CASE(7) MI_NMS_CASE(8)
  }
#undef MI_NMS_CASE
  return true;
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return MI_OK;
  if (bytes > 160 * 1024) return MI_E_PARAM;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? MI_OK : (int)e;
}

int check_common(const void *a, const void *b, int n, int h, int w) {
  if (!a || !b) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0 || (long long)h * w > 0x7fffffffLL) return MI_E_SHAPE;
  if ((long long)n * ceil_div(w, NT_W) * ceil_div(h, NT_H) > 0x7fffffffLL) return MI_E_SHAPE;
  return MI_OK;
}

size_t generic_lds(int radius) {
  return ((size_t)(NT_H + 2 * radius) * (NT_W + 2 * radius) + (size_t)(NT_H + 2 * radius) * NT_W) * 4;
}

}  // namespace

extern "C" int mi_candidate_layout(int h, int w, int *segments, int *segment_capacity) {
  if (h <= 0 || w <= 0 || !segments || !segment_capacity) return MI_E_SHAPE;
  *segments = ceil_div(w, NT_W) * ceil_div(h, NT_H);
  *segment_capacity = SEG_CAP;
  return MI_OK;
}

extern "C" int mi_nms_mask(const float *score, int n, int h, int w, int radius, float *mask,
                           mi_stream_t stream) {
  MI_ENTER();
  int e = check_common(score, mask, n, h, w);
  if (e) return e;
  if (radius < 0 || radius > 24) return MI_E_PARAM;
  if (launch_fast<0>(score, n, h, w, radius, mask, 0.f, 0, nullptr, nullptr, (hipStream_t)stream))
    return mi_launch_status();
  const int tiles_x = ceil_div(w, NT_W), tiles_y = ceil_div(h, NT_H);
  const size_t lds = generic_lds(radius);
  if ((e = allow_lds(nms_kernel<0>, lds)) != MI_OK) return e;
  hipLaunchKernelGGL(nms_kernel<0>, dim3((unsigned)(n * tiles_x * tiles_y)), dim3(256), lds,
                     (hipStream_t)stream, score, h, w, radius, tiles_x, tiles_y, mask, 0.f, 0, nullptr, nullptr);
  return mi_launch_status();
}

extern "C" int mi_nms_candidates(const float *score, int n, int h, int w, int radius, float score_threshold,
                                 int border_margin, uint64_t *cand, uint32_t *count, mi_stream_t stream) {
  MI_ENTER();
  int e = check_common(score, cand, n, h, w);
  if (e) return e;
  if (!count) return MI_E_NULL;
  if (radius < 0 || radius > 24) return MI_E_PARAM;
  const float thr_eff = score_threshold > 0.f ? score_threshold : 0.f;
  if (launch_fast<1>(score, n, h, w, radius, nullptr, thr_eff, border_margin, cand, count, (hipStream_t)stream))
    return mi_launch_status();
  const int tiles_x = ceil_div(w, NT_W), tiles_y = ceil_div(h, NT_H);
  const size_t lds = generic_lds(radius);
  if ((e = allow_lds(nms_kernel<1>, lds)) != MI_OK) return e;
  hipLaunchKernelGGL(nms_kernel<1>, dim3((unsigned)(n * tiles_x * tiles_y)), dim3(256), lds,
                     (hipStream_t)stream, score, h, w, radius, tiles_x, tiles_y, nullptr, thr_eff, border_margin,
                     cand, count);
  return mi_launch_status();
}

extern "C" int mi_select_candidates(const float *score, const float *mask, int n, int h, int w,
                                    float score_threshold, int border_margin, uint64_t *cand,
                                    uint32_t *count, mi_stream_t stream) {
  MI_ENTER();
  int e = check_common(score, mask, n, h, w);
  if (e) return e;
  if (!cand || !count) return MI_E_NULL;
  const int tiles_x = ceil_div(w, NT_W), tiles_y = ceil_div(h, NT_H);
  hipLaunchKernelGGL(select_kernel, dim3((unsigned)(n * tiles_x * tiles_y)), dim3(256), 0, (hipStream_t)stream,
                     score, mask, h, w, tiles_x, tiles_y, score_threshold, border_margin, cand, count);
  return mi_launch_status();
}
