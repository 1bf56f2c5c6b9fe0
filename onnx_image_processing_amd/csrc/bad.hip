// K4: sparse BAD (Box Average Difference) descriptors at keypoints.
// Semantics: reference pytorch_model/descriptor/bad.py:436-576, non-oriented branch,
// sampling_mode="nearest".  The reference builds a dense 8-channel 15x15 box-mean bank over
// the whole image and gathers 2*8*K*P samples from it; here nothing dense is built.
//
// One wave (64 lanes) per keypoint.  Every box of every pair lies inside a small window of the
// replicate-extended image around the keypoint, so the wave stages that window, turns it into a
// summed-area table in LDS and evaluates each box with four LDS reads.  Lane l owns pairs
// l, l+64, ...; a wave ballot turns 64 sign tests into one 64-bit word of the packed
// descriptor, and its popcount gives the L2 norm of the bit vector exactly.
//
// Two kernels:
//  * bad_fast_kernel (hard bits, needs a plan): keypoints with integer coordinates, >= 15 px from
//    the border, on an integer-valued (uint8) patch.  Then no box centre is clamped, every box
//    lies in the 32x32 window [k-16, k+15], the four table corners of each box are the same for
//    every keypoint (precomputed byte offsets), sums are exact int32 and the sign test is
//    D <= floor(thr*area).  4 keypoints per 256-thread workgroup, wave-private LDS (4.5 KiB per
//    wave) and no workgroup barrier, so occupancy is high.  Keypoints it cannot take are flagged.
//  * sparse_bad_kernel (general): 34x34 window, fp64 table (exact for integer images, ~1e-13
//    otherwise), box centres through the exact grid_sample(nearest, border, align_corners)
//    arithmetic, all output modes.  With a status array it only visits the flagged keypoints.
// Both are exact, so which one handled a keypoint cannot be seen in the result.
//
// Window proof (general path): table offsets satisfy -16 <= o - r, o + r <= 15 (boxes stay inside
// the 32x32 patch).  With f = floor(ky), the unclamped centre c = nearbyint(ky + o) lies in
// [f + o, f + o + 1]; clamping c into the image and then taking rows c-r..c+r of the
// replicate-extended image never leaves [f - 16, f + 17].
#include "common.h"

#include <math.h>

#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>

#include "bad_plan_opt.h"

namespace {

constexpr int WIN = 34;          // general window edge
constexpr int WOFF = 16;         // window origin = floor(k) - WOFF
constexpr int SP = WIN + 1;      // general SAT edge (leading zero row/column)
constexpr int FR = 33;           // fast-path SAT rows / columns in use (32x32 window + leading zero row/column)
constexpr int FW = 35;           // its row pitch: odd (the table build walks rows and columns conflict-free) and, of
                                 // 33..39, the pitch with the fewest bank-conflict passes of the pair gathers (-8 % vs 33)

// Device-resident plan for the fast path, built once per pair table by mi_bad_plan_build.  Arrays indexed by
// EXECUTION slot e = round * 64 + lane (bad_plan_opt.h: lane l evaluates its pairs l, l + 64, ... in the order that
// minimises LDS bank conflicts): header, then uint4 offs[P] (eight 16-bit BYTE offsets into the int32 table of 33 rows,
// pitch FW: the four corners that enter s1 - s2 with + in x, y, the four with - in z, w, low half first, in the
// scheduled order), int tint[P] = floor(thr * area), uint32 geom[P] (the pair's table word, for the border branch), then
// uint64 masks[rounds][rounds]: masks[g][G] = the lanes whose round-g pair belongs to the canonical 64-pair group G.
struct BadPlan {
  int geometry_ok;   // every box of the table stays inside the 32x32 patch
  int num_pairs;
  int passes;        // LDS passes per keypoint of the scheduled gathers (num_pairs / 4 = conflict-free)
  int passes_canonical;
};

// ATen grid_sampler semantics used by bad.py:518-556 (align_corners=True, padding "border",
// mode "nearest"): normalise with fp32(2/(size-1+1e-8)), un-normalise, clip, round half even.
__device__ __forceinline__ int nearest_centre(float pos, float scale, int size) {
  const float g = pos * scale - 1.0f;
  float x = ((g + 1.0f) / 2.0f) * (float)(size - 1);
  x = fminf(fmaxf(x, 0.0f), (float)(size - 1));
  return (int)nearbyintf(x);
}

// ---- fast kernel: 4 waves = 4 keypoints per workgroup, no workgroup-level synchronisation -----
// waves_per_eu(8, 8): 41 VGPRs instead of 102 -- the kernel waits on its window gather (58 % of wave cycles
// parked on s_waitcnt), so twice the resident waves, i.e. loads in flight, buys 12 %
// GROUPS = num_pairs / 64 at compile time (0: read it at run time).  With a constant trip count the pair
// loop is straight-line code and the compiler hoists the pair-table loads of later groups above the box
// gathers of earlier ones; with a run-time count every group paid an L1/L2 round trip where it was used.
// PIX = float (the reference's input) or uint8_t (the u8 ingest path: same window, a quarter of the bytes, no
// integrality test -- every uint8 patch is integer-valued).
template <int GROUPS, typename PIX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void bad_fast_kernel(MiSets images, int h, int w,
                                                       MiSets kpt_sets, int k, int total,
                                                       int num_pairs, int normalize,
                                                       const BadPlan *__restrict__ plan,
                                                       const uint32_t *__restrict__ geom,
                                                       float *__restrict__ desc, uint32_t *__restrict__ bits,
                                                       uint8_t *__restrict__ status) {
  __shared__ int sat_all[4][FR * FW];
  // everything per keypoint is wave-uniform; readfirstlane tells the compiler (scalar branches and addresses)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-contiguous order: the keypoints of one image run on one XCD, so their overlapping windows
  // are fetched through one L2 instead of once per XCD
  const int flat = (int)xcd_contiguous_id(blockIdx.x, gridDim.x) * 4 + wave;
  if (flat >= total) return;
  const int img = flat / k;
  const PIX *im = mi_set_item<PIX>(images, img, (size_t)h * w);
  constexpr uint32_t PB = (uint32_t)sizeof(PIX);
  const float *kp = mi_set_item<float>(kpt_sets, img, (size_t)k * 2) + (size_t)(flat - img * k) * 2;
  const float ky = kp[0];
  const float kx = kp[1];
  const int groups = GROUPS ? GROUPS : num_pairs / 64;
  const int words = 2 * groups;
  constexpr int GMAX = GROUPS ? GROUPS : 16;

  if (!plan->geometry_ok) {                                    // a plan whose geometry check failed must not be used:
    if (lane == 0) status[flat] = 0;                           // hand the keypoint to the general kernel
    return;
  }
  if (!(ky >= 0.0f)) {                                         // invalid keypoint: zero descriptor (bad.py:461,570)
    if (bits) for (int q = lane; q < words; q += 64) bits[(size_t)flat * words + q] = 0u;
    if (desc) for (int q = lane; q < num_pairs; q += 64) desc[(size_t)flat * num_pairs + q] = 0.0f;
    if (lane == 0) status[flat] = 1;
    return;
  }
  // integer coordinates inside the image: the 32x32 replicate-clamped window [k-16, k+15] holds every
  // box (a clamped centre only moves towards the keypoint).  >= 15 px from the border no centre is
  // clamped and the precomputed corner offsets apply; closer, the corners are computed per pair.
  const bool onpixel = ky == floorf(ky) && kx == floorf(kx) && ky <= (float)(h - 1) && kx <= (float)(w - 1);
  if (!onpixel) {
    if (lane == 0) status[flat] = 0;
    return;
  }
  const bool interior = ky >= 15.0f && ky <= (float)(h - 15) && kx >= 15.0f && kx <= (float)(w - 15);
  const int oy = (int)ky - 16, ox = (int)kx - 16;
  int *isat = sat_all[wave];
  const int half = lane >> 5, c = lane & 31;
  // lane = (half, column c): 16 rows of its column, all loads issued before use
  // scalar base + 32-bit byte offsets.  Interior windows lie inside the image: one multiply for the lane's first
  // row, then a scalar row step per load; only windows that touch the border pay the per-row clamp and multiply
  // (the kernel is bound by VALU issue: the clamped form was a quarter of its instruction slots).
  // (`interior` allows ky = 15 or ky = h - 15, whose window has one row or column outside the image that no
  // box uses: those take the clamped form too)
  const bool window_inside = ky >= 16.0f && ky <= (float)(h - 16) && kx >= 16.0f && kx <= (float)(w - 16);
  uint32_t off[16];
  if (window_inside) {                                         // wave-uniform
    const uint32_t first = (uint32_t)((oy + 16 * half) * w + ox + c) * PB;
#pragma unroll
    for (int r = 0; r < 16; ++r) off[r] = first + (uint32_t)(r * w) * PB;
  } else {
    const int gx = clampi(ox + c, 0, w - 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) off[r] = (uint32_t)(clampi(oy + 16 * half + r, 0, h - 1) * w + gx) * PB;
  }
  PIX px[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) px[r] = *reinterpret_cast<const PIX *>(reinterpret_cast<const char *>(im) + off[r]);
  const uint4 *plan_offs = reinterpret_cast<const uint4 *>(plan + 1);
  const int *plan_tint = reinterpret_cast<const int *>(plan_offs + num_pairs);
  const uint32_t *plan_geom = reinterpret_cast<const uint32_t *>(plan_tint + num_pairs);
  const unsigned long long *plan_masks = reinterpret_cast<const unsigned long long *>(plan_geom + num_pairs);
  // the first pair-table words ride behind the window gather instead of waiting for the table build
  constexpr int NPRE = GROUPS ? 3 : 0;
  uint4 o_pre[NPRE ? NPRE : 1];
  int t_pre[NPRE ? NPRE : 1];
#pragma unroll
  for (int g = 0; g < NPRE; ++g) {
    o_pre[g] = plan_offs[g * 64 + lane];
    t_pre[g] = plan_tint[g * 64 + lane];
  }
  // uint8-valued window: every pixel has the bits of (float)(uint8)pixel (-0.0 counts as differing and takes the general
  // kernel).  One convert and ONE v_bitop3 (differs |= back ^ px) per pixel behind an asm -- two converts, three compares
  // and their scalar ANDs before (round 4; written in C the optimiser turns the or-chain back into compares)
  uint32_t differs = 0u;
  int col[16];
  int acc = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int v = (int)px[r];
    if constexpr (std::is_same<PIX, float>::value) {
      const float back = (float)(v & 0xff);
      asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(differs) : "v"(back), "v"(px[r]));   // a | (b ^ c)
    }
    acc += v;
    col[r] = acc;
  }
  if (std::is_same<PIX, float>::value && !__all(differs == 0u)) {
    if (lane == 0) status[flat] = 0;
    return;
  }
  const int upper = __shfl(acc, c, 64);            // column total of rows 0..15 (held by half 0)
  if (lane < FR) { isat[lane] = 0; isat[lane * FW] = 0; }
#pragma unroll
  for (int r = 0; r < 16; ++r) isat[(16 * half + r + 1) * FW + c + 1] = col[r] + (half ? upper : 0);
  __builtin_amdgcn_wave_barrier();                 // same wave: DS operations execute in order
  {
    // (the row direction as a DPP prefix sum in registers -- 96 more VALU, 32 fewer LDS instructions -- measured 8 %
    // SLOWER: the kernel saturates VALU issue and the LDS array at the same time, PMC in DESIGN.md K4)
    int *row = isat + (c + 1) * FW + 16 * half + 1;   // lane = (half, row c): 16 entries of row c
    int v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = row[q];
    int racc = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { racc += v[q]; v[q] = racc; }
    const int left = __shfl(racc, c, 64);          // total of columns 0..15 of this row
#pragma unroll
    for (int q = 0; q < 16; ++q) row[q] = v[q] + (half ? left : 0);
  }
  __builtin_amdgcn_wave_barrier();
  const char *sbase = reinterpret_cast<const char *>(isat);
  // round g of lane l evaluates a pair of some canonical 64-pair group G (same lane l): masks[g][G] = the lanes
  // whose round-g pair belongs to group G, so word G of the packed descriptor, in the table's order, is the OR over
  // the rounds of ballot & mask -- scalar work (the masks arrive by s_load), the vector unit only compares
  unsigned long long wordv[GMAX];
  int pop = 0;
#pragma unroll
  for (int g = 0; g < GMAX; ++g) wordv[g] = 0ull;
  auto emit = [&](int g, int d, int tint) {
    const unsigned long long hit = __ballot(d <= tint);                   // bad.py:567
    pop += (int)__popcll(hit);
#pragma unroll
    for (int G = 0; G < GMAX; ++G)
      if (G < groups) wordv[G] |= hit & plan_masks[g * groups + G];
  };
  if (interior) {                                                          // wave-uniform
    auto at = [&](uint32_t byte_off) { return *reinterpret_cast<const int *>(sbase + byte_off); };
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
      if (g < groups) {
        const uint4 o = g < NPRE ? o_pre[g] : plan_offs[g * 64 + lane];
        const int tint = g < NPRE ? t_pre[g] : plan_tint[g * 64 + lane];
        // x, y: the four corners that enter s1 - s2 with +, z, w: the four with - (order chosen by the plan)
        const int plus = (at(o.x & 0xFFFFu) + at(o.x >> 16)) + (at(o.y & 0xFFFFu) + at(o.y >> 16));
        const int minus = (at(o.z & 0xFFFFu) + at(o.z >> 16)) + (at(o.w & 0xFFFFu) + at(o.w >> 16));
        emit(g, plus - minus, tint);
      }
    }
  } else {
    // box centre = clamp(keypoint + offset) into the image (bad.py:518-556 for integer positions),
    // in window coordinates; rows a..b-1 / columns l..r-1 of the table with its leading zero row
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
      if (g < groups) {
        const uint32_t q = plan_geom[g * 64 + lane];
        const int tint = plan_tint[g * 64 + lane];
        const int rad = (int)((q >> 20) & 15u);
        auto box = [&](int offx, int offy) {
          const int cx = clampi((int)kx + offx, 0, w - 1) - ox, cy = clampi((int)ky + offy, 0, h - 1) - oy;
          const int a = cy - rad, b = cy + rad + 1, l = cx - rad, r = cx + rad + 1;
          return (isat[b * FW + r] - isat[a * FW + r]) - (isat[b * FW + l] - isat[a * FW + l]);
        };
        const int s1 = box((int)(q & 31u) - 16, (int)((q >> 10) & 31u) - 16);
        const int s2 = box((int)((q >> 5) & 31u) - 16, (int)((q >> 15) & 31u) - 16);
        emit(g, s1 - s2, tint);
      }
    }
  }
  if (bits) {                                      // lane g stores word g: one store for the whole packed row
    unsigned long long mine = 0ull;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) mine = (lane == g) ? wordv[g] : mine;
    if (lane < groups) reinterpret_cast<unsigned long long *>(bits + (size_t)flat * words)[lane] = mine;
  }
  if (desc) {
    const float inv = normalize ? fmaxf(sqrtf((float)pop), 1e-12f) : 1.0f;       // bad.py:573
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
      if (g < groups) {
        const float v = ((wordv[g] >> lane) & 1ull) ? 1.0f : 0.0f;
        desc[(size_t)flat * num_pairs + g * 64 + lane] = normalize ? v / inv : v;
      }
    }
  }
  if (lane == 0) status[flat] = 1;
}

// ---- general kernel ----------------------------------------------------------------------------
// status == nullptr: one workgroup (one wave) per keypoint.  status != nullptr: one workgroup per
// `chunk` (<= 64) consecutive keypoints; it visits only those the fast kernel flagged (status == 0).
// Small chunks keep the serial depth per wave low: about 6 % of the keypoints are flagged at 640x480
// (those within 15 px of the border), i.e. about one per 16.
template <typename PIX>
__global__ __launch_bounds__(64) void sparse_bad_kernel(MiSets images, int h, int w,
                                                        MiSets kpt_sets, int k, int total,
                                                        const uint32_t *__restrict__ geom,
                                                        const float *__restrict__ thr, int num_pairs,
                                                        int mode, float temperature, int normalize,
                                                        float scale_y, float scale_x,
                                                        float *__restrict__ desc,
                                                        uint32_t *__restrict__ bits,
                                                        const uint8_t *__restrict__ status, int chunk) {
  __shared__ double sat[SP * SP];
  __shared__ float vals[1024];               // un-normalised descriptor row (num_pairs <= 1024)
  const int lane = threadIdx.x;
  const int groups = num_pairs / 64;
  const int words = num_pairs / 32;

  unsigned long long todo = 1ull;
  int first = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);   // one image's keypoints share an XCD's L2
  if (status) {
    first = (int)blockIdx.x * chunk;
    const int mine = first + lane;
    todo = __ballot(lane < chunk && mine < total && status[mine] == 0);
    if (!todo) return;                                         // nothing flagged in this chunk (the common case)
  }
  // pair table words of the first 8 groups: fetched once, reused for every keypoint of this wave
  uint32_t qg[8];
  float tg[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    qg[g] = (g < groups) ? geom[g * 64 + lane] : 0u;
    tg[g] = (g < groups) ? thr[g * 64 + lane] : 0.0f;
  }

  while (todo) {                                               // wave-uniform loop
    const int bit = __ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    const int flat = first + bit;
    const int img = flat / k;
    const PIX *im = mi_set_item<PIX>(images, img, (size_t)h * w);
    const float *kp = mi_set_item<float>(kpt_sets, img, (size_t)k * 2) + (size_t)(flat - img * k) * 2;
    const float ky_raw = kp[0];
    const float kx_raw = kp[1];
    const bool valid = ky_raw >= 0.0f;                                   // bad.py:461
    const float ky = fminf(fmaxf(ky_raw, 0.0f), (float)(h - 1));         // bad.py:464-465
    const float kx = fminf(fmaxf(kx_raw, 0.0f), (float)(w - 1));
    const int oy = (int)floorf(ky) - WOFF, ox = (int)floorf(kx) - WOFF;

    // summed-area table of the replicate-extended window, sat[r+1][c+1] = sum of rows<=r, cols<=c.
    // Lane c < 34 owns window column c (34 loads in flight), column prefix in registers, then
    // lane r < 34 takes row r (stride 35 doubles between lanes: conflict-free for ds_read_b64).
    __syncthreads();                                                     // previous keypoint's reads are done
    for (int i = lane; i < SP; i += 64) { sat[i] = 0.0; sat[i * SP] = 0.0; }
    if (lane < WIN) {
      const int gx = clampi(ox + lane, 0, w - 1);
      PIX px[WIN];
#pragma unroll
      for (int r = 0; r < WIN; ++r) px[r] = im[(size_t)clampi(oy + r, 0, h - 1) * w + gx];
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < WIN; ++r) {
        acc += (double)px[r];
        sat[(r + 1) * SP + (lane + 1)] = acc;
      }
    }
    __syncthreads();
    if (lane < WIN) {
      double *row = sat + (lane + 1) * SP + 1;
      double v[WIN];
#pragma unroll
      for (int c = 0; c < WIN; ++c) v[c] = row[c];
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < WIN; ++c) { acc += v[c]; row[c] = acc; }
    }
    __syncthreads();

    const size_t drow = (size_t)flat * (size_t)num_pairs;
    uint32_t *brow = bits ? bits + (size_t)flat * words : nullptr;
    int pop = 0;
    float sumsq = 0.0f;
    auto eval_group = [&](int g, uint32_t q, float thr_p) {
      const int p = g * 64 + lane;
      const int x1 = (int)(q & 31u) - 16, x2 = (int)((q >> 5) & 31u) - 16;
      const int y1 = (int)((q >> 10) & 31u) - 16, y2 = (int)((q >> 15) & 31u) - 16;
      const int r = (int)((q >> 20) & 15u);
      const int c1y = nearest_centre(ky + (float)y1, scale_y, h) - oy;
      const int c1x = nearest_centre(kx + (float)x1, scale_x, w) - ox;
      const int c2y = nearest_centre(ky + (float)y2, scale_y, h) - oy;
      const int c2x = nearest_centre(kx + (float)x2, scale_x, w) - ox;
      // box rows [c-r, c+r] of the window -> SAT rows c-r and c+r+1 (clamped: cannot trigger for
      // table geometry; keeps arbitrary user tables memory-safe)
      const int a1 = clampi(c1y - r, 0, WIN), b1 = clampi(c1y + r + 1, 0, WIN);
      const int l1 = clampi(c1x - r, 0, WIN), r1 = clampi(c1x + r + 1, 0, WIN);
      const int a2 = clampi(c2y - r, 0, WIN), b2 = clampi(c2y + r + 1, 0, WIN);
      const int l2 = clampi(c2x - r, 0, WIN), r2 = clampi(c2x + r + 1, 0, WIN);
      const double s1 = (sat[b1 * SP + r1] - sat[a1 * SP + r1]) - (sat[b1 * SP + l1] - sat[a1 * SP + l1]);
      const double s2 = (sat[b2 * SP + r2] - sat[a2 * SP + r2]) - (sat[b2 * SP + l2] - sat[a2 * SP + l2]);
      const double area = (double)((2 * r + 1) * (2 * r + 1));
      const double t = (double)thr_p;
      if (mode == MI_BAD_HARD) {
        // bit = (mean1 - mean2 - t <= 0)  <=>  s1 - s2 <= t * area   (t*area exact in fp64)
        const bool bitv = valid && ((s1 - s2) <= t * area);              // bad.py:567,570
        const unsigned long long word = __ballot(bitv);
        pop += (int)__popcll(word);
        if (brow && lane == 0) {
          brow[2 * g] = (uint32_t)word;
          brow[2 * g + 1] = (uint32_t)(word >> 32);
        }
        if (desc) vals[p] = bitv ? 1.0f : 0.0f;
      } else {
        const float c = (float)((s1 - s2) / area - t);                   // bad.py:559
        float v = c;
        if (mode == MI_BAD_SOFT) v = 1.0f / (1.0f + expf(c * temperature));   // sigmoid(-c*T), bad.py:565
        v = valid ? v : 0.0f;
        sumsq += v * v;
        vals[p] = v;
      }
    };
#pragma unroll
    for (int g = 0; g < 8; ++g)
      if (g < groups) eval_group(g, qg[g], tg[g]);
#pragma unroll 1
    for (int g = 8; g < groups; ++g) eval_group(g, geom[g * 64 + lane], thr[g * 64 + lane]);

    if (desc) {
      float inv = 1.0f;
      if (normalize) {                                                   // F.normalize(p=2, eps=1e-12), bad.py:573
        const float ss = (mode == MI_BAD_HARD) ? (float)pop : wave_sum(sumsq);
        inv = fmaxf(sqrtf(ss), 1e-12f);
      }
      // each lane re-reads only what it wrote itself (p = g*64 + lane)
      for (int g = 0; g < groups; ++g) {
        const float v = vals[g * 64 + lane];
        desc[drow + g * 64 + lane] = normalize ? v / inv : v;
      }
    }
  }
}

// ---- plan construction (host) ---------------------------------------------------------------------
// s1 - s2 = [(b,r)1 + (a,l)1 + (a,r)2 + (b,l)2] - [(a,r)1 + (b,l)1 + (b,r)2 + (a,l)2]: four table corners enter
// with +, four with -.  The fast kernel is bound by LDS bank conflicts on exactly these reads (64 scattered addresses
// per instruction, 3.3 passes on average as the table stands), and the addresses are the same for every keypoint, so
// their schedule -- which round a lane evaluates each of its pairs in, and in which order it reads the corners of a
// sign class -- is chosen here, once per table: bad_plan_opt.h (420 -> 184 passes on the 512-pair table, 128 =
// conflict-free; measured per 448 x 512 keypoints: uint8 frames 213 -> 180 us, fp32 frames 244 -> 240 us -- those wait
// on their four times larger window gather instead).  The search takes a few hundred milliseconds, so schedules are kept per table content.
struct HostPlan {
  std::vector<uint32_t> geom;          // key: the table
  bool geometry_ok = false;
  mi::BadGatherSchedule sched;
};

std::vector<uint16_t> table_corners(const std::vector<uint32_t> &geom, bool *all_inside) {
  const int num_pairs = (int)geom.size();
  std::vector<uint16_t> c((size_t)num_pairs * 8);
  bool ok = true;
  auto word = [](int row, int col) { return (uint16_t)(row * FW + col); };
  for (int p = 0; p < num_pairs; ++p) {
    const uint32_t q = geom[p];
    const int x1 = (int)(q & 31u), x2 = (int)((q >> 5) & 31u);
    const int y1 = (int)((q >> 10) & 31u), y2 = (int)((q >> 15) & 31u);
    const int r = (int)((q >> 20) & 15u);
    const bool in = x1 - r >= 0 && x2 - r >= 0 && y1 - r >= 0 && y2 - r >= 0 && x1 + r <= 31 && x2 + r <= 31 &&
                    y1 + r <= 31 && y2 + r <= 31;
    ok = ok && in;
    const int cx1 = in ? x1 : 16, cy1 = in ? y1 : 16, cx2 = in ? x2 : 16, cy2 = in ? y2 : 16, cr = in ? r : 0;
    uint16_t *pos = &c[(size_t)p * 8], *neg = pos + 4;
    pos[0] = word(cy1 + cr + 1, cx1 + cr + 1);   // (b,r) of box 1
    pos[1] = word(cy1 - cr, cx1 - cr);           // (a,l) of box 1
    pos[2] = word(cy2 - cr, cx2 + cr + 1);       // (a,r) of box 2
    pos[3] = word(cy2 + cr + 1, cx2 - cr);       // (b,l) of box 2
    neg[0] = word(cy1 - cr, cx1 + cr + 1);       // (a,r) of box 1
    neg[1] = word(cy1 + cr + 1, cx1 - cr);       // (b,l) of box 1
    neg[2] = word(cy2 + cr + 1, cx2 + cr + 1);   // (b,r) of box 2
    neg[3] = word(cy2 - cr, cx2 - cr);           // (a,l) of box 2
  }
  *all_inside = ok;
  return c;
}

// schedules by table content: a process builds plans for a handful of tables (one per descriptor size), many times
std::shared_ptr<const HostPlan> host_plan_for(const std::vector<uint32_t> &geom) {
  static std::mutex mu;
  static std::vector<std::shared_ptr<const HostPlan>> *cache = new std::vector<std::shared_ptr<const HostPlan>>();
  {
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &hp : *cache)
      if (hp->geom == geom) return hp;
  }
  auto hp = std::make_shared<HostPlan>();
  hp->geom = geom;
  const std::vector<uint16_t> corners = table_corners(geom, &hp->geometry_ok);
  hp->sched = mi::schedule_bad_gathers(corners, (int)geom.size());
  std::lock_guard<std::mutex> lock(mu);
  if (cache->size() >= 16) cache->erase(cache->begin());
  cache->push_back(hp);
  return hp;
}

}  // namespace

extern "C" size_t mi_bad_plan_bytes(int num_pairs) {
  if (num_pairs <= 0) return 0;
  return sizeof(BadPlan) + (size_t)num_pairs * (sizeof(uint4) + sizeof(int) + sizeof(uint32_t)) +
         (size_t)(num_pairs / 64) * (num_pairs / 64) * sizeof(uint64_t);
}

// development aid (mi_debug_bad_plan_passes of the debug library, csrc/hooks.hip): the LDS passes per keypoint of the
// gather schedule for a HOST copy of a pair table, as the table stands and as scheduled -- no GPU involved
int mi_bad_plan_passes_host(const uint32_t *pair_geom_host, int num_pairs, int *canonical, int *scheduled) {
  if (!pair_geom_host || !canonical || !scheduled) return MI_E_NULL;
  if (num_pairs <= 0 || num_pairs % 64 != 0 || num_pairs > 1024) return MI_E_PARAM;
  const auto hp = host_plan_for(std::vector<uint32_t>(pair_geom_host, pair_geom_host + num_pairs));
  for (int e = 0; e < num_pairs; ++e)
    if (hp->sched.exec_pair[e] % 64 != e % 64) return MI_E_PARAM;          // a lane only reorders its own pairs
  *canonical = hp->sched.passes_canonical;
  *scheduled = hp->sched.passes;
  return MI_OK;
}

extern "C" int mi_bad_plan_build(const uint32_t *pair_geom, const float *pair_thr, int num_pairs, void *plan,
                                 mi_stream_t stream) {
  MI_ENTER();
  // Set-up call, once per pair table: it copies the table to the host, builds the plan there (see above)
  // and uploads it, synchronising `stream` twice.  Not capturable into a hipGraph; everything else is.
  if (!pair_geom || !pair_thr || !plan) return MI_E_NULL;
  if (num_pairs <= 0 || num_pairs % 64 != 0 || num_pairs > 1024) return MI_E_PARAM;
  if (((uintptr_t)plan % 16) != 0) return MI_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  std::vector<uint32_t> geom(num_pairs);
  std::vector<float> thr(num_pairs);
  if (hipMemcpyAsync(geom.data(), pair_geom, sizeof(uint32_t) * num_pairs, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipMemcpyAsync(thr.data(), pair_thr, sizeof(float) * num_pairs, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return mi_launch_status() ? mi_launch_status() : MI_E_PARAM;
  const auto sched = host_plan_for(geom);
  std::vector<char> host(mi_bad_plan_bytes(num_pairs));
  BadPlan *hp = reinterpret_cast<BadPlan *>(host.data());
  uint4 *offs = reinterpret_cast<uint4 *>(hp + 1);
  int *tint = reinterpret_cast<int *>(offs + num_pairs);
  uint32_t *xgeom = reinterpret_cast<uint32_t *>(tint + num_pairs);
  uint64_t *masks = reinterpret_cast<uint64_t *>(xgeom + num_pairs);
  const int rounds = num_pairs / 64;
  for (int i = 0; i < rounds * rounds; ++i) masks[i] = 0;
  for (int e = 0; e < num_pairs; ++e) {
    const int p = sched->sched.exec_pair[e], lane = e % 64, round = e / 64;
    const uint16_t *pos = &sched->sched.pos[(size_t)e * 4], *neg = &sched->sched.neg[(size_t)e * 4];
    auto two = [](uint16_t lo, uint16_t hi) { return (uint32_t)(lo * 4u) | ((uint32_t)(hi * 4u) << 16); };   // byte offsets
    offs[e] = make_uint4(two(pos[0], pos[1]), two(pos[2], pos[3]), two(neg[0], neg[1]), two(neg[2], neg[3]));
    const int r = (int)((geom[p] >> 20) & 15u);
    tint[e] = (int)floor((double)thr[p] * (double)((2 * r + 1) * (2 * r + 1)));
    xgeom[e] = geom[p];
    masks[round * rounds + p / 64] |= (uint64_t)1 << lane;
  }
  hp->geometry_ok = sched->geometry_ok ? 1 : 0;
  hp->num_pairs = num_pairs;
  hp->passes = sched->sched.passes;
  hp->passes_canonical = sched->sched.passes_canonical;
  if (hipMemcpyAsync(plan, host.data(), host.size(), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return mi_launch_status() ? mi_launch_status() : MI_E_PARAM;
  return MI_OK;
}

namespace {
template <typename PIX>
int sparse_bad_launch(MiSets image, int n, int h, int w, MiSets keypoints, int k, const uint32_t *pair_geom,
                      const float *pair_thr, int num_pairs, int mode, float temperature, int normalize, float *desc,
                      uint32_t *bits, const void *plan, uint8_t *status, mi_stream_t stream) {
  if (!image.a || !keypoints.a || !pair_geom || !pair_thr) return MI_E_NULL;
  if ((image.per_set < n && !image.b) || (keypoints.per_set < n && !keypoints.b)) return MI_E_NULL;
  if (!desc && !bits) return MI_E_NULL;
  if (n <= 0 || h <= 0 || w <= 0 || k <= 0) return MI_E_SHAPE;
  if (num_pairs <= 0 || num_pairs % 64 != 0 || num_pairs > 1024) return MI_E_PARAM;
  if (mode != MI_BAD_RAW && mode != MI_BAD_SOFT && mode != MI_BAD_HARD) return MI_E_PARAM;
  if (bits && mode != MI_BAD_HARD) return MI_E_PARAM;
  if ((long long)n * k > 0x7fffffffLL) return MI_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int total = n * k;
  // bad.py:469-470: python double 2/(size-1+1e-8), multiplied into an fp32 tensor
  const float scale_y = (float)(2.0 / ((double)(h - 1) + 1e-8));
  const float scale_x = (float)(2.0 / ((double)(w - 1) + 1e-8));
  const bool fast = plan != nullptr && status != nullptr && mode == MI_BAD_HARD && h >= 32 && w >= 32 &&
                    (long long)h * w < (1LL << 30);          // the fast kernel addresses a plane with 32-bit byte offsets
  if (fast) {
    const BadPlan *bp = reinterpret_cast<const BadPlan *>(plan);
    auto fast_kernel = num_pairs == 512 ? bad_fast_kernel<8, PIX> : num_pairs == 256 ? bad_fast_kernel<4, PIX> : bad_fast_kernel<0, PIX>;
    hipLaunchKernelGGL(fast_kernel, dim3((unsigned)ceil_div(total, 4)), dim3(256), 0, s, image, h, w, keypoints,
                       k, total, num_pairs, normalize, bp, pair_geom, desc, bits, status);
    constexpr int CHUNK = 16;
    hipLaunchKernelGGL(sparse_bad_kernel<PIX>, dim3((unsigned)ceil_div(total, CHUNK)), dim3(64), 0, s, image, h, w,
                       keypoints, k, total, pair_geom, pair_thr, num_pairs, mode, temperature, normalize, scale_y,
                       scale_x, desc, bits, status, CHUNK);
  } else {
    hipLaunchKernelGGL(sparse_bad_kernel<PIX>, dim3((unsigned)total), dim3(64), 0, s, image, h, w, keypoints, k, total,
                       pair_geom, pair_thr, num_pairs, mode, temperature, normalize, scale_y, scale_x, desc, bits,
                       nullptr, 1);
  }
  return mi_launch_status();
}
}  // namespace

int mi_sparse_bad_sets(MiSets images, int pix_u8, int n, int h, int w, MiSets keypoints, int k, const uint32_t *pair_geom,
                       const float *pair_thr, int num_pairs, int mode, float temperature, int normalize, float *desc,
                       uint32_t *bits, const void *plan, uint8_t *status, mi_stream_t stream) {
  if (pix_u8)
    return sparse_bad_launch<uint8_t>(images, n, h, w, keypoints, k, pair_geom, pair_thr, num_pairs, mode, temperature,
                                      normalize, desc, bits, plan, status, stream);
  return sparse_bad_launch<float>(images, n, h, w, keypoints, k, pair_geom, pair_thr, num_pairs, mode, temperature,
                                  normalize, desc, bits, plan, status, stream);
}

extern "C" int mi_sparse_bad(const float *image, int n, int h, int w, const float *keypoints, int k,
                             const uint32_t *pair_geom, const float *pair_thr, int num_pairs, int mode,
                             float temperature, int normalize, float *desc, uint32_t *bits, const void *plan,
                             uint8_t *status, mi_stream_t stream) {
  MI_ENTER();
  return sparse_bad_launch<float>(mi_one_set(image, n), n, h, w, mi_one_set(keypoints, n), k, pair_geom, pair_thr, num_pairs, mode, temperature,
                                  normalize, desc, bits, plan, status, stream);
}

// image1 / image2 of a matcher behind one launch: keypoints (2 * per_set, k, 2) and the outputs hold batch a first
extern "C" int mi_sparse_bad_pair(const void *image_a, const void *image_b, int pixels_are_u8, int per_set, int h, int w,
                                  const float *keypoints, int k, const uint32_t *pair_geom, const float *pair_thr,
                                  int num_pairs, int mode, float temperature, int normalize, float *desc, uint32_t *bits,
                                  const void *plan, uint8_t *status, mi_stream_t stream) {
  MI_ENTER();
  if (!image_b) return MI_E_NULL;
  if (per_set <= 0 || per_set > 0x3fffffff) return MI_E_SHAPE;
  return mi_sparse_bad_sets(MiSets{image_a, image_b, per_set}, pixels_are_u8 ? 1 : 0, 2 * per_set, h, w,
                            mi_one_set(keypoints, 2 * per_set), k, pair_geom, pair_thr, num_pairs, mode, temperature,
                            normalize, desc, bits, plan, status, stream);
}

// u8 ingest: the same descriptors from uint8 pixels (a uint8 frame is what the camera delivers; the reference converts
// it to float32 on the host first, sample/visual_odometry.py:65-92)
extern "C" int mi_sparse_bad_u8(const uint8_t *image, int n, int h, int w, const float *keypoints, int k,
                                const uint32_t *pair_geom, const float *pair_thr, int num_pairs, int mode,
                                float temperature, int normalize, float *desc, uint32_t *bits, const void *plan,
                                uint8_t *status, mi_stream_t stream) {
  MI_ENTER();
  return sparse_bad_launch<uint8_t>(mi_one_set(image, n), n, h, w, mi_one_set(keypoints, n), k, pair_geom, pair_thr, num_pairs, mode, temperature,
                                    normalize, desc, bits, plan, status, stream);
}
