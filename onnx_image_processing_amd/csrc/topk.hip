// K3: per-image top-k of the candidate keys -> keypoints.
// Semantics: reference pytorch_model/utils/keypoint_utils.py:94-115 (torch.topk sorted=True,
// index -> (y, x), invalid -> (-1, -1) / score 0) with the build's tie policy: keys are
// (score bits << 32 | inverted linear index), all distinct, so "descending key" means
// score descending, then linear index ascending -- independent of how K2 ordered them.
//
// Input: the segmented candidate buffer of K2 (one segment per 128x32 tile, count per
// segment).  One 1024-thread workgroup per image.  Up to 4096 candidates are gathered into LDS
// (slot ranges handed out by one LDS atomic per segment up front, four segments' loads in flight
// per wave) and bitonic-sorted, with the keys in registers when the padded count is exactly 4096.  Longer lists
// (plateau images can make every pixel a candidate) first run an exact 8-pass MSB radix select
// for the k-th largest key straight from global memory, then sort only the k survivors.
#include "common.h"

#include "hooks.h"
// test hook (mi_debug_set key 9, debug library only): 1 = radix-select the k-th key and sort only the k selected keys
// whenever n > k (default), 0 = always sort every candidate

namespace {

constexpr int TK_THREADS = 1024;
constexpr int TK_WAVES = TK_THREADS / 64;
constexpr int TK_MAX = 4096;
constexpr int TK_SEGS = 1024;        // segments with a slot-table entry (1920x1080 has 510)

__device__ __forceinline__ void bitonic_sort_desc(uint64_t *keys, int npad, int t) {
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < (npad >> 1); i += TK_THREADS) {
        const int pos = 2 * i - (i & (stride - 1));
        const int par = pos + stride;
        const bool desc = ((pos & size) == 0);
        const uint64_t a = keys[pos], b = keys[par];
        if ((a < b) == desc) {
          keys[pos] = b;
          keys[par] = a;
        }
      }
      __syncthreads();
    }
  }
}

// The same network for exactly TK_MAX = 4 * TK_THREADS keys with the keys in registers: thread t owns
// elements 4t..4t+3, so strides 1-2 are register swaps, strides 4-128 are wave shuffles (partner lane =
// lane ^ stride/4) and only strides >= 256 -- 10 of the 78 stages -- go through LDS, ping-ponging two
// buffers so that each needs a single barrier.  Identical result to bitonic_sort_desc (same network).
template <int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_stage_4096(uint64_t (&r)[4], uint64_t *keys, uint64_t *keys2, int t, int &flip) {
  if constexpr (STRIDE < 4) {                        // partner in the same thread
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if ((j ^ STRIDE) > j) {
        const bool desc = (((t * 4 + j) & SIZE) == 0);
        const uint64_t a = r[j], b = r[j ^ STRIDE];
        const bool sw = (a < b) == desc;
        r[j] = sw ? b : a;
        r[j ^ STRIDE] = sw ? a : b;
      }
    }
  } else {
    uint64_t other[4];
    if constexpr (STRIDE < 256) {                    // partner lane in the same wave
#pragma unroll
      for (int j = 0; j < 4; ++j) other[j] = __shfl_xor(r[j], STRIDE >> 2, 64);
    } else {                                         // partner in another wave: through LDS, buffers alternate
      uint64_t *buf = flip ? keys2 : keys;
      flip ^= 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) buf[t * 4 + j] = r[j];
      __syncthreads();
      const int pt = t ^ (STRIDE >> 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) other[j] = buf[pt * 4 + j];
    }
    const bool lower = (t & (STRIDE >> 2)) == 0;     // this element is the lower index of its pair
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool desc = (((t * 4 + j) & SIZE) == 0);
      const bool gt = r[j] > other[j];
      const uint64_t mx = gt ? r[j] : other[j], mn = gt ? other[j] : r[j];
      r[j] = (lower == desc) ? mx : mn;
    }
  }
}

template <int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_merge_4096(uint64_t (&r)[4], uint64_t *keys, uint64_t *keys2, int t, int &flip) {
  if constexpr (STRIDE >= 1) {
    bitonic_stage_4096<SIZE, STRIDE>(r, keys, keys2, t, flip);
    bitonic_merge_4096<SIZE, STRIDE / 2>(r, keys, keys2, t, flip);
  }
}

template <int SIZE>
__device__ __forceinline__ void bitonic_levels_4096(uint64_t (&r)[4], uint64_t *keys, uint64_t *keys2, int t, int &flip) {
  if constexpr (SIZE <= TK_MAX) {
    bitonic_merge_4096<SIZE, SIZE / 2>(r, keys, keys2, t, flip);
    bitonic_levels_4096<SIZE * 2>(r, keys, keys2, t, flip);
  }
}

// The same network for exactly TK_MAX = 4 * TK_THREADS keys with the keys in registers (every stage a
// compile-time instance, so the four keys stay in VGPRs): thread t owns elements 4t..4t+3, strides 1-2 are
// register swaps, strides 4-128 wave shuffles (partner lane = lane ^ stride/4) and only strides >= 256 --
// 10 of the 78 stages -- go through LDS, ping-ponging two buffers so that each needs a single barrier.
// Identical result to bitonic_sort_desc (same network).
__device__ __forceinline__ void bitonic_sort_desc_4096(uint64_t *keys, uint64_t *keys2, int t) {
  uint64_t r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = keys[t * 4 + j];
  int flip = 0;
  bitonic_levels_4096<2>(r, keys, keys2, t, flip);
  __syncthreads();                                          // the last LDS readers are done
#pragma unroll
  for (int j = 0; j < 4; ++j) keys[t * 4 + j] = r[j];
  __syncthreads();
}

// BIG = false: 73 KB of LDS, two workgroups per CU -- images whose candidates are expected to fit the LDS key array
// (640x480: ~3,300).  BIG = true (large images; chosen on the host by the segment count): ONE 128 KB LDS region that
// first holds the 32-bit SCORES of up to TK_BIG_MAX candidates -- the k-th largest score is then found by a radix
// select that never leaves LDS, and one more pass over the candidate lists gathers the keys at or above it into the
// key array (which reuses the region) -- instead of up to eight dependent passes over global memory plus a gather:
// 1080p, K = 1024, ~22,000 candidates per image: 148 -> ~50 us per 128 images.  Every path returns the k largest keys
// in descending order; keys are distinct, so the result does not depend on the path.
constexpr int TK_BIG_MAX = 32768;    // scores the big kernel's LDS region holds (128 KB)

template <bool BIG>
__global__ __launch_bounds__(TK_THREADS) void topk_kernel(const uint64_t *__restrict__ cand,
                                                          const uint32_t *__restrict__ count, int segments,
                                                          uint32_t seg_cap, int w, int k, MiSets kpt_sets,
                                                          float *__restrict__ kscores, int select_mode,
                                                          unsigned long long *prof) {
#define TK_STAMP(i) do { if (prof && blockIdx.x == 0 && threadIdx.x == 0) prof[i] = wall_clock64(); } while (0)
  TK_STAMP(0);
  __shared__ uint64_t raw[BIG ? TK_BIG_MAX / 2 : 2 * TK_MAX];
  uint64_t *const keys = raw;
  uint64_t *const keys2 = raw + TK_MAX;   // second buffer of the register sort's cross-wave stages
  uint32_t *const scores = reinterpret_cast<uint32_t *>(raw);   // BIG: the candidates' scores, before the keys move in
  __shared__ uint32_t seg_cnt[TK_SEGS], seg_base[TK_SEGS];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t wsum[TK_WAVES];
  __shared__ uint64_t s_prefix;
  __shared__ uint32_t s_krem, s_fill, s_done;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int img = blockIdx.x;
  const uint64_t *list = cand + (size_t)img * segments * seg_cap;
  const uint32_t *cnt = count + (size_t)img * segments;

  // total number of candidates of this image; with few enough segments each one also gets its slot range in
  // the LDS key array right here (one LDS atomic per segment), so that the gather below has no dependent
  // global load or atomic in front of its candidate loads
  const bool slots = segments <= TK_SEGS;
  uint32_t part = 0;
  if (t == 0) s_fill = 0u;
  __syncthreads();
  for (int s = t; s < segments; s += TK_THREADS) {
    const uint32_t c = min(cnt[s], seg_cap);
    part += c;
    if (slots) {
      seg_cnt[s] = c;
      seg_base[s] = c ? atomicAdd(&s_fill, c) : 0u;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) wsum[wave] = part;
  __syncthreads();
  uint32_t n = 0;
#pragma unroll
  for (int q = 0; q < TK_WAVES; ++q) n += wsum[q];
  int nsel;
  TK_STAMP(1);

  // ---- BIG: more candidates than the key array holds, but their scores fit the LDS region
  bool in_lds = false;                    // the keys to choose from are already in keys[0..n)
  if (BIG && slots && n > (uint32_t)TK_MAX && n <= (uint32_t)TK_BIG_MAX && k <= TK_MAX) {
    // (a) scores into LDS: a wave takes eight segments at a time, all loads in flight before the first LDS store
    for (int s0 = wave * 8; s0 < segments; s0 += TK_WAVES * 8) {
      uint64_t v[8];
      uint32_t c[8], base[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int sg = s0 + q;
        c[q] = sg < segments ? seg_cnt[sg] : 0u;
        base[q] = sg < segments ? seg_base[sg] : 0u;
        v[q] = 0ull;
        if ((uint32_t)lane < c[q]) v[q] = list[(size_t)sg * seg_cap + lane];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if ((uint32_t)lane < c[q]) scores[base[q] + lane] = (uint32_t)(v[q] >> 32);
        for (uint32_t i = 64 + lane; i < c[q]; i += 64) scores[base[q] + i] = (uint32_t)(list[(size_t)(s0 + q) * seg_cap + i] >> 32);
      }
    }
    if (t == 0) { s_prefix = 0ull; s_krem = (uint32_t)k; s_done = 0u; }
    __syncthreads();
    // (b) the k-th largest SCORE by MSB radix select over the LDS copy (scores repeat; the keys do not)
    uint32_t mask32 = 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (t < 256) hist[t] = 0u;
      __syncthreads();
      const uint32_t prefix = (uint32_t)s_prefix;
      for (uint32_t i0 = 0; i0 < n; i0 += TK_THREADS) {
        const uint32_t i = i0 + t;
        const uint32_t sc = i < n ? scores[i] : 0u;
        const bool act = i < n && ((sc & mask32) == prefix);
        const uint32_t digit = (sc >> shift) & 255u;
        const unsigned long long am = __ballot(act);
        if (am) {                                                // wave-level aggregation, as in the key select below
          const int leader = __ffsll((long long)am) - 1;
          const uint32_t d0 = __shfl(digit, leader, 64);
          const unsigned long long same = __ballot(act && digit == d0);
          if (lane == leader) atomicAdd(&hist[d0], (uint32_t)__popcll(same));
          if (act && digit != d0) atomicAdd(&hist[digit], 1u);
        }
      }
      __syncthreads();
      if (t < 64) {
        const uint32_t krem = s_krem;
        uint32_t c[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c[q] = hist[255 - 4 * t - q]; tot += c[q]; }
        uint32_t incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t up = __shfl_up(incl, o, 64);
          if (t >= o) incl += up;
        }
        uint32_t above = incl - tot;
        if (above < krem && krem <= incl) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (krem <= above + c[q]) {
              s_prefix = (uint64_t)(prefix | ((uint32_t)(255 - 4 * t - q) << shift));
              s_krem = krem - above;
              s_fill = c[q];                                     // keys in the chosen bin
              if (krem - above == c[q]) s_done = 1u;             // the whole bin is wanted: no further digit needed
              break;
            }
            above += c[q];
          }
        }
      }
      mask32 |= (0xFFu << shift);
      __syncthreads();
      if (s_done) break;                                         // workgroup-uniform
    }
    // Every key whose score is >= `floor32` is a candidate for the k places: the k - s_krem keys above the chosen bin
    // plus the bin's s_fill keys, of which s_krem are wanted (all of them unless scores tie exactly at the k-th place).
    const uint32_t floor32 = (uint32_t)s_prefix;                 // digits below the last pass are zero
    const uint32_t n2 = ((uint32_t)k - s_krem) + s_fill;
    __syncthreads();                                             // everybody has read s_prefix / s_krem / s_fill; scores are dead
    if (n2 <= (uint32_t)TK_MAX) {
      // (c) gather those keys from the lists into the key array (which now takes over the region)
      if (t == 0) s_fill = 0u;
      __syncthreads();
      for (int s0 = wave * 8; s0 < segments; s0 += TK_WAVES * 8) {
        uint64_t v[8];
        uint32_t c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int sg = s0 + q;
          c[q] = sg < segments ? seg_cnt[sg] : 0u;
          v[q] = 0ull;
          if ((uint32_t)lane < c[q]) v[q] = list[(size_t)sg * seg_cap + lane];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if ((uint32_t)lane < c[q] && (uint32_t)(v[q] >> 32) >= floor32) keys[atomicAdd(&s_fill, 1u)] = v[q];
          for (uint32_t i = 64 + lane; i < c[q]; i += 64) {
            const uint64_t key = list[(size_t)(s0 + q) * seg_cap + i];
            if ((uint32_t)(key >> 32) >= floor32) keys[atomicAdd(&s_fill, 1u)] = key;
          }
        }
      }
      __syncthreads();
      n = s_fill;                                                // == n2: k <= n <= TK_MAX keys to choose from, in keys[]
      in_lds = true;
    }
    __syncthreads();
  }

  if (in_lds || (n <= (uint32_t)TK_MAX && slots)) {
    // gather: a wave takes four segments at a time, all their loads in flight before the first LDS store
    for (int s0 = wave * 4; !in_lds && s0 < segments; s0 += TK_WAVES * 4) {
      uint64_t v[4];
      uint32_t c[4], base[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sg = s0 + q;
        c[q] = sg < segments ? seg_cnt[sg] : 0u;
        base[q] = sg < segments ? seg_base[sg] : 0u;
        v[q] = 0ull;
        if ((uint32_t)lane < c[q]) v[q] = list[(size_t)sg * seg_cap + lane];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if ((uint32_t)lane < c[q]) keys[base[q] + lane] = v[q];
        for (uint32_t i = 64 + lane; i < c[q]; i += 64) keys[base[q] + i] = list[(size_t)(s0 + q) * seg_cap + i];
      }
    }
    int npad = 2;
    while (npad < (int)n) npad <<= 1;
    __syncthreads();
    TK_STAMP(2);
    if ((select_mode || in_lds) && n >= (uint32_t)k && 4 * k <= TK_MAX && (n > (uint32_t)k || in_lds)) {
      // Far fewer keys are wanted than there are candidates (512 of ~3300 at 640x480): find the k-th largest key
      // by MSB radix select on register copies of the LDS keys (at most 8 passes of 8 bits, typically 4: the
      // passes stop as soon as a whole bin is wanted), move the k winners to the second buffer and sort only
      // those -- 45 stages on 512 keys instead of 78 on 4096.
      uint64_t r[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] = (uint32_t)(t + q * TK_THREADS) < n ? keys[t + q * TK_THREADS] : 0ull;
      if (t == 0) { s_prefix = 0ull; s_krem = (uint32_t)k; s_fill = 0u; s_done = 0u; }
      uint64_t mask = 0ull;
      // n == k (the usual outcome of the big kernel's score select): every key is a winner, nothing to select
      for (int shift = n == (uint32_t)k ? -8 : 56; shift >= 0; shift -= 8) {
        if (t < 256) hist[t] = 0u;
        __syncthreads();
        const uint64_t prefix = s_prefix;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool act = r[q] != 0ull && ((r[q] & mask) == prefix);
          const uint32_t digit = (uint32_t)(r[q] >> shift) & 255u;
          // wave-level aggregation: the leading digits are shared by almost every key
          const unsigned long long am = __ballot(act);
          if (am) {
            const int leader = __ffsll((long long)am) - 1;
            const uint32_t d0 = __shfl(digit, leader, 64);
            const unsigned long long same = __ballot(act && digit == d0);
            if (lane == leader) atomicAdd(&hist[d0], (uint32_t)__popcll(same));
            if (act && digit != d0) atomicAdd(&hist[digit], 1u);
          }
        }
        __syncthreads();
        if (t < 64) {
          // lane l owns bins 255-4l .. 252-4l (descending); find the bin holding the krem-th key
          const uint32_t krem = s_krem;
          uint32_t c[4], tot = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) { c[q] = hist[255 - 4 * t - q]; tot += c[q]; }
          uint32_t incl = tot;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (t >= o) incl += up;
          }
          uint32_t above = incl - tot;
          if (above < krem && krem <= incl) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (krem <= above + c[q]) {
                s_prefix = prefix | ((uint64_t)(255 - 4 * t - q) << shift);
                s_krem = krem - above;
                if (krem - above == c[q]) s_done = 1u;      // the whole bin is wanted: no further digit needed
                break;
              }
              above += c[q];
            }
          }
        }
        mask |= (0xFFull << shift);
        __syncthreads();
        if (s_done) break;                // workgroup-uniform
      }
      if (n == (uint32_t)k) __syncthreads();   // (no pass ran: thread 0's reset of s_prefix / s_fill must still be seen)
      const uint64_t kth = s_prefix;      // keys are distinct: exactly k keys are >= kth
      TK_STAMP(3);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (r[q] != 0ull && r[q] >= kth) {
          const uint32_t slot = atomicAdd(&s_fill, 1u);
          if (slot < (uint32_t)TK_MAX) keys2[slot] = r[q];
        }
      __syncthreads();
      nsel = (int)(s_fill < (uint32_t)k ? s_fill : (uint32_t)k);
      int kpad = 2;
      while (kpad < k) kpad <<= 1;
      for (int i = nsel + t; i < kpad; i += TK_THREADS) keys2[i] = 0ull;
      __syncthreads();
      TK_STAMP(4);
      if (kpad >= 128 && kpad <= TK_THREADS) {
        // Counting-rank sort of the <= 1024 winners (round 3; it replaces a merge-rank sort of 21 shuffle stages plus
        // 7-step binary searches: 7.5 -> ~2 us on the one-pair path): the keys are DISTINCT, so a key's place in the
        // descending order is simply the number of keys above it.  TPK = 1024 / kpad threads share a key, each counting
        // over its slice of the list; a wave's 64 lanes hold 64 consecutive keys and the same slice, so every LDS read
        // of the scan is one broadcast address (two keys per ds_read_b128).  The slices' counts meet in an LDS counter
        // per key (seg_cnt is free by now), then every key is written to its place.
        const int tpk = TK_THREADS / kpad, slice = kpad / tpk;        // kpad is a power of two in [128, 1024]
        const int ki = t & (kpad - 1), part = t / kpad;
        if (t < kpad) seg_cnt[t] = 0u;
        const uint64_t mine = keys2[ki];
        uint32_t above = 0;
        if (mine != 0ull) {
          const ulonglong2 *lst = reinterpret_cast<const ulonglong2 *>(keys2 + part * slice);
#pragma unroll 8
          for (int j = 0; j < slice / 2; ++j) {
            const ulonglong2 two = lst[j];
            above += (two.x > mine ? 1u : 0u) + (two.y > mine ? 1u : 0u);
          }
        }
        __syncthreads();                                              // the counters are zero, every slice is counted
        if (mine != 0ull && above) atomicAdd(&seg_cnt[ki], above);
        __syncthreads();
        if (part == 0 && mine != 0ull) keys[seg_cnt[ki]] = mine;
        __syncthreads();
      } else {
        bitonic_sort_desc(keys2, kpad, t);
        for (int i = t; i < nsel; i += TK_THREADS) keys[i] = keys2[i];      // the epilogue reads `keys`
        __syncthreads();
      }
    } else {
    for (int i = (int)n + t; i < npad; i += TK_THREADS) keys[i] = 0ull;
    __syncthreads();
    if (npad == TK_MAX) bitonic_sort_desc_4096(keys, keys2, t);
    else bitonic_sort_desc(keys, npad, t);
    nsel = (int)n < k ? (int)n : k;
    }
  } else if (n <= (uint32_t)TK_MAX) {
    // more segments than slot-table entries (very large images): a wave walks whole segments
    for (int s = wave; s < segments; s += TK_WAVES) {
      const uint32_t c = min(cnt[s], seg_cap);
      if (c == 0u) continue;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&s_fill, c);
      base = __shfl(base, 0, 64);
      const uint64_t *seg = list + (size_t)s * seg_cap;
      for (uint32_t i = lane; i < c; i += 64) keys[base + i] = seg[i];
    }
    int npad = 2;
    while (npad < (int)n) npad <<= 1;
    __syncthreads();
    for (int i = (int)n + t; i < npad; i += TK_THREADS) keys[i] = 0ull;
    __syncthreads();
    if (npad == TK_MAX) bitonic_sort_desc_4096(keys, keys2, t);
    else bitonic_sort_desc(keys, npad, t);
    nsel = (int)n < k ? (int)n : k;
  } else {
    // exact k-th largest key by MSB radix select (8 digits of 8 bits); n > TK_MAX >= k here
    if (t == 0) { s_prefix = 0ull; s_krem = (uint32_t)k; s_fill = 0u; s_done = 0u; }   // s_fill: the slot table may have used it
    uint64_t mask = 0ull;
    for (int shift = 56; shift >= 0; shift -= 8) {
      if (t < 256) hist[t] = 0u;
      __syncthreads();
      const uint64_t prefix = s_prefix;
      for (int s = wave; s < segments; s += TK_WAVES) {
        const uint32_t c = min(cnt[s], seg_cap);
        const uint64_t *seg = list + (size_t)s * seg_cap;
        for (uint32_t base = 0; base < c; base += 64) {
          const uint32_t i = base + lane;
          bool act = false;
          uint32_t digit = 0;
          if (i < c) {
            const uint64_t key = seg[i];
            act = ((key & mask) == prefix);
            digit = (uint32_t)(key >> shift) & 255u;
          }
          // wave-level aggregation: the leading digits are shared by almost every key
          const unsigned long long am = __ballot(act);
          if (am) {
            const int leader = __ffsll((long long)am) - 1;
            const uint32_t d0 = __shfl(digit, leader, 64);
            const unsigned long long same = __ballot(act && digit == d0);
            if (lane == leader) atomicAdd(&hist[d0], (uint32_t)__popcll(same));
            if (act && digit != d0) atomicAdd(&hist[digit], 1u);
          }
        }
      }
      __syncthreads();
      if (t < 64) {
        // lane l owns bins 255-4l .. 252-4l (descending); find the bin holding the krem-th key
        const uint32_t krem = s_krem;
        uint32_t c[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c[q] = hist[255 - 4 * t - q]; tot += c[q]; }
        uint32_t incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t v = __shfl_up(incl, o, 64);
          if (t >= o) incl += v;
        }
        uint32_t above = incl - tot;
        if (above < krem && krem <= incl) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (krem <= above + c[q]) {
              s_prefix = prefix | ((uint64_t)(255 - 4 * t - q) << shift);
              s_krem = krem - above;
              // the whole bin is wanted: every key with this prefix (low bits anything) is among the k largest,
              // so the prefix itself is the threshold and the remaining digits need no pass
              if (krem - above == c[q]) s_done = 1u;
              break;
            }
            above += c[q];
          }
        }
      }
      mask |= (0xFFull << shift);
      __syncthreads();
      if (s_done) break;                // workgroup-uniform
    }
    const uint64_t kth = s_prefix;   // keys are distinct: exactly k keys are >= kth
    for (int s = wave; s < segments; s += TK_WAVES) {
      const uint32_t c = min(cnt[s], seg_cap);
      const uint64_t *seg = list + (size_t)s * seg_cap;
      for (uint32_t i = lane; i < c; i += 64) {
        const uint64_t key = seg[i];
        if (key >= kth) {
          const uint32_t slot = atomicAdd(&s_fill, 1u);
          if (slot < (uint32_t)TK_MAX) keys[slot] = key;
        }
      }
    }
    __syncthreads();
    nsel = (int)(s_fill < (uint32_t)k ? s_fill : (uint32_t)k);
    int npad = 2;
    while (npad < k) npad <<= 1;
    for (int i = nsel + t; i < npad; i += TK_THREADS) keys[i] = 0ull;
    __syncthreads();
    bitonic_sort_desc(keys, npad, t);
  }

  TK_STAMP(5);
  float *kpts = const_cast<float *>(mi_set_item<float>(kpt_sets, img, (size_t)k * 2));
  for (int j = t; j < k; j += TK_THREADS) {
    float y = -1.0f, x = -1.0f, s = 0.0f;
    if (j < nsel) {
      const uint64_t key = keys[j];
      const uint32_t lin = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
      y = (float)(lin / (uint32_t)w);
      x = (float)(lin % (uint32_t)w);
      s = __uint_as_float((uint32_t)(key >> 32));
    }
    kpts[(size_t)j * 2 + 0] = y;
    kpts[(size_t)j * 2 + 1] = x;
    kscores[(size_t)img * k + j] = s;
  }
  TK_STAMP(6);
#undef TK_STAMP
}

}  // namespace

int mi_topk_keypoints_sets(const uint64_t *cand, const uint32_t *count, int segments, int segment_capacity, int n, int w,
                           int k, MiSets keypoints, float *kscores, mi_stream_t stream) {
  if (!cand || !count || !keypoints.a || (keypoints.per_set < n && !keypoints.b) || !kscores) return MI_E_NULL;
  if (n <= 0 || w <= 0 || segments <= 0) return MI_E_SHAPE;
  if (k <= 0 || k > TK_MAX) return MI_E_PARAM;
  if (segment_capacity <= 0) return MI_E_CAPACITY;
  // the big-LDS kernel for large images (one workgroup per CU; see the kernel's note): more than 150 tiles of 128 x 32
  // pixels (~0.6 Mpx) are expected to yield more candidates than the small kernel's key array holds.  Either kernel is
  // correct for any input; the choice is a matter of speed (test hook key 10 forces one of them).
  const int choice = MI_HOOK(topk_split, -1);
  const bool big = choice < 0 ? segments > 150 : choice != 0;
  if (big)
    hipLaunchKernelGGL(topk_kernel<true>, dim3(n), dim3(TK_THREADS), 0, (hipStream_t)stream, cand, count, segments,
                       (uint32_t)segment_capacity, w, k, keypoints, kscores, MI_HOOK(topk_select, 1),
                       MI_HOOK(topk_prof, (unsigned long long *)nullptr));
  else
    hipLaunchKernelGGL(topk_kernel<false>, dim3(n), dim3(TK_THREADS), 0, (hipStream_t)stream, cand, count, segments,
                       (uint32_t)segment_capacity, w, k, keypoints, kscores, MI_HOOK(topk_select, 1),
                       MI_HOOK(topk_prof, (unsigned long long *)nullptr));
  return mi_launch_status();
}

extern "C" int mi_topk_keypoints(const uint64_t *cand, const uint32_t *count, int segments,
                                 int segment_capacity, int n, int w, int k, float *keypoints, float *kscores,
                                 mi_stream_t stream) {
  MI_ENTER();
  return mi_topk_keypoints_sets(cand, count, segments, segment_capacity, n, w, k, mi_one_set(keypoints, n), kscores, stream);
}
