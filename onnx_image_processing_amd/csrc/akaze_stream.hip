// K9, streaming form: one AKAZE scale (ITERS explicit diffusion steps + Hessian determinant + window-equality NMS,
// reference pytorch_model/detector/akaze.py:98-131, 190-254) as a ROLLING WINDOW that walks down the image.
//
// The LDS-tile form (akaze.hip: akaze_scale_kernel) stages a 64 x 66 tile, recomputes a 9-pixel halo on all four sides
// (1.58x the arithmetic, 1.70x the HBM traffic) and meets a workgroup barrier between every phase.  Here a WAVE owns a
// strip of 128 columns (two adjacent columns per lane) and a range of rows, and every stage of the scale is a software
// pipeline stage that lags the previous one by one row:
//
//   tick t:  L0 row y            (8-byte load per lane, issued three ticks earlier)
//            flux_1 row y-1, L1 row y-2, flux_2 row y-3, L2 row y-4, flux_3 row y-5, L3 row y-6   -> l_out
//            Hessian response row y-7, its row-window maximum, NMS output row y-7-NH              -> scores
//
// Nothing is staged: the 3 x 3 stencils are kept as PARTIAL SUMS in registers.  The reference adds a stencil's taps in
// row-major order (akaze.hip's header note), i.e. top row, middle row, bottom row -- so when a row arrives it finishes
// the stencil of the row above (bottom taps), continues the one centred on it (middle taps) and starts the one below (top
// taps): the same additions in the same order as the tile kernel, bit for bit, with 22 registers of state per diffusion
// step instead of a 3-row window of every map.  Horizontal neighbours come from the adjacent lanes by DPP wave shifts
// (v_mov_b32_dpp wave_shr:1 / wave_shl:1): no LDS, no barrier, no workgroup -- waves are independent.  Only the strip's
// 19 halo columns are recomputed (128 columns for 108 outputs); vertically a wave runs 2 * HALO extra ticks for its
// row range (HALO rows of context above it, and the pipeline's lag of HALO rows below).  HBM traffic: the image rows once (+ the strip halo, mostly L2 hits), l_out and the scores once.
//
// MODE 1 (the last scale): instead of this scale's score map the kernel writes AKAZE.forward's selection across scales
// (akaze.py:436-451) -- best = max over the earlier scales' maps and this one, attain = the set of scales (bit s) whose
// score equals it -- so the separate max-over-scales pass (16 B/px) and the last scale's score map disappear.
#include "akaze_math.h"
#include "common.h"

namespace {

constexpr int ST_COLS = 128;     // tile columns of a wave: 2 per lane
constexpr int ST_UNROLL = 6;     // ticks per loop iteration = lcm of the ring periods (2, 3 and 6)
constexpr int ST_PF = 3;         // rows in flight ahead of the one being consumed

// (mov_dpp, not update_dpp: with bound_ctrl every lane is written, so there is no `old` value to initialise -- update_dpp
// cost one v_mov_b32 0 per shift, 5 % of the kernel's instructions)
__device__ __forceinline__ float lane_left(float v) {        // lane i <- lane i-1; lane 0 <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_right(float v) {       // lane i <- lane i+1; lane 63 <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
// v_max_f32 / v_max3_f32 as written (fmaxf() first quiets a possible signalling NaN: one more instruction per operand)
__device__ __forceinline__ float smax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float smax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// a row of one map as a lane sees it: its own two columns (b, c) and the neighbours' adjacent ones (a, d)
struct Row4 {
  float a, b, c, d;
};
__device__ __forceinline__ Row4 with_edges(float b, float c) { return Row4{lane_left(c), b, c, lane_right(b)}; }
// (left, middle, right) of own column j
#define ROW_L(r, j) ((j) == 0 ? (r).a : (r).b)
#define ROW_M(r, j) ((j) == 0 ? (r).b : (r).c)
#define ROW_R(r, j) ((j) == 0 ? (r).c : (r).d)

// c * x + acc with c in {+-2, +-4}: c * x is exact, so the fused form rounds like the reference's separate multiply + add
#define AXPY(c, x, acc) __builtin_fmaf((c), (x), (acc))

template <int ITERS, int NH, int NP>
struct StreamState {
  static constexpr int RS = (NH == 1) ? 3 : 6;          // row-maximum ring (2 NH + 1 live rows; period divides ST_UNROLL)
  static constexpr int RC = NH + 1;                     // responses waiting for their window to fill
  float2 fifo[ST_PF];                                   // prefetched image rows
  float2 prevq[2][NP > 0 ? NP : 1];                     // select mode: the earlier scales' scores, two ticks ahead
  float lc[ITERS][3][2];                                // L_s at rows y, y-1, y-2 (own columns): the update's centre value
  float gxA[ITERS][2], gxB[ITERS][2], gyT[ITERS][2][2]; // gradient stencils in flight (A: top seen, B: top + middle)
  float dxA[ITERS][2], dxB[ITERS][2], dyT[ITERS][2][2]; // divergence stencils in flight
  float hxxA[2], hxxB[2], hyyA[2], hyyB[2], hxyT[2][2]; // Hessian stencils in flight
  float rm[RS][2];                                      // row-window maxima of the response
  float rc[RC][2];                                      // the responses themselves
};

struct StreamArgs {
  const float *lin, *lin_b;      // images [0, per_set) from lin, the rest from lin_b (two batches behind one launch)
  int per_set;
  float *lout, *scores;
  const float *prev_scores;      // select mode: (num_prev, n, h, w)
  uint8_t *attain;               // select mode: (n, h, w)
  size_t prev_stride;            // n * h * w
  int n, h, w;
  int strips, chunks, rows_per_chunk;
  float kappa, dt, threshold;
};

// Every memory access is a BUFFER instruction on a per-image resource (base = the image's plane, num_records = its
// bytes): an access whose byte offset lies outside the plane reads 0 / is dropped by the hardware's range check.  Rows
// and columns outside the image, halo lanes and rows outside the wave's range are all expressed as the offset ST_BAD --
// zero padding without a select, predicated stores without a branch.  The loop body is one basic block, so the
// compiler's s_waitcnt vmcnt(N) counts are exact and the prefetch distance is real (with `if (inside) store` every
// store was its own branch, the counts collapsed to vmcnt(1) and a row had to arrive within one tick of its request).
constexpr int ST_BAD = 0x40000000;   // >= any plane's bytes (the host checks h * w * 4 < 2^30), and so is ST_BAD + ST_BAD
typedef unsigned int st_u32x2 __attribute__((ext_vector_type(2)));

template <int ITERS, int NH, int NP>
struct StreamCtx {
  __amdgpu_buffer_rsrc_t in, lout, sout, att;
  __amdgpu_buffer_rsrc_t prev[NP > 0 ? NP : 1];
  int h, rowbytes, w;
  int xin, xout, xatt;           // the lane's byte offset in a row for loads / stores / attain stores (ST_BAD: none)
  int ya, yb;                    // output rows of this wave
  unsigned cmask;                // all ones when the lane's columns lie inside the image, else 0
  float kappa, rkappa, dt, thr;
};

__device__ __forceinline__ float2 buf_load2(__amdgpu_buffer_rsrc_t r, int voff) {
  const st_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
  return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ void buf_store2(__amdgpu_buffer_rsrc_t r, int voff, float a, float b) {
  st_u32x2 v;
  v.x = __float_as_uint(a);
  v.y = __float_as_uint(b);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, 0, 0);
}
// Zero padding by BIT MASKS, not selects: `cond ? expensive : 0` makes the compiler branch around the expensive side
// (an exec-masked block per pixel: eight per tick, each a scheduling barrier), and v_cndmask issues at half the rate of
// v_and.  mask = lane's column mask & row mask (wave-uniform): one v_and per predicate and tick, one per masked value.
__device__ __forceinline__ float fmask(float v, unsigned m) { return __uint_as_float(__float_as_uint(v) & m); }
__device__ __forceinline__ unsigned rmask(int y, uint32_t uh) { return (uint32_t)y < uh ? 0xffffffffu : 0u; }
// byte offset of row y of a plane with `rowbytes` per row, ST_BAD when y is not in [lo, hi) (wave-uniform)
__device__ __forceinline__ int row_off(int y, int lo, int hi, int rowbytes) { return (y >= lo && y < hi) ? y * rowbytes : ST_BAD; }

// one tick: image row y_in enters, output row y_in - (2 ITERS + 1 + NH) leaves.  P = tick index mod ST_UNROLL.
// NP < 0: this scale's score map is written; NP >= 0: the selection over NP earlier maps and this one.
template <int ITERS, int NH, int NP, int P>
__device__ __forceinline__ void tick(StreamState<ITERS, NH, NP> &st, const StreamCtx<ITERS, NH, NP> &cx, int y_in) {
  using S = StreamState<ITERS, NH, NP>;
  constexpr int P2 = P % 2, P3 = P % 3;
  const uint32_t uh = (uint32_t)cx.h;
  // ---- the image row (requested ST_PF ticks ago; 0 outside the image: both convolutions zero-pad); the slot is
  // refilled with the row ST_PF further down
  const float2 raw = st.fifo[P3];
  st.fifo[P3] = buf_load2(cx.in, cx.xin + row_off(y_in + ST_PF, 0, cx.h, cx.rowbytes));
  // select mode: the earlier scales' scores of the row that leaves at the end of this tick (requested two ticks ago)
  const int yo = y_in - (2 * ITERS + 1 + NH);
  float2 prev[NP > 0 ? NP : 1];
  if (NP > 0) {
    const int po = cx.xout + row_off(yo + 2, cx.ya, cx.yb, cx.rowbytes);
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      prev[s] = st.prevq[P2][s];
      st.prevq[P2][s] = buf_load2(cx.prev[s], po);
    }
  }
  Row4 row = with_edges(raw.x, raw.y);
  int yrow = y_in;
#pragma unroll
  for (int s = 0; s < ITERS; ++s) {
    // ---- flux of step s at row yrow - 1 (akaze.py:82-96,116): `row` is its bottom row
    const unsigned fin = cx.cmask & rmask(yrow - 1, uh);          // the flux is zero-padded outside the image
    float fx[2], fy[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float l = ROW_L(row, j), m = ROW_M(row, j), r = ROW_R(row, j);
      const float gxv = ((st.gxB[s][j] - l) + r) * 0.125f;
      const float gyv = (AXPY(2.0f, m, st.gyT[s][P2][j] + l) + r) * 0.125f;
      st.gxB[s][j] = AXPY(2.0f, r, AXPY(-2.0f, l, st.gxA[s][j]));
      st.gxA[s][j] = r - l;
      st.gyT[s][P2][j] = AXPY(-2.0f, m, -l) - r;
      const float mag = ak_sqrt_fp<1>(gxv * gxv + gyv * gyv + 1e-8f);
      const float q = ak_div_by(mag, cx.kappa, cx.rkappa);
      const float cond = ak_rcp(1.0f + q * q);
      fx[j] = fmask(cond * gxv, fin);
      fy[j] = fmask(cond * gyv, fin);
      st.lc[s][P3][j] = m;
    }
    // ---- L_{s+1} at row yrow - 2 (akaze.py:125-129): the flux row just made is its bottom row
    const Row4 rx = with_edges(fx[0], fx[1]), ry = with_edges(fy[0], fy[1]);
    const unsigned uin = cx.cmask & rmask(yrow - 2, uh);          // only pixels of the image evolve
    float ln[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float xl = ROW_L(rx, j), xr = ROW_R(rx, j);
      const float yl = ROW_L(ry, j), ym = ROW_M(ry, j), yr = ROW_R(ry, j);
      const float dx = ((st.dxB[s][j] - xl) + xr) * 0.125f;
      const float dy = (AXPY(2.0f, ym, st.dyT[s][P2][j] + yl) + yr) * 0.125f;
      st.dxB[s][j] = AXPY(2.0f, xr, AXPY(-2.0f, xl, st.dxA[s][j]));
      st.dxA[s][j] = xr - xl;
      st.dyT[s][P2][j] = AXPY(-2.0f, ym, -yl) - yr;
      const float centre = st.lc[s][(P3 + 1) % 3][j];             // L_s at row yrow - 2 (stored two ticks ago)
      ln[j] = fmask(centre + cx.dt * (dx + dy), uin);
    }
    row = with_edges(ln[0], ln[1]);
    yrow -= 2;
  }
  // ---- the diffused row yrow = y_in - 2 ITERS
  buf_store2(cx.lout, cx.xout + row_off(yrow, cx.ya, cx.yb, cx.rowbytes), row.b, row.c);
  // ---- Hessian determinant at row yrow - 1 (akaze.py:153-171,196); -inf outside the image (the pool's padding)
  const unsigned hin = cx.cmask & rmask(yrow - 1, uh);
  const unsigned hout = ~hin & 0xff800000u;                       // -inf where hin is 0
  float resp[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float l = ROW_L(row, j), m = ROW_M(row, j), r = ROW_R(row, j);
    const float lxx = (AXPY(-2.0f, m, st.hxxB[j] + l) + r) * 0.0625f;
    const float lyy = (AXPY(2.0f, m, st.hyyB[j] + l) + r) * 0.0625f;
    const float lxy = ((st.hxyT[P2][j] - l) + r) * 0.25f;
    st.hxxB[j] = AXPY(2.0f, r, AXPY(-4.0f, m, AXPY(2.0f, l, st.hxxA[j])));
    st.hyyB[j] = AXPY(-2.0f, r, AXPY(-4.0f, m, AXPY(-2.0f, l, st.hyyA[j])));
    st.hxxA[j] = AXPY(-2.0f, m, l) + r;
    st.hyyA[j] = AXPY(2.0f, m, l) + r;
    st.hxyT[P2][j] = l - r;
    const float det = lxx * lyy - lxy * lxy;
    resp[j] = __uint_as_float((__float_as_uint(det) & hin) | hout);
  }
  // row-window maximum over columns -NH .. +NH (neighbour lanes hold columns -2, -1 | 2, 3)
  {
    const float lb = lane_left(resp[0]), lcv = lane_left(resp[1]);
    const float rb = lane_right(resp[0]), rcv = lane_right(resp[1]);
    const float core = smax(resp[0], resp[1]);
    if (NH == 1) {
      st.rm[P % S::RS][0] = smax(core, lcv);
      st.rm[P % S::RS][1] = smax(core, rb);
    } else {
      const float mid = smax3(core, lcv, rb);
      st.rm[P % S::RS][0] = smax(mid, lb);
      st.rm[P % S::RS][1] = smax(mid, rcv);
    }
    st.rc[P % S::RC][0] = resp[0];
    st.rc[P % S::RC][1] = resp[1];
  }
  // ---- NMS output row yo = yrow - 1 - NH (akaze.py:214-223,245-252)
  float sc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float mx = st.rm[P % S::RS][j];
#pragma unroll
    for (int k = 1; k <= 2 * NH; ++k) mx = smax(mx, st.rm[(P + S::RS * 8 - k) % S::RS][j]);
    const float rv = st.rc[(P + 1) % S::RC][j];                   // the response NH ticks ago
    sc[j] = (rv == mx && rv > cx.thr) ? smax(rv, 0.0f) : 0.0f;
  }
  const int oo = row_off(yo, cx.ya, cx.yb, cx.rowbytes);
  if (NP < 0) {
    buf_store2(cx.sout, cx.xout + oo, sc[0], sc[1]);
  } else {
    // AKAZE.forward's selection across scales (akaze.py:436-451): the maximum and the set of scales that reach it
    float bx = sc[0], by = sc[1];
#pragma unroll
    for (int s = 0; s < NP; ++s) { bx = smax(bx, prev[s].x); by = smax(by, prev[s].y); }
    unsigned ax = (sc[0] == bx) ? (1u << (NP > 0 ? NP : 0)) : 0u, ay = (sc[1] == by) ? (1u << (NP > 0 ? NP : 0)) : 0u;
#pragma unroll
    for (int s = 0; s < NP; ++s) { ax |= (prev[s].x == bx) ? (1u << s) : 0u; ay |= (prev[s].y == by) ? (1u << s) : 0u; }
    buf_store2(cx.sout, cx.xout + oo, bx, by);
    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(ax | (ay << 8)), cx.att,
                                          cx.xatt + row_off(yo, cx.ya, cx.yb, cx.w), 0, 0);
  }
}

template <int ITERS, int NH, int NP>
__global__ __launch_bounds__(64) void akaze_stream_kernel(StreamArgs a) {
  constexpr int HALO = 2 * ITERS + 1 + NH;
  constexpr int HL = (HALO + 1) & ~1;                    // left halo, even: 8-byte aligned lane columns
  constexpr int OUTW = (ST_COLS - HL - HALO) & ~1;       // output columns of a strip
  constexpr int RSRC_FLAGS = 0x00020000;                 // gfx9 raw buffer: DATA_FORMAT = 32, no swizzle, range-checked
  // wave -> (image, row chunk, strip); neighbouring strips / chunks (shared halo) sit on one XCD
  unsigned id = xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int strip = (int)(id % (unsigned)a.strips);
  id /= (unsigned)a.strips;
  const int chunk = (int)(id % (unsigned)a.chunks);
  const int img = (int)(id / (unsigned)a.chunks);
  const int lane = threadIdx.x;
  const int tc = 2 * lane;                               // tile column of the lane's first pixel
  const int x = strip * OUTW - HL + tc;                  // image column (even)
  StreamCtx<ITERS, NH, NP> cx;
  const size_t plane = (size_t)a.h * a.w;
  const int pbytes = (int)(plane * 4);
  const float *in_plane = img < a.per_set ? a.lin + (size_t)img * plane : a.lin_b + (size_t)(img - a.per_set) * plane;
  cx.in = __builtin_amdgcn_make_buffer_rsrc((void *)in_plane, 0, pbytes, RSRC_FLAGS);
  cx.lout = __builtin_amdgcn_make_buffer_rsrc((void *)(a.lout + (size_t)img * plane), 0, pbytes, RSRC_FLAGS);
  cx.sout = __builtin_amdgcn_make_buffer_rsrc((void *)(a.scores + (size_t)img * plane), 0, pbytes, RSRC_FLAGS);
  if (NP >= 0) cx.att = __builtin_amdgcn_make_buffer_rsrc((void *)(a.attain + (size_t)img * plane), 0, (int)plane, RSRC_FLAGS);
#pragma unroll
  for (int s = 0; s < NP; ++s)
    cx.prev[s] = __builtin_amdgcn_make_buffer_rsrc((void *)(a.prev_scores + (size_t)s * a.prev_stride + (size_t)img * plane), 0,
                                                   pbytes, RSRC_FLAGS);
  cx.h = a.h;
  cx.w = a.w;
  cx.rowbytes = a.w * 4;
  cx.ya = chunk * a.rows_per_chunk;
  cx.yb = min(cx.ya + a.rows_per_chunk, a.h);
  const bool cin = x >= 0 && x < a.w;                    // w is even: both columns or neither
  const bool cout = tc >= HL && tc < HL + OUTW && x < a.w;
  cx.cmask = cin ? 0xffffffffu : 0u;
  cx.xin = cin ? x * 4 : ST_BAD;
  cx.xout = cout ? x * 4 : ST_BAD;
  cx.xatt = cout ? x : ST_BAD;
  cx.kappa = a.kappa;
  cx.rkappa = 1.0f / a.kappa;                            // IEEE division: the correctly rounded reciprocal (ak_div_by)
  cx.dt = a.dt;
  cx.thr = a.threshold;

  StreamState<ITERS, NH, NP> st = {};
  const int y_first = cx.ya - HALO;
  const int ticks = cx.yb - cx.ya + 2 * HALO;            // image rows ya - HALO .. yb - 1 + HALO enter; row yb - 1 leaves last
#pragma unroll
  for (int q = 0; q < ST_PF; ++q) st.fifo[q] = buf_load2(cx.in, cx.xin + row_off(y_first + q, 0, cx.h, cx.rowbytes));
  if (NP > 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int s = 0; s < NP; ++s)
        st.prevq[q][s] = buf_load2(cx.prev[s], cx.xout + row_off(y_first + q - HALO, cx.ya, cx.yb, cx.rowbytes));
  }
  for (int t = 0; t < ticks; t += ST_UNROLL) {
    const int y = y_first + t;
    tick<ITERS, NH, NP, 0>(st, cx, y);
    tick<ITERS, NH, NP, 1>(st, cx, y + 1);
    tick<ITERS, NH, NP, 2>(st, cx, y + 2);
    tick<ITERS, NH, NP, 3>(st, cx, y + 3);
    tick<ITERS, NH, NP, 4>(st, cx, y + 4);
    tick<ITERS, NH, NP, 5>(st, cx, y + 5);
  }
}

}  // namespace

// rows per wave: the launch is `n * strips * chunks` independent waves on `simds` SIMDs that hold `waves_per_simd`
// of them; a SIMD needs two waves to issue at full rate and its most loaded round decides.  Cost model in ticks.
static int pick_chunks(int n, int h, int strips, int halo, int simds, int waves_per_simd) {
  int best = 1;
  double best_cost = 1e300;
  const int max_chunks = h / 16 > 0 ? h / 16 : 1;
  for (int c = 1; c <= max_chunks && c <= 64; ++c) {
    const int rows = (h + c - 1) / c;
    const int used = (h + rows - 1) / rows;
    if (used != c) continue;
    const double ticks = (double)((rows + 2 * halo + ST_UNROLL - 1) / ST_UNROLL * ST_UNROLL);
    long long waves = (long long)n * strips * c;
    double cost = 0.0;
    while (waves > 0) {
      const long long cap = (long long)simds * waves_per_simd;
      const long long now = waves < cap ? waves : cap;
      const double k = (double)((now + simds - 1) / simds);
      cost += ticks * (k < 1.85 ? 1.85 : k);
      waves -= now;
    }
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

int mi_akaze_stream_supported(int h, int w, int iterations, int nms_size, const void *l_in, const void *l_out,
                              const void *scores) {
  if (iterations < 1 || iterations > 3 || (nms_size != 3 && nms_size != 5)) return 0;
  if (w < 2 || (w & 1) || h < 1 || (long long)h * w * 4 >= 0x40000000LL) return 0;   // (buffer offsets, ST_BAD)
  if ((((uintptr_t)l_in | (uintptr_t)l_out | (uintptr_t)scores) & 7u) != 0) return 0;
  return 1;
}

// mode 0: scores = this scale's score map.  mode 1: scores = max over prev_scores[0..num_prev) and this scale's map,
// attain = which of them reach it (num_prev <= MI_AKAZE_STREAM_MAX_PREV: one kernel instance per count).  Returns MI_E_PARAM when the streaming form does not apply (caller falls back).
int mi_akaze_scale_stream(const float *l_in, const float *l_in_b, int per_set, int n, int h, int w, int iterations, float kappa, float dt, float threshold,
                          int nms_size, float *l_out, float *scores, int mode, const float *prev_scores, int num_prev,
                          uint8_t *attain, mi_stream_t stream) {
  if (!mi_akaze_stream_supported(h, w, iterations, nms_size, l_in, l_out, scores)) return MI_E_PARAM;
  if (per_set < n && (!l_in_b || ((uintptr_t)l_in_b & 7u) != 0)) return MI_E_PARAM;
  if (mode == 1 && (num_prev < 0 || num_prev > MI_AKAZE_STREAM_MAX_PREV || !attain || (num_prev > 0 && !prev_scores) ||
                    ((uintptr_t)prev_scores & 7u) != 0 || ((uintptr_t)attain & 1u) != 0))
    return MI_E_PARAM;
  const int nh = nms_size / 2, halo = 2 * iterations + 1 + nh;
  const int outw = (ST_COLS - ((halo + 1) & ~1) - halo) & ~1;
  StreamArgs a;
  a.lin = l_in; a.lin_b = l_in_b; a.per_set = per_set; a.lout = l_out; a.scores = scores;
  a.prev_scores = prev_scores; a.attain = attain; a.prev_stride = (size_t)n * h * w;
  a.n = n; a.h = h; a.w = w;
  a.strips = ceil_div(w, outw);
  a.kappa = kappa; a.dt = dt; a.threshold = threshold;
  static int simds = 0;                                   // compute units of the current device x 4 (asked once)
  if (simds == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    (void)hipGetLastError();
    simds = cus * 4;
  }
  a.chunks = pick_chunks(n, h, a.strips, halo, simds, 3);
  a.rows_per_chunk = ceil_div(h, a.chunks);
  const long long blocks = (long long)n * a.strips * a.chunks;
  if (blocks > 0x7fffffffLL) return MI_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
#define AKST1(I, NHALF, NPREV) hipLaunchKernelGGL((akaze_stream_kernel<I, NHALF, NPREV>), dim3((unsigned)blocks), dim3(64), 0, s, a)
#define AKST(I, NHALF)                                                                                              \
  do {                                                                                                              \
    if (mode == 0) AKST1(I, NHALF, -1);                                                                             \
    else if (num_prev == 0) AKST1(I, NHALF, 0);                                                                     \
    else if (num_prev == 1) AKST1(I, NHALF, 1);                                                                     \
    else if (num_prev == 2) AKST1(I, NHALF, 2);                                                                     \
    else AKST1(I, NHALF, 3);                                                                                        \
  } while (0)
  if (iterations == 1) { if (nh == 1) AKST(1, 1); else AKST(1, 2); }
  else if (iterations == 2) { if (nh == 1) AKST(2, 1); else AKST(2, 2); }
  else { if (nh == 1) AKST(3, 1); else AKST(3, 2); }
#undef AKST1
#undef AKST
  return mi_launch_status();
}
