// mi_match_pairs: the whole hot path for a batch of image pairs behind ONE C-ABI call.
// Semantics: reference pytorch_model/feature_detection/match_extraction_wrapper.py:82-113 around
// shi_tomasi_sparse_bad_sinkhorn.py:79-182 with hard-binarised descriptors and the L2 cost, i.e.
//   ShiTomasiScore -> apply_nms_maxpool + select_topk_keypoints -> SparseBAD (bits) -> cost -> 20 x Sinkhorn
//   -> MutualNearestNeighborMatcher,
// for image1[b] against image2[b].  It only sequences the entry points the Python modules call (same
// kernels, same order, same stream), carving every intermediate out of one caller-provided workspace, so a
// host without Python -- a C++/Go/Rust service over cgo/FFI -- gets the path with a single binding and no
// allocation inside the call.  Results are bit-identical to the module path (tests/test_gpu_parity.py).
#include "common.h"

#include <stddef.h>
#include <stdint.h>

#include <type_traits>

namespace {

struct Carver {
  char *base;
  size_t off = 0;
  explicit Carver(void *p) : base(reinterpret_cast<char *>(p)) {}
  template <typename T>
  T *take(size_t count) {
    off = (off + 255) & ~(size_t)255;           // every buffer 256-byte aligned (16 is what the kernels need)
    T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

// Batches up to this many pairs put image1 and image2 behind ONE launch per stage (twice the score / candidate
// workspace, half the dependent launches): the one-pair-per-call latency path.  Larger batches keep one launch per
// image side (every launch already fills the chip; the workspace stays at one side's size).
constexpr int MP_MERGED_MAX_BATCH = 32;

struct Layout {
  int sides;                    // 2 when both images share the detection workspace (merged launches), else 1
  float *score;                 // (sides * batch, h, w); with sides == 1 reused for both images
  uint64_t *cand;               // (batch, segments, capacity)
  uint32_t *count;              // (batch, segments)
  float *kscores;               // (batch, K), scratch (keypoint scores are not an output of the wrapper)
  uint32_t *bits1, *bits2;      // (batch, K, P/32)
  uint8_t *status;              // (sides * batch, K)
  uint32_t *tile_ctr;           // K1's ticket counters (MI_TILE_COUNTER_BYTES, cleared per call)
  uint16_t *dots;               // (batch, K, pitch)
  float *row_info, *col_info;   // (batch, K, 2)
  float *u, *v;                 // (batch, K+1)
  void *sk_ws;
  size_t sk_bytes;
  void *mnn_ws;
  size_t mnn_bytes;
  int segments, capacity, pitch;
  size_t total;
};

int lay_out(void *ws, int batch, int h, int w, int k, int num_pairs, Layout *L) {
  int e = mi_candidate_layout(h, w, &L->segments, &L->capacity);
  if (e) return e;
  L->sides = batch <= MP_MERGED_MAX_BATCH ? 2 : 1;
  const size_t nb = (size_t)L->sides * batch;
  L->pitch = (k + 7) / 8 * 8;
  L->sk_bytes = mi_sinkhorn_dots_workspace_bytes(batch, k, k);
  L->mnn_bytes = mi_mnn_duals_workspace_bytes(batch, k, k);
  if (L->sk_bytes == 0 || L->mnn_bytes == 0) return MI_E_PARAM;       // K > 1024: not the packed form
  Carver c(ws);
  L->score = c.take<float>(nb * h * w);
  L->cand = c.take<uint64_t>(nb * L->segments * L->capacity);
  L->count = c.take<uint32_t>(nb * L->segments);
  L->kscores = c.take<float>(nb * k);
  // bits1 and bits2 are one array (2 * batch, K, P/32): the merged descriptor launch writes both halves
  L->bits1 = c.take<uint32_t>(2 * (size_t)batch * k * (num_pairs / 32));
  L->bits2 = L->bits1 + (size_t)batch * k * (num_pairs / 32);
  L->status = c.take<uint8_t>(nb * k);
  L->tile_ctr = c.take<uint32_t>(MI_TILE_COUNTER_BYTES / 4);
  L->dots = c.take<uint16_t>((size_t)batch * k * L->pitch);
  L->row_info = c.take<float>((size_t)batch * k * 2);
  L->col_info = c.take<float>((size_t)batch * k * 2);
  L->u = c.take<float>((size_t)batch * (k + 1));
  L->v = c.take<float>((size_t)batch * (k + 1));
  L->sk_ws = c.take<char>(L->sk_bytes);
  L->mnn_ws = c.take<char>(L->mnn_bytes);
  L->total = (c.off + 255) & ~(size_t)255;
  return MI_OK;
}

int check_params(const mi_match_params *p) {
  if (!p || !p->pair_geom || !p->pair_thr) return MI_E_NULL;
  if (p->block_size <= 0 || p->block_size % 2 == 0 || p->nms_radius < 0 || p->max_keypoints <= 0) return MI_E_PARAM;
  if (p->num_pairs <= 0 || p->num_pairs % 64 != 0 || p->num_pairs > 1024) return MI_E_PARAM;
  if (p->sinkhorn_iterations <= 0 || !(p->epsilon >= MI_DOTS_MIN_EPSILON) || p->max_matches <= 0) return MI_E_PARAM;
  if ((p->flags & ~(MI_SOLVER_MULTI_LAUNCH | MI_SOLVER_NO_FORK)) != 0) return MI_E_PARAM;
  return MI_OK;
}

}  // namespace

extern "C" size_t mi_match_pairs_workspace_bytes(int batch, int h, int w, const mi_match_params *params) {
  if (batch <= 0 || h <= 0 || w <= 0 || check_params(params) != MI_OK) return 0;
  Layout L;
  if (lay_out(nullptr, batch, h, w, params->max_keypoints, params->num_pairs, &L) != MI_OK) return 0;
  return L.total;
}

namespace {
int corner_of(const float *im, int n, int h, int w, int bs, float *score, uint32_t *ctr, mi_stream_t s) { return mi_corner_response_balanced(im, 0, n, h, w, bs, score, ctr, s); }
int corner_of(const uint8_t *im, int n, int h, int w, int bs, float *score, uint32_t *ctr, mi_stream_t s) { return mi_corner_response_balanced(im, 1, n, h, w, bs, score, ctr, s); }
int bad_bits_of(const float *im, int n, int h, int w, const float *kp, int k, const mi_match_params *p, uint32_t *bits,
                uint8_t *status, mi_stream_t s) {
  return mi_sparse_bad(im, n, h, w, kp, k, p->pair_geom, p->pair_thr, p->num_pairs, MI_BAD_HARD, 0.0f,
                       p->normalize_descriptors, nullptr, bits, p->bad_plan, p->bad_plan ? status : nullptr, s);
}
int bad_bits_of(const uint8_t *im, int n, int h, int w, const float *kp, int k, const mi_match_params *p, uint32_t *bits,
                uint8_t *status, mi_stream_t s) {
  return mi_sparse_bad_u8(im, n, h, w, kp, k, p->pair_geom, p->pair_thr, p->num_pairs, MI_BAD_HARD, 0.0f,
                          p->normalize_descriptors, nullptr, bits, p->bad_plan, p->bad_plan ? status : nullptr, s);
}

template <typename PIX>
int match_pairs_impl(const PIX *image1, const PIX *image2, int batch, int h, int w,
                              const mi_match_params *params, float *keypoints1, float *keypoints2,
                              float *matched1, float *matched2, float *match_scores, uint8_t *match_valid,
                              int32_t *match_ij, void *workspace, size_t workspace_bytes, mi_stream_t stream) {
  if (!image1 || !image2 || !keypoints1 || !keypoints2 || !matched1 || !matched2 || !match_scores || !match_valid ||
      !workspace)
    return MI_E_NULL;
  if (batch <= 0 || h <= 0 || w <= 0) return MI_E_SHAPE;
  int e = check_params(params);
  if (e) return e;
  if (((uintptr_t)workspace % 16) != 0) return MI_E_ALIGN;
  const int k = params->max_keypoints, pbits = params->num_pairs;
  if ((long long)h * w < k) return MI_E_SHAPE;                          // torch.topk's failure mode (k > H*W)
  Layout L;
  if ((e = lay_out(workspace, batch, h, w, k, pbits, &L)) != MI_OK) return e;
  if (workspace_bytes < L.total) return MI_E_CAPACITY;

  constexpr int U8 = std::is_same<PIX, uint8_t>::value ? 1 : 0;
  if (L.sides == 2) {
    // both images of every pair behind one launch per stage (items 0..batch-1 = image1, batch..2*batch-1 = image2)
    const MiSets imgs{image1, image2, batch}, kps{keypoints1, keypoints2, batch};
    const int n2 = 2 * batch;
    if ((e = mi_corner_response_sets(imgs, U8, n2, h, w, params->block_size, L.score, nullptr, stream)) != MI_OK) return e;
    if ((e = mi_nms_candidates(L.score, n2, h, w, params->nms_radius, params->score_threshold, params->border_margin, L.cand,
                               L.count, stream)) != MI_OK)
      return e;
    if ((e = mi_topk_keypoints_sets(L.cand, L.count, L.segments, L.capacity, n2, w, k, kps, L.kscores, stream)) != MI_OK) return e;
    if ((e = mi_sparse_bad_sets(imgs, U8, n2, h, w, kps, k, params->pair_geom, params->pair_thr, pbits, MI_BAD_HARD, 0.0f,
                                params->normalize_descriptors, nullptr, L.bits1, params->bad_plan,
                                params->bad_plan ? L.status : nullptr, stream)) != MI_OK)
      return e;
  } else {
    // one image side per launch; K1 hands its tiles out dynamically (mi_corner_response_balanced clears the counter block)
    const PIX *images[2] = {image1, image2};
    float *kpts[2] = {keypoints1, keypoints2};
    uint32_t *bits[2] = {L.bits1, L.bits2};
    for (int side = 0; side < 2; ++side) {
      // detector/shi_tomasi.py:66-112, utils/keypoint_utils.py:12-117 (mask never materialised)
      if ((e = corner_of(images[side], batch, h, w, params->block_size, L.score, L.tile_ctr, stream)) != MI_OK) return e;
      if ((e = mi_nms_candidates(L.score, batch, h, w, params->nms_radius, params->score_threshold, params->border_margin,
                                 L.cand, L.count, stream)) != MI_OK)
        return e;
      if ((e = mi_topk_keypoints(L.cand, L.count, L.segments, L.capacity, batch, w, k, kpts[side], L.kscores, stream)) !=
          MI_OK)
        return e;
      // descriptor/bad.py:436-576, hard bits, packed
      if ((e = bad_bits_of(images[side], batch, h, w, kpts[side], k, params, bits[side], L.status, stream)) != MI_OK) return e;
    }
  }
  // matching/sinkhorn.py:79-208 in the packed (uint16 dot product) form; P is never written
  // (the cost kernel also clears the hand-off area of the single-launch Sinkhorn: one graph node less per call)
  void *handoff = nullptr;
  const size_t handoff_bytes = mi_sinkhorn_dots_handoff_region(L.sk_ws, batch, k, k, params->flags, &handoff);
  if ((e = mi_cost_dots_bits_zeroing(L.bits1, L.bits2, batch, k, k, pbits, params->normalize_descriptors, L.dots, L.pitch,
                                     L.row_info, L.col_info, handoff, handoff_bytes, stream)) != MI_OK)
    return e;
  const double sqnorm_bound = params->normalize_descriptors ? 1.0 : (double)pbits;
  if ((e = mi_sinkhorn_dots_impl(L.dots, L.row_info, L.col_info, batch, k, k, L.pitch, params->epsilon,
                                 params->unused_score, sqnorm_bound, params->sinkhorn_iterations, L.u, L.v, nullptr, L.sk_ws,
                                 L.sk_bytes, params->flags | (pbits < 1024 ? MI_SOLVER_DOTS_BELOW_1024 : 0), handoff_bytes > 0,
                                 stream)) != MI_OK)
    return e;
  // matching/match_extraction.py:46-184 straight from the duals
  return mi_mnn_from_duals_dots(L.dots, L.row_info, L.col_info, batch, k, k, L.pitch, params->epsilon, L.u, L.v,
                                keypoints1, keypoints2, params->max_matches, params->match_threshold, L.mnn_ws,
                                L.mnn_bytes, mi_sinkhorn_dots_status_word(L.sk_ws, batch, k, k), matched1, matched2,
                                match_scores, match_valid, match_ij, stream);
}
}  // namespace

extern "C" int mi_match_pairs(const float *image1, const float *image2, int batch, int h, int w,
                              const mi_match_params *params, float *keypoints1, float *keypoints2,
                              float *matched1, float *matched2, float *match_scores, uint8_t *match_valid,
                              int32_t *match_ij, void *workspace, size_t workspace_bytes, mi_stream_t stream) {
  MI_ENTER();
  return match_pairs_impl<float>(image1, image2, batch, h, w, params, keypoints1, keypoints2, matched1, matched2,
                                 match_scores, match_valid, match_ij, workspace, workspace_bytes, stream);
}

// u8 ingest: the same call on uint8 frames (same workspace size; results identical to the float32 call on the
// converted frames)
extern "C" int mi_match_pairs_u8(const uint8_t *image1, const uint8_t *image2, int batch, int h, int w,
                                 const mi_match_params *params, float *keypoints1, float *keypoints2,
                                 float *matched1, float *matched2, float *match_scores, uint8_t *match_valid,
                                 int32_t *match_ij, void *workspace, size_t workspace_bytes, mi_stream_t stream) {
  MI_ENTER();
  return match_pairs_impl<uint8_t>(image1, image2, batch, h, w, params, keypoints1, keypoints2, matched1, matched2,
                                   match_scores, match_valid, match_ij, workspace, workspace_bytes, stream);
}
