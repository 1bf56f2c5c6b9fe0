/*
 * mi355x_match.h -- C ABI of the MI355X-native image-matching hot path.
 *
 * The reference (fateshelled/onnx_image_processing) has no FFI: its boundary for this
 * path is the Python nn.Module.forward() signatures under pytorch_model/{detector,utils,
 * descriptor,matching}.  Each entry point below is what a binding for one of those
 * forward()s calls; the reference interface it replaces is cited per function
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd / torch.Tensor.data_ptr() on ROCm);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued
 *     asynchronously and is ordered as if it ran on `stream` (after earlier work on it, before later
 *     work), nothing synchronises, nothing allocates device memory.  mi_sinkhorn_dots (and
 *     mi_match_pairs through it; mi_sinkhorn with a workspace likewise) overlaps the halves of a batch of >= 64 pairs on helper streams
 *     joined back into `stream` by events; those helpers belong to the calling (device, stream) and
 *     are created on its first such call (mi_release_stream_resources frees them).  WHICH streams the
 *     halves run on is tuned per calling stream and per shape (batch, n, m, iterations): the first
 *     9 such calls of a shape try three schedules -- halves on {stream, helper}, on {helper, helper},
 *     unsplit -- three times each, every trial call bracketed by two hipEventRecord on `stream`; later
 *     calls read the elapsed times without waiting (hipEventQuery) and use the fastest schedule; the
 *     decision is re-measured every 8192 calls.  Same results bit for bit whatever the schedule.  Inside
 *     a stream capture nothing is tried, recorded or queried: the capture takes the decision in force,
 *     or the unsplit schedule when there is none.  mi_sinkhorn_dots_schedule reports the schedule in
 *     force, mi_sinkhorn_dots_set_schedule pins one, and the flag MI_SOLVER_NO_FORK keeps every kernel
 *     of the call on `stream` (no helper stream, no event, no tuning);
 *   - threads and devices: the device of `stream` must be the calling thread's current device.  Calls
 *     on DIFFERENT streams may run concurrently from different host threads; calls on the SAME
 *     stream must be serialised by the caller (as for any HIP stream).  Mutable state of the library:
 *     (a) per calling (device, stream): the helper streams / events above and the schedule tuner's
 *     trial times and decisions (mutex-guarded); (b) per device: one cached occupancy query (the
 *     single-launch Sinkhorn form) and the compute-unit count.  No process-wide switches (the
 *     kernel-variant test hooks of include/mi355x_match_debug.h exist only in the separate
 *     libmi355x_match_debug.so);
 *   - co-residency: ONE kernel of this library needs its whole grid resident on the device at the same
 *     time -- the single-launch Sinkhorn form mi_sinkhorn_dots / mi_match_pairs use for <= 8 pairs
 *     (<= 128 workgroups of 512 threads whose bands hand column sums to each other inside the launch).
 *     The library checks on the host that the grid fits the device it sees (occupancy x compute units;
 *     a CU-masked or partitioned device takes the multi-launch form instead), and work on other streams
 *     only delays it: its workgroups become resident as that work drains.  A band that still has not
 *     arrived after ~1 s of polling is a failure, and a loud one: the call's status word
 *     (mi_sinkhorn_dots_status_word) becomes non-zero, the pair's duals are NaN (so is P), and
 *     mi_mnn_from_duals_dots / mi_match_pairs return valid = 0 for every match of the call.  A caller
 *     that cannot accept that risk (a GPU shared with long-running foreign kernels) passes
 *     MI_SOLVER_MULTI_LAUNCH and gets the form with no cross-workgroup dependency;
 *   - tensors are dense row-major float32 unless stated; images are (n, 1, h, w);
 *   - return value: 0 = launched; > 0 = hipError_t of the failed launch; < 0 = MI_E_*
 *     argument error detected on the host before any launch.
 */
#ifndef MI355X_MATCH_H
#define MI355X_MATCH_H

#include <stddef.h>
#include <stdint.h>

/* the exported symbols: the library is built with -fvisibility=hidden, only these declarations are visible */
#define MI_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

typedef void *mi_stream_t;

enum {
  MI_OK = 0,
  MI_E_NULL = -1,     /* required pointer is NULL */
  MI_E_SHAPE = -2,    /* non-positive or inconsistent extent */
  MI_E_PARAM = -3,    /* parameter outside the supported set */
  MI_E_CAPACITY = -4, /* workspace / capacity too small for the request */
  MI_E_ALIGN = -5     /* pointer or pitch not aligned as documented */
};

/* descriptor output modes of mi_sparse_bad (reference descriptor/bad.py:561-567) */
enum { MI_BAD_RAW = 0, MI_BAD_SOFT = 1, MI_BAD_HARD = 2 };
/* distance types (reference matching/sinkhorn.py:95-108) */
enum { MI_DIST_L2 = 0, MI_DIST_L1 = 1 };
/* smallest epsilon of the packed (uint16 dot product) Sinkhorn form, see mi_sinkhorn_dots */
#define MI_DOTS_MIN_EPSILON 0.005
/* flags of mi_sinkhorn_dots / mi_match_params (see "co-residency" above) */
enum {
  MI_SOLVER_DEFAULT = 0,
  MI_SOLVER_MULTI_LAUNCH = 1, /* never the single-launch Sinkhorn form */
  MI_SOLVER_NO_FORK = 2,      /* never fork onto helper streams: every kernel of the call on `stream` (e.g. for a
                                 capture that must not contain cross-stream branches, or a device with one hardware
                                 queue); bit-identical results, the >= 64-pair Sinkhorn runs ~8 % slower */
  MI_SOLVER_DOTS_BELOW_1024 = 4 /* mi_sinkhorn_dots only: the caller vouches that every dot product is < 1024 (descriptors
                                 of at most 1023 bits -- mi_match_pairs sets it by itself from num_bits).  The row
                                 kernel of the batched form then reads a uint16 as the fp16 denormal dot * 2^-24 and
                                 multiplies in one mixed-precision instruction instead of converting first; the same
                                 duals bit for bit.  A value >= 1024 under this flag reads as some other fp16: wrong
                                 duals, no fault. */
};
/* stream schedules of mi_sinkhorn_dots for >= 64 pairs (see the conventions above) */
enum {
  MI_SCHEDULE_UNDECIDED = -1,  /* nothing decided or pinned yet for this stream / shape */
  MI_SCHEDULE_CALLER_HELPER = 0, /* halves on {caller's stream, helper 0} */
  MI_SCHEDULE_TWO_HELPERS = 1, /* halves on {helper 0, helper 1} */
  MI_SCHEDULE_UNSPLIT = 2      /* one part on the caller's stream */
};

MI_API int mi_abi_version(void);
MI_API const char *mi_error_string(int code);
/* Frees the helper streams / events held for (current device, stream), see the conventions above.  Call it
 * before destroying a stream that was passed to mi_sinkhorn_dots / mi_sinkhorn / mi_match_pairs with >= 64 pairs; the
 * stream's helper work must have completed (synchronise the stream first). */
MI_API int mi_release_stream_resources(mi_stream_t stream);
/* The stream schedule in force for mi_sinkhorn_dots calls of this shape on (current device, stream): the pinned
 * schedule, the tuner's decision, or MI_SCHEDULE_UNDECIDED.  Never blocks (finished trials are collected with
 * hipEventQuery).  Hosts log it next to their timings; a capture taken while it is undecided records the unsplit
 * schedule. */
MI_API int mi_sinkhorn_dots_schedule(mi_stream_t stream, int batch, int n, int m, int iterations);
/* Pin `schedule` (MI_SCHEDULE_CALLER_HELPER .. MI_SCHEDULE_UNSPLIT) for every shape on (current device, stream) --
 * nothing is tried or timed while pinned --, or MI_SCHEDULE_UNDECIDED to unpin and forget every decision (tuning starts
 * over).  Creates the stream's helper resources if they do not exist yet (MI_E_CAPACITY when the library already serves
 * 64 caller streams). */
MI_API int mi_sinkhorn_dots_set_schedule(mi_stream_t stream, int schedule);

/* ---- detector/shi_tomasi.py:66-112  ShiTomasiScore.forward ---------------------------------
 * score[n,1,h,w] = max(0, (a+c)/2 - sqrt(((a-c)/2)^2 + b^2 + 1e-10)) of the Sobel structure
 * tensor summed over block_size^2 (replicate padding of image and of the product maps).
 * block_size: positive odd.  Bit-exact vs the reference for uint8-valued input, block 3. */
MI_API int mi_corner_response(const float *image, int n, int h, int w, int block_size, float *score,
                       mi_stream_t stream);
/* ---- u8 ingest (sample/visual_odometry.py:65-92 load_image_from_array, sample/image_matching.py:42-46: a uint8 gray
 * frame is converted to float32 (1,1,H,W) on the host before the model sees it).  The _u8 entry points take the
 * uint8 frame itself: the same results as the float32 entry point on the converted frame, bit for bit, with 1 instead
 * of 4 bytes per pixel read (corner response: 5 instead of 8 B/px of HBM traffic; a pair costs 0.6 instead of
 * 2.5 MB of PCIe when frames are streamed from the host).  mi_convert_u8_f32 is that conversion on the device, for
 * the entry points that have no uint8 form. */
MI_API int mi_corner_response_u8(const uint8_t *image, int n, int h, int w, int block_size, float *score,
                          mi_stream_t stream);
MI_API int mi_convert_u8_f32(const uint8_t *src, long long count, float *dst, mi_stream_t stream);
/* mi_corner_response / _u8 (pixels_are_u8 = 0 / 1) with dynamic tile scheduling for large batches: tile_counter =
 * MI_TILE_COUNTER_BYTES of device memory, 4-byte aligned, of ANY content: the call clears the block on `stream`
 * (with a small kernel -- no entry point of this library issues hipMemsetAsync, whose captured form does not survive
 * hipGraph replays on ROCm 7.2) ahead of the kernel that draws tickets from it, so a block left dirty by a launch that died cannot
 * make a later call skip tiles.  A block must not be shared by calls that may run concurrently (different streams
 * need different blocks).  NULL = the static schedule of the two entry points above.  Same scores; equally sized
 * static shares do not finish together because the SIMDs issue oldest-first, tickets make them. */
#define MI_TILE_COUNTER_BYTES 16640
MI_API int mi_corner_response_balanced(const void *image, int pixels_are_u8, int n, int h, int w, int block_size, float *score,
                                uint32_t *tile_counter, mi_stream_t stream);
/* image1 / image2 of a matcher (two equally shaped batches of per_set images) behind ONE launch, like mi_match_pairs
 * does inside: score is (2 * per_set, h, w), batch a first; NMS and top-k then run on one batch of twice the size.
 * The same per-image results as two calls; half the launches and one tail instead of two (the _pair entries below:
 * mi_sparse_bad_pair, mi_angle_at_keypoints_pair, mi_sparse_bad_oriented_pair; AKAZE: mi_akaze_scale_sets). */
MI_API int mi_corner_response_pair(const void *image_a, const void *image_b, int pixels_are_u8, int per_set, int h, int w,
                            int block_size, float *score, uint32_t *tile_counter, mi_stream_t stream);

/* ---- utils/keypoint_utils.py:12-44  apply_nms_maxpool ---------------------------------------
 * mask = 1.0f where score >= max over the (2r+1)^2 window (outside image = -inf) - 1e-7. */
MI_API int mi_nms_mask(const float *score, int n, int h, int w, int radius, float *mask, mi_stream_t stream);

/* ---- utils/keypoint_utils.py:71-92 (candidate stage of select_topk_keypoints) ---------------
 * Emits one 64-bit key per surviving pixel:
 *     m = score * mask * border ; survive iff m > max(score_threshold, 0)
 *     key = (float_bits(m) << 32) | (0xFFFFFFFF - (y*w + x))
 * The candidate buffer is segmented: mi_candidate_layout(h, w) gives S segments (one per
 * 128x32 image tile) of C slots each; cand is uint64[n][S][C], count is uint32[n][S].  Every
 * count entry is written (no pre-zeroing), a segment holds at most its tile's pixels (cannot
 * overflow), and no global atomics are used; the order inside a segment is unspecified.
 * mi_nms_candidates fuses the NMS of mi_nms_mask (mask never materialised);
 * mi_select_candidates takes an explicit mask (the reference's two-call form). */
MI_API int mi_candidate_layout(int h, int w, int *segments, int *segment_capacity);
MI_API int mi_nms_candidates(const float *score, int n, int h, int w, int radius, float score_threshold,
                      int border_margin, uint64_t *cand, uint32_t *count, mi_stream_t stream);
MI_API int mi_select_candidates(const float *score, const float *mask, int n, int h, int w,
                         float score_threshold, int border_margin, uint64_t *cand, uint32_t *count,
                         mi_stream_t stream);

/* ---- utils/keypoint_utils.py:94-115 (top-k stage of select_topk_keypoints) ------------------
 * For each image: the k largest keys over all its segments, descending => (score desc, linear
 * index asc).  keypoints[n,k,2] = (y, x) as float, (-1,-1) beyond the candidate count;
 * kscores[n,k].  1 <= k <= 4096. */
MI_API int mi_topk_keypoints(const uint64_t *cand, const uint32_t *count, int segments, int segment_capacity,
                      int n, int w, int k, float *keypoints, float *kscores, mi_stream_t stream);

/* ---- descriptor/bad.py:436-576  SparseBAD.forward (non-oriented, sampling_mode="nearest") ---
 * pair_geom[p] = x1 | x2<<5 | y1<<10 | y2<<15 | r<<20 in the 32x32 patch frame (table rows of
 * descriptor/bad_params.py), pair_thr[p] the learned threshold.  num_pairs % 64 == 0, <= 1024.
 * desc (n,k,num_pairs) f32 and/or bits (n,k,num_pairs/32) u32 may be NULL (bits only for HARD).
 * Box sums are exact (fp64 summed-area table over a replicate-clamped 34x34 window).
 * plan (optional, may be NULL): device buffer of mi_bad_plan_bytes(num_pairs) bytes, 16-byte
 * aligned, filled once per pair table by mi_bad_plan_build.  With a plan, HARD-mode keypoints
 * with integer coordinates inside the image that sit on an integer-valued (uint8) patch take an
 * int32 fast path (precomputed table corners when >= 15 px from the border); results are identical.
 * status (optional, required for the fast path): n*k bytes of workspace; the fast kernel marks
 * the keypoints it handled and the general kernel visits the rest.
 * mi_bad_plan_build is the one set-up call of this ABI that synchronises: it reads the table back,
 * orders every pair's table-corner reads on the host so that the fast kernel's LDS gathers hit as few
 * banks twice as possible, and uploads the plan (two stream synchronisations; not hipGraph-capturable). */
MI_API size_t mi_bad_plan_bytes(int num_pairs);
MI_API int mi_bad_plan_build(const uint32_t *pair_geom, const float *pair_thr, int num_pairs, void *plan,
                      mi_stream_t stream);
MI_API int mi_sparse_bad(const float *image, int n, int h, int w, const float *keypoints, int k,
                  const uint32_t *pair_geom, const float *pair_thr, int num_pairs, int mode,
                  float temperature, int normalize, float *desc, uint32_t *bits, const void *plan,
                  uint8_t *status, mi_stream_t stream);

/* u8 ingest form of mi_sparse_bad (see mi_corner_response_u8): identical results from a uint8 image. */
MI_API int mi_sparse_bad_u8(const uint8_t *image, int n, int h, int w, const float *keypoints, int k,
                     const uint32_t *pair_geom, const float *pair_thr, int num_pairs, int mode,
                     float temperature, int normalize, float *desc, uint32_t *bits, const void *plan,
                     uint8_t *status, mi_stream_t stream);
/* mi_sparse_bad / _u8 for image1 / image2 behind one launch (see mi_corner_response_pair): keypoints (2 * per_set, k, 2),
 * status (2 * per_set * k) and the outputs hold batch a first. */
MI_API int mi_sparse_bad_pair(const void *image_a, const void *image_b, int pixels_are_u8, int per_set, int h, int w,
                       const float *keypoints, int k, const uint32_t *pair_geom, const float *pair_thr, int num_pairs,
                       int mode, float temperature, int normalize, float *desc, uint32_t *bits, const void *plan,
                       uint8_t *status, mi_stream_t stream);

/* ---- descriptor/bad.py:62-110,189-218  BADDescriptor.forward (dense, non-oriented) ------------
 * out (n, num_pairs, h, w): the BAD response at every pixel, raw / sigmoid(-c*T) / (c <= 0),
 * box centres clamped into the image, boxes over the replicate-padded image; exact fp64 box sums
 * (the reference's fp32 integral image is itself inexact above 2^24).
 * mi_gather_descriptors: descriptor/bad.py:221-333, (batch,d,h,w) map sampled at keypoints
 * (batch,nk,2) -> (batch,nk,d); bilinear = 0: integer truncation, 1: grid_sample bilinear/border. */
MI_API int mi_bad_dense(const float *image, int n, int h, int w, const uint32_t *pair_geom, const float *pair_thr,
                 int num_pairs, int mode, float temperature, float *out, mi_stream_t stream);
/* descriptor/bad.py:112-187 (_compute_diff_map_oriented): the same map with every pixel's pair offsets
 * rotated by orientation (n,1,h,w) there and the box means sampled bilinearly (exact box sums). */
MI_API int mi_bad_dense_oriented(const float *image, const float *orientation, int n, int h, int w,
                          const uint32_t *pair_geom, const float *pair_thr, int num_pairs, int mode,
                          float temperature, float *out, mi_stream_t stream);
MI_API int mi_gather_descriptors(const float *descriptor_map, int batch, int d, int h, int w, const float *keypoints,
                          int nk, int bilinear, float *out, mi_stream_t stream);

/* ---- orientation/angle_estimation.py:86-172  AngleEstimator.forward --------------------------
 * angle = atan2(m01, m10) of the Gaussian-weighted first moments, conv with ZERO padding.
 * moment_kernels: the module's (2,1,ps,ps) weight buffer (x*G then y*G), patch_size odd <= 31.
 * mi_angle_map writes the dense (n,1,h,w) map; mi_angle_at_keypoints writes theta (n,k) only at
 * the keypoints (what descriptor/bad.py:487-500 samples from the map, same nearest rounding). */
MI_API int mi_angle_map(const float *image, int n, int h, int w, int patch_size, const float *moment_kernels,
                 float *angle, mi_stream_t stream);
MI_API int mi_angle_at_keypoints(const float *image, int n, int h, int w, const float *keypoints, int k,
                          int patch_size, const float *moment_kernels, float *theta, mi_stream_t stream);
/* mi_angle_at_keypoints for image1 / image2 behind one launch (see mi_corner_response_pair): keypoints (2 * per_set, k, 2),
 * theta (2 * per_set, k), batch a first. */
MI_API int mi_angle_at_keypoints_pair(const float *image_a, const float *image_b, int per_set, int h, int w,
                               const float *keypoints, int k, int patch_size, const float *moment_kernels, float *theta,
                               mi_stream_t stream);

/* ---- descriptor/bad.py:487-517  SparseBAD.forward, oriented branch; and sampling_mode "bilinear" --
 * Pair offsets rotated by the keypoint's angle, which comes either from a dense orientation map
 * (n,1,h,w) sampled at the keypoint, or from a per-keypoint array (n,k): exactly one non-NULL.
 * bilinear = 0: box centre = nearest pixel (grid_sample "nearest"); 1: the box means of the four
 * neighbouring centres interpolated as grid_sample "bilinear" does (bad.py:535-549), response and
 * sign test in fp32.  The non-oriented bilinear case is angle 0 for every keypoint.
 * status (optional): n*k bytes of workspace; when given, keypoints on uint8-valued windows are done
 * with an int32 table (half the LDS, same results) and only the rest with the fp64 one.
 * max_reach: an upper bound, in pixels, of |pair offset from the patch centre| + box radius over the pair table, or 0 if
 * unknown.  0 < max_reach <= 22.5 (both reference tables: 22.22) lets the nearest mode use a 48 x 48 instead of a
 * 60 x 60 window per keypoint (same results, more keypoints in flight); a bound the table exceeds is a contract
 * violation (boxes clipped to the window). */
MI_API int mi_sparse_bad_oriented(const float *image, int n, int h, int w, const float *keypoints, int k,
                           const float *orientation_map, const float *keypoint_angles,
                           const uint32_t *pair_geom, const float *pair_thr, int num_pairs, int mode,
                           float temperature, int normalize, int bilinear, float max_reach, float *desc,
                           uint32_t *bits, uint8_t *status, mi_stream_t stream);
/* mi_sparse_bad_oriented with per-keypoint angles for image1 / image2 behind one launch (see mi_corner_response_pair):
 * keypoints (2 * per_set, k, 2), keypoint_angles (2 * per_set, k), status and the outputs hold batch a first. */
MI_API int mi_sparse_bad_oriented_pair(const float *image_a, const float *image_b, int per_set, int h, int w,
                                const float *keypoints, int k, const float *keypoint_angles, const uint32_t *pair_geom,
                                const float *pair_thr, int num_pairs, int mode, float temperature, int normalize,
                                int bilinear, float max_reach, float *desc, uint32_t *bits, uint8_t *status,
                                mi_stream_t stream);

/* ---- matching/sinkhorn.py:79-110,178  cost matrix -> core log-score matrix --------------------
 * z[b, i, j] = -cost(desc1[b,i], desc2[b,j]) / epsilon for i < n, j < m; row pitch `pitch` floats
 * (pitch % 4 == 0, pitch >= m, z 16-byte aligned).  The dustbin row/column of the reference's
 * augmented matrix (sinkhorn.py:187) is a constant and is not stored: mi_sinkhorn takes it as a
 * scalar.  epsilon is a double so that fp32(epsilon) and fp32(-unused/epsilon) round exactly as
 * the reference's Python-float arithmetic does.
 * _bits: descriptors are packed hard bits (num_bits % 32 == 0, <= 4096); `normalized` selects
 *     desc = bit/sqrt(popcount) (cost = 2 - 2 dot/sqrt(pa pb)) or desc = bit (cost = Hamming).
 *     Dot products are exact: popcount(a & b) on the matrix cores (256 / 512 bits: v_mfma_f32_32x32x64_f8f6f4 on
 *     FP4 operands, a bit = the nibble 1.0 / 0.0, fp32 sums <= 4096 exact; other lengths: v_mfma_i32_32x32x32_i8 on 0/1 bytes).
 * _f32: arbitrary float descriptors (n,d)/(m,d); L2 via v_mfma_f32_32x32x2_f32, L1 on the VALU. */
MI_API int mi_cost_logscores_bits(const uint32_t *bits1, const uint32_t *bits2, int batch, int n, int m,
                           int num_bits, int normalized, double epsilon, float *z, int pitch,
                           mi_stream_t stream);
MI_API int mi_cost_logscores_f32(const float *desc1, const float *desc2, int batch, int n, int m, int d,
                          int distance, double epsilon, float *z, int pitch, mi_stream_t stream);

/* ---- matching/sinkhorn.py:112-147,187-206  log-space Sinkhorn with dustbins ------------------
 * z: core log-scores as above.  dustbin_logscore = fp32(-unused_score/epsilon).  u (batch*(n+1))
 * and v (batch*(m+1)) are workspace and return the final duals.  p (batch, n+1, m+1) dense =
 * exp(Z + u + v) over the augmented matrix; may be NULL (duals only).  iterations >= 1.
 * workspace: mi_sinkhorn_workspace_bytes(batch, n, m) bytes, 16-byte aligned, enables the fused
 * iteration that reads Z once per iteration (per-band column partials); with workspace == NULL
 * (or m > 1024, for which the query returns 0) the two-pass form runs -- same results up to
 * fp32 summation order.
 * Streams: with a workspace and batch >= 64 the call runs as two half batches on the caller's stream and the library's
 * per-stream helper streams, like mi_sinkhorn_dots and scheduled by the same self-tuner (its shapes are kept apart
 * from that solver's: mi_sinkhorn_dots_schedule(stream, batch, n, m, iterations + (1 << 20)) reports this solver's
 * decision); same duals whatever the schedule.  mi_sinkhorn_dots_set_schedule(stream, MI_SCHEDULE_UNSPLIT) keeps
 * every launch of both solvers on the caller's stream; inside a stream capture nothing is tried (the decision in
 * force, or unsplit). */
MI_API size_t mi_sinkhorn_workspace_bytes(int batch, int n, int m);
MI_API int mi_sinkhorn(const float *z, int batch, int n, int m, int pitch, float dustbin_logscore,
                int iterations, float *u, float *v, float *p, void *workspace, size_t workspace_bytes,
                mi_stream_t stream);

/* ---- packed-descriptor form of the two calls above (hard-binarised descriptors, L2) -----------
 * mi_cost_dots_bits stores the exact integer dot products popcount(a_i & b_j) as uint16
 * (dots[b, i, j], row pitch `pitch` uint16 elements, pitch % 8 == 0, pitch >= m, 16-byte aligned)
 * and, per descriptor, the pair (scale, squared norm) = (1/sqrt(pop), pop * scale^2) if
 * `normalized` else (1, pop): row_info float[batch][n][2], col_info float[batch][m][2].
 * mi_sinkhorn_dots runs the same iterations as mi_sinkhorn but rebuilds
 *     z = -max(|a|^2 + |b|^2 - 2 * dot * s_a * s_b, 0) * (1/epsilon)
 * in registers on every pass: 2 bytes per matrix element per iteration instead of 4.
 * m <= 1024 (mi_sinkhorn_dots_workspace_bytes returns 0 otherwise: use the fp32 form).
 * epsilon >= MI_DOTS_MIN_EPSILON: the factored z drops the reference's clamp(cost, min=0) (sinkhorn.py:103),
 * which only acts on the rounding noise of identical normalised descriptors (cost = -O(3e-7)); that noise
 * enters z as +O(3e-7/epsilon), inside the 1e-4 parity bound down to epsilon = 0.005 and not below --
 * smaller epsilon is refused with MI_E_PARAM (mi_match_pairs_workspace_bytes returns 0): use the fp32-Z
 * form, which clamps.
 * sqnorm_bound: an upper bound of every squared norm in row_info / col_info (1 for `normalized`
 * descriptors, num_bits otherwise), or 0 if unknown.  With a bound small enough that
 * 2 * sqnorm_bound / epsilon < ~62 the row pass shifts every row of a pair by one analytic bound
 * instead of each row's own maximum (same result to fp32 rounding, fewer instructions); a bound the
 * data exceeds is a contract violation (exponent overflow).  0 always takes the per-row-maximum path.
 * flags: MI_SOLVER_DEFAULT, or MI_SOLVER_MULTI_LAUNCH to rule out the single-launch form (which otherwise runs for
 * batch <= 8, n, m <= 512 when its grid fits the device; same duals bit for bit either way).
 * mi_sinkhorn_dots_status_word: the device address, inside `workspace`, of the call's 32-bit status word -- written by
 * every mi_sinkhorn_dots call with these extents: 0 = solved, non-zero = a hand-off of the single-launch form timed out
 * and u, v (and p) of the affected pairs are NaN.  Read it after synchronising, or hand it to mi_mnn_from_duals_dots. */
MI_API int mi_cost_dots_bits(const uint32_t *bits1, const uint32_t *bits2, int batch, int n, int m, int num_bits,
                      int normalized, uint16_t *dots, int pitch, float *row_info, float *col_info,
                      mi_stream_t stream);
MI_API size_t mi_sinkhorn_dots_workspace_bytes(int batch, int n, int m);
MI_API int mi_sinkhorn_dots(const uint16_t *dots, const float *row_info, const float *col_info, int batch, int n,
                     int m, int pitch, double epsilon, double unused_score, double sqnorm_bound, int iterations,
                     float *u, float *v, float *p, void *workspace, size_t workspace_bytes, int flags,
                     mi_stream_t stream);
MI_API const uint32_t *mi_sinkhorn_dots_status_word(const void *workspace, int batch, int n, int m);

/* ---- matching/sinkhorn.py:317-465  SinkhornMatcherWithFilters (filter stage) -----------------
 * In place on p (batch, n+1, m+1): per row i < n, best/second-best core probability and the
 * dustbin entry decide valid[b,i] (ratio_threshold <= 0 / dustbin_margin < 0 disable a filter);
 * failing rows get core * 0 and dustbin entry 1, as the reference writes them. */
MI_API int mi_match_filters(float *p, int batch, int n, int m, float ratio_threshold, float dustbin_margin,
                     uint8_t *valid, mi_stream_t stream);
/* ---- matching/outlier_filters.py:11-116  probability_ratio_filter / dustbin_margin_filter ------
 * The same two tests as masks only (p is not modified): valid[b,i] for i < n.
 * has_dustbin = 1: p is (batch, n+1, m+1) with the dustbin column (dustbin_margin_filter's argument; both tests
 * available); 0: p is the (batch, n, m) core (probability_ratio_filter's argument; dustbin_margin must be < 0).
 * ratio_threshold <= 0 / dustbin_margin < 0 disable a test. */
MI_API int mi_match_filter_masks(const float *p, int batch, int n, int m, int has_dustbin, float ratio_threshold,
                          float dustbin_margin, uint8_t *valid, mi_stream_t stream);

/* ---- matching/match_extraction.py:72-181  MutualNearestNeighborMatcher.forward --------------
 * p (batch, n+1, m+1); kpts1 (batch,n,2); kpts2 (batch,m,2).  Workspace: row_best (batch*n) u64,
 * col_best (batch*m) u64.  Outputs mk1/mk2 (batch,max_matches,2), scores (batch,max_matches),
 * valid (batch,max_matches) u8, match_ij (batch,max_matches,2) i32 (may be NULL).
 * n <= 4096.  Ties: first index (argmax), then (score desc, row asc) for the top-max_matches.
 * m <= 1024: one pass over p (row and column winners together; col_best, 8-byte aligned, is cleared by the call and
 * filled by 64-bit atomic maxima -- exact, so the order of arrival cannot be seen); larger m: a row and a column pass. */
MI_API int mi_mnn_extract(const float *p, int batch, int n, int m, const float *kpts1, const float *kpts2,
                   int max_matches, float threshold, uint64_t *row_best, uint64_t *col_best,
                   float *mk1, float *mk2, float *scores, uint8_t *valid, int32_t *match_ij,
                   mi_stream_t stream);

/* ---- MutualNearestNeighborMatcher.forward straight from the Sinkhorn duals ---------------------
 * What feature_detection/match_extraction_wrapper.py:82-113 computes (matcher -> P -> mutual NN)
 * without materialising P: P_ij = expf((z_ij + u_i) + v_j) is evaluated in registers exactly as
 * mi_sinkhorn's final pass does, so the outputs are bit-identical to mi_sinkhorn(p != NULL) followed
 * by mi_mnn_extract.  z/pitch (or dots/row_info/col_info/pitch/epsilon) and u, v are what was
 * passed to / returned by mi_sinkhorn (mi_sinkhorn_dots) with p == NULL.  m <= 1024, n <= 4096.
 * workspace: mi_mnn_duals_workspace_bytes(batch, n, m) bytes (0 = unsupported size), 8-byte aligned.
 * solver_status (mi_mnn_from_duals_dots; may be NULL): the status word of the mi_sinkhorn_dots call that produced
 * u, v.  When it is non-zero on the device every match of this call comes back with score -1 / valid 0 / match_ij -1. */
MI_API size_t mi_mnn_duals_workspace_bytes(int batch, int n, int m);
MI_API int mi_mnn_from_duals(const float *z, int batch, int n, int m, int pitch, const float *u, const float *v,
                      const float *kpts1, const float *kpts2, int max_matches, float threshold, void *workspace,
                      size_t workspace_bytes, float *mk1, float *mk2, float *scores, uint8_t *valid,
                      int32_t *match_ij, mi_stream_t stream);
MI_API int mi_mnn_from_duals_dots(const uint16_t *dots, const float *row_info, const float *col_info, int batch, int n,
                           int m, int pitch, double epsilon, const float *u, const float *v, const float *kpts1,
                           const float *kpts2, int max_matches, float threshold, void *workspace,
                           size_t workspace_bytes, const uint32_t *solver_status, float *mk1, float *mk2,
                           float *scores, uint8_t *valid, int32_t *match_ij, mi_stream_t stream);

/* ---- detector/akaze.py  AKAZE (BASELINE config 4), all maps fp32 (n,1,h,w) ---------------------
 * mi_akaze_diffuse: one explicit step of NonLinearDiffusion.forward (akaze.py:98-131):
 *   g = sobel/8 gradients (zero pad), c = 1/(1+(|g|/kappa)^2) with |g| = sqrt(gx^2+gy^2+1e-8),
 *   l_out = l_in + dt * div(c*g) (sobel/8 on the zero-padded flux).  l_out must not alias l_in.
 * mi_akaze_hessian_scores: HessianDetector.forward (akaze.py:227-254): det of the 3x3-kernel
 *   Hessian, kept where it equals the nms_size^2 window maximum (-inf outside the image) and
 *   exceeds threshold, clamped >= 0.  nms_size odd <= 15.
 * mi_akaze_combine: AKAZE.forward's scale selection (akaze.py:442-451) on stacked per-scale maps
 *   (num_scales,n,h,w): scores = max over scales, orientations = mean of the orientations of the
 *   scales attaining the max (orientations/scale_orientations may both be NULL: scores only).
 * mi_akaze_orientation_at_keypoints: the same selection evaluated only at keypoints (n,k,2):
 *   scale_theta (num_scales,n,k) from mi_angle_at_keypoints per scale -> theta (n,k); equal to
 *   sampling the combined map the way descriptor/bad.py:487-500 does. */
MI_API int mi_akaze_diffuse(const float *l_in, int n, int h, int w, float kappa, float dt, float *l_out,
                     mi_stream_t stream);
/* One scale of AKAZE.forward (akaze.py:430-440) in one launch: l_out = `iterations` diffusion steps of l_in,
 * scores = mi_akaze_hessian_scores(l_out); identical maps, 12 instead of 8 * iterations + 8 bytes per pixel of HBM
 * traffic.  Fused for iterations 1..3 and nms_size 3 / 5 / 7 (mi_akaze_scale_fused returns 1) -- as a rolling window
 * that streams down the image (even w, 8-byte aligned maps, nms_size 3 / 5) or on an LDS-resident tile --; other
 * values run the per-step kernels and then need `tmp` (n*h*w floats) when iterations > 1.  l_out must not alias l_in.
 * kappa must lie in [MI_AKAZE_KAPPA_MIN, MI_AKAZE_KAPPA_MAX] (the range the fused kernels' exactly rounded division
 * helpers are verified for; MI_E_PARAM otherwise -- mi_akaze_diffuse + mi_akaze_hessian_scores take any kappa > 0). */
#define MI_AKAZE_KAPPA_MIN 1e-3f
#define MI_AKAZE_KAPPA_MAX 1e6f
MI_API int mi_akaze_scale_fused(int iterations, int nms_size);
MI_API int mi_akaze_scale(const float *l_in, int n, int h, int w, int iterations, float kappa, float dt, float threshold,
                   int nms_size, float *l_out, float *scores, float *tmp, mi_stream_t stream);
/* mi_akaze_scale for the FIRST scale of two equally shaped batches (image1 / image2 of a matcher) behind one launch:
 * l_out and scores are (2 * per_set, h, w), batch a first; every later scale then runs on one batch of twice the size.
 * tmp: per_set * h * w floats, as for mi_akaze_scale. */
MI_API int mi_akaze_scale_sets(const float *l_in_a, const float *l_in_b, int per_set, int h, int w, int iterations,
                        float kappa, float dt, float threshold, int nms_size, float *l_out, float *scores, float *tmp,
                        mi_stream_t stream);
/* The LAST scale with AKAZE.forward's selection across scales (akaze.py:436-451) folded in: l_out as mi_akaze_scale;
 * instead of this scale's score map, best (n,h,w) = max over the num_prev (<= 7) earlier scales' maps prev_scores
 * (num_prev,n,h,w) and this scale's, and attain (n,h,w) uint8: bit s set when earlier scale s reaches that maximum,
 * bit num_prev when this scale does -- what mi_akaze_combine computes from stacked maps, without the stack's last map
 * and without a pass of its own (the streaming form reads the earlier maps where it writes its output row).
 * mi_akaze_orientation_from_attain: mi_akaze_orientation_at_keypoints from `attain` instead of the stacked maps. */
MI_API int mi_akaze_scale_select(const float *l_in, int n, int h, int w, int iterations, float kappa, float dt,
                          float threshold, int nms_size, float *l_out, const float *prev_scores, int num_prev,
                          float *best, uint8_t *attain, float *tmp, mi_stream_t stream);
MI_API int mi_akaze_orientation_from_attain(const uint8_t *attain, const float *scale_theta, int num_scales, int n, int h,
                                     int w, const float *keypoints, int k, float *theta, mi_stream_t stream);
/* The same orientation in ONE launch from the diffused images themselves: scale_images = num_scales maps (n,h,w),
 * scale_stride floats apart (a stacked (S,n,h,w) tensor: n*h*w); per keypoint the patch_size^2 Gaussian moments
 * (mi_angle_at_keypoints) are evaluated only for the scales `attain` names -- typically one of three. */
MI_API int mi_akaze_orientation_select(const float *scale_images, size_t scale_stride, int num_scales,
                                const uint8_t *attain, int n, int h, int w, const float *keypoints, int k,
                                int patch_size, const float *moment_kernels, float *theta, mi_stream_t stream);
MI_API int mi_akaze_hessian_scores(const float *l, int n, int h, int w, float threshold, int nms_size, float *scores,
                            mi_stream_t stream);
MI_API int mi_akaze_combine(const float *scale_scores, const float *scale_orientations, int num_scales, int n, int h,
                     int w, float *scores, float *orientations, mi_stream_t stream);
MI_API int mi_akaze_orientation_at_keypoints(const float *scale_scores, const float *scale_theta, int num_scales,
                                      int n, int h, int w, const float *keypoints, int k, float *theta,
                                      mi_stream_t stream);

/* ---- matching/sinkhorn.py:228-259  SinkhornMatcherWithScores: maxima of P[:n,:m] per row / column */
MI_API int mi_core_maxima(const float *p, int batch, int n, int m, float *row_max, float *col_max, mi_stream_t stream);

/* ---- feature_detection/..._essential_matrix.py:334-360: `count` keypoints (y, x) in pixels -> normalised
 * image coordinates (x, y), the first two rows of k_inv (3x3 row-major, device memory) times [x, y, 1]. */
MI_API int mi_normalise_keypoints(const float *keypoints, long long count, const float *k_inv, float *points,
                           mi_stream_t stream);

/* ---- geometry/essential_matrix_estimator.py:302-431  EssentialMatrixEstimator.forward and the
 * composites' _estimate_essential_matrix (feature_detection/..._essential_matrix.py:184-271) -------
 * Weighted 8-point algorithm on the assignment matrix p (batch, n+1, m+1): bidirectional top_k mask
 * (k-th largest with multiplicity) AND p > 0.01 on the core (times valid1 x valid2 when given, both
 * (batch, n) / (batch, m) bytes or both NULL), Hartley normalisation with the weights' row / column
 * sums, Kronecker-factored normal equations, n_iter steps of shifted power iteration for the minimum
 * eigenvector, denormalisation, projection onto singular values (s, s, 0) with n_iter_manifold steps.
 * pts1 (batch, n, 2), pts2 (batch, m, 2): NORMALISED image coordinates (x, y) = K^-1 [px, py, 1].
 * e (batch, 3, 3).  n, m <= 1024, 1 <= top_k <= min(8, n, m).  Deterministic.
 * workspace (optional): mi_essential_matrix_workspace_bytes(batch, n, m, top_k) bytes, 16-byte aligned (the query returns
 * 0 for top_k > 4: no banded form).  With it the head runs in two launches -- one pass over the matrix spread over the
 * chip (bands of 32 rows: row thresholds, the rows' candidate entries, per-band column lists), then one workgroup per
 * pair on the sparse weights -- instead of one workgroup per pair streaming the matrix four times (~0.6 ms per launch
 * whatever the batch).  Same definition of every quantity; results agree with the workspace-less form to fp32 rounding
 * (a row's / column's few weights are added in a different order).  NULL: the single-launch dense form. */
MI_API size_t mi_essential_matrix_workspace_bytes(int batch, int n, int m, int top_k);
MI_API int mi_essential_matrix(const float *p, int batch, int n, int m, const float *pts1, const float *pts2,
                        const uint8_t *valid1, const uint8_t *valid2, int top_k, int n_iter,
                        int n_iter_manifold, float *e, void *workspace, size_t workspace_bytes, mi_stream_t stream);
/* The same head WITHOUT a materialised P, from the packed-descriptor Sinkhorn solution: dots / row_info / col_info /
 * pitch as written by mi_cost_dots_bits, u (batch, n+1) / v (batch, m+1) the duals of mi_sinkhorn_dots (p = NULL there).
 * Every entry P_ij = exp(z_ij + u_i + v_j) is rebuilt in registers with the solver's own final-pass expression, so E
 * equals mi_essential_matrix on the P that mi_sinkhorn_dots would have written, bit for bit; 2 instead of 4 bytes per
 * entry are read and the (n+1) x (m+1) matrix is neither written nor read back.  epsilon >= MI_DOTS_MIN_EPSILON. */
MI_API int mi_essential_matrix_dots(const uint16_t *dots, const float *row_info, const float *col_info, int pitch,
                             double epsilon, const float *u, const float *v, int batch, int n, int m,
                             const float *pts1, const float *pts2, const uint8_t *valid1, const uint8_t *valid2,
                             int top_k, int n_iter, int n_iter_manifold, float *e, void *workspace,
                             size_t workspace_bytes, mi_stream_t stream);

/* ---- detector/fast.py:198-239  FASTScore.forward (use_nms = False) ----------------------------------
 * score (n,1,h,w) = 1.0 where 9 contiguous pixels of the radius-3 circle (replicate padding) are all
 * >= centre + threshold or all <= centre - threshold, else 0.0.  Bit-identical to the reference.
 * ---- detector/dog.py:100-142  DoGDetector.forward ---------------------------------------------------
 * out (n, num_scales-1, h, w) = differences of consecutive Gaussian blurs of the replicate-padded
 * image.  weights_1d (num_scales, kernel_size): the row sums of the module's normalised 2-D kernels
 * (their exact 1-D factors).  2 <= num_scales <= 8, kernel_size odd <= 49.
 * score (n,1,h,w), optional: max over scales of |DoG| (DoGDetectorWithScore.forward, dog.py:182-204);
 * out or score may be NULL, not both. */
MI_API int mi_fast_score(const float *image, int n, int h, int w, float threshold, float *score, mi_stream_t stream);
MI_API int mi_dog_responses(const float *image, int n, int h, int w, const float *weights_1d, int num_scales,
                     int kernel_size, float *out, float *score, mi_stream_t stream);

/* ---- feature_detection/match_extraction_wrapper.py:82-113 over shi_tomasi_sparse_bad_sinkhorn.py:79-182
 * The whole path for `batch` image pairs in one call: image1[b] vs image2[b], (batch,1,h,w) f32 each ->
 * keypoints1/2 (batch,K,2) as (y,x), matched1/2 (batch,max_matches,2), match_scores (batch,max_matches),
 * match_valid (batch,max_matches) bytes, match_ij (batch,max_matches,2) int32 or NULL.  Hard-binarised
 * descriptors, L2 cost, matches straight from the Sinkhorn duals (P is never written); K <= 1024.
 * It sequences mi_corner_response, mi_nms_candidates, mi_topk_keypoints, mi_sparse_bad (twice each),
 * mi_cost_dots_bits, mi_sinkhorn_dots and mi_mnn_from_duals_dots on `stream`, with every intermediate in
 * `workspace` (mi_match_pairs_workspace_bytes, 16-byte aligned): nothing is allocated, nothing synchronises.
 * Results are bit-identical to calling those entry points one by one (what the Python modules do).
 * pair_geom / pair_thr / bad_plan: device pointers as for mi_sparse_bad (bad_plan may be NULL). */
typedef struct mi_match_params {
  int block_size;            /* ShiTomasiScore(block_size), 3 in the export CLI */
  int nms_radius;            /* apply_nms_maxpool radius */
  int max_keypoints;         /* K */
  float score_threshold;     /* select_topk_keypoints */
  int border_margin;         /* select_topk_keypoints; the matcher's default is the descriptor's max radius (7) */
  int num_pairs;             /* BAD pairs P: 256 or 512 with the reference tables */
  const uint32_t *pair_geom; /* device, P words */
  const float *pair_thr;     /* device, P floats */
  const void *bad_plan;      /* device, mi_bad_plan_build output, or NULL */
  int normalize_descriptors; /* SparseBAD(normalize_descriptors=...) */
  double epsilon;            /* SinkhornMatcher */
  double unused_score;
  int sinkhorn_iterations;
  int max_matches;           /* MutualNearestNeighborMatcher */
  float match_threshold;
  int flags;                 /* MI_SOLVER_DEFAULT or an OR of MI_SOLVER_MULTI_LAUNCH ("co-residency" in the conventions)
                                and MI_SOLVER_NO_FORK */
} mi_match_params;
MI_API size_t mi_match_pairs_workspace_bytes(int batch, int h, int w, const mi_match_params *params);
MI_API int mi_match_pairs(const float *image1, const float *image2, int batch, int h, int w,
                   const mi_match_params *params, float *keypoints1, float *keypoints2, float *matched1,
                   float *matched2, float *match_scores, uint8_t *match_valid, int32_t *match_ij,
                   void *workspace, size_t workspace_bytes, mi_stream_t stream);

/* u8 ingest form of mi_match_pairs: uint8 frames (batch,1,h,w), same workspace, identical results. */
MI_API int mi_match_pairs_u8(const uint8_t *image1, const uint8_t *image2, int batch, int h, int w,
                      const mi_match_params *params, float *keypoints1, float *keypoints2, float *matched1,
                      float *matched2, float *match_scores, uint8_t *match_valid, int32_t *match_ij,
                      void *workspace, size_t workspace_bytes, mi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_MATCH_H */
