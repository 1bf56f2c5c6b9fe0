/*
 * mi355x_match_debug.h -- development / test hooks.  NOT part of the product ABI (include/mi355x_match.h) and NOT in
 * the product library: these entry points exist only in libmi355x_match_debug.so, a second build of the same sources
 * with -DMI_DEBUG_HOOKS (csrc/hooks.h) that also exports everything of include/mi355x_match.h.  Nothing here changes
 * results, only which of two equivalent kernel implementations runs; the setting is process-wide (atomics).
 * tests/ load the debug library to compare the alternatives bit for bit; tools/ to time them.
 */
#ifndef MI355X_MATCH_DEBUG_H
#define MI355X_MATCH_DEBUG_H

#include <stdint.h>

#ifndef MI_API
#define MI_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Choose between equivalent kernel implementations (results are
 * identical).  key 0: reset every selector to the product library's value and switch the probes off.  key 1: corner response, 0 = streaming LDS-DMA kernel (default), 1 = register-
 * staged tile kernel.  key 2: rows per thread of the streaming corner kernel (4, 5 or 8).
 * key 6: number of batch parts mi_sinkhorn_dots runs on separate streams (1..4, default 2).
 * key 7: mi_sinkhorn_dots for <= 8 pairs (n, m <= 512), 1 = single-launch form (default), 0 = multi-launch form.
 * key 8: 1 = the single-launch Sinkhorn kernel records phase time stamps behind its workspace's fail word (tools/).
 * key 9: top-k, 1 = radix-select the k-th key and sort only the k winners when k << candidates
 * (default), 0 = always sort every candidate.
 * key 10: workgroups per image of the top-k histogram passes when the candidates are selected from global memory
 * (-1 = automatic, the default; 1 = the single-workgroup form).
 * key 4: Sinkhorn band kernel, 0 = probability form, lean instruction stream (default), 2 = first
 * probability-form kernel, 1 = log-domain (max,sum) partials (results agree to fp32 rounding).
 * key 11: stream schedule of mi_sinkhorn_dots for >= 64 pairs: -1 = self-tuned per caller stream (default), 0 = halves
 * on {caller's stream, helper 0}, 1 = halves on {helper 0, helper 1}, 2 = unsplit (same duals bit for bit).
 * key 12: fused AKAZE scale (mi_akaze_scale / mi_akaze_scale_select), 0 = streaming rolling-window kernel where it
 * applies (default), 1 = LDS-tile kernel / per-step kernels (same maps bit for bit).
 * key 13: mi_sparse_bad_oriented asked for packed bits only (nearest sampling, 256 / 512 pairs), 0 = the unrolled bits
 * kernel (default), 1 = the generic kernel (same bits).
 * key 14: mi_cost_dots_bits / mi_cost_logscores_bits at 256 / 512 bits: 0 = dot products on the FP4 MFMA (default), 1 = on
 * the int8 MFMA (the same integers).
 * key 15: mi_sinkhorn_dots under MI_SOLVER_DOTS_BELOW_1024: 1 = the row kernel reads the uint16 dots as fp16 denormals
 * (v_fma_mix_f32; default), 0 = converts them first (the same duals bit for bit).
 * key 16: mi_sinkhorn_dots with 512 < m <= 1024 (bounded-shift row kernel): 1 = two waves per row group, one 512-column
 * chunk each, 32-row bands (default), 0 = two chunks per wave, 16-row bands (the sums associate differently: duals equal to rounding).
 * key 17: mi_mnn_from_duals / mi_mnn_from_duals_dots with 512 < m <= 1024: 1 = two waves per row group, one 512-column
 * chunk each (default), 0 = two chunks per wave (the same matches: winners are exact maxima).
 * key 18: the P output of mi_sinkhorn_dots / mi_sinkhorn: 1 = four rows per wave, every load issued up front (default), 0 = one row per
 * wave in a loop of dependent round trips (the same P bit for bit).
 * key 19: mi_mnn_extract with m <= 1024: 1 = row and column winners in one pass over P (default), 0 = a row kernel and a
 * column kernel (the same winners: exact maxima). */
MI_API int mi_debug_set(int key, int value);
/* top-k kernel phase time stamps (100 MHz clock) of workgroup 0 into `buffer` (8 x uint64, device memory); NULL = off */
MI_API int mi_debug_topk_stamps(void *buffer);
/* streaming corner kernel: every workgroup of the persistent grid writes {shader clock, 100 MHz clock} at entry and
 * exit into `buffer` (4 x uint64 per workgroup, <= 2048 workgroups; device memory); NULL = off */
MI_API int mi_debug_clock_probe(void *buffer);
/* fast BAD kernel: LDS passes per keypoint of its gather schedule (csrc/bad_plan_opt.h) for a HOST copy of a pair table,
 * as the table stands and as mi_bad_plan_build schedules it (num_pairs / 4 = conflict-free).  Host only, no GPU. */
MI_API int mi_debug_bad_plan_passes(const uint32_t *pair_geom_host, int num_pairs, int *canonical, int *scheduled);
/* which Sinkhorn form mi_sinkhorn_dots chooses (1 = single launch, 0 = multi launch) for these extents and flags on a
 * device that can hold `blocks_per_cu` workgroups of the single-launch kernel on each of `cus` compute units: the pure
 * host decision function the library applies to the occupancy query's answer.  Host only, no GPU. */
MI_API int mi_debug_sinkhorn_dots_form(int batch, int n, int m, int flags, int blocks_per_cu, int cus);

/* The decision logic of mi_sinkhorn_dots' stream-schedule tuner (csrc/sk_tuner.h) on a fresh state, driven by a
 * script of n_ops operations with injected timings.  Host only, no GPU.  op 0: an eager call of the shape with batch =
 * arg -> out = schedule | (slot & 0xff) << 8 | (entry & 0xff) << 16 (slot = entry = 0xff: no trial handed out);
 * op 1: trial (entry = arg >> 8, slot = arg & 0xff) finished in val milliseconds; op 2: that trial abandoned;
 * op 3: out = the schedule in force for batch = arg (-1: undecided); op 4: out = the schedule a capture would get;
 * op 5: pin schedule arg (-1: unpin and forget), out = 0 / -1. */
MI_API int mi_debug_tuner_script(int n_ops, const int *op, const int *arg, const double *val, int *out);

/* Exhaustive check of the fused AKAZE kernel's exact-rounding helpers (csrc/akaze_math.h) against the IEEE operators:
 * every float whose bit pattern lies in [lo_bits, hi_bits) is tried; which = 0: sqrt, 1: x / kappa through the
 * precomputed correctly rounded reciprocal, 2: 1 / x, 3: the general x / kappa sequence.  *mismatches_u64 (device,
 * zeroed by the caller) receives the number of differing results, *first_bad_u32 (device, preset to 0xFFFFFFFF) the
 * smallest offending bit pattern. */
MI_API int mi_debug_akaze_math_check(int which, float kappa, uint32_t lo_bits, uint32_t hi_bits, void *mismatches_u64,
                                     void *first_bad_u32, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_MATCH_DEBUG_H */
