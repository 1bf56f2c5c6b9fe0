// Kernel-variant selectors and development probes.
//
// Product build (libmi355x_match.so): every selector is a compile-time constant -- the library has no process-wide
// switch that could change which kernel runs -- and the probes are off.
// -DMI_DEBUG_HOOKS (libmi355x_match_debug.so, loaded by tests/ and tools/ only): the same selectors are process-wide
// atomics set through include/mi355x_match_debug.h (mi_debug_set and friends, defined in hooks.hip).
#pragma once

#ifdef MI_DEBUG_HOOKS
#include <atomic>
struct MiHooks {
  std::atomic<int> corner_impl{0};             // key 1: 0 = streaming (LDS-DMA) corner kernel, 1 = register-staged tile kernel
  std::atomic<int> corner_rows{4};             // key 2: rows per thread of the streaming kernel (4, 5, 8)
  std::atomic<int> corner_rows_u8_default{1};  // 1 until key 2 is set: the uint8 kernel then uses its own best (5 rows)
  std::atomic<int> sinkhorn_log_partials{0};   // key 4: band kernel form of mi_sinkhorn
  std::atomic<int> sinkhorn_split{2};          // key 6: batch parts of mi_sinkhorn_dots on separate streams
  std::atomic<int> sinkhorn_persist{1};        // key 7: single-launch form for <= 8 pairs
  std::atomic<int> sinkhorn_stamps{0};         // key 8: phase time stamps of the single-launch kernel
  std::atomic<int> topk_select{1};             // key 9: radix select + sort k instead of sorting every candidate
  std::atomic<int> sinkhorn_schedule{-1};      // key 11: -1 = self-tuned (default); 0 / 1 / 2 = fixed schedule of mi_sinkhorn_dots
  std::atomic<int> akaze_impl{0};              // key 12: 0 = streaming rolling-window AKAZE scale kernel, 1 = LDS-tile kernel
  std::atomic<int> bad_oriented_impl{0};       // key 13: 0 = the matchers' unrolled bits kernel of mi_sparse_bad_oriented, 1 = generic kernel
  std::atomic<int> cost_impl{0};               // key 14: 0 = packed-descriptor dot products on the FP4 MFMA (256 / 512 bits), 1 = int8 MFMA
  std::atomic<int> sinkhorn_mix{1};         // key 15: 1 = under MI_SOLVER_DOTS_BELOW_1024 the row kernel reads the dots as fp16 denormals (v_fma_mix_f32), 0 = converts them
  std::atomic<int> sinkhorn_pair_waves{1};     // key 16: 512 < m <= 1024, bounded-shift row kernel: 1 = two waves per row group with one chunk each, 0 = two chunks per wave
  std::atomic<int> mnn_pair_waves{1};          // key 17: matches from the duals, 512 < m <= 1024: 1 = two waves per row group with one chunk each, 0 = two chunks per wave
  std::atomic<int> sinkhorn_exp_rows{1};       // key 18: P of mi_sinkhorn_dots / mi_sinkhorn: 1 = four rows per wave with every load up front, 0 = one row per wave in a loop
  std::atomic<int> mnn_one_pass{1};            // key 19: mi_mnn_extract: 1 = rows and columns in one pass over P (m <= 1024), 0 = a row kernel and a column kernel
  std::atomic<int> topk_split{-1};             // key 10: workgroups per image of the top-k histogram pass (-1 = automatic)
  std::atomic<unsigned long long *> corner_clk{nullptr};   // mi_debug_clock_probe
  std::atomic<unsigned long long *> topk_prof{nullptr};    // mi_debug_topk_stamps
};
extern MiHooks mi_hooks;   // hooks.hip
#define MI_HOOK(field, product_value) (mi_hooks.field.load(std::memory_order_relaxed))
#else
#define MI_HOOK(field, product_value) (product_value)
#endif
