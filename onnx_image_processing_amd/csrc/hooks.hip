// include/mi355x_match_debug.h: the development / test hooks.  This translation unit exists only in
// libmi355x_match_debug.so (built with -DMI_DEBUG_HOOKS); the product library has neither these entry points nor the
// atomics behind them (csrc/hooks.h).
#ifndef MI_DEBUG_HOOKS
#error "hooks.hip belongs to the debug library only (-DMI_DEBUG_HOOKS)"
#endif
#include "common.h"
#include "hooks.h"

#include "../../include/mi355x_match_debug.h"

MiHooks mi_hooks;

int mi_bad_plan_passes_host(const uint32_t *pair_geom_host, int num_pairs, int *canonical, int *scheduled);   // bad.hip
int mi_sinkhorn_dots_form_host(int batch, int n, int m, int flags, int blocks_per_cu, int cus);                // sinkhorn_dots.hip

MI_API int mi_debug_clock_probe(void *buffer) {
  mi_hooks.corner_clk = reinterpret_cast<unsigned long long *>(buffer);
  return MI_OK;
}

MI_API int mi_debug_topk_stamps(void *buffer) {
  mi_hooks.topk_prof = reinterpret_cast<unsigned long long *>(buffer);
  return MI_OK;
}

MI_API int mi_debug_set(int key, int value) {
  if (key == 0) {                                  // every selector back to the product's value, probes off
    mi_hooks.corner_impl = 0; mi_hooks.corner_rows = 4; mi_hooks.corner_rows_u8_default = 1;
    mi_hooks.sinkhorn_log_partials = 0; mi_hooks.sinkhorn_split = 2; mi_hooks.sinkhorn_persist = 1;
    mi_hooks.sinkhorn_stamps = 0; mi_hooks.topk_select = 1; mi_hooks.topk_split = -1;
    mi_hooks.corner_clk = nullptr; mi_hooks.topk_prof = nullptr;
    return MI_OK;
  }
  if (key == 1) { mi_hooks.corner_impl = value; return MI_OK; }
  if (key == 2 && (value == 4 || value == 5 || value == 8)) { mi_hooks.corner_rows = value; mi_hooks.corner_rows_u8_default = 0; return MI_OK; }
  if (key == 4) { mi_hooks.sinkhorn_log_partials = value; return MI_OK; }
  if (key == 6) { mi_hooks.sinkhorn_split = value; return MI_OK; }
  if (key == 7) { mi_hooks.sinkhorn_persist = value; return MI_OK; }
  if (key == 8) { mi_hooks.sinkhorn_stamps = value; return MI_OK; }
  if (key == 9) { mi_hooks.topk_select = value; return MI_OK; }
  if (key == 10) { mi_hooks.topk_split = value; return MI_OK; }
  return MI_E_PARAM;
}

MI_API int mi_debug_bad_plan_passes(const uint32_t *pair_geom_host, int num_pairs, int *canonical, int *scheduled) {
  return mi_bad_plan_passes_host(pair_geom_host, num_pairs, canonical, scheduled);
}

MI_API int mi_debug_sinkhorn_dots_form(int batch, int n, int m, int flags, int blocks_per_cu, int cus) {
  return mi_sinkhorn_dots_form_host(batch, n, m, flags, blocks_per_cu, cus);
}
