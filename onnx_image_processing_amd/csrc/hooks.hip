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
    mi_hooks.sinkhorn_stamps = 0; mi_hooks.topk_select = 1; mi_hooks.topk_split = -1; mi_hooks.sinkhorn_schedule = -1;
    mi_hooks.akaze_impl = 0;
    mi_hooks.bad_oriented_impl = 0;
    mi_hooks.cost_impl = 0;
    mi_hooks.sinkhorn_mix = 1;
    mi_hooks.sinkhorn_pair_waves = 1;
    mi_hooks.mnn_pair_waves = 1;
    mi_hooks.mnn_one_pass = 1;
    mi_hooks.sinkhorn_exp_rows = 1;
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
  if (key == 12 && (value == 0 || value == 1)) { mi_hooks.akaze_impl = value; return MI_OK; }
  if (key == 13 && (value == 0 || value == 1)) { mi_hooks.bad_oriented_impl = value; return MI_OK; }
  if (key == 14 && (value == 0 || value == 1)) { mi_hooks.cost_impl = value; return MI_OK; }
  if (key == 15 && (value == 0 || value == 1)) { mi_hooks.sinkhorn_mix = value; return MI_OK; }
  if (key == 16 && (value == 0 || value == 1)) { mi_hooks.sinkhorn_pair_waves = value; return MI_OK; }
  if (key == 17 && (value == 0 || value == 1)) { mi_hooks.mnn_pair_waves = value; return MI_OK; }
  if (key == 19 && (value == 0 || value == 1)) { mi_hooks.mnn_one_pass = value; return MI_OK; }
  if (key == 18 && (value == 0 || value == 1)) { mi_hooks.sinkhorn_exp_rows = value; return MI_OK; }
  if (key == 11 && value >= -1 && value <= 2) { mi_hooks.sinkhorn_schedule = value; return MI_OK; }
  return MI_E_PARAM;
}

MI_API int mi_debug_bad_plan_passes(const uint32_t *pair_geom_host, int num_pairs, int *canonical, int *scheduled) {
  return mi_bad_plan_passes_host(pair_geom_host, num_pairs, canonical, scheduled);
}

// ---- the stream-schedule tuner's decision logic (csrc/sk_tuner.h) driven by a script, no GPU involved -------------
#include "sk_tuner.h"
MI_API int mi_debug_tuner_script(int n_ops, const int *op, const int *arg, const double *val, int *out) {
  if (n_ops < 0 || (n_ops > 0 && (!op || !arg || !val || !out))) return MI_E_NULL;
  mi::TunerLogic t;
  int last_entry = -1, last_slot = -1;
  for (int i = 0; i < n_ops; ++i) {
    mi::TunerShape s;
    s.batch = arg[i]; s.n = 512; s.m = 512; s.iterations = 20;
    switch (op[i]) {
      case 0: {                                                 // eager call of shape `arg`: schedule | slot << 8 | entry << 16
        const int sched = t.begin(s, &last_entry, &last_slot);
        out[i] = sched | ((last_slot & 0xff) << 8) | ((last_entry & 0xff) << 16);
        break;
      }
      case 1: t.finish(arg[i] >> 8, arg[i] & 0xff, val[i]); out[i] = 0; break;      // arg = entry << 8 | slot
      case 2: t.abandon(arg[i] >> 8, arg[i] & 0xff); out[i] = 0; break;
      case 3: out[i] = t.current(s); break;
      case 4: out[i] = t.for_capture(s); break;
      case 5: out[i] = t.set(arg[i]) ? 0 : -1; break;
      default: return MI_E_PARAM;
    }
  }
  return MI_OK;
}

MI_API int mi_debug_sinkhorn_dots_form(int batch, int n, int m, int flags, int blocks_per_cu, int cus) {
  return mi_sinkhorn_dots_form_host(batch, n, m, flags, blocks_per_cu, cus);
}

// ---- exhaustive check of csrc/akaze_math.h against the IEEE operators (tests/test_gpu_parity.py) -----------------
#include "akaze_math.h"
namespace {
// which: 0 = ak_sqrt(x) vs sqrtf(x); 1 = ak_div_by(x, kappa, 1 / kappa) vs x / kappa; 2 = ak_rcp(x) vs 1 / x;
// 3 = ak_div(x, kappa) vs x / kappa; 4 / 5 = ak_sqrt_fp<1> / <2>(x) vs sqrtf(x).  Every float with bit pattern in [lo_bits, hi_bits) is tried.
__global__ __launch_bounds__(256) void akaze_math_check_kernel(int which, float kappa, uint32_t lo_bits, uint32_t hi_bits,
                                                               unsigned long long *mismatches, uint32_t *first_bad) {
  const float rk = 1.0f / kappa;
  unsigned long long bad = 0;
  for (unsigned long long b = (unsigned long long)lo_bits + blockIdx.x * 256ull + threadIdx.x; b < hi_bits;
       b += (unsigned long long)gridDim.x * 256ull) {
    const float x = __uint_as_float((uint32_t)b);
    float got, want;
    if (which == 0) { got = ak_sqrt(x); want = sqrtf(x); }
    else if (which == 1) { got = ak_div_by(x, kappa, rk); want = x / kappa; }
    else if (which == 2) { got = ak_rcp(x); want = 1.0f / x; }
    else if (which == 3) { got = ak_div(x, kappa); want = x / kappa; }
    else if (which == 4) { got = ak_sqrt_fp<1>(x); want = sqrtf(x); }
    else { got = ak_sqrt_fp<2>(x); want = sqrtf(x); }
    if (__float_as_uint(got) != __float_as_uint(want)) {
      ++bad;
      atomicMin(first_bad, (uint32_t)b);
    }
  }
  if (bad) atomicAdd(mismatches, bad);
}
}  // namespace

MI_API int mi_debug_akaze_math_check(int which, float kappa, uint32_t lo_bits, uint32_t hi_bits, void *mismatches_u64,
                                     void *first_bad_u32, mi_stream_t stream) {
  MI_ENTER();
  if (!mismatches_u64 || !first_bad_u32) return MI_E_NULL;
  if (which < 0 || which > 5 || !(kappa > 0.0f) || lo_bits >= hi_bits) return MI_E_PARAM;
  hipLaunchKernelGGL(akaze_math_check_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, which, kappa, lo_bits, hi_bits,
                     reinterpret_cast<unsigned long long *>(mismatches_u64), reinterpret_cast<uint32_t *>(first_bad_u32));
  return mi_launch_status();
}
