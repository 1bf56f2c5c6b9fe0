/* A host WITHOUT Python or torch on the C ABI of include/mi355x_match.h: plain C, the HIP runtime for device memory
 * and nothing else.  It is what a C / C++ / cgo / JNI service would write around the hot path
 * ("The whole path in one call") and it is a test: tests/test_gpu_parity.py builds it with hipcc, feeds it
 * the frames and pair table the Python modules get, and compares its match records with theirs bit for bit.
 *
 *   host_match_pairs <libmi355x_match.so> <in.bin> <out.bin>
 * in.bin : int32 header {batch, h, w, max_keypoints, num_pairs, max_matches}, then uint8 frames image1 (batch*h*w),
 *          image2 (batch*h*w), then uint32 pair_geom[num_pairs], float pair_thr[num_pairs].
 * out.bin: float keypoints1 (batch*K*2), keypoints2, matched1 (batch*Mx*2), matched2, scores (batch*Mx), then uint8
 *          valid (batch*Mx), then uint32 solver status word, then int32 rc of every call made.
 * The library is opened with dlopen: the binary has no link-time dependency on it either. */
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355x_match.h"

#define CHECK_HIP(x)                                                    \
  do {                                                                  \
    hipError_t e_ = (x);                                                \
    if (e_ != hipSuccess) {                                             \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
      return 2;                                                         \
    }                                                                   \
  } while (0)

typedef int (*abi_version_fn)(void);
typedef size_t (*plan_bytes_fn)(int);
typedef int (*plan_build_fn)(const uint32_t *, const float *, int, void *, mi_stream_t);
typedef size_t (*ws_bytes_fn)(int, int, int, const mi_match_params *);
typedef int (*match_u8_fn)(const uint8_t *, const uint8_t *, int, int, int, const mi_match_params *, float *, float *, float *,
                           float *, float *, uint8_t *, int32_t *, void *, size_t, mi_stream_t);
typedef const char *(*err_fn)(int);

int main(int argc, char **argv) {
  if (argc != 4) {
    fprintf(stderr, "usage: %s <library> <in.bin> <out.bin>\n", argv[0]);
    return 1;
  }
  void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 1;
  }
  abi_version_fn abi_version = (abi_version_fn)dlsym(lib, "mi_abi_version");
  plan_bytes_fn plan_bytes = (plan_bytes_fn)dlsym(lib, "mi_bad_plan_bytes");
  plan_build_fn plan_build = (plan_build_fn)dlsym(lib, "mi_bad_plan_build");
  ws_bytes_fn ws_bytes = (ws_bytes_fn)dlsym(lib, "mi_match_pairs_workspace_bytes");
  match_u8_fn match_u8 = (match_u8_fn)dlsym(lib, "mi_match_pairs_u8");
  err_fn err_string = (err_fn)dlsym(lib, "mi_error_string");
  if (!abi_version || !plan_bytes || !plan_build || !ws_bytes || !match_u8 || !err_string) {
    fprintf(stderr, "missing symbol\n");
    return 1;
  }
  if (abi_version() != 3) {
    fprintf(stderr, "ABI version %d, expected 3\n", abi_version());
    return 1;
  }

  FILE *in = fopen(argv[2], "rb");
  if (!in) return 1;
  int32_t hdr[6];
  if (fread(hdr, sizeof(int32_t), 6, in) != 6) return 1;
  const int batch = hdr[0], h = hdr[1], w = hdr[2], K = hdr[3], P = hdr[4], Mx = hdr[5];
  const size_t npix = (size_t)batch * h * w;
  uint8_t *f1 = (uint8_t *)malloc(npix), *f2 = (uint8_t *)malloc(npix);
  uint32_t *geom = (uint32_t *)malloc(sizeof(uint32_t) * P);
  float *thr = (float *)malloc(sizeof(float) * P);
  if (fread(f1, 1, npix, in) != npix || fread(f2, 1, npix, in) != npix || fread(geom, 4, P, in) != (size_t)P ||
      fread(thr, 4, P, in) != (size_t)P)
    return 1;
  fclose(in);

  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  uint8_t *d_f1, *d_f2, *d_valid;
  uint32_t *d_geom;
  float *d_thr, *d_kp1, *d_kp2, *d_mk1, *d_mk2, *d_sc;
  void *d_plan, *d_ws;
  CHECK_HIP(hipMalloc((void **)&d_f1, npix));
  CHECK_HIP(hipMalloc((void **)&d_f2, npix));
  CHECK_HIP(hipMalloc((void **)&d_geom, sizeof(uint32_t) * P));
  CHECK_HIP(hipMalloc((void **)&d_thr, sizeof(float) * P));
  CHECK_HIP(hipMalloc(&d_plan, plan_bytes(P)));
  CHECK_HIP(hipMalloc((void **)&d_kp1, sizeof(float) * batch * K * 2));
  CHECK_HIP(hipMalloc((void **)&d_kp2, sizeof(float) * batch * K * 2));
  CHECK_HIP(hipMalloc((void **)&d_mk1, sizeof(float) * batch * Mx * 2));
  CHECK_HIP(hipMalloc((void **)&d_mk2, sizeof(float) * batch * Mx * 2));
  CHECK_HIP(hipMalloc((void **)&d_sc, sizeof(float) * batch * Mx));
  CHECK_HIP(hipMalloc((void **)&d_valid, (size_t)batch * Mx));
  CHECK_HIP(hipMemcpyAsync(d_f1, f1, npix, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_f2, f2, npix, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_geom, geom, sizeof(uint32_t) * P, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_thr, thr, sizeof(float) * P, hipMemcpyHostToDevice, stream));

  int32_t rcs[2];
  rcs[0] = plan_build(d_geom, d_thr, P, d_plan, stream);                 /* once per pair table */
  if (rcs[0]) fprintf(stderr, "mi_bad_plan_build: %s\n", err_string(rcs[0]));

  mi_match_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.block_size = 3;
  prm.nms_radius = 5;
  prm.max_keypoints = K;
  prm.score_threshold = 0.0f;
  prm.border_margin = 7;
  prm.num_pairs = P;
  prm.pair_geom = d_geom;
  prm.pair_thr = d_thr;
  prm.bad_plan = d_plan;
  prm.normalize_descriptors = 1;
  prm.epsilon = 0.05;
  prm.unused_score = 1.0;
  prm.sinkhorn_iterations = 20;
  prm.max_matches = Mx;
  prm.match_threshold = 0.1f;
  prm.flags = MI_SOLVER_DEFAULT;
  const size_t ws = ws_bytes(batch, h, w, &prm);
  if (ws == 0) {
    fprintf(stderr, "mi_match_pairs_workspace_bytes: parameters not covered\n");
    return 1;
  }
  CHECK_HIP(hipMalloc(&d_ws, ws));
  CHECK_HIP(hipMemsetAsync(d_ws, 0xA5, ws, stream));                     /* the workspace may hold anything */
  rcs[1] = match_u8(d_f1, d_f2, batch, h, w, &prm, d_kp1, d_kp2, d_mk1, d_mk2, d_sc, d_valid, NULL, d_ws, ws, stream);
  if (rcs[1]) fprintf(stderr, "mi_match_pairs_u8: %s\n", err_string(rcs[1]));
  CHECK_HIP(hipStreamSynchronize(stream));

  const size_t nk = (size_t)batch * K * 2, nm = (size_t)batch * Mx * 2, ns = (size_t)batch * Mx;
  float *out_f = (float *)malloc(sizeof(float) * (2 * nk + 2 * nm + ns));
  uint8_t *out_v = (uint8_t *)malloc(ns);
  CHECK_HIP(hipMemcpy(out_f, d_kp1, sizeof(float) * nk, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(out_f + nk, d_kp2, sizeof(float) * nk, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(out_f + 2 * nk, d_mk1, sizeof(float) * nm, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(out_f + 2 * nk + nm, d_mk2, sizeof(float) * nm, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(out_f + 2 * nk + 2 * nm, d_sc, sizeof(float) * ns, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(out_v, d_valid, ns, hipMemcpyDeviceToHost));
  FILE *out = fopen(argv[3], "wb");
  if (!out) return 1;
  fwrite(out_f, sizeof(float), 2 * nk + 2 * nm + ns, out);
  fwrite(out_v, 1, ns, out);
  fwrite(rcs, sizeof(int32_t), 2, out);
  fclose(out);
  printf("host_match_pairs: %d pairs %dx%d K=%d, rc %d %d, workspace %zu bytes\n", batch, w, h, K, rcs[0], rcs[1], ws);
  return (rcs[0] || rcs[1]) ? 3 : 0;
}
