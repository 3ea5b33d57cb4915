/* Stand-alone C consumer of libccqp_hip.so: no Python, no torch.  Generates a small SPD problem on
 * the GPU, runs one matvec + epilogue + dots + Box projection through the C ABI and checks them
 * against plain C loops on the host.  Built and run by tests/test_gpu_cabi.py:
 *   gcc -std=c99 tests/c/abi_smoke.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       -Lccqppy_amd -lccqp_hip -L/opt/rocm/lib -lamdhip64 -lm -o abi_smoke
 * (the define is what HIP's own headers require from a C translation unit; only hipMalloc/hipMemcpy
 * come from them).  */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ccqp_hip.h"

#define CHECK(call)                                                            \
  do {                                                                         \
    int rc_ = (call);                                                          \
    if (rc_ != 0) {                                                            \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ccqp_last_error()); \
      return 2;                                                                \
    }                                                                          \
  } while (0)

int main(void) {
  const int64_t n = 1500;
  ccqp_ctx* ctx = NULL;
  CHECK(ccqp_ctx_create(0, &ctx));
  char arch[64];
  int dev = -1, ncu = 0;
  CHECK(ccqp_ctx_info(ctx, &dev, &ncu, arch, sizeof(arch)));
  printf("device %d: %s, %d CUs, ABI %d\n", dev, arch, ncu, ccqp_abi_version());

  double *A, *x, *b, *y, *g, *p;
  if (hipMalloc((void**)&A, n * n * sizeof(double)) || hipMalloc((void**)&x, n * sizeof(double)) ||
      hipMalloc((void**)&b, n * sizeof(double)) || hipMalloc((void**)&y, n * sizeof(double)) ||
      hipMalloc((void**)&g, n * sizeof(double)) || hipMalloc((void**)&p, n * sizeof(double)))
    return 3;
  CHECK(ccqp_gen_spd_rows_f64(ctx, A, n, 0, n, n, 1234, NULL));
  CHECK(ccqp_gen_vec_f64(ctx, x, n, 1234, NULL));
  CHECK(ccqp_gen_vec_f64(ctx, b, n, 99, NULL));
  CHECK(ccqp_gemv_epilogue_f64(ctx, A, n, x, b, y, g, n, n, 0, NULL));
  const double* xs[2] = {x, g};
  const double* ys[2] = {y, g};
  CHECK(ccqp_dots_f64(ctx, 2, xs, ys, n, 0, NULL));
  ccqp_seg seg = {CCQP_OP_BOX, 0, 0, n, NULL, NULL, -0.5, 0.5};
  CHECK(ccqp_proj_elementwise_f64(ctx, &seg, 1.0, x, -0.25, g, p, NULL));   /* P(x - 0.25 g) */
  double s[2];
  CHECK(ccqp_scalars_read(ctx, 0, 2, s, NULL));

  double *hA = malloc(n * n * sizeof(double)), *hx = malloc(n * sizeof(double)), *hb = malloc(n * sizeof(double));
  double *hy = malloc(n * sizeof(double)), *hg = malloc(n * sizeof(double)), *hp = malloc(n * sizeof(double));
  hipMemcpy(hA, A, n * n * sizeof(double), hipMemcpyDeviceToHost);
  hipMemcpy(hx, x, n * sizeof(double), hipMemcpyDeviceToHost);
  hipMemcpy(hb, b, n * sizeof(double), hipMemcpyDeviceToHost);
  hipMemcpy(hy, y, n * sizeof(double), hipMemcpyDeviceToHost);
  hipMemcpy(hg, g, n * sizeof(double), hipMemcpyDeviceToHost);
  hipMemcpy(hp, p, n * sizeof(double), hipMemcpyDeviceToHost);

  int bad = 0;
  double xy = 0.0, gg = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    double acc = 0.0, mag = 0.0;
    for (int64_t j = 0; j < n; ++j) {
      acc += hA[i * n + j] * hx[j];
      mag += fabs(hA[i * n + j] * hx[j]);
      if (hA[i * n + j] != hA[j * n + i]) bad++;          /* generator: symmetric */
    }
    if (hA[i * n + i] != 2.0) bad++;                        /* generator: diagonal */
    if (fabs(acc - hy[i]) > 1e-13 * mag + 1e-300) bad++;    /* matvec */
    if (hg[i] != hy[i] + hb[i]) bad++;                      /* epilogue: one rounding */
    const double u = hx[i] + (-0.25) * hg[i];
    const double pr = u > 0.5 ? 0.5 : (u < -0.5 ? -0.5 : u);
    if (pr != hp[i]) bad++;                                 /* fused step + Box projection */
    xy += hx[i] * hy[i];
    gg += hg[i] * hg[i];
  }
  if (fabs(xy - s[0]) > 1e-12 * fabs(xy) + 1e-12) bad++;
  if (fabs(gg - s[1]) > 1e-12 * fabs(gg)) bad++;
  /* error path: message through ccqp_last_error */
  if (ccqp_gemv_f64(ctx, A, n - 1, x, y, n, n, 0, NULL) != CCQP_ERR_INVALID) bad++;
  printf("x.Ax = %.15g, |g|^2 = %.15g, mismatches = %d, last error: %s\n", s[0], s[1], bad, ccqp_last_error());
  hipFree(A); hipFree(x); hipFree(b); hipFree(y); hipFree(g); hipFree(p);
  CHECK(ccqp_ctx_destroy(ctx));
  return bad ? 1 : 0;
}
