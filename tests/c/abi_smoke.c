/* Stand-alone C consumer of libccqp_hip.so: no Python, no torch.  Generates a small SPD problem on
 * the GPU, runs one matvec + epilogue + dots + Box projection, the CSR matvec, a lock-step batch of three
 * BBPGD solves, a DisjointUnion-like set through the segmented engine, a lock-step SPG batch drawing from a C generator
 * and the 1-rank RCCL exchange through the C ABI
 * and checks them against plain C loops on the host.  Built and run by tests/test_gpu_cabi.py:
 *   gcc -std=c99 tests/c/abi_smoke.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       -Lccqppy_amd -lccqp_hip -L/opt/rocm/lib -lamdhip64 -lm -o abi_smoke
 * (the define is what HIP's own headers require from a C translation unit; only hipMalloc/hipMemcpy
 * come from them).  */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
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

#define HIPCHECK(call)                                          \
  do {                                                          \
    if ((call) != hipSuccess) {                                 \
      fprintf(stderr, "%s failed\n", #call);                    \
      return 3;                                                 \
    }                                                           \
  } while (0)

/* a caller-supplied source of uniform [0,1) doubles for the lock-step SPG driver (any generator with numpy's bitgen_t
 * layout will do; from Python it is a RandomState's own) */
static double lcg_next_double(void* state) {
  uint64_t* s = (uint64_t*)state;
  *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
  return (double)(*s >> 11) / 9007199254740992.0;
}

int main(void) {
  const int64_t n = 1500;
  ccqp_ctx* ctx = NULL;
  CHECK(ccqp_ctx_create(0, &ctx));
  char arch[64];
  int dev = -1, ncu = 0;
  CHECK(ccqp_ctx_info(ctx, &dev, &ncu, arch, sizeof(arch)));
  printf("device %d: %s, %d CUs, ABI %d\n", dev, arch, ncu, ccqp_abi_version());
  if (ccqp_abi_version() != CCQP_ABI_VERSION || CCQP_ABI_VERSION != 2) {
    fprintf(stderr, "ABI version mismatch\n");
    return 1;
  }
  {
    /* tuning options live in the context (nothing in the library reads the environment): set, read back, range check */
    int64_t v = -1;
    int bad_opt = 0;
    CHECK(ccqp_ctx_get_option(ctx, CCQP_OPT_GEMV_ROW_STRIDE, &v));
    if (v != 0) bad_opt++;
    CHECK(ccqp_ctx_set_option(ctx, CCQP_OPT_GEMV_ROW_STRIDE, 4));
    CHECK(ccqp_ctx_get_option(ctx, CCQP_OPT_GEMV_ROW_STRIDE, &v));
    if (v != 4) bad_opt++;
    CHECK(ccqp_ctx_set_option(ctx, CCQP_OPT_GEMV_ROW_STRIDE, 0));
    if (ccqp_ctx_set_option(ctx, CCQP_OPT_GEMV2_SHAPE, 45) != CCQP_ERR_INVALID) bad_opt++;
    if (ccqp_ctx_set_option(ctx, CCQP_OPT_COUNT, 1) != CCQP_ERR_INVALID) bad_opt++;
    if (bad_opt) {
      fprintf(stderr, "context options misbehave\n");
      return 1;
    }
  }

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
  /* sparse operator: validate -> plan -> CSR matvec with epilogue, against a C loop.  Rows of 0..7 entries
   * plus one row of 2500 entries (longer than a window: wave-per-row path); columns (5 i + 3 j) mod n. */
  {
    const int64_t rows = n, cols = n;
    int64_t* hrp = malloc((rows + 1) * sizeof(int64_t));
    hrp[0] = 0;
    for (int64_t i = 0; i < rows; ++i) hrp[i + 1] = hrp[i] + (i == 77 ? 2500 : i % 8);
    const int64_t nnz = hrp[rows];
    int32_t* hc = malloc(nnz * sizeof(int32_t));
    double* hv = malloc(nnz * sizeof(double));
    for (int64_t i = 0; i < rows; ++i)
      for (int64_t e = hrp[i]; e < hrp[i + 1]; ++e) {
        hc[e] = (int32_t)((5 * i + 3 * (e - hrp[i])) % cols);
        hv[e] = 0.25 + (double)((i + 7 * e) % 13) / 16.0;
      }
    int64_t *rp, *plan;
    int32_t* cidx;
    double* vals;
    const int64_t words = ccqp_spmv_plan_words(rows, nnz);
    if (words <= 0 || hipMalloc((void**)&rp, (rows + 1) * sizeof(int64_t)) || hipMalloc((void**)&cidx, nnz * sizeof(int32_t)) ||
        hipMalloc((void**)&vals, nnz * sizeof(double)) || hipMalloc((void**)&plan, words * sizeof(int64_t)))
      return 4;
    hipMemcpy(rp, hrp, (rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice);
    hipMemcpy(cidx, hc, nnz * sizeof(int32_t), hipMemcpyHostToDevice);
    hipMemcpy(vals, hv, nnz * sizeof(double), hipMemcpyHostToDevice);
    double viol = -1.0;
    CHECK(ccqp_csr_validate(ctx, rp, cidx, rows, cols, nnz, 5, NULL));
    CHECK(ccqp_scalars_read(ctx, 5, 1, &viol, NULL));
    if (viol != 0.0) bad++;
    int64_t summary[4] = {-1, -1, -1, -1};
    CHECK(ccqp_spmv_plan_csr(ctx, rp, cidx, rows, nnz, plan, summary, NULL));
    if (summary[0] != 1) bad++;                              /* exactly the 2500-entry row */
    CHECK(ccqp_spmv_csr_f64(ctx, rp, cidx, vals, plan, summary, x, b, y, g, rows, cols, nnz, 0, NULL));
    hipMemcpy(hy, y, n * sizeof(double), hipMemcpyDeviceToHost);
    hipMemcpy(hg, g, n * sizeof(double), hipMemcpyDeviceToHost);
    for (int64_t i = 0; i < rows; ++i) {
      double acc = 0.0, mag = 0.0;
      for (int64_t e = hrp[i]; e < hrp[i + 1]; ++e) {
        acc += hv[e] * hx[hc[e]];
        mag += fabs(hv[e] * hx[hc[e]]);
      }
      if (fabs(acc - hy[i]) > 1e-13 * mag + 1e-300) bad++;
      if (hrp[i + 1] - hrp[i] == 1 && hy[i] != hv[hrp[i]] * hx[hc[hrp[i]]]) bad++;   /* one entry: exact */
      if (hg[i] != hy[i] + hb[i]) bad++;
    }
    /* an out-of-range column must be counted by the validator */
    int32_t oob = (int32_t)cols;
    hipMemcpy(cidx + 3, &oob, sizeof(int32_t), hipMemcpyHostToDevice);
    CHECK(ccqp_csr_validate(ctx, rp, cidx, rows, cols, nnz, 5, NULL));
    CHECK(ccqp_scalars_read(ctx, 5, 1, &viol, NULL));
    if (viol != 1.0) bad++;
    printf("sparse: nnz = %lld, plan words = %lld, long rows = %lld\n", (long long)nnz, (long long)words,
           (long long)summary[0]);
    hipFree(rp); hipFree(cidx); hipFree(vals); hipFree(plan);
    free(hrp); free(hc); free(hv);
  }
  /* lock-step batch of three box-QPs (same A, different b) from plain C: descriptors in device memory, start,
   * queue iterations while polling the progress mirror, then check the KKT conditions of every solution. */
  {
    enum { NB = 3 };
    const double tol = 1e-8, gd = 1e-6, eps10 = 10.0 * 2.220446049250313e-16;
    const int64_t pd = ccqp_bpipe_partials_doubles(n);
    double *vecs, *slabs, *parts, *bs;
    uint32_t* words;
    ccqp_pprob hp3[NB], *dp3;
    if (pd <= 0 || hipMalloc((void**)&vecs, 7 * NB * n * sizeof(double)) || hipMalloc((void**)&slabs, NB * 64 * sizeof(double)) ||
        hipMalloc((void**)&parts, NB * pd * sizeof(double)) || hipMalloc((void**)&words, (NB + 2) * sizeof(uint32_t)) ||
        hipMalloc((void**)&bs, NB * n * sizeof(double)) || hipMalloc((void**)&dp3, sizeof(hp3)))
      return 5;
    hipMemset(vecs, 0, 7 * NB * n * sizeof(double));      /* x0 = 0 */
    hipMemset(slabs, 0, NB * 64 * sizeof(double));
    hipMemset(words, 0, (NB + 2) * sizeof(uint32_t));
    for (int q = 0; q < NB; ++q) {
      CHECK(ccqp_gen_vec_f64(ctx, bs + q * n, n, 500 + q, NULL));
      hp3[q].A = A;
      hp3[q].lda = n;
      hp3[q].b = bs + q * n;
      for (int j = 0; j < 3; ++j) {
        hp3[q].x[j] = vecs + ((int64_t)j * NB + q) * n;
        hp3[q].g[j] = vecs + ((int64_t)(3 + j) * NB + q) * n;
      }
      hp3[q].y = vecs + ((int64_t)6 * NB + q) * n;
      hp3[q].seg = seg;                                    /* Box [-0.5, 0.5] */
      hp3[q].slab = slabs + q * 64;
      hp3[q].partials = parts + q * pd;
      hp3[q].counter = words + q;
      hp3[q].c = 1.0 / (3 * n * gd);
      hp3[q].tol = tol;
      hp3[q].xmin = NULL;                                  /* (BBPGDf only) */
      hp3[q].gmin = NULL;
    }
    hipMemcpy(dp3, hp3, sizeof(hp3), hipMemcpyHostToDevice);
    const int flags = 1;                                   /* hipMalloc + even n: 16-byte aligned rows */
    CHECK(ccqp_bpipe_start_f64(ctx, dp3, NB, n, gd, flags, words + NB, NULL));
    double st[2] = {0.0, -1.0};
    int64_t k = 0;
    while (st[0] < NB && k < 500) {
      ++k;
      CHECK(ccqp_bpipe_iterate_f64(ctx, dp3, NB, n, k, gd, eps10, flags, 1, words + NB, NULL));
      do {
        CHECK(ccqp_bpipe_poll(ctx, st));
      } while (st[0] < NB && k - st[1] >= 4);              /* run at most 4 iterations ahead */
    }
    hipDeviceSynchronize();
    if (st[0] < NB) bad++;
    double hs[64];
    for (int q = 0; q < NB; ++q) {
      hipMemcpy(hs, slabs + q * 64, sizeof(hs), hipMemcpyDeviceToHost);
      const int iters = (int)hs[50];
      if (hs[49] != 1.0 || iters < 1 || sqrt(hs[51]) >= tol) bad++;
      hipMemcpy(hx, hp3[q].x[iters % 3], n * sizeof(double), hipMemcpyDeviceToHost);
      hipMemcpy(hb, hp3[q].b, n * sizeof(double), hipMemcpyDeviceToHost);
      int kkt = 0;
      for (int64_t i = 0; i < n; ++i) {
        double gi = hb[i];
        for (int64_t j = 0; j < n; ++j) gi += hA[i * n + j] * hx[j];
        if (hx[i] < -0.5 || hx[i] > 0.5) kkt++;
        else if (hx[i] == -0.5) kkt += gi < -1e-4;
        else if (hx[i] == 0.5) kkt += gi > 1e-4;
        else kkt += fabs(gi) > 1e-4;
      }
      if (kkt) bad++;
      printf("lock-step batch: problem %d converged after %d iterations, residual %.3g, KKT violations %d\n", q, iters,
             sqrt(hs[51]), kkt);
    }
    hipFree(vecs); hipFree(slabs); hipFree(parts); hipFree(words); hipFree(bs); hipFree(dp3);
  }
  /* a DisjointUnion-like feasible set through the segmented engine: 40 Sphere(3) blocks (one thread each), one
   * Box(200), one Sphere(1000, radius 2) (chunked) -- plan on the host, table / work lists / zeroed workspace in device
   * memory, then P(x - 0.25 g) and the Eq. 25 residual, each against a plain C loop. */
  {
    enum { NS = 42 };
    ccqp_seg hs[NS];
    int64_t off = 0;
    hipMemcpy(hx, x, n * sizeof(double), hipMemcpyDeviceToHost);   /* (the batch section reused the host buffers) */
    hipMemcpy(hg, g, n * sizeof(double), hipMemcpyDeviceToHost);
    for (int q = 0; q < 40; ++q, off += 3) {
      ccqp_seg t = {CCQP_OP_SPHERE, 0, off, 3, NULL, NULL, 0.3, 0.0};
      hs[q] = t;
    }
    ccqp_seg tb = {CCQP_OP_BOX, 0, off, 200, NULL, NULL, -0.25, 0.5};
    hs[40] = tb;
    off += 200;
    ccqp_seg ts = {CCQP_OP_SPHERE, 0, off, 1000, NULL, NULL, 2.0, 0.0};
    hs[41] = ts;
    off += 1000;                                           /* 1320 <= n */
    int64_t nwork = 0, nnorm = 0, npart = 0;
    CHECK(ccqp_segs_plan(hs, NS, NULL, 0, &nwork, NULL, 0, &nnorm, &npart));
    ccqp_segwork* hw = malloc((nwork + 1) * sizeof(ccqp_segwork));
    ccqp_segwork* hn = malloc((nnorm + 1) * sizeof(ccqp_segwork));
    CHECK(ccqp_segs_plan(hs, NS, hw, nwork, &nwork, hn, nnorm, &nnorm, &npart));
    if (nwork != 3 || nnorm != 1 || npart != 2) bad++;    /* run of 40 small blocks; Box chunk; Sphere chunk */
    const int64_t wsd = ccqp_segs_workspace_doubles(npart, nwork);
    ccqp_seg* dtab;
    ccqp_segwork *dw, *dn;
    double *ws, *po;
    if (hipMalloc((void**)&dtab, sizeof(hs)) || hipMalloc((void**)&dw, nwork * sizeof(ccqp_segwork)) ||
        hipMalloc((void**)&dn, (nnorm + 1) * sizeof(ccqp_segwork)) || hipMalloc((void**)&ws, wsd * sizeof(double)) ||
        hipMalloc((void**)&po, n * sizeof(double)))
      return 6;
    hipMemcpy(dtab, hs, sizeof(hs), hipMemcpyHostToDevice);
    hipMemcpy(dw, hw, nwork * sizeof(ccqp_segwork), hipMemcpyHostToDevice);
    hipMemcpy(dn, hn, nnorm * sizeof(ccqp_segwork), hipMemcpyHostToDevice);
    hipMemset(ws, 0, wsd * sizeof(double));
    hipMemset(po, 0, n * sizeof(double));
    CHECK(ccqp_segs_project_f64(ctx, dtab, dw, nwork, dn, nnorm, npart, ws, 1.0, x, -0.25, g, NULL, po, NULL));
    const double cres = 1.0 / (3.0 * (double)off * 1e-6);
    CHECK(ccqp_segs_residual_sq_f64(ctx, dtab, dw, nwork, dn, nnorm, npart, ws, x, g, cres, 1e-6, 7, NULL));
    double r2 = 0.0;
    CHECK(ccqp_scalars_read(ctx, 7, 1, &r2, NULL));
    hipMemcpy(hp, po, n * sizeof(double), hipMemcpyDeviceToHost);
    double want_r2 = 0.0;
    for (int pass = 0; pass < 2; ++pass) {                 /* pass 0: step 0.25 (projection); pass 1: step 1e-6 (residual) */
      const double step = pass == 0 ? 0.25 : 1e-6;
      for (int q = 0; q < NS; ++q) {
        double nrm = 0.0;
        for (int64_t i = 0; i < hs[q].dim; ++i) {
          const double u = hx[hs[q].offset + i] + (-step) * hg[hs[q].offset + i];
          nrm += u * u;
        }
        nrm = sqrt(nrm);
        for (int64_t i = 0; i < hs[q].dim; ++i) {
          const int64_t e = hs[q].offset + i;
          const double u = hx[e] + (-step) * hg[e];
          double pr;
          if (hs[q].kind == CCQP_OP_BOX) pr = u > hs[q].ub_s ? hs[q].ub_s : (u < hs[q].lb_s ? hs[q].lb_s : u);
          else pr = nrm > hs[q].lb_s ? hs[q].lb_s * u / nrm : u;
          if (pass == 0) {
            if (fabs(pr - hp[e]) > 1e-13 * (fabs(pr) + 1e-300)) bad++;
          } else {
            const double v = cres * (hx[e] - pr);
            want_r2 += v * v;
          }
        }
      }
    }
    if (fabs(r2 - want_r2) > 1e-10 * want_r2 + 1e-300) bad++;
    printf("segmented engine: %lld work items, %lld norm items, residual^2 = %.6g (C loop %.6g)\n", (long long)nwork,
           (long long)nnorm, r2, want_r2);
    hipFree(dtab); hipFree(dw); hipFree(dn); hipFree(ws); hipFree(po);
    free(hw); free(hn);
  }
  /* lock-step SPG-QP, two problems (one matrix, two right-hand sides), Box(-0.5, 0.5), draws from a C generator.
   * SPG-QP's own exit test |P(x - alpha g) - x| <= tol is all the reference promises (its step length can collapse
   * far from a KKT point), so the checks are the driver's: both problems stop on that test, draws = matvecs - 3, and a
   * second run of the same problems and generators, cut into driver calls of 7 rounds instead of one call, ends in
   * the same bits. */
  {
    const int B = 2;
    const int64_t nblk = ccqp_bpipe_partials_doubles(n) / 3;
    double *vec, *part, *xchg, *b2;
    uint32_t* words;
    if (hipMalloc((void**)&vec, 4 * B * n * sizeof(double)) || hipMalloc((void**)&part, B * 3 * nblk * sizeof(double)) ||
        hipMalloc((void**)&xchg, 8 * B * sizeof(double)) || hipMalloc((void**)&words, (B + 2) * sizeof(uint32_t)) ||
        hipMalloc((void**)&b2, n * sizeof(double)))
      return 8;
    CHECK(ccqp_gen_vec_f64(ctx, b2, n, 4321, NULL));
    ccqp_pspg hq[2];
    ccqp_pspg* dp;
    if (hipMalloc((void**)&dp, sizeof(hq))) return 8;
    double* xrun[2];
    xrun[0] = malloc(B * n * sizeof(double));
    xrun[1] = malloc(B * n * sizeof(double));
    long long mvs[2][2];
    for (int run = 0; run < 2; ++run) {
      ccqp_spg_host hh[2];
      double queue[2 * 5];
      uint64_t lcg[2] = {12345, 67890};
      ccqp_draw_source src[2];
      const ccqp_draw_source* srcs[2];
      hipMemset(vec, 0, 4 * B * n * sizeof(double));                 /* x0 = 0 */
      hipMemset(words, 0, (B + 2) * sizeof(uint32_t));
      for (int q = 0; q < B; ++q) {
        hq[q].A = A; hq[q].lda = n; hq[q].b = q ? b2 : b;
        hq[q].x = vec + (0 * B + q) * n; hq[q].g = vec + (1 * B + q) * n;
        hq[q].d = vec + (2 * B + q) * n; hq[q].Ad = vec + (3 * B + q) * n;
        hq[q].seg = seg;
        hq[q].partials = part + q * 3 * nblk; hq[q].counter = words + q;
        ccqp_spg_host z = {0};
        hh[q] = z;
        hh[q].tol = 1e-8; hh[q].max_mv = 2000; hh[q].tau = 0.5; hh[q].sigma1 = 0.01; hh[q].sigma2 = 0.5; hh[q].m = 5;
        src[q].state = &lcg[q]; src[q].next_uint64 = NULL; src[q].next_uint32 = NULL; src[q].next_raw = NULL;
        src[q].next_double = lcg_next_double;
        srcs[q] = &src[q];
      }
      hipMemcpy(dp, hq, sizeof(hq), hipMemcpyHostToDevice);
      hipDeviceSynchronize();
      int more = 1, start = 1, calls = 0;
      while (more) {
        CHECK(ccqp_bspg_run_f64(ctx, dp, NULL, hh, queue, 5, srcs, B, n, 0, 0, 1, start, run ? 7 : 100000, xchg,
                                xchg + 4 * B, words + B, &more, NULL));
        start = 0;
        calls++;
      }
      for (int q = 0; q < B; ++q) {
        if (hh[q].status != 1 || !(sqrt(hh[q].dd) <= 1e-8) || hh[q].used != hh[q].mv - 3) bad++;
        mvs[run][q] = (long long)hh[q].mv;
        hipMemcpy(xrun[run] + q * n, hq[q].x, n * sizeof(double), hipMemcpyDeviceToHost);
        if (run == 0)
          printf("lock-step SPG problem %d: %lld matvecs, %lld draws, |d| = %.3g\n", q, (long long)hh[q].mv,
                 (long long)hh[q].used, sqrt(hh[q].dd));
      }
      if ((run == 0) != (calls == 1)) bad++;
    }
    int differ = 0;
    for (int64_t i = 0; i < B * n; ++i) {
      if (xrun[0][i] != xrun[1][i]) differ++;
      if (xrun[0][i] < -0.5 || xrun[0][i] > 0.5) differ++;
    }
    if (differ || mvs[0][0] != mvs[1][0] || mvs[0][1] != mvs[1][1]) bad++;
    printf("lock-step SPG: one driver call vs calls of 7 rounds: %d entries differ\n", differ);
    free(xrun[0]); free(xrun[1]);
    hipFree(vec); hipFree(part); hipFree(xchg); hipFree(words); hipFree(b2); hipFree(dp);
  }
  /* the multi-GPU exchange step with the one rank a single-GPU box allows: bootstrap, sharded matvec (bitwise the
   * plain matvec), in-place collectives on the NULL stream.  Skipped (not failed) where no RCCL can be loaded. */
  if (ccqp_comm_rccl_version() > 0) {
    unsigned char id[CCQP_COMM_ID_BYTES];
    ccqp_comm* comm = NULL;
    CHECK(ccqp_comm_unique_id(id));
    CHECK(ccqp_comm_create(ctx, 1, 0, id, &comm));
    double* y2;
    if (hipMalloc((void**)&y2, n * sizeof(double))) return 7;
    CHECK(ccqp_gemv_f64(ctx, A, n, x, y, n, n, 0, NULL));  /* the plain matvec (y was reused by the sections above) */
    hipMemcpy(hy, y, n * sizeof(double), hipMemcpyDeviceToHost);
    CHECK(ccqp_shard_gemv_allgather_f64(ctx, comm, A, n, x, y2, n, n, n, 0, 0, NULL));
    CHECK(ccqp_comm_allreduce_sum_f64(comm, y2, n, NULL));
    hipDeviceSynchronize();
    double* hy2 = malloc(n * sizeof(double));
    hipMemcpy(hy2, y2, n * sizeof(double), hipMemcpyDeviceToHost);
    for (int64_t i = 0; i < n; ++i)
      if (hy2[i] != hy[i]) bad++;
    printf("rccl %d: sharded matvec through a 1-rank communicator matches the plain matvec\n", ccqp_comm_rccl_version());
    CHECK(ccqp_comm_destroy(comm));
    hipFree(y2);
    free(hy2);
  } else {
    printf("rccl: not available, communicator section skipped\n");
  }
  /* two matvecs in one pass over A (the LDS-staged form forced for this small matrix: variant 0x51001800) against two
     single matvecs: the same bits */
  {
    double *y1, *y2, *g2, *r1, *r2;
    double *h1 = (double*)malloc(n * sizeof(double)), *h2 = (double*)malloc(n * sizeof(double));
    char form[64];
    if (hipMalloc((void**)&y1, n * sizeof(double)) || hipMalloc((void**)&y2, n * sizeof(double)) ||
        hipMalloc((void**)&g2, n * sizeof(double)) || hipMalloc((void**)&r1, n * sizeof(double)) ||
        hipMalloc((void**)&r2, n * sizeof(double)))
      return 1;
    CHECK(ccqp_gemv_f64(ctx, A, n, x, r1, n, n, 0, NULL));
    CHECK(ccqp_gemv_epilogue_f64(ctx, A, n, b, b, y2, r2, n, n, 0, NULL));      /* A.b and A.b + b */
    CHECK(ccqp_gemv2_f64(ctx, A, n, x, b, b, y1, NULL, y2, g2, n, n, 0x51001800, NULL));
    CHECK(ccqp_gemv2_form(ctx, n, n, 0x51001800, form, 64));
    if (strcmp(form, "gemv_lds2_kernel")) bad++;
    CHECK(ccqp_gemv2_form(ctx, n, n, 0, form, 64));
    if (strcmp(form, "two single matvecs")) bad++;
    hipDeviceSynchronize();
    hipMemcpy(h1, r1, n * sizeof(double), hipMemcpyDeviceToHost);
    hipMemcpy(h2, y1, n * sizeof(double), hipMemcpyDeviceToHost);
    if (memcmp(h1, h2, n * sizeof(double))) bad++;
    hipMemcpy(h1, r2, n * sizeof(double), hipMemcpyDeviceToHost);
    hipMemcpy(h2, g2, n * sizeof(double), hipMemcpyDeviceToHost);
    if (memcmp(h1, h2, n * sizeof(double))) bad++;
    printf("two-vector matvec: A.x and A.b + b in one pass, bits of the single matvecs: %s\n", bad ? "NO" : "yes");
    /* three products of the square matrix in one pass (ccqp_gemv3_f64): slot 1 unused */
    {
      const double* xs[3] = {x, NULL, b};
      double* ys[3] = {y1, NULL, y2};
      double* gs[3] = {NULL, NULL, g2};
      hipMemset(y1, 0xff, n * sizeof(double));
      hipMemset(g2, 0xff, n * sizeof(double));
      CHECK(ccqp_gemv3_f64(ctx, A, n, xs, b, ys, gs, n, 0x51001800, NULL));
      CHECK(ccqp_gemv3_form(ctx, n, 0x51001800, form, 64));
      if (strcmp(form, "bp_gemv3_pers_kernel")) bad++;
      CHECK(ccqp_gemv3_form(ctx, n, 0, form, 64));
      if (strcmp(form, "single matvecs")) bad++;
      hipDeviceSynchronize();
      hipMemcpy(h1, r1, n * sizeof(double), hipMemcpyDeviceToHost);
      hipMemcpy(h2, y1, n * sizeof(double), hipMemcpyDeviceToHost);
      if (memcmp(h1, h2, n * sizeof(double))) bad++;
      hipMemcpy(h1, r2, n * sizeof(double), hipMemcpyDeviceToHost);
      hipMemcpy(h2, g2, n * sizeof(double), hipMemcpyDeviceToHost);
      if (memcmp(h1, h2, n * sizeof(double))) bad++;
      printf("three-vector matvec: bits of the single matvecs: %s\n", bad ? "NO" : "yes");
    }
    free(h1); free(h2);
    hipFree(y1); hipFree(y2); hipFree(g2); hipFree(r1); hipFree(r2);
  }
  /* how a shape will be launched: the small matrix in one launch of the VALU form; the headline matrix as ONE
     persistent launch of the LDS-staged form by default, in launch rounds when the VALU form is forced (0xF) */
  {
    int64_t per = 0, launches = 0;
    char form[64];
    CHECK(ccqp_gemv_plan(ctx, n, n, 0, &per, &launches));
    if (per != n || launches != 1) bad++;
    CHECK(ccqp_gemv_form(ctx, n, n, 0, form, 64));
    if (strcmp(form, "gemv_stream_kernel")) bad++;
    CHECK(ccqp_gemv_plan(ctx, 131072, 131072, 0, &per, &launches));
    if (launches != 1 || per != 131072) bad++;
    CHECK(ccqp_gemv_form(ctx, 131072, 131072, 0, form, 64));
    if (strcmp(form, "gemv_lds_kernel")) bad++;
    CHECK(ccqp_gemv_plan(ctx, 131072, 131072, 0xF, &per, &launches));
    if (launches < 2 || per * launches < 131072) bad++;
    CHECK(ccqp_gemv_form(ctx, 131072, 131072, 0xF, form, 64));
    if (strcmp(form, "gemv_stream_kernel")) bad++;
    printf("matvec plan: %lld x %lld -> 1 launch; 131072 x 131072 -> 1 launch (LDS-staged), VALU form: %lld launches of "
           "%lld rows\n", (long long)n, (long long)n, (long long)launches, (long long)per);
    /* the batched matvec of the lock-step forms: one persistent LDS-DMA launch from n = 4096 and 1 GiB on */
    CHECK(ccqp_bpipe_matvec_form(ctx, 128, 8192, 1, form, 64));
    if (strcmp(form, "bp_gemv_pers_kernel")) bad++;
    CHECK(ccqp_bpipe_matvec_form(ctx, 8, 512, 1, form, 64));
    if (strcmp(form, "batched VALU kernel")) bad++;
    CHECK(ccqp_gemvk_form(ctx, 3, 8192, 16384, 0, form, 64));
    if (strcmp(form, "gemv_lds3_kernel")) bad++;
  }
  {
    /* three products of one block in one call (ccqp_gemvk_f64, the LDS-staged form forced onto this small matrix) and
     * the unpack launch of the row-sharded path (one "rank"): every product bitwise its own ccqp_gemv_epilogue_f64 */
    double *w3, *y3, *g3;
    HIPCHECK(hipMalloc((void**)&w3, 3 * n * sizeof(double)));
    HIPCHECK(hipMalloc((void**)&y3, 3 * n * sizeof(double)));
    HIPCHECK(hipMalloc((void**)&g3, 3 * n * sizeof(double)));
    const double* xs[3] = {x, b, p};
    double* ws[3] = {w3, w3 + n, w3 + 2 * n};
    double* none[3] = {NULL, NULL, NULL};
    double* ys[3] = {y3, y3 + n, y3 + 2 * n};
    double* gs[3] = {g3, NULL, g3 + 2 * n};
    CHECK(ccqp_gemvk_f64(ctx, A, n, 3, xs, NULL, ws, none, n, n, 0x51001800, NULL));
    CHECK(ccqp_unpack_gathered_f64(ctx, w3, 3, n, n, ys, b, gs, NULL));
    double* hy = (double*)malloc(3 * n * sizeof(double));
    double* hg = (double*)malloc(3 * n * sizeof(double));
    double* hr = (double*)malloc(2 * n * sizeof(double));
    HIPCHECK(hipMemcpy(hy, y3, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(hg, g3, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    for (int j = 0; j < 3; ++j) {
      CHECK(ccqp_gemv_epilogue_f64(ctx, A, n, xs[j], b, y, g, n, n, 0, NULL));
      HIPCHECK(hipMemcpy(hr, y, n * sizeof(double), hipMemcpyDeviceToHost));
      HIPCHECK(hipMemcpy(hr + n, g, n * sizeof(double), hipMemcpyDeviceToHost));
      if (memcmp(hr, hy + j * n, n * sizeof(double))) bad++;
      if (j != 1 && memcmp(hr + n, hg + j * n, n * sizeof(double))) bad++;
    }
    printf("three products in one call + unpack: mismatches so far %d\n", bad);
    free(hy); free(hg); free(hr);
    hipFree(w3); hipFree(y3); hipFree(g3);
    /* leave y / g as the rest of the program expects them */
    CHECK(ccqp_gemv_epilogue_f64(ctx, A, n, x, b, y, g, n, n, 0, NULL));
  }
  /* error path: message through ccqp_last_error */
  if (ccqp_gemv_f64(ctx, A, n - 1, x, y, n, n, 0, NULL) != CCQP_ERR_INVALID) bad++;
  printf("x.Ax = %.15g, |g|^2 = %.15g, mismatches = %d, last error: %s\n", s[0], s[1], bad, ccqp_last_error());
  hipFree(A); hipFree(x); hipFree(b); hipFree(y); hipFree(g); hipFree(p);
  CHECK(ccqp_ctx_destroy(ctx));
  return bad ? 1 : 0;
}
