// Lock-step BBPGD / PGD for feasible sets that are NOT elementwise: Sphere, Cone, DisjointUnion with blocks of any
// size (opt-in: ccqppy_amd.batch.solve_batch_pipelined picks it for such sets).
//
// ccqp_bpipe.hip advances many dense problems of one size in lock-step with every scalar on the device, but its
// projection is the one-expression-per-element kind.  Here the projection, the fused step and the Eq. 25 residual go
// through the segmented engine (ccqp_segs.h): each problem brings its block table and work lists (struct ccqp_psegs)
// and an iteration of ALL problems is
//   [norm]  chunk sums of |x_cur - alpha g_cur|^2 for the Sphere / Cone blocks      (only if any problem has one)
//   step    x_new = P(x_cur - alpha g_cur)                                          one workgroup per work item
//   matvec  g_new = A x_new + b                                                     the batched matvec of ccqp_bpipe
//   [norm]  chunk sums of |x_new - gd g_new|^2
//   resid   per work item, sum (c (x_new - P(x_new - gd g_new)))^2
//   tail    dx.dx, dx.dg over the reductions' own geometry; the problem's last block adds the work items' residual
//           sums in index order, then alpha = dx.dx / (dx.dg + 10 eps) and the exit test (solvers.py:647-656)
// i.e. 4-6 launches for the whole batch and no host round trip.  Reference: the Python loops of
// CCQPSolverBBPGD.solve / CCQPSolverPGD.solve (solvers.py:583-669, 94-170) around DisjointProjOp / SphereProjOp /
// ConeProjOp.__call__ (solution_spaces.py:540-560, 417-435, 470-492), which the ensemble benchmark
// (benchmarks/benchmark_random_ccqp.py:67-102) runs thousands of times.
// Bits: per element and per sum this is the arithmetic of ccqp_segs_project_f64 / ccqp_segs_residual_sq_f64 /
// ccqp_dots_f64 / the host loop's scalar step, so every problem gets exactly the result of its own solve().
#include "ccqp_lockstep.h"
#include "ccqp_segs.h"

// Operands of a projection launch.  mode 0: the step u = x[ix] - alpha*g[ix] -> x[inew];  mode 1: the residual's
// u = x[ix] - gd*g[ix];  mode 2 (BBPGDf, solvers.py:799-800): the restart u = xmin - gd*gmin -> x[inew], only for
// problems whose slab says "fall back".
struct bsegs_io {
  const double* x;
  const double* y;
  double b;
  bool live;
};

__device__ __forceinline__ bsegs_io bsegs_operands(const ccqp_pprob& q, int mode, int ix, double gd) {
  bsegs_io io = {nullptr, nullptr, 0.0, false};
  if (q.slab[BP_DONE] != 0.0) return io;
  if (mode == 2) {
    if (q.slab[BP_FALLBACK] == 0.0) return io;
    io.x = q.xmin;
    io.y = q.gmin;
    io.b = -gd;
  } else {
    io.x = q.x[ix];
    io.y = q.g[ix];
    io.b = mode == 0 ? -q.slab[BP_ALPHA] : -gd;
  }
  io.live = true;
  return io;
}

// chunk sums of |u|^2 for the Sphere / Cone blocks.  grid (max nnorm, problems)
__global__ __launch_bounds__(CCQP_BLOCK) void bsegs_norm_kernel(const ccqp_pprob* __restrict__ probs,
                                                                const ccqp_psegs* __restrict__ segs, int mode, int ix,
                                                                double gd) {
  __shared__ double lds4[4];
  const ccqp_psegs& ps = segs[blockIdx.y];
  if ((int)blockIdx.x >= ps.nnorm) return;
  const bsegs_io io = bsegs_operands(probs[blockIdx.y], mode, ix, gd);
  if (!io.live) return;
  const ccqp_segwork w = ps.normwork[blockIdx.x];
  const double s = segs_norm_item(ps.table, w, 1.0, io.x, io.b, io.y, lds4);
  if (threadIdx.x == 0) ps.workspace[w.pbase + w.chunk] = s;
}

// x[inew] = P(u), u by `mode` (0: the step from x[icur]; 2: BBPGDf's restart).  grid (max nwork, problems)
__global__ __launch_bounds__(CCQP_BLOCK) void bsegs_step_kernel(const ccqp_pprob* __restrict__ probs,
                                                                const ccqp_psegs* __restrict__ segs, int inew,
                                                                int icur, int mode, double gd) {
  __shared__ double lds4[4];
  __shared__ double bcast;
  const ccqp_psegs& ps = segs[blockIdx.y];
  if ((int)blockIdx.x >= ps.nwork) return;
  const ccqp_pprob& q = probs[blockIdx.y];
  const bsegs_io io = bsegs_operands(q, mode, icur, gd);
  if (!io.live) return;
  const ccqp_segwork w = ps.work[blockIdx.x];
  (void)segs_apply_item<0>(ps.table, w, ps.workspace, 1.0, io.x, io.b, io.y, nullptr, q.x[inew], 0.0, lds4, &bcast);
}

// per work item: sum (c*(x - P(x - gd*g)))^2 -> the residual partials of the problem's workspace
__global__ __launch_bounds__(CCQP_BLOCK) void bsegs_resid_kernel(const ccqp_pprob* __restrict__ probs,
                                                                 const ccqp_psegs* __restrict__ segs, int ix,
                                                                 double gd) {
  __shared__ double lds4[4];
  __shared__ double bcast;
  const ccqp_pprob& q = probs[blockIdx.y];
  if (q.slab[BP_DONE] != 0.0) return;
  const ccqp_psegs& ps = segs[blockIdx.y];
  if ((int)blockIdx.x >= ps.nwork) return;
  const ccqp_segwork w = ps.work[blockIdx.x];
  const double racc = segs_apply_item<1>(ps.table, w, ps.workspace, 1.0, q.x[ix], -gd, q.g[ix], nullptr, nullptr, q.c, lds4,
                                         &bcast);
  const double s = ccqp_block_sum(racc, lds4);
  if (threadIdx.x == 0) ps.workspace[3 * ps.npartials + blockIdx.x] = s;
}

// init != 0: the start-up decision (residual of x0, g0.g0; solvers.py:629-635);  else the tail of iteration `seq`.
// grid (nblk, problems) -- the geometry of ccqp_dots_f64 on n-vectors
__global__ __launch_bounds__(CCQP_BLOCK) void bsegs_tail_kernel(const ccqp_pprob* __restrict__ probs,
                                                                const ccqp_psegs* __restrict__ segs, int inew,
                                                                int icur, double eps10, int64_t n, int64_t chunk,
                                                                int nblk, unsigned* __restrict__ batch,
                                                                double* __restrict__ host, unsigned count, double seq,
                                                                int const_step, int init) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  const ccqp_pprob& q = probs[blockIdx.y];
  const bp_words z = bp_enter(q.slab);
  if (z.done != 0.0) {                      // converged earlier: only the roll-call
    if (blockIdx.x == 0 && threadIdx.x == 0) bp_report(batch, host, count, seq);
    return;
  }
  const ccqp_psegs& ps = segs[blockIdx.y];
  double v[2] = {0.0, 0.0};
  if (init) {
    v[0] = dot_body(q.g[0], q.g[0], n, chunk, blockIdx.x, lds4);
  } else {
    // dx = x_new - x_old, dg = g_new - g_old formed on the fly (each rounded once, like the axpby they replace), the
    // fma chains and thread -> element map of dot_body
    const double* __restrict__ xn = q.x[inew];
    const double* __restrict__ xo = q.x[icur];
    const double* __restrict__ gn = q.g[inew];
    const double* __restrict__ go = q.g[icur];
    const int bx = blockIdx.x;
    CHUNK_RANGE(n)
    double a1 = 0.0, a2 = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
      const double dx = xn[i] - xo[i];
      const double dg = gn[i] - go[i];
      a1 = fma(dx, dx, a1);
      a2 = fma(dx, dg, a2);
    }
    v[0] = ccqp_block_sum(a1, lds4);
    v[1] = ccqp_block_sum(a2, lds4);
  }
  if (!bp_meet(q.partials, q.counter, nblk, init ? 1 : 2, v, lds4, &is_last)) return;
  // the problem's last block: the residual = the work items' sums added in index order (ccqp_segs_residual_sq_f64)
  const double* respart = ps.workspace + 3 * ps.npartials;
  double r;
  if (ps.nwork == 1) r = respart[0];
  else r = partial_sum_body(respart, ps.nwork, lds4);
  if (threadIdx.x != 0) return;
  double* s = q.slab;
  if (init) {
    s[0] = r;
    s[BP_GG] = v[0];
    s[BP_RES2] = r;
    s[BP_ITERS] = 0.0;
    s[BP_RESMIN] = INFINITY;
    if (!const_step) s[BP_ALPHA] = 0.0;
    if (sqrt(r) >= q.tol) {                  // `if res >= tol:` -- a NaN residual skips the loop, as in the reference
      s[BP_DONE] = 0.0;
    } else {
      s[BP_DONE] = 1.0;
      __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bp_report(batch, host, count, seq);
    return;
  }
  const double res = sqrt(r);
  const double den = v[1] + eps10;
  s[0] = r;
  s[1] = v[0];
  s[2] = v[1];
  s[BP_RES2] = r;
  if (!const_step) s[BP_ALPHA] = v[0] / den;
  s[BP_ITERS] = z.iters + 1.0;
  if (res < q.tol) {
    s[BP_DONE] = 1.0;
    __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  bp_report(batch, host, count, seq);
}

// ---- BBPGDf (solvers.py:719-819): the tail in two parts, as in ccqp_bpipe.hip.  Part A: the residual (work-item
// sums in index order) -> exit test, "best so far", "fall back";  keep: xmin / gmin = x_new / g_new where improved;
// restart: x_new = P(xmin - gd gmin) where the step length has collapsed (bsegs_norm / bsegs_step, mode 2);  part B:
// dx.dx, dx.dg with the possibly replaced x_new -> alpha.
__global__ __launch_bounds__(CCQP_BLOCK) void bsegs_tailf_a_kernel(const ccqp_pprob* __restrict__ probs,
                                                                   const ccqp_psegs* __restrict__ segs, double eps10,
                                                                   unsigned* __restrict__ batch) {
  __shared__ double lds4[4];
  const ccqp_pprob& q = probs[blockIdx.x];
  const bp_words z = bp_enter(q.slab);      // (one block per problem; the same entry discipline all the same)
  if (z.done != 0.0) return;
  const ccqp_psegs& ps = segs[blockIdx.x];
  const double* respart = ps.workspace + 3 * ps.npartials;
  const double r = ps.nwork == 1 ? respart[0] : partial_sum_body(respart, ps.nwork, lds4);
  if (threadIdx.x == 0 && bbf_decide_a(q.slab, z, r, q.tol, eps10)) {
    q.slab[BP_ITERS] = z.iters + 1.0;
    q.slab[BP_DONE] = 1.0;
    __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void bsegs_keep_best_kernel(const ccqp_pprob* __restrict__ probs, int inew,
                                                                     int64_t n) {
  const ccqp_pprob& q = probs[blockIdx.y];
  if (q.slab[BP_DONE] != 0.0 || q.slab[BP_IMPROVED] == 0.0) return;
  copy_body(q.x[inew], q.xmin, n, blockIdx.x, gridDim.x);
  copy_body(q.g[inew], q.gmin, n, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(CCQP_BLOCK) void bsegs_tailf_b_kernel(const ccqp_pprob* __restrict__ probs, int inew,
                                                                   int icur, double eps10, int64_t n, int64_t chunk,
                                                                   int nblk, unsigned* __restrict__ batch,
                                                                   double* __restrict__ host, unsigned count,
                                                                   double seq) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  const ccqp_pprob& q = probs[blockIdx.y];
  const bp_words z = bp_enter(q.slab);
  if (z.done != 0.0) {                      // converged (now or earlier): only the roll-call
    if (blockIdx.x == 0 && threadIdx.x == 0) bp_report(batch, host, count, seq);
    return;
  }
  const double* __restrict__ xn = q.x[inew];
  const double* __restrict__ xo = q.x[icur];
  const double* __restrict__ gn = q.g[inew];
  const double* __restrict__ go = q.g[icur];
  const int bx = blockIdx.x;
  CHUNK_RANGE(n)
  double a1 = 0.0, a2 = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
    const double dx = xn[i] - xo[i];
    const double dg = gn[i] - go[i];
    a1 = fma(dx, dx, a1);
    a2 = fma(dx, dg, a2);
  }
  double v[2];
  v[0] = ccqp_block_sum(a1, lds4);
  v[1] = ccqp_block_sum(a2, lds4);
  if (!bp_meet(q.partials, q.counter, nblk, 2, v, lds4, &is_last)) return;
  if (threadIdx.x == 0) {
    const double den = v[1] + eps10;
    q.slab[1] = v[0];
    q.slab[2] = v[1];
    q.slab[BP_ALPHA] = v[0] / den;
    q.slab[BP_ITERS] = z.iters + 1.0;
    bp_report(batch, host, count, seq);
  }
}

#define BSEGS_CHECK()                                                                    \
  BP_CHECK_BATCH();                                                                      \
  CCQP_REQUIRE(dev_segs != nullptr, "NULL segment descriptors");                         \
  CCQP_REQUIRE(max_nwork >= 1 && max_nnorm >= 0 && max_nwork < 2147483647, "bad work-list bounds")

// Start-up of all solves (solvers.py:621-636): g0 = A x0 + b, residual, A g0, alpha0.  Expects x[0] staged, slabs and
// per-problem counters zero, every problem's workspace zero.
extern "C" int ccqp_bsegs_start_f64(ccqp_ctx* ctx, const ccqp_pprob* dev_probs, const ccqp_psegs* dev_segs, int count,
                                    int64_t n, int64_t max_nwork, int64_t max_nnorm, double gd, int flags,
                                    uint32_t* dev_batch, void* stream) {
  BSEGS_CHECK();
  hipStream_t s = (hipStream_t)stream;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  ccqp_lockstep_begin(ctx, dev_batch, s);
  int rc = ccqp_bp_gemv(ctx, dev_probs, count, n, 0, 0, flags, s);
  if (rc) return rc;
  const int const_step = (flags & BP_FLAG_CONST_STEP) != 0;
  if (max_nnorm > 0)
    hipLaunchKernelGGL(bsegs_norm_kernel, dim3((unsigned)max_nnorm, count), dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs,
                       1, 0, gd);
  hipLaunchKernelGGL(bsegs_resid_kernel, dim3((unsigned)max_nwork, count), dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, 0,
                     gd);
  hipLaunchKernelGGL(bsegs_tail_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, 0, 0, 0.0, n,
                     chunk, nblk, dev_batch, ctx->host_slots_dev, (unsigned)count, 0.0, const_step, 1);
  if (!const_step) {
    rc = ccqp_bp_gemv(ctx, dev_probs, count, n, 0, 1, flags, s);
    if (rc) return rc;
    ccqp_bp_init_alpha(dev_probs, count, n, chunk, nblk, s);
  }
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// Iteration `iteration` >= 1 of all live solves; with_tail == 0: the matvec budget ends here (solvers.py:643-644).
extern "C" int ccqp_bsegs_iterate_f64(ccqp_ctx* ctx, const ccqp_pprob* dev_probs, const ccqp_psegs* dev_segs,
                                      int count, int64_t n, int64_t iteration, int64_t max_nwork, int64_t max_nnorm,
                                      double gd, double eps10, int flags, int with_tail, uint32_t* dev_batch,
                                      void* stream) {
  BSEGS_CHECK();
  CCQP_REQUIRE(iteration >= 1, "iterations count from 1");
  hipStream_t s = (hipStream_t)stream;
  const int inew = (int)(iteration % 3), icur = (int)((iteration - 1) % 3);
  const dim3 gw((unsigned)max_nwork, count), gn((unsigned)(max_nnorm > 0 ? max_nnorm : 1), count);
  if (max_nnorm > 0)
    hipLaunchKernelGGL(bsegs_norm_kernel, gn, dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, 0, icur, gd);
  hipLaunchKernelGGL(bsegs_step_kernel, gw, dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, inew, icur, 0, gd);
  const int rc = ccqp_bp_gemv(ctx, dev_probs, count, n, inew, 0, flags, s);
  if (rc) return rc;
  if (with_tail) {
    int64_t chunk;
    int nblk;
    ccqp_red_geometry(n, &chunk, &nblk);
    if (max_nnorm > 0)
      hipLaunchKernelGGL(bsegs_norm_kernel, gn, dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, 1, inew, gd);
    hipLaunchKernelGGL(bsegs_resid_kernel, gw, dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, inew, gd);
    if (flags & BP_FLAG_FALLBACK) {
      hipLaunchKernelGGL(bsegs_tailf_a_kernel, dim3(count), dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, eps10, dev_batch);
      hipLaunchKernelGGL(bsegs_keep_best_kernel, dim3(bp_vec_grid(n), count), dim3(CCQP_BLOCK), 0, s, dev_probs, inew, n);
      if (max_nnorm > 0)
        hipLaunchKernelGGL(bsegs_norm_kernel, gn, dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, 2, inew, gd);
      hipLaunchKernelGGL(bsegs_step_kernel, gw, dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, inew, icur, 2, gd);
      hipLaunchKernelGGL(bsegs_tailf_b_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, inew, icur, eps10,
                         n, chunk, nblk, dev_batch, ctx->host_slots_dev, (unsigned)count, (double)iteration);
    } else {
      hipLaunchKernelGGL(bsegs_tail_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, inew, icur,
                         eps10, n, chunk, nblk, dev_batch, ctx->host_slots_dev, (unsigned)count, (double)iteration,
                         (flags & BP_FLAG_CONST_STEP) != 0, 0);
    }
  }
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// =====================================================================================================================
// Lock-step APGD / anti-relaxation APGD (solvers.py:220-343, 393-533) on non-elementwise sets: the state machine of
// ccqp_bapgd_* (ccqp_bpipe.hip) with the two places where the feasible set enters taken from the segmented engine --
//   * the residual of (x_{k+1}, g1) (solvers.py:321-323): work-item sums by the launches below, added in index order
//     by the deciding block;
//   * the trial step x_{k+1} = P(y_k - t g_k) (:283, :310).
// A round = matvec, [norm,] residual, decide, [norm,] step, apply: up to 7 launches for the whole batch.  The slab
// slots are those of ccqp_bpipe.hip (AP_*, ccqp_lockstep.h).

// what = 0: residual operands (x_{k+1}, g1), for problems whose matvec of this round was A x_{k+1} (phase 1);
// what = 1: trial-step operands (y_k, g_k) with t from the slab, for problems whose next action is a trial step.
struct bsegs_apgd_io {
  const double* x;
  const double* y;
  double b;
  double* out;
  bool live;
};

__device__ __forceinline__ bsegs_apgd_io bsegs_apgd_operands(const ccqp_papgd& q, int what, double gd) {
  bsegs_apgd_io io = {nullptr, nullptr, 0.0, nullptr, false};
  const double* s = q.slab;
  if (s[AP_DONE] != 0.0) return io;
  const int it = (int)s[AP_IT];
  if (what == 0) {
    if (s[AP_PHASE] != 1.0) return io;
    io.x = q.X[(it + 1) & 1];
    io.y = q.g1;
    io.b = -gd;
  } else {
    if (s[AP_ACT] != 1.0) return io;
    io.x = q.Y[it & 1];
    io.y = q.gk;
    io.b = -s[AP_T];
    io.out = q.X[(it + 1) & 1];
  }
  io.live = true;
  return io;
}

__global__ __launch_bounds__(CCQP_BLOCK) void bsegs_apgd_norm_kernel(const ccqp_papgd* __restrict__ probs,
                                                                     const ccqp_psegs* __restrict__ segs, int what,
                                                                     double gd) {
  __shared__ double lds4[4];
  const ccqp_psegs& ps = segs[blockIdx.y];
  if ((int)blockIdx.x >= ps.nnorm) return;
  const bsegs_apgd_io io = bsegs_apgd_operands(probs[blockIdx.y], what, gd);
  if (!io.live) return;
  const ccqp_segwork w = ps.normwork[blockIdx.x];
  const double s = segs_norm_item(ps.table, w, 1.0, io.x, io.b, io.y, lds4);
  if (threadIdx.x == 0) ps.workspace[w.pbase + w.chunk] = s;
}

__global__ __launch_bounds__(CCQP_BLOCK) void bsegs_apgd_apply_kernel(const ccqp_papgd* __restrict__ probs,
                                                                      const ccqp_psegs* __restrict__ segs, int what,
                                                                      double gd) {
  __shared__ double lds4[4];
  __shared__ double bcast;
  const ccqp_psegs& ps = segs[blockIdx.y];
  if ((int)blockIdx.x >= ps.nwork) return;
  const ccqp_papgd& q = probs[blockIdx.y];
  const bsegs_apgd_io io = bsegs_apgd_operands(q, what, gd);
  if (!io.live) return;
  const ccqp_segwork w = ps.work[blockIdx.x];
  if (what == 0) {
    const double racc = segs_apply_item<1>(ps.table, w, ps.workspace, 1.0, io.x, io.b, io.y, nullptr, nullptr, q.c, lds4,
                                           &bcast);
    const double s = ccqp_block_sum(racc, lds4);
    if (threadIdx.x == 0) ps.workspace[3 * ps.npartials + blockIdx.x] = s;
  } else {
    (void)segs_apply_item<0>(ps.table, w, ps.workspace, 1.0, io.x, io.b, io.y, nullptr, io.out, 0.0, lds4, &bcast);
  }
}

// Round `round` >= 1 of all live solves (start-up: ccqp_bapgd_start_f64 -- it involves no projection).
extern "C" int ccqp_bsegs_apgd_round_f64(ccqp_ctx* ctx, const ccqp_papgd* dev_probs, const ccqp_psegs* dev_segs,
                                         int count, int64_t n, int64_t round, int64_t max_nwork, int64_t max_nnorm,
                                         double gd, int flags, uint32_t* dev_batch, void* stream) {
  BSEGS_CHECK();
  CCQP_REQUIRE(round >= 1, "rounds count from 1");
  const bool asked = (flags & CCQP_LOCKSTEP_LOOKAHEAD) != 0;
  flags &= 0xFFFF;
  hipStream_t s = (hipStream_t)stream;
  const dim3 gw((unsigned)max_nwork, count), gn((unsigned)(max_nnorm > 0 ? max_nnorm : 1), count);
  const int rc = ccqp_bapgd_launch_gemv(ctx, dev_probs, count, n, 0, flags, asked, s, true);   // (looks ahead like the elementwise form)
  if (rc) return rc;
  if (max_nnorm > 0)
    hipLaunchKernelGGL(bsegs_apgd_norm_kernel, gn, dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, 0, gd);
  hipLaunchKernelGGL(bsegs_apgd_apply_kernel, gw, dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, 0, gd);
  ccqp_bapgd_launch_decide(ctx, dev_probs, dev_segs, count, n, gd, (double)round, (flags & BP_FLAG_CONST_STEP) != 0,
                           dev_batch, s);
  if (max_nnorm > 0)
    hipLaunchKernelGGL(bsegs_apgd_norm_kernel, gn, dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, 1, gd);
  hipLaunchKernelGGL(bsegs_apgd_apply_kernel, gw, dim3(CCQP_BLOCK), 0, s, dev_probs, dev_segs, 1, gd);
  ccqp_bapgd_launch_spec(dev_probs, count, n, s);                      // (look-ahead runs: y_{k+1} of the trial point)
  ccqp_bapgd_launch_apply(dev_probs, count, n, s);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}
