// Batched + pipelined Barzilai-Borwein solves (opt-in: ccqppy_amd.batch.solve_batch_pipelined): MANY small
// independent dense QPs of one size advance in lock-step, three launches per iteration for ALL of them
// (blockIdx.y = problem), with every scalar on the device (ccqp_pipe.hip explains the single-problem form).
// The host-driven batch schedulers spend ~20 us of Python per problem and round; at n <= 4096 that, not the GPU,
// bounds their throughput (2.4-2.8k problems/s for n = 256..1024).  Here the host issues 3 launches per iteration
// whatever the number of problems and polls two words.
//
// Per problem (struct ccqp_pprob, an array of them in device memory): its own scalar slab (slots 0..3 and 48..51),
// its own partial-sum area and arrival counter, a ring of three x / g buffers.  The blocks of a converged problem
// leave at once.  Every problem -- converged or not -- reports once per launch of a reducing kernel to a batch-wide
// arrival counter; the last arrival publishes {problems converged, sequence number} to pinned host memory.
// Bits: the bodies are those of the single-problem kernels (proj_elem_body, gemv_task, residual_body, dot_body,
// bb_tail_body), the reduction geometry is the same function of n and the matvec's bits do not depend on its
// tiling, so every problem gets exactly the result of its own CCQPSolverBBPGD.solve() (tests/test_gpu_bpipe.py).
#include <string.h>
#include "ccqp_lockstep.h"

__global__ void bpipe_begin_kernel(unsigned* batch) {
  if (threadIdx.x == 0) {
    batch[0] = 0u;
    batch[1] = 0u;
  }
}

void ccqp_lockstep_begin(ccqp_ctx* ctx, uint32_t* dev_batch, hipStream_t s) {
  // the mirror is reset from the host side: a poll issued before the first kernel runs must not see the final
  // state of an earlier batch (whose solves have all been read back by then)
  ctx->host_slots[BP_MIRROR] = 0.0;
  ctx->host_slots[BP_MIRROR + 1] = -1.0;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  hipLaunchKernelGGL(bpipe_begin_kernel, dim3(1), dim3(64), 0, s, dev_batch);
}

// x_new = P(x_cur - alpha g_cur), grid (blocks, problems)
__global__ __launch_bounds__(CCQP_BLOCK) void bpipe_step_kernel(const ccqp_pprob* __restrict__ probs, int inew,
                                                                int icur) {
  const ccqp_pprob& q = probs[blockIdx.y];
  if (q.slab[BP_DONE] != 0.0) return;
  const double b = -q.slab[BP_ALPHA];
  proj_elem_body(bp_segk_of(q), 1.0, q.x[icur], b, q.g[icur], q.x[inew], blockIdx.x, gridDim.x);
}

// grid (blocks, problems); the waves of a problem stride over ITS rows, one row per wave at a time.
//   mode 0:  y = A x[i],  g[i] = y + b        mode 1:  y = A g[i]
// The matrices of a batch are re-read every iteration and fit the 256 MiB of MALL for the sizes this path is
// meant for, so the loads are ordinary (cacheable) ones, unlike the single big matvec's.
template <int U, bool ALIGNED, bool NT>
__global__ __launch_bounds__(CCQP_BLOCK) void bpipe_gemv_kernel(const ccqp_pprob* __restrict__ probs, int i, int mode,
                                                                int64_t n) {
  const ccqp_pprob& q = probs[blockIdx.y];
  if (q.slab[BP_DONE] != 0.0) return;
  const int64_t wave = (int64_t)blockIdx.x * (CCQP_BLOCK / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (CCQP_BLOCK / 64);
  if (mode == 0)
    gemv_body<BP_R(U), BP_U(U), ALIGNED, NT>(q.A, q.lda, q.x[i], q.b, q.y, q.g[i], n, n, BP_TASKS(U, n), wave, nwaves);
  else
    gemv_body<BP_R(U), BP_U(U), ALIGNED, NT>(q.A, q.lda, q.g[i], nullptr, q.y, nullptr, n, n, BP_TASKS(U, n), wave, nwaves);
}

// start-up, first reduction: residual^2 of (x[0], g[0]) and g0.g0; the exit test of solvers.py `if res >= tol`
__global__ __launch_bounds__(CCQP_BLOCK) void bpipe_init_res_kernel(const ccqp_pprob* __restrict__ probs, double gd,
                                                                    int64_t chunk, int nblk,
                                                                    unsigned* __restrict__ batch,
                                                                    double* __restrict__ host, unsigned count,
                                                                    double seq, int const_step) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  const ccqp_pprob& q = probs[blockIdx.y];
  const ccqp_segk sg = bp_segk_of(q);
  double v[2];
  v[0] = residual_body(sg, q.x[0], q.g[0], q.c, gd, chunk, blockIdx.x, lds4);
  v[1] = dot_body(q.g[0], q.g[0], sg.dim, chunk, blockIdx.x, lds4);
  if (!bp_meet(q.partials, q.counter, nblk, 2, v, lds4, &is_last)) return;
  if (threadIdx.x == 0) {
    q.slab[0] = v[0];
    q.slab[BP_GG] = v[1];
    q.slab[BP_RES2] = v[0];
    q.slab[BP_ITERS] = 0.0;
    q.slab[BP_RESMIN] = INFINITY;             // (BBPGDf, :767)
    if (!const_step) q.slab[BP_ALPHA] = 0.0;  // (PGD: the caller has put the constant step there)
    if (sqrt(v[0]) >= q.tol) {               // `if res >= tol:` -- a NaN residual skips the loop, as in the reference
      q.slab[BP_DONE] = 0.0;
    } else {
      q.slab[BP_DONE] = 1.0;
      __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bp_report(batch, host, count, seq);
  }
}

// start-up, second reduction: alpha0 = g0.g0 / g0.(A g0)
__global__ __launch_bounds__(CCQP_BLOCK) void bpipe_init_alpha_kernel(const ccqp_pprob* __restrict__ probs,
                                                                      int64_t n, int64_t chunk, int nblk) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  const ccqp_pprob& q = probs[blockIdx.y];
  const bp_words z = bp_enter(q.slab);
  if (z.done != 0.0) return;
  double v[1];
  v[0] = dot_body(q.g[0], q.y, n, chunk, blockIdx.x, lds4);
  if (!bp_meet(q.partials, q.counter, nblk, 1, v, lds4, &is_last)) return;
  if (threadIdx.x == 0) {
    q.slab[1] = v[0];
    q.slab[BP_ALPHA] = z.gg / v[0];
  }
}

// ---- BBPGDf (solvers.py:696-843): between the residual and the BB dots the reference keeps the best iterate so far
// and, when the step length has collapsed, restarts from it -- so the tail is two reductions with an elementwise
// update folded into the second.  Slab: 4 smallest residual so far, 5 "this iterate is the best", 6 "fall back".
// part B, elementwise + reduction: xmin / gmin = x_new / g_new where improved; x_new = P(xmin - gd gmin) where the
// fallback is due (:799-800, g_new is NOT recomputed); then dx.dx, dx.dg with the possibly replaced x_new
__device__ __forceinline__ void bbf_tail_b_body(const ccqp_pprob& q, const ccqp_segk& sg, int inew, int icur,
                                                bool improved, bool fallback, double gd, int64_t chunk, int bx,
                                                double* lds4, double* dxdx, double* dxdg) {
  double* __restrict__ xn = q.x[inew];
  const double* __restrict__ gn = q.g[inew];
  const double* __restrict__ xo = q.x[icur];
  const double* __restrict__ go = q.g[icur];
  const double mgd = -gd;
  CHUNK_RANGE(sg.dim)
  double a1 = 0.0, a2 = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
    double xi = xn[i];
    const double gi = gn[i];
    if (improved) {
      q.xmin[i] = xi;
      q.gmin[i] = gi;
    }
    if (fallback) {
      const double t1 = 1.0 * q.xmin[i];
      const double t2 = mgd * q.gmin[i];
      const double u = t1 + t2;
      const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
      const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
      xi = ccqp_proj_elem(sg.kind, sg.flags, u, lb, ub);
      xn[i] = xi;
    }
    const double dx = xi - xo[i];
    const double dg = gi - go[i];
    a1 = fma(dx, dx, a1);
    a2 = fma(dx, dg, a2);
  }
  *dxdx = ccqp_block_sum(a1, lds4);
  *dxdg = ccqp_block_sum(a2, lds4);
}

// (bbf_decide_a -- part A's decision -- lives in ccqp_lockstep.h: shared with ccqp_bsegs.hip)
__global__ __launch_bounds__(CCQP_BLOCK) void bpipe_tailf_a_kernel(const ccqp_pprob* __restrict__ probs, int inew,
                                                                   double gd, double eps10, int64_t chunk, int nblk,
                                                                   unsigned* __restrict__ batch) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  const ccqp_pprob& q = probs[blockIdx.y];
  const bp_words z = bp_enter(q.slab);
  if (z.done != 0.0) return;
  double v[1];
  v[0] = residual_body(bp_segk_of(q), q.x[inew], q.g[inew], q.c, gd, chunk, blockIdx.x, lds4);
  if (!bp_meet(q.partials, q.counter, nblk, 1, v, lds4, &is_last)) return;
  if (threadIdx.x == 0 && bbf_decide_a(q.slab, z, v[0], q.tol, eps10)) {
    q.slab[BP_ITERS] = z.iters + 1.0;
    q.slab[BP_DONE] = 1.0;
    __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void bpipe_tailf_b_kernel(const ccqp_pprob* __restrict__ probs, int inew,
                                                                   int icur, double gd, double eps10, int64_t chunk,
                                                                   int nblk, unsigned* __restrict__ batch,
                                                                   double* __restrict__ host, unsigned count,
                                                                   double seq, int step_next) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  __shared__ double alpha_s;
  __shared__ int part_a[3];                 // (step_next: part A's decision -- improved, fall back, converged)
  const ccqp_pprob& q = probs[blockIdx.y];
  const bp_words z = bp_enter(q.slab);
  if (z.done != 0.0) {                      // converged (now or earlier): only the roll-call
    if (blockIdx.x == 0 && threadIdx.x == 0) bp_report(batch, host, count, seq);
    return;
  }
  bool improved = z.improved != 0.0, fallback = z.fallback != 0.0;
  if (step_next) {
    // one-chunk problems: part A (bpipe_tailf_a_kernel: the residual, the exit test, "best so far", "fall back") runs HERE,
    // in the problem's only block, instead of in a launch of its own -- the same body and decision, the flags handed on
    // through LDS instead of the slab
    const double r = residual_body(bp_segk_of(q), q.x[inew], q.g[inew], q.c, gd, chunk, 0, lds4);
    if (threadIdx.x == 0) {
      const bool conv = bbf_decide_a(q.slab, z, r, q.tol, eps10);
      if (conv) {
        q.slab[BP_ITERS] = z.iters + 1.0;
        q.slab[BP_DONE] = 1.0;
        __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bp_report(batch, host, count, seq);
      }
      part_a[0] = sqrt(r) < z.resmin;       // (bbf_decide_a: :793-796)
      part_a[1] = z.alpha < eps10;          // (:799)
      part_a[2] = conv;
    }
    __syncthreads();
    if (part_a[2]) return;
    improved = part_a[0] != 0;
    fallback = part_a[1] != 0;
  }
  double v[2];
  bbf_tail_b_body(q, bp_segk_of(q), inew, icur, improved, fallback, gd, chunk,
                  blockIdx.x, lds4, &v[0], &v[1]);
  if (!bp_meet(q.partials, q.counter, nblk, 2, v, lds4, &is_last)) return;
  if (threadIdx.x == 0) {
    const double den = v[1] + eps10;
    const double alpha = v[0] / den;
    q.slab[1] = v[0];
    q.slab[2] = v[1];
    q.slab[BP_ALPHA] = alpha;
    q.slab[BP_ITERS] = z.iters + 1.0;
    alpha_s = alpha;
    if (!step_next) bp_report(batch, host, count, seq);
  }
  if (step_next) {
    // one-chunk problems (this block is the problem's only one): the step of the NEXT iteration, x = P(x_new - alpha g_new)
    // (bpipe_step_kernel), taken here -- an iteration is then matvec + tail, not step + matvec + tail
    __syncthreads();
    proj_elem_body(bp_segk_of(q), 1.0, q.x[inew], -alpha_s, q.g[inew], q.x[(inew + 1) % 3], 0, 1);
    if (threadIdx.x == 0) bp_report(batch, host, count, seq);
  }
}

// residual^2, dx.dx, dx.dg per problem; the problem's last block derives alpha and the exit test.  grid (nblk, problems)
__global__ __launch_bounds__(CCQP_BLOCK) void bpipe_tail_kernel(const ccqp_pprob* __restrict__ probs, int inew,
                                                                int icur, double gd, double eps10, int64_t chunk,
                                                                int nblk, unsigned* __restrict__ batch,
                                                                double* __restrict__ host, unsigned count,
                                                                double seq, int const_step, int step_next) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  __shared__ double alpha_s;                // (step_next: the step length of the next iteration; NaN = converged, no step)
  const ccqp_pprob& q = probs[blockIdx.y];
  const bp_words z = bp_enter(q.slab);
  if (z.done != 0.0) {                      // converged earlier: only the roll-call
    if (blockIdx.x == 0 && threadIdx.x == 0) bp_report(batch, host, count, seq);
    return;
  }
  double v[3];
  bb_tail_body(bp_segk_of(q), q.x[inew], q.x[icur], q.g[inew], q.g[icur], q.c, gd, chunk, blockIdx.x, lds4, &v[0], &v[1],
               &v[2]);
  if (!bp_meet(q.partials, q.counter, nblk, 3, v, lds4, &is_last)) return;
  if (threadIdx.x == 0) {
    // the host loop's scalar step (solvers.py:647-656): res = sqrt(res2); alpha = dx.dx / (dx.dg + 10 eps)
    const double res = sqrt(v[0]);
    const double den = v[2] + eps10;
    const double alpha = const_step ? z.alpha : v[1] / den;
    q.slab[0] = v[0];
    q.slab[1] = v[1];
    q.slab[2] = v[2];
    q.slab[BP_RES2] = v[0];
    if (!const_step) q.slab[BP_ALPHA] = alpha;
    q.slab[BP_ITERS] = z.iters + 1.0;
    const bool conv = res < q.tol;
    if (conv) {
      q.slab[BP_DONE] = 1.0;
      __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    alpha_s = alpha;
    is_last = conv ? 2 : 1;                 // (2: converged -- no step)
    if (!step_next) bp_report(batch, host, count, seq);
  }
  if (step_next) {
    // one-chunk problems (this block is the problem's only one): the step of the NEXT iteration, x = P(x_new - alpha g_new)
    // (bpipe_step_kernel), taken here -- an iteration is then matvec + tail, not step + matvec + tail
    __syncthreads();
    if (is_last != 2)
      proj_elem_body(bp_segk_of(q), 1.0, q.x[inew], -alpha_s, q.g[inew], q.x[(inew + 1) % 3], 0, 1);
    if (threadIdx.x == 0) bp_report(batch, host, count, seq);
  }
}

// (also used by the segmented lock-step form, ccqp_bsegs.hip)
// (the persistent LDS-DMA form, bp_gemv_pers_kernel of ccqp_lockstep.h: the operands bpipe_gemv_kernel would take)
struct bpipe_resolve {
  const ccqp_pprob* probs;
  int i, mode;
  __device__ bool operator()(int p, gemv_lds_desc* d) const {
    const ccqp_pprob& q = probs[p];
    if (q.slab[BP_DONE] != 0.0) return false;
    d->A = q.A;
    d->lda = q.lda;
    d->y = q.y;
    d->x = mode == 0 ? q.x[i] : q.g[i];
    d->b = mode == 0 ? q.b : nullptr;
    d->g = mode == 0 ? q.g[i] : nullptr;
    return true;
  }
};

int ccqp_bp_gemv(ccqp_ctx* ctx, const ccqp_pprob* probs, int count, int64_t n, int i, int mode, int flags, hipStream_t s) {
  const bp_mv_tuning t = bp_mv_tune(count, n, flags);
  if (bp_pers_default(ctx, t, flags, count, n)) {
    const bpipe_resolve rsv = {probs, i, mode};
    return bp_launch_gemv_pers<4, 4>(ctx, rsv, count, n, 4, 1, s);
  }
  BP_MV_LAUNCH(bpipe_gemv_kernel, probs, i, mode, n);
  return CCQP_OK;
}

void ccqp_bp_init_alpha(const ccqp_pprob* probs, int count, int64_t n, int64_t chunk, int nblk, hipStream_t s) {
  hipLaunchKernelGGL(bpipe_init_alpha_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, probs, n, chunk, nblk);
}

static int bp_gemv(ccqp_ctx* ctx, const ccqp_pprob* probs, int count, int64_t n, int i, int mode, int flags, hipStream_t s) {
  return ccqp_bp_gemv(ctx, probs, count, n, i, mode, flags, s);
}

// Start-up of all solves (solvers.py:626-636): g0 = A x0 + b, residual, A g0, alpha0.  Expects x[0] staged, the
// per-problem counters zero; sequence number 0 is published when every problem has its residual.
extern "C" int ccqp_bpipe_start_f64(ccqp_ctx* ctx, const ccqp_pprob* dev_probs, int count, int64_t n, double gd,
                                    int flags, uint32_t* dev_batch, void* stream) {
  BP_CHECK_BATCH();
  hipStream_t s = (hipStream_t)stream;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  ccqp_lockstep_begin(ctx, dev_batch, s);
  // (DONE is read by the matvec: the caller zeroes the slabs before this call)
  int rc = bp_gemv(ctx, dev_probs, count, n, 0, 0, flags, s);
  if (rc) return rc;
  const int const_step = (flags & BP_FLAG_CONST_STEP) != 0;
  hipLaunchKernelGGL(bpipe_init_res_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, gd, chunk, nblk,
                     dev_batch, ctx->host_slots_dev, (unsigned)count, 0.0, const_step);
  if (!const_step) {                          // PGD (solvers.py:94-170) has no A.g0 / alpha0
    rc = bp_gemv(ctx, dev_probs, count, n, 0, 1, flags, s);
    if (rc) return rc;
    hipLaunchKernelGGL(bpipe_init_alpha_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, n, chunk, nblk);
  }
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// Iteration `iteration` >= 1 of all live solves: x[new] = P(x[cur] - alpha g[cur]); g[new] = A x[new] + b; and,
// unless `with_tail` is 0 (the matvec budget ends here, solvers.py:643-644), the fused tail.
extern "C" int ccqp_bpipe_iterate_f64(ccqp_ctx* ctx, const ccqp_pprob* dev_probs, int count, int64_t n,
                                      int64_t iteration, double gd, double eps10, int flags, int with_tail,
                                      uint32_t* dev_batch, void* stream) {
  BP_CHECK_BATCH();
  CCQP_REQUIRE(iteration >= 1, "iterations count from 1");
  hipStream_t s = (hipStream_t)stream;
  const int inew = (int)(iteration % 3), icur = (int)((iteration - 1) % 3);
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  // one-chunk problems (n <= 2048): the tail of an iteration takes the NEXT iteration's step with it (its only block has
  // the new step length at hand) -- from the second iteration on a round is two launches, matvec + tail, instead of three
  // (context option BPIPE_STEP_IN_TAIL = 1: off, the A/B switch of the tools)
  const int step_in_tail = nblk == 1 && !ccqp_opt(ctx, CCQP_OPT_BPIPE_STEP_IN_TAIL);
  if (!(step_in_tail && iteration >= 2))
    hipLaunchKernelGGL(bpipe_step_kernel, dim3(bp_vec_grid(n), count), dim3(CCQP_BLOCK), 0, s, dev_probs, inew, icur);
  const int rc = bp_gemv(ctx, dev_probs, count, n, inew, 0, flags, s);
  if (rc) return rc;
  if (with_tail) {
    if (flags & BP_FLAG_FALLBACK) {
      if (!step_in_tail)                    // (one-chunk problems: part A runs inside the launch below)
        hipLaunchKernelGGL(bpipe_tailf_a_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, inew, gd, eps10,
                           chunk, nblk, dev_batch);
      hipLaunchKernelGGL(bpipe_tailf_b_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, inew, icur, gd,
                         eps10, chunk, nblk, dev_batch, ctx->host_slots_dev, (unsigned)count, (double)iteration,
                         step_in_tail);
    } else {
      hipLaunchKernelGGL(bpipe_tail_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, inew, icur, gd, eps10,
                         chunk, nblk, dev_batch, ctx->host_slots_dev, (unsigned)count, (double)iteration,
                         (flags & BP_FLAG_CONST_STEP) != 0, step_in_tail);
    }
  }
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// Non-blocking look at the host mirror: out = {problems converged, last sequence number all problems finished}.
extern "C" int ccqp_bpipe_poll(ccqp_ctx* ctx, double* out) {
  CCQP_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
  const double v = *(volatile double*)(ctx->host_slots + BP_MIRROR + 1);      // (one word: bp_report)
  if (v < 0.0) {
    out[0] = 0.0;
    out[1] = -1.0;
  } else {
    const double seq = floor(v / BP_MIRROR_SEQ_UNIT);
    out[0] = v - seq * BP_MIRROR_SEQ_UNIT;
    out[1] = seq;
  }
  return CCQP_OK;
}

// Name of the kernel the batched matvec of the lock-step forms (ccqp_bpipe_* / ccqp_bapgd_* / ccqp_bmprgp_* / ccqp_bspg_*
// and their segmented forms) runs as for `count` problems of size n under `flags` (bit 0 = aligned operands, the rest
// the tuning word): "bp_gemv_pers_kernel" (one persistent LDS-DMA launch for the whole batch) or "batched VALU kernel".
// Bit 16 of `flags` asks for the runs that share passes between products (ccqp_bapgd_* / ccqp_bmprgp_* with their
// look-ahead on): those take the persistent form from n = 1024 on.
// (bit 18 beside bit 16: the sharing run is MPRGP -- it shares passes from a smaller n on than APGD, bp_lookahead_min_n)
extern "C" int ccqp_bpipe_matvec_form(ccqp_ctx* ctx, int count, int64_t n, int flags, char* name, int buflen) {
  CCQP_REQUIRE(name != nullptr && buflen > 0 && count >= 1 && n >= 1, "bad argument");
  const bool shares = (flags >> 16) & 1, mprgp = (flags >> 18) & 1;
  flags &= 0xFFFF;
  const bp_mv_tuning t = bp_mv_tune(count, n, flags);
  const bool lds_tuned = ((flags >> 1) & 3) == 3 && ((flags >> 4) & 7) >= 4;          // (MPRGP's explicit ring codes)
  const char* k = (shares ? bp_pers_sharing(ctx, t, flags, count, n, mprgp) : bp_pers_default(ctx, t, flags, count, n)) ||
                          (lds_tuned && t.aligned && n % 2 == 0 && n >= 128)
                      ? "bp_gemv_pers_kernel" : "batched VALU kernel";
  strncpy(name, k, (size_t)buflen - 1);
  name[buflen - 1] = 0;
  return CCQP_OK;
}

// ring shape of the three-vector persistent launch for `count` problems per launch: <rows per wave><slots><waves>
extern "C" int ccqp_bmv3_shape(ccqp_ctx* ctx, int count) { return bp_gemv3_shape(ctx, count); }

// doubles of partial-sum area one problem needs for vectors of length n
extern "C" int64_t ccqp_bpipe_partials_doubles(int64_t n) {
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n > 0 ? n : 1, &chunk, &nblk);
  return 3 * (int64_t)nblk;
}

// =====================================================================================================================
// Lock-step APGD (solvers.py:220-343): the same idea for the accelerated method, whose iteration is a small state
// machine -- one matvec of y_k, then one matvec per backtracking trial of x_{k+1}.  A ROUND is three launches for
// the whole batch and advances every live problem by exactly one matvec:
//   matvec   phase 0:  A y_k -> (Ayk, gk)            phase 1:  A x_{k+1} -> (Ax1, g1)
//   decide   phase 0:  nothing to reduce; next action = trial step, phase 1
//            phase 1:  the 6 (first trial: 8) dot products and the residual of solvers.py:296-301,321-323 in one
//                      pass; the problem's last block evaluates the sufficient-decrease test and either accepts
//                      (momentum scalars, exit test, L *= 0.9, phase 0) or rejects (L *= 2, another trial)
//   apply    action 1: x_{k+1} = P(y_k - t g_k)       action 2: y_{k+1} = (1 + beta) x_{k+1} - beta x_k
// x and y live in two buffers each, addressed by the parity of the accepted-iteration count `it`, so "swapping the
// iterates" is the increment of `it` by the deciding thread.  State changes are made only by the LAST block of a
// problem to arrive in `decide`, i.e. after every block of that launch has read the state.
// The scalar arithmetic is the host loop's, operation for operation (fp contraction is off in this library).
// (slab slots AP_*: ccqp_lockstep.h)

// d = x0 - ones (solvers.py:262) into the spare x buffer
__device__ __forceinline__ void bapgd_init_d_body(const ccqp_papgd& q, int64_t n, int bx, int nbx) {
  const double* __restrict__ x = q.X[0];
  double* __restrict__ d = q.X[1];
  BODY_LOOP(i, n) {
    const double t1 = 1.0 * x[i];
    const double t2 = -1.0 * 1.0;
    d[i] = t1 + t2;
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void bapgd_init_d_kernel(const ccqp_papgd* __restrict__ probs, int64_t n) {
  bapgd_init_d_body(probs[blockIdx.y], n, blockIdx.x, gridDim.x);
}

template <int U, bool ALIGNED, bool NT>
__device__ __forceinline__ void bapgd_gemv_body(const ccqp_papgd& q, int init, int64_t n, int64_t wave,
                                                int64_t nwaves) {
  if (init) {
    gemv_body<BP_R(U), BP_U(U), ALIGNED, NT>(q.A, q.lda, q.X[1], nullptr, q.Ax1, nullptr, n, n, BP_TASKS(U, n), wave, nwaves);
    return;
  }
  if (q.slab[AP_DONE] != 0.0) return;
  const int it = (int)q.slab[AP_IT];
  if (q.slab[AP_PHASE] == 0.0)
    gemv_body<BP_R(U), BP_U(U), ALIGNED, NT>(q.A, q.lda, q.Y[it & 1], q.b, q.Ayk, q.gk, n, n, BP_TASKS(U, n), wave, nwaves);
  else
    gemv_body<BP_R(U), BP_U(U), ALIGNED, NT>(q.A, q.lda, q.X[(it + 1) & 1], q.b, q.Ax1, q.g1, n, n, BP_TASKS(U, n), wave, nwaves);
}

template <int U, bool ALIGNED, bool NT>
__global__ __launch_bounds__(CCQP_BLOCK) void bapgd_gemv_kernel(const ccqp_papgd* __restrict__ probs, int init,
                                                                int64_t n) {
  bapgd_gemv_body<U, ALIGNED, NT>(probs[blockIdx.y], init, n, (int64_t)blockIdx.x * (CCQP_BLOCK / 64) + (threadIdx.x >> 6),
                                  (int64_t)gridDim.x * (CCQP_BLOCK / 64));
}

// L = |A d| / |d|, t = 1 / L (solvers.py:264-268) and the initial state
// momentum weight beta of solvers.py:313 as a function of theta_k alone (the expressions of the accepting branch of
// bapgd_decide_body, so the bits)
__device__ __forceinline__ double bapgd_beta_of(double thk) {
  const double tt = thk * thk;
  const double root = sqrt(4.0 + tt);
  const double m1 = (-thk) * thk;
  const double m2 = thk * root;
  const double thk1 = 0.5 * (m1 + m2);
  const double num = thk * (1.0 - thk);
  return num / (tt + thk1);
}

// `fuse` (elementwise look-ahead runs): the state after the first take-over -- the start's apply launch takes A.y_0 + b
// over and forms the first trial point, so round 1 starts with the trial matvec (see bapgd_decide_body)
__device__ __forceinline__ void bapgd_init_L_body(const ccqp_papgd& q, int64_t n, int64_t chunk, int nblk, int bx,
                                                  double* lds4, int* is_last_p, double look = 0.0, int fuse = 0) {
  const ap_words z = ap_enter(q.slab);         // (the budget word AP_LIMIT, written by the caller before the start)
  double v[2];
  v[0] = dot_body(q.Ax1, q.Ax1, n, chunk, bx, lds4);
  v[1] = dot_body(q.X[1], q.X[1], n, chunk, bx, lds4);
  if (!bp_meet(q.partials, q.counter, nblk, 2, v, lds4, is_last_p)) return;
  if (threadIdx.x == 0) {
    double* s = q.slab;
    const double L = sqrt(v[0]) / sqrt(v[1]);
    s[0] = v[0];
    s[1] = v[1];
    s[AP_L] = L;
    s[AP_T] = 1.0 / L;
    s[AP_THK] = 1.0;
    s[AP_BETA] = 0.0;
    s[AP_PHASE] = 0.0;
    s[AP_ACT] = 0.0;
    s[AP_IT] = 0.0;
    s[AP_FIRST] = 0.0;
    s[AP_DONE] = 0.0;
    s[AP_MV] = 1.0;
    s[AP_RES2] = 0.0;
    s[AP_RESMIN] = INFINITY;
    s[AP_COPYHAT] = 0.0;
    s[AP_SOLHAT] = 1.0;
    const double looks = (look != 0.0 && q.ga != nullptr && q.Aya != nullptr) ? 1.0 : 0.0;   // (opt-in: the two buffers)
    s[AP_LOOK] = looks;
    s[AP_AHEAD] = looks;                       // (a look-ahead run's start multiplies y_0 beside x_0 - 1: bapgd_resolve3_start)
    s[AP_BETAS] = 0.0;
    if (fuse) {
      if (looks == 0.0 || (z.limit > 0.0 && z.limit < 3.0)) {
        s[AP_DONE] = 2.0;                      // no buffers / no room for one trial: left to the host-driven loop
      } else {
        s[AP_PHASE] = 1.0;                     // as after the phase-0 decision of the unfused form (:275-278)
        s[AP_ACT] = 4.0;
        s[AP_FIRST] = 1.0;
        s[AP_MV] = 2.0;
        s[AP_BETAS] = bapgd_beta_of(1.0);
        s[AP_AHEAD] = 0.0;
      }
    }
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void bapgd_init_L_kernel(const ccqp_papgd* __restrict__ probs, int64_t n,
                                                                  int64_t chunk, int nblk, double look = 0.0, int fuse = 0) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  bapgd_init_L_body(probs[blockIdx.y], n, chunk, nblk, blockIdx.x, lds4, &is_last, look, fuse);
}

// A thread's share of the sums a trial round's decision needs, over chunk `bx` of the problem: one pass, the dot products
// and the residual (residual_body's arithmetic, element for element).  v[0..7] are THREAD sums (the caller adds them over
// the workgroup).  `tshift`: threadIdx.x - tshift is the thread's index within the 256 threads that walk the chunk (0 in
// a 256-thread workgroup).  U: elements whose loads are in flight together.
template <int U>
__device__ __forceinline__ void bapgd_trial_sums(const ccqp_papgd& q, const ap_words& z, int bx, int64_t tshift, double gd,
                                                 int64_t chunk, int ar, const ccqp_psegs* ps, int64_t n_segs, double* v) {
  const ccqp_segk sg = bp_segk_of(q);
  const int64_t n = ps ? n_segs : sg.dim;
  const int it = (int)z.it;
  const bool first = z.first != 0.0;
  const double* __restrict__ x1 = q.X[(it + 1) & 1];
  const double* __restrict__ yk = q.Y[it & 1];
  const double* __restrict__ Ayk = q.Ayk;
  const double* __restrict__ Ax1 = q.Ax1;
  const double* __restrict__ gk = q.gk;
  const double* __restrict__ bv = q.b;
  const double* __restrict__ xk = q.X[it & 1];
  const double* __restrict__ g1 = q.g1;
  const double c = q.c;
  CHUNK_RANGE(n)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
  // (all loads of a thread's U elements first -- ccqp_chunk_walk: the same arithmetic in the same order)
  const bool elem = !ps;
  const double* const vecs[10] = {x1, yk, bv, Ax1, gk, elem ? g1 : nullptr, ar ? xk : nullptr, first ? Ayk : nullptr,
                                  elem ? sg.lb : nullptr, elem ? sg.ub : nullptr};
  ccqp_chunk_walk<10, U>(vecs, lo - tshift, hi, [&](int64_t, const double* e) {
    const double xi = e[0], yi = e[1], bi = e[2];
    if (ar) {                                          // gk.(x_{k+1} - x_k), solvers.py:510
      const double dk = xi - e[6];
      a6 = fma(e[4], dk, a6);
    }
    if (first) {
      a0 = fma(yi, e[7], a0);
      a1 = fma(yi, bi, a1);
    }
    a2 = fma(xi, e[3], a2);
    a3 = fma(xi, bi, a3);
    const double d = xi - yi;
    a4 = fma(e[4], d, a4);
    a5 = fma(d, d, a5);
    if (elem) {
      const double t = gd * e[5];
      const double u = xi - t;
      const double lb = sg.lb ? e[8] : sg.lb_s;
      const double ub = sg.ub ? e[9] : sg.ub_s;
      const double pr = ccqp_proj_elem(sg.kind, sg.flags, u, lb, ub);
      const double dd = xi - pr;
      const double w = c * dd;
      a7 = fma(w, w, a7);
    }
  });
  v[0] = a0;
  v[1] = a1;
  v[2] = a2;
  v[3] = a3;
  v[4] = a4;
  v[5] = a5;
  v[6] = a7;
  v[7] = a6;
}

// The decision of a trial round from the problem's sums v[0..7] (one thread): solvers.py:290-316 (+ :501-512).  Writes the
// state words to `s` -- the problem's slab -- computing from the entry snapshot `z` alone.  `report`: also the batch's
// roll-call (LOCKSTEP).
// `fuse` (elementwise look-ahead runs, LOCKSTEP only): an ACCEPTING round also takes the decision of the phase-0 round
// that would follow -- the product A.y_{k+1} + b is already there (this round's launch computed it), so there is nothing
// to wait for: mv advances by two (the trial's matvec :290 and the product taken over where the reference computes it,
// :275-278), the state becomes "first trial of iteration k + 1" and the apply launch (ACT 4 / 5) takes the product
// over, forms the next trial point and its y.  One round per accepted iteration instead of two; a problem is never in
// phase 0 at a round's start.  The device enforces the budget itself (AP_LIMIT): a problem whose NEXT matvec would
// pass max_mv - 1 stops (AP_DONE = 2) and is left to the host-driven loop, exactly where the unfused run's round cap
// leaves it.
template <bool LOCKSTEP>
__device__ __forceinline__ void bapgd_decide_scalars(const ccqp_papgd& q, double* s, const ap_words& z, const double* v,
                                                     int ar, int fuse, bool report, unsigned* batch, double* host,
                                                     unsigned count, double seq) {
  const int it = (int)z.it;
  const bool first = z.first != 0.0;
  double r1, r2;
  if (first) {
    r1 = v[0] * 0.5;
    r2 = v[1];
    s[AP_R1] = r1;
    s[AP_R2] = r2;
    s[0] = v[0];
    s[1] = v[1];
  } else {
    r1 = z.r1;
    r2 = z.r2;
  }
  for (int k = 2; k < 6; ++k) s[k] = v[k];
  s[6] = v[7];
  s[7] = v[6];
  double L = z.L;
  const double l1 = v[2] * 0.5;
  const double l2 = v[3];
  const double r3 = v[4];
  const double hL = 0.5 * L;
  const double r4 = hL * v[5];
  const double lhs = l1 + l2;
  const double rhs = ((r1 + r2) + r3) + r4;
  s[AP_MV] = z.mv + 1.0;
  if (lhs <= rhs) {                                    // solvers.py:302
    const double thk = z.thk;
    const double tt = thk * thk;
    const double root = sqrt(4.0 + tt);
    const double m1 = (-thk) * thk;
    const double m2 = thk * root;
    const double thk1 = 0.5 * (m1 + m2);               // :312
    const double num = thk * (1.0 - thk);
    const double beta = num / (tt + thk1);             // :313
    const double res = sqrt(v[6]);
    s[AP_RES2] = v[6];
    bool improved = false;
    if (ar && res < z.resmin) {                        // :501-503  xhat = copy(x_{k+1})
      s[AP_RESMIN] = res;
      improved = true;
    }
    if (res < q.tol) {
      s[AP_DONE] = 1.0;
      if (improved) s[AP_SOLHAT] = 0.0;                // the solution is x_{k+1} itself; else the xhat kept so far
      if (LOCKSTEP) __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const bool restart = ar && v[7] > 0.0;           // :510-512  y_{k+1} = copy(x_{k+1}), theta = 1
      const double thnew = restart ? 1.0 : thk1;
      L = L * 0.9;
      s[AP_L] = L;
      s[AP_T] = 1.0 / L;
      s[AP_THK] = thnew;
      s[AP_BETA] = beta;
      s[AP_IT] = (double)(it + 1);
      s[AP_COPYHAT] = improved ? 1.0 : 0.0;
      if (fuse) {
        // the take-over of the round that would follow, decided here: the product is counted (:275-278) and the next
        // round's trial matvec must still fit the budget
        const double mv2 = z.mv + 2.0;
        if (z.limit > 0.0 && mv2 + 1.0 > z.limit) {
          s[AP_DONE] = 2.0;
          if (LOCKSTEP) __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          s[AP_MV] = mv2;
          s[AP_PHASE] = 1.0;
          s[AP_ACT] = restart ? 5.0 : 4.0;
          s[AP_FIRST] = 1.0;
          s[AP_BETAS] = bapgd_beta_of(thnew);
          s[AP_AHEAD] = 0.0;
        }
      } else {
        s[AP_PHASE] = 0.0;
        s[AP_ACT] = restart ? 3.0 : 2.0;
        // this round's launch also computed A.y_{k+1} + b (bapgd_resolve3); after an anti-relaxation restart y_{k+1} is
        // x_{k+1} itself (:510-512) and its product the launch's FIRST one (2: taken from g1 / Ax1)
        if (z.look != 0.0) s[AP_AHEAD] = restart ? 2.0 : 1.0;
      }
    }
  } else {
    L = L * 2.0;
    s[AP_L] = L;
    s[AP_T] = 1.0 / L;
    s[AP_ACT] = 1.0;
    s[AP_FIRST] = 0.0;
    if (fuse && z.limit > 0.0 && (z.mv + 1.0) + 1.0 > z.limit) {   // (the retry's matvec would pass the budget)
      s[AP_DONE] = 2.0;
      if (LOCKSTEP) __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (LOCKSTEP && report) bp_report(batch, host, count, seq);
}

// LOCKSTEP: one of the batch's problems in a lock-step launch (roll-call, converged counter); otherwise the problem
// of a single-launch solve (bx = 0, nblk = 1)
// ps != NULL: the feasible set is not elementwise -- the problem's n comes from the caller, the residual from the
// work-item sums the segmented residual launch (ccqp_bsegs.hip) has left in ps->workspace
// `z`: the caller's entry snapshot of the slab (ap_enter) -- this function only WRITES the slab
// `fuse`: see bapgd_decide_scalars
// UW: elements of the chunk walk whose loads are in flight together (8 = the whole chunk of a thread: the latency-bound
// lock-step launch of a FEW blocks; the single-launch solve -- thousands of workgroups, bound by how many of them a CU
// holds -- takes 2: 237 -> fewer than 128 registers, four workgroups per CU instead of two)
template <bool LOCKSTEP, int UW = 8>
__device__ __forceinline__ void bapgd_decide_body(const ccqp_papgd& q, const ap_words& z, int bx, double gd,
                                                  int64_t chunk, int nblk, unsigned* batch, double* host,
                                                  unsigned count, double seq, double* lds4, int* is_last_p, int ar,
                                                  const ccqp_psegs* ps = nullptr, int64_t n_segs = 0, int fuse = 0) {
  double* s = q.slab;
  double v[8];
  if (z.phase == 0.0) {
    // the matvec of y_k has run; next: the first trial step.  (All blocks meet so that the state changes only
    // after every block has read it.)
    if (!bp_meet(q.partials, q.counter, nblk, 0, v, lds4, is_last_p)) return;
    if (threadIdx.x == 0) {
      s[AP_PHASE] = 1.0;
      s[AP_ACT] = 1.0;
      s[AP_FIRST] = 1.0;
      s[AP_MV] = z.mv + 1.0;
      if (z.look != 0.0) {
        // look-ahead: beta of :313 is a function of theta_k alone -- the apply launch forms y_{k+1} beside the trial
        // point, the next matvec launch multiplies both.  (The expressions of the accepting branch below, so the bits.)
        s[AP_BETAS] = bapgd_beta_of(z.thk);
        s[AP_AHEAD] = 0.0;                                 // (A.y_k + b computed ahead has been taken: bapgd_take_ahead_kernel)
      }
      if (LOCKSTEP) bp_report(batch, host, count, seq);
    }
    return;
  }
  bapgd_trial_sums<UW>(q, z, bx, 0, gd, chunk, ar, ps, n_segs, v);
  ccqp_block_sum_multi<8>(v);
  if (!bp_meet(q.partials, q.counter, nblk, ar ? 8 : 7, v, lds4, is_last_p)) return;
  if (ps) {
    // the problem's last block: residual = the work items' sums in index order (ccqp_segs_residual_sq_f64)
    const double* respart = ps->workspace + 3 * ps->npartials;
    const double r = ps->nwork == 1 ? respart[0] : partial_sum_body(respart, ps->nwork, lds4);
    if (threadIdx.x == 0) v[6] = r;
  }
  if (threadIdx.x == 0) bapgd_decide_scalars<LOCKSTEP>(q, s, z, v, ar, fuse, true, batch, host, count, seq);
}

// (defined below; U: elements of the fused update whose loads are in flight together)
template <int U = 4>
__device__ __forceinline__ void bapgd_apply_body(const ccqp_papgd& q, int bx, int nbx, int64_t n_segs = 0);

__global__ __launch_bounds__(CCQP_BLOCK) void bapgd_decide_kernel(const ccqp_papgd* __restrict__ probs, double gd,
                                                                  int64_t chunk, int nblk,
                                                                  unsigned* __restrict__ batch,
                                                                  double* __restrict__ host, unsigned count,
                                                                  double seq, int ar,
                                                                  const ccqp_psegs* __restrict__ segs = nullptr,
                                                                  int64_t n_segs = 0, int fuse = 0, int update = 0) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  const ccqp_papgd& q = probs[blockIdx.y];
  const ap_words z = ap_enter(q.slab);
  if (z.done != 0.0) {                     // converged (or stopped) earlier: only the roll-call
    if (blockIdx.x == 0 && threadIdx.x == 0) bp_report(batch, host, count, seq);
    return;
  }
  bapgd_decide_body<true>(q, z, blockIdx.x, gd, chunk, nblk, batch, host, count, seq, lds4, &is_last, ar,
                          segs ? segs + blockIdx.y : nullptr, n_segs, fuse);
  if (update) {
    // one-chunk problems (this block is the problem's only one): the update the decision asks for, taken here
    // (bapgd_apply_kernel) -- a fused round is then two launches
    __syncthreads();
    bapgd_apply_body(q, 0, 1);
  }
}

// n_segs > 0: segmented form -- the trial step is taken by the segmented launches (ccqp_bsegs.hip), the vector length
// comes from the caller.  U: elements of the fused update whose loads are in flight together.
template <int U>
__device__ __forceinline__ void bapgd_apply_body(const ccqp_papgd& q, int bx, int nbx, int64_t n_segs) {
  const double* s = q.slab;
  if (s[AP_DONE] != 0.0) return;
  const int it = (int)s[AP_IT];
  const int64_t n = n_segs > 0 ? n_segs : q.seg.dim;
  if (s[AP_ACT] == 1.0) {
    if (n_segs > 0) return;
    const double mt = -s[AP_T];
    proj_elem_body(bp_segk_of(q), 1.0, q.Y[it & 1], mt, q.gk, q.X[(it + 1) & 1], bx, nbx);
    if (s[AP_LOOK] != 0.0) {
      // y_{k+1} = (1 + beta) x_{k+1} - beta x_k (:316) of the trial point just formed, into the OTHER y buffer (y_{k-1},
      // dead): every thread reads back the elements it wrote itself (the same element partition in both bodies)
      const double beta = s[AP_BETAS];
      axpby_body(1.0 + beta, q.X[(it + 1) & 1], -beta, q.X[it & 1], q.Y[(it + 1) & 1], n, bx, nbx);
    }
  } else if (s[AP_ACT] >= 4.0) {
    // fused rounds (bapgd_decide_body): iteration k + 1 accepted AND its first trial due.  `it` is the NEW count:
    // X[it & 1] = x_{k+1}, Y[it & 1] = y_{k+1} as the trial's apply formed it (ACT 5, the anti-relaxation restart
    // :510-512: y_{k+1} = copy(x_{k+1}), and the product to take over is the launch's FIRST one, A.x_{k+1}).
    // Every body below walks the same element partition: a thread reads back only what it wrote itself.
    // ONE pass per element, the loads of four elements in flight (the six separate bodies it replaces -- copy, copy,
    // copy, copy, proj_elem_body, axpby_body -- each waited for its own loads: 5 us per launch at n = 4096): the same
    // operations on the same values in the same order per element.
    const bool restart = s[AP_ACT] == 5.0;
    const bool copyhat = s[AP_COPYHAT] != 0.0;
    const ccqp_segk sg = bp_segk_of(q);
    const double mt = -s[AP_T];
    const double beta = s[AP_BETAS];
    const double pa = 1.0 + beta, pb = -beta;
    const double* __restrict__ xk1 = q.X[it & 1];                  // x_{k+1}
    double* __restrict__ yv = q.Y[it & 1];                         // y_{k+1} (restart: becomes a copy of x_{k+1})
    const double* __restrict__ gsrc = restart ? q.g1 : q.ga;       // A.y_{k+1} + b, computed ahead: taken over
    const double* __restrict__ ysrc = restart ? q.Ax1 : q.Aya;
    double* __restrict__ gk = q.gk;
    double* __restrict__ Ayk = q.Ayk;
    double* __restrict__ x2 = q.X[(it + 1) & 1];                   // x_{k+2} trial (:283)
    double* __restrict__ y2 = q.Y[(it + 1) & 1];                   // y_{k+2} (:316)
    double* __restrict__ xhat = q.xhat;
    for (int64_t i0 = (int64_t)bx * CCQP_BLOCK + threadIdx.x; i0 < n; i0 += (int64_t)U * nbx * CCQP_BLOCK) {
      double ex[U], ey[U], eg[U], ea[U], el[U], eu[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + (int64_t)u * nbx * CCQP_BLOCK;
        const bool ok = i < n;
        ex[u] = ok ? xk1[i] : 0.0;
        ey[u] = (ok && !restart) ? yv[i] : 0.0;
        eg[u] = ok ? gsrc[i] : 0.0;
        ea[u] = ok ? ysrc[i] : 0.0;
        el[u] = (ok && sg.lb) ? sg.lb[i] : sg.lb_s;
        eu[u] = (ok && sg.ub) ? sg.ub[i] : sg.ub_s;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + (int64_t)u * nbx * CCQP_BLOCK;
        if (i >= n) continue;
        const double xv = ex[u];
        if (copyhat) xhat[i] = xv;                                 // xhat = copy(x_{k+1})
        double y1 = ey[u];
        if (restart) {                                             // y_{k+1} = copy(x_{k+1})
          y1 = xv;
          yv[i] = xv;
        }
        gk[i] = eg[u];
        Ayk[i] = ea[u];
        const double t1 = 1.0 * y1;                                // proj_elem_body(1.0, y_{k+1}, -t, g_{k+1})
        const double t2 = mt * eg[u];
        const double w = t1 + t2;
        const double xn = ccqp_proj_elem(sg.kind, sg.flags, w, el[u], eu[u]);
        x2[i] = xn;
        const double s1 = pa * xn;                                 // axpby_body(1 + beta, x_{k+2}, -beta, x_{k+1})
        const double s2 = pb * xv;
        y2[i] = s1 + s2;
      }
    }
  } else {
    if (s[AP_ACT] == 3.0) {                   // anti-relaxation restart: y_{k+1} = copy(x_{k+1})
      copy_body(q.X[it & 1], q.Y[it & 1], n, bx, nbx);
    } else {
      const double beta = s[AP_BETA];
      const double a = 1.0 + beta, b = -beta;
      axpby_body(a, q.X[it & 1], b, q.X[(it + 1) & 1], q.Y[it & 1], n, bx, nbx);
    }
    if (s[AP_COPYHAT] != 0.0) copy_body(q.X[it & 1], q.xhat, n, bx, nbx);   // xhat = copy(x_{k+1})
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void bapgd_apply_kernel(const ccqp_papgd* __restrict__ probs,
                                                                 int64_t n_segs = 0) {
  bapgd_apply_body(probs[blockIdx.y], blockIdx.x, gridDim.x, n_segs);
}

// look-ahead of the SEGMENTED form (ccqp_bsegs.hip takes the trial step itself): y_{k+1} beside the trial point just formed
__global__ __launch_bounds__(CCQP_BLOCK) void bapgd_spec_kernel(const ccqp_papgd* __restrict__ probs, int64_t n) {
  const ccqp_papgd& q = probs[blockIdx.y];
  const double* s = q.slab;
  if (s[AP_DONE] != 0.0 || s[AP_LOOK] == 0.0 || s[AP_ACT] != 1.0) return;
  const int it = (int)s[AP_IT];
  const double beta = s[AP_BETAS];
  axpby_body(1.0 + beta, q.X[(it + 1) & 1], -beta, q.X[it & 1], q.Y[(it + 1) & 1], n, blockIdx.x, gridDim.x);
}

void ccqp_bapgd_launch_spec(const ccqp_papgd* probs, int count, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(bapgd_spec_kernel, dim3(bp_vec_grid(n), count), dim3(CCQP_BLOCK), 0, s, probs, n);
}

// the three launches of an APGD round, for the segmented form's driver (ccqp_bsegs.hip)
void ccqp_bapgd_launch_decide(ccqp_ctx* ctx, const ccqp_papgd* probs, const ccqp_psegs* segs, int count, int64_t n,
                              double gd, double seq, int ar, uint32_t* dev_batch, hipStream_t s) {
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  hipLaunchKernelGGL(bapgd_decide_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, probs, gd, chunk, nblk, dev_batch,
                     ctx->host_slots_dev, (unsigned)count, seq, ar, segs, n);
}

void ccqp_bapgd_launch_apply(const ccqp_papgd* probs, int count, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(bapgd_apply_kernel, dim3(bp_vec_grid(n), count), dim3(CCQP_BLOCK), 0, s, probs, n);
}

// (the persistent LDS-DMA form: the operands bapgd_gemv_body would take)
struct bapgd_resolve {
  const ccqp_papgd* probs;
  int init;
  __device__ bool operator()(int p, gemv_lds_desc* d) const {
    const ccqp_papgd& q = probs[p];
    d->A = q.A;
    d->lda = q.lda;
    if (init) {
      // (init == 2: the start of a run that looks ahead -- only then are `ga` / `Aya` read; a problem WITH the two
      // buffers was multiplied by bapgd_resolve3_start)
      if (init == 2 && q.ga != nullptr && q.Aya != nullptr) return false;
      d->x = q.X[1];
      d->b = nullptr;
      d->y = q.Ax1;
      d->g = nullptr;
      return true;
    }
    if (q.slab[AP_DONE] != 0.0) return false;
    const int it = (int)q.slab[AP_IT];
    const bool trial = q.slab[AP_PHASE] == 0.0;
    if (q.slab[AP_LOOK] != 0.0 && (!trial || q.slab[AP_AHEAD] != 0.0)) return false;   // (bapgd_resolve3 / taken over)
    d->x = trial ? q.Y[it & 1] : q.X[(it + 1) & 1];
    d->b = q.b;
    d->y = trial ? q.Ayk : q.Ax1;
    d->g = trial ? q.gk : q.g1;
    return true;
  }
};

// Look-ahead (plain APGD, elementwise sets, the persistent matvec): the launch that multiplies the trial point x_{k+1} (:290)
// also multiplies y_{k+1} (:275 of the next iteration), which bapgd_apply_body has formed beside it -- one pass over each
// matrix for both.  If the line search rejects the trial point the product is dropped and the retry pairs again.
struct bapgd_resolve3 {
  const ccqp_papgd* probs;
  __device__ bool operator()(int p, gemv_lds3_desc* d) const {
    const ccqp_papgd& q = probs[p];
    const double* s = q.slab;
    if (s[AP_DONE] != 0.0 || s[AP_LOOK] == 0.0 || s[AP_PHASE] == 0.0) return false;
    const int it = (int)s[AP_IT];
    d->A = q.A;
    d->lda = q.lda;
    d->b = q.b;
    d->x[0] = q.X[(it + 1) & 1];
    d->y[0] = q.Ax1;
    d->g[0] = q.g1;
    d->x[1] = q.Y[(it + 1) & 1];
    d->y[1] = q.Aya;
    d->g[1] = q.ga;
    d->x[2] = q.X[(it + 1) & 1];
    d->y[2] = nullptr;
    d->g[2] = nullptr;
    return true;
  }
};

// the start of a look-ahead run: A.(x_0 - 1) (:264) and A.y_0 + b (:275 of the first iteration; y_0 = x_0) in one pass
struct bapgd_resolve3_start {
  const ccqp_papgd* probs;
  __device__ bool operator()(int p, gemv_lds3_desc* d) const {
    const ccqp_papgd& q = probs[p];
    d->A = q.A;
    d->lda = q.lda;
    d->b = q.b;
    d->x[0] = q.X[1];
    d->y[0] = q.Ax1;
    d->g[0] = nullptr;
    d->x[1] = q.Y[0];
    d->y[1] = q.Aya;
    d->g[1] = q.ga;
    d->x[2] = q.X[1];
    d->y[2] = nullptr;
    d->g[2] = nullptr;
    return q.ga != nullptr && q.Aya != nullptr;
  }
};

// the phase-0 round of a problem whose A.y_k + b was computed ahead: take it
__global__ __launch_bounds__(CCQP_BLOCK) void bapgd_take_ahead_kernel(const ccqp_papgd* __restrict__ probs, int64_t n) {
  const ccqp_papgd& q = probs[blockIdx.y];
  const double* s = q.slab;
  if (s[AP_DONE] != 0.0 || s[AP_PHASE] != 0.0 || s[AP_AHEAD] == 0.0) return;
  const double* __restrict__ gfrom = s[AP_AHEAD] == 2.0 ? q.g1 : q.ga;
  const double* __restrict__ yfrom = s[AP_AHEAD] == 2.0 ? q.Ax1 : q.Aya;
  for (int64_t i = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CCQP_BLOCK) {
    q.gk[i] = gfrom[i];
    q.Ayk[i] = yfrom[i];
  }
}

// whether a run looks ahead: the caller asked for it (CCQP_LOCKSTEP_LOOKAHEAD: every problem has its two buffers) and the
// persistent matvec is the run's form (the context's option APGD_LOOKAHEAD = 1: off -- the A/B switch of the tools)
static bool bapgd_looks_ahead(const ccqp_ctx* ctx, int count, int64_t n, int flags, bool asked) {
  return asked && !ccqp_opt(ctx, CCQP_OPT_APGD_LOOKAHEAD) &&
         bp_pers_sharing(ctx, bp_mv_tune(count, n, flags), flags, count, n, false);
}

// The form of a run's FUSED rounds (flags bits 17 + 18 asked for them; every problem has its two buffers):
// 1 = the persistent three-vector launch (bapgd_looks_ahead), 2 = both products of a round from ONE launch of the batched
// VALU body (bp_gemv3_valu_kernel with the descriptors of bapgd_resolve3: batches below the persistent form's sizes whose
// matrices fit the MALL together, bp_valu_sharing) -- the unfused rounds of an iteration are six launches (matvec,
// decision, update, twice), the fused round three --, 0 = the rounds are not fused.
static int bapgd_fused_form(const ccqp_ctx* ctx, int count, int64_t n, int flags, bool asked, bool fuse_asked) {
  if (!asked || !fuse_asked || ccqp_opt(ctx, CCQP_OPT_APGD_LOOKAHEAD)) return 0;
  if (bapgd_looks_ahead(ctx, count, n, flags, asked)) return 1;
  return bp_valu_sharing(ctx, flags, count, n) ? 2 : 0;
}

int ccqp_bapgd_launch_gemv(ccqp_ctx* ctx, const ccqp_papgd* probs, int count, int64_t n, int init, int flags, bool look,
                           hipStream_t s, bool elementwise) {
  const bp_mv_tuning t = bp_mv_tune(count, n, flags);
  const bool ahead = elementwise && bapgd_looks_ahead(ctx, count, n, flags, look);
  if (ahead || bp_pers_default(ctx, t, flags, count, n)) {
    if (init == 0 && ahead) {
      hipLaunchKernelGGL(bapgd_take_ahead_kernel, dim3(bp_vec_grid(n), count), dim3(CCQP_BLOCK), 0, s, probs, n);
      const bapgd_resolve3 rsv3 = {probs};
      // (a three-vector launch that cannot be queued fails the round: the slabs already say "computed ahead")
      const int rc3 = bp_launch_gemv3(ctx, rsv3, count, n, s);
      if (rc3) return rc3;
    }
    const bapgd_resolve rsv = {probs, init};
    return bp_launch_gemv_pers<4, 4>(ctx, rsv, count, n, 4, 1, s);
  }
  BP_MV_LAUNCH(bapgd_gemv_kernel, probs, init, n);
  return CCQP_OK;
}

// Start-up of all solves: d = x0 - 1, A d, L, t.  Expects X[0] = Y[0] = x0, slabs and counters zero.
extern "C" int ccqp_bapgd_start_f64(ccqp_ctx* ctx, const ccqp_papgd* dev_probs, int count, int64_t n, int flags,
                                    uint32_t* dev_batch, void* stream) {
  BP_CHECK_BATCH();
  const bool asked = (flags & CCQP_LOCKSTEP_LOOKAHEAD) != 0, fuse_asked = (flags & CCQP_LOCKSTEP_FUSED_ROUNDS) != 0;
  flags &= 0xFFFF;
  hipStream_t s = (hipStream_t)stream;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  ccqp_lockstep_begin(ctx, dev_batch, s);
  hipLaunchKernelGGL(bapgd_init_d_kernel, dim3(bp_vec_grid(n), count), dim3(CCQP_BLOCK), 0, s, dev_probs, n);
  const int form = bapgd_fused_form(ctx, count, n, flags, asked, fuse_asked);
  const bool look = form == 2 || bapgd_looks_ahead(ctx, count, n, flags, asked);
  const int fuse = form != 0;
  int rc;
  if (form == 2) {
    const bapgd_resolve3_start rsv3 = {dev_probs};
    bp_launch_gemv3_valu(rsv3, count, n, flags, 2, s);
    rc = CCQP_OK;
  } else if (look) {
    // (a problem whose two buffers are NULL takes no part here and is multiplied by the single launch below; with every
    // problem looking ahead that launch finds nothing to do)
    const bapgd_resolve3_start rsv3 = {dev_probs};
    rc = bp_launch_gemv3(ctx, rsv3, count, n, s);
    if (rc) return rc;
    rc = ccqp_bapgd_launch_gemv(ctx, dev_probs, count, n, 2, flags, look, s);   // init = 2: the problems the launch above left out
  } else {
    rc = ccqp_bapgd_launch_gemv(ctx, dev_probs, count, n, 1, flags, look, s);
  }
  if (rc) return rc;
  // (the flag travels in the problems' slabs, set here: AP_LOOK)
  hipLaunchKernelGGL(bapgd_init_L_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, n, chunk, nblk,
                     look ? 1.0 : 0.0, fuse);
  // fused rounds: the first take-over and the first trial point belong to the start (ACT 4 of bapgd_apply_body)
  if (fuse) hipLaunchKernelGGL(bapgd_apply_kernel, dim3(bp_vec_grid(n), count), dim3(CCQP_BLOCK), 0, s, dev_probs);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// form of a run's fused rounds (see the header): 0 = not fused
extern "C" int ccqp_bapgd_fused_form(ccqp_ctx* ctx, int count, int64_t n, int flags) {
  if (count < 1 || n < 1) return 0;
  const bool asked = (flags & CCQP_LOCKSTEP_LOOKAHEAD) != 0, fuse_asked = (flags & CCQP_LOCKSTEP_FUSED_ROUNDS) != 0;
  return bapgd_fused_form(ctx, count, n, flags & 0xFFFF, asked, fuse_asked);
}

// Round `round` >= 1: one matvec for every live problem, the decision, the vector update it asks for.
extern "C" int ccqp_bapgd_round_f64(ccqp_ctx* ctx, const ccqp_papgd* dev_probs, int count, int64_t n, int64_t round,
                                    double gd, int flags, uint32_t* dev_batch, void* stream) {
  BP_CHECK_BATCH();
  CCQP_REQUIRE(round >= 1, "rounds count from 1");
  const bool asked = (flags & CCQP_LOCKSTEP_LOOKAHEAD) != 0;
  const int ar = (flags & BP_FLAG_CONST_STEP) != 0;
  const bool fuse_asked = (flags & CCQP_LOCKSTEP_FUSED_ROUNDS) != 0;
  flags &= 0xFFFF;
  hipStream_t s = (hipStream_t)stream;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  const int form = bapgd_fused_form(ctx, count, n, flags, asked, fuse_asked);
  const int fuse = form != 0;
  int rc;
  if (form == 2) {
    const bapgd_resolve3 rsv3 = {dev_probs};
    bp_launch_gemv3_valu(rsv3, count, n, flags, 2, s);
    rc = CCQP_OK;
  } else if (fuse) {
    // every live problem is at a trial point: ONE launch multiplies x_{k+1} and y_{k+1} of each (nothing to take over
    // here -- the apply launch of the accepting round did --, no single-product launch)
    const bapgd_resolve3 rsv3 = {dev_probs};
    rc = bp_launch_gemv3(ctx, rsv3, count, n, s);
  } else {
    rc = ccqp_bapgd_launch_gemv(ctx, dev_probs, count, n, 0, flags, asked, s);
  }
  if (rc) return rc;
  // (the decision and the update stay two launches: letting the deciding block walk the update alone -- one launch and
  // one kernel boundary fewer -- was measured SLOWER, +4 us per round at n = 4096 and +11 us at n = 8192: one CU moves
  // the ~100 bytes per element at 50-80 GB/s; and ONE block walking the 2 ... 4 chunks of the decision itself, without
  // the inter-workgroup meeting: +5 us at 4096, +16 us at 8192 -- a chunk costs ~5 us wherever it runs;
  // profiles/r04_single_mid.jsonl)
  // (and ONE workgroup per problem doing both, a group of 256 threads per chunk, the chunk sums met in LDS: level with the
  // two launches up to n = 512, 5-7 % behind at 1024 ... 3072, +3 / +14 / +27 us per round at 4096 / 8192 / 16384 --
  // one CU again; round 4, removed)
  // Small one-chunk problems: the update inside the decision launch, whose only block has the new state at hand -- a fused
  // round is then TWO launches.  Whole solves, one problem to 64 (profiles/r04_apgd_update_in_decision.jsonl): +6 ... +12 %
  // at n <= 512, +2 ... +5 % at 1024, -1 ... -5 % at 2048 (one CU moves the update's ~100 bytes per element): taken up to
  // n = 1024 (the context's option APGD_UPDATE_IN_DECISION: 1 never, 2 for every one-chunk problem).  (The first attempt at
  // this -- a kernel of its own with an LDS copy of the slab, before the roll-call was made cheap -- had come out level.)
  const int64_t uopt = ccqp_opt(ctx, CCQP_OPT_APGD_UPDATE_IN_DECISION);
  const int update = nblk == 1 && uopt != 1 && (uopt == 2 || n <= 1024);      // (fused or not)
  hipLaunchKernelGGL(bapgd_decide_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, gd, chunk, nblk,
                     dev_batch, ctx->host_slots_dev, (unsigned)count, (double)round, ar, (const ccqp_psegs*)nullptr,
                     (int64_t)0, fuse, update);
  if (!update) hipLaunchKernelGGL(bapgd_apply_kernel, dim3(bp_vec_grid(n), count), dim3(CCQP_BLOCK), 0, s, dev_probs);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// =====================================================================================================================
// One workgroup per problem, the WHOLE BBPGD solve in one launch (n <= CCQP_RED_CHUNK, i.e. every reduction is one
// block wide).  For matrices of a few hundred rows a lock-step round is bound by its three launches, not by the
// data (which sits in L2 / MALL); here nothing leaves the workgroup: its four waves stride over the rows of the
// matvec, the reductions are the single-block form of the same bodies, phases are separated by workgroup barriers,
// and the scalars live in LDS.  Same bodies, same reduction geometry (one block), same matvec lane map => the same
// bits as the lock-step run and as solve().  The grid is `count` workgroups; the hardware scheduler keeps as many
// resident as fit and starts the next problem where one finishes.
template <bool ALIGNED>
__global__ __launch_bounds__(CCQP_BLOCK) void bsolve_bbpgd_kernel(const ccqp_pprob* __restrict__ probs, double gd,
                                                                  double eps10, int64_t last_iter, int64_t n,
                                                                  int const_step, int fallback) {
  __shared__ double lds4[4];
  __shared__ double sh[4];                    // alpha, res2, flag
  const ccqp_pprob& q = probs[blockIdx.x];
  const ccqp_segk sg = bp_segk_of(q);
  const int64_t wave = threadIdx.x >> 6;
  const int64_t nwaves = CCQP_BLOCK / 64;
  const int64_t chunk = CCQP_RED_CHUNK;       // ccqp_red_geometry(n) for n <= CCQP_RED_CHUNK: one block
  const int64_t ntasks = (n + 3) / 4;         // four rows per wave at a time: loads of four rows in flight
  double* s = q.slab;
  // start-up (solvers.py:625-636)
  gemv_body<4, 1, ALIGNED, false>(q.A, q.lda, q.x[0], q.b, q.y, q.g[0], n, n, ntasks, wave, nwaves);
  __syncthreads();
  {
    const double r = residual_body(sg, q.x[0], q.g[0], q.c, gd, chunk, 0, lds4);
    const double gg = dot_body(q.g[0], q.g[0], n, chunk, 0, lds4);
    if (threadIdx.x == 0) {
      sh[1] = r;
      sh[2] = gg;
      sh[3] = sqrt(r) >= q.tol ? 0.0 : 1.0;   // `if res >= tol:` (a NaN residual skips the loop)
      sh[0] = s[BP_ALPHA];                    // (x0 already meets tol: the step in the slab is stored back as it is)
      s[BP_RESMIN] = INFINITY;                // (BBPGDf, :767)
    }
  }
  __syncthreads();
  int64_t k = 0;
  bool done = sh[3] != 0.0;
  if (!done) {
    if (const_step) {                         // PGD: the caller has put the constant step into the slab
      if (threadIdx.x == 0) sh[0] = s[BP_ALPHA];
    } else {
      gemv_body<4, 1, ALIGNED, false>(q.A, q.lda, q.g[0], nullptr, q.y, nullptr, n, n, ntasks, wave, nwaves);
      __syncthreads();
      const double gAg = dot_body(q.g[0], q.y, n, chunk, 0, lds4);
      if (threadIdx.x == 0) sh[0] = sh[2] / gAg;
    }
    __syncthreads();
    for (;;) {
      ++k;
      const int inew = (int)(k % 3), icur = (int)((k - 1) % 3);
      const double mb = -sh[0];
      proj_elem_body(sg, 1.0, q.x[icur], mb, q.g[icur], q.x[inew], 0, 1);
      __syncthreads();
      gemv_body<4, 1, ALIGNED, false>(q.A, q.lda, q.x[inew], q.b, q.y, q.g[inew], n, n, ntasks, wave, nwaves);
      __syncthreads();
      if (k >= last_iter) break;              // the matvec budget ends before the residual (:643-644)
      if (fallback) {                         // BBPGDf: residual -> best iterate / fall-back -> BB dots
        const double r = residual_body(sg, q.x[inew], q.g[inew], q.c, gd, chunk, 0, lds4);
        if (threadIdx.x == 0) {
          s[BP_ALPHA] = sh[0];
          bp_words z;                         // (one workgroup owns the slab: thread 0 is its only reader and writer)
          z.alpha = sh[0];
          z.resmin = s[BP_RESMIN];
          sh[3] = bbf_decide_a(s, z, r, q.tol, eps10) ? 1.0 : 0.0;
          sh[1] = r;
        }
        __syncthreads();
        if (sh[3] != 0.0) {
          done = true;
          break;
        }
        double s1, s2;
        bbf_tail_b_body(q, sg, inew, icur, s[BP_IMPROVED] != 0.0, s[BP_FALLBACK] != 0.0, gd, chunk, 0, lds4, &s1, &s2);
        if (threadIdx.x == 0) {
          const double den = s2 + eps10;
          s[1] = s1;
          s[2] = s2;
          sh[0] = s1 / den;
        }
        __syncthreads();
        continue;
      }
      double r, s1, s2;
      bb_tail_body(sg, q.x[inew], q.x[icur], q.g[inew], q.g[icur], q.c, gd, chunk, 0, lds4, &r, &s1, &s2);
      if (threadIdx.x == 0) {
        const double den = s2 + eps10;
        s[0] = r;
        s[1] = s1;
        s[2] = s2;
        if (!const_step) sh[0] = s1 / den;
        sh[1] = r;
        sh[3] = sqrt(r) < q.tol ? 1.0 : 0.0;
      }
      __syncthreads();
      if (sh[3] != 0.0) {
        done = true;
        break;
      }
    }
  }
  if (threadIdx.x == 0) {
    // as the lock-step run leaves them: iterations WITH a tail, the last residual, the converged flag
    s[BP_ALPHA] = sh[0];
    s[BP_DONE] = done ? 1.0 : 0.0;
    s[BP_ITERS] = (double)(done || k == 0 ? k : k - 1);
    s[BP_RES2] = sh[1];
  }
}

extern "C" int ccqp_bsolve_bbpgd_f64(ccqp_ctx* ctx, const ccqp_pprob* dev_probs, int count, int64_t n, double gd,
                                     double eps10, int64_t last_iteration, int flags, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && dev_probs != nullptr, "NULL argument");
  CCQP_REQUIRE(count >= 1 && n >= 1 && n <= CCQP_RED_CHUNK, "one workgroup per problem needs n <= 2048");
  CCQP_REQUIRE(last_iteration >= 1, "at least one iteration (solvers.py:638-644)");
  hipStream_t s = (hipStream_t)stream;
  if (flags & 1)
    hipLaunchKernelGGL((bsolve_bbpgd_kernel<true>), dim3(count), dim3(CCQP_BLOCK), 0, s, dev_probs, gd, eps10,
                       last_iteration, n, (flags & BP_FLAG_CONST_STEP) != 0, (flags & BP_FLAG_FALLBACK) != 0);
  else
    hipLaunchKernelGGL((bsolve_bbpgd_kernel<false>), dim3(count), dim3(CCQP_BLOCK), 0, s, dev_probs, gd, eps10,
                       last_iteration, n, (flags & BP_FLAG_CONST_STEP) != 0, (flags & BP_FLAG_FALLBACK) != 0);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// The same for APGD: the three bodies of a lock-step round (matvec by phase, decide, apply) run one after the other
// inside the problem's workgroup, separated by barriers; the state stays in the problem's slab, which only this
// workgroup touches.  `mv_limit`: no round starts once the matvec count has reached it (max_mv - 1: the budget's own
// exit paths are the host-driven loop's).
template <bool ALIGNED>
__global__ __launch_bounds__(CCQP_BLOCK) void bsolve_apgd_kernel(const ccqp_papgd* __restrict__ probs, double gd,
                                                                 double mv_limit, int64_t n, int ar) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  const ccqp_papgd& q = probs[blockIdx.x];
  const int64_t wave = threadIdx.x >> 6;
  const int64_t nwaves = CCQP_BLOCK / 64;
  const int64_t chunk = CCQP_RED_CHUNK;       // ccqp_red_geometry(n) for n <= CCQP_RED_CHUNK: one block
  double* s = q.slab;
  bapgd_init_d_body(q, n, 0, 1);
  __syncthreads();
  bapgd_gemv_body<4, ALIGNED, false>(q, 1, n, wave, nwaves);
  __syncthreads();
  bapgd_init_L_body(q, n, chunk, 1, 0, lds4, &is_last);
  __syncthreads();
  for (;;) {
    const ap_words z = ap_enter(s);            // the round's snapshot (the matvec does not change the slab)
    if (z.done != 0.0 || z.mv >= mv_limit) break;
    bapgd_gemv_body<4, ALIGNED, false>(q, 0, n, wave, nwaves);
    __syncthreads();
    bapgd_decide_body<false, 2>(q, z, 0, gd, chunk, 1, nullptr, nullptr, 0u, 0.0, lds4, &is_last, ar);
    __syncthreads();
    bapgd_apply_body<1>(q, 0, 1);              // (nothing to do for a problem that has just converged)
    __syncthreads();
  }
}

extern "C" int ccqp_bsolve_apgd_f64(ccqp_ctx* ctx, const ccqp_papgd* dev_probs, int count, int64_t n, double gd,
                                    double mv_limit, int flags, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && dev_probs != nullptr, "NULL argument");
  CCQP_REQUIRE(count >= 1 && n >= 1 && n <= CCQP_RED_CHUNK, "one workgroup per problem needs n <= 2048");
  hipStream_t s = (hipStream_t)stream;
  if (flags & 1)
    hipLaunchKernelGGL((bsolve_apgd_kernel<true>), dim3(count), dim3(CCQP_BLOCK), 0, s, dev_probs, gd, mv_limit, n,
                       (flags & BP_FLAG_CONST_STEP) != 0);
  else
    hipLaunchKernelGGL((bsolve_apgd_kernel<false>), dim3(count), dim3(CCQP_BLOCK), 0, s, dev_probs, gd, mv_limit, n,
                       (flags & BP_FLAG_CONST_STEP) != 0);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// Staging of ONE lock-step APGD problem in ONE launch (the device-resident form of a single solve: the ~10 small host-paced
// operations it replaces -- zero the slab and the counters, x0 or zeros into X[0], y_0 = x_0, xhat = ones, the descriptor's
// upload -- cost 140 us of a 1.3 ms solve at n = 4096).  The descriptor travels BY VALUE as a kernel argument and is
// written to `dev_desc`, where the round kernels read it.  words: [0] the problem's arrival counter, [1..2] the batch words.
__global__ __launch_bounds__(CCQP_BLOCK) void bapgd_stage_one_kernel(const ccqp_papgd desc, ccqp_papgd* __restrict__ dev_desc,
                                                                     const double* __restrict__ x0, int64_t n, double limit,
                                                                     uint32_t* __restrict__ words) {
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) *dev_desc = desc;
    if (threadIdx.x < 64) desc.slab[threadIdx.x] = threadIdx.x == AP_LIMIT ? limit : 0.0;
    if (threadIdx.x < 3) words[threadIdx.x] = 0u;
  }
  double* __restrict__ X0 = desc.X[0];
  double* __restrict__ Y0 = desc.Y[0];
  double* __restrict__ xhat = desc.xhat;
  for (int64_t i = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CCQP_BLOCK) {
    const double v = x0 ? x0[i] : 0.0;         // x0 = zeros if None (solvers.py:253-254)
    X0[i] = v;
    Y0[i] = v;                                 // yk = copy(x0) (:257-258)
    if (xhat) xhat[i] = 1.0;                   // anti-relaxation: xhat = ones (:432)
  }
}

extern "C" int ccqp_bapgd_stage_one_f64(ccqp_ctx* ctx, const ccqp_papgd* host_desc, ccqp_papgd* dev_desc, const double* x0,
                                        int64_t n, double limit, uint32_t* words, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && host_desc != nullptr && dev_desc != nullptr && words != nullptr, "NULL argument");
  CCQP_REQUIRE(n >= 1 && host_desc->X[0] != nullptr && host_desc->Y[0] != nullptr && host_desc->slab != nullptr,
               "incomplete descriptor");
  hipLaunchKernelGGL(bapgd_stage_one_kernel, dim3(bp_vec_grid(n)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, *host_desc,
                     dev_desc, x0, n, limit, words);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// The same for ONE problem of the BBPGD-type drivers (ccqp_bpipe_*: PGD / BBPGD / BBPGDf as a batch of one): the
// descriptor by value, the slab zeroed (slot 48 = PGD's constant step), the three words, x0 or zeros into x[0] and -- BBPGDf,
// solvers.py:762-763 -- copies of it into xmin and gmin.
__global__ __launch_bounds__(CCQP_BLOCK) void bpipe_stage_one_kernel(const ccqp_pprob desc, ccqp_pprob* __restrict__ dev_desc,
                                                                     const double* __restrict__ x0, int64_t n, double step,
                                                                     uint32_t* __restrict__ words) {
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) *dev_desc = desc;
    if (threadIdx.x < 64) desc.slab[threadIdx.x] = threadIdx.x == BP_ALPHA ? step : 0.0;
    if (threadIdx.x < 3) words[threadIdx.x] = 0u;
  }
  double* __restrict__ X0 = desc.x[0];
  double* __restrict__ xmin = desc.xmin;
  double* __restrict__ gmin = desc.gmin;
  for (int64_t i = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CCQP_BLOCK) {
    const double v = x0 ? x0[i] : 0.0;
    X0[i] = v;
    if (xmin) xmin[i] = v;
    if (gmin) gmin[i] = v;
  }
}

extern "C" int ccqp_bpipe_stage_one_f64(ccqp_ctx* ctx, const ccqp_pprob* host_desc, ccqp_pprob* dev_desc, const double* x0,
                                        int64_t n, double step, uint32_t* words, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && host_desc != nullptr && dev_desc != nullptr && words != nullptr, "NULL argument");
  CCQP_REQUIRE(n >= 1 && host_desc->x[0] != nullptr && host_desc->slab != nullptr, "incomplete descriptor");
  hipLaunchKernelGGL(bpipe_stage_one_kernel, dim3(bp_vec_grid(n)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, *host_desc,
                     dev_desc, x0, n, step, words);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// =====================================================================================================================
// ONE problem, the WHOLE APGD solve in ONE launch on the whole chip (2048 < n <= 16384: where a pass over the matrix takes
// 27 ... 340 us and the ~35 us of launch gaps and small kernels per lock-step round, or the host's turn-around per
// iteration, are a third to a half of a solve).  One 4-wave workgroup per CU, launched COOPERATIVELY (the runtime
// guarantees that all workgroups are resident together); a round is the fused lock-step round above -- the two-vector
// matvec of the trial point and its momentum point through the per-wave LDS rings (gemv_lds3_multi_body with one matrix),
// the decision in the first nblk workgroups (the same bodies, the same chunk geometry and meeting point, hence the bits
// and counts of the lock-step rounds and of solve()), the elementwise update -- separated by GRID BARRIERS instead of
// kernel boundaries.  Barrier: a monotone arrival counter (dev_batch[0]); one thread per workgroup drains its stores,
// releases at agent scope, takes a ticket and polls with s_sleep; the poll is BOUNDED (~seconds): a workgroup that gives
// up raises dev_batch[1], every other one sees it and leaves too, and the caller falls back to the lock-step rounds --
// every wave reaches an exit whatever happens.
#define APG_SPIN_LIMIT 4000000u
__device__ __forceinline__ bool apg_grid_sync(unsigned* bar, unsigned nwg, unsigned& epoch) {
  __shared__ int ok_s;
  __syncthreads();
  epoch += 1u;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = epoch * nwg;
    bool ok = true;
    for (unsigned spins = 0; __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins) {
      __builtin_amdgcn_s_sleep(2);
      if (spins > APG_SPIN_LIMIT || __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        ok = false;
        break;
      }
    }
    if (!ok) __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

template <int R, int S>
__global__ __launch_bounds__(CCQP_BLOCK) void bsolve_apgd_grid_kernel(const ccqp_papgd* __restrict__ probs, int64_t n,
                                                                      double gd, int ar, int64_t rs, int64_t chunk,
                                                                      int nblk, int64_t max_rounds,
                                                                      unsigned* __restrict__ bar) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char ring[];
  __shared__ double lds4[4];
  __shared__ int is_last;
  constexpr int RING = S * (R + 3) * 1024;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;
  double* res = reinterpret_cast<double*>(ring + wpb * RING) + w * GEMV_LDS_RESCAP;
  gemv_lds3_desc* D = reinterpret_cast<gemv_lds3_desc*>(ring + wpb * RING + wpb * GEMV_LDS_RESCAP * sizeof(double));
  const ccqp_papgd& q = probs[0];
  double* s = q.slab;
  const unsigned nwg = gridDim.x;
  const int bx = blockIdx.x, nbx = gridDim.x;
  const int64_t wave = (int64_t)bx * wpb + w, nwaves = (int64_t)nbx * wpb;
  unsigned epoch = 0;
  // ---- start-up (solvers.py:257-269): d = x0 - 1; A.d and A.y_0 + b in one pass; L, t; the first trial point
  bapgd_init_d_body(q, n, bx, nbx);
  if (!apg_grid_sync(bar, nwg, epoch)) return;
  if (threadIdx.x == 0) {
    const bapgd_resolve3_start r3 = {probs};
    gemv_lds3_desc d;
    (void)r3(0, &d);
    D[0] = d;
  }
  __syncthreads();
  gemv_lds3_multi_body<R, S, true>(D, 1, n, wave, nwaves, ring + w * RING, res, rs);
  if (!apg_grid_sync(bar, nwg, epoch)) return;
  if (bx < nblk) bapgd_init_L_body(q, n, chunk, nblk, bx, lds4, &is_last, 1.0, 1);
  if (!apg_grid_sync(bar, nwg, epoch)) return;
  bapgd_apply_body(q, bx, nbx);
  if (!apg_grid_sync(bar, nwg, epoch)) return;
  // ---- rounds: every workgroup reads the same slab between two barriers, so all of them take the same exit
  for (int64_t round = 0; round < max_rounds; ++round) {
    const ap_words z = ap_enter(s);
    if (z.done != 0.0) break;
    if (threadIdx.x == 0) {
      const bapgd_resolve3 r3 = {probs};
      gemv_lds3_desc d;
      (void)r3(0, &d);
      D[0] = d;
    }
    __syncthreads();
    gemv_lds3_multi_body<R, S, true>(D, 1, n, wave, nwaves, ring + w * RING, res, rs);
    if (!apg_grid_sync(bar, nwg, epoch)) return;
    if (bx < nblk)
      bapgd_decide_body<false>(q, z, bx, gd, chunk, nblk, nullptr, nullptr, 0u, 0.0, lds4, &is_last, ar, nullptr, 0, 1);
    if (!apg_grid_sync(bar, nwg, epoch)) return;
    bapgd_apply_body(q, bx, nbx);              // (nothing to do for a problem that has just converged or stopped)
    if (!apg_grid_sync(bar, nwg, epoch)) return;
  }
#endif
}

// flags: bit 0 (aligned operands: required), bit 3 (the anti-relaxation variant).  The problem carries the two look-ahead
// buffers; slab slot 48 = max_mv - 1 (0: none) as for the fused rounds; dev_batch: two zeroed words.  Results in the slab as
// after ccqp_bapgd_round_f64 (slot 49: 1 converged, 2 stopped at the budget, 0 with dev_batch[1] != 0: the grid barrier
// gave up -- the caller runs the problem some other way).  CCQP_ERR_UNSUPPORTED: no cooperative launch of this shape here.
extern "C" int ccqp_bsolve_apgd_grid_f64(ccqp_ctx* ctx, const ccqp_papgd* dev_probs, int64_t n, double gd, int flags,
                                         int64_t max_rounds, uint32_t* dev_batch, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && dev_probs != nullptr && dev_batch != nullptr, "NULL argument");
  CCQP_REQUIRE((flags & 1) && n >= 128 && n % 2 == 0, "needs aligned operands and an even n >= 128");
  CCQP_REQUIRE(max_rounds >= 1, "max_rounds < 1");
  hipStream_t s = (hipStream_t)stream;
  int coop = 0;
  CCQP_HIP(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, ctx->device));
  if (!coop) {
    ccqp_set_error("%s: the device does not launch cooperatively", __func__);
    return CCQP_ERR_UNSUPPORTED;
  }
  constexpr int R = 4, S = 4, WPB = 4;
  const size_t lds = (size_t)WPB * ((size_t)S * (R + 3) * 1024 + GEMV_LDS_RESCAP * sizeof(double)) + sizeof(gemv_lds3_desc);
  const void* fn = reinterpret_cast<const void*>(bsolve_apgd_grid_kernel<R, S>);
  int rc = ccqp_lds_raise(ctx, fn, lds);
  if (rc) return rc;
  int per_cu = 0;
  CCQP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * WPB, lds));
  if (per_cu < 1) {
    ccqp_set_error("%s: the workgroup does not fit a compute unit", __func__);
    return CCQP_ERR_UNSUPPORTED;
  }
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  CCQP_REQUIRE(nblk <= ctx->num_cu, "more reduction chunks than workgroups");
  ccqp_lockstep_begin(ctx, dev_batch, s);       // (zeroes the two words: barrier counter, give-up flag)
  int ar = (flags & BP_FLAG_CONST_STEP) != 0;
  int64_t rs = gemv_lds_row_stride(ccqp_opt(ctx, CCQP_OPT_GEMV_ROW_STRIDE), n, n, R);
  unsigned* bar = dev_batch;
  void* args[] = {(void*)&dev_probs, (void*)&n, (void*)&gd, (void*)&ar, (void*)&rs, (void*)&chunk, (void*)&nblk,
                  (void*)&max_rounds, (void*)&bar};
  const hipError_t e = hipLaunchCooperativeKernel(fn, dim3((unsigned)ctx->num_cu), dim3(64 * WPB), args, (unsigned)lds, s);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    ccqp_set_error("%s: cooperative launch refused: %s", __func__, hipGetErrorString(e));
    return CCQP_ERR_UNSUPPORTED;
  }
  return CCQP_OK;
}
