// Pipelined Barzilai-Borwein iterations (opt-in: CCQPSolverBBPGD(..., pipeline=True)).
//
// A host-driven BBPGD iteration (solvers.py:636-660) is  P(x - alpha g)  ->  A.x + b  ->  residual and the two BB
// dots  ->  host reads three scalars, computes alpha, tests the residual, launches the next iteration.  At
// n <= 16k the host round trip (a polled read, ~10 us of Python, the launch latency of the next kernel) is 15-50 %
// of the iteration.  Here the scalars never leave the device on the critical path:
//   * the tail kernel's last block derives alpha = dx.dx / (dx.dg + 10 eps) and the exit test
//     sqrt(res2) < tol itself (same IEEE operations as the host's numpy float64 arithmetic), keeps them in the
//     context's scalar slab, and mirrors {iterations finished, converged, res2, alpha} into pinned host memory;
//   * the step / matvec / tail kernels of the NEXT iterations are queued without waiting; each of them first
//     looks at the slab's `converged` word and leaves at once if it is set, so running ahead costs a few empty
//     launches, never a wrong write;
//   * the host only polls the mirror to bound its run-ahead and to learn when to stop queueing.
// The arithmetic, its order and therefore every bit of x, the residual and the iteration count are those of the
// host-driven loop (tests/test_gpu_pipeline.py).
#include "ccqp_bodies.h"

#define PIPE_ALPHA 48   // slab slots of the pipeline state
#define PIPE_DONE 49
#define PIPE_ITERS 50
#define PIPE_RES2 51
#define PIPE_MIRROR (CCQP_NUM_SLOTS + 1)   // host mirror, in the padding behind the slab mirror and the read flag:
                                          // host_slots[65..68] = {iters, done, res2, alpha}

int ccqp_gemv_launch(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, const double* b, double* y,
                     double* g, int64_t rows, int64_t cols, int variant, hipStream_t s, const double* skip);

static inline int pipe_vec_grid(int64_t n) {
  int64_t g = (n + CCQP_BLOCK - 1) / CCQP_BLOCK;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

__global__ void pipe_reset_kernel(double* slots, double* host, double alpha0) {
  if (threadIdx.x == 0) {
    slots[PIPE_ALPHA] = alpha0;
    slots[PIPE_DONE] = 0.0;
    slots[PIPE_ITERS] = 0.0;
    slots[PIPE_RES2] = 0.0;
    host[PIPE_MIRROR + 0] = 0.0;
    host[PIPE_MIRROR + 1] = 0.0;
    host[PIPE_MIRROR + 2] = 0.0;
    host[PIPE_MIRROR + 3] = alpha0;
  }
}

// out = P(x - alpha g), alpha from the slab
__global__ __launch_bounds__(CCQP_BLOCK) void pipe_step_project_kernel(ccqp_segk sg, const double* x, const double* g,
                                                                       double* out, const double* slots) {
  if (slots[PIPE_DONE] != 0.0) return;
  const double b = -slots[PIPE_ALPHA];        // the host path passes -alpha as well
  proj_elem_body(sg, 1.0, x, b, g, out, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(CCQP_BLOCK) void pipe_bb_tail_kernel(ccqp_segk sg, const double* xn, const double* xo,
                                                                  const double* gn, const double* go, double c,
                                                                  double gd, double tol, double eps10, int64_t chunk,
                                                                  double* partials, int nblk, double* slots,
                                                                  double* host) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  // entry snapshot + barrier (ccqp_lockstep.h, the meeting-point invariant): every wave of this block has read the
  // words the problem's last block will change before the block takes its ticket; nothing below reads the slab again
  const double done0 = slots[PIPE_DONE];
  const double iters0 = slots[PIPE_ITERS];
  __syncthreads();
  if (done0 != 0.0) return;                   // (no block of a skipped launch touches the arrival counter)
  double r, s1, s2;
  bb_tail_body(sg, xn, xo, gn, go, c, gd, chunk, blockIdx.x, lds4, &r, &s1, &s2);
  if (nblk > 1) {
    if (threadIdx.x == 0) {
      partials[blockIdx.x] = r;
      partials[nblk + blockIdx.x] = s1;
      partials[2 * nblk + blockIdx.x] = s2;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned* counter = ccqp_red_counters(partials);
      const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      is_last = prev + 1u == (unsigned)nblk;
      if (is_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (!is_last) return;
    double sums[3];
    for (int q = 0; q < 3; ++q) {
      double acc = 0.0;
      for (int i = threadIdx.x; i < nblk; i += CCQP_BLOCK) acc += partials[(int64_t)q * nblk + i];
      sums[q] = ccqp_block_sum(acc, lds4);
    }
    r = sums[0];
    s1 = sums[1];
    s2 = sums[2];
    if (threadIdx.x == 0)
      __hip_atomic_store(ccqp_red_counters(partials), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) {
    // the host loop's scalar step, on the device: res = sqrt(res2); alpha = dx.dx / (dx.dg + 10 eps)
    const double res = sqrt(r);
    const double den = s2 + eps10;
    const double alpha = s1 / den;
    const double iters = iters0 + 1.0;
    const double done = res < tol ? 1.0 : 0.0;
    slots[0] = r;
    slots[1] = s1;
    slots[2] = s2;
    slots[PIPE_RES2] = r;
    slots[PIPE_ALPHA] = alpha;
    slots[PIPE_ITERS] = iters;
    slots[PIPE_DONE] = done;
    host[PIPE_MIRROR + 2] = r;
    host[PIPE_MIRROR + 3] = alpha;
    host[PIPE_MIRROR + 0] = iters;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(host + PIPE_MIRROR + 1, done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

extern "C" int ccqp_pipe_reset(ccqp_ctx* ctx, double alpha0, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  // The mirror is reset from the HOST side as well: ccqp_pipe_poll may run before the reset kernel does, and must
  // not see the final state ("converged") of the previous solve.  (That solve has ended with a blocking read, so
  // none of its kernels is still writing here.)
  ctx->host_slots[PIPE_MIRROR + 0] = 0.0;
  ctx->host_slots[PIPE_MIRROR + 1] = 0.0;
  ctx->host_slots[PIPE_MIRROR + 2] = 0.0;
  ctx->host_slots[PIPE_MIRROR + 3] = alpha0;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  hipLaunchKernelGGL(pipe_reset_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctx->slots, ctx->host_slots_dev,
                     alpha0);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

static int pipe_seg(const ccqp_seg* seg, ccqp_segk* k) {
  CCQP_REQUIRE(seg != nullptr, "seg is NULL");
  CCQP_REQUIRE(seg->kind == CCQP_OP_IDENTITY || seg->kind == CCQP_OP_LOWER || seg->kind == CCQP_OP_UPPER ||
                   seg->kind == CCQP_OP_BOX,
               "the pipelined iteration needs an elementwise feasible set");
  CCQP_REQUIRE(seg->offset == 0 && seg->dim >= 0, "the pipelined iteration works on whole vectors");
  k->kind = seg->kind;
  k->flags = seg->flags;
  k->dim = seg->dim;
  k->lb = seg->lb;
  k->ub = seg->ub;
  k->lb_s = seg->lb_s;
  k->ub_s = seg->ub_s;
  return CCQP_OK;
}

extern "C" int ccqp_pipe_step_project_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x, const double* g,
                                          double* out, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  ccqp_segk k;
  int rc = pipe_seg(seg, &k);
  if (rc) return rc;
  if (k.dim == 0) return CCQP_OK;
  CCQP_REQUIRE(x && g && out, "NULL vector");
  hipLaunchKernelGGL(pipe_step_project_kernel, dim3(pipe_vec_grid(k.dim)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream,
                     k, x, g, out, ctx->slots);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_pipe_gemv_epilogue_f64(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x,
                                           const double* b, double* y, double* g, int64_t rows, int64_t cols,
                                           int variant, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(rows >= 0 && cols >= 0 && lda >= cols, "bad shape");
  CCQP_REQUIRE(rows == 0 || (A && x && y && b && g), "NULL operand");
  CCQP_REQUIRE(y != x && g != x, "output must not alias x");
  if (rows == 0) return CCQP_OK;
  int rc = ccqp_gemv_launch(ctx, A, lda, x, b, y, g, rows, cols, variant, (hipStream_t)stream,
                            ctx->slots + PIPE_DONE);
  if (rc) return rc;
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_pipe_bb_tail_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x_new, const double* x_old,
                                     const double* g_new, const double* g_old, double c, double gd, double tol,
                                     double eps10, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  ccqp_segk k;
  int rc = pipe_seg(seg, &k);
  if (rc) return rc;
  CCQP_REQUIRE(k.dim == 0 || (x_new && x_old && g_new && g_old), "NULL vector");
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(k.dim, &chunk, &nblk);
  hipLaunchKernelGGL(pipe_bb_tail_kernel, dim3(nblk), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, k, x_new, x_old, g_new,
                     g_old, c, gd, tol, eps10, chunk, ctx->partials, nblk, ctx->slots, ctx->host_slots_dev);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// elementwise factors of a composed operator, behind the same word
__global__ __launch_bounds__(CCQP_BLOCK) void pipe_mul_kernel(const double* x, const double* y, double* out, int64_t n,
                                                              const double* slots) {
  if (slots[PIPE_DONE] != 0.0) return;
  const int64_t stride = (int64_t)gridDim.x * CCQP_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; i < n; i += stride) out[i] = x[i] * y[i];
}

__global__ __launch_bounds__(CCQP_BLOCK) void pipe_axpby_kernel(double a, const double* x, double b, const double* y,
                                                                double* out, int64_t n, const double* slots) {
  if (slots[PIPE_DONE] != 0.0) return;
  axpby_body(a, x, b, y, out, n, blockIdx.x, gridDim.x);
}

extern "C" int ccqp_pipe_mul_f64(ccqp_ctx* ctx, const double* x, const double* y, double* out, int64_t n,
                                 void* stream) {
  CCQP_REQUIRE(ctx != nullptr && n >= 0, "bad argument");
  if (n == 0) return CCQP_OK;
  CCQP_REQUIRE(x && y && out, "NULL vector");
  hipLaunchKernelGGL(pipe_mul_kernel, dim3(pipe_vec_grid(n)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, x, y, out, n,
                     ctx->slots);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_pipe_axpby_f64(ccqp_ctx* ctx, double a, const double* x, double b, const double* y, double* out,
                                   int64_t n, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && n >= 0, "bad argument");
  if (n == 0) return CCQP_OK;
  CCQP_REQUIRE(x && y && out, "NULL vector");
  hipLaunchKernelGGL(pipe_axpby_kernel, dim3(pipe_vec_grid(n)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, a, x, b, y,
                     out, n, ctx->slots);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// Non-blocking look at the host mirror: out = {iterations finished, converged (0/1), res2, alpha}.
extern "C" int ccqp_pipe_poll(ccqp_ctx* ctx, double* out) {
  CCQP_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
  volatile double* m = ctx->host_slots + PIPE_MIRROR;
  const double done = m[1];
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  out[0] = m[0];
  out[1] = done;
  out[2] = m[2];
  out[3] = m[3];
  return CCQP_OK;
}

// ---- hipGraph form: with the scalars on the device, the kernel arguments of iteration k+3 are those of iteration
// k (a ring of three x / g buffers), so THREE iterations are captured once and replayed -- one graph launch instead
// of nine kernel launches.  Capture brackets the ordinary ccqp_pipe_* calls on `stream`.
extern "C" int ccqp_pipe_capture_begin(ccqp_ctx* ctx, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return CCQP_OK;
}

extern "C" int ccqp_pipe_capture_end(ccqp_ctx* ctx, void* stream, void** graph_exec) {
  CCQP_REQUIRE(ctx != nullptr && graph_exec != nullptr, "NULL argument");
  *graph_exec = nullptr;
  hipGraph_t graph = nullptr;
  CCQP_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
  hipGraphExec_t exec = nullptr;
  const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  CCQP_HIP(e);
  *graph_exec = exec;
  return CCQP_OK;
}

extern "C" int ccqp_pipe_graph_launch(ccqp_ctx* ctx, void* graph_exec, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && graph_exec != nullptr, "NULL argument");
  CCQP_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return CCQP_OK;
}

extern "C" int ccqp_pipe_graph_destroy(void* graph_exec) {
  if (graph_exec) CCQP_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return CCQP_OK;
}
