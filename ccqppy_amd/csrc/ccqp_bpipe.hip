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
#include "ccqp_bodies.h"

#define BP_ALPHA 48
#define BP_DONE 49
#define BP_ITERS 50
#define BP_RES2 51
#define BP_GG 3                            // g0.g0 between the two start-up reductions
#define BP_MIRROR (CCQP_NUM_SLOTS + 5)     // host_slots[69..70] = {problems converged, sequence number}

__device__ __forceinline__ ccqp_segk bp_segk(const ccqp_pprob& q) {
  ccqp_segk k;
  k.kind = q.seg.kind;
  k.flags = q.seg.flags;
  k.dim = q.seg.dim;
  k.lb = q.seg.lb;
  k.ub = q.seg.ub;
  k.lb_s = q.seg.lb_s;
  k.ub_s = q.seg.ub_s;
  return k;
}

// Block sums v[0..nred) of this block -> sums over the problem's nblk blocks, in the LAST block to arrive (the
// pattern of ccqp_reduce_finish with a per-problem area and counter).  Returns true in that block only.
__device__ __forceinline__ bool bp_finish(const ccqp_pprob& q, int nblk, int nred, double* v, double* lds4,
                                          int* is_last) {
  if (nblk == 1) return true;
  if (threadIdx.x == 0) {
    for (int k = 0; k < nred; ++k) q.partials[(int64_t)k * nblk + blockIdx.x] = v[k];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(q.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *is_last = prev + 1u == (unsigned)nblk;
    if (*is_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!*is_last) return false;
  for (int k = 0; k < nred; ++k) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += CCQP_BLOCK) acc += q.partials[(int64_t)k * nblk + i];
    v[k] = ccqp_block_sum(acc, lds4);
  }
  if (threadIdx.x == 0) __hip_atomic_store(q.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// One thread per problem and launch calls this: the last of `count` publishes the progress of the whole batch.
__device__ __forceinline__ void bp_report(unsigned* batch, double* host, unsigned count, double seq) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned prev = __hip_atomic_fetch_add(batch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (prev + 1u == count) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(batch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned conv = __hip_atomic_load(batch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    host[BP_MIRROR] = (double)conv;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(host + BP_MIRROR + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void bpipe_begin_kernel(unsigned* batch) {
  if (threadIdx.x == 0) {
    batch[0] = 0u;
    batch[1] = 0u;
  }
}

// x_new = P(x_cur - alpha g_cur), grid (blocks, problems)
__global__ __launch_bounds__(CCQP_BLOCK) void bpipe_step_kernel(const ccqp_pprob* __restrict__ probs, int inew,
                                                                int icur) {
  const ccqp_pprob& q = probs[blockIdx.y];
  if (q.slab[BP_DONE] != 0.0) return;
  const double b = -q.slab[BP_ALPHA];
  proj_elem_body(bp_segk(q), 1.0, q.x[icur], b, q.g[icur], q.x[inew], blockIdx.x, gridDim.x);
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
    gemv_body<1, U, ALIGNED, NT>(q.A, q.lda, q.x[i], q.b, q.y, q.g[i], n, n, n, wave, nwaves);
  else
    gemv_body<1, U, ALIGNED, NT>(q.A, q.lda, q.g[i], nullptr, q.y, nullptr, n, n, n, wave, nwaves);
}

// start-up, first reduction: residual^2 of (x[0], g[0]) and g0.g0; the exit test of solvers.py `if res >= tol`
__global__ __launch_bounds__(CCQP_BLOCK) void bpipe_init_res_kernel(const ccqp_pprob* __restrict__ probs, double gd,
                                                                    int64_t chunk, int nblk,
                                                                    unsigned* __restrict__ batch,
                                                                    double* __restrict__ host, unsigned count,
                                                                    double seq) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  const ccqp_pprob& q = probs[blockIdx.y];
  const ccqp_segk sg = bp_segk(q);
  double v[2];
  v[0] = residual_body(sg, q.x[0], q.g[0], q.c, gd, chunk, blockIdx.x, lds4);
  v[1] = dot_body(q.g[0], q.g[0], sg.dim, chunk, blockIdx.x, lds4);
  if (!bp_finish(q, nblk, 2, v, lds4, &is_last)) return;
  if (threadIdx.x == 0) {
    q.slab[0] = v[0];
    q.slab[BP_GG] = v[1];
    q.slab[BP_RES2] = v[0];
    q.slab[BP_ITERS] = 0.0;
    q.slab[BP_ALPHA] = 0.0;
    if (sqrt(v[0]) < q.tol) {
      q.slab[BP_DONE] = 1.0;
      __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      q.slab[BP_DONE] = 0.0;
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
  if (q.slab[BP_DONE] != 0.0) return;
  double v[1];
  v[0] = dot_body(q.g[0], q.y, n, chunk, blockIdx.x, lds4);
  if (!bp_finish(q, nblk, 1, v, lds4, &is_last)) return;
  if (threadIdx.x == 0) {
    q.slab[1] = v[0];
    q.slab[BP_ALPHA] = q.slab[BP_GG] / v[0];
  }
}

// residual^2, dx.dx, dx.dg per problem; the problem's last block derives alpha and the exit test.  grid (nblk, problems)
__global__ __launch_bounds__(CCQP_BLOCK) void bpipe_tail_kernel(const ccqp_pprob* __restrict__ probs, int inew,
                                                                int icur, double gd, double eps10, int64_t chunk,
                                                                int nblk, unsigned* __restrict__ batch,
                                                                double* __restrict__ host, unsigned count,
                                                                double seq) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  const ccqp_pprob& q = probs[blockIdx.y];
  if (q.slab[BP_DONE] != 0.0) {             // converged earlier: only the roll-call
    if (blockIdx.x == 0 && threadIdx.x == 0) bp_report(batch, host, count, seq);
    return;
  }
  double v[3];
  bb_tail_body(bp_segk(q), q.x[inew], q.x[icur], q.g[inew], q.g[icur], q.c, gd, chunk, blockIdx.x, lds4, &v[0], &v[1],
               &v[2]);
  if (!bp_finish(q, nblk, 3, v, lds4, &is_last)) return;
  if (threadIdx.x == 0) {
    // the host loop's scalar step (solvers.py:647-656): res = sqrt(res2); alpha = dx.dx / (dx.dg + 10 eps)
    const double res = sqrt(v[0]);
    const double den = v[2] + eps10;
    q.slab[0] = v[0];
    q.slab[1] = v[1];
    q.slab[2] = v[2];
    q.slab[BP_RES2] = v[0];
    q.slab[BP_ALPHA] = v[1] / den;
    q.slab[BP_ITERS] = q.slab[BP_ITERS] + 1.0;
    if (res < q.tol) {
      q.slab[BP_DONE] = 1.0;
      __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bp_report(batch, host, count, seq);
  }
}

static inline int bp_vec_grid(int64_t n) {
  int64_t g = (n + CCQP_BLOCK - 1) / CCQP_BLOCK;
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return (int)g;
}

// `flags` of the entry points: bit 0 = 16-byte aligned operands; the rest is a TUNING word that never changes a
// result bit (0 = defaults): bits 1-2 matrix loads 0 default / 1 cacheable / 2 non-temporal, bits 4-6 columns
// per wave step in units of 128 (1, 2, 4), bits 8.. workgroups per problem.
static void bp_gemv(const ccqp_pprob* probs, int count, int64_t n, int i, int mode, int flags, hipStream_t s) {
  const bool aligned = flags & 1;
  int nt = (flags >> 1) & 3, U = (flags >> 4) & 7;
  int64_t per = flags >> 8;
  const double bytes = 8.0 * (double)n * (double)n * count;
  if (nt == 0) nt = bytes > 192.0 * 1048576.0 ? 2 : 1;      // a batch that fits the MALL is re-read from there
  if (U == 0) U = n >= 4096 ? 4 : 1;
  const int64_t rows_per_wg = CCQP_BLOCK / 64;
  // many small workgroups: ~64k per launch measured best for n = 256..4096 (profiles/r01g_bpipe_sweep.txt)
  if (per == 0) per = (65536 + count - 1) / count < 8 ? 8 : (65536 + count - 1) / count;
  if (per > (n + rows_per_wg - 1) / rows_per_wg) per = (n + rows_per_wg - 1) / rows_per_wg;
  if (per < 1) per = 1;
  const dim3 grid((unsigned)per, count);
#define BP_GEMV(UU, AL, NTT)                                                                              \
  hipLaunchKernelGGL((bpipe_gemv_kernel<UU, AL, NTT>), grid, dim3(CCQP_BLOCK), 0, s, probs, i, mode, n)
#define BP_GEMV_U(AL, NTT)            \
  do {                                \
    if (U >= 4) BP_GEMV(4, AL, NTT);  \
    else if (U == 2) BP_GEMV(2, AL, NTT); \
    else BP_GEMV(1, AL, NTT);         \
  } while (0)
  if (aligned) {
    if (nt == 2) BP_GEMV_U(true, true);
    else BP_GEMV_U(true, false);
  } else {
    if (nt == 2) BP_GEMV_U(false, true);
    else BP_GEMV_U(false, false);
  }
#undef BP_GEMV_U
#undef BP_GEMV
}

#define BP_CHECK_BATCH()                                                                          \
  CCQP_REQUIRE(ctx != nullptr && dev_probs != nullptr && dev_batch != nullptr, "NULL argument"); \
  CCQP_REQUIRE(count >= 1 && count <= 65535 && n >= 1, "bad batch shape")

// Start-up of all solves (solvers.py:626-636): g0 = A x0 + b, residual, A g0, alpha0.  Expects x[0] staged, the
// per-problem counters zero; sequence number 0 is published when every problem has its residual.
extern "C" int ccqp_bpipe_start_f64(ccqp_ctx* ctx, const ccqp_pprob* dev_probs, int count, int64_t n, double gd,
                                    int flags, uint32_t* dev_batch, void* stream) {
  BP_CHECK_BATCH();
  hipStream_t s = (hipStream_t)stream;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  // the mirror is reset from the host side: a poll issued before the first kernel runs must not see the final
  // state of an earlier batch (whose solves have all been read back by then)
  ctx->host_slots[BP_MIRROR] = 0.0;
  ctx->host_slots[BP_MIRROR + 1] = -1.0;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  hipLaunchKernelGGL(bpipe_begin_kernel, dim3(1), dim3(64), 0, s, dev_batch);
  // (DONE is read by the matvec: the caller zeroes the slabs before this call)
  bp_gemv(dev_probs, count, n, 0, 0, flags, s);
  hipLaunchKernelGGL(bpipe_init_res_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, gd, chunk, nblk,
                     dev_batch, ctx->host_slots_dev, (unsigned)count, 0.0);
  bp_gemv(dev_probs, count, n, 0, 1, flags, s);
  hipLaunchKernelGGL(bpipe_init_alpha_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, n, chunk, nblk);
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
  hipLaunchKernelGGL(bpipe_step_kernel, dim3(bp_vec_grid(n), count), dim3(CCQP_BLOCK), 0, s, dev_probs, inew, icur);
  bp_gemv(dev_probs, count, n, inew, 0, flags, s);
  if (with_tail) {
    int64_t chunk;
    int nblk;
    ccqp_red_geometry(n, &chunk, &nblk);
    hipLaunchKernelGGL(bpipe_tail_kernel, dim3(nblk, count), dim3(CCQP_BLOCK), 0, s, dev_probs, inew, icur, gd, eps10,
                       chunk, nblk, dev_batch, ctx->host_slots_dev, (unsigned)count, (double)iteration);
  }
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// Non-blocking look at the host mirror: out = {problems converged, last sequence number all problems finished}.
extern "C" int ccqp_bpipe_poll(ccqp_ctx* ctx, double* out) {
  CCQP_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
  volatile double* m = ctx->host_slots + BP_MIRROR;
  const double seq = m[1];
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  out[0] = m[0];
  out[1] = seq;
  return CCQP_OK;
}

// doubles of partial-sum area one problem needs for vectors of length n
extern "C" int64_t ccqp_bpipe_partials_doubles(int64_t n) {
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n > 0 ? n : 1, &chunk, &nblk);
  return 3 * (int64_t)nblk;
}
