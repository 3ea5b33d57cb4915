// Kernel BODIES shared by the single-problem kernels (ccqp_vec/proj/mprgp/gemv.hip) and their
// batched twins (ccqp_batch.hip, one problem per blockIdx.y).  A body takes the block coordinates
// (bx of nbx) explicitly, so both launch shapes execute literally the same arithmetic in the same
// order: batched results are bitwise those of one launch per problem.
#pragma once
#include "ccqp_internal.h"

#include <stdlib.h>
#ifdef __HIPCC__
#define BODY_LOOP(i, n) \
  for (int64_t i = (int64_t)bx * CCQP_BLOCK + threadIdx.x; i < (n); i += (int64_t)nbx * CCQP_BLOCK)

// ------------------------------------------------------------------ elementwise
__device__ __forceinline__ void axpby_body(double a, const double* x, double b, const double* y, double* out,
                                           int64_t n, int bx, int nbx) {
  BODY_LOOP(i, n) {
    const double t1 = a * x[i];
    const double t2 = b * y[i];
    out[i] = t1 + t2;
  }
}

__device__ __forceinline__ void scale_body(double a, const double* x, double* out, int64_t n, int bx, int nbx) {
  BODY_LOOP(i, n) out[i] = a * x[i];
}

__device__ __forceinline__ void copy_body(const double* x, double* out, int64_t n, int bx, int nbx) {
  BODY_LOOP(i, n) out[i] = x[i];
}

__device__ __forceinline__ void fill_body(double v, double* out, int64_t n, int bx, int nbx) {
  BODY_LOOP(i, n) out[i] = v;
}

// out = P(a*x + b*y) (y == NULL: out = P(x)), elementwise kinds
__device__ __forceinline__ void proj_elem_body(const ccqp_segk& sg, double a, const double* x, double b,
                                               const double* y, double* out, int bx, int nbx) {
  BODY_LOOP(i, sg.dim) {
    double u = x[i];
    if (y) {
      const double t1 = a * u;
      const double t2 = b * y[i];
      u = t1 + t2;
    }
    const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
    const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
    out[i] = ccqp_proj_elem(sg.kind, sg.flags, u, lb, ub);
  }
}

// out = isclose(x, P(x)) * g
__device__ __forceinline__ void free_gradient_body(const ccqp_segk& sg, const double* x, const double* g,
                                                   double* out, int bx, int nbx) {
  BODY_LOOP(i, sg.dim) {
    const double xi = x[i];
    const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
    const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
    const double p = ccqp_proj_elem(sg.kind, sg.flags, xi, lb, ub);
    out[i] = (ccqp_isclose(xi, p) ? 1.0 : 0.0) * g[i];
  }
}

// out = psi - ((psi*Ap)/pAp)*p
__device__ __forceinline__ void cg_direction_body(const double* psi, const double* Ap, double pAp, const double* p,
                                                  double* out, int64_t n, int bx, int nbx) {
  BODY_LOOP(i, n) {
    const double ps = psi[i];
    const double t1 = ps * Ap[i];
    const double beta = t1 / pAp;
    const double t2 = beta * p[i];
    out[i] = ps - t2;
  }
}

// ------------------------------------------------------------------ stage-1 reductions
// Every stage-1 body reduces elements [bx*chunk, (bx+1)*chunk) with the same thread->element map
// (t, t+256, ...) and returns the block sum in thread 0.
#define CHUNK_RANGE(n)                       \
  const int64_t lo = (int64_t)bx * chunk;    \
  int64_t hi = lo + chunk;                   \
  if (hi > (n)) hi = (n);

// A thread's walk over its elements of a chunk (i = lo + t, lo + t + 256, ...) with the LOADS of U consecutive
// elements issued before the first of them is used: `body(i, v)` is called in increasing i with v[k] = p[k][i] (0.0 for
// a NULL p[k]) -- the arithmetic order of the plain loop, hence the same bits, but U x NV loads in flight per thread
// instead of one dependent group per iteration.  The stage-1 bodies of ONE problem are two to eight blocks whose threads
// take eight elements each: written as a plain loop the compiler waits for every iteration's loads (vmcnt(0) per
// element), ~2 us each -- the 15 us of a lock-step APGD decision at n = 4096 (profiles/r04_trace_rounds_n4096.json).
template <int NV, int U, class F>
__device__ __forceinline__ void ccqp_chunk_walk(const double* const (&p)[NV], int64_t lo, int64_t hi, F body) {
  for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (int64_t)U * CCQP_BLOCK) {
    double v[U][NV];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * CCQP_BLOCK;
#pragma unroll
      for (int k = 0; k < NV; ++k) v[u][k] = (i < hi && p[k] != nullptr) ? p[k][i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * CCQP_BLOCK;
      if (i < hi) body(i, v[u]);
    }
  }
}

__device__ __forceinline__ double dot_body(const double* __restrict__ x, const double* __restrict__ y, int64_t n,
                                           int64_t chunk, int bx, double* lds4) {
  CHUNK_RANGE(n)
  double acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) acc = fma(x[i], y[i], acc);
  return ccqp_block_sum(acc, lds4);
}

__device__ __forceinline__ double partial_sum_body(const double* partials, int nblk, double* lds4) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblk; i += CCQP_BLOCK) acc += partials[i];
  return ccqp_block_sum(acc, lds4);
}

__device__ __forceinline__ double residual_body(const ccqp_segk& sg, const double* x, const double* g, double c,
                                                double gd, int64_t chunk, int bx, double* lds4) {
  CHUNK_RANGE(sg.dim)
  double acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
    const double xi = x[i];
    const double t = gd * g[i];
    const double u = xi - t;
    const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
    const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
    const double p = ccqp_proj_elem(sg.kind, sg.flags, u, lb, ub);
    const double d = xi - p;
    const double v = c * d;
    acc = fma(v, v, acc);
  }
  return ccqp_block_sum(acc, lds4);
}

// Fused tail of a Barzilai-Borwein iteration (solvers.py:647-656 and the BBPGDf / PGD analogues): ONE pass
// over (x_new, x_old, g_new, g_old) yields the three block sums that residual_body on (x_new, g_new),
// dot_body(dx, dx) and dot_body(dx, dg) would give with dx = x_new - x_old, dg = g_new - g_old materialised
// first (each difference is rounded once, exactly like the axpby it replaces): same thread -> element map, same
// fma chains, so the bits are those of the four separate launches.
__device__ __forceinline__ void bb_tail_body(const ccqp_segk& sg, const double* xn, const double* xo,
                                             const double* gn, const double* go, double c, double gd,
                                             int64_t chunk, int bx, double* lds4, double* res2, double* dxdx,
                                             double* dxdg) {
  CHUNK_RANGE(sg.dim)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
    const double xi = xn[i], gi = gn[i];
    const double t = gd * gi;
    const double u = xi - t;
    const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
    const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
    const double p = ccqp_proj_elem(sg.kind, sg.flags, u, lb, ub);
    const double d = xi - p;
    const double v = c * d;
    a0 = fma(v, v, a0);
    const double dx = xi - xo[i];
    const double dg = gi - go[i];
    a1 = fma(dx, dx, a1);
    a2 = fma(dx, dg, a2);
  }
  double v[3] = {a0, a1, a2};
  ccqp_block_sum_multi<3>(v);                 // (same bits as three ccqp_block_sum calls, a third of the barriers)
  *res2 = v[0];
  *dxdx = v[1];
  *dxdg = v[2];
}

// count of i with not isclose(yf_i, P(yf)_i), yf = x - (alpha0 * 0.5^k)*p: this thread's share of block bx
__device__ __forceinline__ double feasible_thread_count(const ccqp_segk& sg, const double* x, const double* p,
                                                        double alpha0, int k, int64_t chunk, int bx) {
  double alpha = alpha0;
  for (int j = 0; j < k; ++j) alpha *= 0.5;  // the reference's `alpha_f *= 0.5`, k times
  CHUNK_RANGE(sg.dim)
  double acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
    const double t = alpha * p[i];
    const double yf = x[i] - t;
    const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
    const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
    const double pr = ccqp_proj_elem(sg.kind, sg.flags, yf, lb, ub);
    acc += ccqp_isclose(yf, pr) ? 0.0 : 1.0;
  }
  return acc;
}

__device__ __forceinline__ double feasible_body(const ccqp_segk& sg, const double* x, const double* p,
                                                double alpha0, int k, int64_t chunk, int bx, double* lds4) {
  return ccqp_block_sum(feasible_thread_count(sg, x, p, alpha0, k, chunk, bx), lds4);
}

// xp = P(x); psi = isclose(x,xp)*g; w = 1 - isclose; returns sum (x-xp)^2
__device__ __forceinline__ double split_body(const ccqp_segk& sg, const double* x, const double* g, double* xp,
                                             double* psi, double* w, int64_t chunk, int bx, double* lds4) {
  CHUNK_RANGE(sg.dim)
  double acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
    const double xi = x[i];
    const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
    const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
    const double p = ccqp_proj_elem(sg.kind, sg.flags, xi, lb, ub);
    const bool close = ccqp_isclose(xi, p);
    xp[i] = p;
    psi[i] = (close ? 1.0 : 0.0) * g[i];
    w[i] = close ? 0.0 : 1.0;
    const double d = xi - p;
    acc = fma(d, d, acc);
  }
  if (sg.dim <= CCQP_SMALL_SEG_MAX) {
    // a tiny block (one chunk): |x - P(x)|^2 as the SEQUENTIAL chain of the thread-per-block kernel
    // (small_segments_kernel, ccqp_proj.hip: what BoxProjOp.normal_vector of such a block -- and the host-driven MPRGP
    // loop for n <= 64 -- computes; numpy's own order for short vectors), so that the one order-sensitive decision of an
    // MPRGP iteration, sqrt(dist2) > 1e-8 (solution_spaces.py:310), has the same bits in every form of the solver
    __syncthreads();
    double sq = 0.0;
    if (threadIdx.x == 0)
      for (int64_t i = 0; i < sg.dim; ++i) {
        const double d = x[i] - xp[i];
        sq = fma(d, d, sq);
      }
    __syncthreads();
    return sq;
  }
  return ccqp_block_sum(acc, lds4);
}

// nv = normal vector from xp (zeros if sqrt(dist2) > 1e-8); returns sum nv*g
__device__ __forceinline__ double normal_dot_body(const ccqp_segk& sg, const double* xp, const double* g,
                                                  double* nvout, double dist2, int64_t chunk, int bx,
                                                  double* lds4) {
  const bool outside = !ccqp_isclose(sqrt(dist2), 0.0);
  const bool has_ub = sg.kind == CCQP_OP_UPPER || sg.kind == CCQP_OP_BOX;
  const bool has_lb = sg.kind == CCQP_OP_LOWER || sg.kind == CCQP_OP_BOX;
  CHUNK_RANGE(sg.dim)
  double acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
    double nv = 0.0;
    if (!outside) {
      const double v = xp[i];
      const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
      const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
      if (has_ub && ccqp_isclose(v, ub)) nv = 1.0;
      else if (has_lb && ccqp_isclose(v, lb)) nv = -1.0;
    }
    nvout[i] = nv;
    acc = fma(nv, g[i], acc);
  }
  return ccqp_block_sum(acc, lds4);
}

// beta = w*(g - min(0, ng)*nv); *sb = sum beta^2, *sp = sum psi^2 (valid in thread 0)
__device__ __forceinline__ void proportion_body(const double* w, const double* g, const double* nv,
                                                const double* psi, double ng, int64_t n, int64_t chunk, int bx,
                                                double* lds4, double* sb, double* sp) {
  const double sc = (ng != ng) ? ng : (ng < 0.0 ? ng : 0.0);  // np.min([0, ng])
  CHUNK_RANGE(n)
  double ab = 0.0, ap = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
    const double t = sc * nv[i];
    const double u = g[i] - t;
    const double be = w[i] * u;
    ab = fma(be, be, ab);
    const double ps = psi[i];
    ap = fma(ps, ps, ap);
  }
  *sb = ccqp_block_sum(ab, lds4);
  *sp = ccqp_block_sum(ap, lds4);
}

// ------------------------------------------------------------------ matvec
typedef double ccqp_d2 __attribute__((ext_vector_type(2)));

template <bool ALIGNED, bool NT>
__device__ __forceinline__ ccqp_d2 ccqp_ld2(const double* p) {
  if constexpr (ALIGNED) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const ccqp_d2*>(p));
    else return *reinterpret_cast<const ccqp_d2*>(p);
  } else {
    ccqp_d2 v;
    if constexpr (NT) {
      v.x = __builtin_nontemporal_load(p);
      v.y = __builtin_nontemporal_load(p + 1);
    } else {
      v.x = p[0];
      v.y = p[1];
    }
    return v;
  }
}

// One wave64 computes one TASK = R consecutive rows.
// Lane l accumulates columns {128k+2l, 128k+2l+1} in increasing k into one accumulator per row.
template <int R, int U, bool ALIGNED, bool NT>
__device__ __forceinline__ void gemv_task(const double* __restrict__ A, int64_t lda, const double* __restrict__ x,
                                          const double* __restrict__ b, double* __restrict__ y,
                                          double* __restrict__ g, int64_t rows, int64_t cols, int64_t task) {
  const int lane = threadIdx.x & 63;
  constexpr int64_t SPAN = 128 * U;
  const int64_t nfull = cols / SPAN;
  {
    const int64_t r0 = task * R;
    const double* rp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t row = r0 + r;
      if (row > rows - 1) row = rows - 1;  // clamp: tail rows re-read a valid row, never stored
      rp[r] = A + row * lda;
    }
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;

    for (int64_t s = 0; s < nfull; ++s) {
      const int64_t c0 = s * SPAN + 2 * lane;
      ccqp_d2 xv[U];
      ccqp_d2 av[R][U];
#pragma unroll
      for (int u = 0; u < U; ++u) xv[u] = ccqp_ld2<ALIGNED, false>(x + c0 + 128 * u);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) av[r][u] = ccqp_ld2<ALIGNED, NT>(rp[r] + c0 + 128 * u);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          acc[r] = fma(av[r][u].x, xv[u].x, acc[r]);
          acc[r] = fma(av[r][u].y, xv[u].y, acc[r]);
        }
    }
    // column tail (< SPAN columns), same lane -> column map and order
    for (int64_t c = nfull * SPAN + 2 * lane; c < cols; c += 128) {
      if (c + 1 < cols) {
        const ccqp_d2 xv = ccqp_ld2<ALIGNED, false>(x + c);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const ccqp_d2 a = ccqp_ld2<ALIGNED, NT>(rp[r] + c);
          acc[r] = fma(a.x, xv.x, acc[r]);
          acc[r] = fma(a.y, xv.y, acc[r]);
        }
      } else {
        const double xs = x[c];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fma(rp[r][c], xs, acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = ccqp_wave_sum(acc[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = r0 + r;
      if (lane == r && row < rows) {
        y[row] = acc[r];
        if (g) g[row] = acc[r] + b[row];  // `A.dot(x) + b`: one more rounding, as numpy
      }
    }
  }
}

// wave `wave` of `nwaves` grid-strides over the tasks of ONE matrix
template <int R, int U, bool ALIGNED, bool NT>
__device__ __forceinline__ void gemv_body(const double* __restrict__ A, int64_t lda, const double* __restrict__ x,
                                          const double* __restrict__ b, double* __restrict__ y,
                                          double* __restrict__ g, int64_t rows, int64_t cols, int64_t ntasks,
                                          int64_t wave, int64_t nwaves) {
  for (int64_t task = wave; task < ntasks; task += nwaves)
    gemv_task<R, U, ALIGNED, NT>(A, lda, x, b, y, g, rows, cols, task);
}
// ---- the matvec with its rows staged through a per-wave LDS ring by LDS-DMA loads (global_load_lds_dwordx4)
// One wave owns R consecutive rows per task, as in gemv_task.  Per 128-column step it issues R + 1 LDS-DMA loads (1 KiB
// of each row, non-temporal, and the matching 1 KiB of x out of L2) into one slot of its PRIVATE ring of S slots -- no
// VGPR destination, so S - 1 steps (21 KiB per wave at R = 2, S = 8) are in flight instead of one or two -- waits with
// a counted `s_waitcnt vmcnt`, reads the slot back with ds_read_b128 and runs the same fma chain: lane l accumulates
// columns {128k + 2l, 128k + 2l + 1} in increasing k into one accumulator per row, hence the SAME BITS as gemv_task.
// The ring runs CONTINUOUSLY across the wave's tasks (the loads of the next rows are in flight while the last steps of
// the current ones are consumed; no drain at a row end), in groups of GEMV_LDS_GROUP results: row sums are parked in
// LDS (`res`) and stored -- with the `+ b` epilogue -- after the group's last load has landed, so no ordinary load or
// store is ever in flight beside the DMA (hipcc would answer with vmcnt(0), and stores count on the same counter).
// No barrier anywhere: a wave reads only what it has loaded itself.  Needs an even cols >= 128 and 16-byte aligned rows / x.
#define GEMV_LDS_RESCAP 512        // row sums a wave parks before it flushes them (doubles of LDS per wave: 4 KiB)
struct gemv_lds_desc {             // operands of one matrix of the multi-matrix form (gemv_lds_multi_body)
  const double* A;
  int64_t lda;
  const double* x;
  const double* b;
  double* y;
  double* g;
};
// Which rows a task takes.  The R row streams a wave reads side by side should lie about 1 MiB apart: the DRAM serves
// R = 4 streams at 1 MiB spacing at 7.18 TB/s, at 64 KiB spacing at 6.83-6.98, at 4 MiB at 6.7, at 128 MiB again at 7.2
// (tools/readbw.py, profiles/r03_read_ceiling.jsonl; spacing between DIFFERENT waves' streams does not matter, they
// drift).  So task T of a matrix with row stride `rs` takes rows first, first + rs, ..., first + (R - 1) rs with
// first = (T / rs) R rs + T mod rs: groups of R rs consecutive rows, dealt to rs tasks.  rs = 1: R consecutive rows.
// The bits of y do not depend on it (they are a function of (row, n) only).
// (`forced` = the context's CCQP_OPT_GEMV_ROW_STRIDE, 0 = by shape: k forces rs = k -- the A/B switch of the tools; 1 = R
// consecutive rows, the only form before round 3's probe)
static inline int64_t gemv_lds_row_stride(int64_t forced, int64_t lda, int64_t rows, int R) {
  if (forced >= 1) {
    int64_t k = forced;
    while (k > 1 && (int64_t)R * k > rows) k /= 2;
    return k < 1 ? 1 : k;
  }
  const double rowbytes = 8.0 * (double)lda;
  int64_t rs = rowbytes <= 1.5 * 1048576.0 ? (int64_t)(1048576.0 / rowbytes + 0.5) : (int64_t)(134217728.0 / rowbytes + 0.5);
  if (rs < 1) rs = 1;
  while (rs > 1 && (int64_t)R * rs > rows) rs /= 2;        // (a matrix with fewer rows than one group)
  return rs;
}
static inline int64_t gemv_lds_tasks(int64_t rows, int R, int64_t rs) {
  return (rows + R * rs - 1) / (R * rs) * rs;
}
struct gemv_lds3_desc {            // operands of one matrix of the three-vector multi-matrix form (gemv_lds3_multi_body)
  const double* A;
  int64_t lda;
  const double* x[3];              // (a matrix with fewer products repeats a vector and leaves y / g of it NULL)
  const double* b;
  double* y[3];
  double* g[3];
};
template <int K>
struct gemv_ldsk_io {              // by-value kernel argument: the K vectors and their outputs (g[j] may be NULL)
  const double* x[K];
  double* y[K];
  double* g[K];
};
#if defined(__HIP_DEVICE_COMPILE__)
template <int R>
__device__ __forceinline__ int64_t gemv_lds_first_row(int64_t T, int64_t rs) {
  const int64_t q = T / rs;
  return q * (R * rs) + (T - q * rs);
}
template <int R, int S, bool NT>
__device__ __forceinline__ void gemv_lds_body(const double* __restrict__ A, int64_t lda, const double* __restrict__ x,
                                              const double* __restrict__ b, double* __restrict__ y,
                                              double* __restrict__ g, int64_t rows, int64_t cols, int64_t ntasks,
                                              int64_t wave, int64_t nwaves, char* mine, double* res, int64_t rs) {
  constexpr int PIECE = 1024;                   // bytes one wave instruction moves: 64 lanes x 16 B
  constexpr int SLOT = (R + 1) * PIECE;
  constexpr int GROUP = GEMV_LDS_RESCAP / R;    // tasks per flush
  const int lane = threadIdx.x & 63;
  // full 128-column steps, then -- cols even, not a multiple of 128 -- one more step in which only the lanes whose two
  // columns exist load and accumulate (gemv_task's column tail: same lane -> column map, same place in the chain)
  const int64_t nfull = cols / 128;
  const int tail = (int)(cols - nfull * 128);
  const bool mine_in_tail = 2 * lane < tail;
  const int64_t nsteps = nfull + (tail ? 1 : 0);
  const int64_t mytasks = wave < ntasks ? (ntasks - wave + nwaves - 1) / nwaves : 0;
  const double* xp = x + 2 * lane;
  for (int64_t k0 = 0; k0 < mytasks; k0 += GROUP) {
    const int kn = (int)(mytasks - k0 < GROUP ? mytasks - k0 : GROUP);
    const int64_t total = (int64_t)kn * nsteps;
    // ---- issue cursor: (task ik of the group, step is), ring slot islot, steps issued so far
    int ik = 0, islot = 0;
    int64_t is = 0, issued = 0;
    const double* irp[R];
    auto rows_of = [&](int k) {
      const int64_t r0 = gemv_lds_first_row<R>(wave + (k0 + k) * nwaves, rs);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int64_t row = r0 + r * rs;
        if (row > rows - 1) row = rows - 1;     // clamp: tail rows re-read a valid row, never stored
        irp[r] = A + row * lda + 2 * lane;
      }
    };
    rows_of(0);
    auto issue_next = [&]() {
      char* slot = mine + islot * SLOT;
      if (is < nfull || mine_in_tail) {         // (the tail step: lanes past the last column issue nothing)
#pragma unroll
        for (int r = 0; r < R; ++r)
          __builtin_amdgcn_global_load_lds(irp[r] + is * 128, (__attribute__((address_space(3))) void*)(slot + r * PIECE),
                                           16, 0, NT ? 2 : 0);
        __builtin_amdgcn_global_load_lds(xp + is * 128, (__attribute__((address_space(3))) void*)(slot + R * PIECE), 16, 0,
                                         0);
      }
      islot = islot + 1 == S ? 0 : islot + 1;
      ++issued;
      if (++is == nsteps) {
        is = 0;
        if (++ik < kn) rows_of(ik);
      }
    };
    for (int p = 0; p < S - 1 && issued < total; ++p) issue_next();
    int cslot = 0;
    for (int kk = 0; kk < kn; ++kk) {
      double acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = 0.0;
      for (int64_t s = 0; s < nsteps; ++s) {
        if (issued < total) {
          issue_next();
          // the S - 1 newest steps may stay in flight: (S - 1) * (R + 1) loads
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * (R + 1)) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last S - 1 steps of a group: drain
        }
        const char* slot = mine + cslot * SLOT;
        if (s < nfull || mine_in_tail) {
          // (all R + 1 pieces asked for back to back, then the fma chains: see gemv_ldsk_body)
          const ccqp_d2 xv = *reinterpret_cast<const ccqp_d2*>(slot + R * PIECE + lane * 16);
          ccqp_d2 av[R];
#pragma unroll
          for (int r = 0; r < R; ++r) av[r] = *reinterpret_cast<const ccqp_d2*>(slot + r * PIECE + lane * 16);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            acc[r] = fma(av[r].x, xv.x, acc[r]);
            acc[r] = fma(av[r].y, xv.y, acc[r]);
          }
        }
        cslot = cslot + 1 == S ? 0 : cslot + 1;
        asm volatile("" ::: "memory");          // the slot's reads stay ahead of the DMA that refills it
      }
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = ccqp_wave_sum(acc[r]);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (lane == r) res[kk * R + r] = acc[r];
    }
    // ---- flush: every DMA of the group has landed (the last steps waited for vmcnt(0))
    for (int j = lane; j < kn * R; j += 64) {
      const int kk = j / R, r = j - kk * R;
      const int64_t row = gemv_lds_first_row<R>(wave + (k0 + kk) * nwaves, rs) + r * rs;
      if (row < rows) {
        const double v = res[j];
        y[row] = v;
        if (g) g[row] = v + b[row];             // `A.dot(x) + b`: one more rounding, as numpy
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the stores count on the DMA's counter)
  }
}
// ---- K right-hand sides in one pass over A (y_j = A x_j; g_j = y_j + b where asked; K = 2, 3): the ring's slot holds
// R row pieces and ALL K x pieces, every row piece is read back once and multiplied into K accumulators.  Each product
// has the bits of its own gemv_lds_body / gemv_task pass (same lane -> column map, same fma chain per accumulator); the
// matrix is read from HBM once instead of K times.  (K = 2: APGD's trial point and its momentum point, solvers.py:275,290;
// K = 3: MPRGP's A.dx, A.x_{k+1} + b, A.p_{k+1} on a rank's row block, solvers.py:1129|1163, 1086, 1100.)
// `rescap` <= GEMV_LDS_RESCAP: row sums a wave parks before it stores them (the flush group: rescap / (K R) tasks).
template <int R, int S, int K, bool NT>
__device__ __forceinline__ void gemv_ldsk_body(const double* __restrict__ A, int64_t lda, const gemv_ldsk_io<K>& io,
                                               const double* __restrict__ b, int64_t rows, int64_t cols, int64_t ntasks,
                                               int64_t wave, int64_t nwaves, char* mine, double* res, int64_t rs,
                                               int rescap) {
  constexpr int PIECE = 1024;
  constexpr int SLOT = (R + K) * PIECE;
  const int GROUP = rescap / (K * R);           // tasks per flush (K sums per row)
  const int lane = threadIdx.x & 63;
  const int64_t nfull = cols / 128;
  const int tail = (int)(cols - nfull * 128);
  const bool mine_in_tail = 2 * lane < tail;
  const int64_t nsteps = nfull + (tail ? 1 : 0);
  const int64_t mytasks = wave < ntasks ? (ntasks - wave + nwaves - 1) / nwaves : 0;
  const double* xp[K];
#pragma unroll
  for (int j = 0; j < K; ++j) xp[j] = io.x[j] + 2 * lane;
  for (int64_t k0 = 0; k0 < mytasks; k0 += GROUP) {
    const int kn = (int)(mytasks - k0 < GROUP ? mytasks - k0 : GROUP);
    const int64_t total = (int64_t)kn * nsteps;
    int ik = 0, islot = 0;
    int64_t is = 0, issued = 0;
    const double* irp[R];
    auto rows_of = [&](int k) {
      const int64_t r0 = gemv_lds_first_row<R>(wave + (k0 + k) * nwaves, rs);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int64_t row = r0 + r * rs;
        if (row > rows - 1) row = rows - 1;     // clamp: tail rows re-read a valid row, never stored
        irp[r] = A + row * lda + 2 * lane;
      }
    };
    rows_of(0);
    auto issue_next = [&]() {
      char* slot = mine + islot * SLOT;
      if (is < nfull || mine_in_tail) {
#pragma unroll
        for (int r = 0; r < R; ++r)
          __builtin_amdgcn_global_load_lds(irp[r] + is * 128, (__attribute__((address_space(3))) void*)(slot + r * PIECE),
                                           16, 0, NT ? 2 : 0);
#pragma unroll
        for (int j = 0; j < K; ++j)
          __builtin_amdgcn_global_load_lds(xp[j] + is * 128,
                                           (__attribute__((address_space(3))) void*)(slot + (R + j) * PIECE), 16, 0, 0);
      }
      islot = islot + 1 == S ? 0 : islot + 1;
      ++issued;
      if (++is == nsteps) {
        is = 0;
        if (++ik < kn) rows_of(ik);
      }
    };
    for (int p = 0; p < S - 1 && issued < total; ++p) issue_next();
    int cslot = 0;
    for (int kk = 0; kk < kn; ++kk) {
      double acc[K][R];
#pragma unroll
      for (int j = 0; j < K; ++j)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[j][r] = 0.0;
      for (int64_t s = 0; s < nsteps; ++s) {
        if (issued < total) {
          issue_next();
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * (R + K)) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char* slot = mine + cslot * SLOT;
        if (s < nfull || mine_in_tail) {
          // ALL R + K pieces of the slot are asked for back to back, then the fma chains run: left to itself the
          // scheduler interleaves reads and fmas in three groups with a full LDS round trip (lgkmcnt(0)) between them,
          // ~300 clocks per step on the wave's critical path -- and a wave that consumes slowly issues its next DMA
          // late (why two waves per CU were 14 % short for the pair, profiles/r04_gemv2_shapes.jsonl)
          ccqp_d2 xv[K], av[R];
#pragma unroll
          for (int j = 0; j < K; ++j) xv[j] = *reinterpret_cast<const ccqp_d2*>(slot + (R + j) * PIECE + lane * 16);
#pragma unroll
          for (int r = 0; r < R; ++r) av[r] = *reinterpret_cast<const ccqp_d2*>(slot + r * PIECE + lane * 16);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
              acc[j][r] = fma(av[r].x, xv[j].x, acc[j][r]);
              acc[j][r] = fma(av[r].y, xv[j].y, acc[j][r]);
            }
          }
        }
        cslot = cslot + 1 == S ? 0 : cslot + 1;
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j][r] = ccqp_wave_sum(acc[j][r]);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (lane == r) {
#pragma unroll
          for (int j = 0; j < K; ++j) res[(kk * R + r) * K + j] = acc[j][r];
        }
    }
    // ---- flush: every DMA of the group has landed (the last steps waited for vmcnt(0))
    for (int i = lane; i < kn * R; i += 64) {
      const int kk = i / R, r = i - kk * R;
      const int64_t row = gemv_lds_first_row<R>(wave + (k0 + kk) * nwaves, rs) + r * rs;
      if (row < rows) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const double v = res[K * i + j];
          io.y[j][row] = v;
          if (io.g[j]) io.g[j][row] = v + b[row];     // `A.dot(x) + b`: one more rounding, as numpy
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
// ---- the same ring over MANY matrices of one size (the batched matvec of the lock-step forms): the wave's tasks are
// T = wave, wave + nwaves, ... over the concatenation of `nlive` matrices (task T = rows [R t, R t + R) of matrix T / tpp,
// t = T % tpp), so the 2 x 256 waves of a persistent grid work through ONE matrix after the other, close together.
// D (in LDS: descriptor reads must not be ordinary loads -- they would drain the ring) holds each matrix's operands.
template <int R, int S, bool NT>
__device__ __forceinline__ void gemv_lds_multi_body(const gemv_lds_desc* D, int nlive, int64_t n, int64_t wave,
                                                    int64_t nwaves, char* mine, double* res, int64_t rs) {
  constexpr int PIECE = 1024;
  constexpr int SLOT = (R + 1) * PIECE;
  constexpr int GROUP = GEMV_LDS_RESCAP / R;
  const int lane = threadIdx.x & 63;
  const int64_t nfull = n / 128;
  const int tail = (int)(n - nfull * 128);
  const bool mine_in_tail = 2 * lane < tail;
  const int64_t nsteps = nfull + (tail ? 1 : 0);
  const int64_t tpp = (n + R * rs - 1) / (R * rs) * rs;       // tasks per matrix (gemv_lds_tasks)
  const int64_t ntasks = (int64_t)nlive * tpp;
  const int64_t mytasks = wave < ntasks ? (ntasks - wave + nwaves - 1) / nwaves : 0;
  for (int64_t k0 = 0; k0 < mytasks; k0 += GROUP) {
    const int kn = (int)(mytasks - k0 < GROUP ? mytasks - k0 : GROUP);
    const int64_t total = (int64_t)kn * nsteps;
    int ik = 0, islot = 0;
    int64_t is = 0, issued = 0;
    const double* irp[R];
    const double* ixp;
    auto rows_of = [&](int k) {
      const int64_t T = wave + (k0 + k) * nwaves;
      const int64_t m = T / tpp, t = T - m * tpp;
      const gemv_lds_desc d = D[m];
      const int64_t r0 = gemv_lds_first_row<R>(t, rs);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int64_t row = r0 + r * rs;
        if (row > n - 1) row = n - 1;           // clamp: tail rows re-read a valid row, never stored
        irp[r] = d.A + row * d.lda + 2 * lane;
      }
      ixp = d.x + 2 * lane;
    };
    rows_of(0);
    auto issue_next = [&]() {
      char* slot = mine + islot * SLOT;
      if (is < nfull || mine_in_tail) {
#pragma unroll
        for (int r = 0; r < R; ++r)
          __builtin_amdgcn_global_load_lds(irp[r] + is * 128, (__attribute__((address_space(3))) void*)(slot + r * PIECE),
                                           16, 0, NT ? 2 : 0);
        __builtin_amdgcn_global_load_lds(ixp + is * 128, (__attribute__((address_space(3))) void*)(slot + R * PIECE), 16, 0,
                                         0);
      }
      islot = islot + 1 == S ? 0 : islot + 1;
      ++issued;
      if (++is == nsteps) {
        is = 0;
        if (++ik < kn) rows_of(ik);
      }
    };
    for (int p = 0; p < S - 1 && issued < total; ++p) issue_next();
    int cslot = 0;
    for (int kk = 0; kk < kn; ++kk) {
      double acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = 0.0;
      for (int64_t s = 0; s < nsteps; ++s) {
        if (issued < total) {
          issue_next();
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * (R + 1)) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char* slot = mine + cslot * SLOT;
        if (s < nfull || mine_in_tail) {
          // (all R + 1 pieces asked for back to back, then the fma chains: see gemv_ldsk_body)
          const ccqp_d2 xv = *reinterpret_cast<const ccqp_d2*>(slot + R * PIECE + lane * 16);
          ccqp_d2 av[R];
#pragma unroll
          for (int r = 0; r < R; ++r) av[r] = *reinterpret_cast<const ccqp_d2*>(slot + r * PIECE + lane * 16);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            acc[r] = fma(av[r].x, xv.x, acc[r]);
            acc[r] = fma(av[r].y, xv.y, acc[r]);
          }
        }
        cslot = cslot + 1 == S ? 0 : cslot + 1;
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = ccqp_wave_sum(acc[r]);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (lane == r) res[kk * R + r] = acc[r];
    }
    for (int j = lane; j < kn * R; j += 64) {
      const int kk = j / R, r = j - kk * R;
      const int64_t T = wave + (k0 + kk) * nwaves;
      const int64_t m = T / tpp, t = T - m * tpp;
      const int64_t row = gemv_lds_first_row<R>(t, rs) + r * rs;
      if (row < n) {
        const gemv_lds_desc d = D[m];
        const double v = res[j];
        d.y[row] = v;
        if (d.g) d.g[row] = v + d.b[row];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
// ---- up to THREE right-hand sides per matrix in one pass, over MANY matrices (gemv_lds_multi_body with the slot of
// gemv_lds2_body widened to three x pieces): y_j = A x_j for j < 3 (g_j = y_j + b where g_j != NULL).  A matrix with fewer
// products repeats a vector (its outputs NULL): the number of DMA loads per step is fixed (the waits are counted).
// Each product has the bits of its own single pass.  (Lock-step MPRGP: A dx, A x_{k+1} and A p_{k+1} of a CG step.)
template <int R, int S, bool NT>
__device__ __forceinline__ void gemv_lds3_multi_body(const gemv_lds3_desc* D, int nlive, int64_t n, int64_t wave,
                                                     int64_t nwaves, char* mine, double* res, int64_t rs) {
  constexpr int PIECE = 1024;
  constexpr int SLOT = (R + 3) * PIECE;
  constexpr int GROUP = GEMV_LDS_RESCAP / (3 * R);
  const int lane = threadIdx.x & 63;
  const int64_t nfull = n / 128;
  const int tail = (int)(n - nfull * 128);
  const bool mine_in_tail = 2 * lane < tail;
  const int64_t nsteps = nfull + (tail ? 1 : 0);
  const int64_t tpp = (n + R * rs - 1) / (R * rs) * rs;       // tasks per matrix (gemv_lds_tasks)
  const int64_t ntasks = (int64_t)nlive * tpp;
  const int64_t mytasks = wave < ntasks ? (ntasks - wave + nwaves - 1) / nwaves : 0;
  for (int64_t k0 = 0; k0 < mytasks; k0 += GROUP) {
    const int kn = (int)(mytasks - k0 < GROUP ? mytasks - k0 : GROUP);
    const int64_t total = (int64_t)kn * nsteps;
    int ik = 0, islot = 0;
    int64_t is = 0, issued = 0;
    const double* irp[R];
    const double* ixp[3];
    auto rows_of = [&](int k) {
      const int64_t T = wave + (k0 + k) * nwaves;
      const int64_t m = T / tpp, t = T - m * tpp;
      const gemv_lds3_desc& d = D[m];
      const int64_t r0 = gemv_lds_first_row<R>(t, rs);
      const double* A = d.A;
      const int64_t lda = d.lda;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int64_t row = r0 + r * rs;
        if (row > n - 1) row = n - 1;           // clamp: tail rows re-read a valid row, never stored
        irp[r] = A + row * lda + 2 * lane;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) ixp[j] = d.x[j] + 2 * lane;
    };
    rows_of(0);
    auto issue_next = [&]() {
      char* slot = mine + islot * SLOT;
      if (is < nfull || mine_in_tail) {
#pragma unroll
        for (int r = 0; r < R; ++r)
          __builtin_amdgcn_global_load_lds(irp[r] + is * 128, (__attribute__((address_space(3))) void*)(slot + r * PIECE),
                                           16, 0, NT ? 2 : 0);
#pragma unroll
        for (int j = 0; j < 3; ++j)
          __builtin_amdgcn_global_load_lds(ixp[j] + is * 128,
                                           (__attribute__((address_space(3))) void*)(slot + (R + j) * PIECE), 16, 0, 0);
      }
      islot = islot + 1 == S ? 0 : islot + 1;
      ++issued;
      if (++is == nsteps) {
        is = 0;
        if (++ik < kn) rows_of(ik);
      }
    };
    for (int p = 0; p < S - 1 && issued < total; ++p) issue_next();
    int cslot = 0;
    for (int kk = 0; kk < kn; ++kk) {
      double acc[3][R];
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[j][r] = 0.0;
      for (int64_t s = 0; s < nsteps; ++s) {
        if (issued < total) {
          issue_next();
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * (R + 3)) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char* slot = mine + cslot * SLOT;
        if (s < nfull || mine_in_tail) {
          ccqp_d2 xv[3], av[R];
#pragma unroll
          for (int j = 0; j < 3; ++j) xv[j] = *reinterpret_cast<const ccqp_d2*>(slot + (R + j) * PIECE + lane * 16);
#pragma unroll
          for (int r = 0; r < R; ++r) av[r] = *reinterpret_cast<const ccqp_d2*>(slot + r * PIECE + lane * 16);
          __builtin_amdgcn_sched_barrier(0);          // (reads back to back, then the fma chains: see gemv_ldsk_body)
#pragma unroll
          for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              acc[j][r] = fma(av[r].x, xv[j].x, acc[j][r]);
              acc[j][r] = fma(av[r].y, xv[j].y, acc[j][r]);
            }
          }
        }
        cslot = cslot + 1 == S ? 0 : cslot + 1;
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[j][r] = ccqp_wave_sum(acc[j][r]);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (lane == r) {
#pragma unroll
          for (int j = 0; j < 3; ++j) res[(kk * R + r) * 3 + j] = acc[j][r];
        }
    }
    for (int i = lane; i < kn * R; i += 64) {
      const int kk = i / R, r = i - kk * R;
      const int64_t T = wave + (k0 + kk) * nwaves;
      const int64_t m = T / tpp, t = T - m * tpp;
      const int64_t row = gemv_lds_first_row<R>(t, rs) + r * rs;
      if (row < n) {
        const gemv_lds3_desc& d = D[m];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const double v = res[3 * i + j];
          if (d.y[j]) d.y[j][row] = v;
          if (d.g[j]) d.g[j][row] = v + d.b[row];
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
#endif  // __HIP_DEVICE_COMPILE__

#endif  // __HIPCC__

// Default (R, U, workgroups per CU) of the matvec for a shape: host-side, shared by both launch forms.
static inline void ccqp_gemv_defaults(int64_t rows, int64_t cols, int* R, int* U, int* bpc) {
  const double elems = (double)rows * (double)cols;
  const bool huge = elems >= 8589934592.0;  // >= 64 GiB of matrix
  if (*R == 0) *R = (huge || cols >= 65536) ? 2 : 1;
  if (*U == 0) *U = (rows < 8192 && cols < 65536) ? 2 : 1;
  if (*bpc == 0) *bpc = huge ? 4 : 8;
}

// Whether the matvec is cut into launches that fill the machine once each (ccqp_gemv.hip: ROUNDS), by shape.
// Measured (MI355X, profiles/r02_gemv_rounds.jsonl): +2.4 % at 131072^2, +1.8 % at 65536 x 131072, +1 % at 65536^2,
// nothing at 32768^2 and below, nothing (bpc 8: -2 %) for 262144-column slabs.
static inline bool ccqp_gemv_rounds_default(int64_t rows, int64_t cols) {
  return (double)rows * (double)cols >= 4294967296.0 && cols <= 131072;
}
