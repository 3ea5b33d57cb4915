// Dense fp64 matvec y = A.x for gfx950 (MI355X): the >96 % hot spot of every CCQP solver
// (reference: `A.dot(v)` at solvers.py:133,148,275,290,625,641,922,938,1086,1100 ...).
//
// Roofline: 2n^2 flop over 8n^2 bytes = 0.25 flop/B  =>  HBM-bound by a factor ~40 on MI355X
// (8 TB/s vs 78.6 TFLOP/s fp64).  The kernel is therefore a pure streaming design:
//   * one wave64 owns R consecutive rows ("task"); lanes stride across the columns with 16-byte
//     loads, so every wave-instruction reads one contiguous 1 KiB piece of a row (fully coalesced);
//   * A goes HBM -> VGPR directly (an LDS round trip would be pure overhead for data used once);
//     it is loaded non-temporally so the stream does not evict x from the XCD's L2;
//   * x (n doubles = 1 MiB at n=131072) is re-read by every task and stays L2-resident (4 MiB/XCD);
//     R rows share each x load, so x traffic out of L2 is 1/R of the A stream;
//   * per-lane partial sums live in registers; a __shfl_xor butterfly folds the 64 lanes.
// Determinism: lane l accumulates columns {128k+2l, 128k+2l+1 : k = 0,1,..} in increasing k with
// one fma per element into ONE accumulator per row, so the bits of y[i] depend on (row i, cols)
// only -- not on R, the unroll U, the grid, the load width or how rows are sharded over GPUs.
// The body lives in ccqp_bodies.h (shared with the batched launch form of ccqp_batch.hip).
#include "ccqp_bodies.h"

template <int R, int U, bool ALIGNED, bool NT>
__global__ __launch_bounds__(CCQP_BLOCK) void gemv_stream_kernel(
    const double* __restrict__ A, int64_t lda, const double* __restrict__ x,
    const double* __restrict__ b, double* __restrict__ y, double* __restrict__ g,
    int64_t rows, int64_t cols, int64_t ntasks, const double* __restrict__ skip) {
  // pipelined solves (ccqp_pipe_*): launches queued behind a converged iteration find the flag set and leave
  if (skip != nullptr && *skip != 0.0) return;
  const int64_t wave = (int64_t)blockIdx.x * (CCQP_BLOCK / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (CCQP_BLOCK / 64);
  gemv_body<R, U, ALIGNED, NT>(A, lda, x, b, y, g, rows, cols, ntasks, wave, nwaves);
}

template <int R, int U, bool ALIGNED, bool NT>
static void launch_one(hipStream_t s, int grid, const double* A, int64_t lda, const double* x,
                       const double* b, double* y, double* g, int64_t rows, int64_t cols, const double* skip) {
  const int64_t ntasks = (rows + R - 1) / R;
  hipLaunchKernelGGL((gemv_stream_kernel<R, U, ALIGNED, NT>), dim3(grid), dim3(CCQP_BLOCK), 0, s, A, lda,
                     x, b, y, g, rows, cols, ntasks, skip);
}

template <int R, int U>
static void launch_ru(bool aligned, bool nt, hipStream_t s, int grid, const double* A, int64_t lda,
                      const double* x, const double* b, double* y, double* g, int64_t rows, int64_t cols,
                      const double* skip) {
  if (aligned) {
    if (nt) launch_one<R, U, true, true>(s, grid, A, lda, x, b, y, g, rows, cols, skip);
    else launch_one<R, U, true, false>(s, grid, A, lda, x, b, y, g, rows, cols, skip);
  } else {
    if (nt) launch_one<R, U, false, true>(s, grid, A, lda, x, b, y, g, rows, cols, skip);
    else launch_one<R, U, false, false>(s, grid, A, lda, x, b, y, g, rows, cols, skip);
  }
}

template <int R>
static int launch_r(int U, bool aligned, bool nt, hipStream_t s, int grid, const double* A, int64_t lda,
                    const double* x, const double* b, double* y, double* g, int64_t rows, int64_t cols,
                    const double* skip) {
  switch (U) {
    case 1: launch_ru<R, 1>(aligned, nt, s, grid, A, lda, x, b, y, g, rows, cols, skip); return 0;
    case 2: launch_ru<R, 2>(aligned, nt, s, grid, A, lda, x, b, y, g, rows, cols, skip); return 0;
    case 4: launch_ru<R, 4>(aligned, nt, s, grid, A, lda, x, b, y, g, rows, cols, skip); return 0;
    default: return 1;
  }
}

// variant word: bits 0-3 R (rows per wave: 1,2,4,8), bits 4-7 U (unroll: 1,2,4), bit 8 = 1 disables
// the non-temporal A loads, bits 12-19 workgroups per CU (0 = default).  0 = tuned default
// (interleaved sweep on MI355X: tools/gemv_sweep.py, profiles/r01_gemv_sweep.md).  Every choice
// yields the same bits, so this is speed only.
int ccqp_gemv_launch(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, const double* b,
                     double* y, double* g, int64_t rows, int64_t cols, int variant, hipStream_t s,
                     const double* skip) {
  int R = variant & 0xF, U = (variant >> 4) & 0xF;
  const bool nt = !((variant >> 8) & 1);
  int bpc = (variant >> 12) & 0xFF;
  ccqp_gemv_defaults(rows, cols, &R, &U, &bpc);
  const bool aligned = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 &&
                       (lda % 2) == 0;
  const int64_t ntasks = (rows + R - 1) / R;
  int64_t blocks = (ntasks + (CCQP_BLOCK / 64) - 1) / (CCQP_BLOCK / 64);
  const int64_t cap = (int64_t)ctx->num_cu * bpc;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  int bad = 0;
  switch (R) {
    case 1: bad = launch_r<1>(U, aligned, nt, s, (int)blocks, A, lda, x, b, y, g, rows, cols, skip); break;
    case 2: bad = launch_r<2>(U, aligned, nt, s, (int)blocks, A, lda, x, b, y, g, rows, cols, skip); break;
    case 4: bad = launch_r<4>(U, aligned, nt, s, (int)blocks, A, lda, x, b, y, g, rows, cols, skip); break;
    case 8: bad = launch_r<8>(U, aligned, nt, s, (int)blocks, A, lda, x, b, y, g, rows, cols, skip); break;
    default: bad = 1;
  }
  if (bad) {
    ccqp_set_error("ccqp_gemv_f64: unsupported variant 0x%x (R=%d, U=%d)", variant, R, U);
    return CCQP_ERR_UNSUPPORTED;
  }
  return CCQP_OK;
}

static int gemv_check(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, double* y,
                      int64_t rows, int64_t cols) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(rows >= 0 && cols >= 0, "negative size");
  CCQP_REQUIRE(lda >= cols, "lda < cols");
  CCQP_REQUIRE(rows == 0 || (A != nullptr && y != nullptr), "NULL matrix/output");
  CCQP_REQUIRE(cols == 0 || rows == 0 || x != nullptr, "NULL x");
  CCQP_REQUIRE(y != x, "y must not alias x");
  return CCQP_OK;
}

extern "C" int ccqp_gemv_f64(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, double* y,
                             int64_t rows, int64_t cols, int variant, void* stream) {
  int rc = gemv_check(ctx, A, lda, x, y, rows, cols);
  if (rc) return rc;
  if (rows == 0) return CCQP_OK;
  rc = ccqp_gemv_launch(ctx, A, lda, x, nullptr, y, nullptr, rows, cols, variant, (hipStream_t)stream, nullptr);
  if (rc) return rc;
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_gemv_epilogue_f64(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x,
                                      const double* b, double* y, double* g, int64_t rows, int64_t cols,
                                      int variant, void* stream) {
  int rc = gemv_check(ctx, A, lda, x, y, rows, cols);
  if (rc) return rc;
  CCQP_REQUIRE(g == nullptr || b != nullptr, "g requested but b is NULL");
  CCQP_REQUIRE(g == nullptr || g != x, "g must not alias x");
  if (rows == 0) return CCQP_OK;
  rc = ccqp_gemv_launch(ctx, A, lda, x, b, y, g, rows, cols, variant, (hipStream_t)stream, nullptr);
  if (rc) return rc;
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}
