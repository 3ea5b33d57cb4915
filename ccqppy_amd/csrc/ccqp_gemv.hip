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

// ------------------------------------------------------------------ MFMA form (A/B variant, bit 9 of the variant word)
// BASELINE config 3 words the matvec as "MFMA GEMV"; this is that kernel, kept selectable so the question is answered
// by a measurement (profiles/r02_gemv_mfma_ab.md) rather than by an argument.  One wave owns 16 consecutive rows.
// v_mfma_f64_16x16x4_f64 computes D(16x16) += Aop(16x4) . Bop(4x16); the matrix tile is the B operand,
//   Bop[k][j] = A[row0 + j][c(k)]      (lane l holds j = l & 15, k = l >> 4 -- one fp64 per lane),
// and x enters through ROW 0 of the A operand only (Aop[0][k] = x[c(k)], Aop[i > 0][k] = 0), so row 0 of D collects
// the 16 row sums: lanes 0..15, accumulator register 0 (f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4*reg).
// A single right-hand side uses 1/16 of the instruction's multiply-adds; that still sustains ~19 TB/s of matrix bytes
// (64 cycles per 512-byte tile and SIMD), i.e. the kernel stays HBM-bound like the VALU form.
// Loads: per 16-column step a lane issues two 16-byte non-temporal loads, columns c0 + 2k + {0,1} and c0 + 8 + 2k +
// {0,1}, so one wave instruction reads 64 contiguous bytes of each of its 16 rows (the VALU form reads 1 KiB of ONE
// row).  Summation order per row: columns in groups of four per MFMA (hardware order inside), groups in increasing
// column order -- NOT the lane-strided chain of the VALU form, so the last bits of y differ between the two forms
// (both within the fp64 bound 4*eps*sqrt(n)*sum|a_ij x_j| the tests state).  Needs cols % 16 == 0 and 16-byte
// aligned A / x / lda; other shapes take the VALU form.
typedef double ccqp_d4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(CCQP_BLOCK) void gemv_mfma_kernel(const double* __restrict__ A, int64_t lda,
                                                               const double* __restrict__ x,
                                                               const double* __restrict__ b, double* __restrict__ y,
                                                               double* __restrict__ g, int64_t rows, int64_t cols,
                                                               int64_t ntasks, const double* __restrict__ skip) {
  if (skip != nullptr && *skip != 0.0) return;
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  const int64_t wave = (int64_t)blockIdx.x * (CCQP_BLOCK / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (CCQP_BLOCK / 64);
  const bool xlane = j == 0;
  constexpr int64_t SPAN = 16 * U;
  for (int64_t task = wave; task < ntasks; task += nwaves) {
    int64_t row = task * 16 + j;
    if (row > rows - 1) row = rows - 1;           // clamp: tail rows re-read a valid row, never stored
    const double* rp = A + row * lda + 2 * kq;
    const double* xp = x + 2 * kq;
    ccqp_d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int64_t c0 = 0; c0 < cols; c0 += SPAN) {
      ccqp_d2 av[U][2], xv[U][2];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        av[u][0] = ccqp_ld2<true, true>(rp + c0 + 16 * u);
        av[u][1] = ccqp_ld2<true, true>(rp + c0 + 16 * u + 8);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        xv[u][0] = ccqp_ld2<true, false>(xp + c0 + 16 * u);
        xv[u][1] = ccqp_ld2<true, false>(xp + c0 + 16 * u + 8);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xlane ? xv[u][h].x : 0.0, av[u][h].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xlane ? xv[u][h].y : 0.0, av[u][h].y, acc, 0, 0, 0);
        }
    }
    // row 0 of D: lanes 0..15 hold y[task*16 + lane] in accumulator register 0
    const int64_t orow = task * 16 + lane;
    if (lane < 16 && orow < rows) {
      y[orow] = acc.x;
      if (g) g[orow] = acc.x + b[orow];
    }
  }
}

// ------------------------------------------------------------------ LDS-staged form (A/B variant, bit 24 of the variant word)
// north_star words the matvec as "coalesced HBM row reads, LDS-staged A-tiles"; this is that kernel, kept selectable
// so the phrase is answered by a measurement (profiles/r03_gemv_lds_ab.md) and not only by the guide's rule for GEMV
// ("operand streamed once and not shared across waves: load straight to VGPRs; an LDS round trip is pure overhead").
// One wave owns R consecutive rows, as in the VALU form.  Per 128-column step it issues R + 1 LDS-DMA loads
// (`global_load_lds_dwordx4`: 1 KiB of each row, non-temporal, and the matching 1 KiB of x out of L2) into one slot of
// its PRIVATE ring of S slots -- no VGPR destination, so S - 1 steps (S = 4: 9 KiB per wave) are in flight instead of
// one -- waits with a counted `s_waitcnt vmcnt`, reads the slot back with ds_read_b128 and runs the same fma chain:
// lane l still accumulates columns {128k + 2l, 128k + 2l + 1} in increasing k into one accumulator per row, hence the
// SAME BITS as the VALU form.  No barrier anywhere (a wave reads only what it has loaded itself).
// LDS: 4 waves x S x (R + 1) KiB = 48 KiB per workgroup at R = 2, S = 4 => 3 workgroups (12 waves) per CU.
// Needs cols % 128 == 0 and 16-byte aligned A / x / lda; other shapes take the VALU form.
template <int R, int S, bool NT>
__global__ __launch_bounds__(CCQP_BLOCK) void gemv_lds_kernel(const double* __restrict__ A, int64_t lda,
                                                              const double* __restrict__ x,
                                                              const double* __restrict__ b, double* __restrict__ y,
                                                              double* __restrict__ g, int64_t rows, int64_t cols,
                                                              int64_t ntasks, const double* __restrict__ skip) {
#if defined(__HIP_DEVICE_COMPILE__)             // (the host pass only needs the launch stub)
  if (skip != nullptr && *skip != 0.0) return;
  constexpr int PIECE = 1024;                   // bytes one wave instruction moves: 64 lanes x 16 B
  constexpr int SLOT = (R + 1) * PIECE;
  __shared__ __attribute__((aligned(16))) char ring[(CCQP_BLOCK / 64) * S * SLOT];
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (CCQP_BLOCK / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (CCQP_BLOCK / 64);
  char* const mine = ring + (threadIdx.x >> 6) * (S * SLOT);
  const int64_t nsteps = cols / 128;
  for (int64_t task = wave; task < ntasks; task += nwaves) {
    const int64_t r0 = task * R;
    const double* rp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t row = r0 + r;
      if (row > rows - 1) row = rows - 1;       // clamp: tail rows re-read a valid row, never stored
      rp[r] = A + row * lda + 2 * lane;
    }
    const double* xp = x + 2 * lane;
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    // every ds_read of the previous task has returned (its values fed the fma chain) before a slot is written again
    auto issue = [&](int64_t step) {
      char* slot = mine + (int)(step % S) * SLOT;
#pragma unroll
      for (int r = 0; r < R; ++r)
        __builtin_amdgcn_global_load_lds(rp[r] + step * 128, (__attribute__((address_space(3))) void*)(slot + r * PIECE),
                                         16, 0, NT ? 2 : 0);
      __builtin_amdgcn_global_load_lds(xp + step * 128, (__attribute__((address_space(3))) void*)(slot + R * PIECE), 16,
                                       0, 0);
    };
    for (int64_t p = 0; p < S - 1 && p < nsteps; ++p) issue(p);
    for (int64_t s = 0; s < nsteps; ++s) {
      if (s + S - 1 < nsteps) {
        issue(s + S - 1);
        // steps s+1 .. s+S-1 may stay in flight: (S - 1) * (R + 1) loads
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * (R + 1)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last S - 1 steps of a row: drain
      }
      const char* slot = mine + (int)(s % S) * SLOT;
      const ccqp_d2 xv = *reinterpret_cast<const ccqp_d2*>(slot + R * PIECE + lane * 16);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const ccqp_d2 av = *reinterpret_cast<const ccqp_d2*>(slot + r * PIECE + lane * 16);
        acc[r] = fma(av.x, xv.x, acc[r]);
        acc[r] = fma(av.y, xv.y, acc[r]);
      }
      asm volatile("" ::: "memory");            // the slot's reads stay ahead of the DMA that refills it
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = ccqp_wave_sum(acc[r]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = r0 + r;
      if (lane == r && row < rows) {
        y[row] = acc[r];
        if (g) g[row] = acc[r] + b[row];
      }
    }
  }
#endif
}

static bool gemv_lds_applicable(const double* A, int64_t lda, const double* x, int64_t cols) {
  return ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 && (lda % 2) == 0 &&
         cols > 0 && (cols % 128) == 0;
}

#define GEMV_LDS_R 2
#define GEMV_LDS_BPC 3                          // workgroups per CU the 48 KiB ring allows

static void launch_lds(int64_t blocks, bool nt, hipStream_t s, const double* A, int64_t lda, const double* x,
                       const double* b, double* y, double* g, int64_t rows, int64_t cols, const double* skip) {
  constexpr int R = GEMV_LDS_R;
  const int64_t ntasks = (rows + R - 1) / R;
  if (nt)
    hipLaunchKernelGGL((gemv_lds_kernel<R, 4, true>), dim3((unsigned)blocks), dim3(CCQP_BLOCK), 0, s, A, lda, x, b, y, g,
                       rows, cols, ntasks, skip);
  else
    hipLaunchKernelGGL((gemv_lds_kernel<R, 4, false>), dim3((unsigned)blocks), dim3(CCQP_BLOCK), 0, s, A, lda, x, b, y,
                       g, rows, cols, ntasks, skip);
}

static bool gemv_mfma_applicable(const double* A, int64_t lda, const double* x, int64_t cols) {
  return ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 && (lda % 2) == 0 &&
         cols > 0 && (cols % 64) == 0;
}

static void launch_mfma(ccqp_ctx* ctx, int U, int bpc, hipStream_t s, const double* A, int64_t lda, const double* x,
                        const double* b, double* y, double* g, int64_t rows, int64_t cols, const double* skip) {
  const int64_t ntasks = (rows + 15) / 16;
  int64_t blocks = (ntasks + (CCQP_BLOCK / 64) - 1) / (CCQP_BLOCK / 64);
  const int64_t cap = (int64_t)ctx->num_cu * (bpc > 0 ? bpc : 4);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  if (U == 4)
    hipLaunchKernelGGL((gemv_mfma_kernel<4>), dim3((unsigned)blocks), dim3(CCQP_BLOCK), 0, s, A, lda, x, b, y, g, rows,
                       cols, ntasks, skip);
  else if (U == 2)
    hipLaunchKernelGGL((gemv_mfma_kernel<2>), dim3((unsigned)blocks), dim3(CCQP_BLOCK), 0, s, A, lda, x, b, y, g, rows,
                       cols, ntasks, skip);
  else
    hipLaunchKernelGGL((gemv_mfma_kernel<1>), dim3((unsigned)blocks), dim3(CCQP_BLOCK), 0, s, A, lda, x, b, y, g, rows,
                       cols, ntasks, skip);
}

// variant word: bits 0-3 R (rows per wave: 1,2,4,8), bits 4-7 U (unroll: 1,2,4), bit 8 = 1 disables
// the non-temporal A loads (matrices of at most 192 MiB are loaded cacheably anyway), bit 9 = 1 selects the MFMA form (16 rows per wave; U = columns/16 per step: 1, 2, 4;
// shapes it does not cover fall through to the VALU form), bits 10-11 / 20-23 the launch rounds (below), bits 12-19
// workgroups per CU (0 = default), bit 24 = 1 selects the LDS-staged form (same bits as the default).  0 = tuned default
// (interleaved sweep on MI355X: tools/gemv_sweep.py, profiles/r01_gemv_sweep.md).  Every choice
// yields the same bits, so this is speed only.
struct gemv_shape {
  int R, U, bpc;
  bool nt, rounds;
  int64_t cap, round_rows;
};

static gemv_shape gemv_decide(const ccqp_ctx* ctx, int64_t rows, int64_t cols, int variant) {
  gemv_shape d;
  d.R = variant & 0xF;
  d.U = (variant >> 4) & 0xF;
  // non-temporal matrix loads keep a big stream out of the caches (+10 % at 128 GiB); a matrix that fits the 256 MiB
  // of MALL is re-read from there by the next matvec of a solve if it was loaded cacheably: 5000^2 36.4 -> 30.9 us,
  // 4096^2 23.6 -> 22.6 us, 8192^2 (512 MiB) 82 us non-temporal vs 91 us cacheable (sweep of 2026-10-04)
  d.nt = !((variant >> 8) & 1) && (double)rows * (double)cols * 8.0 > 192.0 * 1048576.0;
  d.bpc = (variant >> 12) & 0xFF;
  // ROUNDS (bits 10-11: 1 = on, 2 = off, 0 = by shape; bits 20-23: tasks per wave and round, 0 = 1): the rows are
  // cut into blocks that fill the machine exactly once -- every wave takes its share of ONE block, and the next
  // block is the next launch on the stream.  A single persistent launch lets the waves drift apart over a 128 GiB
  // matrix (each strides over the tasks at its own pace); restarting them together every few milliseconds keeps the
  // window of rows in flight compact: 19.5 instead of 19.9-20.0 ms at n = 131072, 9.77 instead of 9.95 ms for a
  // 65536 x 131072 slab, nothing to gain below 32 GiB or for 2 MiB rows (profiles/r02_gemv_rounds.jsonl).
  // The bits of y do not depend on it (they are a function of the row and of n alone).
  const int rounds_mode = (variant >> 10) & 3;
  const bool rounds_shape = ccqp_gemv_rounds_default(rows, cols);
  if (d.bpc == 0 && rounds_mode == 0 && rounds_shape) d.bpc = 8;   // (the persistent form's default for >= 64 GiB: 4)
  ccqp_gemv_defaults(rows, cols, &d.R, &d.U, &d.bpc);
  d.cap = (int64_t)ctx->num_cu * d.bpc;
  int per_wave = (variant >> 20) & 0xF;
  if (per_wave == 0) per_wave = 1;
  d.round_rows = (int64_t)d.R * (CCQP_BLOCK / 64) * d.cap * per_wave;
  d.rounds = (rounds_mode == 1 || (rounds_mode == 0 && rounds_shape)) && rows > d.round_rows;
  return d;
}

// the LDS-staged form has its own rows per wave and workgroups per CU; launch rounds as the default
static int gemv_lds_variant(int variant) {
  int bpc = (variant >> 12) & 0xFF;
  if (bpc == 0 || bpc > GEMV_LDS_BPC) bpc = GEMV_LDS_BPC;
  return (variant & ~0xFF00F) | GEMV_LDS_R | (bpc << 12);
}

extern "C" int ccqp_gemv_plan(ccqp_ctx* ctx, int64_t rows, int64_t cols, int variant, int64_t* rows_per_launch,
                              int64_t* launches) {
  CCQP_REQUIRE(ctx != nullptr && rows_per_launch != nullptr && launches != nullptr, "NULL argument");
  CCQP_REQUIRE(rows >= 0 && cols >= 0, "negative size");
  if (((variant >> 24) & 1) && cols > 0 && cols % 128 == 0) variant = gemv_lds_variant(variant);   // (aligned operands assumed)
  const gemv_shape d = gemv_decide(ctx, rows, cols, variant);
  *rows_per_launch = d.rounds ? d.round_rows : rows;
  *launches = d.rounds ? (rows + d.round_rows - 1) / d.round_rows : (rows > 0 ? 1 : 0);
  return CCQP_OK;
}

int ccqp_gemv_launch(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, const double* b,
                     double* y, double* g, int64_t rows, int64_t cols, int variant, hipStream_t s,
                     const double* skip) {
  if (((variant >> 9) & 1) && gemv_mfma_applicable(A, lda, x, cols)) {
    const int U0 = (variant >> 4) & 0xF;
    launch_mfma(ctx, U0 == 0 ? 4 : U0, (variant >> 12) & 0xFF, s, A, lda, x, b, y, g, rows, cols, skip);
    return CCQP_OK;
  }
  const bool lds = ((variant >> 24) & 1) && gemv_lds_applicable(A, lda, x, cols);
  if (lds) variant = gemv_lds_variant(variant);
  const gemv_shape d = gemv_decide(ctx, rows, cols, variant);
  const int R = d.R, U = d.U;
  const bool nt = d.nt, rounds = d.rounds;
  const int64_t cap = d.cap, round_rows = d.round_rows;
  const bool aligned = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 &&
                       (lda % 2) == 0;
  int bad = 0;
  for (int64_t r0 = 0; r0 < rows && !bad; r0 += rounds ? round_rows : rows) {
    const int64_t nr = rounds && rows - r0 > round_rows ? round_rows : rows - r0;
    const int64_t ntasks = (nr + R - 1) / R;
    int64_t blocks = (ntasks + (CCQP_BLOCK / 64) - 1) / (CCQP_BLOCK / 64);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const double* Ar = A + r0 * lda;
    const double* br = b ? b + r0 : nullptr;
    double* yr = y + r0;
    double* gr = g ? g + r0 : nullptr;
    if (lds) {
      launch_lds(blocks, nt, s, Ar, lda, x, br, yr, gr, nr, cols, skip);
      continue;
    }
    switch (R) {
      case 1: bad = launch_r<1>(U, aligned, nt, s, (int)blocks, Ar, lda, x, br, yr, gr, nr, cols, skip); break;
      case 2: bad = launch_r<2>(U, aligned, nt, s, (int)blocks, Ar, lda, x, br, yr, gr, nr, cols, skip); break;
      case 4: bad = launch_r<4>(U, aligned, nt, s, (int)blocks, Ar, lda, x, br, yr, gr, nr, cols, skip); break;
      case 8: bad = launch_r<8>(U, aligned, nt, s, (int)blocks, Ar, lda, x, br, yr, gr, nr, cols, skip); break;
      default: bad = 1;
    }
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
