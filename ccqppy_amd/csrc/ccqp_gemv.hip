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
// north_star words the matvec as "coalesced HBM row reads, LDS-staged A-tiles"; this is that kernel (body and full
// description: gemv_lds_body, ccqp_bodies.h): the rows travel through a per-wave LDS ring by LDS-DMA loads, many steps in
// flight, continuously across the wave's tasks -- same lane -> column map and fma chain, hence the SAME BITS as the
// VALU form.  The guide's rule for GEMV is "load straight to VGPRs; an LDS round trip is pure overhead"; measured here
// (profiles/r03_gemv_lds_ab.md) the deep ring with ONE workgroup per CU reads the headline matrix FASTER than any
// VGPR form (fewer, longer row streams), which is why it is selectable and, where it wins, the default.
// LDS: waves x (S x (R + 1) KiB ring + 4 KiB of parked row sums) per workgroup.
template <int R, int S, bool NT>
__global__ __launch_bounds__(CCQP_BLOCK) void gemv_lds_kernel(const double* __restrict__ A, int64_t lda,
                                                              const double* __restrict__ x,
                                                              const double* __restrict__ b, double* __restrict__ y,
                                                              double* __restrict__ g, int64_t rows, int64_t cols,
                                                              int64_t ntasks, const double* __restrict__ skip,
                                                              int stagger, int64_t rs) {
#if defined(__HIP_DEVICE_COMPILE__)             // (the host pass only needs the launch stub)
  if (skip != nullptr && *skip != 0.0) return;
  // dynamic LDS: the four waves' rings, then their parked row sums
  extern __shared__ __attribute__((aligned(16))) char ring[];
  constexpr int RING = S * (R + 1) * 1024;
  // (readfirstlane: the wave's task list, step counters and branches become scalar code -- see gemv_lds2_kernel)
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;   // (1, 2 or 4 waves per workgroup)
  const int64_t wave = (int64_t)blockIdx.x * wpb + w;
  const int64_t nwaves = (int64_t)gridDim.x * wpb;
  double* res = reinterpret_cast<double*>(ring + wpb * RING) + w * GEMV_LDS_RESCAP;
  // staggered start: every wave of the machine walks the SAME x (and, row for row, the same column positions) at the
  // same pace, so waves launched together ask the same L2 lines and DRAM pages at the same moments; a start delay of
  // (37 wave mod 64) x stagger x 64 clocks (at most 7.7 us at stagger 4) spreads them: +0.5 ... 1 % on the pure read
  // with the x stream beside it (tools/readbw.py) and on the matvec, at any size
  for (int i = (int)((wave * 37) & 63) * stagger; i > 0; --i) __builtin_amdgcn_s_sleep(1);
  gemv_lds_body<R, S, NT>(A, lda, x, b, y, g, rows, cols, ntasks, wave, nwaves, ring + w * RING, res, rs);
#endif
}

static bool gemv_lds_applicable(const double* A, int64_t lda, const double* x, int64_t cols) {
  return ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 && (lda % 2) == 0 &&
         cols >= 128 && (cols % 2) == 0;
}

// ring shapes selectable by the variant word: bits 25-27 slots S = 4 (code + 1), bits 28-29 rows per wave R = 2, 4, 1, 8
// (codes 0 .. 3), bit 30 / bit 23: 2 / 1 waves per workgroup instead of 4; LDS per workgroup = waves x (S x (R + 1) KiB
// + 4 KiB), at most 160 KiB per CU -- a ring that does not fit (or is not instantiated) is shortened to the next one
struct gemv_lds_shape {
  int R, S, bpc_max, wpb;      // wpb: waves per workgroup
  int stagger;                 // start delay unit of the waves (bits 20-22: 0 = by size, 1 = none, 2 .. 6 = 1, 4, 16, 32, 64)
  size_t lds;
};

static gemv_lds_shape gemv_lds_decode(int variant) {
  gemv_lds_shape h;
  static const int rows_of[4] = {2, 4, 1, 8};
  h.R = rows_of[(variant >> 28) & 3];
  h.wpb = ((variant >> 23) & 1) ? 1 : ((variant >> 30) & 1) ? 2 : CCQP_BLOCK / 64;
  static const int stagger_of[8] = {-1, 0, 1, 4, 16, 32, 64, 64};
  h.stagger = stagger_of[(variant >> 20) & 7];     // (-1: chosen by the launch from the bytes it reads)
  const int want = 4 * (((variant >> 25) & 7) + 1);
  static const int have[4][3] = {{12, 8, 4}, {8, 4, 4}, {16, 8, 4}, {4, 4, 4}};   // instantiated depths, by R code
  const int* row = have[(variant >> 28) & 3];
  h.S = 4;
  for (int i = 0; i < 3; ++i) {
    const size_t need = (size_t)h.wpb * ((size_t)row[i] * (h.R + 1) * 1024 + GEMV_LDS_RESCAP * sizeof(double));
    if (row[i] <= want && need <= 160 * 1024) {
      h.S = row[i];
      break;
    }
  }
  h.lds = (size_t)h.wpb * ((size_t)h.S * (h.R + 1) * 1024 + GEMV_LDS_RESCAP * sizeof(double));
  h.bpc_max = (int)((160 * 1024) / h.lds);
  if (h.bpc_max < 1) h.bpc_max = 1;
  return h;
}

template <int R, int S>
static int launch_lds_rs(ccqp_ctx* ctx, int64_t blocks, bool nt, size_t lds, int wpb, int stagger, hipStream_t s,
                         const double* A, int64_t lda, const double* x, const double* b, double* y, double* g, int64_t rows,
                         int64_t cols, const double* skip) {
  const int64_t rs = gemv_lds_row_stride(ccqp_opt(ctx, CCQP_OPT_GEMV_ROW_STRIDE), lda, rows, R);
  const int64_t ntasks = gemv_lds_tasks(rows, R, rs);
  if (stagger < 0) {
    // by size: the longest delay (63 x stagger x 64 clocks ~ 1.9 us x stagger) stays near 0.3 % of the launch's duration
    const double gib4 = 8.0 * (double)rows * (double)cols / 4294967296.0;
    stagger = gib4 < 1.0 ? 1 : gib4 > 32.0 ? 32 : (int)(gib4 + 0.5);
  }
  // (more than 64 KiB of dynamic LDS has to be asked for, per kernel and device: ccqp_lds_raise)
  if (ccqp_lds_raise(ctx, reinterpret_cast<const void*>(nt ? gemv_lds_kernel<R, S, true> : gemv_lds_kernel<R, S, false>), lds))
    return 1;
  if (nt)
    hipLaunchKernelGGL((gemv_lds_kernel<R, S, true>), dim3((unsigned)blocks), dim3(64 * wpb), lds, s, A, lda, x, b, y,
                       g, rows, cols, ntasks, skip, stagger, rs);
  else
    hipLaunchKernelGGL((gemv_lds_kernel<R, S, false>), dim3((unsigned)blocks), dim3(64 * wpb), lds, s, A, lda, x, b, y,
                       g, rows, cols, ntasks, skip, stagger, rs);
  return 0;
}

static int launch_lds(ccqp_ctx* ctx, const gemv_lds_shape& h, int64_t blocks, bool nt, hipStream_t s, const double* A,
                      int64_t lda, const double* x, const double* b, double* y, double* g, int64_t rows, int64_t cols,
                      const double* skip) {
#define LDS_CASE(RR, SS) \
  if (h.R == RR && h.S == SS)   \
  return launch_lds_rs<RR, SS>(ctx, blocks, nt, h.lds, h.wpb, h.stagger, s, A, lda, x, b, y, g, rows, cols, skip)
  LDS_CASE(1, 4);
  LDS_CASE(1, 8);
  LDS_CASE(1, 16);
  LDS_CASE(2, 4);
  LDS_CASE(2, 8);
  LDS_CASE(2, 12);
  LDS_CASE(4, 4);
  LDS_CASE(4, 8);
  LDS_CASE(8, 4);
#undef LDS_CASE
  return 1;
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
  const gemv_lds_shape h = gemv_lds_decode(variant);
  int bpc = (variant >> 12) & 0xFF;
  if (bpc == 0 || bpc > h.bpc_max) bpc = h.bpc_max;
  return (variant & ~0xFF00F) | h.R | (bpc << 12);
}

// ---- which form a matvec takes.  variant == 0 (the tuned default): matrices of 2^30 elements (8 GiB) or more whose
// shape the LDS-staged form covers take IT -- one persistent launch, one 2-wave workgroup per CU, GEMV_AUTO_LDS below,
// a wave's rows about 1 MiB apart.  Measured against the VALU form on the same box (profiles/r03_gemv_lds_ab.md,
// r03_gemv_threshold.jsonl): +1.4 ... +4.5 % from 16 GiB slabs to the 128 GiB headline matrix; at 8 GiB (with the row
// spacing) 32768^2 +1.3 %, 65536 x 16384 +1 %, 16384 x 65536 +2.6 %; at 4.5 GiB (24576^2) -5 % and at 2 GiB -3 %: the
// persistent grid's ramp.  (variant & 0xF) == 0xF forces the VALU form with its shape defaults (the A/B switch).  Same
// bits whichever form runs.
#define GEMV_AUTO_LDS ((1 << 24) | (1 << 28) | (0 << 25) | (1 << 12) | (2 << 10) | (1 << 30))   // R = 4, S = 4, 2 waves, 1 wg/CU
static int gemv_resolve(int variant, int64_t rows, int64_t cols, bool lds_ok) {
  if ((variant & 0xF) == 0xF) return variant & ~0xF;
  if (variant == 0 && lds_ok && (double)rows * (double)cols >= 1073741824.0) return GEMV_AUTO_LDS;
  return variant;
}

extern "C" int ccqp_gemv_form(ccqp_ctx* ctx, int64_t rows, int64_t cols, int variant, char* name, int buflen) {
  CCQP_REQUIRE(ctx != nullptr && name != nullptr && buflen > 0, "NULL argument");
  const bool lds_ok = cols >= 128 && cols % 2 == 0;             // (16-byte aligned operands assumed)
  variant = gemv_resolve(variant, rows, cols, lds_ok);
  const char* k = ((variant >> 24) & 1) && lds_ok ? "gemv_lds_kernel"
                  : ((variant >> 9) & 1) && cols % 64 == 0 ? "gemv_mfma_kernel" : "gemv_stream_kernel";
  strncpy(name, k, (size_t)buflen - 1);
  name[buflen - 1] = 0;
  return CCQP_OK;
}

extern "C" int ccqp_gemv_plan(ccqp_ctx* ctx, int64_t rows, int64_t cols, int variant, int64_t* rows_per_launch,
                              int64_t* launches) {
  CCQP_REQUIRE(ctx != nullptr && rows_per_launch != nullptr && launches != nullptr, "NULL argument");
  CCQP_REQUIRE(rows >= 0 && cols >= 0, "negative size");
  variant = gemv_resolve(variant, rows, cols, cols >= 128 && cols % 2 == 0);                   // (aligned operands assumed)
  if (((variant >> 24) & 1) && cols >= 128 && cols % 2 == 0) variant = gemv_lds_variant(variant);
  const gemv_shape d = gemv_decide(ctx, rows, cols, variant);
  *rows_per_launch = d.rounds ? d.round_rows : rows;
  *launches = d.rounds ? (rows + d.round_rows - 1) / d.round_rows : (rows > 0 ? 1 : 0);
  return CCQP_OK;
}

int ccqp_gemv_launch(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, const double* b,
                     double* y, double* g, int64_t rows, int64_t cols, int variant, hipStream_t s,
                     const double* skip) {
  variant = gemv_resolve(variant, rows, cols, gemv_lds_applicable(A, lda, x, cols));
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
      const gemv_lds_shape h = gemv_lds_decode(variant);
      int64_t lb = (ntasks + h.wpb - 1) / h.wpb;
      if (lb > cap) lb = cap;
      if (lb < 1) lb = 1;
      bad = launch_lds(ctx, h, lb, nt, s, Ar, lda, x, br, yr, gr, nr, cols, skip);
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

// ---- two (three) right-hand sides in one pass over A (gemv_ldsk_body, ccqp_bodies.h): the LDS-staged form with all x
// pieces in the ring's slot.  Waves per workgroup, workgroups per CU, stagger and row stride as the single form's.
// (up to EIGHT waves per workgroup: launch bounds 512; the parked row sums take `rescap` doubles per wave)
template <int R, int S, bool NT>
__global__ __launch_bounds__(512) void gemv_lds2_kernel(const double* __restrict__ A, int64_t lda,
                                                        const double* __restrict__ x1, const double* __restrict__ x2,
                                                        const double* __restrict__ b, double* __restrict__ y1,
                                                        double* __restrict__ g1, double* __restrict__ y2,
                                                        double* __restrict__ g2, int64_t rows, int64_t cols, int64_t ntasks,
                                                        int stagger, int64_t rs, int rescap) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char ring[];
  constexpr int RING = S * (R + 2) * 1024;
  // (the wave's index through readfirstlane: everything derived from it -- task list, step counters, ring slots, the
  // loop's branches -- is then SCALAR code; as a function of threadIdx it would be vector arithmetic and exec-mask
  // branches, ~2x the instructions of a step on the one wave a SIMD has here)
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * wpb + w;
  const int64_t nwaves = (int64_t)gridDim.x * wpb;
  double* res = reinterpret_cast<double*>(ring + wpb * RING) + w * rescap;
  for (int i = (int)((wave * 37) & 63) * stagger; i > 0; --i) __builtin_amdgcn_s_sleep(1);     // (staggered start)
  gemv_ldsk_io<2> io;
  io.x[0] = x1, io.x[1] = x2, io.y[0] = y1, io.y[1] = y2, io.g[0] = g1, io.g[1] = g2;
  gemv_ldsk_body<R, S, 2, NT>(A, lda, io, b, rows, cols, ntasks, wave, nwaves, ring + w * RING, res, rs, rescap);
#endif
}

// three right-hand sides (a rank's row block under MPRGP's look-ahead, RowShardedOperator.matvec3)
template <int R, int S, bool NT>
__global__ __launch_bounds__(CCQP_BLOCK) void gemv_lds3_kernel(const double* __restrict__ A, int64_t lda,
                                                               const gemv_ldsk_io<3> io, const double* __restrict__ b,
                                                               int64_t rows, int64_t cols, int64_t ntasks, int stagger,
                                                               int64_t rs) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char ring[];
  constexpr int RING = S * (R + 3) * 1024;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * wpb + w;
  const int64_t nwaves = (int64_t)gridDim.x * wpb;
  double* res = reinterpret_cast<double*>(ring + wpb * RING) + w * GEMV_LDS_RESCAP;
  for (int i = (int)((wave * 37) & 63) * stagger; i > 0; --i) __builtin_amdgcn_s_sleep(1);     // (staggered start)
  gemv_ldsk_body<R, S, 3, NT>(A, lda, io, b, rows, cols, ntasks, wave, nwaves, ring + w * RING, res, rs, GEMV_LDS_RESCAP);
#endif
}

template <int R, int S>
static int launch_lds3(ccqp_ctx* ctx, int wpb, bool nt, int stagger, hipStream_t s, const double* A, int64_t lda,
                       const gemv_ldsk_io<3>& io, const double* b, int64_t rows, int64_t cols) {
  const size_t lds = (size_t)wpb * ((size_t)S * (R + 3) * 1024 + GEMV_LDS_RESCAP * sizeof(double));
  const int64_t rs = gemv_lds_row_stride(ccqp_opt(ctx, CCQP_OPT_GEMV_ROW_STRIDE), lda, rows, R);
  const int64_t ntasks = gemv_lds_tasks(rows, R, rs);
  int64_t blocks = (ntasks + wpb - 1) / wpb;
  if (blocks > (int64_t)ctx->num_cu) blocks = ctx->num_cu;
  if (blocks < 1) blocks = 1;
  const int rc = ccqp_lds_raise(
      ctx, reinterpret_cast<const void*>(nt ? gemv_lds3_kernel<R, S, true> : gemv_lds3_kernel<R, S, false>), lds);
  if (rc) return rc;
  if (nt)
    hipLaunchKernelGGL((gemv_lds3_kernel<R, S, true>), dim3((unsigned)blocks), dim3(64 * wpb), lds, s, A, lda, io, b, rows, cols,
                       ntasks, stagger, rs);
  else
    hipLaunchKernelGGL((gemv_lds3_kernel<R, S, false>), dim3((unsigned)blocks), dim3(64 * wpb), lds, s, A, lda, io, b, rows, cols,
                       ntasks, stagger, rs);
  return CCQP_OK;
}

template <int R, int S>
static int launch_lds2(ccqp_ctx* ctx, int wpb, int bpc, bool nt, int stagger, int rescap, hipStream_t s, const double* A,
                       int64_t lda, const double* x1, const double* x2, const double* b, double* y1, double* g1, double* y2,
                       double* g2, int64_t rows, int64_t cols) {
  size_t lds = (size_t)wpb * ((size_t)S * (R + 2) * 1024 + (size_t)rescap * sizeof(double));
  while (wpb > 1 && lds > 160 * 1024) {           // (a ring that does not fit: fewer waves per workgroup)
    wpb -= 1;
    lds = (size_t)wpb * ((size_t)S * (R + 2) * 1024 + (size_t)rescap * sizeof(double));
  }
  if (bpc < 1 || (size_t)bpc * lds > 160 * 1024) bpc = 1;
  const int64_t rs = gemv_lds_row_stride(ccqp_opt(ctx, CCQP_OPT_GEMV_ROW_STRIDE), lda, rows, R);
  const int64_t ntasks = gemv_lds_tasks(rows, R, rs);
  int64_t blocks = (ntasks + wpb - 1) / wpb;
  if (blocks > (int64_t)ctx->num_cu * bpc) blocks = (int64_t)ctx->num_cu * bpc;
  if (blocks < 1) blocks = 1;
  const int rc = ccqp_lds_raise(
      ctx, reinterpret_cast<const void*>(nt ? gemv_lds2_kernel<R, S, true> : gemv_lds2_kernel<R, S, false>), lds);
  if (rc) return rc;
  if (nt)
    hipLaunchKernelGGL((gemv_lds2_kernel<R, S, true>), dim3((unsigned)blocks), dim3(64 * wpb), lds, s, A, lda, x1, x2, b, y1, g1,
                       y2, g2, rows, cols, ntasks, stagger, rs, rescap);
  else
    hipLaunchKernelGGL((gemv_lds2_kernel<R, S, false>), dim3((unsigned)blocks), dim3(64 * wpb), lds, s, A, lda, x1, x2, b, y1,
                       g1, y2, g2, rows, cols, ntasks, stagger, rs, rescap);
  return CCQP_OK;
}

// Which form a pair of matvecs takes: true = ONE pass (gemv_lds2_kernel), wherever the single matvec would take the
// LDS-staged form; false = two passes of the single matvec.
// (the default, variant == 0, from 2^24 elements -- 4096^2 -- on: there ONE pass of the LDS-staged pair kernel beats two
// passes of whichever single form is the default, 27 against 46 us at 4096^2, 89 against 168 us at 8192^2, 0.33 against
// 0.64 ms at 16384^2; at 2048^2 a tie -- profiles/r03_gemv2_small.jsonl)
#define GEMV2_DEFAULT_SHAPE 44
static bool gemv2_default_fused(int64_t rows, int64_t cols) { return (double)rows * (double)cols >= 16777216.0; }

static bool gemv2_fused(int variant, const double* A, int64_t lda, const double* x1, const double* x2, int64_t rows,
                        int64_t cols) {
  const bool ok = gemv_lds_applicable(A, lda, x1, cols) && (reinterpret_cast<uintptr_t>(x2) & 15) == 0;
  if (variant == 0) return ok && gemv2_default_fused(rows, cols);
  variant = gemv_resolve(variant, rows, cols, ok);
  return ok && ((variant >> 24) & 1) && !((variant >> 9) & 1);
}

extern "C" int ccqp_gemv2_form(ccqp_ctx* ctx, int64_t rows, int64_t cols, int variant, char* name, int buflen) {
  CCQP_REQUIRE(ctx != nullptr && name != nullptr && buflen > 0, "NULL argument");
  const bool lds_ok = cols >= 128 && cols % 2 == 0;             // (16-byte aligned operands assumed)
  const bool fused = variant == 0 ? lds_ok && gemv2_default_fused(rows, cols)
                                  : lds_ok && ((gemv_resolve(variant, rows, cols, lds_ok) >> 24) & 1) &&
                                        !((gemv_resolve(variant, rows, cols, lds_ok) >> 9) & 1);
  const char* k = fused ? "gemv_lds2_kernel" : "two single matvecs";
  strncpy(name, k, (size_t)buflen - 1);
  name[buflen - 1] = 0;
  return CCQP_OK;
}

extern "C" int ccqp_gemv2_f64(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x1, const double* x2,
                              const double* b, double* y1, double* g1, double* y2, double* g2, int64_t rows,
                              int64_t cols, int variant, void* stream) {
  int rc = gemv_check(ctx, A, lda, x1, y1, rows, cols);
  if (rc) return rc;
  rc = gemv_check(ctx, A, lda, x2, y2, rows, cols);
  if (rc) return rc;
  CCQP_REQUIRE((g1 == nullptr && g2 == nullptr) || b != nullptr, "g requested but b is NULL");
  CCQP_REQUIRE(y1 != x2 && y2 != x1 && (g1 == nullptr || (g1 != y1 && g1 != y2)) && (g2 == nullptr || (g2 != y1 && g2 != y2)),
               "outputs must not alias each other or the inputs");
  CCQP_REQUIRE(y1 != y2 && (g1 == nullptr || (g1 != g2 && g1 != x1 && g1 != x2)) && (g2 == nullptr || (g2 != x1 && g2 != x2)),
               "outputs must not alias each other or the inputs");
  if (rows == 0) return CCQP_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!gemv2_fused(variant, A, lda, x1, x2, rows, cols)) {
    rc = ccqp_gemv_launch(ctx, A, lda, x1, b, y1, g1, rows, cols, variant, s, nullptr);
    if (rc) return rc;
    rc = ccqp_gemv_launch(ctx, A, lda, x2, b, y2, g2, rows, cols, variant, s, nullptr);
    if (rc) return rc;
    CCQP_LAUNCH_CHECK();
    return CCQP_OK;
  }
  const int v = gemv_lds_variant(variant == 0 ? GEMV_AUTO_LDS : gemv_resolve(variant, rows, cols, true));
  const gemv_lds_shape h = gemv_lds_decode(v);
  int bpc = (v >> 12) & 0xFF;
  int stagger = h.stagger;
  if (stagger < 0) {
    const double gib4 = 8.0 * (double)rows * (double)cols / 4294967296.0;
    stagger = gib4 < 1.0 ? 1 : gib4 > 32.0 ? 32 : (int)(gib4 + 0.5);
  }
  const bool nt = !((v >> 8) & 1) && (double)rows * (double)cols * 8.0 > 192.0 * 1048576.0;
  // waves per workgroup: FOUR by default (twice the single matvec's: a wave's step now feeds two fma chains per row and
  // reads two x pieces, and two waves per CU no longer hide it -- 19.36 against 22.06 ms on the headline matrix,
  // profiles/r03_gemv2_sweep.jsonl); an explicit variant word keeps its own count; the context's options GEMV2_WPB /
  // GEMV2_SHAPE / GEMV2_GROUP are the tools' A/B switches (same bits whatever the shape)
  int wpb = variant == 0 ? CCQP_BLOCK / 64 : h.wpb;
  if (ccqp_opt(ctx, CCQP_OPT_GEMV2_WPB)) wpb = (int)ccqp_opt(ctx, CCQP_OPT_GEMV2_WPB);
  // ring shape <rows per wave><slots>: 44 = the single matvec's (variant bits 28-29 = 3 ask for eight rows: 84)
  int shape = (int)ccqp_opt(ctx, CCQP_OPT_GEMV2_SHAPE);
  if (shape == 0) shape = h.R == 8 ? 84 : GEMV2_DEFAULT_SHAPE;
  int rescap = (int)ccqp_opt(ctx, CCQP_OPT_GEMV2_GROUP);
  if (rescap == 0) rescap = GEMV_LDS_RESCAP;
#define LDS2_CASE(RR, SS) \
  case RR * 10 + SS:      \
    rc = launch_lds2<RR, SS>(ctx, wpb, bpc, nt, stagger, rescap, s, A, lda, x1, x2, b, y1, g1, y2, g2, rows, cols); \
    break
  switch (shape) {
    LDS2_CASE(4, 3);
    LDS2_CASE(4, 6);
    LDS2_CASE(4, 8);
    LDS2_CASE(8, 2);
    LDS2_CASE(8, 3);
    LDS2_CASE(8, 4);
    default:
      LDS2_CASE(4, 4);
  }
#undef LDS2_CASE
  if (rc) return rc;
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// ---- k <= 3 products of a rows x cols block (ccqp_hip.h): one pass where the pair kernel would take one
extern "C" int ccqp_gemvk_form(ccqp_ctx* ctx, int k, int64_t rows, int64_t cols, int variant, char* name, int buflen) {
  CCQP_REQUIRE(ctx != nullptr && name != nullptr && buflen > 0, "NULL argument");
  CCQP_REQUIRE(k >= 1 && k <= 3, "k outside [1, 3]");
  if (k == 1) return ccqp_gemv_form(ctx, rows, cols, variant, name, buflen);
  if (k == 2) return ccqp_gemv2_form(ctx, rows, cols, variant, name, buflen);
  char two[64];
  const int rc = ccqp_gemv2_form(ctx, rows, cols, variant, two, 64);
  if (rc) return rc;
  const char* kn = strcmp(two, "gemv_lds2_kernel") == 0 ? "gemv_lds3_kernel" : "three single matvecs";
  strncpy(name, kn, (size_t)buflen - 1);
  name[buflen - 1] = 0;
  return CCQP_OK;
}

extern "C" int ccqp_gemvk_f64(ccqp_ctx* ctx, const double* A, int64_t lda, int k, const double* const* x, const double* b,
                              double* const* y, double* const* g, int64_t rows, int64_t cols, int variant, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && x != nullptr && y != nullptr && g != nullptr, "NULL argument");
  CCQP_REQUIRE(k >= 1 && k <= 3, "k outside [1, 3]");
  if (k == 1) return ccqp_gemv_epilogue_f64(ctx, A, lda, x[0], b, y[0], g[0], rows, cols, variant, stream);
  if (k == 2) return ccqp_gemv2_f64(ctx, A, lda, x[0], x[1], b, y[0], g[0], y[1], g[1], rows, cols, variant, stream);
  bool fused = true;
  for (int j = 0; j < 3; ++j) {
    const int rc = gemv_check(ctx, A, lda, x[j], y[j], rows, cols);
    if (rc) return rc;
    CCQP_REQUIRE(g[j] == nullptr || b != nullptr, "g requested but b is NULL");
    for (int i = 0; i < 3; ++i) {
      CCQP_REQUIRE(y[j] != x[i] && (g[j] == nullptr || g[j] != x[i]), "outputs must not alias the inputs");
      CCQP_REQUIRE(i == j || (y[j] != y[i] && (g[j] == nullptr || (g[j] != y[i] && g[j] != g[i]))),
                   "outputs must not alias each other");
    }
    CCQP_REQUIRE(g[j] == nullptr || g[j] != y[j], "outputs must not alias each other");
    fused = fused && (reinterpret_cast<uintptr_t>(x[j]) & 15) == 0;
  }
  if (rows == 0) return CCQP_OK;
  hipStream_t s = (hipStream_t)stream;
  fused = fused && gemv2_fused(variant, A, lda, x[0], x[1], rows, cols);
  if (!fused) {
    for (int j = 0; j < 3; ++j) {
      const int rc = ccqp_gemv_launch(ctx, A, lda, x[j], b, y[j], g[j], rows, cols, variant, s, nullptr);
      if (rc) return rc;
    }
    CCQP_LAUNCH_CHECK();
    return CCQP_OK;
  }
  const int v = gemv_lds_variant(variant == 0 ? GEMV_AUTO_LDS : gemv_resolve(variant, rows, cols, true));
  int stagger = gemv_lds_decode(v).stagger;
  if (stagger < 0) {
    const double gib4 = 8.0 * (double)rows * (double)cols / 4294967296.0;
    stagger = gib4 < 1.0 ? 1 : gib4 > 32.0 ? 32 : (int)(gib4 + 0.5);
  }
  const bool nt = !((v >> 8) & 1) && (double)rows * (double)cols * 8.0 > 192.0 * 1048576.0;
  gemv_ldsk_io<3> io;
  for (int j = 0; j < 3; ++j) io.x[j] = x[j], io.y[j] = y[j], io.g[j] = g[j];
  // (the three-vector persistent launch's shape for few matrices: four rows per wave, four slots, four waves)
  const int rc = launch_lds3<4, 4>(ctx, CCQP_BLOCK / 64, nt, stagger, s, A, lda, io, b, rows, cols);
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
