// Symmetric matvec reading ONLY the upper triangle of a fully stored symmetric matrix:
// y = A.x with 4n^2 bytes of HBM traffic instead of 8n^2 ("symmetric-half-storage
// operators").  OPT-IN and NON-PARITY: the Hessian of a convex QP is symmetric, but the reference's
// `A.dot(v)` reads all of A and sums each row left to right; here a_ij (j > i) is read once and feeds
// both y_i and y_j, so the summation order differs from ccqp_gemv_f64 (same value to ~1e-15 relative).
//
// Layout of the work: block-row I = rows [32 I, 32 I + 32).  One wave64 streams the part of these rows
// right of the diagonal block's first 128-column chunk, 128 columns at a time (every wave-instruction
// is still one contiguous 1 KiB piece of a row):
//   * row part    : acc[r] += a_rj x_j            (j >= i)      -> yrow[i] after a lane butterfly
//   * column part : cp_j   += a_rj x_r over the 32 rows (j > i) -> ycol[I][j]   (lane-local: a lane owns
//                   the same two columns in all 32 rows, so no cross-lane traffic at all)
// The rows are further cut into column segments of 8192 (task = 32 rows x <= 8192 columns, so the triangle
// load-balances over the waves); a second kernel adds, for every j, the row partials of its segments and
// ycol[I][j] for I = 0..j/32 in a fixed pattern.  Extra traffic: the ycol slabs, n^2/64 doubles written +
// read = 1/16 of the triangle each.  Everything is a fixed function of n: bitwise reproducible.
#include "ccqp_internal.h"

#define SYMV_RB 32   // rows per block-row (one wave)
#define SYMV_G 16    // rows whose loads are in flight together (sweep: 32/16 best at n >= 32768)

typedef double symv_d2 __attribute__((ext_vector_type(2)));

#define SYMV_CSEG 8192   // columns per task: a task = (block-row I, column segment s) = 32 rows x <= 8192 columns

// grid (ceil(nb/4), nseg): wave (I, s) handles block-row I over the absolute columns of segment s that lie
// right of the block-row's first chunk; (I, s) pairs left of the diagonal exit at once.
// (forcing 3 waves/SIMD with __launch_bounds__(256, 3) spills 230 VGPRs and runs 3x slower: measured)
__global__ __launch_bounds__(CCQP_BLOCK) void symv_upper_kernel(const double* __restrict__ A, int64_t lda,
                                                                const double* __restrict__ x,
                                                                double* __restrict__ yrow,   // [nseg][n]
                                                                double* __restrict__ ycol,   // [nb][n]
                                                                int64_t n, int64_t nb) {
  const int lane = threadIdx.x & 63;
  // readfirstlane makes the block-row index provably wave-uniform: the 32 row bases live in SGPRs and
  // only the lane's column offset needs VGPRs (236 -> far fewer registers, more waves per SIMD)
  const int64_t I = (int64_t)blockIdx.x * (CCQP_BLOCK / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (I >= nb) return;
  const int64_t sidx = blockIdx.y;
  const int64_t I0 = I * SYMV_RB;
  const int64_t cstart = (I0 / 128) * 128;
  int64_t c_lo = sidx * SYMV_CSEG;
  int64_t c_hi = c_lo + SYMV_CSEG;
  if (c_hi > n) c_hi = n;
  if (c_lo < cstart) c_lo = cstart;
  if (c_lo >= c_hi) return;
  double acc[SYMV_RB];
#pragma unroll
  for (int r = 0; r < SYMV_RB; ++r) acc[r] = 0.0;
  double* __restrict__ ycI = ycol + I * n;
  for (int64_t c0 = c_lo; c0 < c_hi; c0 += 128) {
    const int64_t col = c0 + 2 * lane;
    const symv_d2 xv = *reinterpret_cast<const symv_d2*>(x + col);
    const bool diag = c0 < I0 + SYMV_RB;  // this chunk still touches the diagonal block
    symv_d2 cp;
    cp.x = 0.0;
    cp.y = 0.0;
#pragma unroll
    for (int rg = 0; rg < SYMV_RB; rg += SYMV_G) {
      symv_d2 av[SYMV_G];
#pragma unroll
      for (int r = 0; r < SYMV_G; ++r)
        av[r] = __builtin_nontemporal_load(reinterpret_cast<const symv_d2*>(A + (I0 + rg + r) * lda + col));
#pragma unroll
      for (int r = 0; r < SYMV_G; ++r) {
        const int64_t i = I0 + rg + r;
        const double xi = x[i];
        symv_d2 ar = av[r], ac = av[r];
        if (diag) {
          if (col < i) ar.x = 0.0;       // strictly lower part: owned by the mirrored entry
          if (col + 1 < i) ar.y = 0.0;
          if (col <= i) ac.x = 0.0;      // the diagonal itself counts once (row part only)
          if (col + 1 <= i) ac.y = 0.0;
        }
        acc[rg + r] = fma(ar.x, xv.x, acc[rg + r]);
        acc[rg + r] = fma(ar.y, xv.y, acc[rg + r]);
        cp.x = fma(ac.x, xi, cp.x);
        cp.y = fma(ac.y, xi, cp.y);
      }
    }
    *reinterpret_cast<symv_d2*>(ycI + col) = cp;
  }
#pragma unroll
  for (int r = 0; r < SYMV_RB; ++r) acc[r] = ccqp_wave_sum(acc[r]);
  double* __restrict__ yr = yrow + sidx * n;
#pragma unroll
  for (int r = 0; r < SYMV_RB; ++r)
    if (lane == r) yr[I0 + r] = acc[r];
}

// y[j] = sum_{s >= s0(j)} yrow[s][j]  +  sum_{I <= j/32} ycol[I][j]   (both ascending);  g = y + b if requested
__global__ __launch_bounds__(CCQP_BLOCK) void symv_combine_kernel(const double* __restrict__ yrow,
                                                                  const double* __restrict__ ycol,
                                                                  const double* __restrict__ b,
                                                                  double* __restrict__ y, double* __restrict__ g,
                                                                  int64_t n, int64_t nseg) {
  const int64_t j = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x;
  if (j >= n) return;
  const int64_t I0 = (j / SYMV_RB) * SYMV_RB;
  const int64_t s0 = ((I0 / 128) * 128) / SYMV_CSEG;   // first segment the row's block-row takes part in
  double s = 0.0;
  for (int64_t q = s0; q < nseg; ++q) s += yrow[q * n + j];
  // 8 interleaved partial sums keep 8 loads in flight per thread; fixed pattern => deterministic
  const int64_t count = j / SYMV_RB + 1;
  double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int64_t I = 0;
  for (; I + 8 <= count; I += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] += ycol[(I + u) * n + j];
  }
  for (int u = 0; I < count; ++I, ++u) p[u] += ycol[I * n + j];
  s += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  y[j] = s;
  if (g) g[j] = s + b[j];
}

extern "C" int64_t ccqp_symv_workspace_doubles(int64_t n) {
  if (n <= 0) return 0;
  const int64_t nseg = (n + SYMV_CSEG - 1) / SYMV_CSEG;
  return (n / SYMV_RB) * n + nseg * n;
}

extern "C" int ccqp_symv_upper_f64(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, const double* b,
                                   double* y, double* g, int64_t n, double* workspace, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(n >= 0, "negative size");
  if (n == 0) return CCQP_OK;
  CCQP_REQUIRE(A && x && y && workspace, "NULL operand");
  CCQP_REQUIRE(n % 128 == 0, "n must be a multiple of 128");
  CCQP_REQUIRE(lda >= n && lda % 2 == 0, "lda < n or odd");
  CCQP_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(x) |
                 reinterpret_cast<uintptr_t>(workspace)) & 15) == 0, "operands must be 16-byte aligned");
  CCQP_REQUIRE(g == nullptr || b != nullptr, "g requested but b is NULL");
  CCQP_REQUIRE(y != x && g != x, "output must not alias x");
  hipStream_t s = (hipStream_t)stream;
  const int64_t nb = n / SYMV_RB;
  const int64_t nseg = (n + SYMV_CSEG - 1) / SYMV_CSEG;
  double* ycol = workspace;
  double* yrow = workspace + nb * n;
  hipLaunchKernelGGL(symv_upper_kernel, dim3((unsigned)((nb + 3) / 4), (unsigned)nseg), dim3(CCQP_BLOCK), 0, s, A, lda,
                     x, yrow, ycol, n, nb);
  hipLaunchKernelGGL(symv_combine_kernel, dim3((unsigned)((n + CCQP_BLOCK - 1) / CCQP_BLOCK)), dim3(CCQP_BLOCK), 0,
                     s, yrow, ycol, b, y, g, n, nseg);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}
