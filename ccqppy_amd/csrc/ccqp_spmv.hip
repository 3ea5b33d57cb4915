// Sparse (CSR, fp64) form of the solvers' `A.dot(v)` (reference: src/ccqppy/solvers.py:25 -- the solver
// only ever calls `A.dot`, so a scipy.sparse matrix is a legal `A` there).
//
// HBM-bound streaming kernel: every stored entry is read once (8 B value + 4 B column), x is gathered
// through L2 / Infinity Cache, one 8-B row pointer and one 8-B result per row.
//
// Determinism contract (same as the dense matvec): the bits of y[i] are a function of row i's stored
// entries and x ONLY.  The canonical order is: entry e of the row (0-based, in storage order) belongs to
// canonical lane L = e mod 64; lane L accumulates its entries in order with one fma each, starting from
// +0.0; the 64 lane sums are combined by the xor butterfly 32,16,8,4,2,1.  A row is worked on by a
// SUBGROUP of W lanes (W = 4..64, picked from the mean row length); lane l of the subgroup then carries
// the 64/W canonical lanes l, l+W, l+2W, ... in separate accumulators and performs the butterfly steps
// >= W inside its registers, so every W yields the same bits and W is a pure tuning knob -- and so is
// any row-sharding of the matrix.
#include "ccqp_internal.h"

template <int W, bool EPI>
__global__ __launch_bounds__(CCQP_BLOCK) void spmv_csr_kernel(const int64_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ col,
                                                              const double* __restrict__ val,
                                                              const double* __restrict__ x,
                                                              const double* __restrict__ b, double* __restrict__ y,
                                                              double* __restrict__ g, int64_t rows) {
  constexpr int K = 64 / W;                 // canonical lanes per physical lane
  constexpr int ROWS_PER_BLOCK = CCQP_BLOCK / W;
  const int l = threadIdx.x % W;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x / W;
  int64_t beg = 0, end = 0;
  if (row < rows) {                         // no early exit: every lane takes part in the shuffles
    beg = rowptr[row];
    end = rowptr[row + 1];
  }
  double acc[K];
#pragma unroll
  for (int a = 0; a < K; ++a) acc[a] = 0.0;
  for (int64_t e = beg + l; e < end; e += 64) {
#pragma unroll
    for (int a = 0; a < K; ++a) {
      const int64_t idx = e + (int64_t)a * W;
      if (idx < end) {
        const double v = __builtin_nontemporal_load(val + idx);
        const int32_t c = __builtin_nontemporal_load(col + idx);
        acc[a] = fma(v, x[c], acc[a]);
      }
    }
  }
  // butterfly steps 32 .. W: both partners live in this lane
#pragma unroll
  for (int s = K / 2; s >= 1; s >>= 1) {
#pragma unroll
    for (int a = 0; a < s; ++a) acc[a] = acc[a] + acc[a + s];
  }
  double v = acc[0];
  // butterfly steps W/2 .. 1 across the subgroup
#pragma unroll
  for (int s = W / 2; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, W);
  if (l == 0 && row < rows) {
    y[row] = v;
    if (EPI) g[row] = v + b[row];
  }
}

// violations of the CSR invariants, counted into a scalar slot (0 = valid)
__global__ __launch_bounds__(CCQP_BLOCK) void csr_check_rows_kernel(const int64_t* __restrict__ rowptr, int64_t rows,
                                                                    int64_t nnz, double* __restrict__ slot) {
  const int64_t r = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x;
  int bad = 0;
  if (r < rows) {
    const int64_t a = rowptr[r], c = rowptr[r + 1];
    bad = (a < 0) || (c < a) || (c > nnz);
  }
  if (r == 0 && rows > 0) bad |= (rowptr[0] != 0) || (rowptr[rows] != nnz);
  if (bad) atomicAdd(slot, 1.0);
}

__global__ __launch_bounds__(CCQP_BLOCK) void csr_check_cols_kernel(const int32_t* __restrict__ col, int64_t nnz,
                                                                    int64_t cols, double* __restrict__ slot) {
  const int64_t stride = (int64_t)gridDim.x * CCQP_BLOCK;
  int bad = 0;
  for (int64_t e = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; e < nnz; e += stride) {
    const int32_t c = col[e];
    bad += (c < 0) || ((int64_t)c >= cols);
  }
  if (bad) atomicAdd(slot, (double)bad);
}

static int spmv_pick_width(int64_t rows, int64_t nnz, int width) {
  if (width == 4 || width == 8 || width == 16 || width == 32 || width == 64) return width;
  const double mean = rows > 0 ? (double)nnz / (double)rows : 0.0;
  if (mean <= 6.0) return 4;
  if (mean <= 12.0) return 8;
  if (mean <= 24.0) return 16;
  if (mean <= 48.0) return 32;
  return 64;
}

template <int W>
static void spmv_launch(const int64_t* rowptr, const int32_t* col, const double* val, const double* x,
                        const double* b, double* y, double* g, int64_t rows, hipStream_t s) {
  const int64_t rpb = CCQP_BLOCK / W;
  const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
  if (g)
    hipLaunchKernelGGL((spmv_csr_kernel<W, true>), dim3(grid), dim3(CCQP_BLOCK), 0, s, rowptr, col, val, x, b, y, g,
                       rows);
  else
    hipLaunchKernelGGL((spmv_csr_kernel<W, false>), dim3(grid), dim3(CCQP_BLOCK), 0, s, rowptr, col, val, x, b, y,
                       g, rows);
}

extern "C" int ccqp_spmv_csr_f64(ccqp_ctx* ctx, const int64_t* rowptr, const int32_t* col, const double* val,
                                 const double* x, const double* b, double* y, double* g, int64_t rows,
                                 int64_t cols, int64_t nnz, int width, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(rows >= 0 && cols >= 0 && nnz >= 0, "negative size");
  CCQP_REQUIRE(cols <= INT32_MAX, "cols exceeds the 32-bit column index");
  CCQP_REQUIRE(rows == 0 || (rowptr != nullptr && y != nullptr), "NULL rowptr / y");
  CCQP_REQUIRE(nnz == 0 || (col != nullptr && val != nullptr && x != nullptr), "NULL col / val / x");
  CCQP_REQUIRE(y != x, "y must not alias x");
  CCQP_REQUIRE(g == nullptr || b != nullptr, "g requested but b is NULL");
  CCQP_REQUIRE(g == nullptr || g != x, "g must not alias x");
  CCQP_REQUIRE(width == 0 || width == 4 || width == 8 || width == 16 || width == 32 || width == 64,
               "width must be 0 (auto), 4, 8, 16, 32 or 64");
  if (rows == 0) return CCQP_OK;
  CCQP_REQUIRE((rows + 3) / 4 <= (int64_t)INT32_MAX, "too many rows for one launch");
  hipStream_t s = (hipStream_t)stream;
  switch (spmv_pick_width(rows, nnz, width)) {
    case 4: spmv_launch<4>(rowptr, col, val, x, b, y, g, rows, s); break;
    case 8: spmv_launch<8>(rowptr, col, val, x, b, y, g, rows, s); break;
    case 16: spmv_launch<16>(rowptr, col, val, x, b, y, g, rows, s); break;
    case 32: spmv_launch<32>(rowptr, col, val, x, b, y, g, rows, s); break;
    default: spmv_launch<64>(rowptr, col, val, x, b, y, g, rows, s); break;
  }
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_csr_validate(ccqp_ctx* ctx, const int64_t* rowptr, const int32_t* col, int64_t rows,
                                 int64_t cols, int64_t nnz, int slot, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(rows >= 0 && cols >= 0 && nnz >= 0, "negative size");
  CCQP_REQUIRE(ccqp_slot_ok(slot, 1), "slot out of range");
  CCQP_REQUIRE(rowptr != nullptr, "NULL rowptr (rows + 1 entries are required, even for rows == 0)");
  CCQP_REQUIRE(nnz == 0 || col != nullptr, "NULL col");
  hipStream_t s = (hipStream_t)stream;
  CCQP_HIP(hipMemsetAsync(ctx->slots + slot, 0, sizeof(double), s));
  const int64_t rgrid = (rows + 1 + CCQP_BLOCK - 1) / CCQP_BLOCK;
  CCQP_REQUIRE(rgrid <= (int64_t)INT32_MAX, "too many rows for one launch");
  hipLaunchKernelGGL(csr_check_rows_kernel, dim3((unsigned)rgrid), dim3(CCQP_BLOCK), 0, s, rowptr, rows, nnz,
                     ctx->slots + slot);
  if (nnz > 0) {
    int64_t cgrid = (nnz + CCQP_BLOCK - 1) / CCQP_BLOCK;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (cgrid > cap) cgrid = cap;
    hipLaunchKernelGGL(csr_check_cols_kernel, dim3((unsigned)cgrid), dim3(CCQP_BLOCK), 0, s, col, nnz, cols,
                       ctx->slots + slot);
  }
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}
