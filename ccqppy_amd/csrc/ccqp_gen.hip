// Synthetic problem generator: counter-hashed symmetric matrix, row-range
// addressable so each rank fills only its own row block, in place, at HBM write speed.
// The reference's own generator (scipy.stats.wishart, benchmarks/benchmark_random_ccqp.py:59-63)
// is O(n^3) and was only ever run for n <= 12.  CPU statement of the same formulas:
// oracle/gen_ref.py (bit-for-bit equality is asserted by tests/test_gpu_ops.py).
#include <math.h>

#include "ccqp_internal.h"

#define SM_GOLDEN 0x9E3779B97F4A7C15ull
#define SM_M1 0xBF58476D1CE4E5B9ull
#define SM_M2 0x94D049BB133111EBull
#define VEC_STREAM 0xD1B54A32D192ED03ull

__host__ __device__ static inline uint64_t splitmix64(uint64_t z) {
  z += SM_GOLDEN;
  z = (z ^ (z >> 30)) * SM_M1;
  z = (z ^ (z >> 27)) * SM_M2;
  return z ^ (z >> 31);
}

__device__ __forceinline__ double u01(uint64_t key, uint64_t k) {
  return (double)(splitmix64(key + k) >> 11) * 0x1.0p-53;  // exact: 53-bit integer scaled by 2^-53
}

typedef double gen_d2 __attribute__((ext_vector_type(2)));

// One thread produces two adjacent columns (one 16-byte store when aligned).
__global__ __launch_bounds__(CCQP_BLOCK) void gen_rows_kernel(double* A, int64_t lda, int64_t row0,
                                                              int64_t rows, int64_t n, uint64_t key,
                                                              double sqrt_n, int aligned) {
  const int64_t pairs_per_row = (n + 1) / 2;
  const int64_t total = rows * pairs_per_row;
  for (int64_t t = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * CCQP_BLOCK) {
    const int64_t r = t / pairs_per_row;
    const int64_t j0 = (t - r * pairs_per_row) * 2;
    const uint64_t i = (uint64_t)(row0 + r);
    double v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const uint64_t j = (uint64_t)(j0 + e);
      const uint64_t lo = i < j ? i : j, hi = i < j ? j : i;
      const double u = u01(key, lo * (uint64_t)n + hi);
      const double w = 2.0 * u - 1.0;  // exact
      v[e] = (i == j) ? 2.0 : w / sqrt_n;
    }
    double* dst = A + r * lda + j0;
    if (aligned && j0 + 1 < n) {
      gen_d2 p;
      p.x = v[0];
      p.y = v[1];
      __builtin_nontemporal_store(p, reinterpret_cast<gen_d2*>(dst));
    } else {
      dst[0] = v[0];
      if (j0 + 1 < n) dst[1] = v[1];
    }
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void gen_vec_kernel(double* out, int64_t n, uint64_t key) {
  for (int64_t i = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * CCQP_BLOCK) {
    const double u = u01(key, (uint64_t)i);
    out[i] = 1.0 - 2.0 * u;
  }
}

extern "C" int ccqp_gen_spd_rows_f64(ccqp_ctx* ctx, double* A, int64_t lda, int64_t row0, int64_t rows,
                                     int64_t n, uint64_t seed, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(n >= 0 && rows >= 0 && row0 >= 0 && row0 + rows <= n, "row range outside the matrix");
  CCQP_REQUIRE(lda >= n, "lda < n");
  if (rows == 0 || n == 0) return CCQP_OK;
  CCQP_REQUIRE(A != nullptr, "A is NULL");
  const uint64_t key = splitmix64(seed);
  const double sq = sqrt((double)n);  // host libm sqrt: correctly rounded, same value as numpy's
  const int aligned = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda % 2) == 0) ? 1 : 0;
  const int64_t total = rows * ((n + 1) / 2);
  int64_t grid = (total + CCQP_BLOCK - 1) / CCQP_BLOCK;
  const int64_t cap = (int64_t)ctx->num_cu * 16;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(gen_rows_kernel, dim3((int)grid), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, A, lda, row0,
                     rows, n, key, sq, aligned);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_gen_vec_f64(ccqp_ctx* ctx, double* out, int64_t n, uint64_t seed, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(n >= 0, "negative size");
  if (n == 0) return CCQP_OK;
  CCQP_REQUIRE(out != nullptr, "out is NULL");
  const uint64_t key = splitmix64(seed ^ VEC_STREAM);
  int64_t grid = (n + CCQP_BLOCK - 1) / CCQP_BLOCK;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(gen_vec_kernel, dim3((int)grid), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, out, n, key);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}
