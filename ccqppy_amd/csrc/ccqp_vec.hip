// O(n) vector kernels: the axpy family (numpy rounding: one rounding per operation, built with
// -ffp-contract=off) and the deterministic two-stage dot-product reduction.
// These carry < 2 % of the HBM traffic of a solve; they are written for
// exact numpy parity and few launches, not for bandwidth.
#include "ccqp_bodies.h"

static inline int vec_grid(int64_t n) {
  int64_t g = (n + CCQP_BLOCK - 1) / CCQP_BLOCK;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

#define VEC_LOOP(i, n)                                                         \
  for (int64_t i = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; i < (n);   \
       i += (int64_t)gridDim.x * CCQP_BLOCK)

__global__ __launch_bounds__(CCQP_BLOCK) void axpby_kernel(double a, const double* x, double b,
                                                           const double* y, double* out, int64_t n) {
  axpby_body(a, x, b, y, out, n, blockIdx.x, gridDim.x);
}

// two independent axpby of the same length in one launch (blockIdx.y picks the operation)
__global__ __launch_bounds__(CCQP_BLOCK) void axpby2_kernel(double a1, const double* x1, double b1, const double* y1,
                                                            double* out1, double a2, const double* x2, double b2,
                                                            const double* y2, double* out2, int64_t n) {
  if (blockIdx.y == 0)
    axpby_body(a1, x1, b1, y1, out1, n, blockIdx.x, gridDim.x);
  else
    axpby_body(a2, x2, b2, y2, out2, n, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(CCQP_BLOCK) void scale_kernel(double a, const double* x, double* out, int64_t n) {
  scale_body(a, x, out, n, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(CCQP_BLOCK) void copy_kernel(const double* x, double* out, int64_t n) {
  copy_body(x, out, n, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(CCQP_BLOCK) void fill_kernel(double v, double* out, int64_t n) {
  fill_body(v, out, n, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(CCQP_BLOCK) void mul_kernel(const double* x, const double* y, double* out,
                                                         int64_t n) {
  VEC_LOOP(i, n) out[i] = x[i] * y[i];
}

__global__ __launch_bounds__(CCQP_BLOCK) void muldiv_kernel(const double* x, const double* y, double s,
                                                            double* out, int64_t n) {
  VEC_LOOP(i, n) {
    const double t = x[i] * y[i];
    out[i] = t / s;
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void submul_kernel(const double* x, const double* y,
                                                            const double* z, double* out, int64_t n) {
  VEC_LOOP(i, n) {
    const double t = y[i] * z[i];
    out[i] = x[i] - t;
  }
}

#define VEC_PROLOGUE(nullcheck)                                    \
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");                     \
  CCQP_REQUIRE(n >= 0, "negative size");                           \
  if (n == 0) return CCQP_OK;                                      \
  CCQP_REQUIRE(nullcheck, "NULL vector");                          \
  hipStream_t s = (hipStream_t)stream;

extern "C" int ccqp_axpby_f64(ccqp_ctx* ctx, double a, const double* x, double b, const double* y,
                              double* out, int64_t n, void* stream) {
  VEC_PROLOGUE(x && y && out)
  hipLaunchKernelGGL(axpby_kernel, dim3(vec_grid(n)), dim3(CCQP_BLOCK), 0, s, a, x, b, y, out, n);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_axpby2_f64(ccqp_ctx* ctx, double a1, const double* x1, double b1, const double* y1,
                               double* out1, double a2, const double* x2, double b2, const double* y2,
                               double* out2, int64_t n, void* stream) {
  VEC_PROLOGUE(x1 && y1 && out1 && x2 && y2 && out2)
  CCQP_REQUIRE(out1 != out2 && out1 != x2 && out1 != y2 && out2 != x1 && out2 != y1,
               "the two updates must not depend on each other");
  hipLaunchKernelGGL(axpby2_kernel, dim3(vec_grid(n), 2), dim3(CCQP_BLOCK), 0, s, a1, x1, b1, y1, out1, a2, x2, b2, y2,
                     out2, n);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_scale_f64(ccqp_ctx* ctx, double a, const double* x, double* out, int64_t n, void* stream) {
  VEC_PROLOGUE(x && out)
  hipLaunchKernelGGL(scale_kernel, dim3(vec_grid(n)), dim3(CCQP_BLOCK), 0, s, a, x, out, n);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_copy_f64(ccqp_ctx* ctx, const double* x, double* out, int64_t n, void* stream) {
  VEC_PROLOGUE(x && out)
  hipLaunchKernelGGL(copy_kernel, dim3(vec_grid(n)), dim3(CCQP_BLOCK), 0, s, x, out, n);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_fill_f64(ccqp_ctx* ctx, double value, double* out, int64_t n, void* stream) {
  VEC_PROLOGUE(out)
  hipLaunchKernelGGL(fill_kernel, dim3(vec_grid(n)), dim3(CCQP_BLOCK), 0, s, value, out, n);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_mul_f64(ccqp_ctx* ctx, const double* x, const double* y, double* out, int64_t n,
                            void* stream) {
  VEC_PROLOGUE(x && y && out)
  hipLaunchKernelGGL(mul_kernel, dim3(vec_grid(n)), dim3(CCQP_BLOCK), 0, s, x, y, out, n);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_muldiv_f64(ccqp_ctx* ctx, const double* x, const double* y, double sc, double* out,
                               int64_t n, void* stream) {
  VEC_PROLOGUE(x && y && out)
  hipLaunchKernelGGL(muldiv_kernel, dim3(vec_grid(n)), dim3(CCQP_BLOCK), 0, s, x, y, sc, out, n);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_submul_f64(ccqp_ctx* ctx, const double* x, const double* y, const double* z,
                               double* out, int64_t n, void* stream) {
  VEC_PROLOGUE(x && y && z && out)
  hipLaunchKernelGGL(submul_kernel, dim3(vec_grid(n)), dim3(CCQP_BLOCK), 0, s, x, y, z, out, n);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// ------------------------------------------------------------------ dots
struct dot_args {
  const double* x[CCQP_MAX_DOTS];
  const double* y[CCQP_MAX_DOTS];
};

// grid (nblk, k).  Block (b, j) reduces elements [b*chunk, (b+1)*chunk) of pair j: thread t takes
// t, t+256, ... with one fma chain, then wave butterfly, then waves 0..3 in order.
__global__ __launch_bounds__(CCQP_BLOCK) void dots_stage1(dot_args a, int64_t n, int64_t chunk,
                                                          double* dst, int nblk, double* slots) {
  __shared__ double lds4[4];
  const int j = blockIdx.y;
  const double s = dot_body(a.x[j], a.y[j], n, chunk, blockIdx.x, lds4);
  if (threadIdx.x == 0) dst[(int64_t)j * nblk + blockIdx.x] = s;
  if (nblk > 1) ccqp_reduce_finish(dst + (int64_t)j * nblk, nblk, 1, ccqp_red_counters(dst) + j, slots + j, lds4);
}

extern "C" int ccqp_dots_f64(ccqp_ctx* ctx, int k, const double* const* xs, const double* const* ys,
                             int64_t n, int slot0, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(k >= 1 && k <= CCQP_MAX_DOTS, "k outside [1, CCQP_MAX_DOTS]");
  CCQP_REQUIRE(n >= 0, "negative size");
  CCQP_REQUIRE(ccqp_slot_ok(slot0, k), "slot range outside the slab");
  CCQP_REQUIRE(xs && ys, "NULL pointer table");
  hipStream_t s = (hipStream_t)stream;
  dot_args a;
  for (int j = 0; j < CCQP_MAX_DOTS; ++j) {
    a.x[j] = j < k ? xs[j] : nullptr;
    a.y[j] = j < k ? ys[j] : nullptr;
    CCQP_REQUIRE(j >= k || n == 0 || (a.x[j] && a.y[j]), "NULL vector");
  }
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  // one launch either way: with several blocks per dot, the block that finishes last adds the partials
  hipLaunchKernelGGL(dots_stage1, dim3(nblk, k), dim3(CCQP_BLOCK), 0, s, a, n, chunk,
                     nblk == 1 ? ctx->slots + slot0 : ctx->partials, nblk, ctx->slots + slot0);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}
