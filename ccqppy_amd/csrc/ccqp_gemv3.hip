// ccqp_gemv3_f64: up to THREE matvecs with one square matrix in ONE pass over it -- the single-problem entry to the
// three-vector persistent launch of the lock-step forms (bp_gemv3_pers_kernel, ccqp_lockstep.h; gemv_lds3_multi_body,
// ccqp_bodies.h).  Replaces the A.dot(dx), A.dot(xk) and A.dot(p) that follow each other across the end of an MPRGP
// iteration (/root/reference/src/ccqppy/solvers.py:1129 | :1163, :1086, :1100): at the A.dx call x_{k+1} and p_{k+1} are
// known.  Each product has the bits of its own ccqp_gemv_f64 / ccqp_gemv_epilogue_f64 call.  Where the single matvec
// is not the LDS-staged form (small or odd n, unaligned operands) the call is up to three single matvecs.
#include <stdint.h>
#include <string.h>

#include <hip/hip_runtime.h>

#include "ccqp_hip.h"
#include "ccqp_internal.h"
#include "ccqp_lockstep.h"

int ccqp_gemv_launch(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, const double* b, double* y, double* g,
                     int64_t rows, int64_t cols, int variant, hipStream_t s, const double* skip);

struct gemv3_one {                 // "resolver" of a batch of one
  gemv_lds3_desc d;
  __device__ bool operator()(int, gemv_lds3_desc* out) const {
    *out = d;
    return true;
  }
};

// one pass: by default from 2^24 elements on (below), or where the variant word asks for the LDS-staged form (bit 24);
// never under (variant & 0xF) == 0xF or the MFMA bit, nor for unaligned operands / odd or tiny n
static bool gemv3_one_pass(int variant, int64_t n, bool aligned) {
  if (!aligned || n < 128 || n % 2) return false;
  if ((variant & 0xF) == 0xF || ((variant >> 9) & 1)) return false;
  // (the default from 4096^2 on: one pass for three products 31 against 71 us at 4096^2, 94 against 244 us at 8192^2,
  // 0.34 against 0.91 ms at 16384^2; at 2048^2 a tie -- profiles/r03_gemv2_small.jsonl)
  if (variant == 0) return (double)n * (double)n >= 16777216.0;
  return (variant >> 24) & 1;
}

extern "C" int ccqp_gemv3_form(ccqp_ctx* ctx, int64_t n, int variant, char* name, int buflen) {
  CCQP_REQUIRE(ctx != nullptr && name != nullptr && buflen > 0, "NULL argument");
  const char* k = gemv3_one_pass(variant, n, true) ? "bp_gemv3_pers_kernel" : "single matvecs";
  strncpy(name, k, (size_t)buflen - 1);
  name[buflen - 1] = 0;
  return CCQP_OK;
}

extern "C" int ccqp_gemv3_f64(ccqp_ctx* ctx, const double* A, int64_t lda, const double* const* x, const double* b,
                              double* const* y, double* const* g, int64_t n, int variant, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && A != nullptr && x != nullptr && y != nullptr && g != nullptr, "NULL argument");
  CCQP_REQUIRE(n >= 1 && lda >= n, "bad shape");
  hipStream_t s = (hipStream_t)stream;
  bool aligned = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && lda % 2 == 0;
  int used = 0;
  for (int j = 0; j < 3; ++j) {
    if (y[j] == nullptr && g[j] == nullptr) continue;
    CCQP_REQUIRE(x[j] != nullptr && y[j] != nullptr, "a product needs x and y");
    CCQP_REQUIRE(g[j] == nullptr || b != nullptr, "g requested but b is NULL");
    for (int k = 0; k < 3; ++k)
      CCQP_REQUIRE(y[j] != x[k] && (g[j] == nullptr || g[j] != x[k]), "outputs must not alias the inputs");
    aligned = aligned && (reinterpret_cast<uintptr_t>(x[j]) & 15) == 0;
    ++used;
  }
  CCQP_REQUIRE(used >= 1, "no product asked for");
  if (!gemv3_one_pass(variant, n, aligned)) {
    for (int j = 0; j < 3; ++j) {
      if (y[j] == nullptr) continue;
      const int rc = ccqp_gemv_launch(ctx, A, lda, x[j], b, y[j], g[j], n, n, variant, s, nullptr);
      if (rc) return rc;
    }
    CCQP_LAUNCH_CHECK();
    return CCQP_OK;
  }
  gemv3_one rsv;
  rsv.d.A = A;
  rsv.d.lda = lda;
  rsv.d.b = b;
  const double* some = nullptr;
  for (int j = 0; j < 3; ++j)
    if (y[j] != nullptr) some = x[j];
  for (int j = 0; j < 3; ++j) {
    const bool on = y[j] != nullptr;
    rsv.d.x[j] = on ? x[j] : some;            // (an unused slot repeats a vector: the loads per step are fixed)
    rsv.d.y[j] = on ? y[j] : nullptr;
    rsv.d.g[j] = on ? g[j] : nullptr;
  }
  const int rc = bp_launch_gemv3(ctx, rsv, 1, n, s);
  if (rc) return rc;
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}
