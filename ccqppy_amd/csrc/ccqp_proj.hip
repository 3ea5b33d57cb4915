// Projection operators P_Omega, the fused residual and the MPRGP mask / normal-vector helpers.
// Reference: solution_spaces.py (Identity :111-125, Lower :186-201, Upper :262-277, Box :349-366,
// Sphere :417-435, Cone :470-492, Disjoint :540-560, normal_vector :92-98,146-160,222-236,306-322,
// 389-403) and the residual / isclose call sites of solvers.py.
// Built with -ffp-contract=off: every elementwise expression rounds once per operation like numpy.
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

static int seg_check(const ccqp_seg* seg, bool elementwise_only) {
  CCQP_REQUIRE(seg != nullptr, "seg is NULL");
  CCQP_REQUIRE(seg->dim >= 0 && seg->offset >= 0, "negative offset/dim");
  if (elementwise_only)
    CCQP_REQUIRE(seg->kind >= CCQP_OP_IDENTITY && seg->kind <= CCQP_OP_BOX, "kind is not elementwise");
  else
    CCQP_REQUIRE(seg->kind >= CCQP_OP_IDENTITY && seg->kind <= CCQP_OP_CONE, "unknown kind");
  return CCQP_OK;
}

static inline ccqp_segk seg_by_value(const ccqp_seg* seg) {
  ccqp_segk k;
  k.kind = seg->kind;
  k.flags = seg->flags;
  k.dim = seg->dim;
  k.lb = seg->lb;
  k.ub = seg->ub;
  k.lb_s = seg->lb_s;
  k.ub_s = seg->ub_s;
  return k;
}

// ------------------------------------------------------------------ elementwise P(a*x + b*y)
__global__ __launch_bounds__(CCQP_BLOCK) void proj_elem_kernel(ccqp_segk sg, double a, const double* x,
                                                               double b, const double* y, double* out) {
  proj_elem_body(sg, a, x, b, y, out, blockIdx.x, gridDim.x);
}

extern "C" int ccqp_proj_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, double a, const double* x,
                                         double b, const double* y, double* out, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  int rc = seg_check(seg, true);
  if (rc) return rc;
  if (seg->dim == 0) return CCQP_OK;
  CCQP_REQUIRE(x && out, "NULL vector");
  const ccqp_segk k = seg_by_value(seg);
  hipLaunchKernelGGL(proj_elem_kernel, dim3(vec_grid(k.dim)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, k,
                     a, x + seg->offset, b, y ? y + seg->offset : nullptr, out + seg->offset);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// out = P(a*x + b*y) - z : the SPG search direction d = P(x - alpha*g) - x (solvers.py:937) in one launch
__global__ __launch_bounds__(CCQP_BLOCK) void proj_elem_sub_kernel(ccqp_segk sg, double a, const double* x, double b,
                                                                   const double* y, const double* z, double* out) {
  VEC_LOOP(i, sg.dim) {
    const double t1 = a * x[i];
    const double t2 = b * y[i];
    const double u = t1 + t2;
    const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
    const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
    const double p = ccqp_proj_elem(sg.kind, sg.flags, u, lb, ub);
    out[i] = p - z[i];
  }
}

extern "C" int ccqp_proj_elementwise_sub_f64(ccqp_ctx* ctx, const ccqp_seg* seg, double a, const double* x,
                                             double b, const double* y, const double* z, double* out,
                                             void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  int rc = seg_check(seg, true);
  if (rc) return rc;
  if (seg->dim == 0) return CCQP_OK;
  CCQP_REQUIRE(x && y && z && out, "NULL vector");
  const ccqp_segk k = seg_by_value(seg);
  const int64_t o = seg->offset;
  hipLaunchKernelGGL(proj_elem_sub_kernel, dim3(vec_grid(k.dim)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, k, a,
                     x + o, b, y + o, z + o, out + o);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// ------------------------------------------------------------------ Sphere / Cone, phase 2
// r = sqrt(|x|^2) is taken on the device (IEEE correctly rounded fp64 sqrt; checked bit-for-bit
// against numpy in tests/test_gpu_ops.py).
__global__ __launch_bounds__(CCQP_BLOCK) void normscale_kernel(ccqp_segk sg, const double* x, double* out,
                                                               const double* norm2) {
  const double r = sqrt(*norm2);
  if (sg.kind == CCQP_OP_SPHERE) {
    const double radius = sg.lb_s;
    if (r > radius) {
      VEC_LOOP(i, sg.dim) {
        const double t = radius * x[i];
        out[i] = t / r;
      }
    } else {
      VEC_LOOP(i, sg.dim) out[i] = x[i];
    }
  } else {  // CCQP_OP_CONE, solution_spaces.py:484-492
    const double mu = sg.lb_s;
    const double z = x[sg.dim - 1];
    if (mu * z >= r) {
      VEC_LOOP(i, sg.dim) out[i] = x[i];
    } else if ((-z) / mu >= r) {
      VEC_LOOP(i, sg.dim) out[i] = 0.0;
    } else {
      const double num = z + mu * r;
      const double s = num / sg.ub_s;  // ub_s = mu**2 + 1 evaluated by the host in Python
      VEC_LOOP(i, sg.dim) {
        const double e = (i == sg.dim - 1) ? -mu : x[i] / r;
        out[i] = s * e;
      }
    }
  }
}

extern "C" int ccqp_proj_normscale_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x, double* out,
                                       int norm2_slot, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  int rc = seg_check(seg, false);
  if (rc) return rc;
  CCQP_REQUIRE(seg->kind == CCQP_OP_SPHERE || seg->kind == CCQP_OP_CONE, "kind is not Sphere/Cone");
  CCQP_REQUIRE(ccqp_slot_ok(norm2_slot, 1), "slot outside the slab");
  if (seg->dim == 0) return CCQP_OK;
  CCQP_REQUIRE(x && out, "NULL vector");
  CCQP_REQUIRE(x != out || seg->kind == CCQP_OP_SPHERE, "Cone cannot run in place");
  const ccqp_segk k = seg_by_value(seg);
  hipLaunchKernelGGL(normscale_kernel, dim3(vec_grid(k.dim)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, k,
                     x + seg->offset, out + seg->offset, ctx->slots + norm2_slot);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// ------------------------------------------------------------------ many small blocks, 1 thread each
// Sequential s = fma(v_i, v_i, s): what OpenBLAS ddot does for n <= 15.
__device__ __forceinline__ double seq_norm2(const double* v, int64_t d) {
  double s = 0.0;
  for (int64_t i = 0; i < d; ++i) s = fma(v[i], v[i], s);
  return s;
}

__device__ void small_project(const ccqp_seg& sg, const double* x, double* out) {
  const int64_t d = sg.dim;
  if (sg.kind <= CCQP_OP_BOX) {
    for (int64_t i = 0; i < d; ++i) {
      const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
      const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
      out[i] = ccqp_proj_elem(sg.kind, sg.flags, x[i], lb, ub);
    }
    return;
  }
  const double r = sqrt(seq_norm2(x, d));
  if (sg.kind == CCQP_OP_SPHERE) {
    const double radius = sg.lb_s;
    if (r > radius) {
      for (int64_t i = 0; i < d; ++i) {
        const double t = radius * x[i];
        out[i] = t / r;
      }
    } else {
      for (int64_t i = 0; i < d; ++i) out[i] = x[i];
    }
    return;
  }
  const double mu = sg.lb_s;
  const double z = x[d - 1];
  if (mu * z >= r) {
    for (int64_t i = 0; i < d; ++i) out[i] = x[i];
  } else if ((-z) / mu >= r) {
    for (int64_t i = 0; i < d; ++i) out[i] = 0.0;
  } else {
    const double num = z + mu * r;
    const double s = num / sg.ub_s;
    for (int64_t i = 0; i + 1 < d; ++i) {
      const double e = x[i] / r;
      out[i] = s * e;
    }
    out[d - 1] = s * (-mu);
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void small_segments_kernel(const ccqp_seg* table, int64_t nseg,
                                                                    const double* x, double* out, int mode) {
  VEC_LOOP(t, nseg) {
    const ccqp_seg sg = table[t];
    const double* xs = x + sg.offset;
    double* os = out + sg.offset;
    const int64_t d = sg.dim;
    if (d <= 0) continue;
    if (mode == 0) {
      small_project(sg, xs, os);
      continue;
    }
    // mode 1: normal_vector.  xp = P(x) is staged in `out`, then overwritten with the normal.
    if (sg.kind == CCQP_OP_IDENTITY) {
      for (int64_t i = 0; i < d; ++i) os[i] = 0.0;
      continue;
    }
    small_project(sg, xs, os);
    double dist2 = 0.0;
    for (int64_t i = 0; i < d; ++i) {
      const double df = xs[i] - os[i];
      dist2 = fma(df, df, dist2);
    }
    const bool outside = !ccqp_isclose(sqrt(dist2), 0.0);
    if (sg.kind == CCQP_OP_SPHERE) {
      const double len = sqrt(seq_norm2(os, d));
      const bool on = !outside && ccqp_isclose(len, sg.lb_s);
      for (int64_t i = 0; i < d; ++i) os[i] = on ? os[i] / len : 0.0;
      continue;
    }
    for (int64_t i = 0; i < d; ++i) {
      double nv = 0.0;
      if (!outside) {
        const double xp = os[i];
        const bool has_ub = sg.kind == CCQP_OP_UPPER || sg.kind == CCQP_OP_BOX;
        const bool has_lb = sg.kind == CCQP_OP_LOWER || sg.kind == CCQP_OP_BOX;
        const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
        const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
        if (has_ub && ccqp_isclose(xp, ub)) nv = 1.0;
        else if (has_lb && ccqp_isclose(xp, lb)) nv = -1.0;
      }
      os[i] = nv;
    }
  }
}

extern "C" int ccqp_proj_small_segments_f64(ccqp_ctx* ctx, const ccqp_seg* table, int64_t nseg,
                                            const double* x, double* out, int mode, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(nseg >= 0, "negative nseg");
  CCQP_REQUIRE(mode == 0 || mode == 1, "mode must be 0 or 1");
  if (nseg == 0) return CCQP_OK;
  CCQP_REQUIRE(table && x && out, "NULL pointer");
  CCQP_REQUIRE(x != out, "small-segment kernel cannot run in place");
  hipLaunchKernelGGL(small_segments_kernel, dim3(vec_grid(nseg)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream,
                     table, nseg, x, out, mode);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// ------------------------------------------------------------------ residual^2 (elementwise ops)
__global__ __launch_bounds__(CCQP_BLOCK) void residual_stage1(ccqp_segk sg, const double* x, const double* g,
                                                              double c, double gd, int64_t chunk,
                                                              double* dst, int nblk, double* slot) {
  __shared__ double lds4[4];
  const double s = residual_body(sg, x, g, c, gd, chunk, blockIdx.x, lds4);
  if (threadIdx.x == 0) dst[blockIdx.x] = s;
  if (nblk > 1) ccqp_reduce_finish(dst, nblk, 1, ccqp_red_counters(dst), slot, lds4);
}

extern "C" int ccqp_residual_sq_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x,
                                                const double* g, double c, double gd, int slot,
                                                void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  int rc = seg_check(seg, true);
  if (rc) return rc;
  CCQP_REQUIRE(ccqp_slot_ok(slot, 1), "slot outside the slab");
  CCQP_REQUIRE(seg->dim == 0 || (x && g), "NULL vector");
  hipStream_t s = (hipStream_t)stream;
  const ccqp_segk k = seg_by_value(seg);
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(k.dim, &chunk, &nblk);
  const double* xs = x ? x + seg->offset : x;
  const double* gs = g ? g + seg->offset : g;
  hipLaunchKernelGGL(residual_stage1, dim3(nblk), dim3(CCQP_BLOCK), 0, s, k, xs, gs, c, gd, chunk,
                     nblk == 1 ? ctx->slots + slot : ctx->partials, nblk, ctx->slots + slot);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// ------------------------------------------------------------------ MPRGP helpers
__global__ __launch_bounds__(CCQP_BLOCK) void isclose_mask_mul_kernel(const double* a, const double* b,
                                                                      const double* g, int invert,
                                                                      double* out, int64_t n) {
  VEC_LOOP(i, n) {
    const bool close = ccqp_isclose(a[i], b[i]);
    const double m = (close != (invert != 0)) ? 1.0 : 0.0;  // mask, or 1 - mask
    out[i] = g ? m * g[i] : m;
  }
}

extern "C" int ccqp_isclose_mask_mul_f64(ccqp_ctx* ctx, const double* a, const double* b, const double* g,
                                         int invert, double* out, int64_t n, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(n >= 0, "negative size");
  if (n == 0) return CCQP_OK;
  CCQP_REQUIRE(a && b && out, "NULL vector");
  hipLaunchKernelGGL(isclose_mask_mul_kernel, dim3(vec_grid(n)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream,
                     a, b, g, invert, out, n);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

__global__ __launch_bounds__(CCQP_BLOCK) void count_notclose_stage1(const double* a, const double* b,
                                                                    int64_t n, int64_t chunk, double* dst,
                                                                    int nblk, double* slot) {
  __shared__ double lds4[4];
  const int64_t lo = (int64_t)blockIdx.x * chunk;
  int64_t hi = lo + chunk;
  if (hi > n) hi = n;
  double acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) acc += ccqp_isclose(a[i], b[i]) ? 0.0 : 1.0;
  const double s = ccqp_block_sum(acc, lds4);
  if (threadIdx.x == 0) dst[blockIdx.x] = s;
  if (nblk > 1) ccqp_reduce_finish(dst, nblk, 1, ccqp_red_counters(dst), slot, lds4);
}

// ---- fused dots (+ residual): k dot products whose operands may be DIFFERENCES of two vectors, and
// optionally the Eq. 25 residual^2, in one pass.  Pair j: a = a1[j] - a2[j] (a2[j] NULL: a = a1[j]), same for
// b; every difference is rounded once and every pair keeps its own fma chain over the same thread -> element
// map as dot_body, so slab[slot0 + j] has the bits ccqp_dots_f64 gives on materialised differences, and
// slab[slot0 + k] those of ccqp_residual_sq_elementwise_f64.
struct fused_dots_args {
  const double* a1[CCQP_MAX_DOTS];
  const double* a2[CCQP_MAX_DOTS];
  const double* b1[CCQP_MAX_DOTS];
  const double* b2[CCQP_MAX_DOTS];
};

// grid (nblk, k + RES): row j < k is dot pair j, row k the residual -- the rows run side by side like the
// (nblk, k) grid of ccqp_dots_f64; each row has its own arrival counter.
template <bool RES>
__global__ __launch_bounds__(CCQP_BLOCK) void fused_dots_stage1(fused_dots_args a, int k, ccqp_segk sg,
                                                                const double* rx, const double* rg, double c,
                                                                double gd, int64_t n, int64_t chunk, double* dst,
                                                                int nblk, double* slots, double* rslot) {
  __shared__ double lds4[4];
  const int bx = blockIdx.x, j = blockIdx.y;
  double s;
  if (RES && j == k) {
    s = residual_body(sg, rx, rg, c, gd, chunk, bx, lds4);
  } else {
    const int64_t lo = (int64_t)bx * chunk;
    int64_t hi = lo + chunk;
    if (hi > n) hi = n;
    const double* __restrict__ a1 = a.a1[j];
    const double* __restrict__ a2 = a.a2[j];
    const double* __restrict__ b1 = a.b1[j];
    const double* __restrict__ b2 = a.b2[j];
    double acc = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
      double u = a1[i], v = b1[i];
      if (a2) u = u - a2[i];
      if (b2) v = v - b2[i];
      acc = fma(u, v, acc);
    }
    s = ccqp_block_sum(acc, lds4);
  }
  // nblk == 1: dst is the slab itself (row j -> slot0 + j; the residual row -> its own slot)
  double* out = (RES && j == k) ? rslot : slots + j;
  if (nblk == 1) {
    if (threadIdx.x == 0) *out = s;
    return;
  }
  if (threadIdx.x == 0) dst[(int64_t)j * nblk + bx] = s;
  ccqp_reduce_finish(dst + (int64_t)j * nblk, nblk, 1, ccqp_red_counters(dst) + j, out, lds4);
}

extern "C" int ccqp_fused_dots_f64(ccqp_ctx* ctx, int k, const double* const* a1, const double* const* a2,
                                   const double* const* b1, const double* const* b2, int64_t n, int slot0,
                                   const ccqp_seg* res_seg, const double* res_x, const double* res_g, double c,
                                   double gd, int res_slot, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(k >= 1 && k <= CCQP_MAX_DOTS, "k outside [1, CCQP_MAX_DOTS]");
  CCQP_REQUIRE(n >= 0, "negative size");
  const bool res = res_seg != nullptr;
  CCQP_REQUIRE(ccqp_slot_ok(slot0, k) && (!res || ccqp_slot_ok(res_slot, 1)), "slot range outside the slab");
  CCQP_REQUIRE(!res || res_slot < slot0 || res_slot >= slot0 + k, "residual slot overlaps the dot slots");
  CCQP_REQUIRE(a1 && a2 && b1 && b2, "NULL pointer table");
  fused_dots_args a;
  for (int j = 0; j < CCQP_MAX_DOTS; ++j) {
    a.a1[j] = j < k ? a1[j] : nullptr;
    a.a2[j] = j < k ? a2[j] : nullptr;
    a.b1[j] = j < k ? b1[j] : nullptr;
    a.b2[j] = j < k ? b2[j] : nullptr;
    CCQP_REQUIRE(j >= k || n == 0 || (a.a1[j] && a.b1[j]), "NULL vector");
  }
  ccqp_segk sg;
  memset(&sg, 0, sizeof(sg));
  if (res) {
    int rc = seg_check(res_seg, true);
    if (rc) return rc;
    CCQP_REQUIRE(res_seg->offset == 0 && res_seg->dim == n, "the fused residual works on whole vectors of length n");
    CCQP_REQUIRE(n == 0 || (res_x && res_g), "NULL residual operand");
    sg = seg_by_value(res_seg);
  }
  hipStream_t s = (hipStream_t)stream;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  double* rslot = ctx->slots + (res ? res_slot : 0);
  if (res)
    hipLaunchKernelGGL((fused_dots_stage1<true>), dim3(nblk, k + 1), dim3(CCQP_BLOCK), 0, s, a, k, sg, res_x, res_g,
                       c, gd, n, chunk, ctx->partials, nblk, ctx->slots + slot0, rslot);
  else
    hipLaunchKernelGGL((fused_dots_stage1<false>), dim3(nblk, k), dim3(CCQP_BLOCK), 0, s, a, k, sg, res_x, res_g, c,
                       gd, n, chunk, ctx->partials, nblk, ctx->slots + slot0, rslot);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// grid (nblk): three sums per block -> dst[0*nblk + b], dst[1*nblk + b], dst[2*nblk + b]
__global__ __launch_bounds__(CCQP_BLOCK) void bb_tail_stage1(ccqp_segk sg, const double* xn, const double* xo,
                                                             const double* gn, const double* go, double c,
                                                             double gd, int64_t chunk, double* dst, int nblk,
                                                             double* slots) {
  __shared__ double lds4[4];
  double r, s1, s2;
  bb_tail_body(sg, xn, xo, gn, go, c, gd, chunk, blockIdx.x, lds4, &r, &s1, &s2);
  if (threadIdx.x == 0) {
    dst[blockIdx.x] = r;
    dst[nblk + blockIdx.x] = s1;
    dst[2 * nblk + blockIdx.x] = s2;
  }
  if (nblk > 1) ccqp_reduce_finish(dst, nblk, 3, ccqp_red_counters(dst), slots, lds4);
}

extern "C" int ccqp_bb_tail_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x_new,
                                            const double* x_old, const double* g_new, const double* g_old,
                                            double c, double gd, int slot0, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  int rc = seg_check(seg, true);
  if (rc) return rc;
  CCQP_REQUIRE(ccqp_slot_ok(slot0, 3), "slot range outside the slab");
  CCQP_REQUIRE(seg->offset == 0, "the fused tail works on whole vectors (offset must be 0)");
  CCQP_REQUIRE(seg->dim == 0 || (x_new && x_old && g_new && g_old), "NULL vector");
  hipStream_t s = (hipStream_t)stream;
  const ccqp_segk k = seg_by_value(seg);
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(k.dim, &chunk, &nblk);
  hipLaunchKernelGGL(bb_tail_stage1, dim3(nblk), dim3(CCQP_BLOCK), 0, s, k, x_new, x_old, g_new, g_old, c, gd, chunk,
                     nblk == 1 ? ctx->slots + slot0 : ctx->partials, nblk, ctx->slots + slot0);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_count_notclose_f64(ccqp_ctx* ctx, const double* a, const double* b, int64_t n, int slot,
                                       void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(n >= 0, "negative size");
  CCQP_REQUIRE(ccqp_slot_ok(slot, 1), "slot outside the slab");
  CCQP_REQUIRE(n == 0 || (a && b), "NULL vector");
  hipStream_t s = (hipStream_t)stream;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  hipLaunchKernelGGL(count_notclose_stage1, dim3(nblk), dim3(CCQP_BLOCK), 0, s, a, b, n, chunk,
                     nblk == 1 ? ctx->slots + slot : ctx->partials, nblk, ctx->slots + slot);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// grid (nblk, K): block (b, k) counts the not-close entries of yf = x - alpha_k*p over its chunk.
__global__ __launch_bounds__(CCQP_BLOCK) void feasible_steps_stage1(ccqp_segk sg, const double* x,
                                                                    const double* p, double alpha0,
                                                                    int64_t chunk, double* dst, int nblk,
                                                                    double* slots) {
  __shared__ double lds4[4];
  const int k = blockIdx.y;
  const double s = feasible_body(sg, x, p, alpha0, k, chunk, blockIdx.x, lds4);
  if (threadIdx.x == 0) dst[(int64_t)k * nblk + blockIdx.x] = s;
  if (nblk > 1) ccqp_reduce_finish(dst + (int64_t)k * nblk, nblk, 1, ccqp_red_counters(dst) + k, slots + k, lds4);
}

extern "C" int ccqp_feasible_steps_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x,
                                                   const double* p, double alpha0, int K, int slot0,
                                                   void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  int rc = seg_check(seg, true);
  if (rc) return rc;
  CCQP_REQUIRE(K >= 1 && K <= CCQP_MAX_STEPS, "K outside [1, CCQP_MAX_STEPS]");
  CCQP_REQUIRE(ccqp_slot_ok(slot0, K), "slot range outside the slab");
  CCQP_REQUIRE(seg->dim == 0 || (x && p), "NULL vector");
  hipStream_t s = (hipStream_t)stream;
  const ccqp_segk k = seg_by_value(seg);
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(k.dim, &chunk, &nblk);
  // partials area holds CCQP_RED_ROWS * CCQP_RED_MAXBLK doubles: K * nblk must fit
  CCQP_REQUIRE((int64_t)K * nblk <= (int64_t)CCQP_RED_ROWS * CCQP_RED_MAXBLK, "K * blocks exceeds the workspace");
  const double* xs = x ? x + seg->offset : x;
  const double* ps = p ? p + seg->offset : p;
  hipLaunchKernelGGL(feasible_steps_stage1, dim3(nblk, K), dim3(CCQP_BLOCK), 0, s, k, xs, ps, alpha0, chunk,
                     nblk == 1 ? ctx->slots + slot0 : ctx->partials, nblk, ctx->slots + slot0);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

__global__ __launch_bounds__(CCQP_BLOCK) void normal_elem_kernel(ccqp_segk sg, const double* xp, double* out,
                                                                 const double* dist2) {
  // `not np.isclose(np.linalg.norm(x - x_proj), 0)` -> zeros (solution_spaces.py:313-314)
  const bool outside = !ccqp_isclose(sqrt(*dist2), 0.0);
  const bool has_ub = sg.kind == CCQP_OP_UPPER || sg.kind == CCQP_OP_BOX;
  const bool has_lb = sg.kind == CCQP_OP_LOWER || sg.kind == CCQP_OP_BOX;
  VEC_LOOP(i, sg.dim) {
    double nv = 0.0;
    if (!outside) {
      const double v = xp[i];
      const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
      const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
      if (has_ub && ccqp_isclose(v, ub)) nv = 1.0;
      else if (has_lb && ccqp_isclose(v, lb)) nv = -1.0;
    }
    out[i] = nv;
  }
}

extern "C" int ccqp_normal_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* xp, double* out,
                                           int dist2_slot, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  int rc = seg_check(seg, true);
  if (rc) return rc;
  CCQP_REQUIRE(ccqp_slot_ok(dist2_slot, 1), "slot outside the slab");
  if (seg->dim == 0) return CCQP_OK;
  CCQP_REQUIRE(xp && out, "NULL vector");
  const ccqp_segk k = seg_by_value(seg);
  hipLaunchKernelGGL(normal_elem_kernel, dim3(vec_grid(k.dim)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, k,
                     xp + seg->offset, out + seg->offset, ctx->slots + dist2_slot);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

__global__ __launch_bounds__(CCQP_BLOCK) void normal_sphere_kernel(ccqp_segk sg, const double* xp, double* out,
                                                                   const double* dist2, const double* len2) {
  const bool outside = !ccqp_isclose(sqrt(*dist2), 0.0);
  const double len = sqrt(*len2);
  const bool on = !outside && ccqp_isclose(len, sg.lb_s);
  VEC_LOOP(i, sg.dim) out[i] = on ? xp[i] / len : 0.0;
}

extern "C" int ccqp_normal_sphere_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* xp, double* out,
                                      int dist2_slot, int len2_slot, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  int rc = seg_check(seg, false);
  if (rc) return rc;
  CCQP_REQUIRE(seg->kind == CCQP_OP_SPHERE, "kind is not Sphere");
  CCQP_REQUIRE(ccqp_slot_ok(dist2_slot, 1) && ccqp_slot_ok(len2_slot, 1), "slot outside the slab");
  if (seg->dim == 0) return CCQP_OK;
  CCQP_REQUIRE(xp && out, "NULL vector");
  const ccqp_segk k = seg_by_value(seg);
  hipLaunchKernelGGL(normal_sphere_kernel, dim3(vec_grid(k.dim)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, k,
                     xp + seg->offset, out + seg->offset, ctx->slots + dist2_slot, ctx->slots + len2_slot);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

__global__ __launch_bounds__(CCQP_BLOCK) void chopped_gradient_kernel(const double* w, const double* g,
                                                                      const double* nv, const double* ng,
                                                                      double* out, int64_t n) {
  const double v = *ng;
  // np.min([0, v]): NaN propagates, otherwise the smaller of 0 and v
  const double s = (v != v) ? v : (v < 0.0 ? v : 0.0);
  VEC_LOOP(i, n) {
    const double t = s * nv[i];
    const double u = g[i] - t;
    out[i] = w[i] * u;
  }
}

extern "C" int ccqp_chopped_gradient_f64(ccqp_ctx* ctx, const double* w, const double* g, const double* nv,
                                         int ng_slot, double* out, int64_t n, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(n >= 0, "negative size");
  CCQP_REQUIRE(ccqp_slot_ok(ng_slot, 1), "slot outside the slab");
  if (n == 0) return CCQP_OK;
  CCQP_REQUIRE(w && g && nv && out, "NULL vector");
  hipLaunchKernelGGL(chopped_gradient_kernel, dim3(vec_grid(n)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, w,
                     g, nv, ctx->slots + ng_slot, out, n);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// ------------------------------------------------------------------ projected_gradient (8(f)2)
// Free / chopped gradient split of the ELEMENTWISE kinds, solution_spaces.py:162-184, 238-260,
// 324-347 -- including the Box condition's misplaced parenthesis (:340-341):
//   isclose(x, ub) or x > ub or isclose(x, (lb or x < ub))
// where Python's `or` yields lb unless lb == 0, else the boolean (x < ub) as 1.0 / 0.0.
__global__ __launch_bounds__(CCQP_BLOCK) void projected_gradient_kernel(ccqp_segk sg, const double* x,
                                                                        const double* g, const double* nv,
                                                                        double* free_g, double* chopped_g) {
  VEC_LOOP(i, sg.dim) {
    const double xi = x[i];
    const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
    const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
    bool active;
    if (sg.kind == CCQP_OP_LOWER) {
      active = ccqp_isclose(xi, lb);
    } else if (sg.kind == CCQP_OP_UPPER) {
      active = ccqp_isclose(xi, ub);
    } else {
      const double second = (lb != 0.0) ? lb : ((xi < ub) ? 1.0 : 0.0);  // NaN != 0 -> truthy, like Python
      active = ccqp_isclose(xi, ub) || (xi > ub) || ccqp_isclose(xi, second);
    }
    double fr = 0.0, ch = 0.0;
    if (active) {
      const double n = nv[i];
      const double m = n * g[i];
      const double mn = (m < 0.0 || m != m) ? m : 0.0;  // np.min((m, 0))
      const double t = mn * n;
      ch = g[i] - t;
    } else {
      fr = g[i];
    }
    free_g[i] = fr;
    chopped_g[i] = ch;
  }
}

extern "C" int ccqp_projected_gradient_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x,
                                                       const double* g, const double* nv, double* free_g,
                                                       double* chopped_g, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  int rc = seg_check(seg, true);
  if (rc) return rc;
  CCQP_REQUIRE(seg->kind != CCQP_OP_IDENTITY, "Identity has no projected gradient");
  if (seg->dim == 0) return CCQP_OK;
  CCQP_REQUIRE(x && g && nv && free_g && chopped_g, "NULL vector");
  const ccqp_segk k = seg_by_value(seg);
  const int64_t o = seg->offset;
  hipLaunchKernelGGL(projected_gradient_kernel, dim3(vec_grid(k.dim)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, k,
                     x + o, g + o, nv + o, free_g + o, chopped_g + o);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}
