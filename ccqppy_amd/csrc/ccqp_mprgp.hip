// Fused MPRGP steps for ELEMENTWISE feasible sets (Identity / Lower / Upper / Box).
// Reference: solvers.py:1081-1098 (proportioning test), :1132-1135 (CG direction update).
// At n = 8192 (BASELINE config 5) one matvec is 83 us, so the ~25 tiny launches the literal
// operation sequence needs per iteration dominate; these kernels perform the same arithmetic, in
// the same order, with the same reduction geometry (bitwise identical results, asserted in
// tests/test_gpu_batch.py) in 5 launches instead of 14.
#include "ccqp_bodies.h"


static inline int vec_grid(int64_t n) {
  int64_t g = (n + CCQP_BLOCK - 1) / CCQP_BLOCK;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

static inline ccqp_segk segk(const ccqp_seg* seg) {
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

static int elementwise_seg_ok(const ccqp_seg* seg) {
  CCQP_REQUIRE(seg != nullptr, "seg is NULL");
  CCQP_REQUIRE(seg->dim >= 0 && seg->offset >= 0, "negative offset/dim");
  CCQP_REQUIRE(seg->kind >= CCQP_OP_IDENTITY && seg->kind <= CCQP_OP_BOX, "kind is not elementwise");
  return CCQP_OK;
}

// ---- K1: xp = P(x); delta = isclose(x, xp); psi = delta*g; w = 1-delta; sum (x-xp)^2 -> dst
__global__ __launch_bounds__(CCQP_BLOCK) void split_stage1(ccqp_segk sg, const double* x, const double* g,
                                                           double* xp, double* psi, double* w, int64_t chunk,
                                                           double* dst, int nblk, double* slot) {
  __shared__ double lds4[4];
  const double s = split_body(sg, x, g, xp, psi, w, chunk, blockIdx.x, lds4);
  if (threadIdx.x == 0) dst[blockIdx.x] = s;
  if (nblk > 1) ccqp_reduce_finish(dst, nblk, 1, ccqp_red_counters(dst), slot, lds4);
}

extern "C" int ccqp_mprgp_split_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x, const double* g,
                                    double* xp, double* psi, double* w, int dist2_slot, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  int rc = elementwise_seg_ok(seg);
  if (rc) return rc;
  CCQP_REQUIRE(ccqp_slot_ok(dist2_slot, 1), "slot outside the slab");
  CCQP_REQUIRE(seg->dim == 0 || (x && g && xp && psi && w), "NULL vector");
  hipStream_t s = (hipStream_t)stream;
  const ccqp_segk k = segk(seg);
  const int64_t o = seg->offset;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(k.dim, &chunk, &nblk);
  double* dst = nblk == 1 ? ctx->slots + dist2_slot : ctx->partials;
  hipLaunchKernelGGL(split_stage1, dim3(nblk), dim3(CCQP_BLOCK), 0, s, k, x + o, g + o, xp + o, psi + o, w + o,
                     chunk, dst, nblk, ctx->slots + dist2_slot);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// ---- K2: nv = normal_vector from xp (+ the "outside => zeros" rule); sum nv*g -> dst
__global__ __launch_bounds__(CCQP_BLOCK) void normal_dot_stage1(ccqp_segk sg, const double* xp, const double* g,
                                                                double* nvout, const double* dist2,
                                                                int64_t chunk, double* dst, int nblk,
                                                                double* slot) {
  __shared__ double lds4[4];
  const double s = normal_dot_body(sg, xp, g, nvout, *dist2, chunk, blockIdx.x, lds4);
  if (threadIdx.x == 0) dst[blockIdx.x] = s;
  if (nblk > 1) ccqp_reduce_finish(dst, nblk, 1, ccqp_red_counters(dst), slot, lds4);
}

extern "C" int ccqp_mprgp_normal_dot_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* xp, const double* g,
                                         double* nv, int dist2_slot, int ng_slot, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  int rc = elementwise_seg_ok(seg);
  if (rc) return rc;
  CCQP_REQUIRE(ccqp_slot_ok(dist2_slot, 1) && ccqp_slot_ok(ng_slot, 1), "slot outside the slab");
  CCQP_REQUIRE(seg->dim == 0 || (xp && g && nv), "NULL vector");
  hipStream_t s = (hipStream_t)stream;
  const ccqp_segk k = segk(seg);
  const int64_t o = seg->offset;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(k.dim, &chunk, &nblk);
  double* dst = nblk == 1 ? ctx->slots + ng_slot : ctx->partials;
  hipLaunchKernelGGL(normal_dot_stage1, dim3(nblk), dim3(CCQP_BLOCK), 0, s, k, xp + o, g + o, nv + o,
                     ctx->slots + dist2_slot, chunk, dst, nblk, ctx->slots + ng_slot);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// ---- K3: beta = w*(g - min(0, ng)*nv);  (sum beta^2, sum psi^2) -> dst[0..1]
__global__ __launch_bounds__(CCQP_BLOCK) void proportion_stage1(const double* w, const double* g,
                                                                const double* nv, const double* psi,
                                                                const double* ng, int64_t n, int64_t chunk,
                                                                double* dst, int nblk, double* slots) {
  __shared__ double lds4[4];
  double sb, sp;
  proportion_body(w, g, nv, psi, *ng, n, chunk, blockIdx.x, lds4, &sb, &sp);
  if (threadIdx.x == 0) {
    dst[blockIdx.x] = sb;
    dst[nblk + blockIdx.x] = sp;
  }
  if (nblk > 1) ccqp_reduce_finish(dst, nblk, 2, ccqp_red_counters(dst), slots, lds4);
}

extern "C" int ccqp_mprgp_proportion_f64(ccqp_ctx* ctx, const double* w, const double* g, const double* nv,
                                         const double* psi, int ng_slot, int64_t n, int slot0, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(n >= 0, "negative size");
  CCQP_REQUIRE(ccqp_slot_ok(ng_slot, 1) && ccqp_slot_ok(slot0, 2), "slot outside the slab");
  CCQP_REQUIRE(n == 0 || (w && g && nv && psi), "NULL vector");
  hipStream_t s = (hipStream_t)stream;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  double* dst = nblk == 1 ? ctx->slots + slot0 : ctx->partials;
  hipLaunchKernelGGL(proportion_stage1, dim3(nblk), dim3(CCQP_BLOCK), 0, s, w, g, nv, psi, ctx->slots + ng_slot, n,
                     chunk, dst, nblk, ctx->slots + slot0);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// ---- out = isclose(x, P(x)) * g   (free gradient)
__global__ __launch_bounds__(CCQP_BLOCK) void free_gradient_kernel(ccqp_segk sg, const double* x, const double* g,
                                                                   double* out) {
  free_gradient_body(sg, x, g, out, blockIdx.x, gridDim.x);
}

extern "C" int ccqp_free_gradient_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x,
                                                  const double* g, double* out, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  int rc = elementwise_seg_ok(seg);
  if (rc) return rc;
  if (seg->dim == 0) return CCQP_OK;
  CCQP_REQUIRE(x && g && out, "NULL vector");
  const ccqp_segk k = segk(seg);
  const int64_t o = seg->offset;
  hipLaunchKernelGGL(free_gradient_kernel, dim3(vec_grid(k.dim)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, k,
                     x + o, g + o, out + o);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// ---- out = psi - ((psi*Ap)/pAp)*p   (solvers.py:1134-1135, four roundings)
__global__ __launch_bounds__(CCQP_BLOCK) void cg_direction_kernel(const double* psi, const double* Ap, double pAp,
                                                                  const double* p, double* out, int64_t n) {
  cg_direction_body(psi, Ap, pAp, p, out, n, blockIdx.x, gridDim.x);
}

extern "C" int ccqp_cg_direction_f64(ccqp_ctx* ctx, const double* psi, const double* Ap, double pAp,
                                     const double* p, double* out, int64_t n, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(n >= 0, "negative size");
  if (n == 0) return CCQP_OK;
  CCQP_REQUIRE(psi && Ap && p && out, "NULL vector");
  hipLaunchKernelGGL(cg_direction_kernel, dim3(vec_grid(n)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, psi, Ap, pAp,
                     p, out, n);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}
