// Segmented projection engine: P, P(a*x + b*y) (- z), the Eq. 25 residual and normal_vector of ANY feasible set
// -- Sphere, Cone, DisjointUnion with blocks of every size -- in at most 2 (normal_vector: 3) launches, whatever
// the number of blocks.
//
// Reference: DisjointProjOp.__call__ (solution_spaces.py:540-560) loops over its blocks in Python; with 1365
// Sphere(3) blocks that loop is 85 % of a reference solve.  Round 1 put the blocks of dimension
// <= 64 into one launch and gave every larger Sphere / Cone block two launches of its own, i.e. 2000 launches for
// DisjointProjOp(*[SphereProjOp(100)] * 1000).  Here the set is flattened to a device table of `ccqp_seg` and a
// WORK list whose items are one workgroup each:
//   type 0  a run of up to 256 consecutive SMALL blocks (dim <= CCQP_SMALL_SEG_MAX), one thread per block, norms by
//           the sequential chain s = fma(x_i, x_i, s) (= OpenBLAS ddot for n <= 15: bit-exact against numpy);
//   type 1  one 2048-element chunk of a LARGE block (the stage-1 geometry of every reduction, ccqp_red_geometry).
// Launch 1 (norm items only) leaves the chunk sums of |u|^2 in the workspace; launch 2 (all items) finishes each
// block's sum -- every workgroup of a block adds that block's chunk sums itself, in index order, so no atomics and
// no third launch -- and applies the set's branches.  The arithmetic per element and the order of every sum are
// those of the round-1 kernels (dots_stage1 + last-block finish + normscale_kernel, small_segments_kernel,
// proj_elem_kernel), hence the same bits: the 408 operator goldens are unchanged.
// Built with -ffp-contract=off: one rounding per operation, like numpy.
#include "ccqp_segs.h"

static inline bool seg_is_norm(int kind) { return kind == CCQP_OP_SPHERE || kind == CCQP_OP_CONE; }

// ------------------------------------------------------------------ plan (host only, no GPU involved)
extern "C" int ccqp_segs_plan(const ccqp_seg* segs, int64_t nseg, ccqp_segwork* work, int64_t work_cap,
                              int64_t* nwork_out, ccqp_segwork* normwork, int64_t norm_cap, int64_t* nnorm_out,
                              int64_t* npartials_out) {
  CCQP_REQUIRE(nseg >= 0, "negative nseg");
  CCQP_REQUIRE(nseg == 0 || segs != nullptr, "segs is NULL");
  CCQP_REQUIRE(nwork_out && nnorm_out && npartials_out, "NULL output");
  int64_t nwork = 0, nnorm = 0, npart = 0;
  int64_t run0 = -1, runlen = 0;
  auto close_run = [&]() {
    if (runlen > 0) {
      if (work && nwork < work_cap) {
        work[nwork].type = 0;
        work[nwork].seg = (int32_t)run0;
        work[nwork].chunk = (int32_t)runlen;
        work[nwork].pbase = 0;
      }
      ++nwork;
    }
    run0 = -1;
    runlen = 0;
  };
  for (int64_t s = 0; s < nseg; ++s) {
    const ccqp_seg& sg = segs[s];
    CCQP_REQUIRE(sg.dim >= 0 && sg.offset >= 0, "negative offset/dim");
    CCQP_REQUIRE(sg.kind >= CCQP_OP_IDENTITY && sg.kind <= CCQP_OP_CONE, "unknown kind");
    CCQP_REQUIRE(s < INT32_MAX, "too many segments");
    if (sg.dim <= CCQP_SMALL_SEG_MAX) {
      // a run = up to 256 CONSECUTIVE table entries, all small (the kernel skips the empty ones)
      if (runlen == 0) run0 = s;
      if (++runlen == CCQP_BLOCK) close_run();
      continue;
    }
    close_run();
    int64_t chunk;
    int nblk;
    segs_geometry(sg.dim, &chunk, &nblk);
    CCQP_REQUIRE(npart + nblk < INT32_MAX, "too many chunks");
    for (int c = 0; c < nblk; ++c) {
      if (work && nwork < work_cap) {
        work[nwork].type = 1;
        work[nwork].seg = (int32_t)s;
        work[nwork].chunk = c;
        work[nwork].pbase = (int32_t)npart;
      }
      ++nwork;
      if (seg_is_norm(sg.kind)) {
        if (normwork && nnorm < norm_cap) {
          normwork[nnorm].type = 1;
          normwork[nnorm].seg = (int32_t)s;
          normwork[nnorm].chunk = c;
          normwork[nnorm].pbase = (int32_t)npart;
        }
        ++nnorm;
      }
    }
    npart += nblk;
  }
  close_run();
  *nwork_out = nwork;
  *nnorm_out = nnorm;
  *npartials_out = npart;
  if (work) CCQP_REQUIRE(nwork <= work_cap, "work_cap too small");
  if (normwork) CCQP_REQUIRE(nnorm <= norm_cap, "norm_cap too small");
  return CCQP_OK;
}

// workspace (doubles): 3 arrays of npartials chunk sums (|u|^2 | |x - xp|^2 | |xp|^2) + nwork residual partials -- or,
// for the feasible-step search, CCQP_MAX_STEPS x (npartials chunk sums + nwork counts) -- then 8 doubles of arrival
// counters (zero on entry, left at zero)
static inline int64_t segs_counter_offset(int64_t npartials, int64_t nwork) {
  const int64_t a = 3 * npartials + nwork, b = (int64_t)CCQP_MAX_STEPS * (npartials + nwork);
  return a > b ? a : b;
}

extern "C" int64_t ccqp_segs_workspace_doubles(int64_t npartials, int64_t nwork) {
  if (npartials < 0 || nwork < 0) return -1;
  return segs_counter_offset(npartials, nwork) + 8;
}

// ------------------------------------------------------------------ device side (helpers: ccqp_segs.h)
// launch 1: chunk sums of |u|^2 for the Sphere / Cone blocks
__global__ __launch_bounds__(CCQP_BLOCK) void segs_norm_kernel(const ccqp_seg* __restrict__ table,
                                                               const ccqp_segwork* __restrict__ normwork,
                                                               double* __restrict__ part, double a, const double* x,
                                                               double b, const double* y) {
  __shared__ double lds4[4];
  const ccqp_segwork w = normwork[blockIdx.x];
  const double s = segs_norm_item(table, w, a, x, b, y, lds4);
  if (threadIdx.x == 0) part[w.pbase + w.chunk] = s;
}

// launch 2.  EPI 0: out = P(u) (- z);  EPI 1: residual^2 of (x, g) = sum (c*(x - P(x - gd*g)))^2 -> slab slot
template <int EPI>
__global__ __launch_bounds__(CCQP_BLOCK) void segs_apply_kernel(const ccqp_seg* __restrict__ table,
                                                                const ccqp_segwork* __restrict__ work,
                                                                const double* __restrict__ part, double a,
                                                                const double* x, double b, const double* y,
                                                                const double* z, double* out, double c,
                                                                double* respart, int nwork, unsigned* counter,
                                                                double* slot) {
  __shared__ double lds4[4];
  __shared__ double bcast;
  const ccqp_segwork w = work[blockIdx.x];
  const double racc = segs_apply_item<EPI>(table, w, part, a, x, b, y, z, out, c, lds4, &bcast);
  if constexpr (EPI == 1) {
    const double s = ccqp_block_sum(racc, lds4);
    if (nwork == 1) {
      if (threadIdx.x == 0) *slot = s;
      return;
    }
    if (threadIdx.x == 0) respart[blockIdx.x] = s;
    ccqp_reduce_finish(respart, nwork, 1, counter, slot, lds4);
  }
}

// normal_vector, launch 2 of 3: xp = P(x) into `out`; chunk sums of |x - xp|^2 (and |xp|^2 for Spheres).  Small blocks
// finish here (small_segments_kernel mode 1: solution_spaces.py:146-160,222-236,306-322,389-403,92-98).
__global__ __launch_bounds__(CCQP_BLOCK) void segs_normal_project_kernel(const ccqp_seg* __restrict__ table,
                                                                         const ccqp_segwork* __restrict__ work,
                                                                         const double* __restrict__ part,
                                                                         double* __restrict__ dpart,
                                                                         double* __restrict__ lpart, const double* x,
                                                                         double* out) {
  __shared__ double lds4[4];
  __shared__ double bcast;
  const ccqp_segwork w = work[blockIdx.x];
  segs_no_extra none;
  segs_normal_project_item(table, w, part, dpart, lpart, x, out, none, lds4, &bcast);
}

// normal_vector, launch 3 of 3: large blocks only, in place on `out` (which holds xp)
__global__ __launch_bounds__(CCQP_BLOCK) void segs_normal_finish_kernel(const ccqp_seg* __restrict__ table,
                                                                        const ccqp_segwork* __restrict__ work,
                                                                        const double* __restrict__ dpart,
                                                                        const double* __restrict__ lpart,
                                                                        double* out) {
  __shared__ double lds4[4];
  __shared__ double bcast;
  const ccqp_segwork w = work[blockIdx.x];
  segs_normal_finish_item(table, w, dpart, lpart, out, lds4, &bcast);
}

// ------------------------------------------------------------------ C ABI
static int segs_check(ccqp_ctx* ctx, const ccqp_seg* table, const ccqp_segwork* work, int64_t nwork,
                      const ccqp_segwork* normwork, int64_t nnorm, int64_t npartials, const double* ws) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(nwork >= 0 && nnorm >= 0 && npartials >= 0, "negative count");
  CCQP_REQUIRE(nwork < INT32_MAX, "too many work items");
  CCQP_REQUIRE(nwork == 0 || (table && work && ws), "NULL table / work list / workspace");
  CCQP_REQUIRE(nnorm == 0 || normwork, "NULL norm work list");
  return CCQP_OK;
}

static inline unsigned* segs_counter(double* ws, int64_t npartials, int64_t nwork) {
  return reinterpret_cast<unsigned*>(ws + segs_counter_offset(npartials, nwork));
}

extern "C" int ccqp_segs_project_f64(ccqp_ctx* ctx, const ccqp_seg* table, const ccqp_segwork* work, int64_t nwork,
                                     const ccqp_segwork* normwork, int64_t nnorm, int64_t npartials, double* ws,
                                     double a, const double* x, double b, const double* y, const double* z,
                                     double* out, void* stream) {
  int rc = segs_check(ctx, table, work, nwork, normwork, nnorm, npartials, ws);
  if (rc) return rc;
  if (nwork == 0) return CCQP_OK;
  CCQP_REQUIRE(x && out, "NULL vector");
  CCQP_REQUIRE(x != out && y != out, "the segmented projection cannot run in place");
  hipStream_t s = (hipStream_t)stream;
  if (nnorm > 0)
    hipLaunchKernelGGL(segs_norm_kernel, dim3((unsigned)nnorm), dim3(CCQP_BLOCK), 0, s, table, normwork, ws, a, x, b, y);
  hipLaunchKernelGGL((segs_apply_kernel<0>), dim3((unsigned)nwork), dim3(CCQP_BLOCK), 0, s, table, work, ws, a, x, b, y,
                     z, out, 0.0, (double*)nullptr, (int)nwork, (unsigned*)nullptr, (double*)nullptr);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_segs_residual_sq_f64(ccqp_ctx* ctx, const ccqp_seg* table, const ccqp_segwork* work,
                                         int64_t nwork, const ccqp_segwork* normwork, int64_t nnorm,
                                         int64_t npartials, double* ws, const double* x, const double* g, double c,
                                         double gd, int slot, void* stream) {
  int rc = segs_check(ctx, table, work, nwork, normwork, nnorm, npartials, ws);
  if (rc) return rc;
  CCQP_REQUIRE(ccqp_slot_ok(slot, 1), "slot outside the slab");
  hipStream_t s = (hipStream_t)stream;
  if (nwork == 0) {
    CCQP_HIP(hipMemsetAsync(ctx->slots + slot, 0, sizeof(double), s));
    return CCQP_OK;
  }
  CCQP_REQUIRE(x && g, "NULL vector");
  // u = x - gd*g  ==  1.0*x + (-gd)*g, rounded like the axpby it replaces
  if (nnorm > 0)
    hipLaunchKernelGGL(segs_norm_kernel, dim3((unsigned)nnorm), dim3(CCQP_BLOCK), 0, s, table, normwork, ws, 1.0, x, -gd,
                       g);
  hipLaunchKernelGGL((segs_apply_kernel<1>), dim3((unsigned)nwork), dim3(CCQP_BLOCK), 0, s, table, work, ws, 1.0, x,
                     -gd, g, (const double*)nullptr, (double*)nullptr, c, ws + 3 * npartials, (int)nwork,
                     segs_counter(ws, npartials, nwork), ctx->slots + slot);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_segs_normal_f64(ccqp_ctx* ctx, const ccqp_seg* table, const ccqp_segwork* work, int64_t nwork,
                                    const ccqp_segwork* normwork, int64_t nnorm, int64_t npartials, double* ws,
                                    const double* x, double* out, void* stream) {
  int rc = segs_check(ctx, table, work, nwork, normwork, nnorm, npartials, ws);
  if (rc) return rc;
  if (nwork == 0) return CCQP_OK;
  CCQP_REQUIRE(x && out, "NULL vector");
  CCQP_REQUIRE(x != out, "the segmented normal_vector cannot run in place");
  hipStream_t s = (hipStream_t)stream;
  if (nnorm > 0)
    hipLaunchKernelGGL(segs_norm_kernel, dim3((unsigned)nnorm), dim3(CCQP_BLOCK), 0, s, table, normwork, ws, 1.0, x, 0.0,
                       (const double*)nullptr);
  hipLaunchKernelGGL(segs_normal_project_kernel, dim3((unsigned)nwork), dim3(CCQP_BLOCK), 0, s, table, work, ws,
                     ws + npartials, ws + 2 * npartials, x, out);
  if (npartials > 0)
    hipLaunchKernelGGL(segs_normal_finish_kernel, dim3((unsigned)nwork), dim3(CCQP_BLOCK), 0, s, table, work,
                       ws + npartials, ws + 2 * npartials, out);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}


// ------------------------------------------------------------------ MPRGP's feasible-step search on any set
// solvers.py:1112-1118 halves alpha_f until np.all(np.isclose(yf, P(yf))), yf = x - alpha_f*p -- one projection and one
// host synchronisation per candidate in the reference's form.  Here K candidates alpha0 * 0.5^k (the values the loop
// visits: *0.5 is exact) are evaluated in 2 launches: chunk sums of |yf_k|^2 for the Sphere / Cone blocks (grid
// (norm items, K)), then per work item the count of entries that are not close (grid (items, K)); counts are small
// integers in fp64, so their sum does not depend on the order.  slab[slot0 + k] = the count of candidate k.
__global__ __launch_bounds__(CCQP_BLOCK) void segs_feasible_norm_kernel(const ccqp_seg* __restrict__ table,
                                                                        const ccqp_segwork* __restrict__ normwork,
                                                                        double* __restrict__ ws, int64_t npartials,
                                                                        const double* x, const double* p,
                                                                        double alpha0) {
  __shared__ double lds4[4];
  const int k = blockIdx.y;
  double alpha = alpha0;
  for (int j = 0; j < k; ++j) alpha *= 0.5;
  const ccqp_segwork w = normwork[blockIdx.x];
  const double s = segs_norm_item(table, w, 1.0, x, -alpha, p, lds4);
  if (threadIdx.x == 0) ws[(int64_t)k * npartials + w.pbase + w.chunk] = s;
}

__global__ __launch_bounds__(CCQP_BLOCK) void segs_feasible_count_kernel(const ccqp_seg* __restrict__ table,
                                                                         const ccqp_segwork* __restrict__ work,
                                                                         double* __restrict__ ws, int64_t npartials,
                                                                         int nwork, int K, const double* x,
                                                                         const double* p, double alpha0,
                                                                         unsigned* counters, double* slots) {
  __shared__ double lds4[4];
  __shared__ double bcast;
  const int k = blockIdx.y;
  double alpha = alpha0;
  for (int j = 0; j < k; ++j) alpha *= 0.5;
  const ccqp_segwork w = work[blockIdx.x];
  const double racc = segs_apply_item<2>(table, w, ws + (int64_t)k * npartials, 1.0, x, -alpha, p, nullptr, nullptr, 0.0,
                                         lds4, &bcast);
  const double s = ccqp_block_sum(racc, lds4);
  if (nwork == 1) {
    if (threadIdx.x == 0) slots[k] = s;
    return;
  }
  double* cnt = ws + (int64_t)K * npartials + (int64_t)k * nwork;
  if (threadIdx.x == 0) cnt[blockIdx.x] = s;
  ccqp_reduce_finish(cnt, nwork, 1, counters + k, slots + k, lds4);
}

extern "C" int ccqp_segs_feasible_steps_f64(ccqp_ctx* ctx, const ccqp_seg* table, const ccqp_segwork* work,
                                            int64_t nwork, const ccqp_segwork* normwork, int64_t nnorm,
                                            int64_t npartials, double* ws, const double* x, const double* p,
                                            double alpha0, int K, int slot0, void* stream) {
  int rc = segs_check(ctx, table, work, nwork, normwork, nnorm, npartials, ws);
  if (rc) return rc;
  CCQP_REQUIRE(K >= 1 && K <= CCQP_MAX_STEPS, "K outside [1, CCQP_MAX_STEPS]");
  CCQP_REQUIRE(ccqp_slot_ok(slot0, K), "slot range outside the slab");
  hipStream_t s = (hipStream_t)stream;
  if (nwork == 0) {
    CCQP_HIP(hipMemsetAsync(ctx->slots + slot0, 0, (size_t)K * sizeof(double), s));
    return CCQP_OK;
  }
  CCQP_REQUIRE(x && p, "NULL vector");
  if (nnorm > 0)
    hipLaunchKernelGGL(segs_feasible_norm_kernel, dim3((unsigned)nnorm, K), dim3(CCQP_BLOCK), 0, s, table, normwork, ws,
                       npartials, x, p, alpha0);
  hipLaunchKernelGGL(segs_feasible_count_kernel, dim3((unsigned)nwork, K), dim3(CCQP_BLOCK), 0, s, table, work, ws,
                     npartials, (int)nwork, K, x, p, alpha0, segs_counter(ws, npartials, nwork), ctx->slots + slot0);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}
