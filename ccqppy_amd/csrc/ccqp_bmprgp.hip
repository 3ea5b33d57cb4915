// Lock-step MPRGP (solvers.py:1026-1200; README's "MPRGP-BB", BASELINE config 5): many equal-size dense problems,
// every scalar and every branch decision on the device.  MPRGP's iteration is a state machine with 2-4 matvecs --
//     g = A x_k + b;  proportioning test  ->  CG step: A p; feasible?  full step + A dx  |  expansion half-step,
//     A x_{k+1} + b, A dx
// -- so, as for APGD (ccqp_bpipe.hip), a ROUND advances every live problem by ONE matvec (none while a problem
// repeats its feasibility search) and is the same five launches whatever the states of the problems:
//     matvec | stage 1 | stage 2 | stage 3 | apply
// where a stage is a reduction over the problem's vectors whose LAST block takes the decisions of the host loop with
// the host loop's arithmetic, and `apply` is the elementwise update the decisions ask for.  Which stage does what:
//     state                 matvec            stage 1                    stage 2           stage 3
//     M1                    A x_k + b -> g_k  split: P(x), psi, 1-delta  normal, n.g       beta.beta, psi.psi -> branch
//     AP                    A p               psi.p, p.Ap -> alpha_cg    feasible steps    [half step: BB dots]
//     FEAS (search again)   --                --                         feasible steps    [half step: BB dots]
//     G1   (half step)      A x_{k+1} + b     --
//     ADXC / ADXH           A dx              dx.dx, dx.Adx, residual -> BB step, exit test, swap
// (The proportioning step needs psi.psi <= 0 or NaN -- the iterates are always feasible, so beta.beta = 0 -- and does
// not occur on finite data: a problem that asks for it leaves the lock-step run for the host-driven loop.)
// x_k / x_{k+1} and g_k / g_{k+1} live in two buffers each, addressed by the parity of the outer-iteration count.
// The bodies are those of the single-problem kernels (ccqp_bodies.h), so every problem gets bit for bit the result
// of its own CCQPSolverMPRGP.solve() (tests/test_gpu_bmprgp.py).
#include <stdlib.h>

#include "ccqp_lockstep.h"
#include "ccqp_segs.h"

// slab slots of one problem
#define MP_S0 0          // dx.dx | beta.beta
#define MP_S1 1          // dx.Adx | psi.psi
#define MP_S2 2          // residual^2 of the tail
#define MP_NG 8          // n(x).g
#define MP_DIST2 10      // |x - P(x)|^2
#define MP_COUNTS 16     // 8 feasibility counts
#define MP_ABB 32        // alpha_bb
#define MP_ACG 33        // alpha_cg
#define MP_PAP 34        // p.Ap
#define MP_AF 35         // feasibility search: first candidate of the next pass / the accepted alpha_f
#define MP_AHALF 36      // BB step of the expansion half-step
#define MP_TRIED 37
#define MP_STATE 38
#define MP_ACT 39
#define MP_IT 40         // outer iterations finished (parity addresses x_k / g_k)
#define MP_EXEC 52       // matvecs executed (the reference's count skips A dx)
// Look-ahead (round 3, the persistent LDS-DMA matvec only): at the A.dx launch of an iteration x_{k+1} and p_{k+1} are
// final, so that launch also computes A.x_{k+1} + b and A.p_{k+1} -- ONE pass over A with three right-hand sides
// (bp_gemv3_pers_kernel) instead of three passes per CG step.  MP_LOOK: this run looks ahead; MP_AHEAD: what the next
// rounds find computed -- bit 0: A.x_k + b, parked in the OTHER gradient buffer (the recurrence gradient, still needed
// by the residual of the A.dx round, sits where it belongs: the M1 round copies instead of multiplying), bit 1: A.p in
// Ap, bit 2: A.x_k + b already in place (after an expansion step, whose :1152 matvec is the loop top's :1086 again;
// likewise the start's :1065).  Every product has the bits of its own pass; the counters advance where the reference
// computes the product (the stage of the M1 / AP round).
#define MP_AHEAD 53
#define MP_LOOK 54
// fused rounds (round 4, flags bit 18: small batches of one-chunk problems): the matvec budget the DEVICE enforces,
// max_mv - 1 (0: none) -- a launch may pass through several rounds, so the host's launch count no longer bounds the count
#define MP_LIMIT 55
#define MP_DONE BP_DONE  // 49: 1 converged, 2 the feasibility search cannot terminate (:1113-1118), 3 proportioning step
#define MP_MV BP_ITERS   // 50: the reference's mv_count
#define MP_RES2 BP_RES2  // 51

enum { ST_M1 = 0, ST_M1_N, ST_M1_P, ST_AP, ST_FEAS_NOW, ST_FEAS_WAIT, ST_HALF3, ST_G1, ST_ADXC, ST_ADXH,
       ST_TAIL /* segmented form only: A dx done, dx.dx / dx.Adx known, the residual on its way */ };
enum { ACT_NONE = 0, ACT_CGFULL, ACT_HALF, ACT_AFTER_G1 };

// Candidate steps alpha_f * 0.5^k per feasibility pass.  The host-driven loop tries 8 per launch; the values visited
// and the first feasible one are the same whatever the pass size, and a larger pass saves whole rounds (the search
// typically needs ~20 halvings).  The host loop gives up after 151 passes of 8 = candidates 0..1207.
#define MP_STEPS 24     // a multiple of 8
#define MP_MAX_CANDIDATES 1208

// entry snapshot of a problem's steering words (ccqp_lockstep.h: the meeting-point invariant)
struct mp_words {
  double done, state, it, mv, exec, dist2, ng, af, tried, acg, s0, s1, ahead, look, limit;
};
__device__ __forceinline__ mp_words mp_enter(const double* __restrict__ slab) {
  mp_words z;
  z.done = slab[MP_DONE];
  z.state = slab[MP_STATE];
  z.it = slab[MP_IT];
  z.mv = slab[MP_MV];
  z.exec = slab[MP_EXEC];
  z.dist2 = slab[MP_DIST2];
  z.ng = slab[MP_NG];
  z.af = slab[MP_AF];
  z.tried = slab[MP_TRIED];
  z.acg = slab[MP_ACG];
  z.s0 = slab[MP_S0];
  z.s1 = slab[MP_S1];
  z.ahead = slab[MP_AHEAD];
  z.look = slab[MP_LOOK];
  z.limit = slab[MP_LIMIT];
  __syncthreads();
  return z;
}

__device__ __forceinline__ double mp_lin(double a, double x, double b, double y) {   // axpby_body, one element
  const double t1 = a * x;
  const double t2 = b * y;
  return t1 + t2;
}

__device__ __forceinline__ double mp_free(const ccqp_segk& sg, double xi, double gi, int64_t i) {  // free_gradient_body
  const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
  const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
  const double p = ccqp_proj_elem(sg.kind, sg.flags, xi, lb, ub);
  return (ccqp_isclose(xi, p) ? 1.0 : 0.0) * gi;
}

// ---------------------------------------------------------------------------------------------------- start-up
// x_k = P(x0), x_{k+1} = copy (solvers.py:1063-1064)
__device__ __forceinline__ void bmprgp_init_x_body(const ccqp_pmprgp& q, int bx, int nbx) {
  const ccqp_segk sg = bp_segk_of(q);
  const double* __restrict__ x0 = q.X[1];     // staged by the caller
  BODY_LOOP(i, sg.dim) {
    const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
    const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
    const double v = ccqp_proj_elem(sg.kind, sg.flags, x0[i], lb, ub);
    q.X[0][i] = v;
    q.X[1][i] = v;
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void bmprgp_init_x_kernel(const ccqp_pmprgp* __restrict__ probs) {
  bmprgp_init_x_body(probs[blockIdx.y], blockIdx.x, gridDim.x);
}

// which matvec a problem runs in this launch, by `init` / its state: false = none
struct bmprgp_mv {
  const double* x;
  const double* b;
  double* y;
  double* g;
};
__device__ __forceinline__ bool bmprgp_mv_operands(const ccqp_pmprgp& q, int init, bmprgp_mv* m) {
  m->b = nullptr;
  m->g = nullptr;
  if (init == 1) {                            // g_k = A x_k + b
    m->x = q.X[0];
    m->b = q.b;
    m->y = q.ytmp;
    m->g = q.G[0];
    return true;
  }
  if (q.slab[MP_DONE] != 0.0) return false;
  if (init == 2) {                            // A g_k (solvers.py:1077)
    m->x = q.G[0];
    m->y = q.Adx;
    return true;
  }
  const int it = (int)q.slab[MP_IT];
  const int ahead = (int)q.slab[MP_AHEAD];
  switch ((int)q.slab[MP_STATE]) {
    case ST_M1:
      if (ahead & 5) return false;            // computed ahead (bmprgp_resolve3, bmprgp_take_ahead_kernel)
      m->x = q.X[it & 1];
      m->b = q.b;
      m->y = q.ytmp;
      m->g = q.G[it & 1];
      return true;
    case ST_AP:
      if (ahead & 2) return false;
      m->x = q.p;
      m->y = q.Ap;
      return true;
    case ST_G1:
      m->x = q.X[(it + 1) & 1];
      m->b = q.b;
      m->y = q.ytmp;
      m->g = q.G[(it + 1) & 1];
      return true;
    case ST_ADXC:
    case ST_ADXH:
      if (q.slab[MP_LOOK] != 0.0) return false;   // the three-vector launch of this round (bmprgp_resolve3)
      m->x = q.dx;
      m->y = q.Adx;
      return true;
    default:
      return false;                           // ST_FEAS_WAIT: no matvec this round
  }
}

template <int U, bool ALIGNED, bool NT>
__device__ __forceinline__ void bmprgp_gemv_body(const ccqp_pmprgp& q, int init, int64_t n, int64_t wave,
                                                 int64_t nwaves) {
  bmprgp_mv m;
  if (!bmprgp_mv_operands(q, init, &m)) return;
  gemv_body<BP_R(U), BP_U(U), ALIGNED, NT>(q.A, q.lda, m.x, m.b, m.y, m.g, n, n, BP_TASKS(U, n), wave, nwaves);
}

template <int U, bool ALIGNED, bool NT>
__global__ __launch_bounds__(CCQP_BLOCK) void bmprgp_gemv_kernel(const ccqp_pmprgp* __restrict__ probs, int init,
                                                                 int64_t n) {
  bmprgp_gemv_body<U, ALIGNED, NT>(probs[blockIdx.y], init, n, (int64_t)blockIdx.x * (CCQP_BLOCK / 64) + (threadIdx.x >> 6),
                                   (int64_t)gridDim.x * (CCQP_BLOCK / 64));
}

// The same matvec with the rows staged through per-wave LDS rings by LDS-DMA loads (gemv_lds_body, ccqp_bodies.h: same
// bits): few long-lived workgroups, 15-21 KiB in flight per wave.  Tuning word: loads = 3; bits 4-6 pick the ring
// (0: R = 2, S = 8, 4 waves per workgroup; 1: R = 4, S = 4, 2 waves; 2: R = 4, S = 4, 4 waves; 3: R = 2, S = 4, 4 waves).
// Needs an even n >= 128 and aligned operands.  An A/B form: it ties the VALU form at config 5's shape
// (profiles/r03_bmv_sweep_lds.jsonl) -- 64 KiB rows leave the ring little to stream between two row ends.
template <int R, int S>
__global__ __launch_bounds__(CCQP_BLOCK) void bmprgp_gemv_lds_kernel(const ccqp_pmprgp* __restrict__ probs, int init,
                                                                     int64_t n, int64_t ntasks, int64_t rs) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char ring[];
  const ccqp_pmprgp& q = probs[blockIdx.y];
  bmprgp_mv m;
  if (!bmprgp_mv_operands(q, init, &m)) return;
  constexpr int RING = S * (R + 1) * 1024;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;
  double* res = reinterpret_cast<double*>(ring + wpb * RING) + w * GEMV_LDS_RESCAP;
  gemv_lds_body<R, S, true>(q.A, q.lda, m.x, m.b, m.y, m.g, n, n, ntasks, (int64_t)blockIdx.x * wpb + w,
                            (int64_t)gridDim.x * wpb, ring + w * RING, res, rs);
#endif
}

template <int R, int S>
static int bmprgp_launch_gemv_lds(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, int count, int64_t n, int init, int flags,
                                  int wpb, hipStream_t s) {
  const size_t lds = (size_t)wpb * ((size_t)S * (R + 1) * 1024 + GEMV_LDS_RESCAP * sizeof(double));
  const int rcl = ccqp_lds_raise(ctx, reinterpret_cast<const void*>(bmprgp_gemv_lds_kernel<R, S>), lds);
  if (rcl) return rcl;
  // workgroups per problem: the tuning word's, else one workgroup per CU over the whole batch (at least 1 per problem)
  int64_t per = flags >> 8;
  if (per <= 0) per = ctx->num_cu / count > 0 ? ctx->num_cu / count : 1;
  const int64_t rs = gemv_lds_row_stride(ccqp_opt(ctx, CCQP_OPT_GEMV_ROW_STRIDE), n, n, R);   // (the wave's R rows about 1 MiB apart, ccqp_bodies.h)
  const int64_t tasks = gemv_lds_tasks(n, R, rs);
  if (per > (tasks + wpb - 1) / wpb) per = (tasks + wpb - 1) / wpb;
  if (ccqp_mvt_on) ccqp_mvt_mark(s, 0);
  hipLaunchKernelGGL((bmprgp_gemv_lds_kernel<R, S>), dim3((unsigned)per, count), dim3(64 * wpb), lds, s, dev_probs, init,
                     n, tasks, rs);
  if (ccqp_mvt_on) ccqp_mvt_mark(s, 1);
  return CCQP_OK;
}

// The PERSISTENT form (bp_gemv_pers_kernel, ccqp_lockstep.h; tuning word: loads = 3, ring codes 4-7; code 5 is the
// default from n = 4096 on): which operands problem p multiplies in this launch
struct bmprgp_resolve {
  const ccqp_pmprgp* probs;
  int init;
  __device__ bool operator()(int p, gemv_lds_desc* d) const {
    bmprgp_mv m;
    if (!bmprgp_mv_operands(probs[p], init, &m)) return false;
    d->A = probs[p].A;
    d->lda = probs[p].lda;
    d->x = m.x;
    d->b = m.b;
    d->y = m.y;
    d->g = m.g;
    return true;
  }
};

// The three-vector launch of a look-ahead run (MP_LOOK): a problem in its A.dx round multiplies dx, p_{k+1} and -- after a
// CG step -- x_{k+1} (+ b, into the gradient buffer the recurrence gradient is NOT in) in one pass over its matrix.
struct bmprgp_resolve3 {
  const ccqp_pmprgp* probs;
  __device__ bool operator()(int p, gemv_lds3_desc* d) const {
    const ccqp_pmprgp& q = probs[p];
    const double* s = q.slab;
    if (s[MP_DONE] != 0.0 || s[MP_LOOK] == 0.0) return false;
    const int st = (int)s[MP_STATE];
    if (st != ST_ADXC && st != ST_ADXH) return false;
    const int it = (int)s[MP_IT];
    d->A = q.A;
    d->lda = q.lda;
    d->b = q.b;
    d->x[0] = q.dx;                           // :1129 / :1163
    d->y[0] = q.Adx;
    d->g[0] = nullptr;
    d->x[1] = q.p;                            // :1100 of the next iteration (p_{k+1}: bmprgp_apply_body has formed it)
    d->y[1] = q.Ap;
    d->g[1] = nullptr;
    if (st == ST_ADXC) {                      // :1086 of the next iteration
      d->x[2] = q.X[(it + 1) & 1];
      d->y[2] = q.ytmp;
      d->g[2] = q.G[it & 1];
    } else {                                  // (after an expansion step A.x_{k+1} + b is in place: :1152)
      d->x[2] = q.dx;
      d->y[2] = nullptr;
      d->g[2] = nullptr;
    }
    return true;
  }
};

// the start of a look-ahead run: A.g_0 (:1077) and A.p_0 (:1100 of the first iteration; p_0 depends on x_0 and g_0 only)
struct bmprgp_resolve3_start {
  const ccqp_pmprgp* probs;
  __device__ bool operator()(int p, gemv_lds3_desc* d) const {
    const ccqp_pmprgp& q = probs[p];
    if (q.slab[MP_DONE] != 0.0) return false;
    d->A = q.A;
    d->lda = q.lda;
    d->b = q.b;
    d->x[0] = q.G[0];
    d->y[0] = q.Adx;
    d->x[1] = q.p;
    d->y[1] = q.Ap;
    d->x[2] = q.G[0];
    d->y[2] = nullptr;
    d->g[0] = d->g[1] = d->g[2] = nullptr;
    return true;
  }
};

// the M1 round of a problem whose A.x_k + b was computed ahead into the other gradient buffer: take it
__global__ __launch_bounds__(CCQP_BLOCK) void bmprgp_take_ahead_kernel(const ccqp_pmprgp* __restrict__ probs, int64_t n) {
  const ccqp_pmprgp& q = probs[blockIdx.y];
  const double* s = q.slab;
  if (s[MP_DONE] != 0.0 || (int)s[MP_STATE] != ST_M1 || !((int)s[MP_AHEAD] & 1)) return;
  const int it = (int)s[MP_IT];
  const double* __restrict__ from = q.G[(it + 1) & 1];
  double* __restrict__ to = q.G[it & 1];
  for (int64_t i = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CCQP_BLOCK) to[i] = from[i];
}

// whether a run looks ahead: the persistent matvec is its form (the context's option MPRGP_LOOKAHEAD = 1: off, the A/B
// switch of the tools)
static bool bmprgp_looks_ahead(const ccqp_ctx* ctx, int count, int64_t n, int flags) {
  return !ccqp_opt(ctx, CCQP_OPT_MPRGP_LOOKAHEAD) && bp_pers_sharing(ctx, bp_mv_tune(count, n, flags), flags, count, n, true);
}

// ---- fused rounds of small batches (round 4; flags bit 18 = CCQP_LOCKSTEP_FUSED_ROUNDS: slab slot 55 holds max_mv - 1)
// A batch of one-chunk problems below the persistent matvec's sizes is bound by the NUMBER of its launches.  The
// look-ahead alone does not help it (measured: it saves passes over the matrices, not rounds -- the M1 and A.p rounds of a
// CG step remain, without a matvec); what helps is to pass THROUGH those rounds in the launch that already holds the
// problem: the fused stage launch (one workgroup per problem, bmprgp_tail_fused_kernel) goes on to the next round as
// long as that round's product is there -- A.x_k + b and A.p computed ahead by the A.dx launch, or a feasibility search
// that repeats -- taking the product over itself (what bmprgp_take_ahead_kernel does between launches).  A CG step is
// then TWO launches -- one batched VALU launch with the three products of the A.dx round as grid slices
// (bp_gemv3_valu_kernel; a problem in any other state multiplies its one product in slice 0), one stage launch -- instead
// of six (matvec + stages, three times).  The device enforces the budget (MP_LIMIT): a problem whose next round would
// pass max_mv - 1 stops with MP_DONE = 4 and is left to the host-driven loop, where the unfused run's launch cap leaves
// it.  Same bodies, same order per problem: the bits and counts of the unfused rounds.
struct bmprgp_resolve3_any {
  const ccqp_pmprgp* probs;
  __device__ bool operator()(int p, gemv_lds3_desc* d) const {
    const bmprgp_resolve3 r3 = {probs};
    if (r3(p, d)) return true;
    bmprgp_mv m;
    if (!bmprgp_mv_operands(probs[p], 0, &m)) return false;
    d->A = probs[p].A;
    d->lda = probs[p].lda;
    d->b = m.b;
    d->x[0] = d->x[1] = d->x[2] = m.x;
    d->y[0] = m.y;
    d->g[0] = m.g;
    d->y[1] = d->y[2] = nullptr;
    d->g[1] = d->g[2] = nullptr;
    return true;
  }
};

#define MP_FUSED_PASSES 4                     // rounds one fused launch passes through at most
static bool bmprgp_fused_rounds(const ccqp_ctx* ctx, int count, int64_t n, int flags, bool fuse_asked) {
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  // (no byte limit, unlike APGD's fused rounds: nothing is read twice here that the unfused rounds read once -- the three
  // products of an A.dx round are three passes either way; forced beyond 256 MiB: 1024 / 4096 x 256 +6 / +11 %, 100 x 1024
  // +6 %, 16 / 24 x 2048 +10 %, 60 x 2048 +-0 -- up to where the persistent look-ahead takes the run, 1 GiB from n = 512)
  return fuse_asked && !ccqp_opt(ctx, CCQP_OPT_MPRGP_LOOKAHEAD) && ccqp_opt(ctx, CCQP_OPT_MPRGP_FUSED_TAIL) != 1 &&
         ccqp_opt(ctx, CCQP_OPT_SMALL_SHARING) != 1 && (flags & ~(1 | 8 | 128)) == 0 && nblk <= 1 &&
         !bmprgp_looks_ahead(ctx, count, n, flags);
}

template <int R, int S>
static int bmprgp_launch_gemv_pers(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, int count, int64_t n, int init, int wpb,
                                   int bpc, hipStream_t s) {
  const bmprgp_resolve rsv = {dev_probs, init};
  return bp_launch_gemv_pers<R, S>(ctx, rsv, count, n, wpb, bpc, s);
}

static int bmprgp_launch_gemv_on(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, int count, int64_t n, int init, int flags,
                                 hipStream_t s, bool elementwise) {
  const bp_mv_tuning t = bp_mv_tune(count, n, flags);
  const bool ahead = elementwise && bmprgp_looks_ahead(ctx, count, n, flags);
  if (ahead || bp_pers_default(ctx, t, flags, count, n)) {
    if (init == 0 && ahead) {
      hipLaunchKernelGGL(bmprgp_take_ahead_kernel, dim3(bp_vec_grid(n), count), dim3(CCQP_BLOCK), 0, s, dev_probs, n);
      const bmprgp_resolve3 rsv3 = {dev_probs};
      const int rc3 = bp_launch_gemv3(ctx, rsv3, count, n, s);
      if (rc3) return rc3;
    }
    return bmprgp_launch_gemv_pers<4, 4>(ctx, dev_probs, count, n, init, 4, 1, s);
  }
  if (((flags >> 1) & 3) == 3 && t.aligned && n % 2 == 0 && n >= 128) {
    const int code = (flags >> 4) & 7;
    if (code >= 4) {   // persistent: 4 = R 4, S 4, 2 waves; 5 = 4 waves; 6 = R 2, S 8, 4 waves; 7 = R 4, S 4, 2 waves, 2 wg/CU
      if (code == 5) return bmprgp_launch_gemv_pers<4, 4>(ctx, dev_probs, count, n, init, 4, 1, s);
      if (code == 6) return bmprgp_launch_gemv_pers<2, 8>(ctx, dev_probs, count, n, init, 4, 1, s);
      if (code == 7) return bmprgp_launch_gemv_pers<4, 4>(ctx, dev_probs, count, n, init, 2, 2, s);
      return bmprgp_launch_gemv_pers<4, 4>(ctx, dev_probs, count, n, init, 2, 1, s);
    }
    switch (code) {
      case 1: return bmprgp_launch_gemv_lds<4, 4>(ctx, dev_probs, count, n, init, flags, 2, s);
      case 2: return bmprgp_launch_gemv_lds<4, 4>(ctx, dev_probs, count, n, init, flags, 4, s);
      case 3: return bmprgp_launch_gemv_lds<2, 4>(ctx, dev_probs, count, n, init, flags, 4, s);
      default: return bmprgp_launch_gemv_lds<2, 8>(ctx, dev_probs, count, n, init, flags, 4, s);
    }
  }
  BP_MV_LAUNCH(bmprgp_gemv_kernel, dev_probs, init, n);
  return CCQP_OK;
}

// (ccqp_batch_serialize_matvecs on: the launch goes to the device's matvec stream, chained behind `s`, and `s` waits for it
// -- two half-batches on two streams then multiply one after the other while each one's decision stages and vector
// update go by under the other's matvec)
static int bmprgp_launch_gemv(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, int count, int64_t n, int init, int flags,
                              hipStream_t s, bool elementwise = true) {
  hipStream_t ms = s;
  int rc = ccqp_mv_stream_enter(ctx, s, &ms);
  if (rc) return rc;
  rc = bmprgp_launch_gemv_on(ctx, dev_probs, count, n, init, flags, ms, elementwise);
  if (rc) return rc;
  return ccqp_mv_stream_leave(ctx, s, ms);
}

// residual of the start (exit test :1070-1073)
template <bool LOCKSTEP>
__device__ __forceinline__ void bmprgp_init_res_body(const ccqp_pmprgp& q, double gd, int64_t chunk, int nblk, int bx,
                                                     unsigned* batch, double* lds4, int* is_last_p, double look = 0.0) {
  double v[1];
  v[0] = residual_body(bp_segk_of(q), q.X[0], q.G[0], q.c, gd, chunk, bx, lds4);
  if (!bp_meet(q.partials, q.counter, nblk, 1, v, lds4, is_last_p)) return;
  if (threadIdx.x == 0) {
    double* s = q.slab;
    s[MP_S2] = v[0];
    s[MP_RES2] = v[0];
    s[MP_MV] = 1.0;
    s[MP_EXEC] = 1.0;
    s[MP_IT] = 0.0;
    s[MP_STATE] = (double)ST_M1;
    s[MP_ACT] = (double)ACT_NONE;
    s[MP_LOOK] = look;
    // (the loop top's A.x_0 + b, :1086, is the start's :1065 again: in place; A.p_0 comes with the start's A.g_0)
    s[MP_AHEAD] = look != 0.0 ? 6.0 : 0.0;
    if (sqrt(v[0]) >= q.tol) {               // `if res >= tol:` -- a NaN residual skips the loop, as in the reference
      s[MP_DONE] = 0.0;
    } else {
      s[MP_DONE] = 1.0;
      if (LOCKSTEP) __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void bmprgp_init_res_kernel(const ccqp_pmprgp* __restrict__ probs, double gd,
                                                                     int64_t chunk, int nblk,
                                                                     unsigned* __restrict__ batch, double look) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  bmprgp_init_res_body<true>(probs[blockIdx.y], gd, chunk, nblk, blockIdx.x, batch, lds4, &is_last, look);
}

// alpha_bb = g.g / g.Ag (:1077), then g_{k+1} = copy(g_k) and p = phi(x_k, g_k) (:1066, :1081)
__device__ __forceinline__ void bmprgp_init_alpha_body(const ccqp_pmprgp& q, int64_t n, int64_t chunk, int nblk, int bx,
                                                       double* lds4, int* is_last_p) {
  const mp_words z = mp_enter(q.slab);
  if (z.done != 0.0) return;
  double v[2];
  v[0] = dot_body(q.G[0], q.G[0], n, chunk, bx, lds4);
  v[1] = dot_body(q.G[0], q.Adx, n, chunk, bx, lds4);
  if (!bp_meet(q.partials, q.counter, nblk, 2, v, lds4, is_last_p)) return;
  if (threadIdx.x == 0) {
    double* s = q.slab;
    s[MP_S0] = v[0];
    s[MP_S1] = v[1];
    s[MP_ABB] = v[0] / v[1];
    s[MP_MV] = 2.0;
    s[MP_EXEC] = 2.0;
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void bmprgp_init_alpha_kernel(const ccqp_pmprgp* __restrict__ probs, int64_t n,
                                                                       int64_t chunk, int nblk) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  bmprgp_init_alpha_body(probs[blockIdx.y], n, chunk, nblk, blockIdx.x, lds4, &is_last);
}

__device__ __forceinline__ void bmprgp_init_p_body(const ccqp_pmprgp& q, int bx, int nbx) {
  if (q.slab[MP_DONE] != 0.0) return;
  const ccqp_segk sg = bp_segk_of(q);
  BODY_LOOP(i, sg.dim) {
    const double gi = q.G[0][i];
    q.G[1][i] = gi;
    q.p[i] = mp_free(sg, q.X[0][i], gi, i);
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void bmprgp_init_p_kernel(const ccqp_pmprgp* __restrict__ probs) {
  bmprgp_init_p_body(probs[blockIdx.y], blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------------- the stages
// tail of an outer iteration (solvers.py:1184-1192): BB step length, exit test, swap of the iterates
template <bool LOCKSTEP>
// (`slab_copy`: the fused-rounds launch keeps the problem's slab in LDS -- see bmprgp_tail_fused_kernel)
__device__ __forceinline__ void mp_tail(const ccqp_pmprgp& q, const mp_words& z, double s0, double s1, double r2,
                                        double eps10, unsigned* batch, double* slab_copy = nullptr) {
  double* s = q.slab;
  if (slab_copy) s = slab_copy;
  const double den = s1 + eps10;
  s[MP_ABB] = s0 / den;
  s[MP_S2] = r2;
  s[MP_RES2] = r2;
  if (sqrt(r2) < q.tol) {
    s[MP_DONE] = 1.0;
    if (LOCKSTEP) __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    s[MP_IT] = z.it + 1.0;
    s[MP_STATE] = (double)ST_M1;
    // (what this round's three-vector launch computed ahead, bmprgp_resolve3)
    // (the segmented form passes through ST_TAIL and has set the word in its A.dx round, mps_reduce_a_kernel)
    if (z.look != 0.0 && ((int)z.state == ST_ADXC || (int)z.state == ST_ADXH)) s[MP_AHEAD] = (int)z.state == ST_ADXC ? 3.0 : 6.0;
  }
}

// stage = 1, 2, 3; grid (nblk, problems)
// LOCKSTEP: one of the batch's problems in a lock-step launch (roll-call in stage 3, converged counter); otherwise
// the problem of a single-launch solve (bx = 0, nblk = 1)
// SEQ: one workgroup walks the problem's chunks itself, one call per chunk (bmprgp_tail_fused_kernel): the meeting point
// is bp_meet_seq on `seqpart` (LDS, MP_STEPS x nblk doubles) -- same partial sums, same order, same bits
#define MP_MEET(NRED) \
  (SEQ ? bp_meet_seq(seqpart, bx, nblk, NRED, v, lds4) : bp_meet(q.partials, q.counter, nblk, NRED, v, lds4, is_last_p))
template <bool LOCKSTEP, bool SEQ = false>
__device__ __forceinline__ void bmprgp_stage_body(const ccqp_pmprgp& q, int stage, int bx, double gd, double eps10,
                                                  int64_t chunk, int nblk, unsigned* batch, double* host,
                                                  unsigned count, double seq, double* lds4, int* is_last_p,
                                                  double* seqpart = nullptr, double* slab_copy = nullptr) {
  double* s = q.slab;                         // written only (by the deciding thread); read through `z` alone
  if (slab_copy) s = slab_copy;               // (the fused-rounds launch: the workgroup's LDS copy of the slab)
  const mp_words z = mp_enter(s);
  if (z.done != 0.0) {
    if (LOCKSTEP && host && stage == 3 && bx == 0 && threadIdx.x == 0) bp_report(batch, host, count, seq);
    return;
  }
  const ccqp_segk sg = bp_segk_of(q);
  const int64_t n = sg.dim;
  const int st = (int)z.state;
  const int it = (int)z.it;
  double* xk = q.X[it & 1];
  double* xk1 = q.X[(it + 1) & 1];
  double* gk = q.G[it & 1];
  double* gk1 = q.G[(it + 1) & 1];
  double v[MP_STEPS];
  if (stage == 1) {
    switch (st) {
      case ST_M1: {                           // :1090-1094
        v[0] = split_body(sg, xk, gk, q.txk, q.psi, q.w, chunk, bx, lds4);
        if (!MP_MEET(1)) return;
        if (threadIdx.x == 0) {
          s[MP_DIST2] = v[0];
          s[MP_ACT] = (double)ACT_NONE;
          s[MP_MV] = z.mv + 1.0;
          s[MP_EXEC] = z.exec + 1.0;
          s[MP_STATE] = (double)ST_M1_N;
          s[MP_AHEAD] = (double)((int)z.ahead & 2);       // (A.x_k + b taken, whichever way it came)
        }
        return;
      }
      case ST_AP: {                           // :1100-1106
        v[0] = dot_body(q.psi, q.p, n, chunk, bx, lds4);
        v[1] = dot_body(q.p, q.Ap, n, chunk, bx, lds4);
        if (!MP_MEET(2)) return;
        if (threadIdx.x == 0) {
          const double acg = v[0] / v[1];
          s[MP_S0] = v[0];
          s[MP_S1] = v[1];
          s[MP_ACG] = acg;
          s[MP_PAP] = v[1];
          s[MP_AF] = acg + eps10;             // :1111
          s[MP_TRIED] = 0.0;
          s[MP_ACT] = (double)ACT_NONE;
          s[MP_MV] = z.mv + 1.0;
          s[MP_EXEC] = z.exec + 1.0;
          s[MP_STATE] = (double)ST_FEAS_NOW;
          s[MP_AHEAD] = (double)((int)z.ahead & ~2);
        }
        return;
      }
      case ST_FEAS_WAIT:
      case ST_G1: {
        if (!MP_MEET(0)) return;
        if (threadIdx.x == 0) {
          if (st == ST_G1) {
            s[MP_ACT] = (double)ACT_AFTER_G1;
            s[MP_MV] = z.mv + 1.0;
            s[MP_EXEC] = z.exec + 1.0;
            s[MP_STATE] = (double)ST_ADXH;
          } else {
            s[MP_ACT] = (double)ACT_NONE;
            s[MP_STATE] = (double)ST_FEAS_NOW;
          }
        }
        return;
      }
      case ST_ADXC:
      case ST_ADXH: {                         // :1129 / :1163, and the tail's residual
        {
          // dot_body twice and residual_body in one pass, one set of barriers
          const double* __restrict__ dxv = q.dx;
          const double* __restrict__ Adxv = q.Adx;
          const double c = q.c;
          CHUNK_RANGE(n)
          double a0 = 0.0, a1 = 0.0, a2 = 0.0;
          for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
            const double di = dxv[i];
            a0 = fma(di, di, a0);
            a1 = fma(di, Adxv[i], a1);
            const double xi = xk1[i];
            const double t = gd * gk1[i];
            const double u = xi - t;
            const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
            const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
            const double pr = ccqp_proj_elem(sg.kind, sg.flags, u, lb, ub);
            const double dd = xi - pr;
            const double w = c * dd;
            a2 = fma(w, w, a2);
          }
          v[0] = a0;
          v[1] = a1;
          v[2] = a2;
          ccqp_block_sum_multi<3>(v);
        }
        if (!MP_MEET(3)) return;
        if (threadIdx.x == 0) {
          s[MP_S0] = v[0];
          s[MP_S1] = v[1];
          s[MP_ACT] = (double)ACT_NONE;
          s[MP_EXEC] = z.exec + 1.0;
          mp_tail<LOCKSTEP>(q, z, v[0], v[1], v[2], eps10, batch, slab_copy);
        }
        return;
      }
      default:
        return;
    }
  }
  if (stage == 2) {
    if (st == ST_M1_N) {                      // :1095
      v[0] = normal_dot_body(sg, q.txk, gk, q.nv, z.dist2, chunk, bx, lds4);
      if (!MP_MEET(1)) return;
      if (threadIdx.x == 0) {
        s[MP_NG] = v[0];
        s[MP_STATE] = (double)ST_M1_P;
      }
    } else if (st == ST_FEAS_NOW) {           // :1112-1118, MP_STEPS candidates per pass
      const double a0 = z.af;
      {
        // feasible_body for the MP_STEPS candidates with ONE pair of barriers instead of one per candidate: the wave
        // sums go to LDS as they come, the four of each candidate are added (wave 0..3, as ccqp_block_sum) at the end
        __shared__ double wsum[4 * MP_STEPS];
        const int wv = threadIdx.x >> 6;
        for (int k = 0; k < MP_STEPS; ++k) {
          const double acc = ccqp_wave_sum(feasible_thread_count(sg, xk, q.p, a0, k, chunk, bx));
          if ((threadIdx.x & 63) == 0) wsum[4 * k + wv] = acc;
        }
        __syncthreads();
        if (threadIdx.x == 0)
          for (int k = 0; k < MP_STEPS; ++k)
            v[k] = ((wsum[4 * k] + wsum[4 * k + 1]) + wsum[4 * k + 2]) + wsum[4 * k + 3];
        __syncthreads();
      }
      if (!MP_MEET(MP_STEPS)) return;
      if (threadIdx.x == 0) {
        double af = a0;
        int found = 0, at = 0;
        for (int k = 0; k < MP_STEPS; ++k) {
          if (k < 8) s[MP_COUNTS + k] = v[k];
          if (!found) {
            if (v[k] == 0.0) {
              found = 1;
              at = k;
            } else {
              af *= 0.5;
            }
          }
        }
        s[MP_AF] = af;
        const double tried = z.tried;
        if (found && tried + (double)at < (double)MP_MAX_CANDIDATES) {
          if (z.acg <= af) {                  // :1120
            s[MP_ACT] = (double)ACT_CGFULL;
            s[MP_STATE] = (double)ST_ADXC;
          } else {
            s[MP_STATE] = (double)ST_HALF3;
          }
        } else if (found || tried + (double)MP_STEPS >= (double)MP_MAX_CANDIDATES) {
          // alpha_f has underflowed without a feasible candidate: the reference never exits, the host loop raises
          s[MP_DONE] = 2.0;
          if (LOCKSTEP) __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          s[MP_TRIED] = tried + (double)MP_STEPS;
          s[MP_STATE] = (double)ST_FEAS_WAIT;
        }
      }
    }
    return;
  }
  // stage 3
  if (st == ST_M1_P) {                        // :1096-1099
    proportion_body(q.w, gk, q.nv, q.psi, z.ng, n, chunk, bx, lds4, &v[0], &v[1]);
    if (!MP_MEET(2)) return;
    if (threadIdx.x == 0) {
      s[MP_S0] = v[0];
      s[MP_S1] = v[1];
      if (v[0] < v[1]) {
        s[MP_STATE] = (double)ST_AP;
      } else {
        // The proportioning step (:1165-1182).  beta is masked by 1 - isclose(x, P(x)) and the iterates are always
        // feasible, so beta.beta = 0 and this needs psi.psi <= 0 or NaN: it does not occur on finite data.  Such a
        // problem leaves the lock-step run and is solved by the host-driven loop.
        s[MP_DONE] = 3.0;
        if (LOCKSTEP) __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (LOCKSTEP && host) bp_report(batch, host, count, seq);
    }
    return;
  }
  if (st == ST_HALF3) {                       // :1140-1147: xh, gh, dx, dg formed on the fly, rounded as the axpbys
    const double maf = -z.af;
    CHUNK_RANGE(n)
    double a0 = 0.0, a1 = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
      const double xi = xk[i], gi = gk[i];
      const double xh = mp_lin(1.0, xi, maf, q.p[i]);
      const double gh = mp_lin(1.0, gi, maf, q.Ap[i]);
      const double dx = mp_lin(1.0, xh, -1.0, xi);
      const double dg = mp_lin(1.0, gh, -1.0, gi);
      a0 = fma(dx, dx, a0);
      a1 = fma(dx, dg, a1);
    }
    v[0] = ccqp_block_sum(a0, lds4);
    v[1] = ccqp_block_sum(a1, lds4);
    if (!MP_MEET(2)) return;
    if (threadIdx.x == 0) {
      const double den = v[1] + eps10;
      s[MP_S0] = v[0];
      s[MP_S1] = v[1];
      s[MP_AHALF] = v[0] / den;
      s[MP_ACT] = (double)ACT_HALF;
      s[MP_STATE] = (double)ST_G1;
      if (LOCKSTEP && host) bp_report(batch, host, count, seq);
    }
    return;
  }
  // nothing to reduce in this stage: the roll-call only.  (One designated thread: no state is written here.)
  // (host == NULL: the fused-rounds launch reports once, after its last pass)
  if (LOCKSTEP && host && bx == 0 && threadIdx.x == 0) bp_report(batch, host, count, seq);
}

__global__ __launch_bounds__(CCQP_BLOCK) void bmprgp_stage_kernel(const ccqp_pmprgp* __restrict__ probs, int stage,
                                                                  double gd, double eps10, int64_t chunk, int nblk,
                                                                  unsigned* __restrict__ batch,
                                                                  double* __restrict__ host, unsigned count,
                                                                  double seq) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  bmprgp_stage_body<true>(probs[blockIdx.y], stage, blockIdx.x, gd, eps10, chunk, nblk, batch, host, count, seq, lds4,
                          &is_last);
}

#undef MP_MEET

__device__ __forceinline__ void bmprgp_apply_body(const ccqp_pmprgp& q, int bx, int nbx, const double* slab_copy = nullptr) {
  const double* s = q.slab;
  if (slab_copy) s = slab_copy;
  if (s[MP_DONE] != 0.0) return;
  const int act = (int)s[MP_ACT];
  if (act == ACT_NONE) return;
  const ccqp_segk sg = bp_segk_of(q);
  const int it = (int)s[MP_IT];
  const double* __restrict__ xk = q.X[it & 1];
  double* __restrict__ xk1 = q.X[(it + 1) & 1];
  const double* __restrict__ gk = q.G[it & 1];
  double* __restrict__ gk1 = q.G[(it + 1) & 1];
  if (act == ACT_CGFULL) {             // :1125-1135
    const double macg = -s[MP_ACG];
    const double pAp = s[MP_PAP];
    BODY_LOOP(i, sg.dim) {
      const double xi = xk[i], pi = q.p[i], Api = q.Ap[i];
      const double y = mp_lin(1.0, xi, macg, pi);
      const double g1 = mp_lin(1.0, gk[i], macg, Api);
      xk1[i] = y;
      gk1[i] = g1;
      q.dx[i] = mp_lin(1.0, y, -1.0, xi);
      const double ps = mp_free(sg, y, g1, i);
      const double t1 = ps * Api;             // cg_direction_body
      const double beta = t1 / pAp;
      const double t2 = beta * pi;
      q.p[i] = ps - t2;
    }
  } else if (act == ACT_HALF) {               // :1140-1141, :1148
    const double maf = -s[MP_AF];
    const double mal = -s[MP_AHALF];
    BODY_LOOP(i, sg.dim) {
      const double xh = mp_lin(1.0, xk[i], maf, q.p[i]);
      const double gh = mp_lin(1.0, gk[i], maf, q.Ap[i]);
      const double u = mp_lin(1.0, xh, mal, gh);
      const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
      const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
      xk1[i] = ccqp_proj_elem(sg.kind, sg.flags, u, lb, ub);
    }
  } else if (act == ACT_AFTER_G1) {           // :1157-1162
    BODY_LOOP(i, sg.dim) {
      const double x1 = xk1[i];
      q.p[i] = mp_free(sg, x1, gk1[i], i);
      q.dx[i] = mp_lin(1.0, x1, -1.0, xk[i]);
    }
  }
}

__global__ __launch_bounds__(CCQP_BLOCK) void bmprgp_apply_kernel(const ccqp_pmprgp* __restrict__ probs) {
  bmprgp_apply_body(probs[blockIdx.y], blockIdx.x, gridDim.x);
}

// The three stages and the update of a round in ONE launch, one workgroup per problem: the workgroup walks the problem's
// chunks itself (bp_meet_seq), so the stages need no inter-workgroup meeting and follow each other behind workgroup
// barriers: four launches, with their gaps, become one.  Same bodies, same partial sums in the same order: the bits of
// the four launches.  Used where a round is bound by its launches -- problems of ONE chunk (n <= 2048), however many
// (round 4: one problem to 48 problems, fused against four launches, whole solves: n = 128 / 256 +16 ... +22 %, 1024
// +6 ... +11 %, 2048 +3 ... +4 %, profiles/r04_mprgp_fused_tail_small.jsonl; until then only from 64 problems on); at
// config 5's shape (4 chunks, 128 problems) one workgroup walking 12 chunk passes measured SLOWER than the four
// launches over 512 workgroups (not-matvec time of a solve 6.7 against 4.7 ms, profiles/r03_config5_fused_tail_ab.json),
// so longer problems keep the four launches (MP_FUSED_MAX_BLK is the kernel's capacity, MP_FUSED_USE_BLK the rule).
#define MP_FUSED_MAX_BLK 8
#define MP_FUSED_USE_BLK 1
__global__ __launch_bounds__(CCQP_BLOCK) void bmprgp_tail_fused_kernel(const ccqp_pmprgp* __restrict__ probs, double gd,
                                                                       double eps10, int64_t chunk, int nblk,
                                                                       unsigned* __restrict__ batch,
                                                                       double* __restrict__ host, unsigned count,
                                                                       double seq, int passes) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  __shared__ double seqpart[MP_STEPS * MP_FUSED_MAX_BLK];
  const ccqp_pmprgp& q = probs[blockIdx.x];
  if (passes <= 0) {
    // one round (the stage launch of the ordinary lock-step rounds)
    for (int stage = 1; stage <= 3; ++stage) {
      for (int bx = 0; bx < nblk; ++bx)
        bmprgp_stage_body<true, true>(q, stage, bx, gd, eps10, chunk, nblk, batch, host, count, seq, lds4, &is_last, seqpart);
      __syncthreads();
    }
    bmprgp_apply_body(q, 0, 1);
    return;
  }
  // fused rounds (one chunk per problem): pass through the rounds whose product is already there.  The problem's slab is
  // this workgroup's alone for the launch: it works on a copy in LDS (every stage reads fifteen steering words, and a
  // launch runs up to twelve stages -- from global memory that was a round trip each time) and writes it back at the end.
  __shared__ double slab_s[CCQP_NUM_SLOTS];
  if (threadIdx.x < CCQP_NUM_SLOTS) slab_s[threadIdx.x] = q.slab[threadIdx.x];
  __syncthreads();
  const int64_t n = q.seg.dim;
  for (int pass = 0; pass < passes; ++pass) {
    // (the words below were written by this workgroup's own thread 0 before the last barrier, or by earlier launches)
    double* s = slab_s;
    const double done = s[MP_DONE], mv = s[MP_MV], limit = s[MP_LIMIT];
    const int st = (int)s[MP_STATE], ahead = (int)s[MP_AHEAD], it = (int)s[MP_IT];
    int go = 1;
    if (done != 0.0) {
      go = 0;
    } else if (pass > 0) {
      // the round that would follow: only if it needs no matvec launch
      go = (st == ST_M1 && (ahead & 5)) || (st == ST_AP && (ahead & 2)) || st == ST_FEAS_WAIT;
    }
    const bool counts = st == ST_M1 || st == ST_AP || st == ST_G1;        // (the rounds whose stage 1 counts a matvec)
    if (go && counts && limit > 0.0 && mv + 1.0 > limit) {
      // the budget: this problem is left to the host-driven loop (whose exits are the reference's)
      __syncthreads();                          // (every thread has read the words above)
      if (threadIdx.x == 0) {
        s[MP_DONE] = 4.0;
        __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      go = 0;
    }
    if (!go) break;                             // (uniform: every thread read the same words)
    if (st == ST_M1 && (ahead & 1)) {
      // A.x_k + b was computed ahead into the OTHER gradient buffer: take it (bmprgp_take_ahead_kernel)
      const double* __restrict__ from = q.G[(it + 1) & 1];
      double* __restrict__ to = q.G[it & 1];
      for (int64_t i = threadIdx.x; i < n; i += CCQP_BLOCK) to[i] = from[i];
    }
    __syncthreads();
    // (an A.dx or half-step round is over after its first stage: the other two would only read the state again)
    const int stages = (st == ST_ADXC || st == ST_ADXH || st == ST_G1) ? 1 : 3;
    for (int stage = 1; stage <= stages; ++stage) {
      bmprgp_stage_body<true, true>(q, stage, 0, gd, eps10, chunk, 1, batch, nullptr, count, seq, lds4, &is_last, seqpart,
                                    slab_s);
      __syncthreads();
    }
    bmprgp_apply_body(q, 0, 1, slab_s);
    __syncthreads();
  }
  __syncthreads();
  if (threadIdx.x < CCQP_NUM_SLOTS) q.slab[threadIdx.x] = slab_s[threadIdx.x];
  if (threadIdx.x == 0) bp_report(batch, host, count, seq);
}

// ---------------------------------------------------------------------------------------------------- entry points
// Staging of ONE lock-step MPRGP problem in ONE launch (the device-resident form of a single solve, as
// ccqp_bapgd_stage_one_f64): the descriptor by value to `dev_desc`, the slab zeroed, x0 (NULL: zeros) into X[1], the
// problem's arrival counter and the two batch words zeroed; slab slot 55 = `limit` (fused rounds: max_mv - 1, 0: none).
__global__ __launch_bounds__(CCQP_BLOCK) void bmprgp_stage_one_kernel(const ccqp_pmprgp desc, ccqp_pmprgp* __restrict__ dev_desc,
                                                                      const double* __restrict__ x0, int64_t n, double limit,
                                                                      uint32_t* __restrict__ words) {
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) *dev_desc = desc;
    if (threadIdx.x < 64) desc.slab[threadIdx.x] = threadIdx.x == MP_LIMIT ? limit : 0.0;
    if (threadIdx.x < 3) words[threadIdx.x] = 0u;
  }
  double* __restrict__ X1 = desc.X[1];
  for (int64_t i = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CCQP_BLOCK)
    X1[i] = x0 ? x0[i] : 0.0;
}

extern "C" int ccqp_bmprgp_stage_one_f64(ccqp_ctx* ctx, const ccqp_pmprgp* host_desc, ccqp_pmprgp* dev_desc, const double* x0,
                                         int64_t n, double limit, uint32_t* words, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && host_desc != nullptr && dev_desc != nullptr && words != nullptr, "NULL argument");
  CCQP_REQUIRE(n >= 1 && host_desc->X[1] != nullptr && host_desc->slab != nullptr, "incomplete descriptor");
  hipLaunchKernelGGL(bmprgp_stage_one_kernel, dim3(bp_vec_grid(n)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, *host_desc,
                     dev_desc, x0, n, limit, words);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// whether a run's rounds are fused (see the header)
extern "C" int ccqp_bmprgp_fused_form(ccqp_ctx* ctx, int count, int64_t n, int flags) {
  if (count < 1 || n < 1) return 0;
  return bmprgp_fused_rounds(ctx, count, n, flags & 0xFFFF, (flags & CCQP_LOCKSTEP_FUSED_ROUNDS) != 0) ? 2 : 0;
}

// Start-up of all solves (solvers.py:1063-1081).  Expects x0 staged in X[1], slabs and counters zero.
extern "C" int ccqp_bmprgp_start_f64(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, int count, int64_t n, double gd,
                                     int flags, uint32_t* dev_batch, void* stream) {
  BP_CHECK_BATCH();
  const bool fuse_asked = (flags & CCQP_LOCKSTEP_FUSED_ROUNDS) != 0;
  flags &= 0xFFFF;
  hipStream_t s = (hipStream_t)stream;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  const dim3 vgrid(bp_vec_grid(n), count), rgrid(nblk, count), blk(CCQP_BLOCK);
  ccqp_lockstep_begin(ctx, dev_batch, s);
  hipLaunchKernelGGL(bmprgp_init_x_kernel, vgrid, blk, 0, s, dev_probs);
  int rc = bmprgp_launch_gemv(ctx, dev_probs, count, n, 1, flags, s);
  if (rc) return rc;
  const bool fused = bmprgp_fused_rounds(ctx, count, n, flags, fuse_asked);
  const bool look = fused || bmprgp_looks_ahead(ctx, count, n, flags);
  hipLaunchKernelGGL(bmprgp_init_res_kernel, rgrid, blk, 0, s, dev_probs, gd, chunk, nblk, dev_batch, look ? 1.0 : 0.0);
  if (look) {
    // p_0 first (it needs x_0 and g_0 only), then A.g_0 and A.p_0 in one pass over every matrix
    hipLaunchKernelGGL(bmprgp_init_p_kernel, vgrid, blk, 0, s, dev_probs);
    hipStream_t ms = s;
    rc = ccqp_mv_stream_enter(ctx, s, &ms);
    if (rc) return rc;
    const bmprgp_resolve3_start rsv3 = {dev_probs};
    if (fused)
      bp_launch_gemv3_valu(rsv3, count, n, flags, 2, ms);      // (fused rounds: the products as slices of a VALU launch)
    else
      rc = bp_launch_gemv3(ctx, rsv3, count, n, ms);
    if (rc) return rc;
    rc = ccqp_mv_stream_leave(ctx, s, ms);
    if (rc) return rc;
    hipLaunchKernelGGL(bmprgp_init_alpha_kernel, rgrid, blk, 0, s, dev_probs, n, chunk, nblk);
  } else {
    rc = bmprgp_launch_gemv(ctx, dev_probs, count, n, 2, flags, s);
    if (rc) return rc;
    hipLaunchKernelGGL(bmprgp_init_alpha_kernel, rgrid, blk, 0, s, dev_probs, n, chunk, nblk);
    hipLaunchKernelGGL(bmprgp_init_p_kernel, vgrid, blk, 0, s, dev_probs);
  }
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// Round `round` >= 1: at most one matvec for every live problem, the three decision stages, the vector update.
extern "C" int ccqp_bmprgp_round_f64(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, int count, int64_t n, int64_t round,
                                     double gd, double eps10, int flags, uint32_t* dev_batch, void* stream) {
  BP_CHECK_BATCH();
  CCQP_REQUIRE(round >= 1, "rounds count from 1");
  const bool fuse_asked = (flags & CCQP_LOCKSTEP_FUSED_ROUNDS) != 0;
  flags &= 0xFFFF;
  hipStream_t s = (hipStream_t)stream;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  const dim3 rgrid(nblk, count), blk(CCQP_BLOCK);
  if (bmprgp_fused_rounds(ctx, count, n, flags, fuse_asked)) {
    // TWO launches: every live problem's product(s) -- the three of an A.dx round as grid slices --, then the stage
    // launch that passes through the rounds whose products are there
    hipStream_t ms = s;
    int rcf = ccqp_mv_stream_enter(ctx, s, &ms);
    if (rcf) return rcf;
    const bmprgp_resolve3_any rsv = {dev_probs};
    bp_launch_gemv3_valu(rsv, count, n, flags, 3, ms);
    rcf = ccqp_mv_stream_leave(ctx, s, ms);
    if (rcf) return rcf;
    hipLaunchKernelGGL(bmprgp_tail_fused_kernel, dim3(count), blk, 0, s, dev_probs, gd, eps10, chunk, nblk, dev_batch,
                       ctx->host_slots_dev, (unsigned)count, (double)round, (int)MP_FUSED_PASSES);
    CCQP_LAUNCH_CHECK();
    return CCQP_OK;
  }
  const int rc = bmprgp_launch_gemv(ctx, dev_probs, count, n, 0, flags, s);
  if (rc) return rc;
  // (by shape; the context's option MPRGP_FUSED_TAIL: 1 keeps the four launches, 2 fuses wherever the kernel can run --
  // the A/B switches of the tools and of tests/test_gpu_bmprgp.py)
  const bool allow_fused = ccqp_opt(ctx, CCQP_OPT_MPRGP_FUSED_TAIL) != 1;
  const bool force_fused = ccqp_opt(ctx, CCQP_OPT_MPRGP_FUSED_TAIL) == 2;
  const bool fuse = allow_fused && nblk <= (force_fused ? MP_FUSED_MAX_BLK : MP_FUSED_USE_BLK);
  if (fuse) {
    hipLaunchKernelGGL(bmprgp_tail_fused_kernel, dim3(count), blk, 0, s, dev_probs, gd, eps10, chunk, nblk, dev_batch,
                       ctx->host_slots_dev, (unsigned)count, (double)round, 0);
  } else {
    for (int stage = 1; stage <= 3; ++stage)
      hipLaunchKernelGGL(bmprgp_stage_kernel, rgrid, blk, 0, s, dev_probs, stage, gd, eps10, chunk, nblk, dev_batch,
                         ctx->host_slots_dev, (unsigned)count, (double)round);
    hipLaunchKernelGGL(bmprgp_apply_kernel, dim3(bp_vec_grid(n), count), blk, 0, s, dev_probs);
  }
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// n <= CCQP_RED_CHUNK: the whole solve of every problem in ONE launch, one workgroup per problem -- the bodies of a
// lock-step round run back to back inside the workgroup, separated by barriers (ccqp_bpipe.hip explains the form).
// No round starts once the matvec count has reached `mv_limit` (max_mv - 1).
template <bool ALIGNED>
__global__ __launch_bounds__(CCQP_BLOCK) void bsolve_mprgp_kernel(const ccqp_pmprgp* __restrict__ probs, double gd,
                                                                  double eps10, double mv_limit, int64_t n) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  const ccqp_pmprgp& q = probs[blockIdx.x];
  const int64_t wave = threadIdx.x >> 6;
  const int64_t nwaves = CCQP_BLOCK / 64;
  const int64_t chunk = CCQP_RED_CHUNK;       // ccqp_red_geometry(n) for n <= CCQP_RED_CHUNK: one block
  double* s = q.slab;
  bmprgp_init_x_body(q, 0, 1);
  __syncthreads();
  bmprgp_gemv_body<4, ALIGNED, false>(q, 1, n, wave, nwaves);
  __syncthreads();
  bmprgp_init_res_body<false>(q, gd, chunk, 1, 0, nullptr, lds4, &is_last);
  __syncthreads();
  if (s[MP_DONE] != 0.0) return;
  bmprgp_gemv_body<4, ALIGNED, false>(q, 2, n, wave, nwaves);
  __syncthreads();
  bmprgp_init_alpha_body(q, n, chunk, 1, 0, lds4, &is_last);
  __syncthreads();
  bmprgp_init_p_body(q, 0, 1);
  __syncthreads();
  for (;;) {
    if (s[MP_MV] >= mv_limit) break;
    bmprgp_gemv_body<4, ALIGNED, false>(q, 0, n, wave, nwaves);
    __syncthreads();
    for (int stage = 1; stage <= 3; ++stage) {
      bmprgp_stage_body<false>(q, stage, 0, gd, eps10, chunk, 1, nullptr, nullptr, 0u, 0.0, lds4, &is_last);
      __syncthreads();
    }
    if (s[MP_DONE] != 0.0) break;
    bmprgp_apply_body(q, 0, 1);
    __syncthreads();
  }
}

extern "C" int ccqp_bsolve_mprgp_f64(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, int count, int64_t n, double gd,
                                     double eps10, double mv_limit, int flags, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && dev_probs != nullptr, "NULL argument");
  CCQP_REQUIRE(count >= 1 && n >= 1 && n <= CCQP_RED_CHUNK, "one workgroup per problem needs n <= 2048");
  hipStream_t s = (hipStream_t)stream;
  if (flags & 1)
    hipLaunchKernelGGL((bsolve_mprgp_kernel<true>), dim3(count), dim3(CCQP_BLOCK), 0, s, dev_probs, gd, eps10, mv_limit,
                       n);
  else
    hipLaunchKernelGGL((bsolve_mprgp_kernel<false>), dim3(count), dim3(CCQP_BLOCK), 0, s, dev_probs, gd, eps10,
                       mv_limit, n);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// =====================================================================================================================
// Lock-step MPRGP on feasible sets that are NOT elementwise: Sphere, DisjointUnion with blocks of any size (Cone has no
// normal vector: solution_spaces.py:465 raises, and so does the host-driven loop such a problem is left to).
// Reference: CCQPSolverMPRGP.solve (solvers.py:1026-1200) around SphereProjOp / DisjointProjOp.__call__ and
// .normal_vector (solution_spaces.py:389-403, 417-435, 512-525, 540-560) -- per iteration four projections, one normal
// vector and up to dozens of feasibility probes, each a Python loop over the blocks.
// The state machine, slab and matvec kernel are those of the elementwise form above; every place the set enters goes
// through the segmented engine (ccqp_segs.h) with per-problem block tables (struct ccqp_psegs, as ccqp_bsegs.hip).  A
// ROUND is ten launches for the whole batch, each acting on a problem according to its state:
//   1 matvec        A x_k + b | A p | A x_{k+1} + b | A dx                                      (bmprgp_gemv_kernel)
//   2 reduce A      M1: counters -> M1_N;  AP: psi.p, p.Ap -> alpha_cg, FEAS_NOW;  G1: -> "after G1";  FEAS_WAIT ->
//                   FEAS_NOW;  ADXC / ADXH: dx.dx, dx.Adx -> TAIL
//   3 seg norm 1    chunk sums of |u|^2 for the Sphere blocks, u = x_k | x_k - alpha_f 0.5^k p (K candidates) |
//                   x_{k+1} - gd g_{k+1} | x_{k+1}
//   4 seg apply 1   M1_N: P(x_k) -> psi, 1 - delta, normal vector (small blocks finished, large ones: sums);
//                   FEAS_NOW: counts of entries with not isclose(y_f, P(y_f)) per work item and candidate;
//                   TAIL: per work item sum (c (x - P(x - gd g)))^2;  after G1: p = isclose(x, P(x)) g, dx = x_{k+1} - x_k
//   5 seg finish    M1_N: normal vector of the large blocks
//   6 reduce B      M1_N: n.g -> M1_P;  FEAS_NOW: the decision (full CG step | half step | another pass | give up);
//                   TAIL: residual = work-item sums in index order, BB step, exit test, swap -> M1
//   7 reduce C      M1_P: beta.beta, psi.psi -> AP (| proportioning: left to the host loop);  HALF3: BB dots of the
//                   half step;  and the batch's roll-call
//   8 apply         full CG step: x_{k+1}, g_{k+1}, dx;  half step: x_h, g_h into scratch vectors
//   9 seg norm 2    |y|^2 (full step)  |  |x_h - alpha g_h|^2 (half step)
//  10 seg apply 2   full step: p = psi_y - (psi_y Ap / p.Ap) p with psi_y = isclose(y, P(y)) g_{k+1};  half step:
//                   x_{k+1} = P(x_h - alpha g_h)
// Bits: per element and per sum the arithmetic of the host-driven loop's launches (ccqp_segs_project_f64,
// ccqp_isclose_mask_mul_f64, ccqp_segs_normal_f64, ccqp_dots_f64, ccqp_chopped_gradient_f64,
// ccqp_segs_feasible_steps_f64, ccqp_cg_direction_f64, ccqp_segs_residual_sq_f64) and the host's scalar steps, so
// every problem gets exactly the result of its own CCQPSolverMPRGP.solve() (tests/test_gpu_bsegs.py).
#define MPS_STEPS CCQP_MAX_STEPS          // feasibility candidates per pass (the workspace holds this many norm rows)

// what a segmented launch of the FIRST group (3-5) has to do for this problem
enum { SEG1_NONE = 0, SEG1_NORMAL, SEG1_FEAS, SEG1_TAIL, SEG1_AFTER_G1 };
__device__ __forceinline__ int mps_what1(const double* s) {
  if (s[MP_DONE] != 0.0) return SEG1_NONE;
  switch ((int)s[MP_STATE]) {
    case ST_M1_N: return SEG1_NORMAL;
    case ST_FEAS_NOW: return SEG1_FEAS;
    case ST_TAIL: return SEG1_TAIL;
    case ST_ADXH: return (int)s[MP_ACT] == ACT_AFTER_G1 ? SEG1_AFTER_G1 : SEG1_NONE;
    default: return SEG1_NONE;
  }
}

// start-up modes of the segmented launches (mode > 0); mode 0 = by state (round, first group), -1 = by action (second)
enum { MPS_BY_STATE = 0, MPS_START_X = 1, MPS_START_RES = 2, MPS_START_P = 3, MPS_BY_ACTION = -1 };

// chunk sums of |u|^2 for the Sphere / Cone blocks.  grid (max nnorm, problems, K)
__global__ __launch_bounds__(CCQP_BLOCK) void mps_norm_kernel(const ccqp_pmprgp* __restrict__ probs,
                                                              const ccqp_psegs* __restrict__ segs, int mode,
                                                              double gd) {
  __shared__ double lds4[4];
  const ccqp_psegs& ps = segs[blockIdx.y];
  if ((int)blockIdx.x >= ps.nnorm) return;
  const ccqp_pmprgp& q = probs[blockIdx.y];
  const double* s = q.slab;
  const int k = blockIdx.z;
  const double* x = nullptr;
  const double* y = nullptr;
  double b = 0.0;
  int64_t row = 0;                             // which row of npartials chunk sums of the workspace
  if (mode == MPS_START_X) {
    x = q.dx;                                  // x0, staged there by the caller
  } else if (mode == MPS_START_RES) {
    if (s[MP_DONE] != 0.0) return;
    x = q.X[0];
    y = q.G[0];
    b = -gd;
  } else if (mode == MPS_START_P) {
    if (s[MP_DONE] != 0.0) return;
    x = q.X[0];
  } else if (mode == MPS_BY_STATE) {
    const int it = (int)s[MP_IT];
    switch (mps_what1(s)) {
      case SEG1_NORMAL: x = q.X[it & 1]; break;
      case SEG1_FEAS: {
        double alpha = s[MP_AF];
        for (int j = 0; j < k; ++j) alpha *= 0.5;   // the reference's `alpha_f *= 0.5`, k times
        x = q.X[it & 1];
        y = q.p;
        b = -alpha;
        row = k;
        break;
      }
      case SEG1_TAIL: x = q.X[(it + 1) & 1]; y = q.G[(it + 1) & 1]; b = -gd; break;
      case SEG1_AFTER_G1: x = q.X[(it + 1) & 1]; break;
      default: return;
    }
    if (k > 0 && row == 0) return;             // only the feasibility search has more than one candidate
  } else {
    if (s[MP_DONE] != 0.0) return;
    const int act = (int)s[MP_ACT];
    const int it = (int)s[MP_IT];
    if (act == ACT_CGFULL) {
      x = q.X[(it + 1) & 1];                   // y = x_k - alpha_cg p, written by the apply launch
    } else if (act == ACT_HALF) {
      x = q.txk;                               // x_h
      y = q.w;                                 // g_h
      b = -s[MP_AHALF];
    } else {
      return;
    }
  }
  const ccqp_segwork w = ps.normwork[blockIdx.x];
  const double sum = segs_norm_item(ps.table, w, 1.0, x, b, y, lds4);
  if (threadIdx.x == 0) ps.workspace[row * ps.npartials + w.pbase + w.chunk] = sum;
}

// ---- per-element epilogues of the apply launches (segs_apply_emit: emit(gi, u, P(u)))
struct mps_emit_start_x {      // x_k = P(x0), x_{k+1} = copy (:1063-1064)
  double* x0;
  double* x1;
  __device__ __forceinline__ void operator()(int64_t i, double, double p) {
    x0[i] = p;
    x1[i] = p;
  }
};
struct mps_emit_residual {     // sum (c (x - P(x - gd g)))^2, this thread's share
  const double* x;
  double c, acc;
  __device__ __forceinline__ void operator()(int64_t i, double, double p) {
    const double d = x[i] - p;
    const double v = c * d;
    acc = fma(v, v, acc);
  }
};
struct mps_emit_free {         // out = isclose(x, P(x)) * g   [+ g1 = copy(g)]  [+ dx = x - xk]
  const double* g;
  double* out;
  double* gcopy;
  const double* xk;
  double* dx;
  __device__ __forceinline__ void operator()(int64_t i, double u, double p) {
    const double gi = g[i];
    out[i] = (ccqp_isclose(u, p) ? 1.0 : 0.0) * gi;
    if (gcopy) gcopy[i] = gi;
    if (dx) dx[i] = mp_lin(1.0, u, -1.0, xk[i]);
  }
};
struct mps_emit_count {        // entries with not isclose(y_f, P(y_f))
  double acc;
  __device__ __forceinline__ void operator()(int64_t, double u, double p) { acc += ccqp_isclose(u, p) ? 0.0 : 1.0; }
};
struct mps_emit_split {        // psi = isclose(x, P(x)) g, w = 1 - isclose  (:1090-1094)
  const double* g;
  double* psi;
  double* w;
  __device__ __forceinline__ void operator()(int64_t i, double x, double xp) const {
    const bool close = ccqp_isclose(x, xp);
    psi[i] = (close ? 1.0 : 0.0) * g[i];
    w[i] = close ? 0.0 : 1.0;
  }
};
struct mps_emit_cg {           // psi_y = isclose(y, P(y)) g1;  p = psi_y - (psi_y Ap / pAp) p   (:1132-1135)
  const double* g1;
  const double* Ap;
  double* p;
  double pAp;
  __device__ __forceinline__ void operator()(int64_t i, double u, double pr) {
    const double ps = (ccqp_isclose(u, pr) ? 1.0 : 0.0) * g1[i];
    const double t1 = ps * Ap[i];
    const double beta = t1 / pAp;
    const double t2 = beta * p[i];
    p[i] = ps - t2;
  }
};
struct mps_emit_store {
  double* out;
  __device__ __forceinline__ void operator()(int64_t i, double, double p) { out[i] = p; }
};

// grid (max nwork, problems, K)
__global__ __launch_bounds__(CCQP_BLOCK) void mps_apply_kernel(const ccqp_pmprgp* __restrict__ probs,
                                                               const ccqp_psegs* __restrict__ segs, int mode,
                                                               double gd) {
  __shared__ double lds4[4];
  __shared__ double bcast;
  const ccqp_psegs& ps = segs[blockIdx.y];
  if ((int)blockIdx.x >= ps.nwork) return;
  const ccqp_pmprgp& q = probs[blockIdx.y];
  const double* s = q.slab;
  const int k = blockIdx.z;
  const ccqp_segwork w = ps.work[blockIdx.x];
  double* ws = ps.workspace;
  if (mode == MPS_START_X) {
    mps_emit_start_x e = {q.X[0], q.X[1]};
    segs_apply_emit(ps.table, w, ws, 1.0, q.dx, 0.0, nullptr, e, lds4, &bcast);
    return;
  }
  if (s[MP_DONE] != 0.0) return;
  const int it = (int)s[MP_IT];
  if (mode == MPS_START_RES) {
    mps_emit_residual e = {q.X[0], q.c, 0.0};
    segs_apply_emit(ps.table, w, ws, 1.0, q.X[0], -gd, q.G[0], e, lds4, &bcast);
    const double sum = ccqp_block_sum(e.acc, lds4);
    if (threadIdx.x == 0) ws[3 * ps.npartials + blockIdx.x] = sum;
    return;
  }
  if (mode == MPS_START_P) {                   // g_{k+1} = copy(g_k), p = phi(x_k, g_k)  (:1066, :1081)
    mps_emit_free e = {q.G[0], q.p, q.G[1], nullptr, nullptr};
    segs_apply_emit(ps.table, w, ws, 1.0, q.X[0], 0.0, nullptr, e, lds4, &bcast);
    return;
  }
  if (mode == MPS_BY_ACTION) {
    const int act = (int)s[MP_ACT];
    if (act == ACT_CGFULL) {
      mps_emit_cg e = {q.G[(it + 1) & 1], q.Ap, q.p, s[MP_PAP]};
      segs_apply_emit(ps.table, w, ws, 1.0, q.X[(it + 1) & 1], 0.0, nullptr, e, lds4, &bcast);
    } else if (act == ACT_HALF) {
      mps_emit_store e = {q.X[(it + 1) & 1]};
      segs_apply_emit(ps.table, w, ws, 1.0, q.txk, -s[MP_AHALF], q.w, e, lds4, &bcast);
    }
    return;
  }
  switch (mps_what1(s)) {
    case SEG1_NORMAL: {
      if (k > 0) return;
      mps_emit_split e = {q.G[it & 1], q.psi, q.w};
      segs_normal_project_item(ps.table, w, ws, ws + ps.npartials, ws + 2 * ps.npartials, q.X[it & 1], q.nv, e, lds4,
                               &bcast);
      return;
    }
    case SEG1_FEAS: {
      double alpha = s[MP_AF];
      for (int j = 0; j < k; ++j) alpha *= 0.5;
      mps_emit_count e = {0.0};
      segs_apply_emit(ps.table, w, ws + (int64_t)k * ps.npartials, 1.0, q.X[it & 1], -alpha, q.p, e, lds4, &bcast);
      const double sum = ccqp_block_sum(e.acc, lds4);
      if (threadIdx.x == 0) ws[(int64_t)MPS_STEPS * ps.npartials + (int64_t)k * ps.nwork + blockIdx.x] = sum;
      return;
    }
    case SEG1_TAIL: {
      if (k > 0) return;
      mps_emit_residual e = {q.X[(it + 1) & 1], q.c, 0.0};
      segs_apply_emit(ps.table, w, ws, 1.0, q.X[(it + 1) & 1], -gd, q.G[(it + 1) & 1], e, lds4, &bcast);
      const double sum = ccqp_block_sum(e.acc, lds4);
      if (threadIdx.x == 0) ws[3 * ps.npartials + blockIdx.x] = sum;
      return;
    }
    case SEG1_AFTER_G1: {                      // :1157-1162
      if (k > 0) return;
      mps_emit_free e = {q.G[(it + 1) & 1], q.p, nullptr, q.X[it & 1], q.dx};
      segs_apply_emit(ps.table, w, ws, 1.0, q.X[(it + 1) & 1], 0.0, nullptr, e, lds4, &bcast);
      return;
    }
    default:
      return;
  }
}

// normal vector of the large blocks (launch 5).  grid (max nwork, problems)
__global__ __launch_bounds__(CCQP_BLOCK) void mps_normal_finish_kernel(const ccqp_pmprgp* __restrict__ probs,
                                                                       const ccqp_psegs* __restrict__ segs) {
  __shared__ double lds4[4];
  __shared__ double bcast;
  const ccqp_psegs& ps = segs[blockIdx.y];
  if ((int)blockIdx.x >= ps.nwork || ps.npartials == 0) return;
  const ccqp_pmprgp& q = probs[blockIdx.y];
  if (mps_what1(q.slab) != SEG1_NORMAL) return;
  const ccqp_segwork w = ps.work[blockIdx.x];
  segs_normal_finish_item(ps.table, w, ps.workspace + ps.npartials, ps.workspace + 2 * ps.npartials, q.nv, lds4, &bcast);
}

// the work items' sums added in index order (ccqp_reduce_finish's order; one item: the sum itself) -- all threads call
__device__ __forceinline__ double mps_item_sum(const double* part, int nwork, double* lds4) {
  if (nwork == 1) return part[0];
  return partial_sum_body(part, nwork, lds4);
}

// start-up decision: residual of (x_k, g_k), exit test (:1070-1073).  One block per problem.
__global__ __launch_bounds__(CCQP_BLOCK) void mps_init_res_kernel(const ccqp_pmprgp* __restrict__ probs,
                                                                  const ccqp_psegs* __restrict__ segs,
                                                                  unsigned* __restrict__ batch, double look) {
  __shared__ double lds4[4];
  const ccqp_pmprgp& q = probs[blockIdx.x];
  const ccqp_psegs& ps = segs[blockIdx.x];
  const double r = mps_item_sum(ps.workspace + 3 * ps.npartials, ps.nwork, lds4);
  if (threadIdx.x != 0) return;
  double* s = q.slab;
  s[MP_S2] = r;
  s[MP_RES2] = r;
  s[MP_MV] = 1.0;
  s[MP_EXEC] = 1.0;
  s[MP_IT] = 0.0;
  s[MP_STATE] = (double)ST_M1;
  s[MP_ACT] = (double)ACT_NONE;
  s[MP_LOOK] = look;                           // (look-ahead: as the elementwise form, bmprgp_init_res_body)
  s[MP_AHEAD] = look != 0.0 ? 6.0 : 0.0;
  if (sqrt(r) >= q.tol) {                      // `if res >= tol:` -- a NaN residual skips the loop, as in the reference
    s[MP_DONE] = 0.0;
  } else {
    s[MP_DONE] = 1.0;
    __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// launch 2.  grid (nblk, problems)
__global__ __launch_bounds__(CCQP_BLOCK) void mps_reduce_a_kernel(const ccqp_pmprgp* __restrict__ probs, int64_t n,
                                                                  double eps10, int64_t chunk, int nblk) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  const ccqp_pmprgp& q = probs[blockIdx.y];
  double* s = q.slab;                          // written only (by the deciding thread); read through `z` alone
  const mp_words z = mp_enter(s);
  if (z.done != 0.0) return;
  const int bx = blockIdx.x;
  double v[2];
  switch ((int)z.state) {
    case ST_M1:                                // g_k = A x_k + b has run (:1086) -- or was computed ahead
      if (!bp_meet(q.partials, q.counter, nblk, 0, v, lds4, &is_last)) return;
      if (threadIdx.x == 0) {
        s[MP_ACT] = (double)ACT_NONE;
        s[MP_MV] = z.mv + 1.0;
        s[MP_EXEC] = z.exec + 1.0;
        s[MP_STATE] = (double)ST_M1_N;
        s[MP_AHEAD] = (double)((int)z.ahead & 2);
      }
      return;
    case ST_AP:                                // :1100-1111
      v[0] = dot_body(q.psi, q.p, n, chunk, bx, lds4);
      v[1] = dot_body(q.p, q.Ap, n, chunk, bx, lds4);
      if (!bp_meet(q.partials, q.counter, nblk, 2, v, lds4, &is_last)) return;
      if (threadIdx.x == 0) {
        const double acg = v[0] / v[1];
        s[MP_S0] = v[0];
        s[MP_S1] = v[1];
        s[MP_ACG] = acg;
        s[MP_PAP] = v[1];
        s[MP_AF] = acg + eps10;
        s[MP_TRIED] = 0.0;
        s[MP_ACT] = (double)ACT_NONE;
        s[MP_MV] = z.mv + 1.0;
        s[MP_EXEC] = z.exec + 1.0;
        s[MP_STATE] = (double)ST_FEAS_NOW;
        s[MP_AHEAD] = (double)((int)z.ahead & ~2);
      }
      return;
    case ST_G1:                                // g_{k+1} = A x_{k+1} + b has run (:1152)
    case ST_FEAS_WAIT:
      if (!bp_meet(q.partials, q.counter, nblk, 0, v, lds4, &is_last)) return;
      if (threadIdx.x == 0) {
        if ((int)z.state == ST_G1) {
          s[MP_ACT] = (double)ACT_AFTER_G1;
          s[MP_MV] = z.mv + 1.0;
          s[MP_EXEC] = z.exec + 1.0;
          s[MP_STATE] = (double)ST_ADXH;
        } else {
          s[MP_ACT] = (double)ACT_NONE;
          s[MP_STATE] = (double)ST_FEAS_NOW;
        }
      }
      return;
    case ST_ADXC:
    case ST_ADXH:                              // A dx has run (:1129 / :1163): the BB dots; the residual follows
      v[0] = dot_body(q.dx, q.dx, n, chunk, bx, lds4);
      v[1] = dot_body(q.dx, q.Adx, n, chunk, bx, lds4);
      if (!bp_meet(q.partials, q.counter, nblk, 2, v, lds4, &is_last)) return;
      if (threadIdx.x == 0) {
        s[MP_S0] = v[0];
        s[MP_S1] = v[1];
        s[MP_ACT] = (double)ACT_NONE;
        s[MP_EXEC] = z.exec + 1.0;
        s[MP_STATE] = (double)ST_TAIL;
        // (what this round's three-vector launch computed ahead, bmprgp_resolve3; harmless if the tail finds convergence)
        if (z.look != 0.0) s[MP_AHEAD] = (int)z.state == ST_ADXC ? 3.0 : 6.0;
      }
      return;
    default:
      return;
  }
}

// launch 6.  grid (nblk, problems)
__global__ __launch_bounds__(CCQP_BLOCK) void mps_reduce_b_kernel(const ccqp_pmprgp* __restrict__ probs,
                                                                  const ccqp_psegs* __restrict__ segs, int64_t n,
                                                                  double eps10, int64_t chunk, int nblk,
                                                                  unsigned* __restrict__ batch) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  __shared__ double cnt[MPS_STEPS];
  const ccqp_pmprgp& q = probs[blockIdx.y];
  const ccqp_psegs& ps = segs[blockIdx.y];
  double* s = q.slab;                          // written only (by the deciding thread); read through `z` alone
  const mp_words z = mp_enter(s);
  if (z.done != 0.0) return;
  const int it = (int)z.it;
  double v[1];
  switch ((int)z.state) {
    case ST_M1_N:                              // n(x_k).g_k (:1095-1096)
      v[0] = dot_body(q.nv, q.G[it & 1], n, chunk, blockIdx.x, lds4);
      if (!bp_meet(q.partials, q.counter, nblk, 1, v, lds4, &is_last)) return;
      if (threadIdx.x == 0) {
        s[MP_NG] = v[0];
        s[MP_STATE] = (double)ST_M1_P;
      }
      return;
    case ST_FEAS_NOW: {                        // :1112-1120, MPS_STEPS candidates per pass
      if (!bp_meet(q.partials, q.counter, nblk, 0, v, lds4, &is_last)) return;
      const double* counts = ps.workspace + (int64_t)MPS_STEPS * ps.npartials;
      for (int k = 0; k < MPS_STEPS; ++k) {    // (small integers: their sum does not depend on the order)
        const double c = mps_item_sum(counts + (int64_t)k * ps.nwork, ps.nwork, lds4);
        if (threadIdx.x == 0) cnt[k] = c;
      }
      if (threadIdx.x != 0) return;
      double af = z.af;
      int found = 0, at = 0;
      for (int k = 0; k < MPS_STEPS; ++k) {
        if (k < 8) s[MP_COUNTS + k] = cnt[k];
        if (!found) {
          if (cnt[k] == 0.0) {
            found = 1;
            at = k;
          } else {
            af *= 0.5;
          }
        }
      }
      s[MP_AF] = af;
      const double tried = z.tried;
      if (found && tried + (double)at < (double)MP_MAX_CANDIDATES) {
        if (z.acg <= af) {                     // :1120
          s[MP_ACT] = (double)ACT_CGFULL;
          s[MP_STATE] = (double)ST_ADXC;
        } else {
          s[MP_STATE] = (double)ST_HALF3;
        }
      } else if (found || tried + (double)MPS_STEPS >= (double)MP_MAX_CANDIDATES) {
        s[MP_DONE] = 2.0;                      // alpha_f has underflowed: the reference never exits, the host loop raises
        __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        s[MP_TRIED] = tried + (double)MPS_STEPS;
        s[MP_STATE] = (double)ST_FEAS_WAIT;
      }
      return;
    }
    case ST_TAIL: {                            // :1184-1192
      if (!bp_meet(q.partials, q.counter, nblk, 0, v, lds4, &is_last)) return;
      const double r = mps_item_sum(ps.workspace + 3 * ps.npartials, ps.nwork, lds4);
      if (threadIdx.x == 0) mp_tail<true>(q, z, z.s0, z.s1, r, eps10, batch);
      return;
    }
    default:
      return;
  }
}

// launch 7.  grid (nblk, problems); the batch's roll-call
__global__ __launch_bounds__(CCQP_BLOCK) void mps_reduce_c_kernel(const ccqp_pmprgp* __restrict__ probs, int64_t n,
                                                                  double eps10, int64_t chunk, int nblk,
                                                                  unsigned* __restrict__ batch,
                                                                  double* __restrict__ host, unsigned count,
                                                                  double seq) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  const ccqp_pmprgp& q = probs[blockIdx.y];
  double* s = q.slab;                          // written only (by the deciding thread); read through `z` alone
  const mp_words z = mp_enter(s);
  const int bx = blockIdx.x;
  if (z.done != 0.0) {
    if (bx == 0 && threadIdx.x == 0) bp_report(batch, host, count, seq);
    return;
  }
  const int it = (int)z.it;
  const double* gk = q.G[it & 1];
  double v[2];
  if ((int)z.state == ST_M1_P) {               // :1096-1099
    proportion_body(q.w, gk, q.nv, q.psi, z.ng, n, chunk, bx, lds4, &v[0], &v[1]);
    if (!bp_meet(q.partials, q.counter, nblk, 2, v, lds4, &is_last)) return;
    if (threadIdx.x == 0) {
      s[MP_S0] = v[0];
      s[MP_S1] = v[1];
      if (v[0] < v[1]) {
        s[MP_STATE] = (double)ST_AP;
      } else {
        s[MP_DONE] = 3.0;                      // a proportioning step (:1165-1182): left to the host-driven loop
        __hip_atomic_fetch_add(batch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      bp_report(batch, host, count, seq);
    }
    return;
  }
  if ((int)z.state == ST_HALF3) {              // :1140-1147
    const double maf = -z.af;
    const double* xk = q.X[it & 1];
    CHUNK_RANGE(n)
    double a0 = 0.0, a1 = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
      const double xi = xk[i], gi = gk[i];
      const double xh = mp_lin(1.0, xi, maf, q.p[i]);
      const double gh = mp_lin(1.0, gi, maf, q.Ap[i]);
      const double dx = mp_lin(1.0, xh, -1.0, xi);
      const double dg = mp_lin(1.0, gh, -1.0, gi);
      a0 = fma(dx, dx, a0);
      a1 = fma(dx, dg, a1);
    }
    v[0] = ccqp_block_sum(a0, lds4);
    v[1] = ccqp_block_sum(a1, lds4);
    if (!bp_meet(q.partials, q.counter, nblk, 2, v, lds4, &is_last)) return;
    if (threadIdx.x == 0) {
      const double den = v[1] + eps10;
      s[MP_S0] = v[0];
      s[MP_S1] = v[1];
      s[MP_AHALF] = v[0] / den;
      s[MP_ACT] = (double)ACT_HALF;
      s[MP_STATE] = (double)ST_G1;
      bp_report(batch, host, count, seq);
    }
    return;
  }
  if (bx == 0 && threadIdx.x == 0) bp_report(batch, host, count, seq);
}

// launch 8: the elementwise part of the update.  grid (blocks, problems)
__global__ __launch_bounds__(CCQP_BLOCK) void mps_apply_elem_kernel(const ccqp_pmprgp* __restrict__ probs, int64_t n) {
  const ccqp_pmprgp& q = probs[blockIdx.y];
  const double* s = q.slab;
  if (s[MP_DONE] != 0.0) return;
  const int act = (int)s[MP_ACT];
  const int it = (int)s[MP_IT];
  const double* __restrict__ xk = q.X[it & 1];
  const double* __restrict__ gk = q.G[it & 1];
  const int bx = blockIdx.x, nbx = gridDim.x;
  if (act == ACT_CGFULL) {                     // :1125-1128
    double* __restrict__ xk1 = q.X[(it + 1) & 1];
    double* __restrict__ gk1 = q.G[(it + 1) & 1];
    const double macg = -s[MP_ACG];
    BODY_LOOP(i, n) {
      const double xi = xk[i];
      const double y = mp_lin(1.0, xi, macg, q.p[i]);
      xk1[i] = y;
      gk1[i] = mp_lin(1.0, gk[i], macg, q.Ap[i]);
      q.dx[i] = mp_lin(1.0, y, -1.0, xi);
    }
  } else if (act == ACT_HALF) {                // :1140-1141 (x_h, g_h into scratch: psi / w / nv / txk are dead here)
    const double maf = -s[MP_AF];
    BODY_LOOP(i, n) {
      q.txk[i] = mp_lin(1.0, xk[i], maf, q.p[i]);
      q.w[i] = mp_lin(1.0, gk[i], maf, q.Ap[i]);
    }
  }
}

#define MPS_CHECK()                                                                       \
  BP_CHECK_BATCH();                                                                       \
  CCQP_REQUIRE(dev_segs != nullptr, "NULL segment descriptors");                          \
  CCQP_REQUIRE(max_nwork >= 1 && max_nnorm >= 0 && max_nwork < 2147483647, "bad work-list bounds")

// Start-up of all solves (solvers.py:1063-1081).  Expects x0 staged in `dx`, slabs, counters and every problem's
// segment workspace zero.
extern "C" int ccqp_bsegs_mprgp_start_f64(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, const ccqp_psegs* dev_segs,
                                          int count, int64_t n, int64_t max_nwork, int64_t max_nnorm, double gd,
                                          int flags, uint32_t* dev_batch, void* stream) {
  MPS_CHECK();
  hipStream_t s = (hipStream_t)stream;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  const dim3 gw((unsigned)max_nwork, count), gn((unsigned)(max_nnorm > 0 ? max_nnorm : 1), count), blk(CCQP_BLOCK);
  ccqp_lockstep_begin(ctx, dev_batch, s);
  if (max_nnorm > 0) hipLaunchKernelGGL(mps_norm_kernel, gn, blk, 0, s, dev_probs, dev_segs, (int)MPS_START_X, gd);
  hipLaunchKernelGGL(mps_apply_kernel, gw, blk, 0, s, dev_probs, dev_segs, (int)MPS_START_X, gd);
  int rc = bmprgp_launch_gemv(ctx, dev_probs, count, n, 1, flags, s, false);
  if (rc) return rc;
  if (max_nnorm > 0) hipLaunchKernelGGL(mps_norm_kernel, gn, blk, 0, s, dev_probs, dev_segs, (int)MPS_START_RES, gd);
  hipLaunchKernelGGL(mps_apply_kernel, gw, blk, 0, s, dev_probs, dev_segs, (int)MPS_START_RES, gd);
  const bool look = bmprgp_looks_ahead(ctx, count, n, flags);
  hipLaunchKernelGGL(mps_init_res_kernel, dim3(count), blk, 0, s, dev_probs, dev_segs, dev_batch, look ? 1.0 : 0.0);
  if (look) {
    // p_0 first (it needs x_0 and g_0 only), then A.g_0 and A.p_0 in one pass over every matrix (as ccqp_bmprgp_start_f64)
    if (max_nnorm > 0) hipLaunchKernelGGL(mps_norm_kernel, gn, blk, 0, s, dev_probs, dev_segs, (int)MPS_START_P, gd);
    hipLaunchKernelGGL(mps_apply_kernel, gw, blk, 0, s, dev_probs, dev_segs, (int)MPS_START_P, gd);
    hipStream_t ms = s;
    rc = ccqp_mv_stream_enter(ctx, s, &ms);
    if (rc) return rc;
    const bmprgp_resolve3_start rsv3 = {dev_probs};
    rc = bp_launch_gemv3(ctx, rsv3, count, n, ms);
    if (rc) return rc;
    rc = ccqp_mv_stream_leave(ctx, s, ms);
    if (rc) return rc;
    hipLaunchKernelGGL(bmprgp_init_alpha_kernel, dim3(nblk, count), blk, 0, s, dev_probs, n, chunk, nblk);
  } else {
    rc = bmprgp_launch_gemv(ctx, dev_probs, count, n, 2, flags, s, false);
    if (rc) return rc;
    hipLaunchKernelGGL(bmprgp_init_alpha_kernel, dim3(nblk, count), blk, 0, s, dev_probs, n, chunk, nblk);
    if (max_nnorm > 0) hipLaunchKernelGGL(mps_norm_kernel, gn, blk, 0, s, dev_probs, dev_segs, (int)MPS_START_P, gd);
    hipLaunchKernelGGL(mps_apply_kernel, gw, blk, 0, s, dev_probs, dev_segs, (int)MPS_START_P, gd);
  }
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// Round `round` >= 1 of all live solves: the ten launches above.
extern "C" int ccqp_bsegs_mprgp_round_f64(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, const ccqp_psegs* dev_segs,
                                          int count, int64_t n, int64_t round, int64_t max_nwork, int64_t max_nnorm,
                                          double gd, double eps10, int flags, uint32_t* dev_batch, void* stream) {
  MPS_CHECK();
  CCQP_REQUIRE(round >= 1, "rounds count from 1");
  hipStream_t s = (hipStream_t)stream;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  const dim3 rgrid(nblk, count), blk(CCQP_BLOCK);
  const dim3 gwk((unsigned)max_nwork, count, MPS_STEPS), gnk((unsigned)(max_nnorm > 0 ? max_nnorm : 1), count, MPS_STEPS);
  const dim3 gw((unsigned)max_nwork, count), gn((unsigned)(max_nnorm > 0 ? max_nnorm : 1), count);
  const int rc = bmprgp_launch_gemv(ctx, dev_probs, count, n, 0, flags, s, true);    // (looks ahead like the elementwise form)
  if (rc) return rc;
  hipLaunchKernelGGL(mps_reduce_a_kernel, rgrid, blk, 0, s, dev_probs, n, eps10, chunk, nblk);
  if (max_nnorm > 0) hipLaunchKernelGGL(mps_norm_kernel, gnk, blk, 0, s, dev_probs, dev_segs, (int)MPS_BY_STATE, gd);
  hipLaunchKernelGGL(mps_apply_kernel, gwk, blk, 0, s, dev_probs, dev_segs, (int)MPS_BY_STATE, gd);
  hipLaunchKernelGGL(mps_normal_finish_kernel, gw, blk, 0, s, dev_probs, dev_segs);
  hipLaunchKernelGGL(mps_reduce_b_kernel, rgrid, blk, 0, s, dev_probs, dev_segs, n, eps10, chunk, nblk, dev_batch);
  hipLaunchKernelGGL(mps_reduce_c_kernel, rgrid, blk, 0, s, dev_probs, n, eps10, chunk, nblk, dev_batch,
                     ctx->host_slots_dev, (unsigned)count, (double)round);
  hipLaunchKernelGGL(mps_apply_elem_kernel, dim3(bp_vec_grid(n), count), blk, 0, s, dev_probs, n);
  if (max_nnorm > 0) hipLaunchKernelGGL(mps_norm_kernel, gn, blk, 0, s, dev_probs, dev_segs, (int)MPS_BY_ACTION, gd);
  hipLaunchKernelGGL(mps_apply_kernel, gw, blk, 0, s, dev_probs, dev_segs, (int)MPS_BY_ACTION, gd);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// rows of partial sums (each ccqp_bpipe_partials_doubles(n) / 3 doubles long) one MPRGP problem needs
extern "C" int ccqp_bmprgp_partial_rows(void) { return MP_STEPS; }
