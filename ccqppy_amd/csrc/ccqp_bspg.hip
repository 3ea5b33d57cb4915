// Lock-step SPG-QP (solvers.py:878-975) for many dense problems of one size, with the host in the loop ONCE per
// iteration of the whole batch (opt-in: ccqppy_amd.batch.solve_batch_pipelined picks it for CCQPSolverSPG solvers that
// carry their own np.random.RandomState).
//
// Why the host stays in the loop: the step length of SPG-QP is `np.random.uniform(sigma1, min(betahat, sigma2))` with
// betahat = tau*beta + sqrt(tau**2 * beta**2 + 2*xi) -- a draw of numpy's legacy Mersenne twister and two calls of
// libm's pow(), neither of which a kernel can reproduce bit for bit (pow(x, 2) != x*x for ~1 argument in 1200, and
// glibc's pow is a table-driven routine of that particular libm).  What the device CAN do is all the vector work of
// all problems at once.  One round =
//   ctl     the host's control words (alpha, betak, mode per problem) from pinned memory into device memory
//   step    x += betak d, g += betak Ad of the PREVIOUS iteration (:961-962), then d = P(x - alpha g) - x (:937)
//           [segmented sets: update, norm, apply -- the projection goes through ccqp_segs.h]
//   matvec  Ad = A d                                                       (the batched matvec of ccqp_bpipe.hip)
//   dots    d.d, d.Ad, d.g per problem (:944-946); the last block of the last problem copies all of them to pinned
//           host memory and raises the round's sequence number
// and then the host, in C, problem after problem: exit test, fmax, xi, beta, betahat, the draw, fk, the queue, alpha
// (:949-966) -- ~20 ns per problem instead of ~20 us of Python per problem and scalar read (the host-driven batch
// scheduler's 2.7 k problems/s).  Reference: the Python loop of CCQPSolverSPG.solve as the ensemble benchmark runs it
// (benchmarks/benchmark_random_ccqp.py:67-102).
// Bits: the bodies are those of the single-problem kernels (axpby_body's and proj_elem_sub_kernel's expressions,
// gemv_task, dot_body with the reduction geometry of ccqp_dots_f64), the host arithmetic is the Python loop's,
// operation for operation, on the same libm: every problem gets exactly its own CCQPSolverSPG.solve().
#include <math.h>

#include "ccqp_lockstep.h"
#include "ccqp_segs.h"

static_assert(sizeof(ccqp_pspg) == 128 && sizeof(ccqp_spg_host) == 120, "descriptor layout changed: update _lib.py");

#define SPG_ALPHA 0
#define SPG_BETAK 1
#define SPG_MODE 2      // 0 skip | 1 update (if due) + direction + matvec + dots | 2 update only (the budget's last round)
#define SPG_UPDATE 3

// ------------------------------------------------------------------------------------------------- host arithmetic
// libm's pow through a pointer the compiler cannot see through: clang folds pow(x, 2.0) into x*x, which is NOT what
// Python's and numpy's scalar `**` compute
static double (*volatile spg_pow)(double, double) = pow;

// Python's max() over the queue (oldest first): the first element unless a later one compares greater
static inline double spg_queue_max(const double* q, int len) {
  double m = q[0];
  for (int i = 1; i < len; ++i)
    if (q[i] > m) m = q[i];
  return m;
}

static inline void spg_queue_append(ccqp_spg_host* h, double* q, double v) {   // deque(maxlen=m).append
  if (h->qlen == h->m) {
    for (int i = 1; i < h->m; ++i) q[i - 1] = q[i];
    h->qlen -= 1;
  }
  q[h->qlen++] = v;
}

// solvers.py:949-966 for one problem (its matvec of this iteration is already counted)
// (the draw: from `src` at the moment the loop would take it, else the given u)
static void spg_host_step(ccqp_spg_host* h, double* q, double dd, double dAd, double dg, const ccqp_draw_source* src,
                          double u) {
  h->dd = dd;
  h->have_dd = 1;
  h->pending = 0;
  if (sqrt(dd) <= h->tol) {                             // :949
    h->status = 1;
    return;
  }
  const double fmax = spg_queue_max(q, h->qlen);         // :953
  const double xi = (fmax - h->fk) / dAd;                // :954
  const double beta = (-dg) / dAd;                       // :955
  const double t2 = spg_pow(h->tau, 2.0);                // :958  tau ** 2
  const double b2 = spg_pow(beta, 2.0);                  //       beta ** 2
  const double prod = t2 * b2;
  const double twoxi = 2.0 * xi;
  const double arg = prod + twoxi;
  const double tb = h->tau * beta;
  const double betahat = tb + sqrt(arg);
  const double high = h->sigma2 < betahat ? h->sigma2 : betahat;   // Python's min(betahat, sig2): NaN stays
  const double range = high - h->sigma1;                 // RandomState.uniform: fscale = high - low ...
  if (!isfinite(range)) {                                // ... OverflowError('Range exceeds valid bounds')
    h->status = 3;
    return;
  }
  if (src) u = src->next_double(src->state);
  const double ru = range * u;                           // random_uniform: low + range * next_double
  const double betak = h->sigma1 + ru;
  h->used += 1;
  const double bb = betak * betak;                       // :963 (sic)  betak * betak * dg + 0.5 * (betak ** 2) * dAd
  const double t3 = bb * dg;
  const double hp = 0.5 * spg_pow(betak, 2.0);
  const double t4 = hp * dAd;
  const double inc = t3 + t4;
  h->fk = h->fk + inc;
  spg_queue_append(h, q, h->fk);                         // :964
  h->alpha = dd / dAd;                                   // :966
  h->betak = betak;
  h->pending = 1;                                        // :961-962 are applied by the next round's step kernel
}

extern "C" int ccqp_bspg_host_step(ccqp_spg_host* host, double* queue, int64_t qstride, const double* dots,
                                   const ccqp_draw_source* const* sources, const double* u,
                                   const unsigned char* live, int count) {
  CCQP_REQUIRE(host && queue && dots && (u || sources) && count >= 0, "NULL argument");
  for (int p = 0; p < count; ++p) {
    if (live && !live[p]) continue;
    CCQP_REQUIRE(host[p].m >= 1 && host[p].m <= qstride && host[p].qlen >= 1 && host[p].qlen <= host[p].m,
                 "bad queue geometry");
    CCQP_REQUIRE(!sources || (sources[p] && sources[p]->next_double), "NULL draw source");
    spg_host_step(host + p, queue + (int64_t)p * qstride, dots[3 * p], dots[3 * p + 1], dots[3 * p + 2],
                  sources ? sources[p] : nullptr, u ? u[p] : 0.0);
  }
  return CCQP_OK;
}

// ------------------------------------------------------------------------------------------------------- kernels
__global__ __launch_bounds__(CCQP_BLOCK) void bspg_ctl_kernel(const double* __restrict__ host_ctl,
                                                              double* __restrict__ ctl, int n4) {
  const int i = blockIdx.x * CCQP_BLOCK + threadIdx.x;
  if (i < n4) ctl[i] = host_ctl[i];
}

// grid (blocks, problems).  DIRECTION: also d = P(x - alpha g) - x for an elementwise set
// host_ctl != NULL: the round's control words are still in the pinned host buffer -- every block reads its problem's four
// from there and the problem's first block leaves them in `ctl` for the launches that follow (instead of a copy launch
// of its own in front: four launches per round become three)
template <bool DIRECTION>
__global__ __launch_bounds__(CCQP_BLOCK) void bspg_step_kernel(const ccqp_pspg* __restrict__ probs, double* __restrict__ ctl,
                                                               int64_t n, const double* __restrict__ host_ctl = nullptr) {
  double cw[4];
  {
    const double* src = (host_ctl ? host_ctl : ctl) + 4 * blockIdx.y;
#pragma unroll
    for (int k = 0; k < 4; ++k) cw[k] = src[k];
    if (host_ctl && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) ctl[4 * blockIdx.y + k] = cw[k];
    }
  }
  const double* c = cw;
  const int mode = (int)c[SPG_MODE];
  if (mode == 0) return;
  const bool upd = c[SPG_UPDATE] != 0.0;
  const bool dir = DIRECTION && mode == 1;
  if (!upd && !dir) return;
  const ccqp_pspg& q = probs[blockIdx.y];
  const double betak = c[SPG_BETAK];
  const double ma = -c[SPG_ALPHA];
  const ccqp_segk sg = bp_segk_of(q);
  double* x = q.x;
  double* g = q.g;
  double* d = q.d;
  const double* Ad = q.Ad;
  const int bx = blockIdx.x, nbx = gridDim.x;
  BODY_LOOP(i, n) {                   // (n, not sg.dim: a segmented run leaves `seg` empty)
    double xi = x[i], gi = g[i];
    if (upd) {                        // ccqp_axpby2_f64(1, x, betak, d, x, 1, g, betak, Ad, g)
      const double t1 = 1.0 * xi;
      const double t2 = betak * d[i];
      xi = t1 + t2;
      const double t3 = 1.0 * gi;
      const double t4 = betak * Ad[i];
      gi = t3 + t4;
      x[i] = xi;
      g[i] = gi;
    }
    if (dir) {                        // ccqp_proj_elementwise_sub_f64(seg, 1, x, -alpha, g, x, d)
      const double t1 = 1.0 * xi;
      const double t2 = ma * gi;
      const double u = t1 + t2;
      const double lb = sg.lb ? sg.lb[i] : sg.lb_s;
      const double ub = sg.ub ? sg.ub[i] : sg.ub_s;
      const double p = ccqp_proj_elem(sg.kind, sg.flags, u, lb, ub);
      d[i] = p - xi;
    }
  }
}

// segmented sets: chunk sums of |x - alpha g|^2 for the Sphere / Cone blocks.  grid (max nnorm, problems)
__global__ __launch_bounds__(CCQP_BLOCK) void bspg_norm_kernel(const ccqp_pspg* __restrict__ probs,
                                                               const ccqp_psegs* __restrict__ segs,
                                                               const double* __restrict__ ctl) {
  __shared__ double lds4[4];
  const double* c = ctl + 4 * blockIdx.y;
  if ((int)c[SPG_MODE] != 1) return;
  const ccqp_psegs& ps = segs[blockIdx.y];
  if ((int)blockIdx.x >= ps.nnorm) return;
  const ccqp_pspg& q = probs[blockIdx.y];
  const ccqp_segwork w = ps.normwork[blockIdx.x];
  const double s = segs_norm_item(ps.table, w, 1.0, q.x, -c[SPG_ALPHA], q.g, lds4);
  if (threadIdx.x == 0) ps.workspace[w.pbase + w.chunk] = s;
}

// segmented sets: d = P(x - alpha g) - x.  grid (max nwork, problems)
__global__ __launch_bounds__(CCQP_BLOCK) void bspg_apply_kernel(const ccqp_pspg* __restrict__ probs,
                                                                const ccqp_psegs* __restrict__ segs,
                                                                const double* __restrict__ ctl) {
  __shared__ double lds4[4];
  __shared__ double bcast;
  const double* c = ctl + 4 * blockIdx.y;
  if ((int)c[SPG_MODE] != 1) return;
  const ccqp_psegs& ps = segs[blockIdx.y];
  if ((int)blockIdx.x >= ps.nwork) return;
  const ccqp_pspg& q = probs[blockIdx.y];
  const ccqp_segwork w = ps.work[blockIdx.x];
  (void)segs_apply_item<0>(ps.table, w, ps.workspace, 1.0, q.x, -c[SPG_ALPHA], q.g, q.x, q.d, 0.0, lds4, &bcast);
}

// grid (blocks, problems).  which 0: Ad = A x, g = Ad + b (start-up);  1: Ad = A g (start-up);  2: Ad = A d
template <int U, bool ALIGNED, bool NT>
__global__ __launch_bounds__(CCQP_BLOCK) void bspg_gemv_kernel(const ccqp_pspg* __restrict__ probs,
                                                               const double* __restrict__ ctl, int which, int64_t n) {
  if ((int)ctl[4 * blockIdx.y + SPG_MODE] != 1) return;
  const ccqp_pspg& q = probs[blockIdx.y];
  const int64_t wave = (int64_t)blockIdx.x * (CCQP_BLOCK / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (CCQP_BLOCK / 64);
  if (which == 0)
    gemv_body<BP_R(U), BP_U(U), ALIGNED, NT>(q.A, q.lda, q.x, q.b, q.Ad, q.g, n, n, BP_TASKS(U, n), wave, nwaves);
  else
    gemv_body<BP_R(U), BP_U(U), ALIGNED, NT>(q.A, q.lda, which == 1 ? q.g : q.d, nullptr, q.Ad, nullptr, n, n,
                                             BP_TASKS(U, n), wave, nwaves);
}

// grid (nblk, problems) -- the geometry of ccqp_dots_f64 on n-vectors.  start: {g.x, g.g, g.Ad} (:923-924), else
// {d.d, d.Ad, d.g} (:944-946).  `nlive` of the `count` problems behind the pointers take part; the last of them to
// finish publishes every problem's four words to pinned host memory and then the sequence number to `flag`.
__global__ __launch_bounds__(CCQP_BLOCK) void bspg_dots_kernel(const ccqp_pspg* __restrict__ probs,
                                                               const double* __restrict__ ctl,
                                                               double* __restrict__ out, double* __restrict__ host_out,
                                                               int start, int64_t n, int64_t chunk, int nblk,
                                                               unsigned* __restrict__ batch, double* __restrict__ flag,
                                                               unsigned nlive, int count, double seq) {
  __shared__ double lds4[4];
  __shared__ int is_last;
  __shared__ int publish;
  if ((int)ctl[4 * blockIdx.y + SPG_MODE] != 1) return;
  const ccqp_pspg& q = probs[blockIdx.y];
  // three dot_body passes in one: per thread the same elements in the same order into one fma chain per product, then
  // the block sums of ccqp_block_sum taken together (ccqp_block_sum_multi: same bits, two barriers instead of six)
  const double* __restrict__ u = start ? q.g : q.d;     // {g.x, g.g, g.Ad}  |  {d.d, d.Ad, d.g}
  const double* __restrict__ w0 = start ? q.x : q.d;
  const double* __restrict__ w1 = start ? q.g : q.Ad;
  const double* __restrict__ w2 = start ? q.Ad : q.g;
  double v[3] = {0.0, 0.0, 0.0};
  {
    const int bx = blockIdx.x;
    CHUNK_RANGE(n)
    for (int64_t i = lo + threadIdx.x; i < hi; i += CCQP_BLOCK) {
      const double ui = u[i];
      v[0] = fma(ui, w0[i], v[0]);
      v[1] = fma(ui, w1[i], v[1]);
      v[2] = fma(ui, w2[i], v[2]);
    }
  }
  ccqp_block_sum_multi<3>(v);
  if (!bp_meet(q.partials, q.counter, nblk, 3, v, lds4, &is_last)) return;
  // the problem's last block; the batch-wide roll-call decides which block copies the results out
  if (threadIdx.x == 0) {
    double* o = out + 4 * blockIdx.y;
    o[0] = v[0];
    o[1] = v[1];
    o[2] = v[2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(batch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    publish = prev + 1u == nlive;
    if (publish) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(batch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!publish) return;
  for (int i = threadIdx.x; i < 4 * count; i += CCQP_BLOCK) host_out[i] = out[i];
  // this block's stores have reached host memory before the flag does (as publish_slots_kernel, ccqp_ctx.hip)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// --------------------------------------------------------------------------------------------------------- driver
static int bspg_exchange(ccqp_ctx* ctx, size_t doubles) {
  if (ctx->xchg_doubles >= doubles) return CCQP_OK;
  if (ctx->xchg_host) {
    CCQP_HIP(hipHostFree(ctx->xchg_host));
    ctx->xchg_host = nullptr;
    ctx->xchg_doubles = 0;
  }
  CCQP_HIP(hipHostMalloc(&ctx->xchg_host, doubles * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
  CCQP_HIP(hipHostGetDevicePointer((void**)&ctx->xchg_host_dev, ctx->xchg_host, 0));
  ctx->xchg_doubles = doubles;
  return CCQP_OK;
}

static int bspg_wait(volatile double* flag, double seq, hipStream_t s) {
  for (int spin = 0; spin < 4000000; ++spin) {
    if (*flag == seq) {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      return CCQP_OK;
    }
    __builtin_ia32_pause();
  }
  CCQP_HIP(hipStreamSynchronize(s));                    // mirror silent for a long time: wait for the queue itself
  CCQP_REQUIRE(*flag == seq, "the device never reported the round");
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return CCQP_OK;
}

// (the persistent LDS-DMA form, bp_gemv_pers_kernel of ccqp_lockstep.h: the operands bspg_gemv_kernel would take)
struct bspg_resolve {
  const ccqp_pspg* probs;
  const double* ctl;
  int which;
  __device__ bool operator()(int p, gemv_lds_desc* d) const {
    if ((int)ctl[4 * p + SPG_MODE] != 1) return false;
    const ccqp_pspg& q = probs[p];
    d->A = q.A;
    d->lda = q.lda;
    d->y = q.Ad;
    d->x = which == 0 ? q.x : which == 1 ? q.g : q.d;
    d->b = which == 0 ? q.b : nullptr;
    d->g = which == 0 ? q.g : nullptr;
    return true;
  }
};

static int bspg_gemv(ccqp_ctx* ctx, const ccqp_pspg* probs, const double* ctl, int count, int64_t n, int which, int flags,
                     hipStream_t s) {
  const bp_mv_tuning t = bp_mv_tune(count, n, flags);
  if (bp_pers_default(ctx, t, flags, count, n)) {
    const bspg_resolve rsv = {probs, ctl, which};
    return bp_launch_gemv_pers<4, 4>(ctx, rsv, count, n, 4, 1, s);
  }
  BP_MV_LAUNCH(bspg_gemv_kernel, probs, ctl, which, n);
  return CCQP_OK;
}

// One group of consecutive problems of the batch: its rounds are queued and awaited on their own, so that while the
// host does the scalar work of one group the device runs the round of the other.
struct bspg_group {
  int p0, cnt;            // problems [p0, p0 + cnt)
  double seq;             // rounds with a dots launch so far (the value the group's flag shows after the last one)
  volatile double* flag;  // host_slots[BP_MIRROR + 1 + g]
  double* flag_dev;
  uint32_t* batch;        // the group's arrival counter
  bool inflight;          // a round with a dots launch is queued and not yet consumed
};

// Staging of ONE lock-step SPG problem in ONE launch (the device-resident form of a single solve, as
// ccqp_bapgd_stage_one_f64): the descriptor by value to `dev_desc`, x0 (NULL: zeros) into x, the problem's arrival counter
// and the two batch words, the eight exchange words (control | dot products) zeroed.
__global__ __launch_bounds__(CCQP_BLOCK) void bspg_stage_one_kernel(const ccqp_pspg desc, ccqp_pspg* __restrict__ dev_desc,
                                                                    const double* __restrict__ x0, int64_t n,
                                                                    uint32_t* __restrict__ words, double* __restrict__ xchg) {
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) *dev_desc = desc;
    if (threadIdx.x < 3) words[threadIdx.x] = 0u;
    if (threadIdx.x < 8) xchg[threadIdx.x] = 0.0;
  }
  double* __restrict__ x = desc.x;
  for (int64_t i = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CCQP_BLOCK)
    x[i] = x0 ? x0[i] : 0.0;
}

extern "C" int ccqp_bspg_stage_one_f64(ccqp_ctx* ctx, const ccqp_pspg* host_desc, ccqp_pspg* dev_desc, const double* x0,
                                       int64_t n, uint32_t* words, double* xchg, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && host_desc != nullptr && dev_desc != nullptr && words != nullptr && xchg != nullptr,
               "NULL argument");
  CCQP_REQUIRE(n >= 1 && host_desc->x != nullptr, "incomplete descriptor");
  hipLaunchKernelGGL(bspg_stage_one_kernel, dim3(bp_vec_grid(n)), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, *host_desc,
                     dev_desc, x0, n, words, xchg);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_bspg_run_f64(ccqp_ctx* ctx, const ccqp_pspg* dev_probs, const ccqp_psegs* dev_segs,
                                 ccqp_spg_host* host, double* queue, int64_t qstride,
                                 const ccqp_draw_source* const* sources, int count, int64_t n, int64_t max_nwork,
                                 int64_t max_nnorm, int flags, int start, int64_t max_rounds, double* dev_ctl,
                                 double* dev_out, uint32_t* dev_batch, int* more, void* stream) {
  BP_CHECK_BATCH();
  CCQP_REQUIRE(host && queue && sources && dev_ctl && dev_out && more, "NULL argument");
  for (int p = 0; p < count; ++p)
    CCQP_REQUIRE(sources[p] && sources[p]->next_double, "NULL draw source");
  CCQP_REQUIRE(dev_segs == nullptr || (max_nwork >= 1 && max_nwork <= 65535 && max_nnorm >= 0 && max_nnorm <= 65535),
               "bad work-list sizes");
  for (int p = 0; p < count; ++p)
    CCQP_REQUIRE(host[p].m >= 1 && host[p].m <= qstride, "queue capacity m outside [1, qstride]");
  hipStream_t s = (hipStream_t)stream;
  int rc = bspg_exchange(ctx, (size_t)8 * count);
  if (rc) return rc;
  double* hctl = ctx->xchg_host;                         // pinned: control words out, dot products in
  double* hout = ctx->xchg_host + (size_t)4 * count;
  double* hctl_dev = ctx->xchg_host_dev;
  double* hout_dev = ctx->xchg_host_dev + (size_t)4 * count;
  int64_t chunk;
  int nblk;
  ccqp_red_geometry(n, &chunk, &nblk);
  const dim3 blk(CCQP_BLOCK);
  ccqp_lockstep_begin(ctx, dev_batch, s);
  ctx->host_slots[BP_MIRROR + 2] = -1.0;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  *more = 0;
  // two groups from 32 problems on (flags bit 3: one group) -- a half batch of that size still fills the machine's
  // launch slots, and the host's turn-around (~50 us per round) disappears behind the other half's kernels
  const int ngroups = (count >= 32 && !(flags & 8)) ? 2 : 1;
  const int mvflags = flags & ~8;
  bspg_group grp[2];
  for (int g = 0; g < ngroups; ++g) {
    grp[g].p0 = g == 0 ? 0 : count / 2;
    grp[g].cnt = ngroups == 1 ? count : (g == 0 ? count / 2 : count - count / 2);
    grp[g].seq = 0.0;
    grp[g].flag = ctx->host_slots + BP_MIRROR + 1 + g;
    grp[g].flag_dev = ctx->host_slots_dev + BP_MIRROR + 1 + g;
    grp[g].batch = dev_batch + g;
    grp[g].inflight = false;
  }

  if (start) {                                           // the whole batch as one group
    unsigned nlive = 0;
    for (int p = 0; p < count; ++p) {
      double* c = hctl + 4 * p;
      c[SPG_ALPHA] = 0.0;
      c[SPG_BETAK] = 0.0;
      c[SPG_MODE] = host[p].status == 0 ? 1.0 : 0.0;
      c[SPG_UPDATE] = 0.0;
      nlive += host[p].status == 0;
    }
    if (nlive) {
      __atomic_thread_fence(__ATOMIC_RELEASE);
      hipLaunchKernelGGL(bspg_ctl_kernel, dim3((4 * count + CCQP_BLOCK - 1) / CCQP_BLOCK), blk, 0, s, hctl_dev, dev_ctl,
                         4 * count);
      rc = bspg_gemv(ctx, dev_probs, dev_ctl, count, n, 0, mvflags, s);
      if (!rc) rc = bspg_gemv(ctx, dev_probs, dev_ctl, count, n, 1, mvflags, s);
      if (rc) return rc;
      hipLaunchKernelGGL(bspg_dots_kernel, dim3(nblk, count), blk, 0, s, dev_probs, dev_ctl, dev_out, hout_dev, 1, n,
                         chunk, nblk, grp[0].batch, grp[0].flag_dev, nlive, count, 1.0);
      CCQP_LAUNCH_CHECK();
      rc = bspg_wait(grp[0].flag, 1.0, s);
      if (rc) return rc;
      grp[0].seq = 1.0;
      for (int p = 0; p < count; ++p) {
        ccqp_spg_host* h = host + p;
        if (h->status != 0) continue;
        const double* o = hout + 4 * p;
        h->fk = o[0];                                    // sic: g.x, :923
        h->alpha = o[1] / o[2];                          // :924
        h->mv = 2;
        h->used = 0;
        h->pending = 0;
        h->have_dd = 0;
        h->qlen = 1;
        queue[(int64_t)p * qstride] = h->fk;
      }
    }
  }

  // queue one round of a group; returns 1 if the group had anything to do
  auto launch_round = [&](bspg_group& G) -> int {
    const int p0 = G.p0, cnt = G.cnt;
    unsigned nlive = 0;
    bool any = false;
    for (int p = p0; p < p0 + cnt; ++p) {
      ccqp_spg_host* h = host + p;
      double* c = hctl + 4 * p;
      c[SPG_ALPHA] = h->alpha;
      c[SPG_BETAK] = h->betak;
      c[SPG_UPDATE] = h->pending ? 1.0 : 0.0;
      if (h->status != 0) {
        c[SPG_MODE] = 0.0;
      } else if ((double)(h->mv + 1) >= h->max_mv) {     // :938-941: the matvec that ends the budget (counted; its
        h->mv += 1;                                      // result is never used, so it is not executed here)
        h->status = h->have_dd ? 2 : 3;                  // (no d.d yet: the reference dies with UnboundLocalError)
        h->pending = 0;
        c[SPG_MODE] = 2.0;
        any = true;
      } else {
        h->mv += 1;
        c[SPG_MODE] = 1.0;
        nlive += 1;
        any = true;
      }
    }
    if (!any) return 0;
    const ccqp_pspg* probs = dev_probs + p0;
    const ccqp_psegs* segs = dev_segs ? dev_segs + p0 : nullptr;
    double* ctl = dev_ctl + 4 * p0;
    const dim3 vgrid(bp_vec_grid(n), cnt);
    __atomic_thread_fence(__ATOMIC_RELEASE);
    if (segs) {
      hipLaunchKernelGGL(bspg_ctl_kernel, dim3((4 * cnt + CCQP_BLOCK - 1) / CCQP_BLOCK), blk, 0, s, hctl_dev + 4 * p0, ctl,
                         4 * cnt);
      hipLaunchKernelGGL(bspg_step_kernel<false>, vgrid, blk, 0, s, probs, ctl, n, (const double*)nullptr);
      if (nlive) {
        if (max_nnorm > 0)
          hipLaunchKernelGGL(bspg_norm_kernel, dim3((unsigned)max_nnorm, cnt), blk, 0, s, probs, segs, ctl);
        hipLaunchKernelGGL(bspg_apply_kernel, dim3((unsigned)max_nwork, cnt), blk, 0, s, probs, segs, ctl);
      }
    } else {
      // (elementwise sets: the step launch fetches the control words itself)
      hipLaunchKernelGGL(bspg_step_kernel<true>, vgrid, blk, 0, s, probs, ctl, n, (const double*)(hctl_dev + 4 * p0));
    }
    if (!nlive) return 2;                                // only final updates: nothing will be reported
    G.seq += 1.0;
    if (bspg_gemv(ctx, probs, ctl, cnt, n, 2, mvflags, s)) return -1;      // (the message is set)
    hipLaunchKernelGGL(bspg_dots_kernel, dim3(nblk, cnt), blk, 0, s, probs, ctl, dev_out + 4 * p0, hout_dev + 4 * p0, 0,
                       n, chunk, nblk, G.batch, G.flag_dev, nlive, cnt, G.seq);
    G.inflight = true;
    return 1;
  };
  // wait for a group's queued round and do the scalar part of the iteration for its problems
  auto finish_round = [&](bspg_group& G) -> int {
    int r = bspg_wait(G.flag, G.seq, s);
    if (r) return r;
    G.inflight = false;
    for (int p = G.p0; p < G.p0 + G.cnt; ++p) {
      if (hctl[4 * p + SPG_MODE] != 1.0) continue;
      const double* o = hout + 4 * p;
      spg_host_step(host + p, queue + (int64_t)p * qstride, o[0], o[1], o[2], sources[p], 0.0);
    }
    return CCQP_OK;
  };

  for (int64_t round = 0;; ++round) {
    const bool stop = round >= max_rounds;
    int busy = 0;
    for (int g = 0; g < ngroups; ++g) {
      bspg_group& G = grp[g];
      if (G.inflight) {                                  // (queued in the previous turn of this loop)
        rc = finish_round(G);
        if (rc) return rc;
      }
      if (stop) continue;
      const int what = launch_round(G);
      if (what < 0) return CCQP_ERR_HIP;
      CCQP_LAUNCH_CHECK();
      if (what == 2) {
        // the control words of this group must not be overwritten while the copy kernel may still be reading them
        CCQP_HIP(hipStreamSynchronize(s));
      }
      busy += what != 0;
    }
    if (stop) {
      bool any = false;
      for (int p = 0; p < count; ++p) any = any || host[p].status == 0;
      *more = any ? 1 : 0;
      break;
    }
    if (!busy) break;
  }
  // x of every finished problem is final once the stream has drained
  CCQP_HIP(hipStreamSynchronize(s));
  return CCQP_OK;
}
