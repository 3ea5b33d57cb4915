// Batched launch form: ONE launch executes the same operation for many independent problems
// (blockIdx.y / .z = problem).  Used by the batch-sharded mode (BASELINE config 5: 1024 independent
// n = 8192 QPs, 128 per GPU): at that size a matvec is 83 us and a solve needs ~170 O(n) kernels, so
// per-problem launches leave the GPU half idle.  The host scheduler (ccqppy_amd/batch.py) records
// the operations of many in-flight solves and hands groups of identical operations to
// ccqp_batch_run_f64; every group becomes one launch (plus one stage-2 launch for reductions).
// The bodies are the ones of the single-problem kernels (ccqp_bodies.h): results are bitwise equal.
#include "ccqp_bodies.h"

#define BATCH_RING 16384  // descriptors in the device ring (one flush uses a few hundred)

#define BATCH_QUARTERS 4
struct ccqp_batch {
  ccqp_bop* dev;     // device ring
  ccqp_bop* host;    // pinned staging ring
  int64_t cursor;
  // one event per ring quarter, recorded behind the last flush that used descriptors of that quarter: before a later
  // flush overwrites a quarter (the ring has wrapped) the host waits for that event, so descriptors still unread by a
  // queued copy or kernel are never overwritten -- however many flushes a caller queues without synchronising
  hipEvent_t ev[BATCH_QUARTERS];
  bool ev_set[BATCH_QUARTERS];
};

static ccqp_batch* batch_of(ccqp_ctx* ctx) {
  if (ctx->batch) return ctx->batch;
  ccqp_batch* b = new ccqp_batch();
  b->dev = nullptr;
  b->host = nullptr;
  b->cursor = 0;
  for (int q = 0; q < BATCH_QUARTERS; ++q) {
    b->ev[q] = nullptr;
    b->ev_set[q] = false;
  }
  bool ok = hipMalloc(&b->dev, sizeof(ccqp_bop) * BATCH_RING) == hipSuccess &&
            hipHostMalloc(&b->host, sizeof(ccqp_bop) * BATCH_RING, hipHostMallocDefault) == hipSuccess;
  for (int q = 0; ok && q < BATCH_QUARTERS; ++q)
    ok = hipEventCreateWithFlags(&b->ev[q], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    if (b->dev) (void)hipFree(b->dev);
    if (b->host) (void)hipHostFree(b->host);
    for (int q = 0; q < BATCH_QUARTERS; ++q)
      if (b->ev[q]) (void)hipEventDestroy(b->ev[q]);
    delete b;
    return nullptr;
  }
  ctx->batch = b;
  return b;
}

void ccqp_batch_free(ccqp_ctx* ctx) {
  if (!ctx->batch) return;
  (void)hipFree(ctx->batch->dev);
  (void)hipHostFree(ctx->batch->host);
  for (int q = 0; q < BATCH_QUARTERS; ++q)
    if (ctx->batch->ev[q]) (void)hipEventDestroy(ctx->batch->ev[q]);
  delete ctx->batch;
  ctx->batch = nullptr;
}

__device__ __forceinline__ ccqp_segk segk_of(const ccqp_bop& o) {
  ccqp_segk k;
  k.kind = o.seg.kind;
  k.flags = o.seg.flags;
  k.dim = o.seg.dim;
  k.lb = o.seg.lb;
  k.ub = o.seg.ub;
  k.lb_s = o.seg.lb_s;
  k.ub_s = o.seg.ub_s;
  return k;
}

// ------------------------------------------------------------------ elementwise groups
__global__ __launch_bounds__(CCQP_BLOCK) void b_elementwise(const ccqp_bop* t, int opcode) {
  const ccqp_bop& o = t[blockIdx.y];
  const int bx = blockIdx.x, nbx = gridDim.x;
  switch (opcode) {
    case CCQP_BOP_AXPBY:
      axpby_body(o.s[0], o.p[0], o.s[1], o.p[1], const_cast<double*>(o.p[2]), o.n, bx, nbx);
      break;
    case CCQP_BOP_SCALE:
      scale_body(o.s[0], o.p[0], const_cast<double*>(o.p[1]), o.n, bx, nbx);
      break;
    case CCQP_BOP_COPY:
      copy_body(o.p[0], const_cast<double*>(o.p[1]), o.n, bx, nbx);
      break;
    case CCQP_BOP_FILL:
      fill_body(o.s[0], const_cast<double*>(o.p[0]), o.n, bx, nbx);
      break;
    case CCQP_BOP_PROJ_ELEM: {
      const int64_t off = o.seg.offset;
      proj_elem_body(segk_of(o), o.s[0], o.p[0] + off, o.s[1], o.p[1] ? o.p[1] + off : nullptr,
                     const_cast<double*>(o.p[2]) + off, bx, nbx);
      break;
    }
    case CCQP_BOP_FREE_GRADIENT_ELEM: {
      const int64_t off = o.seg.offset;
      free_gradient_body(segk_of(o), o.p[0] + off, o.p[1] + off, const_cast<double*>(o.p[2]) + off, bx, nbx);
      break;
    }
    case CCQP_BOP_CG_DIRECTION:
      cg_direction_body(o.p[0], o.p[1], o.s[0], o.p[2], const_cast<double*>(o.p[3]), o.n, bx, nbx);
      break;
    default:
      break;
  }
}

// ------------------------------------------------------------------ reduction groups
// grid (nblk, k, count).  dst of block (b, j) of problem z: the problem's partials (nblk > 1) or slab.
__global__ __launch_bounds__(CCQP_BLOCK) void b_reduce_stage1(const ccqp_bop* t, int opcode, int64_t chunk, int nblk) {
  __shared__ double lds4[4];
  const ccqp_bop& o = t[blockIdx.z];
  const int bx = blockIdx.x, j = blockIdx.y;
  double* dst = nblk == 1 ? o.slab + o.slot[0] : o.partials;
  double s = 0.0;
  switch (opcode) {
    case CCQP_BOP_DOTS:
      s = dot_body(o.p[2 * j], o.p[2 * j + 1], o.n, chunk, bx, lds4);
      break;
    case CCQP_BOP_RESIDUAL_ELEM: {
      const int64_t off = o.seg.offset;
      s = residual_body(segk_of(o), o.p[0] + off, o.p[1] + off, o.s[0], o.s[1], chunk, bx, lds4);
      break;
    }
    case CCQP_BOP_FEASIBLE_ELEM: {
      const int64_t off = o.seg.offset;
      s = feasible_body(segk_of(o), o.p[0] + off, o.p[1] + off, o.s[0], j, chunk, bx, lds4);
      break;
    }
    case CCQP_BOP_MPRGP_SPLIT: {
      const int64_t off = o.seg.offset;
      s = split_body(segk_of(o), o.p[0] + off, o.p[1] + off, const_cast<double*>(o.p[2]) + off,
                     const_cast<double*>(o.p[3]) + off, const_cast<double*>(o.p[4]) + off, chunk, bx, lds4);
      break;
    }
    case CCQP_BOP_MPRGP_NORMAL_DOT: {
      const int64_t off = o.seg.offset;
      s = normal_dot_body(segk_of(o), o.p[0] + off, o.p[1] + off, const_cast<double*>(o.p[2]) + off,
                          o.slab[o.slot[1]], chunk, bx, lds4);
      break;
    }
    case CCQP_BOP_MPRGP_PROPORTION: {
      double sb, sp;
      proportion_body(o.p[0], o.p[1], o.p[2], o.p[3], o.slab[o.slot[1]], o.n, chunk, bx, lds4, &sb, &sp);
      if (threadIdx.x == 0) {
        dst[bx] = sb;
        dst[nblk + bx] = sp;
      }
      if (nblk > 1) ccqp_reduce_finish(dst, nblk, 2, ccqp_red_counters(o.partials), o.slab + o.slot[0], lds4);
      return;
    }
    default:
      break;
  }
  if (threadIdx.x == 0) dst[(int64_t)j * nblk + bx] = s;
  // several blocks per reduction: the block of this problem's reduction j that finishes last adds the partials
  if (nblk > 1)
    ccqp_reduce_finish(dst + (int64_t)j * nblk, nblk, 1, ccqp_red_counters(o.partials) + j, o.slab + o.slot[0] + j,
                       lds4);
}

// ------------------------------------------------------------------ matvec group
// Measured at n = 8192 (scratch microbenchmark, MI355X): ONE launch streaming 8-32 matrices -- whether
// with blockIdx.y = problem or with a 1-D grid walking the problems in order -- reaches 5.4-6.1 TB/s,
// because the waves drift apart and several matrices are streamed at once (DRAM page locality), while
// the same matvecs as back-to-back single-matrix launches reach 6.5 TB/s (a launch costs the host
// ~2 us, the kernel runs 83 us).  So a matvec group is issued as `count` launches of the
// single-problem kernel from this C loop: no Python between them, identical bits by construction.
int ccqp_gemv_launch(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, const double* b, double* y,
                     double* g, int64_t rows, int64_t cols, int variant, hipStream_t s,
                     const double* skip);

static inline int vec_grid(int64_t n) {
  int64_t g = (n + CCQP_BLOCK - 1) / CCQP_BLOCK;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

static int launch_group(ccqp_ctx* ctx, int opcode, int count, const ccqp_bop* host, const ccqp_bop* dev,
                        hipStream_t s) {
  const ccqp_bop& h = host[0];
  switch (opcode) {
    case CCQP_BOP_AXPBY:
    case CCQP_BOP_SCALE:
    case CCQP_BOP_COPY:
    case CCQP_BOP_FILL:
    case CCQP_BOP_CG_DIRECTION: {
      int64_t nmax = 0;
      for (int i = 0; i < count; ++i) nmax = host[i].n > nmax ? host[i].n : nmax;
      if (nmax == 0) return CCQP_OK;
      hipLaunchKernelGGL(b_elementwise, dim3(vec_grid(nmax), count), dim3(CCQP_BLOCK), 0, s, dev, opcode);
      return CCQP_OK;
    }
    case CCQP_BOP_PROJ_ELEM:
    case CCQP_BOP_FREE_GRADIENT_ELEM: {
      int64_t nmax = 0;
      for (int i = 0; i < count; ++i) nmax = host[i].seg.dim > nmax ? host[i].seg.dim : nmax;
      if (nmax == 0) return CCQP_OK;
      hipLaunchKernelGGL(b_elementwise, dim3(vec_grid(nmax), count), dim3(CCQP_BLOCK), 0, s, dev, opcode);
      return CCQP_OK;
    }
    case CCQP_BOP_DOTS:
    case CCQP_BOP_RESIDUAL_ELEM:
    case CCQP_BOP_FEASIBLE_ELEM:
    case CCQP_BOP_MPRGP_SPLIT:
    case CCQP_BOP_MPRGP_NORMAL_DOT:
    case CCQP_BOP_MPRGP_PROPORTION: {
      const bool seg_sized = opcode != CCQP_BOP_DOTS && opcode != CCQP_BOP_MPRGP_PROPORTION;
      const int64_t n = seg_sized ? h.seg.dim : h.n;
      int k = 1;                       // reductions per problem
      if (opcode == CCQP_BOP_DOTS || opcode == CCQP_BOP_FEASIBLE_ELEM) k = h.k;
      if (opcode == CCQP_BOP_MPRGP_PROPORTION) k = 2;
      const bool kvar = opcode == CCQP_BOP_DOTS || opcode == CCQP_BOP_FEASIBLE_ELEM;
      for (int i = 0; i < count; ++i) {
        const int64_t ni = seg_sized ? host[i].seg.dim : host[i].n;
        CCQP_REQUIRE(ni == n && (!kvar || host[i].k == h.k), "batched reduction: problems differ in size");
        CCQP_REQUIRE(ccqp_slot_ok(host[i].slot[0], k), "slot range outside the slab");
      }
      CCQP_REQUIRE(k >= 1 && k <= CCQP_MAX_STEPS, "batched reduction: bad k");
      int64_t chunk;
      int nblk;
      ccqp_red_geometry(n, &chunk, &nblk);
      CCQP_REQUIRE((int64_t)k * nblk <= (int64_t)CCQP_RED_ROWS * CCQP_RED_MAXBLK, "k * blocks exceeds the workspace");
      const int gy = opcode == CCQP_BOP_MPRGP_PROPORTION ? 1 : k;
      hipLaunchKernelGGL(b_reduce_stage1, dim3(nblk, gy, count), dim3(CCQP_BLOCK), 0, s, dev, opcode, chunk, nblk);
      return CCQP_OK;
    }
    case CCQP_BOP_GEMV: {
      // All matvec groups of the process go through ONE dedicated stream (event-chained to the
      // caller's stream): matvecs of different lane groups then run one after the other at full
      // bandwidth instead of overlapping and splitting it, while the small kernels of the other
      // groups still overlap with them.
      // (one stream per device; the host side of this library is driven by one thread per process)
      hipStream_t ms = s;
      {
        const int rc = ccqp_mv_stream_enter(ctx, s, &ms);
        if (rc) return rc;
      }
      for (int i = 0; i < count; ++i) {
        const ccqp_bop& o = host[i];
        CCQP_REQUIRE(o.n >= 0 && o.n2 >= 0 && o.lda >= o.n2, "batched matvec: bad shape");
        if (o.n == 0) continue;
        CCQP_REQUIRE(o.p[0] && o.p[3] && (o.n2 == 0 || o.p[1]), "batched matvec: NULL operand");
        CCQP_REQUIRE(o.p[4] == nullptr || o.p[2] != nullptr, "batched matvec: g requested but b is NULL");
        int rc = ccqp_gemv_launch(ctx, o.p[0], o.lda, o.p[1], o.p[2], const_cast<double*>(o.p[3]),
                                  const_cast<double*>(o.p[4]), o.n, o.n2, 0, ms, nullptr);
        if (rc) return rc;
      }
      return ccqp_mv_stream_leave(ctx, s, ms);
    }
    default:
      ccqp_set_error("ccqp_batch_run_f64: unknown opcode %d", opcode);
      return CCQP_ERR_UNSUPPORTED;
  }
}

extern "C" int ccqp_batch_run_f64(ccqp_ctx* ctx, int ngroups, const int* opcodes, const int* counts,
                                  const ccqp_bop* ops, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(ngroups >= 0, "negative group count");
  if (ngroups == 0) return CCQP_OK;
  CCQP_REQUIRE(opcodes && counts && ops, "NULL argument");
  int64_t total = 0;
  for (int gidx = 0; gidx < ngroups; ++gidx) {
    CCQP_REQUIRE(counts[gidx] >= 1 && counts[gidx] <= 65535, "group size outside [1, 65535]");
    total += counts[gidx];
  }
  CCQP_REQUIRE(total <= BATCH_RING / 4, "too many descriptors in one flush");
  ccqp_batch* b = batch_of(ctx);
  if (!b) {
    ccqp_set_error("ccqp_batch_run_f64: cannot allocate the descriptor ring");
    return CCQP_ERR_HIP;
  }
  if (b->cursor + total > BATCH_RING) b->cursor = 0;
  const int64_t qlen = BATCH_RING / BATCH_QUARTERS;
  const int q0 = (int)(b->cursor / qlen), q1 = (int)((b->cursor + total - 1) / qlen);
  for (int q = q0; q <= q1; ++q)
    if (b->ev_set[q]) CCQP_HIP(hipEventSynchronize(b->ev[q]));   // (returns at once unless the ring has lapped the GPU)
  ccqp_bop* hst = b->host + b->cursor;
  ccqp_bop* dev = b->dev + b->cursor;
  b->cursor += total;
  memcpy(hst, ops, sizeof(ccqp_bop) * (size_t)total);
  hipStream_t s = (hipStream_t)stream;
  CCQP_HIP(hipMemcpyAsync(dev, hst, sizeof(ccqp_bop) * (size_t)total, hipMemcpyHostToDevice, s));
  int64_t at = 0;
  for (int gidx = 0; gidx < ngroups; ++gidx) {
    int rc = launch_group(ctx, opcodes[gidx], counts[gidx], hst + at, dev + at, s);
    if (rc) return rc;
    at += counts[gidx];
  }
  CCQP_LAUNCH_CHECK();
  for (int q = q0; q <= q1; ++q) {
    CCQP_HIP(hipEventRecord(b->ev[q], s));
    b->ev_set[q] = true;
  }
  return CCQP_OK;
}

extern "C" int ccqp_batch_serialize_matvecs(ccqp_ctx* ctx, int on) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  ctx->mv_serialize = on ? 1 : 0;
  return CCQP_OK;
}

extern "C" int ccqp_ctx_workspace(ccqp_ctx* ctx, double** slab, double** partials) {
  CCQP_REQUIRE(ctx != nullptr && slab != nullptr && partials != nullptr, "NULL argument");
  *slab = ctx->slots;
  *partials = ctx->partials;
  return CCQP_OK;
}

// ------------------------------------------------------------------ gathered scalar read
// The lanes of a group each own a scalar slab; one tiny kernel collects the slots every lane asked
// for into one staging area, ONE device->host copy brings them over (instead of one copy per lane).
#define GATHER_MAX 64       // lanes per gather
#define GATHER_WIDTH 16     // slots per lane

struct gather_args {
  const double* slab[GATHER_MAX];
  int slot0[GATHER_MAX];
  int count[GATHER_MAX];
};

__global__ void gather_kernel(gather_args a, int n, double* out) {
  const int lane = blockIdx.x;
  const int j = threadIdx.x;
  if (lane < n && j < a.count[lane]) out[lane * GATHER_WIDTH + j] = a.slab[lane][a.slot0[lane] + j];
}

extern "C" int ccqp_scalars_gather_async(ccqp_ctx* ctx, int n, double* const* slabs, const int* slot0,
                                         const int* count, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(n >= 1 && n <= GATHER_MAX, "n outside [1, 64]");
  CCQP_REQUIRE(slabs && slot0 && count, "NULL argument");
  if (!ctx->gather_dev) {
    CCQP_HIP(hipMalloc(&ctx->gather_dev, sizeof(double) * GATHER_MAX * GATHER_WIDTH));
    CCQP_HIP(hipHostMalloc(&ctx->gather_host, sizeof(double) * GATHER_MAX * GATHER_WIDTH, hipHostMallocDefault));
  }
  gather_args a;
  for (int i = 0; i < n; ++i) {
    CCQP_REQUIRE(slabs[i] != nullptr, "NULL slab");
    CCQP_REQUIRE(count[i] >= 0 && count[i] <= GATHER_WIDTH && ccqp_slot_ok(slot0[i], count[i]), "bad slot range");
    a.slab[i] = slabs[i];
    a.slot0[i] = slot0[i];
    a.count[i] = count[i];
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gather_kernel, dim3(n), dim3(GATHER_WIDTH), 0, s, a, n, ctx->gather_dev);
  CCQP_LAUNCH_CHECK();
  CCQP_HIP(hipMemcpyAsync(ctx->gather_host, ctx->gather_dev, sizeof(double) * (size_t)n * GATHER_WIDTH,
                          hipMemcpyDeviceToHost, s));
  CCQP_HIP(hipEventRecord(ctx->read_event, s));
  return CCQP_OK;
}

extern "C" int ccqp_scalars_gather_host(ccqp_ctx* ctx, int n, double* host_out) {
  CCQP_REQUIRE(ctx != nullptr && host_out != nullptr, "NULL argument");
  CCQP_REQUIRE(n >= 1 && n <= GATHER_MAX && ctx->gather_host != nullptr, "no gather pending");
  CCQP_HIP(hipEventSynchronize(ctx->read_event));
  memcpy(host_out, ctx->gather_host, sizeof(double) * (size_t)n * GATHER_WIDTH);
  return CCQP_OK;
}
