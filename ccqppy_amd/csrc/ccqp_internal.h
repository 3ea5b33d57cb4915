// Internal declarations shared by the translation units of libccqp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "ccqp_hip.h"

#define CCQP_BLOCK 256          // 4 wave64 per workgroup
#define CCQP_RED_CHUNK 2048     // elements per stage-1 reduction block (8 per thread)
#define CCQP_RED_MAXBLK 1024    // stage-1 blocks per reduction (partials per dot)
#define CCQP_RED_COUNTERS 32    // arrival counters behind the partials (one per reduction of a launch)
#define CCQP_RED_ROWS (CCQP_MAX_STEPS + 1)   // rows of CCQP_RED_MAXBLK partials: up to 16 sums per launch + 1
#define CCQP_PARTIALS_DOUBLES ((size_t)CCQP_RED_ROWS * CCQP_RED_MAXBLK + CCQP_RED_COUNTERS / 2)

#define CCQP_LDS_KERNELS 128       // distinct kernels with more than 64 KiB of dynamic LDS a context can keep track of

struct ccqp_batch;                 // descriptor ring of the batched launch form (ccqp_batch.hip)
void ccqp_batch_free(struct ccqp_ctx* ctx);

struct ccqp_ctx {
  ccqp_batch* batch;
  int mv_serialize;     // batched matvec groups go through the process-wide matvec stream
  hipEvent_t mv_ev_in, mv_ev_out;
  double* gather_dev;   // staging of ccqp_scalars_gather_async (device / pinned host)
  double* gather_host;
  int device;
  int num_cu;
  char arch[64];
  double* slots;      // device, CCQP_NUM_SLOTS doubles
  double* partials;   // device, CCQP_PARTIALS_DOUBLES doubles: partial sums, then the arrival counters
  double* host_slots; // pinned, coherent host mirror (CCQP_NUM_SLOTS doubles + the publish flag behind them)
  double* host_slots_dev;       // the same memory as the device sees it
  unsigned long long read_seq;  // sequence number of the last published read
  hipEvent_t read_event;  // marks the end of the last ccqp_scalars_read_async copy
  double* xchg_host;      // pinned, coherent exchange area of the host-in-the-loop lock-step driver (ccqp_bspg.hip),
  double* xchg_host_dev;  // grown on demand; the same memory as the device sees it
  size_t xchg_doubles;
  int64_t opt[CCQP_OPT_COUNT];  // tuning options (ccqp_ctx_set_option); 0 = the tuned default
  // kernels whose dynamic-LDS limit this context has raised on ITS device (hipFuncAttributeMaxDynamicSharedMemorySize is
  // kept per device by the runtime; a context belongs to one device and is driven by one host thread)
  struct {
    const void* fn;
    size_t bytes;
  } lds_raised[CCQP_LDS_KERNELS];
  int lds_nraised;
};

// tuning option `key` of a context (0 = default; a NULL context has only defaults)
static inline int64_t ccqp_opt(const struct ccqp_ctx* ctx, int key) { return ctx ? ctx->opt[key] : 0; }

// Makes sure kernel `fn` may be launched with `bytes` of dynamic LDS on the context's device: the first launch of a
// kernel that needs more than 64 KiB (and any later one that needs more than was asked for) raises the function's
// attribute, once per context -- never per process (ccqp_ctx.hip).
int ccqp_lds_raise(struct ccqp_ctx* ctx, const void* fn, size_t bytes);

void ccqp_set_error(const char* fmt, ...);

// ccqp_batch_serialize_matvecs: with the switch on, a group of matvec launches leaves the caller's stream `s` for the
// process-wide matvec stream of the context's device (*ms; event-chained behind what `s` holds) and `s` waits for the
// group afterwards; with it off *ms = s and both calls do nothing (ccqp_ctx.hip)
int ccqp_mv_stream_enter(struct ccqp_ctx* ctx, hipStream_t s, hipStream_t* ms);
int ccqp_mv_stream_leave(struct ccqp_ctx* ctx, hipStream_t s, hipStream_t ms);

// matvec timing hook (ccqp_mv_timing_begin / _end, ccqp_ctx.hip): event pairs around the batched matvec launches
extern int ccqp_mvt_on;
void ccqp_mvt_mark(hipStream_t s, int end);

#define CCQP_REQUIRE(cond, msg)                                   \
  do {                                                            \
    if (!(cond)) {                                                \
      ccqp_set_error("%s: %s", __func__, msg);                    \
      return CCQP_ERR_INVALID;                                    \
    }                                                             \
  } while (0)

#define CCQP_HIP(call)                                                            \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      ccqp_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e_)); \
      return CCQP_ERR_HIP;                                                        \
    }                                                                             \
  } while (0)

#define CCQP_LAUNCH_CHECK()                                                        \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      ccqp_set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(e_)); \
      return CCQP_ERR_HIP;                                                         \
    }                                                                              \
  } while (0)

static inline int ccqp_slot_ok(int slot, int count) {
  return slot >= 0 && count >= 0 && slot + count <= CCQP_NUM_SLOTS;
}

// Stage-1 geometry of every reduction: a function of n ONLY (determinism contract).
static inline void ccqp_red_geometry(int64_t n, int64_t* chunk, int* nblk) {
  int64_t c = CCQP_RED_CHUNK;
  if ((n + c - 1) / c > CCQP_RED_MAXBLK) {
    c = (n + CCQP_RED_MAXBLK - 1) / CCQP_RED_MAXBLK;
    c = (c + CCQP_BLOCK - 1) / CCQP_BLOCK * CCQP_BLOCK;
  }
  int64_t nb = n > 0 ? (n + c - 1) / c : 1;
  *chunk = c;
  *nblk = (int)nb;
}

#ifdef __HIPCC__
// ---------------------------------------------------------------- device helpers
// Sum over the 64 lanes of a wave; butterfly => every lane holds the same bits.
__device__ __forceinline__ double ccqp_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over a 256-thread workgroup in a fixed order (wave 0..3); result valid in thread 0.
__device__ __forceinline__ double ccqp_block_sum(double v, double* lds4) {
  v = ccqp_wave_sum(v);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) lds4[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) s = ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
  __syncthreads();
  return s;
}

// K sums at once: the same wave butterflies and the same (wave 0..3) order per value as K calls of ccqp_block_sum,
// hence the same bits, but two workgroup barriers instead of 2 K (results valid in thread 0).
template <int K>
__device__ __forceinline__ void ccqp_block_sum_multi(double* v) {
  __shared__ double lds[4 * K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = ccqp_wave_sum(v[k]);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) lds[4 * k + wave] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = ((lds[4 * k] + lds[4 * k + 1]) + lds[4 * k + 2]) + lds[4 * k + 3];
  }
  __syncthreads();
}

// Arrival counters of a reduction workspace (`partials` of a context): they sit behind the partial sums.
__device__ __forceinline__ unsigned* ccqp_red_counters(double* partials) {
  return reinterpret_cast<unsigned*>(partials + (size_t)CCQP_RED_ROWS * CCQP_RED_MAXBLK);
}

// Single-launch reductions.  Thread 0 of every block has just stored the block's partial sum(s); the block
// that arrives LAST at the counter adds the `nred` rows of `nblk` partials -- always in index order, whichever
// block that is, so the result is the one the former second kernel produced -- and writes out[0..nred).
// Protocol (guide, "inter-workgroup visibility"): release fence by the storing thread, agent-scope counter,
// acquire fence by one lane, barrier, then plain loads.  The counter is left at zero for the next launch.
__device__ __forceinline__ void ccqp_reduce_finish(const double* partials, int nblk, int nred, unsigned* counter,
                                                   double* out, double* lds4) {
  __shared__ int is_last;
  // Structural form of the invariant the lock-step kernels rely on (ccqp_lockstep.h:bp_meet): every wave of this
  // block has finished reading whatever the launch reads at its start (skip / state words, operands) before the
  // block takes its ticket -- the last block to arrive may overwrite such words right after this function.  The
  // callers' bodies happen to contain a barrier already; this one makes it independent of them.
  __syncthreads();
  if (threadIdx.x == 0) {
    // the guide's counter hand-off (cdna_hip_programming.md, in-launch reduction / Guideline 16), to the letter:
    // drain the store, agent-scope release, drain again (ROCm 7.2 can drop the fence's own wait), relaxed
    // agent-scope ticket; the last arriver takes an agent-scope acquire before anybody loads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = prev + 1u == (unsigned)nblk;
    if (is_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!is_last) return;
  for (int r = 0; r < nred; ++r) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += CCQP_BLOCK) acc += partials[(int64_t)r * nblk + i];
    const double s = ccqp_block_sum(acc, lds4);
    if (threadIdx.x == 0) out[r] = s;
  }
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// np.isclose(a, b) with the default rtol=1e-5, atol=1e-8 (numpy/_core/numeric.py):
//   (|a-b| <= atol + rtol*|b|) & isfinite(b) | (a == b)
__device__ __forceinline__ bool ccqp_isclose(double a, double b) {
  const double d = fabs(a - b);
  const double t = 1e-5 * fabs(b);
  const double tol = 1e-8 + t;
  return ((d <= tol) && isfinite(b)) || (a == b);
}

struct ccqp_segk {  // by-value kernel copy of a descriptor, block-local pointers
  int kind, flags;
  int64_t dim;
  const double* lb;
  const double* ub;
  double lb_s, ub_s;
};

// bound * mask exactly as numpy evaluates it: float bound -> float multiply (bound*0.0 keeps the
// bound's sign / turns inf into NaN); integer-typed bound -> integer multiply -> +0.0 or bound.
__device__ __forceinline__ double ccqp_bound_term(double bound, bool mask, bool is_int) {
  if (is_int) return mask ? bound : 0.0;
  return bound * (mask ? 1.0 : 0.0);
}

// Literal evaluation of the reference's elementwise projections (solution_spaces.py:200-201,
// 276-277, 363-366), one rounding per operation (this file is built with -ffp-contract=off).
__device__ __forceinline__ double ccqp_proj_elem(int kind, int flags, double x, double lb, double ub) {
  switch (kind) {
    case CCQP_OP_LOWER: {
      const bool m = x < lb;
      const double t1 = ccqp_bound_term(lb, m, flags & CCQP_SEG_LB_INT);
      const double t2 = x * (m ? 0.0 : 1.0);
      return t1 + t2;
    }
    case CCQP_OP_UPPER: {
      const bool m = x > ub;
      const double t1 = ccqp_bound_term(ub, m, flags & CCQP_SEG_UB_INT);
      const double t2 = x * (m ? 0.0 : 1.0);
      return t1 + t2;
    }
    case CCQP_OP_BOX: {
      const bool hi = x > ub;
      const bool lo = x < lb;
      const bool li = flags & CCQP_SEG_LB_INT, ui = flags & CCQP_SEG_UB_INT;
      const double t1 = ccqp_bound_term(lb, lo, li);
      const double t2 = ccqp_bound_term(ub, hi, ui);
      const double s = t1 + t2;
      double t3 = x * (hi ? 0.0 : 1.0);
      t3 = t3 * (lo ? 0.0 : 1.0);
      return s + t3;
    }
    default:
      return x;  // identity
  }
}
#endif  // __HIPCC__
