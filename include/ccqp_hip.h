/*
 * ccqp_hip.h -- C ABI of libccqp_hip.so, the MI355X (gfx950) engine behind the
 * CCQPpy-compatible `ccqppy_amd.solvers.CCQPSolver*` / `ccqppy_amd.solution_spaces.*ProjOp`.
 *
 * The reference (palmerb4/CCQPpy) has no FFI: its hot path is numpy -> OpenBLAS calls made
 * from Python.  Each entry point below replaces one family of those numpy calls; the
 * reference call sites are cited as file:line relative to /root/reference/src/ccqppy/.
 * INTEGRATION.md shows the ctypes stub a CCQPpy maintainer would add.
 *
 * Conventions
 *  - plain C, no torch / C++ types; all vectors are fp64, contiguous, DEVICE pointers;
 *    sizes are int64_t; `stream` is a hipStream_t passed as void*.
 *  - every call returns 0 (CCQP_OK) or an error code; the message is available from
 *    ccqp_last_error() (thread-local).  Nothing here allocates or frees caller memory.
 *  - every call is asynchronous on `stream` except ccqp_scalars_read(), which copies the
 *    requested scalar slots to the host and waits for the stream.
 *  - reductions (dot products, norms, counts) land in the context's device "scalar slab":
 *    `slot` arguments index doubles in that slab.  Their summation order is a function of
 *    the vector length only (never of the grid, the GPU count or timing), so results are
 *    bitwise reproducible run to run and rank to rank.
 *  - elementwise kernels round after every arithmetic operation exactly like numpy
 *    (no FMA contraction); only the matvec and the reductions use fused multiply-add.
 */
#ifndef CCQP_HIP_H
#define CCQP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: struct ccqp_papgd grew `ga` / `Aya` and the look-ahead of the lock-step forms became an explicit request (flags
 *    bit 17, CCQP_LOCKSTEP_LOOKAHEAD) instead of "ga != NULL"; ccqp_bpipe_matvec_form takes the context; tuning / A-B
 *    switches are per-context options (ccqp_ctx_set_option) instead of environment variables.
 *    Additive since (same version): ccqp_b{apgd,pipe,mprgp,spg}_stage_one_f64, ccqp_bapgd_fused_form,
 *    ccqp_bmprgp_fused_form, ccqp_read_f64, ccqp_gemvk_f64 / ccqp_shard_gemvk_allgather_f64 / ccqp_unpack_gathered_f64,
 *    the options of CCQP_OPT_*; `flags` bit 18 (CCQP_LOCKSTEP_FUSED_ROUNDS) is a request of ccqp_bmprgp_start_f64 /
 *    _round_f64 too (their tuning word ends at bit 15, as that of ccqp_bapgd_*). */
#define CCQP_ABI_VERSION 2

enum {
  CCQP_OK = 0,
  CCQP_ERR_INVALID = 1,     /* bad argument (null pointer, negative size, slot out of range ...) */
  CCQP_ERR_HIP = 2,         /* a HIP runtime call failed */
  CCQP_ERR_UNSUPPORTED = 3  /* valid request this build does not implement */
};

#define CCQP_NUM_SLOTS 64   /* doubles in the scalar slab */
#define CCQP_MAX_DOTS 8     /* dot products per ccqp_dots_f64 call */

typedef struct ccqp_ctx ccqp_ctx;

/* ---- context ------------------------------------------------------------------------ */
int ccqp_abi_version(void);
const char* ccqp_last_error(void);
/* Binds to HIP device `device` (the caller keeps that device current when calling in). */
int ccqp_ctx_create(int device, ccqp_ctx** out);
int ccqp_ctx_destroy(ccqp_ctx* ctx);
/* num_cu: compute units of the device; arch: e.g. "gfx950" copied into buf (<= buflen). */
int ccqp_ctx_info(ccqp_ctx* ctx, int* device, int* num_cu, char* arch_buf, int buflen);
/* Per-context tuning options -- the A/B switches of tools/ and tests/.  None of them changes a result bit; value 0 is
 * always "the tuned default".  They are read by the launches made with THIS context only (nothing in the library reads
 * the environment); not thread-safe against launches in flight on the same context (one host thread per context).
 *   GEMV_ROW_STRIDE        k >= 1: a wave's R row streams lie k rows apart (LDS-staged matvecs; default: ~1 MiB apart)
 *   GEMV2_WPB              1..8 waves per workgroup of the two-vector launch ccqp_gemv2_f64 (default 4)
 *   GEMV2_SHAPE            <R><S> of its ring as a two-digit number: 44, 43, 46, 48, 82, 83, 84 (rows per wave, ring slots)
 *   GEMV2_GROUP            row sums a wave parks before it stores them: 64 .. 512 (default 512)
 *   BMV3_SHAPE             <R><S><W> of the three-vector persistent launch: 824, 833, 444, 434
 *   LOOKAHEAD_MIN_N        smallest n at which lock-step APGD looks ahead (>= 128; default 1024)
 *   MPRGP_LOOKAHEAD_MIN_N  the same for lock-step MPRGP (default 512)
 *   MPRGP_LOOKAHEAD, APGD_LOOKAHEAD   1 = the lock-step form does not look ahead (A/B of the shared passes)
 *   BMV_STAGGER            1 = no staggered start of the persistent batched matvec's waves
 *   MPRGP_FUSED_TAIL       1 = never the fused stage launch, 2 = wherever it can run (default: by shape)
 *   PERS1_MIN_N            smallest n at which a lock-step batch of ONE problem takes the persistent multi-vector matvec
 *   BMV_LOADS              matrix loads of the persistent batched matvecs: 1 = non-temporal, 2 = plain (default: by size)
 *   SMALL_SHARING          lock-step APGD BELOW the persistent matvec's sizes: fused rounds whose two products come from one
 *                          launch of the VALU kernel: 1 = never, 2 = whatever the batch (default: batches with <= 256 MiB
 *                          of matrices)
 *   PERS_MIN_MIB           MiB of matrices from which a lock-step batch of n >= 4096 takes the persistent matvec (512)
 *   BPIPE_STEP_IN_TAIL     1 = lock-step PGD / BBPGD / BBPGDf always launch their step (default: problems of one reduction
 *                          chunk, n <= 2048, take the next iteration's step in the tail launch)
 *   APGD_UPDATE_IN_DECISION lock-step APGD rounds update inside the decision launch: 1 = never, 2 = every one-chunk
 *                          problem (default: n <= 1024) */
enum {
  CCQP_OPT_GEMV_ROW_STRIDE = 0,
  CCQP_OPT_GEMV2_WPB = 1,
  CCQP_OPT_GEMV2_SHAPE = 2,
  CCQP_OPT_GEMV2_GROUP = 3,
  CCQP_OPT_BMV3_SHAPE = 4,
  CCQP_OPT_LOOKAHEAD_MIN_N = 5,
  CCQP_OPT_MPRGP_LOOKAHEAD_MIN_N = 6,
  CCQP_OPT_MPRGP_LOOKAHEAD = 7,
  CCQP_OPT_APGD_LOOKAHEAD = 8,
  CCQP_OPT_BMV_STAGGER = 9,
  CCQP_OPT_MPRGP_FUSED_TAIL = 10,
  CCQP_OPT_PERS1_MIN_N = 11,
  CCQP_OPT_BMV_LOADS = 12,
  CCQP_OPT_SMALL_SHARING = 13,
  CCQP_OPT_PERS_MIN_MIB = 14,
  CCQP_OPT_BPIPE_STEP_IN_TAIL = 15,
  CCQP_OPT_APGD_UPDATE_IN_DECISION = 16,
  CCQP_OPT_COUNT = 17
};
int ccqp_ctx_set_option(ccqp_ctx* ctx, int key, int64_t value);
int ccqp_ctx_get_option(ccqp_ctx* ctx, int key, int64_t* value);
/* Copy slots [slot0, slot0+count) to host_out and wait for `stream` (the only sync point). */
int ccqp_scalars_read(ccqp_ctx* ctx, int slot0, int count, double* host_out, void* stream);
/* Split form for callers that interleave several solves on several streams (batch-sharded mode):
 * ccqp_scalars_read_async enqueues the device->pinned-host copy on `stream`, records the context's
 * event behind it and returns at once; ccqp_scalars_host waits for that event (a no-op once the copy
 * has landed) and copies the values out of the pinned mirror.  One context per stream. */
int ccqp_scalars_read_async(ccqp_ctx* ctx, int slot0, int count, void* stream);
int ccqp_scalars_host(ccqp_ctx* ctx, int slot0, int count, double* host_out);
/* Blocking read of count <= 64 doubles at ANY device address through the context's pinned mirror (publish + poll, as
 * ccqp_scalars_read): e.g. the result words of a lock-step problem's own slab. */
int ccqp_read_f64(ccqp_ctx* ctx, const double* dev, int count, double* host_out, void* stream);
/* Device address of the slab (for kernels of the caller's own / for tests). */
int ccqp_scalars_devptr(ccqp_ctx* ctx, double** out);

/* ---- a1: dense matvec  y[0:rows] = A[rows x cols] . x --------------------------------
 * replaces `A.dot(v)` at solvers.py:133,148,264,275,290,440,452,467,625,635,641,764,775,781,
 * 922,924,938,1065,1077,1086,1100,1129,1152,1163,1172,1174.
 * A is row-major with leading dimension lda (doubles).  Row sums are accumulated in an order
 * that depends on `cols` only, so a row block [r0, r1) of A gives bitwise the same y[r0:r1]
 * as the full matrix does -> row-sharded multi-GPU runs reproduce the 1-GPU iterates.
 * variant: 0 = tuned default; other values select tuning variants and the A/B forms (MFMA: bit 9; rows staged through
 * LDS rings by LDS-DMA loads: bit 24, with bits 20-22 its waves' start stagger -- 0 = by size, 1 = none, 2 .. 6 = 1, 4,
 * 16, 32, 64 units; (variant & 0xF) == 0xF: the VALU form even where the default would not take it)
 * -- see DESIGN.md 3.1 and ccqp_gemv.hip.  Default by size: from 8 GiB on (an even number of columns, 16-byte aligned rows and x) the
 * LDS-staged form as ONE persistent launch; below, the VALU form -- as several launches over row blocks that fill the
 * machine once ("rounds") where the LDS form does not apply to a >= 32 GiB matrix.  Same bits whichever runs.
 */
int ccqp_gemv_f64(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, double* y,
                  int64_t rows, int64_t cols, int variant, void* stream);
/* How ccqp_gemv_f64 / ccqp_gemv_epilogue_f64 will launch a rows x cols matvec (the streaming form): the rows every
 * launch but the last covers, and the number of launches -- what a profile's per-launch figures refer to. */
int ccqp_gemv_plan(ccqp_ctx* ctx, int64_t rows, int64_t cols, int variant, int64_t* rows_per_launch,
                   int64_t* launches);
/* Name of the kernel a rows x cols matvec runs as ("gemv_stream_kernel" | "gemv_lds_kernel" | "gemv_mfma_kernel";
 * 16-byte aligned operands assumed): the key of its rows in a rocprofv3 table. */
int ccqp_gemv_form(ccqp_ctx* ctx, int64_t rows, int64_t cols, int variant, char* name, int buflen);
/* TWO matvecs with the same matrix in ONE pass over it: y1 = A x1, y2 = A x2 (g_i = y_i + b where g_i != NULL).
 * Replaces the two `A.dot(...)` of an APGD iteration (solvers.py:275 A.dot(yk) of the NEXT iteration and :290
 * A.dot(xkp1): y_{k+1} = (1 + beta) x_{k+1} - beta x_k is known as soon as x_{k+1} is, :313-316).  Each product has
 * the bits ccqp_gemv_f64 / ccqp_gemv_epilogue_f64 give it; the matrix is read from HBM once (ccqp_gemv2_form says
 * "gemv_lds2_kernel": aligned operands, an even number of columns, by default from 2^24 elements on -- or where the
 * variant word asks for the LDS-staged form), otherwise the call is two single matvecs.
 * y1, y2, g1, g2 must not alias each other or the inputs. */
int ccqp_gemv2_f64(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x1, const double* x2, const double* b,
                   double* y1, double* g1, double* y2, double* g2, int64_t rows, int64_t cols, int variant, void* stream);
int ccqp_gemv2_form(ccqp_ctx* ctx, int64_t rows, int64_t cols, int variant, char* name, int buflen);
/* Up to THREE matvecs with one SQUARE matrix in ONE pass over it: y[j] = A x[j] (g[j] = y[j] + b where g[j] != NULL) for
 * every j < 3 with y[j] != NULL.  Replaces the A.dot(dx), A.dot(xk) and A.dot(p) that follow each other across the end
 * of an MPRGP iteration (solvers.py:1129 | :1163, then :1086 and :1100 of the next iteration: at the A.dx call x_{k+1}
 * and p_{k+1} are known).  Each product has the bits of its own ccqp_gemv_f64 / ccqp_gemv_epilogue_f64 call; one pass
 * (ccqp_gemv3_form: "bp_gemv3_pers_kernel") for aligned operands and an even n, by default from n = 4096 on,
 * otherwise single matvecs.  Outputs must not alias inputs. */
int ccqp_gemv3_f64(ccqp_ctx* ctx, const double* A, int64_t lda, const double* const* x, const double* b,
                   double* const* y, double* const* g, int64_t n, int variant, void* stream);
int ccqp_gemv3_form(ccqp_ctx* ctx, int64_t n, int variant, char* name, int buflen);
/* k = 1, 2 or 3 matvecs with one rows x cols BLOCK (any shape: a rank's row block) in one pass over it: y[j] = A x[j],
 * g[j] = y[j] + b where g[j] != NULL, every j < k given.  k = 1 / 2: ccqp_gemv_epilogue_f64 / ccqp_gemv2_f64; k = 3:
 * "gemv_lds3_kernel" wherever the pair kernel would take one pass, else three single matvecs (ccqp_gemvk_form).  Each
 * product has the bits of its own ccqp_gemv_f64 call.  Replaces, on a row-sharded matrix, the A.dot(dx), A.dot(xk) and
 * A.dot(p) across the end of an MPRGP iteration (solvers.py:1129 | :1163, :1086, :1100). */
int ccqp_gemvk_f64(ccqp_ctx* ctx, const double* A, int64_t lda, int k, const double* const* x, const double* b,
                   double* const* y, double* const* g, int64_t rows, int64_t cols, int variant, void* stream);
int ccqp_gemvk_form(ccqp_ctx* ctx, int k, int64_t rows, int64_t cols, int variant, char* name, int buflen);
/* Measurement hook (kernel-level roofline figures measured live, not only under a profiler): between
 * begin and end every BATCHED matvec launch of the lock-step drivers (ccqp_bpipe_*, ccqp_bapgd_*, ccqp_bmprgp_*,
 * ccqp_bsegs_*, ccqp_bspg_*) is bracketed by a pair of HIP events on its own stream, up to `max_pairs` of them (later
 * launches are not timed).  ccqp_mv_timing_end waits for the last recorded event, writes the elapsed milliseconds of
 * the first min(cap, recorded) launches to ms_out, the number recorded to *count, and switches the hook off.  One
 * recorder per process (one process per GPU); never changes a result bit. */
int ccqp_mv_timing_begin(int max_pairs);
int ccqp_mv_timing_end(double* ms_out, int cap, int* count);
/* Same with a fused epilogue: y = A.x ; if g != NULL: g = y + b, one extra rounding exactly like
 * numpy's `A.dot(x) + b` (solvers.py:133,148,280,625,641,922,1065,1090).  b, y, g have `rows` entries. */
int ccqp_gemv_epilogue_f64(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x,
                           const double* b, double* y, double* g, int64_t rows, int64_t cols,
                           int variant, void* stream);

/* ---- a2: elementwise axpy family -------------------------------------------------------
 * out = a*x + b*y with three roundings (mul, mul, add) -- covers `Ax+b`, `x - alpha*g`,
 * `x - y`, `(1+beta)*x1 - beta*x0`, `xk += beta*dk` (solvers.py:139,145,280,283,299,316,
 * 653-654,937,961-962 ...).  1.0*x and -1.0*x are exact, so the numpy forms are reproduced
 * bit for bit.  out may alias x or y. */
int ccqp_axpby_f64(ccqp_ctx* ctx, double a, const double* x, double b, const double* y,
                   double* out, int64_t n, void* stream);
/* Two independent updates of the same length in one launch (SPG's `xk += beta*dk; gk += beta*Adk`,
 * solvers.py:961-962); neither output may be an operand of the other update. */
int ccqp_axpby2_f64(ccqp_ctx* ctx, double a1, const double* x1, double b1, const double* y1, double* out1,
                    double a2, const double* x2, double b2, const double* y2, double* out2, int64_t n,
                    void* stream);
/* out = a*x (one rounding) -- the `c * (...)` of the residual, solvers.py:138 */
int ccqp_scale_f64(ccqp_ctx* ctx, double a, const double* x, double* out, int64_t n, void* stream);
int ccqp_copy_f64(ccqp_ctx* ctx, const double* x, double* out, int64_t n, void* stream);
int ccqp_fill_f64(ccqp_ctx* ctx, double value, double* out, int64_t n, void* stream);
/* out = x * y (elementwise, one rounding). */
int ccqp_mul_f64(ccqp_ctx* ctx, const double* x, const double* y, double* out, int64_t n, void* stream);
/* out = (x * y) / s   -- `psi_y * Ap / p.dot(Ap)` solvers.py:1134 */
int ccqp_muldiv_f64(ccqp_ctx* ctx, const double* x, const double* y, double s, double* out,
                    int64_t n, void* stream);
/* out = x - y * z      -- `psi_y - beta * p` with VECTOR beta, solvers.py:1135 */
int ccqp_submul_f64(ccqp_ctx* ctx, const double* x, const double* y, const double* z, double* out,
                    int64_t n, void* stream);

/* ---- a3: dot / norm reductions ------------------------------------------------------------
 * slab[slot0 + j] = xs[j] . ys[j]  for j < k <= CCQP_MAX_DOTS, ONE launch for all k (with several blocks per
 * dot, the block that finishes last adds the partial sums in index order).
 * xs / ys are HOST arrays of device pointers.  Replaces `.dot` / `np.dot` / the squared
 * `np.linalg.norm` at solvers.py:264-265,285-286,296-301,510,635,655-656,923-924,944-946,
 * 1077,1098,1106,1129,1146-1147,1163,1171-1172. */
int ccqp_dots_f64(ccqp_ctx* ctx, int k, const double* const* xs, const double* const* ys,
                  int64_t n, int slot0, void* stream);

/* ---- a5..a8: projection operators -------------------------------------------------------- */
enum {
  CCQP_OP_IDENTITY = 0, /* solution_spaces.py:111-125 */
  CCQP_OP_LOWER = 1,    /* :186-201  lb*m + x*(1-m), m = x<lb */
  CCQP_OP_UPPER = 2,    /* :262-277 */
  CCQP_OP_BOX = 3,      /* :349-366  lb*lo + ub*hi + x*(1-hi)*(1-lo) */
  CCQP_OP_SPHERE = 4,   /* :417-435  (radius*x)/|x| if |x| > radius */
  CCQP_OP_CONE = 5      /* :470-492  reproduced as is, incl. the whole-vector norm (:484) */
};
#define CCQP_SEG_LB_INT 1 /* lower bound was integer-typed in Python: lb*mask is integer math */
#define CCQP_SEG_UB_INT 2
#define CCQP_SEG_P0_INT 4 /* Sphere radius / Cone aspect ratio was a Python int */

/* One contiguous block x[offset : offset+dim) and the convex set it is projected on.
 * DisjointProjOp (solution_spaces.py:495-560) = a list of these with running offsets. */
typedef struct ccqp_seg {
  int32_t kind;
  int32_t flags;
  int64_t offset;
  int64_t dim;
  const double* lb; /* device, dim doubles, or NULL -> lb_s broadcast */
  const double* ub;
  double lb_s;      /* scalar lower bound | sphere radius | cone aspect ratio */
  double ub_s;
} ccqp_seg;

/* out[seg] = P(a*x + b*y)[seg] for an ELEMENTWISE kind (Identity/Lower/Upper/Box); y may be NULL
 * (then out = P(x), a and b ignored).  Fuses the step `P(x - alpha*g)` (solvers.py:145,283,310,
 * 638,778,937,1148,1167) with the projection; literal expression order => bit-exact. */
int ccqp_proj_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, double a, const double* x,
                              double b, const double* y, double* out, void* stream);
/* out[seg] = P(a*x + b*y)[seg] - z[seg]: SPG's direction `P(xk - alpha*gk) - xk` (solvers.py:937) without the
 * intermediate vector; the subtraction is a separate rounding, as in numpy. */
int ccqp_proj_elementwise_sub_f64(ccqp_ctx* ctx, const ccqp_seg* seg, double a, const double* x, double b,
                                  const double* y, const double* z, double* out, void* stream);
/* Sphere / Cone, second phase: slab[norm2_slot] must already hold |x[seg]|^2 (ccqp_dots_f64 on the
 * block).  Computes r = sqrt(slab[norm2_slot]) on the device and applies the op's branches. */
int ccqp_proj_normscale_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x, double* out,
                            int norm2_slot, void* stream);
/* Many small blocks in ONE launch, one thread per block (dim <= CCQP_SMALL_SEG_MAX), any kind.
 * `table` is a DEVICE array of nseg descriptors.  Norms of blocks with dim <= 15 use the
 * sequential chain s = fma(x_i, x_i, s) that OpenBLAS ddot performs for such lengths, so small
 * Sphere/Cone blocks are bit-exact against numpy.  mode 0: out = P(x); mode 1: out = normal_vector(x)
 * (solution_spaces.py:146-160,222-236,306-322,389-403,92-98). */
#define CCQP_SMALL_SEG_MAX 64
int ccqp_proj_small_segments_f64(ccqp_ctx* ctx, const ccqp_seg* table, int64_t nseg, const double* x,
                                 double* out, int mode, void* stream);

/* ---- (e1) row-sharded multi-GPU matvec, one process per GPU, RCCL over xGMI (ccqp_comm.hip) -------------------
 * The exchange step of SURVEY.md 8(e)1 behind the C ABI (the Python shells default to torch.distributed's RCCL
 * binding; this is the same step without torch, and their opt-in VectorExchange(direct=True)).  RCCL is bound at run
 * time: without a loadable librccl the calls fail with CCQP_ERR_UNSUPPORTED.  Bootstrap: rank 0 obtains a unique id
 * and ships its 128 bytes to the other ranks by any means (MPI, a file, torch.distributed); every rank then
 * creates its communicator on its context's device.  All calls are asynchronous on `stream`. */
#define CCQP_COMM_ID_BYTES 128
typedef struct ccqp_comm ccqp_comm;
int ccqp_comm_rccl_version(void);                      /* 0: no RCCL available */
int ccqp_comm_unique_id(unsigned char* id128);
int ccqp_comm_create(ccqp_ctx* ctx, int nranks, int rank, const unsigned char* id128, ccqp_comm** out);
int ccqp_comm_destroy(ccqp_comm* comm);
int ccqp_comm_info(const ccqp_comm* comm, int* nranks, int* rank);
/* y holds `count` doubles per rank, rank r's at y[r*count ...): in-place all-gather / in-place sum. */
int ccqp_comm_allgather_f64(ccqp_comm* comm, double* y, int64_t count, void* stream);
int ccqp_comm_allreduce_sum_f64(ccqp_comm* comm, double* buf, int64_t count, void* stream);
/* One row-sharded `A.dot(v)` (solvers.py:133,...): y[rank*per + i] = A_local[i,:] . x for i < rows_local
 * (per = ceil(n / nranks) >= rows_local), then ONE exchange on the same stream -- mode 0: in-place all-gather of `per`
 * doubles per rank; mode 1: the other slabs zeroed + all-reduce (adds exact zeros).  y must hold per * nranks doubles;
 * its first n entries are the assembled vector.  Each row sum comes from one wave in an order that depends on the row
 * only, so the result is bitwise the single-GPU ccqp_gemv_f64 for every rank count. */
int ccqp_shard_gemv_allgather_f64(ccqp_ctx* ctx, ccqp_comm* comm, const double* A_local, int64_t lda, const double* x,
                                  double* y, int64_t rows_local, int64_t per, int64_t cols, int variant, int mode,
                                  void* stream);

/* The k products of an APGD iteration (k = 2: solvers.py:290 and :275 of the next iteration, y_{k+1} is known as soon as
 * x_{k+1} is) or across the end of an MPRGP iteration (k <= 3: :1129 | :1163, :1086, :1100) on a row-sharded matrix with
 * ONE exchange: the local k-vector pass (ccqp_gemvk_f64) writes its slabs straight into this rank's part of `work`
 * (k * per * nranks doubles of device memory owned by the caller; rank r's slab of product j at work[(r k + j) per ...)),
 * one in-place all-gather of k * per doubles per rank, one unpack launch that scatters the slabs into the replicated
 * vectors y[j] (cols entries each) and forms g[j] = y[j] + b where g[j] != NULL -- the same single add as the fused
 * epilogue, so every product is bitwise its own single-GPU ccqp_gemv_epilogue_f64.  Per iteration: 1 matvec launch +
 * 1 collective + 1 unpack launch.  ccqp_unpack_gathered_f64 is that last launch by itself (for callers that bring their
 * own collective, e.g. torch.distributed). */
int ccqp_shard_gemvk_allgather_f64(ccqp_ctx* ctx, ccqp_comm* comm, const double* A_local, int64_t lda, int k,
                                   const double* const* x, const double* b, double* const* y, double* const* g,
                                   int64_t rows_local, int64_t per, int64_t cols, int variant, double* work, void* stream);
int ccqp_shard_gemv2_allgather_f64(ccqp_ctx* ctx, ccqp_comm* comm, const double* A_local, int64_t lda, const double* x1,
                                   const double* x2, const double* b, double* y1, double* g1, double* y2, double* g2,
                                   int64_t rows_local, int64_t per, int64_t cols, int variant, double* work, void* stream);
int ccqp_unpack_gathered_f64(ccqp_ctx* ctx, const double* gathered, int k, int64_t per, int64_t n, double* const* y,
                             const double* b, double* const* g, void* stream);

/* ---- a5-a8 for ANY feasible set in <= 2 launches: the segmented engine (ccqp_segs.hip) -----------------------
 * Replaces the Python loop of DisjointProjOp.__call__ / normal_vector (solution_spaces.py:540-560, 512-525) and
 * the dot + scale pairs of Sphere / Cone (:417-435, 470-492) for tables with blocks of EVERY size; the residual
 * form replaces `np.linalg.norm(c*(x - P(x - gd*g)))` (solvers.py:137-139,321-323,629-631,647-648,1070-1072).
 * A set is a DEVICE table of ccqp_seg plus a work list (ccqp_segs_plan, host side): one workgroup per item --
 *   type 0: `chunk` consecutive table entries starting at `seg`, all of dim <= CCQP_SMALL_SEG_MAX, one thread each;
 *   type 1: chunk number `chunk` (stage-1 geometry of the reductions: 2048 elements) of the large block `seg`,
 *           whose chunk sums live at workspace[pbase ...].
 * Per element the arithmetic, and per block the order of every sum, are those of ccqp_proj_small_segments_f64,
 * ccqp_dots_f64 + ccqp_proj_normscale_f64 and ccqp_proj_elementwise_f64: the same bits. */
typedef struct ccqp_segwork {
  int32_t type;
  int32_t seg;
  int32_t chunk;
  int32_t pbase;
} ccqp_segwork;
/* Host-side planning (no GPU work): fills `work` (all items, table order) and `normwork` (the type-1 items of Sphere
 * / Cone blocks) up to their capacities and ALWAYS returns the needed counts; call once with NULL lists to size
 * them.  `segs` is the HOST copy of the table. */
int ccqp_segs_plan(const ccqp_seg* segs, int64_t nseg, ccqp_segwork* work, int64_t work_cap, int64_t* nwork,
                   ccqp_segwork* normwork, int64_t norm_cap, int64_t* nnorm, int64_t* npartials);
/* doubles of device workspace for a plan; must be ZERO when first used (arrival counter), is left so. */
int64_t ccqp_segs_workspace_doubles(int64_t npartials, int64_t nwork);
/* out = P(a*x + b*y) - z over the whole table; y == NULL: P(x); z == NULL: no subtraction (SPG's direction
 * solvers.py:937 otherwise).  x, y must not alias out. */
int ccqp_segs_project_f64(ccqp_ctx* ctx, const ccqp_seg* table, const ccqp_segwork* work, int64_t nwork,
                          const ccqp_segwork* normwork, int64_t nnorm, int64_t npartials, double* workspace,
                          double a, const double* x, double b, const double* y, const double* z, double* out,
                          void* stream);
/* slab[slot] = sum_i (c*(x_i - P(x - gd*g)_i))^2: per work item a block sum (chunk: thread t takes elements
 * t, t+256, ...; small blocks: one fma chain per block), the items added in index order by the last block. */
int ccqp_segs_residual_sq_f64(ccqp_ctx* ctx, const ccqp_seg* table, const ccqp_segwork* work, int64_t nwork,
                              const ccqp_segwork* normwork, int64_t nnorm, int64_t npartials, double* workspace,
                              const double* x, const double* g, double c, double gd, int slot, void* stream);
/* out = normal_vector(x) over the whole table (Identity / Lower / Upper / Box / Sphere blocks; a Cone block has no
 * normal in the reference, solution_spaces.py:465 -- the caller raises before coming here). 3 launches. */
int ccqp_segs_normal_f64(ccqp_ctx* ctx, const ccqp_seg* table, const ccqp_segwork* work, int64_t nwork,
                         const ccqp_segwork* normwork, int64_t nnorm, int64_t npartials, double* workspace,
                         const double* x, double* out, void* stream);
/* MPRGP's feasible-step search (solvers.py:1112-1118) on any set, K <= CCQP_MAX_STEPS candidates in 2 launches:
 * slab[slot0 + k] = #{i : not isclose(yf_i, P(yf)_i)}, yf = x - (alpha0 * 0.5^k) * p.  The host takes the first k whose
 * count is 0 (the candidates are exactly the values the reference's halving loop visits). */
int ccqp_segs_feasible_steps_f64(ccqp_ctx* ctx, const ccqp_seg* table, const ccqp_segwork* work, int64_t nwork,
                                 const ccqp_segwork* normwork, int64_t nnorm, int64_t npartials, double* workspace,
                                 const double* x, const double* p, double alpha0, int K, int slot0, void* stream);

/* ---- a4: residual (Mazhar 2015 Eq. 25 as coded, solvers.py:138-139 etc.) --------------------
 * slab[slot] = sum_i v_i^2, v = c*(x - P(x - gd*g)) for an ELEMENTWISE kind over the block;
 * the caller takes the square root (np.linalg.norm = sqrt(v.v)). */
int ccqp_residual_sq_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x,
                                     const double* g, double c, double gd, int slot, void* stream);
/* k dot products whose operands may be DIFFERENCES of two vectors, plus (res_seg != NULL) the Eq. 25
 * residual^2 of (res_x, res_g), in ONE launch: slab[slot0 + j] = (a1[j] - a2[j]) . (b1[j] - b2[j]) with a NULL
 * a2[j] / b2[j] meaning "no subtraction"; slab[res_slot] = residual^2.  Each difference is rounded once (like
 * the axpby that would materialise it) and each pair keeps the fma chain of ccqp_dots_f64, so the results are
 * bit for bit those of the separate launches.  Used for the per-iteration probes of APGD (solvers.py:296-302,
 * 321-323).  res_seg must describe the whole vector (offset 0, dim n). */
int ccqp_fused_dots_f64(ccqp_ctx* ctx, int k, const double* const* a1, const double* const* a2,
                        const double* const* b1, const double* const* b2, int64_t n, int slot0,
                        const ccqp_seg* res_seg, const double* res_x, const double* res_g, double c, double gd,
                        int res_slot, void* stream);
/* Fused tail of a Barzilai-Borwein iteration (solvers.py:647-656; BBPGDf :787-806): ONE launch gives
 *   slab[slot0]   = the Eq. 25 residual^2 of (x_new, g_new)     (as ccqp_residual_sq_elementwise_f64)
 *   slab[slot0+1] = dx . dx,   slab[slot0+2] = dx . dg          (as ccqp_dots_f64 on the differences)
 * with dx = x_new - x_old, dg = g_new - g_old formed on the fly (each rounded once, like the axpby they
 * replace): bit for bit the results of the four separate launches.  Whole vectors only (seg->offset == 0). */
int ccqp_bb_tail_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x_new, const double* x_old,
                                 const double* g_new, const double* g_old, double c, double gd, int slot0,
                                 void* stream);

/* ---- a12: MPRGP helpers (solvers.py:1081-1182; np.isclose rtol=1e-5, atol=1e-8) -----------
 * mask_i = isclose(a_i, b_i) = (|a-b| <= 1e-8 + 1e-5*|b| and isfinite(b)) or a == b.
 * out = (invert ? 1-mask : mask) * g   (g == NULL -> out = the 0.0/1.0 mask itself). */
int ccqp_isclose_mask_mul_f64(ccqp_ctx* ctx, const double* a, const double* b, const double* g,
                              int invert, double* out, int64_t n, void* stream);
/* slab[slot] = number of i with NOT isclose(a_i, b_i)   (np.all(...) <=> 0), solvers.py:1115 */
int ccqp_count_notclose_f64(ccqp_ctx* ctx, const double* a, const double* b, int64_t n, int slot,
                            void* stream);
/* MPRGP's feasible-step bisection (solvers.py:1112-1118) evaluated for K candidate steps in ONE
 * launch: for k < K (<= 16), alpha_k = alpha0 * 0.5^k (exact, the same values the reference's
 * `alpha_f *= 0.5` loop visits), yf = x - alpha_k*p, slab[slot0+k] = #{i : not isclose(yf_i, P(yf)_i)}.
 * ELEMENTWISE kinds only.  The host takes the first k whose count is 0. */
#define CCQP_MAX_STEPS 16
int ccqp_feasible_steps_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x,
                                        const double* p, double alpha0, int K, int slot0, void* stream);
/* normal_vector of an ELEMENTWISE kind over its block: xp = P(x) (already computed),
 * slab[dist2_slot] = |x - xp|^2 over the block.  If sqrt(dist2) > 1e-8 (x outside) -> zeros, else
 * +1 where isclose(xp_i, ub_i) elif -1 where isclose(xp_i, lb_i) (solution_spaces.py:306-322). */
int ccqp_normal_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* xp, double* out,
                                int dist2_slot, void* stream);
/* Sphere normal (solution_spaces.py:389-403): slab[dist2_slot] as above, slab[len2_slot] = |xp|^2. */
int ccqp_normal_sphere_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* xp, double* out,
                           int dist2_slot, int len2_slot, void* stream);
/* out = w * (g - s*nv) with s = min(0, slab[ng_slot]) (NaN propagates like np.min),
 * `(1 - delta) * (gk - np.min([0, n.dot(gk)]) * n)` solvers.py:1096-1097; w is a 0/1 vector. */
int ccqp_chopped_gradient_f64(ccqp_ctx* ctx, const double* w, const double* g, const double* nv,
                              int ng_slot, double* out, int64_t n, void* stream);

/* Fused forms of the MPRGP proportioning test for ELEMENTWISE kinds (solvers.py:1093-1098); same
 * arithmetic, order and reduction geometry as the unfused calls above (bitwise identical results):
 *  split      : xp = P(x); psi = isclose(x,xp)*g; w = 1 - isclose(x,xp); slab[dist2_slot] = |x-xp|^2
 *  normal_dot : nv = normal_vector (from xp, slab[dist2_slot]); slab[ng_slot] = nv . g
 *  proportion : beta = w*(g - min(0, slab[ng_slot])*nv); slab[slot0] = beta.beta, slab[slot0+1] = psi.psi */
int ccqp_mprgp_split_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x, const double* g,
                         double* xp, double* psi, double* w, int dist2_slot, void* stream);
int ccqp_mprgp_normal_dot_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* xp, const double* g,
                              double* nv, int dist2_slot, int ng_slot, void* stream);
int ccqp_mprgp_proportion_f64(ccqp_ctx* ctx, const double* w, const double* g, const double* nv,
                              const double* psi, int ng_slot, int64_t n, int slot0, void* stream);
/* out = isclose(x, P(x)) * g for an ELEMENTWISE kind (`delta * g`, solvers.py:1081-1082,1132-1133,1157-1158) */
int ccqp_free_gradient_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x,
                                       const double* g, double* out, void* stream);
/* out = psi - ((psi*Ap)/pAp)*p   -- solvers.py:1134-1135 in one pass (four roundings, as numpy) */
int ccqp_cg_direction_f64(ccqp_ctx* ctx, const double* psi, const double* Ap, double pAp,
                          const double* p, double* out, int64_t n, void* stream);

/* projected_gradient of an ELEMENTWISE kind (solution_spaces.py:162-184,238-260,324-347; the Box
 * condition keeps the reference's misplaced parenthesis :340-341): nv = normal_vector(x) of the block;
 * active entries get chopped = g - min(nv*g, 0)*nv, free = 0; the others free = g, chopped = 0. */
int ccqp_projected_gradient_elementwise_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x,
                                            const double* g, const double* nv, double* free_g,
                                            double* chopped_g, void* stream);

/* ---- opt-in: pipelined Barzilai-Borwein iterations (same bits, fewer host round trips) ------------------
 * A BBPGD iteration (solvers.py:636-660) whose scalars stay on the device: ccqp_pipe_bb_tail_f64 derives
 * alpha = dx.dx / (dx.dg + eps10) and the exit test sqrt(res2) < tol in its last block, keeps them in slab slots
 * 48..51 {alpha, converged, iterations, res2} (and 0..2 as ccqp_bb_tail_elementwise_f64) and mirrors them into
 * pinned host memory; ccqp_pipe_step_project_f64 (out = P(x - alpha g), alpha from the slab) and
 * ccqp_pipe_gemv_epilogue_f64 can therefore be queued for the NEXT iterations without waiting -- every pipe
 * kernel leaves at once when the slab says "converged".  ccqp_pipe_poll is a non-blocking look at the mirror:
 * out[0..3] = {iterations finished, converged (0/1), res2, alpha}.  Elementwise sets, whole vectors only. */
int ccqp_pipe_reset(ccqp_ctx* ctx, double alpha0, void* stream);
int ccqp_pipe_step_project_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x, const double* g, double* out,
                               void* stream);
int ccqp_pipe_gemv_epilogue_f64(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, const double* b,
                                double* y, double* g, int64_t rows, int64_t cols, int variant, void* stream);
int ccqp_pipe_bb_tail_f64(ccqp_ctx* ctx, const ccqp_seg* seg, const double* x_new, const double* x_old,
                          const double* g_new, const double* g_old, double c, double gd, double tol, double eps10,
                          void* stream);
/* the other matvec forms and the elementwise factors of a composed operator behind the same "converged" word */
int ccqp_pipe_spmv_csr_f64(ccqp_ctx* ctx, const int64_t* rowptr, const int32_t* col, const double* val,
                           const int64_t* plan, const int64_t* summary, const double* x, const double* b, double* y,
                           double* g, int64_t rows, int64_t cols, int64_t nnz, int width, void* stream);
int ccqp_pipe_mul_f64(ccqp_ctx* ctx, const double* x, const double* y, double* out, int64_t n, void* stream);
int ccqp_pipe_axpby_f64(ccqp_ctx* ctx, double a, const double* x, double b, const double* y, double* out, int64_t n,
                        void* stream);
int ccqp_pipe_poll(ccqp_ctx* ctx, double* out);
/* hipGraph form of the same: bracket the ccqp_pipe_* calls of THREE consecutive iterations (a ring of three x / g
 * buffers makes iteration k+3 a repeat of iteration k, argument for argument) with capture_begin / capture_end on
 * `stream`, then replay with ccqp_pipe_graph_launch -- one launch instead of nine. */
int ccqp_pipe_capture_begin(ccqp_ctx* ctx, void* stream);
int ccqp_pipe_capture_end(ccqp_ctx* ctx, void* stream, void** graph_exec);
int ccqp_pipe_graph_launch(ccqp_ctx* ctx, void* graph_exec, void* stream);
int ccqp_pipe_graph_destroy(void* graph_exec);

/* ---- opt-in: MANY equal-size dense BBPGD solves (solvers.py:560-693) in lock-step, scalars on the device ---
 * The pipelined iteration above, batched: blockIdx.y = problem, three launches per iteration for the whole batch
 * (four for the BBPGDf variant),
 * a converged problem's blocks leave at once.  `dev_probs` is a DEVICE array of `count` descriptors; `dev_batch`
 * two device words owned by the caller.  Per-problem slab (>= 52 doubles, zeroed before start): slots 0..2 as
 * ccqp_bb_tail_elementwise_f64, 48..51 = {alpha, converged, iterations finished, res2}.  Each problem's results are
 * bit for bit those of its own solve(): x = x[iterations % 3] (converged) or x[last iteration % 3] (budget). */
typedef struct ccqp_pprob {
  const double* A;   /* n x n, row-major, leading dimension lda */
  int64_t lda;
  const double* b;
  double* x[3];      /* ring; x[0] holds x0 on entry */
  double* g[3];
  double* y;
  ccqp_seg seg;      /* elementwise kind, offset 0, dim n */
  double* slab;
  double* partials;  /* ccqp_bpipe_partials_doubles(n) doubles */
  uint32_t* counter; /* one word, zero on entry */
  double c;          /* residual scale 1/(3 n gd), solvers.py:138 */
  double tol;
  double* xmin;      /* BBPGDf only (flags bit 7): best iterate so far and its gradient, both = x0 on entry */
  double* gmin;      /* (solvers.py:762-763 -- sic, gmin starts as a copy of x0); NULL otherwise */
} ccqp_pprob;
int64_t ccqp_bpipe_partials_doubles(int64_t n);
/* Name of the kernel the batched matvec of ALL lock-step forms runs as for `count` problems of size n under `flags`
 * (bit 0 = aligned operands, the rest the tuning word): "bp_gemv_pers_kernel" -- ONE persistent LDS-DMA launch whose
 * waves walk the live problems' matrices (rows through per-wave LDS rings, a wave's 4 rows about 1 MiB apart); the
 * tuned default from n = 4096 and 512 MiB of matrices on (the runs that look ahead -- ccqp_bapgd_*, ccqp_bmprgp_* and their
 * segmented forms -- already from n = 1024: set bit 16 of `flags` to ask for those) -- or "batched VALU kernel".  Same
 * bits either way. */
int ccqp_bpipe_matvec_form(ccqp_ctx* ctx, int count, int64_t n, int flags, char* name, int buflen);
/* Ring shape of the three-vector persistent launch (bp_gemv3_pers_kernel<R, S>, the look-ahead of ccqp_bapgd_* /
 * ccqp_bmprgp_*) for `count` problems per launch, as the number <R><S><waves per workgroup>: 824 from 32 problems on
 * (config 5's share: 128), 444 below (and for the single-problem ccqp_gemv3_f64); the context's option BMV3_SHAPE
 * overrides.  NOT an error code. */
int ccqp_bmv3_shape(ccqp_ctx* ctx, int count);
/* Form of the FUSED lock-step APGD rounds (ccqp_bapgd_round_f64 with CCQP_LOCKSTEP_LOOKAHEAD | _FUSED_ROUNDS in `flags`:
 * one round -- the two products of an iteration from one launch, the decision, the update -- per accepted iteration) for
 * `count` problems of size n: 1 = the persistent three-vector matvec (where ccqp_bpipe_matvec_form with bit 16 names it),
 * 2 = both products from one launch of the batched VALU kernel (batches below those sizes whose matrices fit the MALL
 * together, 256 MiB: bound by the number of their launches; the context's option SMALL_SHARING overrides), 0 = this run's
 * rounds are not fused.  NOT an error code. */
int ccqp_bapgd_fused_form(ccqp_ctx* ctx, int count, int64_t n, int flags);
/* g0 = A x0 + b, residual (exit test), A g0, alpha0 = g0.g0 / g0.A g0  (solvers.py:626-636).  `flags` bit 0
 * promises 16-byte aligned A rows and vectors (A, x, g pointers % 16 == 0 and lda even); the other bits are a
 * tuning word of the matvec that changes no result bit (0 = defaults; layout in csrc/ccqp_lockstep.h).  Bit 3 (8)
 * turns the run into PGD (solvers.py:94-170): the caller stores each problem's constant step in slab slot 48, the
 * start-up has no A g0 / alpha0 and the tail leaves the step alone (same flag for iterate and bsolve).  Bit 7 (128)
 * selects BBPGDf (solvers.py:696-843): slab slots 4..6 {smallest residual, best-so-far, fall back}, the best iterate
 * kept in xmin / gmin, the restart from it when the step length drops below 10 eps. */
int ccqp_bpipe_start_f64(ccqp_ctx* ctx, const ccqp_pprob* dev_probs, int count, int64_t n, double gd, int flags,
                         uint32_t* dev_batch, void* stream);
/* iteration >= 1: x[it%3] = P(x[(it-1)%3] - alpha g[..]); g[it%3] = A x[it%3] + b; with_tail != 0: residual, BB
 * dots, alpha, exit test (with_tail == 0 where the matvec budget ends before the residual, solvers.py:643-644). */
int ccqp_bpipe_iterate_f64(ccqp_ctx* ctx, const ccqp_pprob* dev_probs, int count, int64_t n, int64_t iteration,
                           double gd, double eps10, int flags, int with_tail, uint32_t* dev_batch, void* stream);
/* ---- lock-step BBPGD / PGD for NON-elementwise feasible sets (Sphere, Cone, DisjointUnion; ccqp_bsegs.hip) ----------
 * The same scheduler with the projection, the fused step and the Eq. 25 residual taken from the segmented engine:
 * problem p brings its block table and work lists (ccqp_segs_plan) in dev_segs[p]; its `seg` in dev_probs[p] is
 * unused.  4-6 launches per iteration for the whole batch (grids of max_nwork / max_nnorm work items per problem;
 * a problem with fewer items idles the rest).  Replaces the Python loops solvers.py:583-669 / 94-170 around
 * solution_spaces.py:540-560, 417-435, 470-492 as run by benchmarks/benchmark_random_ccqp.py:67-102.  Per problem
 * the bits of its own solve() (ccqp_segs_project_f64 + ccqp_segs_residual_sq_f64 + ccqp_dots_f64 + the host loop's
 * scalar arithmetic).  `workspace` must be zero on entry.  Polling, slab slots, flags bit 0 / bit 3 (PGD) / bit 7
 * (BBPGDf: xmin / gmin of ccqp_pprob; the restart P(xmin - gd gmin) is a segmented launch of its own) and the
 * matvec tuning word: as ccqp_bpipe_*. */
typedef struct ccqp_psegs {
  const ccqp_seg* table;
  const ccqp_segwork* work;
  const ccqp_segwork* normwork;
  double* workspace;   /* ccqp_segs_workspace_doubles(npartials, nwork) doubles */
  int64_t npartials;
  int32_t nwork, nnorm;
} ccqp_psegs;
int ccqp_bsegs_start_f64(ccqp_ctx* ctx, const ccqp_pprob* dev_probs, const ccqp_psegs* dev_segs, int count, int64_t n,
                         int64_t max_nwork, int64_t max_nnorm, double gd, int flags, uint32_t* dev_batch,
                         void* stream);
int ccqp_bsegs_iterate_f64(ccqp_ctx* ctx, const ccqp_pprob* dev_probs, const ccqp_psegs* dev_segs, int count,
                           int64_t n, int64_t iteration, int64_t max_nwork, int64_t max_nnorm, double gd,
                           double eps10, int flags, int with_tail, uint32_t* dev_batch, void* stream);
/* non-blocking: out[0..1] = {problems converged, last iteration every problem has finished (-1: none yet)} */
int ccqp_bpipe_poll(ccqp_ctx* ctx, double* out);
/* The WHOLE solve of every problem in ONE launch, one workgroup per problem (n <= 2048: every reduction is one block
 * wide): for matrices of a few hundred rows, where a lock-step iteration is bound by its launches.  Same descriptors
 * (partials / counter unused), same slab contents on return, same bits.  `last_iteration` = the iteration at which
 * the matvec budget ends (max_mv - 1, at least 1; a huge value for "no budget"). */
int ccqp_bsolve_bbpgd_f64(ccqp_ctx* ctx, const ccqp_pprob* dev_probs, int count, int64_t n, double gd, double eps10,
                          int64_t last_iteration, int flags, void* stream);

/* The same for APGD (solvers.py:220-343), whose iteration is a small state machine (one matvec of y_k, then one
 * matvec per backtracking trial of x_{k+1}).  A ROUND advances every live problem by exactly one matvec: matvec,
 * decide (all dot products + residual of :296-301,321-323 in one pass, sufficient-decrease test :302, momentum
 * :312-313, exit test, L update), apply (trial step x_{k+1} = P(y_k - t g_k)  or  y_{k+1} = (1+beta) x_{k+1} -
 * beta x_k).  Slab (>= 52 doubles, zeroed): 0..7 the host loop's eight scalars, 32..37 {L, t, theta_k, beta, r1,
 * r2}, 38..40 {phase, action, accepted iterations}, 49..51 {converged (1; 2: stopped at the budget, fused rounds),
 * matvecs so far, res2}.  A converged
 * problem's solution is X[(accepted iterations + 1) % 2].  ccqp_bpipe_poll reports {converged, round}.
 * `flags` bit 3 (8) selects the anti-relaxation variant (solvers.py:370-557; slots 41..43 {smallest residual,
 * copy pending, solution is xhat}): restart of the momentum where gk.(x_{k+1} - x_k) > 0, best iterate kept in xhat;
 * its solution is xhat if slot 43 is set, X[...] otherwise. */
typedef struct ccqp_papgd {
  const double* A;
  int64_t lda;
  const double* b;
  double* X[2];      /* x_k / x_{k+1} by parity of the accepted-iteration count; X[0] = x0 on entry */
  double* Y[2];      /* y_k / y_{k+1};                                            Y[0] = x0 on entry */
  double* gk;        /* A y_k + b */
  double* g1;        /* A x_{k+1} + b */
  double* Ayk;
  double* Ax1;
  double* xhat;      /* anti-relaxation only (flags bit 3): best iterate so far, ones on entry (solvers.py:432) */
  ccqp_seg seg;      /* elementwise kind, offset 0, dim n */
  double* slab;
  double* partials;  /* 8 * ccqp_bpipe_partials_doubles(n) / 3 doubles */
  uint32_t* counter; /* one word, zero on entry */
  double c;
  double tol;
  double* ga;        /* look-ahead (flags bit 17 of start / round, CCQP_LOCKSTEP_LOOKAHEAD): A y_{k+1} + b and A y_{k+1}, */
  double* Aya;       /* computed in the pass that computes A x_{k+1}.  Read ONLY when the flag is set -- then both must be
                        valid n-vectors for every problem (the argument check refuses NULL); ignored otherwise */
} ccqp_papgd;
/* flags bit 17 of ccqp_bapgd_start_f64 / ccqp_bapgd_round_f64 / ccqp_bsegs_apgd_round_f64 / ccqp_bsolve_apgd_f64: the
 * caller has given every problem the two look-ahead buffers and asks the run to share passes where the persistent
 * batched matvec is its form.  The same value must be passed to every call of a run. */
#define CCQP_LOCKSTEP_LOOKAHEAD (1 << 17)
/* flags bit 18 of ccqp_bapgd_start_f64 / ccqp_bapgd_round_f64 (elementwise sets, with bit 17): FUSED ROUNDS -- a round
 * that accepts iteration k also takes A.y_{k+1} + b over and forms the next trial point, so an accepted iteration is ONE
 * round (3 launches) instead of two (10).  A round may then count two matvecs, so the device enforces the budget itself:
 * the caller writes max_mv - 1 (0 = none) into slab slot 48 of every problem before the start; a problem whose next
 * matvec would pass it stops with slot 49 = 2 (left to the host-driven loop, as the unfused run's round cap leaves it) and
 * counts as finished in ccqp_bpipe_poll.  The same value for every call of a run.  Same bits, same counts. */
#define CCQP_LOCKSTEP_FUSED_ROUNDS (1 << 18)
int ccqp_bapgd_start_f64(ccqp_ctx* ctx, const ccqp_papgd* dev_probs, int count, int64_t n, int flags,
                         uint32_t* dev_batch, void* stream);
int ccqp_bapgd_round_f64(ccqp_ctx* ctx, const ccqp_papgd* dev_probs, int count, int64_t n, int64_t round, double gd,
                         int flags, uint32_t* dev_batch, void* stream);
/* The same round for NON-elementwise sets (ccqp_bsegs.hip): matvec, [norm,] residual, decide, [norm,] trial step,
 * apply -- the residual (solvers.py:321-323) and the trial step (:283,:310) go through the segmented engine, everything
 * else is the state machine above; start-up: ccqp_bapgd_start_f64 (it involves no projection).  flags bit 3: the
 * anti-relaxation variant.  Per problem the bits of its own solve(). */
int ccqp_bsegs_apgd_round_f64(ccqp_ctx* ctx, const ccqp_papgd* dev_probs, const ccqp_psegs* dev_segs, int count,
                              int64_t n, int64_t round, int64_t max_nwork, int64_t max_nnorm, double gd, int flags,
                              uint32_t* dev_batch, void* stream);
/* n <= 2048: the whole solve of every problem in ONE launch, one workgroup per problem (as ccqp_bsolve_bbpgd_f64).
 * No round starts once a problem's matvec count has reached `mv_limit` (max_mv - 1; a huge value for "no budget"):
 * such a problem is left unconverged for the host-driven loop. */
int ccqp_bsolve_apgd_f64(ccqp_ctx* ctx, const ccqp_papgd* dev_probs, int count, int64_t n, double gd, double mv_limit,
                         int flags, void* stream);

/* ONE problem of 128 <= n (even, aligned: flags bit 0), the whole APGD solve (flags bit 3: the anti-relaxation variant) in
 * ONE cooperative launch on the whole chip: one 4-wave workgroup per CU, the rounds of the fused lock-step form (the
 * two-vector persistent matvec of the trial point and its momentum point, the decision, the update) separated by grid
 * barriers instead of kernel boundaries -- the device-resident form of a single solve of 2048 < n <= 16384, where launch
 * gaps / host turn-around are a third to a half of a solve.  The problem carries ga / Aya; slab slot 48 = max_mv - 1
 * (0: none); dev_batch: two words (barrier counter, give-up flag).  On return of the stream: slab as after
 * ccqp_bapgd_round_f64 (slot 49: 1 converged | 2 stopped at the budget; 0 with dev_batch[1] != 0: the barrier gave up
 * after seconds -- solve the problem another way).  CCQP_ERR_UNSUPPORTED: no cooperative launch of this shape on this
 * device.  Replaces the loop solvers.py:271-334 (:446-523).  Per problem the bits and counts of its own solve(). */
/* Staging of ONE lock-step APGD problem in one launch: *host_desc (a HOST copy of the descriptor, its pointers to device
 * memory) is written to dev_desc; the problem's slab is zeroed with slot 48 = limit (max_mv - 1, 0: none); X[0] = Y[0] =
 * x0 (NULL: zeros, solvers.py:253-258); xhat = ones where given (:432); words[0..2] (the problem's arrival counter, the
 * two batch words) = 0.  Then ccqp_bapgd_start_f64 / ccqp_bapgd_round_f64 (or ccqp_bsolve_apgd_grid_f64) on dev_desc. */
int ccqp_bapgd_stage_one_f64(ccqp_ctx* ctx, const ccqp_papgd* host_desc, ccqp_papgd* dev_desc, const double* x0, int64_t n,
                             double limit, uint32_t* words, void* stream);
/* The same for ONE problem of the ccqp_bpipe_* drivers (PGD / BBPGD / BBPGDf as a batch of one): *host_desc to dev_desc,
 * the slab zeroed with slot 48 = step (PGD's constant step, solvers.py:71-194; 0 otherwise), x[0] = x0 (NULL: zeros) and,
 * where given, xmin = gmin = x0 (BBPGDf, :762-763), words[0..2] = 0.  Then ccqp_bpipe_start_f64 / ccqp_bpipe_iterate_f64. */
int ccqp_bpipe_stage_one_f64(ccqp_ctx* ctx, const ccqp_pprob* host_desc, ccqp_pprob* dev_desc, const double* x0, int64_t n,
                             double step, uint32_t* words, void* stream);
int ccqp_bsolve_apgd_grid_f64(ccqp_ctx* ctx, const ccqp_papgd* dev_probs, int64_t n, double gd, int flags,
                              int64_t max_rounds, uint32_t* dev_batch, void* stream);

/* And for MPRGP (solvers.py:1026-1200, README's "MPRGP-BB", BASELINE config 5), elementwise sets with array
 * bounds.  A ROUND = at most one matvec per live problem (A x_k + b | A p | A x_{k+1} + b | A dx, by the problem's
 * state), three decision stages (proportioning test :1090-1099; alpha_cg :1106 and the feasible-step search
 * :1112-1118; the expansion half-step's BB dots :1146-1147; the tail :1184-1192) and the elementwise update they
 * ask for.  Slab (>= 56 doubles, zeroed): 0..2 the tail's three scalars, 8 n.g, 10 |x - P(x)|^2, 16..23 the first
 * eight feasibility counts of the last pass, 32..37 {alpha_bb, alpha_cg, p.Ap, alpha_f, half-step alpha, candidates
 * tried}, 38..40 {state, action, outer iterations}, 49 {0 running | 1 converged | 2 the feasibility search cannot terminate | 3 a proportioning step
 * (:1165-1182, unreachable on finite data) is due: left to the host-driven loop}, 50 the reference's
 * mv_count, 51 res2, 52 matvecs executed.  A converged problem's solution is X[(outer iterations + 1) % 2]. */
typedef struct ccqp_pmprgp {
  const double* A;
  int64_t lda;
  const double* b;
  double* X[2];      /* x_k / x_{k+1} by parity of the outer-iteration count; x0 staged in X[1] on entry */
  double* G[2];      /* g_k / g_{k+1} */
  double* p;
  double* Ap;
  double* psi;
  double* w;
  double* nv;
  double* txk;
  double* dx;
  double* Adx;
  double* ytmp;      /* A x of the "+ b" matvecs (unused by the loop, written by the matvec kernel) */
  ccqp_seg seg;      /* elementwise kind, offset 0, dim n */
  double* slab;
  double* partials;  /* ccqp_bmprgp_partial_rows() * ccqp_bpipe_partials_doubles(n) / 3 doubles */
  uint32_t* counter; /* one word, zero on entry */
  double c;
  double tol;
} ccqp_pmprgp;
int ccqp_bmprgp_partial_rows(void);
/* MPRGP in lock-step on feasible sets that are NOT elementwise (Sphere, DisjointUnion with blocks of any size; a Cone
 * block has no normal vector, solution_spaces.py:465): the state machine, slab and descriptors of ccqp_bmprgp_* with
 * the projections, the normal vector (solution_spaces.py:389-403, 512-525), the feasible-step search (solvers.py:
 * 1112-1118, CCQP_MAX_STEPS candidates per pass) and the Eq. 25 residual taken from the segmented engine -- per problem
 * a struct ccqp_psegs as for ccqp_bsegs_*.  A round is ten launches for the whole batch.  start: x0 staged in `dx`
 * (NOT in X[1]), slabs, counters and segment workspaces zero.  Polling, flags, status words: as ccqp_bmprgp_*.  Per
 * problem the bits of its own CCQPSolverMPRGP.solve(). */
int ccqp_bsegs_mprgp_start_f64(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, const struct ccqp_psegs* dev_segs, int count,
                               int64_t n, int64_t max_nwork, int64_t max_nnorm, double gd, int flags,
                               uint32_t* dev_batch, void* stream);
int ccqp_bsegs_mprgp_round_f64(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, const struct ccqp_psegs* dev_segs, int count,
                               int64_t n, int64_t round, int64_t max_nwork, int64_t max_nnorm, double gd, double eps10,
                               int flags, uint32_t* dev_batch, void* stream);
/* (ccqp_bmprgp_*: a run whose batched matvec is the persistent form -- ccqp_bpipe_matvec_form, default tuning word --
 * LOOKS AHEAD: the A.dx launch of an iteration (solvers.py:1129 / :1163) also computes A.p_{k+1} (:1100 of the next
 * iteration) and, after a CG step, A.x_{k+1} + b (:1086), three right-hand sides in one pass over each matrix; the
 * start's A.x_0 + b (:1065) serves the first :1086 and A.g_0 (:1077) carries A.p_0.  Slab words 53 / 54 hold what is
 * computed ahead.  Every product has the bits of its own pass and the two counters advance where the reference computes
 * the product, so results and counts are unchanged; ccqp_ctx_set_option(ctx, CCQP_OPT_MPRGP_LOOKAHEAD, 1) turns it off.  The
 * segmented form ccqp_bsegs_mprgp_* looks ahead in the same way.) */
/* Staging of ONE lock-step MPRGP problem in one launch (as ccqp_bapgd_stage_one_f64): *host_desc to dev_desc, the slab
 * zeroed with slot 55 = limit (fused rounds: max_mv - 1, 0: none), X[1] = x0 (NULL: zeros), words[0..2] (arrival counter,
 * the two batch words) = 0.  Then ccqp_bmprgp_start_f64 /
 * ccqp_bmprgp_round_f64 on dev_desc. */
int ccqp_bmprgp_stage_one_f64(ccqp_ctx* ctx, const ccqp_pmprgp* host_desc, ccqp_pmprgp* dev_desc, const double* x0, int64_t n,
                              double limit, uint32_t* words, void* stream);
/* FUSED ROUNDS of ccqp_bmprgp_start_f64 / ccqp_bmprgp_round_f64 (`flags` bit 18, CCQP_LOCKSTEP_FUSED_ROUNDS: the caller has
 * put max_mv - 1 -- 0: no budget -- into slab slot 55 of every problem): for a batch of one-chunk problems (n <= 2048)
 * below the persistent matvec's sizes (context option SMALL_SHARING = 1: off) a round is TWO launches -- one batched VALU
 * launch with the products of every live problem (the three of an A.dx round as grid slices: the run looks ahead), one
 * stage launch that passes THROUGH the rounds whose product is already there (up to four) -- and a CG step one round
 * instead of three.  The device enforces the budget: a problem whose next round would pass max_mv - 1 ends with slab
 * slot 49 = 4 and counts as finished in ccqp_bpipe_poll (the caller runs it some other way).  Same bits, same counts.
 * ccqp_bmprgp_fused_form: 2 if a run of `count` problems of size n under `flags` takes this form, else 0 (NOT an error
 * code). */
int ccqp_bmprgp_fused_form(ccqp_ctx* ctx, int count, int64_t n, int flags);
int ccqp_bmprgp_start_f64(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, int count, int64_t n, double gd, int flags,
                          uint32_t* dev_batch, void* stream);
int ccqp_bmprgp_round_f64(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, int count, int64_t n, int64_t round, double gd,
                          double eps10, int flags, uint32_t* dev_batch, void* stream);
/* n <= 2048: the whole solve of every problem in ONE launch, one workgroup per problem (as ccqp_bsolve_apgd_f64). */
int ccqp_bsolve_mprgp_f64(ccqp_ctx* ctx, const ccqp_pmprgp* dev_probs, int count, int64_t n, double gd, double eps10,
                          double mv_limit, int flags, void* stream);

/* ---- lock-step SPG-QP (solvers.py:878-975) with the HOST in the loop once per iteration of the whole batch ----
 * SPG's step needs what the device cannot reproduce bit for bit: a draw of numpy's legacy generator and libm's
 * pow() (`beta ** 2`, `betak ** 2`, :958,:963 -- pow(x, 2) differs from x*x in ~1 of 1200 arguments).  So the
 * vector work of ALL problems runs in lock-step on the device (4 launches per iteration for the whole batch: control
 * words, update + direction, matvec, three dot products per problem published to pinned host memory) and the scalar
 * part of the iteration (:949-966) runs on the host inside ccqp_bspg_run_f64, in C, problem after problem, on the
 * same libm as the Python loop.  The draws are taken from each problem's OWN generator at the moment the loop would
 * take them: `sources[p]` points to a record with the layout of numpy's public `bitgen_t` (numpy/random/bitgen.h;
 * from Python: the "BitGenerator" capsule of a np.random.RandomState), whose next_double is the double
 * `RandomState.uniform` consumes (betak = sigma1 + (high - sigma1) * u, as numpy's random_uniform) -- one call per
 * accepted iteration, so every generator is left exactly where the problem's own solve() would leave it.  With one
 * shared stream the draws of problem p would start where problem p-1 stopped, which is not known while they advance
 * together: one source per problem.
 * Per problem the bits of its own CCQPSolverSPG.solve() with that generator; replaces the Python loop as run by
 * benchmarks/benchmark_random_ccqp.py:67-102 over its SPG cells. */
typedef struct ccqp_pspg {
  const double* A;   /* n x n, row-major, leading dimension lda */
  int64_t lda;
  const double* b;
  double* x;         /* x_k; x0 on entry */
  double* g;         /* g_k = A x_k + b */
  double* d;         /* d_k = P(x_k - alpha g_k) - x_k */
  double* Ad;        /* A d_k (A g_0 during the start-up) */
  ccqp_seg seg;      /* elementwise kind, offset 0, dim n (unused when dev_segs is given) */
  double* partials;  /* ccqp_bpipe_partials_doubles(n) doubles */
  uint32_t* counter; /* one word, zero on entry */
} ccqp_pspg;
/* host-side state of one problem; the caller fills the five parameters, m, and zeroes the rest */
typedef struct ccqp_spg_host {
  double tol, max_mv, tau, sigma1, sigma2; /* solvers.py:856-873; max_mv may be INFINITY */
  double alpha, betak, fk, dd;             /* step length, last draw, model value (sic :923,:963), last d.d */
  int64_t mv, used, reserved_;             /* mv_count; draws consumed */
  int32_t m, qlen;                         /* capacity / fill of the problem's queue of model values (:931-932) */
  int32_t status;  /* 0 running | 1 converged (:949) | 2 matvec budget (:940) | 3 hand back to the host-driven loop
                      (the draw would raise OverflowError, or the budget ends before the first d.d: UnboundLocalError) */
  int32_t pending; /* x += betak d; g += betak Ad (:961-962) still to be applied by the next round */
  int32_t have_dd, pad_;
} ccqp_spg_host;
/* a source of uniform [0,1) doubles: the layout of numpy's bitgen_t */
typedef struct ccqp_draw_source {
  void* state;
  uint64_t (*next_uint64)(void* state);
  uint32_t (*next_uint32)(void* state);
  double (*next_double)(void* state);
  uint64_t (*next_raw)(void* state);
} ccqp_draw_source;
/* The scalar part of one iteration for the problems with live[p] != 0 (host only, no GPU work; ccqp_bspg_run_f64
 * calls the same code): dots[3p..3p+2] = {d.d, d.Ad, d.g}; the draw comes from sources[p], or -- sources == NULL --
 * is u[p]; queue: count rows of `qstride` >= m doubles, oldest value first. */
int ccqp_bspg_host_step(ccqp_spg_host* host, double* queue, int64_t qstride, const double* dots,
                        const ccqp_draw_source* const* sources, const double* u, const unsigned char* live,
                        int count);
/* Staging of ONE lock-step SPG problem in one launch (as ccqp_bapgd_stage_one_f64): *host_desc to dev_desc, x = x0 (NULL:
 * zeros), words[0..2] (arrival counter, the two batch words) = 0, xchg[0..7] (dev_ctl | dev_out of ccqp_bspg_run_f64 for
 * one problem) = 0.  Then ccqp_bspg_run_f64 with count = 1 on the same stream. */
int ccqp_bspg_stage_one_f64(ccqp_ctx* ctx, const ccqp_pspg* host_desc, ccqp_pspg* dev_desc, const double* x0, int64_t n,
                            uint32_t* words, double* xchg, void* stream);
/* Runs rounds until every problem has left status 0 (*more = 0) or `max_rounds` rounds have been run (*more = 1:
 * call again with start = 0; lets the caller stay interruptible).  start != 0 first does the start-up (:921-932: g0,
 * A g0, fk = g0.x0 (sic), alpha = g0.g0 / g0.A g0, mv = 2).
 * dev_ctl / dev_out: 4 * count doubles of device memory each; dev_batch: two device words;
 * dev_segs != NULL: the feasible sets as plans of the segmented engine (Sphere / Cone / DisjointUnion; max_nwork,
 * max_nnorm as ccqp_bsegs_*), two more launches per round.  `flags`: bit 0 + the matvec's tuning word, as
 * ccqp_bpipe_*; from 32 problems on the batch advances as two alternating halves (the device runs one half's round
 * while the host does the other half's scalars) unless bit 3 (8) is set -- same results either way.
 * BLOCKING (it waits for every round); the results are x (dev_probs[p].x) and host[p]. */
int ccqp_bspg_run_f64(ccqp_ctx* ctx, const ccqp_pspg* dev_probs, const ccqp_psegs* dev_segs, ccqp_spg_host* host,
                      double* queue, int64_t qstride, const ccqp_draw_source* const* sources, int count, int64_t n,
                      int64_t max_nwork, int64_t max_nnorm, int flags, int start, int64_t max_rounds, double* dev_ctl,
                      double* dev_out, uint32_t* dev_batch, int* more, void* stream);

/* ---- opt-in, NON-parity: symmetric matvec reading only the upper triangle -----------
 * y = A.x for a fully stored SYMMETRIC matrix, 4n^2 bytes of HBM traffic instead of 8n^2; optional
 * g = y + b.  a_ij (j > i) is read once and feeds y_i and y_j, so the summation order differs from
 * ccqp_gemv_f64 (the reference's `A.dot(v)` never assumes symmetry).  n % 128 == 0; `workspace` =
 * ccqp_symv_workspace_doubles(n) doubles of device memory owned by the caller (n^2/32 + n*ceil(n/8192)). */
int64_t ccqp_symv_workspace_doubles(int64_t n);
int ccqp_symv_upper_f64(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, const double* b,
                        double* y, double* g, int64_t n, double* workspace, void* stream);

/* ---- sparse operator: CSR fp64 matvec (operator generality) ------------------------
 * The reference only calls `A.dot(v)` (solvers.py:25,133,...), so a scipy.sparse matrix is a legal A
 * there; this is that call for a CSR matrix resident in HBM: y = A.x for rows [0, rows) of a
 * rows x cols matrix (rowptr: rows+1 int64, rowptr[0] == 0; col: nnz int32; val: nnz doubles), with
 * the optional epilogue g = y + b.
 *   plan: ccqp_spmv_plan_words(rows, nnz) int64 words of device memory owned by the caller, filled ONCE
 *         per matrix by ccqp_spmv_plan_csr: which rows each workgroup streams through LDS, which windows
 *         gather x in transposed order, the list of rows that take the wave-per-row launch instead, and --
 *         when `col` is given -- for every window whose entries reuse their columns (each distinct column
 *         1.5 times or more, at most 1024 of them) the sorted distinct columns plus a 16-bit local column
 *         per entry: such "local-x" windows load each x value once into LDS and stream 10 instead of 12
 *         bytes per entry; for the other windows, whether all rows have one length 2 <= L <= 256 and, position by
 *         position, the same offset col - row (banded / circulant / regular-grid matrices): such "diagonal"
 *         windows keep the L offsets once and the matvec reads NO column index for them (8 instead of 12 bytes per
 *         entry).  The plan takes ~2.3 + 2 bytes per stored entry.
 *   summary (optional, 4 host words): {rows on the wave-per-row path, windows in transposed order,
 *         local-x windows, diagonal windows}.  Asking for it makes the plan call blocking; handing it to the
 *         matvec (which reads word 0 only) lets it skip the second launch when no row needs it (NULL = unknown:
 *         launched, the kernel reads the count).
 * Bits of y[i]: a function of row i's stored entries (in storage order) and x only -- not of the plan,
 * not of `width` (lanes that reduce one row: 0 = from the mean row length; 4, 8, 16, 32, 64 to force),
 * not of how the rows are sharded over GPUs: p_e = fl(val_e * x[col_e]); canonical lane e mod 64 adds its
 * products in order from +0.0; the 64 lanes are combined by the xor butterfly 32, 16, 8, 4, 2, 1.
 * Traffic per call: 12 B per stored entry + 16 B per row + 8 B per column (x once). */
int64_t ccqp_spmv_plan_words(int64_t rows, int64_t nnz);
int ccqp_spmv_plan_csr(ccqp_ctx* ctx, const int64_t* rowptr, const int32_t* col, int64_t rows, int64_t nnz,
                       int64_t* plan, int64_t* summary, void* stream);
/* Tuning / A-B switch on a finished plan (same bits of y either way): flags bit 0 = send every "diagonal" window back
 * to the generic paths; `summary` (may be NULL) is updated to match. */
int ccqp_spmv_plan_set(ccqp_ctx* ctx, int64_t* plan, int64_t nnz, int flags, int64_t* summary, void* stream);
int ccqp_spmv_csr_f64(ccqp_ctx* ctx, const int64_t* rowptr, const int32_t* col, const double* val,
                      const int64_t* plan, const int64_t* summary, const double* x, const double* b, double* y,
                      double* g, int64_t rows, int64_t cols, int64_t nnz, int width, void* stream);
/* Counts violated CSR invariants into scalar slot `slot` (0.0 = valid): rowptr[0] == 0, rowptr
 * non-decreasing, rowptr[rows] == nnz, 0 <= col[e] < cols.  Call once before the first matvec on
 * untrusted index arrays: the matvec itself does not bounds-check its gathers. */
int ccqp_csr_validate(ccqp_ctx* ctx, const int64_t* rowptr, const int32_t* col, int64_t rows, int64_t cols,
                      int64_t nnz, int slot, void* stream);

/* ---- batched launch form (batch-sharded mode, BASELINE config 5) --------------------------------
 * One descriptor per (problem, operation); a GROUP = `count` descriptors of the same opcode and shape,
 * executed by ONE launch (blockIdx.y/z = problem) with the bodies of the single-problem kernels, so the
 * results are bitwise those of one call per problem.  Reductions write into the scalar slab of the
 * problem's OWN context (`slab`/`partials` from ccqp_ctx_workspace).  Operand meaning per opcode:
 *   GEMV        p0=A p1=x p2=b|NULL p3=y p4=g|NULL; n=rows n2=cols lda       (ccqp_gemv_epilogue_f64)
 *   AXPBY       p0=x p1=y p2=out; s0=a s1=b; n          SCALE p0=x p1=out; s0=a      COPY p0=x p1=out
 *   FILL        p0=out; s0=value                        CG_DIRECTION p0=psi p1=Ap p2=p p3=out; s0=pAp
 *   DOTS        p[2j], p[2j+1] for j < k; n; slot0      PROJ_ELEM seg; p0=x p1=y|NULL p2=out; s0=a s1=b
 *   RESIDUAL_ELEM seg; p0=x p1=g; s0=c s1=gd; slot0     FEASIBLE_ELEM seg; p0=x p1=p; s0=alpha0; k; slot0
 *   MPRGP_SPLIT seg; p0=x p1=g p2=xp p3=psi p4=w; slot0=dist2
 *   MPRGP_NORMAL_DOT seg; p0=xp p1=g p2=nv; slot0=ng, slot1=dist2
 *   MPRGP_PROPORTION p0=w p1=g p2=nv p3=psi; n; slot0 (2 results), slot1=ng
 *   FREE_GRADIENT_ELEM seg; p0=x p1=g p2=out */
enum {
  CCQP_BOP_GEMV = 1, CCQP_BOP_AXPBY, CCQP_BOP_SCALE, CCQP_BOP_COPY, CCQP_BOP_FILL, CCQP_BOP_DOTS,
  CCQP_BOP_PROJ_ELEM, CCQP_BOP_RESIDUAL_ELEM, CCQP_BOP_FEASIBLE_ELEM, CCQP_BOP_MPRGP_SPLIT,
  CCQP_BOP_MPRGP_NORMAL_DOT, CCQP_BOP_MPRGP_PROPORTION, CCQP_BOP_FREE_GRADIENT_ELEM, CCQP_BOP_CG_DIRECTION
};
typedef struct ccqp_bop {
  const double* p[16];
  double s[4];
  int64_t n, n2, lda;
  double* slab;      /* scalar slab of the problem's context */
  double* partials;  /* reduction workspace of the problem's context */
  int32_t slot[4];
  int32_t k;
  int32_t reserved;
  ccqp_seg seg;
} ccqp_bop;
/* on != 0: the matvec groups of ccqp_batch_run_f64 calls and the batched matvecs of ccqp_bmprgp_* /
 * ccqp_bsegs_mprgp_* calls made with this context are issued on one stream per device, event-chained to
 * the caller's stream, so that matvecs of different lane groups / half-batches never overlap each other
 * (they would only split the HBM bandwidth) while their small launches still do. */
int ccqp_batch_serialize_matvecs(ccqp_ctx* ctx, int on);
/* Device addresses of a context's scalar slab and reduction workspace (to fill ccqp_bop). */
int ccqp_ctx_workspace(ccqp_ctx* ctx, double** slab, double** partials);
/* Upload all descriptors once, then launch group after group on `stream`:
 * group i = opcodes[i] applied to the next counts[i] descriptors of `ops` (host array).
 * Contract: ONE stream per context (a context's descriptor ring, scalar slab and reduction workspace are ordered by
 * that stream only).  Any number of calls may be queued without synchronising: the ring (16384 descriptors, at most
 * 4096 per call) keeps an event per quarter and the call waits on the host before it reuses a quarter whose
 * descriptors a queued copy or kernel may still read. */
int ccqp_batch_run_f64(ccqp_ctx* ctx, int ngroups, const int* opcodes, const int* counts,
                       const ccqp_bop* ops, void* stream);

/* Scalar read of many contexts at once (the lanes of a batched group): slots
 * [slot0[i], slot0[i]+count[i]) (count <= 16) of slab i are collected by one kernel and copied to the
 * host with ONE transfer; ccqp_scalars_gather_host waits for it and returns n x 16 doubles (row i =
 * lane i).  `ctx` owns the staging buffers and the event; n <= 64. */
int ccqp_scalars_gather_async(ccqp_ctx* ctx, int n, double* const* slabs, const int* slot0,
                              const int* count, void* stream);
int ccqp_scalars_gather_host(ccqp_ctx* ctx, int n, double* host_out);

/* ---- synthetic problems (replaces benchmarks/benchmark_random_ccqp.py:59-63,
 * whose Wishart generator is O(n^3)) ---------------------------------------------------------
 * Rows [row0, row0+rows) of the n x n counter-hashed symmetric matrix:
 *   A[i][j] = (2u(min*n+max) - 1)/sqrt(n) (i != j), A[i][i] = 2 ; u = (splitmix64 >> 11) * 2^-53. */
int ccqp_gen_spd_rows_f64(ccqp_ctx* ctx, double* A, int64_t lda, int64_t row0, int64_t rows,
                          int64_t n, uint64_t seed, void* stream);
/* out[i] = 1 - 2u_x(i), i < n  (the unconstrained solution x*, benchmark_random_ccqp.py:61) */
int ccqp_gen_vec_f64(ccqp_ctx* ctx, double* out, int64_t n, uint64_t seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CCQP_HIP_H */
