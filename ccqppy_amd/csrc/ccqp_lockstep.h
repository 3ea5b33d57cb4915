// Shared pieces of the lock-step batch drivers (ccqp_bpipe.hip: BBPGD and APGD; ccqp_bmprgp.hip: MPRGP): slab slots,
// the per-problem meeting point, the batch-wide roll-call with its host mirror, the batched matvec's tuning.
#pragma once
#include "ccqp_bodies.h"

#define BP_ALPHA 48
#define BP_DONE 49
#define BP_ITERS 50
#define BP_RES2 51
#define BP_GG 3                            // g0.g0 between the two start-up reductions
#define BP_RESMIN 4                        // BBPGDf: smallest residual so far, "this iterate is the best", "fall back"
#define BP_IMPROVED 5
#define BP_FALLBACK 6
#define BP_FLAG_CONST_STEP 8                // `flags` bit 3: BBPGD entry points -> PGD (constant step from the slab);
                                            // APGD entry points -> the anti-relaxation variant
#define BP_FLAG_FALLBACK 128                // `flags` bit 7 of the BBPGD entry points: BBPGDf (needs xmin / gmin)
                                            // (bits 1-2, 4-6 and 8.. belong to the matvec's tuning word)
// slab slots of the lock-step APGD state machine (ccqp_bpipe.hip, ccqp_bsegs.hip)
#define AP_L 32
#define AP_T 33
#define AP_THK 34
#define AP_BETA 35
#define AP_R1 36
#define AP_R2 37
#define AP_PHASE 38
#define AP_ACT 39
#define AP_IT 40
#define AP_RESMIN 41      // anti-relaxation (solvers.py:370-557): smallest residual so far ...
#define AP_COPYHAT 42     // ... "apply: xhat = x_{k+1}" ...
#define AP_SOLHAT 43      // ... and, at convergence, whether the solution is xhat (1) or x_{k+1} itself (0)
#define AP_FIRST 44
// look-ahead of a lock-step APGD run (ccqp_bpipe.hip): this run pairs A.x_{k+1} with A.y_{k+1} | the next phase-0 round finds
// A.y_k + b computed (in `ga` / `Aya`) | beta of :313, computed when the trial step is (it depends on theta_k only)
#define AP_LOOK 45
#define AP_AHEAD 46
#define AP_BETAS 47
// fused rounds (elementwise look-ahead runs, flags bit 18): the matvec budget the DEVICE enforces, max_mv - 1 (0: none) --
// a round that takes a product over counts two matvecs, so the host's round count no longer bounds the matvec count
#define AP_LIMIT 48
#define AP_DONE BP_DONE   // 49: 0 running | 1 converged | 2 stopped one short of the budget (left to the host-driven loop)
#define AP_MV BP_ITERS    // 50: matvecs so far (the reference's mv_count)
#define AP_RES2 BP_RES2   // 51
#define BP_MIRROR (CCQP_NUM_SLOTS + 5)     // host_slots[70] = sequence number * 65536 + problems converged (bp_report)

// the descriptor records are built on the host with numpy dtypes derived from ctypes mirrors (ccqppy_amd/_lib.py,
// tests/test_lockstep_host_cpu.py pins the same numbers)
static_assert(sizeof(ccqp_seg) == 56 && sizeof(ccqp_pprob) == 192 && sizeof(ccqp_papgd) == 208 &&
                  sizeof(ccqp_pmprgp) == 224,
              "descriptor layout changed: update ccqppy_amd/_lib.py and the tests");

#ifdef __HIPCC__
// ---- the meeting-point invariant, structurally ----------------------------------------------------------------------
// A kernel with a meeting point (bp_meet) may have its problem's slab CHANGED by the block that arrives last while
// other blocks of the same launch are still starting.  Every such kernel (and every body function it is made of)
// therefore takes ONE by-value snapshot of the steering words at its entry -- every thread, before anything else --
// followed by a workgroup barrier (bp_enter / ap_enter / mp_enter below), and works from the snapshot alone: no thread
// reads the slab again; the deciding thread only WRITES it, computing from the snapshot (each word is written at most
// once per launch, by that one thread, so the snapshot is the current value wherever it is used).  The rule "a function
// with a meeting point never reads the slab" is enforced textually by tests/test_lockstep_lint_cpu.py.  (Kernels
// without a meeting point -- matvec, apply, step -- change nothing and may read the slab wherever they like: launches
// are ordered by the stream.)  Unused words of a snapshot cost nothing: their loads are dead code.
struct bp_words {                            // BBPGD / PGD / BBPGDf (ccqp_bpipe.hip, ccqp_bsegs.hip)
  double done, alpha, iters, gg, resmin, improved, fallback;
};
__device__ __forceinline__ bp_words bp_enter(const double* __restrict__ slab) {
  bp_words z;
  z.done = slab[BP_DONE];
  z.alpha = slab[BP_ALPHA];
  z.iters = slab[BP_ITERS];
  z.gg = slab[BP_GG];
  z.resmin = slab[BP_RESMIN];
  z.improved = slab[BP_IMPROVED];
  z.fallback = slab[BP_FALLBACK];
  __syncthreads();
  return z;
}

struct ap_words {                            // APGD / anti-relaxation APGD
  double done, phase, mv, it, first, r1, r2, L, thk, resmin, look, limit;
};
__device__ __forceinline__ ap_words ap_enter(const double* __restrict__ slab) {
  ap_words z;
  z.done = slab[AP_DONE];
  z.phase = slab[AP_PHASE];
  z.mv = slab[AP_MV];
  z.it = slab[AP_IT];
  z.first = slab[AP_FIRST];
  z.r1 = slab[AP_R1];
  z.r2 = slab[AP_R2];
  z.L = slab[AP_L];
  z.thk = slab[AP_THK];
  z.resmin = slab[AP_RESMIN];
  z.look = slab[AP_LOOK];
  z.limit = slab[AP_LIMIT];
  __syncthreads();
  return z;
}

// Block sums v[0..nred) of this block -> sums over the problem's nblk blocks, in the LAST block to arrive (the
// pattern of ccqp_reduce_finish with a per-problem area and counter).  Returns true in that block only, with the
// sums valid in thread 0.  nred == 0: a plain "all blocks have passed" meeting point -- state that every block reads
// at the start of a launch may be changed by the block this returns true in.
__device__ __forceinline__ bool bp_meet(double* partials, unsigned* counter, int nblk, int nred, double* v,
                                        double* lds4, int* is_last) {
  // Every wave of this block has read the state words of the launch (phase / state / done) before anybody may change
  // them: the block that turns out to be the last changes them right after this function, and a path without a
  // reduction (nred == 0) has no barrier of its own.  (Without a barrier a late wave read the NEW state once per
  // ~10^4 decisions under a saturated memory system and took another branch: an APGD solve that never converged.)
  // The callers' entry snapshot (*_enter above) already ends in a barrier; this one keeps the meeting point safe
  // on its own.
  __syncthreads();
  if (nblk == 1) return true;
  if (threadIdx.x == 0) {
    for (int k = 0; k < nred; ++k) partials[(int64_t)k * nblk + blockIdx.x] = v[k];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *is_last = prev + 1u == (unsigned)nblk;
    if (*is_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!*is_last) return false;
  // eight sums at a time: per value the arithmetic of ccqp_block_sum (hence the bits), two workgroup barriers for the
  // eight instead of sixteen -- the decision launch of a single problem spends ~2 us less here
  for (int k0 = 0; k0 < nred; k0 += 8) {
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = 0.0;
      if (k0 + j < nred)
        for (int i = threadIdx.x; i < nblk; i += CCQP_BLOCK) a[j] += partials[(int64_t)(k0 + j) * nblk + i];
    }
    ccqp_block_sum_multi<8>(a);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (k0 + j < nred) v[k0 + j] = a[j];
  }
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// The same meeting point for ONE workgroup that walks the problem's nblk chunks itself, one call per chunk (the fused
// stage launch of ccqp_bmprgp.hip): the chunk sums go to LDS (`part`, nred x nblk doubles), the call for the LAST chunk
// adds them in index order with the code above -- the same partial sums in the same order, hence the same bits -- and
// returns true; the earlier calls return false.  No inter-workgroup traffic, no counter.
__device__ __forceinline__ bool bp_meet_seq(double* part, int bx, int nblk, int nred, double* v, double* lds4) {
  __syncthreads();
  if (nblk == 1) return true;
  if (threadIdx.x == 0)
    for (int k = 0; k < nred; ++k) part[k * nblk + bx] = v[k];
  __syncthreads();
  if (bx != nblk - 1) return false;
  for (int k = 0; k < nred; ++k) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += CCQP_BLOCK) acc += part[k * nblk + i];
    v[k] = ccqp_block_sum(acc, lds4);
  }
  return true;
}

// ---- BBPGDf (solvers.py:696-843)
// part A: the residual, the exit test, "best so far", "fall back" (thread 0 of the problem's last block)
__device__ __forceinline__ bool bbf_decide_a(double* s, const bp_words& z, double r, double tol, double eps10) {
  const double res = sqrt(r);
  s[0] = r;
  s[BP_RES2] = r;
  if (res < tol) return true;
  if (res < z.resmin) {                       // :793-796
    s[BP_RESMIN] = res;
    s[BP_IMPROVED] = 1.0;
  } else {
    s[BP_IMPROVED] = 0.0;
  }
  s[BP_FALLBACK] = z.alpha < eps10 ? 1.0 : 0.0;       // :799
  return false;
}


template <typename PROB>
__device__ __forceinline__ ccqp_segk bp_segk_of(const PROB& q) {
  ccqp_segk k;
  k.kind = q.seg.kind;
  k.flags = q.seg.flags;
  k.dim = q.seg.dim;
  k.lb = q.seg.lb;
  k.ub = q.seg.ub;
  k.lb_s = q.seg.lb_s;
  k.ub_s = q.seg.ub_s;
  return k;
}

// One thread per problem and launch calls this: the last of `count` publishes the progress of the whole batch -- ONE
// 8-byte word in the pinned host mirror, sequence number * 65536 + problems converged (count <= 65535), so the host can
// never see one without the other and nothing has to be ordered around the store.  Only atomics travel between the
// reporting threads (the arrival ticket `batch[0]`, the cumulative converged count `batch[1]`), so there is no data to
// release or acquire; the one ordering that matters -- a thread's own increment of batch[1], issued before this call, is
// performed before its ticket can be seen -- is the s_waitcnt (a write leaves vmcnt when L2 has it).  The host consumes
// nothing else on the strength of this word: results are read behind the stream (ccqp_bpipe_poll's callers).  (Round 4:
// the release / acquire fences, the second host store and its waits cost the deciding thread ~3 us per launch.)
#define BP_MIRROR_SEQ_UNIT 65536.0
__device__ __forceinline__ void bp_report(unsigned* batch, double* host, unsigned count, double seq) {
  if (count == 1u) {
    // a batch of ONE (the device-resident single solves): nobody to wait for -- no ticket, no wait for the thread's
    // earlier stores; the converged count is the word this very thread may just have incremented (same address, program
    // order).  One round trip instead of two and a drain: ~2 us of a ~15 us round at n <= 1024.
    const unsigned conv = __hip_atomic_load(batch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(host + BP_MIRROR + 1, seq * BP_MIRROR_SEQ_UNIT + (double)conv, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned prev = __hip_atomic_fetch_add(batch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (prev + 1u == count) {
    __hip_atomic_store(batch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned conv = __hip_atomic_load(batch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(host + BP_MIRROR + 1, seq * BP_MIRROR_SEQ_UNIT + (double)conv, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

#endif  // __HIPCC__

static inline int bp_vec_grid(int64_t n) {
  int64_t g = (n + CCQP_BLOCK - 1) / CCQP_BLOCK;
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return (int)g;
}

// `flags` of the entry points: bit 0 = 16-byte aligned operands; the rest is a TUNING word that never changes a
// result bit (0 = defaults): bits 1-2 matrix loads 0 default / 1 cacheable / 2 non-temporal, bits 4-6 columns
// per wave step in units of 128 (1, 2, 4; 7 / 6 = two / four rows per wave, one step; 3 / 5 = two rows per wave, two / four
// steps), bits 8.. workgroups per problem.
struct bp_mv_tuning {
  bool aligned;
  int nt, U;
  dim3 grid;
};

static bp_mv_tuning bp_mv_tune(int count, int64_t n, int flags) {
  bp_mv_tuning t;
  t.aligned = flags & 1;
  t.nt = (flags >> 1) & 3;
  t.U = (flags >> 4) & 7;
  int64_t per = flags >> 8;
  const double bytes = 8.0 * (double)n * (double)n * count;
  if (t.nt == 0) t.nt = bytes > 192.0 * 1048576.0 ? 2 : 1;  // a batch that fits the MALL is re-read from there
  const int64_t rows_per_wg = CCQP_BLOCK / 64;
  // many small workgroups: ~64k per launch measured best for n = 256..4096 (profiles/r01g_bpipe_sweep.txt)
  if (per == 0) per = (65536 + count - 1) / count < 8 ? 8 : (65536 + count - 1) / count;
  if (per > (n + rows_per_wg - 1) / rows_per_wg) per = (n + rows_per_wg - 1) / rows_per_wg;
  if (per < 1) per = 1;
  // few waves in flight (a batch of one, or few problems): more loads per lane; large matrices: two rows per wave
  // (short rows: several rows per wave amortise the per-row reduction, profiles/r01g_bpipe_sweep.txt)
  // (n >= 4096: two rows per wave with TWO steps in flight -- 4 loads per lane instead of 2: 9.90 against 10.04 ms for
  // config 5's 128 x 8192^2 launch, profiles/r03_bmv_sweep.jsonl)
  if (t.U == 0) t.U = per * count * rows_per_wg < 16384 ? 4 : (n >= 4096 ? 3 : n <= 256 ? 6 : n <= 512 ? 7 : 1);
  // several rows per wave: no more workgroups than there are tasks (a workgroup beyond that would load its problem's
  // descriptor and state word only to leave -- at n = 256, four rows per wave, that was three workgroups out of four)
  const int64_t rows_per_wave = (t.U == 7 || t.U == 3 || t.U == 5) ? 2 : t.U == 6 ? 4 : 1;
  const int64_t tasks = (n + rows_per_wave - 1) / rows_per_wave;
  if (per > (tasks + rows_per_wg - 1) / rows_per_wg) per = (tasks + rows_per_wg - 1) / rows_per_wg;
  // very short rows (n <= 128): a task is only 2-4 KiB of matrix, so let a wave take several (~8 KiB) as long as
  // the batch still brings >= 4096 workgroups -- the fixed cost of a workgroup (descriptor, state word) is paid
  // less often: +5-10 % at n = 64 / 128 (sweep of 2026-10-04, profiles/r02c_bpipe_small_n.txt)
  if ((flags >> 8) == 0) {
    const int64_t want = 8192 / (rows_per_wave * n * 8);
    if (want > 1) {
      int64_t p2 = (tasks + rows_per_wg * want - 1) / (rows_per_wg * want);
      if (p2 < 2) p2 = 2;
      if (p2 < per && p2 * count >= 4096) per = p2;
    }
  }
  t.grid = dim3((unsigned)per, count);
  return t;
}

// The matvec kernels' first template argument is a CODE: 1, 2, 4 = one row per wave and that many 128-column
// steps in flight; 8 / 9 = two / four rows per wave, one step (8: the large-matrix form of ccqp_gemv.hip); 10 / 11 = two
// rows per wave, two / four steps in flight (tuning-word values 3 / 5).
#define BP_R(C) ((C) == 8 || (C) == 10 || (C) == 11 ? 2 : (C) == 9 ? 4 : 1)
#define BP_U(C) ((C) == 10 ? 2 : (C) == 11 ? 4 : (C) >= 8 ? 1 : (C))
#define BP_TASKS(C, n) (((n) + BP_R(C) - 1) / BP_R(C))

// launches KERNEL<code, ALIGNED, NT>(args...) for the tuning `t`
// (BP_EXTRA_TARGS: further template arguments of KERNEL, redefined around a BP_MV_LAUNCH by the launcher that needs them)
#define BP_EXTRA_TARGS
#define BP_MV_LAUNCH1(KERNEL, UU, AL, NTT, ...) \
  hipLaunchKernelGGL((KERNEL<UU, AL, NTT BP_EXTRA_TARGS>), t.grid, dim3(CCQP_BLOCK), 0, s, __VA_ARGS__)
#define BP_MV_LAUNCH_U(KERNEL, AL, NTT, ...)                          \
  do {                                                                \
    if (t.U == 7) BP_MV_LAUNCH1(KERNEL, 8, AL, NTT, __VA_ARGS__);     \
    else if (t.U == 6) BP_MV_LAUNCH1(KERNEL, 9, AL, NTT, __VA_ARGS__); \
    else if (t.U == 3) BP_MV_LAUNCH1(KERNEL, 10, AL, NTT, __VA_ARGS__); \
    else if (t.U == 5) BP_MV_LAUNCH1(KERNEL, 11, AL, NTT, __VA_ARGS__); \
    else if (t.U >= 4) BP_MV_LAUNCH1(KERNEL, 4, AL, NTT, __VA_ARGS__); \
    else if (t.U == 2) BP_MV_LAUNCH1(KERNEL, 2, AL, NTT, __VA_ARGS__); \
    else BP_MV_LAUNCH1(KERNEL, 1, AL, NTT, __VA_ARGS__);              \
  } while (0)
#define BP_MV_LAUNCH(KERNEL, ...)                                          \
  do {                                                                     \
    if (ccqp_mvt_on) ccqp_mvt_mark(s, 0);                                  \
    BP_MV_LAUNCH_A(KERNEL, __VA_ARGS__);                                   \
    if (ccqp_mvt_on) ccqp_mvt_mark(s, 1);                                  \
  } while (0)
#define BP_MV_LAUNCH_A(KERNEL, ...)                                        \
  do {                                                                     \
    if (t.aligned) {                                                       \
      if (t.nt == 2) BP_MV_LAUNCH_U(KERNEL, true, true, __VA_ARGS__);      \
      else BP_MV_LAUNCH_U(KERNEL, true, false, __VA_ARGS__);               \
    } else {                                                               \
      if (t.nt == 2) BP_MV_LAUNCH_U(KERNEL, false, true, __VA_ARGS__);     \
      else BP_MV_LAUNCH_U(KERNEL, false, false, __VA_ARGS__);              \
    }                                                                      \
  } while (0)

// ---- the batched matvec as ONE persistent LDS-DMA launch (round 3): one 4-wave workgroup per CU for the whole batch;
// the waves walk the live problems' matrices one after the other (gemv_lds_multi_body, ccqp_bodies.h: rows through
// per-wave LDS rings, a wave's 4 rows about 1 MiB apart) instead of every problem having workgroups of its own -- the
// single big matvec's best shape (few, long row streams) carried over to many matrices.  `Resolve` names a problem's
// operands: bool operator()(int p, gemv_lds_desc* d) const, false = the problem does not multiply in this launch.  The
// operands are resolved once per workgroup, into LDS, before the first DMA (an ordinary load beside the DMA stream
// would make hipcc drain it).  Same bits as the VALU kernels (a function of (row, n) only).
#define BMV_PERS_MAX 512                        // problems per launch (48 B of LDS each); larger batches: several launches
template <int R, int S, class Resolve, bool NT = true>
__global__ __launch_bounds__(CCQP_BLOCK) void bp_gemv_pers_kernel(const Resolve rsv, int first, int count, int64_t n,
                                                                  int64_t rs, float stagger_per_problem) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char ring[];
  __shared__ int nlive_s;
  constexpr int RING = S * (R + 1) * 1024;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6, lane = threadIdx.x & 63;
  double* res = reinterpret_cast<double*>(ring + wpb * RING) + w * GEMV_LDS_RESCAP;
  gemv_lds_desc* D = reinterpret_cast<gemv_lds_desc*>(ring + wpb * RING + wpb * GEMV_LDS_RESCAP * sizeof(double));
  // the live problems, in order, compacted by wave 0 (ballot + prefix over 64 problems at a time)
  if (w == 0) {
    int base = 0;
    for (int p0 = 0; p0 < count; p0 += 64) {
      const int p = p0 + lane;
      gemv_lds_desc d;
      bool live = false;
      if (p < count) live = rsv(first + p, &d);
      const unsigned long long mask = __ballot(live);
      if (live) D[base + __popcll(mask & ((1ull << lane) - 1ull))] = d;
      base += __popcll(mask);
    }
    if (lane == 0) nlive_s = base;
  }
  __syncthreads();                               // (ordinary loads are done; from here on only DMA and LDS)
  const int nlive = nlive_s;
  if (nlive == 0) return;
  // staggered start (gemv_lds_kernel, ccqp_gemv.hip): the waves walk the same column positions at the same pace; the
  // delay unit follows the bytes the LIVE problems make this launch read (a round in which one problem of 64 is left
  // must not wait as the full batch would); a launch that reads less than 1 GiB starts at once -- there the delay of
  // up to 63 sleeps only lengthens the tail (single APGD solves, staggered -> not: n = 4096 1.053 -> 1.044 ms, 6144
  // 1.830 -> 1.768, 8192 2.402 -> 2.384, but 16384 -- 2 GiB -- 6.37 -> 6.61: profiles/r04_single_stagger.txt)
  float su = stagger_per_problem * (float)nlive;
  const int stagger = (stagger_per_problem <= 0.f || su < 0.25f) ? 0 : su < 1.f ? 1 : su > 32.f ? 32 : (int)(su + 0.5f);
  for (int i = (int)((((int64_t)blockIdx.x * wpb + w) * 37) & 63) * stagger; i > 0; --i) __builtin_amdgcn_s_sleep(1);
  gemv_lds_multi_body<R, S, NT>(D, nlive, n, (int64_t)blockIdx.x * wpb + w, (int64_t)gridDim.x * wpb, ring + w * RING, res, rs);
#endif
}

// The tuned default of every lock-step solver (no loads / rows-x-steps / workgroups field in the tuning word; bits 3
// and 7 are solver variants): from n = 4096 and 512 MiB of matrices on (1 GiB until round 4), the persistent form.  Measured against the VALU
// kernels: the launch alone (tools/bmv_sweep.py, profiles/r03_bmv_sweep_stride.jsonl) 128 x 8192 +1.7 %, 512 x 4096
// +3.4 %, 16 x 8192 +6.5 %, 4 x 8192 +12 %; whole solves (tools/bench_lockstep_matvec_forms.py,
// profiles/r03_lockstep_matvec_forms.jsonl) n = 4096 / 8192: APGD, MPRGP, SPG, BBPGD +2 ... +6 %.  At n = 1024 / 2048 the
// launch gains 1.5-2.7 % with every problem live, but whole solves come out even (-4 ... +5 %, solver by solver), and
// 512 x 512 loses 5 %: those stay on the VALU kernels, as do batches small enough to be re-read from the MALL.
// `min_n`: solvers that LOOK AHEAD (APGD, MPRGP: two or three products per pass over a matrix) take the persistent form
// from n = 1024 on -- there its single-product launch ties with the VALU kernels, and a shared pass replaces two or
// three of them (profiles/r03_lockstep_matvec_forms_small.jsonl).
// Round 4: the byte threshold is 512 MiB (was 1 GiB; the context's option PERS_MIN_MIB moves it -- the A/B switch of the
// tools).  Whole solves between the two, persistent against VALU (profiles/r04_lockstep_forms_512mib.jsonl,
// r04_single_pers_512mib.txt): 4 x 4096 (512 MiB) BBPGD +17 %, SPG +20 %; 6 x 4096 +11 / +15 %; 3 x 5120 +6 %; 1 x 8192
// +4 ... +7 % (PGD / BBPGD / BBPGDf single solves 2.41 -> 2.25, 1.40 -> 1.32, 1.46 -> 1.38 ms); 2 x 6144 -9 % (and one
// 6144 -8 %, one 4096 +-0: below the threshold).
static inline bool bp_pers_default(const ccqp_ctx* ctx, const bp_mv_tuning& t, int flags, int count, int64_t n,
                                   int64_t min_n = 4096) {
  // (below n = 4096 -- the runs that share passes, `min_n` -- the threshold stays where it was measured: 1 GiB)
  const int64_t mib = ccqp_opt(ctx, CCQP_OPT_PERS_MIN_MIB);
  return (flags & ~(1 | 8 | 128)) == 0 && t.aligned && n % 2 == 0 && n >= min_n &&
         8.0 * (double)n * (double)n * (double)count >= 1048576.0 * (double)(mib ? mib : n >= 4096 ? 512 : 1024);
}
// smallest n at which a lock-step run that shares passes takes the persistent form (the context's options LOOKAHEAD_MIN_N
// for APGD, MPRGP_LOOKAHEAD_MIN_N for MPRGP are the tools' A/B switches).  MPRGP shares up to three products per pass and
// gains from n = 512 on (+5 ... +6 % there, profiles/r03_lockstep_matvec_forms_tiny.jsonl); APGD shares two and loses
// 3-8 % at n = 512, so its threshold stays at 1024.
static inline int64_t bp_lookahead_min_n(const ccqp_ctx* ctx, bool mprgp = false) {
  const int64_t x = ccqp_opt(ctx, mprgp ? CCQP_OPT_MPRGP_LOOKAHEAD_MIN_N : CCQP_OPT_LOOKAHEAD_MIN_N);
  return x >= 128 ? x : (mprgp ? 512 : 1024);
}

// Runs that SHARE passes between products (lock-step APGD / MPRGP with their look-ahead): the persistent form where
// bp_pers_default says so from the run's own threshold on, and -- whatever the bytes, down to a batch of ONE problem --
// from n = 4096 on (the context's option PERS1_MIN_N moves that): one pass of the multi-vector launch replaces two or
// three single ones there (4096^2: 27 against 46 us for a pair, 31 against 71 us for a triple, profiles/r03_gemv2_small.jsonl),
// which is what makes a device-resident single solve of 2048 < n <= 16384 faster than the host-driven loop.
static inline bool bp_pers_sharing(const ccqp_ctx* ctx, const bp_mv_tuning& t, int flags, int count, int64_t n, bool mprgp) {
  if (bp_pers_default(ctx, t, flags, count, n, bp_lookahead_min_n(ctx, mprgp))) return true;
  const int64_t m = ccqp_opt(ctx, CCQP_OPT_PERS1_MIN_N) ? ccqp_opt(ctx, CCQP_OPT_PERS1_MIN_N) : 4096;
  return (flags & ~(1 | 8 | 128)) == 0 && t.aligned && n % 2 == 0 && n >= m;
}

// Cache policy of the persistent launches' MATRIX loads: non-temporal (a stream that must not evict the vectors from L2)
// unless the context's option BMV_LOADS says otherwise (1 non-temporal, 2 plain) ...
static inline bool bp_pers_cacheable(const ccqp_ctx* ctx, int count, int64_t n) {
  const int64_t o = ccqp_opt(ctx, CCQP_OPT_BMV_LOADS);
  (void)count;
  (void)n;
  return o == 2;
}

template <int R, int S, class Resolve>
static int bp_launch_gemv_pers(ccqp_ctx* ctx, const Resolve& rsv, int count, int64_t n, int wpb, int bpc, hipStream_t s) {
  const int chunk = count < BMV_PERS_MAX ? count : BMV_PERS_MAX;
  const size_t lds = (size_t)wpb * ((size_t)S * (R + 1) * 1024 + GEMV_LDS_RESCAP * sizeof(double)) +
                     (size_t)chunk * sizeof(gemv_lds_desc);
  // (the kernel also holds a few bytes of static LDS: ask for what is needed)
  const bool keep = bp_pers_cacheable(ctx, count, n);
  const int rcl = keep ? ccqp_lds_raise(ctx, reinterpret_cast<const void*>(bp_gemv_pers_kernel<R, S, Resolve, false>), lds)
                       : ccqp_lds_raise(ctx, reinterpret_cast<const void*>(bp_gemv_pers_kernel<R, S, Resolve, true>), lds);
  if (rcl) return rcl;
  // start delay unit of the waves per live problem: one unit per 4 GiB read, as the single matvec's (ccqp_gemv.hip)
  const float stagger = ccqp_opt(ctx, CCQP_OPT_BMV_STAGGER) ? 0.f : (float)(8.0 * (double)n * (double)n / 4294967296.0);
  const int64_t rs = gemv_lds_row_stride(ccqp_opt(ctx, CCQP_OPT_GEMV_ROW_STRIDE), n, n, R);
  if (ccqp_mvt_on) ccqp_mvt_mark(s, 0);
  for (int first = 0; first < count; first += BMV_PERS_MAX) {    // (more than BMV_PERS_MAX problems: one launch per group)
    const int m = count - first < BMV_PERS_MAX ? count - first : BMV_PERS_MAX;
    if (keep)
      hipLaunchKernelGGL((bp_gemv_pers_kernel<R, S, Resolve, false>), dim3((unsigned)(ctx->num_cu * bpc)), dim3(64 * wpb), lds,
                         s, rsv, first, m, n, rs, stagger);
    else
      hipLaunchKernelGGL((bp_gemv_pers_kernel<R, S, Resolve, true>), dim3((unsigned)(ctx->num_cu * bpc)), dim3(64 * wpb), lds,
                         s, rsv, first, m, n, rs, stagger);
  }
  if (ccqp_mvt_on) ccqp_mvt_mark(s, 1);
  return CCQP_OK;
}

// ---- the same persistent launch with up to THREE products per problem in one pass over its matrix (gemv_lds3_multi_body):
// `Resolve3` fills a gemv_lds3_desc, false = the problem takes no part.  At most BMV3_PERS_MAX problems per launch
// (96 B of LDS each), more in several launches.
#define BMV3_PERS_MAX 256
template <int R, int S, class Resolve3, bool NT = true>
__global__ __launch_bounds__(CCQP_BLOCK) void bp_gemv3_pers_kernel(const Resolve3 rsv, int first, int count, int64_t n,
                                                                   int64_t rs, float stagger_per_problem) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char ring[];
  __shared__ int nlive_s;
  constexpr int RING = S * (R + 3) * 1024;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6, lane = threadIdx.x & 63;
  double* res = reinterpret_cast<double*>(ring + wpb * RING) + w * GEMV_LDS_RESCAP;
  gemv_lds3_desc* D = reinterpret_cast<gemv_lds3_desc*>(ring + wpb * RING + wpb * GEMV_LDS_RESCAP * sizeof(double));
  if (w == 0) {
    int base = 0;
    for (int p0 = 0; p0 < count; p0 += 64) {
      const int p = p0 + lane;
      gemv_lds3_desc d;
      bool live = false;
      if (p < count) live = rsv(first + p, &d);
      const unsigned long long mask = __ballot(live);
      if (live) D[base + __popcll(mask & ((1ull << lane) - 1ull))] = d;
      base += __popcll(mask);
    }
    if (lane == 0) nlive_s = base;
  }
  __syncthreads();
  const int nlive = nlive_s;
  if (nlive == 0) return;
  float su = stagger_per_problem * (float)nlive;
  const int stagger = (stagger_per_problem <= 0.f || su < 0.25f) ? 0 : su < 1.f ? 1 : su > 32.f ? 32 : (int)(su + 0.5f);
  for (int i = (int)((((int64_t)blockIdx.x * wpb + w) * 37) & 63) * stagger; i > 0; --i) __builtin_amdgcn_s_sleep(1);
  gemv_lds3_multi_body<R, S, NT>(D, nlive, n, (int64_t)blockIdx.x * wpb + w, (int64_t)gridDim.x * wpb, ring + w * RING, res, rs);
#endif
}

template <int R, int S, class Resolve3>
static int bp_launch_gemv3_pers(ccqp_ctx* ctx, const Resolve3& rsv, int count, int64_t n, int wpb, hipStream_t s) {
  const int chunk = count < BMV3_PERS_MAX ? count : BMV3_PERS_MAX;
  const size_t lds = (size_t)wpb * ((size_t)S * (R + 3) * 1024 + GEMV_LDS_RESCAP * sizeof(double)) +
                     (size_t)chunk * sizeof(gemv_lds3_desc);
  if (lds > 160 * 1024) {
    ccqp_set_error("%s: ring does not fit", __func__);
    return CCQP_ERR_UNSUPPORTED;
  }
  const bool keep = bp_pers_cacheable(ctx, count, n);
  const int rcl = keep ? ccqp_lds_raise(ctx, reinterpret_cast<const void*>(bp_gemv3_pers_kernel<R, S, Resolve3, false>), lds)
                       : ccqp_lds_raise(ctx, reinterpret_cast<const void*>(bp_gemv3_pers_kernel<R, S, Resolve3, true>), lds);
  if (rcl) return rcl;
  const float stagger = ccqp_opt(ctx, CCQP_OPT_BMV_STAGGER) ? 0.f : (float)(8.0 * (double)n * (double)n / 4294967296.0);
  const int64_t rs = gemv_lds_row_stride(ccqp_opt(ctx, CCQP_OPT_GEMV_ROW_STRIDE), n, n, R);
  if (ccqp_mvt_on) ccqp_mvt_mark(s, 0);
  for (int first = 0; first < count; first += BMV3_PERS_MAX) {
    const int m = count - first < BMV3_PERS_MAX ? count - first : BMV3_PERS_MAX;
    if (keep)
      hipLaunchKernelGGL((bp_gemv3_pers_kernel<R, S, Resolve3, false>), dim3((unsigned)ctx->num_cu), dim3(64 * wpb), lds, s, rsv,
                         first, m, n, rs, stagger);
    else
      hipLaunchKernelGGL((bp_gemv3_pers_kernel<R, S, Resolve3, true>), dim3((unsigned)ctx->num_cu), dim3(64 * wpb), lds, s, rsv,
                         first, m, n, rs, stagger);
  }
  if (ccqp_mvt_on) ccqp_mvt_mark(s, 1);
  return CCQP_OK;
}

// ---- the products of a sharing launch from the batched VALU body, blockIdx.z = the product (round 4)
// A batch whose matrices fit the MALL together (256 MiB: one problem of n < 4096, 32 of n = 1024, 512 of n = 256) is
// bound by the NUMBER of its launches (~5 us each), not by its bytes: a second product re-reads the matrix from the
// MALL / L2.  What the persistent three-vector launch shares for large matrices -- the pass -- is worth nothing here; what
// lock-step APGD's fused rounds save is ROUNDS (one per accepted iteration instead of two).  So below the persistent
// form's sizes the same `Resolve3` descriptors are multiplied by the VALU body of the single-product launches (gemv_body:
// the bits are a function of (row, n) only), one grid slice per product.  `nz`: products a descriptor can name.
// (MPRGP's look-ahead through this launch was measured too and is NOT taken: it saves passes, not rounds -- the M1 and
// A.p rounds of a CG step remain, without a matvec -- and adds a launch per round: whole solves 0.81x at n = 256, 0.93-0.98x
// at 1024, +-1 % at 2048 / 3072, one problem to 512; it would take fused MPRGP rounds)
template <int U, bool ALIGNED, bool NT, class Resolve3>
__global__ __launch_bounds__(CCQP_BLOCK) void bp_gemv3_valu_kernel(const Resolve3 rsv, int64_t n) {
  gemv_lds3_desc d;
  if (!rsv((int)blockIdx.y, &d)) return;
  const int z = blockIdx.z;
  const double* x = z == 0 ? d.x[0] : z == 1 ? d.x[1] : d.x[2];
  double* y = z == 0 ? d.y[0] : z == 1 ? d.y[1] : d.y[2];
  double* g = z == 0 ? d.g[0] : z == 1 ? d.g[1] : d.g[2];
  if (y == nullptr) return;
  gemv_body<BP_R(U), BP_U(U), ALIGNED, NT>(d.A, d.lda, x, g ? d.b : nullptr, y, g, n, n, BP_TASKS(U, n),
                                           (int64_t)blockIdx.x * (CCQP_BLOCK / 64) + (threadIdx.x >> 6),
                                           (int64_t)gridDim.x * (CCQP_BLOCK / 64));
}

// whether a run that shares passes takes this form: default tuning word, the batch's matrices <= 256 MiB (the context's
// option SMALL_SHARING: 1 never, 2 whatever the batch).  Measured on whole lock-step APGD solves, fused against unfused
// rounds (tools/bench_small_sharing.py, profiles/r04_apgd_fused_small.jsonl): n = 256: 4 problems +31 %, 32 +23 %,
// 128 (64 MiB) +16 %, 512 (256 MiB) +-0; n = 1024: 4 +24 %, 16 (128 MiB) +7 %, 32 +-0, 100 -6 %; n = 2048: 4 (128 MiB)
// +10 %, 16 -1 %; n = 3072: 4 +2 %, 12 -4 %.
#define BP_SMALL_SHARING_BYTES 268435456.0
static inline bool bp_valu_sharing(const ccqp_ctx* ctx, int flags, int count, int64_t n) {
  const int64_t o = ccqp_opt(ctx, CCQP_OPT_SMALL_SHARING);
  if (o == 1 || (flags & ~(1 | 8 | 128)) != 0) return false;
  return o == 2 || 8.0 * (double)n * (double)n * (double)count <= BP_SMALL_SHARING_BYTES;
}

template <class Resolve3>
static void bp_launch_gemv3_valu(const Resolve3& rsv, int count, int64_t n, int flags, int nz, hipStream_t s) {
  bp_mv_tuning t = bp_mv_tune(count, n, flags);
  t.grid.z = (unsigned)nz;
#undef BP_EXTRA_TARGS
#define BP_EXTRA_TARGS , Resolve3
  BP_MV_LAUNCH(bp_gemv3_valu_kernel, rsv, n);
#undef BP_EXTRA_TARGS
#define BP_EXTRA_TARGS
}

// ring shape of the three-vector launch: rows per wave, slots, waves per workgroup.  EIGHT rows per wave, two slots, four
// waves: a step brings 8 KiB of matrix with 3 KiB of vectors instead of 4 + 3 -- against <4, 4> x 4 waves +1.5 % on
// config 5's share (788 -> 800 problems/s), +2...+4 % on lock-step MPRGP / APGD batches of n = 1024 ... 8192, nine cases
// of nine (profiles/r03_bmv3_shapes.jsonl; <8, 3> x 3 waves loses 1-7 %, <4, 3> x 4 is within +-1 %).  The bits do not
// depend on the shape.  With FEW matrices per launch the waves crowd on one matrix at a time and the bytes a wave keeps
// in flight count for more: <4, 4> x 4 waves (12 KiB per wave against 8) is 2-5 % ahead for a single problem of
// n <= 16384 and level at n = 32768 (ccqp_gemv3_f64), level to +1 % for 2 ... 16 problems -- kept below 32 problems.
// The context's option BMV3_SHAPE = <R><S><W> is the A/B switch of the tools.
static inline int bp_gemv3_shape(const ccqp_ctx* ctx, int count) {
  const int forced = (int)ccqp_opt(ctx, CCQP_OPT_BMV3_SHAPE);
  return forced ? forced : count >= 32 ? 824 : 444;
}
template <class Resolve3>
static int bp_launch_gemv3(ccqp_ctx* ctx, const Resolve3& rsv, int count, int64_t n, hipStream_t s) {
  switch (bp_gemv3_shape(ctx, count)) {
    case 833: return bp_launch_gemv3_pers<8, 3>(ctx, rsv, count, n, 3, s);
    case 444: return bp_launch_gemv3_pers<4, 4>(ctx, rsv, count, n, 4, s);
    case 434: return bp_launch_gemv3_pers<4, 3>(ctx, rsv, count, n, 4, s);
    default: return bp_launch_gemv3_pers<8, 2>(ctx, rsv, count, n, 4, s);
  }
}

#define BP_CHECK_BATCH()                                                                          \
  CCQP_REQUIRE(ctx != nullptr && dev_probs != nullptr && dev_batch != nullptr, "NULL argument"); \
  CCQP_REQUIRE(count >= 1 && count <= 65535 && n >= 1, "bad batch shape")


// zeroes the two batch words and resets the host mirror (ccqp_bpipe.hip)
void ccqp_lockstep_begin(ccqp_ctx* ctx, uint32_t* dev_batch, hipStream_t s);
// the batched matvec (mode 0: y = A x[i], g[i] = y + b; mode 1: y = A g[i]) and the alpha0 reduction of the BBPGD
// start-up, as launched by ccqp_bpipe_*: shared with the segmented form (ccqp_bsegs.hip)
int ccqp_bp_gemv(ccqp_ctx* ctx, const ccqp_pprob* probs, int count, int64_t n, int i, int mode, int flags, hipStream_t s);
void ccqp_bp_init_alpha(const ccqp_pprob* probs, int count, int64_t n, int64_t chunk, int nblk, hipStream_t s);
// the launches of a lock-step APGD round (ccqp_bpipe.hip), shared with the segmented form: `segs` != NULL takes the
// residual from the segmented residual launch and leaves the trial step to the segmented step launch
// (`look`: the caller asked for the look-ahead -- CCQP_LOCKSTEP_LOOKAHEAD -- and gave every problem its two buffers)
int ccqp_bapgd_launch_gemv(ccqp_ctx* ctx, const ccqp_papgd* probs, int count, int64_t n, int init, int flags, bool look,
                           hipStream_t s, bool elementwise = true);
void ccqp_bapgd_launch_decide(ccqp_ctx* ctx, const ccqp_papgd* probs, const ccqp_psegs* segs, int count, int64_t n,
                              double gd, double seq, int ar, uint32_t* dev_batch, hipStream_t s);
void ccqp_bapgd_launch_apply(const ccqp_papgd* probs, int count, int64_t n, hipStream_t s);
void ccqp_bapgd_launch_spec(const ccqp_papgd* probs, int count, int64_t n, hipStream_t s);   // (look-ahead: y_{k+1} of a trial point)
