// Sparse (CSR, fp64) form of the solvers' `A.dot(v)` (reference: src/ccqppy/solvers.py:25 -- the solver
// only ever calls `A.dot`, so a scipy.sparse matrix is a legal `A` there).
//
// HBM-bound: every stored entry is read once (8 B value + 4 B column), x is gathered through L2 /
// Infinity Cache, one 8-B row pointer and one 8-B result per row.
//
// Structure ("CSR-stream"): the entry range is cut into windows of SPMV_CAP entries; workgroup i owns
// the rows that START in window i (a plan computed once per matrix, ccqp_spmv_plan_csr).  It streams its
// entries with all 256 threads -- fully coalesced whatever the row lengths, 8 independent (value, column)
// loads per thread in flight, then 8 gathers -- parks the products in LDS, and subgroups of W lanes reduce
// one row each out of LDS.  Where the rows of a window have (nearly) the same length and at least ~28
// entries (plan word 4), the x gathers are issued in TRANSPOSED order -- consecutive lanes take the same
// entry position of consecutive rows, (value, column) coming through LDS -- because banded / stencil-like
// matrices then gather consecutive x, which the L1 merges; in entry order every lane of a gather is a cache
// access of its own.  The columns travel through the LDS region that later holds the products, the gathered x
// travel back through it to the threads that hold the values: no LDS beyond the products themselves.  A window's last row bypasses LDS when it is longer than SPMV_LONG entries (one wave
// per row, second launch).
//
// Windows whose entries reuse their columns (plan word 4, SPMV_LOCALX) gather each DISTINCT x once into LDS and
// read a 16-bit local column per entry ("local x" mode).
//
// Measured (profiles/r01_spmv_*): 5.8-6.3 TB/s of algorithmic bytes on a 27-point-stencil pattern (local-x
// windows); 4.0-4.6 TB/s on hashed-circulant matrices with 8..128 entries per row, which reuse no column inside
// a window (4.9 TB/s when x fits one XCD's L2, n <= 512k); 0.7-1.3 TB/s when the columns are scrambled (every
// gather a cache miss).  The same kernel with the gather replaced by arithmetic streams at 7.2 TB/s; with
// it the time scales with ALL bytes delivered through the vector-memory path -- 12 B of stream + 8 B gathered
// per entry: 7 TB/s x 12/20 = 4.2 TB/s -- and does not respond to the gather order (1.6 vs 0.85 L1 accesses
// per entry), to occupancy (5 vs 8 workgroups per CU) or to W: the gathered x is the bound, not the HBM
// stream.  Variants that were built and measured slower: a CSR-vector kernel (subgroup per row straight from
// global memory: 2-3 TB/s, one load per lane in flight and half-empty load instructions), several rows per
// subgroup, and two persistent software-pipelined forms (next window's loads in flight during the reduce:
// 100+ VGPRs, 4 workgroups per CU, 2.6-3.3 TB/s); for local-x windows, 16-byte packed loads of values, local
// columns and distinct columns (12 instead of 26 vector-memory instructions per thread, plan data laid out to
// match): 0.129 vs 0.123 ms on the 2M-row stencil -- that mode already moves ~6 TB/s of real HBM traffic.
//
// Determinism contract (same as the dense matvec): the bits of y[i] are a function of row i's stored
// entries (in storage order) and x ONLY: product p_e = fl(val_e * x[col_e]); canonical lane L = e mod 64
// adds its products in order starting from +0.0; the 64 lane sums are combined by the xor butterfly
// 32,16,8,4,2,1.  Lane l of a W-lane subgroup carries the 64/W canonical lanes l, l+W, ... in separate
// accumulators and does the butterfly steps >= W in registers, so W, the window plan and any
// row-sharding of the matrix are pure tuning / layout choices.
#include "ccqp_internal.h"

#define SPMV_LDS 2048              // products a workgroup parks in LDS (8 per thread)
#define SPMV_LONG 256              // a window's last row bypasses LDS (one wave per row) if it is longer than this
#define SPMV_CAP (SPMV_LDS - SPMV_LONG)   // window size in entries: rows starting in it hold < SPMV_LDS entries
#define SPMV_SLOTS (SPMV_LDS / CCQP_BLOCK)   // entries per thread in the streaming phase
#define SPMV_PLAN_WORDS 5          // per window: first row, end row (LDS path), first entry, end entry (LDS path),
                                   // longest row if the window takes the transposed gather order, else 0
#define SPMV_UNIFORM (1ll << 32)   // flag in plan word 4: all rows of the window have the same length
#define SPMV_LOCALX (1ll << 33)    // flag in plan word 4: the window gathers x through LDS ("local x" mode); the
                                   // number of distinct columns sits in bits 40-55
#define SPMV_DIAG (1ll << 34)      // flag in plan word 4: "diagonal" window -- every row has the same length L (bits
                                   // 40-55) and the same column OFFSETS col - row, stored once (L int32) where a
                                   // local-x window keeps its distinct columns; the matvec then reads no column
                                   // index at all for the window
#define SPMV_DIAG_MAXLEN 256       // longest row of a diagonal window (its offsets live in 1 KiB of LDS)
#define SPMV_DIAG_MINLEN 2         // shortest row of a diagonal window
#define SPMV_SHORT_MAX 64          // diagonal windows with rows of at most this many entries take the threads-per-row
                                   // path (spmv_short_rows: 1 / 2 / 4 threads per row up to 16 / 32 / 64 entries):
                                   // round 2's generic phases were VALU-bound there (80 % VALU busy, 1.6 L1 accesses
                                   // per entry at 9 entries per row, profiles/r01_spmv_16777216k8_*)
#define SPMV_NODIAG 1              // ccqp_spmv_plan_set: keep every window off the diagonal paths (tuning / A-B switch)
#define SPMV_XS 1024               // most distinct columns of a local-x window (their x values share the LDS of the
                                   // products: 8 KB)
#define SPMV_TRANSPOSE_ROWS 64     // windows with more rows than this (shorter rows) stay in entry order: measured
                                   // slower in transposed order even with the uniform-length shortcut (9 entries
                                   // per row: 0.57 vs 0.49 ms at n = 16.8M)
#define SPMV_ROWS_LDS 1790         // row starts of a window live in LDS (16-bit) up to this many rows: with the
                                   // products 19.9 KB per workgroup = 8 workgroups per CU

// ------------------------------------------------------------------------------------------ plan
// first row r in [0, rows] whose start rowptr[r] is >= target (rows if none)
__device__ __forceinline__ int64_t spmv_lower_bound(const int64_t* __restrict__ rowptr, int64_t rows, int64_t target) {
  int64_t lo = 0, hi = rows;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (rowptr[mid] < target) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(CCQP_BLOCK) void spmv_plan_kernel(const int64_t* __restrict__ rowptr, int64_t rows,
                                                               int64_t nwin, int64_t* __restrict__ plan,
                                                               unsigned long long* __restrict__ nlong,
                                                               unsigned long long* __restrict__ ntransposed,
                                                               int64_t* __restrict__ long_rows) {
  const int64_t i = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x;
  if (i >= nwin) return;
  const int64_t rb = spmv_lower_bound(rowptr, rows, i * SPMV_CAP);
  int64_t re = spmv_lower_bound(rowptr, rows, (i + 1) * SPMV_CAP);
  // LDS capacity: the rows before the window's last row span < SPMV_CAP entries, so only the LAST row can
  // overflow the SPMV_LDS products a workgroup parks; if it is longer than SPMV_LONG it takes the
  // wave-per-row path instead (a row longer than SPMV_CAP is always the last row of its window)
  if (re > rb && rowptr[re] - rowptr[re - 1] > SPMV_LONG) {
    re -= 1;
    long_rows[atomicAdd(nlong, 1ull)] = re;
  }
  const int64_t nzb = rowptr[rb], nze = rowptr[re];
  // transposed gather order: the (position, row) grid of the window must fit the 8 grid points per thread,
  // i.e. the rows have (nearly) the same length; measured to pay from ~28 entries per row on
  int64_t longest = 0;
  const int64_t nr = re - rb;
  if (nr > 0 && nr <= SPMV_TRANSPOSE_ROWS && nze > nzb) {
    int64_t shortest = nze - nzb;
    for (int64_t r = rb; r < re; ++r) {
      const int64_t len = rowptr[r + 1] - rowptr[r];
      longest = len > longest ? len : longest;
      shortest = len < shortest ? len : shortest;
    }
    const bool uniform = longest == shortest;           // e = row * length + position: no row-start lookups
    if (nr * longest > SPMV_LDS) longest = 0;
    if (longest > 0) atomicAdd(ntransposed, 1ull);
    if (longest > 0 && uniform) longest |= SPMV_UNIFORM;
  }
  plan[SPMV_PLAN_WORDS * i + 0] = rb;
  plan[SPMV_PLAN_WORDS * i + 1] = re;
  plan[SPMV_PLAN_WORDS * i + 2] = nzb;
  plan[SPMV_PLAN_WORDS * i + 3] = nze;
  plan[SPMV_PLAN_WORDS * i + 4] = longest;
}

static inline int64_t spmv_windows(int64_t nnz) { return nnz / SPMV_CAP + 1; }
// plan layout (int64 words): [5 per window][4 counters: long rows, transposed windows, local-x windows, diagonal windows]
// [long-row list, one slot per window][distinct columns, SPMV_XS int32 per window][16-bit local column per entry]
static inline int64_t spmv_off_counts(int64_t nwin) { return SPMV_PLAN_WORDS * nwin; }
static inline int64_t spmv_off_long(int64_t nwin) { return spmv_off_counts(nwin) + 4; }
static inline int64_t spmv_off_ucol(int64_t nwin) { return spmv_off_long(nwin) + nwin; }
static inline int64_t spmv_off_lcol(int64_t nwin) { return spmv_off_ucol(nwin) + nwin * (SPMV_XS / 2); }

extern "C" int64_t ccqp_spmv_plan_words(int64_t rows, int64_t nnz) {
  if (rows < 0 || nnz < 0) return -1;
  const int64_t nwin = spmv_windows(nnz);
  return spmv_off_lcol(nwin) + (nnz + 3) / 4;
}

// Local-x analysis, one workgroup per window: sort the window's column indices (bitonic, in LDS), keep the
// distinct ones, and -- if they fit SPMV_XS and every x value is used 1.5 times or more -- store them
// (sorted) plus, for every entry, the 16-bit position of its column in that list.  The matvec then loads
// each distinct x ONCE with coalesced gathers, parks it in LDS, and reads 2 bytes instead of 4 per entry.
__global__ __launch_bounds__(CCQP_BLOCK) void spmv_plan_localx_kernel(const int64_t* __restrict__ rowptr,
                                                                      const int32_t* __restrict__ col,
                                                                      int64_t* __restrict__ plan,
                                                                      int32_t* __restrict__ ucol,
                                                                      uint16_t* __restrict__ lcol,
                                                                      unsigned long long* __restrict__ nlocal) {
  __shared__ int32_t key[SPMV_LDS];
  __shared__ int ndistinct;
  __shared__ int notdiag;
  const int64_t w = blockIdx.x;
  const int64_t nzb = plan[SPMV_PLAN_WORDS * w + 2];
  const int E = (int)(plan[SPMV_PLAN_WORDS * w + 3] - nzb);
  if (E <= 0) return;
  const int t = threadIdx.x;
  for (int i = t; i < SPMV_LDS; i += CCQP_BLOCK) key[i] = i < E ? col[nzb + i] : INT32_MAX;
  __syncthreads();
  for (int k = 2; k <= SPMV_LDS; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < SPMV_LDS; i += CCQP_BLOCK) {
        const int partner = i ^ j;
        if (partner > i) {
          const int32_t a = key[i], b = key[partner];
          const bool ascending = (i & k) == 0;
          if ((a > b) == ascending) {
            key[i] = b;
            key[partner] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  if (t == 0) {
    int u = 0;
    for (int i = 0; i < E; ++i)
      if (i == 0 || key[i] != key[i - 1]) key[u++] = key[i];      // u <= i: in place
    ndistinct = u;
  }
  __syncthreads();
  const int cnt = ndistinct;
  int32_t* uc = ucol + w * SPMV_XS;
  if (cnt > SPMV_XS || 2 * E < 3 * cnt) {                         // too many columns, or too little reuse
    // ---- "diagonal" window?  All rows of the same length L <= SPMV_DIAG_MAXLEN and, position by position, the same
    // offset col - row (banded / circulant / regular-grid matrices): keep the L offsets once; the matvec rebuilds
    // every column index as row + offset and streams 8 instead of 12 bytes per entry.
    // (measured +17 ... +36 % at 17-129 entries per row -- in transposed and in entry order alike --
    // profiles/r02_spmv_sweep.jsonl; rows of up to SPMV_SHORT_MAX entries take a path of their own, spmv_short_rows)
    const int64_t rb = plan[SPMV_PLAN_WORDS * w + 0];
    const int64_t nr = plan[SPMV_PLAN_WORDS * w + 1] - rb;
    if (nr < 1 || E % nr != 0) return;
    const int L = (int)(E / nr);
    if (L < SPMV_DIAG_MINLEN || L > SPMV_DIAG_MAXLEN) return;
    // short rows take the thread-per-row path, which transposes the window's values through LDS with the row stride
    // padded to an odd number of doubles: the padded window must fit (it does not for full windows of 2, 4, 6 entries
    // per row, which stay on the generic paths)
    if (L <= SPMV_SHORT_MAX && nr * (int64_t)(L | 1) > SPMV_LDS) return;
    if (t == 0) notdiag = 0;
    __syncthreads();
    int bad = 0;
    for (int64_t r = t; r < nr; r += CCQP_BLOCK) bad |= (rowptr[rb + r + 1] - rowptr[rb + r]) != L;
    for (int i = t; i < E; i += CCQP_BLOCK) {
      const int r = i / L, pos = i - r * L;
      bad |= ((int64_t)col[nzb + i] - (rb + r)) != ((int64_t)col[nzb + pos] - rb);
    }
    if (bad) notdiag = 1;                                         // (benign race: every writer stores 1)
    __syncthreads();
    if (notdiag) return;
    for (int i = t; i < L; i += CCQP_BLOCK) uc[i] = (int32_t)((int64_t)col[nzb + i] - rb);
    if (t == 0) {
      plan[SPMV_PLAN_WORDS * w + 4] |= SPMV_DIAG | ((int64_t)L << 40);
      atomicAdd(nlocal + 1, 1ull);
    }
    return;
  }
  for (int i = t; i < cnt; i += CCQP_BLOCK) uc[i] = key[i];
  for (int i = t; i < E; i += CCQP_BLOCK) {
    const int32_t c = col[nzb + i];
    int lo = 0, hi = cnt - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (key[mid] < c) lo = mid + 1; else hi = mid;
    }
    lcol[nzb + i] = (uint16_t)lo;
  }
  if (t == 0) {
    plan[SPMV_PLAN_WORDS * w + 4] |= SPMV_LOCALX | ((int64_t)cnt << 40);
    atomicAdd(nlocal, 1ull);
  }
}

extern "C" int ccqp_spmv_plan_csr(ccqp_ctx* ctx, const int64_t* rowptr, const int32_t* col, int64_t rows,
                                  int64_t nnz, int64_t* plan, int64_t* summary, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(rows >= 0 && nnz >= 0, "negative size");
  CCQP_REQUIRE(rowptr != nullptr && plan != nullptr, "NULL rowptr / plan");
  hipStream_t s = (hipStream_t)stream;
  const int64_t nwin = spmv_windows(nnz);
  int64_t* count = plan + spmv_off_counts(nwin);
  CCQP_HIP(hipMemsetAsync(count, 0, 4 * sizeof(int64_t), s));
  const int64_t grid = (nwin + CCQP_BLOCK - 1) / CCQP_BLOCK;
  CCQP_REQUIRE(grid <= (int64_t)INT32_MAX, "matrix too large for one plan launch");
  hipLaunchKernelGGL(spmv_plan_kernel, dim3((unsigned)grid), dim3(CCQP_BLOCK), 0, s, rowptr, rows, nwin, plan,
                     (unsigned long long*)count, (unsigned long long*)count + 1, plan + spmv_off_long(nwin));
  CCQP_REQUIRE(nwin <= (int64_t)INT32_MAX, "matrix too large for one plan launch");
  if (col != nullptr && nnz > 0)             // local-x analysis (optional: without col the windows keep their mode)
    hipLaunchKernelGGL(spmv_plan_localx_kernel, dim3((unsigned)nwin), dim3(CCQP_BLOCK), 0, s, rowptr, col, plan,
                       reinterpret_cast<int32_t*>(plan + spmv_off_ucol(nwin)),
                       reinterpret_cast<uint16_t*>(plan + spmv_off_lcol(nwin)), (unsigned long long*)count + 2);
  CCQP_LAUNCH_CHECK();
  if (summary) {                             // blocking: lets the matvec skip the wave-per-row launch
    CCQP_HIP(hipMemcpyAsync(summary, count, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    CCQP_HIP(hipStreamSynchronize(s));
  }
  return CCQP_OK;
}

// Tuning / A-B switch on a finished plan: SPMV_NODIAG sends every diagonal window back to the generic paths (same
// bits -- the paths differ in what they read, not in what they add); `summary`, if given, is updated to match.
__global__ __launch_bounds__(CCQP_BLOCK) void spmv_plan_nodiag_kernel(int64_t* __restrict__ plan, int64_t nwin,
                                                                      int64_t* __restrict__ ndiag) {
  const int64_t w = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x;
  if (w < nwin && (plan[SPMV_PLAN_WORDS * w + 4] & SPMV_DIAG))
    plan[SPMV_PLAN_WORDS * w + 4] &= 0xffffffffll | SPMV_UNIFORM;   // keep the transposed-order words, drop flag + length
  if (w == 0) *ndiag = 0;
}

extern "C" int ccqp_spmv_plan_set(ccqp_ctx* ctx, int64_t* plan, int64_t nnz, int flags, int64_t* summary,
                                  void* stream) {
  CCQP_REQUIRE(ctx != nullptr && plan != nullptr, "NULL argument");
  CCQP_REQUIRE(nnz >= 0, "negative size");
  CCQP_REQUIRE((flags & ~SPMV_NODIAG) == 0, "unknown flag");
  if (flags & SPMV_NODIAG) {
    const int64_t nwin = spmv_windows(nnz);
    const int64_t grid = (nwin + CCQP_BLOCK - 1) / CCQP_BLOCK;
    hipLaunchKernelGGL(spmv_plan_nodiag_kernel, dim3((unsigned)grid), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, plan,
                       nwin, plan + spmv_off_counts(nwin) + 3);
    CCQP_LAUNCH_CHECK();
    if (summary) summary[3] = 0;
  }
  return CCQP_OK;
}

// ------------------------------------------------------------------------------------------ matvec
// v of the lane `s` places up, for the butterfly steps inside a 16-lane row: DPP register moves instead of
// two ds_bpermute round trips per step.  Lane 0 of a subgroup (the only lane whose sum is stored) sees the
// same operands as in the xor butterfly, and a + b == b + a bitwise.
template <int CTRL>
__device__ __forceinline__ double spmv_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

template <int W>
__device__ __forceinline__ double spmv_subgroup_sum(double v) {
#pragma unroll
  for (int s = W / 2; s >= 16; s >>= 1) v = v + __shfl_xor(v, s, W);
  if (W >= 16) v = v + spmv_dpp<0x128>(v);   // row_ror:8   == lane ^ 8 within the row
  if (W >= 8) v = v + spmv_dpp<0x104>(v);    // row_shl:4   lane l <- lane l + 4
  if (W >= 4) v = v + spmv_dpp<0x4E>(v);     // quad_perm [2,3,0,1] == lane ^ 2
  if (W >= 2) v = v + spmv_dpp<0xB1>(v);     // quad_perm [1,0,3,2] == lane ^ 1
  return v;
}

// ---- diagonal window with SHORT rows (L <= SPMV_SHORT_MAX entries each, all rows alike, col = row + offs[position]):
// S = 1, 2 or 4 THREADS per row (L <= 16, 32, 64), each taking the positions p = s, s + S, s + 2S, ... (at most 16).
// The values arrive coalesced in entry order and are transposed through LDS (row stride padded to an odd number of
// doubles: conflict-free); thread (i, s) then gathers x[row_i + offs[p]] -- the threads of consecutive rows read
// consecutive x, i.e. every gather instruction is S coalesced runs -- and adds its products in the canonical order in
// registers: no row pointers, no column stream, no products in LDS, no 16-accumulator subgroup loops.
// Canonical order for a row of L <= 64 entries: entry p sits alone in canonical lane p (lane sum = +0.0 + product), the
// other lanes hold +0.0.  Thread s owns the lanes s + S j, so the butterfly steps 32 ... S pair lanes of ONE thread
// (slots j and j + h, h = 8, 4, 2, 1 -- for S < 4 the first steps only add +0.0 to sums that are never -0.0: skipped),
// and the steps S/2 ... 1 pair neighbouring threads (DPP, as spmv_subgroup_sum).  Lane 0's operands; a + b == b + a.
template <int S, bool EPI>
__device__ __forceinline__ void spmv_short_rows(const double* __restrict__ val, const int32_t* __restrict__ uc,
                                                const double* __restrict__ x, const double* __restrict__ b,
                                                double* __restrict__ y, double* __restrict__ g, double* lds,
                                                int32_t* offs, int64_t rb, int64_t nzb, int E, int L, int64_t elast) {
  constexpr int Q = 16;                        // positions per thread
  const int t = threadIdx.x;
  const int nr = E / L;
  const int LP = L | 1;
  double v[SPMV_SLOTS];
#pragma unroll
  for (int j = 0; j < SPMV_SLOTS; ++j) {
    const int64_t e = nzb + t + j * CCQP_BLOCK;
    v[j] = __builtin_nontemporal_load(val + (e < elast ? e : elast));
  }
  if (t < L) offs[t] = uc[t];
  {
    const int qs = CCQP_BLOCK / L, ms = CCQP_BLOCK % L;
    int r = t / L, pos = t - r * L;
#pragma unroll
    for (int j = 0; j < SPMV_SLOTS; ++j) {
      if (t + j * CCQP_BLOCK < E) lds[r * LP + pos] = v[j];
      r += qs;
      pos += ms;
      if (pos >= L) {
        pos -= L;
        r += 1;
      }
    }
  }
  __syncthreads();
  const int sl = t % S;
  // (all threads of a wave run the loop together: the DPP steps read lanes of rows past the end, whose q are +0.0)
  for (int i0 = 0; i0 < nr; i0 += CCQP_BLOCK / S) {
    const int i = i0 + t / S;
    const bool live = i < nr;
    const double* __restrict__ xr = x + rb + (live ? i : 0);
    const double* vr = lds + (live ? i : 0) * LP;
    double q[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const int p = sl + S * j;
      q[j] = (live && p < L) ? xr[offs[p]] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const int p = sl + S * j;
      if (live && p < L) {
        const double pr = vr[p] * q[j];
        q[j] = 0.0 + pr;
      }
    }
#pragma unroll
    for (int h = Q / 2; h >= 1; h >>= 1) {
#pragma unroll
      for (int j = 0; j < h; ++j) q[j] = q[j] + q[j + h];
    }
    double sum = q[0];
    if (S >= 4) sum = sum + spmv_dpp<0x4E>(sum);     // quad_perm [2,3,0,1] == lane ^ 2
    if (S >= 2) sum = sum + spmv_dpp<0xB1>(sum);     // quad_perm [1,0,3,2] == lane ^ 1
    if (live && sl == 0) {
      y[rb + i] = sum;
      if (EPI) g[rb + i] = sum + b[rb + i];
    }
  }
}

template <int W, bool EPI>
__global__ __launch_bounds__(CCQP_BLOCK) void spmv_stream_kernel(const int64_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ col,
                                                                 const double* __restrict__ val,
                                                                 const int64_t* __restrict__ plan,
                                                                 const int32_t* __restrict__ ucol,
                                                                 const uint16_t* __restrict__ lcol,
                                                                 const double* __restrict__ x,
                                                                 const double* __restrict__ b, double* __restrict__ y,
                                                                 double* __restrict__ g, int64_t last,
                                                                 const double* __restrict__ skip) {
  constexpr int K = 64 / W;                 // canonical lanes per physical lane
  constexpr int SUBGROUPS = CCQP_BLOCK / W;
  if (skip != nullptr && *skip != 0.0) return;      // pipelined solves: queued behind a converged iteration
  __shared__ double prod[SPMV_LDS];
  __shared__ uint16_t rstart[SPMV_ROWS_LDS + 2];      // row starts relative to the window
  __shared__ int32_t offs[SPMV_DIAG_MAXLEN];          // column offsets of a diagonal window
  // transposed windows pass the column indices through the first half of `prod` before any product exists
  int32_t* const colv = reinterpret_cast<int32_t*>(prod);
  const int64_t* pw = plan + (int64_t)SPMV_PLAN_WORDS * blockIdx.x;
  const int64_t rb = pw[0], re = pw[1], nzb = pw[2];
  const int E = (int)(pw[3] - nzb);          // < SPMV_LDS by construction of the plan
  const int longest = (int)(pw[4] & 0xffffffff);
  const bool uniform = (pw[4] & SPMV_UNIFORM) != 0;
  const bool localx = (pw[4] & SPMV_LOCALX) != 0;
  const bool diag = (pw[4] & SPMV_DIAG) != 0;
  if (diag && (int)((pw[4] >> 40) & 0xffff) <= SPMV_SHORT_MAX) {
    const int Ls = (int)((pw[4] >> 40) & 0xffff);
    const int32_t* ucw = ucol + (int64_t)blockIdx.x * SPMV_XS;
    if (E <= 0) return;
    if (Ls <= 16) spmv_short_rows<1, EPI>(val, ucw, x, b, y, g, prod, offs, rb, nzb, E, Ls, nzb + E - 1);
    else if (Ls <= 32) spmv_short_rows<2, EPI>(val, ucw, x, b, y, g, prod, offs, rb, nzb, E, Ls, nzb + E - 1);
    else spmv_short_rows<4, EPI>(val, ucw, x, b, y, g, prod, offs, rb, nzb, E, Ls, nzb + E - 1);
    return;
  }
  const int t = threadIdx.x;
  const int l = t % W;
  const int64_t nrows = re - rb;
  const bool rows_in_lds = nrows <= SPMV_ROWS_LDS;
  // an empty window (E == 0) may sit at the very end of the arrays: clamp to the last stored entry
  const int64_t elast = E > 0 ? nzb + E - 1 : (nzb < last ? nzb : last);
  // row pointers of the window: first in the load queue, needed last
  int64_t rp[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t i = t + j * CCQP_BLOCK;
    rp[j] = rowptr[rb + (i < nrows ? i : nrows)];
  }
  if (localx) {
    // ---- "local x" window: the distinct columns of the window (sorted, from the plan) are gathered ONCE,
    // coalesced, into LDS; every entry then multiplies its value with an LDS read through a 16-bit local
    // index.  Per entry 8 B + 2 B are streamed instead of 8 B + 4 B + an 8-B global gather.
    const int ucnt = (int)((pw[4] >> 40) & 0xffff);
    const int32_t* __restrict__ uc = ucol + (int64_t)blockIdx.x * SPMV_XS;
    int32_t ucv[SPMV_XS / CCQP_BLOCK];
#pragma unroll
    for (int q = 0; q < SPMV_XS / CCQP_BLOCK; ++q) {
      const int k = t + q * CCQP_BLOCK;
      ucv[q] = uc[k < ucnt ? k : ucnt - 1];
    }
    double v[SPMV_SLOTS];
    uint16_t lc[SPMV_SLOTS];
#pragma unroll
    for (int j = 0; j < SPMV_SLOTS; ++j) {
      const int64_t e = nzb + t + j * CCQP_BLOCK;
      const int64_t es = e < elast ? e : elast;
      v[j] = __builtin_nontemporal_load(val + es);
      lc[j] = __builtin_nontemporal_load(lcol + es);
    }
    if (rows_in_lds) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (t + j * CCQP_BLOCK <= nrows) rstart[t + j * CCQP_BLOCK] = (uint16_t)(rp[j] - nzb);
      for (int i = t + 2 * CCQP_BLOCK; i <= (int)nrows; i += CCQP_BLOCK)
        rstart[i] = (uint16_t)(rowptr[rb + i] - nzb);
    }
    double xg[SPMV_XS / CCQP_BLOCK];
#pragma unroll
    for (int q = 0; q < SPMV_XS / CCQP_BLOCK; ++q) xg[q] = x[ucv[q]];
    double* const xs = prod;                  // the first SPMV_XS doubles of `prod`, until the products exist
#pragma unroll
    for (int q = 0; q < SPMV_XS / CCQP_BLOCK; ++q) {
      const int k = t + q * CCQP_BLOCK;
      if (k < ucnt) xs[k] = xg[q];
    }
    __syncthreads();
    double pr[SPMV_SLOTS];
#pragma unroll
    for (int j = 0; j < SPMV_SLOTS; ++j) pr[j] = v[j] * xs[lc[j]];
    __syncthreads();                          // every x value has been read: `prod` may be overwritten
#pragma unroll
    for (int j = 0; j < SPMV_SLOTS; ++j) {
      const int e = t + j * CCQP_BLOCK;
      if (e < E) prod[e] = pr[j];
    }
  } else if (diag) {
    // ---- "diagonal" window: col = row + offset[position], the offsets (L int32, from the plan) parked in LDS.  No
    // column stream (8 instead of 12 bytes per entry) and, in transposed order, no pass of the columns through LDS.
    const int L = (int)((pw[4] >> 40) & 0xffff);
    const int32_t* __restrict__ uc = ucol + (int64_t)blockIdx.x * SPMV_XS;
    const int32_t myoff = uc[t < L ? t : L - 1];
    double v[SPMV_SLOTS];
#pragma unroll
    for (int j = 0; j < SPMV_SLOTS; ++j) {
      const int64_t e = nzb + t + j * CCQP_BLOCK;
      const int64_t es = e < elast ? e : elast;
      v[j] = __builtin_nontemporal_load(val + es);
    }
    if (rows_in_lds) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (t + j * CCQP_BLOCK <= nrows) rstart[t + j * CCQP_BLOCK] = (uint16_t)(rp[j] - nzb);
      for (int i = t + 2 * CCQP_BLOCK; i <= (int)nrows; i += CCQP_BLOCK)
        rstart[i] = (uint16_t)(rowptr[rb + i] - nzb);
    }
    if (t < L) offs[t] = myoff;
    __syncthreads();
    if (longest > 0) {
      // transposed order: thread p of the (position, row) grid takes position p / nr of row p % nr -- consecutive
      // lanes gather consecutive x
      const int nr = (int)nrows;
      const int q = CCQP_BLOCK / nr, m = CCQP_BLOCK % nr;
      int row = t % nr, pos = t / nr;
      int e[SPMV_SLOTS];
      bool ok[SPMV_SLOTS];
      int64_t cc[SPMV_SLOTS];
      double xv[SPMV_SLOTS];
#pragma unroll
      for (int j = 0; j < SPMV_SLOTS; ++j) {
        ok[j] = pos < L;
        e[j] = ok[j] ? row * L + pos : 0;
        cc[j] = rb + (ok[j] ? row : 0) + offs[ok[j] ? pos : 0];
        row += m;
        pos += q;
        if (row >= nr) {
          row -= nr;
          pos += 1;
        }
      }
#pragma unroll
      for (int j = 0; j < SPMV_SLOTS; ++j) xv[j] = x[cc[j]];
#pragma unroll
      for (int j = 0; j < SPMV_SLOTS; ++j)
        if (ok[j]) prod[e[j]] = xv[j];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < SPMV_SLOTS; ++j) {
        const int eo = t + j * CCQP_BLOCK;
        if (eo < E) prod[eo] = v[j] * prod[eo];
      }
    } else {
      // entry order: slot j of thread t is entry t + 256 j; its (row, position) follows from slot j-1's by adding
      // (256 / L, 256 % L) with one carry -- one integer division per thread instead of one per entry
      double xv[SPMV_SLOTS];
      const int qs = CCQP_BLOCK / L, ms = CCQP_BLOCK % L;
      int r = t / L, pos = t - r * L;
      const int rlast = (E - 1) / L, plast = (E - 1) - rlast * L;
#pragma unroll
      for (int j = 0; j < SPMV_SLOTS; ++j) {
        const bool in = t + j * CCQP_BLOCK < E;              // past the end: re-read the window's last entry
        xv[j] = x[rb + (in ? r : rlast) + offs[in ? pos : plast]];
        r += qs;
        pos += ms;
        if (pos >= L) {
          pos -= L;
          r += 1;
        }
      }
#pragma unroll
      for (int j = 0; j < SPMV_SLOTS; ++j) {
        const int e = t + j * CCQP_BLOCK;
        if (e < E) prod[e] = v[j] * xv[j];
      }
    }
  } else {
  // phase 1: ALL (value, column) loads of the window and its row pointers back to back, then all
  // gathers -- one pass through the load -> gather latency chain per workgroup.  Every load is
  // unconditional with a clamped index (a slot past the end of the window re-reads its last entry, one
  // cache line for the whole wave): loads inside conditional code make the compiler wait for them one by
  // one, and this kernel lives on the number of loads in flight.
  double v[SPMV_SLOTS];
  int32_t c[SPMV_SLOTS];
#pragma unroll
  for (int j = 0; j < SPMV_SLOTS; ++j) {
    const int64_t e = nzb + t + j * CCQP_BLOCK;
    const int64_t es = e < elast ? e : elast;
    v[j] = __builtin_nontemporal_load(val + es);
    c[j] = __builtin_nontemporal_load(col + es);
  }
  if (rows_in_lds) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (t + j * CCQP_BLOCK <= nrows) rstart[t + j * CCQP_BLOCK] = (uint16_t)(rp[j] - nzb);
    for (int i = t + 2 * CCQP_BLOCK; i <= (int)nrows; i += CCQP_BLOCK)      // > 511 rows in one window: rare
      rstart[i] = (uint16_t)(rowptr[rb + i] - nzb);
  }
  if (longest > 0) {
    // transposed window: (a) the column indices -> LDS
#pragma unroll
    for (int j = 0; j < SPMV_SLOTS; ++j) {
      const int e = t + j * CCQP_BLOCK;
      if (e < E) colv[e] = c[j];
    }
    __syncthreads();
    // (b) thread p of the (position, row) grid takes position p / nr of row p % nr: consecutive lanes
    // gather for consecutive rows (banded / stencil matrices: consecutive x, which the L1 merges -- in
    // entry order every lane of a gather is a cache access of its own).  nr * longest <= SPMV_LDS, so the
    // 8 grid points per thread cover the grid; gathers unconditional (clamped) as above.
    const int nr = (int)nrows;
    const int q = CCQP_BLOCK / nr, m = CCQP_BLOCK % nr;
    int row = t % nr, pos = t / nr;
    int e[SPMV_SLOTS];
    bool ok[SPMV_SLOTS];
    int32_t cc[SPMV_SLOTS];
    double xv[SPMV_SLOTS];
#pragma unroll
    for (int j = 0; j < SPMV_SLOTS; ++j) {
      if (uniform) {                          // the grid is exactly the window: a pure transposition
        ok[j] = pos < longest;
        e[j] = ok[j] ? row * longest + pos : 0;
      } else {
        const int s0 = rstart[row], s1 = rstart[row + 1];
        ok[j] = pos < s1 - s0;
        e[j] = ok[j] ? s0 + pos : 0;
      }
      cc[j] = colv[e[j]];
      row += m;
      pos += q;
      if (row >= nr) {
        row -= nr;
        pos += 1;
      }
    }
#pragma unroll
    for (int j = 0; j < SPMV_SLOTS; ++j) xv[j] = x[cc[j]];
    __syncthreads();                          // every column index has been read: `prod` may be overwritten
    // (c) the gathered x travel back to the entries' owners (who hold the values), again through LDS
#pragma unroll
    for (int j = 0; j < SPMV_SLOTS; ++j)
      if (ok[j]) prod[e[j]] = xv[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SPMV_SLOTS; ++j) {
      const int eo = t + j * CCQP_BLOCK;
      if (eo < E) prod[eo] = v[j] * prod[eo];
    }
  } else {
    double xv[SPMV_SLOTS];
#pragma unroll
    for (int j = 0; j < SPMV_SLOTS; ++j) xv[j] = x[c[j]];
#pragma unroll
    for (int j = 0; j < SPMV_SLOTS; ++j) {
      const int e = t + j * CCQP_BLOCK;
      if (e < E) prod[e] = v[j] * xv[j];
    }
  }
  }
  __syncthreads();
  // phase 2: one row per subgroup out of LDS, canonical order
  for (int64_t i = t / W; i < nrows; i += SUBGROUPS) {
    int beg, end;
    if (rows_in_lds) {
      beg = rstart[i];
      end = rstart[i + 1];
    } else {                                // a window with thousands of empty rows
      beg = (int)(rowptr[rb + i] - nzb);
      end = (int)(rowptr[rb + i + 1] - nzb);
    }
    double acc[K];
#pragma unroll
    for (int a = 0; a < K; ++a) acc[a] = 0.0;
    for (int e = beg + l; e < end; e += 64) {
#pragma unroll
      for (int a = 0; a < K; ++a)
        if (e + a * W < end) acc[a] = acc[a] + prod[e + a * W];
    }
    // butterfly steps 32 .. W: both partners live in this lane
#pragma unroll
    for (int s = K / 2; s >= 1; s >>= 1) {
#pragma unroll
      for (int a = 0; a < s; ++a) acc[a] = acc[a] + acc[a + s];
    }
    // butterfly steps W/2 .. 1 across the subgroup
    const double sum = spmv_subgroup_sum<W>(acc[0]);
    if (l == 0) {
      if (rows_in_lds) {
        if (end > beg) prod[beg] = sum;     // parked in the row's own first slot (nobody reads it any more)
      } else {
        y[rb + i] = sum;
        if (EPI) g[rb + i] = sum + b[rb + i];
      }
    }
  }
  if (rows_in_lds) {
    // phase 3: coalesced output -- lane 0 of every W-lane subgroup storing 8 bytes costs ~20 % of the kernel
    // at 9 entries per row
    __syncthreads();
    for (int i = t; i < (int)nrows; i += CCQP_BLOCK) {
      const int beg = rstart[i], end = rstart[i + 1];
      const double sum = end > beg ? prod[beg] : 0.0;
      y[rb + i] = sum;
      if (EPI) g[rb + i] = sum + b[rb + i];
    }
  }
}

// rows longer than SPMV_CAP entries: one wave per row, straight from global memory, same canonical order
template <bool EPI>
__global__ __launch_bounds__(CCQP_BLOCK) void spmv_long_rows_kernel(const int64_t* __restrict__ rowptr,
                                                                    const int32_t* __restrict__ col,
                                                                    const double* __restrict__ val,
                                                                    const int64_t* __restrict__ nlong,
                                                                    const int64_t* __restrict__ long_rows,
                                                                    const double* __restrict__ x,
                                                                    const double* __restrict__ b,
                                                                    double* __restrict__ y, double* __restrict__ g,
                                                                    const double* __restrict__ skip) {
  if (skip != nullptr && *skip != 0.0) return;
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (CCQP_BLOCK / 64);
  const int64_t n = *nlong;
  for (int64_t i = (int64_t)blockIdx.x * (CCQP_BLOCK / 64) + (threadIdx.x >> 6); i < n; i += nwaves) {
    const int64_t r = long_rows[i];
    const int64_t beg = rowptr[r], end = rowptr[r + 1];
    double acc = 0.0;
    for (int64_t base = beg; base < end; base += 4 * 64) {
      double v[4], xv[4];
      int32_t c[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t e = base + j * 64 + lane;
        const int64_t es = e < end ? e : end - 1;
        v[j] = __builtin_nontemporal_load(val + es);
        c[j] = __builtin_nontemporal_load(col + es);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[j] = x[c[j]];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double p = v[j] * xv[j];
        if (base + j * 64 + lane < end) acc = acc + p;
      }
    }
    const double sum = ccqp_wave_sum(acc);
    if (lane == 0) {
      y[r] = sum;
      if (EPI) g[r] = sum + b[r];
    }
  }
}

// violations of the CSR invariants, counted into a scalar slot (0 = valid)
__global__ __launch_bounds__(CCQP_BLOCK) void csr_check_rows_kernel(const int64_t* __restrict__ rowptr, int64_t rows,
                                                                    int64_t nnz, double* __restrict__ slot) {
  const int64_t r = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x;
  int bad = 0;
  if (r < rows) {
    const int64_t a = rowptr[r], c = rowptr[r + 1];
    bad = (a < 0) || (c < a) || (c > nnz);
  }
  if (r == 0) bad |= (rowptr[0] != 0) || (rowptr[rows] != nnz);
  if (bad) atomicAdd(slot, 1.0);
}

__global__ __launch_bounds__(CCQP_BLOCK) void csr_check_cols_kernel(const int32_t* __restrict__ col, int64_t nnz,
                                                                    int64_t cols, double* __restrict__ slot) {
  const int64_t stride = (int64_t)gridDim.x * CCQP_BLOCK;
  int bad = 0;
  for (int64_t e = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; e < nnz; e += stride) {
    const int32_t c = col[e];
    bad += (c < 0) || ((int64_t)c >= cols);
  }
  if (bad) atomicAdd(slot, (double)bad);
}

static int spmv_pick_width(int64_t rows, int64_t nnz, int width) {
  if (width == 4 || width == 8 || width == 16 || width == 32 || width == 64) return width;
  const double mean = rows > 0 ? (double)nnz / (double)rows : 0.0;
  // phase 2 wants (rows of a window) x W ~ the 256 threads: W ~ mean / 7 (measured: 4 at 9 entries per
  // row, 8-16 at 33, 16 at 129, flat beyond)
  if (mean <= 12.0) return 4;
  if (mean <= 48.0) return 8;
  if (mean <= 192.0) return 16;
  if (mean <= 512.0) return 32;
  return 64;
}

template <int W>
static void spmv_launch(unsigned grid, const int64_t* rowptr, const int32_t* col, const double* val,
                        const int64_t* plan, const double* x, const double* b, double* y, double* g, int64_t nnz,
                        hipStream_t s, const double* skip) {
  const int64_t nwin = spmv_windows(nnz);
  const int32_t* ucol = reinterpret_cast<const int32_t*>(plan + spmv_off_ucol(nwin));
  const uint16_t* lcol = reinterpret_cast<const uint16_t*>(plan + spmv_off_lcol(nwin));
  if (g)
    hipLaunchKernelGGL((spmv_stream_kernel<W, true>), dim3(grid), dim3(CCQP_BLOCK), 0, s, rowptr, col, val, plan,
                       ucol, lcol, x, b, y, g, nnz - 1, skip);
  else
    hipLaunchKernelGGL((spmv_stream_kernel<W, false>), dim3(grid), dim3(CCQP_BLOCK), 0, s, rowptr, col, val, plan,
                       ucol, lcol, x, b, y, g, nnz - 1, skip);
}

static int spmv_csr_impl(ccqp_ctx* ctx, const int64_t* rowptr, const int32_t* col, const double* val,
                         const int64_t* plan, const int64_t* summary, const double* x, const double* b, double* y,
                         double* g, int64_t rows, int64_t cols, int64_t nnz, int width, void* stream,
                         const double* skip) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(rows >= 0 && cols >= 0 && nnz >= 0, "negative size");
  CCQP_REQUIRE(cols <= INT32_MAX, "cols exceeds the 32-bit column index");
  CCQP_REQUIRE(rows == 0 || (rowptr != nullptr && y != nullptr && plan != nullptr), "NULL rowptr / y / plan");
  CCQP_REQUIRE(nnz == 0 || (col != nullptr && val != nullptr && x != nullptr), "NULL col / val / x");
  CCQP_REQUIRE(y != x, "y must not alias x");
  CCQP_REQUIRE(g == nullptr || b != nullptr, "g requested but b is NULL");
  CCQP_REQUIRE(g == nullptr || g != x, "g must not alias x");
  CCQP_REQUIRE(width == 0 || width == 4 || width == 8 || width == 16 || width == 32 || width == 64,
               "width must be 0 (auto), 4, 8, 16, 32 or 64");
  if (rows == 0) return CCQP_OK;
  const int64_t nwin = spmv_windows(nnz);
  CCQP_REQUIRE(nwin <= (int64_t)INT32_MAX, "matrix too large for one launch");
  hipStream_t s = (hipStream_t)stream;
  if (nnz == 0) {                            // all rows empty: nothing to stream
    CCQP_HIP(hipMemsetAsync(y, 0, (size_t)rows * sizeof(double), s));
    if (g) CCQP_HIP(hipMemcpyAsync(g, b, (size_t)rows * sizeof(double), hipMemcpyDeviceToDevice, s));
    return CCQP_OK;
  }
  const unsigned grid = (unsigned)nwin;
  // summary (host copy of the plan's two counters) unknown: assume rows for the wave-per-row launch exist
  const int64_t n_long_rows = summary ? summary[0] : -1;
  switch (spmv_pick_width(rows, nnz, width)) {
    case 4: spmv_launch<4>(grid, rowptr, col, val, plan, x, b, y, g, nnz, s, skip); break;
    case 8: spmv_launch<8>(grid, rowptr, col, val, plan, x, b, y, g, nnz, s, skip); break;
    case 16: spmv_launch<16>(grid, rowptr, col, val, plan, x, b, y, g, nnz, s, skip); break;
    case 32: spmv_launch<32>(grid, rowptr, col, val, plan, x, b, y, g, nnz, s, skip); break;
    default: spmv_launch<64>(grid, rowptr, col, val, plan, x, b, y, g, nnz, s, skip); break;
  }
  if (n_long_rows != 0) {                    // < 0: unknown to the caller, the kernel reads the count
    const int64_t* count = plan + spmv_off_counts(nwin);
    int64_t lg = n_long_rows < 0 ? ctx->num_cu : (n_long_rows + 3) / 4;
    if (lg > (int64_t)ctx->num_cu * 8) lg = (int64_t)ctx->num_cu * 8;
    if (g)
      hipLaunchKernelGGL((spmv_long_rows_kernel<true>), dim3((unsigned)lg), dim3(CCQP_BLOCK), 0, s, rowptr, col, val,
                         count, plan + spmv_off_long(nwin), x, b, y, g, skip);
    else
      hipLaunchKernelGGL((spmv_long_rows_kernel<false>), dim3((unsigned)lg), dim3(CCQP_BLOCK), 0, s, rowptr, col, val,
                         count, plan + spmv_off_long(nwin), x, b, y, g, skip);
  }
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

extern "C" int ccqp_spmv_csr_f64(ccqp_ctx* ctx, const int64_t* rowptr, const int32_t* col, const double* val,
                                 const int64_t* plan, const int64_t* summary, const double* x, const double* b,
                                 double* y, double* g, int64_t rows, int64_t cols, int64_t nnz, int width,
                                 void* stream) {
  return spmv_csr_impl(ctx, rowptr, col, val, plan, summary, x, b, y, g, rows, cols, nnz, width, stream, nullptr);
}

// the same behind the pipeline's "converged" word (ccqp_pipe.hip): a no-op once slab slot 49 is set
extern "C" int ccqp_pipe_spmv_csr_f64(ccqp_ctx* ctx, const int64_t* rowptr, const int32_t* col, const double* val,
                                      const int64_t* plan, const int64_t* summary, const double* x, const double* b,
                                      double* y, double* g, int64_t rows, int64_t cols, int64_t nnz, int width,
                                      void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  return spmv_csr_impl(ctx, rowptr, col, val, plan, summary, x, b, y, g, rows, cols, nnz, width, stream,
                       ctx->slots + 49);
}

extern "C" int ccqp_csr_validate(ccqp_ctx* ctx, const int64_t* rowptr, const int32_t* col, int64_t rows,
                                 int64_t cols, int64_t nnz, int slot, void* stream) {
  CCQP_REQUIRE(ctx != nullptr, "ctx is NULL");
  CCQP_REQUIRE(rows >= 0 && cols >= 0 && nnz >= 0, "negative size");
  CCQP_REQUIRE(ccqp_slot_ok(slot, 1), "slot out of range");
  CCQP_REQUIRE(rowptr != nullptr, "NULL rowptr (rows + 1 entries are required, even for rows == 0)");
  CCQP_REQUIRE(nnz == 0 || col != nullptr, "NULL col");
  hipStream_t s = (hipStream_t)stream;
  CCQP_HIP(hipMemsetAsync(ctx->slots + slot, 0, sizeof(double), s));
  const int64_t rgrid = (rows + 1 + CCQP_BLOCK - 1) / CCQP_BLOCK;
  CCQP_REQUIRE(rgrid <= (int64_t)INT32_MAX, "too many rows for one launch");
  hipLaunchKernelGGL(csr_check_rows_kernel, dim3((unsigned)rgrid), dim3(CCQP_BLOCK), 0, s, rowptr, rows, nnz,
                     ctx->slots + slot);
  if (nnz > 0) {
    int64_t cgrid = (nnz + CCQP_BLOCK - 1) / CCQP_BLOCK;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (cgrid > cap) cgrid = cap;
    hipLaunchKernelGGL(csr_check_cols_kernel, dim3((unsigned)cgrid), dim3(CCQP_BLOCK), 0, s, col, nnz, cols,
                       ctx->slots + slot);
  }
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}
