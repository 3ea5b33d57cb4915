// Row-sharded multi-GPU matvec behind the C ABI: one process per GPU, RCCL over xGMI.
//
// rank r keeps rows [r*per, r*per + rows_r) of A (per = ceil(n / ranks)), every n-vector is
// replicated, ONE exchange per matvec reassembles y.  The Python shells do this with torch.distributed (backend
// "nccl" = RCCL); this file is the same step for callers that have no torch -- and an opt-in for those that do
// (VectorExchange(direct=True)): the collective is issued by RCCL itself on the CALLER's stream, right behind the
// GEMV, with no staging copy (in-place all-gather: each rank's slab already sits at its offset of y), no second
// stream and no event hand-off.  The reference has nothing to cite here (no collectives); the call it
// replaces is still `A.dot(v)` (solvers.py:133,148,275,290,625,641,922,938,1086,1100,...).
//
// RCCL is bound at run time (dlopen): a process that already holds a copy (PyTorch ships one) must use THAT copy, and
// libccqp_hip.so stays loadable on a box without RCCL -- the ccqp_comm_* calls then fail with CCQP_ERR_UNSUPPORTED.
// Because each row sum is produced by one wave in an order that depends on the row alone, the gathered y is bitwise
// the single-GPU y for every rank count.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "ccqp_internal.h"

int ccqp_gemv_launch(ccqp_ctx* ctx, const double* A, int64_t lda, const double* x, const double* b, double* y,
                     double* g, int64_t rows, int64_t cols, int variant, hipStream_t s, const double* skip);

// ---- k products per exchange: rank r's slab of product j sits at gathered[(r k + j) per ...) (rows_r <= per valid words).
// One launch scatters the gathered slabs into the k replicated vectors and forms g_j = y_j + b where asked -- the same single
// add as the matvec's fused epilogue / the axpby it replaces, hence the same bits.
struct unpack_io {
  double* y[3];
  double* g[3];
};
__global__ __launch_bounds__(CCQP_BLOCK) void unpack_gathered_kernel(const double* __restrict__ gathered, int k, int64_t per,
                                                                     int64_t n, const unpack_io io,
                                                                     const double* __restrict__ b) {
  for (int64_t i = (int64_t)blockIdx.x * CCQP_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CCQP_BLOCK) {
    const int64_t r = i / per, off = i - r * per;
    const double bi = b ? b[i] : 0.0;
    for (int j = 0; j < k; ++j) {
      const double v = gathered[(r * k + j) * per + off];
      io.y[j][i] = v;
      if (io.g[j]) io.g[j][i] = v + bi;
    }
  }
}

struct ccqp_comm {
  ncclComm_t nccl;
  int nranks, rank, device;
};

namespace {
struct rccl_api {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
} g_rccl;

bool rccl_load() {
  if (g_rccl.handle) return true;
  // a copy already mapped into the process first (PyTorch's librccl.so has no soname; ROCm's is librccl.so.1)
  const char* names[] = {"librccl.so", "librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  for (const char* n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return false;
  rccl_api a;
  a.handle = h;
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
  a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  a.GetVersion = reinterpret_cast<decltype(a.GetVersion)>(dlsym(h, "ncclGetVersion"));
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.AllReduce || !a.GetErrorString)
    return false;
  g_rccl = a;
  return true;
}
}  // namespace

#define CCQP_NEED_RCCL()                                                                            \
  do {                                                                                              \
    if (!rccl_load()) {                                                                             \
      ccqp_set_error("%s: no RCCL library (librccl.so / librccl.so.1) could be loaded", __func__);  \
      return CCQP_ERR_UNSUPPORTED;                                                                  \
    }                                                                                               \
  } while (0)

#define CCQP_NCCL(call)                                                                         \
  do {                                                                                          \
    ncclResult_t r_ = (call);                                                                   \
    if (r_ != ncclSuccess) {                                                                    \
      ccqp_set_error("%s: %s failed: %s", __func__, #call, g_rccl.GetErrorString(r_));          \
      return CCQP_ERR_HIP;                                                                      \
    }                                                                                           \
  } while (0)

// 0 if RCCL is not available, else its version number (e.g. 22606)
extern "C" int ccqp_comm_rccl_version(void) {
  if (!rccl_load()) return 0;
  int v = 0;
  if (g_rccl.GetVersion && g_rccl.GetVersion(&v) == ncclSuccess) return v;
  return 1;
}

extern "C" int ccqp_comm_unique_id(unsigned char* id128) {
  CCQP_REQUIRE(id128 != nullptr, "id buffer is NULL");
  CCQP_NEED_RCCL();
  ncclUniqueId id;
  CCQP_NCCL(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == CCQP_COMM_ID_BYTES, "ncclUniqueId size changed");
  memcpy(id128, &id, sizeof(id));
  return CCQP_OK;
}

extern "C" int ccqp_comm_create(ccqp_ctx* ctx, int nranks, int rank, const unsigned char* id128, ccqp_comm** out) {
  CCQP_REQUIRE(ctx != nullptr && id128 != nullptr && out != nullptr, "NULL argument");
  CCQP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank outside [0, nranks)");
  *out = nullptr;
  CCQP_NEED_RCCL();
  int prev = 0;
  CCQP_HIP(hipGetDevice(&prev));
  CCQP_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t r = g_rccl.CommInitRank(&c, nranks, id, rank);
  (void)hipSetDevice(prev);
  if (r != ncclSuccess) {
    ccqp_set_error("ccqp_comm_create: ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    return CCQP_ERR_HIP;
  }
  ccqp_comm* k = new ccqp_comm();
  k->nccl = c;
  k->nranks = nranks;
  k->rank = rank;
  k->device = ctx->device;
  *out = k;
  return CCQP_OK;
}

extern "C" int ccqp_comm_destroy(ccqp_comm* comm) {
  if (!comm) return CCQP_OK;
  if (g_rccl.handle && comm->nccl) (void)g_rccl.CommDestroy(comm->nccl);
  delete comm;
  return CCQP_OK;
}

extern "C" int ccqp_comm_info(const ccqp_comm* comm, int* nranks, int* rank) {
  CCQP_REQUIRE(comm != nullptr, "comm is NULL");
  if (nranks) *nranks = comm->nranks;
  if (rank) *rank = comm->rank;
  return CCQP_OK;
}

// y holds `count` doubles per rank, rank r's at y[r*count ...): in-place all-gather on `stream`
extern "C" int ccqp_comm_allgather_f64(ccqp_comm* comm, double* y, int64_t count, void* stream) {
  CCQP_REQUIRE(comm != nullptr, "comm is NULL");
  CCQP_REQUIRE(count >= 0, "negative count");
  if (count == 0) return CCQP_OK;
  CCQP_REQUIRE(y != nullptr, "NULL buffer");
  CCQP_NEED_RCCL();
  CCQP_NCCL(g_rccl.AllGather(y + (int64_t)comm->rank * count, y, (size_t)count, ncclFloat64, comm->nccl,
                             (hipStream_t)stream));
  return CCQP_OK;
}

extern "C" int ccqp_comm_allreduce_sum_f64(ccqp_comm* comm, double* buf, int64_t count, void* stream) {
  CCQP_REQUIRE(comm != nullptr, "comm is NULL");
  CCQP_REQUIRE(count >= 0, "negative count");
  if (count == 0) return CCQP_OK;
  CCQP_REQUIRE(buf != nullptr, "NULL buffer");
  CCQP_NEED_RCCL();
  CCQP_NCCL(g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat64, ncclSum, comm->nccl, (hipStream_t)stream));
  return CCQP_OK;
}

// One row-sharded matvec: y[rank*per + i] = A_local[i, :] . x for i < rows_local, then the exchange, all on `stream`.
// y must hold per * nranks doubles (rows_local <= per; only the last non-empty rank may be short, so the first
// `cols`-many... the first n entries ARE the assembled vector; the padding behind a short rank's rows is zeroed).
// mode 0: in-place all-gather of `per` doubles per rank;
// mode 1: the rest of y zeroed, then an all-reduce (north_star's wording; adds exact zeros, `nranks` times the bytes).
extern "C" int ccqp_shard_gemv_allgather_f64(ccqp_ctx* ctx, ccqp_comm* comm, const double* A_local, int64_t lda,
                                             const double* x, double* y, int64_t rows_local, int64_t per,
                                             int64_t cols, int variant, int mode, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && comm != nullptr, "NULL context / communicator");
  CCQP_REQUIRE(rows_local >= 0 && per >= rows_local && cols >= 0 && lda >= cols, "bad shape");
  CCQP_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (all-gather) or 1 (padded all-reduce)");
  CCQP_REQUIRE(y != nullptr && y != x, "NULL y / y aliases x");
  CCQP_REQUIRE(rows_local == 0 || (A_local != nullptr && x != nullptr), "NULL matrix / x");
  hipStream_t s = (hipStream_t)stream;
  double* mine = y + (int64_t)comm->rank * per;
  if (rows_local > 0) {
    int rc = ccqp_gemv_launch(ctx, A_local, lda, x, nullptr, mine, nullptr, rows_local, cols, variant, s, nullptr);
    if (rc) return rc;
    CCQP_LAUNCH_CHECK();
  }
  // a short last rank: the rest of its slab is padding -- zeroed, so that no rank ever receives uninitialised words
  // beyond the assembled vector (they lie at y[n ...) of every rank after the exchange)
  if (mode == 0 && rows_local < per)
    CCQP_HIP(hipMemsetAsync(mine + rows_local, 0, (size_t)(per - rows_local) * sizeof(double), s));
  if (comm->nranks == 1) return CCQP_OK;
  if (mode == 0) return ccqp_comm_allgather_f64(comm, y, per, stream);
  const int64_t total = per * comm->nranks, lo = (int64_t)comm->rank * per, hi = lo + rows_local;
  if (lo > 0) CCQP_HIP(hipMemsetAsync(y, 0, (size_t)lo * sizeof(double), s));
  if (hi < total) CCQP_HIP(hipMemsetAsync(y + hi, 0, (size_t)(total - hi) * sizeof(double), s));
  return ccqp_comm_allreduce_sum_f64(comm, y, total, stream);
}

extern "C" int ccqp_unpack_gathered_f64(ccqp_ctx* ctx, const double* gathered, int k, int64_t per, int64_t n,
                                        double* const* y, const double* b, double* const* g, void* stream) {
  CCQP_REQUIRE(ctx != nullptr && gathered != nullptr && y != nullptr, "NULL argument");
  CCQP_REQUIRE(k >= 1 && k <= 3 && per >= 1 && n >= 0, "bad shape");
  unpack_io io;
  for (int j = 0; j < 3; ++j) {
    io.y[j] = j < k ? y[j] : nullptr;
    io.g[j] = j < k && g ? g[j] : nullptr;
    CCQP_REQUIRE(j >= k || y[j] != nullptr, "NULL output vector");
    CCQP_REQUIRE(io.g[j] == nullptr || b != nullptr, "g requested but b is NULL");
  }
  if (n == 0) return CCQP_OK;
  int64_t grid = (n + CCQP_BLOCK - 1) / CCQP_BLOCK;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(unpack_gathered_kernel, dim3((unsigned)grid), dim3(CCQP_BLOCK), 0, (hipStream_t)stream, gathered, k, per, n,
                     io, b);
  CCQP_LAUNCH_CHECK();
  return CCQP_OK;
}

// k <= 3 row-sharded products with ONE exchange: the local k-vector pass (ccqp_gemvk_f64: one pass over the row block
// where the pair kernel applies) writes its slabs straight into this rank's part of `work`, one in-place all-gather of
// k * per doubles per rank, one unpack launch that also forms g_j = y_j + b.
extern "C" int ccqp_shard_gemvk_allgather_f64(ccqp_ctx* ctx, ccqp_comm* comm, const double* A_local, int64_t lda, int k,
                                              const double* const* x, const double* b, double* const* y, double* const* g,
                                              int64_t rows_local, int64_t per, int64_t cols, int variant, double* work,
                                              void* stream) {
  CCQP_REQUIRE(ctx != nullptr && comm != nullptr && x != nullptr && y != nullptr && work != nullptr, "NULL argument");
  CCQP_REQUIRE(k >= 1 && k <= 3, "k outside [1, 3]");
  CCQP_REQUIRE(rows_local >= 0 && per >= rows_local && per >= 1 && cols >= 0 && lda >= cols, "bad shape");
  CCQP_REQUIRE(cols <= per * comm->nranks, "the row blocks do not cover the vector");
  double* mine = work + (int64_t)comm->rank * k * per;
  double* slab[3] = {mine, mine + per, mine + 2 * per};
  double* none[3] = {nullptr, nullptr, nullptr};
  if (rows_local > 0) {
    const int rc = ccqp_gemvk_f64(ctx, A_local, lda, k, x, nullptr, slab, none, rows_local, cols, variant, stream);
    if (rc) return rc;
  }
  if (comm->nranks > 1) {
    const int rc = ccqp_comm_allgather_f64(comm, work, (int64_t)k * per, stream);
    if (rc) return rc;
  }
  return ccqp_unpack_gathered_f64(ctx, work, k, per, cols, y, b, g, stream);
}

extern "C" int ccqp_shard_gemv2_allgather_f64(ccqp_ctx* ctx, ccqp_comm* comm, const double* A_local, int64_t lda,
                                              const double* x1, const double* x2, const double* b, double* y1, double* g1,
                                              double* y2, double* g2, int64_t rows_local, int64_t per, int64_t cols,
                                              int variant, double* work, void* stream) {
  const double* x[3] = {x1, x2, nullptr};
  double* y[3] = {y1, y2, nullptr};
  double* g[3] = {g1, g2, nullptr};
  return ccqp_shard_gemvk_allgather_f64(ctx, comm, A_local, lda, 2, x, b, y, g, rows_local, per, cols, variant, work, stream);
}
