// Context, error reporting and the scalar slab of libccqp_hip.so.
#include <stdarg.h>

#include "ccqp_internal.h"

static thread_local char g_err[512] = "";

void ccqp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Blocking scalar read without the copy engine and without an interrupt-driven stream wait: one wave
// copies the requested slots into the pinned, coherent host mirror and then raises a sequence flag; the host
// polls the flag (ccqp_scalars_read).  13.5 us -> ~5 us per read on an idle stream, and every solver
// iteration has one to three of them.
__global__ void publish_slots_kernel(const double* __restrict__ slots, double* __restrict__ host, int slot0,
                                     int count, unsigned long long seq) {
  const int i = threadIdx.x;
  if (i < count) host[slot0 + i] = slots[slot0 + i];
  // the wave's stores have reached host memory before the flag does: drain, system-scope release, drain again
  // (ROCm 7.2 can drop the fence's own wait -- guide, Guideline 16 pitfall 12)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (i == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(host + CCQP_NUM_SLOTS), seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- matvec timing hook
int ccqp_mvt_on = 0;
static hipEvent_t* g_mvt_ev = nullptr;
static int g_mvt_cap = 0, g_mvt_n = 0;

void ccqp_mvt_mark(hipStream_t s, int end) {
  if (!ccqp_mvt_on || g_mvt_n >= g_mvt_cap) return;
  (void)hipEventRecord(g_mvt_ev[2 * g_mvt_n + end], s);
  if (end) ++g_mvt_n;
}

// All matvec groups of a process that ask for it go through ONE dedicated stream per device (event-chained to the
// caller's stream): matvecs of different lanes / half-batches then run one after the other at full bandwidth instead of
// overlapping and splitting it, while the small kernels of the other lanes still overlap with them.
// (the host side of this library is driven by one thread per process)
int ccqp_mv_stream_enter(ccqp_ctx* ctx, hipStream_t s, hipStream_t* ms) {
  *ms = s;
  if (!ctx->mv_serialize) return CCQP_OK;
  static hipStream_t mv_streams[64] = {nullptr};
  CCQP_REQUIRE(ctx->device >= 0 && ctx->device < 64, "device index out of range");
  hipStream_t& mv_stream = mv_streams[ctx->device];
  if (!mv_stream) {
    int prev = 0;
    CCQP_HIP(hipGetDevice(&prev));
    CCQP_HIP(hipSetDevice(ctx->device));
    const hipError_t e = hipStreamCreateWithFlags(&mv_stream, hipStreamNonBlocking);
    (void)hipSetDevice(prev);
    CCQP_HIP(e);
  }
  if (mv_stream == s) return CCQP_OK;
  if (!ctx->mv_ev_in) {
    CCQP_HIP(hipEventCreateWithFlags(&ctx->mv_ev_in, hipEventDisableTiming));
    CCQP_HIP(hipEventCreateWithFlags(&ctx->mv_ev_out, hipEventDisableTiming));
  }
  CCQP_HIP(hipEventRecord(ctx->mv_ev_in, s));
  CCQP_HIP(hipStreamWaitEvent(mv_stream, ctx->mv_ev_in, 0));
  *ms = mv_stream;
  return CCQP_OK;
}

int ccqp_mv_stream_leave(ccqp_ctx* ctx, hipStream_t s, hipStream_t ms) {
  if (ms == s) return CCQP_OK;
  CCQP_HIP(hipEventRecord(ctx->mv_ev_out, ms));
  CCQP_HIP(hipStreamWaitEvent(s, ctx->mv_ev_out, 0));
  return CCQP_OK;
}

extern "C" int ccqp_mv_timing_begin(int max_pairs) {
  CCQP_REQUIRE(max_pairs >= 1 && max_pairs <= (1 << 20), "max_pairs outside [1, 2^20]");
  CCQP_REQUIRE(!ccqp_mvt_on, "a timing session is already open");
  if (max_pairs > g_mvt_cap) {
    hipEvent_t* ev = new hipEvent_t[2 * (size_t)max_pairs];
    for (int i = 0; i < 2 * g_mvt_cap; ++i) ev[i] = g_mvt_ev[i];
    for (int i = 2 * g_mvt_cap; i < 2 * max_pairs; ++i) {
      const hipError_t e = hipEventCreate(&ev[i]);
      if (e != hipSuccess) {                      // the session does not open; the old table stays as it was
        for (int j = 2 * g_mvt_cap; j < i; ++j) (void)hipEventDestroy(ev[j]);
        delete[] ev;
        ccqp_set_error("%s: hipEventCreate failed: %s", __func__, hipGetErrorString(e));
        return CCQP_ERR_HIP;
      }
    }
    delete[] g_mvt_ev;
    g_mvt_ev = ev;
    g_mvt_cap = max_pairs;
  }
  g_mvt_n = 0;
  ccqp_mvt_on = 1;
  return CCQP_OK;
}

extern "C" int ccqp_mv_timing_end(double* ms_out, int cap, int* count) {
  CCQP_REQUIRE(ccqp_mvt_on, "no timing session is open");
  ccqp_mvt_on = 0;
  if (count) *count = g_mvt_n;
  if (g_mvt_n == 0) return CCQP_OK;
  CCQP_HIP(hipEventSynchronize(g_mvt_ev[2 * (g_mvt_n - 1) + 1]));
  for (int i = 0; i < g_mvt_n && i < cap && ms_out; ++i) {
    float ms = 0.f;
    CCQP_HIP(hipEventElapsedTime(&ms, g_mvt_ev[2 * i], g_mvt_ev[2 * i + 1]));
    ms_out[i] = (double)ms;
  }
  return CCQP_OK;
}

extern "C" int ccqp_abi_version(void) { return CCQP_ABI_VERSION; }

extern "C" const char* ccqp_last_error(void) { return g_err; }

extern "C" int ccqp_ctx_create(int device, ccqp_ctx** out) {
  CCQP_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  int ndev = 0;
  CCQP_HIP(hipGetDeviceCount(&ndev));
  CCQP_REQUIRE(device >= 0 && device < ndev, "no such HIP device");
  int prev = 0;
  CCQP_HIP(hipGetDevice(&prev));
  CCQP_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  CCQP_HIP(hipGetDeviceProperties(&prop, device));
  ccqp_ctx* c = new ccqp_ctx();
  c->device = device;
  c->num_cu = prop.multiProcessorCount;
  strncpy(c->arch, prop.gcnArchName, sizeof(c->arch) - 1);
  c->arch[sizeof(c->arch) - 1] = 0;
  c->slots = nullptr;
  c->partials = nullptr;
  c->host_slots = nullptr;
  c->host_slots_dev = nullptr;
  c->read_seq = 0;
  c->read_event = nullptr;
  c->batch = nullptr;
  c->mv_serialize = 0;
  c->mv_ev_in = nullptr;
  c->mv_ev_out = nullptr;
  c->gather_dev = nullptr;
  c->gather_host = nullptr;
  c->xchg_host = nullptr;
  c->xchg_host_dev = nullptr;
  c->xchg_doubles = 0;
  for (int k = 0; k < CCQP_OPT_COUNT; ++k) c->opt[k] = 0;
  c->lds_nraised = 0;
  hipError_t e = hipMalloc(&c->slots, CCQP_NUM_SLOTS * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&c->partials, CCQP_PARTIALS_DOUBLES * sizeof(double));
  if (e == hipSuccess) e = hipMemset(c->partials, 0, CCQP_PARTIALS_DOUBLES * sizeof(double));
  if (e == hipSuccess)
    e = hipHostMalloc(&c->host_slots, (CCQP_NUM_SLOTS + 8) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) memset(c->host_slots, 0, (CCQP_NUM_SLOTS + 8) * sizeof(double));
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&c->host_slots_dev, c->host_slots, 0);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->read_event, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMemset(c->slots, 0, CCQP_NUM_SLOTS * sizeof(double));
  if (e != hipSuccess) {
    ccqp_set_error("ccqp_ctx_create: allocation failed: %s", hipGetErrorString(e));
    if (c->slots) (void)hipFree(c->slots);
    if (c->partials) (void)hipFree(c->partials);
    if (c->host_slots) (void)hipHostFree(c->host_slots);
    if (c->read_event) (void)hipEventDestroy(c->read_event);
    delete c;
    (void)hipSetDevice(prev);
    return CCQP_ERR_HIP;
  }
  (void)hipSetDevice(prev);
  *out = c;
  return CCQP_OK;
}

extern "C" int ccqp_ctx_destroy(ccqp_ctx* c) {
  if (!c) return CCQP_OK;
  (void)hipFree(c->slots);
  (void)hipFree(c->partials);
  (void)hipHostFree(c->host_slots);
  (void)hipEventDestroy(c->read_event);
  ccqp_batch_free(c);
  if (c->mv_ev_in) (void)hipEventDestroy(c->mv_ev_in);
  if (c->mv_ev_out) (void)hipEventDestroy(c->mv_ev_out);
  if (c->gather_dev) (void)hipFree(c->gather_dev);
  if (c->gather_host) (void)hipHostFree(c->gather_host);
  if (c->xchg_host) (void)hipHostFree(c->xchg_host);
  delete c;
  return CCQP_OK;
}

extern "C" int ccqp_ctx_set_option(ccqp_ctx* c, int key, int64_t value) {
  CCQP_REQUIRE(c != nullptr, "ctx is NULL");
  CCQP_REQUIRE(key >= 0 && key < CCQP_OPT_COUNT, "no such option");
  bool ok = value >= 0;
  switch (key) {
    case CCQP_OPT_GEMV2_WPB: ok = value >= 0 && value <= 8; break;
    case CCQP_OPT_GEMV2_SHAPE:
      ok = value == 0 || value == 44 || value == 43 || value == 46 || value == 48 || value == 82 || value == 83 || value == 84;
      break;
    case CCQP_OPT_GEMV2_GROUP: ok = value == 0 || (value >= 64 && value <= 512 && value % 16 == 0); break;
    case CCQP_OPT_BMV3_SHAPE: ok = value == 0 || value == 824 || value == 833 || value == 444 || value == 434; break;
    case CCQP_OPT_LOOKAHEAD_MIN_N:
    case CCQP_OPT_MPRGP_LOOKAHEAD_MIN_N:
    case CCQP_OPT_PERS1_MIN_N: ok = value == 0 || value >= 128; break;
    case CCQP_OPT_PERS_MIN_MIB: ok = value >= 0 && value <= 1048576; break;
    case CCQP_OPT_MPRGP_LOOKAHEAD:
    case CCQP_OPT_APGD_LOOKAHEAD:
    case CCQP_OPT_BPIPE_STEP_IN_TAIL:
    case CCQP_OPT_BMV_STAGGER: ok = value == 0 || value == 1; break;
    case CCQP_OPT_MPRGP_FUSED_TAIL:
    case CCQP_OPT_BMV_LOADS:
    case CCQP_OPT_APGD_UPDATE_IN_DECISION:
    case CCQP_OPT_SMALL_SHARING: ok = value >= 0 && value <= 2; break;
    default: break;
  }
  CCQP_REQUIRE(ok, "value outside the option's range");
  c->opt[key] = value;
  return CCQP_OK;
}

extern "C" int ccqp_ctx_get_option(ccqp_ctx* c, int key, int64_t* value) {
  CCQP_REQUIRE(c != nullptr && value != nullptr, "NULL argument");
  CCQP_REQUIRE(key >= 0 && key < CCQP_OPT_COUNT, "no such option");
  *value = c->opt[key];
  return CCQP_OK;
}

int ccqp_lds_raise(ccqp_ctx* c, const void* fn, size_t bytes) {
  CCQP_REQUIRE(c != nullptr && fn != nullptr, "NULL argument");
  if (bytes > 160 * 1024) {
    ccqp_set_error("ccqp_lds_raise: %zu bytes of LDS asked for, a compute unit has 160 KiB", bytes);
    return CCQP_ERR_UNSUPPORTED;
  }
  int at = -1;
  for (int i = 0; i < c->lds_nraised; ++i)
    if (c->lds_raised[i].fn == fn) {
      if (c->lds_raised[i].bytes >= bytes) return CCQP_OK;
      at = i;
      break;
    }
  if (bytes <= 64 * 1024 && at < 0) return CCQP_OK;       // (within the default limit: nothing to ask for)
  // the attribute belongs to the device that is current: the caller keeps the context's device current (ccqp_hip.h)
  int dev = -1;
  CCQP_HIP(hipGetDevice(&dev));
  if (dev != c->device) {
    ccqp_set_error("ccqp_lds_raise: device %d is current, the context belongs to device %d", dev, c->device);
    return CCQP_ERR_INVALID;
  }
  CCQP_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  if (at < 0) {
    if (c->lds_nraised == CCQP_LDS_KERNELS) return CCQP_OK;   // (table full: the attribute is set, only not remembered)
    at = c->lds_nraised++;
    c->lds_raised[at].fn = fn;
  }
  c->lds_raised[at].bytes = bytes;
  return CCQP_OK;
}

extern "C" int ccqp_ctx_info(ccqp_ctx* c, int* device, int* num_cu, char* arch_buf, int buflen) {
  CCQP_REQUIRE(c != nullptr, "ctx is NULL");
  if (device) *device = c->device;
  if (num_cu) *num_cu = c->num_cu;
  if (arch_buf && buflen > 0) {
    strncpy(arch_buf, c->arch, (size_t)buflen - 1);
    arch_buf[buflen - 1] = 0;
  }
  return CCQP_OK;
}

extern "C" int ccqp_scalars_read(ccqp_ctx* c, int slot0, int count, double* host_out, void* stream) {
  CCQP_REQUIRE(c != nullptr, "ctx is NULL");
  CCQP_REQUIRE(host_out != nullptr, "host_out is NULL");
  CCQP_REQUIRE(ccqp_slot_ok(slot0, count), "slot range outside the slab");
  hipStream_t s = (hipStream_t)stream;
  CCQP_REQUIRE(count <= 64, "at most 64 slots per read");
  const unsigned long long seq = ++c->read_seq;
  hipLaunchKernelGGL(publish_slots_kernel, dim3(1), dim3(64), 0, s, c->slots, c->host_slots_dev, slot0, count, seq);
  CCQP_LAUNCH_CHECK();
  // poll briefly (the common case at small and medium n: the GPU is a few microseconds behind), then
  // fall back to a sleeping wait -- the flag is certainly up once the stream has drained
  volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(c->host_slots + CCQP_NUM_SLOTS);
  bool up = false;
  for (int spin = 0; spin < 20000; ++spin) {
    if (*flag == seq) {
      up = true;
      break;
    }
    __builtin_ia32_pause();
  }
  if (!up) {
    CCQP_HIP(hipStreamSynchronize(s));
    CCQP_REQUIRE(*flag == seq, "published scalars did not arrive");
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (int i = 0; i < count; ++i) host_out[i] = c->host_slots[slot0 + i];
  return CCQP_OK;
}

// Blocking read of `count` <= 64 doubles at ANY device address through the context's pinned mirror (the same publish +
// poll as ccqp_scalars_read: ~5 us on a drained stream instead of the ~25 us of a pageable device-to-host copy): the
// result words of a lock-step problem's own slab.
extern "C" int ccqp_read_f64(ccqp_ctx* c, const double* dev, int count, double* host_out, void* stream) {
  CCQP_REQUIRE(c != nullptr && dev != nullptr && host_out != nullptr, "NULL argument");
  CCQP_REQUIRE(count >= 1 && count <= CCQP_NUM_SLOTS, "between 1 and 64 doubles per read");
  hipStream_t s = (hipStream_t)stream;
  const unsigned long long seq = ++c->read_seq;
  hipLaunchKernelGGL(publish_slots_kernel, dim3(1), dim3(64), 0, s, dev, c->host_slots_dev, 0, count, seq);
  CCQP_LAUNCH_CHECK();
  volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(c->host_slots + CCQP_NUM_SLOTS);
  bool up = false;
  for (int spin = 0; spin < 20000; ++spin) {
    if (*flag == seq) {
      up = true;
      break;
    }
    __builtin_ia32_pause();
  }
  if (!up) {
    CCQP_HIP(hipStreamSynchronize(s));
    CCQP_REQUIRE(*flag == seq, "published words did not arrive");
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (int i = 0; i < count; ++i) host_out[i] = c->host_slots[i];
  return CCQP_OK;
}

extern "C" int ccqp_scalars_read_async(ccqp_ctx* c, int slot0, int count, void* stream) {
  CCQP_REQUIRE(c != nullptr, "ctx is NULL");
  CCQP_REQUIRE(ccqp_slot_ok(slot0, count), "slot range outside the slab");
  if (count > 0)
    CCQP_HIP(hipMemcpyAsync(c->host_slots + slot0, c->slots + slot0, (size_t)count * sizeof(double),
                            hipMemcpyDeviceToHost, (hipStream_t)stream));
  CCQP_HIP(hipEventRecord(c->read_event, (hipStream_t)stream));
  return CCQP_OK;
}

extern "C" int ccqp_scalars_host(ccqp_ctx* c, int slot0, int count, double* host_out) {
  CCQP_REQUIRE(c != nullptr, "ctx is NULL");
  CCQP_REQUIRE(host_out != nullptr, "host_out is NULL");
  CCQP_REQUIRE(ccqp_slot_ok(slot0, count), "slot range outside the slab");
  CCQP_HIP(hipEventSynchronize(c->read_event));   // returns at once if the copy has completed
  for (int i = 0; i < count; ++i) host_out[i] = c->host_slots[slot0 + i];
  return CCQP_OK;
}

extern "C" int ccqp_scalars_devptr(ccqp_ctx* c, double** out) {
  CCQP_REQUIRE(c != nullptr && out != nullptr, "NULL argument");
  *out = c->slots;
  return CCQP_OK;
}
