// tools/readbw.py: what read bandwidth does this MI355X give a kernel that does nothing but stream 128 GiB?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef double v2 __attribute__((ext_vector_type(2)));
// each wave sums `seg` doubles starting at base + wave*seg; 16 B per lane per step, U steps in flight
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const double* __restrict__ a, int64_t seg, int64_t nseg, double* out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  double acc = 0.0;
  for (int64_t s = wave; s < nseg; s += nw) {
    const v2* p = reinterpret_cast<const v2*>(a + s * seg) + lane;
    const int64_t steps = seg / 128;
    for (int64_t k = 0; k < steps; k += U) {
      v2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + (k + u) * 64) : p[(k + u) * 64];
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
  }
  if (acc == 123.456) out[wave] = acc;   // never true for this data: keeps the loads alive
}
extern "C" int readbw_launch(const double* a, int64_t seg, int64_t nseg, double* out, int grid, int U, int nt, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (U == 1 && nt) hipLaunchKernelGGL((read_kernel<1, true>), dim3(grid), dim3(256), 0, s, a, seg, nseg, out);
  else if (U == 2 && nt) hipLaunchKernelGGL((read_kernel<2, true>), dim3(grid), dim3(256), 0, s, a, seg, nseg, out);
  else if (U == 4 && nt) hipLaunchKernelGGL((read_kernel<4, true>), dim3(grid), dim3(256), 0, s, a, seg, nseg, out);
  else if (U == 8 && nt) hipLaunchKernelGGL((read_kernel<8, true>), dim3(grid), dim3(256), 0, s, a, seg, nseg, out);
  else if (U == 2) hipLaunchKernelGGL((read_kernel<2, false>), dim3(grid), dim3(256), 0, s, a, seg, nseg, out);
  else return 1;
  return hipGetLastError() != hipSuccess;
}

// ---- the same question for LDS-DMA loads (global_load_lds_dwordx4: no VGPR destination): a persistent grid whose waves
// each stream R consecutive segments at a time through a private ring of S slots (the matvec's LDS form without x, the
// LDS reads, the fma chains and the butterflies); nothing consumes the data
// blocked = 0: wave w takes tasks w, w + nw, ... (the machine reads a moving band of nw * R segments); 1: wave w takes
// the contiguous run of tasks [w * mytasks, (w + 1) * mytasks) (nw streams spread evenly over the whole array)
template <int R, int S>
__global__ void read_lds_kernel(const double* __restrict__ a, int64_t seg, int64_t nseg, int blocked) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  char* mine = lds + (threadIdx.x >> 6) * (S * R * 1024);
  const int64_t wave = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * wpb;
  const int64_t ntasks = nseg / R, steps = seg / 128;
  const int64_t mytasks = blocked ? ntasks / nw : wave < ntasks ? (ntasks - wave + nw - 1) / nw : 0;
  const int64_t total = mytasks * steps;
  const int64_t dtask = blocked ? 1 : nw;
  int64_t task = blocked ? wave * mytasks : wave, k = 0;
  int slot = 0;
  for (int64_t i = 0; i < total; ++i) {
    const double* p = a + task * R * seg + k * 128 + 2 * lane;
#pragma unroll
    for (int r = 0; r < R; ++r)
      __builtin_amdgcn_global_load_lds(p + r * seg, (__attribute__((address_space(3))) void*)(mine + (slot * R + r) * 1024), 16, 0, 2);
    slot = slot + 1 == S ? 0 : slot + 1;
    if (++k == steps) {
      k = 0;
      task += dtask;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * R) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int R, int S>
static int launch_lds(const double* a, int64_t seg, int64_t nseg, int grid, int wpb, int blocked, hipStream_t s) {
  if (blocked && (nseg / R) % ((int64_t)grid * wpb)) return 4;
  const size_t bytes = (size_t)wpb * S * R * 1024;
  if (hipFuncSetAttribute((const void*)read_lds_kernel<R, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return 2;
  hipLaunchKernelGGL((read_lds_kernel<R, S>), dim3(grid), dim3(64 * wpb), bytes, s, a, seg, nseg, blocked);
  return hipGetLastError() != hipSuccess;
}
extern "C" int readbw_lds_launch(const double* a, int64_t seg, int64_t nseg, int grid, int wpb, int R, int S, int blocked, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (seg % 128 || nseg % R || (size_t)wpb * S * R * 1024 > 160 * 1024) return 3;
  if (R == 4 && S == 4) return launch_lds<4, 4>(a, seg, nseg, grid, wpb, blocked, s);
  if (R == 4 && S == 8) return launch_lds<4, 8>(a, seg, nseg, grid, wpb, blocked, s);
  if (R == 2 && S == 8) return launch_lds<2, 8>(a, seg, nseg, grid, wpb, blocked, s);
  if (R == 8 && S == 4) return launch_lds<8, 4>(a, seg, nseg, grid, wpb, blocked, s);
  if (R == 1 && S == 16) return launch_lds<1, 16>(a, seg, nseg, grid, wpb, blocked, s);
  if (R == 2 && S == 4) return launch_lds<2, 4>(a, seg, nseg, grid, wpb, blocked, s);
  if (R == 1 && S == 8) return launch_lds<1, 8>(a, seg, nseg, grid, wpb, blocked, s);
  if (R == 4 && S == 2) return launch_lds<4, 2>(a, seg, nseg, grid, wpb, blocked, s);
  return 1;
}

// ---- the LDS-DMA ring with DYNAMIC task assignment: a wave draws its next task (R consecutive segments) from a global
// counter one task ahead (a returning atomic, issued by inline asm so that hipcc does not answer with vmcnt(0); its
// result is read a whole task later, behind counted waits that every older memory operation has passed by then)
template <int R, int S>
__global__ void read_lds_dyn_kernel(const double* __restrict__ a, int64_t seg, int64_t nseg, unsigned* counter) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63;
  char* mine = lds + (threadIdx.x >> 6) * (S * R * 1024);
  const int64_t ntasks = nseg / R, steps = seg / 128;
  const unsigned one = 1u;
  unsigned cur, nxt;
  cur = 0u;
  nxt = 0u;
  if (lane == 0)
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(cur) : "v"(counter), "v"(one) : "memory");
  cur = __builtin_amdgcn_readfirstlane(cur);
  int slot = 0;
  while ((int64_t)cur < ntasks) {
    if (lane == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(nxt) : "v"(counter), "v"(one) : "memory");
    const double* p = a + (int64_t)cur * R * seg + 2 * lane;
    for (int64_t k = 0; k < steps; ++k) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        __builtin_amdgcn_global_load_lds(p + r * seg + k * 128, (__attribute__((address_space(3))) void*)(mine + (slot * R + r) * 1024), 16, 0, 2);
      slot = slot + 1 == S ? 0 : slot + 1;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * R) : "memory");
    }
    // steps >= S: the atomic above is older than every load still in flight
    asm volatile("" : "+v"(nxt));
    cur = __builtin_amdgcn_readfirstlane(nxt);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int R, int S>
static int launch_lds_dyn(const double* a, int64_t seg, int64_t nseg, int grid, int wpb, unsigned* counter, hipStream_t s) {
  const size_t bytes = (size_t)wpb * S * R * 1024;
  if (seg / 128 < S) return 5;
  if (hipFuncSetAttribute((const void*)read_lds_dyn_kernel<R, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return 2;
  if (hipMemsetAsync(counter, 0, 4, s) != hipSuccess) return 6;
  hipLaunchKernelGGL((read_lds_dyn_kernel<R, S>), dim3(grid), dim3(64 * wpb), bytes, s, a, seg, nseg, counter);
  return hipGetLastError() != hipSuccess;
}
extern "C" int readbw_lds_dyn_launch(const double* a, int64_t seg, int64_t nseg, int grid, int wpb, int R, int S, unsigned* counter, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (seg % 128 || nseg % R || (size_t)wpb * S * R * 1024 > 160 * 1024) return 3;
  if (R == 4 && S == 4) return launch_lds_dyn<4, 4>(a, seg, nseg, grid, wpb, counter, s);
  if (R == 2 && S == 8) return launch_lds_dyn<2, 8>(a, seg, nseg, grid, wpb, counter, s);
  if (R == 1 && S == 16) return launch_lds_dyn<1, 16>(a, seg, nseg, grid, wpb, counter, s);
  return 1;
}

// ---- the static ring plus the matvec's x stream: per step one more DMA load, the matching 1 KiB of ONE shared vector
// (out of L2, requested by every wave of the machine at about the same time), optionally with the waves' starts staggered
template <int R, int S>
__global__ void read_lds_x_kernel(const double* __restrict__ a, const double* __restrict__ x, int64_t seg, int64_t nseg,
                                  int stagger) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  char* mine = lds + (threadIdx.x >> 6) * (S * (R + 1) * 1024);
  const int64_t wave = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * wpb;
  const int64_t ntasks = nseg / R, steps = seg / 128;
  const int64_t mytasks = wave < ntasks ? (ntasks - wave + nw - 1) / nw : 0;
  const int64_t total = mytasks * steps;
  if (stagger > 0) {
    const int turns = (int)((wave * 37) % 64) * stagger;    // up to 63 * stagger sleeps of 64 clocks
    for (int i = 0; i < turns; ++i) __builtin_amdgcn_s_sleep(1);
  }
  int64_t task = wave, k = 0;
  int slot = 0;
  for (int64_t i = 0; i < total; ++i) {
    const double* p = a + task * R * seg + k * 128 + 2 * lane;
#pragma unroll
    for (int r = 0; r < R; ++r)
      __builtin_amdgcn_global_load_lds(p + r * seg, (__attribute__((address_space(3))) void*)(mine + (slot * (R + 1) + r) * 1024), 16, 0, 2);
    __builtin_amdgcn_global_load_lds(x + k * 128 + 2 * lane, (__attribute__((address_space(3))) void*)(mine + (slot * (R + 1) + R) * 1024), 16, 0, 0);
    slot = slot + 1 == S ? 0 : slot + 1;
    if (++k == steps) {
      k = 0;
      task += nw;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * (R + 1)) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
extern "C" int readbw_lds_x_launch(const double* a, const double* x, int64_t seg, int64_t nseg, int grid, int wpb, int stagger,
                                   void* stream) {
  hipStream_t s = (hipStream_t)stream;
  constexpr int R = 4, S = 4;
  const size_t bytes = (size_t)wpb * S * (R + 1) * 1024;
  if (seg % 128 || nseg % R || bytes > 160 * 1024) return 3;
  if (hipFuncSetAttribute((const void*)read_lds_x_kernel<R, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return 2;
  hipLaunchKernelGGL((read_lds_x_kernel<R, S>), dim3(grid), dim3(64 * wpb), bytes, s, a, x, seg, nseg, stagger);
  return hipGetLastError() != hipSuccess;
}

// ---- does the SPACING of the R segments a wave reads side by side matter?  Task t of the static ring takes segments
// {(t / rstride) * R * rstride + t % rstride + r * rstride : r < R}: rstride = 1 is R adjacent segments (the matvec's
// R consecutive rows), rstride = 16 with 64 KiB segments puts them 1 MiB apart, as adjacent rows of the headline matrix are
template <int R, int S>
__global__ void read_lds_rs_kernel(const double* __restrict__ a, int64_t seg, int64_t nseg, int64_t rstride) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  char* mine = lds + (threadIdx.x >> 6) * (S * R * 1024);
  const int64_t wave = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * wpb;
  const int64_t ntasks = nseg / R, steps = seg / 128;
  const int64_t mytasks = wave < ntasks ? (ntasks - wave + nw - 1) / nw : 0;
  const int64_t total = mytasks * steps;
  int64_t task = wave, k = 0;
  int slot = 0;
  for (int64_t i = 0; i < total; ++i) {
    const int64_t first = (task / rstride) * R * rstride + task % rstride;
    const double* p = a + first * seg + k * 128 + 2 * lane;
#pragma unroll
    for (int r = 0; r < R; ++r)
      __builtin_amdgcn_global_load_lds(p + r * rstride * seg, (__attribute__((address_space(3))) void*)(mine + (slot * R + r) * 1024), 16, 0, 2);
    slot = slot + 1 == S ? 0 : slot + 1;
    if (++k == steps) {
      k = 0;
      task += nw;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * R) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int R, int S>
static int launch_lds_rs(const double* a, int64_t seg, int64_t nseg, int grid, int wpb, int64_t rstride, hipStream_t s) {
  const size_t bytes = (size_t)wpb * S * R * 1024;
  nseg -= nseg % (R * rstride);                 // (whole groups only: the caller counts the bytes the same way)
  if (hipFuncSetAttribute((const void*)read_lds_rs_kernel<R, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return 2;
  hipLaunchKernelGGL((read_lds_rs_kernel<R, S>), dim3(grid), dim3(64 * wpb), bytes, s, a, seg, nseg, rstride);
  return hipGetLastError() != hipSuccess;
}
extern "C" int readbw_lds_rs_launch(const double* a, int64_t seg, int64_t nseg, int grid, int wpb, int R, int S, int64_t rstride,
                                    void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (seg % 128 || rstride < 1 || (size_t)wpb * S * R * 1024 > 160 * 1024) return 3;
  if (R == 4 && S == 4) return launch_lds_rs<4, 4>(a, seg, nseg, grid, wpb, rstride, s);
  if (R == 2 && S == 8) return launch_lds_rs<2, 8>(a, seg, nseg, grid, wpb, rstride, s);
  if (R == 2 && S == 4) return launch_lds_rs<2, 4>(a, seg, nseg, grid, wpb, rstride, s);
  return 1;
}
