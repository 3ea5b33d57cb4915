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
