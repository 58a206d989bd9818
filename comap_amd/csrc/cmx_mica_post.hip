// What Mica does with the all-pairs MI matrix once it exists (CoMap/Mica.cpp:346-363, 549-607): per-site average MI
// (the APC / RCW corrections of :656-657 are products of these), and the "z-score" null that turns every pair of the
// data set itself into one draw of the null distribution.  Bandwidth-bound passes over the n x n matrix.
#include <hip/hip_runtime.h>

#include "cmx_device.h"

namespace cmx {

namespace {

__device__ __forceinline__ double block_sum_256(double v, double* sm) {
  for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  const double t = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  return t;
}

// averageMI[i] = sum_{j != i} MI(i, j) / (n - 1)   (Mica.cpp:349-361); one workgroup per site.
// Element (i, j) is read from the upper triangle (row min(i,j), column max(i,j)): the matrix is symmetric and the
// intra entry points of the engine only guarantee that triangle.
__global__ __launch_bounds__(256) void mica_average_kernel(const double* __restrict__ mi, size_t n, size_t ld,
                                                           double* __restrict__ avg) {
  __shared__ double sm[4];
  const size_t i = blockIdx.x;
  double s = 0.0;
  for (size_t j = threadIdx.x; j < n; j += 256)
    if (j != i) s += j > i ? mi[i * ld + j] : mi[j * ld + i];
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) avg[i] = s / (double)(n - 1);
}

// fullAverageMI = mean(averageMI)   (Mica.cpp:363); one workgroup, fixed order
__global__ __launch_bounds__(256) void mica_full_average_kernel(const double* __restrict__ avg, size_t n, double* __restrict__ full) {
  __shared__ double sm[4];
  double s = 0.0;
  for (size_t j = threadIdx.x; j < n; j += 256) s += avg[j];
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) *full = s / (double)n;
}

// z-score null (Mica.cpp:565-603): pair (i, j), j > i, in row-major order -> (statistic, min key); which: 0 MI,
// 1 MIp = MI - avg_i avg_j / full, 2 MIc = MI / (avg_i avg_j / 2)
__global__ __launch_bounds__(256) void mica_zscore_kernel(int which, const double* __restrict__ mi, size_t n, size_t ld,
                                                          const double* __restrict__ avg, const double* __restrict__ full,
                                                          const double* __restrict__ key, double* __restrict__ out_stat,
                                                          double* __restrict__ out_key) {
  const size_t i = blockIdx.y;
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j <= i || j >= n) return;
  const size_t p = i * n - i * (i + 1) / 2 + (j - i - 1);
  double v = mi[i * ld + j];
  if (which == 1) v -= avg[i] * avg[j] / *full;
  else if (which == 2) v /= avg[i] * avg[j] / 2.;
  out_stat[p] = v;
  out_key[p] = fmin(key[i], key[j]);
}

}  // namespace

hipError_t launch_mica_average(const double* d_mi, size_t n, size_t ld, double* d_avg, double* d_full, hipStream_t stream) {
  hipLaunchKernelGGL(mica_average_kernel, dim3((unsigned)n), dim3(256), 0, stream, d_mi, n, ld, d_avg);
  hipLaunchKernelGGL(mica_full_average_kernel, dim3(1), dim3(256), 0, stream, d_avg, n, d_full);
  return hipGetLastError();
}

hipError_t launch_mica_zscore(int which, const double* d_mi, size_t n, size_t ld, const double* d_avg, const double* d_full,
                              const double* d_key, double* d_stat, double* d_outkey, hipStream_t stream) {
  hipLaunchKernelGGL(mica_zscore_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, stream, which, d_mi, n,
                     ld, d_avg, d_full, d_key, d_stat, d_outkey);
  return hipGetLastError();
}

}  // namespace cmx
