// matvec with the matrix in a VGPR ring (25 tiles x 16 doubles, one double per lane and tile, replicated over the 4
// DPP rows), operands broadcast with v_fmac_f64_dpp row_newbcast; next matrix streamed in behind the FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
template <int S, int T>
__device__ __forceinline__ void tile(double (&R)[25], const double* nextM, const double (&x)[S], double (&y)[S]) {
  constexpr int NB = S / 4, bi = T / NB, bj = T % NB;
  // wait for this tile (issued 25 loads ago), 16 FMAs, then refill the ring slot with the next matrix's tile
  asm volatile(
      "s_waitcnt vmcnt(24)\n\t"
      "v_fmac_f64_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %4, %5 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %4, %5 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %4, %6 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %4, %6 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %4, %6 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %4, %7 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %4, %7 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %4, %7 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %4, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %4, %8 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %4, %8 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
      "global_load_dwordx2 %4, %9, off offset:%10"
      : "+v"(y[4 * bi + 0]), "+v"(y[4 * bi + 1]), "+v"(y[4 * bi + 2]), "+v"(y[4 * bi + 3]), "+v"(R[T])
      : "v"(x[4 * bj + 0]), "v"(x[4 * bj + 1]), "v"(x[4 * bj + 2]), "v"(x[4 * bj + 3]), "v"(nextM), "i"(T * 128)
      : "memory");
}
template <int S, int T>
__device__ __forceinline__ void tiles(double (&R)[25], const double* nextM, const double (&x)[S], double (&y)[S]) {
  if constexpr (T < (S / 4) * (S / 4)) { tile<S, T>(R, nextM, x, y); tiles<S, T + 1>(R, nextM, x, y); }
}
template <int S>
__global__ __launch_bounds__(64) void k(const double* __restrict__ M, int nmat, double* out, int iters, double* chk) {
  double x[S], y[S], R[25];
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < S; ++i) x[i] = 1.0 + 1e-3 * (lane + i);
  int m = blockIdx.x % nmat;
  const double* lanebase = M + (lane & 15);
  // prime the ring with matrix m
#pragma unroll
  for (int t = 0; t < 25; ++t) asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(R[t]) : "v"(lanebase + (size_t)m * S * S), "i"(t * 128) : "memory");
  for (int it = 0; it < iters; ++it) {
    const int mn = (m + 1 == nmat) ? 0 : m + 1;
#pragma unroll
    for (int i = 0; i < S; ++i) y[i] = 0.0;
    tiles<S, 0>(R, lanebase + (size_t)mn * S * S, x, y);
    if (chk && it == 0) {
#pragma unroll
      for (int i = 0; i < S; ++i) chk[(blockIdx.x * 64 + lane) * S + i] = y[i];
    }
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = y[i];
    m = mn;
  }
  asm volatile("s_waitcnt vmcnt(0)");
  double s = 0;
#pragma unroll
  for (int i = 0; i < S; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + R[0] * 1e-30;
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); int CUS = prop.multiProcessorCount;
  const int S = 20, nmat = 1000;
  std::vector<double> P((size_t)nmat * 400), Ap((size_t)nmat * 400);
  for (size_t i = 0; i < P.size(); ++i) P[i] = 0.01 * ((i * 7919) % 101) - 0.3;
  for (int m = 0; m < nmat; ++m)
    for (int t = 0; t < 25; ++t)
      for (int kk = 0; kk < 16; ++kk) Ap[(size_t)m * 400 + t * 16 + kk] = P[(size_t)m * 400 + (4 * (t / 5) + kk / 4) * S + 4 * (t % 5) + kk % 4];
  double *M, *out, *chk;
  hipMalloc(&M, Ap.size() * 8); hipMemcpy(M, Ap.data(), Ap.size() * 8, hipMemcpyHostToDevice);
  hipMalloc(&out, 8 * 64 * CUS * 16); hipMalloc(&chk, 8 * 64 * S);
  k<S><<<1, 64>>>(M, nmat, out, 1, chk);
  std::vector<double> hc(64 * S); hipMemcpy(hc.data(), chk, hc.size() * 8, hipMemcpyDeviceToHost);
  double err = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < S; ++r) { double a = 0; for (int c = 0; c < S; ++c) a += P[r * S + c] * (1.0 + 1e-3 * (l + c)); err = fmax(err, fabs(a - hc[l * S + r])); }
  printf("dpp matvec check max err %g\n", err);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
  for (int wpc : {4, 8, 12, 16}) {
    int blocks = CUS * wpc, iters = 2000;
    k<S><<<blocks, 64>>>(M, nmat, out, 10, nullptr);
    hipEventRecord(e0); k<S><<<blocks, 64>>>(M, nmat, out, iters, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("dpp-ring matvec20 waves/CU=%2d  %.2f TFLOP/s\n", wpc, 2.0 * S * S * iters * (double)blocks * 64 / ms / 1e9);
  }
  return 0;
}
