// Micro-benchmarks that size the design of the mapping kernel (round 1):
//   1. v_fma_f64 issue rate (VALU fp64 peak)
//   2. v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 issue rate
//   3. matvec 20x20 with the matrix arriving through scalar loads (wave-uniform operand) -- the
//      inner loop shape of map_kernel
// Build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_f64.hip -o /tmp/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_fma(double* out, int iters, double a, double b) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mfma16(double* out, int iters, double a, double b) {
  d4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
  double av = a + threadIdx.x, bv = b - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mfma4(double* out, int iters, double a, double b) {
  double acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0;
  double av = a + threadIdx.x, bv = b - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// matvec with wave-uniform matrix: y[r] = sum_c M[m][r][c] x[c];  M streamed through the scalar path
template <int S>
__global__ void k_matvec_s(const double* __restrict__ M, int nmat, double* out, int iters) {
  double x[S], y[S];
#pragma unroll
  for (int i = 0; i < S; ++i) x[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  int m = blockIdx.x % nmat;
  for (int it = 0; it < iters; ++it) {
    const double* Mm = M + (size_t)m * S * S;
#pragma unroll
    for (int r = 0; r < S; ++r) {
      double a = 0;
#pragma unroll
      for (int c = 0; c < S; ++c) a = __builtin_fma(Mm[r * S + c], x[c], a);
      y[r] = a;
    }
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = y[i];
    m = (m + 1 == nmat) ? 0 : m + 1;
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < S; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// ---- scalar-path matvec with hand-issued s_load_dwordx16 ping-pong (4x4-block packed matrix) ----
typedef int s16 __attribute__((ext_vector_type(16)));
typedef double d8 __attribute__((ext_vector_type(8)));
#define SLOAD32(p, off, r0, r1) \
  asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4" : "=&s"(r0), "=&s"(r1) : "s"(p), "i"(off), "i"((off) + 64))
#define SWAIT2(r0, r1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0), "+s"(r1))

template <int S, bool TR, int T>
__device__ __forceinline__ void mv_tile(const s16& r0, const s16& r1, const double (&x)[S], double (&y)[S]) {
  constexpr int NB = S / 4;
  constexpr int bi = T / NB, bj = T % NB;
  d8 lo = __builtin_bit_cast(d8, r0), hi = __builtin_bit_cast(d8, r1);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = k / 4, j = k % 4;
    double a = k < 8 ? lo[k] : hi[k - 8];
    if (!TR) y[4 * bi + i] = __builtin_fma(a, x[4 * bj + j], y[4 * bi + i]);
    else     y[4 * bj + j] = __builtin_fma(a, x[4 * bi + i], y[4 * bj + j]);
  }
}

template <int S, bool TR, int T>
__device__ __forceinline__ void mv_steps(const double* A, s16& a0, s16& a1, s16& b0, s16& b1, const double (&x)[S], double (&y)[S]) {
  constexpr int NT = (S / 4) * (S / 4);
  if constexpr (T < NT) {
    // tile T is resident in (a0,a1); prefetch T+1 into (b0,b1), compute T, wait.
    if constexpr (T + 1 < NT) { SLOAD32(A, (T + 1) * 128, b0, b1); __builtin_amdgcn_sched_barrier(0); }
    mv_tile<S, TR, T>(a0, a1, x, y);
    if constexpr (T + 1 < NT) {
      SWAIT2(b0, b1);
      mv_steps<S, TR, T + 1>(A, b0, b1, a0, a1, x, y);
    }
  }
}

template <int S, bool TR>
__device__ __forceinline__ void matvec_sasm(const double* A, const double (&x)[S], double (&y)[S]) {
  s16 a0, a1, b0, b1;
#pragma unroll
  for (int i = 0; i < S; ++i) y[i] = 0;
  SLOAD32(A, 0, a0, a1);
  SWAIT2(a0, a1);
  mv_steps<S, TR, 0>(A, a0, a1, b0, b1, x, y);
}

template <int S>
__global__ void k_matvec_asm(const double* __restrict__ M, int nmat, double* out, int iters) {
  double x[S], y[S];
#pragma unroll
  for (int i = 0; i < S; ++i) x[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  int m = blockIdx.x % nmat;
  for (int it = 0; it < iters; ++it) {
    const double* Mm = M + (size_t)m * S * S;
    matvec_sasm<S, false>(Mm, x, y);
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = y[i];
    m = (m + 1 == nmat) ? 0 : m + 1;
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < S; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// correctness probe for the packed layout: y = P x with P given row-major on the host
template <int S>
__global__ void k_matvec_asm_check(const double* __restrict__ Apacked, const double* xin, double* yout, double* ytout) {
  double x[S], y[S];
#pragma unroll
  for (int i = 0; i < S; ++i) x[i] = xin[i] + threadIdx.x;
  matvec_sasm<S, false>(Apacked, x, y);
#pragma unroll
  for (int i = 0; i < S; ++i) yout[threadIdx.x * S + i] = y[i];
  matvec_sasm<S, true>(Apacked, x, y);
#pragma unroll
  for (int i = 0; i < S; ++i) ytout[threadIdx.x * S + i] = y[i];
}

// check the f64 MFMA fragment layout: C = A(16x4) * B(4x16)
__global__ void k_layout(const double* A, const double* B, double* C) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];   // A[i = l&15][k = l>>4]
  double b = B[(l >> 4) * 16 + (l & 15)];  // B[k = l>>4][j = l&15]
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];  // row=(l>>4)+4r, col=l&15
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  double* out;
  CK(hipMalloc(&out, sizeof(double) * 256 * 8192));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms;
  const int CUS = prop.multiProcessorCount;
  for (int wpc : {4, 8, 16}) {  // waves per CU
    int blocks = CUS * wpc / 4, threads = 256, iters = 20000;
    k_fma<<<blocks, threads>>>(out, 10, 1.0000001, 1e-9);
    CK(hipEventRecord(e0));
    k_fma<<<blocks, threads>>>(out, iters, 1.0000001, 1e-9);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    double fl = 2.0 * 16 * iters * (double)blocks * threads;
    printf("v_fma_f64      waves/CU=%2d  %.2f TFLOP/s\n", wpc, fl / ms / 1e9);
    k_mfma16<<<blocks, threads>>>(out, 10, 1.0, 1.0);
    CK(hipEventRecord(e0));
    k_mfma16<<<blocks, threads>>>(out, iters / 4, 1e-3, 1e-3);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    fl = 2.0 * 16 * 16 * 4 * 4 * (iters / 4) * (double)blocks * (threads / 64);
    printf("mfma_f64_16x16x4 waves/CU=%2d  %.2f TFLOP/s\n", wpc, fl / ms / 1e9);
    k_mfma4<<<blocks, threads>>>(out, 10, 1.0, 1.0);
    CK(hipEventRecord(e0));
    k_mfma4<<<blocks, threads>>>(out, iters / 4, 1e-3, 1e-3);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    fl = 2.0 * 4 * 4 * 4 * 4 * 8 * (iters / 4) * (double)blocks * (threads / 64);
    printf("mfma_f64_4x4x4_4b waves/CU=%2d  %.2f TFLOP/s\n", wpc, fl / ms / 1e9);
  }
  // scalar-fed matvec
  {
    const int S = 20, nmat = 1000;  // 3.2 MB of matrices: L2-resident working set like P/J of cfg2
    std::vector<double> hM((size_t)nmat * S * S);
    for (size_t i = 0; i < hM.size(); ++i) hM[i] = ((i % (S + 1)) == 0 ? 0.9 : 0.1 / S);
    double* M;
    CK(hipMalloc(&M, hM.size() * 8));
    CK(hipMemcpy(M, hM.data(), hM.size() * 8, hipMemcpyHostToDevice));
    for (int wpc : {4, 8, 12, 16}) {
      int threads = 64, blocks = CUS * wpc, iters = 2000;
      k_matvec_s<S><<<blocks, threads>>>(M, nmat, out, 10);
      CK(hipEventRecord(e0));
      k_matvec_s<S><<<blocks, threads>>>(M, nmat, out, iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      double fl = 2.0 * S * S * iters * (double)blocks * threads;
      printf("matvec20 scalar-fed waves/CU=%2d  %.2f TFLOP/s  (%.1f GB/s of matrix through the scalar path)\n", wpc,
             fl / ms / 1e9, 8.0 * S * S * iters * (double)blocks / ms / 1e6);
    }

    for (int wpc : {4, 8, 12, 16}) {
      int threads = 64, blocks = CUS * wpc, iters = 2000;
      k_matvec_asm<S><<<blocks, threads>>>(M, nmat, out, 10);
      CK(hipEventRecord(e0));
      k_matvec_asm<S><<<blocks, threads>>>(M, nmat, out, iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      double fl = 2.0 * S * S * iters * (double)blocks * threads;
      printf("matvec20 asm-sload waves/CU=%2d  %.2f TFLOP/s\n", wpc, fl / ms / 1e9);
    }
    {  // packed-layout correctness
      std::vector<double> P(S * S), Ap(S * S), x(S), y(64 * S), yt(64 * S);
      for (int i = 0; i < S * S; ++i) P[i] = 0.01 * ((i * 7919) % 101) - 0.3;
      for (int i = 0; i < S; ++i) x[i] = 0.5 + i;
      for (int t = 0; t < (S / 4) * (S / 4); ++t)
        for (int k = 0; k < 16; ++k) Ap[t * 16 + k] = P[(4 * (t / (S / 4)) + k / 4) * S + 4 * (t % (S / 4)) + k % 4];
      double *dA, *dx, *dy, *dyt;
      CK(hipMalloc(&dA, S * S * 8)); CK(hipMalloc(&dx, S * 8)); CK(hipMalloc(&dy, 64 * S * 8)); CK(hipMalloc(&dyt, 64 * S * 8));
      CK(hipMemcpy(dA, Ap.data(), S * S * 8, hipMemcpyHostToDevice));
      CK(hipMemcpy(dx, x.data(), S * 8, hipMemcpyHostToDevice));
      k_matvec_asm_check<S><<<1, 64>>>(dA, dx, dy, dyt);
      CK(hipMemcpy(y.data(), dy, 64 * S * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(yt.data(), dyt, 64 * S * 8, hipMemcpyDeviceToHost));
      double e1m = 0, e2m = 0;
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < S; ++r) {
          double a = 0, b = 0;
          for (int c = 0; c < S; ++c) { a += P[r * S + c] * (x[c] + l); b += P[c * S + r] * (x[c] + l); }
          e1m = fmax(e1m, fabs(a - y[l * S + r])); e2m = fmax(e2m, fabs(b - yt[l * S + r]));
        }
      printf("asm matvec check: max err fwd %g  transposed %g\n", e1m, e2m);
    }
    const int S4 = 4;
    for (int wpc : {8, 16}) {
      int threads = 64, blocks = CUS * wpc, iters = 20000;
      k_matvec_s<S4><<<blocks, threads>>>(M, nmat, out, 10);
      CK(hipEventRecord(e0));
      k_matvec_s<S4><<<blocks, threads>>>(M, nmat, out, iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      double fl = 2.0 * S4 * S4 * iters * (double)blocks * threads;
      printf("matvec4  scalar-fed waves/CU=%2d  %.2f TFLOP/s\n", wpc, fl / ms / 1e9);
    }
  }
  // layout check
  {
    std::vector<double> A(64), B(64), C(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = i + 1; B[i] = 0.5 * i - 3; }
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j)
        for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dC;
    CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(dA, dB, dC);
    CK(hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost));
    double err = 0;
    for (int i = 0; i < 256; ++i) err = fmax(err, fabs(C[i] - R[i]));
    printf("f64 mfma 16x16x4 layout check max err %g\n", err);
  }
  return 0;
}
