// Micro-benchmark that sizes the nucleotide mapping kernel (round 3): lane = site, the 4x4 operators of
// a branch arrive through the scalar cache (s_load_dwordx16, wave-uniform) and are applied with v_fma_f64 taking SGPR
// operands; messages of a block of nodes live in LDS ([slot][lane][4] doubles).  One "visit" = the outside-pass work of
// one internal node: W = J^T U, count = sum W o Ma o Mb, Up = P^T U, Ua = Up o Mb, Ub = Up o Ma  (48 FMA/MUL + LDS traffic).
// Reports cycles per visit and the fp64 rate at 1..4 waves per SIMD, with and without the LDS traffic, and for an inside
// visit (M = P (Ma o Mb)).
// Build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_nuc.hip -o /tmp/ubench_nuc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double d16 __attribute__((ext_vector_type(16)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef const d16 __attribute__((address_space(4)))* cd16p;

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

__device__ __forceinline__ void mv_t(const d16& A, const double (&x)[4], double (&y)[4]) {   // y = A^T x
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double a = A[j] * x[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) a = __builtin_fma(A[4 * i + j], x[i], a);
    y[j] = a;
  }
}
__device__ __forceinline__ void mv_n(const d16& A, const double (&x)[4], double (&y)[4]) {   // y = A x
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double a = A[4 * i] * x[0];
#pragma unroll
    for (int j = 1; j < 4; ++j) a = __builtin_fma(A[4 * i + j], x[j], a);
    y[i] = a;
  }
}
__device__ __forceinline__ void lds_read(const unsigned char* base, int slot, int lane, double (&v)[4]) {
  const d2* p = reinterpret_cast<const d2*>(base + ((size_t)slot * 64 + lane) * 32);
  const d2 a = p[0], b = p[1];
  v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
}
__device__ __forceinline__ void lds_write(unsigned char* base, int slot, int lane, const double (&v)[4]) {
  d2* p = reinterpret_cast<d2*>(base + ((size_t)slot * 64 + lane) * 32);
  d2 a, b;
  a[0] = v[0]; a[1] = v[1]; b[0] = v[2]; b[1] = v[3];
  p[0] = a; p[1] = b;
}

// MODE 0: outside visit with LDS messages; 1: outside visit, messages stay in registers (no LDS); 2: inside visit with LDS
template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k_visit(const double* __restrict__ ops, int nops, int nvisit, int nslots,
                                                    double* __restrict__ out, long long* __restrict__ cyc) {
  const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned char* my = smem + (size_t)wib * nslots * 64 * 32;
  double U[4], Ma[4], Mb[4], W[4], Up[4], acc = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { U[i] = 0.25 + 1e-3 * lane + 1e-2 * i; Ma[i] = 0.9; Mb[i] = 1.1; }
  for (int s = 0; s < nslots; ++s) lds_write(my, s, lane, U);
  __syncthreads();
  const cd16p O = (cd16p)ops;
  int o = (blockIdx.x * 4 + wib) % nops;
  d16 J = O[2 * o], P = O[2 * o + 1];
  const long long t0 = (long long)__builtin_readcyclecounter();
  int sa = 0, sb = 1;
  for (int v = 0; v < nvisit; ++v) {
    const int on = (o + 1 == nops) ? 0 : o + 1;
    const d16 Jn = O[2 * on], Pn = O[2 * on + 1];   // next visit's operators (loop-carried, loaded one visit ahead)
    if (MODE == 0 || MODE == 2) { lds_read(my, sa, lane, Ma); lds_read(my, sb, lane, Mb); }
    if (MODE == 2) {
      double D[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) D[i] = Ma[i] * Mb[i];
      mv_n(P, D, W);
      lds_write(my, sa, lane, W);
      acc += W[0];
    } else {
      mv_t(J, U, W);
      double c = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) c = __builtin_fma(W[i] * Ma[i], Mb[i], c);
      acc += c;
      mv_t(P, U, Up);
      double Ua[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { Ua[i] = Up[i] * Mb[i]; U[i] = Up[i] * Ma[i]; }
      if (MODE == 0) lds_write(my, sb, lane, Ua);
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { Mb[i] = Ua[i] * 0.999; Ma[i] = Ma[i] * 1.0001; }
      }
      // keep the values bounded
#pragma unroll
      for (int i = 0; i < 4; ++i) U[i] = U[i] * 0.5 + 0.25;
    }
    J = Jn; P = Pn; o = on;
    sa = (sa + 1 == nslots) ? 0 : sa + 1;
    sb = (sb + 1 == nslots) ? 0 : sb + 1;
  }
  const long long t1 = (long long)__builtin_readcyclecounter();
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc + U[0] + U[1] + U[2] + U[3];
  if (lane == 0) cyc[blockIdx.x * 4 + wib] = t1 - t0;
}

template <int MODE, int WPS>
static int run(const double* dops, int nops, int CUS, double* out, long long* dcyc, int nslots) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int blocks = CUS * WPS, nvisit = 20000;
  const size_t lds = (size_t)4 * nslots * 64 * 32;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_visit<MODE, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / WPS));
  k_visit<MODE, WPS><<<blocks, 256, lds>>>(dops, nops, 100, nslots, out, dcyc);
  CK(hipEventRecord(e0));
  k_visit<MODE, WPS><<<blocks, 256, lds>>>(dops, nops, nvisit, nslots, out, dcyc);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> c(blocks * 4);
  CK(hipMemcpy(c.data(), dcyc, sizeof(long long) * c.size(), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto x : c) mean += (double)x;
  mean /= c.size();
  const double ninstr = MODE == 2 ? 20.0 : 48.0;   // f64 VALU instructions of the visit proper
  const double fl = 2.0 * ninstr * 64 * nvisit * (double)blocks * 4;
  printf("mode %d waves/SIMD %d slots %d: %.3f ms, %.0f cycles per visit per wave, %.2f TFLOP/s (visit = %.0f f64 instr)\n", MODE, WPS, nslots,
         ms, mean / nvisit, fl / ms / 1e9, ninstr);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int CUS = prop.multiProcessorCount;
  printf("device %s CUs %d\n", prop.name, CUS);
  const int nops = 2048;   // 2048 x 256 B = 512 KB of operators: beyond the scalar cache, inside L2 (a class pass of cfg4 streams 130 KB)
  std::vector<double> h((size_t)nops * 32);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((i % 5) == 0 ? 0.7 : 0.1) + 1e-4 * (double)(i % 97);
  double *dops, *out;
  long long* dcyc;
  CK(hipMalloc(&dops, h.size() * 8));
  CK(hipMemcpy(dops, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, sizeof(double) * 256 * CUS * 8));
  CK(hipMalloc(&dcyc, sizeof(long long) * 4 * CUS * 8));
  if (run<1, 1>(dops, nops, CUS, out, dcyc, 1)) return 1;
  if (run<1, 2>(dops, nops, CUS, out, dcyc, 1)) return 1;
  if (run<1, 3>(dops, nops, CUS, out, dcyc, 1)) return 1;
  if (run<1, 4>(dops, nops, CUS, out, dcyc, 1)) return 1;
  if (run<0, 2>(dops, nops, CUS, out, dcyc, 8)) return 1;
  if (run<0, 3>(dops, nops, CUS, out, dcyc, 5)) return 1;
  if (run<0, 4>(dops, nops, CUS, out, dcyc, 4)) return 1;
  if (run<2, 2>(dops, nops, CUS, out, dcyc, 8)) return 1;
  if (run<2, 4>(dops, nops, CUS, out, dcyc, 4)) return 1;
  return 0;
}
