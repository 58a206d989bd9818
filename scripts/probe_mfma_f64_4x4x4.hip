// Probe of the lane mapping of v_mfma_f64_4x4x4_4b_f64 (4 blocks of D[4x4] += A[4x4] . B[4x4]) on gfx950.
// A = one-hot at (block, lane a), B = one-hot at (block, lane b): prints which D lane receives the product.
// hipcc --offload-arch=gfx950 -O2 scripts/probe_mfma_f64_4x4x4.hip -o build/probe_mfma && build/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main() {
  double *a, *b, *d;
  hipMallocManaged(&a, 64 * 8); hipMallocManaged(&b, 64 * 8); hipMallocManaged(&d, 64 * 8);
  // for each (la, lb) within block 0: find the D lane; prints tables A-lane -> (i,k), B-lane -> (k,n), D-lane -> (i,n)
  int dl[16][16];
  for (int la = 0; la < 16; ++la)
    for (int lb = 0; lb < 16; ++lb) {
      for (int q = 0; q < 64; ++q) { a[q] = 0; b[q] = 0; }
      a[la] = 1.0; b[lb] = 1.0;
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
      hipDeviceSynchronize();
      int hit = -1, nh = 0;
      for (int q = 0; q < 64; ++q) if (d[q] != 0.0) { hit = q; ++nh; }
      dl[la][lb] = nh == 1 ? hit : (nh == 0 ? -1 : -2);
    }
  printf("D lane for (A lane row, B lane col), block 0 (-1: no product, i.e. k mismatch)\n");
  for (int la = 0; la < 16; ++la) { for (int lb = 0; lb < 16; ++lb) printf("%3d", dl[la][lb]); printf("\n"); }
  // cross-block check: A in block 1 lane 16, B in block 0 lane 0
  for (int q = 0; q < 64; ++q) { a[q] = 0; b[q] = 0; }
  a[16] = 1.0; b[0] = 1.0;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
  hipDeviceSynchronize();
  int nh = 0; for (int q = 0; q < 64; ++q) nh += d[q] != 0.0;
  printf("cross-block products: %d\n", nh);
  for (int q = 0; q < 64; ++q) { a[q] = 0; b[q] = 0; }
  a[16 + 5] = 2.0; b[16 + 6] = 3.0;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
  hipDeviceSynchronize();
  for (int q = 0; q < 64; ++q) if (d[q] != 0.0) printf("block1: A lane 5, B lane 6 -> D lane %d value %g\n", q, d[q]);
  return 0;
}
