// Per-lane pair statistics shared by the mapping kernels (null mode), the diagonal-pair and the group kernels
// (CoMap/Statistics.h:164-329).  Device code only.
#pragma once
#include <hip/hip_runtime.h>

#include "cmx_device.h"

namespace cmx {

// per-lane statistic between two count columns (row strides ld1 / ld2), CoMap/Statistics.h
__device__ __forceinline__ double pair_stat_strided(int kind, double param, int B, int K, const double* __restrict__ c1,
                                                    size_t ld1, const double* __restrict__ c2, size_t ld2,
                                                    const double* __restrict__ mv = nullptr /* [2][B], kind 6 */) {
  switch (kind) {
    case 0: case 4: case 6: {  // Correlation / Covariance / CorrectedCorrelation: VectorTools::cor, two-pass on type 0
      // CorrectedCorrelation (Statistics.h:176-204) first subtracts a per-branch mean vector from either operand
      const double* u1 = kind == 6 ? mv : nullptr;
      const double* u2 = kind == 6 ? mv + B : nullptr;
      double m1 = 0, m2 = 0;
#pragma unroll 8
      for (int b = 0; b < B; ++b) {
        m1 += c1[(size_t)b * K * ld1] - (u1 ? u1[b] : 0.0);
        m2 += c2[(size_t)b * K * ld2] - (u2 ? u2[b] : 0.0);
      }
      m1 /= B; m2 /= B;
      double sxy = 0, sxx = 0, syy = 0;
#pragma unroll 8
      for (int b = 0; b < B; ++b) {
        const double dx = c1[(size_t)b * K * ld1] - (u1 ? u1[b] : 0.0) - m1, dy = c2[(size_t)b * K * ld2] - (u2 ? u2[b] : 0.0) - m2;
        sxy += dx * dy; sxx += dx * dx; syy += dy * dy;
      }
      const double cov = sxy / (B - 1);
      if (kind == 4) return cov;
      return cov / (sqrt(sxx / (B - 1)) * sqrt(syy / (B - 1)));
    }
    case 9: {  // scalar product (VectorTools::scalar)
      double sxy = 0;
      for (int b = 0; b < B; ++b) sxy += c1[(size_t)b * K * ld1] * c2[(size_t)b * K * ld2];
      return sxy;
    }
    case 3: {  // Cosinus
      double sxy = 0, sxx = 0, syy = 0;
      for (int b = 0; b < B; ++b) {
        const double x = c1[(size_t)b * K * ld1], y = c2[(size_t)b * K * ld2];
        sxy += x * y; sxx += x * x; syy += y * y;
      }
      return sxy / (sqrt(sxx) * sqrt(syy));
    }
    case 7: {  // EuclidianDistance
      double d = 0;
      for (int b = 0; b < B; ++b) {
        double t1 = 0, t2 = 0;
        for (int k = 0; k < K; ++k) { t1 += c1[((size_t)b * K + k) * ld1]; t2 += c2[((size_t)b * K + k) * ld2]; }
        d = __builtin_fma(t2 - t1, t2 - t1, d);
      }
      return sqrt(d);
    }
    case 1: case 2: case 5: {
      double s1 = 0, s2 = 0, s3 = 0, cc = 0, n11 = 0, r1 = 0, r2 = 0;
      bool bad = false;
      for (int b = 0; b < B; ++b) {
        double t1 = 0, t2 = 0;
        for (int k = 0; k < K; ++k) { t1 += c1[((size_t)b * K + k) * ld1]; t2 += c2[((size_t)b * K + k) * ld2]; }
        s1 += t1 * t1; s2 += t2 * t2; s3 += (t1 + t2) * (t1 + t2);
        if (t1 >= 1.0 && t2 >= 1.0) cc += 1.0;
        if (!(t1 >= 0.0 && t1 < 10000.0) || !(t2 >= 0.0 && t2 < 10000.0)) bad = true;
        const double i1 = t1 >= param ? 1.0 : 0.0, i2 = t2 >= param ? 1.0 : 0.0;
        n11 += i1 * i2; r1 += i1; r2 += i2;
      }
      if (kind == 1) return 1.0 - sqrt(s3) / (sqrt(s1) + sqrt(s2));
      if (kind == 2) return cc;
      if (bad) return __builtin_nan("");
      const double np = B;
      const double cell[4] = {n11, r1 - n11, r2 - n11, np - r1 - r2 + n11};
      const double ma[4] = {r1, r1, np - r1, np - r1}, mb[4] = {r2, np - r2, r2, np - r2};
      double s = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (cell[q] > 0) s += (cell[q] / np) * log(cell[q] * np / (ma[q] * mb[q]));
      return s / log(2.7182818);
    }
  }
  return __builtin_nan("");
}

__device__ __forceinline__ double pair_stat_lane(int kind, double param, int B, int K, const double* __restrict__ c1,
                                                 const double* __restrict__ c2) {
  return pair_stat_strided(kind, param, B, K, c1, (size_t)kWave, c2, (size_t)kWave);
}

}  // namespace cmx
