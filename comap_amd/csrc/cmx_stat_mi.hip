// DiscreteMutualInformationStatistic with an arbitrary bounds vector (CoMap/Statistics.h:307-327): per-branch totals are
// binned with Domain(bounds)::getIndex (CoMap/Domain.cpp:113-122) and the statistic is VectorTools::miDiscrete of the two
// class vectors -- sum over the observed cells (a, c) of (n_ac / B) log(n_ac B / (n_a n_c)) / log(2.7182818).  The factory
// builds two kinds of bounds (CoMap/CoETools.cpp:577-593): {0, threshold, 10000} (two classes; served by the indicator
// Gram of pair_gram_kernel) and, for nijt = Label, -0.5, 0.5, .., S(S-1) + 0.5 (13 classes for nucleotides, 381 for
// proteins).  This file serves any bounds vector.
//
// mi_classify_kernel: counts -> one 32-bit word per (branch, site): class index (low half; 0xFFFF = total outside the
//   bounds, the reference's OutOfRangeException -> the statistic of every pair of that site is NaN) and the number of
//   branches of the site that share the class (high half: the marginal count n_a the cell terms need).
// Pairs: one wave per pair at a time.  The joint table of a pair has at most B occupied cells out of up to 381^2, so it is
//   a hash table in LDS keyed by (a, c): every lane inserts its branches (CAS on the key, add on the count, min on the
//   first branch index of the cell); the lane that owns a cell's FIRST branch evaluates the cell's term -- same expression
//   as the restatement, so a table whose cells all satisfy n_ac B == n_a n_c gives exactly 0 as it does there -- and the
//   terms are summed as 2^-46 fixed-point integers: the value of a pair depends on its joint table only, not on the order
//   of the branches, nor on which kernel, row block or rank computed it.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "cmx_device.h"

namespace cmx {

constexpr uint32_t kMiEmpty = 0xFFFFFFFFu;

// Domain::getIndex: first i >= 1 with x < bounds[i] -> i - 1; outside [bounds[0], bounds[nb-1]) -> -1.  (Binary search:
// the same index as the reference's linear scan for non-decreasing bounds, which the Domain constructor enforces.)
__device__ __forceinline__ int mi_domain_index(const double* __restrict__ bounds, int nb, double x) {
  if (!(x >= bounds[0]) || !(x < bounds[nb - 1])) return -1;
  int lo = 1, hi = nb - 1;   // answer in [lo, hi]: bounds[hi] > x
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (x < bounds[mid]) hi = mid; else lo = mid + 1;
  }
  return lo - 1;
}

__global__ void mi_classify_kernel(const double* __restrict__ counts, size_t n, size_t ldc, int B, int K,
                                   const double* __restrict__ bounds, int nb, uint32_t* __restrict__ cls, size_t ldx,
                                   uint8_t* __restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool anybad = false;
  for (int b = 0; b < B; ++b) {
    double t = 0.0;
    for (int k = 0; k < K; ++k) t += counts[((size_t)b * K + k) * ldc + i];
    const int c = mi_domain_index(bounds, nb, t);
    anybad |= c < 0;
    cls[(size_t)b * ldx + i] = c < 0 ? 0xFFFFu : (uint32_t)c;
  }
  bad[i] = anybad ? 1 : 0;
  // marginal counts: branches of this site in the same class (B^2 / 2 compares per site, once per call)
  for (int b = 0; b < B; ++b) {
    const uint32_t a = cls[(size_t)b * ldx + i] & 0xFFFFu;
    uint32_t m = 0;
    for (int q = 0; q < B; ++q) m += ((cls[(size_t)q * ldx + i] & 0xFFFFu) == a) ? 1u : 0u;
    cls[(size_t)b * ldx + i] = a | (m << 16);
  }
}

// LDS of one wave: keys[cap] | cnt[cap] | first[cap] | slot_of_branch[B rounded up to 64]
__host__ __device__ inline int mi_table_cap(int B) {
  int cap = 128;
  while (cap < 2 * B) cap <<= 1;
  return cap;
}
__host__ __device__ inline size_t mi_lds_bytes(int B) { return (size_t)mi_table_cap(B) * 12 + (size_t)((B + 63) / 64 * 64) * 4; }

struct MiTable {
  uint32_t *keys, *cnt, *first, *slot;
  int cap;
};
__device__ __forceinline__ MiTable mi_table(uint8_t* lds, int B) {
  MiTable t;
  t.cap = mi_table_cap(B);
  t.keys = reinterpret_cast<uint32_t*>(lds);
  t.cnt = t.keys + t.cap;
  t.first = t.cnt + t.cap;
  t.slot = t.first + t.cap;
  return t;
}
__device__ __forceinline__ void mi_table_init(const MiTable& t, int lane) {
  for (int q = lane; q < t.cap; q += kWave) { t.keys[q] = kMiEmpty; t.cnt[q] = 0; t.first[q] = kMiEmpty; }
}

// statistic of one pair; w1 / w2: the packed class words of the two sites (element b at w[b * ld]).  All 64 lanes of the
// wave take part and all return the value.  The table must be empty on entry and is empty again on return.
__device__ __forceinline__ double mi_pair_wave(const MiTable& t, int lane, int B, const uint32_t* __restrict__ w1, size_t ld1,
                                               const uint32_t* __restrict__ w2, size_t ld2) {
  const unsigned mask = (unsigned)t.cap - 1u;
  for (int b = lane; b < B; b += kWave) {
    const uint32_t a = w1[(size_t)b * ld1] & 0xFFFFu, c = w2[(size_t)b * ld2] & 0xFFFFu;
    const uint32_t key = (a << 16) | c;
    unsigned h = (key * 0x9E3779B1u) >> 7 & mask;
    for (;;) {
      const uint32_t old = atomicCAS(&t.keys[h], kMiEmpty, key);
      if (old == kMiEmpty || old == key) break;
      h = (h + 1u) & mask;
    }
    atomicAdd(&t.cnt[h], 1u);
    atomicMin(&t.first[h], (uint32_t)b);
    t.slot[b] = h;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  // cell terms are summed in 2^-46 fixed point: an integer sum does not depend on the order of the cells, so two pairs
  // with the same joint table get the same bits whatever order their branches come in -- ties between a statistic and a
  // null value (frequent with integer tables: the p-value counts "null < statistic" strictly, CoETools.cpp:715) are then
  // ties here exactly where they are ties in exact arithmetic.  A term that is exactly 0 stays exactly 0.
  const double np = (double)B;
  long long acc = 0;
  for (int b = lane; b < B; b += kWave) {
    const unsigned h = t.slot[b];
    if (t.first[h] == (uint32_t)b) {
      const double nac = (double)t.cnt[h], na = (double)(w1[(size_t)b * ld1] >> 16), nc = (double)(w2[(size_t)b * ld2] >> 16);
      acc += __double2ll_rn((nac / np) * log(nac * np / (na * nc)) * 70368744177664.0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  for (int b = lane; b < B; b += kWave) {
    const unsigned h = t.slot[b];
    t.keys[h] = kMiEmpty; t.cnt[h] = 0; t.first[h] = kMiEmpty;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
  return ((double)acc * (1.0 / 70368744177664.0)) / log(2.7182818);
}

constexpr int kMiTile = 32;   // columns per workgroup

// rows irow0 .. of the full matrix against all columns: out[il * ldo + j].  intra: 0 two data sets, 1 NaN for j <= i,
// 2 pairs j <= i are left untouched (row blocks: the caller only reads j > i)
__global__ __launch_bounds__(kWave) void mi_pairs_block_kernel(int B, const uint32_t* __restrict__ cls1, const uint8_t* __restrict__ bad1,
                                                               size_t ld1, const uint32_t* __restrict__ cls2,
                                                               const uint8_t* __restrict__ bad2, size_t n2, size_t ld2, int intra,
                                                               double* __restrict__ out, size_t ldo, size_t irow0) {
  extern __shared__ __attribute__((aligned(16))) uint8_t mi_smem[];
  const int lane = threadIdx.x;
  const size_t il = blockIdx.y, i = irow0 + il, j0 = (size_t)blockIdx.x * kMiTile;
  const size_t jend = std::min(n2, j0 + kMiTile);
  if (intra && jend <= i + 1) {
    if (intra == 1)
      for (size_t j = j0 + lane; j < jend; j += kWave) out[il * ldo + j] = __builtin_nan("");
    return;
  }
  const MiTable t = mi_table(mi_smem, B);
  mi_table_init(t, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const bool badi = bad1[il] != 0;
  for (size_t j = j0; j < jend; ++j) {
    double v;
    if (intra && j <= i) {
      if (intra == 2) continue;
      v = __builtin_nan("");
    } else if (badi || bad2[j]) {
      v = __builtin_nan("");
    } else {
      v = mi_pair_wave(t, lane, B, cls1 + il, ld1, cls2 + j, ld2);
    }
    if (lane == 0) out[il * ldo + j] = v;
  }
}

// pairs (p of data set 1, p of data set 2), p < n: the statistic of a null distribution's replicate pairs
__global__ __launch_bounds__(kWave) void mi_pairs_diag_kernel(int B, const uint32_t* __restrict__ cls1, const uint8_t* __restrict__ bad1,
                                                              size_t ld1, const uint32_t* __restrict__ cls2,
                                                              const uint8_t* __restrict__ bad2, size_t ld2, size_t n,
                                                              double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t mi_smem[];
  const int lane = threadIdx.x;
  const MiTable t = mi_table(mi_smem, B);
  mi_table_init(t, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const size_t p0 = (size_t)blockIdx.x * kMiTile, pend = std::min(n, p0 + kMiTile);
  for (size_t p = p0; p < pend; ++p) {
    const double v = (bad1[p] || bad2[p]) ? __builtin_nan("") : mi_pair_wave(t, lane, B, cls1 + p, ld1, cls2 + p, ld2);
    if (lane == 0) out[p] = v;
  }
}

// Statistic::getValueForGroup of an AbstractMinimumStatistic (CoMap/Statistics.h:121-133): the smallest pairwise value,
// pairs (i, j < i) in the reference's order; "val < mini" with a NaN val never wins
__global__ __launch_bounds__(kWave) void mi_group_kernel(int B, const uint32_t* __restrict__ cls, const uint8_t* __restrict__ bad, size_t ld,
                                                         const int64_t* __restrict__ offsets, const int32_t* __restrict__ sites,
                                                         double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t mi_smem[];
  const int lane = threadIdx.x;
  const MiTable t = mi_table(mi_smem, B);
  mi_table_init(t, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const size_t g = blockIdx.x;
  const int32_t* mem = sites + offsets[g];
  const int m = (int)(offsets[g + 1] - offsets[g]);
  double best = __builtin_inf();
  for (int i = 1; i < m; ++i)
    for (int j = 0; j < i; ++j) {
      if (bad[mem[i]] || bad[mem[j]]) continue;
      const double v = mi_pair_wave(t, lane, B, cls + mem[i], ld, cls + mem[j], ld);
      if (v < best) best = v;
    }
  if (lane == 0) out[g] = best;
}

hipError_t launch_mi_classify(const double* d_counts, size_t n, size_t ldc, int B, int K, const double* d_bounds, int nb,
                              uint32_t* d_cls, size_t ldx, uint8_t* d_bad, hipStream_t stream) {
  hipLaunchKernelGGL(mi_classify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_counts, n, ldc, B, K, d_bounds, nb,
                     d_cls, ldx, d_bad);
  return hipGetLastError();
}

static hipError_t mi_lds_attr(const void* fn, size_t lds) {
  if (lds <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

hipError_t launch_mi_pairs_block(int B, const uint32_t* d_cls1, const uint8_t* d_bad1, size_t nrows, size_t ld1, const uint32_t* d_cls2,
                                 const uint8_t* d_bad2, size_t n2, size_t ld2, int intra, double* d_out, size_t ldo, size_t irow0,
                                 hipStream_t stream) {
  const size_t lds = mi_lds_bytes(B);
  hipError_t e = mi_lds_attr(reinterpret_cast<const void*>(&mi_pairs_block_kernel), lds);
  if (e != hipSuccess) return e;
  for (size_t r0 = 0; r0 < nrows; r0 += 65535) {   // grid.y limit
    const size_t rb = std::min<size_t>(65535, nrows - r0);
    hipLaunchKernelGGL(mi_pairs_block_kernel, dim3((unsigned)((n2 + kMiTile - 1) / kMiTile), (unsigned)rb), dim3(kWave), lds, stream, B,
                       d_cls1 + r0, d_bad1 + r0, ld1, d_cls2, d_bad2, n2, ld2, intra, d_out + r0 * ldo, ldo, irow0 + r0);
  }
  return hipGetLastError();
}

hipError_t launch_mi_pairs_diag(int B, const uint32_t* d_cls1, const uint8_t* d_bad1, size_t ld1, const uint32_t* d_cls2,
                                const uint8_t* d_bad2, size_t ld2, size_t n, double* d_out, hipStream_t stream) {
  const size_t lds = mi_lds_bytes(B);
  hipError_t e = mi_lds_attr(reinterpret_cast<const void*>(&mi_pairs_diag_kernel), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(mi_pairs_diag_kernel, dim3((unsigned)((n + kMiTile - 1) / kMiTile)), dim3(kWave), lds, stream, B, d_cls1, d_bad1, ld1,
                     d_cls2, d_bad2, ld2, n, d_out);
  return hipGetLastError();
}

hipError_t launch_mi_group(int B, const uint32_t* d_cls, const uint8_t* d_bad, size_t ld, const int64_t* d_offsets,
                           const int32_t* d_sites, size_t ngroups, double* d_out, hipStream_t stream) {
  if (ngroups == 0) return hipSuccess;
  const size_t lds = mi_lds_bytes(B);
  hipError_t e = mi_lds_attr(reinterpret_cast<const void*>(&mi_group_kernel), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(mi_group_kernel, dim3((unsigned)ngroups), dim3(kWave), lds, stream, B, d_cls, d_bad, ld, d_offsets, d_sites, d_out);
  return hipGetLastError();
}

size_t mi_bounds_lds_bytes(int B) { return mi_lds_bytes(B); }

}  // namespace cmx
