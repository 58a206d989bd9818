// Clustering analysis on the device: the distance matrix of CoMap/CoMap.cpp:432-440, the
// agglomeration bpp::HierarchicalClustering runs for CoMap.cpp:460-472, and the per-group properties of
// ClusterTools.cpp:302-320 / Distance.h:113-126, :353-366, :393-413 -- for a BATCH of matrices at once, which is what
// the null of ClusterTools::computeGlobalDistanceDistribution (ClusterTools.cpp:221-292) needs: every replicate is an
// independent n x n problem, so one workgroup owns one replicate and the chip works on hundreds of them.
//
// The agglomeration is a chain of n-1 dependent steps; what a step costs is latency, not bandwidth.  Per step the
// owning workgroup (a) finds the closest pair from per-row minima kept in LDS, (b) rewrites row/column i of its
// matrix in HBM with the linkage update while collecting the new minimum of row i on the fly, (c) rescans a row
// whose cached minimum was invalidated only if it becomes the head of the queue (lazy row minima).  Ties go to the first pair in index order, as the
// reference's scan does (CoMap/Cluster.cpp:55-79).  All integer/compare work plus one fp64 formula per element that
// is evaluated with explicit roundings, so the result is bit-identical to the CPU restatement.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <type_traits>
#include <cmath>

#include "cmx_device.h"

namespace cmx {

namespace {

#ifndef HC_THREADS
#define HC_THREADS 512
#endif
constexpr int kHcThreads = HC_THREADS;
constexpr int kHcWaves = kHcThreads / kWave;
constexpr int kHcMaxPerThread = (CMX_CLUSTER_MAX_SITES + kHcThreads - 1) / kHcThreads;   // columns of a row per thread
constexpr int kHcSeg = 512;       // columns of one rescan unit: 8 loads in flight per lane
constexpr int kHcMaxSeg = (CMX_CLUSTER_MAX_SITES + kHcSeg - 1) / kHcSeg;

__device__ __forceinline__ bool key_less(double v1, int c1, double v2, int c2) { return v1 < v2 || (v1 == v2 && c1 < c2); }

// Wave-wide minimum without LDS round trips: xor-butterfly over lane bits 0..3 with DPP (quad_perm / row_ror inside a
// row of 16 lanes), bits 4 and 5 with v_permlane16_swap / v_permlane32_swap.  Every lane ends up with the result.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppRor4 = 0x124, kDppRor8 = 0x128;
__device__ __forceinline__ double wave_min_f64(double v) {
  v = fmin(v, dpp_f64<kDppXor1>(v));
  v = fmin(v, dpp_f64<kDppXor2>(v));
  v = fmin(v, dpp_f64<kDppRor4>(v));
  v = fmin(v, dpp_f64<kDppRor8>(v));
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v = fmin(__hiloint2double((int)h16[0], (int)l16[0]), __hiloint2double((int)h16[1], (int)l16[1]));
  const unsigned lo2 = (unsigned)__double2loint(v), hi2 = (unsigned)__double2hiint(v);
  const auto l32 = __builtin_amdgcn_permlane32_swap(lo2, lo2, false, false);
  const auto h32 = __builtin_amdgcn_permlane32_swap(hi2, hi2, false, false);
  return fmin(__hiloint2double((int)h32[0], (int)l32[0]), __hiloint2double((int)h32[1], (int)l32[1]));
}
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, dpp_i32<kDppXor1>(v));
  v = min(v, dpp_i32<kDppXor2>(v));
  v = min(v, dpp_i32<kDppRor4>(v));
  v = min(v, dpp_i32<kDppRor8>(v));
  const auto s16 = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  v = min((int)s16[0], (int)s16[1]);
  const auto s32 = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return min((int)s32[0], (int)s32[1]);
}
// lexicographic (value, index) minimum across the wave (values are never NaN here; indices are >= 0 or INT_MAX)
__device__ __forceinline__ void wave_argmin(double& v, int& c) {
  const double m = wave_min_f64(v);
  c = wave_min_i32(v == m ? c : INT_MAX);
  v = m;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// upper triangle holds the pair statistic (j > i); turn it into the distance, mirror it, zero the diagonal.
// One 32 x 32 tile per workgroup, transposed through LDS so both the read and the mirrored write are coalesced.
__global__ __launch_bounds__(256) void dist_finish_kernel(double* Dall, size_t n, size_t ld, size_t mat_stride, double comp,
                                                          int negate) {
  __shared__ double tile[32][33];
  const size_t nt = (n + 31) / 32;
  const size_t by = blockIdx.x / nt, bx = blockIdx.x % nt;
  if (bx < by) return;
  double* D = Dall + (size_t)blockIdx.y * mat_stride;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const size_t i = by * 32 + r, j = bx * 32 + tx;
    double v = 0.0;
    if (i < n && j < n && i < j) {
      v = D[i * ld + j];
      v = negate ? comp - v : v;
    }
    tile[r][tx] = v;
    if (i < n && j < n && i <= j) D[i * ld + j] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const size_t row = bx * 32 + r, col = by * 32 + tx;
    if (row < n && col < n && row > col) D[row * ld + col] = tile[tx][r];
  }
}

// one wave per (matrix, row): NaN -> +inf over the whole row, nearest neighbour among the columns right of the diagonal
__global__ __launch_bounds__(256) void hc_rowmin_kernel(double* Dall, size_t ld, size_t mat_stride, int n, size_t total_rows,
                                                        double* grmin, int* gnn) {
  const int lane = threadIdx.x & 63;
  const size_t nw = (size_t)gridDim.x * (blockDim.x / kWave);
  for (size_t w = (size_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6); w < total_rows; w += nw) {
    const size_t m = w / n;
    const int r = (int)(w % n);
    double* row = Dall + m * mat_stride + (size_t)r * ld;
    double bv = INFINITY;
    int bc = INT_MAX;
    for (int c = lane; c < n; c += kWave) {
      double v = row[c];
      if (v != v) { v = INFINITY; row[c] = v; }
      if (c > r && key_less(v, c, bv, bc)) { bv = v; bc = c; }
    }
    wave_argmin(bv, bc);
    if (lane == 0) { grmin[w] = bv; gnn[w] = bc == INT_MAX ? -1 : bc; }
  }
}

struct HcArgs {
  double* D;           // [batch] matrices, destroyed
  size_t ld, mat_stride;
  int n;
  const double* grmin; // [batch][n]
  const int* gnn;      // [batch][n]
  int32_t* merge;      // [batch][n-1][2]
  double* dmax;        // [batch][n-1]
  int32_t* size;       // [batch][n-1]
};

template <int LINK>
__device__ __forceinline__ double linkage_update(double x, double y, double ni, double nj) {
#pragma clang fp contract(off)   // (ni*x + nj*y)/(ni + nj) with every operation rounded on its own, like the CPU
  if (LINK == CMX_LINK_COMPLETE) return fmax(x, y);
  if (LINK == CMX_LINK_SINGLE) return fmin(x, y);
  const double p = ni * x, q = nj * y;
  const double s = p + q;
  return s / (ni + nj);
}

// Row minima are kept LAZILY (the scheme of Muellner's generic linkage algorithm, arXiv:1109.2378 sec. 3.3): when a
// join removes or enlarges the cached minimum of a row, the row is only marked stale and its old minimum kept as a
// lower bound -- for these three linkages the entries of a row never drop below the row's previous minimum unless
// the new entry itself is the smaller one, which is handled on the spot.  A stale row is rescanned when (and only
// when) its bound makes it the head of the queue.  Exactness of the tie rule: the head is the smallest (bound, row);
// a stale row elsewhere has true minimum >= its bound, so it cannot precede a fresh head in (value, row) order.
// Barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global store of the wave
// (s_waitcnt vmcnt(0)); inside a step nobody reads what was just stored to the matrix or to the outputs, so that wait
// is taken once per step -- at the full barrier that precedes the next loads from the matrix -- instead of three times.
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kStale = -2;   // nn[r]: minimum unknown, rmin[r] is a lower bound
constexpr int kNone = -1;    // nn[r]: no live column right of the diagonal

// PER = columns of a row per thread (compile-time, so the per-thread row slices live in registers without predicated
// dead iterations): the launcher picks the smallest instantiated value >= ceil(n / kHcThreads)
template <int LINK, int PER>
__global__ __launch_bounds__(kHcThreads) void hclust_kernel(HcArgs a) {
  extern __shared__ double hc_smem[];
  const int n = a.n;
  double* rmin = hc_smem;                          // [n] smallest distance right of the diagonal in row r (or a bound)
  int* nn = reinterpret_cast<int*>(rmin + n);      // [n] its column, kNone, kStale
  int* cid = nn + n;                               // [n] node id of the cluster living in slot r, -1: merged away
  int* csz = cid + n;                              // [n] its number of leaves
  __shared__ double pv[2][kHcWaves];
  __shared__ int pi[2][kHcWaves];
  __shared__ double sv[kHcMaxSeg];
  __shared__ int sc[kHcMaxSeg];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* D = a.D + (size_t)blockIdx.x * a.mat_stride;
  const size_t ld = a.ld;
  const size_t ob = (size_t)blockIdx.x * (size_t)(n - 1);
  const int nseg = (n + kHcSeg - 1) / kHcSeg;
  const unsigned ld32 = (unsigned)ld;                  // n <= CMX_CLUSTER_MAX_SITES: element offsets fit 32 bits
  for (int r = tid; r < n; r += kHcThreads) {
    rmin[r] = a.grmin[(size_t)blockIdx.x * n + r];
    nn[r] = a.gnn[(size_t)blockIdx.x * n + r];
    cid[r] = r;
    csz[r] = 1;
  }
  __syncthreads();
  for (int step = 0; step < n - 1; ++step) {
    // ---- head of the queue: a reduction over n LDS entries; rescan it while it is stale (each pass freshens one row)
    double gv;
    int i;
    for (;;) {
      double bv = INFINITY;
      int br = INT_MAX;
#pragma unroll
      for (int t = 0; t < PER; ++t) {      // merged-away rows carry nn = kNone
        const int r = tid + t * kHcThreads;
        if (r < n && nn[r] != kNone && key_less(rmin[r], r, bv, br)) { bv = rmin[r]; br = r; }
      }
      wave_argmin(bv, br);
      if (lane == 0) { pv[0][wave] = bv; pi[0][wave] = br; }
      __syncthreads();   // full: the matrix stores of the previous step are complete before anyone loads from it
      gv = pv[0][0];
      i = pi[0][0];
#pragma unroll
      for (int w = 1; w < kHcWaves; ++w)
        if (key_less(pv[0][w], pi[0][w], gv, i)) { gv = pv[0][w]; i = pi[0][w]; }
      if (nn[i] != kStale) break;
      // one wave per segment of kHcSeg columns, all loads of a segment in flight together
      for (int u = wave; u < nseg; u += kHcWaves) {
        const int base = i + 1 + u * kHcSeg;
        const double* row = D + (size_t)i * ld;
        double x[kHcSeg / kWave];
#pragma unroll
        for (int t = 0; t < kHcSeg / kWave; ++t) {
          const int k = base + lane + t * kWave;
          x[t] = (k < n && cid[k < n ? k : 0] >= 0) ? row[k] : INFINITY;
        }
        double v = INFINITY;
        int c = INT_MAX;
#pragma unroll
        for (int t = 0; t < kHcSeg / kWave; ++t) {
          const int k = base + lane + t * kWave;
          if (k < n && cid[k < n ? k : 0] >= 0 && key_less(x[t], k, v, c)) { v = x[t]; c = k; }
        }
        wave_argmin(v, c);
        if (lane == 0) { sv[u] = v; sc[u] = c; }
      }
      barrier_lds();
      if (tid == 0) {
        double v = sv[0];
        int c = sc[0];
        for (int g = 1; g < nseg; ++g)
          if (key_less(sv[g], sc[g], v, c)) { v = sv[g]; c = sc[g]; }
        rmin[i] = v;
        nn[i] = c == INT_MAX ? kNone : c;
      }
      barrier_lds();
    }
    const int j = nn[i];
    const double ni = (double)csz[i], nj = (double)csz[j];
    // ---- linkage update of row/column i; the new minimum of row i falls out of the same loop.  All loads of the
    // step are issued before the first store (the compiler cannot prove the stores do not alias the next loads)
    double cv = INFINITY;
    int cc = INT_MAX;
    double xi[PER], xj[PER];
    const double* rowi = D + (size_t)i * ld;
    const double* rowj = D + (size_t)j * ld;
    double* coli = D + i;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int k = tid + t * kHcThreads;
      const bool on = k < n && k != i && k != j && cid[k < n ? k : 0] >= 0;
      xi[t] = on ? rowi[k] : 0.0;
      xj[t] = on ? rowj[k] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int k = tid + t * kHcThreads;
      if (k >= n || k == i || k == j || cid[k] < 0) continue;
      const double nw = linkage_update<LINK>(xi[t], xj[t], ni, nj);
      D[(size_t)i * ld + k] = nw;
      coli[(unsigned)k * ld32] = nw;
      if (k > i) {
        if (key_less(nw, k, cv, cc)) { cv = nw; cc = k; }
        if (k < j && nn[k] == j) nn[k] = kStale;                     // its neighbour disappears; the bound stands
      } else {
        const int nk = nn[k];
        if (nk == i || nk == j) {
          // (k, i) took the place of the cached minimum: still the minimum unless it grew
          if (nw <= rmin[k]) { rmin[k] = nw; nn[k] = i; }
          else nn[k] = kStale;
        } else if (nk == kStale) {
          if (nw < rmin[k]) { rmin[k] = nw; nn[k] = i; }              // below every other entry of the row: fresh again
        } else if (key_less(nw, i, rmin[k], nk)) { rmin[k] = nw; nn[k] = i; }
      }
    }
    wave_argmin(cv, cc);
    if (lane == 0) { pv[1][wave] = cv; pi[1][wave] = cc; }
    barrier_lds();
    if (tid == 0) {
      double v = pv[1][0];
      int c = pi[1][0];
      for (int w = 1; w < kHcWaves; ++w)
        if (key_less(pv[1][w], pi[1][w], v, c)) { v = pv[1][w]; c = pi[1][w]; }
      rmin[i] = v;
      nn[i] = c == INT_MAX ? kNone : c;
      a.merge[(ob + step) * 2] = cid[i];
      a.merge[(ob + step) * 2 + 1] = cid[j];
      a.dmax[ob + step] = gv;
      const int sz = csz[i] + csz[j];
      a.size[ob + step] = sz;
      csz[i] = sz;
      cid[i] = n + step;
      cid[j] = -1;
      nn[j] = kNone;
    }
    barrier_lds();
  }
}

// per-group properties, one workgroup per replicate, merges in order (a node's sons are always older merges)
struct PropArgs {
  int n, B, K, dist_kind;
  const int32_t* merge;   // [batch][n-1][2]
  const double* dmax;     // [batch][n-1]
  const double* norm;     // site r of replicate b at norm[b*site_stride + r]
  const double* counts;   // [B*K][ldc], site r of replicate b in column b*site_stride + r
  size_t ldc, site_stride;
  double* sigma;          // [batch][2n-1][B] scratch (compensation only)
  double* stat;           // [batch][n-1]
  double* nmin;           // [batch][n-1]
};

__global__ __launch_bounds__(256) void cluster_props_kernel(PropArgs a) {
  extern __shared__ double pr_smem[];
  const int n = a.n, B = a.B;
  double* nm = pr_smem;               // [2n-1] smallest norm in the subtree
  double* sn = nm + (2 * n - 1);      // [2n-1] sum of norms
  __shared__ double part[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t b = blockIdx.x, ob = b * (size_t)(n - 1);
  const bool comp = a.dist_kind == CMX_DIST_COMPENSATION;
  double* sg = comp ? a.sigma + b * (size_t)(2 * n - 1) * B : nullptr;
  for (int r = tid; r < n; r += 256) nm[r] = sn[r] = a.norm[b * a.site_stride + r];
  if (comp)
    for (int br = 0; br < B; ++br)
      for (int r = tid; r < n; r += 256) {
        double s = 0.0;
        for (int k = 0; k < a.K; ++k) s += a.counts[(size_t)(br * a.K + k) * a.ldc + b * a.site_stride + r];
        sg[(size_t)r * B + br] = s;
      }
  __syncthreads();
  for (int m = 0; m < n - 1; ++m) {
    const int x = a.merge[(ob + m) * 2], y = a.merge[(ob + m) * 2 + 1];
    double st;
    if (comp) {
      double sq = 0.0;
      for (int t = tid; t < B; t += 256) {     // thread t owns component t of every node's vector
        const double s = sg[(size_t)x * B + t] + sg[(size_t)y * B + t];
        sg[(size_t)(n + m) * B + t] = s;
        sq += s * s;
      }
      sq = wave_sum(sq);
      if (lane == 0) part[wave] = sq;
      __syncthreads();
      st = 1.0 - sqrt(part[0] + part[1] + part[2] + part[3]) / (sn[x] + sn[y]);
    } else {
      const double d = a.dmax[ob + m];
      st = a.dist_kind == CMX_DIST_EUCLIDIAN ? d : 1.0 - d;
    }
    if (tid == 0) {
      nm[n + m] = fmin(nm[x], nm[y]);
      sn[n + m] = sn[x] + sn[y];
      a.stat[ob + m] = st;
      a.nmin[ob + m] = nm[n + m];
    }
    __syncthreads();
  }
}

}  // namespace

size_t hclust_lds_bytes(int n) { return (size_t)n * (sizeof(double) + 3 * sizeof(int)); }
size_t cluster_props_lds_bytes(int n) { return (size_t)(2 * n - 1) * 2 * sizeof(double); }

hipError_t launch_dist_finish(int dist_kind, double* d_D, size_t n, size_t ld, size_t mat_stride, size_t batch,
                              hipStream_t stream) {
  const size_t nt = (n + 31) / 32;
  hipLaunchKernelGGL(dist_finish_kernel, dim3((unsigned)(nt * nt), (unsigned)batch), dim3(256), 0, stream, d_D, n, ld,
                     mat_stride, 1.0, dist_kind == CMX_DIST_EUCLIDIAN ? 0 : 1);
  return hipGetLastError();
}

hipError_t launch_hclust(int linkage, double* d_D, size_t n, size_t ld, size_t mat_stride, size_t batch, double* d_rmin,
                         int* d_nn, int32_t* d_merge, double* d_dmax, int32_t* d_size, hipStream_t stream) {
  const size_t rows = batch * n;
  const unsigned grid = (unsigned)std::min<size_t>((rows + 3) / 4, 256 * 32);
  hipLaunchKernelGGL(hc_rowmin_kernel, dim3(grid), dim3(256), 0, stream, d_D, ld, mat_stride, (int)n, rows, d_rmin, d_nn);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  HcArgs a{d_D, ld, mat_stride, (int)n, d_rmin, d_nn, d_merge, d_dmax, d_size};
  const size_t lds = hclust_lds_bytes((int)n);
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e2 != hipSuccess) return e2;
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(kHcThreads), lds, stream, a);
    return hipGetLastError();
  };
  const int per = (int)((n + kHcThreads - 1) / kHcThreads);
  auto pick = [&](auto link) -> hipError_t {
    constexpr int L = decltype(link)::value;
    if (per <= 1) return go(hclust_kernel<L, 1>);
    if (per <= 2) return go(hclust_kernel<L, 2>);
    if (per <= 4) return go(hclust_kernel<L, 4>);
    if (per <= 6) return go(hclust_kernel<L, 6>);
    return go(hclust_kernel<L, kHcMaxPerThread>);
  };
  switch (linkage) {
    case CMX_LINK_COMPLETE: return pick(std::integral_constant<int, CMX_LINK_COMPLETE>());
    case CMX_LINK_SINGLE: return pick(std::integral_constant<int, CMX_LINK_SINGLE>());
    case CMX_LINK_AVERAGE: return pick(std::integral_constant<int, CMX_LINK_AVERAGE>());
  }
  return hipErrorInvalidValue;
}

hipError_t launch_cluster_props(int dist_kind, int n, int B, int K, size_t batch, const int32_t* d_merge, const double* d_dmax,
                                const double* d_norm, const double* d_counts, size_t ldc, size_t site_stride, double* d_sigma,
                                double* d_stat, double* d_nmin, hipStream_t stream) {
  PropArgs a{n, B, K, dist_kind, d_merge, d_dmax, d_norm, d_counts, ldc, site_stride, d_sigma, d_stat, d_nmin};
  const size_t lds = cluster_props_lds_bytes(n);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cluster_props_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(cluster_props_kernel, dim3((unsigned)batch), dim3(256), lds, stream, a);
  return hipGetLastError();
}

}  // namespace cmx
