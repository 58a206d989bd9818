// Clustering analysis on the device: the distance matrix of CoMap/CoMap.cpp:432-440, the
// agglomeration bpp::HierarchicalClustering runs for CoMap.cpp:460-472, and the per-group properties of
// ClusterTools.cpp:302-320 / Distance.h:113-126, :353-366, :393-413 -- for a BATCH of matrices at once, which is what
// the null of ClusterTools::computeGlobalDistanceDistribution (ClusterTools.cpp:221-292) needs: every replicate is an
// independent n x n problem, so one workgroup owns one replicate and the chip works on hundreds of them.
//
// The agglomeration is a chain of n-1 dependent steps; what a step costs is latency, not bandwidth.  Per step the
// owning workgroup (a) finds the closest pair from per-row minima kept in LDS, (b) rewrites row/column i of its
// matrix in HBM with the linkage update while collecting the new minimum of row i on the fly, (c) rescans only
// the rows whose cached nearest neighbour was invalidated.  Ties go to the first pair in index order, as the
// reference's scan does (CoMap/Cluster.cpp:55-79).  All integer/compare work plus one fp64 formula per element that
// is evaluated with explicit roundings, so the result is bit-identical to the CPU restatement.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>

#include "cmx_device.h"

namespace cmx {

namespace {

constexpr int kHcThreads = 512;
constexpr int kHcWaves = kHcThreads / kWave;

__device__ __forceinline__ bool key_less(double v1, int c1, double v2, int c2) { return v1 < v2 || (v1 == v2 && c1 < c2); }

// lexicographic (value, index) minimum across the wave; every lane ends up with the result
__device__ __forceinline__ void wave_argmin(double& v, int& c) {
#pragma unroll
  for (int off = 32; off; off >>= 1) {
    const double ov = __shfl_xor(v, off);
    const int oc = __shfl_xor(c, off);
    if (key_less(ov, oc, v, c)) { v = ov; c = oc; }
  }
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

template <int LINK>
__global__ __launch_bounds__(kHcThreads) void hclust_kernel(HcArgs a) {
  extern __shared__ double hc_smem[];
  const int n = a.n;
  double* rmin = hc_smem;                          // [n] smallest distance right of the diagonal in row r
  int* nn = reinterpret_cast<int*>(rmin + n);      // [n] its column, -1: none
  int* cid = nn + n;                               // [n] node id of the cluster living in slot r, -1: merged away
  int* csz = cid + n;                              // [n] its number of leaves
  int* list = csz + n;                             // [n] rows to rescan this step
  __shared__ double pv[2][kHcWaves];
  __shared__ int pi[2][kHcWaves];
  __shared__ int lcount;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* D = a.D + (size_t)blockIdx.x * a.mat_stride;
  const size_t ld = a.ld;
  const size_t ob = (size_t)blockIdx.x * (size_t)(n - 1);
  for (int r = tid; r < n; r += kHcThreads) {
    rmin[r] = a.grmin[(size_t)blockIdx.x * n + r];
    nn[r] = a.gnn[(size_t)blockIdx.x * n + r];
    cid[r] = r;
    csz[r] = 1;
  }
  __syncthreads();
  for (int step = 0; step < n - 1; ++step) {
    // ---- closest pair: rows carry their own minimum, so this is a reduction over n LDS entries
    double bv = INFINITY;
    int br = INT_MAX;
    for (int r = tid; r < n; r += kHcThreads)
      if (cid[r] >= 0 && nn[r] >= 0 && key_less(rmin[r], r, bv, br)) { bv = rmin[r]; br = r; }
    wave_argmin(bv, br);
    if (lane == 0) { pv[0][wave] = bv; pi[0][wave] = br; }
    if (tid == 0) lcount = 0;
    __syncthreads();
    double gv = pv[0][0];
    int i = pi[0][0];
#pragma unroll
    for (int w = 1; w < kHcWaves; ++w)
      if (key_less(pv[0][w], pi[0][w], gv, i)) { gv = pv[0][w]; i = pi[0][w]; }
    const int j = nn[i];
    const double ni = (double)csz[i], nj = (double)csz[j];
    // ---- linkage update of row/column i; the new minimum of row i falls out of the same loop
    double cv = INFINITY;
    int cc = INT_MAX;
    for (int k = tid; k < n; k += kHcThreads) {
      if (k == i || k == j || cid[k] < 0) continue;
      const double nw = linkage_update<LINK>(D[(size_t)i * ld + k], D[(size_t)j * ld + k], ni, nj);
      D[(size_t)i * ld + k] = nw;
      D[(size_t)k * ld + i] = nw;
      if (k > i) {
        if (key_less(nw, k, cv, cc)) { cv = nw; cc = k; }
        if (k < j && nn[k] == j) list[atomicAdd(&lcount, 1)] = k;     // its neighbour disappears
      } else {
        const int nk = nn[k];
        if (nk == i || nk == j) {
          // (k, i) took the place of the cached minimum: still the minimum unless it grew
          if (nw <= rmin[k]) { rmin[k] = nw; nn[k] = i; }
          else list[atomicAdd(&lcount, 1)] = k;
        } else if (key_less(nw, i, rmin[k], nk)) { rmin[k] = nw; nn[k] = i; }
      }
    }
    wave_argmin(cv, cc);
    if (lane == 0) { pv[1][wave] = cv; pi[1][wave] = cc; }
    __syncthreads();
    // ---- bookkeeping (one lane) and rescans (one wave per invalidated row)
    if (tid == 0) {
      double v = pv[1][0];
      int c = pi[1][0];
      for (int w = 1; w < kHcWaves; ++w)
        if (key_less(pv[1][w], pi[1][w], v, c)) { v = pv[1][w]; c = pi[1][w]; }
      rmin[i] = v;
      nn[i] = c == INT_MAX ? -1 : c;
      a.merge[(ob + step) * 2] = cid[i];
      a.merge[(ob + step) * 2 + 1] = cid[j];
      a.dmax[ob + step] = gv;
      const int sz = csz[i] + csz[j];
      a.size[ob + step] = sz;
      csz[i] = sz;
      cid[i] = n + step;
      cid[j] = -1;
    }
    const int nl = lcount;
    for (int idx = wave; idx < nl; idx += kHcWaves) {
      const int r = list[idx];
      const double* row = D + (size_t)r * ld;
      double v = INFINITY;
      int c = INT_MAX;
      for (int k = r + 1 + lane; k < n; k += kWave)
        if (k != j && cid[k] >= 0 && key_less(row[k], k, v, c)) { v = row[k]; c = k; }
      wave_argmin(v, c);
      if (lane == 0) { rmin[r] = v; nn[r] = c == INT_MAX ? -1 : c; }
    }
    __syncthreads();
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

size_t hclust_lds_bytes(int n) { return (size_t)n * (sizeof(double) + 4 * sizeof(int)); }
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
  switch (linkage) {
    case CMX_LINK_COMPLETE: return go(hclust_kernel<CMX_LINK_COMPLETE>);
    case CMX_LINK_SINGLE: return go(hclust_kernel<CMX_LINK_SINGLE>);
    case CMX_LINK_AVERAGE: return go(hclust_kernel<CMX_LINK_AVERAGE>);
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
