// The non-default mapping variants (nijt.average / nijt.joint, "for benchmarking only" in the reference).  First
// nijt.average = no, nijt.joint = yes: LegacySubstitutionMappingTools::computeSubstitutionVectorsNoAveraging (call sites
// CoMap/CoETools.cpp:395-403 for the observed data, CoMap/AnalysisTools.cpp:598-610 inside the null).  "For benchmarking
// only" says the reference -- but it is the only way it runs nijt = Label with the MI statistic (CoETools.cpp:577-588).
//   per branch b (father f, son n), site i:  pxy(x, y) = sum_c p_c U_b(i,c,x) P_c,b(x,y) D_n(i,c,y);
//   (x*, y*) = first maximum of pxy in row-major order (MatrixTools::whichMax);  count(b, i, k) = N^k(x*, y*; t_b).
// The algorithm is bpp-phyl's (absent from the reference tree): restated in oracle/oracle.c orc_map_sites_noavg, which
// tests/test_oracle_noavg.py pins to the definition by brute force; this file follows that restatement loop for loop.
// This is NOT the hot path: three plain kernels, one thread per (site, class) or (site, branch), every
// per-node vector in a global scratch of [class][node][state][site] (coalesced over sites; operators are wave-uniform
// and come through the scalar cache).  The likelihood, posterior rate and rate class of a site do not depend on the
// mapping variant and still come from the mapping kernel.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "cmx_device.h"

namespace cmx {

constexpr int kPlainStatesDev = 64;   // == kPlainStates (cmx_host_model.h): padded state count of the plain path

namespace {

template <int S>
__global__ __launch_bounds__(256) void noavg_inside_kernel(const NoAvgArgs a) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= a.nsites) return;
  const int c = blockIdx.y, nn = a.nn;
  const size_t ch = a.chunk;
  double* Dc = a.D + (size_t)c * nn * S * ch;
  double* Mc = a.M + (size_t)c * nn * S * ch;
  const uint32_t all = S >= 32 ? 0xffffffffu : ((1u << (S & 31)) - 1u);
  for (int n = 0; n < nn; ++n) {
    double d[S];
    if (a.first_child[n] < 0) {
      const unsigned code = a.aln[(size_t)a.taxon_of[n] * a.ld + a.site0 + j];
      if constexpr (S > 32) {   // plain path (padded states): no mask table, every code >= Sreal is an unknown
#pragma unroll
        for (int x = 0; x < S; ++x) d[x] = code < (unsigned)a.Sreal ? (double)((unsigned)x == code) : (double)(x < a.Sreal);
      } else {
        const unsigned row = code < (unsigned)(S + max_ambig(S)) ? code : (unsigned)(S + max_ambig(S) - 1);
        const uint32_t m = code < (unsigned)S ? (1u << code) : (a.masks ? a.masks[row] : all);
#pragma unroll
        for (int x = 0; x < S; ++x) d[x] = (double)((m >> x) & 1u);
      }
    } else {
#pragma unroll
      for (int x = 0; x < S; ++x) d[x] = 1.0;
      for (int e = a.first_child[n]; e >= 0; e = a.next_sib[e]) {
#pragma unroll
        for (int x = 0; x < S; ++x) d[x] *= Mc[((size_t)e * S + x) * ch + j];
      }
    }
#pragma unroll
    for (int x = 0; x < S; ++x) Dc[((size_t)n * S + x) * ch + j] = d[x];
    if (n != a.root) {
      const double* Pn = a.P + ((size_t)c * a.B + n) * S * S;
#pragma unroll
      for (int x = 0; x < S; ++x) {
        double s = 0.0;
#pragma unroll
        for (int z = 0; z < S; ++z) s += Pn[x * S + z] * d[z];
        Mc[((size_t)n * S + x) * ch + j] = s;
      }
    }
  }
}

template <int S>
__global__ __launch_bounds__(256) void noavg_outside_kernel(const NoAvgArgs a) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= a.nsites) return;
  const int c = blockIdx.y, nn = a.nn;
  const size_t ch = a.chunk;
  const double* Mc = a.M + (size_t)c * nn * S * ch;
  double* Uc = a.U + (size_t)c * nn * S * ch;
  double* Upc = a.Up + (size_t)c * nn * S * ch;
#pragma unroll
  for (int x = 0; x < S; ++x) Upc[((size_t)a.root * S + x) * ch + j] = a.pi[x];
  for (int f = nn - 1; f >= 0; --f) {
    if (a.first_child[f] < 0) continue;
    double upf[S];
#pragma unroll
    for (int x = 0; x < S; ++x) upf[x] = Upc[((size_t)f * S + x) * ch + j];
    for (int n = a.first_child[f]; n >= 0; n = a.next_sib[n]) {
      double u[S];
#pragma unroll
      for (int x = 0; x < S; ++x) u[x] = upf[x];
      for (int m = a.first_child[f]; m >= 0; m = a.next_sib[m])
        if (m != n) {
#pragma unroll
          for (int x = 0; x < S; ++x) u[x] *= Mc[((size_t)m * S + x) * ch + j];
        }
#pragma unroll
      for (int x = 0; x < S; ++x) Uc[((size_t)n * S + x) * ch + j] = u[x];
      if (a.first_child[n] >= 0) {
        const double* Pn = a.P + ((size_t)c * a.B + n) * S * S;
#pragma unroll
        for (int z = 0; z < S; ++z) {
          double s = 0.0;
#pragma unroll
          for (int x = 0; x < S; ++x) s += Pn[x * S + z] * u[x];
          Upc[((size_t)n * S + z) * ch + j] = s;
        }
      }
    }
  }
}

template <int S>
__global__ __launch_bounds__(256) void noavg_pick_kernel(const NoAvgArgs a) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= a.nsites) return;
  const int b = blockIdx.y, nn = a.nn, C = a.C;
  const size_t ch = a.chunk;
  double best = -__builtin_inf();
  int bidx = 0;
  for (int x = 0; x < S; ++x)
    for (int y = 0; y < S; ++y) {
      double s = 0.0;
      for (int c = 0; c < C; ++c) {
        const double u = a.U[(((size_t)c * nn + b) * S + x) * ch + j];
        const double d = a.D[(((size_t)c * nn + b) * S + y) * ch + j];
        s += a.probs[c] * ((u * a.P[((size_t)c * a.B + b) * S * S + x * S + y]) * d);
      }
      if (s > best) { best = s; bidx = x * S + y; }
    }
  for (int k = 0; k < a.K; ++k)
    a.counts[((size_t)b * a.K + k) * a.ldc + a.site0 + j] = a.N1[((size_t)b * a.K + k) * S * S + bidx];
}

// ---- nijt.joint = no (computeSubstitutionVectorsMarginal / ...NoAveragingMarginal, CoETools.cpp:399-405).  Restated in
// oracle/oracle.c orc_map_sites_marginal (pinned to its definition by tests/test_oracle_marginal.py; parity unpinned against
// the reference, which ships no output of these variants):
//   post_n(c, x) = p_c Up_n(c, x) D_n(c, x) / L  at an internal node (DRTreeLikelihoodTools::
//   getPosteriorProbabilitiesPerStatePerRate: the likelihood re-rooted at the node; Up_root = pi), e(x) p_c / sum e at a leaf;
//   Marginal:            count(b, k) = sum_c sum_x sum_y post_f(c, x) post_n(c, y) N^k(x, y; r_c t_b)
//   NoAveragingMarginal: x*(n) = first maximum of sum_c post_n(c, x) (leaf: of e);  count(b, k) = N^k(x*(f), x*(n); t_b)
// One thread per (site, branch); same global scratch as the NoAveraging kernels above.
template <int S>
__global__ __launch_bounds__(256) void marginal_kernel(const NoAvgArgs a) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= a.nsites) return;
  const int b = blockIdx.y, nn = a.nn, C = a.C, f = a.parent[b];
  const size_t ch = a.chunk;
  const bool leaf = a.first_child[b] < 0;
  double L = 0.0;
  for (int c = 0; c < C; ++c) {
    double s = 0.0;
    for (int x = 0; x < S; ++x) s += a.pi[x] * a.D[(((size_t)c * nn + a.root) * S + x) * ch + j];
    L += a.probs[c] * s;
  }
  double se = 0.0;
  if (leaf)
    for (int x = 0; x < S; ++x) se += a.D[((size_t)b * S + x) * ch + j];   // the leaf vector is the same in every class
  if (a.mode == kVariantMarginal) {
    for (int k = 0; k < a.K; ++k) {
      double v = 0.0;
      for (int c = 0; c < C; ++c) {
        const double* Nk = a.NC + (((size_t)c * a.B + b) * a.K + k) * S * S;
        double pn[S];
#pragma unroll
        for (int y = 0; y < S; ++y) {
          const double d = a.D[(((size_t)c * nn + b) * S + y) * ch + j];
          pn[y] = leaf ? d * a.probs[c] / se : a.Up[(((size_t)c * nn + b) * S + y) * ch + j] * d * a.probs[c] / L;
        }
        for (int x = 0; x < S; ++x) {
          const double pf = a.Up[(((size_t)c * nn + f) * S + x) * ch + j] * a.D[(((size_t)c * nn + f) * S + x) * ch + j] * a.probs[c] / L;
          double s = 0.0;
#pragma unroll
          for (int y = 0; y < S; ++y) s += Nk[x * S + y] * pn[y];
          v += pf * s;
        }
      }
      a.counts[((size_t)b * a.K + k) * a.ldc + a.site0 + j] = v;
    }
    return;
  }
  // marginal ancestral states of the father and of the node: first maximum over the states
  int xs = 0, ys = 0;
  double bf = -__builtin_inf(), bn = -__builtin_inf();
  for (int x = 0; x < S; ++x) {
    double sf = 0.0, sn = 0.0;
    for (int c = 0; c < C; ++c) {
      sf += a.Up[(((size_t)c * nn + f) * S + x) * ch + j] * a.D[(((size_t)c * nn + f) * S + x) * ch + j] * a.probs[c] / L;
      if (!leaf) sn += a.Up[(((size_t)c * nn + b) * S + x) * ch + j] * a.D[(((size_t)c * nn + b) * S + x) * ch + j] * a.probs[c] / L;
    }
    if (leaf) sn = a.D[((size_t)b * S + x) * ch + j];
    if (sf > bf) { bf = sf; xs = x; }
    if (sn > bn) { bn = sn; ys = x; }
  }
  for (int k = 0; k < a.K; ++k)
    a.counts[((size_t)b * a.K + k) * a.ldc + a.site0 + j] = a.N1[((size_t)b * a.K + k) * S * S + xs * S + ys];
}

// ---- the default mapping (computeSubstitutionVectors: nijt.average = yes, nijt.joint = yes) on the same per-node vectors,
// for the alphabets the matrix-core walk does not serve (codon models, CoETools.cpp:95-100):
//   count(b, i, k) = sum_c p_c sum_xy U_b(i,c,x) (P o N^k)_c,b(x,y) D_n(i,c,y) / L_i
// One thread per (site, branch).
template <int S>
__global__ __launch_bounds__(256) void joint_kernel(const NoAvgArgs a) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= a.nsites) return;
  const int b = blockIdx.y, nn = a.nn, C = a.C;
  const size_t ch = a.chunk;
  double L = 0.0;
  for (int c = 0; c < C; ++c) {
    double s = 0.0;
    for (int x = 0; x < S; ++x) s += a.pi[x] * a.D[(((size_t)c * nn + a.root) * S + x) * ch + j];
    L += a.probs[c] * s;
  }
  for (int k = 0; k < a.K; ++k) {
    double v = 0.0;
    for (int c = 0; c < C; ++c) {
      const double* PNk = a.PN + (((size_t)c * a.B + b) * a.K + k) * S * S;
      double d[S];
#pragma unroll
      for (int y = 0; y < S; ++y) d[y] = a.D[(((size_t)c * nn + b) * S + y) * ch + j];
      double vc = 0.0;
      for (int x = 0; x < S; ++x) {
        double s = 0.0;
#pragma unroll
        for (int y = 0; y < S; ++y) s += PNk[x * S + y] * d[y];
        vc += a.U[(((size_t)c * nn + b) * S + x) * ch + j] * s;
      }
      v += a.probs[c] * vc;
    }
    a.counts[((size_t)b * a.K + k) * a.ldc + a.site0 + j] = v / L;
  }
}
// site scalars of the plain path: log-likelihood, posterior rate, rate class with the largest posterior (first maximum)
template <int S>
__global__ __launch_bounds__(256) void site_scalars_kernel(const NoAvgArgs a) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= a.nsites) return;
  double L = 0.0, pr = 0.0, best = -1.0;
  int bc = 0;
  for (int c = 0; c < a.C; ++c) {
    double s = 0.0;
    for (int x = 0; x < S; ++x) s += a.pi[x] * a.D[(((size_t)c * a.nn + a.root) * S + x) * a.chunk + j];
    const double w = a.probs[c] * s;
    L += w;
    pr += a.rates[c] * w;
    if (w > best) { best = w; bc = c; }
  }
  if (a.logL) a.logL[a.site0 + j] = log(L);
  if (a.post_rate) a.post_rate[a.site0 + j] = pr / L;
  if (a.rate_class) a.rate_class[a.site0 + j] = bc;
}

// computeNormForSite over the (branch-major) counts: sqrt(sum_b (sum_k count)^2), branches in order
__global__ __launch_bounds__(256) void counts_norm_kernel(const double* __restrict__ counts, size_t ldc, int B, int K, size_t n,
                                                          double* __restrict__ norm) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  double nrm = 0.0;
  for (int b = 0; b < B; ++b) {
    double tot = 0.0;
    for (int k = 0; k < K; ++k) tot += counts[((size_t)b * K + k) * ldc + j];
    nrm += tot * tot;
  }
  norm[j] = sqrt(nrm);
}

}  // namespace

size_t noavg_scratch_doubles(int S, int C, int nn, size_t chunk) { return 4 * (size_t)C * nn * S * chunk; }

hipError_t launch_map_noavg(NoAvgArgs a, size_t nsites_total, double* scratch, double* d_norm, hipStream_t stream) {
  const size_t per = (size_t)a.C * a.nn * a.S * a.chunk;
  a.D = scratch; a.M = scratch + per; a.U = scratch + 2 * per; a.Up = scratch + 3 * per;
  for (size_t s0 = 0; s0 < nsites_total; s0 += a.chunk) {
    a.site0 = s0;
    a.nsites = std::min(a.chunk, nsites_total - s0);
    const unsigned gx = (unsigned)((a.nsites + 255) / 256);
    if (a.S == 20) {
      hipLaunchKernelGGL(noavg_inside_kernel<20>, dim3(gx, a.C), dim3(256), 0, stream, a);
      hipLaunchKernelGGL(noavg_outside_kernel<20>, dim3(gx, a.C), dim3(256), 0, stream, a);
      if (a.mode == kVariantNoAvg) hipLaunchKernelGGL(noavg_pick_kernel<20>, dim3(gx, a.B), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL(marginal_kernel<20>, dim3(gx, a.B), dim3(256), 0, stream, a);
    } else if (a.S == 4) {
      hipLaunchKernelGGL(noavg_inside_kernel<4>, dim3(gx, a.C), dim3(256), 0, stream, a);
      hipLaunchKernelGGL(noavg_outside_kernel<4>, dim3(gx, a.C), dim3(256), 0, stream, a);
      if (a.mode == kVariantNoAvg) hipLaunchKernelGGL(noavg_pick_kernel<4>, dim3(gx, a.B), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL(marginal_kernel<4>, dim3(gx, a.B), dim3(256), 0, stream, a);
    } else if (a.S == kPlainStatesDev) {
      hipLaunchKernelGGL(noavg_inside_kernel<kPlainStatesDev>, dim3(gx, a.C), dim3(256), 0, stream, a);
      if (a.logL || a.post_rate || a.rate_class) hipLaunchKernelGGL(site_scalars_kernel<kPlainStatesDev>, dim3(gx), dim3(256), 0, stream, a);
      if (a.counts) {
        hipLaunchKernelGGL(noavg_outside_kernel<kPlainStatesDev>, dim3(gx, a.C), dim3(256), 0, stream, a);
        if (a.mode == kVariantJoint) hipLaunchKernelGGL(joint_kernel<kPlainStatesDev>, dim3(gx, a.B), dim3(256), 0, stream, a);
        else if (a.mode == kVariantNoAvg) hipLaunchKernelGGL(noavg_pick_kernel<kPlainStatesDev>, dim3(gx, a.B), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(marginal_kernel<kPlainStatesDev>, dim3(gx, a.B), dim3(256), 0, stream, a);
      }
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (d_norm && a.counts)
    hipLaunchKernelGGL(counts_norm_kernel, dim3((unsigned)((nsites_total + 255) / 256)), dim3(256), 0, stream, a.counts, a.ldc, a.B,
                       a.K, nsites_total, d_norm);
  return hipGetLastError();
}

}  // namespace cmx
