// map_nuc_kernel: the 4-state (nucleotide) mapping kernel of the engine (gfx950).  Design and program: cmx_nuc.h.
//
// Replaces, per site: DRHomogeneousTreeLikelihood::initialize, the outside pass,
// LegacySubstitutionMappingTools::computeSubstitutionVectors and computeNormForSite (call sites CoMap/CoETools.cpp:397,
// CoMap/AnalysisTools.cpp:592-612; algorithm SURVEY.md A.2 / A.3 / A.6) and, in null mode, the replicate body of
// AnalysisTools::getNullDistributionIntraDR (AnalysisTools.cpp:587-653: map two batches, score site j against site j).
//
// lane = site.  A message is 4 doubles = 8 VGPRs of the lane.  The 4x4 operator of a branch is wave-uniform: it is read
// with s_load_dwordx16 (x2) through the scalar cache and applied with v_fma_f64 whose multiplicand is the SGPR pair --
// no operator staging, no cross-lane instruction anywhere in the walk.  The messages of a block of <= NB internal nodes
// live in this wave's LDS slots (2 KiB per message); only block roots touch HBM.  Leaves: the leaf vector e (1 for every state compatible with the
// symbol; a resolved symbol is one-hot) is multiplied by the leaf branch's operator like any other message, which serves
// every ambiguity code without extra table rows.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "cmx_nuc.h"
#include "cmx_pairstat.h"

namespace cmx {

typedef double nuc_d16 __attribute__((ext_vector_type(16)));
typedef double nuc_d2 __attribute__((ext_vector_type(2)));
typedef int nuc_i4 __attribute__((ext_vector_type(4)));
typedef int nuc_i8 __attribute__((ext_vector_type(8)));
typedef int nuc_i16 __attribute__((ext_vector_type(16)));
// wave-uniform read-only data goes through the CONSTANT address space: hipcc then emits s_load (scalar cache)
typedef const nuc_d16 __attribute__((address_space(4)))* nuc_cop;
typedef const nuc_i4 __attribute__((address_space(4)))* nuc_ci4;
typedef const nuc_i8 __attribute__((address_space(4)))* nuc_ci8;
typedef const nuc_i16 __attribute__((address_space(4)))* nuc_ci16;
typedef const double __attribute__((address_space(4)))* nuc_cdbl;

// ---- message slots of the current block, in LDS: [slot][2][64 lanes][2 doubles] per wave -- each of the two accesses of
// a message is one conflict-free ds_read_b128 / ds_write_b128.  (Register arrays indexed with s_set_gpr_idx were tried
// first: hipcc does emit them, but every dynamic insert into a second or third 16-double vector copies whole 32-register
// tuples around the control flow -- 256 VGPRs and 900-2000 bytes of scratch per lane.)
extern __shared__ __attribute__((aligned(16))) uint8_t nuc_smem[];
struct NucSlots {
  uint8_t* base;   // this wave's slots + 16 * lane
  __device__ __forceinline__ void get(int s, double (&r)[4]) const {
    const nuc_d2* q = reinterpret_cast<const nuc_d2*>(base + (size_t)s * 2048);
    const nuc_d2 a = q[0], b = q[64];
    r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
  }
  __device__ __forceinline__ void put(int s, const double (&r)[4]) const {
    nuc_d2* q = reinterpret_cast<nuc_d2*>(base + (size_t)s * 2048);
    nuc_d2 a, b;
    a[0] = r[0]; a[1] = r[1]; b[0] = r[2]; b[1] = r[3];
    q[0] = a; q[64] = b;
  }
};

__device__ __forceinline__ void nuc_mv_n(const nuc_d16& A, const double (&x)[4], double (&y)[4]) {   // y = A x
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double s = A[4 * i] * x[0];
#pragma unroll
    for (int j = 1; j < 4; ++j) s = __builtin_fma(A[4 * i + j], x[j], s);
    y[i] = s;
  }
}
__device__ __forceinline__ void nuc_mv_t(const nuc_d16& A, const double (&x)[4], double (&y)[4]) {   // y = A^T x
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double s = A[j] * x[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) s = __builtin_fma(A[4 * i + j], x[i], s);
    y[j] = s;
  }
}

// block-root messages in HBM: [slot][2][64 lanes][2 doubles] -- each of the two accesses of a message is one coalesced KiB
__device__ __forceinline__ void nuc_root_load(const double* base /* + 2 * lane */, int slot, double (&r)[4]) {
  const nuc_d2 a = *reinterpret_cast<const nuc_d2*>(base + (size_t)slot * 256);
  const nuc_d2 b = *reinterpret_cast<const nuc_d2*>(base + (size_t)slot * 256 + 128);
  r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
}
__device__ __forceinline__ void nuc_root_store(double* base, int slot, const double (&r)[4]) {
  nuc_d2 a, b;
  a[0] = r[0]; a[1] = r[1]; b[0] = r[2]; b[1] = r[3];
  *reinterpret_cast<nuc_d2*>(base + (size_t)slot * 256) = a;
  *reinterpret_cast<nuc_d2*>(base + (size_t)slot * 256 + 128) = b;
}

// what one class pass of one wave needs to find its data
struct NucPass {
  nuc_cop ops;             // operators of this class
  const double* WM;        // root messages of this class, + 2 * lane
  double* WMw;
  const double* WU;
  double* WUw;
  const uint8_t* gcodes;   // symbol of taxon t of this lane's site at gcodes[t * gstride]
  size_t gstride;
  const uint32_t* masks;
};

// leaf vector of this lane's site: 1.0 for every state compatible with the symbol (DR likelihood leaf initialisation)
__device__ __forceinline__ void nuc_leaf_e(const NucPass& p, int tx, double (&e)[4]) {
  const unsigned code = p.gcodes[(size_t)tx * p.gstride];
  unsigned m = code < 4u ? (1u << code) : 0xFu;
  if (p.masks != nullptr && __ballot(code >= 4u) != 0ull) {
    const unsigned mk = p.masks[code];
    m = code < 4u ? m : mk;
  }
#pragma unroll
  for (int z = 0; z < 4; ++z) e[z] = ((m >> z) & 1u) ? 1.0 : 0.0;
}

__device__ __forceinline__ void nuc_child(const NucSlots& sl, const NucPass& p, int kind, int arg, int tx, double (&M)[4]) {
  if (kind == NK_LEAF) {
    double e[4];
    nuc_leaf_e(p, tx, e);
    const nuc_d16 A = p.ops[arg];
    nuc_mv_n(A, e, M);
  } else if (kind == NK_SLOT) {
    sl.get(arg, M);
  } else {
    nuc_root_load(p.WM, arg, M);
  }
}

// inside visit of one internal node: M = P (M_a o M_b); phase 1 keeps block roots in HBM and the root likelihood
__device__ __forceinline__ void nuc_inside(const NucSlots& sl, const NucPass& p, const nuc_i8 r, bool phase1, nuc_cdbl pi, double& Lc) {
  const int fl = r[NI_FLAGS];
  double Ma[4], Mb[4], D[4], M[4];
  nuc_child(sl, p, fl & 3, r[NI_A], r[NI_ATX], Ma);
  nuc_child(sl, p, (fl >> 2) & 3, r[NI_B], r[NI_BTX], Mb);
#pragma unroll
  for (int x = 0; x < 4; ++x) D[x] = Ma[x] * Mb[x];
  if (fl & NF_ROOT) {
    if (phase1) {
      double s = pi[0] * D[0];
#pragma unroll
      for (int x = 1; x < 4; ++x) s = __builtin_fma(pi[x], D[x], s);
      Lc = s;
    }
    return;
  }
  if (fl & NF_PSEUDO) {
#pragma unroll
    for (int x = 0; x < 4; ++x) M[x] = D[x];
  } else {
    const nuc_d16 A = p.ops[r[NI_OP]];
    nuc_mv_n(A, D, M);
  }
  if (fl & NF_BLOCKROOT) {
    if (phase1) nuc_root_store(p.WMw, r[NI_DST], M);
  } else {
    sl.put(r[NI_DST], M);
  }
}

// where the counts of this lane's site go: row r at dst[r * stride]; written only by active lanes
struct NucCnt {
  double* dst;
  size_t stride;
  bool active;
};
// count rows accumulate over the classes in class order: class 0 writes, the others read-modify-write a row that the
// previous class of this block visit left in L2.  `last`: the row is final -- its value is returned for the norm.
__device__ __forceinline__ double nuc_count(const NucCnt& cd, int row, int c, double w, double v) {
  double* q = cd.dst + (size_t)row * cd.stride;
  double nv = w * v;
  if (c != 0) nv += cd.active ? *q : 0.0;
  if (cd.active) *q = nv;
  return nv;
}

__device__ __forceinline__ void nuc_outside(const NucSlots& sl, const NucPass& p, const nuc_i16 r, int c, bool last, double w, int K,
                                            nuc_cdbl pi, const NucCnt& cd, double& nrm) {
  const int fl = r[NO_FLAGS], ka = fl & 3, kb = (fl >> 2) & 3;
  double U[4], Ma[4], Mb[4], Up[4];
  if (!(fl & NF_ROOT)) {
    if (fl & NF_BLOCKROOT) nuc_root_load(p.WU, r[NO_USRC], U);
    else sl.get(r[NO_USRC], U);
  }
  nuc_child(sl, p, ka, r[NO_A], r[NO_ATX], Ma);
  nuc_child(sl, p, kb, r[NO_B], r[NO_BTX], Mb);
  if (fl & NF_ROOT) {
#pragma unroll
    for (int x = 0; x < 4; ++x) Up[x] = pi[x];
  } else if (fl & NF_PSEUDO) {
#pragma unroll
    for (int x = 0; x < 4; ++x) Up[x] = U[x];
  } else {
    double tot = 0.0;
    for (int k = 0; k < K; ++k) {
      const nuc_d16 J = p.ops[r[NO_OPJ] + k];
      double W[4];
      nuc_mv_t(J, U, W);
      double s = (W[0] * Ma[0]) * Mb[0];
#pragma unroll
      for (int x = 1; x < 4; ++x) s = __builtin_fma(W[x] * Ma[x], Mb[x], s);
      tot += nuc_count(cd, r[NO_ROW] + k, c, w, s);
    }
    if (last) nrm = __builtin_fma(tot, tot, nrm);
    const nuc_d16 P = p.ops[r[NO_OPP]];
    nuc_mv_t(P, U, Up);
  }
  // outside messages of the children: U_a = Up o M_b, U_b = Up o M_a
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const int kind = side ? kb : ka, disp = r[side ? NO_BDISP : NO_ADISP];
    double Uc[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) Uc[x] = Up[x] * (side ? Ma[x] : Mb[x]);
    if (kind == NK_LEAF) {
      // count of the leaf branch: sum_x U[x] (J e)[x]
      double e[4];
      nuc_leaf_e(p, r[side ? NO_BTX : NO_ATX], e);
      const int row = r[side ? NO_BROW : NO_AROW];
      double tot = 0.0;
      for (int k = 0; k < K; ++k) {
        const nuc_d16 J = p.ops[disp + k];
        double Je[4];
        nuc_mv_n(J, e, Je);
        double s = Uc[0] * Je[0];
#pragma unroll
        for (int x = 1; x < 4; ++x) s = __builtin_fma(Uc[x], Je[x], s);
        tot += nuc_count(cd, row + k, c, w, s);
      }
      if (last) nrm = __builtin_fma(tot, tot, nrm);
    } else if (kind == NK_HBM) {
      nuc_root_store(p.WUw, disp, Uc);
    } else {
      sl.put(disp, Uc);
    }
  }
}

// Maps the 64 sites of this wave.  On return the count rows at cd hold n(b, site, k) and the scalars are per lane.
__device__ __forceinline__ void nuc_map_sites(const NucSlots& sl, const NucArgs& a, double* WM, double* WU, const NucCnt& cd, const uint8_t* gcodes,
                                              size_t gstride, int lane, double& L_out, double& pr_out, int& rc_out, double& norm_out) {
  const NucDev& m = a.m;
  const int C = m.C, K = m.K;
  const nuc_cdbl pi = (nuc_cdbl)m.pi, probs = (nuc_cdbl)m.probs, rates = (nuc_cdbl)m.rates;
  const nuc_ci4 blk = (nuc_ci4)m.blk;
  const nuc_ci8 irec = (nuc_ci8)m.irec;
  const nuc_ci16 orec = (nuc_ci16)m.orec;
  NucPass p;
  p.gcodes = gcodes;
  p.gstride = gstride;
  p.masks = a.masks;
  const size_t cls_stride = (size_t)m.nroots * 256;   // doubles of root messages per class
  // ---- phase 1: inside pass, class by class; block roots -> HBM
  double Lsum = 0.0, prsum = 0.0, best = -1.0;
  int bestc = 0;
  for (int c = 0; c < C; ++c) {
    p.ops = (nuc_cop)m.ops + (size_t)c * m.nops;
    p.WM = p.WMw = WM + c * cls_stride + 2 * lane;
    p.WU = p.WUw = WU + c * cls_stride + 2 * lane;
    double Lc = 0.0;
    for (int b = 0; b < m.nblocks; ++b) {
      const nuc_i4 bd = blk[b];
      for (int i = 0; i < bd[1]; ++i) nuc_inside(sl, p, irec[bd[0] + i], true, pi, Lc);
    }
    const double pc = probs[c];
    Lsum += pc * Lc;
    prsum += rates[c] * pc * Lc;
    if (pc * Lc > best) { best = pc * Lc; bestc = c; }   // first maximum wins (getRateClassWithMaxPostProbPerSite)
  }
  const double rL = 1.0 / Lsum;
  // ---- phase 2: outside pass + counts, blocks top-down; per block every class: recompute the block's inside messages
  // into the registers, then walk it from its root's outside message
  double nrm = 0.0;
  for (int b = m.nblocks - 1; b >= 0; --b) {
    const nuc_i4 bd = blk[b];
    for (int c = 0; c < C; ++c) {
      p.ops = (nuc_cop)m.ops + (size_t)c * m.nops;
      p.WM = p.WMw = WM + c * cls_stride + 2 * lane;
      p.WU = p.WUw = WU + c * cls_stride + 2 * lane;
      const double w = probs[c] * rL;
      double dummy = 0.0;
      for (int i = 0; i + 1 < bd[1]; ++i) nuc_inside(sl, p, irec[bd[0] + i], false, pi, dummy);
      for (int i = 0; i < bd[3]; ++i) nuc_outside(sl, p, orec[bd[2] + i], c, c == C - 1, w, K, pi, cd, nrm);
    }
  }
  L_out = Lsum;
  pr_out = prsum * rL;
  rc_out = bestc;
  norm_out = sqrt(nrm);
}

// LDS per workgroup = 4 waves x NB slots x 2 KiB; the CU's 160 KiB then hold 160 / (8 NB) workgroups, i.e. that many waves
// per SIMD: NB = 10 -> 2, NB = 6 -> 3, NB = 5 -> 4.  Larger blocks mean fewer block roots through HBM, fewer waves to hide
// the scalar loads behind (measured trade-off: DESIGN.md 4.9).
int nuc_waves_per_simd(int NB) { return std::max(1, std::min(4, 160 / (8 * NB))); }

template <int WPS, bool NULLMODE>
__global__ __launch_bounds__(256, WPS) void map_nuc_kernel(const NucArgs a) {
  const NucDev& m = a.m;
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = blockIdx.x * 4 + wib, nwaves = gridDim.x * 4;
  const size_t wsz = (size_t)m.C * m.nroots * 256, BK = (size_t)m.B * m.K;
  double* WM = a.ws.WM + (size_t)wave * wsz;
  double* WU = a.ws.WU + (size_t)wave * wsz;
  double* cnt0 = a.ws.cnt + (size_t)wave * 2 * BK * 64;
  double* cnt1 = cnt0 + BK * 64;
  NucSlots sl;
  sl.base = nuc_smem + (size_t)wib * m.NB * 2048 + 16 * lane;
  const size_t nblk = (a.nsites + 63) / 64;
  for (size_t sb = wave; sb < nblk; sb += nwaves) {
    const size_t site = sb * 64 + lane;
    const bool active = site < a.nsites;
    const size_t s = active ? site : a.nsites - 1;
    if (!NULLMODE) {
      NucCnt cd;
      if (a.counts) { cd.dst = a.counts + s; cd.stride = a.ldc; cd.active = active; }
      else { cd.dst = cnt0 + lane; cd.stride = 64; cd.active = true; }
      double L, pr, nrm;
      int rc;
      nuc_map_sites(sl, a, WM, WU, cd, a.aln + s, a.ld, lane, L, pr, rc, nrm);
      if (active) {
        if (a.logL) a.logL[s] = log(L);
        if (a.post_rate) a.post_rate[s] = pr;
        if (a.rate_class) a.rate_class[s] = rc;
        if (a.norm) a.norm[s] = nrm;
      }
    } else {
      // null pair s: replicate s / rep_ram, column s % rep_ram of both batches; only the minima over the two batches
      // leave the loop (AnalysisTools.cpp:643-652)
      double prmin = 0.0, nmin = 0.0;
      int rcmin = 0;
      const size_t rep_local = s / a.rep_ram, j = s % a.rep_ram;
      for (int h = 0; h < 2; ++h) {
        const uint8_t* gbase = a.supplied + ((rep_local * 2 + h) * (size_t)m.T) * a.rep_ram + j;
        NucCnt cd;
        cd.dst = (h ? cnt1 : cnt0) + lane; cd.stride = 64; cd.active = true;
        double L, pr, nrm;
        int rc;
        nuc_map_sites(sl, a, WM, WU, cd, gbase, a.rep_ram, lane, L, pr, rc, nrm);
        if (h == 0) { prmin = pr; nmin = nrm; rcmin = rc; }
        else { prmin = pr < prmin ? pr : prmin; nmin = nrm < nmin ? nrm : nmin; rcmin = rc < rcmin ? rc : rcmin; }
      }
      const double stat = pair_stat_strided(a.stat_kind, a.stat_param, m.B, m.K, cnt0 + lane, (size_t)64, cnt1 + lane, (size_t)64, a.stat_mean);
      if (active) {
        a.null_stat[s] = stat;
        if (a.null_rcmin) a.null_rcmin[s] = rcmin;
        if (a.null_prmin) a.null_prmin[s] = prmin;
        if (a.null_nmin) a.null_nmin[s] = nmin;
      }
    }
  }
}

template <int WPS>
static hipError_t launch_map_nuc_w(const NucArgs& a, bool null_mode, dim3 grid, size_t lds, hipStream_t stream) {
  const void* fn = null_mode ? reinterpret_cast<const void*>(&map_nuc_kernel<WPS, true>) : reinterpret_cast<const void*>(&map_nuc_kernel<WPS, false>);
  // per launch: the attribute belongs to the current device (a process may hold contexts on several)
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / WPS);
  if (e != hipSuccess) return e;
  if (null_mode) hipLaunchKernelGGL((map_nuc_kernel<WPS, true>), grid, dim3(256), lds, stream, a);
  else hipLaunchKernelGGL((map_nuc_kernel<WPS, false>), grid, dim3(256), lds, stream, a);
  return hipGetLastError();
}

hipError_t launch_map_nuc(const NucArgs& a, bool null_mode, int grid_blocks, hipStream_t stream) {
  const int wps = nuc_waves_per_simd(a.m.NB);
  const size_t lds = (size_t)4 * a.m.NB * 2048;
  if (lds * wps > 160 * 1024) return hipErrorInvalidValue;
  dim3 grid(grid_blocks);
  switch (wps) {
    case 1: return launch_map_nuc_w<1>(a, null_mode, grid, lds, stream);
    case 2: return launch_map_nuc_w<2>(a, null_mode, grid, lds, stream);
    case 3: return launch_map_nuc_w<3>(a, null_mode, grid, lds, stream);
    default: return launch_map_nuc_w<4>(a, null_mode, grid, lds, stream);
  }
}

}  // namespace cmx
