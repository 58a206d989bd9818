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

// what a wave needs to find its data (wave-uniform pointers; per-lane offsets are added at each use)
struct NucCtx {
  const NucDev* m;
  double* WM;              // this wave's root messages, + 2 * lane
  double* WU;
  const uint8_t* gcodes;   // symbol of taxon t of this lane's site at gcodes[t * gstride]
  size_t gstride;
  const uint32_t* masks;
  size_t cls_root;         // doubles of root messages per class
  size_t cls_ltab;         // doubles of leaf tables per class
};

// where the counts of this lane's site go: row r at dst[r * stride]; written only by active lanes
struct NucCnt {
  double* dst;
  size_t stride;
  bool active;
};

// what is read one packet ahead of a visit: the messages it takes from outside its block (leaf gathers, block-root
// messages), its leaf children's count columns, its own outside message when it is a block root, the old count values
typedef double nuc_d4 __attribute__((ext_vector_type(4)));
struct NucAhead {   // (first-class vector members, named per side: hipcc left an array-of-arrays version of this in scratch memory)
  nuc_d4 GA, GB, GJA, GJB, GU;
  double OC0, OC1, OC2;
};

__device__ __forceinline__ nuc_d4 nuc_leaf_gather(const NucCtx& x, int c, int tx, int w, unsigned nib) {
  const double* q = x.m->ltab + (size_t)c * x.cls_ltab + ((size_t)tx * (x.m->K + 1) + w) * 64 + nib * 4;
  const nuc_d2 a = *reinterpret_cast<const nuc_d2*>(q), b = *reinterpret_cast<const nuc_d2*>(q + 2);
  nuc_d4 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
  return r;
}
__device__ __forceinline__ nuc_d4 nuc_root_load4(const double* base /* + 2 * lane */, int slot) {
  const nuc_d2 a = *reinterpret_cast<const nuc_d2*>(base + (size_t)slot * 256);
  const nuc_d2 b = *reinterpret_cast<const nuc_d2*>(base + (size_t)slot * 256 + 128);
  nuc_d4 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
  return r;
}

template <int SIDE>
__device__ __forceinline__ void nuc_read_ahead_child(const NucCtx& x, const nuc_i8 r, int c, unsigned long long sw, bool outside, nuc_d4& g,
                                                     nuc_d4& gj) {
  const int fl = r[PK_FLAGS];
  const int kind = SIDE ? (fl >> 2) & 3 : fl & 3, arg = r[SIDE ? PK_B : PK_A];
  if (kind == NK_LEAF) {
    const unsigned nib = (unsigned)(sw >> (4 * r[SIDE ? PK_BPOS : PK_APOS])) & 15u;
    g = nuc_leaf_gather(x, c, arg, 0, nib);
    if (outside) gj = nuc_leaf_gather(x, c, arg, 1, nib);
  } else if (kind == NK_HBM && !(fl & (SIDE ? NF_NOPF_B : NF_NOPF_A))) {
    g = nuc_root_load4(x.WM + c * x.cls_root, arg);
  }
}

// issues the reads of packet r that may run one packet ahead (ahead = true: during the previous packet, when the leaf
// masks of a block's first packet are still the "next" ones)
__device__ __forceinline__ void nuc_read_ahead(const NucCtx& x, const NucCnt& cd, const nuc_i8 r, unsigned long long symw,
                                               unsigned long long symn, NucAhead& o) {
  const int fl = r[PK_FLAGS];
  if (fl & (NF_BLOCKPKT | NF_END)) return;
  const int c = (unsigned)fl >> 24, K = x.m->K;
  const unsigned long long sw = (fl & NF_SWAPSYM) ? symn : symw;
  const bool outside = (fl & NF_OUTSIDE) != 0;
  nuc_read_ahead_child<0>(x, r, c, sw, outside, o.GA, o.GJA);
  nuc_read_ahead_child<1>(x, r, c, sw, outside, o.GB, o.GJB);
  if (outside) {
    if ((fl & NF_BLOCKROOT) && !(fl & (NF_ROOT | NF_NOPF_U))) o.GU = nuc_root_load4(x.WU + c * x.cls_root, r[PK_SLOT]);
    if (!(fl & (NF_FIRSTCLASS | NF_NOPF_CNT))) {
      if (r[PK_NODE] >= 0) o.OC0 = cd.dst[(size_t)r[PK_NODE] * K * cd.stride];
      if ((fl & 3) == NK_LEAF) o.OC1 = cd.dst[(size_t)(r[PK_KIDS] & 0xffff) * K * cd.stride];
      if (((fl >> 2) & 3) == NK_LEAF) o.OC2 = cd.dst[(size_t)((unsigned)r[PK_KIDS] >> 16) * K * cd.stride];
    }
  }
}

// per-site results that live across the packets of a wave-task
struct NucSite {
  double Lsum, prsum, best, rL, nrm;
  int bestc;
};

// count rows accumulate over the classes in class order: the first class writes, the others add to what the previous
// class of this block visit left (in L2); `old` was read one packet ahead unless the packet forbids it
__device__ __forceinline__ double nuc_count(const NucCnt& cd, int row, bool first, bool fresh, double old, double w, double v) {
  double* q = cd.dst + (size_t)row * cd.stride;
  double nv = w * v;
  if (!first) nv += fresh ? *q : old;
  if (cd.active) *q = nv;
  return nv;
}

// outside message of child SIDE: U_c = Up o (message of the other child); a leaf child's branch is counted, an internal
// child's message goes to its slot, a lower block's root's to HBM
template <int SIDE>
__device__ __forceinline__ void nuc_dispose(const NucSlots& sl, const NucCtx& x, const NucCnt& cd, const nuc_i8 r, const nuc_d4 gj, double oc,
                                            unsigned long long symw, NucSite& st, const double (&Up)[4], const double (&Mo)[4],
                                            bool first, bool last, bool fresh, double w) {
  const int fl = r[PK_FLAGS], c = (unsigned)fl >> 24, K = x.m->K;
  const int kind = SIDE ? (fl >> 2) & 3 : fl & 3, arg = r[SIDE ? PK_B : PK_A];
  double Uc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) Uc[q] = Up[q] * Mo[q];
  if (kind == NK_LEAF) {
    // count of the leaf branch: sum_x U[x] (J e)[x], (J e) from the leaf table
    const int leaf = (int)(((unsigned)r[PK_KIDS] >> (16 * SIDE)) & 0xffffu);
    double tot = 0.0;
    for (int k = 0; k < K; ++k) {
      double Je[4];
      if (k == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Je[q] = gj[q];
      } else {
        const nuc_d4 t = nuc_leaf_gather(x, c, arg, 1 + k, (unsigned)(symw >> (4 * r[SIDE ? PK_BPOS : PK_APOS])) & 15u);
#pragma unroll
        for (int q = 0; q < 4; ++q) Je[q] = t[q];
      }
      double s = Uc[0] * Je[0];
#pragma unroll
      for (int q = 1; q < 4; ++q) s = __builtin_fma(Uc[q], Je[q], s);
      tot += nuc_count(cd, leaf * K + k, first, fresh || k > 0, oc, w, s);
    }
    if (last) st.nrm = __builtin_fma(tot, tot, st.nrm);
  } else if (kind == NK_HBM) {
    nuc_root_store(x.WU + c * x.cls_root, arg, Uc);
  } else {
    sl.put(arg, Uc);
  }
}

// one visit packet
__device__ __forceinline__ void nuc_visit(const NucSlots& sl, const NucCtx& x, const NucCnt& cd, const nuc_i8 r, const NucAhead& pa,
                                          unsigned long long symw, NucSite& st) {
  const NucDev& m = *x.m;
  const int fl = r[PK_FLAGS], ka = fl & 3, kb = (fl >> 2) & 3, c = (unsigned)fl >> 24, K = m.K;
  const nuc_cop ops = (nuc_cop)m.ops + (size_t)c * m.nops;
  const nuc_cdbl pi = (nuc_cdbl)m.pi, probs = (nuc_cdbl)m.probs, rates = (nuc_cdbl)m.rates;
  double Ma[4], Mb[4];
  if (ka == NK_SLOT) sl.get(r[PK_A], Ma);
  else if (ka == NK_HBM && (fl & NF_NOPF_A)) nuc_root_load(x.WM + c * x.cls_root, r[PK_A], Ma);
  else {
#pragma unroll
    for (int q = 0; q < 4; ++q) Ma[q] = pa.GA[q];
  }
  if (kb == NK_SLOT) sl.get(r[PK_B], Mb);
  else if (kb == NK_HBM && (fl & NF_NOPF_B)) nuc_root_load(x.WM + c * x.cls_root, r[PK_B], Mb);
  else {
#pragma unroll
    for (int q = 0; q < 4; ++q) Mb[q] = pa.GB[q];
  }
  if (!(fl & NF_OUTSIDE)) {
    // ---- inside visit: M = P (M_a o M_b); phase 1 keeps block roots in HBM and the root likelihood
    double D[4], M[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) D[q] = Ma[q] * Mb[q];
    if (fl & NF_ROOT) {
      if (fl & NF_PHASE1) {
        double s = pi[0] * D[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) s = __builtin_fma(pi[q], D[q], s);
        const double pc = probs[c];
        st.Lsum += pc * s;
        st.prsum += rates[c] * pc * s;
        if (pc * s > st.best) { st.best = pc * s; st.bestc = c; }   // first maximum wins (getRateClassWithMaxPostProbPerSite)
        if (fl & NF_FINISH1) st.rL = 1.0 / st.Lsum;
      }
      return;
    }
    if (fl & NF_PSEUDO) {
#pragma unroll
      for (int q = 0; q < 4; ++q) M[q] = D[q];
    } else {
      const nuc_d16 A = ops[r[PK_NODE] * (K + 1)];
      nuc_mv_n(A, D, M);
    }
    if (fl & NF_BLOCKROOT) {
      if (fl & NF_PHASE1) nuc_root_store(x.WM + c * x.cls_root, r[PK_SLOT], M);
    } else {
      sl.put(r[PK_SLOT], M);
    }
    return;
  }
  // ---- outside visit: W = J^T U, count = sum W o M_a o M_b, Up = P^T U, U_a = Up o M_b, U_b = Up o M_a
  const bool first = (fl & NF_FIRSTCLASS) != 0, last = (fl & NF_LASTCLASS) != 0, fresh = (fl & NF_NOPF_CNT) != 0;
  const double w = probs[c] * st.rL;
  double U[4], Up[4];
  if (!(fl & NF_ROOT)) {
    if (!(fl & NF_BLOCKROOT)) sl.get(r[PK_SLOT], U);
    else if (fl & NF_NOPF_U) nuc_root_load(x.WU + c * x.cls_root, r[PK_SLOT], U);
    else {
#pragma unroll
      for (int q = 0; q < 4; ++q) U[q] = pa.GU[q];
    }
  }
  if (fl & NF_ROOT) {
#pragma unroll
    for (int q = 0; q < 4; ++q) Up[q] = pi[q];
  } else if (fl & NF_PSEUDO) {
#pragma unroll
    for (int q = 0; q < 4; ++q) Up[q] = U[q];
  } else {
    const int node = r[PK_NODE];
    double tot = 0.0;
    for (int k = 0; k < K; ++k) {
      const nuc_d16 J = ops[node * (K + 1) + 1 + k];
      double W[4];
      nuc_mv_t(J, U, W);
      double s = (W[0] * Ma[0]) * Mb[0];
#pragma unroll
      for (int q = 1; q < 4; ++q) s = __builtin_fma(W[q] * Ma[q], Mb[q], s);
      tot += nuc_count(cd, node * K + k, first, fresh || k > 0, pa.OC0, w, s);
    }
    if (last) st.nrm = __builtin_fma(tot, tot, st.nrm);
    const nuc_d16 P = ops[node * (K + 1)];
    nuc_mv_t(P, U, Up);
  }
  nuc_dispose<0>(sl, x, cd, r, pa.GJA, pa.OC1, symw, st, Up, Mb, first, last, fresh, w);
  nuc_dispose<1>(sl, x, cd, r, pa.GJB, pa.OC2, symw, st, Up, Ma, first, last, fresh, w);
}

// leaf list of a block: the 4-bit compatibility masks of its leaves' symbols, one 64-bit word per lane
__device__ __forceinline__ unsigned long long nuc_leaf_masks(const NucCtx& x, const nuc_i8 r) {
  const int nl = (unsigned)r[PK_FLAGS] >> 24;
  unsigned long long w = 0;
#pragma unroll
  for (int q = 0; q < kNucMaxLeaves; ++q) {
    if (q < nl) {
      const int tx = (r[1 + q / 2] >> (16 * (q & 1))) & 0xffff;
      const unsigned code = x.gcodes[(size_t)tx * x.gstride];
      unsigned mk = code < 4u ? (1u << code) : 0xFu;
      if (x.masks != nullptr && __ballot(code >= 4u) != 0ull) {
        const unsigned t = x.masks[code] & 0xFu;
        mk = code < 4u ? mk : t;
      }
      w |= (unsigned long long)mk << (4 * q);
    }
  }
  return w;
}

// one step of the stream: packet rc with what was read ahead for it (use); reads ahead for packet rn into fill.
// Returns true at the end of the stream.
__device__ __forceinline__ bool nuc_step(const NucSlots& sl, const NucCtx& x, const NucCnt& cd, const nuc_i8 rc, const nuc_i8 rn,
                                         const NucAhead& use, NucAhead& fill, unsigned long long& symw, unsigned long long& symn,
                                         NucSite& st) {
  const int fl = rc[PK_FLAGS];
  if (fl & NF_END) return true;
  if (fl & NF_BLOCKPKT) {
    symn = nuc_leaf_masks(x, rc);
    nuc_read_ahead(x, cd, rn, symw, symn, fill);
    return false;
  }
  if (fl & NF_SWAPSYM) symw = symn;
  nuc_read_ahead(x, cd, rn, symw, symn, fill);
  nuc_visit(sl, x, cd, rc, use, symw, st);
  return false;
}

// Maps the 64 sites of this wave: one walk of the packet stream.  On return the count rows at cd hold n(b, site, k) and
// the scalars are per lane.
__device__ __forceinline__ void nuc_map_sites(const NucSlots& sl, const NucArgs& a, double* WM, double* WU, const NucCnt& cd,
                                              const uint8_t* gcodes, size_t gstride, int lane, double& L_out, double& pr_out,
                                              int& rc_out, double& norm_out) {
  const NucDev& m = a.m;
  NucCtx x;
  x.m = &m;
  x.WM = WM + 2 * lane;
  x.WU = WU + 2 * lane;
  x.gcodes = gcodes;
  x.gstride = gstride;
  x.masks = a.masks;
  x.cls_root = (size_t)m.nroots * 256;
  x.cls_ltab = (size_t)m.T * (m.K + 1) * 64;
  NucSite st;
  st.Lsum = 0.0; st.prsum = 0.0; st.best = -1.0; st.rL = 0.0; st.nrm = 0.0; st.bestc = 0;
  unsigned long long symw = ~0ull, symn = ~0ull;
  NucAhead A, B;
  const nuc_d4 z4 = {0.0, 0.0, 0.0, 0.0};
  A.GA = A.GB = A.GJA = A.GJB = A.GU = z4;
  A.OC0 = A.OC1 = A.OC2 = 0.0;
  B = A;
  const nuc_ci8 pk = (nuc_ci8)m.pk;
  nuc_i8 r0 = pk[0], r1 = pk[1];
  for (int i = 0;; i += 2) {
    const nuc_i8 r2 = pk[i + 2];                  // records are read two packets ahead
    if (nuc_step(sl, x, cd, r0, r1, A, B, symw, symn, st)) break;
    r0 = pk[i + 3];
    if (nuc_step(sl, x, cd, r1, r2, B, A, symw, symn, st)) break;
    r1 = r0;
    r0 = r2;
  }
  L_out = st.Lsum;
  pr_out = st.prsum * st.rL;
  rc_out = st.bestc;
  norm_out = sqrt(st.nrm);
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
