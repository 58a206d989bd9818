// Hand-written gfx950 (CDNA4) kernels of the engine.  fp64 throughout (the reference computes in double and its
// DR likelihood has no per-node rescaling, CoMap/CoETools.cpp:244).
//
// map_kernel<S, MODE>
//   one wavefront = 64 sites (lane == site) x all rate classes in turn; four independent waves per workgroup.
//   Replaces, per site: DRHomogeneousTreeLikelihood::initialize (post-order "inside" pass), the pre-order
//   "outside" pass, LegacySubstitutionMappingTools::computeSubstitutionVectors and computeNormForSite
//   (call sites CoMap/CoETools.cpp:397, CoMap/AnalysisTools.cpp:592-612; algorithm SURVEY.md A.2/A.3/A.6).
//   The SxS operator of an edge is the same for all 64 lanes: every operator use of a class pass (products on
//   internal edges, row gathers by observed symbol on leaf edges) is listed by the host in one op stream and staged
//   in LDS by LDS-DMA one op ahead; products run on the matrix cores (v_mfma_f64_4x4x4_4b, 100 per 20x20 product) in
//   a layout where lane = (state inside a 4-state tile, site of a group of 16).
//   Matrices are packed host-side in 4x4 blocks so that one copy serves both P.d (inside) and P^T.u (outside).
//   Inside vectors / outside messages that must survive go to a per-wave HBM workspace as [S/2][lane][2]: every
//   access is one fully coalesced 1 KiB row; loads are prefetched into LDS by DMA under a host-built schedule.
//   DESIGN.md 4.1 has the full description and the measurements.
//   MODE == kModeNull: map (x2 batches) -> per-pair statistic of AnalysisTools::getNullDistributionIntraDR
//   (CoMap/AnalysisTools.cpp:587-653) per wave, on alignments simulated beforehand by simulate_lds_kernel /
//   simulate_blocked_kernel at full occupancy (round 1 simulated inside the mapping wave).
// pair_gram_kernel: all-pairs statistic as X.X^T on v_mfma_f64_16x16x4_f64 with per-statistic epilogues
//   (CoMap/Statistics.h:164-329; loops CoMap/CoETools.cpp:672-692, 786-810).
// mica_mfma_kernel: column mutual information as a one-hot Gram on v_mfma_i32_32x32x32_i8 (CoMap/Mica.cpp:349-361).
#include <algorithm>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "cmx_device.h"
#include "cmx_lanes.h"
#include "cmx_pairstat.h"
#include "cmx_walk.h"

namespace cmx {

// ------------------------------------------------------------------------------------------------ ring matvec
typedef double d8 __attribute__((ext_vector_type(8)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));


// Every edge of the tree applies one SxS operator that is the same for all 64 lanes: a matrix-vector product on
// internal edges, a row gather by observed symbol on leaf edges.  The sequence of operators of a class pass is known
// (host-built op stream, m.msched), so each operator is DMAed into LDS one op ahead: two S*S*8-byte stage buffers per
// wave, op i reads buffer i&1 while the DMA of op i+1 fills the other (global_load_lds_dwordx4, 1 KiB per
// instruction, no VGPRs).  A product reads the A operand of each 4x4 tile with one ds_read_b64 and runs on
// v_mfma_f64_4x4x4_4b (see "Matrix-core layout" below); a leaf op reads the values its lane holds of row `symbol` of
// the transposed matrix.  The symbols of the next leaf op are DMAed the same way (global_load_lds_ubyte into a
// 256-byte slot).
// History: an s_load-fed v_fma_f64 version ran at 39 TFLOP/s; tiles in a VGPR ring applied with
// v_fmac_f64_dpp row_newbcast (53 TFLOP/s in isolation) needed either fixed registers -- amdgpu_num_vgpr turned out not
// to be a hard limit, the compiler reused them under pressure -- or 50 more loop-carried registers than two waves per
// SIMD can afford; the LDS-staged DPP version (lane = site) reached 16.7 ms per launch, the matrix-core layout 15.5 ms;
// gathering leaf rows straight from L2 (lane-divergent 160-byte rows, latency exposed at every leaf) cost 5 of 21 ms.

typedef __attribute__((address_space(1))) const void* cmx_gptr;
typedef __attribute__((address_space(3))) void* cmx_lptr;

template <int S>
struct MatStage {
  static constexpr int NROW = S + max_ambig(S);              // rows of a transposed leaf operator (states + ambiguity ids)
  static constexpr int UNIT = mat_unit(S);                   // doubles per device matrix
  static constexpr int BYTES = UNIT * 8;                     // one operator
  static constexpr int FULL = BYTES / 1024;                  // full 1 KiB DMA rows
  static constexpr int TAIL = (BYTES % 1024) / 16;           // lanes of the last, partial row
  static constexpr int ROWS = FULL + (TAIL ? 1 : 0);         // VMEM instructions per operator
  static_assert(BYTES % 16 == 0, "operator size must be a multiple of 16 bytes");
};
constexpr int kCodeSlotBytes = 4 * kWave;  // one dword per lane

// LDS-DMA is issued through inline asm on purpose.  hipcc models __builtin_amdgcn_global_load_lds as a write to
// LDS that any later ds_read may alias, and puts s_waitcnt vmcnt(0) in front of the next LDS read: with one dynamic LDS
// array every operator, workspace vector and symbol would then be waited for right after the NEXT one was requested,
// i.e. the whole L2/HBM latency exposed once per op.  An asm DMA is invisible to that bookkeeping; the ordering
// between a DMA and the LDS reads of its data is kept by the counted waits below (wait_vm), and compiler-issued
// vmcnt waits for its own loads only become stricter (it undercounts what is in flight).
// One instruction moves 64 lanes x 16 B = 1 KiB; the instruction offset advances the global and the LDS address alike,
// so consecutive 1 KiB rows need one M0 value and one address register.
__device__ __forceinline__ uint32_t lds_addr(const uint8_t* p) {
  // wave-uniform by construction; readfirstlane tells the compiler so (M0 is written from an SGPR)
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)p);
}
template <int NROWS>
__device__ __forceinline__ void dma_rows16(const void* g /* per lane: row 0 address of this lane */, uint32_t lds) {
  static_assert(NROWS >= 1 && NROWS <= 4, "instruction offset range");
  if constexpr (NROWS == 1)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(g) : "memory");
  else if constexpr (NROWS == 2)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024" ::"s"(lds), "v"(g) : "memory");
  else if constexpr (NROWS == 3)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %1, off offset:2048" ::"s"(lds), "v"(g) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072" ::"s"(lds), "v"(g) : "memory");
}

// DMA the operator at `src` (global, wave-uniform) into the stage buffer `buf` (LDS, wave-uniform)
template <int S>
__device__ __forceinline__ void mat_dma_l(const double* src, uint32_t l, int lane);
template <int S>
__device__ __forceinline__ void mat_dma(const double* src, uint8_t* buf, int lane) { mat_dma_l<S>(src, lds_addr(buf), lane); }
template <int S>
__device__ __forceinline__ void mat_dma_l(const double* src, uint32_t l /* LDS byte address, wave-uniform */, int lane) {
  const double* g = src + 2 * lane;
  if constexpr (MatStage<S>::FULL > 0) dma_rows16<MatStage<S>::FULL>(g, l);
  if constexpr (MatStage<S>::TAIL > 0) {
    if (lane < MatStage<S>::TAIL) dma_rows16<1>(g + MatStage<S>::FULL * 128, l + MatStage<S>::FULL * 1024);
  }
}
// the first BYTES bytes of an operator only (compile-time): products of a class-fused model need their NB diagonal tiles,
// leaf ops on fully resolved alignments (the null's simulated ones) the S0 state rows of the transposed operator
template <int BYTES>
struct MatPart {
  static constexpr int FULL = BYTES / 1024, TAIL = (BYTES % 1024 + 15) / 16, ROWS = FULL + (TAIL ? 1 : 0);
};
template <int BYTES>
__device__ __forceinline__ void mat_dma_part(const double* src, uint32_t l, int lane) {
  const double* g = src + 2 * lane;
  if constexpr (MatPart<BYTES>::FULL > 0) dma_rows16<MatPart<BYTES>::FULL>(g, l);
  if constexpr (MatPart<BYTES>::TAIL > 0) {
    if (lane < MatPart<BYTES>::TAIL) dma_rows16<1>(g + MatPart<BYTES>::FULL * 128, l + MatPart<BYTES>::FULL * 1024);
  }
}
// DMA one symbol per lane (address p is per lane) into dword `lane` of the code slot.  The symbols of the null are
// written by this very wave (simulation) through the same address, hence the L1-bypassing cache policy (sc0 sc1).
__device__ __forceinline__ void code_dma_l(const uint8_t* p, uint32_t l) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, off sc0 sc1" ::"s"(l), "v"(p) : "memory");
}
__device__ __forceinline__ void code_dma(const uint8_t* p, uint8_t* slot) { code_dma_l(p, lds_addr(slot)); }

// s_waitcnt vmcnt(n) needs an immediate: pick the largest supported threshold <= allowed (waiting for more is safe)
template <int N>
__device__ __forceinline__ void waitcnt_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
template <int S, int VL>
__device__ __forceinline__ void wait_vm(int allowed) {
  // what can have been issued after X: the next operator (1 .. R DMA rows, +1 with symbols) and up to three workspace
  // vectors (stores or loads, H instructions each).  The operator part varies per op since round 4 (products of a fused
  // model and leaf ops of resolved alignments stage a part of the unit), so every count up to R + 1 + H is exact and the
  // steps above are the multiples of H the vectors add; a smaller immediate than `allowed` only waits for more.
  constexpr int R = MatStage<S>::ROWS, H = (VL + 1) / 2;
  static_assert(R + 1 + 2 * H < 64, "vmcnt is a 6-bit counter");
  static_assert(R + 1 <= 6, "operator rows + symbols");
  const int v = allowed % H < 7 ? allowed % H : 6, q = allowed / H;   // (H >= 7 for every instantiation but S = 4: there v < H anyway)
  if (q == 0) {
    if (v <= 3) { if (v <= 1) { if (v == 0) waitcnt_vm<0>(); else waitcnt_vm<1>(); } else { if (v == 2) waitcnt_vm<2>(); else waitcnt_vm<3>(); } }
    else { if (v == 4) waitcnt_vm<4>(); else if (v == 5) waitcnt_vm<5>(); else waitcnt_vm<6>(); }
  } else if (q == 1) {
    if (v <= 3) { if (v <= 1) { if (v == 0) waitcnt_vm<H>(); else waitcnt_vm<H + 1>(); } else { if (v == 2) waitcnt_vm<H + 2>(); else waitcnt_vm<H + 3>(); } }
    else { if (v == 4) waitcnt_vm<H + 4>(); else if (v == 5) waitcnt_vm<H + 5>(); else waitcnt_vm<H + 6>(); }
  } else if (q == 2 && 2 * H + 6 < 64) {
    if (v <= 3) { if (v <= 1) { if (v == 0) waitcnt_vm<2 * H>(); else waitcnt_vm<2 * H + 1>(); } else { if (v == 2) waitcnt_vm<2 * H + 2>(); else waitcnt_vm<2 * H + 3>(); } }
    else { if (v == 4) waitcnt_vm<2 * H + 4>(); else if (v == 5) waitcnt_vm<2 * H + 5>(); else waitcnt_vm<2 * H + 6>(); }
  } else {
    // three vectors or more in flight: whatever the counter can still tell apart (it saturates at 63)
    constexpr int TOP = (3 * H + 4 < 64) ? 3 * H : (2 * H + 6 < 64 ? 2 * H + 6 : 2 * H);
    waitcnt_vm<TOP>();
  }
}

// ---- Matrix-core layout of an S-vector of the wave's 64 sites.  v_mfma_f64_4x4x4_4b computes four independent
// D[4x4] += A[4x4] . B[4x4]; measured on gfx950 (scripts/probe_mfma_f64_4x4x4.hip): A lane = 16 k + 4 blk + i,
// B lane = 16 k + 4 blk + n, D lane = 16 i + 4 blk + n.  With i / k = state inside a 4-state tile and (blk, n) = site inside
// a group of 16, B and D share one layout:
//     register v[sb * 4 + g] of lane l holds state 4 sb + (l >> 4) of site 16 g + (l & 15)
// (sb = state tile, g = site group).  A product is one MFMA per (output tile, input tile, site group): 100 matrix
// instructions instead of 400 DPP-broadcast FMAs, the accumulators of the four site groups are independent chains, and
// the other wave's vector work issues while they run.  The operator tile is read from the same packed 4x4 blocks as
// before, each lane taking the one element its A slot needs.  Elementwise work is layout-blind; sums over states
// become a reduce-scatter over the lane bits 4 and 5 that leaves lane l with the total of site l.

// reduce-scatter of per-site-group partial sums over the four state-in-tile lanes: lane l returns the total of site l.
// v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second one,
// v_permlane16_swap does the same for the odd / even rows of 16 lanes: after swap(a, b) the sum a + b holds, in the
// lanes that keep a's site group, own + partner's a, and in the others own + partner's b -- no LDS round trip.
template <int NG>
__device__ __forceinline__ double reduce_sites(const double (&p)[NG]) {
  if constexpr (NG == 4) {
    double p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
    swap32(p0, p2);            // lanes < 32: own p0, partner's p0   | lanes >= 32: partner's p2, own p2
    swap32(p1, p3);
    double k0 = p0 + p2, k1 = p1 + p3;   // site groups 0 / 1 in the lower half of the wave, 2 / 3 in the upper
    swap16(k0, k1);            // even rows: own k0, partner's k0    | odd rows: partner's k1, own k1
    return k0 + k1;            // site group lane >> 4: lane l holds site l
  } else if constexpr (NG == 3) {
    // three site groups (48 sites per wave, three waves per SIMD): the four-group exchange with an empty fourth group;
    // lanes 48 .. 63 end with nothing of interest
    double p0 = p[0], p1 = p[1], p2 = p[2], p3 = 0.0;
    swap32(p0, p2);
    swap32(p1, p3);
    double k0 = p0 + p2, k1 = p1 + p3;
    swap16(k0, k1);
    return k0 + k1;
  } else if constexpr (NG == 2) {
    // two site groups (32 sites per wave): lanes l and l ^ 16 both end with the total of site 16 (l >> 5) + (l & 15)
    double p0 = p[0], p1 = p[1];
    swap32(p0, p1);            // lanes < 32: own p0, partner's p0   | lanes >= 32: partner's p1, own p1
    double k = p0 + p1, q = k;
    swap16(k, q);              // even rows: k own, q = partner's (odd row) k   | odd rows: k = partner's, q own
    return k + q;
  } else {
    // one site group (16 sites per wave, small alignments): all four lanes l, l ^ 16, l ^ 32, l ^ 48 end with the total of site l & 15
    static_assert(NG == 1, "site groups per wave");
    double p0 = p[0], p1 = p0;
    swap32(p0, p1);            // p0 = the lower half's value everywhere, p1 = the upper half's
    double k = p0 + p1, q = k;
    swap16(k, q);
    return k + q;
  }
}

// step q of a product: output tile o = q / NB, input tile i = q % NB; the stored tile is (o, i), or (i, o) for the
// transposed product.  Tiles are read three steps ahead of their use (an LDS read takes ~100+ cycles and longer under
// load, the four MFMAs of a step 64); the empty asm pins the read in front of the MFMAs it overlaps.
// DIAG (class-fused nucleotide model: one 4-state tile per class, the operator is block diagonal): only the NB diagonal
// tiles are applied, step q = tile (q, q) -- the same bits as the dense product, whose other tiles are exact zeros.
template <int S, bool TR, bool DIAG>
__device__ __forceinline__ constexpr int mfma_tile(int q) {
  return DIAG ? q : (TR ? (q % (S / 4)) * (S / 4) + q / (S / 4) : q);   // (DIAG: the host stores tile (q, q) at position q)
}
template <int S, bool TR, int NG, bool DIAG, int Q>
__device__ __forceinline__ void mfma_steps(const uint8_t* tile0, double m0, double m1, double m2, const double (&x)[S / 4 * NG],
                                           double (&y)[S / 4 * NG]) {
  constexpr int NB = S / 4, NT = DIAG ? NB : NB * NB;
  if constexpr (Q < NT) {
    constexpr int o = DIAG ? Q : Q / NB, i = DIAG ? Q : Q % NB;
    double m3 = 0.0;
    if constexpr (Q + 3 < NT) {
      m3 = *reinterpret_cast<const double*>(tile0 + mfma_tile<S, TR, DIAG>(Q + 3) * 128);
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int g = 0; g < NG; ++g)
      y[o * NG + g] = __builtin_amdgcn_mfma_f64_4x4x4f64(m0, x[i * NG + g], (DIAG || i == 0) ? 0.0 : y[o * NG + g], 0, 0, 0);
    mfma_steps<S, TR, NG, DIAG, Q + 1>(tile0, m1, m2, m3, x, y);
  }
}

// y = M x (TR = false) or y = M^T x (TR = true) with M = the packed matrix staged in buf (already landed)
template <int S, bool TR, int NG, bool DIAG>
__device__ __forceinline__ void matvec_stage(const uint8_t* buf, int lane, const double (&x)[S / 4 * NG],
                                             double (&y)[S / 4 * NG]) {
  static_assert(S % 4 == 0, "state count must be a multiple of 4");
  constexpr int NT = DIAG ? S / 4 : (S / 4) * (S / 4);
  // element (row r, column c) of a packed tile sits at (4 r + c) * 8; the A slot of lane l is row l & 3, column l >> 4
  // (transposed product: the transposed tile, i.e. row l >> 4, column l & 3)
  const uint8_t* tile0 = buf + (TR ? (4 * (lane >> 4) + (lane & 3)) : (4 * (lane & 3) + (lane >> 4))) * 8;
  mfma_steps<S, TR, NG, DIAG, 0>(tile0, *reinterpret_cast<const double*>(tile0 + mfma_tile<S, TR, DIAG>(0) * 128),
                       NT > 1 ? *reinterpret_cast<const double*>(tile0 + mfma_tile<S, TR, DIAG>(1) * 128) : 0.0,
                       NT > 2 ? *reinterpret_cast<const double*>(tile0 + mfma_tile<S, TR, DIAG>(2) * 128) : 0.0, x, y);
}

// Message of a leaf edge from the transposed operator staged in buf ([z][x] = M[x][z], rows >= S: ambiguity ids):
// lane l needs, for each site group g, element 4 sb + (l >> 4) of the row named by the symbol of site 16 g + (l & 15).
// SET out = row, MUL out = row o in (in may be out), DOT returns sum_x in[x] * row[x] of site l (reduce_sites).
enum { LEAF_SET = 0, LEAF_MUL = 1, LEAF_DOT = 2 };
template <int S, int MODE, int NG>
__device__ __forceinline__ double leaf_apply(const uint8_t* buf, const uint8_t* codes /* slot + 4 * (lane & 15) */, int lane,
                                             const double (&in)[S / 4 * NG], double (&out)[S / 4 * NG]) {
  constexpr int NB = S / 4;
  double part[NG];
  // all symbols first (one LDS round trip), then per site group the row values and their use: the reads of the next
  // group do not depend on anything computed for the previous one
  unsigned code[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) code[g] = codes[g * (NG == 3 ? 64 : 256 / NG)];   // one dword per LANE in the slot, group g at 64 / NG * g (three groups: as four)
  const double* r[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const unsigned row = code[g] < (unsigned)MatStage<S>::NROW ? code[g] : (unsigned)(MatStage<S>::NROW - 1);
    // rows are stored state-in-tile major: state 4 sb + s4 at position s4 * NB + sb (cmx_host_model.cpp)
    r[g] = reinterpret_cast<const double*>(buf + row * (leaf_row_stride(S) * 8)) + (lane >> 4) * NB;
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    part[g] = 0.0;
    double v[NB];
#pragma unroll
    for (int sb = 0; sb < NB; ++sb) v[sb] = r[g][sb];
#pragma unroll
    for (int sb = 0; sb < NB; ++sb) {
      if (MODE == LEAF_SET) out[sb * NG + g] = v[sb];
      else if (MODE == LEAF_MUL) out[sb * NG + g] = v[sb] * in[sb * NG + g];
      else part[g] = __builtin_fma(in[sb * NG + g], v[sb], part[g]);
    }
  }
  if (MODE != LEAF_DOT) return 0.0;
  return reduce_sites<NG>(part);
}

// Row of a cherry table (cmx_walk.h): the row is named by the symbols of the cherry's two leaves, 4 * symbol(l1) + symbol(l2)
// (fully resolved alignments only: both symbols are states of the 4-letter alphabet), the columns are a leaf row's.
template <int S, int MODE, int NG>
__device__ __forceinline__ double cherry_apply(const uint8_t* buf, const uint8_t* codes1, const uint8_t* codes2, int lane,
                                               const double (&in)[S / 4 * NG], double (&out)[S / 4 * NG]) {
  constexpr int NB = S / 4;
  double part[NG];
  unsigned c1[NG], c2[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) { c1[g] = codes1[g * (NG == 3 ? 64 : 256 / NG)]; c2[g] = codes2[g * (NG == 3 ? 64 : 256 / NG)]; }
  const double* r[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const unsigned row = ((c1[g] & 3u) << 2) | (c2[g] & 3u);
    r[g] = reinterpret_cast<const double*>(buf + row * (leaf_row_stride(S) * 8)) + (lane >> 4) * NB;
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    part[g] = 0.0;
    double v[NB];
#pragma unroll
    for (int sb = 0; sb < NB; ++sb) v[sb] = r[g][sb];
#pragma unroll
    for (int sb = 0; sb < NB; ++sb) {
      if (MODE == LEAF_SET) out[sb * NG + g] = v[sb];
      else part[g] = __builtin_fma(in[sb * NG + g], v[sb], part[g]);
    }
  }
  if (MODE != LEAF_DOT) return 0.0;
  return reduce_sites<NG>(part);
}

// ------------------------------------------------------------------------------------------------ small helpers
// Workspace vectors are stored as [S/2][64 lanes][2 doubles]: one 16-byte access per lane and row, 1 KiB per
// wave-instruction, the same image in HBM and (for prefetched vectors) in LDS.  An odd length (16-site tasks: S = 5) ends
// with a row of single doubles.
// (Round 4 tried [S][64] with 64-bit accesses: hipcc puts v_mov copies and the s_waitcnt for them right behind the 128-bit
// loads at 3 of the 5 load sites of map_kernel<20, null> -- the loaded quads do not match the registers the vector's phi
// with the other arms of the child switch got -- and 64-bit loads remove copies and early waits alike.  Twice the VMEM
// instructions cost what that saved: target launch 445 -> 450 ms, same box, profiles/r04_ab_workspace_64bit.log.)
template <int S>
__device__ __forceinline__ void load_vec(const double* p /* slice base */, double (&v)[S], int lane) {
  p += 2 * lane;
#pragma unroll
  for (int i = 0; i < S / 2; ++i) {
    const d2 t = *reinterpret_cast<const d2*>(p + (size_t)i * 2 * kWave);
    v[2 * i] = t[0];
    v[2 * i + 1] = t[1];
  }
  if constexpr (S % 2) v[S - 1] = p[(size_t)(S / 2) * 2 * kWave - lane];
}
template <int S>
__device__ __forceinline__ void store_vec(double* p, const double (&v)[S], int lane) {
  p += 2 * lane;
#pragma unroll
  for (int i = 0; i < S / 2; ++i) {
    d2 t;
    t[0] = v[2 * i];
    t[1] = v[2 * i + 1];
    *reinterpret_cast<d2*>(p + (size_t)i * 2 * kWave) = t;
  }
  if constexpr (S % 2) p[(size_t)(S / 2) * 2 * kWave - lane] = v[S - 1];
}

// asynchronous HBM -> LDS copy of one workspace vector (S/2 LDS-DMA instructions, no VGPR destination)
template <int S>
__device__ __forceinline__ void prefetch_vec_lds(const double* p /* slice base + 2*lane */, uint8_t* lds /* wave-uniform */) {
  constexpr int R = S / 2;
  const uint32_t l = lds_addr(lds);
#pragma unroll
  for (int i = 0; i + 4 <= R; i += 4) dma_rows16<4>(p + (size_t)i * 2 * kWave, l + i * 1024);
  if constexpr (R % 4 != 0) dma_rows16<R % 4>(p + (size_t)(R / 4) * 4 * 2 * kWave, l + (R / 4) * 4 * 1024);
}
template <int S>
__device__ __forceinline__ void read_vec_lds(const uint8_t* lds /* wave-uniform */, int lane, double (&v)[S]) {
#pragma unroll
  for (int i = 0; i < S / 2; ++i) {
    const d2 t = *reinterpret_cast<const d2*>(lds + (size_t)(i * kWave + lane) * 16);
    v[2 * i] = t[0];
    v[2 * i + 1] = t[1];
  }
}

// Philox2x32-10 (Random123): counter = (g_lo, g_hi[14:0] | draw << 15), key = seed_lo ^ seed_hi * 0x9E3779B9 ^ 'CMX2'.
// Same scheme as oracle/oracle.c (DESIGN.md "RNG").  One 32 x 32 -> 64 multiply per round: the 4x32 variant (two per
// round, 128 output bits) cost the fused null kernel 4.7 % of its time in quarter-rate integer multiplies.
// g < 2^47, draw < 2^17.
//   draw 0 (rate class / continuous rate) and 1 (root state): counter from the site's g, the 64 output bits give one
//   53-bit uniform (philox_uniform);
//   draw 2 + node (state at the lower end of a branch): the sites 2k and 2k + 1 SHARE the call with counter g >> 1 and
//   take its first and second output word as a 32-bit uniform (philox_node_uniform): the node draws are 99 % of a
//   simulation's calls, a category's probability is resolved to 2^-32 either way, and a thread that holds both sites of
//   a pair (simulate_lds_kernel) makes one call for two draws.
__device__ __forceinline__ void philox_words(uint64_t seed, uint64_t g, uint32_t draw, uint32_t& w0, uint32_t& w1) {
  uint32_t c0 = (uint32_t)g, c1 = ((uint32_t)(g >> 32) & 0x7fffu) | (draw << 15);
  uint32_t k = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u) ^ 0x434d5832u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p = (uint64_t)0xD256D193u * (uint64_t)c0;   // one v_mad_u64_u32 instead of v_mul_hi + v_mul_lo
    c0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p >> 32), k, c1, 0x96);   // three-way xor in one v_bitop3_b32
    c1 = (uint32_t)p;
    k += 0x9E3779B9u;
  }
  w0 = c0;
  w1 = c1;
}
__device__ __forceinline__ double philox_uniform(uint64_t seed, uint64_t g, uint32_t draw) {
  uint32_t c0, c1;
  philox_words(seed, g, draw, c0, c1);
  const uint64_t bits = (((uint64_t)c0 << 32) | c1) >> 11;
  return (double)bits * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double philox_node_uniform(uint64_t seed, uint64_t g, uint32_t node) {
  uint32_t w0, w1;
  philox_words(seed, g >> 1, 2u + node, w0, w1);
  return (double)((g & 1) ? w1 : w0) * (1.0 / 4294967296.0);
}

template <class CumPtr>
__device__ __forceinline__ int draw_index(double u, CumPtr cum, int n) {
  int idx = 0;
  for (int j = 0; j < n - 1; ++j) idx += (u >= cum[j]) ? 1 : 0;
  return idx;
}
// the same index (#{ j < n-1 : u >= cum[j] }, cum non-decreasing) found from a guide table: entry k = the number of
// leading running sums that are <= k/32, so the scan for a u in [k/32, (k+1)/32) starts there -- one or two reads of the
// lane's own row instead of n - 1 (the rows are lane-divergent 160-byte gathers: the simulator's whole cost)
__device__ __forceinline__ int draw_guided(double u, const double* __restrict__ cum, const uint8_t* __restrict__ guide, int n) {
  int idx = guide[(int)(u * 32.0)];
  while (idx < n - 1 && u >= cum[idx]) ++idx;
  return idx;
}

// ------------------------------------------------------------------------------------------------ mapping core
extern __shared__ __attribute__((aligned(16))) uint8_t cmx_smem[];

// Read-only, wave-uniform metadata (tree program, schedules, pi, class rates) is read through the CONSTANT address
// space so that hipcc emits s_load (scalar cache, lgkmcnt) instead of global_load + v_readfirstlane: the latter would
// queue every dependent tree-walk step behind the HBM prefetches outstanding on vmcnt.
typedef const int __attribute__((address_space(4)))* cmx_cint;
typedef const double __attribute__((address_space(4)))* cmx_cdbl;
// per-visited-node record of the tree walk (cmx_walk.h): 16 ints, one scalar load
typedef int cmx_i16 __attribute__((ext_vector_type(16)));
typedef int cmx_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sload_rec(cmx_cint p, cmx_i16& r) {
  asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p) : "memory");
}

struct ConstModel {
  cmx_cint taxon_of, parent, nrec, msched, ldsched;
  cmx_cdbl pi, rates, probs, cum_pi, cum_probs;
  // tables: the cherry-table walk's operator stream (class-fused nucleotide models on resolved alignments)
  __device__ __forceinline__ explicit ConstModel(const DevModel& m, bool tables = false)
      : taxon_of((cmx_cint)m.taxon_of), parent((cmx_cint)m.parent), nrec((cmx_cint)m.nrec),
        msched((cmx_cint)(tables ? m.msched_r : m.msched)), ldsched((cmx_cint)m.ldsched), pi((cmx_cdbl)m.pi), rates((cmx_cdbl)m.rates),
        probs((cmx_cdbl)m.probs), cum_pi((cmx_cdbl)m.cum_pi), cum_probs((cmx_cdbl)m.cum_probs) {}
};
// Operator-stream bookkeeping of a wave (wave-uniform, lives across site blocks).  vs counts the VMEM instructions this
// code issued and knows about (DMA rows, vector stores); an asynchronous transfer remembers vs right after its issue,
// and "wait for X" is s_waitcnt vmcnt(vs - X_seq): VMEM completes in order, so the instructions issued after X may stay
// in flight.  Instructions that are not counted only make the wait stricter.
// (The phase timers, the ablation switches and the round-1 fused-simulator arrangement that produced the figures quoted in
// DESIGN.md were stripped from this file in round 3: scripts/experiments/ablations/README.md names the tagged source.)
struct OpState {
  int pre_mat, pre_tx;  // op-stream entry of the NEXT op, loaded one op early (its latency hides behind the current op)
  unsigned par;      // stage buffer / code slot of the current operator op
  unsigned vs;       // counted VMEM instructions issued so far
  unsigned cur_seq;  // vs right after the current operator op's operator (and symbols) were requested
};

// Device backend of the tree walk (cmx_walk.h): registers R0..R3 are S-vectors of the wave's sites in the matrix-core
// layout.  The walk names what it wants; the operator stream (m.msched) and the load schedule (m.ldsched) -- recorded
// from the same walk on the host and checked numerically there -- say where it is.
// RESOLVED: every symbol of the alignment is a state (the null's simulated or supplied alignments): leaf ops then stage the
// S0 state rows of the transposed operator only, not its ambiguity rows
template <int S, int FUSE, int NG, bool RESOLVED>
struct DevWalk {
  static constexpr int VL = S / 4 * NG, kSites = 16 * NG;
  static constexpr int kRow = NG == 3 ? 64 : kSites;   // lanes per row of the wave's per-site scratch arrays (three groups: lanes 48 .. 63 own unused columns)
  static constexpr bool DIAG = FUSE > 1 && S / FUSE == 4;
  static constexpr int kProductBytes = DIAG ? (S / 4) * 128 : MatStage<S>::BYTES;                      // diagonal tiles come first
  static constexpr int kLeafBytes = RESOLVED ? (S / FUSE) * leaf_row_stride(S) * 8 : MatStage<S>::BYTES;
#ifndef CMX_NO_CHERRY_TABLES
#define CMX_NO_CHERRY_TABLES 0     // 1: A/B builds without the cherry tables (scripts/ab_libs.sh)
#endif
  static constexpr int kVecInstrs = (VL + 1) / 2;   // VMEM instructions of one workspace vector
  static constexpr bool kCherryTables = RESOLVED && DIAG && !CMX_NO_CHERRY_TABLES;   // cmx_walk.h: cset / cdot instead of a cherry's operator ops
  static constexpr int kCherryBytes = 16 * leaf_row_stride(S) * 8;   // a table's 16 rows (one per symbol pair)
  static constexpr int kCherryFlag = 0x40000000;                     // stream entry: taxon of l1 | taxon of l2 << 15 | flag
  static_assert(kCherryBytes <= MatStage<S>::BYTES, "a cherry table fits a stage buffer");
  double R0[VL], R1[VL], R2[VL], R3[VL];
  OpState& os;
  const ConstModel& cm;
  const double* pi;            // root frequencies (vector loads: lane-dependent index)
  int nmv;
  const double *mat_c, *mat_after;
  double *wsM, *wsU, *pcnt;
  const uint8_t* gcodes;
  size_t gstride;
  uint8_t *stage, *cslot, *cslot2;
  uint32_t lds_stage, lds_codes, lds_codes2;
  int lane, c, c_end;
  double pc;
  double Lg[FUSE];
  int mi;

  __device__ __forceinline__ DevWalk(OpState& os_, const ConstModel& cm_) : os(os_), cm(cm_) {}

  // The lane index as an opaque value: address arithmetic built on it is redone at every use (a few VALU instructions)
  // instead of being hoisted out of the walk as a dozen loop-invariant 64-bit per-lane addresses, which the register
  // allocator then spills -- and every reload of a spilled address is a scratch load whose vmcnt(0) wait drains the
  // DMAs in flight.
  __device__ __forceinline__ int vlane() const {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  }
  // site of the lane inside the wave's block (see map_sites_wave)
  __device__ __forceinline__ int vsidx() const {
    const int l = vlane();
    return NG >= 3 ? l : (NG == 2 ? (((l >> 5) << 4) | (l & 15)) : (l & 15));
  }
  template <int I>
  __device__ __forceinline__ double (&reg())[VL] {
    if constexpr (I == 0) return R0;
    else if constexpr (I == 1) return R1;
    else if constexpr (I == 2) return R2;
    else return R3;
  }
  __device__ __forceinline__ void begin_pass() {
    mi = 0;
#pragma unroll
    for (int g = 0; g < FUSE; ++g) Lg[g] = 0.0;
  }
  __device__ __forceinline__ void rec(int v, int (&r)[16]) const {
    cmx_i16 q;
    sload_rec(cm.nrec + v * 16, q);
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = q[i];
  }
  // tells the compiler a register's old value is dead here (defines it without an instruction)
  // (ONE asm statement: hipcc puts an `s_nop 0` behind every inline asm, and a wave issues one instruction per four
  // cycles -- sixty one-element kills per visited node were sixty issue slots of the walk's chain: target launch -0.6 %, same box)
  template <int R>
  __device__ __forceinline__ void kill() {
    double(&r)[VL] = reg<R>();
    static_assert(VL == 4 || VL == 5 || VL == 8 || VL == 12 || VL == 15 || VL == 16 || VL == 20, "kill() names every element in one statement");
    if constexpr (VL == 12)
      asm volatile("" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]));
    else if constexpr (VL == 15)
      asm volatile("" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]));
    else
    if constexpr (VL == 5)
      asm volatile("" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]));
    else if constexpr (VL == 4)
      asm volatile("" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]));
    else if constexpr (VL == 8)
      asm volatile("" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]));
    else if constexpr (VL == 16)
      asm volatile("" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]),
                        "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]));
    else
      asm volatile("" : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]),
                        "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]), "=v"(r[16]),
                        "=v"(r[17]), "=v"(r[18]), "=v"(r[19]));
  }
  // One operator op: request the operator (and symbols) of the NEXT op of the stream -- entry mi + 1, or entry 0 of the
  // next class / next site block -- into the other buffer, then wait for this op's operator.  Returns its buffer.
  __device__ __forceinline__ const uint8_t* op_begin(unsigned& nseq) {
    const bool more = (mi + 1 < nmv);
    const int emat = os.pre_mat, etx = os.pre_tx;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS reads of the other buffer are done
    unsigned issued;
    {
      const double* src = (more ? mat_c : mat_after) + emat;   // element offset, premultiplied on the host
      const uint32_t dst = lds_stage + (os.par ^ 1u) * MatStage<S>::BYTES;
      if constexpr (kProductBytes == MatStage<S>::BYTES && kLeafBytes == MatStage<S>::BYTES) {
        mat_dma_l<S>(src, dst, vlane());
        issued = MatStage<S>::ROWS;
      } else if (etx >= 0) {   // (wave-uniform: the next op is a leaf op -- or a cherry-table op)
        if (kCherryTables && (etx & kCherryFlag)) {
          mat_dma_part<kCherryBytes>(src, dst, vlane());
          issued = MatPart<kCherryBytes>::ROWS;
        } else {
          mat_dma_part<kLeafBytes>(src, dst, vlane());
          issued = MatPart<kLeafBytes>::ROWS;
        }
      } else {
        mat_dma_part<kProductBytes>(src, dst, vlane());
        issued = MatPart<kProductBytes>::ROWS;
      }
    }
    if (etx >= 0 && (more || c + 1 < c_end)) {
      if (kCherryTables && (etx & kCherryFlag)) {
        code_dma_l(gcodes + (size_t)(etx & 0x7fff) * gstride, lds_codes + (os.par ^ 1u) * kCodeSlotBytes);
        code_dma_l(gcodes + (size_t)((etx >> 15) & 0x7fff) * gstride, lds_codes2 + (os.par ^ 1u) * kCodeSlotBytes);
        issued += 2;
      } else {
        code_dma_l(gcodes + (size_t)etx * gstride, lds_codes + (os.par ^ 1u) * kCodeSlotBytes);
        issued += 1;
      }
    }
    {  // entry two ops ahead; the stream carries its first two entries again after the last (no wrap test)
      const int i2 = more ? mi + 2 : 1;
      os.pre_mat = cm.msched[2 * i2];
      os.pre_tx = cm.msched[2 * i2 + 1];
    }
    wait_vm<S, VL>((int)(os.vs - os.cur_seq + issued));
    os.vs += issued;
    nseq = os.vs;
    return stage + os.par * MatStage<S>::BYTES;
  }
  __device__ __forceinline__ void op_end(unsigned nseq) {
    os.par ^= 1u;
    os.cur_seq = nseq;
    ++mi;
  }
  template <int SRC, int DST, bool TR>
  __device__ __forceinline__ void mv(int, int) {
    unsigned nseq;
    const uint8_t* buf = op_begin(nseq);
    matvec_stage<S, TR, NG, DIAG>(buf, vlane(), reg<SRC>(), reg<DST>()); asm volatile("" :: "v"(reg<DST>()[0]), "v"(reg<DST>()[VL - 1]));
    op_end(nseq);
  }
  template <int MODE, int SRC, int DST>
  __device__ __forceinline__ double leaf(void) {
    unsigned nseq;
    const uint8_t* buf = op_begin(nseq);
    const int vl = vlane();
    const double tot = leaf_apply<S, MODE, NG>(buf, cslot + os.par * kCodeSlotBytes + 4 * (vl & 15), vl, reg<SRC>(), reg<DST>());
    asm volatile("" :: "v"(tot), "v"(reg<DST>()[0]), "v"(reg<DST>()[VL - 1]));
    op_end(nseq);
    return tot;
  }
  template <int D> __device__ __forceinline__ void lset(int, int) { (void)leaf<LEAF_SET, D, D>(); }
  template <int SRC, int D> __device__ __forceinline__ void lmul(int, int) { (void)leaf<LEAF_MUL, SRC, D>(); }
  template <int SRC> __device__ __forceinline__ void ldot(int, int, int row) {
    const double tot = leaf<LEAF_DOT, SRC, SRC>();
    pcnt[(size_t)row * kRow + vsidx()] = pc * tot;
  }
  // cherry-table ops (cmx_walk.h): the table staged like a leaf operator, its row named by two symbols
  template <int MODE, int SRC, int DST>
  __device__ __forceinline__ double cherry(void) {
    unsigned nseq;
    const uint8_t* buf = op_begin(nseq);
    const int vl = vlane();
    const double tot = cherry_apply<S, MODE, NG>(buf, cslot + os.par * kCodeSlotBytes + 4 * (vl & 15), cslot2 + os.par * kCodeSlotBytes + 4 * (vl & 15),
                                                 vl, reg<SRC>(), reg<DST>());
    asm volatile("" :: "v"(tot), "v"(reg<DST>()[0]), "v"(reg<DST>()[VL - 1]));
    op_end(nseq);
    return tot;
  }
  template <int D> __device__ __forceinline__ void cset(int, int, int) { (void)cherry<LEAF_SET, D, D>(); }
  template <int SRC> __device__ __forceinline__ void cdot(int, int, int, int, int row) {
    const double tot = cherry<LEAF_DOT, SRC, SRC>();
    pcnt[(size_t)row * kRow + vsidx()] = pc * tot;
  }
  // workspace vector -> register: plain global loads straight into the destination register, which the walk has
  // declared dead (kill) long before -- in the outside pass the two sibling messages of a node are requested before the
  // first product and first read after it, so their latency hides behind ~100 MFMAs without a landing buffer.  The
  // loads are compiler-visible (hipcc waits for them before the first use); they are counted in vs like every VMEM
  // instruction this code issues so that the counted waits of the operator stream let them stay in flight.
  template <int D>
  __device__ __forceinline__ void load(int arr, int slot) {
    {
      const int vl = vlane();
      load_vec<VL>((arr ? wsU : wsM) + (size_t)slot * VL * kWave, reg<D>(), vl);
    }
    os.vs += kVecInstrs;
  }
  template <int SRC>
  __device__ __forceinline__ void store(int arr, int slot) {
    {
      const int vl = vlane();
      store_vec<VL>((arr ? wsU : wsM) + (size_t)slot * VL * kWave, reg<SRC>(), vl);
    }
    os.vs += kVecInstrs;
  }
  template <int D, int SRC> __device__ __forceinline__ void mov() {
#pragma unroll
    for (int x = 0; x < VL; ++x) reg<D>()[x] = reg<SRC>()[x];
  }
  template <int D, int SRC> __device__ __forceinline__ void mul() {
#pragma unroll
    for (int x = 0; x < VL; ++x) reg<D>()[x] *= reg<SRC>()[x];
  }
  __device__ __forceinline__ void mulup() {
#pragma unroll
    for (int x = 0; x < VL; ++x) { R1[x] *= R3[x]; R2[x] *= R3[x]; }
  }
  template <int D> __device__ __forceinline__ void setpi() {
#pragma unroll
    for (int sb = 0; sb < S / 4; ++sb) {
      const double pv = pi[(4 * sb + (vlane() >> 4)) % (S / FUSE)];
#pragma unroll
      for (int g = 0; g < NG; ++g) reg<D>()[sb * NG + g] = pv;
    }
  }
  template <int SRC> __device__ __forceinline__ void rootl() {
    // lane l holds state 4 sb + (l >> 4) of its four sites: weight it with that state's frequency
    if constexpr (FUSE == 1) {
      double p_[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) p_[g] = 0.0;
#pragma unroll
      for (int sb = 0; sb < S / 4; ++sb) {
        const double pv = pi[4 * sb + (vlane() >> 4)];
#pragma unroll
        for (int g = 0; g < NG; ++g) p_[g] = __builtin_fma(pv, reg<SRC>()[sb * NG + g], p_[g]);
      }
      Lg[0] = reduce_sites<NG>(p_);
    } else {   // one 4-state tile per fused class
      const double pv = pi[vlane() >> 4];
#pragma unroll
      for (int sb = 0; sb < FUSE; ++sb) {
        double p_[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) p_[g] = pv * reg<SRC>()[sb * NG + g];
        Lg[sb] = reduce_sites<NG>(p_);
      }
    }
  }
  __device__ __forceinline__ void dot3(int row) {
    double p_[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) p_[g] = 0.0;
#pragma unroll
    for (int sb = 0; sb < S / 4; ++sb)
#pragma unroll
      for (int g = 0; g < NG; ++g) p_[g] = __builtin_fma(R3[sb * NG + g] * R1[sb * NG + g], R2[sb * NG + g], p_[g]);
    pcnt[(size_t)row * kRow + vsidx()] = pc * reduce_sites<NG>(p_);
  }
};

// Maps the 64 sites of this wave (symbol of taxon t at gcodes[t * gstride], per lane) for all rate classes: one walk of
// the tree (cmx_walk.h) per class.  Classes [c_begin, c_end) are processed; c_after is the class of the first pass that
// follows this call (its first operator is requested by the last operator op here).  With finalize, on return
// cnt[(b*K+k)*64 + lane] holds the final counts n(b, site, k) and the scalars are per lane; without (class-split
// observed mode) only part[] and L_out = sum of p_c L_c over the processed classes are produced.
// part: [C][B*K][64] per-class joint counts (written once each, summed at the end: no read-modify-write in the loop).
template <int S, int FUSE, int NG, bool RESOLVED>
__device__ __forceinline__ void map_sites_wave(const MapArgs& a, double* __restrict__ wsM, double* __restrict__ wsU,
                                               double* __restrict__ part, double* __restrict__ cnt, int lds_off,
                                               const uint8_t* __restrict__ gcodes, size_t gstride, int lane,
                                               OpState& os, double& L_out, double& pr_out, int& rc_out,
                                               double& norm_out, int c_begin, int c_end, int c_after, bool finalize) {
  const DevModel& m = a.m;
  constexpr bool kTables = DevWalk<S, FUSE, NG, RESOLVED>::kCherryTables;   // (the launcher guarantees m.msched_r then)
  const ConstModel cm(m, kTables);
  constexpr int kSites = 16 * NG;      // sites per wave (NG site groups of 16; an S-vector is S / 4 * NG doubles per lane)
  constexpr int kRow = NG == 3 ? 64 : kSites;   // row length of the wave's per-site scratch arrays
  // site of this lane inside the wave's block for per-site scalars and arrays: NG = 4: the lane itself; NG = 2: lanes
  // l and l ^ 16 both carry site 16 (l >> 5) + (l & 15) and do the per-site work redundantly (identical values)
  const int sidx = NG >= 3 ? lane : (NG == 2 ? (((lane >> 5) << 4) | (lane & 15)) : (lane & 15));
  const int C = m.C, K = m.K;
  double Lsum = 0.0, prsum = 0.0, best = -1.0;
  int bestc = 0;
  DevWalk<S, FUSE, NG, RESOLVED> be(os, cm);
  be.pi = m.pi;
  be.nmv = kTables ? m.nmv_r : m.nmv;
  be.wsM = wsM;
  be.wsU = wsU;
  be.gcodes = gcodes;
  be.gstride = gstride;
  be.stage = cmx_smem + lds_off;                                 // two operator buffers
  be.cslot = be.stage + 2 * MatStage<S>::BYTES;                  // two symbol slots
  be.cslot2 = be.cslot + 2 * kCodeSlotBytes;                     // ... and two for the second leaf of a cherry-table op
  be.lds_stage = lds_addr(be.stage);                             // wave-uniform LDS byte addresses (SGPRs)
  be.lds_codes = lds_addr(be.cslot);
  be.lds_codes2 = lds_addr(be.cslot2);
  be.lane = lane;
  be.c_end = c_end;
  {  // symbols of op 0 if it is a leaf op (the previous site block could not request them).  Everything this wave
     // wrote before (simulated symbols) is in L2 first; the request is not counted, so drain it here.
    const int tx0 = cm.msched[1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tx0 >= 0) {
      if (kTables && (tx0 & 0x40000000)) {
        code_dma(gcodes + (size_t)(tx0 & 0x7fff) * gstride, be.cslot + os.par * kCodeSlotBytes);
        code_dma(gcodes + (size_t)((tx0 >> 15) & 0x7fff) * gstride, be.cslot2 + os.par * kCodeSlotBytes);
      } else {
        code_dma(gcodes + (size_t)tx0 * gstride, be.cslot + os.par * kCodeSlotBytes);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  for (int c = c_begin; c < c_end; ++c) {
    // operators of this class and of the class of the pass that follows (its first operator is requested by our last op)
    be.mat_c = m.MAT + (size_t)c * m.MC * MatStage<S>::UNIT;
    be.mat_after = m.MAT + (size_t)((c + 1 < c_end) ? c + 1 : c_after) * m.MC * MatStage<S>::UNIT;
    const double pc = FUSE > 1 ? 1.0 : cm.probs[c];  // fused: the class probabilities are folded into the count operators
    be.pc = pc;
    be.c = c;
    be.pcnt = part + (size_t)c * m.B * K * kRow;   // wave-uniform; the lane's site is added at each use
    be.begin_pass();
    walk_pass(be, m.NV, K);
    if constexpr (FUSE == 1) {
      const double Lc = be.Lg[0];
      Lsum += pc * Lc;
      prsum += cm.rates[c] * pc * Lc;
      if (pc * Lc > best) { best = pc * Lc; bestc = c; }  // first maximum wins (getRateClassWithMaxPostProbPerSite)
    } else {
#pragma unroll
      for (int g = 0; g < FUSE; ++g) {
        const int cc = c * FUSE + g;
        if (cc < m.C0) {
          const double pg = cm.probs[cc];
          Lsum += pg * be.Lg[g];
          prsum += cm.rates[cc] * pg * be.Lg[g];
          if (pg * be.Lg[g] > best) { best = pg * be.Lg[g]; bestc = cc; }
        }
      }
    }
  }
  if (!finalize) {   // class-split mode: the pass's sums, its best class and that class's weight
    L_out = Lsum;
    pr_out = prsum;
    rc_out = bestc;
    norm_out = best;
    return;
  }
  // ---------------- sum the classes in class order, divide by the site likelihood, norm (computeNormForSite)
  // Rows r = b*K + k are taken sixteen at a time and classes four at a time so that 64 independent loads are in flight
  // (this phase is pure L2 latency); the sums run in the same order as a plain (b, k, c) loop nest.
  double nrm = 0.0, tot = 0.0;
  const int BK = m.B * K;
  int kk = 0;  // r % K
  constexpr int RC = 16;
  for (int r0 = 0; r0 < BK; r0 += RC) {
    double v[RC];
#pragma unroll
    for (int u = 0; u < RC; ++u) v[u] = 0.0;
    int c = 0;
    for (; c + 4 <= C; c += 4) {
      double t[4][RC];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double* pp = part + ((size_t)(c + q) * BK + r0) * kRow + sidx;
#pragma unroll
        for (int u = 0; u < RC; ++u) t[q][u] = pp[(size_t)(r0 + u < BK ? u : 0) * kRow];
      }
#pragma unroll
      for (int u = 0; u < RC; ++u) v[u] = (((v[u] + t[0][u]) + t[1][u]) + t[2][u]) + t[3][u];
    }
    for (; c < C; ++c) {
      const double* pp = part + ((size_t)c * BK + r0) * kRow + sidx;
#pragma unroll
      for (int u = 0; u < RC; ++u) v[u] += pp[(size_t)(r0 + u < BK ? u : 0) * kRow];
    }
#pragma unroll
    for (int u = 0; u < RC; ++u) {
      if (r0 + u < BK) {
        const double q = v[u] / Lsum;
        cnt[(size_t)(r0 + u) * kRow + sidx] = q;
        tot += q;
        if (++kk == K) {
          nrm = __builtin_fma(tot, tot, nrm);
          tot = 0.0;
          kk = 0;
        }
      }
    }
  }
  L_out = Lsum;
  pr_out = prsum / Lsum;
  rc_out = bestc;
  norm_out = sqrt(nrm);
}

// site groups of 16 per wave: 4 (64 sites, two waves per SIMD) or 2 (32 sites: the four live S-vectors take 80 registers
// instead of 160 and three waves fit a SIMD; operators are then staged per 32 sites)
template <int S>
constexpr int map_lds_fixed() { return 2 * MatStage<S>::BYTES + 4 * kCodeSlotBytes; }   // stage buffers + symbol slots (two leaves)
// + the simulator's node states (one byte per node and site) when they fit what is left of the CU's LDS share

template <int S, int MODE, int FUSE, int NG = map_ng(S)>
__global__ __launch_bounds__(kWave * kWavesPerBlock, map_waves_per_simd(S)) void map_kernel(const MapArgs a) {
  constexpr int VL = S / 4 * NG, kSites = 16 * NG;
  constexpr int kRow = NG == 3 ? 64 : kSites;
  const DevModel& m = a.m;
  // the null's alignments are fully resolved: class-fused nucleotide models walk them with the cherry tables' stream
  constexpr bool kTables = MODE == kModeNull && DevWalk<S, FUSE, NG, true>::kCherryTables;
  const ConstModel cm(m, kTables);
  const int lane = threadIdx.x & (kWave - 1);
  const int sidx = NG >= 3 ? lane : (NG == 2 ? (((lane >> 5) << 4) | (lane & 15)) : (lane & 15));   // site of this lane in the wave's block
  // wave index through readfirstlane: the compiler cannot see that threadIdx.x >> 6 is wave-uniform and would keep every
  // per-wave base pointer below as a per-lane 64-bit VGPR pair (spilled, and reloaded from scratch in the hot loop)
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = blockIdx.x * kWavesPerBlock + wib;
  const int nwaves = gridDim.x * kWavesPerBlock;
  double* wsD = a.ws.D + (size_t)wave * m.NIW * VL * kWave;
  double* wsU = a.ws.U + (size_t)wave * m.NIW * VL * kWave;
  double* cnt0 = a.ws.cnt + (size_t)wave * 2 * m.B * m.K * kRow;
  double* cnt1 = cnt0 + (size_t)m.B * m.K * kRow;
  double* part = a.ws.part + (size_t)wave * m.C * m.B * m.K * kRow;
  // LDS per wave: workspace prefetch buffer (S*64*8 B), two operator stage buffers, two symbol slots
  const int lds_off = wib * a.lds_per_wave;
  const size_t nblocks = (a.nsites + kSites - 1) / kSites;
  // request the first op's operator (class 0, entry 0); every op then requests the next one
  OpState os;
  os.par = 0;
  {
    const int c0 = (MODE == kModeObservedSplit) ? wave % m.C : 0;   // class of this wave's first pass
    mat_dma<S>(m.MAT + (size_t)c0 * m.MC * MatStage<S>::UNIT + cm.msched[0], cmx_smem + lds_off, lane);
  }
  os.vs = MatStage<S>::ROWS;
  os.cur_seq = os.vs;
  os.pre_mat = cm.msched[2];   // entry 1 (the stream is padded with copies of its first entries)
  os.pre_tx = cm.msched[3];
  if (MODE == kModeObservedSplit) {
    const size_t ntasks = nblocks * (size_t)m.C, BK = (size_t)m.B * m.K;
    for (size_t task = wave; task < ntasks; task += nwaves) {
      const size_t sb = task / m.C;
      const int c = (int)(task % m.C);
      const size_t site = sb * kSites + sidx;
      const size_t s = site < a.nsites ? site : a.nsites - 1;
      double L, pr, nrm;
      int rc;
      map_sites_wave<S, FUSE, NG, false>(a, wsD, wsU, a.split_part + sb * m.C * BK * kSites, nullptr, lds_off, a.aln + s, a.ld, lane, os, L, pr,
                        rc, nrm, c, c + 1, (int)((task + nwaves) % m.C), false);
      a.split_lc[task * kSites + sidx] = L;
      a.split_lc[(ntasks + task) * kSites + sidx] = pr;
      a.split_lc[(2 * ntasks + task) * kSites + sidx] = nrm;          // weight of the pass's best class
      a.split_lc[(3 * ntasks + task) * kSites + sidx] = (double)rc;   // ... and its index
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  for (size_t sb = wave; sb < nblocks; sb += nwaves) {
    const size_t site = sb * kSites + sidx;
    const bool active = site < a.nsites && sidx < kSites;
    const size_t s = active ? site : a.nsites - 1;
    if (MODE == kModeObserved) {
      double L, pr, nrm;
      int rc;
      map_sites_wave<S, FUSE, NG, false>(a, wsD, wsU, part, cnt0, lds_off, a.aln + s, a.ld, lane, os, L, pr, rc, nrm, 0, m.C, 0, true);
      if (active) {
        if (a.logL) a.logL[s] = log(L);
        if (a.post_rate) a.post_rate[s] = pr;
        if (a.rate_class) a.rate_class[s] = rc;
        if (a.norm) a.norm[s] = nrm;
        if (a.counts)
          for (int r = 0; r < m.B * m.K; ++r) a.counts[(size_t)r * a.ldc + s] = cnt0[(size_t)r * kRow + sidx];
      }
    } else {
      // null pair q = s: replicate rep, column j; simulated-site index g_h = ((rep*2 + h)*rep_ram + j).
      // Only the minima over the two batches leave the loop (AnalysisTools.cpp:643-652) -- fewer live registers.
      double prmin = 0.0, nmin = 0.0;
      int rcmin = 0;
      for (int h = 0; h < 2; ++h) {
        const size_t rep_local = s / a.rep_ram, j = s % a.rep_ram;
        const uint8_t* gbase;
        size_t gstride;
        // the alignments were simulated by simulate_lds_kernel / simulate_blocked_kernel (cmx_null_simulate_dev) or
        // supplied by the caller: [replicate][batch][taxon][rep_ram]
        gbase = a.supplied + ((rep_local * 2 + h) * (size_t)m.T) * a.rep_ram + j;
        gstride = a.rep_ram;
        double L, pr, nrm;
        int rc;
        map_sites_wave<S, FUSE, NG, true>(a, wsD, wsU, part, h ? cnt1 : cnt0, lds_off, gbase, gstride, lane, os, L, pr, rc, nrm, 0, m.C, 0, true);
        if (h == 0) { prmin = pr; nmin = nrm; rcmin = rc; }
        else { prmin = pr < prmin ? pr : prmin; nmin = nrm < nmin ? nrm : nmin; rcmin = rc < rcmin ? rc : rcmin; }
      }
      const double stat = pair_stat_strided(a.stat_kind, a.stat_param, m.B, m.K, cnt0 + sidx, (size_t)kRow, cnt1 + sidx, (size_t)kRow, a.stat_mean);
      if (active) {
        a.null_stat[s] = stat;
        if (a.null_rcmin) a.null_rcmin[s] = rcmin;
        if (a.null_prmin) a.null_prmin[s] = prmin;
        if (a.null_nmin) a.null_nmin[s] = nmin;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last op left an operator DMA in flight
}

// LDS per mapping wave: operator stage buffers + symbol slots, plus (null mode) the simulator's node states when
// nn * 64 bytes still fit the workgroup's share of the CU (that many workgroups per CU as waves per SIMD)
int map_lds_per_wave(int S, int nn, int mode) {
  const int fixed = S == 20 ? map_lds_fixed<20>() : (S == 16 ? map_lds_fixed<16>() : map_lds_fixed<4>());
  const int share = 160 * 1024 / map_waves_per_simd(S) / kWavesPerBlock;
  const int states = (nn * map_sites_per_wave(S) + 15) / 16 * 16;
  (void)share; (void)states; (void)mode;   // the node states of a simulator inside the wave: not needed any more
  return fixed;
}

hipError_t launch_map(const MapArgs& a_in, int mode, int grid_blocks, hipStream_t stream) {
  dim3 grid(grid_blocks), block(kWave * kWavesPerBlock);
  MapArgs a = a_in;
  a.lds_per_wave = map_lds_per_wave(a.m.S, a.m.nn, mode);
  const size_t lds = (size_t)kWavesPerBlock * (size_t)a.lds_per_wave;
  const int lim = 160 * 1024 / map_waves_per_simd(a.m.S);  // dynamic LDS a workgroup may use (that many workgroups per CU)
  if ((int)lds > lim) return hipErrorInvalidValue;
#define CMX_LAUNCH(S_, MODE_, F_)                                                                             \
  do {                                                                                                        \
    /* per launch: the attribute belongs to the current device (a process may hold contexts on several) */    \
    const hipError_t ea_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&map_kernel<S_, MODE_, F_>),     \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lim);              \
    if (ea_ != hipSuccess) return ea_;                                                                        \
    hipLaunchKernelGGL((map_kernel<S_, MODE_, F_>), grid, block, lds, stream, a);                             \
  } while (0)
  // (the null instantiations of the class-fused layouts read the cherry-table walk's stream)
  if (mode == kModeNull && a.m.fuse > 1 && a.m.msched_r == nullptr) return hipErrorInvalidValue;
#define CMX_LAUNCH_MODES(S_, F_)                                            \
  do {                                                                      \
    if (mode == kModeObserved) CMX_LAUNCH(S_, kModeObserved, F_);           \
    else if (mode == kModeObservedSplit) CMX_LAUNCH(S_, kModeObservedSplit, F_); \
    else CMX_LAUNCH(S_, kModeNull, F_);                                     \
  } while (0)
  if (a.m.S == 20 && a.m.fuse == 1 && mode == kModeObservedSplit && a.split_sites == 16) {
    // 16-site wave-tasks (one site group): small alignments, four times the waves
    const hipError_t ea_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&map_kernel<20, kModeObservedSplit, 1, 1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    if (ea_ != hipSuccess) return ea_;
    hipLaunchKernelGGL((map_kernel<20, kModeObservedSplit, 1, 1>), grid, block, lds, stream, a);
  } else if (a.m.S == 20 && a.m.fuse == 1) CMX_LAUNCH_MODES(20, 1);
  else if (a.m.S == 20 && a.m.fuse == 5) CMX_LAUNCH_MODES(20, 5);
  else if (a.m.S == 16 && a.m.fuse == 4) CMX_LAUNCH_MODES(16, 4);
  else if (a.m.S == 4 && a.m.fuse == 1) CMX_LAUNCH_MODES(4, 1);
  else return hipErrorInvalidValue;
#undef CMX_LAUNCH_MODES
#undef CMX_LAUNCH
  return hipGetLastError();
}

// class-split observed mode: sums the per-class results in class order exactly as map_sites_wave's own epilogue does.
// 64 sites x 16 branch lanes per workgroup (round 4; one thread per site walked B x C dependent loads, 0.2 ms for 2 000
// sites): thread (site, w) sums the classes of the branches b = w mod 16 and leaves each branch's total in LDS, then the
// site's first thread adds the squares in branch order -- every sum runs in the order of the plain (b, k, c) loop nest.
constexpr int kFinBranchLanes = 16, kFinChunk = 128;
__global__ __launch_bounds__(64 * kFinBranchLanes) void map_finalize_kernel(const MapArgs a) {
  const DevModel& m = a.m;
  __shared__ double tots[kFinChunk][64];
  const int tx = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t s = (size_t)blockIdx.x * 64 + tx;
  const bool live = s < a.nsites;
  const size_t sc = live ? s : a.nsites - 1;
  const size_t kS = (size_t)a.split_sites;   // sites per wave-task of the mapping kernel
  const size_t sb = sc / kS, lane = sc % kS, BK = (size_t)m.B * m.K;
  const size_t nblocks = (a.nsites + kS - 1) / kS, ntasks = nblocks * (size_t)m.C;
  double Lsum = 0.0, prsum = 0.0, best = -1.0;
  int bestc = 0;
  for (int c = 0; c < m.C; ++c) {
    const size_t t = sb * m.C + c;
    Lsum += a.split_lc[t * kS + lane];
    prsum += a.split_lc[(ntasks + t) * kS + lane];
    const double bv = a.split_lc[(2 * ntasks + t) * kS + lane];
    if (bv > best) { best = bv; bestc = (int)a.split_lc[(3 * ntasks + t) * kS + lane]; }
  }
  const double* part = a.split_part + sb * m.C * BK * kS + lane;
  double nrm = 0.0;
  for (int b0 = 0; b0 < m.B; b0 += kFinChunk) {
    const int be = b0 + kFinChunk < m.B ? b0 + kFinChunk : m.B;
    for (int b = b0 + w; b < be; b += kFinBranchLanes) {
      double tot = 0.0;
      for (int k = 0; k < m.K; ++k) {
        const size_t r = (size_t)b * m.K + k;
        double v = 0.0;
        for (int c = 0; c < m.C; ++c) v += part[((size_t)c * BK + r) * kS];
        v /= Lsum;
        if (a.counts && live) a.counts[r * a.ldc + s] = v;
        tot += v;
      }
      tots[b - b0][tx] = tot;
    }
    __syncthreads();
    if (w == 0)
      for (int b = b0; b < be; ++b) nrm = __builtin_fma(tots[b - b0][tx], tots[b - b0][tx], nrm);
    __syncthreads();
  }
  if (w != 0 || !live) return;
  if (a.logL) a.logL[s] = log(Lsum);
  if (a.post_rate) a.post_rate[s] = prsum / Lsum;
  if (a.rate_class) a.rate_class[s] = bestc;
  if (a.norm) a.norm[s] = sqrt(nrm);
}

hipError_t launch_map_finalize(const MapArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(map_finalize_kernel, dim3((unsigned)((a.nsites + 63) / 64)), dim3(64 * kFinBranchLanes), 0, stream, a);
  return hipGetLastError();
}

// Rows S0 .. S0+A-1 of every transposed leaf operator: sum of the rows of the states compatible with ambiguity id a
// (what the DR likelihood's leaf initialisation does for B/Z/X/gap).  One thread per (class, leaf operator, a, x);
// a row has `rowlen` values (= S0, or S0 * fuse for the class-fused nucleotide layout).
__global__ void extend_leaf_rows_kernel(double* MAT, int C, int MC, int first_leaf, int nleaf, int S0, int rowlen, int A,
                                        int unit, int stride /* doubles between rows: leaf_row_stride */, const uint32_t* __restrict__ masks) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)C * nleaf * A * rowlen;
  if (i >= total) return;
  const int x = (int)(i % rowlen);
  const int a = (int)((i / rowlen) % A);
  const int l = (int)((i / ((size_t)rowlen * A)) % nleaf);
  const int c = (int)(i / ((size_t)rowlen * A * nleaf));
  double* M = MAT + ((size_t)c * MC + first_leaf + l) * unit;
  const uint32_t mk = masks ? masks[S0 + a] : 0xffffffffu;
  double v = 0.0;
  for (int z = 0; z < S0; ++z)
    if ((mk >> z) & 1u) v += M[(size_t)z * stride + x];
  M[(size_t)(S0 + a) * stride + x] = v;
}

hipError_t launch_extend_leaf_rows(const DevModel& m, const uint32_t* d_masks, hipStream_t stream) {
  const int A = max_ambig(m.S0), nleaf = m.T + m.K * m.T;
  const size_t total = (size_t)m.C * nleaf * A * m.S;
  hipLaunchKernelGGL(extend_leaf_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, m.MAT, m.C, m.MC,
                     m.NI + m.NI * m.K, nleaf, m.S0, m.S, A, mat_unit(m.S), leaf_row_stride(m.S), d_masks);
  return hipGetLastError();
}

// statistic of the pairs (j of data set 1, j of data set 2), j < n, and the minima the null file carries
// (AnalysisTools.cpp:728-732): counts are branch-major [B*K][ld]
__global__ void pair_diag_kernel(int kind, double param, int B, int K, const double* __restrict__ c1, size_t ld1,
                                 const double* __restrict__ c2, size_t ld2, size_t n, const int32_t* __restrict__ rc1,
                                 const int32_t* __restrict__ rc2, const double* __restrict__ pr1,
                                 const double* __restrict__ pr2, const double* __restrict__ nm1,
                                 const double* __restrict__ nm2, double* __restrict__ stat, int32_t* __restrict__ rcmin,
                                 double* __restrict__ prmin, double* __restrict__ nmin, const double* __restrict__ mv) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (stat) stat[j] = pair_stat_strided(kind, param, B, K, c1 + j, ld1, c2 + j, ld2, mv);
  if (rcmin) rcmin[j] = rc1[j] < rc2[j] ? rc1[j] : rc2[j];
  if (prmin) prmin[j] = pr1[j] < pr2[j] ? pr1[j] : pr2[j];
  if (nmin) nmin[j] = nm1[j] < nm2[j] ? nm1[j] : nm2[j];
}

hipError_t launch_pair_diag(int kind, double param, int B, int K, const double* c1, size_t ld1, const double* c2, size_t ld2,
                            size_t n, const int32_t* rc1, const int32_t* rc2, const double* pr1, const double* pr2,
                            const double* nm1, const double* nm2, double* stat, int32_t* rcmin, double* prmin, double* nmin,
                            const double* d_mean, hipStream_t stream) {
  hipLaunchKernelGGL(pair_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, kind, param, B, K, c1, ld1,
                     c2, ld2, n, rc1, rc2, pr1, pr2, nm1, nm2, stat, rcmin, prmin, nmin, d_mean);
  return hipGetLastError();
}

// Statistic of a group of sites, Statistic::getValueForGroup: the smallest pairwise value over the group
// (AbstractMinimumStatistic, CoMap/Statistics.h:121-133) or, for Compensation, the closed form of Statistics.h:267-294.
// One wave per group; group g owns sites[offsets[g] .. offsets[g+1]) (columns of the branch-major counts).
__global__ __launch_bounds__(kWave) void group_stat_kernel(int kind, double param, int B, int K, const double* __restrict__ counts,
                                                          size_t ld, const int64_t* __restrict__ offsets,
                                                          const int32_t* __restrict__ sites, double* __restrict__ out,
                                                          const double* __restrict__ mv) {
  const size_t g = blockIdx.x;
  const int lane = threadIdx.x;
  const int32_t* mem = sites + offsets[g];
  const int m = (int)(offsets[g + 1] - offsets[g]);
  if (kind == CMX_STAT_COMPENSATION) {
    double sumnorms = 0.0, sq2 = 0.0;
    for (int j = 0; j < m; ++j) {
      double q = 0.0;
      for (int b = lane; b < B; b += kWave) {
        double t = 0.0;
        for (int k = 0; k < K; ++k) t += counts[((size_t)b * K + k) * ld + mem[j]];
        q += t * t;
      }
      for (int off = 32; off; off >>= 1) q += __shfl_xor(q, off);
      sumnorms += sqrt(q);
    }
    for (int b = lane; b < B; b += kWave) {
      double t = 0.0;
      for (int j = 0; j < m; ++j)
        for (int k = 0; k < K; ++k) t += counts[((size_t)b * K + k) * ld + mem[j]];
      sq2 += t * t;
    }
    for (int off = 32; off; off >>= 1) sq2 += __shfl_xor(sq2, off);
    if (lane == 0) out[g] = 1.0 - sqrt(sq2) / sumnorms;
    return;
  }
  // pairs (i, j), j < i, in the reference's order; "val < mini" with a NaN val never wins, so NaN pairs are skipped
  double best = __builtin_inf();
  const int npairs = m * (m - 1) / 2;
  for (int p = lane; p < npairs; p += kWave) {
    int i = (int)((1.0 + sqrt(1.0 + 8.0 * (double)p)) / 2.0);
    while (i * (i - 1) / 2 > p) --i;
    while ((i + 1) * i / 2 <= p) ++i;
    const int j = p - i * (i - 1) / 2;
    const double v = pair_stat_strided(kind, param, B, K, counts + mem[i], ld, counts + mem[j], ld, mv);
    if (v < best) best = v;
  }
  for (int off = 32; off; off >>= 1) {
    const double o = __shfl_xor(best, off);
    if (o < best) best = o;
  }
  if (lane == 0) out[g] = best;
}

hipError_t launch_group_stats(int kind, double param, int B, int K, const double* d_counts, size_t ld, const int64_t* d_offsets,
                              const int32_t* d_sites, size_t ngroups, double* d_out, const double* d_mean, hipStream_t stream) {
  if (ngroups == 0) return hipSuccess;
  hipLaunchKernelGGL(group_stat_kernel, dim3((unsigned)ngroups), dim3(kWave), 0, stream, kind, param, B, K, d_counts, ld,
                     d_offsets, d_sites, d_out, d_mean);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ stand-alone simulator
// rep_ram != 0: the n sites are blocks of rep_ram (the replicates of one side of a null, side by side in the alignment);
// block r holds the global sites g0 + r * gstep ..  (one launch for all replicates of a side: round 3 launched per replicate)
__global__ void simulate_kernel(const DevModel m, uint64_t seed, uint64_t g0, size_t n, uint8_t* aln, size_t ld,
                                int32_t* classes, uint8_t* states, size_t rep_ram, uint64_t gstep) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint64_t g = rep_ram ? g0 + (uint64_t)(j / rep_ram) * gstep + (uint64_t)(j % rep_ram) : g0 + j;
  const int S = m.S0;
  const int c = draw_index(philox_uniform(seed, g, 0), m.cum_probs, m.C0);
  if (classes) classes[j] = c;
  states[(size_t)m.root * ld + j] = (uint8_t)draw_index(philox_uniform(seed, g, 1), m.cum_pi, S);
  for (int node = m.nn - 2; node >= 0; --node) {
    const int x = states[(size_t)m.parent[node] * ld + j];
    const size_t row = ((size_t)c * m.nn + node) * S + x;
    const int y = draw_guided(philox_node_uniform(seed, g, (uint32_t)node), m.CP + row * S, m.CPG + row * 32, S);
    states[(size_t)node * ld + j] = (uint8_t)y;
    const int tx = m.taxon_of[node];
    if (tx >= 0) aln[(size_t)tx * ld + j] = (uint8_t)y;
  }
}

hipError_t launch_simulate(const DevModel& m, uint64_t seed, uint64_t g0, size_t n, uint8_t* d_aln, size_t ld,
                           int32_t* d_classes, uint8_t* d_states, hipStream_t stream, size_t rep_ram, uint64_t gstep) {
  const int block = 256;
  const int grid = (int)((n + block - 1) / block);
  hipLaunchKernelGGL(simulate_kernel, dim3(grid), dim3(block), 0, stream, m, seed, g0, n, d_aln, ld, d_classes, d_states, rep_ram, gstep);
  return hipGetLastError();
}

// The null's simulator since round 2 (cmx_null_intra_dev): the SAME draws as simulate_kernel / the fused loop of
// map_kernel<S, null>, but one thread per site at full occupancy instead of 64 sites inside a mapping wave that holds
// half a SIMD's registers.  Inside the mapping kernel the simulator was 7.8 % of a wave's time, all of it dependent L2
// gathers that two waves per SIMD cannot hide; here thousands of waves hide them (cfg3, same box: fused 12.96 ms,
// mapping of supplied alignments 11.98 ms + this kernel).  Nodes are drawn level by level in groups of four (m.simg),
// four running sums per round trip of the search, exactly as the fused loop does.
// Sites s = 0 .. of one null launch: g = g0 + s; (replicate, batch) block s / blk, column s % blk of an alignment stored
// as [block][taxon][blk] -- the layout map_kernel<S, null> reads supplied alignments in.
__global__ __launch_bounds__(256) void simulate_blocked_kernel(const DevModel m, uint64_t seed, uint64_t g0, size_t s0, size_t n,
                                                               size_t blk, uint8_t* __restrict__ aln,
                                                               uint8_t* __restrict__ states /*[nn][n]*/) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const size_t s = s0 + j;
  const uint64_t g = g0 + s;
  const int S0 = m.S0;
  uint8_t* out = aln + (s / blk) * (size_t)m.T * blk + s % blk;
  const int c = draw_index(philox_uniform(seed, g, 0), m.cum_probs, m.C0);
  states[(size_t)m.root * n + j] = (uint8_t)draw_index(philox_uniform(seed, g, 1), m.cum_pi, S0);
  for (int gi = 0; gi < m.nsimg; ++gi) {
    const cmx_cint q = (cmx_cint)m.simg + gi * 16;   // [0..3] node, [4..7] its parent, [8..11] its taxon or -1
    int x[4], idx[4];
    double u[4];
    size_t row[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) x[jj] = states[(size_t)q[4 + jj] * n + j];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) u[jj] = philox_node_uniform(seed, g, (uint32_t)q[jj]);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      row[jj] = ((size_t)c * m.nn + q[jj]) * S0 + x[jj];
      idx[jj] = m.CPG[row[jj] * 32 + (int)(u[jj] * 32.0)];
    }
    bool any;
    do {
      double cv[4][4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int d = 0; d < 4; ++d) cv[jj][d] = m.CP[row[jj] * S0 + idx[jj] + d];   // the table is padded by 4 sums
      any = false;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        bool go = true;
        int adv = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          go = go && idx[jj] + d < S0 - 1 && u[jj] >= cv[jj][d];
          adv += go ? 1 : 0;
        }
        idx[jj] += adv;
        any |= adv == 4;
      }
    } while (any);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      states[(size_t)q[jj] * n + j] = (uint8_t)idx[jj];
      if (q[8 + jj] >= 0) out[(size_t)q[8 + jj] * blk] = (uint8_t)idx[jj];
    }
  }
}

// The same simulator with the tables of the node being drawn in LDS.  A draw needs one guide byte and a few running
// sums of ONE of C * S rows of its node; gathered from L2 by every thread that is ~100 bytes of traffic per draw
// (2.5e9 draws per target launch: the gather kernel above was L2-bound, 22 ms).  Here a workgroup draws 1 024 sites
// (four per thread), node by node, parents first: the node's C * S rows (12.8 KB for proteins) and guide bytes are copied
// into LDS once per workgroup, double-buffered (global -> registers while the current node is drawn, registers -> LDS
// behind a barrier), and every search runs on LDS.  Same draws, same states as simulate_kernel.
constexpr int kSimStep = 2;        // running sums per search and LDS round trip in simulate_lds_kernel
constexpr int kSimLdsChunks = 8;   // 16-byte pieces of a node's tables per thread (256 threads): up to 32 KiB per buffer
template <int SPT, int NCH /* 16-byte pieces of a node's tables per thread */, int NT /* threads */, int WAVES>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void simulate_lds_kernel(const DevModel m, uint64_t seed, uint64_t g0, size_t s0, size_t n,
                                                           size_t blk, uint8_t* __restrict__ aln, uint8_t* __restrict__ states) {
  extern __shared__ __attribute__((aligned(16))) uint8_t sim_smem[];
  const int S0 = m.S0, C0 = m.C0, tid = threadIdx.x;
  const int rowb = S0 * 8, tabb = C0 * S0 * rowb, guib = C0 * S0 * 32;       // bytes: one row of sums, all rows, all guides
  const int bufb = (tabb + guib + 15) & ~15, nch = bufb / 16;
  // piece q (16 bytes) of node `node`'s tables: the sums of class q / (S0 * rowb / 16) ... are contiguous per class in CP,
  // the guide bytes per class in CPG
  auto piece_src = [&](int node, int q) -> const cmx_i4* {
    const int off = q * 16;
    if (off < tabb) {
      const int c = off / (S0 * rowb), r = off % (S0 * rowb);
      return reinterpret_cast<const cmx_i4*>(reinterpret_cast<const uint8_t*>(m.CP + ((size_t)c * m.nn + node) * S0 * S0) + r);
    }
    const int o2 = off - tabb, c = o2 / (S0 * 32), r = o2 % (S0 * 32);
    return reinterpret_cast<const cmx_i4*>(m.CPG + ((size_t)c * m.nn + node) * S0 * 32 + r);
  };
  size_t j[SPT];
  uint64_t g[SPT];
  int cls[SPT];
  uint8_t* out[SPT];
  bool on[SPT];
#pragma unroll
  for (int k = 0; k < SPT; ++k) {
    // a thread's sites are neighbours: sites 2 q and 2 q + 1 share a Philox call for their node draws (g0 + s0 is even:
    // launch_simulate_blocked checks it), and their states are neighbouring bytes
    j[k] = (size_t)blockIdx.x * SPT * NT + (size_t)(SPT * tid + k);
    on[k] = j[k] < n;
    // (n is even: a thread's two sites are both inside or both outside; the outside ones repeat the last pair)
    const size_t jj = on[k] ? j[k] : n - 2 + (k & 1), s = s0 + jj;
    j[k] = jj;
    g[k] = g0 + s;
    out[k] = aln + (s / blk) * (size_t)m.T * blk + s % blk;
    cls[k] = draw_index(philox_uniform(seed, g[k], 0), m.cum_probs, C0);
    if (on[k]) states[(size_t)m.root * n + jj] = (uint8_t)draw_index(philox_uniform(seed, g[k], 1), m.cum_pi, S0);
  }
  // this thread's pieces of a node's tables: where they start for node 0, and the node's stride (sums: S0 * S0 doubles,
  // guides: S0 * 32 bytes) -- loop-invariant, so that the node loop adds one 24-bit product instead of dividing and
  // multiplying per node (quarter-rate 32-bit multiplies were a quarter of the loop's vector cycles)
  const uint8_t* psrc[NCH];
  unsigned pstride[NCH];
  bool pok[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int q = tid + NT * i;
    pok[i] = q < nch && q * 16 < tabb + guib;
    psrc[i] = reinterpret_cast<const uint8_t*>(piece_src(0, pok[i] ? q : 0));
    pstride[i] = (unsigned)(q * 16 < tabb ? S0 * rowb : S0 * 32);
  }
  int crow[SPT];
#pragma unroll
  for (int k = 0; k < SPT; ++k) crow[k] = cls[k] * S0;
  // tables of the first node
  const cmx_cint ord = (cmx_cint)m.simord;
  int buf = 0;
  for (int q = tid; q < nch; q += NT)
    reinterpret_cast<cmx_i4*>(sim_smem)[q] = (q * 16 < tabb + guib) ? *piece_src(ord[0], q) : cmx_i4{0, 0, 0, 0};
  __syncthreads();
  // nodes level by level (m.simord): the parent's state was written a whole level ago, not by the previous iteration
  for (int it = 0; it < m.nn - 1; ++it) {
    const int node = ord[it];
    const bool more = it + 1 < m.nn - 1;
    cmx_i4 nxt[NCH];
    if (more) {
      const int nnode = ord[it + 1];
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (pok[i]) nxt[i] = *reinterpret_cast<const cmx_i4*>(psrc[i] + __umul24((unsigned)nnode, pstride[i]));
      }
    }
    const double* T_ = reinterpret_cast<const double*>(sim_smem + (size_t)buf * bufb);
    const uint8_t* G_ = sim_smem + (size_t)buf * bufb + tabb;
    const int par = ((cmx_cint)m.parent)[node], tx = ((cmx_cint)m.taxon_of)[node];   // scalar loads: the products with n and blk stay scalar
    // the SPT searches side by side: parents' states, uniforms, guide bytes, then kSimStep running sums per search and round
    // trip (a `while (u >= cum[idx]) ++idx` per site is a chain of dependent LDS reads under a divergent branch: 36
    // branches and 110 scalar instructions per wave and draw).  Reads past a row's end stay inside the buffer (the guide
    // bytes follow the sums) and are not counted.
    int x[SPT], idx[SPT];
    double u[SPT];
    const double* cum[SPT];
    // the states of a thread's two sites are neighbouring bytes at an even address (n, j[0] even): one 16-bit access
    static_assert(SPT % 2 == 0, "sites in pairs");
    const uint8_t* sp = states + (size_t)par * n;
#pragma unroll
    for (int k = 0; k < SPT; k += 2) {
      const unsigned xx = *reinterpret_cast<const unsigned short*>(sp + (unsigned)j[k]);
      x[k] = (int)(xx & 0xffu);
      x[k + 1] = (int)(xx >> 8);
    }
#pragma unroll
    for (int k = 0; k < SPT; k += 2) {
      // (g[k] is even and g[k + 1] its neighbour -- or g[k] again, the clamped slot behind the last site of an odd n)
      uint32_t w0, w1;
      philox_words(seed, g[k] >> 1, 2u + (uint32_t)node, w0, w1);
      u[k] = (double)((g[k] & 1) ? w1 : w0) * (1.0 / 4294967296.0);
      if (k + 1 < SPT) {
        u[k + 1] = (double)((g[k + 1] & 1) ? w1 : w0) * (1.0 / 4294967296.0);
      }
    }
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
      const int row = crow[k] + x[k];
      idx[k] = G_[row * 32 + (int)(u[k] * 32.0)];
      cum[k] = T_ + __mul24(row, S0);
    }
    bool any;
    do {
      double cv[SPT][kSimStep];
#pragma unroll
      for (int k = 0; k < SPT; ++k)
#pragma unroll
        for (int d = 0; d < kSimStep; ++d) cv[k][d] = cum[k][idx[k] + d];
      any = false;
#pragma unroll
      for (int k = 0; k < SPT; ++k) {
        bool go = true;
        int adv = 0;
#pragma unroll
        for (int d = 0; d < kSimStep; ++d) {
          go = go && idx[k] + d < S0 - 1 && u[k] >= cv[k][d];
          adv += go ? 1 : 0;
        }
        idx[k] += adv;
        any |= adv == kSimStep;
      }
    } while (any);
    uint8_t* sn = states + (size_t)node * n;
#pragma unroll
    for (int k = 0; k < SPT; k += 2)
      if (on[k]) {
        *reinterpret_cast<unsigned short*>(sn + (unsigned)j[k]) = (unsigned short)(idx[k] | (idx[k + 1] << 8));
        if (tx >= 0) {
          if (((blk | (size_t)(uintptr_t)aln) & 1) == 0) {   // (an even rep_ram: the pair sits in one replicate block, at an even address)
            *reinterpret_cast<unsigned short*>(out[k] + (size_t)tx * blk) = (unsigned short)(idx[k] | (idx[k + 1] << 8));
          } else {
            out[k][(size_t)tx * blk] = (uint8_t)idx[k];
            out[k + 1][(size_t)tx * blk] = (uint8_t)idx[k + 1];
          }
        }
      }
    if (more) {
      __syncthreads();   // nobody reads the other buffer any more (it held the previous node)
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int q = tid + NT * i;
        if (q < nch && q * 16 < tabb + guib) reinterpret_cast<cmx_i4*>(sim_smem + (size_t)(buf ^ 1) * bufb)[q] = nxt[i];
      }
      __syncthreads();
      buf ^= 1;
    }
  }
}

// nsites sites with global indices g0 .. in passes of at most `chunk` (the states scratch holds nn * chunk bytes)
hipError_t launch_simulate_blocked(const DevModel& m, uint64_t seed, uint64_t g0, size_t nsites, size_t blk, uint8_t* d_aln,
                                   uint8_t* d_states, size_t chunk, hipStream_t stream) {
  // tables in LDS when a node's rows + guides fit 8 pieces per thread (every model this engine takes: S <= 20, C <= 8)
  const size_t bufb = ((size_t)m.C0 * m.S0 * (m.S0 * 8 + 32) + 15) & ~(size_t)15;
  static const bool gather = [] { const char* e = getenv("CMX_SIM_GATHER"); return e && e[0] == '1'; }();   // A/B timing
  // (DNA: 512-byte tables that sit in L1 / L2 anyway, and 2 barriers x 511 nodes: gathering is faster, cfg4 step 11.7 vs 13.0 ms)
  const bool lds = !gather && m.S0 > 4 && bufb <= (size_t)kSimLdsChunks * 256 * 16;
  for (size_t s0 = 0; s0 < nsites; s0 += chunk) {
    const size_t n = std::min(chunk, nsites - s0);
    // tables in LDS (a node's tables are copied once per 1 024 sites) from about two workgroups per CU on (cfg3's 500 000
    // sites: 0.61 ms against the gather kernel's 0.66); a workgroup's walk over the nodes with two barriers each takes
    // ~0.6 ms however few there are, so below that the gather kernel, one thread per site and no barrier, is quicker
    static const size_t lds_min = [] { const char* e = getenv("CMX_SIM_LDS_MIN"); return e ? (size_t)atoll(e) : (size_t)450000; }();   // (override: A/B timing)
    // (the LDS kernel pairs the sites 2 k, 2 k + 1 of the global numbering and stores a pair's symbols as one 16-bit word at
    // column s0 + j of its replicate block: g0, s0 and n all have to be even, not just g0 + s0)
    if (lds && n >= lds_min && (g0 & 1) == 0 && (s0 & 1) == 0 && (n & 1) == 0) {
      // 512 threads with two sites each: 56 registers = eight waves per SIMD (the kernel is bound by vector issue -- half of
      // it Philox's quarter-rate multiplies -- once enough waves hide the LDS round trips: four sites per thread at three
      // waves per SIMD 14.8 ms per target step, at five 11.6, this shape 10.2)
      if (bufb <= (size_t)2 * 512 * 16)
        hipLaunchKernelGGL((simulate_lds_kernel<2, 2, 512, 8>), dim3((unsigned)((n + 1023) / 1024)), dim3(512), 2 * bufb, stream, m, seed, g0, s0,
                           n, blk, d_aln, d_states);
      else
        hipLaunchKernelGGL((simulate_lds_kernel<4, kSimLdsChunks, 256, 4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 2 * bufb, stream, m,
                           seed, g0, s0, n, blk, d_aln, d_states);
    }
    else
      hipLaunchKernelGGL(simulate_blocked_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, m, seed, g0, s0, n, blk,
                         d_aln, d_states);
  }
  return hipGetLastError();
}

// ---- simulations.continuous = yes (CoMap/CoMap.cpp:146, 213: NonHomogeneousSequenceSimulator::enableContinuousRates).
// Every site draws its own rate from the CONTINUOUS Gamma(alpha, beta = alpha) distribution (Invariant(Gamma): rate 0 with
// probability p_inv, else the Gamma draw divided by 1 - p_inv) and every branch uses exp(Q r t) of that very rate: the
// row of the parent's state is rebuilt from the generator's eigensystem at each node (S exponentials + S^2 multiply-adds)
// -- there is no table to look up.  Same counter RNG and draw numbering as the discrete simulator (draw 0 = rate).

__host__ __device__ inline void cmx_gamma_pq(double a, double x, double* p, double* q) {
  /* regularised incomplete gamma, lower P and upper Q = 1 - P, each from the expansion that gives it without
   * cancellation: series for x < a + 1 (P), Lentz continued fraction otherwise (Q) */
  if (x <= 0.0) { *p = 0.0; *q = 1.0; return; }
  const double pre = exp(-x + a * log(x) - lgamma(a));
  if (x < a + 1.0) {
    double term = 1.0 / a, sum = term;
    for (int n = 1; n < 1000; ++n) {
      term *= x / (a + n);
      sum += term;
      if (fabs(term) < fabs(sum) * 1e-17) break;
    }
    *p = sum * pre;
    *q = 1.0 - *p;
    return;
  }
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 1000; ++i) {
    const double an = -(double)i * ((double)i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < tiny) d = tiny;
    c = b + an / c;
    if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  *q = pre * h;
  *p = 1.0 - *q;
}
/* "x is below the u-quantile": decided on the tail that carries the information (P < u, or Q > 1 - u for u > 1/2) */
__host__ __device__ inline int cmx_gamma_below(double a, double x, double u) {
  double p, q;
  cmx_gamma_pq(a, x, &p, &q);
  return u <= 0.5 ? p < u : q > 1.0 - u;
}
/* quantile of Gamma(shape a, scale 1): bracket [lo, 2 lo] by doubling / halving from 1, then 110 bisection steps
 * (deterministic, no tolerance test) */
__host__ __device__ inline double cmx_gamma_quantile(double a, double u) {
  if (u <= 0.0) return 0.0;
  double lo = 1.0, hi;
  if (cmx_gamma_below(a, lo, u)) {
    for (int i = 0; i < 1100 && cmx_gamma_below(a, 2.0 * lo, u); ++i) lo *= 2.0;
    hi = 2.0 * lo;
  } else {
    hi = lo;
    lo = 0.5 * hi;
    for (int i = 0; i < 1070 && !cmx_gamma_below(a, lo, u); ++i) { hi = lo; lo *= 0.5; }
  }
  for (int i = 0; i < 110; ++i) {
    const double mid = 0.5 * (lo + hi);
    if (cmx_gamma_below(a, mid, u)) lo = mid; else hi = mid;
  }
  return 0.5 * (lo + hi);
}

__global__ void simulate_continuous_kernel(const DevModel m, uint64_t seed, uint64_t g0, size_t n, double alpha, double p_inv,
                                           uint8_t* aln, size_t ld, double* rates, uint8_t* states) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint64_t g = g0 + j;
  const int S = m.S0;
  const double u0 = philox_uniform(seed, g, 0);
  double r = 0.0;
  if (u0 >= p_inv) r = cmx_gamma_quantile(alpha, (u0 - p_inv) / (1.0 - p_inv)) / alpha / (1.0 - p_inv);
  if (rates) rates[j] = r;
  states[(size_t)m.root * ld + j] = (uint8_t)draw_index(philox_uniform(seed, g, 1), m.cum_pi, S);
  for (int node = m.nn - 2; node >= 0; --node) {
    const int x = states[(size_t)m.parent[node] * ld + j];
    const double u = philox_node_uniform(seed, g, (uint32_t)node);
    const size_t mo = (size_t)m.model_of[node];
    const double *V = m.eigV + mo * S * S + (size_t)x * S, *Vi = m.eigVi + mo * S * S, *lam = m.eigLam + mo * S;
    const double rt = r * m.blen[node];
    double w[20];                         // S <= 20: V[x][k] exp(lambda_k r t)
    for (int k = 0; k < S; ++k) w[k] = V[k] * exp(lam[k] * rt);
    // index = #{ y < S-1 : u >= cum_y } with cum the running sum of the row P(x, .) -- the discrete simulator's rule
    int idx = 0;
    double cum = 0.0;
    for (int y = 0; y < S - 1; ++y) {
      double pxy = 0.0;
      for (int k = 0; k < S; ++k) pxy += w[k] * Vi[(size_t)k * S + y];
      cum += pxy;
      idx += (u >= cum) ? 1 : 0;
    }
    states[(size_t)node * ld + j] = (uint8_t)idx;
    const int tx = m.taxon_of[node];
    if (tx >= 0) aln[(size_t)tx * ld + j] = (uint8_t)idx;
  }
}

hipError_t launch_simulate_continuous(const DevModel& m, uint64_t seed, uint64_t g0, size_t n, double alpha, double p_inv,
                                      uint8_t* d_aln, size_t ld, double* d_rates, uint8_t* d_states, hipStream_t stream) {
  const int block = 128;
  hipLaunchKernelGGL(simulate_continuous_kernel, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, stream, m, seed, g0, n,
                     alpha, p_inv, d_aln, ld, d_rates, d_states);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ pair statistics
// prep: X[b][i] (Bp rows, zero padded) and per-site scalars s (sum of squares) and r (row sum of indicators)
//   kind 0/4: X = type-0 count - mean;  3: X = type-0 count;  1: X = per-branch total;
//   2: X = [total >= 1];  5: X = [total >= threshold], r = sum X, s = NaN flag when a total leaves [0, 10000)
__global__ void pair_prep_kernel(int kind, double param, const double* __restrict__ counts, size_t n, size_t ldc, int B,
                                 int K, double* __restrict__ X, size_t ldx, int Bp, double* __restrict__ sv,
                                 double* __restrict__ rv, const double* __restrict__ mvec /* [B] or null */, size_t blk) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // blk > 0: the sites are `blk`-site blocks side by side (replicates of the clustering null); every block gets its own
  // [Bp][ldx] operand so that a block's rows stay ldx * 8 bytes apart (not the whole batch's row length: 128 rows
  // 4 MB apart thrash the TLB and land on one L2 channel)
  if (blk) X += (i / blk) * ((size_t)Bp * ldx) - (i / blk) * blk;
  double mean = 0.0;
  if (kind == 0 || kind == 4) {   // (CorrectedCorrelation arrives as kind 0 with its mean vector in mvec)
    for (int b = 0; b < B; ++b) mean += counts[(size_t)b * K * ldc + i] - (mvec ? mvec[b] : 0.0);
    mean /= B;
  }
  double s = 0.0, r = 0.0;
  bool bad = false;
  for (int b = 0; b < B; ++b) {
    double v;
    if (kind == 0 || kind == 4) v = counts[(size_t)b * K * ldc + i] - (mvec ? mvec[b] : 0.0) - mean;
    else if (kind == 3 || kind == 9) v = counts[(size_t)b * K * ldc + i];
    else {
      double t = 0.0;
      for (int k = 0; k < K; ++k) t += counts[((size_t)b * K + k) * ldc + i];
      if (kind == 1 || kind == 7) v = t;
      else if (kind == 2) v = t >= 1.0 ? 1.0 : 0.0;
      else {
        v = t >= param ? 1.0 : 0.0;
        if (!(t >= 0.0 && t < 10000.0)) bad = true;
      }
    }
    X[(size_t)b * ldx + i] = v;
    s += v * v;
    r += v;
  }
  for (int b = B; b < Bp; ++b) X[(size_t)b * ldx + i] = 0.0;
  sv[i] = bad ? __builtin_nan("") : s;
  rv[i] = r;
}

hipError_t launch_pair_prep(int kind, double param, const double* d_counts, size_t n, size_t ldc, int B, int K,
                            double* d_X, size_t ldx, int Bp, double* d_s, double* d_r, const double* d_mvec,
                            hipStream_t stream, size_t blk) {
  const int block = 256;
  hipLaunchKernelGGL(pair_prep_kernel, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, stream, kind, param,
                     d_counts, n, ldc, B, K, d_X, ldx, Bp, d_s, d_r, d_mvec, blk);
  return hipGetLastError();
}

// the factor of a statistic that depends on one site only (fi, fj of pair_epilogue)
__device__ __forceinline__ double pair_site_factor(int kind, int B, double s) {
  if (kind == 0) return sqrt(s / (B - 1));
  if (kind == 3 || kind == 1) return sqrt(s);
  return 0.0;
}
__device__ __forceinline__ double pair_epilogue(int kind, int B, double g, double si, double sj, double ri, double rj, double fi,
                                                double fj) {
  switch (kind) {
    case 0: {
      const double cov = g / (B - 1);
      return cov / (fi * fj);
    }
    case 4: return g / (B - 1);
    case 9: return g;
    case 3: return g / (fi * fj);
    case 1: {
      double s3 = si + sj + 2.0 * g;
      if (s3 < 0.0) s3 = 0.0;
      return 1.0 - sqrt(s3) / (fi + fj);
    }
    case 2: return g;
    case 5: {
      if (si != si || sj != sj) return __builtin_nan("");
      const double np = B;
      const double cell[4] = {g, ri - g, rj - g, np - ri - rj + g};
      const double ma[4] = {ri, ri, np - ri, np - ri}, mb[4] = {rj, np - rj, rj, np - rj};
      double s = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (cell[q] > 0) s += (cell[q] / np) * log(cell[q] * np / (ma[q] * mb[q]));
      return s / log(2.7182818);
    }
  }
  return __builtin_nan("");
}

// One wave computes a 64x64 tile of G = X1^T-rows . X2-rows on v_mfma_f64_16x16x4_f64 (A[i][k]: lane = i + 16k,
// C[row = (lane>>4) + 4r][col = lane & 15]); operands come straight from L2 (X is a few MB), prefetched one k-step
// ahead; 16 MFMAs per 8 operand loads.
// Four tiles (four waves) per workgroup: single-wave workgroups made the launch dispatch-bound (262 144 of them for a
// batch of 256 matrices of 2 000 sites ran at 0.1-0.5 resident waves per SIMD, profiles/r01_cluster_null_pmc_summary.json).
__global__ __launch_bounds__(4 * kWave) void pair_gram_kernel(int kind, int B, int Bp, const double* __restrict__ X1,
                                                         const double* __restrict__ s1, const double* __restrict__ r1,
                                                         size_t n1, size_t ldx1, const double* __restrict__ X2,
                                                         const double* __restrict__ s2, const double* __restrict__ r2,
                                                         size_t n2, size_t ldx2, int intra, double* __restrict__ out,
                                                         size_t ldo, size_t zsite, size_t zout, size_t zx, size_t irow0) {
  // blockIdx.z: independent blocks of sites (clustering null: one per replicate): per-site vectors side by side
  // (zsite apart), operands zx apart
  X1 += blockIdx.z * zx; s1 += blockIdx.z * zsite; r1 += blockIdx.z * zsite;
  X2 += blockIdx.z * zx; s2 += blockIdx.z * zsite; r2 += blockIdx.z * zsite;
  out += blockIdx.z * zout;
  const int lane = threadIdx.x & 63;
  const size_t ti = blockIdx.y, tj = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t i0 = ti * 64, j0 = tj * 64;
  if (j0 >= n2) return;
  const double nanv = __builtin_nan("");
  // irow0: the rows are rows irow0 .. of the full matrix (row-block / multi-GPU processing); "below the diagonal" is
  // then j <= irow0 + i.  A tile wholly below it is skipped (intra == 2: the caller never reads it) or NaN-filled.
  if (intra == 2 && j0 + 63 < irow0 + i0) return;   // (clustering: mirrored distances; row blocks: only j > i is read)
  if (intra && irow0 == 0 && tj < ti) {  // strictly below the diagonal: NaN fill (reference loop is j > i, CoETools.cpp:680)
    for (int r = 0; r < 64; ++r) {
      const size_t i = i0 + r, j = j0 + lane;
      if (i < n1 && j < n2) out[i * ldo + j] = nanv;
    }
    return;
  }
  const int li = lane & 15, lk = lane >> 4;
  size_t ia[4], jb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    size_t i = i0 + 16 * t + li, j = j0 + 16 * t + li;
    ia[t] = i < n1 ? i : n1 - 1;
    jb[t] = j < n2 ? j : n2 - 1;
  }
  d4 acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[p][q] = (d4){0.0, 0.0, 0.0, 0.0};
  double a[4], b[4], an[4], bn[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    a[t] = X1[(size_t)lk * ldx1 + ia[t]];
    b[t] = X2[(size_t)lk * ldx2 + jb[t]];
  }
  for (int k0 = 0; k0 < Bp; k0 += 4) {
    const int kn = (k0 + 4 < Bp) ? k0 + 4 : k0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      an[t] = X1[(size_t)(kn + lk) * ldx1 + ia[t]];
      bn[t] = X2[(size_t)(kn + lk) * ldx2 + jb[t]];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[p], b[q], acc[p][q], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) { a[t] = an[t]; b[t] = bn[t]; }
  }
  // Epilogue: the per-site factors of the statistic (a square root and a division each) are computed once per row
  // and column of the tile, not once per pair -- the same operations on the same operands, so the values do not
  // change, but 64 pairs per lane no longer repeat them (they were three quarters of the kernel's instructions).
  double sjv[4], rjv[4], fj[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const size_t j = j0 + 16 * q + li;
    sjv[q] = s2[j < n2 ? j : n2 - 1];
    rjv[q] = r2[j < n2 ? j : n2 - 1];
    fj[q] = pair_site_factor(kind, B, sjv[q]);
  }
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t i = i0 + 16 * p + lk + 4 * r;
      if (i >= n1) continue;
      const double si = s1[i], ri = r1[i];
      const double fi = pair_site_factor(kind, B, si);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t j = j0 + 16 * q + li;
        if (j < n2) {
          double v = pair_epilogue(kind, B, acc[p][q][r], si, sjv[q], ri, rjv[q], fi, fj[q]);
          if (intra && j <= irow0 + i) v = nanv;
          out[i * ldo + j] = v;
        }
      }
    }
}

constexpr int kEuclidRows = 8;
// EuclidianDistance (CoMap/Distance.h:157-171): sqrt(sum_b (tot2_b - tot1_b)^2) over the per-branch totals.  Computed from
// the differences themselves, not from the Gram matrix: ||a||^2 + ||b||^2 - 2 a.b loses all digits for near-identical
// vectors.  X = the totals operand of pair_prep_kernel (kind 1), [Bp][ldx]; one thread per pair, row i broadcast.
__global__ __launch_bounds__(256) void pair_euclid_kernel(int B, const double* __restrict__ X1, size_t n1, size_t ldx1,
                                                         const double* __restrict__ X2, size_t n2, size_t ldx2, int intra,
                                                         double* __restrict__ out, size_t ldo, size_t zx,
                                                         size_t zout) {
  // one wave = kEuclidRows rows x 64 columns: the column operand is loaded once per branch and used for all rows, the
  // row operands are wave-uniform (scalar loads); per pair the arithmetic is the same chain of FMAs in branch order
  X1 += blockIdx.z * zx; X2 += blockIdx.z * zx; out += blockIdx.z * zout;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t i0 = ((size_t)blockIdx.y * 4 + wave) * kEuclidRows, j = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  if (i0 >= n1) return;
  if (intra == 2 && (size_t)blockIdx.x * 64 + 63 <= i0) return;   // whole tile in the lower triangle: left to the caller
  const size_t jc = j < n2 ? j : n2 - 1;
  double d[kEuclidRows];
#pragma unroll
  for (int r = 0; r < kEuclidRows; ++r) d[r] = 0.0;
  for (int b = 0; b < B; ++b) {
    const double t2 = X2[(size_t)b * ldx2 + jc];
    const double* row = X1 + (size_t)b * ldx1 + i0;
#pragma unroll
    for (int r = 0; r < kEuclidRows; ++r) {
      const double t = t2 - row[i0 + r < n1 ? r : 0];
      d[r] = __builtin_fma(t, t, d[r]);
    }
  }
  if (j >= n2) return;
#pragma unroll
  for (int r = 0; r < kEuclidRows; ++r) {
    const size_t i = i0 + r;
    if (i >= n1 || (intra == 2 && j <= i)) continue;
    out[i * ldo + j] = (!intra || j > i) ? sqrt(d[r]) : __builtin_nan("");
  }
}

// nblk > 1: nblk independent site blocks of n1 (= n2) sites, block z at site offset z * zsite, output at z * zout
hipError_t launch_pair_gram(int kind, int B, int Bp, const double* d_X1, const double* d_s1, const double* d_r1,
                            size_t n1, size_t ldx1, const double* d_X2, const double* d_s2, const double* d_r2,
                            size_t n2, size_t ldx2, int intra, double* d_out, size_t ldo, hipStream_t stream,
                            size_t nblk, size_t zsite, size_t zout, size_t zx, size_t irow0) {
  for (size_t z0 = 0; z0 < nblk; z0 += 65535) {     // grid.z limit
    const unsigned gz = (unsigned)std::min<size_t>(65535, nblk - z0);
    const size_t so = z0 * zsite, xo = z0 * zx;
    double* out = d_out + z0 * zout;
    if (kind == CMX_STAT_EUCLIDIAN_DISTANCE) {
      hipLaunchKernelGGL(pair_euclid_kernel, dim3((unsigned)((n2 + 63) / 64), (unsigned)((n1 + 4 * kEuclidRows - 1) / (4 * kEuclidRows)), gz), dim3(256), 0, stream, B,
                         d_X1 + xo, n1, ldx1, d_X2 + xo, n2, ldx2, intra, out, ldo, zx, zout);
    } else {
      dim3 grid((unsigned)((n2 + 255) / 256), (unsigned)((n1 + 63) / 64), gz);
      hipLaunchKernelGGL(pair_gram_kernel, grid, dim3(4 * kWave), 0, stream, kind, B, Bp, d_X1 + xo, d_s1 + so, d_r1 + so, n1,
                         ldx1, d_X2 + xo, d_s2 + so, d_r2 + so, n2, ldx2, intra, out, ldo, zsite, zout, zx, irow0);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// ------------------------------------------------------------------------------------------------ p-values
__global__ void max_reduce_kernel(const double* __restrict__ x, size_t n, double* out) {
  __shared__ double sm[256];
  double v = -__builtin_inf();
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) v = x[i] > v ? x[i] : v;
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = sm[threadIdx.x + s] > sm[threadIdx.x] ? sm[threadIdx.x + s] : sm[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sm[0];
}

hipError_t launch_max_reduce(const double* d_x, size_t n, double* d_out, hipStream_t stream) {
  hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(256), 0, stream, d_x, n, d_out);
  return hipGetLastError();
}

// Domain(0, maxnorm, n)::getIndex (CoMap/Domain.cpp:46-59, 113-122) with the reference's operation order and no
// fused multiply-add, so that class indices are bit-exact.  -1 == OutOfRangeException.
__device__ __forceinline__ int domain_index(double maxi, int n, double x) {
  const double mini = 0.0;
  const double w = __ddiv_rn(__dsub_rn(maxi, mini), (double)n);
  if (x < mini || x >= __dadd_rn(mini, __dmul_rn((double)n, w))) return -1;
  for (int i = 1; i < n + 1; ++i)
    if (x < __dadd_rn(mini, __dmul_rn((double)i, w))) return i - 1;
  return -1;
}

__global__ void null_classify_kernel(const double* __restrict__ stat, const double* __restrict__ nmin, size_t nnull,
                                     const double* __restrict__ maxnorm, int nclasses, uint32_t* __restrict__ cls,
                                     uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[65];
  for (int i = threadIdx.x; i <= nclasses; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nnull; q += (size_t)gridDim.x * blockDim.x) {
    const int c = (stat[q] != stat[q]) ? -1 : domain_index(*maxnorm, nclasses, nmin[q]);
    const uint32_t cc = c < 0 ? (uint32_t)nclasses : (uint32_t)c;
    cls[q] = cc;
    atomicAdd(&lh[cc], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= nclasses; i += blockDim.x)
    if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

hipError_t launch_null_classify(const double* d_stat, const double* d_nmin, size_t nnull, const double* d_maxnorm,
                                int nclasses, uint32_t* d_cls, uint32_t* d_hist, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(d_hist, 0, sizeof(uint32_t) * (nclasses + 1), stream);
  if (e != hipSuccess) return e;
  if (nnull == 0) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<size_t>((nnull + 255) / 256, 1024);
  hipLaunchKernelGGL(null_classify_kernel, dim3(blocks), dim3(256), 0, stream, d_stat, d_nmin, nnull, d_maxnorm,
                     nclasses, d_cls, d_hist);
  return hipGetLastError();
}

// stable LSD: sort by statistic, then by class -> classes contiguous, each ascending (CoETools.cpp:650-652)
hipError_t sort_null_by_class(void* d_tmp, size_t& tmp_bytes, double* d_stat_in, double* d_stat_tmp, uint32_t* d_cls_in,
                              uint32_t* d_cls_tmp, size_t n, hipStream_t stream) {
  if (d_tmp == nullptr) {
    size_t b1 = 0, b2 = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, b1, d_stat_in, d_stat_tmp, d_cls_in, d_cls_tmp, n, 0, 64, stream);
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(nullptr, b2, d_cls_tmp, d_cls_in, d_stat_tmp, d_stat_in, n, 0, 8, stream);
    tmp_bytes = b1 > b2 ? b1 : b2;
    return e;
  }
  hipError_t e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_stat_in, d_stat_tmp, d_cls_in, d_cls_tmp, n, 0, 64, stream);
  if (e != hipSuccess) return e;
  return rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_cls_tmp, d_cls_in, d_stat_tmp, d_stat_in, n, 0, 8, stream);
}

// p = (nsim - #{null < stat} + 1) / (nsim + 1), strict '<' (CoETools.cpp:712-717); the reference scans linearly,
// the sorted class makes it a lower bound.  The bin of the statistic (equal-width bins in its value, one per eight sorted
// values of the class) bounds the search to the handful of values inside that bin: two loads for the bin and about four
// for the search, where a binary search over a class of 10^6 values sends twenty divergent loads per pair through the
// vector cache.  null_bin is monotone in v, in the table's construction and here alike, so the count is the same.
__device__ __forceinline__ uint32_t null_bin(double v, const NullClass& c) {
  const double x = (v - c.lo) * c.scale;
  if (!(x > 0.0)) return 0u;             // below the first bin, or not a number
  return x >= (double)(c.nb - 1) ? c.nb - 1 : (uint32_t)x;
}
// the null values of the pair's norm class that are < st (CoETools.cpp:712-717) and the class's size; false: the smaller
// norm lies outside the Domain (NA, CoETools.cpp:718-720)
__device__ __forceinline__ bool null_below(const NullTable& nt, double ni, double nj, double st, uint32_t* below, uint32_t* ns) {
  const double mn = ni < nj ? ni : nj;
  const int cat = domain_index(*nt.maxnorm, nt.nclasses, mn);
  if (cat < 0) return false;
  const NullClass c = nt.cls[cat];
  uint32_t l2 = c.off, h2 = c.off + c.ns;
  if (c.nb > 1) {
    const uint32_t* bs = nt.bins + c.boff + null_bin(st, c);
    l2 = bs[0];
    h2 = bs[1];
  }
  while (l2 < h2) {
    const uint32_t mid = (l2 + h2) >> 1;
    if (nt.sorted[mid] < st) l2 = mid + 1; else h2 = mid;
  }
  *below = l2 - c.off;
  *ns = c.ns;
  return true;
}
__device__ __forceinline__ void null_pvalue(const NullTable& nt, double ni, double nj, double st, double* pvalue, int32_t* nsim) {
  uint32_t below, ns;
  if (!null_below(nt, ni, nj, st, &below, &ns)) { *pvalue = __builtin_nan(""); *nsim = 0; return; }
  *pvalue = (double)(ns - below + 1) / (double)(ns + 1);
  *nsim = (int32_t)ns;
}
__global__ void pvalue_kernel(const double* __restrict__ stat, size_t ldo, const double* __restrict__ norms, size_t n, const NullTable nt,
                              double* __restrict__ pvalue, int32_t* __restrict__ nsim, size_t irow0) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = irow0 + blockIdx.y;          // row of the full matrix; the arrays hold rows irow0 ..
  if (j >= n) return;
  const size_t o = (size_t)blockIdx.y * ldo + j;
  if (j <= i) { pvalue[o] = __builtin_nan(""); nsim[o] = 0; return; }
  null_pvalue(nt, norms[i], norms[j], stat[o], pvalue + o, nsim + o);
}

hipError_t launch_pvalues(const double* d_stat, size_t ldo, const double* d_norms, size_t n, const NullTable& nt, double* d_pvalue,
                          int32_t* d_nsim, hipStream_t stream, size_t irow0, size_t nrows) {
  if (nrows == 0) nrows = n;
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)nrows);
  hipLaunchKernelGGL(pvalue_kernel, grid, dim3(256), 0, stream, d_stat, ldo, d_norms, n, nt, d_pvalue, d_nsim, irow0);
  return hipGetLastError();
}

// class offsets and bin ranges: the bins span the class's values between its 1/64 and 63/64 quantiles (the tails fall
// into the first and last bin), so that a few extreme values do not stretch them
__global__ void null_classes_kernel(const double* __restrict__ sorted, const uint32_t* __restrict__ hist, int nclasses,
                                    NullClass* __restrict__ cls) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t o = 0;
  for (int k = 0; k < nclasses; ++k) {
    NullClass c;
    c.off = o; c.ns = hist[k]; c.nb = 1; c.boff = (o >> kNullBinShift) + 2 * (uint32_t)k; c.lo = 0.0; c.scale = 0.0;
    if (c.ns >= 64) {
      const double qlo = sorted[o + (c.ns >> 6)], qhi = sorted[o + c.ns - 1 - (c.ns >> 6)];
      const uint32_t nb = c.ns >> kNullBinShift;
      const double scale = (double)nb / (qhi - qlo);
      if (qhi > qlo && scale > 0.0 && scale < 1.0e300 && qlo > -1.0e300) { c.nb = nb; c.lo = qlo; c.scale = scale; }
    }
    cls[k] = c;
    o += c.ns;
  }
}
// bins[b] = first position of the class whose value's bin is >= b, for b = 0 .. nb (bins[nb] = the class's end): one
// thread per bin searches the class (a thread per sorted value filling the bins up to its own is cheaper on a smooth
// null and serial on one with gaps)
__global__ void null_bins_kernel(const double* __restrict__ sorted, const NullClass* __restrict__ cls, uint32_t* __restrict__ bins) {
  const NullClass c = cls[blockIdx.y];
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (c.nb <= 1 || b > c.nb) return;
  uint32_t lo = c.off, hi = c.off + c.ns;
  if (b == 0) hi = lo;
  if (b == c.nb) lo = hi;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (null_bin(sorted[mid], c) < b) lo = mid + 1; else hi = mid;
  }
  bins[c.boff + b] = lo;
}
hipError_t launch_null_index(const double* d_sorted, const uint32_t* d_hist, int nclasses, size_t nnull, NullClass* d_cls,
                             uint32_t* d_bins, hipStream_t stream) {
  hipLaunchKernelGGL(null_classes_kernel, dim3(1), dim3(64), 0, stream, d_sorted, d_hist, nclasses, d_cls);
  if (nnull)   // (the classes' sizes live on the device: the grid covers the largest number of bins any class can have)
    hipLaunchKernelGGL(null_bins_kernel, dim3((unsigned)((nnull >> kNullBinShift) / 256 + 1), (unsigned)nclasses), dim3(256), 0, stream,
                       d_sorted, d_cls, d_bins);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ compacted pair rows
// CoETools.cpp:672-724 as two passes of kPairRowSegs waves per row i: count the pairs (i, j > i) that pass the filters,
// exclusive scan of the counts (rocPRIM), then write the rows at their final position -- the reference's (i, j) order.
__device__ __forceinline__ bool pair_passes(const cmx_pair_filters& f, int ci, double ri, int cj, double rj, double st) {
  if (cj < f.min_rate_class || rj < f.min_rate) return false;
  if (f.max_rate_class_diff >= 0 && abs(cj - ci) > f.max_rate_class_diff) return false;
  if (f.max_rate_diff >= 0.0 && fabs(rj - ri) > f.max_rate_diff) return false;
  return !(fabs(st) < f.min_statistic);
}
static_assert(sizeof(cmx_pair_row) == 48 && offsetof(cmx_pair_row, stat) == 8 && offsetof(cmx_pair_row, rc_min) == 16 &&
                  offsetof(cmx_pair_row, nsim) == 20 && offsetof(cmx_pair_row, pr_min) == 24 &&
                  offsetof(cmx_pair_row, n_min) == 32 && offsetof(cmx_pair_row, pvalue) == 40,
              "pair_rows_kernel packs a row as three 16-byte words");
template <bool WRITE>
__global__ __launch_bounds__(64) void pair_rows_kernel(const double* __restrict__ stat, size_t ldo,
                                                       const double* __restrict__ pvalue, const int32_t* __restrict__ nsim,
                                                       size_t n, const int32_t* __restrict__ rc, const double* __restrict__ pr,
                                                       const double* __restrict__ norm, cmx_pair_filters f,
                                                       unsigned long long* __restrict__ rowcount /* counts, then offsets */,
                                                       cmx_pair_row* __restrict__ rows, size_t capacity, size_t irow0,
                                                       const unsigned long long* __restrict__ base, const NullTable nt) {
  // stat / pvalue / nsim hold rows irow0 .. of the full matrix (local row = blockIdx.x / kPairRowSegs); rows are appended
  // after *base.  A row's columns i + 1 .. n - 1 are cut into kPairRowSegs runs of whole 64-column steps, one wave each
  // (one wave per row leaves a 1677-row block of the 25000-site job at 6 waves per CU, all waiting on their loads); the
  // counting pass counts per run, so the runs of a row, and the rows, still land in (i, j) order.
  const size_t il = blockIdx.x / kPairRowSegs, i = irow0 + il;
  const int lane = threadIdx.x;
  const int ci = rc[i];
  const double ri = pr[i];
  const size_t seg = ((n - i - 1 + kPairRowSegs - 1) / kPairRowSegs + 63) / 64 * 64;
  const size_t jb = i + 1 + (blockIdx.x % kPairRowSegs) * seg, je = jb + seg < n ? jb + seg : n;
  unsigned long long run = WRITE ? rowcount[blockIdx.x] + (base ? *base : 0ull) : 0ull;
  const bool row_ok = !(ci < f.min_rate_class || ri < f.min_rate);
  // the passing pairs of one 64-column step are packed in LDS and leave as one contiguous run of 16-byte stores (a
  // 48-byte row per lane is a 48-byte-strided store otherwise); a rows buffer that is not 16-byte aligned gets the rows
  // one per lane
  __shared__ cmx_i4 stage[WRITE ? 64 * 3 : 1];
  const bool packed = ((uintptr_t)rows & 15) == 0;
  if (row_ok)
    for (size_t j0 = jb; j0 < je; j0 += 64) {
      const size_t j = j0 + lane;
      bool ok = false;
      double st = 0.0;
      if (j < je) {
        st = stat[il * ldo + j];
        ok = pair_passes(f, ci, ri, rc[j], pr[j], st);
      }
      const unsigned long long m = __ballot(ok);
      if (WRITE && m && run < capacity) {
        const int slot = __popcll(m & ((1ull << lane) - 1ull));
        if (ok) {
          cmx_pair_row r;
          r.i = (int32_t)i; r.j = (int32_t)j; r.stat = st;
          r.rc_min = ci < rc[j] ? ci : rc[j];
          r.pr_min = ri < pr[j] ? ri : pr[j];
          r.n_min = norm[i] < norm[j] ? norm[i] : norm[j];
          r.pvalue = pvalue ? pvalue[il * ldo + j] : __builtin_nan("");
          r.nsim = nsim ? nsim[il * ldo + j] : 0;
          if (nt.sorted) null_pvalue(nt, norm[i], norm[j], st, &r.pvalue, &r.nsim);   // only for the pairs that are written
          if (packed) {
            const long long s8 = __double_as_longlong(r.stat), p8 = __double_as_longlong(r.pr_min),
                            n8 = __double_as_longlong(r.n_min), v8 = __double_as_longlong(r.pvalue);
            stage[3 * slot + 0] = cmx_i4{r.i, r.j, (int)s8, (int)(s8 >> 32)};
            stage[3 * slot + 1] = cmx_i4{r.rc_min, r.nsim, (int)p8, (int)(p8 >> 32)};
            stage[3 * slot + 2] = cmx_i4{(int)n8, (int)(n8 >> 32), (int)v8, (int)(v8 >> 32)};
          } else if (run + slot < capacity) {
            rows[run + slot] = r;
          }
        }
        if (packed) {
          const unsigned long long room = capacity - run, cnt = (unsigned long long)__popcll(m);
          const int nq = 3 * (int)(cnt < room ? cnt : room);
          cmx_i4* dst = reinterpret_cast<cmx_i4*>(rows + run);
#pragma unroll
          for (int q = lane; q < 192; q += 64)
            if (q < nq) dst[q] = stage[q];
        }
      }
      run += __popcll(m);
    }
  if (!WRITE && lane == 0) rowcount[blockIdx.x] = run;
}

__global__ void pair_rows_total_kernel(const unsigned long long* __restrict__ offsets, const unsigned long long* __restrict__ last_count,
                                       size_t n, unsigned long long* __restrict__ total, const unsigned long long* __restrict__ base) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *total = (base ? *base : 0ull) + offsets[n - 1] + *last_count;
}

// nrows rows irow0 .. irow0 + nrows - 1 of an n-column matrix (nrows == 0: the whole matrix).  d_base (device, may be
// null): number of rows already in d_rows -- this block's rows are appended behind them and *d_count becomes the new
// total, so that consecutive row blocks fill one array in the reference's (i, j) order.
hipError_t launch_pair_rows(const double* d_stat, size_t ldo, const double* d_pvalue, const int32_t* d_nsim, size_t n,
                            const int32_t* d_rc, const double* d_pr, const double* d_norm, const cmx_pair_filters& f,
                            unsigned long long* d_rowcount /*[nrows * kPairRowSegs + 1]*/, void* d_tmp, size_t& tmp_bytes, cmx_pair_row* d_rows,
                            size_t capacity, unsigned long long* d_count, hipStream_t stream, size_t irow0, size_t nrows,
                            const unsigned long long* d_base, const NullTable* d_inline_null) {
  // d_inline_null: the write pass looks the p-values up itself (no dense p-value / Nsim block, no pvalue_kernel)
  const NullTable nt = d_inline_null ? *d_inline_null : NullTable{nullptr, nullptr, nullptr, nullptr, 0};
  if (nrows == 0) nrows = n;
  const size_t nruns = nrows * kPairRowSegs;
  if (d_tmp == nullptr) {
    return rocprim::exclusive_scan(nullptr, tmp_bytes, d_rowcount, d_rowcount, 0ull, nruns, rocprim::plus<unsigned long long>(), stream);
  }
  hipLaunchKernelGGL((pair_rows_kernel<false>), dim3((unsigned)nruns), dim3(64), 0, stream, d_stat, ldo, d_pvalue, d_nsim, n, d_rc,
                     d_pr, d_norm, f, d_rowcount, d_rows, capacity, irow0, d_base, nt);
  // keep the last row's count (the scan overwrites it) to form the total
  hipError_t e = hipMemcpyAsync(d_rowcount + nruns, d_rowcount + nruns - 1, sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream);
  if (e != hipSuccess) return e;
  e = rocprim::exclusive_scan(d_tmp, tmp_bytes, d_rowcount, d_rowcount, 0ull, nruns, rocprim::plus<unsigned long long>(), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((pair_rows_kernel<true>), dim3((unsigned)nruns), dim3(64), 0, stream, d_stat, ldo, d_pvalue, d_nsim, n, d_rc,
                     d_pr, d_norm, f, d_rowcount, d_rows, capacity, irow0, d_base, nt);
  // (after the writes: d_count may be the very word d_base points to)
  hipLaunchKernelGGL(pair_rows_total_kernel, dim3(1), dim3(64), 0, stream, d_rowcount, d_rowcount + nruns, nruns, d_count, d_base);
  return hipGetLastError();
}

// ---- compact pair records (round 4): the unfiltered pair loop of CoETools.cpp:672-724 as 16 bytes per pair -- the statistic,
// the number of null values below it and the size of its null class -- at the pair's position in (i, j) order.  Without
// filters that position is arithmetic (no counting pass, no scan), and everything else a statistics.txt row holds (i, j,
// RCmin, PRmin, Nmin, the p-value's quotient) is a function of per-site arrays the host already has:
// cmx_expand_compact_rows rebuilds the 48-byte rows bit for bit.  A third of the bytes cross PCIe.
static_assert(sizeof(cmx_pair_compact) == 16, "one 16-byte store per pair");
__global__ void pair_compact_kernel(const double* __restrict__ stat, size_t ldo, size_t n, const double* __restrict__ norm, const NullTable nt,
                                    cmx_pair_compact* __restrict__ out, size_t capacity, size_t irow0, size_t row_begin) {
  const size_t i = irow0 + blockIdx.y, j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j <= i || j >= n) return;
  // pairs of the rows [row_begin, i) come first
  const size_t at = (i - row_begin) * (n - 1) - (i * (i - 1) - row_begin * (row_begin - 1)) / 2 + (j - i - 1);
  if (at >= capacity) return;
  cmx_pair_compact r;
  r.stat = stat[(size_t)blockIdx.y * ldo + j];
  r.below = 0xffffffffu;   // NA (or no null given): PValue NaN, Nsim 0
  r.nsim = 0;
  if (nt.sorted) {
    uint32_t below, ns;
    if (null_below(nt, norm[i], norm[j], r.stat, &below, &ns)) { r.below = below; r.nsim = ns; }
  }
  out[at] = r;
}
hipError_t launch_pair_compact(const double* d_stat, size_t ldo, size_t n, const double* d_norm, const NullTable* nt, cmx_pair_compact* d_out,
                               size_t capacity, hipStream_t stream, size_t irow0, size_t nrows, size_t row_begin) {
  const NullTable t = nt ? *nt : NullTable{nullptr, nullptr, nullptr, nullptr, 0};
  hipLaunchKernelGGL(pair_compact_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nrows), dim3(256), 0, stream, d_stat, ldo, n, d_norm, t,
                     d_out, capacity, irow0, row_begin);
  return hipGetLastError();
}

// ---- rows of the inter-gene statistics file: CoETools::computeInterStats' pair loop (CoMap/CoETools.cpp:786-828) on the
// device, the same two passes as pair_rows_kernel.  Row i of data set 1 against columns [0, n2) of data set 2, or against
// column i alone (independant_comparisons, :796-797).  Filters: min rate class / rate per data set, max differences per
// pair, |stat| >= statistic.min.  Nmin = min(norm1[i], norm2[j]); with f.reference_norm_quirk the reference's own column,
// min(norms1[i], norms2[i]) (:803 reads norms2[i]; beyond the second data set's end the index is clamped where the
// reference reads out of bounds).
__device__ __forceinline__ bool inter_passes(const cmx_inter_filters& f, int ci, double ri, int cj, double rj, double st) {
  if (cj < f.min_rate_class2 || rj < f.min_rate2) return false;
  if (f.max_rate_class_diff >= 0 && abs(cj - ci) > f.max_rate_class_diff) return false;
  if (f.max_rate_diff >= 0.0 && fabs(rj - ri) > f.max_rate_diff) return false;
  return !(fabs(st) < f.min_statistic);
}
template <bool WRITE>
__global__ __launch_bounds__(64) void inter_rows_kernel(const double* __restrict__ stat, size_t ldo, size_t n2,
                                                        const int32_t* __restrict__ rc1, const double* __restrict__ pr1,
                                                        const double* __restrict__ nm1, const int32_t* __restrict__ rc2,
                                                        const double* __restrict__ pr2, const double* __restrict__ nm2,
                                                        cmx_inter_filters f, unsigned long long* __restrict__ rowcount,
                                                        cmx_pair_row* __restrict__ rows, size_t capacity, size_t irow0,
                                                        const unsigned long long* __restrict__ base) {
  const size_t il = blockIdx.x, i = irow0 + il;
  const int lane = threadIdx.x;
  const int ci = rc1[i];
  const double ri = pr1[i];
  unsigned long long run = WRITE ? rowcount[il] + (base ? *base : 0ull) : 0ull;
  const bool diag = f.independent_comparisons != 0;
  const size_t jb = diag ? i : 0, je = diag ? i + 1 : n2;
  if (!(ci < f.min_rate_class1 || ri < f.min_rate1))
    for (size_t j0 = jb; j0 < je; j0 += 64) {
      const size_t j = j0 + lane;
      bool ok = false;
      double st = 0.0;
      if (j < je) {
        st = diag ? stat[il] : stat[il * ldo + j];
        ok = inter_passes(f, ci, ri, rc2[j], pr2[j], st);
      }
      const unsigned long long m = __ballot(ok);
      if (WRITE && ok) {
        const unsigned long long pos = run + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < capacity) {
          cmx_pair_row r;
          r.i = (int32_t)i; r.j = (int32_t)j; r.stat = st;
          r.rc_min = ci < rc2[j] ? ci : rc2[j];
          r.pr_min = ri < pr2[j] ? ri : pr2[j];
          const double nj = nm2[f.reference_norm_quirk ? (i < n2 ? i : n2 - 1) : j];
          r.n_min = nm1[i] < nj ? nm1[i] : nj;
          r.pvalue = __builtin_nan("");
          r.nsim = 0;
          rows[pos] = r;
        }
      }
      run += __popcll(m);
    }
  if (!WRITE && lane == 0) rowcount[il] = run;
}

// rows irow0 .. irow0 + nrows - 1 of data set 1; d_stat: [nrows][ldo] (or [nrows] for independant comparisons); d_base
// / d_count as in launch_pair_rows
hipError_t launch_inter_rows(const double* d_stat, size_t ldo, size_t n2, const int32_t* d_rc1, const double* d_pr1, const double* d_nm1,
                             const int32_t* d_rc2, const double* d_pr2, const double* d_nm2, const cmx_inter_filters& f,
                             unsigned long long* d_rowcount, void* d_tmp, size_t& tmp_bytes, cmx_pair_row* d_rows, size_t capacity,
                             unsigned long long* d_count, hipStream_t stream, size_t irow0, size_t nrows,
                             const unsigned long long* d_base) {
  if (d_tmp == nullptr)
    return rocprim::exclusive_scan(nullptr, tmp_bytes, d_rowcount, d_rowcount, 0ull, nrows, rocprim::plus<unsigned long long>(), stream);
  hipLaunchKernelGGL((inter_rows_kernel<false>), dim3((unsigned)nrows), dim3(64), 0, stream, d_stat, ldo, n2, d_rc1, d_pr1, d_nm1, d_rc2,
                     d_pr2, d_nm2, f, d_rowcount, d_rows, capacity, irow0, d_base);
  hipError_t e = hipMemcpyAsync(d_rowcount + nrows, d_rowcount + nrows - 1, sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream);
  if (e != hipSuccess) return e;
  e = rocprim::exclusive_scan(d_tmp, tmp_bytes, d_rowcount, d_rowcount, 0ull, nrows, rocprim::plus<unsigned long long>(), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((inter_rows_kernel<true>), dim3((unsigned)nrows), dim3(64), 0, stream, d_stat, ldo, n2, d_rc1, d_pr1, d_nm1, d_rc2,
                     d_pr2, d_nm2, f, d_rowcount, d_rows, capacity, irow0, d_base);
  hipLaunchKernelGGL(pair_rows_total_kernel, dim3(1), dim3(64), 0, stream, d_rowcount, d_rowcount + nrows, nrows, d_count, d_base);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ Mica column MI
// SiteTools::mutualInformation / jointEntropy / entropy (resolveUnknowns = true), natural log (Mica.cpp:93-95).
// LDS-table kernel (the general path: any ambiguity code, fractional counts in fp64): one wave per (column i, 16 columns
// j); the 16 joint tables of A x A doubles live in LDS ([cell][pair slot]), four lanes share a pair and spread the
// (fractional) unit counts of their quarter of the taxa into its table with LDS atomics.  The MFMA path below serves the
// columns without partial ambiguity codes; this kernel then only sees the pairs that involve a flagged column (and
// returns after one load when no column is flagged).
template <int A>
__global__ __launch_bounds__(64) void mi_columns_kernel(int T, const uint32_t* __restrict__ masks,
                                                        const uint8_t* __restrict__ aln1, size_t n1, size_t ld1,
                                                        const uint8_t* __restrict__ aln2, size_t n2, size_t ld2,
                                                        int intra, double* __restrict__ mi, double* __restrict__ hj,
                                                        size_t ldo, const uint8_t* __restrict__ flag1,
                                                        const uint8_t* __restrict__ flag2, const int* __restrict__ anyflag) {
  // with flags (MFMA path active) this kernel only serves pairs that involve a column with ambiguous symbols; when no
  // column at all is flagged every workgroup leaves after one load
  if (anyflag && *anyflag == 0) return;
  // LDS: joint table [A*A][16 lanes] fp64 per quarter-wave = A*A*16*8 B (51 KB for A = 20): 16 pairs per block pass
  extern __shared__ double tab[];
  const int lane = threadIdx.x;
  const int sub = lane & 15;       // pair slot
  const int part = lane >> 4;      // 4 lanes cooperate on one pair: taxa are split in 4 strides
  const size_t nbj = (n2 + 15) / 16, ntiles = nbj * n1;
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t i = tile / nbj;
    const size_t jb = tile % nbj;
    if (flag1) {
      bool any = flag1[i] != 0;
      for (size_t q = jb * 16; !any && q < jb * 16 + 16 && q < n2; ++q) any = flag2[q] != 0;
      if (!any) continue;
    }
    const size_t j = jb * 16 + sub;
    const bool valid = j < n2 && (!intra || j > i);
    const size_t jj = j < n2 ? j : n2 - 1;
    __syncthreads();   // the previous tile's readers are done with the table
    for (int q = lane; q < A * A * 16; q += 64) tab[q] = 0.0;
    __syncthreads();
    for (int t = part; t < T; t += 4) {
      const unsigned c1 = aln1[(size_t)t * ld1 + i], c2 = aln2[(size_t)t * ld2 + jj];
      if (c1 < (unsigned)A && c2 < (unsigned)A) {
        atomicAdd(&tab[(c1 * A + c2) * 16 + sub], 1.0);
      } else {
        const uint32_t m1 = c1 < (unsigned)A ? (1u << c1) : masks[c1], m2 = c2 < (unsigned)A ? (1u << c2) : masks[c2];
        const double w = 1.0 / (double)(__popc(m1) * __popc(m2));
        for (int a = 0; a < A; ++a)
          if ((m1 >> a) & 1u)
            for (int b = 0; b < A; ++b)
              if ((m2 >> b) & 1u) atomicAdd(&tab[(a * A + b) * 16 + sub], w);
      }
    }
    __syncthreads();
    if (part == 0) {
      double p1[A], p2[A];
#pragma unroll
      for (int a = 0; a < A; ++a) { p1[a] = 0.0; p2[a] = 0.0; }
#pragma unroll
      for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < A; ++b) {
          const double v = tab[(a * A + b) * 16 + sub];
          p1[a] += v;
          p2[b] += v;
        }
      double s = 0.0, h = 0.0;
#pragma unroll
      for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < A; ++b) {
          const double pab = tab[(a * A + b) * 16 + sub] / T;
          if (pab > 0.0) {
            s += pab * log(pab / ((p1[a] / T) * (p2[b] / T)));
            h -= pab * log(pab);
          }
        }
      if (j < n2 && (!flag1 || flag1[i] || flag2[j])) {
        mi[i * ldo + j] = valid ? s : __builtin_nan("");
        hj[i * ldo + j] = valid ? h : __builtin_nan("");
      }
    }
  }
}

// MI / joint entropy of LISTED column pairs (idx1[p], idx2[p]): the building block of Mica's null distributions
// (non-parametric bootstrap = random pairs of the data, Mica.cpp:399-468; parametric bootstrap = column j of one
// simulated alignment against column j of another, Mica.cpp:469-548).  Same table arithmetic as mi_columns_kernel.
template <int A>
__global__ __launch_bounds__(64) void mi_pairs_kernel(int T, const uint32_t* __restrict__ masks,
                                                      const uint8_t* __restrict__ aln1, size_t ld1,
                                                      const uint8_t* __restrict__ aln2, size_t ld2,
                                                      const int64_t* __restrict__ idx1, const int64_t* __restrict__ idx2,
                                                      size_t npairs, double* __restrict__ mi, double* __restrict__ hj) {
  extern __shared__ double tab[];
  const int lane = threadIdx.x, sub = lane & 15, part = lane >> 4;
  const size_t p = (size_t)blockIdx.x * 16 + sub;
  const size_t pp = p < npairs ? p : npairs - 1;
  const size_t i = (size_t)idx1[pp], j = (size_t)idx2[pp];
  for (int q = lane; q < A * A * 16; q += 64) tab[q] = 0.0;
  __syncthreads();
  for (int t = part; t < T; t += 4) {
    const unsigned c1 = aln1[(size_t)t * ld1 + i], c2 = aln2[(size_t)t * ld2 + j];
    if (c1 < (unsigned)A && c2 < (unsigned)A) {
      atomicAdd(&tab[(c1 * A + c2) * 16 + sub], 1.0);
    } else {
      const uint32_t m1 = c1 < (unsigned)A ? (1u << c1) : masks[c1], m2 = c2 < (unsigned)A ? (1u << c2) : masks[c2];
      const double w = 1.0 / (double)(__popc(m1) * __popc(m2));
      for (int a = 0; a < A; ++a)
        if ((m1 >> a) & 1u)
          for (int b = 0; b < A; ++b)
            if ((m2 >> b) & 1u) atomicAdd(&tab[(a * A + b) * 16 + sub], w);
    }
  }
  __syncthreads();
  if (part == 0 && p < npairs) {
    double p1[A], p2[A];
#pragma unroll
    for (int a = 0; a < A; ++a) { p1[a] = 0.0; p2[a] = 0.0; }
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int b = 0; b < A; ++b) {
        const double v = tab[(a * A + b) * 16 + sub];
        p1[a] += v;
        p2[b] += v;
      }
    double s = 0.0, h = 0.0;
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int b = 0; b < A; ++b) {
        const double pab = tab[(a * A + b) * 16 + sub] / T;
        if (pab > 0.0) {
          s += pab * log(pab / ((p1[a] / T) * (p2[b] / T)));
          h -= pab * log(pab);
        }
      }
    mi[p] = s;
    hj[p] = h;
  }
}

hipError_t launch_mi_pairs(int A, int T, const uint32_t* d_masks, const uint8_t* d_aln1, size_t ld1, const uint8_t* d_aln2,
                           size_t ld2, const int64_t* d_idx1, const int64_t* d_idx2, size_t npairs, double* d_mi,
                           double* d_hj, hipStream_t stream) {
  dim3 grid((unsigned)((npairs + 15) / 16));
  const size_t lds = sizeof(double) * A * A * 16;
  if (A == 20)
    hipLaunchKernelGGL((mi_pairs_kernel<20>), grid, dim3(64), lds, stream, T, d_masks, d_aln1, ld1, d_aln2, ld2, d_idx1, d_idx2,
                       npairs, d_mi, d_hj);
  else if (A == 4)
    hipLaunchKernelGGL((mi_pairs_kernel<4>), grid, dim3(64), lds, stream, T, d_masks, d_aln1, ld1, d_aln2, ld2, d_idx1, d_idx2,
                       npairs, d_mi, d_hj);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---- MFMA path (SURVEY 8d "Mica MI"): for columns without ambiguous symbols the joint table of a column pair is one
// 32x32 block of the Gram matrix of one-hot matrices, H_i (32 x T) . H_j^T, exact in int8 x int8 -> int32.  With integer
// counts c the entropies need no logarithm at run time: sum_ab p_ab ln p_ab = (1/T) sum_ab f(c_ab) - ln T with
// f(c) = c ln c read from a (T+1)-entry table, so MI = ln T + (sum_ab f(c_ab) - sum_a f(c_a) - sum_b f(c_b)) / T and
// the epilogue is a layout-free sum over the accumulator registers.
typedef int cmx_i4 __attribute__((ext_vector_type(4)));
typedef int cmx_i16v __attribute__((ext_vector_type(16)));
constexpr int kMicaK = 32;   // taxa per MFMA step (v_mfma_i32_32x32x32_i8)

// one block per column: one-hot rows H[col][a][t] (a < 32, t < Tp, zero padded).  A symbol compatible with EVERY state
// (gap, X, N: mask = all ones) becomes pseudo-state A -- row A of the one-hot matrix -- and the column is marked in
// gap[]: its pairs stay on the matrix cores and the epilogue spreads the pseudo-state's counts (weight 1/A per state,
// the fractional counts of SiteTools::getCounts(.., resolveUnknowns = true)).  Any other ambiguity code sets flag[]
// (pairs of that column go to the LDS-table kernel).  S[col] = sum_a f(count_a) with the fractional counts.
__global__ __launch_bounds__(256) void mica_onehot_kernel(int A, int T, int Tp, const uint32_t* __restrict__ masks,
                                                          const uint8_t* __restrict__ aln, size_t ld,
                                                          int8_t* __restrict__ H, uint8_t* __restrict__ codes /*[n][Tp]: the one-hot row of each taxon (state, A = unknown), 63 = none*/,
                                                          uint8_t* __restrict__ flag,
                                                          uint8_t* __restrict__ gap, double* __restrict__ S,
                                                          int* __restrict__ anyflag, size_t n) {
  __shared__ int cnt[33];
  __shared__ int amb;
  const size_t i = blockIdx.x;
  if (i >= n) {   // the columns of padding behind the last one (kMicaCodePad): "no row" everywhere
    for (int t = threadIdx.x; t < Tp; t += blockDim.x) codes[i * (size_t)Tp + t] = 63;
    return;
  }
  const uint32_t full = (1u << A) - 1u;
  if (threadIdx.x < 33) cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) amb = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < Tp; t += blockDim.x) {
    unsigned c = t < T ? aln[(size_t)t * ld + i] : 63u;
    if (t < T) {
      if (c >= (unsigned)A) {
        if ((masks[c] & full) == full) c = (unsigned)A;     // unknown: pseudo-state
        else { amb = 1; c = 63u; }
      }
      if (c <= (unsigned)A) atomicAdd(&cnt[c], 1);
    }
    codes[i * (size_t)Tp + t] = (uint8_t)c;
    if (H) {
#pragma unroll
      for (int a = 0; a < 32; ++a) H[(i * 32 + a) * (size_t)Tp + t] = (int8_t)((c == (unsigned)a) ? 1 : 0);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    const double g = (double)cnt[A] / (double)A;
    for (int a = 0; a < A; ++a) {
      const double c = (double)cnt[a] + g;
      if (c > 0.0) s += c * log(c);
    }
    S[i] = s;
    flag[i] = (uint8_t)amb;
    gap[i] = (uint8_t)(cnt[A] > 0 ? 1 : 0);
    if (amb) atomicOr(anyflag, 1);
  }
}

// The same classification without the one-hot matrices (what the packed kernels need: codes, flags, column sums), 64 columns
// per workgroup: the alignment is [taxon][column], the codes [column][taxon].  mica_onehot_kernel reads a column with
// 256 one-byte loads from 256 cache lines (and counts with 256 LDS atomics on 21 addresses); here a wave reads 64 columns
// of one taxon in one line, counts per (column, state), and a tile of 64 taxa x 64 columns is turned in LDS so that the
// codes leave as 64-byte rows too.  S is summed in the same state order: the same bits.
__global__ __launch_bounds__(256) void mica_codes_kernel(int A, int T, int Tp, const uint32_t* __restrict__ masks,
                                                         const uint8_t* __restrict__ aln, size_t ld, uint8_t* __restrict__ codes,
                                                         uint8_t* __restrict__ flag, uint8_t* __restrict__ gap, double* __restrict__ S,
                                                         int* __restrict__ anyflag, size_t n, size_t npad /* columns behind n that get "no row" codes */) {
  constexpr int kRow = 68;               // bytes per taxon of the tile: 17 dwords, so that a column is read conflict-free
  constexpr int kCnt = 35;               // counts per column (states, the unknown; odd stride)
  __shared__ __attribute__((aligned(4))) uint8_t tile[64 * kRow];
  __shared__ int cnt[64 * kCnt];
  __shared__ int amb[64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t i0 = (size_t)blockIdx.x * 64, i = i0 + lane;
  const uint32_t full = (1u << A) - 1u;
  for (int k = threadIdx.x; k < 64 * kCnt; k += 256) cnt[k] = 0;
  if (threadIdx.x < 64) amb[threadIdx.x] = 0;
  __syncthreads();
  for (int tc = 0; tc < Tp; tc += 64) {
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {       // wave w: taxa tc + 4 r + w, one column per lane
      const int tl = 4 * r + w, t = tc + tl;
      unsigned c = 63u;
      if (t < T && i < n) {
        c = aln[(size_t)t * ld + i];
        if (c >= (unsigned)A) {
          if ((masks[c] & full) == full) c = (unsigned)A;     // unknown: pseudo-state
          else { amb[lane] = 1; c = 63u; }
        }
        if (c <= (unsigned)A) atomicAdd(&cnt[lane * kCnt + (int)c], 1);
      }
      tile[tl * kRow + lane] = (uint8_t)c;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {       // wave w: columns 4 r + w, one taxon per lane
      const int col = 4 * r + w;
      if (i0 + col < n + npad && tc + lane < Tp) codes[(i0 + col) * (size_t)Tp + tc + lane] = tile[lane * kRow + col];
    }
    __syncthreads();
  }
  if (threadIdx.x < 64 && i < n) {
    const int* cc = cnt + lane * kCnt;
    double s = 0.0;
    const double g = (double)cc[A] / (double)A;
    for (int a = 0; a < A; ++a) {
      const double c = (double)cc[a] + g;
      if (c > 0.0) s += c * log(c);
    }
    S[i] = s;
    flag[i] = (uint8_t)amb[lane];
    gap[i] = (uint8_t)(cc[A] > 0 ? 1 : 0);
    if (amb[lane]) atomicOr(anyflag, 1);
  }
}

// f[c] = c ln c for the integer counts 0..T, followed by f2[m] = (m / A^2) ln(m / A^2) for m = 0..A^2 T: the fractional
// count of a pair with unknowns is c_ab = N_ab + (N_aA + N_Ab) / A + N_AA / A^2 = m / A^2 with the INTEGER
// m = A^2 N_ab + A (N_aA + N_Ab) + N_AA, so those pairs need no logarithm at run time either
__global__ void mica_ftable_kernel(int T, int A, double* __restrict__ f, int* __restrict__ anyflag) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) *anyflag = 0;
  if (c <= T) f[c] = c > 1 ? (double)c * log((double)c) : 0.0;
  const int M = A * A * T;
  if (c <= M) {
    const double v = (double)c / (double)(A * A);
    const double fv = c > 0 ? v * log(v) : 0.0;
    f[T + 1 + c] = fv;
    // third table (cmx_mica4.hip): a zero, then f2[M0 ..], so that "m below M0" can be a load of entry 0
    const int M0 = M + 1 < kMicaLdsF2 ? M + 1 : kMicaLdsF2;
    double* hi = f + (T + 1) + (M + 1);
    if (c == 0) hi[0] = 0.0;
    if (c >= M0) hi[c - M0 + 1] = fv;
  }
}

// One workgroup (8 waves) per 8 x 4 tile of column pairs (8 columns of the first alignment, 4 of the second), wave w owns
// the 2 x 2 sub-tile (rows 2*(w/2).., columns 2*(w%2)..): 64 accumulator registers, so that two workgroups share a CU
// (four waves per SIMD) and one workgroup's barriers and table epilogue overlap the other's MFMAs.  The 12 operand tiles
// of a k-step (64 lanes x 16 B each) go through LDS once per workgroup.
constexpr int kMicaTileI = 8, kMicaTileJ = 4;
template <int A>
__global__ __launch_bounds__(512, 2) void mica_mfma_kernel(int T, int Tp, const int8_t* __restrict__ H1, size_t n1,
                                                           const uint8_t* __restrict__ flag1, const uint8_t* __restrict__ gap1,
                                                           const double* __restrict__ S1,
                                                           const int8_t* __restrict__ H2, size_t n2,
                                                           const uint8_t* __restrict__ flag2, const uint8_t* __restrict__ gap2,
                                                           const double* __restrict__ S2,
                                                           const double* __restrict__ ftab_g, int intra,
                                                           double* __restrict__ mi, double* __restrict__ hj, size_t ldo) {
  extern __shared__ __attribute__((aligned(16))) uint8_t mica_smem[];
  double* ftab = reinterpret_cast<double*>(mica_smem);                       // [T + 1]
  cmx_i4* ops = reinterpret_cast<cmx_i4*>(mica_smem + (((size_t)(T + 1) * 8 + 15) & ~(size_t)15));  // [2][12][64]
  constexpr int NOP = kMicaTileI + kMicaTileJ;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wi = w >> 1, wj = w & 1;
  for (int c = tid; c <= T; c += 512) ftab[c] = ftab_g[c];
  const size_t i0 = (size_t)blockIdx.y * kMicaTileI, j0 = (size_t)blockIdx.x * kMicaTileJ;
  if (intra && j0 + kMicaTileJ <= i0 + 1) {   // no pair with j > i in this tile: only the NaN convention of the intra layout
    if (tid < kMicaTileI * kMicaTileJ) {
      const size_t i = i0 + tid / kMicaTileJ, j = j0 + tid % kMicaTileJ;
      if (i < n1 && j < n2 && !flag1[i] && !flag2[j]) {
        mi[i * ldo + j] = __builtin_nan("");
        hj[i * ldo + j] = __builtin_nan("");
      }
    }
    return;
  }
  // does any column of this tile carry unknowns?  (independent loads, issued here so that the main loop hides them)
  unsigned gapbits = 0;
#pragma unroll
  for (int c = 0; c < kMicaTileI; ++c) gapbits |= gap1[i0 + c < n1 ? i0 + c : n1 - 1];
#pragma unroll
  for (int c = 0; c < kMicaTileJ; ++c) gapbits |= gap2[j0 + c < n2 ? j0 + c : n2 - 1];
  // loader role of this thread: 12 operand tiles x 64 lanes = 768 slots of 16 bytes, threads 0..383 take two each
  // (operand tile q = slot / 64: q < 8 column i0 + q of H1, else column j0 + q - 8 of H2; a lane's 16 bytes are row
  // (lane % 32), taxa group (lane / 32) of the one-hot matrix)
  const bool loader = tid < NOP * 32;
  const int q = loader ? tid >> 5 : 0;
  const size_t col = q < kMicaTileI ? (i0 + q < n1 ? i0 + q : n1 - 1) : (j0 + q - kMicaTileI < n2 ? j0 + q - kMicaTileI : n2 - 1);
  const int8_t* Hq = (q < kMicaTileI ? H1 : H2) + col * 32 * (size_t)Tp;
  const int l0 = 2 * (tid & 31);
  cmx_i16v acc[2][2];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[ii][jj][v] = 0;
  // Operand fetch runs THREE k-steps ahead of the MFMAs (registers -> LDS at the top of each step): with one step of
  // lookahead every k-step exposed most of an L2 / HBM round trip behind its barrier -- eight of them per tile at 256
  // taxa were three quarters of the kernel's time, the matrix cores idle meanwhile.
  constexpr int kAhead = 3;
  cmx_i4 st[kAhead][2] = {};
  auto fetch = [&](cmx_i4 (&dst)[2], int ks) {
    if (loader && ks < Tp) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int l = l0 + u;
        dst[u] = *reinterpret_cast<const cmx_i4*>(Hq + (size_t)(l & 31) * Tp + ks + 16 * (l >> 5));
      }
    }
  };
#pragma unroll
  for (int d = 0; d < kAhead; ++d) fetch(st[d], d * kMicaK);
  int buf = 0;
  auto step = [&](cmx_i4 (&cur)[2], int ks) {
    if (loader) {
#pragma unroll
      for (int u = 0; u < 2; ++u) ops[(buf * NOP + q) * 64 + l0 + u] = cur[u];
    }
    __syncthreads();
    fetch(cur, ks + kAhead * kMicaK);
    cmx_i4 a[2], b[2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) a[ii] = ops[(buf * NOP + 2 * wi + ii) * 64 + lane];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) b[jj] = ops[(buf * NOP + kMicaTileI + 2 * wj + jj) * 64 + lane];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) acc[ii][jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ii], b[jj], acc[ii][jj], 0, 0, 0);
    buf ^= 1;
  };
  for (int ks = 0; ks < Tp; ks += kAhead * kMicaK) {
    step(st[0], ks);
    if (ks + kMicaK < Tp) step(st[1], ks + kMicaK);
    if (ks + 2 * kMicaK < Tp) step(st[2], ks + 2 * kMicaK);
  }
  const double lnT = log((double)T), invT = 1.0 / (double)T;
  // sum_ab f(c_ab) of the wave's four pairs: 16 table lookups per lane and pair, then ONE reduce-scatter for all four
  // (lane bits 5 and 4 with v_permlane32/16_swap: row r of 16 lanes ends up with pair r; bits 3..0 with DPP row
  // rotations) instead of four butterfly reductions through ds_bpermute
  double ps[4];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      // register v of a lane holds table row 8 (v / 4) + v % 4 (+ 4 in the upper half of the wave): rows past the
      // pseudo-state (row A) are padding of the 32-row tile, structurally zero -- no lookup for them
      double s = 0.0;
#pragma unroll
      for (int v = 0; v < 16; ++v)
        if (8 * (v / 4) + v % 4 <= A) s += ftab[acc[ii][jj][v]];   // compile-time after unrolling
      ps[2 * ii + jj] = s;
    }
  swap32(ps[0], ps[2]);
  swap32(ps[1], ps[3]);
  double k0 = ps[0] + ps[2], k1 = ps[1] + ps[3];
  swap16(k0, k1);
  double s = k0 + k1;
  s += mica_dpp_f64<0x128>(s);   // row_ror:8
  s += mica_dpp_f64<0x124>(s);   // row_ror:4
  s += mica_dpp_f64<0x122>(s);   // row_ror:2
  s += mica_dpp_f64<0x121>(s);   // row_ror:1
  const bool blockgap = gapbits != 0;
  // Pairs with unknowns (pseudo-state A): the integer table N (states + pseudo-state, from the same accumulators) is
  // expanded into the fractional counts c_ab = N_ab + (N_aA + N_Ab) / A + N_AA / A^2 and f is evaluated with a
  // logarithm per cell.  v_mfma_i32_32x32x32_i8 leaves D[row][col] in register v of lane l with
  // row = 8 (v / 4) + 4 (l / 32) + v % 4, col = l % 32.
  {
    if (blockgap) {
      __syncthreads();                                  // the operand buffers are free now: reuse them as count tables
      int* tile = reinterpret_cast<int*>(ops) + w * 448;   // (A + 1)^2 <= 441 ints per wave
      const int A1 = A + 1;
      const double* f2 = ftab_g + T + 1;   // (m / A^2) ln(m / A^2), global (A^2 T + 1 entries, L2-resident)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = r >> 1, jj = r & 1;
        const size_t i = i0 + 2 * wi + ii, j = j0 + 2 * wj + jj;
        if (!(gap1[i < n1 ? i : n1 - 1] || gap2[j < n2 ? j : n2 - 1])) continue;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int row = 8 * (v / 4) + 4 * (lane / 32) + v % 4, col = lane % 32;
          if (row <= A && col <= A) tile[row * A1 + col] = acc[ii][jj][v];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int gg = tile[A * A1 + A];
        int mm[(A * A + 63) / 64];   // all table indices first, then all gathers (not one dependent L2 round trip each)
#pragma unroll
        for (int k = 0; k < (A * A + 63) / 64; ++k) {
          const int e = lane + 64 * k, x = e / A, y = e % A;
          mm[k] = e < A * A ? A * A * tile[x * A1 + y] + A * (tile[x * A1 + A] + tile[A * A1 + y]) + gg : 0;   // f2[0] = 0
        }
        double sg = 0.0;
#pragma unroll
        for (int k = 0; k < (A * A + 63) / 64; ++k) sg += f2[mm[k]];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sg += __shfl_xor(sg, off, 64);
        if ((lane >> 4) == r) s = sg;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  {
    const int r = lane >> 4;
    const size_t i = i0 + 2 * wi + (r >> 1), j = j0 + 2 * wj + (r & 1);
    if ((lane & 15) == 0 && i < n1 && j < n2 && !flag1[i] && !flag2[j]) {
      const bool valid = !intra || j > i;
      mi[i * ldo + j] = valid ? lnT + (s - S1[i] - S2[j]) * invT : __builtin_nan("");
      hj[i * ldo + j] = valid ? lnT - s * invT : __builtin_nan("");
    }
  }
}

// ---- protein alphabet, packed tiles.  A column needs 21 one-hot rows (20 states + the pseudo-state of unknowns), a
// 32-row MFMA tile per column wastes a third of the rows and (32/21)^2 of the matrix work AND of the table epilogue.
// Here THREE columns share a 64-row block (rows 21 c + state, row 63 = zero): a wave's 2 x 2 MFMA tiles are the 64 x 64
// Gram block of 3 x 3 column pairs (9 pairs where the one-column-per-tile kernel has 4), same operand traffic, same
// accumulators.  Workgroup tile: 12 columns of the first alignment x 6 of the second (4 x 2 blocks, one per wave).
// The epilogue sums f(count) per (column of the block row, column of the block column): an accumulator register's row
// block is known at compile time up to the lane's half (rows + 4 in lanes >= 32), its column block from the lane.
constexpr int kMica3I = 12, kMica3J = 6, kMicaP = 21;
__device__ __forceinline__ void mica3_tile(int T, int Tp, const uint8_t* __restrict__ C1, size_t n1,
                                                            const uint8_t* __restrict__ flag1, const uint8_t* __restrict__ gap1,
                                                            const double* __restrict__ S1,
                                                            const uint8_t* __restrict__ C2, size_t n2,
                                                            const uint8_t* __restrict__ flag2, const uint8_t* __restrict__ gap2,
                                                            const double* __restrict__ S2,
                                                            const double* __restrict__ ftab_g, int intra,
                                                            double* __restrict__ mi, double* __restrict__ hj, size_t ldo,
                                                            unsigned ntx, unsigned tlin) {
  extern __shared__ __attribute__((aligned(16))) uint8_t mica_smem[];
  constexpr int A = 20, P = kMicaP;
  double* ftab = reinterpret_cast<double*>(mica_smem);                       // [T + 1]
  cmx_i4* ops = reinterpret_cast<cmx_i4*>(mica_smem + (((size_t)(T + 1) * 8 + 15) & ~(size_t)15));  // [4][12][64]
  constexpr int NOP = 12, NI = 8;   // operand tiles per k-step: 8 of the first alignment (4 blocks), 4 of the second
  uint8_t* codes = reinterpret_cast<uint8_t*>(ops + 4 * NOP * 64) + 16384;   // the unknowns' path lays 8 x 8 KiB over the operand buffers + 16 KiB          // [18][Tp]: the tile's columns, one byte per taxon
  double* Scol = reinterpret_cast<double*>(codes + (size_t)(kMica3I + kMica3J) * Tp);   // [18] S of the tile's columns (12 + 6)
  int* fcol = reinterpret_cast<int*>(Scol + 18);   // [18] bit 0 partial ambiguity codes, bit 1 unknowns, bit 2 past the end
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wi = w >> 1, wj = w & 1;
  const size_t i0 = (size_t)(tlin / ntx) * kMica3I, j0 = (size_t)(tlin % ntx) * kMica3J;
  if (intra && j0 + kMica3J <= i0 + 1) {   // no pair with j > i in this tile: only the NaN convention of the intra layout
    if (tid < kMica3I * kMica3J) {
      const size_t i = i0 + tid / kMica3J, j = j0 + tid % kMica3J;
      if (i < n1 && j < n2 && !flag1[i] && !flag2[j]) {
        mi[i * ldo + j] = __builtin_nan("");
        hj[i * ldo + j] = __builtin_nan("");
      }
    }
    return;
  }
  // The operands are NOT read as one-hot matrices (21 rows x T bytes per column, 34 GB through the L2 for 5000 x 5000
  // columns -- the workgroups spent their lives waiting for them): a column travels as its T symbol bytes (the one-hot
  // row of each taxon), all 18 columns of the tile in one round trip, and the loader threads expand them to one-hot
  // operand tiles in LDS k-step by k-step (a dword of four symbols XOR the row's state, zero-byte test -> 0x01 bytes).
  const int chunks = Tp / 16;                  // 16-byte pieces per column
  for (int e = tid; e < (kMica3I + kMica3J) * chunks; e += 512) {
    const int c = e / chunks, o = e % chunks;
    const bool fi = c < kMica3I;
    const size_t want = fi ? i0 + c : j0 + (c - kMica3I);
    const size_t col = want < (fi ? n1 : n2) ? want : (fi ? n1 : n2) - 1;
    reinterpret_cast<cmx_i4*>(codes)[e] = *reinterpret_cast<const cmx_i4*>((fi ? C1 : C2) + col * (size_t)Tp + 16 * o);
  }
  for (int c = tid; c <= T; c += 512) ftab[c] = ftab_g[c];
  // per-column scalars, once per workgroup (not per thread, and not at the very end where their latency would show)
  if (tid < 18) {
    const bool fi = tid < kMica3I;
    const size_t c = fi ? i0 + tid : j0 + (tid - kMica3I), nc = fi ? n1 : n2;
    const size_t cc = c < nc ? c : nc - 1;
    Scol[tid] = (fi ? S1 : S2)[cc];
    fcol[tid] = (int)(fi ? flag1 : flag2)[cc] | ((int)(fi ? gap1 : gap2)[cc] << 1) | (c < nc ? 0 : 4);
  }
  // loader role: 12 operand tiles x 64 lanes = 768 slots of 16 bytes, threads 0..383 take two each.  Operand tile q:
  // q < 8: rows 32 (q % 2) .. + 31 of block q / 2 of the first alignment, else of block (q - 8) / 2 of the second; packed
  // row R = column R / 21 of the block, one-hot row R % 21; R = 63 is padding (state 31 matches no symbol).
  const bool loader = tid < NOP * 32;
  const int q = loader ? tid >> 5 : 0;
  const int l0 = 2 * (tid & 31);
  unsigned srow[2];      // the row's state, replicated in the four bytes of a dword
  const uint8_t* crow[2];  // the row's column in `codes`, at this lane's taxa group
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int R = 32 * (q & 1) + ((l0 + u) & 31);
    const int cb = R / P, st_ = R == 63 ? 31 : R % P;
    const int slot = (q < NI ? 3 * (q >> 1) : kMica3I + 3 * ((q - NI) >> 1)) + (cb < 3 ? cb : 2);
    srow[u] = (unsigned)st_ * 0x01010101u;
    crow[u] = codes + (size_t)slot * Tp + 16 * (l0 >> 5);
  }
  cmx_i16v acc[2][2];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[ii][jj][v] = 0;
  __syncthreads();   // symbols, table and column scalars are in LDS
  // two k-steps per barrier: four operand buffers, the pair being multiplied and the pair being expanded
  int buf = 0;
  for (int ks = 0; ks < Tp; ks += 2 * kMicaK) {
    const bool two = ks + kMicaK < Tp;
    if (loader) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 0 || two) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const cmx_i4 sy = *reinterpret_cast<const cmx_i4*>(crow[u] + ks + h * kMicaK);
            cmx_i4 oh;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const unsigned x = (unsigned)sy[d] ^ srow[u];
              const unsigned t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);   // 0x80 where the byte of x is zero
              oh[d] = (int)(t >> 7);
            }
            ops[((buf + h) * NOP + q) * 64 + l0 + u] = oh;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 0 || two) {
        cmx_i4 a[2], b[2];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) a[ii] = ops[((buf + h) * NOP + 2 * wi + ii) * 64 + lane];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) b[jj] = ops[((buf + h) * NOP + NI + 2 * wj + jj) * 64 + lane];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[ii][jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ii], b[jj], acc[ii][jj], 0, 0, 0);
      }
    }
    buf ^= 2;
  }
  const double lnT = log((double)T), invT = 1.0 / (double)T;
  const bool hi = lane >= 32;
  const int cl = lane & 31;
  // per lane: sums by (column a of the block row, MFMA tile column jj).  Register v of tile (ii, jj) is packed row
  // R = 32 ii + 8 (v / 4) + v % 4 (+ 4 if hi), packed column 32 jj + cl.
  double pa[3][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int R0 = 32 * ii + 8 * (v / 4) + v % 4, a0 = R0 / P, a1 = (R0 + 4) / P;
        const double val = ftab[acc[ii][jj][v]];
        if (a0 == a1) {
          pa[a0][jj] += val;
        } else {
          pa[a0][jj] += hi ? 0.0 : val;
          if (a1 < 3) pa[a1][jj] += hi ? val : 0.0;   // a1 == 3: packed row 63, padding (its counts are zero)
        }
      }
  // by column b of the block column: tile 0 holds packed columns 0..31 (b = 0 for cl < 21, else 1), tile 1 holds 32..63
  // (b = 1 for cl < 10, else 2; packed column 63 is padding)
  double sres[3];   // after the reductions: lanes with lane >> 4 == r hold pair 4 g + r in sres[g] (pair = 3 a + b)
  {
    double t[9];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      t[3 * a + 0] = cl < 21 ? pa[a][0] : 0.0;
      t[3 * a + 1] = (cl >= 21 ? pa[a][0] : 0.0) + (cl < 10 ? pa[a][1] : 0.0);
      t[3 * a + 2] = cl >= 10 ? pa[a][1] : 0.0;
    }
    sres[0] = mica_reduce4(t[0], t[1], t[2], t[3]);
    sres[1] = mica_reduce4(t[4], t[5], t[6], t[7]);
    sres[2] = mica_reduce4(t[8], 0.0, 0.0, 0.0);
  }
  int gapbits = 0;
#pragma unroll
  for (int c = 0; c < kMica3I + kMica3J; ++c) gapbits |= fcol[c] & 2;
  if (gapbits != 0) {
    // Pairs with unknowns: the fractional counts of resolveUnknowns = true are m / A^2 with the integer
    // m = A^2 N_ab + A (N_aG + N_Gb) + N_GG (G = the pseudo-state, row / column 20 of the pair's 21 x 21 block), and
    // sum_ab f2[m] comes from the second table (global, L2-resident).  The wave drops its 64 x 64 accumulator block into
    // LDS once (16-bit counts: T <= 2047), every pair with an unknown reads its cells from there; the nine sums are
    // reduced together like the fast path's.
    __syncthreads();                                     // the operand buffers are free now
    uint16_t* t16 = reinterpret_cast<uint16_t*>(ops) + (size_t)w * 4096;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int R = 32 * ii + 8 * (v / 4) + v % 4 + (hi ? 4 : 0), C = 32 * jj + cl;
          t16[R * 64 + C] = (uint16_t)acc[ii][jj][v];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const double* f2 = ftab_g + T + 1;   // (m / A^2) ln(m / A^2), A^2 T + 1 entries
    double sg[9];
#pragma unroll
    for (int pr = 0; pr < 9; ++pr) {
      sg[pr] = 0.0;
      const int a = pr / 3, b = pr % 3;
      if (!((fcol[3 * wi + a] | fcol[kMica3I + 3 * wj + b]) & 2)) continue;   // wave-uniform
      const uint16_t* tb = t16 + (P * a) * 64 + P * b;
      const int gg = tb[A * 64 + A];
      int mm[(A * A + 63) / 64];   // all table indices first, then all gathers
#pragma unroll
      for (int q_ = 0; q_ < (A * A + 63) / 64; ++q_) {
        const int e = lane + 64 * q_, x = e / A, y = e % A;
        mm[q_] = e < A * A ? A * A * (int)tb[x * 64 + y] + A * ((int)tb[x * 64 + A] + (int)tb[A * 64 + y]) + gg : 0;   // f2[0] = 0
      }
#pragma unroll
      for (int q_ = 0; q_ < (A * A + 63) / 64; ++q_) sg[pr] += f2[mm[q_]];
    }
    const double g0 = mica_reduce4(sg[0], sg[1], sg[2], sg[3]);
    const double g1 = mica_reduce4(sg[4], sg[5], sg[6], sg[7]);
    const double g2 = mica_reduce4(sg[8], 0.0, 0.0, 0.0);
    // lanes with lane >> 4 == r hold pair 4 g + r: take the general sum where that pair has an unknown
    const int r_ = lane >> 4;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int pr = 4 * g + r_;
      if (pr < 9 && ((fcol[3 * wi + pr / 3] | fcol[kMica3I + 3 * wj + pr % 3]) & 2)) sres[g] = g == 0 ? g0 : (g == 1 ? g1 : g2);
    }
  }
  if ((lane & 15) == 0) {
    const int r = lane >> 4;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int pr = 4 * g + r;
      if (pr < 9) {
        const int ci = 3 * wi + pr / 3, cj = kMica3I + 3 * wj + pr % 3;
        if (!((fcol[ci] | fcol[cj]) & 5)) {              // inside both alignments, no partial ambiguity code
          const size_t i = i0 + ci, j = j0 + (cj - kMica3I);
          const bool valid = !intra || j > i;
          const double s = sres[g];
          mi[i * ldo + j] = valid ? lnT + (s - Scol[ci] - Scol[cj]) * invT : __builtin_nan("");
          hj[i * ldo + j] = valid ? lnT - s * invT : __builtin_nan("");
        }
      }
    }
  }
}

#define CMX_MICA3_PARAMS                                                                                                      \
  int T, int Tp, const uint8_t *__restrict__ C1, size_t n1, const uint8_t *__restrict__ flag1, const uint8_t *__restrict__ gap1, \
      const double *__restrict__ S1, const uint8_t *__restrict__ C2, size_t n2, const uint8_t *__restrict__ flag2,              \
      const uint8_t *__restrict__ gap2, const double *__restrict__ S2, const double *__restrict__ ftab_g, int intra,             \
      double *__restrict__ mi, double *__restrict__ hj, size_t ldo, unsigned ntx
#define CMX_MICA3_ARGS T, Tp, C1, n1, flag1, gap1, S1, C2, n2, flag2, gap2, S2, ftab_g, intra, mi, hj, ldo, ntx
// alignments of more than 256 taxa, and the A/B switch CMX_MICA_TILES=3 (up to 256 taxa: cmx_mica4.hip)
__global__ __launch_bounds__(512, 4) void mica_mfma3_kernel(CMX_MICA3_PARAMS, unsigned ntiles, unsigned per_xcd) {
  // XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx.x % 8) and each XCD has its own L2:
  // XCD x takes a contiguous run of the row-major tile order, so that the tiles which complete an output cache line (a
  // tile writes 48-byte pieces of 12 rows) and re-read the same symbol columns meet in one L2.  At 5000 x 5000 x 256 the
  // launch time did not change (6.31 ms either way): kept for the traffic, not for the time.
  const unsigned tlin = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || tlin >= ntiles) return;
  mica3_tile(CMX_MICA3_ARGS, tlin);
}
// Column entropies (SiteTools::entropy per site; Mica.cpp:349-361 h1 / h2).  Two columns per wave, lane 32 c + a = state a of
// column c: every state's frequency is summed over the taxa in taxon order by its own lane and the A terms are added in
// state order by one lane -- the sums of the one-thread-per-column loop this replaces, bit for bit, in a quarter of its
// time (that loop was 256 x 20 predicated adds per thread on 79 waves: 0.09 ms per alignment of 5 000 columns, twice per
// Mica call, next to a 2.5 ms kernel).
template <int A>
__global__ __launch_bounds__(64) void column_entropy_kernel(int T, const uint32_t* __restrict__ masks, const uint8_t* __restrict__ aln,
                                                            size_t n, size_t ld, double* __restrict__ h) {
  static_assert(A <= 32, "a state per lane, two columns per wave");
  const int lane = threadIdx.x, a = lane & 31, c = lane >> 5;
  const size_t i = 2 * (size_t)blockIdx.x + c, ic = i < n ? i : n - 1;
  double p = 0.0;
  for (int t0 = 0; t0 < T; t0 += 16) {   // sixteen symbols in flight (the lanes of a column read the same byte)
    unsigned cs[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) cs[u] = t0 + u < T ? aln[(size_t)(t0 + u) * ld + ic] : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const unsigned sy = cs[u];
      if (sy == 0xffffffffu) continue;
      const uint32_t m = sy < (unsigned)A ? (1u << sy) : masks[sy];
      const double w = sy < (unsigned)A ? 1.0 : 1.0 / (double)__popc(m);
      if ((m >> a) & 1u) p += w;
    }
  }
  const double term = a < A && p > 0.0 ? (p / T) * log(p / T) : 0.0;   // (s - 0.0 == s: the states that never occur)
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < A; ++k) s -= __shfl(term, 32 * c + k, 64);
  if (a == 0 && i < n) h[i] = s;
}

// CMX_MICA_TILES=1: the one-column-per-tile kernel for proteins too (A/B timing of the packed kernel; same results)
static bool mica_one_column_tiles() {
  static const bool v = [] { const char* e = getenv("CMX_MICA_TILES"); return e && e[0] == '1'; }();
  return v;
}

// CMX_MICA_TILES=3: the eight-wave packed kernel for every tile (A/B timing of the four-wave kernel; same results)
static bool mica_eight_wave_tiles() {
  static const bool v = [] { const char* e = getenv("CMX_MICA_TILES"); return e && e[0] == '3'; }();
  return v;
}

// the one-hot matrices H [n][32][Tp] are operands of the one-column-per-tile kernel only (DNA, and CMX_MICA_TILES=1)
bool mica_needs_onehot(int A, int Tp) { return mica_one_column_tiles() || !(A == 20 || (A == 4 && Tp <= 256)); }

hipError_t launch_mi_columns(int A, int T, const uint32_t* d_masks, const uint8_t* d_aln1, size_t n1, size_t ld1,
                             const uint8_t* d_aln2, size_t n2, size_t ld2, int intra, double* d_mi, double* d_hj,
                             size_t ldo, double* d_h1, double* d_h2, const MicaWork* work, hipStream_t stream) {
  const size_t ntiles = ((n2 + 15) / 16) * n1;
  dim3 grid((unsigned)std::min<size_t>(ntiles, 8192));
  const size_t lds = sizeof(double) * A * A * 16;
  const int* anyf = (work && work->H1) ? work->anyflag : nullptr;
  // MFMA path for the columns without ambiguous symbols (work->H1 etc. non-null); the LDS kernel then only serves
  // the pairs that involve an ambiguous column.
  const uint8_t *f1 = nullptr, *f2 = nullptr;
  if (work && work->H1) {
    const int Tp = work->Tp;
    const bool needH = mica_needs_onehot(A, Tp);   // the packed protein kernels expand the symbol bytes themselves
    hipLaunchKernelGGL(mica_ftable_kernel, dim3((unsigned)((A * A * T) / 256 + 1)), dim3(256), 0, stream, T, A, work->ftab, work->anyflag);
    if (needH) {
      hipLaunchKernelGGL(mica_onehot_kernel, dim3((unsigned)(n1 + kMicaCodePad)), dim3(256), 0, stream, A, T, Tp, d_masks, d_aln1, ld1, work->H1, work->C1, work->flag1,
                         work->gap1, work->S1, work->anyflag, n1);
      if (!intra)
        hipLaunchKernelGGL(mica_onehot_kernel, dim3((unsigned)(n2 + kMicaCodePad)), dim3(256), 0, stream, A, T, Tp, d_masks, d_aln2, ld2, work->H2, work->C2,
                           work->flag2, work->gap2, work->S2, work->anyflag, n2);
    } else {   // codes, flags and column sums only: 64 columns per workgroup
      hipLaunchKernelGGL(mica_codes_kernel, dim3((unsigned)((n1 + kMicaCodePad + 63) / 64)), dim3(256), 0, stream, A, T, Tp, d_masks, d_aln1, ld1, work->C1, work->flag1,
                         work->gap1, work->S1, work->anyflag, n1, (size_t)kMicaCodePad);
      if (!intra)
        hipLaunchKernelGGL(mica_codes_kernel, dim3((unsigned)((n2 + kMicaCodePad + 63) / 64)), dim3(256), 0, stream, A, T, Tp, d_masks, d_aln2, ld2, work->C2,
                           work->flag2, work->gap2, work->S2, work->anyflag, n2, (size_t)kMicaCodePad);
    }
    dim3 g2((unsigned)((n2 + kMicaTileJ - 1) / kMicaTileJ), (unsigned)((n1 + kMicaTileI - 1) / kMicaTileI));
    const size_t lds2 = (((size_t)(T + 1) * 8 + 15) & ~(size_t)15) + 2 * (kMicaTileI + kMicaTileJ) * 64 * sizeof(cmx_i4);
    if (A == 20 && !mica_one_column_tiles()) {
      const unsigned ntx = (unsigned)((n2 + kMica3J - 1) / kMica3J), nty = (unsigned)((n1 + kMica3I - 1) / kMica3I);
      const unsigned ntiles = ntx * nty, per_xcd = (ntiles + 7) / 8;
      const size_t lds3 = lds2 + 2 * (kMica3I / 3 * 2 + kMica3J / 3 * 2) * 64 * sizeof(cmx_i4) + 16384 + (size_t)(kMica3I + kMica3J) * Tp + 18 * sizeof(double) + 20 * sizeof(int);
      if (lds3 > 64 * 1024) {
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&mica_mfma3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        if (ea != hipSuccess) return ea;
      }
      if (mica4_serves(A, Tp, n1, n2) && work->info1 && work->img2 && !mica_eight_wave_tiles()) {
        // up to 256 taxa: the four-wave kernel (cmx_mica4.hip), unknowns included
        const hipError_t e4 = launch_mica4(T, work, n1, n2, intra, d_mi, d_hj, ldo, stream);
        if (e4 != hipSuccess) return e4;
      } else
      hipLaunchKernelGGL(mica_mfma3_kernel, dim3(8 * per_xcd), dim3(512), lds3, stream, T, Tp, work->C1, n1,
                         work->flag1, work->gap1, work->S1, intra ? work->C1 : work->C2, n2, intra ? work->flag1 : work->flag2,
                         intra ? work->gap1 : work->gap2, intra ? work->S1 : work->S2, work->ftab, intra, d_mi, d_hj, ldo, ntx, ntiles, per_xcd);
    } else if (A == 20)
      hipLaunchKernelGGL(mica_mfma_kernel<20>, g2, dim3(512), lds2, stream, T, Tp, work->H1, n1,
                         work->flag1, work->gap1, work->S1, intra ? work->H1 : work->H2, n2, intra ? work->flag1 : work->flag2,
                         intra ? work->gap1 : work->gap2, intra ? work->S1 : work->S2, work->ftab, intra, d_mi, d_hj, ldo);
    else if (A == 4 && Tp <= 256 && !mica_one_column_tiles()) {
      // up to 256 taxa: the four-wave nucleotide kernel (cmx_mica4.hip), unknowns included (no one-hot matrices exist here)
      if (!mica_dna4_serves(A, Tp, n1, n2)) return hipErrorInvalidValue;
      const hipError_t e4 = launch_mica_dna4(T, work, n1, n2, intra, d_mi, d_hj, ldo, stream);
      if (e4 != hipSuccess) return e4;
    } else
      hipLaunchKernelGGL(mica_mfma_kernel<4>, g2, dim3(512), lds2, stream, T, Tp, work->H1, n1,
                         work->flag1, work->gap1, work->S1, intra ? work->H1 : work->H2, n2, intra ? work->flag1 : work->flag2,
                         intra ? work->gap1 : work->gap2, intra ? work->S1 : work->S2, work->ftab, intra, d_mi, d_hj, ldo);
    f1 = work->flag1;
    f2 = intra ? work->flag1 : work->flag2;
  }
  if (A == 20) {
    hipLaunchKernelGGL((mi_columns_kernel<20>), grid, dim3(64), lds, stream, T, d_masks, d_aln1, n1, ld1, d_aln2, n2,
                       ld2, intra, d_mi, d_hj, ldo, f1, f2, anyf);
    if (d_h1) hipLaunchKernelGGL((column_entropy_kernel<20>), dim3((unsigned)((n1 + 1) / 2)), dim3(64), 0, stream, T, d_masks, d_aln1, n1, ld1, d_h1);
    if (d_h2) hipLaunchKernelGGL((column_entropy_kernel<20>), dim3((unsigned)((n2 + 1) / 2)), dim3(64), 0, stream, T, d_masks, d_aln2, n2, ld2, d_h2);
  } else if (A == 4) {
    hipLaunchKernelGGL((mi_columns_kernel<4>), grid, dim3(64), lds, stream, T, d_masks, d_aln1, n1, ld1, d_aln2, n2,
                       ld2, intra, d_mi, d_hj, ldo, f1, f2, anyf);
    if (d_h1) hipLaunchKernelGGL((column_entropy_kernel<4>), dim3((unsigned)((n1 + 1) / 2)), dim3(64), 0, stream, T, d_masks, d_aln1, n1, ld1, d_h1);
    if (d_h2) hipLaunchKernelGGL((column_entropy_kernel<4>), dim3((unsigned)((n2 + 1) / 2)), dim3(64), 0, stream, T, d_masks, d_aln2, n2, ld2, d_h2);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace cmx
