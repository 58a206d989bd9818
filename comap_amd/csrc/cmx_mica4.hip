// Mica column MI for the protein alphabet, third arrangement (Mica.cpp:349-361, 646-689 -> per pair
// SiteTools::mutualInformation / jointEntropy): the one-hot Gram of two symbol columns on v_mfma_i32_32x32x32_i8, with
//
//   * persistent workgroups of FOUR waves (one per SIMD), each walking a run of tiles of 12 columns of the first
//     alignment x 3 columns of the second; wave w owns the 3 x 3 pairs of its block of three first-alignment columns;
//   * the first alignment's operands held in REGISTERS for the whole run: three columns x 20 states = 60 of the 64 rows
//     of a block (no pseudo-state row for the unknowns: see below), up to 8 k-steps x 2 row tiles x 4 registers (16 k-steps
//     = 512 taxa at one workgroup per CU), expanded
//     once per run straight from the symbol bytes.  The k-loop reads only the second alignment's operands from LDS
//     (2 ds_read_b128 per 4 MFMAs) -- the 8-wave kernel (cmx_kernels.hip, mica_mfma3_kernel) expanded all twelve operand
//     tiles of every tile again, and its busiest SIMDs spent more issue cycles on that than on the matrix products;
//   * the first operand's "one" is 8, the second's 1: an accumulator holds 8 x count, which IS the LDS address of
//     f(count) (the table sits at LDS address 0) -- no shift and no add in front of the 64 lookups per lane;
//   * the second alignment's operands expanded ONCE per call into an image of the tiles' LDS buffers (mica4_image_kernel),
//     so that a workgroup loads operands, not symbols (round 4);
//   * the tile loop software-pipelined INSIDE the wave (round 4; the loop itself is commented where it stands): a tile's
//     accumulators are two halves, the matrix core fills one while the vector unit empties the other, and one barrier per
//     tile is all the workgroup shares;
//   * a wave's nine pair totals by an LDS transpose: every lane dumps its three partial sums per half into the pair's row
//     of forty, four lanes per pair read ten each (round 3 selected nine values per lane and reduce-scattered them over
//     the wave: a quarter of the kernel's time);
//   * 20 rows per column put the column boundaries on the accumulator registers' row quads (rows 20 and 60 fall between
//     the two lane halves of a register, row 40 between registers): every accumulator register belongs to one column of the
//     wave's block, but for one quad per half that belongs to one by lane half.
//
// UNKNOWNS (gap, X: symbols compatible with every state -- what real alignments are full of) take the same road in a second
// instantiation (WEIGHTED).  The fractional counts of resolveUnknowns = true are c_ab = N_ab + (N_aG + N_Gb) / A + N_GG / A^2
// = m / A^2 with the integer m = A^2 N_ab + A (N_aG + N_Gb) + N_GG, and m IS a Gram: the sum over the taxa of u_a(t) v_b(t)
// with u, v = A where the symbol is the state and 1 where it is an unknown.  Expanded with those weights (20 and 1 fit a
// byte) the accumulators hold m (8 m with the weights doubled on one side and quadrupled on the other), and the epilogue
// looks f2[m] = (m / A^2) ln(m / A^2) up instead of f[N]: the first 4096 entries of that table sit in LDS (cells of up to
// ten taxa: most of them), the rest is gathered from the table in global memory (L2-resident) by the registers that hold
// such a cell, under their own EXEC mask (a copy of the table's tail behind a zero entry serves them).  No pseudo-state
// row, no dump of the accumulators to LDS, no per-pair gather of cells (the eight-wave kernel's way).  The plain
// instantiation walks the tiles where neither the workgroup's twelve first-alignment columns nor the tile's three carry
// an unknown, the weighted one all the others (it is exact for pairs without unknowns too); a tile that is not an
// instantiation's is not even fetched by it.
//
// The columns are tiled in SORTED order (those without unknowns first: mica_sort_columns_kernel below) and every result is
// written at its original (i, j).
//
// Pairs with a column that carries PARTIAL ambiguity codes (B, Z, R, Y ...) are left to the LDS-table kernel
// (launch_mi_columns).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <utility>

#include "cmx_device.h"
#include "cmx_lanes.h"

namespace cmx {

constexpr int kM4I = 12, kM4J = 3, kM4Rows = 20;
// The cells of a pair sum to 400 T, so at most 400 T / M0 <= 25 of them reach M0 = 4096 at T <= 256: a wave's nine pairs
// queue at most 225 cells per tile.
constexpr int kM4Corr = 12;
// (sixteen k-steps = up to 512 taxa: 50 such cells per pair, 450 per wave and tile)
constexpr int m4_qcap(int KS) { return KS > 8 ? 464 : 232; }
// Symbol codes of the SORTED column arrays the protein kernels read (mica_gather_columns_kernel recodes them): states
// 0..19, "no row" 31 (padding taxa and columns), the unknown 32 -- a bit of its own, so that the weighted expansion
// takes its unit from the symbol byte with a shift and a mask; 30 is the state of the padding rows (matches nothing)
constexpr unsigned kM4None = 31u, kM4Unknown = 32u, kM4PadRow = 30u;
constexpr unsigned kM4NoneX4 = kM4None * 0x01010101u, kM4PadRowX4 = kM4PadRow * 0x01010101u;
constexpr unsigned kM4FinRow = 40 * 8, kM4FinBytes = 9 * kM4FinRow + 16;   // a wave's nine rows of forty partial sums (+ a spare slot)
constexpr unsigned kM4MaxChunk = 64;   // tiles per run: one lane of a wave per tile when the run's tile info is loaded

// 16 symbols (four dwords) against one state.  Symbols and states are < 64, so 0x80 - (symbol ^ state) has bit 7 set
// exactly where they match and no byte borrows.
// plain: bytes `one` where they match (SHIFT / MASK move bit 7 to the one's place)
template <int SHIFT, unsigned MASK>
__device__ __forceinline__ cmx_i4 m4_expand(const cmx_i4 sy, unsigned srow) {
  cmx_i4 oh;
#pragma unroll
  for (int d = 0; d < 4; ++d) oh[d] = (int)(((0x80808080u - ((unsigned)sy[d] ^ srow)) >> SHIFT) & MASK);
  return oh;
}
// weighted: UNIT x A (A = 20) where the symbol is the state, UNIT where it is the unknown; `live` = 0 for the padding rows.
// UNIT is 2 on the first side and 4 on the second, so that the accumulators hold 8 m, the byte offset of f2[m]
template <unsigned UNIT>
__device__ __forceinline__ cmx_i4 m4_expand_weighted(const cmx_i4 sy, unsigned srow, unsigned live) {
  static_assert(kM4Unknown == 32u, "the unknown's bit is bit 5");
  cmx_i4 oh;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    // 20 UNIT = 5 x (4 UNIT): the match bit moved to the place of 4 UNIT, OR itself two places higher (bytes do not
    // carry); the unknown's bit 5 moved to the place of UNIT: seven instructions per dword (ten with the unknown compared
    // like a state)
    const unsigned eq = ((0x80808080u - ((unsigned)sy[d] ^ srow)) >> (UNIT == 2 ? 4 : 3)) & (0x01010101u * (4u * UNIT));
    oh[d] = (int)((((unsigned)sy[d] >> (UNIT == 2 ? 4 : 3)) & (live * UNIT)) | ((eq << 2) | eq));
  }
  return oh;
}
// f(0), f(1), ... f(N - 1) with compile-time arguments (std::integral_constant): the steps of the pipelined tile loop
template <class F, int... I>
__device__ __forceinline__ void m4_static_for_impl(F& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void m4_static_for(F&& f) {
  m4_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---- columns sorted by "has unknowns" before tiling.  A tile is the plain instantiation's only if none of its 15 columns
// carries an unknown: with unknowns in a tenth of the columns, scattered, that is one tile in five.  A STABLE partition
// (columns without unknowns first, the others behind them, each group in its original order -- neighbours stay
// neighbours, so a tile's three output columns still share cache lines) makes the blocks homogeneous but one.
// order[k] = original column at sorted position k.  One workgroup: a scan over n flags.
__global__ __launch_bounds__(1024) void mica_sort_columns_kernel(const uint8_t* __restrict__ gap, size_t n, unsigned* __restrict__ order,
                                                                unsigned* __restrict__ anygap /* 1: some column has unknowns */) {
  __shared__ unsigned wsum[16], base[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // first pass: number of columns without unknowns
  unsigned cnt = 0;
  for (size_t i = tid; i < n; i += 1024) cnt += gap[i] ? 0u : 1u;
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) wsum[w] = cnt;
  __syncthreads();
  if (tid == 0) {
    unsigned t = 0;
    for (int k = 0; k < 16; ++k) t += wsum[k];
    base[0] = 0;        // next position among the columns without unknowns
    base[1] = t;        // next position among the others
    *anygap = t < n ? 1u : 0u;
  }
  __syncthreads();
  for (size_t i0 = 0; i0 < n; i0 += 1024) {
    const size_t i = i0 + tid;
    const bool in = i < n, g = in && gap[i] != 0, c = in && !g;
    const unsigned long long mc = __ballot(c), mg = __ballot(g);
    if (lane == 0) { wsum[w] = (unsigned)__popcll(mc); }
    __syncthreads();
    unsigned offc = 0, totc = 0;
    for (int k = 0; k < 16; ++k) { if (k < w) offc += wsum[k]; totc += wsum[k]; }
    // (the others: position in the chunk minus the clean ones before it)
    const unsigned before = (unsigned)(64 * w + lane), cleanb = offc + (unsigned)__popcll(mc & ((1ull << lane) - 1ull));
    if (c) order[base[0] + cleanb] = (unsigned)i;
    if (g) order[base[1] + (before - cleanb) - 0u] = (unsigned)i;
    (void)mg;
    __syncthreads();
    if (tid == 0) {
      const unsigned inchunk = (unsigned)((n - i0) < 1024 ? (n - i0) : 1024);
      base[0] += totc;
      base[1] += inchunk - totc;
    }
    __syncthreads();
  }
}
// symbol bytes and column sums in sorted order, the padding columns behind them ("no row" everywhere)
__global__ __launch_bounds__(256) void mica_gather_columns_kernel(const unsigned* __restrict__ order, size_t n, int Tp,
                                                                  const uint8_t* __restrict__ C, const double* __restrict__ S,
                                                                  uint8_t* __restrict__ Cs, double* __restrict__ Ss) {
  const size_t k = blockIdx.x;
  if (k >= n) {
    for (int t = threadIdx.x; t < Tp; t += 256) Cs[k * (size_t)Tp + t] = (uint8_t)kM4None;
    return;
  }
  const size_t i = order[k];
  for (int t = threadIdx.x; t < Tp / 16; t += 256) {
    // mica_onehot_kernel's codes (unknown = 20, no row = 63) -> this file's (kM4Unknown, kM4None)
    cmx_i4 x = reinterpret_cast<const cmx_i4*>(C + i * (size_t)Tp)[t];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned v = (unsigned)x[d];
      const unsigned isu = ((0x80808080u - (v ^ 0x14141414u)) >> 7) & 0x01010101u, isn = ((0x80808080u - (v ^ 0x3f3f3f3fu)) >> 7) & 0x01010101u;
      x[d] = (int)(v ^ (isu * (20u ^ kM4Unknown)) ^ (isn * (63u ^ kM4None)));
    }
    reinterpret_cast<cmx_i4*>(Cs + k * (size_t)Tp)[t] = x;
  }
  if (threadIdx.x == 0) Ss[k] = S[i];
}
// per block of three SORTED columns (block k = sorted positions 3k .. 3k + 2): bits 0..2 column not served here (partial
// ambiguity codes or past the end), bits 3..5 column has unknowns
__global__ void mica_blockinfo_kernel(const unsigned* __restrict__ order, const uint8_t* __restrict__ flag, const uint8_t* __restrict__ gap,
                                      size_t n, size_t nblocks, unsigned* __restrict__ info) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nblocks) return;
  unsigned v = 0;
  for (int c = 0; c < 3; ++c) {
    const size_t p = 3 * k + c;
    if (p >= n || flag[order[p]]) v |= 1u << c;
    else if (gap[order[p]]) v |= 8u << c;
  }
  info[k] = v;
}
// intra layout: NaN wherever j <= i (the kernels below write each unordered pair once, at (min, max) of its columns)
__global__ void mica_nan_lower_kernel(size_t n, double* __restrict__ mi, double* __restrict__ hj, size_t ldo) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j <= i && j < n) {
    mi[i * ldo + j] = __builtin_nan("");
    hj[i * ldo + j] = __builtin_nan("");
  }
}

// ---- the second alignment's operands, expanded ONCE per call: the image of a tile is what the tile's operand buffer holds in
// LDS -- [2 KS operand tiles q][64 lanes] x 16 bytes; q = column tile q / KS, k-step q % KS; lane: packed column 32 (q / KS)
// + (lane & 31) = column C / 20 of the tile's three, state C % 20 (C >= 60: padding), taxa 32 (q % KS) + 16 (lane >> 5) ..
// + 15 -- so that the workgroups load operands instead of symbols and spend no vector instruction on expanding them (a
// tile was expanded once per block of twelve first-alignment columns: 417 times at cfg 5, 64 of the plain instantiation's
// ~200 vector instructions per wave and tile).  16 KB per tile at 256 taxa, one image per instantiation (the weights
// differ); the workgroups that run together walk the same chunk of tiles (run order in the kernel), so the image is read
// from L2.
template <int KS, bool WEIGHTED>
__global__ __launch_bounds__(256) void mica4_image_kernel(int Tp, const uint8_t* __restrict__ C2, cmx_i4* __restrict__ img) {
  constexpr int NQ = 2 * KS;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, cl = lane & 31, nks = Tp / 32;
  const size_t jt = blockIdx.x;
  for (int q = w; q < NQ; q += 4) {
    const int C = 32 * (q / KS) + cl;
    const unsigned srow = (C < 60 ? (unsigned)(C % kM4Rows) : kM4PadRow) * 0x01010101u;
    cmx_i4 raw = {(int)kM4NoneX4, (int)kM4NoneX4, (int)kM4NoneX4, (int)kM4NoneX4};
    // (the symbol arrays carry columns of padding behind the last one: no clamp)
    if (q % KS < nks) raw = *reinterpret_cast<const cmx_i4*>(C2 + (jt * kM4J + (C < 60 ? C / kM4Rows : 2)) * (size_t)Tp + 32 * (q % KS) + 16 * (lane >> 5));
    img[(jt * NQ + q) * 64 + lane] = WEIGHTED ? m4_expand_weighted<4>(raw, srow, srow == kM4PadRowX4 ? 0u : 0x01010101u)
                                              : m4_expand<7, 0x01010101u>(raw, srow);
  }
}

template <int KS, bool WEIGHTED>   // k-steps of 32 taxa: Tp <= 32 KS
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KS > 8 ? 1 : 2, KS > 8 ? 1 : 2))) void mica_mfma4_kernel(
    int T, int Tp, const uint8_t* __restrict__ C1, size_t n1, const unsigned* __restrict__ info1, const double* __restrict__ S1,
    const cmx_i4* __restrict__ img2, size_t n2, const unsigned* __restrict__ info2, const double* __restrict__ S2,
    const unsigned* __restrict__ order1, const unsigned* __restrict__ order2, const double* __restrict__ ftab_g, int intra,
    double* __restrict__ mi, double* __restrict__ hj, size_t ldo, unsigned nJ, unsigned chunk, unsigned nchunks, unsigned nruns,
    const unsigned* __restrict__ anygap1, const unsigned* __restrict__ anygap2) {
  // no column with unknowns anywhere: nothing for the weighted instantiation to do (its walk over the runs to find that out
  // was 0.03 ms of a 2.6 ms call)
  if (WEIGHTED && *anygap1 == 0 && *anygap2 == 0) return;
  // C1 / S1 / S2 / info1 / info2 and the columns behind img2 are in SORTED column order (mica_sort_columns_kernel); order1 /
  // order2 name the original column of a sorted position, which is where the results go
  extern __shared__ __attribute__((aligned(16))) uint8_t m4_smem[];   // the kernel's only LDS object: LDS address 0
  constexpr int NQ = 2 * KS;            // operand tiles of the second alignment per tile: 2 column tiles x KS k-steps
  constexpr bool DMA = WEIGHTED;        // how the next tile's operands reach LDS (the tile loop below)
  constexpr bool FINI = !WEIGHTED;      // a finished tile's results are made beside the next tile's products (the tile loop below)
  constexpr int SPT = NQ / 4;           // slots per thread
  // LDS address 0: plain f[0 .. T]; weighted f2[0 .. M0) followed by one zero entry (what a cell >= M0 reads there)
  const int M0 = 400 * T + 1 < kMicaLdsF2 ? 400 * T + 1 : kMicaLdsF2;
  const size_t ftab_bytes = ((size_t)(WEIGHTED ? M0 + 1 : T + 1) * 8 + 15) & ~(size_t)15;
  cmx_i4* ops = reinterpret_cast<cmx_i4*>(m4_smem + ftab_bytes);                 // [2][NQ][64]
  double* s2t = reinterpret_cast<double*>(ops + 2 * NQ * 64);                       // [4][4] S of the tile's columns (ring of four tiles)
  unsigned* j2t = reinterpret_cast<unsigned*>(s2t + 16);                          // [4][4] their original column indices
  // weighted: per wave the sums of its nine pairs' cells of M0 or more (gathered from global memory at the end of a tile),
  // and the queue of those cells
  double* corr = reinterpret_cast<double*>(j2t + 16);                             // [4][kM4Corr]
  constexpr int kM4QCap = m4_qcap(KS);
  unsigned* bigq = reinterpret_cast<unsigned*>(corr + 4 * kM4Corr);               // [4][kM4QCap]
  // per wave: the lanes' partial sums of a finished tile by pair (9 rows of 40 doubles), transposed through LDS into the
  // pairs' totals (finalize below)
  uint8_t* fin = reinterpret_cast<uint8_t*>(WEIGHTED ? reinterpret_cast<uint8_t*>(bigq + 4 * kM4QCap) : reinterpret_cast<uint8_t*>(j2t + 16));   // [4][kM4FinBytes]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool hi = lane >= 32;
  const int cl = lane & 31;
  if (WEIGHTED) {
    for (int c = tid; c <= M0; c += 256) reinterpret_cast<double*>(m4_smem)[c] = c < M0 ? ftab_g[T + 1 + c] : 0.0;
    if (tid < 4 * kM4Corr) corr[tid] = 0.0;
  } else
    for (int c = tid; c <= T; c += 256) reinterpret_cast<double*>(m4_smem)[c] = ftab_g[c];
  const unsigned M8 = 8u * (unsigned)M0;
  const double lnT = log((double)T), invT = 1.0 / (double)T;
  const int nks = Tp / 32;
  // the second alignment's operand image (mica4_image_kernel) through a buffer descriptor: this thread's 16 bytes of operand
  // tile q = w + 4 m in a VGPR offset, the tile in an SGPR offset
  const __amdgpu_buffer_rsrc_t rc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<cmx_i4*>(img2), 0, 0x7fffffff, 0x00020000);
  const unsigned ioff = (unsigned)((w * 64 + lane) * 16);
  // f2[m] for m >= M0: the third table holds a zero and then f2[M0 ..]; byte offset max(8 m - (8 M0 - 8), 0)
  const char* f2hi = reinterpret_cast<const char*>(ftab_g + (T + 1) + (400 * T + 1));
  // first alignment: row tile ii, packed row 32 ii + cl
  unsigned asrow[2];
  int acol[2];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const int R = 32 * ii + cl;
    asrow[ii] = (R < 60 ? (unsigned)(R % kM4Rows) : kM4PadRow) * 0x01010101u;
    acol[ii] = R < 60 ? R / kM4Rows : 2;
  }
  // where a lane's partial sums of column tile J go (first-alignment column a = 0; a adds 3 rows): pair row 3 a + b, b by
  // the lane's packed column (tile 0: columns 0..19 | 20..31, tile 1: 32..39 | 40..59; 60..63 are padding and write nothing), 40 slots per row
  const unsigned finw = (unsigned)(fin - m4_smem) + kM4FinBytes * (unsigned)w;   // (m4_smem is LDS address 0)
  const unsigned hs = hi ? 1u : 0u;
  const unsigned fina[2] = {finw + (cl < 20 ? 8u * (20u * hs + (unsigned)cl) : kM4FinRow + 8u * (12u * hs + (unsigned)(cl - 20))),
                            finw + (cl < 8 ? kM4FinRow + 8u * (24u + 8u * hs + (unsigned)cl)
                                           : 2u * kM4FinRow + 8u * (20u * hs + (unsigned)(cl - 8)))};
  // readers: lanes 4 p .. 4 p + 3 sum the forty partial sums of pair p = 3 a + b, ten each; lane 4 p writes the pair's results
  const unsigned pl = (unsigned)lane >> 2, plc = pl < 9 ? pl : 8u;
  const unsigned finr = finw + plc * kM4FinRow + 80u * ((unsigned)lane & 3u);
  const int al = (int)(plc / 3), bl = (int)(plc % 3);
  const bool writer = (lane & 3) == 0 && pl < 9;
  for (unsigned run = blockIdx.x; run < nruns; run += gridDim.x) {
    const unsigned nI = nruns / nchunks, I = run % nI, ch = run / nI;   // workgroups running together share a chunk of tiles
    const size_t i0 = (size_t)I * kM4I;
    unsigned jt0 = ch * chunk;
    const unsigned jt1 = jt0 + chunk < nJ ? jt0 + chunk : nJ;
    __syncthreads();   // the previous run's reads of the operand buffers and tile scalars are done (and the table is in LDS)
    if (intra) {
      // tiles J < 4 I hold no pair with a later second column (sorted positions): each unordered pair is computed once
      const unsigned first = 4 * I;
      if (jt0 < first) jt0 = first;
    }
    if (jt0 >= jt1) continue;
    // who needs what: this wave's block, the workgroup's four blocks, the run's tiles (one lane per tile, chunk <= 64)
    const unsigned inf1 = __builtin_amdgcn_readfirstlane(info1[(size_t)I * (kM4I / 3) + w]);   // (info arrays are padded to whole tiles)
    const int bad1 = inf1 & 7;
    unsigned anyA = 0;   // some block of the workgroup (that is served at all) has unknowns
#pragma unroll
    for (int k = 0; k < kM4I / 3; ++k) {
      const unsigned v = info1[(size_t)I * (kM4I / 3) + k];
      if ((v & 7) != 7) anyA |= (v >> 3) != 0;
    }
    const unsigned tinfo = jt0 + lane < jt1 ? info2[jt0 + lane] : 7u;
    // Every tile is walked by exactly one instantiation, with all four waves at work: the plain one where neither the
    // workgroup's twelve columns nor the tile's three carry an unknown, the weighted one everywhere else (it is exact for
    // the pairs without unknowns too: f2[400 N] = f[N]).  Splitting by (block, tile) instead made both instantiations walk
    // most tiles of a partly gapped alignment with some waves idle: 9.0 ms at 10 % gapped columns, more than with all of them.
    const bool tile_mine = (tinfo & 7) != 7 && (((tinfo >> 3) != 0 || anyA) == WEIGHTED);
    unsigned long long need = __ballot(tile_mine);
    if (need == 0) continue;
    // the lane's pair: its first-alignment column (original index, column sum)
    const size_t il = i0 + 3 * w + al, ilc = il < n1 ? il : n1 - 1;
    const double s1l = S1[ilc];
    const unsigned i1l = order1[ilc];
    const bool ok1 = writer && !((bad1 >> al) & 1);
    cmx_i4 areg[2][KS];
    {
      cmx_i4 raw[2][KS];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        // (the symbol arrays carry twelve columns of padding behind the last one: no clamp)
        const uint8_t* src = C1 + (i0 + 3 * w + acol[ii]) * (size_t)Tp + 16 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          raw[ii][ks] = ks < nks ? *reinterpret_cast<const cmx_i4*>(src + 32 * ks) : cmx_i4{(int)kM4NoneX4, (int)kM4NoneX4, (int)kM4NoneX4, (int)kM4NoneX4};
      }
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          areg[ii][ks] = WEIGHTED ? m4_expand_weighted<2>(raw[ii][ks], asrow[ii], asrow[ii] == kM4PadRowX4 ? 0u : 0x01010101u)
                                  : m4_expand<4, 0x08080808u>(raw[ii][ks], asrow[ii]);
    }
    // The operands of the next tile, two ways.  Plain instantiation: into registers (sixteen of them, held from behind the
    // first half of one tile to the first half of the next, where they are stored to the other operand buffer four
    // registers every fourth step).  Weighted instantiation: global -> LDS without passing through registers
    // (global_load_lds_dwordx4: lane l's 16 bytes land at base + 16 l, the image's order), requested in front of the first
    // half, waited for in front of the next barrier (vmcnt(0)).  Each on the other's road is slower, measured same box: the
    // plain one on DMA 2.95 -> 3.26 ms (the DMA costs the issuing wave more than a register load), the weighted one on
    // registers 4.48 -> 4.75 ms (256 registers, 12 spilled).  Inline asm for the reason given in cmx_kernels.hip: a DMA the
    // compiler knows of makes it wait for every outstanding load before the next LDS read.
    cmx_i4 braw[DMA ? 1 : SPT];
    double s2r = 0.0;
    unsigned j2r = 0;
    auto fetch_scalars = [&](unsigned jt) __attribute__((always_inline)) {   // (an iteration ahead of expand_scalars)
      if (tid < kM4J) {
        const size_t j = (size_t)jt * kM4J + tid, jc = j < n2 ? j : n2 - 1;
        s2r = S2[jc];
        j2r = order2[jc];
      }
    };
    auto fetch = [&](unsigned jt, int nbuf) __attribute__((always_inline)) {
      const unsigned soff = jt * (unsigned)(NQ * 1024);   // uniform
#pragma unroll
      for (int m = 0; m < SPT; ++m) {
        if (DMA) {
          const cmx_i4* dst = ops + (nbuf * NQ + w + 4 * m) * 64;
          const uint8_t* g = reinterpret_cast<const uint8_t*>(img2) + soff + ioff + 4096u * (unsigned)m;
          const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)reinterpret_cast<const uint8_t*>(dst));
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory");
        } else
          braw[m] = __builtin_bit_cast(cmx_i4, __builtin_amdgcn_raw_buffer_load_b128(rc2, ioff + 4096u * (unsigned)m, soff, 0));
      }
      fetch_scalars(jt);
    };
    auto expand_scalars = [&](unsigned slot) __attribute__((always_inline)) {   // ring of four tiles: the tile in the products, the one before it (its
      if (tid < kM4J) {                          // results not yet written) and the one being loaded
        s2t[4 * slot + tid] = s2r;
        j2t[4 * slot + tid] = j2r;
      }
    };
    // ---- the tile loop, software-pipelined inside the wave.  The accumulators of a tile are two halves (column tile J = 0,
    // 1: 32 registers each).  While the matrix core fills one half, the vector unit empties the other: the products of
    // (tile t, J = 0) run beside the lookups of (tile t - 1, J = 1) and the next tile's operands on their way to LDS; the products
    // of (tile t, J = 1) beside the lookups of (tile t, J = 0).  One MFMA per step, the step's share of the vector work
    // behind it, no instruction moved across a step (sched_barrier): a 32 x 32 x 32 product holds the matrix pipe for 32
    // cycles and the issue port for 8, the rest of the gap was idle unless the SIMD's other wave happened to be in its
    // epilogue (measured: one workgroup per CU 6.1 ms, two 3.8 -- a wave alone spent half its tile waiting).
    cmx_i16v acc[2][2];
    const cmx_i16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc[0][1] = zero;   // "the tile before the first": all cells 0, f = 0
    acc[1][1] = zero;
    double pa[3] = {0.0, 0.0, 0.0}, pm = 0.0;   // the sums of the half being emptied, by column a of the wave's block (pm: see `sum`)
    // weighted, cells of M0 or more: they read the zero behind the LDS table, and the lanes that hold one queue (8 m, pair)
    // in the wave's LDS queue -- a compare and a branch per register, the rest only where some lane has one; the queued
    // cells are gathered from global memory when the tile's sums are reduced.  (m4_smem is LDS address 0: the queue's LDS
    // byte address is its offset.)  Inline asm: written as a branch or a per-lane `if`, the compiler spilled 100 - 230
    // registers around the 64 regions.
    const unsigned qbeg = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<const uint8_t*>(bigq) - m4_smem) + 4u * kM4QCap * (unsigned)w);
    unsigned qp = qbeg;
    const unsigned corrb = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<const uint8_t*>(corr) - m4_smem) + 8u * kM4Corr * (unsigned)w);
    // the third table moved back by 8 M0 - 8 bytes: the accumulator itself (8 m) is then the gather's offset, entry 8 M0 the first real one
    const unsigned long long f2hi_a = (unsigned long long)f2hi - (M8 - 8u);
    const unsigned long long f2hi_u = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(f2hi_a >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((unsigned)f2hi_a);
    // pair of a cell = 3 a + b: b by the lane's packed column (tile 0: columns 0..19 | 20..31, tile 1: 32..39 | 40..59)
    unsigned btag[2] = {cl < 20 ? 0u : 1u, cl < 8 ? 1u : 2u}, hi3 = hi ? 3u : 0u;
    asm volatile("" : "+v"(btag[0]), "+v"(btag[1]), "+v"(hi3));   // (or 64 loop-invariant sums stay live across the tiles)

    // one accumulator register of half JJ: register v of tile (ii, JJ) is packed row R0 = 32 ii + 8 (v / 4) + v % 4 in the
    // lower lane half and R0 + 4 in the upper one, packed column 32 JJ + cl.  Rows 16..19 | 20..23 are the one register quad
    // whose halves belong to different columns (0 | 1); rows 60..63 are padding (count 0, f = 0).
    // plain: the accumulator is 8 x count = the LDS address of f(count).  weighted: 8 m, the LDS address of f2[m] for m < M0
    auto look = [&](auto jc, auto ec) __attribute__((always_inline)) -> double {
      constexpr int JJ = decltype(jc)::value, e = decltype(ec)::value, ii = e / 16, v = e % 16;
      const unsigned a8 = (unsigned)acc[ii][JJ][v];
      const double val = *reinterpret_cast<const __attribute__((address_space(3))) double*>(static_cast<uintptr_t>(WEIGHTED ? (a8 < M8 ? a8 : M8) : a8));
      if (WEIGHTED) {
        constexpr int R0 = 32 * ii + 8 * (v / 4) + v % 4, a0 = R0 / kM4Rows, a1 = (R0 + 4) / kM4Rows;
        constexpr bool mixed = !(a0 == a1 || a1 == 3);   // a0 == 0, a1 == 1  (rows 60..63 are padding: never large)
        const unsigned tag = mixed ? btag[JJ] + hi3 : btag[JJ];
        unsigned long long sv;
        unsigned t, q, cnt;
        asm volatile("v_cmp_le_u32_e32 vcc, %[m8], %[a]\n\t"
                     "s_cbranch_vccz .Lmq%=\n\t"
                     "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                     "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                     "v_lshl_add_u32 %[t], %[t], 2, %[qp]\n\t"
                     "v_lshl_or_b32 %[e], %[a], 1, %[tag]\n\t"
                     "v_add_u32_e32 %[e], %[a3], %[e]\n\t"
                     "s_and_saveexec_b64 %[sv], vcc\n\t"
                     "ds_write_b32 %[t], %[e]\n\t"
                     "s_mov_b64 exec, %[sv]\n\t"
                     "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                     "s_lshl2_add_u32 %[qp], %[cnt], %[qp]\n"
                     ".Lmq%=:"
                     : [qp] "+s"(qp), [t] "=&v"(t), [e] "=&v"(q), [sv] "=&s"(sv), [cnt] "=&s"(cnt)
                     : [a] "v"(acc[ii][JJ][v]), [m8] "s"(M8), [tag] "v"(tag), [a3] "n"(mixed ? 0 : 3 * a0)
                     : "vcc", "scc", "memory");
      }
      return val;
    };
    auto sum = [&](auto ec, double val) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value, ii = e / 16, v = e % 16;
      constexpr int R0 = 32 * ii + 8 * (v / 4) + v % 4, a0 = R0 / kM4Rows, a1 = (R0 + 4) / kM4Rows;
      if (a0 == a1 || a1 == 3) pa[a0] += val;
      else pm += val;    // a0 == 0 in the lower lane half, 1 in the upper one
    };
    // a half is emptied: the lane's three sums to its slots of the pair rows
    auto dump = [&](auto jc) __attribute__((always_inline)) {
      constexpr int JJ = decltype(jc)::value;
      pa[0] += hi ? 0.0 : pm;
      pa[1] += hi ? pm : 0.0;
      pm = 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        // (packed columns 60..63 are padding: no pair; their lanes' zeros go to the wave's spare slot -- a select, not a
        // branch: with control flow at the end of a half the register allocator spilled)
        const unsigned at = fina[JJ] + 3u * kM4FinRow * (unsigned)a;
        *reinterpret_cast<__attribute__((address_space(3))) double*>(static_cast<uintptr_t>(JJ == 1 && cl >= 28 ? finw + 9u * kM4FinRow : at)) = pa[a];
        pa[a] = 0.0;
      }
    };
    // ---- the sums of a finished tile (both halves emptied and dumped) -> its nine pairs' results.  `slot`: the tile's place in
    // the scalar ring.  Weighted instantiation: in a row between the halves (its queue of large cells is the finished tile's
    // until it is emptied here).
    auto finalize_row = [&](unsigned jtp, unsigned slot) __attribute__((always_inline)) {
      const unsigned inf2 = __builtin_amdgcn_readlane(tinfo, jtp - jt0);
      const int bad2 = inf2 & 7;
      // the queued cells: one gather for (nearly always) all of them, in flight during the reductions below.  Straight-line
      // asm under an EXEC mask (no lanes when the queue is empty), for the register allocator's sake as above.
      const unsigned nq = WEIGHTED ? __builtin_amdgcn_readfirstlane((qp - qbeg) >> 2) : 0u;
      unsigned ge = 0;
      double gv = 0.0;
      if (WEIGHTED) {
        unsigned long long sv;
        unsigned t;
        asm volatile("v_cmp_gt_u32_e32 vcc, %[nq], %[lane]\n\t"
                     "s_and_saveexec_b64 %[sv], vcc\n\t"
                     "v_lshl_add_u32 %[t], %[lane], 2, %[qb]\n\t"
                     "ds_read_b32 %[e], %[t]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_lshrrev_b32_e32 %[t], 1, %[e]\n\t"
                     "v_and_b32_e32 %[t], 0xfffffff8, %[t]\n\t"
                     "global_load_dwordx2 %[g], %[t], %[base]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [e] "+v"(ge), [g] "+v"(gv), [t] "=&v"(t), [sv] "=&s"(sv)
                     : [nq] "s"(nq), [lane] "v"(lane), [qb] "s"(qbeg), [base] "s"(f2hi_u)
                     : "vcc", "memory");
      }
      // forty partial sums per pair, ten per lane, the four lanes of a pair by two quad swaps (LDS of one wave: in order
      // behind the dumps)
      double sres;
      {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const __attribute__((address_space(3))) d2* rp = reinterpret_cast<const __attribute__((address_space(3))) d2*>(static_cast<uintptr_t>(finr));
        const d2 v0 = rp[0], v1 = rp[1], v2 = rp[2], v3 = rp[3], v4 = rp[4];
        sres = (((v0[0] + v0[1]) + (v1[0] + v1[1])) + ((v2[0] + v2[1]) + (v3[0] + v3[1]))) + (v4[0] + v4[1]);
        sres += mica_dpp_f64<0xB1>(sres);   // quad_perm [1, 0, 3, 2]
        sres += mica_dpp_f64<0x4E>(sres);   // quad_perm [2, 3, 0, 1]
      }
      if (WEIGHTED) {
        // the gathered values into the wave's per-pair sums (LDS atomics of one wave: program order, lane order); beyond
        // the first 64 queued cells (rare) a loop of the same steps with the round trip exposed
        unsigned long long sv;
        unsigned t, e2, sq;
        double g2;
        asm volatile("v_cmp_gt_u32_e32 vcc, %[nq], %[lane]\n\t"
                     "s_and_saveexec_b64 %[sv], vcc\n\t"
                     "v_and_b32_e32 %[t], 15, %[e]\n\t"
                     "v_lshl_add_u32 %[t], %[t], 3, %[cb]\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "ds_add_f64 %[t], %[g]\n\t"
                     "s_mov_b64 exec, %[sv]\n\t"
                     "s_movk_i32 %[sq], 64\n"
                     ".Lmr%=:\n\t"
                     "s_cmp_ge_u32 %[sq], %[nq]\n\t"
                     "s_cbranch_scc1 .Lmd%=\n\t"
                     "v_add_u32_e32 %[t], %[sq], %[lane]\n\t"
                     "v_cmp_gt_u32_e32 vcc, %[nq], %[t]\n\t"
                     "s_and_saveexec_b64 %[sv], vcc\n\t"
                     "v_lshl_add_u32 %[t], %[t], 2, %[qb]\n\t"
                     "ds_read_b32 %[e2], %[t]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_lshrrev_b32_e32 %[t], 1, %[e2]\n\t"
                     "v_and_b32_e32 %[t], 0xfffffff8, %[t]\n\t"
                     "global_load_dwordx2 %[g2], %[t], %[base]\n\t"
                     "v_and_b32_e32 %[e2], 15, %[e2]\n\t"
                     "v_lshl_add_u32 %[e2], %[e2], 3, %[cb]\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "ds_add_f64 %[e2], %[g2]\n\t"
                     "s_mov_b64 exec, %[sv]\n\t"
                     "s_add_u32 %[sq], %[sq], 64\n\t"
                     "s_branch .Lmr%=\n"
                     ".Lmd%=:"
                     : [t] "=&v"(t), [e2] "=&v"(e2), [g2] "=&v"(g2), [sv] "=&s"(sv), [sq] "=&s"(sq), "+v"(sres)
                     : [nq] "s"(nq), [lane] "v"(lane), [e] "v"(ge), [g] "v"(gv), [cb] "s"(corrb), [qb] "s"(qbeg), [base] "s"(f2hi_u)
                     : "vcc", "scc", "memory");
        qp = qbeg;
      }
      // intra: the pair is this tile's if its second column comes later in SORTED order; it is written at (smaller,
      // larger) ORIGINAL column (MI and the joint entropy are symmetric; j <= i holds NaN, mica_nan_lower_kernel)
      if (ok1 && !((bad2 >> bl) & 1) && (!intra || (size_t)jtp * kM4J + bl > il)) {
        const size_t oi = i1l, oj = j2t[4 * slot + bl];
        const size_t i = intra && oj < oi ? oj : oi, j = intra && oj < oi ? oi : oj;
        const double sp = WEIGHTED ? sres + corr[kM4Corr * w + plc] : sres;
        mi[i * ldo + j] = lnT + (sp - s1l - s2t[4 * slot + bl]) * invT;
        hj[i * ldo + j] = lnT - sp * invT;
      }
      if (WEIGHTED && lane < kM4Corr) corr[kM4Corr * w + lane] = 0.0;   // (after the reads above: one wave, LDS in order)
    };
    // Plain instantiation: in four pieces at four steps of the NEXT tile's second half (the LDS round trips and the stores
    // behind products: in a row they were 620 of a tile's 3 870 cycles).  `have`: there is such a tile (not in a run's first
    // iteration).  Forty partial sums per pair, ten per lane (LDS of one wave: in order behind the dumps), in two rounds.
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 f_v0 = {0.0, 0.0}, f_v1 = f_v0, f_v2 = f_v0;
    double f_sres = 0.0, f_s2 = 0.0;
    unsigned f_oj = 0;
    auto fin = [&](auto pc, unsigned jtp, unsigned slot, bool have) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value;
      const __attribute__((address_space(3))) d2* rp = reinterpret_cast<const __attribute__((address_space(3))) d2*>(static_cast<uintptr_t>(finr));
      if (P == 0) {
        f_v0 = rp[0];
        f_v1 = rp[1];
        f_v2 = rp[2];
      } else if (P == 1) {
        f_sres = ((f_v0[0] + f_v0[1]) + (f_v1[0] + f_v1[1])) + (f_v2[0] + f_v2[1]);
        f_v0 = rp[3];
        f_v1 = rp[4];
        f_oj = j2t[4 * slot + bl];   // and the pair's scalars
        f_s2 = s2t[4 * slot + bl];
      } else if (P == 2) {
        f_sres += (f_v0[0] + f_v0[1]) + (f_v1[0] + f_v1[1]);
        f_sres += mica_dpp_f64<0xB1>(f_sres);   // the four lanes of a pair: quad_perm [1, 0, 3, 2],
        f_sres += mica_dpp_f64<0x4E>(f_sres);   // [2, 3, 0, 1]
      } else {
        const unsigned inf2 = __builtin_amdgcn_readlane(tinfo, jtp - jt0);
        const int bad2 = inf2 & 7;
        // intra: as above
        if (have && ok1 && !((bad2 >> bl) & 1) && (!intra || (size_t)jtp * kM4J + bl > il)) {
          const size_t oi = i1l, oj = f_oj;
          const size_t i = intra && oj < oi ? oj : oi, j = intra && oj < oi ? oi : oj;
          mi[i * ldo + j] = lnT + (f_sres - s1l - f_s2) * invT;
          hj[i * ldo + j] = lnT - f_sres * invT;
        }
      }
    };
    auto finalize = [&](unsigned jtp, unsigned slot) __attribute__((always_inline)) {
      if (FINI) m4_static_for<4>([&](auto pc) { fin(pc, jtp, slot, true); });
      else finalize_row(jtp, slot);
    };

    constexpr int NS = 2 * KS;        // steps (products) of a half
    constexpr int RPS = 32 / NS;      // accumulator registers of the other half emptied per step
    // the products of half J of the tile in operand buffer `buf`, the lookups of half 1 - J (of the previous tile for J = 0)
    // and, for EXP, the next tile's operands from registers into operand buffer `nbuf`
    auto half = [&](auto jc, auto expc, auto finc, int buf, int nbuf, unsigned jtp, unsigned pslot, bool have_p) __attribute__((always_inline)) {
      constexpr int J = decltype(jc)::value, O = 1 - J;
      constexpr bool EXP = decltype(expc)::value, FIN = decltype(finc)::value;
      const cmx_i4* ob = ops + (buf * NQ + J * KS) * 64 + lane;
      cmx_i4 nb = ob[0], bb = nb;
      // a lookup's value is summed LAG steps after the lookup is issued (one LDS round trip is about two products)
      constexpr int LAG = 2;   // (1 .. 4 measured alike)
      double vr[LAG + 1][RPS];
      m4_static_for<NS>([&](auto sc) {
        constexpr int st = decltype(sc)::value, ii = st & 1, ks = st >> 1;
        if (ii == 0) {
          bb = nb;
          if (ks + 1 < KS) nb = ob[(ks + 1) * 64];   // operands one k-step ahead of the products
        }
        acc[ii][J] = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[ii][ks], bb, ks ? acc[ii][J] : zero, 0, 0, 0);
        m4_static_for<RPS>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          vr[st % (LAG + 1)][r] = look(std::integral_constant<int, O>{}, std::integral_constant<int, st * RPS + r>{});
        });
        if (st >= LAG)
          m4_static_for<RPS>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            sum(std::integral_constant<int, (st >= LAG ? st - LAG : 0) * RPS + r>{}, vr[(st + 1) % (LAG + 1)][r]);
          });
        if (FIN)   // a piece of the previous tile's results at every fourth of the half
          m4_static_for<4>([&](auto pc) {
            if (decltype(pc)::value * NS / 4 == st) fin(pc, jtp, pslot, have_p);
          });
        if (EXP && st % 4 == 3) ops[(nbuf * NQ + w + 4 * (st / 4)) * 64 + lane] = braw[DMA ? 0 : st / 4];   // operand tile w + 4 m, m = st / 4
        __builtin_amdgcn_sched_barrier(0);
      });
      m4_static_for<LAG * RPS>([&](auto rc) {
        constexpr int r = decltype(rc)::value, st = NS - LAG + r / RPS;
        sum(std::integral_constant<int, st * RPS + r % RPS>{}, vr[st % (LAG + 1)][r % RPS]);
      });
      // (an asm statement with vector outputs counts as divergent in all its outputs; across the loop's back edge the
      // queue pointer has to be visibly uniform or it is given a vector register)
      if (WEIGHTED) qp = __builtin_amdgcn_readfirstlane(qp);
      dump(std::integral_constant<int, O>{});
    };
    // the sums of a finished tile (both halves emptied) -> its nine pairs' results; `slot`: its place in the scalar ring
    unsigned jt = jt0 + (unsigned)__builtin_ctzll(need), jn = 0, jtp = 0;
    need &= need - 1;
    fetch(jt, 0);
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else {
#pragma unroll
      for (int m = 0; m < SPT; ++m) ops[(w + 4 * m) * 64 + lane] = braw[DMA ? 0 : m];
    }
    expand_scalars(0);
    bool have_p = false, have_n = need != 0;
    if (have_n) {
      jn = jt0 + (unsigned)__builtin_ctzll(need);
      need &= need - 1;
      if (!DMA) fetch(jn, 1);
    }
    int buf = 0;
    unsigned slot = 0;
    const std::false_type no{};
    const std::true_type yes{};
    const std::integral_constant<int, 0> h0{};
    const std::integral_constant<int, 1> h1{};
    for (;;) {
      __syncthreads();   // this tile's operands and scalars are in LDS; every wave is done with the other operand buffer
      const unsigned pslot = (slot + 3) & 3;
      if (DMA) {
        // weighted: request the next tile's operands | first half | previous tile's results | second half | wait
        if (have_n) fetch(jn, buf ^ 1);
        half(h0, no, no, buf, buf, 0u, 0u, false);
        if (have_p) finalize(jtp, pslot);
        half(h1, no, no, buf, buf, 0u, 0u, false);
        if (have_n) expand_scalars((slot + 1) & 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's operands are in LDS (in front of the barrier)
        have_p = true;
        jtp = jt;
        if (!have_n) break;
        jt = jn;
        have_n = need != 0;
        if (have_n) {
          jn = jt0 + (unsigned)__builtin_ctzll(need);
          need &= need - 1;
        }
      } else {
        // plain: first half + the next tile's operands from registers to LDS | request the tile after it |
        // second half + the previous tile's results
        if (have_n) {
          half(h0, yes, no, buf, buf ^ 1, 0u, 0u, false);
          expand_scalars((slot + 1) & 3);
        } else
          half(h0, no, no, buf, buf, 0u, 0u, false);
        const unsigned jcur = jt;
        const bool had_n = have_n;
        if (have_n) {
          jt = jn;
          have_n = need != 0;
          if (have_n) {
            jn = jt0 + (unsigned)__builtin_ctzll(need);
            need &= need - 1;
            fetch(jn, 0);   // (the registers are free: the first half stored what they held)
          }
        }
        half(h1, no, yes, buf, buf, jtp, pslot, have_p);
        have_p = true;
        jtp = jcur;
        if (!had_n) break;
      }
      buf ^= 1;
      slot = (slot + 1) & 3;
    }
    // the last tile's second half
    m4_static_for<4>([&](auto bc) {
      constexpr int b0 = decltype(bc)::value * 8;
      double val[8];
      m4_static_for<8>([&](auto rc) { val[decltype(rc)::value] = look(std::integral_constant<int, 1>{}, std::integral_constant<int, b0 + decltype(rc)::value>{}); });
      m4_static_for<8>([&](auto rc) { sum(std::integral_constant<int, b0 + decltype(rc)::value>{}, val[decltype(rc)::value]); });
    });
    if (WEIGHTED) qp = __builtin_amdgcn_readfirstlane(qp);
    dump(std::integral_constant<int, 1>{});
    finalize(jtp, slot);
  }
}
// ---- nucleotides.  Four states: SIXTEEN columns share a 64-row block, a wave's 64 x 64 accumulator block is the Gram of
// 16 x 16 column pairs (nine for proteins), and the table of the weighted form -- m = 16 N_ab + 4 (N_aG + N_Gb) + N_GG <= 16 T,
// 4 097 entries at 256 taxa -- fits the LDS whole: ONE instantiation serves columns with and without unknowns (weights 8 / 2
// on the first side, 16 / 4 on the second: the accumulator is 8 m, the LDS address of f2[m]), nothing is gathered from
// global memory, nothing needs sorting.  A workgroup's tile is 64 columns of the first alignment x 16 of the second =
// 1 024 pairs.  Epilogue per lane: the four state rows of a first-alignment column are four consecutive accumulator
// registers (4 lookups, 3 adds), the four states of a second-alignment column are four neighbouring lanes (two DPP adds
// within the quad); sixteen sums per lane, 256 pairs per wave, stored as 128-byte row pieces.  The one-column-per-tile
// kernel this replaces (cmx_kernels.hip, mica_mfma_kernel<4>) used 5 of 32 rows of every operand tile: 12.2 ms for
// 5 000 x 5 000 columns x 256 taxa.
constexpr int kD4I = 64, kD4J = 16;
template <unsigned KNOWN, unsigned UNK>
__device__ __forceinline__ cmx_i4 d4_expand(const cmx_i4 sy, unsigned srow) {
  cmx_i4 oh;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const unsigned eq = ((0x80808080u - ((unsigned)sy[d] ^ srow)) >> 7) & 0x01010101u;
    const unsigned un = ((0x80808080u - ((unsigned)sy[d] ^ 0x04040404u)) >> 7) & 0x01010101u;
    oh[d] = (int)(eq * KNOWN + un * UNK);   // bytes do not carry: at most KNOWN (a symbol is a state or the unknown)
  }
  return oh;
}
template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void mica_dna4_kernel(
    int T, int Tp, const uint8_t* __restrict__ C1, size_t n1, const uint8_t* __restrict__ flag1, const double* __restrict__ S1,
    const uint8_t* __restrict__ C2, size_t n2, const uint8_t* __restrict__ flag2, const double* __restrict__ S2,
    const double* __restrict__ ftab_g, int intra, double* __restrict__ mi, double* __restrict__ hj, size_t ldo, unsigned nJ,
    unsigned chunk, unsigned nchunks, unsigned nruns) {
  extern __shared__ __attribute__((aligned(16))) uint8_t d4_smem[];   // the kernel's only LDS object: LDS address 0
  constexpr int NQ = 2 * KS, SPT = NQ / 4;
  const int M = 16 * T + 1;                                             // entries of f2
  const size_t tab_bytes = ((size_t)M * 8 + 15) & ~(size_t)15;
  cmx_i4* ops = reinterpret_cast<cmx_i4*>(d4_smem + tab_bytes);         // [2][NQ][64]
  double* s1t = reinterpret_cast<double*>(ops + 2 * NQ * 64);           // [64] S of the workgroup's first-alignment columns
  double* s2t = s1t + kD4I;                                             // [2][16] S of the tile's columns
  int* ok1 = reinterpret_cast<int*>(s2t + 2 * kD4J);                    // [64] column is served (inside, no partial ambiguity code)
  int* ok2 = ok1 + kD4I;                                                // [2][16]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, cl = lane & 31;
  for (int c = tid; c < M; c += 256) reinterpret_cast<double*>(d4_smem)[c] = ftab_g[T + 1 + c];
  const double lnT = log((double)T), invT = 1.0 / (double)T;
  const int nks = Tp / 32;
  const __amdgpu_buffer_rsrc_t rc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(C2), 0, 0x7fffffff, 0x00020000);
  unsigned bsrow[SPT], boff[SPT];
#pragma unroll
  for (int m = 0; m < SPT; ++m) {
    const int q = w + 4 * m, C = 32 * (q / KS) + cl;     // packed column C: column C / 4 of the tile, state C % 4
    bsrow[m] = (unsigned)(C % 4) * 0x01010101u;
    boff[m] = (unsigned)((C / 4) * Tp + 32 * (q % KS) + 16 * hi);
  }
  for (unsigned run = blockIdx.x; run < nruns; run += gridDim.x) {
    const unsigned I = run / nchunks, ch = run % nchunks;
    const size_t i0 = (size_t)I * kD4I;
    unsigned jt0 = ch * chunk;
    const unsigned jt1 = jt0 + chunk < nJ ? jt0 + chunk : nJ;
    __syncthreads();   // the previous run's reads of LDS are done (and the table is in LDS)
    if (intra && jt0 < 4 * I) jt0 = 4 * I;   // tiles J < 4 I hold no pair with j > i (mica_nan_lower_kernel wrote their NaN)
    if (jt0 >= jt1) continue;
    if (tid < kD4I) {
      const size_t i = i0 + tid;
      s1t[tid] = S1[i < n1 ? i : n1 - 1];
      ok1[tid] = i < n1 && !flag1[i];
    }
    // the wave's sixteen columns of the first alignment, expanded once per run: row tile ii, packed row 32 ii + cl
    cmx_i4 areg[2][KS];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int R = 32 * ii + cl;
      const uint8_t* src = C1 + (i0 + 16 * w + R / 4) * (size_t)Tp + 16 * hi;   // (64 columns of padding behind the last one)
      const unsigned srow = (unsigned)(R % 4) * 0x01010101u;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const cmx_i4 raw = ks < nks ? *reinterpret_cast<const cmx_i4*>(src + 32 * ks) : cmx_i4{0x3f3f3f3f, 0x3f3f3f3f, 0x3f3f3f3f, 0x3f3f3f3f};
        areg[ii][ks] = d4_expand<8, 2>(raw, srow);
      }
    }
    cmx_i4 braw[SPT];
    double s2r = 0.0;
    int ok2r = 0;
    auto fetch = [&](unsigned jt) {
      const unsigned soff = jt * (unsigned)(kD4J * Tp);
#pragma unroll
      for (int m = 0; m < SPT; ++m)
        braw[m] = (w + 4 * m) % KS < nks ? __builtin_bit_cast(cmx_i4, __builtin_amdgcn_raw_buffer_load_b128(rc2, boff[m], soff, 0))
                                         : cmx_i4{0x3f3f3f3f, 0x3f3f3f3f, 0x3f3f3f3f, 0x3f3f3f3f};
      if (tid < kD4J) {
        const size_t j = (size_t)jt * kD4J + tid;
        s2r = S2[j < n2 ? j : n2 - 1];
        ok2r = j < n2 && !flag2[j < n2 ? j : n2 - 1];
      }
    };
    auto expand = [&](int buf) {
#pragma unroll
      for (int m = 0; m < SPT; ++m) ops[(buf * NQ + w + 4 * m) * 64 + lane] = d4_expand<16, 4>(braw[m], bsrow[m]);
      if (tid < kD4J) {
        s2t[kD4J * buf + tid] = s2r;
        ok2[kD4J * buf + tid] = ok2r;
      }
    };
    fetch(jt0);
    expand(0);
    int buf = 0;
    for (unsigned jt = jt0; jt < jt1; ++jt) {
      __syncthreads();   // this tile's operands and scalars (and, in the first tile, the run's) are in LDS
      const bool more = jt + 1 < jt1;
      if (more) fetch(jt + 1);
      cmx_i16v acc[2][2];
      {
        const cmx_i16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const cmx_i4* ob = ops + buf * NQ * 64 + lane;
        cmx_i4 nb0 = ob[0], nb1 = ob[KS * 64];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const cmx_i4 bb0 = nb0, bb1 = nb1;
          if (ks + 1 < KS) {
            nb0 = ob[(ks + 1) * 64];
            nb1 = ob[(KS + ks + 1) * 64];
          }
          asm volatile("" ::: "memory");
          acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[0][ks], bb0, ks ? acc[0][0] : zero, 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[1][ks], bb0, ks ? acc[1][0] : zero, 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[0][ks], bb1, ks ? acc[0][1] : zero, 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[1][ks], bb1, ks ? acc[1][1] : zero, 0, 0, 0);
        }
      }
      if (more) expand(buf ^ 1);
      // Register v of accumulator tile (ii, jj) is packed row 32 ii + 8 (v / 4) + v % 4 (+ 4 in the upper lane half), packed
      // column 32 jj + cl: registers 4 vq .. 4 vq + 3 are the four states of first-alignment column 8 ii + 2 vq + hi, lanes
      // 4 k .. 4 k + 3 the four states of second-alignment column 8 jj + k.
      const size_t j0 = (size_t)jt * kD4J;
      const int sel = cl & 3;            // lanes 0 / 1 of a quad store the pair of column tile 0 / 1
      const int b = 8 * (sel & 1) + (cl >> 2);
      const size_t j = j0 + b;
      const bool okj = sel < 2 && ok2[kD4J * buf + b] != 0;
      const double s2v = s2t[kD4J * buf + b];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int vq = 0; vq < 4; ++vq) {
          double ps[2];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            double v4[4];
#pragma unroll
            for (int vs = 0; vs < 4; ++vs)   // the accumulator is 8 m = the LDS address of f2[m]
              v4[vs] = *reinterpret_cast<const __attribute__((address_space(3))) double*>(static_cast<uintptr_t>((unsigned)acc[ii][jj][4 * vq + vs]));
            double s4 = (v4[0] + v4[1]) + (v4[2] + v4[3]);
            s4 += mica_dpp_f64<0xB1>(s4);   // quad_perm [1, 0, 3, 2]
            s4 += mica_dpp_f64<0x4E>(s4);   // quad_perm [2, 3, 0, 1]
            ps[jj] = s4;
          }
          const int al = 16 * w + 8 * ii + 2 * vq + hi;   // column of the workgroup's 64
          const size_t i = i0 + al;
          if (okj && ok1[al] != 0 && (!intra || j > i)) {
            const double sp = sel ? ps[1] : ps[0];
            mi[i * ldo + j] = lnT + (sp - s1t[al] - s2v) * invT;
            hj[i * ldo + j] = lnT - sp * invT;
          }
        }
      buf ^= 1;
    }
  }
}

size_t mica_dna4_lds_bytes(int T, int KS) {
  return (((size_t)(16 * T + 1) * 8 + 15) & ~(size_t)15) + (size_t)2 * 2 * KS * 64 * sizeof(cmx_i4) + (kD4I + 2 * kD4J) * sizeof(double) +
         (kD4I + 2 * kD4J) * sizeof(int);
}
template <int KS>
static hipError_t launch_mica_dna4_ks(int T, int Tp, const MicaWork* wk, size_t n1, size_t n2, int intra, double* d_mi, double* d_hj,
                                      size_t ldo, unsigned chunk, unsigned grid, hipStream_t stream) {
  const size_t lds = mica_dna4_lds_bytes(T, KS);
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mica_dna4_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const unsigned nJ = (unsigned)((n2 + kD4J - 1) / kD4J), nI = (unsigned)((n1 + kD4I - 1) / kD4I);
  const unsigned nchunks = (nJ + chunk - 1) / chunk, nruns = nI * nchunks;
  if (intra) hipLaunchKernelGGL(mica_nan_lower_kernel, dim3((unsigned)((n1 + 255) / 256), (unsigned)n1), dim3(256), 0, stream, n1, d_mi, d_hj, ldo);
  hipLaunchKernelGGL(mica_dna4_kernel<KS>, dim3(grid < nruns ? grid : nruns), dim3(256), lds, stream, T, Tp, wk->C1, n1, wk->flag1, wk->S1,
                     intra ? wk->C1 : wk->C2, n2, intra ? wk->flag1 : wk->flag2, intra ? wk->S1 : wk->S2, wk->ftab, intra, d_mi, d_hj, ldo, nJ,
                     chunk, nchunks, nruns);
  return hipGetLastError();
}
// nucleotides, Tp <= 256 (eight k-steps of operand registers; the whole weighted table in LDS), byte offsets within 31 bits
bool mica_dna4_serves(int A, int Tp, size_t n1, size_t n2) {
  return A == 4 && Tp <= 256 && (std::max(n1, n2) + kMicaCodePad) * (size_t)Tp < 0x7fffffffull;
}
hipError_t launch_mica_dna4(int T, const MicaWork* wk, size_t n1, size_t n2, int intra, double* d_mi, double* d_hj, size_t ldo,
                            hipStream_t stream) {
  const unsigned chunk = 16, grid = 512;   // 16 tiles of 64 x 16 columns per run; two workgroups per CU
  const int Tp = wk->Tp;
  if (Tp <= 64) return launch_mica_dna4_ks<2>(T, Tp, wk, n1, n2, intra, d_mi, d_hj, ldo, chunk, grid, stream);
  if (Tp <= 128) return launch_mica_dna4_ks<4>(T, Tp, wk, n1, n2, intra, d_mi, d_hj, ldo, chunk, grid, stream);
  return launch_mica_dna4_ks<8>(T, Tp, wk, n1, n2, intra, d_mi, d_hj, ldo, chunk, grid, stream);
}

size_t mica4_lds_bytes(int T, int KS, bool weighted) {
  const int M0 = 400 * T + 1 < kMicaLdsF2 ? 400 * T + 1 : kMicaLdsF2;
  return (((size_t)(weighted ? M0 + 1 : T + 1) * 8 + 15) & ~(size_t)15) + (size_t)2 * 2 * KS * 64 * sizeof(cmx_i4) + 16 * sizeof(double) + 16 * sizeof(unsigned) +
         (weighted ? 4 * kM4Corr * sizeof(double) + 4 * m4_qcap(KS) * sizeof(unsigned) : 0) + 4 * kM4FinBytes;
}

template <int KS, bool WEIGHTED>
static hipError_t launch_mica4_one(int T, int Tp, const MicaWork* wk, size_t n1, size_t n2, int intra, double* d_mi, double* d_hj,
                                   size_t ldo, unsigned nJ, unsigned chunk, unsigned nchunks, unsigned nruns, unsigned grid,
                                   hipStream_t stream) {
  const size_t lds = mica4_lds_bytes(T, KS, WEIGHTED);
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mica_mfma4_kernel<KS, WEIGHTED>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  // the second alignment's operand image of this instantiation (its two halves of wk->img2)
  cmx_i4* img = reinterpret_cast<cmx_i4*>(wk->img2) + (WEIGHTED ? (size_t)nJ * 2 * KS * 64 : 0);
  hipLaunchKernelGGL((mica4_image_kernel<KS, WEIGHTED>), dim3(nJ), dim3(256), 0, stream, Tp, intra ? wk->Cs1 : wk->Cs2, img);
  hipLaunchKernelGGL((mica_mfma4_kernel<KS, WEIGHTED>), dim3(grid < nruns ? grid : nruns), dim3(256), lds, stream, T, Tp, wk->Cs1, n1,
                     wk->info1, wk->Ss1, img, n2, intra ? wk->info1 : wk->info2, intra ? wk->Ss1 : wk->Ss2,
                     wk->order1, intra ? wk->order1 : wk->order2, wk->ftab, intra, d_mi, d_hj, ldo, nJ, chunk, nchunks, nruns,
                     wk->info1 + (n1 + 11) / 12 * 4, intra ? wk->info1 + (n1 + 11) / 12 * 4 : wk->info2 + (n2 + 11) / 12 * 4);
  return hipGetLastError();
}

template <int KS>
static hipError_t launch_mica4_ks(int T, int Tp, const MicaWork* wk, size_t n1, size_t n2, int intra, double* d_mi, double* d_hj,
                                  size_t ldo, unsigned chunk, unsigned grid, hipStream_t stream) {
  const unsigned nJ = (unsigned)((n2 + kM4J - 1) / kM4J), nI = (unsigned)((n1 + kM4I - 1) / kM4I);
  const unsigned nchunks = (nJ + chunk - 1) / chunk, nruns = nI * nchunks;
  // block info, padded to whole tiles (blocks past the end: all three columns "not served")
  const size_t nb1 = (size_t)nI * (kM4I / 3), nb2 = nJ;
  // (the info arrays have four words to spare behind the blocks of the four-wave tiling: the first holds "some column has unknowns")
  hipLaunchKernelGGL(mica_sort_columns_kernel, dim3(1), dim3(1024), 0, stream, wk->gap1, n1, wk->order1, wk->info1 + (n1 + 11) / 12 * 4);
  hipLaunchKernelGGL(mica_gather_columns_kernel, dim3((unsigned)(n1 + kMicaCodePad)), dim3(256), 0, stream, wk->order1, n1, Tp, wk->C1, wk->S1,
                     wk->Cs1, wk->Ss1);
  hipLaunchKernelGGL(mica_blockinfo_kernel, dim3((unsigned)((nb1 + 255) / 256)), dim3(256), 0, stream, wk->order1, wk->flag1, wk->gap1, n1, nb1,
                     wk->info1);
  if (!intra) {
    hipLaunchKernelGGL(mica_sort_columns_kernel, dim3(1), dim3(1024), 0, stream, wk->gap2, n2, wk->order2, wk->info2 + (n2 + 11) / 12 * 4);
    hipLaunchKernelGGL(mica_gather_columns_kernel, dim3((unsigned)(n2 + kMicaCodePad)), dim3(256), 0, stream, wk->order2, n2, Tp, wk->C2, wk->S2,
                       wk->Cs2, wk->Ss2);
    hipLaunchKernelGGL(mica_blockinfo_kernel, dim3((unsigned)((nb2 + 255) / 256)), dim3(256), 0, stream, wk->order2, wk->flag2, wk->gap2, n2, nb2,
                       wk->info2);
  } else {
    hipLaunchKernelGGL(mica_nan_lower_kernel, dim3((unsigned)((n1 + 255) / 256), (unsigned)n1), dim3(256), 0, stream, n1, d_mi, d_hj, ldo);
  }
  hipError_t e = launch_mica4_one<KS, false>(T, Tp, wk, n1, n2, intra, d_mi, d_hj, ldo, nJ, chunk, nchunks, nruns, grid, stream);
  if (e == hipSuccess) e = launch_mica4_one<KS, true>(T, Tp, wk, n1, n2, intra, d_mi, d_hj, ldo, nJ, chunk, nchunks, nruns, grid, stream);
  return e;
}

// proteins, Tp <= 256 (eight k-steps of operand registers), byte offsets within 31 bits; the caller serves the pairs with
// partial ambiguity codes
static int mica4_ksteps(int Tp) { return Tp <= 64 ? 2 : (Tp <= 128 ? 4 : (Tp <= 256 ? 8 : 16)); }
// bytes of MicaWork::img2: two images (plain, weighted) of 2 KS KB per tile of three columns of the second alignment
size_t mica4_image_bytes(int Tp, size_t n2) { return 2 * ((n2 + kM4J - 1) / kM4J) * (size_t)(2 * mica4_ksteps(Tp)) * 1024; }
bool mica4_serves(int A, int Tp, size_t n1, size_t n2) {
  return A == 20 && Tp <= 512 && (std::max(n1, n2) + kMicaCodePad) * (size_t)Tp < 0x7fffffffull && mica4_image_bytes(Tp, n2) / 2 < 0x7fffffffull;
}

hipError_t launch_mica4(int T, const MicaWork* wk, size_t n1, size_t n2, int intra, double* d_mi, double* d_hj, size_t ldo,
                        hipStream_t stream) {
  static const unsigned chunk = [] {
    const char* e = getenv("CMX_MICA4_CHUNK");
    const unsigned c = e ? (unsigned)atoi(e) : 32u;
    return c < 1 ? 1u : (c > kM4MaxChunk ? kM4MaxChunk : c);
  }();
  // two workgroups per CU at a time (254 registers): a multiple of 512
  static const unsigned grid = [] { const char* e = getenv("CMX_MICA4_GRID"); return e ? (unsigned)atoi(e) : 1024u; }();
  const int Tp = wk->Tp;
  if (Tp <= 64) return launch_mica4_ks<2>(T, Tp, wk, n1, n2, intra, d_mi, d_hj, ldo, chunk, grid, stream);
  if (Tp <= 128) return launch_mica4_ks<4>(T, Tp, wk, n1, n2, intra, d_mi, d_hj, ldo, chunk, grid, stream);
  if (Tp <= 256) return launch_mica4_ks<8>(T, Tp, wk, n1, n2, intra, d_mi, d_hj, ldo, chunk, grid, stream);
  // 257 .. 512 taxa: sixteen k-steps of operand registers, one workgroup per CU (the pipelined tile loop keeps a lone wave's
  // matrix core and vector unit busy together)
  return launch_mica4_ks<16>(T, Tp, wk, n1, n2, intra, d_mi, d_hj, ldo, chunk, grid, stream);
}

}  // namespace cmx
