// Shared host/device declarations of the MI355X engine (internal; the public surface is include/comap_mi355x.h).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/comap_mi355x.h"
#include <stddef.h>
#include <stdint.h>

namespace cmx {

constexpr int kWave = 64;           // gfx950 wavefront
constexpr int kWavesPerBlock = 4;   // mapping kernels: 4 independent waves per 256-thread workgroup
// Ambiguous symbols (alignment codes >= S) are served from extra rows S .. S+A-1 appended to every transposed leaf
// operator: row S+a = sum of the rows of the states compatible with ambiguity id a (filled per call from the caller's
// mask table, default "every state").  A = 12 for nucleotides (IUPAC + gap), 4 for proteins (B, Z, J, X/gap).
constexpr int max_ambig(int S) { return S == 4 ? 12 : 4; }
// A transposed leaf operator is read row by row, the row named by a site's symbol: sixteen sites of a lane group read
// sixteen rows at once.  With rows of S doubles (40 dwords for S = 20, 32 for the class-fused 16) rows 8 (2) apart fall
// on the same LDS banks -- 61 % of the mapping kernel's LDS cycles were bank conflicts.  One double of padding per row
// (42 / 34 dwords) moves the period to 32 rows: no two rows of an operator share a bank.
constexpr int leaf_row_stride(int S) { return S + 1; }             // doubles per row of a transposed leaf operator
constexpr int mat_unit(int S) { return (S + max_ambig(S)) * leaf_row_stride(S); }   // doubles per device matrix
#ifndef CMX_WAVES_PER_SIMD
#define CMX_WAVES_PER_SIMD 2       // resident mapping waves per SIMD for 20 states (1: 512-register budget, 2: 256)
#endif
#ifndef CMX_WAVES_PER_SIMD_S4
#define CMX_WAVES_PER_SIMD_S4 3    // nucleotide vectors are 8 registers: the kernel is latency-bound, more waves help
#endif
#ifndef CMX_NG
#define CMX_NG 4                   // site groups of 16 per mapping wave for >= 16 device states: 4 (64 sites) or 2 (32 sites)
#endif
constexpr int map_ng(int S) { return S >= 16 ? CMX_NG : 4; }
constexpr int map_sites_per_wave(int S) { return 16 * map_ng(S); }
// (three waves per SIMD were tried for the 16-state class-fused nucleotide layout -- vectors of 32 registers: at 168
// registers the kernel spills 213 of them and the cfg 4 launch went from 7.7 to 10.7 ms)
constexpr int map_waves_per_simd(int S) { return S == 4 ? CMX_WAVES_PER_SIMD_S4 : ((map_ng(S) == 2 || map_ng(S) == 3) ? 3 : CMX_WAVES_PER_SIMD); }

// Device-resident model + tree program.  All pointers are device pointers.
struct DevModel {
  int S, C, K, nn, B, T, NI, NIW, NV, root;  // NI internal nodes (operator slots), NIW workspace slots (+ pseudo nodes), NV visited nodes
  // S, C are the DEVICE view: with fuse > 1 (nucleotides, >= 4 rate classes) S = S0 * fuse concatenated per-class states
  // and C = ceil(C0 / fuse) passes; S0, C0 are the model's own state and class counts (simulator, rates, probs, pi)
  int S0, C0, fuse;
  // tree (wave-uniform, read through the scalar cache; simulator only)
  const int* taxon_of;     // [nn]  alignment row of a leaf, -1 for internal nodes
  const int* parent;       // [nn]
  // matrices, [C][MC][mat_unit(S)]: per class a block of packed P | packed (P o N^k) | leaf P^T | leaf (P o N^k)^T
  // (cmx_host_model.cpp); a matrix use DMAs mat_unit(S)*8 bytes from MAT + (class*MC + index)*mat_unit(S) into LDS
  double* MAT;
  int MC;
  // tree walk of one rate-class pass (cmx_walk.h; built and checked by cmx_host_model.cpp)
  const int* nrec;         // [NV][16] per-visited-node records
  const int* msched;       // operator uses in program order: pairs (element offset in the class block, taxon or -1),
                           // followed by copies of its first two pairs (the op two ahead is read without a wrap test)
  int nmv;                 // number of pairs
  const int* msched_r;     // the cherry-table walk's stream (class-fused nucleotide models, resolved alignments: the null), or null
  int nmv_r;
  const int* ldsched;      // [nloads + 2] workspace loads: bit 31 prefetchable, bit 30 array (0 M, 1 U), low 24 bits slot
  // simulator: running sums of the rows of P, [C][nn][S(x)][S], and a 32-entry guide table per row (see draw_guided)
  const double* CP;
  const uint8_t* CPG;      // [C][nn][S(x)][32]
  const int* simg;         // [nsimg][16] groups of four nodes of equal depth: nodes | parents | taxa (or -1) | pad
  int nsimg;
  const int* simord;       // [nn - 1] the non-root nodes level by level: a node's parent was drawn a whole level earlier
  // continuous-rate simulator: eigensystems of the generators [NM][S*S] / [NM][S], generator and length of each branch
  const double *eigV, *eigVi, *eigLam;
  const int* model_of;     // [B]
  const double* blen;      // [nn]
  const double* pi;        // [S]
  const double* rates;     // [C]
  const double* probs;     // [C]
  const double* cum_pi;    // [S]
  const double* cum_probs; // [C]
};

// Per-wave workspace strides (in elements); every wave owns one slice of each array.
struct Workspace {
  double* D;        // [waves][NIW][S][64]  messages M_n = P_n D_n of internal nodes (inside pass)
  double* U;        // [waves][NIW][S][64]  outside messages arriving at internal nodes
  double* cnt;      // [waves][2][B*K][64]  final counts of the wave's sites (two batches for the null)
  double* part;     // [waves][C][B*K][64]  per-class joint counts, summed in class order at the end
  uint8_t* st;      // [waves][nn][64]      simulated states
  uint8_t* aln;     // [waves][T][64]       simulated leaf states
  int waves;
};

// kModeObservedSplit: one (site block, rate class) per wave-task for small alignments (an alignment of 2 000 sites is
// 32 site blocks: one task per wave would be four class passes of pure latency), classes summed by map_finalize_kernel
enum MapMode { kModeObserved = 0, kModeNull = 1, kModeObservedSplit = 2 };

struct MapArgs {
  DevModel m;
  Workspace ws;
  // observed mode
  const uint8_t* aln;      // [T][ld]
  size_t ld;
  size_t nsites;           // observed: sites; null: (rep_end-rep_begin)*rep_ram null pairs
  double* counts;          // [B*K][ldc] or null
  size_t ldc;
  double* logL;            // [nsites] or null
  double* post_rate;
  int32_t* rate_class;
  double* norm;
  // class-split observed mode: [nblocks][C][B*K][64] per-class counts, [2][nblocks][C][64] p_c L_c and r_c p_c L_c
  double* split_part;
  double* split_lc;
  int split_sites;         // sites per wave-task of the class-split launch: map_sites_per_wave(S), or 16 for small alignments
  int lds_per_wave;        // bytes of dynamic LDS per wave (set by launch_map)
  // null mode
  int stat_kind;
  double stat_param;       // discrete-MI threshold
  const double* stat_mean; // CorrectedCorrelation: [2][B] mean vectors of the two operands (device), else null
  uint64_t seed;
  size_t rep_begin, rep_ram;
  const uint8_t* supplied; // [nrep][2][T][rep_ram] or null
  double* null_stat;       // [nsites]
  int32_t* null_rcmin;
  double* null_prmin;
  double* null_nmin;
};

// launchers (cmx_kernels.hip)
int map_lds_per_wave(int S, int nn, int mode);
hipError_t launch_map(const MapArgs& a, int mode, int grid_blocks, hipStream_t stream);
hipError_t launch_map_finalize(const MapArgs& a, hipStream_t stream);
hipError_t launch_simulate_blocked(const DevModel& m, uint64_t seed, uint64_t g0, size_t nsites, size_t blk, uint8_t* d_aln,
                                   uint8_t* d_states, size_t chunk, hipStream_t stream);
// fills rows S.. of every leaf operator from d_masks[S .. S+max_ambig(S)) (null: every state compatible)
hipError_t launch_extend_leaf_rows(const DevModel& m, const uint32_t* d_masks, hipStream_t stream);
hipError_t launch_pair_diag(int kind, double param, int B, int K, const double* c1, size_t ld1, const double* c2, size_t ld2,
                            size_t n, const int32_t* rc1, const int32_t* rc2, const double* pr1, const double* pr2,
                            const double* nm1, const double* nm2, double* stat, int32_t* rcmin, double* prmin, double* nmin,
                            const double* d_mean, hipStream_t stream);
hipError_t launch_group_stats(int kind, double param, int B, int K, const double* d_counts, size_t ld, const int64_t* d_offsets,
                              const int32_t* d_sites, size_t ngroups, double* d_out, const double* d_mean, hipStream_t stream);
// DiscreteMI with a bounds vector (cmx_stat_mi.hip): class words [B][ldx] (class | marginal count << 16) + per-site
// out-of-range flags; all-pairs block, diagonal pairs, groups
hipError_t launch_mi_classify(const double* d_counts, size_t n, size_t ldc, int B, int K, const double* d_bounds, int nb,
                              uint32_t* d_cls, size_t ldx, uint8_t* d_bad, hipStream_t stream);
hipError_t launch_mi_pairs_block(int B, const uint32_t* d_cls1, const uint8_t* d_bad1, size_t nrows, size_t ld1, const uint32_t* d_cls2,
                                 const uint8_t* d_bad2, size_t n2, size_t ld2, int intra, double* d_out, size_t ldo, size_t irow0,
                                 hipStream_t stream);
hipError_t launch_mi_pairs_diag(int B, const uint32_t* d_cls1, const uint8_t* d_bad1, size_t ld1, const uint32_t* d_cls2,
                                const uint8_t* d_bad2, size_t ld2, size_t n, double* d_out, hipStream_t stream);
hipError_t launch_mi_group(int B, const uint32_t* d_cls, const uint8_t* d_bad, size_t ld, const int64_t* d_offsets,
                           const int32_t* d_sites, size_t ngroups, double* d_out, hipStream_t stream);
// nijt.average = no (cmx_variants.hip): the no-averaging mapping as plain kernels over a global scratch
// mode: which LegacySubstitutionMappingTools function (CoETools.cpp:395-405)
// kVariantJoint: the default mapping (computeSubstitutionVectors: averaged, joint) for the alphabets the matrix-core walk
// does not serve
enum { kVariantNoAvg = 0 /* NoAveraging */, kVariantMarginal = 1 /* Marginal */, kVariantNoAvgMarginal = 2 /* NoAveragingMarginal */,
       kVariantJoint = 3 };
struct NoAvgArgs {
  int S, C, K, nn, B, root, mode;
  int Sreal;              // states of the alphabet; < S on the plain path, whose operators are padded with zeros to S = 64
  const double* PN;       // [C][B][K][S*S] joint counts P o N^k (kVariantJoint)
  const double* rates;    // [C] (site scalars)
  double *logL, *post_rate;   // [ld...] per site, optional: likelihood, posterior rate, rate class (plain path)
  int32_t* rate_class;
  const int *first_child, *next_sib, *taxon_of, *parent;
  const double* P;        // [C][B][S*S] row-major transition matrices
  const double* N1;       // [B][K][S*S] conditional counts at the branch length itself
  const double* NC;       // [C][B][K][S*S] conditional counts at r_c t_b (kVariantMarginal)
  const double *pi, *probs;
  const uint32_t* masks;  // compatibility masks of the codes >= S (NULL: every state)
  const uint8_t* aln;
  size_t ld, site0, nsites, chunk;
  double *D, *M, *U, *Up;  // [C][nn][S][chunk]
  double* counts;          // [B*K][ldc]
  size_t ldc;
};
size_t noavg_scratch_doubles(int S, int C, int nn, size_t chunk);
hipError_t launch_map_noavg(NoAvgArgs a, size_t nsites_total, double* scratch, double* d_norm, hipStream_t stream);
// Mica post-processing (cmx_mica_post.hip)
hipError_t launch_mica_average(const double* d_mi, size_t n, size_t ld, double* d_avg, double* d_full, hipStream_t stream);
hipError_t launch_mica_zscore(int which, const double* d_mi, size_t n, size_t ld, const double* d_avg, const double* d_full,
                              const double* d_key, double* d_stat, double* d_outkey, hipStream_t stream);
int mica_perm_max_taxa();
hipError_t launch_mica_colcount(const uint8_t* d_aln, int T, size_t n, size_t ld, int A, const uint8_t* d_emap, uint16_t* d_cnt,
                                uint16_t* d_ext, uint8_t* d_hasamb, int* d_bad, hipStream_t stream);
hipError_t launch_mica_colorder(const uint8_t* d_aln, int T, size_t n, size_t ld, int A, const uint8_t* d_emap, const uint16_t* d_ext,
                                uint16_t* d_order, hipStream_t stream);
bool mica_perm_opening_fits(int T, int A);   // the four-pairs-per-wave opening pass fits the LDS (else nperm must be preset to -1)
size_t mica_perm_general_lds(int T, int A, int namb);
hipError_t launch_mica_perm_general(const uint8_t* d_aln, int T, size_t n, size_t ld, int A, const uint8_t* d_emap,
                                    const uint16_t* d_ext, const uint16_t* d_order, const uint8_t* d_hasamb, const uint32_t* d_emask,
                                    const uint32_t* d_ewgt, const long long* d_F, uint32_t L, int namb, uint32_t max_perm, uint64_t seed,
                                    size_t pair_begin, size_t pair_end, double* d_pvalue, int32_t* d_nperm, int cu_count,
                                    hipStream_t stream);
hipError_t launch_mica_perm(const uint8_t* d_aln, int T, size_t n, size_t ld, int A, const uint16_t* d_colcnt,
                            const uint8_t* d_hasamb, const long long* d_dF, bool nperm_preset, uint32_t max_perm, uint64_t seed,
                            size_t pair_begin, size_t pair_end, double* d_pvalue, int32_t* d_nperm, int cu_count, hipStream_t stream);
// clustering (cmx_cluster.hip)
size_t hclust_lds_bytes(int n);
size_t cluster_props_lds_bytes(int n);
hipError_t launch_dist_finish(int dist_kind, double* d_D, size_t n, size_t ld, size_t mat_stride, size_t batch,
                              hipStream_t stream);
hipError_t launch_hclust(int linkage, double* d_D, size_t n, size_t ld, size_t mat_stride, size_t batch, double* d_rmin,
                         int* d_nn, int32_t* d_merge, double* d_dmax, int32_t* d_size, hipStream_t stream);
hipError_t launch_cluster_props(int dist_kind, int n, int B, int K, size_t batch, const int32_t* d_merge, const double* d_dmax,
                                const double* d_norm, const double* d_counts, size_t ldc, size_t site_stride, double* d_sigma,
                                double* d_stat, double* d_nmin, hipStream_t stream);
hipError_t launch_simulate(const DevModel& m, uint64_t seed, uint64_t g0, size_t n, uint8_t* d_aln, size_t ld,
                           int32_t* d_classes, uint8_t* d_states, hipStream_t stream, size_t rep_ram = 0, uint64_t gstep = 0);
hipError_t launch_simulate_continuous(const DevModel& m, uint64_t seed, uint64_t g0, size_t n, double alpha, double p_inv,
                                      uint8_t* d_aln, size_t ld, double* d_rates, uint8_t* d_states, hipStream_t stream);
hipError_t launch_pair_prep(int kind, double param, const double* d_counts, size_t n, size_t ldc, int B, int K,
                            double* d_X, size_t ldx, int Bp, double* d_s, double* d_r, const double* d_mvec,
                            hipStream_t stream, size_t blk = 0);
hipError_t launch_pair_gram(int kind, int B, int Bp, const double* d_X1, const double* d_s1, const double* d_r1,
                            size_t n1, size_t ldx1, const double* d_X2, const double* d_s2, const double* d_r2,
                            size_t n2, size_t ldx2, int intra, double* d_out, size_t ldo, hipStream_t stream,
                            size_t nblk = 1, size_t zsite = 0, size_t zout = 0, size_t zx = 0, size_t irow0 = 0);
hipError_t launch_max_reduce(const double* d_x, size_t n, double* d_out, hipStream_t stream);
hipError_t launch_null_classify(const double* d_stat, const double* d_nmin, size_t nnull, const double* d_maxnorm,
                                int nclasses, uint32_t* d_cls, uint32_t* d_hist, hipStream_t stream);
// The null distribution prepared for p-value lookups: statistics sorted by (class, value), and per class a table of
// bins of equal width in the statistic's value, one bin per kNullBinSize sorted values: bins[b] = index of the first value
// of the class that falls into bin b or later.  A lookup computes its bin and searches the few values inside it.
constexpr int kNullBinShift = 3, kNullBinSize = 1 << kNullBinShift;
constexpr int kPairRowSegs = 8;   // waves per row in the pair-row passes (pair_rows_kernel)
struct NullClass {
  uint32_t off, ns;         // the class's stretch of `sorted`
  uint32_t nb, boff;        // number of bins (>= 1), and where its nb + 1 entries start in `bins`
  double lo, scale;         // bin of v = clamp(floor((v - lo) * scale), 0, nb - 1)
};
struct NullTable {
  const double* sorted;     // [nnull] ascending inside each class, classes in order
  const NullClass* cls;     // [nclasses]
  const uint32_t* bins;     // [(nnull >> kNullBinShift) + 2 * nclasses + 2]
  const double* maxnorm;    // upper bound of the Domain of the norms
  int nclasses;
};
hipError_t launch_null_index(const double* d_sorted, const uint32_t* d_hist, int nclasses, size_t nnull, NullClass* d_cls,
                             uint32_t* d_bins, hipStream_t stream);
hipError_t launch_pvalues(const double* d_stat, size_t ldo, const double* d_norms, size_t n, const NullTable& nt, double* d_pvalue,
                          int32_t* d_nsim, hipStream_t stream, size_t irow0 = 0, size_t nrows = 0);
hipError_t sort_null_by_class(void* d_tmp, size_t& tmp_bytes, double* d_stat_in, double* d_stat_tmp, uint32_t* d_cls_in,
                              uint32_t* d_cls_tmp, size_t n, hipStream_t stream);
hipError_t launch_pair_rows(const double* d_stat, size_t ldo, const double* d_pvalue, const int32_t* d_nsim, size_t n,
                            const int32_t* d_rc, const double* d_pr, const double* d_norm, const cmx_pair_filters& f,
                            unsigned long long* d_rowcount /*[n + 1]*/, void* d_tmp, size_t& tmp_bytes, cmx_pair_row* d_rows,
                            size_t capacity, unsigned long long* d_count, hipStream_t stream, size_t irow0 = 0, size_t nrows = 0,
                            const unsigned long long* d_base = nullptr, const NullTable* d_inline_null = nullptr);
hipError_t launch_pair_compact(const double* d_stat, size_t ldo, size_t n, const double* d_norm, const NullTable* nt, cmx_pair_compact* d_out,
                               size_t capacity, hipStream_t stream, size_t irow0, size_t nrows, size_t row_begin);
hipError_t launch_inter_rows(const double* d_stat, size_t ldo, size_t n2, const int32_t* d_rc1, const double* d_pr1, const double* d_nm1,
                             const int32_t* d_rc2, const double* d_pr2, const double* d_nm2, const cmx_inter_filters& f,
                             unsigned long long* d_rowcount, void* d_tmp, size_t& tmp_bytes, cmx_pair_row* d_rows, size_t capacity,
                             unsigned long long* d_count, hipStream_t stream, size_t irow0, size_t nrows,
                             const unsigned long long* d_base);
hipError_t launch_mi_pairs(int A, int T, const uint32_t* d_masks, const uint8_t* d_aln1, size_t ld1, const uint8_t* d_aln2,
                           size_t ld2, const int64_t* d_idx1, const int64_t* d_idx2, size_t npairs, double* d_mi,
                           double* d_hj, hipStream_t stream);
// scratch of the MFMA Mica path (all device pointers; H1 null = LDS-table kernel only)
constexpr int kMicaLdsF2 = 4096;   // entries of f2 the weighted four-wave kernel keeps in LDS (m < 4096: cells of up to ten taxa)
constexpr int kMicaCodePad = 64;   // columns of "no row" symbols behind the last column of C1 / C2 (the four-wave kernels read whole tiles: 12 / 64 columns)
struct MicaWork {
  int8_t *H1, *H2;         // one-hot [n][32][Tp] int8 (one-column-per-tile kernel)
  uint8_t *C1, *C2;        // [n + kMicaCodePad][Tp] one-hot row of each taxon (state, A = unknown, 255 = none): the packed protein kernel's operands
  uint8_t *flag1, *flag2;  // [n] column has ambiguous symbols other than "unknown" (-> LDS-table kernel)
  uint8_t *gap1, *gap2;    // [n] column has unknowns (gap / X / N: compatible with every state; handled on the matrix cores)
  double *S1, *S2;         // [n] sum_a f(count_a)
  double* ftab;            // [T + 1] c ln c, then [A*A*T + 1] f2[m] = (m / A^2) ln(m / A^2) (pairs with unknowns), then 0 and f2[M0 ..] again
  int* anyflag;            // some column of either alignment has ambiguous symbols
  unsigned *info1, *info2;   // per block of three SORTED columns: not-served and has-unknowns bits (cmx_mica4.hip; NULL: not used)
  unsigned *order1, *order2; // [n] original column of a sorted position (columns without unknowns first, stable)
  uint8_t *Cs1, *Cs2;        // [n + kMicaCodePad][Tp] symbol bytes in sorted order
  double *Ss1, *Ss2;         // [n] column sums in sorted order
  void* img2;                // mica4_image_bytes(Tp, n2): the second alignment's expanded operands by tile (cmx_mica4.hip)
  int Tp;                  // T rounded up to a multiple of 32 (taxa per MFMA step)
};
bool mica_needs_onehot(int A, int Tp);   // whether launch_mi_columns reads MicaWork::H1 / H2 for this alphabet
// cmx_mica4.hip: the four-wave protein kernel (unknowns included; partial ambiguity codes are not served)
bool mica4_serves(int A, int Tp, size_t n1, size_t n2);
size_t mica4_image_bytes(int Tp, size_t n2);
hipError_t launch_mica4(int T, const MicaWork* wk, size_t n1, size_t n2, int intra, double* d_mi, double* d_hj, size_t ldo,
                        hipStream_t stream);
// the four-wave nucleotide kernel (unknowns included; partial ambiguity codes are not served)
bool mica_dna4_serves(int A, int Tp, size_t n1, size_t n2);
hipError_t launch_mica_dna4(int T, const MicaWork* wk, size_t n1, size_t n2, int intra, double* d_mi, double* d_hj, size_t ldo,
                            hipStream_t stream);
hipError_t launch_mi_columns(int A, int T, const uint32_t* d_masks, const uint8_t* d_aln1, size_t n1, size_t ld1,
                             const uint8_t* d_aln2, size_t n2, size_t ld2, int intra, double* d_mi, double* d_hj,
                             size_t ldo, double* d_h1, double* d_h2, const MicaWork* work, hipStream_t stream);

}  // namespace cmx
