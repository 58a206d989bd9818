// Shared host/device declarations of the MI355X engine (internal; the public surface is include/comap_mi355x.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cmx {

constexpr int kWave = 64;           // gfx950 wavefront
constexpr int kWavesPerBlock = 4;   // mapping kernels: 4 independent waves per 256-thread workgroup
#ifndef CMX_WAVES_PER_SIMD
#define CMX_WAVES_PER_SIMD 2       // resident mapping waves per SIMD (1: 512-register budget, 2: 256)
#endif

// Device-resident model + tree program.  All pointers are device pointers.
struct DevModel {
  int S, C, K, nn, B, T, NI, NV, root;  // NI internal nodes (slots), NV of them visited by the traversal
  // tree program (wave-uniform, read through the scalar cache)
  const int* int_post;     // [NI]  internal nodes in post-order, root last
  const int* first_child;  // [nn]
  const int* next_sib;     // [nn]
  const int* taxon_of;     // [nn]  alignment row of a leaf, -1 for internal nodes
  const int* slot;         // [nn]  internal nodes: 0..NI-1 (root = NI-1); leaves: -1
  const int* parent;       // [nn]
  // per (class, internal node): 4x4-block-packed matrices for the scalar-operand matvec
  const double* MAT;       // packed matrices: P_node [C][NI][S*S] at 0, (P_node o N^k_node) [C][NI][K][S*S] at joff
  size_t joff;
  // matrix products of one class pass in program order: bit 31 set = (P o N^k) with index slot*K + k, else P[slot]
  const int* msched;
  int nmv;
  const int* nrec;         // [NV][32] per-visited-node records (enum REC_* in cmx_kernels.hip)
  // per (class, taxon): transposed matrices for the per-lane leaf gather, [z][x] = M[x][z]
  const double* LPT;       // [C][T][S][S]
  const double* LJT;       // [C][K][T][S][S]
  // simulator: running sums of the rows of P, [C][nn][S(x)][S]
  const double* CP;
  const double* pi;        // [S]
  const double* rates;     // [C]
  const double* probs;     // [C]
  const double* cum_pi;    // [S]
  const double* cum_probs; // [C]
  // workspace-load schedule of one rate-class pass (see CMX_POP in cmx_kernels.hip)
  const int* ldsched;      // [nloads] bit31 prefetchable, bit30 array (0 D, 1 U), low 24 bits slot
  int nloads;
};

// Per-wave workspace strides (in elements); every wave owns one slice of each array.
struct Workspace {
  double* D;        // [waves][NI][S][64]   inside (post-order) conditional likelihoods of internal nodes
  double* U;        // [waves][NI][S][64]   outside messages arriving at internal nodes
  double* cnt;      // [waves][2][B*K][64]  final counts of the wave's sites (two batches for the null)
  double* part;     // [waves][C][B*K][64]  per-class joint counts, summed in class order at the end
  uint8_t* st;      // [waves][nn][64]      simulated states
  uint8_t* aln;     // [waves][T][64]       simulated leaf states
  int waves;
};

enum MapMode { kModeObserved = 0, kModeNull = 1 };

struct MapArgs {
  DevModel m;
  Workspace ws;
  // observed mode
  const uint8_t* aln;      // [T][ld]
  size_t ld;
  size_t nsites;           // observed: sites; null: (rep_end-rep_begin)*rep_ram null pairs
  const uint32_t* masks;   // ambiguity masks (may be null when all codes < S)
  int codes_in_lds;        // leaf symbols of a wave's sites staged in LDS (fits when map_lds_bytes <= 80 KiB)
  double* counts;          // [B*K][ldc] or null
  size_t ldc;
  double* logL;            // [nsites] or null
  double* post_rate;
  int32_t* rate_class;
  double* norm;
  // null mode
  int stat_kind;
  double stat_param;       // discrete-MI threshold
  uint64_t seed;
  size_t rep_begin, rep_ram;
  const uint8_t* supplied; // [nrep][2][T][rep_ram] or null
  double* null_stat;       // [nsites]
  int32_t* null_rcmin;
  double* null_prmin;
  double* null_nmin;
};

// launchers (cmx_kernels.hip)
size_t map_lds_bytes(int S, int T, bool codes_in_lds);
hipError_t launch_map(const MapArgs& a, int mode, int grid_blocks, hipStream_t stream);
hipError_t launch_simulate(const DevModel& m, uint64_t seed, uint64_t g0, size_t n, uint8_t* d_aln, size_t ld,
                           int32_t* d_classes, uint8_t* d_states /*[nn][ld]*/, hipStream_t stream);
hipError_t launch_pair_prep(int kind, double param, const double* d_counts, size_t n, size_t ldc, int B, int K,
                            double* d_X, size_t ldx, int Bp, double* d_s, double* d_r, hipStream_t stream);
hipError_t launch_pair_gram(int kind, int B, int Bp, const double* d_X1, const double* d_s1, const double* d_r1,
                            size_t n1, size_t ldx1, const double* d_X2, const double* d_s2, const double* d_r2,
                            size_t n2, size_t ldx2, int intra, double* d_out, size_t ldo, hipStream_t stream);
hipError_t launch_max_reduce(const double* d_x, size_t n, double* d_out, hipStream_t stream);
hipError_t launch_null_classify(const double* d_stat, const double* d_nmin, size_t nnull, const double* d_maxnorm,
                                int nclasses, uint32_t* d_cls, uint32_t* d_hist, hipStream_t stream);
hipError_t launch_pvalues(const double* d_stat, size_t ldo, const double* d_norms, size_t n, const double* d_maxnorm,
                          int nclasses, const double* d_sorted, const uint32_t* d_hist, double* d_pvalue,
                          int32_t* d_nsim, hipStream_t stream);
hipError_t sort_null_by_class(void* d_tmp, size_t& tmp_bytes, double* d_stat_in, double* d_stat_tmp, uint32_t* d_cls_in,
                              uint32_t* d_cls_tmp, size_t n, hipStream_t stream);
hipError_t launch_mi_columns(int A, int T, const uint32_t* d_masks, const uint8_t* d_aln1, size_t n1, size_t ld1,
                             const uint8_t* d_aln2, size_t n2, size_t ld2, int intra, double* d_mi, double* d_hj,
                             size_t ldo, double* d_h1, double* d_h2, hipStream_t stream);

}  // namespace cmx
