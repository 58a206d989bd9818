// Nucleotide (4-state) mapping path: host program + device arguments (internal).
//
// Why a second mapping kernel.  The 20-state kernel stages SxS operators in LDS and runs products on the
// matrix cores; for S = 4 an operator is 128 bytes = 32 SGPRs, a message is 4 doubles = 8 VGPRs, and that machinery spends
// 20 scalar + vector instructions per matrix instruction (cfg 4: 0.13 of the fp64 roof, 18x the algorithmic bytes).
// Here: lane = site; the 4x4 operators of a branch arrive through the scalar cache (s_load_dwordx16) and are applied with
// v_fma_f64 taking SGPR operands -- no LDS, no cross-lane traffic; and the per-node messages, which made the old kernel
// stream 80 KB per site through HBM, never leave the register file: the tree's internal nodes are cut into connected
// BLOCKS of <= NB nodes whose messages live in VGPR arrays addressed with s_set_gpr_idx (wave-uniform slot index).
//   phase 1 (inside):  per rate class, blocks bottom-up; only the message of a block's ROOT goes to HBM.
//   phase 2 (outside): blocks top-down; per block and class the block's inside messages are RECOMPUTED into the
//                      registers (20 FMAs per node, cheaper than any memory round trip), then the outside pass of the
//                      block runs from them; only the outside message of a lower block's root goes to HBM.
// Counts of a branch are accumulated over the rate classes in the per-wave count rows (read-modify-write of a row that
// the previous class of the same block visit left in L2), already weighted with p_c / L_site, so no per-class partial
// counts exist.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "cmx_host_model.h"

namespace cmx {

// child / source kinds of a visit record
enum { NK_LEAF = 0, NK_SLOT = 1, NK_HBM = 2, NK_NONE = 3 };
// inside record, 8 ints
enum { NI_FLAGS = 0, NI_A = 1, NI_ATX = 2, NI_B = 3, NI_BTX = 4, NI_OP = 5, NI_DST = 6, NI_PAD = 7 };
// flags of an inside record: bits 0-1 kind of child A, 2-3 kind of child B, 4 pseudo (no operator), 5 tree root,
// 6 block root (destination = HBM root slot NI_DST, else register slot NI_DST)
enum { NF_PSEUDO = 16, NF_ROOT = 32, NF_BLOCKROOT = 64 };
// outside record, 16 ints
enum { NO_FLAGS = 0, NO_USRC = 1, NO_A = 2, NO_ATX = 3, NO_B = 4, NO_BTX = 5, NO_OPJ = 6, NO_OPP = 7, NO_ROW = 8,
       NO_ADISP = 9, NO_AROW = 10, NO_BDISP = 11, NO_BROW = 12 };
// flags of an outside record: bits 0-1 kind of child A, 2-3 kind of child B, 4 pseudo, 5 tree root, 6 block root (U comes
// from HBM root slot NO_USRC, or is pi at the tree root; else from register slot NO_USRC)
// child disposal: LEAF -> counts of the leaf branch (NO_xDISP = operator index of its first count operator, NO_xROW its
// first count row); SLOT -> U into register slot NO_xDISP; HBM -> U into HBM root slot NO_xDISP

struct NucProgram {
  int NB = 0;                 // block capacity (register slots)
  int C = 0, K = 0, B = 0, T = 0;
  int nblocks = 0, nroots = 0, nops = 0;
  std::vector<int> blk;       // [nblocks][4]: first inside record, inside records, first outside record, outside records;
                              // blocks in phase-1 order (bottom-up); phase 2 walks them backwards
  std::vector<int> irec;      // [..][8]
  std::vector<int> orec;      // [..][16]
  std::vector<double> ops;    // [C][nops][16] row-major 4x4: branch b: P at b*(K+1), P o N^k at b*(K+1) + 1 + k
  // per class pass, for flop / traffic accounting
  size_t n_apply = 0;         // 4x4 operator applications (inside of phase 1 + recompute + outside, leaves included)
  size_t n_root_loads = 0, n_root_stores = 0;   // 32-byte-per-site messages through HBM
};

// builds the program for a 4-state model; empty string on success
std::string build_nuc_program(const HostModel& hm, int NB, NucProgram* out);
// runs the program numerically on the host for one random site and compares likelihood and every count with a direct
// pruning computation (empty string when they agree)
std::string verify_nuc_program(const HostModel& hm, const NucProgram& np);

// ---- device side
struct NucDev {
  int C, K, B, T, nblocks, nroots, nops, NB;
  const double* ops;
  const int* blk;
  const int* irec;
  const int* orec;
  const double *pi, *rates, *probs;
};
struct NucWs {
  double* WM;    // [waves][C][nroots][64][4]  inside messages of block roots
  double* WU;    // [waves][C][nroots][64][4]  outside messages of block roots
  double* cnt;   // [waves][2][B*K][64]        final counts of the wave's sites (two batches for the null)
  int waves;
};
struct NucArgs {
  NucDev m;
  NucWs ws;
  const uint8_t* aln;      // observed: [T][ld]
  size_t ld, nsites;
  const uint32_t* masks;   // compatibility masks of the codes >= 4 (null: every state)
  double* counts;          // [B*K][ldc] or null
  size_t ldc;
  double* logL;
  double* post_rate;
  int32_t* rate_class;
  double* norm;
  // null mode (AnalysisTools.cpp:587-653)
  int stat_kind;
  double stat_param;
  const double* stat_mean;
  size_t rep_ram;
  const uint8_t* supplied; // [nrep][2][T][rep_ram]
  double* null_stat;
  int32_t* null_rcmin;
  double* null_prmin;
  double* null_nmin;
};
int nuc_waves_per_simd(int NB);
hipError_t launch_map_nuc(const NucArgs& a, bool null_mode, int grid_blocks, hipStream_t stream);

}  // namespace cmx
