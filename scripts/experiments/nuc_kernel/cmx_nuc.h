// Nucleotide (4-state) mapping path: host program + device arguments (internal).
//
// Why a second mapping kernel.  The 20-state kernel stages SxS operators in LDS and runs products on the
// matrix cores; for S = 4 an operator is 128 bytes = 32 SGPRs, a message is 4 doubles = 8 VGPRs, and that machinery spends
// 20 scalar + vector instructions per matrix instruction (cfg 4: 0.13 of the fp64 roof, 18x the algorithmic bytes).
// Here: lane = site; the 4x4 operators of a branch arrive through the scalar cache (s_load_dwordx16) and are applied with
// v_fma_f64 taking SGPR operands -- no operator staging, no cross-lane traffic; and the per-node messages, which made
// the old kernel stream 80 KB per site through HBM, stay on the CU: the tree's internal nodes are cut into connected
// BLOCKS of <= NB nodes whose messages live in the wave's LDS slots.
//   phase 1 (inside):  blocks bottom-up, every rate class of a block in turn; only the message of a block's ROOT goes
//                      to HBM.
//   phase 2 (outside): blocks top-down; per block and class the block's inside messages are RECOMPUTED into the slots
//                      (20 FMAs per node, cheaper than any memory round trip), then the outside pass of the block runs
//                      from them; only the outside message of a lower block's root goes to HBM.
// Counts of a branch are accumulated over the rate classes in the count rows (read-modify-write of a row that the
// previous class of the same block visit left in L2), already weighted with p_c / L_site: no per-class partial counts.
// Leaves: the symbols of a block's <= 14 leaves are read once per block into one 64-bit word per lane (4-bit
// compatibility masks), and a leaf's message is one 32-byte gather from a table indexed by that mask ([class][taxon]
// [P | P o N^k][16 masks][4]: the operator's columns summed over the compatible states) -- every ambiguity code is served,
// no symbol load sits on a visit's critical path.
// The whole walk of a wave is ONE linear stream of 32-byte packets (inside visit / outside visit / leaf list of a
// block), so the next packets are known two ahead: records are read two packets ahead, the messages a visit takes from
// outside the block (leaf gathers, block-root messages) and the count rows it updates one packet ahead.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "cmx_host_model.h"

namespace cmx {

// child / source kinds of a visit packet
enum { NK_LEAF = 0, NK_SLOT = 1, NK_HBM = 2 };
// packet, 8 ints
enum { PK_FLAGS = 0, PK_A = 1, PK_APOS = 2, PK_B = 3, PK_BPOS = 4, PK_NODE = 5, PK_SLOT = 6, PK_KIDS = 7 };
// PK_A / PK_B: leaf -> taxon, PK_xPOS its position in the block's leaf list; slot -> LDS slot; hbm -> root slot.  The
//   outside pass disposes of a child's outside message at the same place (slot / root slot), or counts a leaf's branch.
// PK_NODE: branch id of the node (operators P at id * (K + 1), P o N^k at + 1 + k; count rows id * K + k), -1 for a
//   pseudo node or the tree root.  PK_SLOT: inside: destination (slot, or root slot for a block root); outside: source of
//   the node's outside message (slot, or root slot for a block root).  PK_KIDS: leaf children's branch ids, A | B << 16.
enum {
  NF_PSEUDO = 1 << 4,      // zero-length branch of a split multifurcation: no operator, no count
  NF_ROOT = 1 << 5,        // tree root
  NF_BLOCKROOT = 1 << 6,
  NF_OUTSIDE = 1 << 7,     // outside visit (else inside visit)
  NF_PHASE1 = 1 << 8,      // inside visit of phase 1: a block root stores its message, the tree root yields L_c
  NF_SWAPSYM = 1 << 9,     // first packet of a block: the leaf masks read ahead become the current ones
  NF_LASTCLASS = 1 << 10,  // outside: count rows are final after this packet (norm)
  NF_FIRSTCLASS = 1 << 11, // outside: count rows are written, not accumulated
  NF_BLOCKPKT = 1 << 12,   // leaf list of a block: up to 14 taxa, 16 bits each, in ints 1..7; count in bits 24..31
  NF_END = 1 << 13,
  NF_NOPF_A = 1 << 14,     // the message of child A / B, the node's outside message, the count rows were written by the packet
  NF_NOPF_B = 1 << 15,     //   right before this one: read them in the visit itself, not one packet ahead
  NF_NOPF_U = 1 << 16,
  NF_NOPF_CNT = 1 << 17,
  NF_FINISH1 = 1 << 18     // last packet of phase 1
};
constexpr int kNucMaxLeaves = 14;   // leaves of one block (16-bit taxa in 7 ints)
constexpr int kNucMaxNB = kNucMaxLeaves - 1;

struct NucProgram {
  int NB = 0;                 // block capacity (LDS slots per wave)
  int C = 0, K = 0, B = 0, T = 0;
  int nblocks = 0, nroots = 0, nops = 0, npk = 0;
  std::vector<int> pk;        // [npk + 2][8] the packet stream (END + one pad packet at the end)
  std::vector<double> ops;    // [C][nops][16] row-major 4x4: branch b: P at b*(K+1), P o N^k at b*(K+1) + 1 + k
  std::vector<double> ltab;   // [C][T][K+1][16][4] leaf tables: entry (mask m) = sum over the states z in m of column z
  // per class pass, for flop / traffic accounting
  size_t n_apply = 0;         // 4x4 operator applications on internal branches (inside of phase 1 + recompute + outside)
  size_t n_leaf = 0;          // leaf-table gathers
  size_t n_root_loads = 0, n_root_stores = 0;   // 32-byte-per-site messages through HBM
};

// builds the program for a 4-state model; empty string on success
std::string build_nuc_program(const HostModel& hm, int NB, NucProgram* out);
// runs the packet stream numerically on the host for one random site exactly as the device does (prefetch-ahead reads
// included: a message or count row marked for reading one packet ahead is READ one packet ahead) and compares likelihood
// and every count with a direct pruning computation (empty string when they agree)
std::string verify_nuc_program(const HostModel& hm, const NucProgram& np);

// ---- device side
struct NucDev {
  int C, K, B, T, nroots, nops, NB;
  const double* ops;
  const double* ltab;
  const int* pk;
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
