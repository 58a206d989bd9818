// Host-side model preparation for the engine (internal).  Everything here runs once per context:
// eigen-decomposition of the reversible generator, per (branch, class) transition matrices P = exp(Q r_c t_b),
// joint count matrices P o N^k (SubstitutionCountInterface::getAllNumbersOfSubstitutions, called per branch and
// class inside computeSubstitutionVectors in the reference -- here cached for all replicates), packing into the
// device layouts, and the tree program.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/comap_mi355x.h"

namespace cmx {

constexpr int kPlainStates = 64;   // device state count of the plain path (alphabets other than 4 / 20 states, up to 64)

struct HostModel {
  int S = 0, C = 0, K = 0, nn = 0, B = 0, T = 0, NI = 0, root = 0;
  // Device view.  Nucleotide models with >= 4 rate classes are mapped `fuse` classes at a time: the per-class 4-vectors
  // of a site are concatenated into one dS = 4*fuse vector and the per-class operators become the diagonal blocks of
  // one dS x dS operator (class probabilities folded into the count operators), so that one pass of the 20-state
  // machinery does the work of `fuse` class passes.  Otherwise dS = S, dC = C, fuse = 1.
  int dS = 0, dC = 0, fuse = 1;
  // Alphabets other than 4 / 20 states (codon models, CoETools.cpp:95-100): no matrix-core walk is built for them; the
  // plain kernels of cmx_variants.hip map their sites with the states padded to kPlainStates (operators padded with zeros).
  bool plain = false;
  std::vector<int> parent, first_child, next_sib, taxon_of, slot, int_post;
  std::vector<double> blen, pi, rates, probs, cum_pi, cum_probs;
  std::vector<double> P;    // [C][B][S*S] row-major (x -> y)
  std::vector<double> PN;   // [C][B][K][S*S] P o N^k
  std::vector<double> N1;   // [B][K][S*S] N^k at the branch length itself (rate 1): the no-averaging mapping
  std::vector<double> NC;   // [C][B][K][S*S] N^k at r_c t_b (conditional, not P o N): the marginal mapping (nijt.joint = no)
  std::vector<double> MAT;  // [C][MC][S*S]     device matrices: packed P | packed PN | leaf P^T | leaf PN^T (cmx_host_model.cpp)
  int MC = 0;               // matrices per (device) class block = NI + NI*K + T + K*T
  std::vector<double> eigV, eigVi, eigLam;   // [NM][S*S], [NM][S*S], [NM][S]: right / left eigenvectors and eigenvalues of the generators
  std::vector<int> model_of;                  // [B] generator of each branch (all 0 for a homogeneous model)
  std::vector<double> CP;   // [C][nn][S][S]    running row sums of P
  std::vector<uint8_t> CPG; // [C][nn][S][32]   guide table of the simulator's inverse-CDF search (cmx_kernels.hip: draw_guided)
  std::vector<int> simg;    // [nsimg][16] simulator: groups of four nodes of equal depth (DevModel::simg)
  std::vector<int> simord;  // [nn - 1] the non-root nodes by depth (DevModel::simord)
  int NV = 0;               // visited nodes of the binary device tree (internal, not inlined; pseudo nodes included)
  int NIW = 0;              // workspace slots: NI + pseudo nodes of split multifurcations
  std::vector<int> nrec;    // [NV][16] per-visited-node records (cmx_walk.h)
  std::vector<int> msched;  // operator uses of one class pass in program order: (matrix index, taxon or -1) pairs
  std::vector<int> ldsched; // workspace loads of one class pass: bit 31 prefetchable, bit 30 array, low 24 bits slot
  // Cherry tables (cmx_walk.h; class-fused nucleotide models only): an inlined cherry's message and outside visit as rows
  // of 1 + 3 K tables indexed by its two leaves' symbols, for fully resolved alignments (the null's).  The tables follow
  // the leaf operators in a class block (matrix index cherry_base + cherry_of[node] * (1 + 3 K) + table); the walk that
  // uses them has its own operator stream.  ncherry = 0: no tables (proteins, nucleotide models with < 4 classes).
  std::vector<int> cherry_of;   // [nn] cherry index of an inlined cherry node, else -1
  int ncherry = 0, cherry_base = 0;
  std::vector<int> msched_r;    // operator stream of the cherry-table walk; entry (matrix, taxon | 0x40000000 | tx1 | tx2 << 15 | -1)
  // per class pass, for traffic / flop accounting (_r: the cherry-table walk)
  size_t n_loads = 0, n_stores = 0, n_products = 0, n_leaf_ops = 0, n_products_r = 0, n_leaf_ops_r = 0;
};

// The walk of a rate-class pass lives in cmx_walk.h.  build_records: the per-node records it reads; record_walk: the
// operator stream and load schedule the device follows (a dry run of the walk); verify_walk: the walk run numerically
// on the host from the device layouts against a direct pruning computation (empty string when they agree).
void build_records(HostModel* hm);
void record_walk(HostModel* hm);
std::string verify_walk(const HostModel& hm);

// returns empty string on success, otherwise the error message (status in *code)
std::string build_host_model(const cmx_model* model, const cmx_tree* tree, HostModel* out, int* code);

}  // namespace cmx
