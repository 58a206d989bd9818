// Host-side model preparation (see cmx_host_model.h).  Plain C++17, no device code.
#include "cmx_host_model.h"
#include "cmx_device.h"
#include "cmx_walk.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>

namespace cmx {
namespace {

using Mat = std::vector<double>;

Mat matmul(int n, const Mat& A, const Mat& B) {
  Mat C((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k) {
      const double a = A[(size_t)i * n + k];
      for (int j = 0; j < n; ++j) C[(size_t)i * n + j] += a * B[(size_t)k * n + j];
    }
  return C;
}

// cyclic Jacobi eigen-solver for a symmetric matrix (n <= 64 here); columns of U are eigenvectors
void jacobi(int n, Mat A, Mat* U, std::vector<double>* lam) {
  U->assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) (*U)[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 200; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += A[(size_t)i * n + i] * A[(size_t)i * n + i];
      for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j];
    }
    if (off <= 1e-40 * (diag + 1e-300)) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
          A[(size_t)k * n + p] = c * akp - s * akq;
          A[(size_t)k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
          A[(size_t)p * n + k] = c * apk - s * aqk;
          A[(size_t)q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double ukp = (*U)[(size_t)k * n + p], ukq = (*U)[(size_t)k * n + q];
          (*U)[(size_t)k * n + p] = c * ukp - s * ukq;
          (*U)[(size_t)k * n + q] = s * ukp + c * ukq;
        }
      }
  }
  lam->resize(n);
  for (int i = 0; i < n; ++i) (*lam)[i] = A[(size_t)i * n + i];
}

// row-major SxS -> 4x4-block packed (tile t = (bi, bj), element k = (k/4, k%4)); layout consumed by matvec_stage.
// diag (class-fused nucleotide models: the operator is block diagonal, one 4-state tile per class): only the NB diagonal
// tiles are stored, tile (q, q) at position q -- the kernel then stages NB * 128 bytes per product instead of the unit
void pack_blocks(int S, const double* M, double* out, bool diag) {
  const int NB = S / 4;
  if (diag) {
    for (int q = 0; q < NB; ++q)
      for (int k = 0; k < 16; ++k) out[q * 16 + k] = M[(size_t)(4 * q + k / 4) * S + 4 * q + k % 4];
    return;
  }
  for (int t = 0; t < NB * NB; ++t)
    for (int k = 0; k < 16; ++k) out[t * 16 + k] = M[(size_t)(4 * (t / NB) + k / 4) * S + 4 * (t % NB) + k % 4];
}

}  // namespace

// ------------------------------------------------------------------------------------------------ tree program
// The walk of a rate-class pass is written once (cmx_walk.h).  Here: the per-node records it reads, the Recorder
// backend that lists its operators and workspace loads in program order (what the device follows), and the Numeric
// backend + direct computation that check the whole thing before a context is accepted.
void build_records(HostModel* hm) {
  const int root = hm->root, nn = hm->nn;
  // ---- binary device tree: nodes 0 .. nn-1 are the tree's own, nn .. are pseudo nodes (zero-length branches) that
  // split a node with k > 2 children c1 .. ck into ((..((c1, c2), c3) ..), ck)
  std::vector<std::array<int, 2>> ch(nn, {-1, -1});
  auto kids0 = [&](int n) { std::vector<int> v; for (int e = hm->first_child[n]; e >= 0; e = hm->next_sib[e]) v.push_back(e); return v; };
  for (int n = 0; n < nn; ++n) {
    if (hm->taxon_of[n] >= 0) continue;
    const std::vector<int> c = kids0(n);
    int left = c[0];
    for (size_t i = 1; i + 1 < c.size(); ++i) {
      ch.push_back({left, c[i]});
      left = (int)ch.size() - 1;
    }
    ch[n] = {left, c.back()};
  }
  const int nd = (int)ch.size();
  auto is_leaf = [&](int n) { return n < nn && hm->taxon_of[n] >= 0; };
  auto pseudo = [&](int n) { return n >= nn; };
  // workspace slots: the tree's internal nodes keep their operator slot, pseudo nodes follow
  std::vector<int> wslot(nd, -1);
  for (int n = 0; n < nn; ++n) wslot[n] = hm->slot[n];
  for (int n = nn; n < nd; ++n) wslot[n] = hm->NI + (n - nn);
  hm->NIW = hm->NI + (nd - nn);
  // inlined cherries: a (real, non-root) internal node with two leaf children is never visited
  std::vector<char> inlined(nd, 0);
  for (int n = 0; n < nn; ++n)
    if (!is_leaf(n) && n != root && is_leaf(ch[n][0]) && is_leaf(ch[n][1])) inlined[n] = 1;
  hm->cherry_of.assign(nn, -1);
  hm->ncherry = 0;
  for (int n = 0; n < nn; ++n)
    if (inlined[n]) hm->cherry_of[n] = hm->ncherry++;
  auto kind = [&](int e) { return is_leaf(e) ? (int)KIND_LEAF : (inlined[e] ? (int)KIND_CHERRY : (int)KIND_STORED); };
  // post-order of the visited nodes (explicit stack: caterpillar trees are deep)
  std::vector<int> visited, parent_d(nd, -1);
  {
    std::vector<std::pair<int, int>> st;
    st.push_back({root, 0});
    while (!st.empty()) {
      auto& top = st.back();
      const int n = top.first;
      if (is_leaf(n) || inlined[n]) { st.pop_back(); continue; }
      if (top.second < 2) {
        const int e = ch[n][top.second++];
        parent_d[e] = n;
        st.push_back({e, 0});
      } else {
        visited.push_back(n);
        st.pop_back();
      }
    }
  }
  const int NV = (int)visited.size();
  hm->NV = NV;
  hm->nrec.assign((size_t)NV * 16, -1);
  auto fill_child = [&](int* d, int e) {
    d[CH_KIND] = kind(e); d[CH_NODE] = pseudo(e) ? -1 : e; d[CH_SLOT] = is_leaf(e) ? -1 : wslot[e];
    d[CH_L1] = d[CH_L2] = -1;
    if (kind(e) == KIND_CHERRY) { d[CH_L1] = ch[e][0]; d[CH_L2] = ch[e][1]; }
  };
  for (int v = 0; v < NV; ++v) {
    const int n = visited[v];
    int* r = &hm->nrec[(size_t)v * 16];
    r[REC_NODE] = pseudo(n) ? -1 : n; r[REC_SLOT] = wslot[n]; r[2] = 2; r[REC_FLAGS] = 0;
    if (n == root) r[REC_FLAGS] |= FLAG_ROOT;
    if (pseudo(n)) r[REC_FLAGS] |= FLAG_PSEUDO;
    // the child visited right before n (inside pass) = right after n (outside pass): its vectors stay in registers
    int a = ch[n][0], b = ch[n][1];
    if (v > 0 && parent_d[visited[v - 1]] == n) {
      if (visited[v - 1] == a) std::swap(a, b);
      r[REC_FLAGS] |= FLAG_HAND;
    }
    if (v + 1 < NV && parent_d[n] == visited[v + 1]) r[REC_FLAGS] |= FLAG_U_HANDED;
    fill_child(r + REC_A, a);
    fill_child(r + REC_B, b);
  }
}

namespace {
// ---- Recorder: the operator stream (matrix index in a class block, taxon or -1) and the workspace loads of a pass
template <bool CT>
struct Recorder {
  static constexpr bool kCherryTables = CT;
  HostModel* hm;
  long t = 0;
  std::vector<long> store_time[2];
  struct Ld { int arr, slot; long t, src; };
  std::vector<Ld> loads;
  explicit Recorder(HostModel* h) : hm(h) { store_time[0].assign(h->NIW, -1); store_time[1].assign(h->NIW, -1); }
  // matrix indices inside a class block (HostModel::MAT): P[slot] | J[slot*K+k] | leaf P^T[taxon] | leaf J^T[k*T+taxon]
  int mat_internal(int node, int which) const { return which < 0 ? hm->slot[node] : hm->NI + hm->slot[node] * hm->K + which; }
  int mat_leaf(int leaf, int which) const {
    const int tx = hm->taxon_of[leaf];
    return which < 0 ? hm->NI + hm->NI * hm->K + tx : hm->NI + hm->NI * hm->K + hm->T + which * hm->T + tx;
  }
  void op(int mat, int tx) {
    std::vector<int>& ms = CT ? hm->msched_r : hm->msched;
    ms.push_back(mat); ms.push_back(tx); ++t;
  }
  void leaf_op() { (CT ? hm->n_leaf_ops_r : hm->n_leaf_ops)++; }
  void rec(int v, int (&r)[16]) const { for (int i = 0; i < 16; ++i) r[i] = hm->nrec[(size_t)v * 16 + i]; }
  template <int D> void lset(int leaf, int which) { op(mat_leaf(leaf, which), hm->taxon_of[leaf]); leaf_op(); }
  template <int S, int D> void lmul(int leaf, int which) { op(mat_leaf(leaf, which), hm->taxon_of[leaf]); leaf_op(); }
  template <int S> void ldot(int leaf, int which, int) { op(mat_leaf(leaf, which), hm->taxon_of[leaf]); leaf_op(); }
  // cherry-table ops: the table's matrix index, both taxa in the stream entry
  int mat_cherry(int node, int table) const { return hm->cherry_base + hm->cherry_of[node] * (1 + 3 * hm->K) + table; }
  int tx_cherry(int l1, int l2) const { return 0x40000000 | hm->taxon_of[l1] | (hm->taxon_of[l2] << 15); }
  template <int D> void cset(int node, int l1, int l2) { op(mat_cherry(node, 0), tx_cherry(l1, l2)); leaf_op(); }
  template <int S> void cdot(int node, int l1, int l2, int table, int) { op(mat_cherry(node, table), tx_cherry(l1, l2)); leaf_op(); }
  template <int S, int D, bool TR> void mv(int node, int which) { op(mat_internal(node, which), -1); (CT ? hm->n_products_r : hm->n_products)++; }
  template <int D> void load(int arr, int slot) { loads.push_back({arr, slot, t++, store_time[arr][slot]}); if (!CT) hm->n_loads++; }
  template <int S> void store(int arr, int slot) { store_time[arr][slot] = t++; if (!CT) hm->n_stores++; }
  template <int D, int S> void mov() {}
  template <int D, int S> void mul() {}
  void mulup() {}
  template <int D> void setpi() {}
  template <int S> void rootl() {}
  void dot3(int) {}
  template <int R> void kill() {}
};

// ---- Numeric: the pass in plain doubles for one site, operators read from HostModel::MAT in their device layouts
// and selected by the recorded stream exactly as the device selects them; every register starts as NaN.
template <bool CT>
struct Numeric {
  static constexpr bool kCherryTables = CT;
  const HostModel& hm;
  int dS, NB;
  size_t MU;
  const double* blk;
  std::vector<int> code;                   // symbol per taxon
  std::vector<double> R[4], ws[2], cnt, Lg;
  std::vector<char> counted;
  size_t mi = 0, fi = 0;
  std::string err;
  explicit Numeric(const HostModel& h) : hm(h), dS(h.dS), NB(h.dS / 4), MU((size_t)mat_unit(h.dS)), blk(h.MAT.data()) {
    const double nan = std::nan("");
    for (auto& r : R) r.assign(dS, nan);
    ws[0].assign((size_t)h.NIW * dS, nan);
    ws[1].assign((size_t)h.NIW * dS, nan);
    cnt.assign((size_t)h.B * h.K, nan);
    counted.assign((size_t)h.B * h.K, 0);
    Lg.assign(h.fuse, nan);
    code.resize(h.T);
    for (int t = 0; t < h.T; ++t) {      // splitmix-style hash of the taxon index: no global RNG state
      uint64_t z = 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
      code[t] = (int)(z % (uint64_t)h.S);
    }
  }
  void fail(const std::string& m) { if (err.empty()) err = "tree-walk self-check failed: " + m; }
  void rec(int v, int (&r)[16]) const { for (int i = 0; i < 16; ++i) r[i] = hm.nrec[(size_t)v * 16 + i]; }
  // the operator the stream stages for this op; `want` = what the walk asked for
  const std::vector<int>& stream() const { return CT ? hm.msched_r : hm.msched; }
  int staged(int want_mat, int want_tx) {
    if (2 * mi + 1 >= stream().size()) { fail("more operator uses than the stream holds"); return -1; }
    const int mat = stream()[2 * mi], tx = stream()[2 * mi + 1];
    ++mi;
    if (mat < 0 || mat >= hm.MC) { fail("operator index out of range"); return -1; }
    if (mat != want_mat || tx != want_tx) { fail("op " + std::to_string(mi - 1) + " finds the wrong operator staged"); return -1; }
    return mat;
  }
  int leaf_mat(int leaf, int which) {
    if (leaf < 0 || leaf >= hm.nn || hm.taxon_of[leaf] < 0) { fail("leaf op on a non-leaf"); return -1; }
    const int tx = hm.taxon_of[leaf];
    return staged(which < 0 ? hm.NI + hm.NI * hm.K + tx : hm.NI + hm.NI * hm.K + hm.T + which * hm.T + tx, tx);
  }
  // cherry tables: row = 4 * symbol(l1) + symbol(l2), columns as in a leaf row
  int cherry_mat(int node, int l1, int l2, int table) {
    if (node < 0 || node >= hm.nn || hm.cherry_of[node] < 0 || table < 0 || table > 3 * hm.K) { fail("cherry op on a node without tables"); return -1; }
    return staged(hm.cherry_base + hm.cherry_of[node] * (1 + 3 * hm.K) + table, 0x40000000 | hm.taxon_of[l1] | (hm.taxon_of[l2] << 15));
  }
  double cherryrow(int mat, int l1, int l2, int X) const {
    return blk[(size_t)mat * MU + (size_t)(4 * code[hm.taxon_of[l1]] + code[hm.taxon_of[l2]]) * leaf_row_stride(dS) + (X % 4) * NB + X / 4];
  }
  template <int D> void cset(int node, int l1, int l2) {
    const int mat = cherry_mat(node, l1, l2, 0);
    if (mat < 0) return;
    for (int x = 0; x < dS; ++x) R[D][x] = cherryrow(mat, l1, l2, x);
  }
  template <int S> void cdot(int node, int l1, int l2, int table, int row) {
    const int mat = cherry_mat(node, l1, l2, table);
    if (mat < 0) return;
    double s = 0;
    for (int x = 0; x < dS; ++x) s += R[S][x] * cherryrow(mat, l1, l2, x);
    count_row(row, s);
  }
  double leafrow(int mat, int leaf, int X) const { return blk[(size_t)mat * MU + (size_t)code[hm.taxon_of[leaf]] * leaf_row_stride(dS) + (X % 4) * NB + X / 4]; }
  double packed(int mat, int r, int c) const {
    if (hm.fuse > 1)   // diagonal tiles only (pack_blocks), the others are exact zeros of the block-diagonal operator
      return r / 4 == c / 4 ? blk[(size_t)mat * MU + (size_t)(r / 4) * 16 + (r % 4) * 4 + c % 4] : 0.0;
    return blk[(size_t)mat * MU + ((size_t)(r / 4) * NB + c / 4) * 16 + (r % 4) * 4 + c % 4];
  }
  void count_row(int row, double v) {
    if (row < 0 || row >= hm.B * hm.K) return fail("count row out of range");
    if (counted[row]) return fail("branch counted twice");
    counted[row] = 1;
    cnt[row] = v;
  }
  template <int D> void lset(int leaf, int which) {
    const int mat = leaf_mat(leaf, which);
    if (mat < 0) return;
    for (int x = 0; x < dS; ++x) R[D][x] = leafrow(mat, leaf, x);
  }
  template <int S, int D> void lmul(int leaf, int which) {
    const int mat = leaf_mat(leaf, which);
    if (mat < 0) return;
    for (int x = 0; x < dS; ++x) R[D][x] = R[S][x] * leafrow(mat, leaf, x);
  }
  template <int S> void ldot(int leaf, int which, int row) {
    const int mat = leaf_mat(leaf, which);
    if (mat < 0) return;
    double s = 0;
    for (int x = 0; x < dS; ++x) s += R[S][x] * leafrow(mat, leaf, x);
    count_row(row, s);
  }
  template <int S, int D, bool TR> void mv(int node, int which) {
    if (node < 0 || node >= hm.nn || hm.slot[node] < 0) return fail("product on a leaf or pseudo branch");
    const int mat = staged(which < 0 ? hm.slot[node] : hm.NI + hm.slot[node] * hm.K + which, -1);
    if (mat < 0) return;
    std::vector<double> out(dS, 0.0);
    for (int r = 0; r < dS; ++r)
      for (int c = 0; c < dS; ++c) out[r] += (TR ? packed(mat, c, r) : packed(mat, r, c)) * R[S][c];
    R[D] = out;
  }
  template <int D> void load(int arr, int slot) {
    if (slot < 0 || slot >= hm.NIW) return fail("workspace slot out of range");
    if (fi >= hm.ldsched.size()) return fail("more workspace loads than scheduled");
    const int w = hm.ldsched[fi++];
    if (((w >> 30) & 1) != arr || (w & 0xffffff) != slot) return fail("load " + std::to_string(fi - 1) + " names the wrong vector");
    R[D].assign(&ws[arr][(size_t)slot * dS], &ws[arr][(size_t)slot * dS] + dS);
  }
  template <int S> void store(int arr, int slot) {
    if (slot < 0 || slot >= hm.NIW) return fail("workspace slot out of range");
    std::copy(R[S].begin(), R[S].end(), &ws[arr][(size_t)slot * dS]);
  }
  template <int D, int S> void mov() { R[D] = R[S]; }
  template <int D, int S> void mul() { for (int x = 0; x < dS; ++x) R[D][x] *= R[S][x]; }
  void mulup() { for (int x = 0; x < dS; ++x) { R[1][x] *= R[3][x]; R[2][x] *= R[3][x]; } }
  template <int D> void setpi() { for (int x = 0; x < dS; ++x) R[D][x] = hm.pi[x % hm.S]; }
  template <int S> void rootl() {
    for (int g = 0; g < hm.fuse; ++g) { double s = 0; for (int x = 0; x < hm.S; ++x) s += hm.pi[x] * R[S][g * hm.S + x]; Lg[g] = s; }
  }
  void dot3(int row) { double s = 0; for (int x = 0; x < dS; ++x) s += R[3][x] * R[1][x] * R[2][x]; count_row(row, s); }
  template <int Rg> void kill() { R[Rg].assign(dS, std::nan("")); }   // a killed register must not be read again
};
}  // namespace

// records -> operator stream + load schedule (with prefetchability) by a dry run of the walk
void record_walk(HostModel* hm) {
  hm->msched.clear();
  hm->msched_r.clear();
  hm->ldsched.clear();
  hm->n_loads = hm->n_stores = hm->n_products = hm->n_leaf_ops = hm->n_products_r = hm->n_leaf_ops_r = 0;
  if (hm->cherry_base > 0) {   // the cherry-table walk's own operator stream (same loads and stores)
    Recorder<true> rt(hm);
    walk_pass(rt, hm->NV, hm->K);
  }
  Recorder<false> rc(hm);
  walk_pass(rc, hm->NV, hm->K);
  for (size_t j = 0; j < rc.loads.size(); ++j) {
    const Recorder<false>::Ld& e = rc.loads[j];
    unsigned w = (unsigned)e.slot | (e.arr ? 0x40000000u : 0u);
    // prefetchable: its producer store is issued before the previous load (where the prefetch is issued)
    if (j > 0 && e.src >= 0 && e.src < rc.loads[j - 1].t) w |= 0x80000000u;
    hm->ldsched.push_back((int)w);
  }
}

// Runs the walk numerically for one random site of device class 0 and compares site likelihood and all joint counts
// with a direct pruning computation from the row-major hm.P / hm.PN.  Empty string when they agree.
std::string verify_walk(const HostModel& hm) {
  const int S = hm.S, F = hm.fuse, K = hm.K, nn = hm.nn, B = hm.B, root = hm.root;
  const size_t S2 = (size_t)S * S;
  if ((int)hm.nrec.size() != hm.NV * 16) return "tree-walk self-check failed: record table size";
  Numeric<false> nm(hm);
  walk_pass(nm, hm.NV, K);
  if (!nm.err.empty()) return nm.err;
  if (2 * nm.mi != hm.msched.size()) return "tree-walk self-check failed: unused operators in the stream";
  if (nm.fi != hm.ldsched.size()) return "tree-walk self-check failed: unused workspace loads in the schedule";
  // the cherry-table walk: same site, same reference
  Numeric<true> nt(hm);
  const bool tables = !hm.msched_r.empty();
  if (tables) {
    walk_pass(nt, hm.NV, K);
    if (!nt.err.empty()) return nt.err + " (cherry-table walk)";
    if (2 * nt.mi != hm.msched_r.size()) return "tree-walk self-check failed: unused operators in the cherry-table stream";
    if (nt.fi != hm.ldsched.size()) return "tree-walk self-check failed: the cherry-table walk loads other workspace vectors";
  }
  const std::vector<int>& code = nm.code;
  std::vector<double> ref((size_t)B * K, 0.0), Lref(F, 0.0);
  auto kids = [&](int n) { std::vector<int> v; for (int e = hm.first_child[n]; e >= 0; e = hm.next_sib[e]) v.push_back(e); return v; };
  for (int g = 0; g < F && g < hm.C; ++g) {   // true classes g of device class 0
    std::vector<double> D((size_t)nn * S), M((size_t)nn * S), U((size_t)nn * S), Up((size_t)nn * S);
    for (int n = 0; n < nn; ++n) {
      double* Dn = &D[(size_t)n * S];
      if (hm.taxon_of[n] >= 0) for (int x = 0; x < S; ++x) Dn[x] = x == code[hm.taxon_of[n]] ? 1.0 : 0.0;
      else { for (int x = 0; x < S; ++x) Dn[x] = 1.0; for (int e : kids(n)) for (int x = 0; x < S; ++x) Dn[x] *= M[(size_t)e * S + x]; }
      if (n != root) {
        const double* P = &hm.P[((size_t)g * B + n) * S2];
        for (int x = 0; x < S; ++x) { double s = 0; for (int z = 0; z < S; ++z) s += P[(size_t)x * S + z] * Dn[z]; M[(size_t)n * S + x] = s; }
      }
    }
    for (int x = 0; x < S; ++x) { Lref[g] += hm.pi[x] * D[(size_t)root * S + x]; Up[(size_t)root * S + x] = hm.pi[x]; }
    const double wgt = F > 1 ? hm.probs[g] : 1.0;   // fused: class probabilities are folded into the count operators
    for (int f = nn - 1; f >= 0; --f) {
      if (hm.taxon_of[f] >= 0) continue;
      const std::vector<int> c = kids(f);
      for (int n : c) {
        double* Un = &U[(size_t)n * S];
        for (int x = 0; x < S; ++x) Un[x] = Up[(size_t)f * S + x];
        for (int m : c) if (m != n) for (int x = 0; x < S; ++x) Un[x] *= M[(size_t)m * S + x];
        for (int k = 0; k < K; ++k) {
          const double* PN = &hm.PN[(((size_t)g * B + n) * K + k) * S2];
          double tot = 0;
          for (int x = 0; x < S; ++x) { double s = 0; for (int y = 0; y < S; ++y) s += PN[(size_t)x * S + y] * D[(size_t)n * S + y]; tot += Un[x] * s; }
          ref[(size_t)n * K + k] += wgt * tot;
        }
        if (hm.taxon_of[n] < 0) {
          const double* P = &hm.P[((size_t)g * B + n) * S2];
          for (int z = 0; z < S; ++z) { double s = 0; for (int x = 0; x < S; ++x) s += P[(size_t)x * S + z] * Un[x]; Up[(size_t)n * S + z] = s; }
        }
      }
    }
  }
  auto close = [](double a, double b) { return std::fabs(a - b) <= 1e-9 * (std::fabs(a) + std::fabs(b)) + 1e-290; };
  for (int g = 0; g < F && g < hm.C; ++g)
    if (!close(nm.Lg[g], Lref[g])) return "tree-walk self-check failed: site likelihood differs from the direct computation";
  for (size_t r = 0; r < ref.size(); ++r) {
    if (!nm.counted[r]) return "tree-walk self-check failed: a branch is never counted";
    if (!close(nm.cnt[r], ref[r])) return "tree-walk self-check failed: joint count of branch " + std::to_string(r / K) + " differs from the direct computation";
  }
  if (tables) {
    for (int g = 0; g < F && g < hm.C; ++g)
      if (!close(nt.Lg[g], Lref[g])) return "tree-walk self-check failed: site likelihood of the cherry-table walk differs from the direct computation";
    for (size_t r = 0; r < ref.size(); ++r) {
      if (!nt.counted[r]) return "tree-walk self-check failed: the cherry-table walk never counts a branch";
      if (!close(nt.cnt[r], ref[r])) return "tree-walk self-check failed: cherry-table count of branch " + std::to_string(r / K) + " differs from the direct computation";
    }
  }
  return std::string();
}

std::string build_host_model(const cmx_model* model, const cmx_tree* tree, HostModel* hm, int* code) {
  *code = CMX_ERR_INVALID;
  if (!model || !tree) return "model and tree are required";
  const int S = model->nstates, C = model->nclasses;
  const int K = (model->nmodels > 0 ? model->Bks : model->Bk) ? model->ntypes : 1;
  if (S < 2 || S > kPlainStates) {
    *code = CMX_ERR_UNSUPPORTED;
    return "nstates must be between 2 and " + std::to_string(kPlainStates) + " (4 and 20 run on the matrix cores, the others on the plain kernels); got " + std::to_string(S);
  }
  hm->plain = S != 4 && S != 20;
  if (C < 1 || C > 64) return "nclasses out of range";
  if (K < 1 || K > 64) return "ntypes out of range";
  if (!model->rates || !model->probs) return "rates and probs are required";
  if (model->nmodels <= 0 && (!model->Q || !model->pi)) return "Q, pi, rates and probs are required";
  const int nn = tree->nnodes, T = tree->ntaxa;
  if (nn < 3 || T < 2 || !tree->parent || !tree->blen || !tree->leaf_of_taxon) return "tree is incomplete";
  if (nn > 65535) return "tree too large";
  hm->S = S; hm->C = C; hm->K = K; hm->nn = nn; hm->B = nn - 1; hm->T = T; hm->root = nn - 1;
  hm->parent.assign(tree->parent, tree->parent + nn);
  hm->blen.assign(tree->blen, tree->blen + nn);
  // ---- tree checks: post-order, root last
  if (hm->parent[nn - 1] != -1) return "parent[root] must be -1 with the root last";
  for (int i = 0; i < nn - 1; ++i) {
    if (hm->parent[i] <= i || hm->parent[i] >= nn) return "nodes must be in post-order (parent id > child id)";
    if (!(hm->blen[i] >= 0.0) || !std::isfinite(hm->blen[i])) return "branch lengths must be finite and >= 0";
  }
  hm->first_child.assign(nn, -1);
  hm->next_sib.assign(nn, -1);
  std::vector<int> last(nn, -1), nchild(nn, 0);
  for (int i = 0; i < nn - 1; ++i) {
    const int p = hm->parent[i];
    if (hm->first_child[p] < 0) hm->first_child[p] = i; else hm->next_sib[last[p]] = i;
    last[p] = i;
    nchild[p]++;
  }
  hm->taxon_of.assign(nn, -1);
  for (int t = 0; t < T; ++t) {
    const int n = tree->leaf_of_taxon[t];
    if (n < 0 || n >= nn || nchild[n] != 0) return "leaf_of_taxon must name leaves";
    if (hm->taxon_of[n] >= 0) return "leaf_of_taxon has duplicates";
    hm->taxon_of[n] = t;
  }
  hm->slot.assign(nn, -1);
  hm->int_post.clear();
  for (int i = 0; i < nn; ++i) {
    if (nchild[i] == 0) {
      if (hm->taxon_of[i] < 0) return "every leaf needs an alignment row";
    } else {
      if (nchild[i] < 2 && i != nn - 1) return "internal nodes need at least two children";
      hm->slot[i] = (int)hm->int_post.size();
      hm->int_post.push_back(i);
    }
  }
  if (nchild[nn - 1] < 2) return "the root needs at least two children";
  hm->NI = (int)hm->int_post.size();
  if (!hm->plain) {
    build_records(hm);
    // cherry tables only for the class-fused nucleotide layout (16 symbol pairs; 400 for proteins would not fit a stage buffer)
    // (a fused model without a single cherry still gets the second stream -- identical to the first: the null's kernel
    // instantiation reads it unconditionally)
    hm->cherry_base = (S == 4 && C >= 4) ? hm->NI + hm->NI * K + T + K * T : 0;
    if (hm->cherry_base == 0) hm->ncherry = 0;
    record_walk(hm);
  }
  {  // simulator: nodes by depth, four of a level at a time (a level's draws only need the level above); a short group is
     // padded by repeating its last node (drawing a node twice gives the same state twice)
    std::vector<int> depth(nn, 0);
    int maxd = 0;
    for (int i = nn - 2; i >= 0; --i) { depth[i] = depth[hm->parent[i]] + 1; maxd = std::max(maxd, depth[i]); }
    std::vector<std::vector<int>> level(maxd + 1);
    for (int i = nn - 2; i >= 0; --i) level[depth[i]].push_back(i);
    hm->simg.clear();
    hm->simord.clear();
    for (int d = 1; d <= maxd; ++d) hm->simord.insert(hm->simord.end(), level[d].begin(), level[d].end());
    for (int d = 1; d <= maxd; ++d)
      for (size_t i = 0; i < level[d].size(); i += 4) {
        int g[16];
        for (int j = 0; j < 4; ++j) {
          const int n = level[d][std::min(i + j, level[d].size() - 1)];
          g[j] = n; g[4 + j] = hm->parent[n]; g[8 + j] = hm->taxon_of[n]; g[12 + j] = 0;
        }
        hm->simg.insert(hm->simg.end(), g, g + 16);
      }
  }
  // ---- model checks
  const bool nh = model->nmodels > 0;
  const int NM = nh ? model->nmodels : 1;
  if (nh && (!model->Qs || !model->pis || !model->model_of_branch || !model->root_freqs))
    return "non-homogeneous model: Qs, pis, model_of_branch and root_freqs are required";
  if (!nh && (model->Qs || model->pis || model->Bks || model->model_of_branch || model->root_freqs))
    return "non-homogeneous fields are set but nmodels is 0";
  // frequencies at the root: the stationary ones, or the model set's root frequency set
  const double* rootf = nh ? model->root_freqs : model->pi;
  hm->pi.assign(rootf, rootf + S);
  hm->rates.assign(model->rates, model->rates + C);
  hm->probs.assign(model->probs, model->probs + C);
  double spi = 0, spr = 0;
  for (double v : hm->pi) { if (!(v > 0)) return "pi must be positive"; spi += v; }
  for (double v : hm->probs) { if (!(v >= 0)) return "probs must be >= 0"; spr += v; }
  if (std::fabs(spi - 1.0) > 1e-6 || std::fabs(spr - 1.0) > 1e-6) return "pi and probs must sum to one";
  for (double v : hm->rates) if (!(v >= 0) || !std::isfinite(v)) return "rates must be finite and >= 0";
  const size_t S2 = (size_t)S * S;
  std::vector<int> model_of(hm->B, 0);
  if (nh)
    for (int b = 0; b < hm->B; ++b) {
      model_of[b] = model->model_of_branch[b];
      if (model_of[b] < 0 || model_of[b] >= NM) return "model_of_branch: generator index out of range";
    }
  // ---- per generator: checks and eigen-decomposition through the symmetrised generator
  struct Eig { Mat V, Vi; std::vector<double> lam; std::vector<Mat> W; };
  std::vector<Eig> eig(NM);
  for (int m = 0; m < NM; ++m) {
    const double* Qp = nh ? model->Qs + (size_t)m * S2 : model->Q;
    const double* pip = nh ? model->pis + (size_t)m * S : model->pi;
    const double* Bp = nh ? (model->Bks ? model->Bks + (size_t)m * K * S2 : nullptr) : model->Bk;
    Mat Q(Qp, Qp + S2);
    double sp = 0;
    for (int x = 0; x < S; ++x) { if (!(pip[x] > 0)) return "pi must be positive"; sp += pip[x]; }
    if (std::fabs(sp - 1.0) > 1e-6) return "pi and probs must sum to one";
    for (int x = 0; x < S; ++x) {
      double rs = 0;
      for (int y = 0; y < S; ++y) {
        rs += Q[(size_t)x * S + y];
        const double a = pip[x] * Q[(size_t)x * S + y], b = pip[y] * Q[(size_t)y * S + x];
        if (std::fabs(a - b) > 1e-9 * (std::fabs(a) + std::fabs(b) + 1e-300) + 1e-14)
          return "Q must be reversible with respect to pi (pi_x Q_xy == pi_y Q_yx)";
      }
      if (std::fabs(rs) > 1e-9) return "rows of Q must sum to zero";
    }
    std::vector<Mat> Bk(K);
    for (int k = 0; k < K; ++k) {
      if (Bp) Bk[k].assign(Bp + k * S2, Bp + (k + 1) * S2);
      else { Bk[k] = Q; for (int x = 0; x < S; ++x) Bk[k][(size_t)x * S + x] = 0.0; }
    }
    Mat A(S2), U;
    for (int x = 0; x < S; ++x)
      for (int y = 0; y < S; ++y) A[(size_t)x * S + y] = std::sqrt(pip[x]) * Q[(size_t)x * S + y] / std::sqrt(pip[y]);
    for (int x = 0; x < S; ++x)
      for (int y = x + 1; y < S; ++y) A[(size_t)x * S + y] = A[(size_t)y * S + x] = 0.5 * (A[(size_t)x * S + y] + A[(size_t)y * S + x]);
    Eig& e = eig[m];
    jacobi(S, A, &U, &e.lam);
    e.V.resize(S2); e.Vi.resize(S2);
    for (int x = 0; x < S; ++x)
      for (int j = 0; j < S; ++j) {
        e.V[(size_t)x * S + j] = U[(size_t)x * S + j] / std::sqrt(pip[x]);
        e.Vi[(size_t)j * S + x] = U[(size_t)x * S + j] * std::sqrt(pip[x]);
      }
    e.W.resize(K);   // Vinv B_k V
    for (int k = 0; k < K; ++k) e.W[k] = matmul(S, matmul(S, e.Vi, Bk[k]), e.V);
  }
  hm->model_of = model_of;
  hm->eigV.clear(); hm->eigVi.clear(); hm->eigLam.clear();
  for (int m = 0; m < NM; ++m) {
    hm->eigV.insert(hm->eigV.end(), eig[m].V.begin(), eig[m].V.end());
    hm->eigVi.insert(hm->eigVi.end(), eig[m].Vi.begin(), eig[m].Vi.end());
    hm->eigLam.insert(hm->eigLam.end(), eig[m].lam.begin(), eig[m].lam.end());
  }
  // ---- per (class, branch) matrices, each branch with its own generator
  const int B = hm->B;
  hm->P.assign((size_t)C * B * S2, 0.0);
  hm->PN.assign((size_t)C * B * K * S2, 0.0);
  Mat E(S2), Phi(S2);
  // P(t) of branch b and, per substitution type, either P o N^k (the joint count operator of the averaged mapping) or the
  // conditional expectation N^k itself (nijt.average = no picks single entries of it)
  auto branch_mats = [&](int b, double t, double* P, double* outK /*[K][S2]*/, bool conditional) {
    const Eig& e = eig[model_of[b]];
    const Mat &V = e.V, &Vi = e.Vi;
    const std::vector<double>& lam = e.lam;
    const std::vector<Mat>& W = e.W;
    for (int x = 0; x < S; ++x)
      for (int j = 0; j < S; ++j) E[(size_t)x * S + j] = V[(size_t)x * S + j] * std::exp(lam[j] * t);
    Mat Pm = matmul(S, E, Vi);
    std::memcpy(P, Pm.data(), sizeof(double) * S2);
    for (int k = 0; k < K; ++k) {
      double* Ok = outK + (size_t)k * S2;
      if (model->count_method == CMX_COUNT_NAIVE) {
        for (size_t i = 0; i < S2; ++i) {
          const double nxy = (i / S == i % S) ? 0.0 : (model->naive_weights ? model->naive_weights[i] : 1.0);
          Ok[i] = conditional ? nxy : P[i] * nxy;
        }
        continue;
      }
      // J = V [ (Vinv B V) o Phi ] Vinv; Phi through expm1 (accurate O(t^2) diagonal on 1e-6 branches)
      for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
          const double d = (lam[i] - lam[j]) * t;
          const double phi = std::fabs(d) < 1e-14 ? t * std::exp(lam[i] * t) : t * std::exp(lam[j] * t) * std::expm1(d) / d;
          Phi[(size_t)i * S + j] = W[k][(size_t)i * S + j] * phi;
        }
      Mat J = matmul(S, matmul(S, V, Phi), Vi);
      for (size_t i = 0; i < S2; ++i) {
        double nxy = J[i] / P[i];  // conditional count; Bio++ guards: non-finite -> 0, unweighted negatives -> 0
        if (std::isnan(nxy) || std::isinf(nxy)) nxy = 0.0;
        if (model->clamp_negative && nxy < 0.0) nxy = 0.0;
        Ok[i] = conditional ? nxy : P[i] * nxy;
      }
    }
  };
  for (int c = 0; c < C; ++c)
    for (int b = 0; b < B; ++b)
      branch_mats(b, hm->blen[b] * hm->rates[c], &hm->P[((size_t)c * B + b) * S2], &hm->PN[((size_t)c * B + b) * K * S2], false);
  // N^k(x, y; t_b) at the branch length itself: computeSubstitutionVectorsNoAveraging reads single entries of it; and at
  // r_c t_b per class: computeSubstitutionVectorsMarginal weights it with the two marginal posteriors
  hm->N1.assign((size_t)B * K * S2, 0.0);
  hm->NC.assign((size_t)C * B * K * S2, 0.0);
  {
    Mat P1(S2);
    for (int b = 0; b < B; ++b) branch_mats(b, hm->blen[b], P1.data(), &hm->N1[(size_t)b * K * S2], true);
    for (int c = 0; c < C; ++c)
      for (int b = 0; b < B; ++b) branch_mats(b, hm->blen[b] * hm->rates[c], P1.data(), &hm->NC[((size_t)c * B + b) * K * S2], true);
  }
  // ---- device layouts
  // One allocation, per class a block of MC matrices of S*S doubles (the unit the kernel DMAs into LDS):
  //   [0, NI)                    P of internal edges, 4x4-block packed (matrix-vector products)
  //   [NI, NI + NI*K)            P o N^k of internal edges, packed, index slot*K + k
  //   [.., + T)                  P of leaf edges transposed, [z][x] = P[x][z] (per-lane row gather by observed symbol)
  //   [.., + K*T)                P o N^k of leaf edges transposed, index k*T + taxon
  const int NI = hm->NI;
  //   [.., + ncherry*(1+3K))     cherry tables of the class-fused nucleotide layout (cmx_walk.h), 16 rows (symbol pair) each
  const int MC = hm->plain ? 0 : NI + NI * K + T + K * T + hm->ncherry * (1 + 3 * K);
  hm->MC = MC;
  hm->fuse = (S == 4 && C >= 4) ? (C == 4 ? 4 : 5) : 1;
  const int F = hm->fuse;
  const int dS = S * F, dC = (C + F - 1) / F;
  hm->dS = dS;
  hm->dC = dC;
  const size_t MU = hm->plain ? 0 : (size_t)mat_unit(dS);   // doubles per device matrix: dS*dS plus max_ambig(dS) extra leaf rows
  const size_t dS2 = (size_t)dS * dS;
  hm->MAT.assign((size_t)dC * MC * MU, 0.0);
  hm->CP.assign((size_t)C * nn * S2 + 4, 0.0);   // + 4: the fused simulator reads four running sums at a time
  for (int c = 0; c < C; ++c)
    for (int b = 0; b < B; ++b) {
      const double* P = &hm->P[((size_t)c * B + b) * S2];
      double* cp = &hm->CP[((size_t)c * nn + b) * S2];
      for (int x = 0; x < S; ++x) {
        double cum = 0.0;
        for (int y = 0; y < S; ++y) { cum += P[(size_t)x * S + y]; cp[(size_t)x * S + y] = cum; }
      }
    }
  Mat dense(dS2);
  for (int dc = 0; dc < (hm->plain ? 0 : dC); ++dc) {
    double* blk = &hm->MAT[(size_t)dc * MC * MU];
    for (int b = 0; b < B; ++b) {
      // operator of edge b for device class dc: diagonal blocks g = 0..F-1 <-> true class dc*F + g (classes beyond C
      // pad the last group: identity transition, zero weight).  which = -1: P, k >= 0: w_c (P o N^k), w_c = class
      // probability when fused (the kernel then uses weight 1 per pass), else 1.
      auto block = [&](int g, int which) -> const double* {
        const int c = dc * F + g;
        if (c >= C) return nullptr;
        return which < 0 ? &hm->P[((size_t)c * B + b) * S2] : &hm->PN[(((size_t)c * B + b) * K + which) * S2];
      };
      auto weight = [&](int g, int which) { return (which >= 0 && F > 1) ? hm->probs[dc * F + g] : 1.0; };
      if (hm->taxon_of[b] >= 0) {
        const int tx = hm->taxon_of[b];
        for (int which = -1; which < K; ++which) {
          double* lt = blk + (size_t)(which < 0 ? NI + NI * K + tx : NI + NI * K + T + which * T + tx) * MU;
          for (int g = 0; g < F; ++g) {
            const double* M = block(g, which);
            for (int x = 0; x < S; ++x)
              for (int z = 0; z < S; ++z) {  // row = observed state z, column = device state X = (class g, state x): transposed.
                // Columns are stored state-in-tile major (X % 4) * (dS / 4) + X / 4: the values one lane of the
                // matrix-core layout needs from a row are contiguous.
                const int X = g * S + x, pos = (X % 4) * (dS / 4) + X / 4;
                lt[(size_t)z * leaf_row_stride(dS) + pos] = M ? weight(g, which) * M[(size_t)x * S + z] : (which < 0 && x == z ? 1.0 : 0.0);
              }
          }
        }
      } else {
        const int sl = hm->slot[b];
        for (int which = -1; which < K; ++which) {
          std::fill(dense.begin(), dense.end(), 0.0);
          for (int g = 0; g < F; ++g) {
            const double* M = block(g, which);
            for (int x = 0; x < S; ++x)
              for (int y = 0; y < S; ++y)
                dense[(size_t)(g * S + x) * dS + g * S + y] =
                    M ? weight(g, which) * M[(size_t)x * S + y] : (which < 0 && x == y ? 1.0 : 0.0);
          }
          pack_blocks(dS, dense.data(), blk + (size_t)(which < 0 ? sl : NI + sl * K + which) * MU, F > 1);
        }
      }
    }
    // cherry tables: row 4 s1 + s2 (symbols of the cherry's leaves l1, l2), column X = (class g, state x) stored like a
    // leaf row; class weights are folded into the count operators as everywhere in the fused layout
    for (int n = 0; n < nn && hm->ncherry > 0; ++n) {
      if (hm->cherry_of[n] < 0) continue;
      int l1 = -1, l2 = -1;
      for (int e = hm->first_child[n]; e >= 0; e = hm->next_sib[e]) { if (l1 < 0) l1 = e; else l2 = e; }
      double* tab = blk + (size_t)(hm->cherry_base + hm->cherry_of[n] * (1 + 3 * K)) * MU;
      for (int g = 0; g < F; ++g) {
        const int c = dc * F + g;
        for (int s1 = 0; s1 < S; ++s1)
          for (int s2 = 0; s2 < S; ++s2)
            for (int x = 0; x < S; ++x) {
              const int X = g * S + x, pos = (X % 4) * (dS / 4) + X / 4;
              const size_t at = (size_t)(4 * s1 + s2) * leaf_row_stride(dS) + pos;
              if (c >= C) {   // padding class: identity transitions, zero weight -- message = [x == s1 == s2], no counts
                tab[at] = (x == s1 && x == s2) ? 1.0 : 0.0;
                for (int q = 1; q <= 3 * K; ++q) tab[(size_t)q * MU + at] = 0.0;
                continue;
              }
              const double w = hm->probs[c];
              const double* Pn = &hm->P[((size_t)c * B + n) * S2];
              const double* P1 = &hm->P[((size_t)c * B + l1) * S2];
              const double* P2 = &hm->P[((size_t)c * B + l2) * S2];
              double m = 0.0;
              for (int y = 0; y < S; ++y) m += Pn[(size_t)x * S + y] * P1[(size_t)y * S + s1] * P2[(size_t)y * S + s2];
              tab[at] = m;
              for (int k = 0; k < K; ++k) {
                const double* Jn = &hm->PN[(((size_t)c * B + n) * K + k) * S2];
                const double* J1 = &hm->PN[(((size_t)c * B + l1) * K + k) * S2];
                const double* J2 = &hm->PN[(((size_t)c * B + l2) * K + k) * S2];
                double tj = 0.0, t1 = 0.0, t2 = 0.0;
                for (int y = 0; y < S; ++y) {
                  tj += Jn[(size_t)x * S + y] * P1[(size_t)y * S + s1] * P2[(size_t)y * S + s2];
                  t1 += Pn[(size_t)x * S + y] * J1[(size_t)y * S + s1] * P2[(size_t)y * S + s2];
                  t2 += Pn[(size_t)x * S + y] * P1[(size_t)y * S + s1] * J2[(size_t)y * S + s2];
                }
                tab[(size_t)(1 + k) * MU + at] = w * tj;
                tab[(size_t)(1 + K + k) * MU + at] = w * t1;
                tab[(size_t)(1 + 2 * K + k) * MU + at] = w * t2;
              }
            }
      }
    }
  }
  // guide table of the simulator's inverse-CDF search: entry k of a row = the number of leading running sums that are
  // <= k/32, i.e. where the linear scan "index = #{j < S-1 : u >= cum[j]}" may start for any u in [k/32, (k+1)/32)
  hm->CPG.assign((size_t)C * nn * S * 32, 0);
  for (size_t r = 0; r < (size_t)C * nn * S; ++r) {
    const double* cum = &hm->CP[r * S];
    for (int k = 0; k < 32; ++k) {
      int st = 0;
      while (st < S - 1 && cum[st] <= k / 32.0) ++st;
      hm->CPG[r * 32 + k] = (uint8_t)st;
    }
  }
  if (!hm->plain) {
    const std::string bad = verify_walk(*hm);
    if (!bad.empty()) return bad;
  }
  hm->cum_pi.resize(S);
  hm->cum_probs.resize(C);
  double cum = 0.0;
  for (int x = 0; x < S; ++x) { cum += hm->pi[x]; hm->cum_pi[x] = cum; }
  cum = 0.0;
  for (int c = 0; c < C; ++c) { cum += hm->probs[c]; hm->cum_probs[c] = cum; }
  *code = CMX_OK;
  return std::string();
}

}  // namespace cmx
